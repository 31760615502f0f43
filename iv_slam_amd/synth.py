"""Seeded synthetic KITTI-shaped stereo frames (SURVEY.md §8(d) "Synthetic inputs").

Left image = gradient base + random grey rectangles + 3x3 corner dots + uniform noise, with a flat
top band (forces the `<=3 keypoints -> minThFAST` fallback of ORB/src/ORBextractor.cc:1047-1052)
and a repeated texture strip (forces FAST-response ties in retainBest).  Right image = left shifted
by a per-row-band integer disparity + independent noise, so L/R Hamming + SAD matching succeeds.
Cost map = sum of seeded Gaussian blobs scaled to u8.

numpy only; no reference code involved.  Deterministic for a given (seed, index).
"""
import numpy as np

KITTI = dict(width=1242, height=375, bf=386.1448, fx=718.856)


def _rng(seed, idx):
    return np.random.Generator(np.random.PCG64([0x5EED0000 + int(seed), int(idx)]))


def make_left(width, height, seed=0, idx=0, n_rect=400, n_dots=200, flat_rows=60):
    r = _rng(seed, idx)
    yy, xx = np.mgrid[0:height, 0:width]
    img = (40 + 120 * xx / max(width - 1, 1) + 50 * yy / max(height - 1, 1)).astype(np.float32)
    n_rect = max(4, int(n_rect * (width * height) / (1242 * 375)))
    n_dots = max(4, int(n_dots * (width * height) / (1242 * 375)))
    for _ in range(n_rect):
        w = int(r.integers(4, 81)); h = int(r.integers(4, 81))
        x = int(r.integers(0, max(1, width - 4))); y = int(r.integers(0, max(1, height - 4)))
        img[y:y + h, x:x + w] = float(r.integers(0, 256))
    # repeated texture strip: identical 16x16 tiles -> many equal FAST responses
    ty = min(height - 1, int(height * 0.55)); th = min(48, height - ty)
    tile = r.integers(60, 200, size=(16, 16)).astype(np.float32)
    tile[4:12, 4:12] = 230.0
    reps = (th + 15) // 16, (width // 2 + 15) // 16
    tex = np.tile(tile, reps)[:th, :width // 2]
    img[ty:ty + th, width // 4:width // 4 + tex.shape[1]] = tex
    for _ in range(n_dots):
        x = int(r.integers(2, max(3, width - 4))); y = int(r.integers(2, max(3, height - 4)))
        img[y:y + 3, x:x + 3] = float(r.choice([10, 245]))
    # flat band on top (after the objects so it really is flat)
    fr = min(flat_rows, height // 4)
    img[:fr, :] = 128.0
    img[:fr, :] += 3.0 * np.sin(xx[:fr, :] / 37.0)
    # weak corners (contrast 14 < iniThFAST=20, > minThFAST=7): only the minThFAST re-run finds them
    for _ in range(max(6, width // 30)):
        x = int(r.integers(20, max(21, width - 24))); y = int(r.integers(min(20, fr - 4), max(min(20, fr - 4) + 1, fr - 4)))
        img[y:y + 3, x:x + 3] += float(r.choice([-14, 14]))
    noise = r.integers(-2, 3, size=img.shape)
    return np.clip(np.rint(img) + noise, 0, 255).astype(np.uint8)


def make_right(left, seed=0, idx=0, band=48, dmin=4, dmax=64):
    r = _rng(seed + 7919, idx)
    h, w = left.shape
    right = np.empty_like(left)
    for y0 in range(0, h, band):
        d = int(r.integers(dmin, dmax + 1))
        blk = left[y0:y0 + band]
        sh = np.empty_like(blk)
        sh[:, :w - d] = blk[:, d:]          # a point at uL appears at uR = uL - d
        sh[:, w - d:] = blk[:, w - 1:w]
        right[y0:y0 + band] = sh
    noise = r.integers(-2, 3, size=right.shape)
    return np.clip(right.astype(np.int16) + noise, 0, 255).astype(np.uint8)


def make_pair(width=1242, height=375, seed=0, idx=0):
    left = make_left(width, height, seed, idx)
    return left, make_right(left, seed, idx)


def make_cost_map(width, height, seed=0, idx=0, n_blobs=6):
    r = _rng(seed + 104729, idx)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    acc = np.zeros((height, width), np.float32)
    for _ in range(n_blobs):
        cx = r.uniform(0, width); cy = r.uniform(0, height)
        s = r.uniform(0.05, 0.25) * width
        acc += r.uniform(0.3, 1.0) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    acc = acc / max(float(acc.max()), 1e-6)
    return np.clip(np.rint(acc * 255.0), 0, 255).astype(np.uint8)


def make_stream(n_pairs, width=1242, height=375, seed=0):
    """(n_pairs, 2, H, W) u8: [:,0] left, [:,1] right."""
    out = np.empty((n_pairs, 2, height, width), np.uint8)
    for i in range(n_pairs):
        out[i, 0], out[i, 1] = make_pair(width, height, seed, i)
    return out


def make_natural(width=1242, height=375, seed=0, idx=0, beta=1.0, foliage=False):
    """Natural-image statistics (r06: the generators above are rectangles, dots and noise): a random-phase field whose amplitude
    spectrum falls as 1 / f^beta (beta = 1: the 1 / f^2 POWER law of natural scenes -- corners at every scale, smooth large-scale
    shading, no flat regions), histogram-stretched to u8.  foliage=True multiplies in occlusion structure: a second, thresholded
    low-frequency field switches between two differently lit copies (sharp leaf-like boundaries at many scales, the corner
    density of vegetation).  numpy FFT in float64; deterministic for a given (seed, idx, size)."""
    r = _rng(seed + 15485863, idx)

    def field(b):
        white = r.standard_normal((height, width))
        fy = np.fft.fftfreq(height)[:, None]; fx = np.fft.rfftfreq(width)[None, :]
        f = np.sqrt(fx * fx + fy * fy); f[0, 0] = 1.0
        spec = np.fft.rfft2(white) / f ** b
        spec[0, 0] = 0.0
        out = np.fft.irfft2(spec, s=(height, width))
        return (out - out.mean()) / max(float(out.std()), 1e-12)

    img = field(beta)
    if foliage:
        mask = field(1.6) > 0.0
        other = field(beta)
        img = np.where(mask, 0.6 * img + 0.9, 0.9 * other - 0.7)
    lo, hi = np.percentile(img, [0.5, 99.5])
    return np.clip(np.rint((img - lo) / max(hi - lo, 1e-12) * 255.0), 0, 255).astype(np.uint8)
