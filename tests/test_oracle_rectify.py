"""CPU tests of the rectification oracle (cv::initUndistortRectifyMap + cv::remap INTER_LINEAR restated in
oracle/ivf_oracle.c): known answers and an independent numpy restatement."""
import numpy as np

import oracle_lib as O

# stereo calibration of the reference's Jackal configuration (jackal_visual_odom_stereo_inference.yaml: LEFT.K/R/P and the
# commented-out distortion vector); values are data
K_L = [527.873518, 0.0, 482.823413, 0.0, 527.276819, 298.033945, 0.0, 0.0, 1.0]
D_L = [-0.153137, 0.075666, -0.000227, -0.000320, 0.0]
R_L = [0.999940, -0.003244, -0.010471, 0.003318, 0.999970, 0.007064, 0.010448, -0.007098, 0.999920]
P_L = [528.955512, 0.0, 479.748173, 0.0, 0.0, 528.955512, 298.607571, 0.0, 0.0, 0.0, 1.0, 0.0]


def np_remap(img, m1, m2):
    """independent restatement in numpy integer arithmetic (1 channel)"""
    sh, sw = img.shape
    fx = np.rint(m1.astype(np.float32) * np.float32(32)).astype(np.int64)
    fy = np.rint(m2.astype(np.float32) * np.float32(32)).astype(np.int64)
    sx = np.clip(fx >> 5, -32768, 32767); sy = np.clip(fy >> 5, -32768, 32767)
    ax = fx & 31; ay = fy & 31
    w = np.stack([(32 - ay) * (32 - ax) * 32, (32 - ay) * ax * 32, ay * (32 - ax) * 32, ay * ax * 32], -1)
    zero = (ax == 0) & (ay == 0)
    w[zero] = [32767, 0, 0, 1]
    acc = np.zeros(m1.shape, np.int64)
    for t, (dx, dy) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)]):
        xx = sx + dx; yy = sy + dy
        ok = (xx >= 0) & (xx < sw) & (yy >= 0) & (yy < sh)
        tap = np.where(ok, img[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)].astype(np.int64), 0)
        acc += tap * w[..., t]
    return ((acc + (1 << 14)) >> 15).astype(np.uint8)


def test_weight_table():
    T = O.remap_weight_table().astype(np.int64)
    assert (T.sum(1) == 32768).all()
    assert T[0].tolist() == [32767, 0, 0, 1]                      # 2^15 saturates to short; the repair lands on the last entry
    a = np.arange(1, 1024); fx = a & 31; fy = a >> 5
    closed = np.stack([(32 - fy) * (32 - fx), (32 - fy) * fx, fy * (32 - fx), fy * fx], -1) * 32
    assert (T[1:] == closed).all()


def test_identity_shift_and_border():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(37, 53)).astype(np.uint8)
    yy, xx = np.mgrid[0:37, 0:53].astype(np.float32)
    assert np.array_equal(O.remap_bilinear(img, xx, yy), img)
    half = O.remap_bilinear(img, xx + 0.5, yy)
    a = img.astype(int); b = np.concatenate([a[:, 1:], np.zeros((37, 1), int)], 1)      # the tap right of the last column is the border (0)
    assert np.array_equal(half, ((a + b + 1) >> 1).astype(np.uint8))
    far = O.remap_bilinear(img, xx + 1000, yy - 500)
    assert not far.any()
    # a position exactly between two 1/32 steps rounds to the even one (cvRound)
    m = xx + np.float32(1 / 64)
    out = O.remap_bilinear(img, m, yy)
    assert np.array_equal(out, np_remap(img, m, yy))


def test_remap_against_numpy_restatement():
    rng = np.random.default_rng(2)
    for (sh, sw, h, w) in [(64, 80, 50, 71), (120, 160, 120, 160)]:
        img = rng.integers(0, 256, size=(sh, sw)).astype(np.uint8)
        m1 = (rng.random((h, w)) * (sw + 8) - 4).astype(np.float32)
        m2 = (rng.random((h, w)) * (sh + 8) - 4).astype(np.float32)
        m1[0, :5] = [-1.0, -0.5, sw - 1, sw - 0.5, 1e6]; m2[0, :5] = [0, 0, 0, 0, -1e6]
        assert np.array_equal(O.remap_bilinear(img, m1, m2), np_remap(img, m1, m2))
    # 3 channels = 3 independent planes
    img3 = rng.integers(0, 256, size=(40, 50, 3)).astype(np.uint8)
    m1 = (rng.random((33, 47)) * 52 - 1).astype(np.float32); m2 = (rng.random((33, 47)) * 42 - 1).astype(np.float32)
    out = O.remap_bilinear(img3, m1, m2)
    for c in range(3):
        assert np.array_equal(out[..., c], np_remap(np.ascontiguousarray(img3[..., c]), m1, m2))


def np_rectify_map(K, D, R, P, w, h):
    """direct (non-incremental) evaluation in double"""
    K = np.asarray(K).reshape(3, 3); P = np.asarray(P).reshape(3, -1)[:, :3]; R = np.eye(3) if R is None else np.asarray(R).reshape(3, 3)
    iR = np.linalg.inv(P @ R)
    d = np.zeros(12); d[:len(D)] = D
    j, i = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    X = iR[0, 0] * j + iR[0, 1] * i + iR[0, 2]; Y = iR[1, 0] * j + iR[1, 1] * i + iR[1, 2]; W = iR[2, 0] * j + iR[2, 1] * i + iR[2, 2]
    x = X / W; y = Y / W; r2 = x * x + y * y
    kr = (1 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2) / (1 + ((d[7] * r2 + d[6]) * r2 + d[5]) * r2)
    xd = x * kr + d[2] * 2 * x * y + d[3] * (r2 + 2 * x * x) + d[8] * r2 + d[9] * r2 * r2
    yd = y * kr + d[2] * (r2 + 2 * y * y) + d[3] * 2 * x * y + d[10] * r2 + d[11] * r2 * r2
    return K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]


def test_init_undistort_rectify_map():
    # no distortion, no rotation, P = K: the identity map (up to float rounding of the accumulated row walk)
    m1, m2 = O.init_undistort_rectify_map(K_L, None, None, K_L, (960, 600))
    yy, xx = np.mgrid[0:600, 0:960]
    assert np.abs(m1 - xx).max() < 1e-3 and np.abs(m2 - yy).max() < 1e-3
    # the Jackal left camera with its distortion vector and rectifying rotation
    for D in (D_L, D_L[:4], D_L + [0.01, -0.02, 0.003], D_L + [0.01, -0.02, 0.003, 1e-4, -2e-4, 3e-4, -1e-4]):
        m1, m2 = O.init_undistort_rectify_map(K_L, D, R_L, P_L, (960, 600))
        e1, e2 = np_rectify_map(K_L, D, R_L, P_L, 960, 600)
        assert m1.dtype == np.float32 and m1.shape == (600, 960)
        assert np.abs(m1 - e1).max() < 2e-3 and np.abs(m2 - e2).max() < 2e-3
    # the principal ray maps to the source principal point
    m1, m2 = O.init_undistort_rectify_map(K_L, D_L, None, K_L, (960, 600))
    assert abs(m1[298, 483] - 483) < 0.05 and abs(m2[298, 483] - 298) < 0.05
