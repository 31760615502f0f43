"""ctypes bindings for oracle/libivf_oracle.so (TEST INFRASTRUCTURE: the checker, never the product)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])
assert KP_DTYPE.itemsize == 24


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int), ("scale_factor", C.c_float), ("nlevels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int), ("enable_introspection", C.c_int)]


class Bounds(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float)]


def _build():
    if os.environ.get("IVF_ORACLE_SO"):
        # bench.py's cpu_baseline worker: the same source built -march=native for the host it is timed on
        return os.environ["IVF_ORACLE_SO"], os.path.join(ORACLE_DIR, "libstl_pin.so")
    so = os.path.join(ORACLE_DIR, "libivf_oracle.so")
    pin = os.path.join(ORACLE_DIR, "libstl_pin.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("ivf_oracle.c", "ivf_oracle.h", "stl_pin.cpp")]
    newest = max(os.path.getmtime(s) for s in srcs)
    if not (os.path.exists(so) and os.path.exists(pin)) or min(os.path.getmtime(so), os.path.getmtime(pin)) < newest:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so, pin


def source_build_id():
    """What orc_build_id() must return for the checker sources on disk (same files and order as oracle/Makefile IDSRCS)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("ivf_oracle.c", "ivf_oracle.h", "stl_pin.cpp", os.path.join("..", "include", "ivf_pattern31.inc")):
        with open(os.path.join(ORACLE_DIR, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_so, _pin = _build()
lib = C.CDLL(_so)
pin = C.CDLL(_pin)
lib.orc_build_id.restype = C.c_char_p; lib.orc_build_id.argtypes = []
BUILD_ID = lib.orc_build_id().decode()
if BUILD_ID != source_build_id():
    raise ImportError("oracle library %s is stale: built from sources %s, the sources on disk are %s -- rebuild with `make -C oracle`"
                      % (_so, BUILD_ID, source_build_id()))

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
vp = C.c_void_p

lib.orc_cv_round_f.restype = C.c_int; lib.orc_cv_round_f.argtypes = [C.c_float]
lib.orc_cv_round_d.restype = C.c_int; lib.orc_cv_round_d.argtypes = [C.c_double]
lib.orc_fast_atan2.restype = C.c_float; lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
lib.orc_cosf.restype = C.c_float; lib.orc_cosf.argtypes = [C.c_float]
lib.orc_logf.restype = C.c_float; lib.orc_logf.argtypes = [C.c_float]
lib.orc_sinf.restype = C.c_float; lib.orc_sinf.argtypes = [C.c_float]
lib.orc_fast_score_map.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
lib.orc_fast_detect.restype = C.c_int
lib.orc_fast_detect.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
lib.orc_resize_linear_8u.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int]
lib.orc_gauss7_8u.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
lib.orc_nth_element_resp.argtypes = [vp, C.c_int, C.c_int]
lib.orc_set_opencv_variant.argtypes = [C.c_int, C.c_int, C.c_int]; lib.orc_set_opencv_variant.restype = None


def set_opencv_variant(blur=0, retain=0, atan=0):
    """process-wide OpenCV-version switches of the oracle (0,0,0 = OpenCV >= 3.4.2 / 4.x)."""
    lib.orc_set_opencv_variant(int(blur), int(retain), int(atan))


lib.orc_retain_best.restype = C.c_int; lib.orc_retain_best.argtypes = [vp, C.c_int, C.c_int]
lib.orc_hamming256.restype = C.c_int; lib.orc_hamming256.argtypes = [vp, vp]
lib.orc_bit_pattern_31.restype = C.POINTER(C.c_int8)
lib.orc_extractor_create.restype = vp; lib.orc_extractor_create.argtypes = [C.POINTER(Params)]
lib.orc_extractor_destroy.argtypes = [vp]
lib.orc_extractor_levels.restype = C.c_int; lib.orc_extractor_levels.argtypes = [vp]
lib.orc_extractor_tables.argtypes = [vp, vp, vp, vp, vp, vp, vp]
lib.orc_extract.restype = C.c_int
lib.orc_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int)]
lib.orc_pyramid_level.restype = C.c_int
lib.orc_pyramid_level.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.orc_quality_level.restype = C.c_int
lib.orc_quality_level.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.orc_level_count.restype = C.c_int; lib.orc_level_count.argtypes = [vp, C.c_int]
lib.orc_stereo_match.restype = C.c_int
lib.orc_stereo_match.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, C.c_float, C.c_float, vp, vp]
lib.orc_search_by_projection.restype = C.c_int
lib.orc_search_by_projection.argtypes = [vp, vp, vp, C.c_int, C.POINTER(Bounds), C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                         vp, vp, C.c_int, vp, C.POINTER(C.c_int)]
lib.orc_features_in_area.restype = C.c_int
lib.orc_features_in_area.argtypes = [vp, C.c_int, C.POINTER(Bounds), C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                     vp, C.c_int]
lib.orc_three_maxima.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.orc_search_map_points.restype = C.c_int
lib.orc_search_map_points.argtypes = [vp, vp, vp, C.c_int, C.POINTER(Bounds), C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                      C.c_float, vp, C.POINTER(C.c_int)]
lib.orc_update_quality_scores.argtypes = [vp, C.c_int, vp, vp]
lib.orc_search_for_initialization.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.POINTER(Bounds), vp, C.c_int, C.c_float, C.c_int,
                                              vp, C.POINTER(C.c_int)]
lib.orc_distinctive_descriptor.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.orc_search_keyframe_points.argtypes = [vp, vp, C.c_int, C.POINTER(Bounds), C.c_int, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int)]
lib.orc_search_by_sim3.argtypes = [vp, vp, C.c_int, C.POINTER(Bounds), vp, vp, C.c_int, C.POINTER(Bounds)] + [vp] * 12 + [vp, C.POINTER(C.c_int)]
lib.orc_search_by_bow.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_float, C.c_int,
                                  vp, C.POINTER(C.c_int)]
lib.orc_search_by_bow_keyframes.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                            C.c_float, C.c_int, vp, C.POINTER(C.c_int)]
lib.orc_search_for_triangulation.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                             vp, C.c_float, C.c_float, vp, vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int)]
lib.orc_search_by_projection_reloc.argtypes = [vp, vp, C.c_int, C.POINTER(Bounds), C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int,
                                               vp, C.POINTER(C.c_int)]
lib.orc_bow_transform.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp]
lib.orc_fuse_candidates.argtypes = [vp, vp, vp, C.c_int, C.POINTER(Bounds), vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
pin.stl_retain_best.restype = C.c_int; pin.stl_retain_best.argtypes = [vp, C.c_int, C.c_int]
pin.stl_nth_element.argtypes = [vp, C.c_int, C.c_int]


def ptr(a):
    return a.ctypes.data_as(vp) if a is not None else None


def fast_score_map(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib.orc_fast_score_map(ptr(img), img.shape[1], img.shape[1], img.shape[0], threshold, ptr(out))
    return out


def fast_detect(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros(cap, KP_DTYPE)
    n = lib.orc_fast_detect(ptr(img), img.shape[1], img.shape[1], img.shape[0], threshold, ptr(out), cap)
    return out[:n]


def resize_linear(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros((dh, dw), np.uint8)
    lib.orc_resize_linear_8u(ptr(img), img.shape[1], img.shape[1], img.shape[0], ptr(out), dw, dw, dh)
    return out


def gauss7(img):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib.orc_gauss7_8u(ptr(img), img.shape[1], img.shape[1], img.shape[0], ptr(out), img.shape[1])
    return out


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib.orc_hamming256(ptr(a), ptr(b))


def pattern31():
    return np.ctypeslib.as_array(lib.orc_bit_pattern_31(), shape=(1024,)).copy()


class Extractor:
    """Oracle counterpart of ORB_SLAM2::ORBextractor (ORB/include/ORBextractor.h:51-126)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, introspection=False):
        self.params = Params(nfeatures, scale_factor, nlevels, ini_th, min_th, int(bool(introspection)))
        self.h = lib.orc_extractor_create(C.byref(self.params))
        if not self.h:
            raise ValueError("bad extractor params")
        self.nlevels = nlevels
        self.nfeatures = nfeatures

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_extractor_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        sc = np.zeros(n, np.float32); inv = np.zeros(n, np.float32); s2 = np.zeros(n, np.float32)
        is2 = np.zeros(n, np.float32); nf = np.zeros(n, np.int32); um = np.zeros(16, np.int32)
        lib.orc_extractor_tables(self.h, ptr(sc), ptr(inv), ptr(s2), ptr(is2), ptr(nf), ptr(um))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, features_per_level=nf, umax=um)

    def __call__(self, img, cost=None, cap=None):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        cap = cap or max(2 * self.nfeatures, 64)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        if cost is not None:
            cost = np.ascontiguousarray(cost, np.uint8)
            assert cost.shape == img.shape
        rc = lib.orc_extract(self.h, ptr(img), w, h, w, ptr(cost), w, ptr(kps), ptr(desc), cap, C.byref(n))
        if rc != 0:
            raise RuntimeError("orc_extract rc=%d" % rc)
        return kps[:n.value].copy(), desc[:n.value].copy()

    def _level(self, fn, level):
        d = vp(); w = C.c_int(); h = C.c_int()
        if fn(self.h, level, C.byref(d), C.byref(w), C.byref(h)) != 0:
            return None
        buf = (C.c_uint8 * (w.value * h.value)).from_address(d.value)
        return np.frombuffer(buf, np.uint8).reshape(h.value, w.value).copy()

    def pyramid(self, level):
        return self._level(lib.orc_pyramid_level, level)

    def quality_pyramid(self, level):
        return self._level(lib.orc_quality_level, level)

    def level_counts(self):
        return [lib.orc_level_count(self.h, l) for l in range(self.nlevels)]


def stereo_match(eL, eR, kpL, descL, kpR, descR, bf, b):
    kpL = np.ascontiguousarray(kpL); kpR = np.ascontiguousarray(kpR)
    descL = np.ascontiguousarray(descL); descR = np.ascontiguousarray(descR)
    ur = np.zeros(len(kpL), np.float32); dp = np.zeros(len(kpL), np.float32)
    rc = lib.orc_stereo_match(eL.h, eR.h, ptr(kpL), len(kpL), ptr(descL), ptr(kpR), len(kpR), ptr(descR),
                              bf, b, ptr(ur), ptr(dp))
    if rc != 0:
        raise RuntimeError("orc_stereo_match rc=%d" % rc)
    return ur, dp


def features_in_area(kps, bounds, x, y, r, min_level, max_level):
    kps = np.ascontiguousarray(kps)
    out = np.zeros(max(len(kps), 1), np.int32)
    bd = Bounds(*bounds)
    n = lib.orc_features_in_area(ptr(kps), len(kps), C.byref(bd), x, y, r, min_level, max_level, ptr(out), len(out))
    return out[:n].copy()


def search_by_projection(cur_kps, cur_desc, cur_uright, bounds, q, check_orientation=True, cur_assign=None):
    """q: dict with u,v,ur,radius,min_level,max_level,angle,desc,valid,blocks (numpy arrays)."""
    cur_kps = np.ascontiguousarray(cur_kps); cur_desc = np.ascontiguousarray(cur_desc, np.uint8)
    cur_uright = np.ascontiguousarray(cur_uright, np.float32)
    n_cur = len(cur_kps); n_q = len(q["u"])
    assign = np.full(n_cur, -1, np.int32) if cur_assign is None else np.ascontiguousarray(cur_assign, np.int32).copy()
    arrs = dict(u=np.float32, v=np.float32, ur=np.float32, radius=np.float32, min_level=np.int32, max_level=np.int32,
                angle=np.float32, desc=np.uint8, valid=np.uint8, blocks=np.uint8)
    qq = {k: np.ascontiguousarray(q[k], t) for k, t in arrs.items()}
    bd = Bounds(*bounds)
    nm = C.c_int(0)
    lib.orc_search_by_projection(ptr(cur_kps), ptr(cur_desc), ptr(cur_uright), n_cur, C.byref(bd), n_q,
                                 ptr(qq["u"]), ptr(qq["v"]), ptr(qq["ur"]), ptr(qq["radius"]),
                                 ptr(qq["min_level"]), ptr(qq["max_level"]), ptr(qq["angle"]), ptr(qq["desc"]),
                                 ptr(qq["valid"]), ptr(qq["blocks"]), int(check_orientation), ptr(assign), C.byref(nm))
    return assign, nm.value


def search_map_points(cur_kps, cur_desc, cur_uright, bounds, q, nn_ratio, cur_assign=None):
    """q: dict with u,v,ur,radius,level,desc,valid,blocks."""
    cur_kps = np.ascontiguousarray(cur_kps); cur_desc = np.ascontiguousarray(cur_desc, np.uint8)
    cur_uright = np.ascontiguousarray(cur_uright, np.float32)
    n_cur = len(cur_kps); n_q = len(q["u"])
    assign = np.full(n_cur, -1, np.int32) if cur_assign is None else np.ascontiguousarray(cur_assign, np.int32).copy()
    arrs = dict(u=np.float32, v=np.float32, ur=np.float32, radius=np.float32, level=np.int32, desc=np.uint8,
                valid=np.uint8, blocks=np.uint8)
    qq = {k: np.ascontiguousarray(q[k], t) for k, t in arrs.items()}
    bd = Bounds(*bounds)
    nm = C.c_int(0)
    lib.orc_search_map_points(ptr(cur_kps), ptr(cur_desc), ptr(cur_uright), n_cur, C.byref(bd), n_q, ptr(qq["u"]), ptr(qq["v"]),
                              ptr(qq["ur"]), ptr(qq["radius"]), ptr(qq["level"]), ptr(qq["desc"]), ptr(qq["valid"]),
                              ptr(qq["blocks"]), nn_ratio, ptr(assign), C.byref(nm))
    return assign, nm.value


def search_for_initialization(kps1, desc1, kps2, desc2, bounds2, prev_matched, window_size, nn_ratio=0.9, check_orientation=True):
    """ORBmatcher::SearchForInitialization on flat arrays -> (vnMatches12, vbPrevMatched updated, nmatches)."""
    k1 = np.ascontiguousarray(kps1); k2 = np.ascontiguousarray(kps2)
    d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
    prev = np.ascontiguousarray(prev_matched, np.float32).reshape(-1, 2).copy()
    m12 = np.full(len(k1), -1, np.int32); nm = C.c_int(0); bd = Bounds(*bounds2)
    lib.orc_search_for_initialization(ptr(k1), ptr(d1), len(k1), ptr(k2), ptr(d2), len(k2), C.byref(bd), ptr(prev), int(window_size),
                                      nn_ratio, int(check_orientation), ptr(m12), C.byref(nm))
    return m12, prev, nm.value


def distinctive_descriptor(desc):
    """MapPoint::ComputeDistinctiveDescriptors core -> (best index, its median distance)."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    bi = C.c_int(0); bm = C.c_int(0)
    rc = lib.orc_distinctive_descriptor(ptr(d), len(d), C.byref(bi), C.byref(bm))
    assert rc == 0
    return bi.value, bm.value


def search_keyframe_points(kf_kps, kf_desc, bounds, q, matched=None):
    """q: dict with u, v, radius, level, desc, valid -> (vpMatched as query indices, nmatches)."""
    k = np.ascontiguousarray(kf_kps); d = np.ascontiguousarray(kf_desc, np.uint8)
    m = np.full(len(k), -1, np.int32) if matched is None else np.ascontiguousarray(matched, np.int32).copy()
    t = dict(u=np.float32, v=np.float32, radius=np.float32, level=np.int32, desc=np.uint8, valid=np.uint8)
    qq = {a: np.ascontiguousarray(q[a], b) for a, b in t.items()}
    nm = C.c_int(0); bd = Bounds(*bounds)
    lib.orc_search_keyframe_points(ptr(k), ptr(d), len(k), C.byref(bd), len(qq["u"]), ptr(qq["u"]), ptr(qq["v"]), ptr(qq["radius"]),
                                   ptr(qq["level"]), ptr(qq["desc"]), ptr(qq["valid"]), ptr(m), C.byref(nm))
    return m, nm.value


def fuse_candidates(kf_kps, kf_desc, kf_uright, bounds, inv_level_sigma2, q):
    """q: dict with u, v, ur, radius, level, desc, valid -> (best_idx, best_dist) per query."""
    k = np.ascontiguousarray(kf_kps); d = np.ascontiguousarray(kf_desc, np.uint8)
    gate = inv_level_sigma2 is not None
    ur = np.ascontiguousarray(kf_uright, np.float32) if gate else None
    sg = np.ascontiguousarray(inv_level_sigma2, np.float32) if gate else None
    t = dict(u=np.float32, v=np.float32, radius=np.float32, level=np.int32, desc=np.uint8, valid=np.uint8)
    if gate: t["ur"] = np.float32
    qq = {a: np.ascontiguousarray(q[a], b) for a, b in t.items()}
    qq.setdefault("ur", None)
    n = len(qq["u"]); bi = np.full(n, -1, np.int32); bdist = np.full(n, 256, np.int32); bd = Bounds(*bounds)
    lib.orc_fuse_candidates(ptr(k), ptr(d), ptr(ur), len(k), C.byref(bd), ptr(sg), n, ptr(qq["u"]), ptr(qq["v"]), ptr(qq["ur"]),
                            ptr(qq["radius"]), ptr(qq["level"]), ptr(qq["desc"]), ptr(qq["valid"]), ptr(bi), ptr(bdist))
    return bi, bdist


def _sim3_q(q):
    t = dict(u=np.float32, v=np.float32, radius=np.float32, level=np.int32, desc=np.uint8, valid=np.uint8)
    return {a: np.ascontiguousarray(q[a], b) for a, b in t.items()}


def search_by_sim3(k1, d1, b1, k2, d2, b2, q12, q21):
    """q12 / q21: dicts with u, v, radius, level, desc, valid per keypoint slot of KF1 / KF2 -> (matches12, nfound)."""
    k1 = np.ascontiguousarray(k1); k2 = np.ascontiguousarray(k2)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    a = _sim3_q(q12); b = _sim3_q(q21)
    m = np.full(len(k1), -1, np.int32); nf = C.c_int(0); bd1 = Bounds(*b1); bd2 = Bounds(*b2)
    lib.orc_search_by_sim3(ptr(k1), ptr(d1), len(k1), C.byref(bd1), ptr(k2), ptr(d2), len(k2), C.byref(bd2),
                           ptr(a["u"]), ptr(a["v"]), ptr(a["radius"]), ptr(a["level"]), ptr(a["desc"]), ptr(a["valid"]),
                           ptr(b["u"]), ptr(b["v"]), ptr(b["radius"]), ptr(b["level"]), ptr(b["desc"]), ptr(b["valid"]),
                           ptr(m), C.byref(nf))
    return m, nf.value


def feature_vector_csr(fv):
    """{node id: [feature indices]} (a DBoW2::FeatureVector) -> (node ids ascending, starts, indices) int32 arrays."""
    nodes = sorted(fv)
    start = np.zeros(len(nodes) + 1, np.int32); idx = []
    for k, nd in enumerate(nodes):
        idx.extend(fv[nd]); start[k + 1] = len(idx)
    return np.array(nodes, np.int32), start, np.array(idx, np.int32)


def search_by_bow(kf_kps, kf_desc, kf_has_mp, kf_fv, f_kps, f_desc, f_fv, nn_ratio=0.7, check_orientation=True):
    kk = np.ascontiguousarray(kf_kps); kd = np.ascontiguousarray(kf_desc, np.uint8); hm = np.ascontiguousarray(kf_has_mp, np.uint8)
    fk = np.ascontiguousarray(f_kps); fd = np.ascontiguousarray(f_desc, np.uint8)
    kn, ks, ki = feature_vector_csr(kf_fv); fn, fs, fi = feature_vector_csr(f_fv)
    m = np.full(len(fk), -1, np.int32); nm = C.c_int(0)
    lib.orc_search_by_bow(ptr(kk), ptr(kd), ptr(hm), len(kk), ptr(kn), ptr(ks), ptr(ki), len(kn), ptr(fk), ptr(fd), len(fk),
                          ptr(fn), ptr(fs), ptr(fi), len(fn), nn_ratio, int(check_orientation), ptr(m), C.byref(nm))
    return m, nm.value


def search_by_bow_keyframes(k1, d1, has1, fv1, k2, d2, has2, fv2, nn_ratio=0.75, check_orientation=True):
    k1 = np.ascontiguousarray(k1); k2 = np.ascontiguousarray(k2)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    h1 = np.ascontiguousarray(has1, np.uint8); h2 = np.ascontiguousarray(has2, np.uint8)
    n1_, s1_, i1_ = feature_vector_csr(fv1); n2_, s2_, i2_ = feature_vector_csr(fv2)
    m = np.full(len(k1), -1, np.int32); nm = C.c_int(0)
    lib.orc_search_by_bow_keyframes(ptr(k1), ptr(d1), ptr(h1), len(k1), ptr(n1_), ptr(s1_), ptr(i1_), len(n1_),
                                    ptr(k2), ptr(d2), ptr(h2), len(k2), ptr(n2_), ptr(s2_), ptr(i2_), len(n2_),
                                    nn_ratio, int(check_orientation), ptr(m), C.byref(nm))
    return m, nm.value


def search_for_triangulation(k1, d1, has1, st1, fv1, k2, d2, has2, st2, fv2, F12, ex, ey, scale2, sigma2_2, only_stereo, check_orientation):
    k1 = np.ascontiguousarray(k1); k2 = np.ascontiguousarray(k2)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    h1 = np.ascontiguousarray(has1, np.uint8); h2 = np.ascontiguousarray(has2, np.uint8)
    s1 = np.ascontiguousarray(st1, np.uint8); s2 = np.ascontiguousarray(st2, np.uint8)
    F = np.ascontiguousarray(F12, np.float32).reshape(9); sc = np.ascontiguousarray(scale2, np.float32); sg = np.ascontiguousarray(sigma2_2, np.float32)
    a, b, c = feature_vector_csr(fv1); e, f, g = feature_vector_csr(fv2)
    m = np.full(len(k1), -1, np.int32); nm = C.c_int(0)
    lib.orc_search_for_triangulation(ptr(k1), ptr(d1), ptr(h1), ptr(s1), len(k1), ptr(a), ptr(b), ptr(c), len(a),
                                     ptr(k2), ptr(d2), ptr(h2), ptr(s2), len(k2), ptr(e), ptr(f), ptr(g), len(e),
                                     ptr(F), ex, ey, ptr(sc), ptr(sg), int(only_stereo), int(check_orientation), ptr(m), C.byref(nm))
    return m, nm.value


def search_by_projection_reloc(cur_kps, cur_desc, bounds, q, orb_dist, check_orientation, cur_assign=None):
    """q: dict with u, v, radius, level, angle, desc, valid -> (mvpMapPoints as query indices, nmatches)."""
    k = np.ascontiguousarray(cur_kps); d = np.ascontiguousarray(cur_desc, np.uint8)
    a = np.full(len(k), -1, np.int32) if cur_assign is None else np.ascontiguousarray(cur_assign, np.int32).copy()
    t = dict(u=np.float32, v=np.float32, radius=np.float32, level=np.int32, angle=np.float32, desc=np.uint8, valid=np.uint8)
    qq = {x: np.ascontiguousarray(q[x], y) for x, y in t.items()}
    nm = C.c_int(0); bd = Bounds(*bounds)
    lib.orc_search_by_projection_reloc(ptr(k), ptr(d), len(k), C.byref(bd), len(qq["u"]), ptr(qq["u"]), ptr(qq["v"]), ptr(qq["radius"]),
                                       ptr(qq["level"]), ptr(qq["angle"]), ptr(qq["desc"]), ptr(qq["valid"]), int(orb_dist),
                                       int(check_orientation), ptr(a), C.byref(nm))
    return a, nm.value


def make_vocabulary(k, depth, seed, early_leaf_frac=0.0, stop_frac=0.0):
    """Synthetic DBoW2-shaped vocabulary tree as flat arrays: dict(child_start, child, desc, word, weight, depth)."""
    rng = np.random.default_rng(seed)
    children = [[]]; level = [0]; frontier = [0]
    for lv in range(1, depth + 1):
        nxt = []
        for node in frontier:
            if lv > 1 and rng.uniform() < early_leaf_frac: continue            # a leaf above the last level
            kk = k if rng.uniform() > 0.3 else int(rng.integers(1, k + 1))     # ragged branching
            for _ in range(kk):
                children.append([]); level.append(lv); children[node].append(len(children) - 1); nxt.append(len(children) - 1)
        frontier = nxt
    n = len(children)
    start = np.zeros(n + 1, np.int32); flat = []
    for i in range(n):
        flat.extend(children[i]); start[i + 1] = len(flat)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    word = np.full(n, -1, np.int32); weight = np.zeros(n, np.float64); w = 0
    for i in range(n):
        if not children[i]:
            word[i] = w; w += 1
            weight[i] = 0.0 if rng.uniform() < stop_frac else float(rng.uniform(0.1, 9.0))
    return dict(child_start=start, child=np.array(flat, np.int32), desc=desc, word=word, weight=weight, depth=depth, n_words=w)


def bow_transform(voc, desc, levelsup=4):
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); n = len(d)
    wid = np.zeros(n, np.int32); nid = np.zeros(n, np.int32); wt = np.zeros(n, np.float64)
    rc = lib.orc_bow_transform(len(voc["word"]), ptr(voc["child_start"]), ptr(voc["child"]), ptr(voc["desc"]), ptr(voc["word"]),
                               ptr(voc["weight"]), voc["depth"], ptr(d), n, levelsup, ptr(wid), ptr(nid), ptr(wt))
    assert rc == 0
    return wid, nid, wt


def init_undistort_rectify_map(K, D, R, P, size):
    w, h = int(size[0]), int(size[1])
    K = np.ascontiguousarray(np.asarray(K, np.float64).reshape(3, 3)); P = np.ascontiguousarray(np.asarray(P, np.float64).reshape(3, -1)[:, :3])
    Rm = None if R is None else np.ascontiguousarray(np.asarray(R, np.float64).reshape(3, 3))
    d = np.zeros(0, np.float64) if D is None else np.ascontiguousarray(np.asarray(D, np.float64).ravel())
    m1 = np.empty((h, w), np.float32); m2 = np.empty((h, w), np.float32)
    lib.orc_init_undistort_rectify_map.restype = C.c_int
    rc = lib.orc_init_undistort_rectify_map(ptr(K), ptr(d) if len(d) else None, len(d), ptr(Rm) if Rm is not None else None, ptr(P),
                                            w, h, ptr(m1), ptr(m2))
    assert rc == 0
    return m1, m2


def remap_bilinear(img, map1, map2):
    img = np.ascontiguousarray(img, np.uint8); cn = 1 if img.ndim == 2 else img.shape[2]
    m1 = np.ascontiguousarray(map1, np.float32); m2 = np.ascontiguousarray(map2, np.float32)
    out = np.empty(m1.shape + ((cn,) if img.ndim == 3 else ()), np.uint8)
    lib.orc_remap_bilinear_u8.restype = None
    lib.orc_remap_bilinear_u8(ptr(img), img.shape[1], img.shape[0], img.strides[0], cn, ptr(m1), ptr(m2), m1.shape[1], m1.shape[0],
                              ptr(out), out.strides[0])
    return out


def remap_weight_table():
    t = np.zeros(4096, np.int16)
    lib.orc_remap_weight_table.restype = None
    lib.orc_remap_weight_table(ptr(t))
    return t.reshape(1024, 4)


def gray_from_color(img, rgb, cv3=False):
    """cvtColor(img, CV_RGB2GRAY if rgb else CV_BGR2GRAY) on an [H, W, 3] u8 image (Tracking.cc:272-295; A-13)"""
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty(img.shape[:2], np.uint8)
    lib.orc_gray_from_color.restype = None
    lib.orc_gray_from_color(ptr(img), img.shape[1], img.shape[0], img.strides[0], ptr(out), out.strides[0], int(bool(rgb)), int(bool(cv3)))
    return out


def c_round(v):
    """std::round of a float array (Frame.cc:130-131 rounds the keypoint position with it): halves AWAY from zero --
    np.rint rounds them to even, which differs on x.5 coordinates (levels > 0 produce them)."""
    v = np.asarray(v, dtype=np.float64)
    return (np.sign(v) * np.floor(np.abs(v) + 0.5)).astype(np.int64)
