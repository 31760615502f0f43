"""Batched, device-resident stereo front end (ivf_frontend_* in include/ivfront.h).

PyTorch is used only as the device-memory / stream / torch.distributed plumbing: the images live in
torch.uint8 CUDA(HIP) tensors whose raw pointers are handed to the C-ABI.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, ExtractorParams, FrontendConfig, check, ptr


class StereoFrontend:
    """Per pair: ORBextractor L/R (ORB/src/Frame.cc:115-125) + mvKeyQualScore (:130-143) +
    ComputeStereoMatches (:758-932), for up to max_pairs pairs per launch sequence."""

    def __init__(self, width, height, max_pairs, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7,
                 enableIntrospection=False, bf=386.1448, b=None, fx=718.856, device_id=0):
        self._lib = _lib.load()
        b = b if b is not None else bf / fx                      # mb = mbf/fx (Frame.cc:410)
        left = ExtractorParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, int(bool(enableIntrospection)))
        right = ExtractorParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, 0)   # Tracking.cc:182-183
        self.cfg = FrontendConfig(left, right, width, height, max_pairs, bf, b, device_id)
        h = C.c_void_p()
        check(self._lib.ivf_frontend_create(C.byref(self.cfg), C.byref(h)))
        self._h = h
        self.width, self.height, self.max_pairs, self.nfeatures = width, height, max_pairs, nfeatures
        self.device_id = device_id

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ivf_frontend_destroy(self._h)
            self._h = None

    def run(self, left, right, cost=None, stream_ptr=None):
        """left/right/cost: torch.uint8 tensors [n,H,W] on this device, unit column stride; rows and images may be padded
        (views of larger tensors) as long as the three share their strides -- the C-ABI takes one (image, row) stride pair.
        Asynchronous."""
        n = left.shape[0]
        assert left.dtype == right.dtype and left.shape == right.shape
        assert left.shape[1] == self.height and left.shape[2] == self.width
        st = left.stride()
        assert st[2] == 1 and right.stride() == st, "unit column stride, same strides left / right"
        cp = None
        if cost is not None:
            assert cost.shape == left.shape and cost.stride() == st
            cp = cost.data_ptr()
        check(self._lib.ivf_frontend_run(self._h, left.data_ptr(), right.data_ptr(), cp, st[0], st[1], n, stream_ptr))
        self._n = n

    COLOR_GRAY, COLOR_BGR, COLOR_RGB, COLOR_CV3 = 0, 1, 2, 4        # include/ivfront.h: IVF_COLOR_*

    def run_color(self, left, right, cost=None, stream_ptr=None, rgb=False, cv3=False):
        """Like run(), with the grey conversion of Tracking::GrabImageStereo (Tracking.cc:272-295) fused into the ingest: a side given as
        [n,H,W,3] u8 (interleaved; rgb=False: bytes B,G,R -> CV_BGR2GRAY, True: R,G,B -> CV_RGB2GRAY; cv3: OpenCV <= 3.x coefficients) is
        converted on the device, a side given as [n,H,W] is grey.  Each side and the cost maps carry their own strides."""
        n = left.shape[0]

        def side(t):
            assert t.shape[0] == n and t.shape[1] == self.height and t.shape[2] == self.width
            st = t.stride()
            if t.dim() == 4:
                assert t.shape[3] == 3 and st[3] == 1 and st[2] == 3, "interleaved 3-channel pixels"
                return t.data_ptr(), (self.COLOR_RGB if rgb else self.COLOR_BGR) | (self.COLOR_CV3 if cv3 else 0), st[0], st[1]
            assert st[2] == 1
            return t.data_ptr(), self.COLOR_GRAY, st[0], st[1]
        lp, lc, li, lr = side(left); rp, rc, ri, rr = side(right)
        cp, ci, cr = None, 0, 0
        if cost is not None:
            assert cost.dim() == 3 and cost.shape[0] == n and cost.stride()[2] == 1
            cp, ci, cr = cost.data_ptr(), cost.stride()[0], cost.stride()[1]
        check(self._lib.ivf_frontend_run_color(self._h, lp, lc, li, lr, rp, rc, ri, rr, cp, ci, cr, n, stream_ptr))
        self._n = n

    def cost_plane(self, n, stream_ptr=None):
        """The pitched level-0 cost plane of the batch context the NEXT run uses, as a torch u8 view [n, height, width] (rows / images padded): a producer on
        the device (IntrospectionFCN.forward_device(..., cost_u8=view)) writes the maps there and run_color(..., cost=view) skips their ingest.
        `stream_ptr` waits until that context's previous batch is done with the plane."""
        import torch
        p = C.c_void_p(); ist = C.c_size_t(); rst = C.c_int()
        check(self._lib.ivf_frontend_cost_plane(self._h, C.byref(p), C.byref(ist), C.byref(rst), stream_ptr))
        key = (p.value, n)
        if getattr(self, "_plane_views", None) is None:
            self._plane_views = {}
        if key not in self._plane_views:
            from ._lib import device_view_u8
            self._plane_views[key] = device_view_u8(p.value, (n, self.height, self.width), (ist.value, rst.value, 1), self.device_id)
        return self._plane_views[key]

    def set_opencv_variant(self, blur=0, retain_best=0, atan2=0):
        check(self._lib.ivf_frontend_set_opencv_variant(self._h, int(blur), int(retain_best), int(atan2)))

    def sync(self):
        check(self._lib.ivf_frontend_sync(self._h))

    def last_fast_ms(self):
        return float(self._lib.ivf_frontend_last_fast_ms(self._h))

    def fast_ms_stats(self, last_n=0):
        """(sum_ms, n) of the FAST+NMS launch over the last `last_n` runs (HIP events on the run's stream)."""
        s = C.c_double(0); n = C.c_int(0)
        check(self._lib.ivf_frontend_fast_ms_stats(self._h, last_n, C.byref(s), C.byref(n)))
        return s.value, n.value

    def fetch(self, pair, side, age=0):
        """Results of one image of the run `age` runs back (0 = last; a run stays held until two further runs were enqueued)."""
        cap = self.nfeatures
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        ur = np.zeros(cap, np.float32); dp = np.zeros(cap, np.float32); q = np.zeros(cap, np.float32)
        n = C.c_int(0)
        check(self._lib.ivf_frontend_fetch_of(self._h, int(age), pair, side, ptr(kps), ptr(desc), cap, C.byref(n), ptr(ur), ptr(dp), ptr(q)))
        n = n.value
        out = dict(kps=kps[:n].copy(), desc=desc[:n].copy(), quality=q[:n].copy())
        if side == 0:
            out.update(uright=ur[:n].copy(), depth=dp[:n].copy())
        return out

    def gather_record_bytes(self):
        rec = C.c_size_t(0)
        check(self._lib.ivf_frontend_pack_gather_block(self._h, None, 0, C.byref(rec), None))
        return rec.value

    STREAM_OF_BATCH = C.c_void_p(-1)        # pack on the internal stream the batch ran on (include/ivfront.h: IVF_STREAM_OF_BATCH)

    def batch_stream(self, age=0):
        """hipStream_t (as int) of the internal stream the batch `age` runs back was enqueued on."""
        return self._lib.ivf_frontend_batch_stream(self._h, int(age))

    def pack_gather_block(self, block, stream_ptr=None, age=0):
        """block: torch.uint8 tensor of >= n_pairs*record_bytes on this device; age = which run (0 last, 1 the one before)."""
        rec = C.c_size_t(0)
        check(self._lib.ivf_frontend_pack_gather_block_of(self._h, int(age), block.data_ptr(), block.numel(), C.byref(rec), stream_ptr))
        return rec.value


def unpack_gather_records(buf, nfeatures):
    """Decode records packed by ivf_frontend_pack_gather_block (host numpy uint8 buffer) -> list of dicts."""
    rec = 16 + nfeatures * 24 + nfeatures * 32 + nfeatures * 8
    buf = np.ascontiguousarray(buf, np.uint8).reshape(-1, rec)
    out = []
    for r in buf:
        n = int(r[:4].view(np.int32)[0])
        kps = r[16:16 + nfeatures * 24].view(KP_DTYPE)[:n].copy()
        desc = r[16 + nfeatures * 24:16 + nfeatures * 56].reshape(nfeatures, 32)[:n].copy()
        ur = r[16 + nfeatures * 56:16 + nfeatures * 60].view(np.float32)[:n].copy()
        dp = r[16 + nfeatures * 60:].view(np.float32)[:n].copy()
        out.append(dict(n=n, kps=kps, desc=desc, uright=ur, depth=dp))
    return out
