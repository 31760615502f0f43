"""Host-side mirror of the introspection-function call contract (ORB/Examples/Stereo/stereo_kitti.cc:231-247,
493-514): load weights once, then `cost_u8 = fcn(bgr_u8)` per frame.  All compute is in libivfront.so (HIP)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr


class IntrospectionFCN:
    def __init__(self, weights_blob, in_size, out_size=None, max_batch=1, device_id=0):
        """weights_blob: flat f32 array (iv_slam_amd.fcn_weights.pack_blob / tools/export_fcn_weights.py).
        in_size/out_size: (height, width); out_size defaults to in_size (the KITTI/Jackal mains' contract)."""
        self._lib = _lib.load()
        blob = np.ascontiguousarray(weights_blob, np.float32).reshape(-1)
        self.in_h, self.in_w = in_size
        self.out_h, self.out_w = out_size or in_size
        self.max_batch, self.device_id = max_batch, device_id
        h = C.c_void_p()
        check(self._lib.ivf_fcn_create(ptr(blob), blob.size, self.in_w, self.in_h, self.out_w, self.out_h, max_batch,
                                       device_id, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ivf_fcn_destroy(self._h)
            self._h = None

    def __call__(self, bgr_u8, want_f32=False):
        """bgr_u8: HxWx3 u8 host array -> cost map u8 HxW (and the f32 map when want_f32)."""
        img = np.ascontiguousarray(bgr_u8, np.uint8)
        assert img.shape == (self.in_h, self.in_w, 3)
        u8 = np.zeros((self.out_h, self.out_w), np.uint8)
        f32 = np.zeros((self.out_h, self.out_w), np.float32) if want_f32 else None
        check(self._lib.ivf_fcn_forward(self._h, ptr(img), self.in_w, self.in_h, self.in_w * 3, ptr(u8), self.out_w, ptr(f32)))
        return (u8, f32) if want_f32 else u8

    def forward_device(self, bgr, cost_u8=None, cost_f32=None, stream_ptr=None):
        """bgr: torch.uint8 [n,H,W,3] on the device (rows / images may be padded); cost_u8 [n,outH,outW] u8 and/or cost_f32 f32 tensors (contiguous)."""
        n = bgr.shape[0]
        assert tuple(bgr.shape[1:]) == (self.in_h, self.in_w, 3)
        st = bgr.stride()
        assert st[3] == 1 and st[2] == 3, "interleaved BGR pixels; rows and images may be padded (a view of a larger tensor)"
        if cost_u8 is not None and not cost_u8.is_contiguous():
            # a padded view, e.g. the front end's own cost plane (StereoFrontend.cost_plane): the maps are written through its strides
            cs = cost_u8.stride()
            assert cost_f32 is None and cs[2] == 1 and tuple(cost_u8.shape) == (n, self.out_h, self.out_w)
            check(self._lib.ivf_fcn_forward_device_strided(self._h, bgr.data_ptr(), st[0], st[1], n, cost_u8.data_ptr(), cs[0], cs[1], stream_ptr))
            return
        check(self._lib.ivf_fcn_forward_device(self._h, bgr.data_ptr(), st[0], st[1], n,
                                               None if cost_u8 is None else cost_u8.data_ptr(),
                                               None if cost_f32 is None else cost_f32.data_ptr(), stream_ptr))

    def status(self, stream_ptr=None):
        """waits for the stream and raises IvfError(IVF_E_STATE) if a forward of this handle since the last check drove an un-clamped
        activation out of the f16 range (|x| >= 65504: the split-f16 products of the next layer would be wrong); clears the flag."""
        check(self._lib.ivf_fcn_status(self._h, stream_ptr))

    # --- measurement aid (bench.py): HIP events around the 960 -> 160 fused depthwise+projection launch
    def probe_enable(self):
        check(self._lib.ivf_fcn_probe_enable(self._h))

    def probe_select(self, which):
        """0 = block 15 (k_fcn_irbd4<true>, runs twice per forward), 1 = block 17 (k_fcn_irbd4h): what probe_stats / probe_info report"""
        check(self._lib.ivf_fcn_probe_select(self._h, int(which)))

    def probe_stats(self, last_n=0):
        """(summed ms, launches, batch size) of the last `last_n` probed forwards (0 = all kept)."""
        s = C.c_double(0); n = C.c_int(0); b = C.c_int(0)
        check(self._lib.ivf_fcn_probe_stats(self._h, last_n, C.byref(s), C.byref(n), C.byref(b)))
        return s.value, n.value, b.value

    def probe_info(self):
        """(kernel name as dispatched, algorithmic HBM bytes per image) of the launch the probe brackets."""
        name = C.create_string_buffer(128); b = C.c_double(0)
        check(self._lib.ivf_fcn_probe_info(self._h, name, 128, C.byref(b)))
        return name.value.decode(), b.value
