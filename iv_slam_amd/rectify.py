"""Stereo rectification in front of the extractor: cv::initUndistortRectifyMap + cv::remap(INTER_LINEAR) as the
reference's driver uses them (introspective_ORB_SLAM/Examples/Stereo/stereo_kitti.cc:285-343, :462-468, :519-521).
The map is built on the host once; the per-frame remap is a HIP kernel behind libivfront's C-ABI."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr


def initUndistortRectifyMap(K, D, R, P, size):
    """cv::initUndistortRectifyMap(K, D, R, P, size, CV_32F) -> (map1, map2), both [height][width] float32.
    `size` = (width, height) like cv::Size; D may be None/empty (or 4, 5, 8, 12 coefficients); R None = identity;
    P: 3x3 or 3x4 (the driver passes P.rowRange(0,3).colRange(0,3), :289)."""
    lib = _lib.load()
    w, h = int(size[0]), int(size[1])
    K = np.ascontiguousarray(np.asarray(K, np.float64).reshape(3, 3))
    P = np.ascontiguousarray(np.asarray(P, np.float64).reshape(3, -1)[:, :3])
    Rm = None if R is None else np.ascontiguousarray(np.asarray(R, np.float64).reshape(3, 3))
    d = np.zeros(0, np.float64) if D is None else np.ascontiguousarray(np.asarray(D, np.float64).ravel())
    m1 = np.empty((h, w), np.float32); m2 = np.empty((h, w), np.float32)
    check(lib.ivf_init_undistort_rectify_map(ptr(K), ptr(d) if len(d) else None, len(d), ptr(Rm) if Rm is not None else None,
                                             ptr(P), w, h, ptr(m1), ptr(m2)))
    return m1, m2


class Remap:
    """cv::remap(src, dst, map1, map2, cv::INTER_LINEAR) with the maps resident on the device.  One object per
    (map, source size, channel count); call it with an [H][W] or [H][W][3] uint8 image."""

    def __init__(self, map1, map2, src_shape, channels=1, device_id=0):
        self._lib = _lib.load()
        m1 = np.ascontiguousarray(map1, np.float32); m2 = np.ascontiguousarray(map2, np.float32)
        if m1.ndim != 2 or m1.shape != m2.shape:
            raise AssertionError("map1 / map2 must be [height][width] float32 of the same shape")
        self.shape = m1.shape; self.src_shape = (int(src_shape[0]), int(src_shape[1])); self.channels = int(channels)
        self.device_id = device_id
        h = C.c_void_p()
        check(self._lib.ivf_remap_create(ptr(m1), ptr(m2), m1.shape[1], m1.shape[0], self.src_shape[1], self.src_shape[0],
                                         self.channels, device_id, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ivf_remap_destroy(self._h)
            self._h = None

    def __call__(self, image):
        img = np.ascontiguousarray(image, np.uint8)
        want = self.src_shape + ((3,) if self.channels == 3 else ())
        if img.shape != want:
            raise AssertionError("image shape %s, remap built for %s" % (img.shape, want))
        out = np.empty(self.shape + ((3,) if self.channels == 3 else ()), np.uint8)
        check(self._lib.ivf_remap_apply(self._h, ptr(img), img.strides[0], ptr(out), out.strides[0]))
        return out

    def apply_device(self, src, out=None):
        """torch uint8 CUDA tensors, [N][H][W] (1 channel) or [N][H][W][3]; runs on torch's current stream."""
        import torch
        if not (src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()):
            raise AssertionError("contiguous uint8 CUDA tensor expected")
        n = src.shape[0]
        tail = (3,) if self.channels == 3 else ()
        if tuple(src.shape[1:]) != self.src_shape + tail:
            raise AssertionError("image shape %s, remap built for %s" % (tuple(src.shape[1:]), self.src_shape + tail))
        if out is None:
            out = torch.empty((n,) + self.shape + tail, dtype=torch.uint8, device=src.device)
        srow = self.src_shape[1] * self.channels; drow = self.shape[1] * self.channels
        st = torch.cuda.current_stream(src.device).cuda_stream
        check(self._lib.ivf_remap_apply_device(self._h, C.c_void_p(src.data_ptr()), srow, srow * self.src_shape[0],
                                               C.c_void_p(out.data_ptr()), drow, drow * self.shape[0], n, C.c_void_p(st)))
        return out

    def fixed_maps(self):
        """(xy int16 [H][W][2], alpha uint16 [H][W]) -- what cv::convertMaps(map1, map2, CV_16SC2) would give."""
        xy = np.empty(self.shape + (2,), np.int16); al = np.empty(self.shape, np.uint16)
        check(self._lib.ivf_remap_get_fixed_maps(self._h, ptr(xy), ptr(al)))
        return xy, al
