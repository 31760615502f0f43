"""Batched, device-resident tracker step (ivf_tracker_* in include/ivfront.h): for every (last, cur) pair of gather records
the matcher part of Tracking::TrackWithMotionModel (ORB/src/Tracking.cc:1303-1330) -- UpdateLastFrame's stereo points,
ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, false) (ORB/src/ORBmatcher.cc:1372-1518), the retry with the wider
window -- in one launch sequence on the device.  torch tensors are only the device-memory plumbing."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Bounds, TrackConfig, check


def record_bytes(nfeatures):
    return int(_lib.load().ivf_track_record_bytes(int(nfeatures)))


class BatchTracker:
    """TrackWithMotionModel's matcher call for up to max_pairs frame pairs per launch sequence.

    th = 7 for stereo (Tracking.cc:1313-1317), th_retry = 2 * th and retry_below = 20 (Tracking.cc:1320-1330) are the
    reference's values; th_depth = mThDepth selects UpdateLastFrame's close-point rule (localization mode)."""

    def __init__(self, nfeatures, scale_factors, fx, fy, cx, cy, bf, bounds, max_pairs, th=7.0, th_retry=None, retry_below=20,
                 check_orientation=True, th_depth=0.0, points_block=True, b=None, device_id=0):
        self._lib = _lib.load()
        sf = np.ascontiguousarray(scale_factors, np.float32)
        cfg = TrackConfig()
        cfg.nfeatures = nfeatures; cfg.nlevels = len(sf)
        for i, v in enumerate(sf):
            cfg.scale_factors[i] = float(v)
        cfg.fx, cfg.fy, cfg.cx, cfg.cy, cfg.bf = fx, fy, cx, cy, bf
        cfg.b = b if b is not None else bf / fx                          # mb = mbf / fx (Frame.cc:410)
        cfg.bounds = Bounds(*bounds)
        cfg.th = th; cfg.th_retry = 2.0 * th if th_retry is None else th_retry
        cfg.retry_below = int(retry_below); cfg.check_orientation = int(bool(check_orientation))
        cfg.th_depth = th_depth; cfg.points_block = int(bool(points_block)); cfg.max_pairs = max_pairs; cfg.device_id = device_id
        h = C.c_void_p()
        check(self._lib.ivf_tracker_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.nfeatures, self.max_pairs, self.record_bytes = nfeatures, max_pairs, record_bytes(nfeatures)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ivf_tracker_destroy(self._h)
            self._h = None

    def run(self, records, pairs, assign, nmatches, poses=None, point_flags=None, point_quality=None, key_quality=None, stream_ptr=None):
        """records: torch.uint8 [n_records * record_bytes] (a packed block or the all-gathered buffer); pairs: torch.int32
        [n_pairs, 2] = (last, cur) record indices; assign: torch.int32 [n_pairs, nfeatures]; nmatches: torch.int32 [n_pairs];
        poses: torch.float32 [n_pairs, 2, 12] = per pair {LastFrame.mTcw, CurrentFrame.mTcw prior}, row-major 3x4, or None = zero
        motion; point_flags: torch.uint8 [n_records, nfeatures] or None; point_quality / key_quality: torch.float32
        [n_pairs, nfeatures] in/out (UpdateQualityScores on the device) or None.  Asynchronous on the given stream."""
        n_rec = records.numel() // self.record_bytes
        n_pairs = pairs.shape[0]
        assert pairs.dtype.is_floating_point is False and pairs.element_size() == 4 and pairs.is_contiguous()
        assert assign.element_size() == 4 and assign.numel() >= n_pairs * self.nfeatures and nmatches.numel() >= n_pairs
        assert poses is None or (poses.is_contiguous() and poses.numel() == n_pairs * 24), "poses are per PAIR: [n_pairs, 2, 12]"
        for q in (point_quality, key_quality):
            assert q is None or (q.is_contiguous() and q.element_size() == 4 and q.numel() >= n_pairs * self.nfeatures)
        check(self._lib.ivf_tracker_run(self._h, records.data_ptr(), self.record_bytes, n_rec, pairs.data_ptr(), n_pairs,
                                        None if poses is None else poses.data_ptr(),
                                        None if point_flags is None else point_flags.data_ptr(),
                                        None if point_quality is None else point_quality.data_ptr(),
                                        None if key_quality is None else key_quality.data_ptr(),
                                        assign.data_ptr(), nmatches.data_ptr(), stream_ptr))

    def search_local(self, records, frames, points, point_offsets, max_points_per_frame, assign, nmatches, poses=None, occupied=None,
                     th=1.0, nn_ratio=0.8, cos_limit=0.5, point_quality=None, key_quality=None, stream_ptr=None):
        """Tracking::SearchLocalPoints (ORB/src/Tracking.cc:2088-2132) for n_frames frames at once: Frame::isInFrustum of every local
        map point, then ORBmatcher(nn_ratio).SearchByProjection(F, vpMapPoints, th).  frames: torch.int32 [n_frames] record indices;
        points: torch.uint8 view of LOCAL_POINT_DTYPE records (80 B each); point_offsets: torch.int32 [n_frames + 1] (CSR);
        poses: torch.float32 [n_frames, 12] (the pose of frame SLOT f) or None; occupied: torch.uint8 [n_frames, nfeatures] or None;
        assign: torch.int32 [n_frames, nfeatures] (index within the frame's point range or -1); nmatches: torch.int32 [n_frames];
        point_quality: torch.float32 indexed like points, key_quality: torch.float32 [n_frames, nfeatures] (both in/out) or None.
        Asynchronous on the given stream."""
        n_rec = records.numel() // self.record_bytes
        n_frames = frames.numel()
        assert frames.element_size() == 4 and point_offsets.element_size() == 4 and point_offsets.numel() >= n_frames + 1
        assert assign.element_size() == 4 and assign.numel() >= n_frames * self.nfeatures and nmatches.numel() >= n_frames
        assert poses is None or (poses.is_contiguous() and poses.numel() == n_frames * 12), "poses are per frame SLOT: [n_frames, 12]"
        check(self._lib.ivf_tracker_search_local(self._h, records.data_ptr(), self.record_bytes, n_rec, frames.data_ptr(), n_frames,
                                                 None if poses is None else poses.data_ptr(), points.data_ptr(), point_offsets.data_ptr(),
                                                 int(max_points_per_frame), None if occupied is None else occupied.data_ptr(),
                                                 float(th), float(nn_ratio), float(cos_limit),
                                                 None if point_quality is None else point_quality.data_ptr(),
                                                 None if key_quality is None else key_quality.data_ptr(),
                                                 assign.data_ptr(), nmatches.data_ptr(), stream_ptr))
