"""ctypes binding of libivfront.so (include/ivfront.h).  There is no Python/CPU fallback: if the HIP
library is missing or no GPU is usable, calls raise."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IVFRONT_LIB") or os.path.join(_HERE, "libivfront.so")      # IVFRONT_LIB: experiment builds (tools/)

IVF_OK, IVF_E_INVALID, IVF_E_CAPACITY, IVF_E_GEOMETRY, IVF_E_NO_DEVICE, IVF_E_STATE = 0, -1, -2, -3, -4, -5
MAX_LEVELS = 16

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])


class ExtractorParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("enable_introspection", C.c_int32)]


class Bounds(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float)]


class FrontendConfig(C.Structure):
    _fields_ = [("left", ExtractorParams), ("right", ExtractorParams), ("width", C.c_int32), ("height", C.c_int32),
                ("max_pairs", C.c_int32), ("bf", C.c_float), ("b", C.c_float), ("device_id", C.c_int32)]


# ivf_local_point (include/ivfront.h): one local map point of Tracking::SearchLocalPoints, 80 bytes
LOCAL_POINT_DTYPE = np.dtype([("pos", np.float32, 3), ("normal", np.float32, 3), ("min_distance", np.float32), ("max_distance", np.float32),
                              ("desc", np.uint8, 32), ("flags", np.int32), ("pad", np.int32, 3)])
assert LOCAL_POINT_DTYPE.itemsize == 80


class TrackConfig(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("nlevels", C.c_int32), ("scale_factors", C.c_float * MAX_LEVELS),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float), ("b", C.c_float),
                ("bounds", Bounds), ("th", C.c_float), ("th_retry", C.c_float), ("retry_below", C.c_int32),
                ("check_orientation", C.c_int32), ("th_depth", C.c_float), ("points_block", C.c_int32),
                ("max_pairs", C.c_int32), ("device_id", C.c_int32)]


class IvfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ivfront error %d: %s" % (code, msg))
        self.code = code


vp = C.c_void_p
_SIGS = {
    "ivf_version": (C.c_int, []),
    "ivf_last_error": (C.c_char_p, []),
    "ivf_device_count": (C.c_int, []),
    "ivf_debug_launch_count": (C.c_longlong, []),
    "ivf_build_id": (C.c_char_p, []),
    "ivf_build_flags": (C.c_char_p, []),
    "ivf_debug_scratch_slots": (C.c_int, []),
    "ivf_extractor_create": (C.c_int, [C.POINTER(ExtractorParams), C.c_int, C.POINTER(vp)]),
    "ivf_extractor_destroy": (None, [vp]),
    "ivf_extractor_set_opencv_variant": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
    "ivf_frontend_set_opencv_variant": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
    "ivf_extractor_get_levels": (C.c_int, [vp]),
    "ivf_extractor_get_scale_factor": (C.c_float, [vp]),
    "ivf_extractor_get_scale_tables": (C.c_int, [vp, vp, vp, vp, vp]),
    "ivf_extractor_get_feature_tables": (C.c_int, [vp, vp, vp]),
    "ivf_extract": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int)]),
    "ivf_extractor_pyramid_level": (C.c_int, [vp, C.c_int, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ivf_extractor_quality_level": (C.c_int, [vp, C.c_int, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ivf_extractor_blur_level": (C.c_int, [vp, C.c_int, vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ivf_extractor_level_counts": (C.c_int, [vp, vp]),
    "ivf_stereo_match": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, C.c_float, C.c_float, vp, vp]),
    "ivf_hamming": (C.c_int, [vp, vp]),
    "ivf_hamming_pairs": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int]),
    "ivf_features_in_area": (C.c_int, [vp, C.c_int, C.POINTER(Bounds), C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                       vp, C.c_int, C.POINTER(C.c_int)]),
    "ivf_search_by_projection": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(Bounds), C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                           vp, vp, C.c_int, vp, C.POINTER(C.c_int), C.c_int]),
    "ivf_search_by_projection_ex": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(Bounds), C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                              vp, vp, C.c_int, vp, vp, C.POINTER(C.c_int), C.c_int]),
    "ivf_search_map_points": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(Bounds), C.c_int, vp, vp, vp, vp, vp, vp, vp, vp,
                                        C.c_float, vp, C.POINTER(C.c_int), C.c_int]),
    "ivf_update_quality_scores": (C.c_int, [vp, C.c_int, vp, vp, C.c_int]),
    "ivf_search_for_initialization": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, C.POINTER(Bounds), vp, C.c_int, C.c_float, C.c_int,
                                                vp, C.POINTER(C.c_int), C.c_int]),
    "ivf_distinctive_descriptor": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    "ivf_search_keyframe_points": (C.c_int, [vp, vp, C.c_int, C.POINTER(Bounds), C.c_int, vp, vp, vp, vp, vp, vp, vp,
                                             C.POINTER(C.c_int), C.c_int]),
    "ivf_search_by_sim3": (C.c_int, [vp, vp, C.c_int, C.POINTER(Bounds), vp, vp, C.c_int, C.POINTER(Bounds)] + [vp] * 12 +
                           [vp, C.POINTER(C.c_int), C.c_int]),
    "ivf_search_by_bow": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_float, C.c_int,
                                    vp, C.POINTER(C.c_int), C.c_int]),
    "ivf_search_by_bow_keyframes": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                              C.c_float, C.c_int, vp, C.POINTER(C.c_int), C.c_int]),
    "ivf_search_for_triangulation": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int,
                                               vp, C.c_float, C.c_float, vp, vp, C.c_int, C.c_int, C.c_int, vp, C.POINTER(C.c_int), C.c_int]),
    "ivf_search_by_projection_reloc": (C.c_int, [vp, vp, C.c_int, C.POINTER(Bounds), C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int,
                                                 vp, C.POINTER(C.c_int), C.c_int]),
    "ivf_vocabulary_create": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "ivf_vocabulary_destroy": (None, [vp]),
    "ivf_bow_transform": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "ivf_bow_vectors": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int), vp, vp, vp, C.c_int, C.POINTER(C.c_int)]),
    "ivf_fuse_candidates": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(Bounds), vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp,
                                      vp, vp, C.c_int]),
    "ivf_frame_create": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(Bounds), C.c_int, C.POINTER(vp)]),
    "ivf_frame_destroy": (None, [vp]),
    "ivf_frame_count": (C.c_int, [vp]),
    "ivf_frame_grid": (C.c_int, [vp, vp, vp]),
    "ivf_frame_search_by_projection": (C.c_int, [vp, C.c_int] + [vp] * 10 + [C.c_int, vp, C.POINTER(C.c_int)]),
    "ivf_frame_search_map_points": (C.c_int, [vp, C.c_int] + [vp] * 8 + [C.c_float, vp, C.POINTER(C.c_int)]),
    "ivf_frame_create_from_frontend": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(Bounds), C.POINTER(vp)]),
    "ivf_frame_search_keyframe_points": (C.c_int, [vp, C.c_int] + [vp] * 7 + [C.POINTER(C.c_int)]),
    "ivf_frame_fuse_candidates": (C.c_int, [vp, vp, C.c_int, C.c_int] + [vp] * 9),
    "ivf_frame_search_by_sim3": (C.c_int, [vp, vp] + [vp] * 12 + [vp, C.POINTER(C.c_int)]),
    "ivf_frame_search_by_projection_reloc": (C.c_int, [vp, C.c_int] + [vp] * 7 + [C.c_int, C.c_int, vp, C.POINTER(C.c_int)]),
    "ivf_init_undistort_rectify_map": (C.c_int, [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp]),
    "ivf_remap_create": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "ivf_remap_destroy": (None, [vp]),
    "ivf_remap_apply": (C.c_int, [vp, vp, C.c_int, vp, C.c_int]),
    "ivf_remap_apply_device": (C.c_int, [vp, vp, C.c_int, C.c_size_t, vp, C.c_int, C.c_size_t, C.c_int, vp]),
    "ivf_remap_get_fixed_maps": (C.c_int, [vp, vp, vp]),
    "ivf_test_retain_best": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int]),
    "ivf_frontend_create": (C.c_int, [C.POINTER(FrontendConfig), C.POINTER(vp)]),
    "ivf_frontend_destroy": (None, [vp]),
    "ivf_frontend_run": (C.c_int, [vp, vp, vp, vp, C.c_size_t, C.c_int, C.c_int, vp]),
    "ivf_frontend_cost_plane": (C.c_int, [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int), vp]),
    "ivf_fcn_forward_device_strided": (C.c_int, [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, C.c_size_t, C.c_int, vp]),
    "ivf_frontend_run_color": (C.c_int, [vp, vp, C.c_int, C.c_size_t, C.c_int, vp, C.c_int, C.c_size_t, C.c_int, vp, C.c_size_t, C.c_int, C.c_int, vp]),
    "ivf_frontend_sync": (C.c_int, [vp]),
    "ivf_frontend_device_results": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                              C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int)]),
    "ivf_frontend_fetch": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int), vp, vp, vp]),
    "ivf_frontend_fetch_of": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int), vp, vp, vp]),
    "ivf_frontend_last_fast_ms": (C.c_float, [vp]),
    "ivf_frontend_fast_ms_stats": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "ivf_frontend_pack_gather_block": (C.c_int, [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), vp]),
    "ivf_frontend_pack_gather_block_of": (C.c_int, [vp, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t), vp]),
    "ivf_frontend_batch_stream": (vp, [vp, C.c_int]),
    "ivf_track_record_bytes": (C.c_size_t, [C.c_int]),
    "ivf_tracker_create": (C.c_int, [C.POINTER(TrackConfig), C.POINTER(vp)]),
    "ivf_tracker_destroy": (None, [vp]),
    "ivf_tracker_run": (C.c_int, [vp, vp, C.c_size_t, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "ivf_tracker_search_local": (C.c_int, [vp, vp, C.c_size_t, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, vp, C.c_float, C.c_float,
                                           C.c_float, vp, vp, vp, vp, vp]),
    "ivf_fcn_create": (C.c_int, [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "ivf_fcn_destroy": (None, [vp]),
    "ivf_fcn_forward": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "ivf_fcn_forward_device": (C.c_int, [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp, vp]),
    "ivf_fcn_status": (C.c_int, [vp, vp]),
    "ivf_fcn_probe_enable": (C.c_int, [vp]),
    "ivf_fcn_probe_select": (C.c_int, [vp, C.c_int]),
    "ivf_fcn_probe_info": (C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.c_double)]),
    "ivf_fcn_probe_stats": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None


EXPERIMENT_LIB_PATH = os.path.join(_HERE, "libivfront_exp.so")     # make -C iv_slam_amd/csrc EXPERIMENT=1 (kernel-variant tests, tools/)


def source_build_id(flags=""):
    """What ivf_build_id() must return for the sources on disk (same files, same order as iv_slam_amd/csrc/Makefile IDSRCS) built
    with the variant flags `flags` ("" = the product, "-DIVF_EXPERIMENT" = the experiment build).  None when the sources are
    not beside the package (an installed / copied package: nothing to compare with)."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    inc = os.path.join(os.path.dirname(_HERE), "include")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip"))) + [os.path.join(csrc, "ivf_device.h")] + sorted(glob.glob(os.path.join(inc, "*")))
    if not os.path.isfile(os.path.join(inc, "ivfront.h")) or not os.path.isfile(os.path.join(csrc, "ivf_device.h")):
        return None
    h = hashlib.sha256()
    try:
        for f in files:
            with open(f, "rb") as fh:
                h.update(fh.read())
    except OSError:
        return None
    h.update(flags.encode())
    return h.hexdigest()[:16]


def load():
    """Load libivfront.so; raises (never falls back) when the HIP extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libivfront.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C iv_slam_amd/csrc).  There is no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own HIP runtime under the same soname (libamdhip64.so.7).
        # Two HIP runtimes in one process cannot both own the GPU, so when torch is installed it is
        # imported first and libivfront.so binds to the runtime torch already loaded; without torch the
        # system runtime in /opt/rocm/lib is used (RUNPATH of the .so).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        # provenance: a library built from other sources OR other flags than the product's must not produce a result.  The id covers
        # the variant flags, so the default library must have been built with none.  IVFRONT_LIB = an experiment build, compiled with
        # other flags on purpose: its id is checked against the sources + ITS OWN flags; a mismatch there is reported, not refused
        # (A/B runs against an older library, tools/ab_old_lib.sh).  Sources absent (installed package): nothing to check against.
        built, flags = lib.ivf_build_id().decode(), lib.ivf_build_flags().decode()
        if not os.environ.get("IVFRONT_LIB"):
            want = source_build_id("")
            if want is not None and (built != want or flags):
                raise ImportError("libivfront.so is stale or not the product build: built from sources+flags %s (flags %r), the sources on disk "
                                  "give %s -- rebuild with `make -C iv_slam_amd/csrc` (python -c 'import __graft_entry__ as g; g.build()')"
                                  % (built, flags, want))
        else:
            want = source_build_id(flags)
            if want is not None and built != want:
                import warnings
                warnings.warn("IVFRONT_LIB=%s was built from other sources than the ones on disk (%s vs %s, flags %r): results are "
                              "those of that build" % (LIB_PATH, built, want, flags))
        _lib = lib
    return _lib


def check(rc):
    if rc != IVF_OK:
        raise IvfError(rc, load().ivf_last_error().decode("utf-8", "replace"))


def ptr(a):
    return a.ctypes.data_as(vp) if a is not None else None


class _DeviceArray:
    """a raw device pointer with shape / byte strides, as the CUDA array interface torch.as_tensor understands"""
    def __init__(self, ptr, shape, strides):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "strides": tuple(strides), "typestr": "|u1", "data": (int(ptr), False), "version": 3}


def device_view_u8(ptr, shape, strides, device_id=0):
    """torch u8 view (no copy, no ownership) of device memory the library owns -- e.g. the front end's cost plane (ivf_frontend_cost_plane)"""
    import torch
    return torch.as_tensor(_DeviceArray(ptr, shape, strides), device=torch.device("cuda", device_id))
