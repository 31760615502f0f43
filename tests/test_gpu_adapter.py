"""The C++ adapter (include/ivfront_orbslam.hpp) COMPILED AND RUN against mock Frame / KeyFrame / MapPoint types that carry
the reference's member names: ALL of ORBmatcher's reference signatures (4 x SearchByProjection, 2 x SearchByBoW,
SearchForInitialization, SearchForTriangulation, SearchBySim3, 2 x Fuse, UpdateQualityScores) end to end -- projection loop in
the adapter, window search on the GPU, bookkeeping on the mocks -- against oracle/projection_oracle.py + the C oracle."""
import subprocess

import numpy as np
import pytest

import adapter_scenario as AS
import oracle_lib as O
from iv_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    import iv_slam_amd
    assert iv_slam_amd.load().ivf_device_count() >= 1
    return AS.build_driver(tmp_path_factory.mktemp("adapter") / "adapter_driver")


@pytest.mark.parametrize("seed,forward", [(5, True), (6, False)])
def test_orbmatcher_reference_signatures_end_to_end(driver, tmp_path, seed, forward):
    S, blob = AS.make(O, synth, seed, forward)
    (tmp_path / "s.bin").write_bytes(blob)
    r = subprocess.run([driver, str(tmp_path / "s.bin"), str(tmp_path / "r.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(tmp_path / "r.bin", np.int32).astype(np.int64)
    want, counts = AS.expected(O, S)
    assert got.shape == want.shape
    assert np.array_equal(got, want), "first difference at %d" % int(np.nonzero(got != want)[0][0])
    # the scenario exercises every call for real
    assert counts["cur_last"] > 50 and counts["local"] > 5 and counts["reloc"] > 20 and counts["kf_sim3"] > 20 and counts["fused"] > 20, counts
    assert counts["bow"] > 50 and counts["bow_kf"] > 20 and counts["init"] > 50 and counts["triangulation"] > 50, counts
    assert counts["sim3"] > 5 and counts["fuse_sim3"] > 10, counts


# ---- ORB_SLAM2::ORBextractor + ivf::ComputeStereoMatches of the adapter, executed ---------------------------------------------
class _Blob:
    def __init__(self, b):
        self.b, self.o = b, 0

    def take(self, dtype, n=1):
        a = np.frombuffer(self.b, dtype, n, self.o); self.o += a.nbytes
        return a

    def i32(self):
        return int(self.take(np.int32)[0])

    def kps(self):
        from iv_slam_amd._lib import KP_DTYPE
        return self.take(KP_DTYPE, self.i32()).copy()

    def mat(self):
        r, c = self.i32(), self.i32()
        return self.take(np.uint8, r * c).reshape(r, c).copy()

    def arr(self, dtype):
        n = self.i32()
        return self.take(dtype, n // np.dtype(dtype).itemsize).copy()


@pytest.mark.parametrize("size,n,ini", [((1242, 375), 1000, 20), ((640, 240), 500, 12)])
def test_orbextractor_adapter_executed_two_threads_masks_pyramids_stereo(tmp_path, size, n, ini):
    """ORBextractor.h:57-92 as the adapter re-creates it, RUN against the mock cv types: operator() with and without a mask on
    two threads (Frame.cc:116-124), mvImagePyramid / mvQualityImagePyramid, mbCopyPyramids = false, the getters, the empty image,
    then ivf::ComputeStereoMatches -- byte for byte against the oracle."""
    import os
    import struct
    import iv_slam_amd
    assert iv_slam_amd.load().ivf_device_count() >= 1
    root = AS.ROOT
    exe = str(tmp_path / "extractor_driver")
    lib_dir = os.path.join(root, "iv_slam_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-ffp-contract=off", "-Wall", "-pthread", "-I", os.path.join(root, "include"),
                           "-I", os.path.join(root, "tests", "cv_mock"), os.path.join(root, "tests", "adapter", "extractor_driver.cpp"),
                           "-o", exe, "-L", lib_dir, "-livfront", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    w, h = size
    bf, b = 386.1448, 386.1448 / 718.856
    L, R = synth.make_pair(w, h, seed=61, idx=0)
    cost = synth.make_cost_map(w, h, seed=61, idx=0)
    (tmp_path / "s.bin").write_bytes(struct.pack("<6i3f", w, h, n, ini, 7, 8, 1.2, bf, b) + L.tobytes() + R.tobytes() + cost.tobytes())
    r = subprocess.run([exe, str(tmp_path / "s.bin"), str(tmp_path / "r.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    B = _Blob((tmp_path / "r.bin").read_bytes())
    oL = O.Extractor(n, 1.2, 8, ini, 7, introspection=True); oR = O.Extractor(n, 1.2, 8, ini, 7)
    okL, odL = oL(L, cost); okR, odR = oR(R, None)
    kL, dL, kR, dR = B.kps(), B.mat(), B.kps(), B.mat()
    assert kL.tobytes() == okL.tobytes() and np.array_equal(dL, odL), "left (with mask)"
    assert kR.tobytes() == okR.tobytes() and np.array_equal(dR, odR), "right (mask ignored: introspection off)"
    assert len(kL) > n // 2
    nl = B.i32()
    assert nl == 8
    for l in range(nl):
        assert np.array_equal(B.mat(), oL.pyramid(l)), "mvImagePyramid[%d] left" % l
        assert np.array_equal(B.mat(), oL.quality_pyramid(l)), "mvQualityImagePyramid[%d]" % l
        assert np.array_equal(B.mat(), oR.pyramid(l)), "mvImagePyramid[%d] right" % l
    t = oL.tables()
    for key in ("scale", "inv_scale", "sigma2", "inv_sigma2"):
        assert B.arr(np.float32).tobytes() == t[key].astype(np.float32).tobytes(), key
    assert B.take(np.float32)[0] == np.float32(1.2)
    our, odp = O.stereo_match(oL, oR, okL, odL, okR, odR, bf, b)
    assert B.arr(np.float32).tobytes() == our.tobytes() and B.arr(np.float32).tobytes() == odp.tobytes(), "ivf::ComputeStereoMatches"
    assert (our >= 0).sum() > 50
    ok2, od2 = O.Extractor(n, 1.2, 8, ini, 7, introspection=True)(L, None)
    k2, d2 = B.kps(), B.mat()
    assert k2.tobytes() == ok2.tobytes() and np.array_equal(d2, od2), "same handle, no mask"
    assert k2.tobytes() != kL.tobytes()                                   # the cost map really changed the selection
    o3 = O.Extractor(n, 1.2, 8, ini, 7, introspection=True); ok3, _ = o3(R, None)
    assert B.kps().tobytes() == ok3.tobytes()
    assert B.i32() == 1, "mbCopyPyramids = false must leave the public pyramid members alone"
    assert np.array_equal(B.mat(), o3.pyramid(2)), "CopyPyramidLevel"
    assert (B.i32(), B.i32()) == (3, 2), "empty image: silent return, outputs untouched"
    assert B.o == len(B.b)
