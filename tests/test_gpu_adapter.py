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
