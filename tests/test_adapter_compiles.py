"""include/ivfront_orbslam.hpp (the C++ adapter that re-creates ORB_SLAM2::ORBextractor / ORBmatcher on the C-ABI) must
compile and LINK against libivfront.so with every ORBmatcher signature instantiated on mock Frame / KeyFrame / MapPoint
types (tests/adapter/adapter_driver.cpp).  OpenCV is absent here, so a test-only stand-in provides cv::Mat / cv::KeyPoint.
Running it needs a GPU: tests/test_gpu_adapter.py."""
import os
import subprocess

import adapter_scenario as AS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adapter_driver_builds_and_links(tmp_path):
    exe = AS.build_driver(tmp_path / "adapter_driver")
    assert os.path.getsize(exe) > 10000
    # all 11 search / fuse signatures of ORB/include/ORBmatcher.h:44-89 are in the binary (mangled member names)
    syms = subprocess.check_output(["nm", "-C", exe], text=True)
    for name in ("SearchByProjection", "SearchByBoW", "SearchForInitialization", "SearchForTriangulation", "SearchBySim3", "Fuse"):       # (UpdateQualityScores is inlined)
        assert "ORBmatcherT<ORB_SLAM2::Frame, ORB_SLAM2::KeyFrame, ORB_SLAM2::MapPoint>::" + name in syms, name


def test_extractor_adapter_compiles(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "ivfront_orbslam.hpp"\n'
                   'int use(ORB_SLAM2::ORBextractor* e, cv::Mat& im, std::vector<cv::KeyPoint>& k, cv::Mat& d) {\n'
                   '  (*e)(im, cv::Mat(), k, d); e->mbCopyPyramids = false;\n'
                   '  std::vector<float> a, b; ivf::ComputeStereoMatches(e, e, k, d, k, d, 386.f, 0.54f, a, b);\n'
                   '  return e->GetLevels() + (int)e->mvQualityImagePyramid.size(); }\n')
    subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "tests", "cv_mock"), str(src)])


def test_cpp_exchange_step_compiles_against_rccl_and_matches_integration_md(tmp_path):
    """r06: the C++ form of the multi-GPU exchange step (pack -> ncclAllGather on the batch's stream into [G * P records | carry] -> carry
    hand-over -> ivf_tracker_run) compiles against include/ivfront.h, the HIP runtime headers and <rccl/rccl.h> with plain g++ (no GPU, no
    hipcc), and INTEGRATION.md section 6 shows exactly the code between its markers.  tests/test_gpu_track.py RUNS it at world = 1."""
    import pytest
    src = os.path.join(ROOT, "tests", "adapter", "exchange_rccl.cpp")
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no RCCL headers here")
    subprocess.check_call(["g++", "-std=c++14", "-c", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           src, "-o", str(tmp_path / "x.o")])
    syms = subprocess.check_output(["nm", "-C", str(tmp_path / "x.o")], text=True)
    for name in ("ncclAllGather", "ivf_frontend_pack_gather_block", "ivf_frontend_batch_stream", "ivf_tracker_run", "hipMemcpyAsync", "hipStreamWaitEvent"):
        assert name in syms, name
    text = open(src).read()
    snippet = text.split("// [integration-snippet-begin]\n")[1].split("// [integration-snippet-end]")[0]
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert snippet.strip() in md, "INTEGRATION.md section 6 does not show tests/adapter/exchange_rccl.cpp's snippet verbatim"
