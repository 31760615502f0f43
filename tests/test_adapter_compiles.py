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
