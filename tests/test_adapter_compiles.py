"""include/ivfront_orbslam.hpp (the C++ adapter that re-creates ORB_SLAM2::ORBextractor / ORBmatcher on the
C-ABI) must at least compile; OpenCV is absent here, so a test-only type stand-in is used for the syntax check."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_adapter_header_compiles(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "ivfront_orbslam.hpp"\n'
                   'int use(ORB_SLAM2::ORBextractor* e, cv::Mat& im, std::vector<cv::KeyPoint>& k, cv::Mat& d) {\n'
                   '  (*e)(im, cv::Mat(), k, d); ORB_SLAM2::ORBmatcher m(0.9f, true);\n'
                   '  std::vector<float> a, b; ivf::ComputeStereoMatches(e, e, k, d, k, d, 386.f, 0.54f, a, b);\n'
                   '  return e->GetLevels() + ORB_SLAM2::ORBmatcher::DescriptorDistance(d, d); }\n')
    subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "tests", "cv_mock"), str(src)])
