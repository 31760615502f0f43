extern "C" const char* ivf_build_id(void) { return "9447e6b6886149d1"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_TRACK_TIMING"; }
