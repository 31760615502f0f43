extern "C" const char* ivf_build_id(void) { return "b8b1ad8d35a0223a"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_TRACK_DEPTH=4"; }
