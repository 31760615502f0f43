extern "C" const char* ivf_build_id(void) { return "99f45d5eb16d20ef"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_TRACK_DEPTH=3"; }
