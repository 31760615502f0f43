extern "C" const char* ivf_build_id(void) { return "ed0bbb0cb8f3fb9e"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_EXPERIMENT"; }
