extern "C" const char* ivf_build_id(void) { return "0c5f02947843274a"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_EXPERIMENT"; }
