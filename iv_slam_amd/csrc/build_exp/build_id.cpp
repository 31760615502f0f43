extern "C" const char* ivf_build_id(void) { return "dcc9b46fb9ef3b39"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_EXPERIMENT"; }
