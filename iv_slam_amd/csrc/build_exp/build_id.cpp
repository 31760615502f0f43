extern "C" const char* ivf_build_id(void) { return "63b0e4aa6e61e1d3"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_EXPERIMENT"; }
