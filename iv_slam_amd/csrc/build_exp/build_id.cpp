extern "C" const char* ivf_build_id(void) { return "75dbe86f0f2b4418"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_EXPERIMENT"; }
