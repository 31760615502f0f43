extern "C" const char* ivf_build_id(void) { return "66ee99debc325985"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_EXPERIMENT"; }
