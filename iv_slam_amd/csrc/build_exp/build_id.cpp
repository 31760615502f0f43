extern "C" const char* ivf_build_id(void) { return "07f7424bb452bb00"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_EXPERIMENT"; }
