extern "C" const char* ivf_build_id(void) { return "1c184ab113c46a46"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_EXPERIMENT"; }
