extern "C" const char* ivf_build_id(void) { return "92b066b4a97e8d76"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_EXPERIMENT"; }
