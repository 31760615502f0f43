extern "C" const char* ivf_build_id(void) { return "1baef5a95b08f262"; }
extern "C" const char* ivf_build_flags(void) { return "-DIVF_EXPERIMENT"; }
