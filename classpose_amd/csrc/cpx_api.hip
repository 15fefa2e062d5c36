// ABI bookkeeping for libclasspose_hip.
#include "cpx_common.h"
thread_local char cpx_err_buf[256] = {0};
extern "C" int cpx_abi_version(void) { return 2; }
extern "C" const char *cpx_last_error(void) { return cpx_err_buf; }
