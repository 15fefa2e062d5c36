// ABI bookkeeping for libclasspose_hip.
#include "cpx_common.h"
thread_local char cpx_err_buf[256] = {0};
extern "C" int cpx_abi_version(void) { return 3; }
extern "C" const char *cpx_last_error(void) { return cpx_err_buf; }
// build provenance: hash of the sources this library was compiled from (csrc/Makefile: BUILD_ID)
#ifndef CPX_BUILD_ID
#define CPX_BUILD_ID "unknown"
#endif
#ifdef CPX_DEBUG
extern "C" const char *cpx_build_id(void) { return CPX_BUILD_ID "+debug"; }
#else
extern "C" const char *cpx_build_id(void) { return CPX_BUILD_ID; }
#endif
