// api.hip -- version / error plumbing of the C ABI.
#include <stdarg.h>

#include "common.h"

thread_local char g_vmr_err[256] = "";

int vmr_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_vmr_err, sizeof(g_vmr_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" int vmr_version(void) { return 100; }
extern "C" const char* vmr_last_error(void) { return g_vmr_err; }
extern "C" int vmr_sizeof_gemm_desc(void) { return (int)sizeof(vmr_gemm_t); }
