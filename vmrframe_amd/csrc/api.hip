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

// ---- test utility: fill the whole LDS of every CU with a 32-bit pattern (e.g. a NaN).  A kernel that reads LDS it
// never wrote then produces the pattern instead of whatever the previous kernel happened to leave there: the
// determinism tests poison LDS with two different patterns around each operator and require bit-equal results.
__global__ __launch_bounds__(256) void poison_lds_kernel(uint32_t pattern, int words, uint32_t* sink) {
  extern __shared__ uint32_t lds_words[];
  for (int i = threadIdx.x; i < words; i += 256) lds_words[i] = pattern;
  __syncthreads();
  if (sink && lds_words[(threadIdx.x * 97) % words] == 0x00C0FFEEu) sink[0] = 1;   // keeps the stores alive
}

extern "C" int vmr_debug_poison_lds(uint32_t pattern, void* scratch_u32, void* stream) {
  VMR_CHECK(scratch_u32, "vmr_debug_poison_lds: needs a 4-byte device scratch word");
  const int bytes = 160 * 1024;        // one workgroup owns a CU's whole LDS
  static thread_local bool set = false;
  if (!set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(poison_lds_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return vmr_fail(-5, "vmr_debug_poison_lds: hipFuncSetAttribute: %s", hipGetErrorString(e));
    set = true;
  }
  hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), bytes, (hipStream_t)stream, pattern, bytes / 4,
                     (uint32_t*)scratch_u32);
  VMR_LAUNCH_CHECK();
  return 0;
}
