// fill_probe.hip -- per-CU operand fill rate on gfx950: global -> LDS by LDS-DMA (global_load_lds_dwordx4) against
// global -> VGPR (global_load_dwordx4) [-> LDS by ds_write_b128], in the GEMM's shape: 256-thread workgroups, one
// 36-KiB "stage" per step, two-stage ring (request step s+1, wait for step s, barrier).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_probe scratch/fill_probe.hip && /tmp/fill_probe
// Patterns: "stream" = every workgroup walks its own region (HBM / Infinity Cache), "shared" = all workgroups walk the
// same 2-MiB region (L2 hits after the first pass).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int STAGE = 36 * 1024;      // bytes per step per workgroup (160 x 64 + 128 x 64 bf16)
constexpr int NI = STAGE / 1024 / 4;  // wave-instructions per wave per step (4 waves): 9

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const unsigned char* __restrict__ src, int64_t wg_stride, int64_t span, int steps,
                                                unsigned int* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned char* base = src + (int64_t)blockIdx.x * wg_stride;
  u32x4 acc = {0u, 0u, 0u, 0u};
  u32x4 r[NI];
  auto off = [&](int s, int i) { return ((int64_t)s * STAGE + (wid * NI + i) * 1024 + lane * 16) % span; };
  if (MODE == 0) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(base + off(0, i)), (lvoid_t*)(smem + (wid * NI + i) * 1024), 16, 0, 0);
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) r[i] = *reinterpret_cast<const u32x4*>(base + off(0, i));
  }
  for (int s = 0; s < steps; ++s) {
    unsigned char* cur = smem + (s & 1) * STAGE;
    unsigned char* nxt = smem + ((s + 1) & 1) * STAGE;
    if (MODE == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_global_load_lds((gvoid_t*)(base + off(s + 1, i)), (lvoid_t*)(nxt + (wid * NI + i) * 1024), 16, 0, 0);
      acc[0] ^= *reinterpret_cast<const unsigned int*>(cur + threadIdx.x * 4);   // (touch the stage)
    } else {
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < NI; ++i) *reinterpret_cast<u32x4*>(cur + (wid * NI + i) * 1024 + lane * 16) = r[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) acc ^= r[i];
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) r[i] = *reinterpret_cast<const u32x4*>(base + off(s + 1, i));
      if (MODE == 1) acc[0] ^= *reinterpret_cast<const unsigned int*>(cur + threadIdx.x * 4);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE != 0) {
#pragma unroll
    for (int i = 0; i < NI; ++i) acc ^= r[i];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

// LDS-DMA ring with DEPTH stages of SB bytes in flight (DEPTH + 1 buffers), counted waits: does the fill rate follow the bytes in flight?
template <int DEPTH, int SB>
__global__ __launch_bounds__(256, 1) void probe_ring(const unsigned char* __restrict__ src, int64_t wg_stride, int64_t span, int steps,
                                                     unsigned int* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NIR = SB / 1024 / 4;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned char* base = src + (int64_t)blockIdx.x * wg_stride;
  unsigned int acc = 0;
  auto issue = [&](int s) {
    unsigned char* dst = smem + (s % (DEPTH + 1)) * SB;
#pragma unroll
    for (int i = 0; i < NIR; ++i)
      __builtin_amdgcn_global_load_lds((gvoid_t*)(base + ((int64_t)s * SB + (wid * NIR + i) * 1024 + lane * 16) % span),
                                       (lvoid_t*)(dst + (wid * NIR + i) * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) issue(s);
  for (int s = 0; s < steps; ++s) {
    // stage s landed (this wave's part); the DEPTH - 1 younger ones stay in flight
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * NIR) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(s + DEPTH);
    acc ^= *reinterpret_cast<const unsigned int*>(smem + (s % (DEPTH + 1)) * SB + threadIdx.x * 4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345u) sink[0] = 1;
}

template <int DEPTH, int SB>
void run_ring(const unsigned char* buf, int nwg, int64_t wg_stride, int64_t span, int64_t bytes_per_wg, unsigned int* sink) {
  const int smem = (DEPTH + 1) * SB, steps = (int)(bytes_per_wg / SB);
  hipFuncSetAttribute((const void*)probe_ring<DEPTH, SB>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe_ring<DEPTH, SB>), dim3(nwg), dim3(256), smem, 0, buf, wg_stride, span, steps, sink);
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe_ring<DEPTH, SB>), dim3(nwg), dim3(256), smem, 0, buf, wg_stride, span, steps, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nwg * steps * SB * reps;
  printf("ring  %d x %3d KiB in flight (%3d KiB), %s: %8.1f us/launch  %6.2f TB/s  %6.1f GB/s per CU\n", DEPTH, SB / 1024,
         DEPTH * SB / 1024, wg_stride ? "stream" : "shared", ms * 1e3 / reps, bytes / (ms * 1e-3) / 1e12,
         bytes / (ms * 1e-3) / 1e9 / (nwg < 256 ? nwg : 256));
}

// Width of the access: the same ring with W bytes per lane per instruction (4 / 8 / 16), VGPR loads, one stage of 64 such
// instructions per wave in flight: is the per-wave ceiling a byte rate or an instruction rate?
template <int W>
__global__ __launch_bounds__(256, 2) void probe_width(const unsigned char* __restrict__ src, int64_t span, int steps,
                                                      unsigned int* __restrict__ sink) {
  typedef __attribute__((ext_vector_type(W / 4))) unsigned int vec_t;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  unsigned int acc = 0;
  constexpr int NI = 16;
  for (int s = 0; s < steps; ++s) {
    vec_t r[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
      r[i] = *reinterpret_cast<const vec_t*>(src + (((int64_t)(blockIdx.x * 4 + wid) * steps + s) * NI + i) * 64 * W % span + lane * W);
#pragma unroll
    for (int i = 0; i < NI; ++i) acc ^= r[i][0];
  }
  if (acc == 0x12345u) sink[0] = 1;
}

template <int W>
void run_width(const unsigned char* buf, int nwg, int64_t span, unsigned int* sink) {
  const int steps = 64;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe_width<W>, dim3(nwg), dim3(256), 0, 0, buf, span, steps, sink);
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(probe_width<W>, dim3(nwg), dim3(256), 0, 0, buf, span, steps, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)nwg * 4 * steps * 16 * reps;
  printf("width %2d B/lane, %4d workgroups: %7.1f us/launch  %6.1f GB/s per CU  %6.1f ns per wave-instruction\n", W, nwg, ms * 1e3 / reps,
         instr * 64 * W / (ms * 1e-3) / 1e9 / 256, ms * 1e-3 / reps / (steps * 16.0) * 1e9);
}

template <int MODE>
void run(const char* name, const unsigned char* buf, int nwg, int64_t wg_stride, int64_t span, int steps, unsigned int* sink) {
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(256), 2 * STAGE, 0, buf, wg_stride, span, steps, sink);
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(256), 2 * STAGE, 0, buf, wg_stride, span, steps, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nwg * steps * STAGE * reps;
  const int cus = nwg < 256 ? nwg : 256;
  printf("%-34s wgs %4d steps %4d: %8.1f us/launch  %6.2f TB/s  %6.1f GB/s per CU\n", name, nwg, steps, ms * 1e3 / reps,
         bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / cus);
}

int main() {
  const int64_t total = 512ll * 64 * STAGE;   // the largest grid's private regions
  unsigned char* buf; unsigned int* sink;
  hipMalloc(&buf, total + (4 << 20)); hipMalloc(&sink, 4);
  hipMemset(buf, 1, total + (4 << 20));
  for (int nwg : {64, 256, 512}) {
    const int steps = 64;
    const int64_t priv = (int64_t)steps * STAGE;              // 2.25 MiB per workgroup
    printf("--- %d workgroups\n", nwg);
    run<0>("stream  LDS-DMA", buf, nwg, priv, priv + STAGE, steps, sink);
    run<1>("stream  VGPR + ds_write", buf, nwg, priv, priv + STAGE, steps, sink);
    run<2>("stream  VGPR only", buf, nwg, priv, priv + STAGE, steps, sink);
    run<0>("shared  LDS-DMA", buf, nwg, 0, 2 << 20, steps, sink);
    run<1>("shared  VGPR + ds_write", buf, nwg, 0, 2 << 20, steps, sink);
    run<2>("shared  VGPR only", buf, nwg, 0, 2 << 20, steps, sink);
  }
  printf("--- 256 workgroups of 256 threads, ONE per CU, LDS-DMA ring\n");
  const int64_t per = 64ll * STAGE;
  for (int64_t stride : {(int64_t)0, per}) {
    const int64_t span = stride ? per + (64 << 10) : (2 << 20);
    run_ring<1, 16384>(buf, 256, stride, span, per, sink);
    run_ring<2, 16384>(buf, 256, stride, span, per, sink);
    run_ring<4, 16384>(buf, 256, stride, span, per, sink);
    run_ring<8, 16384>(buf, 256, stride, span, per, sink);
    run_ring<1, 32768>(buf, 256, stride, span, per, sink);
    run_ring<2, 32768>(buf, 256, stride, span, per, sink);
    run_ring<3, 32768>(buf, 256, stride, span, per, sink);
    run_ring<4, 32768>(buf, 256, stride, span, per, sink);
    run_ring<1, 65536>(buf, 256, stride, span, per, sink);
  }
  printf("--- waves per CU: 256-thread workgroups, 1 / 2 / 4 / 8 per CU (grid = 256 x that), one 16- or 8-KiB stage in flight each\n");
  for (int64_t stride : {(int64_t)0, (int64_t)(16 * STAGE)}) {
    const int64_t span = stride ? stride : (2 << 20);
    for (int k : {1, 2, 4}) run_ring<1, 16384>(buf, 256 * k, stride, span, 16ll * STAGE, sink);
    for (int k : {1, 2, 4, 8}) run_ring<1, 8192>(buf, 256 * k, stride, span, 16ll * STAGE, sink);
    for (int k : {4, 8}) run_ring<2, 4096>(buf, 256 * k, stride, span, 16ll * STAGE, sink);
  }
  printf("--- access width (L2-resident 2 MiB, 16 loads in flight per wave)\n");
  for (int nwg : {256, 512, 1024}) {
    run_width<4>(buf, nwg, 2 << 20, sink);
    run_width<8>(buf, nwg, 2 << 20, sink);
    run_width<16>(buf, nwg, 2 << 20, sink);
  }
  return 0;
}
