// What one SIMD of gfx950 issues per cycle: wave64 integer / float VALU instructions from 1..8 resident wavefronts.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip && ./valu_rate
// Every wavefront runs ITER x 8 independent dependent-chains (so that one wavefront alone is bound by issue, not by the
// latency of its own previous instruction); the kernel reports its own cycle count (s_memtime) for the instruction count.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096;
template <int KIND> __global__ __launch_bounds__(64) void k(unsigned *out, unsigned long long *cyc, unsigned seed) {
  unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
  const unsigned long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < ITER; ++i) {
    if (KIND == 0) { // v_add_u32
      asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                   "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed));
    } else if (KIND == 1) { // v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"((float)seed));
    } else if (KIND == 2) { // v_lshlrev_b64
      unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3;
      asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3\n"
                   "v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3\n"
                   : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(seed & 1));
      a0 += (unsigned)b0; a1 += (unsigned)b1; a2 += (unsigned)b2; a3 += (unsigned)b3;
    } else if (KIND == 3) { // v_cvt_f32_i32 / v_cvt_u32_f32
      asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_u32_f32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_u32_f32 %1, %1\n"
                   "v_cvt_f32_i32 %2, %2\n v_cvt_u32_f32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_cvt_u32_f32 %3, %3\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    } else { // v_mul_lo_u32
      asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                   "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(seed | 1));
    }
  }
  const unsigned long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND> int run(const char *name) {
  hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  for (int w : {1, 2, 3, 4, 6, 8}) {
    const int blocks = cus * 4 * w;
    unsigned *out; unsigned long long *cyc;
    CHK(hipMalloc(&out, (size_t)blocks * 64 * 4)); CHK(hipMalloc(&cyc, (size_t)blocks * 8));
    // LDS keeps w wavefronts per SIMD resident at most: 160 KiB / (4 w) per 64-thread workgroup
    const size_t lds = (size_t)(160 * 1024) / (4 * w) / 2 * 2 - 1024;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), lds > 65536 ? 65536 : lds, 0, out, cyc, 1u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), lds > 65536 ? 65536 : lds, 0, out, cyc, 1u);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CHK(hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double instr = (double)ITER * 8;
    printf("%-14s %d wavefront(s) per SIMD: %.2f shader-clock cycles per instruction and wavefront -> %.2f cycles per instruction per SIMD  (kernel %.3f ms)\n",
           name, w, mean / instr, mean / instr / w, ms);
    CHK(hipFree(out)); CHK(hipFree(cyc));
  }
  return 0;
}
int main() {
  if (run<0>("v_add_u32")) return 1;
  if (run<1>("v_fma_f32")) return 1;
  if (run<2>("v_lshlrev_b64")) return 1;
  if (run<3>("v_cvt f32<->int")) return 1;
  if (run<4>("v_mul_lo_u32")) return 1;
  return 0;
}
