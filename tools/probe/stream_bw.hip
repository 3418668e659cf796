// Does a bandwidth-bound kernel run at the same speed on every HIP stream?  (The streaming transforms are 10 % slower in
// some library contexts for their whole life; a context = a stream.)  A strided row walk like theirs: 4096 workgroups of
// one wavefront, each walks 256 rows of 1 KiB pieces 7680 bytes apart, reads a buffer and writes another.
//   hipcc --offload-arch=gfx950 -O3 -o stream_bw stream_bw.hip && ./stream_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void walk(const uint4 *in, uint4 *out, int rows, size_t row_q, size_t wg_q) {
  const size_t base = (size_t)blockIdx.x * wg_q + threadIdx.x;
  for (int r = 0; r < rows; ++r) {
    uint4 v = in[base + (size_t)r * row_q];
    v.x ^= v.y;
    out[base + (size_t)r * row_q] = v;
  }
}
int main() {
  const int n = 4096, rows = 256;
  const size_t row_q = 7680 / 16, wg_q = (size_t)rows * row_q; // each workgroup its own band of rows, lanes 16 bytes apart
  const size_t bytes = (size_t)n * wg_q * 16;
  uint4 *a, *b;
  hipMalloc((void **)&a, bytes); hipMalloc((void **)&b, bytes);
  hipMemset(a, 1, bytes);
  for (int s = 0; s < 10; ++s) {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, worst = 0.f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0, st);
      hipLaunchKernelGGL(walk, dim3(n), dim3(64), 0, st, a, b, rows, row_q, wg_q);
      hipEventRecord(e1, st);
      hipStreamSynchronize(st);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) { best = ms < best ? ms : best; worst = ms > worst ? ms : worst; }
    }
    printf("stream %d: %.3f - %.3f ms  (%.2f TB/s read + write)\n", s, best, worst, 2.0 * bytes / best / 1e9);
  }
  return 0;
}
