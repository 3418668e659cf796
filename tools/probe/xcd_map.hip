// Which XCD does workgroup b of a launch run on, stream by stream?  (The streaming transforms' work list assumes b mod 8.)
//   hipcc --offload-arch=gfx950 -o xcd_map xcd_map.hip && ./xcd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(unsigned *out) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x & 0xF;
  // a little work so that workgroups overlap
  float a = (float)threadIdx.x;
  for (int i = 0; i < 2000; ++i) a = a * 1.0001f + 0.5f;
  if (a == 123.f) out[0] = 0;
}
int main() {
  const int n = 4096, ns = 8;
  unsigned *d;
  hipMalloc((void **)&d, n * 4);
  std::vector<unsigned> h(n);
  for (int s = 0; s < ns; ++s) {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, st, d);
      hipStreamSynchronize(st);
      hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
      int off = (int)h[0], ok = 0, cnt[8] = {0};
      for (int b = 0; b < n; ++b) { ok += ((int)h[b] == (b + off) % 8); cnt[h[b] & 7]++; }
      printf("stream %d rep %d: first 16:", s, rep);
      for (int b = 0; b < 16; ++b) printf(" %u", h[b]);
      printf(" | b mod 8 (+%d) holds for %d of %d | per XCD:", off, ok, n);
      for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
      printf("\n");
    }
  }
  return 0;
}
