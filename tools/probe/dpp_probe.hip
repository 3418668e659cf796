// Prints what v_mov_b32_dpp wave_shr:1 / wave_shl:1 deliver on this GPU (lane i <- lane i-1 / i+1; lanes without a
// source keep `old`).  The streaming DWT kernels rely on exactly this.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *o) {
  int v = threadIdx.x * 10;
  int l = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false); // wave_shr:1
  int r = __builtin_amdgcn_update_dpp(-2, v, 0x130, 0xf, 0xf, false); // wave_shl:1
  o[threadIdx.x] = l;
  o[64 + threadIdx.x] = r;
}
int main() {
  int *d, h[128];
  hipMalloc(&d, sizeof h);
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("wave_shr:1 :"); for (int i = 0; i < 64; ++i) printf(" %d", h[i]); printf("\n");
  printf("wave_shl:1 :"); for (int i = 0; i < 64; ++i) printf(" %d", h[64 + i]); printf("\n");
  return 0;
}
