// Microbenchmark: does a wave64 VALU instruction cost fewer cycles when whole 16-lane quarters of EXEC are off?
// One workgroup of 64 threads per CU-ish grid; each lane runs a long chain of independent FMAs under a lane mask.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out, int active, int iters) {
  const int lane = threadIdx.x & 63;
  float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
  long long t0 = 0, t1 = 0;
  if (lane < active) {
    t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll 16
      for (int u = 0; u < 16; ++u) {
        a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 1.0001f, 0.5f); a2 = __builtin_fmaf(a2, 1.0001f, 0.5f);
        a3 = __builtin_fmaf(a3, 1.0001f, 0.5f); a4 = __builtin_fmaf(a4, 1.0001f, 0.5f); a5 = __builtin_fmaf(a5, 1.0001f, 0.5f);
        a6 = __builtin_fmaf(a6, 1.0001f, 0.5f); a7 = __builtin_fmaf(a7, 1.0001f, 0.5f);
      }
    }
    t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (lane == 0 && blockIdx.x == 0) ((long long *)out)[4096] = t1 - t0;
  }
}
int main() {
  float *d;
  hipMalloc(&d, 1 << 20);
  const int iters = 2000;
  for (int waves = 1; waves <= 2; ++waves)
    for (int active : {64, 48, 32, 16, 13, 8, 1}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      k<<<256 * 4 * waves, 64>>>(d, active, 10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      k<<<256 * 4 * waves, 64>>>(d, active, iters);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      long long cyc;
      hipMemcpy(&cyc, (char *)d + 4096 * 8, 8, hipMemcpyDeviceToHost);
      printf("waves/SIMD %d active lanes %2d : %.3f ms, %.2f shader-clock ticks per VALU instr (wave 0)\n", waves, active, ms,
             (double)cyc / (iters * 16.0 * 8.0));
    }
  return 0;
}
