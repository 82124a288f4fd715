// Microbenchmark: cycles per wave64 fp32 FMA for 1/2/4/8 independent dependency chains, 1 / 2 / 4 waves per SIMD.
// BUILD WITH -fno-slp-vectorize (as the library is): plain -O3 turns 2+ chains into v_pk_fma_f32 and the figure per FMA halves --
// that artefact is where rounds 1-2 got "independent FMAs issue every 2.5-3 cycles" from.  Without it (round 3, MI355X):
//   1 wave : 6.0 (one chain) / 4.8-5.0 (2-8 chains);  2 waves: 4.4-4.8 per SIMD;  4 waves: 4.15-4.4 per SIMD
// i.e. a SIMD issues one non-packed fp32 wave64 instruction per ~4.2 cycles, whatever the waves or their ILP.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int C>
__global__ void k(float *out, int iters) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64 / C; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) a[c] = __builtin_fmaf(a[c], 1.0001f, 0.5f);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int C>
void run(float *d, int wps) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  // one workgroup per CU (256 CUs), 4*wps waves each => wps waves per SIMD
  k<C><<<256, 256 * wps>>>(d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<C><<<256, 256 * wps>>>(d, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double instr = (double)iters * 64;
  printf("chains %d waves/SIMD %d : %.3f ms  -> %.2f ns per FMA per wave, %.2f cycles at 2.4 GHz (per SIMD: %.2f cycles per FMA)\n", C, wps, ms,
         ms * 1e6 / instr, ms * 1e6 / instr * 2.4, ms * 1e6 / instr * 2.4 / wps);
}
int main() {
  float *d;
  hipMalloc(&d, 1 << 22);
  for (int wps = 1; wps <= 4; wps *= 2) { run<1>(d, wps); run<2>(d, wps); run<4>(d, wps); run<8>(d, wps); }
  return 0;
}
