// Microbenchmark: throughput of packed fp32 (v_pk_fma_f32) against plain v_fma_f32 on gfx950, 1 and 2 waves per SIMD,
// 8 independent chains per lane.  Prints ns per instruction per wave and the flops ratio.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ void k(float *out, int iters) {
  f2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x + i, (float)threadIdx.x - i};
  const f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (PK) a[j] = __builtin_elementwise_fma(a[j], m, c);
        else a[j].x = __builtin_fmaf(a[j].x, 1.0001f, 0.5f);
      }
  }
  f2 s = {0, 0};
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <bool PK>
void run(float *d, int wps) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<PK><<<256, 256 * wps>>>(d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<PK><<<256, 256 * wps>>>(d, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double instr = (double)iters * 64;
  printf("%s waves/SIMD %d : %.3f ms -> %.2f ns per instruction per wave (per SIMD %.2f ns)\n", PK ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms,
         ms * 1e6 / instr, ms * 1e6 / instr / wps);
}
int main() {
  float *d;
  hipMalloc(&d, 1 << 22);
  for (int wps : {1, 2, 4}) { run<false>(d, wps); run<true>(d, wps); }
  return 0;
}
