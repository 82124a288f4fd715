// Microbenchmark: ISSUE cost in shader cycles (s_memtime) of v_fma_f32 vs v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 for ONE wave's
// stream, 1 and 2 waves per SIMD, 8 independent chains per lane.  Cycles, not wall time: the clock ramps and sags.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, int iters) {
  f2 a[8];
  for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x + i, (float)threadIdx.x - i};
  const f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MODE == 1) a[j] = __builtin_elementwise_fma(a[j], m, c);
        else if (MODE == 2) a[j] = a[j] * m;
        else if (MODE == 3) a[j] = a[j] + c;
        else a[j].x = __builtin_fmaf(a[j].x, 1.0001f, 0.5f);
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f2 s = {0, 0};
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(float *d, unsigned long long *c, int wps) {
  const int iters = 20000;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, d, c, iters);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, d, c, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256];
  (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < 256; ++i) avg += (double)h[i];
  const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"};
  // wall time: the s_memtime tick is not wall time once several waves share a SIMD (it stays at 4.4 per instruction per wave)
  printf("%-13s waves/SIMD %d : %.2f ticks per instruction per wave; wall %.3f ms = %.2f ns per instruction per SIMD (%.2f cycles at 2.4 GHz)\n",
         names[MODE], wps, avg / 256.0 / ((double)iters * 64), ms, ms * 1e6 / ((double)iters * 64 * wps), ms * 1e6 / ((double)iters * 64 * wps) * 2.4);
}
int main() {
  float *d; unsigned long long *c;
  (void)hipMalloc(&d, 1 << 22); (void)hipMalloc(&c, 256 * 8);
  for (int wps : {1, 2, 3, 4}) /* 4 waves per SIMD = 1024 threads, the largest workgroup */ { run<0>(d, c, wps); run<1>(d, c, wps); run<2>(d, c, wps); run<3>(d, c, wps); }
  return 0;
}
