// Microbenchmark: what does a NON-vector instruction cost a wave whose stream is a chain of fp32 vector instructions?  The rollout
// kernels' critical waves run alone (or with one partner) on a SIMD, so every instruction of their stream -- scalar ALU, exec-mask
// bookkeeping, branches, satisfied waits, register moves -- may take an issue slot of its own.  Per loop iteration: 64 v_fma_f32 (8
// chains) + 32 extra instructions of one kind, interleaved 2 : 1.  s_memtime ticks per iteration, 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)
template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float m = 1.0001f, c = 0.5f, t = 0.f;
  int s = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#define FMA2 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n"
#define FMA2b "v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
#define FMA2c "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
#define FMA2d "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define BODY(X) asm volatile(REP8(FMA2 X FMA2b X FMA2c X FMA2d X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "v"(t), "s"(s) : "scc", "vcc", "s40", "s41", "s42", "s43", "v100", "v101")
    if (MODE == 0) BODY("");
    else if (MODE == 1) BODY("s_add_u32 s40, s40, 1\n");
    else if (MODE == 2) BODY("s_nop 0\n");
    else if (MODE == 3) BODY("v_mov_b32 v100, v101\n");
    else if (MODE == 4) BODY("s_waitcnt lgkmcnt(0)\n");
    else if (MODE == 5) BODY("s_and_saveexec_b64 s[42:43], vcc\n s_or_b64 exec, exec, s[42:43]\n");  // (two instructions per slot: 64 extras)
    else if (MODE == 6) BODY("s_mul_i32 s40, s41, s41\n");
    else if (MODE == 7) BODY("s_cbranch_scc1 1f\n1:\n");  // never mind taken or not: falls to the next instruction either way
    else if (MODE == 8) BODY("v_cmp_lt_f32 vcc, %10, %10\n");
    else if (MODE == 9) BODY("v_cndmask_b32 v100, v101, v101, vcc\n");
    else if (MODE == 10) BODY("v_mul_f32 v100, v101, v101\n");  // an independent vector instruction: the baseline for 'costs a slot'
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(float *d, unsigned long long *c, int wps, double *base) {
  const int iters = 4000;
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
  const char *names[] = {"64 v_fma_f32 alone", "+ 32 s_add_u32", "+ 32 s_nop 0", "+ 32 v_mov_b32", "+ 32 s_waitcnt (satisfied)", "+ 32 x (s_and_saveexec, s_or exec)",
                         "+ 32 s_mul_i32", "+ 32 s_cbranch_scc1 (to next)", "+ 32 v_cmp_lt_f32", "+ 32 v_cndmask_b32", "+ 32 v_mul_f32 (independent)"};
  const double ns_iter = ms * 1e6 / iters / wps;  // per wave-iteration per SIMD
  if (MODE == 0) *base = ns_iter;
  const int extras = MODE == 5 ? 64 : 32;
  printf("waves/SIMD %d  %-36s wall %.3f ms = %7.2f ns per iteration per SIMD;  extra per added instruction: %5.2f ns = %5.2f cycles at 2.4 GHz  (a v_fma here: %.2f cycles)\n",
         wps, names[MODE], ms, ns_iter, MODE ? (ns_iter - *base) / extras : 0.0, MODE ? (ns_iter - *base) / extras * 2.4 : 0.0, *base / 64 * 2.4);
}
int main() {
  float *d; unsigned long long *c;
  (void)hipMalloc(&d, 1 << 22); (void)hipMalloc(&c, 256 * 8);
  for (int wps : {1, 2}) {
    double base = 0;
    run<0>(d, c, wps, &base); run<1>(d, c, wps, &base); run<2>(d, c, wps, &base); run<3>(d, c, wps, &base); run<4>(d, c, wps, &base); run<5>(d, c, wps, &base);
    run<6>(d, c, wps, &base); run<7>(d, c, wps, &base); run<8>(d, c, wps, &base); run<9>(d, c, wps, &base); run<10>(d, c, wps, &base);
  }
  return 0;
}
