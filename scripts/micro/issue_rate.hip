// Microbenchmark: issue rate of fp32 vector instructions by OPERAND KIND and encoding, 1 .. 4 waves per SIMD, 8 independent chains per lane.
// Round 3's valu_chain.hip compiled to `v_fma_f32 v, v, s4, 0.5` (an SGPR and an inline constant) and found "one per ~4.2 cycles whatever
// the waves"; the rollout kernels' instructions mostly read VGPRs only.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
template <int MODE>
__global__ void k(float *out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float m = 1.0001f, c = 0.5f;
  float sm = __builtin_amdgcn_readfirstlane(__float_as_int(out[0])) ;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#define OPS8(I) I(%0) I(%1) I(%2) I(%3) I(%4) I(%5) I(%6) I(%7)
#define RUN(I) asm volatile(REP8(OPS8(I)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "s"(sm))
#define I_FMA_VVV(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define I_FMAC_E32(r) "v_fmac_f32_e32 " #r ", %8, %9\n"
#define I_MUL_E32(r) "v_mul_f32_e32 " #r ", " #r ", %8\n"
#define I_ADD_E32(r) "v_add_f32_e32 " #r ", " #r ", %8\n"
#define I_FMA_VSV(r) "v_fma_f32 " #r ", " #r ", %10, %9\n"
#define I_FMA_VVC(r) "v_fma_f32 " #r ", " #r ", %8, 0.5\n"
#define I_FMA_VSC(r) "v_fma_f32 " #r ", " #r ", %10, 0.5\n"
#define I_MUL_S(r) "v_mul_f32_e32 " #r ", %10, " #r "\n"
#define I_FMAAK(r) "v_fmaak_f32 " #r ", " #r ", %8, 0x3f000001\n"
#define I_PKFMA(r) "v_mov_b32 " #r ", " #r "\n"
    if (MODE == 0) RUN(I_FMA_VVV);
    else if (MODE == 1) RUN(I_FMAC_E32);
    else if (MODE == 2) RUN(I_MUL_E32);
    else if (MODE == 3) RUN(I_ADD_E32);
    else if (MODE == 4) RUN(I_FMA_VSV);
    else if (MODE == 5) RUN(I_FMA_VVC);
    else if (MODE == 6) RUN(I_FMA_VSC);
    else if (MODE == 7) RUN(I_MUL_S);
    else if (MODE == 8) RUN(I_FMAAK);
    else if (MODE == 9) RUN(I_PKFMA);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE>
void run(float *d, int wps) {
  const int iters = 4000;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, d, iters);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, d, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const char *names[] = {"v_fma_f32 v,v,v,v (VOP3)", "v_fmac_f32_e32 v,v,v", "v_mul_f32_e32 v,v,v", "v_add_f32_e32 v,v,v", "v_fma_f32 v,v,S,v", "v_fma_f32 v,v,v,0.5",
                         "v_fma_f32 v,v,S,0.5 (round 3's)", "v_mul_f32_e32 v,S,v", "v_fmaak_f32 v,v,v,literal", "v_mov_b32 v,v"};
  const double ns = ms * 1e6 / ((double)iters * 64 * wps);
  printf("waves/SIMD %d  %-34s wall %.3f ms = %.3f ns per instruction per SIMD = %.2f cycles at 2.4 GHz\n", wps, names[MODE], ms, ns, ns * 2.4);
}
int main() {
  float *d;
  (void)hipMalloc(&d, 1 << 22);
  (void)hipMemset(d, 0, 1 << 22);
  for (int wps : {1, 2, 3, 4}) {
    run<0>(d, wps); run<1>(d, wps); run<2>(d, wps); run<3>(d, wps); run<4>(d, wps); run<5>(d, wps); run<6>(d, wps); run<7>(d, wps); run<8>(d, wps); run<9>(d, wps);
  }
  return 0;
}
