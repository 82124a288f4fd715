// Microbenchmark: how fast does ONE wave (per SIMD) run the real per-body arithmetic of a forward step, without LDS hand-overs,
// global memory or cross-wave waits?  Variants: V=0 integrate_fwd only; V=1 + joint_fwd (parent record read from LDS, static);
// V=2 + contact_point_fwd for one candidate per lane (the round-3 fused body).  W waves per SIMD.  Prints shader cycles per
// iteration (s_memtime) -- divide by the loop's instruction count from the ISA (scripts/isa_mix.py on the .s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../ppr-diffphys_amd/csrc/pd_device.h"

template <int V>
__global__ __launch_bounds__(512) void k(PdDevModel m, float *out, unsigned long long *cyc, int iters, float dt) {
  __shared__ float rec[16 * PD_RECF + 64];
  __shared__ float4 cull[16];
  const int l = threadIdx.x & 15;
  for (int i = threadIdx.x; i < 16 * PD_RECF + 64; i += blockDim.x) rec[i] = 0.01f * (i % 13);
  for (int i = threadIdx.x; i < 16; i += blockDim.x) { rec[i * PD_RECF + 6] = 1.0f; rec[i * PD_RECF + 16] = 1.f; rec[i * PD_RECF + 20] = 1.f; rec[i * PD_RECF + 24] = 1.f; cull[i] = make_float4(0.3f, 0.f, 1.f, 0.f); }
  __syncthreads();
  BodyConst c;
  c.type = PD_JOINT_REVOLUTE; c.parent = l > 0 ? (l - 1) / 3 * 3 : -1; c.pidx = c.parent >= 0 ? c.parent : 0;
  c.com = V3(0.01f, 0.02f, 0.f); c.axis = V3(1, 0, 0); c.axis_pj = V3(1, 0, 0); c.p_pj = V3(0.1f, -0.2f, 0.05f); c.q_pj = Q4(0, 0, 0, 1);
  c.q_off = Q4(0, 0, 0, 1); c.com_par = c.com; c.reach = 0.2f; c.sphere = make_float4(0, 0, 0, 0.1f);
  for (int k = 0; k < 3; ++k) { c.lim[k].lo = -1e30f; c.lim[k].up = 1e30f; c.lim[k].ke = 0.f; c.lim[k].kd = 0.f; }
  BodyState s;
  s.p = V3(0.1f * l, 0.4f, 0.f); s.r = Q4(0, 0, 0, 1); s.w = V3(0.1f, 0.2f, 0.3f); s.v = V3(0.f, -0.1f, 0.f);
  float Rm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  v3 rc = c.com;
  float I[9] = {0.01f, 0, 0, 0, 0.02f, 0, 0, 0, 0.015f}, invI[9] = {100.f, 0, 0, 0, 50.f, 0, 0, 0, 66.f};
  float tgt[1] = {0.1f}, act[1] = {0.f}, ke[1] = {220.f}, kd[1] = {2.f};
  float4 P = make_float4(0.01f * l, -0.3f, 0.02f, 0.f), M = make_float4(1e4f, 0.f, 1e2f, 1.f);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float accx = 0.f;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    v3 ft = V3(0.f, 0.f, 0.f), ff = ft;
    if (V >= 2) {
      ContactOut o;
      const int pb = (l * 7 + it) & 15;
      const bool t = contact_point_fwd(rec + pb * PD_RECF, cull[pb], P, M, o);
      ft = ft - (t ? o.t : V3(0, 0, 0)); ff = ff - (t ? o.f : V3(0, 0, 0));
    }
    if (V >= 1) {
      v3 wp_t, wp_f, wc_t, wc_f;
      joint_fwd<PD_JT_REVOLUTE, false, true>(m, c, s, rc, Rm, rec, tgt, act, ke, kd, wp_t, wp_f, wc_t, wc_f);
      ft = ft - wc_t; ff = ff - wc_f;
      accx += wp_t.x + wp_f.y;
    }
    float sink, R1[9];
    unsigned mask;
    s = integrate_fwd(m, c, s, Rm, rc, ft, ff, 2.0f, I, invI, dt, R1, rc, sink, mask);
#pragma unroll
    for (int k = 0; k < 9; ++k) Rm[k] = R1[k];
    accx += sink + (float)mask;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.p.x + s.r.w + accx;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(PdDevModel m, float *d, unsigned long long *c, int wps) {
  const int iters = 2000;
  hipLaunchKernelGGL(k<V>, dim3(256), dim3(256 * wps), 0, 0, m, d, c, 10, 5e-4f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(256), dim3(256 * wps), 0, 0, m, d, c, iters, 5e-4f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256];
  hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < 256; ++i) avg += (double)h[i];
  avg /= 256.0 * iters;
  printf("variant %d waves/SIMD %d : %.3f ms, %.0f shader cycles per iteration per wave (clock %.2f GHz)\n", V, wps, ms, avg, avg * iters / (ms * 1e6));
}
int main() {
  PdDevModel m{};
  m.nb = 13; m.gx = 0; m.gy = -9.8f; m.gz = 0; m.attach_ke = 16000.f; m.attach_kd = 200.f;
  float *d; unsigned long long *c;
  hipMalloc(&d, 1 << 22); hipMalloc(&c, 256 * 8);
  for (int wps = 1; wps <= 2; ++wps) { run<0>(m, d, c, wps); run<1>(m, d, c, wps); run<2>(m, d, c, wps); }
  return 0;
}
