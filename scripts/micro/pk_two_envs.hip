// Microbenchmark behind the "two envs per LANE" lead (EXPERIMENTS.md, round 5): the body integration of one sim step
// (integrator_euler.py:21-91: quaternion rotations, cross products, 3x3 products, normalisation, clamps -- ~190 instructions, the mix of the
// rollout kernels' body wave) written ONCE over a scalar type T and instantiated with
//     T = float   : a lane owns one env;  run at TWO waves per SIMD, as the headline kernels do (256 VGPRs each)
//     T = float2  : a lane owns the same body of TWO envs, every value an ext_vector_type(2) -> clang emits v_pk_{fma,mul,add}_f32 for the
//                   arithmetic with no register-pairing moves (the pair IS the layout); run at ONE wave per SIMD (twice the registers)
// Prints shader cycles per iteration per wave and the time for the same number of env-steps: the ratio is what the rewrite could win on the
// arithmetic (hand-overs, LDS and memory not included).   hipcc --offload-arch=gfx950 -O3 -o pk_two_envs pk_two_envs.hip && ./pk_two_envs
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
template <class T> struct S;
template <> struct S<float> {
  static __device__ __forceinline__ float rsqrt(float x) { return 1.0f / sqrtf(x); }
  static __device__ __forceinline__ float clamp(float x, float l) { return fminf(fmaxf(x, -l), l); }
  static __device__ __forceinline__ float nz(float x) { return x != 0.0f ? 1.0f : 0.0f; }
  static __device__ __forceinline__ float sum(float x) { return x; }
  static constexpr int ENVS = 1;
};
template <> struct S<f2> {
  static __device__ __forceinline__ f2 rsqrt(f2 x) { return f2{1.0f / sqrtf(x.x), 1.0f / sqrtf(x.y)}; }
  static __device__ __forceinline__ f2 clamp(f2 x, float l) { return f2{fminf(fmaxf(x.x, -l), l), fminf(fmaxf(x.y, -l), l)}; }
  static __device__ __forceinline__ f2 nz(f2 x) { return f2{x.x != 0.0f ? 1.0f : 0.0f, x.y != 0.0f ? 1.0f : 0.0f}; }
  static __device__ __forceinline__ float sum(f2 x) { return x.x + x.y; }
  static constexpr int ENVS = 2;
};
template <class T> struct V3 { T x, y, z; };
template <class T> struct Q4 { T x, y, z, w; };
#define DEV template <class T> __device__ __forceinline__
DEV V3<T> add(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV V3<T> sub(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV V3<T> mul(V3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
DEV T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV V3<T> cross(V3<T> a, V3<T> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DEV V3<T> rot(Q4<T> q, V3<T> v) {
  V3<T> u{q.x, q.y, q.z};
  const T two = T(2.0f);
  return add(add(mul(v, two * q.w * q.w - T(1.0f)), mul(cross(u, v), two * q.w)), mul(u, two * dot(u, v)));
}
DEV V3<T> rot_inv(Q4<T> q, V3<T> v) {
  V3<T> u{q.x, q.y, q.z};
  const T two = T(2.0f);
  return add(sub(mul(v, two * q.w * q.w - T(1.0f)), mul(cross(u, v), two * q.w)), mul(u, two * dot(u, v)));
}
DEV V3<T> matvec(const T *M, V3<T> a) {
  return {M[0] * a.x + M[1] * a.y + M[2] * a.z, M[3] * a.x + M[4] * a.y + M[5] * a.z, M[6] * a.x + M[7] * a.y + M[8] * a.z};
}

template <class T>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters, float dtf) {
  const int l = threadIdx.x & 63;
  const T dt = T(dtf);
  V3<T> p{T(0.1f * l), T(0.4f), T(0.0f)}, w{T(0.1f), T(0.2f + 0.001f * l), T(0.3f)}, v{T(0.0f), T(-0.1f), T(0.0f)};
  Q4<T> r{T(0.0f), T(0.01f * l), T(0.0f), T(1.0f)};
  const V3<T> com{T(0.01f), T(0.02f), T(0.0f)}, g{T(0.0f), T(-9.8f), T(0.0f)};
  T I[9] = {T(0.01f), T(0.0f), T(0.0f), T(0.0f), T(0.02f), T(0.0f), T(0.0f), T(0.0f), T(0.015f)};
  T invI[9] = {T(100.f), T(0.0f), T(0.0f), T(0.0f), T(50.f), T(0.0f), T(0.0f), T(0.0f), T(66.f)};
  const T inv_m = T(2.0f);
  V3<T> ft{T(0.01f), T(0.0f), T(0.02f)}, ff{T(0.0f), T(19.0f), T(0.1f)};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const V3<T> x_com = add(p, rot(r, com));
    V3<T> v1 = add(v, mul(add(mul(ff, inv_m), mul(g, S<T>::nz(inv_m))), dt));
    const V3<T> x1 = add(x_com, mul(v1, dt));
    const V3<T> wb = rot_inv(r, w);
    const V3<T> tb = sub(rot_inv(r, ft), cross(wb, matvec(I, wb)));
    V3<T> w1 = rot(r, add(wb, mul(matvec(invI, tb), dt)));
    // r1 = normalize(r + quat(w1, 0) * r * 0.5 dt)
    const T h = T(0.5f) * dt;
    Q4<T> dq{w1.x * r.w + w1.y * r.z - w1.z * r.y, w1.y * r.w + w1.z * r.x - w1.x * r.z, w1.z * r.w + w1.x * r.y - w1.y * r.x,
             -(w1.x * r.x + w1.y * r.y + w1.z * r.z)};
    Q4<T> r1{r.x + dq.x * h, r.y + dq.y * h, r.z + dq.z * h, r.w + dq.w * h};
    const T il = S<T>::rsqrt(r1.x * r1.x + r1.y * r1.y + r1.z * r1.z + r1.w * r1.w);
    r1 = {r1.x * il, r1.y * il, r1.z * il, r1.w * il};
    w1 = mul(w1, T(1.0f) - T(0.1f) * dt);
    w1 = {S<T>::clamp(w1.x, 10.f), S<T>::clamp(w1.y, 10.f), S<T>::clamp(w1.z, 10.f)};
    v1 = {S<T>::clamp(v1.x, 10.f), S<T>::clamp(v1.y, 10.f), S<T>::clamp(v1.z, 10.f)};
    p = sub(x1, rot(r1, com));
    r = r1; w = w1; v = v1;
    ft = {ft.x + p.x * T(1e-3f), ft.y, ft.z - w.y * T(1e-3f)};  // (keep the wrench live and data-dependent)
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = S<T>::sum(p.x + r.w + w.z + v.y);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class T>
double run(float *d, unsigned long long *c, int waves_per_simd, const char *tag) {
  const int iters = 4000, blocks = 256;
  hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256 * waves_per_simd), 0, 0, d, c, 10, 5e-4f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256 * waves_per_simd), 0, 0, d, c, iters, 5e-4f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256];
  hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < blocks; ++i) avg += (double)h[i];
  avg /= (double)blocks * iters;
  const double env_steps = (double)blocks * 256 * waves_per_simd * S<T>::ENVS * iters;
  printf("%-34s %d wave(s) per SIMD: %7.0f cycles per iteration per wave, %.3f ms, %.3e lane-env-steps/s\n", tag, waves_per_simd, avg, ms, env_steps / (ms * 1e-3));
  return env_steps / (ms * 1e-3);
}
int main() {
  float *d; unsigned long long *c;
  hipMalloc(&d, 1 << 22); hipMalloc(&c, 256 * 8);
  const double a1 = run<float>(d, c, 1, "float  (one env per lane)");
  const double a2 = run<float>(d, c, 2, "float  (one env per lane)");
  const double b1 = run<f2>(d, c, 1, "float2 (two envs per lane, packed)");
  const double b2 = run<f2>(d, c, 2, "float2 (two envs per lane, packed)");
  printf("packed, one wave per SIMD, against scalar at two waves per SIMD (today's configuration): %.2f x;  both at two waves: %.2f x;  scalar 2 vs 1 waves: %.2f x\n",
         b1 / a2, b2 / a2, a2 / a1);
  return 0;
}
