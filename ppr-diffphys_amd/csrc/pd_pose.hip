// SE(3) pose algebra of the loss plumbing as single launches.
//
// The reference composes these from dqtorch's CUDA quaternion kernels and torch ops (diffphys/dp_utils.py:22-31
// compose_delta, :60-73 rotate_frame, :76-84 rotate_frame_vel; diffphys/geom_utils.py:148-203 se3_vec2mat / se3_mat2vec);
// the torch composition is ~200-450 launches forward and twice that backward per call on a few ten thousand poses:
// launch latency only.  Here every op is ONE launch forward and ONE launch for the vector-Jacobian product.
//
// The functions are written once, as templates over the scalar type.  The forward kernel instantiates them with float; the
// VJP kernel instantiates them with a dual number (value, one tangent) and sweeps the input directions -- forward-mode
// differentiation of exactly the code that produced the value, so the branch taken by the value (small-angle series, the
// best-conditioned quaternion candidate, clamps) is the branch that is differentiated, with torch's subgradient choices
// (d|a|/da = 0 at 0, sqrt_pos' = 0 at x <= 0, clamp' = 1 at the bound).  13-14 sweeps of ~150 flops per pose: nothing.
//
// Conventions (same as diffphys_amd/geom_utils.py): 7-vector = (p, q) with the quaternion's real part LAST; 6-vector =
// (p, axis-angle); matrices row-major; quaternion_to_matrix divides by |q|^2 (mocap quaternions arrive un-normalised).
#include <hip/hip_runtime.h>
#include "../../include/ppr_diffphys.h"

namespace {

struct Dual { float v, d; };
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { const float q = a.v / b.v; return {q, (a.d - q * b.d) / b.v}; }
__device__ __forceinline__ Dual operator+(float a, Dual b) { return {a + b.v, b.d}; }
__device__ __forceinline__ Dual operator-(float a, Dual b) { return {a - b.v, -b.d}; }
__device__ __forceinline__ Dual operator*(float a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator/(float a, Dual b) { const float q = a / b.v; return {q, -q * b.d / b.v}; }
__device__ __forceinline__ Dual operator*(Dual a, float b) { return {a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator/(Dual a, float b) { return {a.v / b, a.d / b}; }

__device__ __forceinline__ float val(float x) { return x; }
__device__ __forceinline__ float val(Dual x) { return x.v; }
__device__ __forceinline__ float t_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ Dual t_sqrt(Dual x) { const float s = sqrtf(x.v); return {s, 0.5f * x.d / s}; }
__device__ __forceinline__ float t_sin(float x) { return sinf(x); }
__device__ __forceinline__ Dual t_sin(Dual x) { return {sinf(x.v), cosf(x.v) * x.d}; }
__device__ __forceinline__ float t_cos(float x) { return cosf(x); }
__device__ __forceinline__ Dual t_cos(Dual x) { return {cosf(x.v), -sinf(x.v) * x.d}; }
__device__ __forceinline__ void set_const(float &x, float c) { x = c; }
__device__ __forceinline__ void set_const(Dual &x, float c) { x = {c, 0.0f}; }

// (r, i, j, k) real-FIRST -> 3x3 row-major   (geom_utils.quaternion_to_matrix)
template <class T>
__device__ __forceinline__ void quat_to_mat(T r, T i, T j, T k, T *R) {
  const T two_s = 2.0f / (r * r + i * i + j * j + k * k);
  R[0] = 1.0f - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
  R[3] = two_s * (i * j + k * r); R[4] = 1.0f - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
  R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1.0f - two_s * (i * i + j * j);
}

// axis-angle -> quaternion real-first   (geom_utils.axis_angle_to_quaternion: series below 1e-6 rad)
template <class T>
__device__ __forceinline__ void aa_to_quat(const T *a, T *q) {
  const T n2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  T ang;
  if (val(n2) > 0.0f) ang = t_sqrt(n2); else set_const(ang, 0.0f);  // torch.norm: zero subgradient at 0
  const T half = 0.5f * ang;
  T s;
  if (fabsf(val(ang)) < 1e-6f) s = 0.5f - ang * ang / 48.0f; else s = t_sin(half) / ang;
  q[0] = t_cos(half); q[1] = a[0] * s; q[2] = a[1] * s; q[3] = a[2] * s;
}

// 3x3 row-major -> quaternion real-first, the best-conditioned of the four candidate forms   (geom_utils.matrix_to_quaternion)
template <class T>
__device__ __forceinline__ void mat_to_quat(const T *m, T *q) {
  const T x[4] = {1.0f + m[0] + m[4] + m[8], 1.0f + m[0] - m[4] - m[8], 1.0f - m[0] + m[4] - m[8], 1.0f - m[0] - m[4] + m[8]};
  T qa[4];
  int best = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (val(x[k]) > 0.0f) qa[k] = t_sqrt(x[k]); else set_const(qa[k], 0.0f);  // _sqrt_pos
    if (k > 0 && val(qa[k]) > val(qa[best])) best = k;                        // argmax: first maximum
  }
  const T a = m[7] - m[5], b = m[2] - m[6], c = m[3] - m[1], d = m[3] + m[1], e = m[2] + m[6], f = m[5] + m[7];
  T cand[4], den = qa[best];
  if (best == 0) { cand[0] = qa[0] * qa[0]; cand[1] = a; cand[2] = b; cand[3] = c; }
  else if (best == 1) { cand[0] = a; cand[1] = qa[1] * qa[1]; cand[2] = d; cand[3] = e; }
  else if (best == 2) { cand[0] = b; cand[1] = d; cand[2] = qa[2] * qa[2]; cand[3] = f; }
  else { cand[0] = c; cand[1] = e; cand[2] = f; cand[3] = qa[3] * qa[3]; }
  if (!(val(den) >= 0.1f)) set_const(den, 0.1f);  // clamp(min=0.1)
  den = 2.0f * den;
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = cand[k] / den;
}

// (p, q real-last) 7-vector -> R, p
template <class T>
__device__ __forceinline__ void pose7(const T *v, T *R, T *p) {
  quat_to_mat(v[6], v[3], v[4], v[5], R);
  p[0] = v[0]; p[1] = v[1]; p[2] = v[2];
}

// T = T_a @ T_b as the 4x4 product computes it, then se3_mat2vec(outdim 7)
template <class T>
__device__ __forceinline__ void compose_out7(const T *Ra, const T *pa, const T *Rb, const T *pb, T *out) {
  T R[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) R[3 * i + j] = Ra[3 * i] * Rb[j] + Ra[3 * i + 1] * Rb[3 + j] + Ra[3 * i + 2] * Rb[6 + j];
    out[i] = Ra[3 * i] * pb[0] + Ra[3 * i + 1] * pb[1] + Ra[3 * i + 2] * pb[2] + pa[i];
  }
  T q[4];
  mat_to_quat(R, q);
  out[3] = q[1]; out[4] = q[2]; out[5] = q[3]; out[6] = q[0];
}

enum { OP_COMPOSE_DELTA = PD_POSE_COMPOSE_DELTA, OP_ROTATE_FRAME = PD_POSE_ROTATE_FRAME, OP_ROTATE_VEL = PD_POSE_ROTATE_VEL };

template <int OP> struct Shape;
template <> struct Shape<OP_COMPOSE_DELTA> { static constexpr int NA = 7, NB = 6, NO = 7; };  // a = target pose, b = delta (p, axis-angle)
template <> struct Shape<OP_ROTATE_FRAME> { static constexpr int NA = 7, NB = 7, NO = 7; };   // a = global pose, b = target pose
template <> struct Shape<OP_ROTATE_VEL> { static constexpr int NA = 7, NB = 6, NO = 6; };     // a = global pose, b = (linear, angular)

template <int OP, class T>
__device__ __forceinline__ void pose_fn(const T *a, const T *b, T *out) {
  if (OP == OP_COMPOSE_DELTA) {  // se3_mat2vec(se3_vec2mat(delta) @ se3_vec2mat(target))   dp_utils.py:22-31
    T q[4], R1[9], R2[9], p2[3];
    aa_to_quat(b + 3, q);
    quat_to_mat(q[0], q[1], q[2], q[3], R1);
    pose7(a, R2, p2);
    compose_out7(R1, b, R2, p2, out);
  } else if (OP == OP_ROTATE_FRAME) {  // se3_mat2vec(se3_vec2mat(global) @ se3_vec2mat(target))   dp_utils.py:60-73
    T Rg[9], pg[3], R2[9], p2[3];
    pose7(a, Rg, pg);
    pose7(b, R2, p2);
    compose_out7(Rg, pg, R2, p2, out);
  } else {  // both halves rotated by the rotation of global   dp_utils.py:76-84
    T Rg[9];
    quat_to_mat(a[6], a[3], a[4], a[5], Rg);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 3; ++i) out[3 * h + i] = Rg[3 * i] * b[3 * h] + Rg[3 * i + 1] * b[3 * h + 1] + Rg[3 * i + 2] * b[3 * h + 2];
  }
}

template <int OP>
__global__ __launch_bounds__(256) void k_pose_fwd(int n, const float *__restrict__ a, int a_stride, const float *__restrict__ b,
                                                  float *__restrict__ out) {
  using S = Shape<OP>;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float av[S::NA], bv[S::NB], o[S::NO];
#pragma unroll
  for (int k = 0; k < S::NA; ++k) av[k] = a[(size_t)i * a_stride + k];
#pragma unroll
  for (int k = 0; k < S::NB; ++k) bv[k] = b[(size_t)i * S::NB + k];
  pose_fn<OP, float>(av, bv, o);
#pragma unroll
  for (int k = 0; k < S::NO; ++k) out[(size_t)i * S::NO + k] = o[k];
}

template <int OP>
__global__ __launch_bounds__(256) void k_pose_vjp(int n, const float *__restrict__ a, int a_stride, const float *__restrict__ b,
                                                  const float *__restrict__ g_out, float *__restrict__ g_a, float *__restrict__ g_b) {
  using S = Shape<OP>;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Dual av[S::NA], bv[S::NB], o[S::NO];
  float g[S::NO];
#pragma unroll
  for (int k = 0; k < S::NA; ++k) av[k] = {a[(size_t)i * a_stride + k], 0.0f};
#pragma unroll
  for (int k = 0; k < S::NB; ++k) bv[k] = {b[(size_t)i * S::NB + k], 0.0f};
#pragma unroll
  for (int k = 0; k < S::NO; ++k) g[k] = g_out[(size_t)i * S::NO + k];
  for (int dir = 0; dir < S::NA + S::NB; ++dir) {  // not unrolled: one copy of the function body
    const bool in_a = dir < S::NA;
    if (in_a ? g_a == nullptr : g_b == nullptr) continue;
#pragma unroll
    for (int k = 0; k < S::NA; ++k) av[k].d = (in_a && k == dir) ? 1.0f : 0.0f;
#pragma unroll
    for (int k = 0; k < S::NB; ++k) bv[k].d = (!in_a && k == dir - S::NA) ? 1.0f : 0.0f;
    pose_fn<OP, Dual>(av, bv, o);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < S::NO; ++k) s += g[k] * o[k].d;
    if (in_a) g_a[(size_t)i * S::NA + dir] = s; else g_b[(size_t)i * S::NB + (dir - S::NA)] = s;
  }
}

// Lowest ground-contact candidate of each pose set: min over candidates c of  p_y[body(c)] + (R(q[body(c)]) point_c)_y - dist_c.
// One workgroup per pose set; ties go to the lowest candidate index; a NaN height wins (propagates like torch.min).
__global__ __launch_bounds__(256) void k_foot_height(int nb, int nc, const float *__restrict__ body_q, const int *__restrict__ c_body,
                                                     const float *__restrict__ c_point, const float *__restrict__ c_dist,
                                                     float *__restrict__ h_out, int *__restrict__ arg_out) {
  __shared__ float s_h[256];
  __shared__ int s_i[256];
  const float *X = body_q + (size_t)blockIdx.x * nb * 7;
  float best = INFINITY;
  int best_i = 0x7fffffff;
  bool best_nan = false;
  for (int c = threadIdx.x; c < nc; c += 256) {
    const float *x = X + (size_t)c_body[c] * 7;
    const float qx = x[3], qy = x[4], qz = x[5], w = x[6];
    const float px = c_point[3 * c], py = c_point[3 * c + 1], pz = c_point[3 * c + 2];
    const float rot_y = py * (2.0f * w * w - 1.0f) + 2.0f * w * (qz * px - qx * pz) + 2.0f * qy * (qx * px + qy * py + qz * pz);
    const float h = x[1] + rot_y - c_dist[c];
    const bool is_nan = h != h;
    if (!best_nan && (is_nan || h < best)) { best = h; best_i = c; best_nan = is_nan; }
  }
  s_h[threadIdx.x] = best; s_i[threadIdx.x] = best_i;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const float h0 = s_h[threadIdx.x], h1 = s_h[threadIdx.x + off];
      const int i0 = s_i[threadIdx.x], i1 = s_i[threadIdx.x + off];
      const bool n0 = h0 != h0, n1 = h1 != h1;
      const bool take1 = n0 || n1 ? (n1 && (!n0 || i1 < i0)) : (h1 < h0 || (h1 == h0 && i1 < i0));
      if (take1) { s_h[threadIdx.x] = h1; s_i[threadIdx.x] = i1; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { h_out[blockIdx.x] = s_h[0]; arg_out[blockIdx.x] = s_i[0]; }
}

// d height / d body_q: non-zero for the body of the arg-min candidate only (the gradient torch.min routes to its index)
__global__ __launch_bounds__(256) void k_foot_height_vjp(int n, int nb, const float *__restrict__ body_q, const int *__restrict__ c_body,
                                                         const float *__restrict__ c_point, const int *__restrict__ arg,
                                                         const float *__restrict__ g_h, float *__restrict__ g_body_q) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // (pose set, body)
  if (i >= n * nb) return;
  const int set = i / nb, body = i - set * nb, c = arg[set];
  float o[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c >= 0 && c_body[c] == body) {
    const float *x = body_q + (size_t)i * 7;
    const float qx = x[3], qy = x[4], qz = x[5], w = x[6], g = g_h[set];
    const float px = c_point[3 * c], py = c_point[3 * c + 1], pz = c_point[3 * c + 2];
    o[1] = g;
    o[3] = g * (-2.0f * w * pz + 2.0f * qy * px);
    o[4] = g * (2.0f * (qx * px + qy * py + qz * pz) + 2.0f * qy * py);
    o[5] = g * (2.0f * w * px + 2.0f * qy * pz);
    o[6] = g * (4.0f * w * py + 2.0f * (qz * px - qx * pz));
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) g_body_q[(size_t)i * 7 + k] = o[k];
}

template <int OP>
int launch_pose(int n, const float *a, int a_bcast, const float *b, float *out, const float *g_out, float *g_a, float *g_b, hipStream_t st) {
  const dim3 grid((n + 255) / 256), block(256);
  const int a_stride = a_bcast ? 0 : Shape<OP>::NA;
  if (out)
    hipLaunchKernelGGL(k_pose_fwd<OP>, grid, block, 0, st, n, a, a_stride, b, out);
  else
    hipLaunchKernelGGL(k_pose_vjp<OP>, grid, block, 0, st, n, a, a_stride, b, g_out, g_a, g_b);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

int pose_dispatch(int op, int n, const float *a, int a_bcast, const float *b, float *out, const float *g_out, float *g_a, float *g_b, void *stream) {
  if (n < 0 || op < 0 || op > 2) return 1;
  if (n == 0) return 0;
  if (!a || !b) return 1;
  hipStream_t st = (hipStream_t)stream;
  switch (op) {
    case OP_COMPOSE_DELTA: return launch_pose<OP_COMPOSE_DELTA>(n, a, a_bcast, b, out, g_out, g_a, g_b, st);
    case OP_ROTATE_FRAME: return launch_pose<OP_ROTATE_FRAME>(n, a, a_bcast, b, out, g_out, g_a, g_b, st);
    default: return launch_pose<OP_ROTATE_VEL>(n, a, a_bcast, b, out, g_out, g_a, g_b, st);
  }
}

// Column sums of a row-major [n][k] matrix -- the bias gradient of a linear layer (n samples) and the gradient of a broadcast operand.
// Two fixed-order stages, no atomics: workgroup (column slab of 32, row slice s of PD_COLSUM_SLICES) -- thread (phase p = tid / 32, column
// c = tid % 32) adds the rows p, p + 8, ... of its slice (a 128-byte coalesced read per row and phase, four independent partial sums per
// thread to keep loads in flight), the 8 phases are added in LDS in the order 0 .. 7 -> ws[s][col]; the second launch adds the slices in the
// order 0 .. S-1.  The same bits every time, eagerly and inside a captured HIP graph (torch's multi-block sum(0) re-arms a semaphore with a
// memset that a graph replay does not re-execute on this stack).  Small n: one slice, straight to out, one launch.
__global__ __launch_bounds__(256) void k_colsum(int n, int k, int rows_per_slice, const float *__restrict__ x, float *__restrict__ dst) {
  __shared__ float part[8][33];
  const int c = threadIdx.x & 31, p = threadIdx.x >> 5, col = blockIdx.x * 32 + c;
  const int r0 = blockIdx.y * rows_per_slice, r1 = min(n, r0 + rows_per_slice);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < k) {
    const float *xc = x + col;
    int r = r0 + p;
    for (; r + 24 < r1; r += 32) {
      s0 += xc[(size_t)r * k]; s1 += xc[(size_t)(r + 8) * k]; s2 += xc[(size_t)(r + 16) * k]; s3 += xc[(size_t)(r + 24) * k];
    }
    for (; r < r1; r += 8) s0 += xc[(size_t)r * k];
  }
  part[p][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (p == 0 && col < k) {
    float t = part[0][c];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += part[q][c];
    dst[(size_t)blockIdx.y * k + col] = t;
  }
}
__global__ __launch_bounds__(64) void k_colsum_final(int k, int slices, const float *__restrict__ ws, float *__restrict__ out) {
  const int col = blockIdx.x * 64 + threadIdx.x;
  if (col >= k) return;
  float v[PD_COLSUM_SLICES];   // all the slices' loads in flight at once (32 dependent round trips otherwise: 7 us for 32 KB), then the fixed order
#pragma unroll
  for (int s = 0; s < PD_COLSUM_SLICES; ++s) v[s] = s < slices ? ws[(size_t)s * k + col] : 0.0f;
  float t = v[0];
#pragma unroll
  for (int s = 1; s < PD_COLSUM_SLICES; ++s) t = s < slices ? t + v[s] : t;
  out[col] = t;
}

}  // namespace

extern "C" int pd_colsum(int n, int k, const float *x_dev, float *out_dev, float *ws_dev, void *stream) {
  if (n < 0 || k < 0) return 1;
  if (k == 0) return 0;
  if (!out_dev || (n > 0 && !x_dev)) return 1;
  hipStream_t st = (hipStream_t)stream;
  const unsigned slabs = (unsigned)((k + 31) / 32);
  if (n <= 1024 || !ws_dev) {  // one slice: straight to out
    hipLaunchKernelGGL(k_colsum, dim3(slabs, 1), dim3(256), 0, st, n, k, n > 0 ? n : 1, x_dev, out_dev);
    return hipGetLastError() == hipSuccess ? 0 : 2;
  }
  const int rows = (((n + PD_COLSUM_SLICES - 1) / PD_COLSUM_SLICES) + 7) & ~7;   // a multiple of the 8 row phases
  const int slices = (n + rows - 1) / rows;
  hipLaunchKernelGGL(k_colsum, dim3(slabs, (unsigned)slices), dim3(256), 0, st, n, k, rows, x_dev, ws_dev);
  hipLaunchKernelGGL(k_colsum_final, dim3((unsigned)((k + 63) / 64)), dim3(64), 0, st, k, slices, ws_dev, out_dev);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int pd_pose_op(int op, int n, const float *a_dev, int a_broadcast, const float *b_dev, float *out_dev, void *stream) {
  if (n > 0 && !out_dev) return 1;
  return pose_dispatch(op, n, a_dev, a_broadcast, b_dev, out_dev, nullptr, nullptr, nullptr, stream);
}

extern "C" int pd_pose_op_vjp(int op, int n, const float *a_dev, int a_broadcast, const float *b_dev, const float *g_out_dev, float *g_a_dev,
                              float *g_b_dev, void *stream) {
  if (n > 0 && (!g_out_dev || (!g_a_dev && !g_b_dev))) return 1;
  return pose_dispatch(op, n, a_dev, a_broadcast, b_dev, nullptr, g_out_dev, g_a_dev, g_b_dev, stream);
}

extern "C" int pd_foot_height(int n, int nb, int nc, const float *body_q_dev, const int *c_body_dev, const float *c_point_dev,
                              const float *c_dist_dev, float *height_dev, int *arg_dev, void *stream) {
  if (n < 0 || nb <= 0 || nc <= 0) return 1;
  if (n == 0) return 0;
  if (!body_q_dev || !c_body_dev || !c_point_dev || !c_dist_dev || !height_dev || !arg_dev) return 1;
  hipLaunchKernelGGL(k_foot_height, dim3(n), dim3(256), 0, (hipStream_t)stream, nb, nc, body_q_dev, c_body_dev, c_point_dev, c_dist_dev,
                     height_dev, arg_dev);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int pd_foot_height_vjp(int n, int nb, const float *body_q_dev, const int *c_body_dev, const float *c_point_dev, const int *arg_dev,
                                  const float *g_height_dev, float *g_body_q_dev, void *stream) {
  if (n < 0 || nb <= 0) return 1;
  if (n == 0) return 0;
  if (!body_q_dev || !c_body_dev || !c_point_dev || !arg_dev || !g_height_dev || !g_body_q_dev) return 1;
  const long long total = (long long)n * nb;
  hipLaunchKernelGGL(k_foot_height_vjp, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, nb, body_q_dev, c_body_dev,
                     c_point_dev, arg_dev, g_height_dev, g_body_q_dev);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
