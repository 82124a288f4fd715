// Weight and bias gradient of a linear layer over n samples -- gw[m][kin] = sum_s g[s][m] x[s][kin], gb[m] = sum_s g[s][m] -- as ONE pair of
// launches on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, each 32 x 32 tile a k-ordered fmaf chain, so the result is a
// fixed-order sum: the same bits every time, eagerly and in a replayed HIP graph).
//
// Why it exists: the reference's real workload (main.py:86: 10 envs x 760 steps) evaluates three live time-MLPs on 7 600 samples; their 30
// weight gradients are 256 x 7 600 x {256, 512} products -- a tiny output and a long reduction, the shape a BLAS handles worst: rocBLAS takes
// 25.8 + 4.8 us (split-K + a reduction kernel), hipBLASLt 55 us, plus a second launch pair for the bias: 13 % of a training iteration.
//
// Mapping.  The reduction (sample) index is the MFMA's k: lane l of a wave holds A[i = l & 31][k = l >> 5] = g[s0 + (l >> 5)][m0 + (l & 31)] and
// B[k][j] = x[s0 + (l >> 5)][n0 + (l & 31)] -- for both operands consecutive lanes read consecutive floats of one row of g / x: every operand
// load is a coalesced 128-byte row segment straight from global memory / L2 into the MFMA's operand register, no LDS, no transpose.
// A wave owns a 64 x 64 output tile (2 x 2 MFMA tiles, 64 accumulator registers), a workgroup 2 x 2 waves = 128 x 128; the sample axis is cut
// into `slices` so that the launch has ~one workgroup per CU (256 x 256 output: 4 tiles x 64 slices).  Per pair of samples a wave issues 4
// loads and 4 MFMAs (256 matrix-core cycles); loads run one 8-pair block ahead of the MFMAs (registers double-buffered).  The bias gradient
// rides along: the A operand IS g, so the waves of the first column of tiles add it up as they go.  Partial tiles go to the workspace
// [slice][m][kin]; the second launch adds the slices in the order 0 .. S-1 (all loads of a thread in flight at once).
#include <hip/hip_runtime.h>
#include "../../include/ppr_diffphys.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PD_WG_U 8   // sample pairs per register block (16: 23.5 / 33.6 us at n = 7 600 against 22.7 / 31.0 -- a slice there is only 60 pairs; 48 against 50 us at n = 25 600)

struct WgradArgs {
  int n, m, kin, slices, rows_per_slice;  // rows_per_slice: even
  const float *g, *x;
  float *ws_w, *ws_b;                     // [slices][m][kin], [slices][m]
};

__device__ __forceinline__ void wg_load(const WgradArgs &a, int s, int s_half, int s_end, const float *ga, const float *xb, float (&A0)[PD_WG_U], float (&A1)[PD_WG_U],
                                        float (&B0)[PD_WG_U], float (&B1)[PD_WG_U]) {
#pragma unroll
  for (int u = 0; u < PD_WG_U; ++u) {
    const int r = s + 2 * u + s_half;      // this lane's sample: even k of the pair in lanes 0-31, odd in lanes 32-63
    const bool ok = r < s_end;
    const size_t og = (size_t)(ok ? r : 0) * a.m, ox = (size_t)(ok ? r : 0) * a.kin;   // (row 0 always exists: n >= 1)
    A0[u] = ga[og]; A1[u] = ga[og + 32]; B0[u] = xb[ox]; B1[u] = xb[ox + 32];   // (past the slice: row 0, masked where it is USED -- a select
  }                                                                              //  here would make the MFMAs in front wait for these loads)
}
// one register block: 8 sample pairs x (2 x 2 tiles); rows past the slice contribute zero
#define PD_WG_BLOCK(BUF, S0)                                                                    \
  _Pragma("unroll") for (int u = 0; u < PD_WG_U; ++u) {                                         \
    const bool ok = (S0) + 2 * u + half < s_end_all;                                            \
    const float a0 = ok ? A0[BUF][u] : 0.f, a1 = ok ? A1[BUF][u] : 0.f, b0 = B0[BUF][u], b1 = B1[BUF][u]; \
    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);                       \
    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);                       \
    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);                       \
    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);                       \
    sb0 += a0; sb1 += a1;                                                                       \
  }

__global__ __launch_bounds__(256) void k_linear_wgrad(WgradArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * 128 + wm * 64, n0 = blockIdx.y * 128 + wn * 64, slice = blockIdx.z;
  const int half = lane >> 5, c = lane & 31;
  const int s_begin = slice * a.rows_per_slice, s_end_all = min(a.n, s_begin + a.rows_per_slice);
  // lane's view: sample s + half, column m0 + c (+32) of g / n0 + c (+32) of x
  const float *ga = a.g + m0 + c, *xb = a.x + n0 + c;
  const int s_end = s_end_all;
  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
  float sb0 = 0.f, sb1 = 0.f;              // bias partial sums (columns m0 + c, m0 + 32 + c), this lane's half of the samples
  float A0[2][PD_WG_U], A1[2][PD_WG_U], B0[2][PD_WG_U], B1[2][PD_WG_U];
  int s = s_begin;
  wg_load(a, s, half, s_end, ga, xb, A0[0], A1[0], B0[0], B1[0]);
  for (; s < s_end_all; s += 4 * PD_WG_U) {
    wg_load(a, s + 2 * PD_WG_U, half, s_end, ga, xb, A0[1], A1[1], B0[1], B1[1]);
    __builtin_amdgcn_sched_barrier(0);   // (the next block's 32 loads are issued BEFORE this block's MFMAs: the scheduler otherwise sinks them to their uses)
    PD_WG_BLOCK(0, s)
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 * PD_WG_U >= s_end_all) break;
    wg_load(a, s + 4 * PD_WG_U, half, s_end, ga, xb, A0[0], A1[0], B0[0], B1[0]);
    __builtin_amdgcn_sched_barrier(0);
    PD_WG_BLOCK(1, s + 2 * PD_WG_U)
    __builtin_amdgcn_sched_barrier(0);
  }
  // C/D map of the 32 x 32 MFMA: register r of lane l is C[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
  float *w = a.ws_w + ((size_t)slice * a.m + m0) * a.kin + n0 + c;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    w[(size_t)row * a.kin] = acc00[r];
    w[(size_t)row * a.kin + 32] = acc01[r];
    w[(size_t)(row + 32) * a.kin] = acc10[r];
    w[(size_t)(row + 32) * a.kin + 32] = acc11[r];
  }
  if (a.ws_b && blockIdx.y == 0 && wn == 0) {  // even samples (lanes 0-31) + odd samples (lanes 32-63), in that order
    const float o0 = __shfl_down(sb0, 32), o1 = __shfl_down(sb1, 32);
    if (half == 0) {
      a.ws_b[(size_t)slice * a.m + m0 + c] = sb0 + o0;
      a.ws_b[(size_t)slice * a.m + m0 + 32 + c] = sb1 + o1;
    }
  }
}

// out[i] = ws[0][i] + ws[1][i] + ... in that order; blocks past the weight entries do the bias
__global__ __launch_bounds__(256) void k_linear_wgrad_reduce(int slices, int count_w, int count_b, const float *__restrict__ ws_w, const float *__restrict__ ws_b,
                                                             float *__restrict__ gw, float *__restrict__ gb) {
  int i = blockIdx.x * 256 + threadIdx.x;
  const float *src = ws_w;
  float *dst = gw;
  int count = count_w;
  if (i >= count_w) { i -= count_w; src = ws_b; dst = gb; count = count_b; }
  if (i >= count) return;
  float t = src[i];
  for (int s0 = 1; s0 < slices; s0 += 32) {   // 32 loads in flight per thread (8: the kernel was eight dependent round trips long, 5.7 us)
    float v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = s0 + u < slices ? src[(size_t)(s0 + u) * count + i] : 0.f;
#pragma unroll
    for (int u = 0; u < 32; ++u) t = s0 + u < slices ? t + v[u] : t;
  }
  dst[i] = t;
}

inline int wgrad_slices(int n, int m, int kin) {
  const int tiles = (m / 128) * (kin / 128);
  int s = (256 + tiles - 1) / tiles;            // ~ one workgroup per CU (swept 64 .. 512 workgroups at n = 7 600: 46 / 29 / 24 / 23 / 29 / 30 us)
  const int max_s = (n + 15) / 16;              // at least 16 samples per slice
  if (s > max_s) s = max_s;
  return s < 1 ? 1 : s;
}

}  // namespace

extern "C" size_t pd_linear_wgrad_workspace_floats(int n, int m, int kin) {
  if (n <= 0 || m <= 0 || kin <= 0 || m % 128 || kin % 128) return 0;   // unsupported shape: the caller keeps its BLAS path
  return (size_t)wgrad_slices(n, m, kin) * ((size_t)m * kin + m);
}

extern "C" int pd_linear_wgrad(int n, int m, int kin, const float *g_dev, const float *x_dev, float *gw_dev, float *gb_dev, float *ws_dev, void *stream) {
  if (pd_linear_wgrad_workspace_floats(n, m, kin) == 0) return 1;
  if (!g_dev || !x_dev || !gw_dev || !ws_dev) return 1;
  WgradArgs a{};
  a.n = n; a.m = m; a.kin = kin; a.g = g_dev; a.x = x_dev;
  a.slices = wgrad_slices(n, m, kin);
  a.rows_per_slice = (((n + a.slices - 1) / a.slices) + 1) & ~1;
  a.slices = (n + a.rows_per_slice - 1) / a.rows_per_slice;
  a.ws_w = ws_dev;
  a.ws_b = gb_dev ? ws_dev + (size_t)a.slices * m * kin : nullptr;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_linear_wgrad, dim3(m / 128, kin / 128, a.slices), dim3(256), 0, st, a);
  const int count_w = m * kin, count_b = gb_dev ? m : 0;
  hipLaunchKernelGGL(k_linear_wgrad_reduce, dim3((count_w + 255) / 256 + (count_b + 255) / 256), dim3(256), 0, st, a.slices, count_w, count_b, a.ws_w, a.ws_b,
                     gw_dev, gb_dev);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
