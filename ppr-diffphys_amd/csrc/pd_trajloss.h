// Device code of the trajectory-loss launches of row f4, shared by pd_loss.hip (the plain launches) and pd_kernels.hip
// (the launches that also carry the FK of the control reference as extra workgroups: k_reduce_fk / k_seeds_fk).
#pragma once
#include <hip/hip_runtime.h>

struct TrajReduceArgs {
  int bs, F;
  const float *table;      // [bs][F] per-frame losses the rollout kernel wrote
  float *reduced, *scale;  // [4], [bs][F] (scale may be null)
  int clip;                // reduce_loss(clip=...): 1 on the rollout's path (dp_model.py:779)
  float *table_after;      // null, or where the table goes after the clip's assignments (the reference truncates its argument in place)
};
struct TrajSeedsArgs {
  int bs, nb, F;
  const float *seed_pos, *scale, *gain, *adj_pos, *adj_vel;
  float *work;
  int nblocks;             // workgroups of the seeds pass (a combined launch appends its FK workgroups after them)
};

// ---- reduce_loss(loss_traj, clip=True) of the reference (diffphys/dp_utils.py:93-110) on the [bs][F] table the rollout kernel wrote
// (k_rollout_fwd<..., LOSS>), ONE workgroup, and each entry's share of the result -- what the adjoint rollout scales its seeds with:
//   th        10 x the (lower, like torch.median) median of the positive entries of ENV 0; env 0 has none: NaN, and nothing is clipped
//   clipping  per env, entries from the first one above th on are ASSIGNED zero (loss_seq[i, clip_idx:] = 0)
//   value     mean of the positive entries left when their sum is positive, else the mean of all entries
//   scale     0 for an assigned-zero entry; else 1 / N_pos for a positive entry (0 for the others) or 1 / (bs F) in the "else" case
// reduced[0..3] = value, th, N_pos, number of clipped envs.  Sums are double, in a fixed order (run-to-run reproducible).
#define PD_REDUCE_LDS_BYTES (128 * 1024)  // the [bs][F] table of one reduce_loss launch held in LDS when it fits
// IN_LDS: the table is copied into LDS once (coalesced) and the three passes read it there -- read where it lies, every thread walks
// its envs' rows with F dependent global loads per pass (measured at 4096 x 4: 26.8 us against the rollout's 220)
template <bool IN_LDS>
__device__ __forceinline__ void traj_loss_reduce_block(const TrajReduceArgs &r, float *s_tab) {  // one workgroup (any multiple of 64 threads up to 1024)
  const int bs = r.bs, F = r.F;
  const float *table = r.table;  // (pd_reduce_loss truncates it in place: no restrict)
  float *__restrict__ reduced = r.reduced, *__restrict__ scale = r.scale;
  __shared__ float s_med;
  __shared__ double s_sum[16], s_tot[16];
  __shared__ int s_cnt[16], s_clip[16];
  const int tid = threadIdx.x, NT = blockDim.x;
  const size_t n_all = (size_t)bs * F;
  if (tid == 0) s_med = 0.f;
  if (IN_LDS) {  // sixteen loads in flight per thread (one at a time: a memory latency each, 16 of them at 4096 x 4)
    for (size_t i0 = tid; i0 < n_all; i0 += (size_t)16 * NT) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = i0 + (size_t)u * NT < n_all ? table[i0 + (size_t)u * NT] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) if (i0 + (size_t)u * NT < n_all) s_tab[i0 + (size_t)u * NT] = v[u];
    }
  }
  __syncthreads();
  auto tab = [&](size_t i) { return IN_LDS ? s_tab[i] : table[i]; };
  // the threshold comes from env 0 ALONE (the reference computes it at i == 0 and never again, dp_utils.py:98-100): 10 x the lower
  // median (torch.median) of env 0's positive entries by rank counting (ties broken by index: ranks are distinct).  Env 0 without a
  // positive entry: the reference's median of an empty selection is NaN (torch >= 1.8; tests/golden/ref_host_reduce_loss.npz holds what
  // it returned), `th == 0` is never true again and `loss > NaN` is false: NOTHING is clipped in the whole batch.  Same here.
  float th = __builtin_nanf("");  // clip == 0: stays NaN, nothing exceeds it
  if (bs > 0 && r.clip) {
    int np = 0;
    for (int j = 0; j < F; ++j) np += tab(j) > 0.f;
    for (int i = tid; i < F; i += NT) {
      const float v = tab(i);
      if (!(v > 0.f)) continue;
      int rank = 0;
      for (int j = 0; j < F; ++j) {
        const float u = tab(j);
        rank += (u > 0.f) && (u < v || (u == v && j < i));
      }
      if (rank == (np - 1) / 2) s_med = v;
    }
    __syncthreads();
    if (np > 0) th = s_med * 10.f;
  }
  // per env: first exceedance, then the sums over what is left
  double sum = 0.0, tot = 0.0;
  int cnt = 0, clip = 0;
  for (int e = tid; e < bs; e += NT) {
    bool cut = false;
    for (int f = 0; f < F; ++f) {
      const float v = tab((size_t)e * F + f);
      if (!cut && v > th) { cut = true; ++clip; }
      const float w = cut ? 0.f : v;
      tot += (double)w;
      if (w > 0.f) { sum += (double)w; ++cnt; }
    }
  }
  // fixed-order sums: butterfly inside the wave, then the (at most 16) wave totals in index order
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    sum += __shfl_xor(sum, w); tot += __shfl_xor(tot, w); cnt += __shfl_xor(cnt, w); clip += __shfl_xor(clip, w);
  }
  if ((tid & 63) == 0) { s_sum[tid >> 6] = sum; s_tot[tid >> 6] = tot; s_cnt[tid >> 6] = cnt; s_clip[tid >> 6] = clip; }
  __syncthreads();
  double S = 0.0, T = 0.0;
  int N = 0, C = 0;
  for (int w = 0; w < (NT + 63) / 64; ++w) { S += s_sum[w]; T += s_tot[w]; N += s_cnt[w]; C += s_clip[w]; }
  const bool pos_case = T > 0.0;
  const long long all = (long long)bs * F;
  if (tid == 0) {
    reduced[0] = pos_case ? (float)(S / (double)(N > 0 ? N : 1)) : (all > 0 ? (float)(T / (double)all) : 0.f);
    reduced[1] = th; reduced[2] = (float)N; reduced[3] = (float)C;
  }
  const float share_pos = N > 0 ? 1.0f / (float)N : 0.f, share_all = all > 0 ? 1.0f / (float)all : 0.f;
  if (IN_LDS) {  // the scales go out coalesced: each thread rewrites its envs' rows in LDS first
    for (int e = tid; e < bs; e += NT) {
      bool cut = false;
      for (int f = 0; f < F; ++f) {
        const float v = s_tab[(size_t)e * F + f];
        if (!cut && v > th) cut = true;
        if (r.table_after) r.table_after[(size_t)e * F + f] = cut ? 0.f : v;  // (standalone pd_reduce_loss only: small tables)
        s_tab[(size_t)e * F + f] = cut ? 0.f : (pos_case ? (v > 0.f ? share_pos : 0.f) : share_all);
      }
    }
    __syncthreads();
    if (scale) for (size_t i = tid; i < n_all; i += NT) scale[i] = s_tab[i];
  } else {
    for (int e = tid; e < bs; e += NT) {
      bool cut = false;
      for (int f = 0; f < F; ++f) {
        const float v = table[(size_t)e * F + f];
        if (!cut && v > th) cut = true;
        if (scale) scale[(size_t)e * F + f] = cut ? 0.f : (pos_case ? (v > 0.f ? share_pos : 0.f) : share_all);
        if (r.table_after) r.table_after[(size_t)e * F + f] = cut ? 0.f : v;
      }
    }
  }
}

// ---- the frame seeds of an adjoint rollout that follows pd_rollout_forward_traj_loss, written where the plain adjoint kernel reads
// its adj_pos / adj_vel rows:  work_pos [F][bs*nb][7] = gain[0] * scale[env][frame] / nb * seed_pos (+ adj_pos),  work_vel [F][bs*nb][6]
// = adj_vel or 0.  A zero share is an assignment in the reference (loss_seq[i, idx:] = 0, loss_traj[outseq_idx] = 0): nothing flows
// through it, not 0 * inf.  A few MB, one pass: 4-8 us at the headline size.
__device__ __forceinline__ void traj_seeds_block(const TrajSeedsArgs &s, int block) {  // 256 threads per workgroup, s.nblocks of them
  const size_t N = (size_t)s.bs * s.nb, n_pos = (size_t)s.F * N * 7, n_all = n_pos + (size_t)s.F * N * 6;
  const float g = s.gain[0] / (float)s.nb;
  for (size_t i = (size_t)block * 256 + threadIdx.x; i < n_all; i += (size_t)s.nblocks * 256) {
    if (i < n_pos) {
      const size_t row = i / 7, f = row / N, e = (row % N) / s.nb;
      const float k = g * s.scale[e * s.F + f];
      const float v = k != 0.0f ? k * s.seed_pos[i] : 0.0f;
      s.work[i] = s.adj_pos ? v + s.adj_pos[i] : v;
    } else {
      s.work[i] = s.adj_vel ? s.adj_vel[i - n_pos] : 0.0f;
    }
  }
}
inline int pd_traj_seeds_blocks(int bs, int nb, int nframes) {
  const size_t n = (size_t)nframes * bs * nb * 13;
  return (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
}
