// Fused per-frame pose loss: se3_loss of the reference (diffphys/dp_utils.py:113-138) and its
// gradient with respect to BOTH poses in one launch -- the torch composition is ~25 launches forward and ~40 backward
// on tensors of a few hundred thousand elements, i.e. pure launch latency next to a 0.3 ms rollout.
//   loss = |p_pred - p_gt|^2 + rot_ratio * acos(clamp((trace(R_pred R_gt^T) - 1) / 2, -1 + 1e-4, 1 - 1e-4)),   0 where an input is NaN
// DIM = 7: rotations are real-last quaternions (normalised by the conversion, 2 / |q|^2, like the reference's
// quaternion_to_matrix); DIM = 6: axis-angle vectors (first-order series below 1e-6 rad, like axis_angle_to_quaternion).
// One element per lane; HBM-bound by construction (2 * DIM floats in, 1 + 2 * DIM out), launch-latency-bound in practice.
#include <hip/hip_runtime.h>
#include "../../include/ppr_diffphys.h"
#include "pd_se3.h"
#include "pd_trajloss.h"

namespace {

template <int DIM>
__global__ __launch_bounds__(256) void k_se3_loss(int n, const float *__restrict__ pred, const float *__restrict__ gt, float rot_ratio,
                                                  float *__restrict__ loss, float *__restrict__ g_pred, float *__restrict__ g_gt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float p[DIM], g[DIM], op[DIM], og[DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k) { p[k] = pred[(size_t)i * DIM + k]; g[k] = gt[(size_t)i * DIM + k]; }
  loss[i] = pd_se3::se3_loss_eval<DIM>(p, g, rot_ratio, g_pred || g_gt, g_pred ? op : nullptr, g_gt ? og : nullptr);
  if (g_pred) {
#pragma unroll
    for (int k = 0; k < DIM; ++k) g_pred[(size_t)i * DIM + k] = op[k];
  }
  if (g_gt) {
#pragma unroll
    for (int k = 0; k < DIM; ++k) g_gt[(size_t)i * DIM + k] = og[k];
  }
}

}  // namespace
extern "C" int pd_se3_loss(int n, int dim, const float *pred_dev, const float *gt_dev, float rot_ratio, float *loss_dev, float *g_pred_dev,
                           float *g_gt_dev, void *stream) {
  if (n < 0 || (dim != 6 && dim != 7)) return 1;
  if (n == 0) return 0;
  if (!pred_dev || !gt_dev || !loss_dev) return 1;
  const dim3 grid((n + 255) / 256), block(256);
  if (dim == 7)
    hipLaunchKernelGGL(k_se3_loss<7>, grid, block, 0, (hipStream_t)stream, n, pred_dev, gt_dev, rot_ratio, loss_dev, g_pred_dev, g_gt_dev);
  else
    hipLaunchKernelGGL(k_se3_loss<6>, grid, block, 0, (hipStream_t)stream, n, pred_dev, gt_dev, rot_ratio, loss_dev, g_pred_dev, g_gt_dev);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// ---- reduce_loss(clip=True) on the rollout's [bs][F] table and the seeds of the adjoint that follows: device code in pd_trajloss.h
namespace {
template <bool IN_LDS>
__global__ __launch_bounds__(1024) void k_traj_loss_reduce(TrajReduceArgs r) {
  extern __shared__ float s_tab[];
  traj_loss_reduce_block<IN_LDS>(r, s_tab);
}
__global__ __launch_bounds__(256) void k_traj_seeds(TrajSeedsArgs s) { traj_seeds_block(s, blockIdx.x); }
}  // namespace

static int reduce_launch_impl(const TrajReduceArgs &r, hipStream_t st) {
  const size_t bytes = (size_t)r.bs * r.F * sizeof(float);
  if (bytes <= PD_REDUCE_LDS_BYTES) {
    // the attribute is PER DEVICE: a process that uses a second GPU must raise it there too (a 4096 x 4 table is already 64 KiB)
    static bool done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    if (dev >= 64 || !done[dev]) {
      if (hipFuncSetAttribute((const void *)k_traj_loss_reduce<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PD_REDUCE_LDS_BYTES) != hipSuccess) return 2;
      if (dev < 64) done[dev] = true;
    }
    hipLaunchKernelGGL(k_traj_loss_reduce<true>, dim3(1), dim3(1024), bytes, st, r);
  } else {
    hipLaunchKernelGGL(k_traj_loss_reduce<false>, dim3(1), dim3(1024), 0, st, r);
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" __attribute__((visibility("hidden"))) int pd_traj_loss_reduce_launch(int bs, int nframes, const float *table, float *reduced, float *scale, hipStream_t st) {
  if (bs < 0 || nframes < 0 || !reduced) return 1;
  if ((size_t)bs * nframes > 0 && (!table || !scale)) return 1;
  return reduce_launch_impl(TrajReduceArgs{bs, nframes, table, reduced, scale, 1, nullptr}, st);
}

// reduce_loss of the reference (diffphys/dp_utils.py:93-110) on ANY [bs][F] device table, as its own entry: the same one-workgroup code
// the rollout's trajectory-loss path runs, with the reference's in-place truncation of its argument.
extern "C" int pd_reduce_loss(int bs, int nframes, float *table_dev, int clip, float *reduced_dev, float *scale_dev, void *stream) {
  if (bs < 0 || nframes < 0 || !reduced_dev) return 1;
  if ((size_t)bs * nframes > 0 && !table_dev) return 1;
  return reduce_launch_impl(TrajReduceArgs{bs, nframes, table_dev, reduced_dev, scale_dev, clip ? 1 : 0, clip ? table_dev : nullptr}, (hipStream_t)stream);
}

extern "C" __attribute__((visibility("hidden"))) int pd_traj_seeds_launch(int bs, int nb, int nframes, const float *seed_pos, const float *scale, const float *gain,
                                                                         const float *adj_pos, const float *adj_vel, float *work, hipStream_t st) {
  if ((size_t)nframes * bs * nb == 0) return 0;
  const int blocks = pd_traj_seeds_blocks(bs, nb, nframes);
  const TrajSeedsArgs s{bs, nb, nframes, seed_pos, scale, gain, adj_pos, adj_vel, work, blocks};
  hipLaunchKernelGGL(k_traj_seeds, dim3(blocks), dim3(256), 0, st, s);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
