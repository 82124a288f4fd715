// Kernel argument blocks and launcher prototypes shared by the kernel TUs and the host TU.
#pragma once
#include "pd_device.h"

// Rollout kernels: 8 waves per workgroup.  Waves 0..3 ("body waves") own the per-body state of 64/SEGW envs each;
// wave 4+i ("contact wave") runs the ground-contact sweeps for the envs of wave i, concurrently with wave i's joint
// work, on the same SIMD (a workgroup's waves are dealt to the 4 SIMDs cyclically).  FK kernels: 4 body waves only.
#define PD_BWAVES 4
#define PD_BLOCK (2 * PD_BWAVES * 64)
#define PD_FK_BLOCK (PD_BWAVES * 64)
#define PD_BLOCK3 (3 * PD_BWAVES * 64)  // 3-role adjoint (k_rollout_bwd3): integrate, contact and joint wave per env group

// Forward sweeps log their hit list (count + up to PD_HITLOG-1 entries) so that the adjoint replays it instead of
// repeating the three cull levels; count -1 = did not fit, the adjoint then culls again for that wave.
#define PD_HITLOG 32

enum { PD_K_ROLLOUT_FWD = 0, PD_K_ROLLOUT_BWD = 1, PD_K_FK_FWD = 2, PD_K_FK_BWD = 3 };

// Which rollout launches run wave-specialised (pd_kernels.hip: launch_jt) -- shared with the host so that it can report
// the launch geometry.  Adjoint: revolute-only robots.  Forward: every joint mix while a CU holds at most one workgroup.
constexpr bool pd_split(int jt) { return jt == PD_JT_REVOLUTE; }
inline bool pd_split_launch(int kind, int jt, int nblocks, int cu_count) {
  return pd_split(jt) || (kind == PD_K_ROLLOUT_FWD && nblocks <= cu_count);
}
inline int pd_block_threads(int kind, int jt, int nblocks, int cu_count, int variant = 0) {
  if (kind == PD_K_ROLLOUT_BWD) return pd_split(jt) ? (variant == 3 ? PD_BLOCK3 : PD_BLOCK) : (variant == 9 ? PD_FK_BLOCK : PD_BLOCK);
  return (kind == PD_K_ROLLOUT_FWD && pd_split_launch(kind, jt, nblocks, cu_count)) ? PD_BLOCK : PD_FK_BLOCK;
}

#define PD_TRAJ_FLOATS 20  // floats of saved trajectory per body-step: 5 float4 planes (pd_kernels.hip: PD_TRAJ_G)

struct RolloutArgs {
  int bs, nsteps, nframes;
  float dt;
  const float *q_init, *qd_init, *torques, *res_f, *refs, *target_ke, *target_kd, *inv_mass, *inertia, *inv_inertia;
  const int *frame_of_step;
  float *ws;                           // workspace: saved trajectory [T][5 planes][N] float4, N = bs*nb, then the hit log
  float *wp_pos, *wp_vel, *grf, *jaf;  // forward outputs (grf/jaf may be null)
  // backward only
  const float *adj_pos, *adj_vel;
  float *g_q_init, *g_qd_init, *g_torques, *g_res_f, *g_refs, *g_ke, *g_kd, *g_inv_mass, *g_inertia, *g_inv_inertia;
  int *hitlog;              // workspace tail: per (step, env) the compacted contact hit list of the forward sweep
  unsigned long long *dbg;  // diagnostic builds only (-DPD_STAMPS): per-phase cycle sums, [block][8]
  int variant;              // A/B experiments (pd_debug_set_variant, not in the public header): which adjoint kernel a revolute robot runs
};


// Batched FK (ForwardKinematics, dp_model.py:1022-1130 of the reference): n articulations, one per segment.
struct FkArgs {
  int n;
  const float *joint_q, *joint_qd;
  float *body_q, *body_qd;                 // forward outputs [n][nb][7] / [n][nb][6]
  const float *adj_body_q, *adj_body_qd;   // backward inputs
  float *g_joint_q, *g_joint_qd;           // backward outputs
};


hipError_t pd_launch_seg16(int kind, int jt, const PdDevModel &m, const void *args, int nblocks, size_t lds, hipStream_t st);
hipError_t pd_launch_seg32(int kind, int jt, const PdDevModel &m, const void *args, int nblocks, size_t lds, hipStream_t st);
hipError_t pd_launch_seg64(int kind, int jt, const PdDevModel &m, const void *args, int nblocks, size_t lds, hipStream_t st);
hipError_t pd_set_lds_seg16(int jt, int bytes);
hipError_t pd_set_lds_seg32(int jt, int bytes);
hipError_t pd_set_lds_seg64(int jt, int bytes);
