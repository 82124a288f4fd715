// Kernel argument blocks and launcher prototypes shared by the kernel TUs and the host TU.
#pragma once
#include "pd_device.h"
#include "pd_trajloss.h"

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

enum { PD_K_ROLLOUT_FWD = 0, PD_K_ROLLOUT_BWD = 1, PD_K_FK_FWD = 2, PD_K_FK_BWD = 3,
       // row f4: the FK of the control reference rides on the trajectory-loss launches that already sit between the rollout launches --
       // workgroup 0 of PD_K_REDUCE_FK is reduce_loss (pd_trajloss.h), the rest are FK forward workgroups; PD_K_SEEDS_FK = the adjoint's
       // seeds pass followed by FK backward workgroups
       PD_K_REDUCE_FK = 4, PD_K_SEEDS_FK = 5 };
#define PD_REDUCE_BLOCK 1024  // threads of the reduce_loss workgroup; the FK workgroups of that launch run 16 body waves

// Which rollout launches run wave-specialised -- shared by the kernel TUs and the host.
//   forward : revolute-only robots always; other joint mixes while a CU holds at most one full workgroup (the latency
//             regime: human at 1024 envs -32 %), else the unsplit kernel packs twice as many body waves per SIMD
//   adjoint : revolute-only robots the 2-role kernel (body + contact wave); other joint mixes the 2-role k_rollout_bwd3
//             (integrate + contacts wave, joint wave).  The variants that were measured and rejected (3-role, early hand-over, unsplit
//             compound adjoint: EXPERIMENTS.md) left the sources in round 5; `git log` has them (commit 837bfbd and before).
constexpr bool pd_split(int jt) { return jt == PD_JT_REVOLUTE; }
// the specialised instantiations (one joint type) are only launched for PLAIN models: non-FREE joints all hang on a body, child
// joint frames are not rotated (pd_host.hip)
constexpr bool pd_parented(int jt) { return jt == PD_JT_REVOLUTE || jt == PD_JT_COMPOUND; }
enum { PD_KV_FWD_SPLIT = 0, PD_KV_FWD_UNSPLIT, PD_KV_BWD_2ROLE, PD_KV_BWD_2ROLE_EARLY, PD_KV_BWD_3ROLE, PD_KV_BWD3_2ROLE, PD_KV_BWD_UNSPLIT, PD_KV_FK,
       PD_KV_FWD_QUAD, PD_KV_BWD_QUAD };  // quad-lane (four lanes per body) small-batch kernels, 64-lane mapping, revolute-only plain models
struct PdLaunchCfg {
  int kernel;    // PD_KV_*
  int roles;     // waves per env group
  int groups;    // env groups (of 64 / segw envs) per workgroup, 1 .. PD_BWAVES
  int nblocks, threads;
  size_t lds;    // dynamic LDS bytes per workgroup
};
// Env groups per workgroup: as many as it takes to cover the batch with one workgroup per compute unit, at most PD_BWAVES.
// A small batch (512 Laikago envs = 128 groups) then occupies 128 CUs with one group each instead of 32 CUs with four.
inline int pd_groups_per_wg(int n_groups, int cu_count) {
  int g = cu_count > 0 ? (n_groups + cu_count - 1) / cu_count : PD_BWAVES;
  return g < 1 ? 1 : (g > PD_BWAVES ? PD_BWAVES : g);
}
inline int pd_kernel_variant(int kind, int jt, int n_groups, int cu_count) {
  // (the unsplit forward exists for compound-only robots alone: the generic instantiation, all joint types in one kernel, needs 7-14
  // VGPRs more than a wave may have at two waves per SIMD -- scratch traffic in its step loop -- so generic robots take the split kernel
  // at every batch size: no shipped rollout kernel spills vector registers, round 5)
  if (kind == PD_K_ROLLOUT_FWD) return (jt != PD_JT_COMPOUND || n_groups <= PD_BWAVES * cu_count) ? PD_KV_FWD_SPLIT : PD_KV_FWD_UNSPLIT;
  if (kind == PD_K_ROLLOUT_BWD) {
    return pd_split(jt) ? PD_KV_BWD_2ROLE : PD_KV_BWD3_2ROLE;
  }
  return PD_KV_FK;
}
// The forward kernels' cull wave (wave-specialised kernels of revolute-only robots, pd_kernels.hip CULLW) keeps its candidate list and a
// SHORT tile list in the room the adjoint's wider per-hit slots leave in the shared per-env LDS size: the tile list's capacity in entries
// (the kernel computes the same from its pointers).  Below PD_CULLW_MIN_CAP the launch stays with two roles.
#define PD_CULLW_MIN_CAP 48
inline int pd_fwd_cull_cap(int nb, int segw, int list_cap, int env_lds_floats, int rec, int w6) {
  const int spec_off = ((4 + rec + 2 * w6) * nb + w6 + 3) & ~3;
  return env_lds_floats - (spec_off + 8 * nb + 4 + list_cap + 8 * segw + 6 * segw + 8 * segw);
}
inline int pd_variant_roles(int kv) {
  return kv == PD_KV_BWD_3ROLE ? 3 : ((kv == PD_KV_FWD_UNSPLIT || kv == PD_KV_BWD_UNSPLIT || kv == PD_KV_FK) ? 1 : 2);
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
  // trajectory loss evaluated where the frame poses are produced (pd_rollout_forward_traj_loss / _backward_traj_loss; row f4):
  // forward (LOSS instantiations only): se3_loss of every frame pose against loss_target [bs][F][nb][7], its unscaled gradients to
  // loss_seed_pos [F][bs*nb][7] (the layout of adj_pos) / loss_seed_gt [bs][F][nb][7] (may be null), the per-frame mean over the
  // env's bodies to loss_table [bs][F] (0 where loss_outseq [bs][F] is set; may be null)
  const float *loss_target;
  const unsigned char *loss_outseq;
  float loss_rot_ratio;
  float *loss_seed_pos, *loss_seed_gt, *loss_table;
  unsigned long long *dbg;  // diagnostic builds only (-DPD_STAMPS): per-phase cycle sums, [block][8]
};


// Batched FK (ForwardKinematics, dp_model.py:1022-1130 of the reference): n articulations, one per segment.
struct FkArgs {
  int n;
  const float *joint_q, *joint_qd;
  float *body_q, *body_qd;                 // forward outputs [n][nb][7] / [n][nb][6]
  const float *adj_body_q, *adj_body_qd;   // backward inputs
  float *g_joint_q, *g_joint_qd;           // backward outputs
  // perm_bs > 0: the n = F * perm_bs articulations come frame-major (row f * bs + e, the layout of ForwardKinematics' rj_q [F, bs, nq])
  // and the body rows (outputs, adjoint inputs) are env-major, row e * F + f -- the [bs, F, nb, .] tensors phys_model.forward uses
  // (dp_model.py:1093-1094 of the reference permutes and copies)
  int perm_bs;
};
struct ReduceFkArgs { FkArgs fk; TrajReduceArgs red; int in_lds; };
struct SeedsFkArgs { FkArgs fk; TrajSeedsArgs seeds; };


hipError_t pd_launch_seg16(int kind, int jt, const PdDevModel &m, const void *args, const PdLaunchCfg &cfg, hipStream_t st);
hipError_t pd_launch_seg32(int kind, int jt, const PdDevModel &m, const void *args, const PdLaunchCfg &cfg, hipStream_t st);
hipError_t pd_launch_seg64(int kind, int jt, const PdDevModel &m, const void *args, const PdLaunchCfg &cfg, hipStream_t st);
hipError_t pd_set_lds_seg16(int jt, int bytes);
hipError_t pd_set_lds_seg32(int jt, int bytes);
hipError_t pd_set_lds_seg64(int jt, int bytes);
// the same launchers of the PD_NUM_LITERAL objects (rollout kinds only; pd_math.h PD_POLICY)
hipError_t pd_launch_seg16_literal(int kind, int jt, const PdDevModel &m, const void *args, const PdLaunchCfg &cfg, hipStream_t st);
hipError_t pd_launch_seg32_literal(int kind, int jt, const PdDevModel &m, const void *args, const PdLaunchCfg &cfg, hipStream_t st);
hipError_t pd_launch_seg64_literal(int kind, int jt, const PdDevModel &m, const void *args, const PdLaunchCfg &cfg, hipStream_t st);
hipError_t pd_set_lds_seg16_literal(int jt, int bytes);
hipError_t pd_set_lds_seg32_literal(int jt, int bytes);
hipError_t pd_set_lds_seg64_literal(int jt, int bytes);
