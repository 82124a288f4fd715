/*
 * ppr_diffphys.h -- C ABI of libpprdiffphys_hip.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the ONE hot path of gengshan-y/ppr-diffphys:
 * the batched semi-implicit-Euler rigid-body rollout, its reverse-mode adjoint,
 * and forward kinematics.  Plain pointers and sizes only; every `float*` /
 * `int*` argument named *_dev is DEVICE memory (HBM) owned by the caller
 * (in the Python host: torch tensors), `stream` is a hipStream_t passed as void*.
 * All functions return 0 on success, non-zero on error (pd_last_error() has the text).
 * Nothing here allocates, frees or synchronises in the launch path (graph-capturable);
 * only pd_model_create / pd_model_destroy touch the allocator.
 *
 * What each entry point replaces in the reference (file:line under /root/reference):
 *
 *   pd_model_create      wp.sim.ModelBuilder.finalize + Model.collide
 *                        (diffphys/dp_model.py:384-401; arrays per SURVEY.md App. A.2)
 *   pd_rollout_forward   ForwardWarp.forward (diffphys/dp_model.py:1146-1249): eval_fk, then per step
 *                        clear_forces + wp_add (:1210-1221) + SemiImplicitIntegrator.simulate
 *                        (diffphys/integrator_euler.py:579-620: eval_body_contacts :93-179,
 *                        eval_body_joints :289-451, integrate_bodies :21-91), frame gather (:1241-1248)
 *   pd_rollout_backward  ForwardWarp.backward (diffphys/dp_model.py:1251-1400): wp.Tape.backward over the
 *                        same launches, gradients of the 11 inputs
 *   pd_fk_forward/backward  ForwardKinematics.forward/backward (diffphys/dp_model.py:1022-1130),
 *                        i.e. warp.sim.articulation.eval_fk and its tape adjoint
 *
 * Layouts are the reference's flat env-major ones:
 *   q_init [bs*nq]   qd_init [bs*nqd] (root twist as (w, v))
 *   torques, refs [T][bs*nqd]         res_f [T][bs*nb][6] (tau, f)
 *   target_ke, target_kd [bs*nqd]     body_mass, body_inv_mass [bs*nb]
 *   body_inertia, body_inv_inertia [bs*nb][3][3] row-major
 *   wp_pos [F][bs*nb][7] (p, q=xyzw)  wp_vel [F][bs*nb][6] (w, v)   grf, jaf [F][bs*nb][6]
 * The saved trajectory (workspace) is internal SoA: see pd_rollout_workspace_floats.
 */
#ifndef PPR_DIFFPHYS_H
#define PPR_DIFFPHYS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PD_ABI_VERSION 9   /* 2: host frame2step (validated), per-model timing, per-env joint_X_p binding; 3: pose algebra + foot height; 4: gradient post-processing (remove_nan / FK clamp) inside the kernels, pd_build_id; 5: pd_model_contact_order (decoding the hit log), pd_rollout_forward_traj_loss / pd_rollout_backward_traj_loss (row f4), pd_model_set_kernel_family (quad-lane small-batch kernels); 6: pd_rollout_forward_traj_loss_fk / pd_rollout_backward_traj_loss_fk (the FK of the control reference rides on the trajectory-loss launches); 7: pd_reduce_loss (reduce_loss on any table, threshold from env 0 as the reference takes it), pd_model_set_numeric_policy (the reference's literal acos forms as a run-time mode); 8: pd_colsum; 9: pd_linear_wgrad (the time-MLPs' weight + bias gradients on the fp32 matrix cores) */

/* Articulation template: HOST pointers, copied by pd_model_create.  One template for all envs. */
typedef struct pd_model_desc {
  int nb, nq, nqd;              /* bodies (= joints), joint coords, joint dofs per env */
  int nc, nmat;                 /* ground-contact candidate points, materials */
  const int *joint_type;        /* [nb] Warp codes: 1 revolute, 3 fixed, 4 free, 5 compound */
  const int *joint_parent;      /* [nb] -1 for the root; parents precede children */
  const int *joint_q_start;     /* [nb] */
  const int *joint_qd_start;    /* [nb] */
  const float *joint_X_p;       /* [nb][7] parent-frame joint transform (p, q) */
  const float *joint_X_c;       /* [nb][7] child-frame joint transform */
  const float *joint_axis;      /* [nb][3] */
  const float *body_com;        /* [nb][3] */
  const float *joint_limit_lower, *joint_limit_upper, *joint_limit_ke, *joint_limit_kd; /* [nqd] */
  const int *contact_body;      /* [nc] body index of each candidate point */
  const float *contact_point;   /* [nc][3] body frame */
  const float *contact_dist;    /* [nc] shape thickness */
  const int *contact_material;  /* [nc] index into shape_materials */
  const float *shape_materials; /* [nmat][4] (ke, kd, kf, mu) */
  float gravity[3];
  float joint_attach_ke, joint_attach_kd;
} pd_model_desc;

typedef struct pd_model pd_model; /* opaque; owns the device copy of the template */

int pd_abi_version(void);
/* "<git HEAD at build time>+<first 16 hex digits of the sha256 over the library's sources>" (csrc/Makefile: SRCS, in that order).
 * A loader that also has the sources (tests, __graft_entry__.smoke) compares the hash: a stale binary cannot pass for the tree's. */
const char *pd_build_id(void);
const char *pd_last_error(void);

int pd_model_create(const pd_model_desc *desc, pd_model **out);
void pd_model_destroy(pd_model *m);
/* Lanes of a 64-wide wavefront given to one articulation: 16, 32 or 64 (>= nb).  0 = default
 * (smallest that fits).  64 is the literal "one articulation per wavefront" mapping. */
int pd_model_set_segment_width(pd_model *m, int lanes);
int pd_model_get_segment_width(const pd_model *m);

/* Kernel family of the rollout launches.  0 (default): chosen per launch by batch size -- revolute-only robots with at most 16 bodies
 * (Laikago) run the QUAD-LANE kernels (four lanes per body, one articulation per wavefront: a shorter instruction stream per step,
 * three times the lane-cycles per env) while the batch fits one workgroup per compute unit (<= 4 x CUs envs = 1 024 on MI355X) and the
 * lane-per-body kernels above that; 1: lane per body always; 2: quad-lane wherever the robot is eligible (tests, A/B timing).
 * pd_model_get_kernel_family returns the setting; *eligible (may be NULL) = 1 when the robot has quad-lane kernels at all. */
int pd_model_set_kernel_family(pd_model *m, int family);
int pd_model_get_kernel_family(const pd_model *m, int *eligible);

/* Numeric policy of the rollout launches of this model (ABI v7) -- HOW two expressions of the joint pass are evaluated in fp32, forward
 * and adjoint, both kernel families; the functions are the same, a float64 evaluation of either policy gives the same numbers:
 *   PD_NUM_STABLE  (default)  revolute twist angle q = 2 sign(d) atan2(|axis| |d|, r.w), d = r.xyz . axis; FIXED joint's angular error
 *                  r.xyz * 2 atan2(|r.xyz|, r.w) / |r.xyz| -- accurate to an ulp at joint angle 0 / at the joint's operating point
 *   PD_NUM_LITERAL the reference's text, diffphys/integrator_euler.py:385-400: twist = normalize((axis d, r.w)), q = 2 acos(twist.w)
 *                  sign(axis . twist.xyz); FIXED: normalize(r.xyz) * acos(r.w) * 2; adjoints through acos' (0 at |x| = 1, argument clamped
 *                  to [-1, 1]: the policy of both modes) and the normalisations.  In fp32 acos near 1 turns one ulp of twist.w into 7e-4
 *                  rad of a small joint angle: use it for a side-by-side run against the reference on Warp,
 *                  expect it 1e-3 .. 1 away from a float64 evaluation in long-horizon gradients where PD_NUM_STABLE is 1e-5 .. 1e-3.
 * May be changed between launches; a forward pass and its adjoint must run under the same policy. */
#define PD_NUM_STABLE 0
#define PD_NUM_LITERAL 1
int pd_model_set_numeric_policy(pd_model *m, int policy);
int pd_model_get_numeric_policy(const pd_model *m);

/* Floats of caller-provided workspace that pd_rollout_forward fills and pd_rollout_backward reads:
 * per step the 13-float body state and the 6-float body wrench as five float4 planes [step][plane][bs*nb], followed
 * by the forward sweep's contact hit log (32 ints per env-step) that the adjoint replays.  The base must be 16-byte
 * aligned. */
size_t pd_rollout_workspace_floats(const pd_model *m, int bs, int nsteps);

/* Inspection of the saved trajectory (tests, diagnostics): the hit log behind the planes holds, per (step, env), 32 ints --
 * [0] the number n of contact candidates that touched in that step (eval_body_contacts' `c <= 0`, integrator_euler.py:130-133;
 * -1 when more than 31 did: the adjoint then sweeps again), [1..n] packed entries  point | material << 16 | body << 24  with
 * `point` an index into the DEVICE contact table (candidates grouped by body, spatially ordered inside a body).
 * order_host[i] (HOST, capacity >= nc ints) receives the index into the template's contact_* arrays of device entry i. */
int pd_model_contact_order(const pd_model *m, int *order_host, int capacity);

/* Per-env joint_X_p (the lab4d path rebinds env.joint_X_p from a torch tensor before every rollout,
 * diffphys/dp_interface.py:465): joint_X_p_dev is [n_envs][nb][7] DEVICE memory owned by the caller and read by every
 * later launch on this model until it is re-bound; NULL / 0 returns to the template's joint_X_p.  A rollout then needs
 * bs == n_envs; FK articulation i uses env i % n_envs (the reference runs eval_fk frame by frame on the same n_envs
 * model).  Host-side pointer swap only: no copy, no synchronisation, no rebuild of the device model. */
int pd_model_bind_joint_X_p(pd_model *m, const float *joint_X_p_dev, int n_envs);

/* frame2step_host: [nframes] HOST ints, frame f is state frame2step[f] (ForwardWarp reads self.frame2step,
 * diffphys/dp_model.py:1231-1248).  Validated before anything is launched: each entry in 0..nsteps, no step twice
 * (a violation returns non-zero, nothing is written).  State `nsteps` (the state after the last step) is a legal frame
 * for wp_pos / wp_vel; its grf / jaf rows are zero -- the reference appends no force snapshot for it (:1225-1228).
 * The device-side step->frame table is cached per model and (nsteps, frame2step); the first call with a new key uploads
 * it with one synchronous copy, later calls neither allocate nor synchronise (warm up before capturing a graph).
 * Sizes: bs = 0 and nsteps = 0 are legal (nothing / FK only).  Largest batch of ONE call: bs x bodies < 2^27 and bs < 2^24 (the kernels
 * add 32-bit lane offsets to 64-bit step bases; 10 M Laikago envs) -- a larger bs is refused (non-zero, "batch too large"), never
 * wrapped; the workspace and every tensor may lie past 4 GiB (tests: 1 048 576 envs x 8 steps, 9.8 GB of workspace). */
int pd_rollout_forward(const pd_model *m, int bs, int nsteps, float dt,
                       const float *q_init_dev, const float *qd_init_dev, const float *torques_dev,
                       const float *res_f_dev, const float *refs_dev, const float *target_ke_dev,
                       const float *target_kd_dev, const float *body_inv_mass_dev,
                       const float *body_inertia_dev, const float *body_inv_inertia_dev,
                       int nframes, const int *frame2step_host, float *workspace_dev,
                       float *wp_pos_dev, float *wp_vel_dev, float *grf_dev, float *jaf_dev, void *stream);

/* Gradients are OVERWRITTEN (not accumulated).  g_*_dev mirror the input shapes.  body_mass has no
 * direct gradient (integrator_euler.py:43 loads it, nothing uses it), so there is no g_body_mass.
 * Every stored gradient has been through the boundary's remove_nan (diffphys/dp_model.py:1294-1384 with
 * diffphys/dp_utils.py:43-57, clip=False): NaN -> 0, +-inf kept -- applied by the kernel at its stores (ABI v4; up to v3
 * the caller had to scrub). */
int pd_rollout_backward(const pd_model *m, int bs, int nsteps, float dt,
                        const float *q_init_dev, const float *qd_init_dev, const float *torques_dev,
                        const float *refs_dev, const float *target_ke_dev, const float *target_kd_dev,
                        const float *body_inv_mass_dev, const float *body_inertia_dev,
                        const float *body_inv_inertia_dev, int nframes, const int *frame2step_host,
                        const float *workspace_dev, const float *adj_pos_dev, const float *adj_vel_dev,
                        float *g_q_init_dev, float *g_qd_init_dev, float *g_torques_dev, float *g_res_f_dev,
                        float *g_refs_dev, float *g_target_ke_dev, float *g_target_kd_dev,
                        float *g_body_inv_mass_dev, float *g_body_inertia_dev, float *g_body_inv_inertia_dev,
                        void *stream);

/* The rollout with the trajectory loss evaluated where the frame poses are produced.  Of the reference's
 * loss terms only loss_traj back-propagates through the rollout (diffphys/dp_model.py:777-779; pos_state / vel_state / distill use
 * sim_position.detach(), :795,802):   loss_traj = reduce_loss(se3_loss(sim_position, target_position).mean(-1) with outseq entries
 * zeroed, clip=True)   (se3_loss diffphys/dp_utils.py:113-138, reduce_loss :93-110).
 * pd_rollout_forward_traj_loss = pd_rollout_forward plus, inside the same rollout launch, at every frame state: se3_loss of each body
 * pose against target_pos_dev [bs][F][nb][7] (p, real-last quaternion), its UNSCALED gradients to seed_pos_dev [F][bs*nb][7] (the
 * layout of adj_pos) and seed_gt_dev [bs][F][nb][7] (d / d target; may be NULL), the mean over the env's bodies to
 * loss_table_dev [bs][F] (0 where outseq_dev [bs][F] bytes are non-zero; outseq_dev may be NULL); then one small launch that does
 * reduce_loss on the table: reduced_dev[4] = { loss_traj, the clip threshold (10 x the lower median of the positive entries of
 * ENV 0, as the reference takes it, dp_utils.py:98-100; NaN when env 0 has none, and then -- like the reference -- no env is
 * clipped), the number of positive entries left, the number of clipped envs } and scale_dev [bs][F] =
 * d loss_traj / d table entry (0 for entries the clip / outseq assigned zero).  wp_pos / wp_vel / grf / jaf as pd_rollout_forward.
 * pd_rollout_backward_traj_loss = pd_rollout_backward whose frame seeds are  g_loss_dev[0] * scale[env][frame] / nb * seed_pos
 * (g_loss_dev: DEVICE scalar, the upstream gradient of loss_traj, e.g. traj_wt) PLUS the rows of adj_pos_dev / adj_vel_dev when
 * those are given (both or neither; NULL = no other term reaches the poses); they are built by a small launch into
 * seed_work_dev (F * bs * nb * 13 floats of caller-owned scratch) in front of the adjoint rollout launch.  No pose or seed passes
 * through the host framework between the launches, no se3_loss launch, no reduce_loss ops. */
int pd_rollout_forward_traj_loss(const pd_model *m, int bs, int nsteps, float dt,
                                 const float *q_init_dev, const float *qd_init_dev, const float *torques_dev,
                                 const float *res_f_dev, const float *refs_dev, const float *target_ke_dev,
                                 const float *target_kd_dev, const float *body_inv_mass_dev,
                                 const float *body_inertia_dev, const float *body_inv_inertia_dev,
                                 int nframes, const int *frame2step_host, float *workspace_dev,
                                 float *wp_pos_dev, float *wp_vel_dev, float *grf_dev, float *jaf_dev,
                                 const float *target_pos_dev, const unsigned char *outseq_dev, float rot_ratio,
                                 float *seed_pos_dev, float *seed_gt_dev, float *loss_table_dev, float *reduced_dev,
                                 float *scale_dev, void *stream);
int pd_rollout_backward_traj_loss(const pd_model *m, int bs, int nsteps, float dt,
                                  const float *q_init_dev, const float *qd_init_dev, const float *torques_dev,
                                  const float *refs_dev, const float *target_ke_dev, const float *target_kd_dev,
                                  const float *body_inv_mass_dev, const float *body_inertia_dev,
                                  const float *body_inv_inertia_dev, int nframes, const int *frame2step_host,
                                  const float *workspace_dev, const float *adj_pos_dev, const float *adj_vel_dev,
                                  const float *seed_pos_dev, const float *scale_dev, const float *g_loss_dev,
                                  float *seed_work_dev, float *g_q_init_dev, float *g_qd_init_dev, float *g_torques_dev, float *g_res_f_dev,
                                  float *g_refs_dev, float *g_target_ke_dev, float *g_target_kd_dev,
                                  float *g_body_inv_mass_dev, float *g_body_inertia_dev, float *g_body_inv_inertia_dev,
                                  void *stream);

/* Row f4, second half: the FK of the control reference (ForwardKinematics.apply(queried_q, queried_qd, env), diffphys/dp_model.py:758
 * of the reference: F x bs chains per iteration, independent of the rollout) without a launch of its own.  The two entries below are
 * the *_traj_loss entries above with one more argument; with fk != NULL and fk->n > 0
 *   forward : the small launch that follows the rollout launch runs reduce_loss in its first workgroup and FK forward in the others;
 *   backward: the small launch in front of the adjoint rollout builds the seeds in its first workgroups and runs FK backward
 *             (with ForwardKinematics.backward's post-processing, see pd_fk_backward) in the others.
 * The articulations come FRAME-major, joint_q [F][bs][nq] / joint_qd [F][bs][nqd] (n = F * bs; F need not be the rollout's nframes) --
 * the layout of ForwardKinematics' rj_q / rj_qd -- and the body rows are ENV-major, [bs][F][nb][7] / [bs][F][nb][6] (what the
 * reference gets from permute(1, 0, 2, 3) at dp_model.py:1093-1094, and what its losses consume): no permuted copy on either side.
 * Forward reads joint_q / joint_qd, writes body_q / body_qd; backward reads joint_q / joint_qd / adj_body_q / adj_body_qd, writes
 * g_joint_q / g_joint_qd; the other fields may be NULL.  fk == NULL or fk->n == 0: exactly the *_traj_loss entries. */
typedef struct pd_fk_ride {
  int n, bs;
  const float *joint_q_dev, *joint_qd_dev;
  float *body_q_dev, *body_qd_dev;
  const float *adj_body_q_dev, *adj_body_qd_dev;
  float *g_joint_q_dev, *g_joint_qd_dev;
} pd_fk_ride;
int pd_rollout_forward_traj_loss_fk(const pd_model *m, int bs, int nsteps, float dt,
                                    const float *q_init_dev, const float *qd_init_dev, const float *torques_dev,
                                    const float *res_f_dev, const float *refs_dev, const float *target_ke_dev,
                                    const float *target_kd_dev, const float *body_inv_mass_dev,
                                    const float *body_inertia_dev, const float *body_inv_inertia_dev,
                                    int nframes, const int *frame2step_host, float *workspace_dev,
                                    float *wp_pos_dev, float *wp_vel_dev, float *grf_dev, float *jaf_dev,
                                    const float *target_pos_dev, const unsigned char *outseq_dev, float rot_ratio,
                                    float *seed_pos_dev, float *seed_gt_dev, float *loss_table_dev, float *reduced_dev,
                                    float *scale_dev, const pd_fk_ride *fk, void *stream);
int pd_rollout_backward_traj_loss_fk(const pd_model *m, int bs, int nsteps, float dt,
                                     const float *q_init_dev, const float *qd_init_dev, const float *torques_dev,
                                     const float *refs_dev, const float *target_ke_dev, const float *target_kd_dev,
                                     const float *body_inv_mass_dev, const float *body_inertia_dev,
                                     const float *body_inv_inertia_dev, int nframes, const int *frame2step_host,
                                     const float *workspace_dev, const float *adj_pos_dev, const float *adj_vel_dev,
                                     const float *seed_pos_dev, const float *scale_dev, const float *g_loss_dev,
                                     float *seed_work_dev, float *g_q_init_dev, float *g_qd_init_dev, float *g_torques_dev, float *g_res_f_dev,
                                     float *g_refs_dev, float *g_target_ke_dev, float *g_target_kd_dev,
                                     float *g_body_inv_mass_dev, float *g_body_inertia_dev, float *g_body_inv_inertia_dev,
                                     const pd_fk_ride *fk, void *stream);

/* n independent articulations: joint_q [n][nq], joint_qd [n][nqd] -> body_q [n][nb][7], body_qd [n][nb][6] */
int pd_fk_forward(const pd_model *m, int n, const float *joint_q_dev, const float *joint_qd_dev,
                  float *body_q_dev, float *body_qd_dev, void *stream);
/* The gradients come out with ForwardKinematics.backward's post-processing (diffphys/dp_model.py:1109-1110, 1122-1123):
 * NaN -> 0, then values above 1 -> 1 (an upper clamp only) -- applied by the kernel at its stores (ABI v4). */
int pd_fk_backward(const pd_model *m, int n, const float *joint_q_dev, const float *joint_qd_dev,
                   const float *adj_body_q_dev, const float *adj_body_qd_dev,
                   float *g_joint_q_dev, float *g_joint_qd_dev, void *stream);

/* Fused per-frame pose loss and its gradients (replaces the torch composition of
 * se3_loss, diffphys/dp_utils.py:113-138):  n elements of `dim` floats, dim = 7 (p, real-last quaternion) or 6
 * (p, axis-angle).  loss[n]; g_pred / g_gt [n][dim] = d loss / d input (either may be NULL).  Entries with a NaN input
 * give loss 0 and gradients 0 x (the raw gradient), i.e. NaN where the local derivative is not finite and 0 elsewhere -- what autograd
 * makes of the reference's `loss[nanid] = 0` (dp_utils.py:137): the NaN reaches the adjoint rollout as a seed there too and is scrubbed
 * at that boundary (remove_nan), which drops the env's whole gradient. */
int pd_se3_loss(int n, int dim, const float *pred_dev, const float *gt_dev, float rot_ratio, float *loss_dev,
                float *g_pred_dev, float *g_gt_dev, void *stream);

/* reduce_loss of the reference (diffphys/dp_utils.py:93-110) on any [bs][nframes] device table (ABI v7) -- the same one-workgroup code
 * the trajectory-loss entries run after the rollout, callable on its own (the reference also reduces its other loss terms with it,
 * dp_model.py:797-809, clip = 0 there).  clip != 0: the threshold is 10 x the lower median (torch.median) of env 0's positive entries,
 * taken once and used for every env; each env's entries from the first one above it on are ASSIGNED zero IN table_dev (the reference
 * truncates its argument in place); env 0 without a positive entry: the threshold is NaN and nothing is clipped (what the reference
 * does on torch >= 1.8: tests/golden/ref_host_reduce_loss.npz holds its outputs).  The value is the mean of the positive entries
 * left when the sum of all entries is positive, else the mean of all entries (NaN entries make that NaN, as in the reference).
 * reduced_dev[4] = { value, threshold, positive entries left, clipped envs }; scale_dev [bs][nframes] (may be NULL) = d value /
 * d entry (0 for assigned entries).  An empty table gives 0. */
int pd_reduce_loss(int bs, int nframes, float *table_dev, int clip, float *reduced_dev, float *scale_dev, void *stream);

/* SE(3) pose algebra of the loss plumbing, one launch per op and one per vector-Jacobian
 * product; replaces the reference's compositions of dqtorch kernels and torch ops:
 *   PD_POSE_COMPOSE_DELTA  a = target pose [n][7] (p, real-last quaternion), b = delta [n][6] (p, axis-angle) -> out [n][7]
 *                          = se3_mat2vec(se3_vec2mat(delta) @ se3_vec2mat(target))        diffphys/dp_utils.py:22-31
 *   PD_POSE_ROTATE_FRAME   a = global pose, b = target pose [n][7] -> out [n][7] = T_global @ T_target   dp_utils.py:60-73
 *   PD_POSE_ROTATE_VEL     a = global pose, b = (linear, angular) [n][6] -> out [n][6], both halves rotated by the
 *                          rotation of the global pose                                    dp_utils.py:76-84
 * with se3_vec2mat / se3_mat2vec / quaternion <-> matrix as in diffphys/geom_utils.py:148-203 (quaternions are divided
 * by |q|^2, the best-conditioned of the four matrix->quaternion forms is taken).  a_broadcast != 0: `a` is ONE 7-vector
 * shared by all n elements.  The VJP writes g_a [n][7] (per element also when `a` is broadcast: the caller sums) and
 * g_b [n][6 or 7]; either may be NULL. */
enum { PD_POSE_COMPOSE_DELTA = 0, PD_POSE_ROTATE_FRAME = 1, PD_POSE_ROTATE_VEL = 2 };
int pd_pose_op(int op, int n, const float *a_dev, int a_broadcast, const float *b_dev, float *out_dev, void *stream);
int pd_pose_op_vjp(int op, int n, const float *a_dev, int a_broadcast, const float *b_dev, const float *g_out_dev,
                   float *g_a_dev, float *g_b_dev, void *stream);

/* Lowest ground-contact candidate of each of n pose sets (get_foot_height / the reg_foot term, diffphys/dp_model.py:
 * 574-579,762,814, evaluated on the contact candidates instead of posed visual meshes):  height[i] = min over candidates c of
 * body_q[i][c_body[c]].p.y + (R(body_q[i][c_body[c]].q) c_point[c]).y - c_dist[c];  arg[i] = the minimising candidate
 * (lowest index on ties).  body_q [n][nb][7].  The VJP writes g_body_q [n][nb][7] (zero except the arg-min body). */
int pd_foot_height(int n, int nb, int nc, const float *body_q_dev, const int *c_body_dev, const float *c_point_dev,
                   const float *c_dist_dev, float *height_dev, int *arg_dev, void *stream);
int pd_foot_height_vjp(int n, int nb, const float *body_q_dev, const int *c_body_dev, const float *c_point_dev,
                       const int *arg_dev, const float *g_height_dev, float *g_body_q_dev, void *stream);

/* out[c] = sum over the n rows of x[n][k] (row-major), rows added in a fixed order: the bias gradient of the time-MLPs' linear layers
 * (reference: torch's autograd of nn.Linear in diffphys/lab4d_utils.py BaseMLP) and the gradient of an operand that was broadcast over
 * n poses (global_q in rotate_frame, diffphys/dp_utils.py:113-138) -- as one launch that gives the same bits eagerly and in a replayed
 * HIP graph.  n = 0 writes zeros.  ws_dev: PD_COLSUM_SLICES * k floats of caller-owned scratch (row slices are summed there first, then the
 * slices in order); NULL, or n <= 1024: one slice, one launch -- the result of a given (n, k, ws given or not) never depends on anything else. */
#define PD_COLSUM_SLICES 32
int pd_colsum(int n, int k, const float *x_dev, float *out_dev, float *ws_dev, void *stream);

/* Weight and bias gradient of a linear layer y = x W^T + b over n samples, on the fp32 matrix cores (csrc/pd_mlp.hip):
 *   gw[m][kin] = sum_s g[s][m] x[s][kin],   gb[m] = sum_s g[s][m]   (gb_dev may be NULL)
 * g_dev [n][m] = dL/dy, x_dev [n][kin], row-major, contiguous.  Fixed summation order: the same bits on every call, eagerly and in a replayed
 * HIP graph.  Reference: torch autograd of nn.Linear in the time-MLPs (diffphys/lab4d_utils.py BaseMLP / TimeMLP; torch_utils.py:116-180) at
 * the training window of main.py:86 (n = 7 600).  Shapes: m and kin multiples of 128; pd_linear_wgrad_workspace_floats returns 0 for any other
 * shape (the caller keeps its BLAS path) and otherwise the floats of caller-owned scratch ws_dev needs. */
size_t pd_linear_wgrad_workspace_floats(int n, int m, int kin);
int pd_linear_wgrad(int n, int m, int kin, const float *g_dev, const float *x_dev, float *gw_dev, float *gb_dev, float *ws_dev, void *stream);

/* Device time (ms) of this model's last `kind` launch, measured with hipEvents recorded on the launch stream around
 * the kernel: kind 0 = rollout forward, 1 = rollout backward.  Enabled per model by pd_model_set_timing(m, 1); used by
 * bench.py.  Returns a negative value when nothing was timed.  (Synchronises on the end event.) */
int pd_model_set_timing(pd_model *m, int on);
float pd_last_kernel_ms(const pd_model *m, int kind);

/* Launch geometry of the last rollout launch of `kind` on this model (bench.py reports it beside the roofline):
 * out[0] workgroups, out[1] threads per workgroup, out[2] dynamic LDS bytes per workgroup, out[3] envs per workgroup. */
int pd_last_launch_info(const pd_model *m, int kind, int out[4]);

#ifdef __cplusplus
}
#endif
#endif
