/*
 * ORACLE (test infrastructure, not product code): plain-C restatement of the
 * reference's hot path with a hand-written reverse sweep, compiled once with
 * REAL=float (the fp32 twin and the timed CPU baseline) and once with
 * REAL=double.
 *
 * PARITY UNPINNED: warp_lang==0.7.2 (requirements.txt:12 of the reference) is
 * not vendored and not installable here, and the reference ships no tests or
 * golden vectors.  This file follows, statement by statement,
 *   /root/reference/diffphys/integrator_euler.py:21-91    integrate_bodies
 *   /root/reference/diffphys/integrator_euler.py:93-179   eval_body_contacts
 *   /root/reference/diffphys/integrator_euler.py:234-286  quat_twist, quat_decompose, eval_joint_force
 *   /root/reference/diffphys/integrator_euler.py:289-451  eval_body_joints
 *   /root/reference/diffphys/integrator_euler.py:491-620  compute_forces / simulate
 *   /root/reference/diffphys/dp_model.py:1133-1249        wp_add, ForwardWarp.forward
 *   /root/reference/diffphys/dp_model.py:1251-1400        ForwardWarp.backward (what the tape returns)
 *   warp.sim.articulation.eval_fk                         SURVEY.md Appendix A.3 (recall)
 * and the Warp built-ins / adjoint rules of SURVEY.md Appendix A.1.  It keeps
 * the reference's algorithm: every step's state and body_f are stored, each
 * step is four passes (res_f add, contacts, joints, integrate), the reverse
 * sweep replays the adjoint of each pass in reverse order.  Envs are
 * independent, so the outer loop is over envs (OpenMP) -- results are identical
 * to pass-major order.
 *
 * It is validated against oracle/ref_torch.py (float64 autograd), see
 * tests/test_oracle_c_vs_torch.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load the library built from this file.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#if defined(REAL_IS_DOUBLE)
#define R_SQRT sqrt
#define R_ACOS acos
#define R_ASIN asin
#define R_ATAN2 atan2
#define R_SIN sin
#define R_COS cos
#else
#define R_SQRT sqrtf
#define R_ACOS acosf
#define R_ASIN asinf
#define R_ATAN2 atan2f
#define R_SIN sinf
#define R_COS cosf
#endif

enum { JOINT_PRISMATIC = 0, JOINT_REVOLUTE = 1, JOINT_BALL = 2, JOINT_FIXED = 3, JOINT_FREE = 4, JOINT_COMPOUND = 5 };

typedef struct { real x, y, z; } v3;
typedef struct { real x, y, z, w; } qt;

typedef struct {
  int nb, nq, nqd, nc, nmat;
  int *joint_type, *joint_parent, *q_start, *qd_start;
  real *X_p, *X_c, *axis, *com;
  real *limit_lower, *limit_upper, *limit_ke, *limit_kd;
  int *c_body, *c_mat;
  real *c_point, *c_dist, *materials;
  real gravity[3];
  real attach_ke, attach_kd;
} RefTemplate;

/* ------------------------------------------------------------------ vec / quat */
static inline v3 V(real x, real y, real z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vscale(v3 a, real s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline real vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline real vlen(v3 a) { return R_SQRT(vdot(a, a)); }
static inline v3 vnormalize(v3 a) { real l = vlen(a); return l > (real)0 ? vscale(a, (real)1 / l) : V(0, 0, 0); }
static inline void vacc(v3 *a, v3 b) { a->x += b.x; a->y += b.y; a->z += b.z; }
static inline qt Q(real x, real y, real z, real w) { qt r = {x, y, z, w}; return r; }
static inline v3 qv(qt q) { return V(q.x, q.y, q.z); }
static inline qt qadd(qt a, qt b) { return Q(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline qt qscale(qt a, real s) { return Q(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline void qacc(qt *a, qt b) { a->x += b.x; a->y += b.y; a->z += b.z; a->w += b.w; }
static inline qt qconj(qt q) { return Q(-q.x, -q.y, -q.z, q.w); }
static inline qt qmul(qt a, qt b) {
  return Q(a.w * b.x + b.w * a.x + a.y * b.z - a.z * b.y, a.w * b.y + b.w * a.y + a.z * b.x - a.x * b.z,
           a.w * b.z + b.w * a.z + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
static inline real qdot(qt a, qt b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline qt qnormalize(qt q) { real l = R_SQRT(qdot(q, q)); return qscale(q, (real)1 / l); }
static inline v3 qrot(qt q, v3 v) {
  v3 u = qv(q); real w = q.w;
  return vadd(vadd(vscale(v, (real)2 * w * w - (real)1), vscale(vcross(u, v), (real)2 * w)), vscale(u, (real)2 * vdot(u, v)));
}
static inline v3 qrot_inv(qt q, v3 v) {
  v3 u = qv(q); real w = q.w;
  return vadd(vsub(vscale(v, (real)2 * w * w - (real)1), vscale(vcross(u, v), (real)2 * w)), vscale(u, (real)2 * vdot(u, v)));
}
static inline qt q_axis_angle(v3 axis, real ang) {
  real s = R_SIN(ang * (real)0.5), c = R_COS(ang * (real)0.5);
  return Q(axis.x * s, axis.y * s, axis.z * s, c);
}
/* acos / asin POLICY -- a NAMED DEVIATION from SURVEY.md Appendix A.1, whose recall of warp 0.7.2 is "no guards:
 * acos(x > 1) = NaN and the adjoint -1/sqrt(1-x^2) = -inf at |x| = 1".  This build's recall of the same builtins is the
 * opposite (argument clamped to [-1, 1], adjoint contribution dropped where sqrt(1-x^2) is not > 0), and the product
 * kernels follow it (pd_math.h).  Neither recall can be checked here (Warp is absent), so the oracle runs BOTH:
 *   ref_set_acos_policy(0)  guarded (default; what the HIP kernels do)
 *   ref_set_acos_policy(1)  unguarded, as SURVEY App. A.1 recalls Warp
 * and tests/test_oracle_known_answers.py measures where they differ: identical on inputs away from |x| = 1, NaN
 * gradients (scrubbed to 0 by remove_nan at the boundary) for an env whose joint passes through angle 0 in fp32. */
static int g_acos_unguarded = 0;
void ref_set_acos_policy(int unguarded) { g_acos_unguarded = unguarded; }
/* Evaluation of the revolute twist angle (integrator_euler.py:394-400).  0 (default): literally -- normalize, q = 2 acos(twist.w) sign(..),
 * adjoint through acos' and the normalisation.  1: the SAME function as q = 2 sign(d) atan2(|axis| |d|, r.w), d = r.xyz . axis, with its
 * partials 2 |axis| r.w / (r.w^2 + y^2) and -2 |axis| d / (..): no cancellation near angle 0 (one ulp of twist.w is 7e-4 rad there).
 * In float64 the two agree to rounding; in fp32 the second is what the HIP kernels evaluate (pd_math.h twist_angle), so the fp32
 * build with this switch on is the like-for-like "plain fp32 evaluation" the tests measure the kernels' arithmetic against. */
static int g_twist_atan2 = 0;
void ref_set_twist_eval(int use_atan2) { g_twist_atan2 = use_atan2; }
static inline real twist_angle_atan2(v3 axis, real da, real w, real *dq_dda, real *dq_dw) {
  const real L = vlen(axis), y = L * (da < (real)0 ? -da : da), den = w * w + y * y;
  const real sgn = da < (real)0 ? (real)-1 : (real)1;
  if (dq_dda) { *dq_dda = den > (real)0 ? (real)2 * L * w / den : (real)0; *dq_dw = den > (real)0 ? -(real)2 * L * da / den : (real)0; }
  return (real)2 * sgn * R_ATAN2(y, w);
}
/* The FIXED joint's angular error (integrator_euler.py:385-390), normalize(r.xyz) * 2 acos(r.w).  Literally (switch 0, default) it is
 * ill-posed for the quaternions a simulation holds: r = conj(q_p) q_c of fp32-normalised q has |r| = 1 + O(1e-7), and at the joint's operating
 * point (angle error ~ 0) acos(r.w) turns that norm error into +-9e-4 rad of "angle" (or clamps it to 0) -- in float64 as much as in fp32.
 * With ref_set_twist_eval(1) it is evaluated as the same function of a unit quaternion in its scale-invariant form
 *     ang_err = v h,   h = 2 atan2(|v|, w) / |v|   (series of atan(x) / x, x = |v| / w, below x = 1e-2),
 * with the partials  d ang_err / d v = h I + (h_s / s) v v^T,  d ang_err / d w = v h_w,  h_w = -2 / (|v|^2 + w^2)  -- what the HIP kernels
 * evaluate (pd_math.h fixed_ang_h).  Returns h; hs_over_s / h_w may be NULL. */
static inline real fixed_ang_h(v3 v, real w, real *hs_over_s, real *h_w) {
  const real s2 = vdot(v, v), den = s2 + w * w;
  real h = (real)0, hss = (real)0;
  if (w > (real)0 && s2 < (real)1e-4 * w * w) {
    const real x2 = s2 / (w * w);
    const real u = (real)1 - x2 * ((real)1 / (real)3 - x2 * ((real)1 / (real)5 - x2 / (real)7));
    const real upx = -(real)2 / (real)3 + x2 * ((real)4 / (real)5 - x2 * (real)6 / (real)7);   /* u'(x) / x */
    h = (real)2 * u / w; hss = (real)2 * upx / (w * w * w);
  } else if (s2 > (real)0) {
    const real sl = R_SQRT(s2), phi = R_ATAN2(sl, w);
    h = (real)2 * phi / sl; hss = (real)2 * (w / den - phi / sl) / s2;
  }
  if (hs_over_s) { *hs_over_s = hss; *h_w = den > (real)0 ? -(real)2 / den : (real)0; }
  return h;
}
/* Conditioning probe (tests): with this on, ref_rollout_forward rounds every state it stores to fp32 (value kept in `real`).  In
 * the float64 build that is the LEAST any fp32 implementation does to a rollout -- one rounding per state component and step,
 * exact arithmetic otherwise -- so how far the gradients of an env move under it measures how ill-conditioned the env is.
 * 2 = the same with each value moved to an adjacent fp32 number (alternating sides): a second sample of the same perturbation. */
static int g_round_states = 0;
void ref_set_state_rounding(int on) { g_round_states = on; }
int ref_get_acos_policy(void) { return g_acos_unguarded; }
static inline real inv_sqrt_1mx2(real x) {
  real d = R_SQRT((real)1 - x * x);
  if (g_acos_unguarded) return (real)1 / d;
  return d > (real)0 ? (real)1 / d : (real)0;
}
static inline real clampr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline real acos_c(real x) { return g_acos_unguarded ? R_ACOS(x) : R_ACOS(clampr(x, (real)-1, (real)1)); }
static inline real asin_c(real x) { return g_acos_unguarded ? R_ASIN(x) : R_ASIN(clampr(x, (real)-1, (real)1)); }
static inline real clamp_pass(real x, real lo, real hi) { return (x < lo || x > hi) ? (real)0 : (real)1; }

/* adjoints (accumulate into adj_* like Warp's generated code) */
static inline void adj_vcross(v3 a, v3 b, v3 *adj_a, v3 *adj_b, v3 g) {
  if (adj_a) vacc(adj_a, vcross(b, g));
  if (adj_b) vacc(adj_b, vcross(g, a));
}
static inline void adj_qmul(qt a, qt b, qt *adj_a, qt *adj_b, qt g) {
  if (adj_a) qacc(adj_a, qmul(g, qconj(b)));
  if (adj_b) qacc(adj_b, qmul(qconj(a), g));
}
static inline void adj_qrot(qt q, v3 v, qt *adj_q, v3 *adj_v, v3 g) {
  v3 u = qv(q); real w = q.w;
  if (adj_v) vacc(adj_v, qrot_inv(q, g));
  if (adj_q) {
    real uv = vdot(u, v), ug = vdot(u, g);
    v3 au = vadd(vscale(vcross(v, g), (real)2 * w), vscale(vadd(vscale(g, uv), vscale(v, ug)), (real)2));
    adj_q->x += au.x; adj_q->y += au.y; adj_q->z += au.z;
    adj_q->w += (real)4 * w * vdot(v, g) + (real)2 * vdot(vcross(u, v), g);
  }
}
static inline void adj_qrot_inv(qt q, v3 v, qt *adj_q, v3 *adj_v, v3 g) {
  v3 u = qv(q); real w = q.w;
  if (adj_v) vacc(adj_v, qrot(q, g));
  if (adj_q) {
    real uv = vdot(u, v), ug = vdot(u, g);
    v3 au = vadd(vscale(vcross(v, g), -(real)2 * w), vscale(vadd(vscale(g, uv), vscale(v, ug)), (real)2));
    adj_q->x += au.x; adj_q->y += au.y; adj_q->z += au.z;
    adj_q->w += (real)4 * w * vdot(v, g) - (real)2 * vdot(vcross(u, v), g);
  }
}
static inline void adj_qnormalize(qt q, qt *adj_q, qt g) {
  real l = R_SQRT(qdot(q, q)); real il = (real)1 / l; qt n = qscale(q, il);
  real ng = qdot(n, g);
  qacc(adj_q, qscale(qadd(g, qscale(n, -ng)), il));
}
static inline void adj_vnormalize(v3 a, v3 *adj_a, v3 g) {
  real l = vlen(a);
  if (l > (real)0) { real il = (real)1 / l; v3 n = vscale(a, il); vacc(adj_a, vscale(vsub(g, vscale(n, vdot(n, g))), il)); }
}
static inline void adj_vlen(v3 a, v3 *adj_a, real g) { vacc(adj_a, vscale(vnormalize(a), g)); }
/* q_axis_angle(axis, ang): adjoint to axis and angle */
static inline void adj_q_axis_angle(v3 axis, real ang, v3 *adj_axis, real *adj_ang, qt g) {
  real s = R_SIN(ang * (real)0.5), c = R_COS(ang * (real)0.5);
  v3 gv = qv(g);
  if (adj_axis) vacc(adj_axis, vscale(gv, s));
  if (adj_ang) *adj_ang += (real)0.5 * (c * vdot(axis, gv) - s * g.w);
}

static inline v3 ld3(const real *p) { return V(p[0], p[1], p[2]); }
static inline qt ld4(const real *p) { return Q(p[0], p[1], p[2], p[3]); }
static inline void st3(real *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline void st4(real *p, qt a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w; }
static inline void add3(real *p, v3 a) { p[0] += a.x; p[1] += a.y; p[2] += a.z; }
static inline void add4(real *p, qt a) { p[0] += a.x; p[1] += a.y; p[2] += a.z; p[3] += a.w; }
static inline void mat_vec(const real *M, v3 a, v3 *o) {
  *o = V(M[0] * a.x + M[1] * a.y + M[2] * a.z, M[3] * a.x + M[4] * a.y + M[5] * a.z, M[6] * a.x + M[7] * a.y + M[8] * a.z);
}
static inline v3 matT_vec(const real *M, v3 a) {
  return V(M[0] * a.x + M[3] * a.y + M[6] * a.z, M[1] * a.x + M[4] * a.y + M[7] * a.z, M[2] * a.x + M[5] * a.y + M[8] * a.z);
}
static inline void add_outer(real *M, v3 a, v3 b) {
  M[0] += a.x * b.x; M[1] += a.x * b.y; M[2] += a.x * b.z;
  M[3] += a.y * b.x; M[4] += a.y * b.y; M[5] += a.y * b.z;
  M[6] += a.z * b.x; M[7] += a.z * b.y; M[8] += a.z * b.z;
}

/* ------------------------------------------------------------------ template */
static void *dup_mem(const void *src, size_t n) { void *p = malloc(n ? n : 1); if (n) memcpy(p, src, n); return p; }

RefTemplate *ref_template_create(int nb, int nq, int nqd, int nc, int nmat, const int *joint_type, const int *joint_parent,
                                 const int *q_start, const int *qd_start, const real *X_p, const real *X_c, const real *axis,
                                 const real *com, const real *limit_lower, const real *limit_upper, const real *limit_ke,
                                 const real *limit_kd, const int *c_body, const real *c_point, const real *c_dist,
                                 const int *c_mat, const real *materials, const real *gravity, real attach_ke, real attach_kd) {
  RefTemplate *t = (RefTemplate *)calloc(1, sizeof(RefTemplate));
  t->nb = nb; t->nq = nq; t->nqd = nqd; t->nc = nc; t->nmat = nmat;
  t->joint_type = dup_mem(joint_type, sizeof(int) * nb);
  t->joint_parent = dup_mem(joint_parent, sizeof(int) * nb);
  t->q_start = dup_mem(q_start, sizeof(int) * nb);
  t->qd_start = dup_mem(qd_start, sizeof(int) * nb);
  t->X_p = dup_mem(X_p, sizeof(real) * nb * 7);
  t->X_c = dup_mem(X_c, sizeof(real) * nb * 7);
  t->axis = dup_mem(axis, sizeof(real) * nb * 3);
  t->com = dup_mem(com, sizeof(real) * nb * 3);
  t->limit_lower = dup_mem(limit_lower, sizeof(real) * nqd);
  t->limit_upper = dup_mem(limit_upper, sizeof(real) * nqd);
  t->limit_ke = dup_mem(limit_ke, sizeof(real) * nqd);
  t->limit_kd = dup_mem(limit_kd, sizeof(real) * nqd);
  t->c_body = dup_mem(c_body, sizeof(int) * nc);
  t->c_mat = dup_mem(c_mat, sizeof(int) * nc);
  t->c_point = dup_mem(c_point, sizeof(real) * nc * 3);
  t->c_dist = dup_mem(c_dist, sizeof(real) * nc);
  t->materials = dup_mem(materials, sizeof(real) * nmat * 4);
  memcpy(t->gravity, gravity, sizeof(real) * 3);
  t->attach_ke = attach_ke; t->attach_kd = attach_kd;
  return t;
}
void ref_template_destroy(RefTemplate *t) {
  if (!t) return;
  free(t->joint_type); free(t->joint_parent); free(t->q_start); free(t->qd_start); free(t->X_p); free(t->X_c);
  free(t->axis); free(t->com); free(t->limit_lower); free(t->limit_upper); free(t->limit_ke); free(t->limit_kd);
  free(t->c_body); free(t->c_mat); free(t->c_point); free(t->c_dist); free(t->materials); free(t);
}
int ref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
int ref_sizeof_real(void) { return (int)sizeof(real); }

/* ------------------------------------------------------------------------ FK
 * One articulation: joint_q [nq], joint_qd [nqd] -> body_q [nb][7], body_qd [nb][6]. */
static void compound_axes(qt q_off, const real *jq, v3 *a0, v3 *a1, v3 *a2, qt *q0, qt *q1, qt *q2) {
  *a0 = qrot(q_off, V(1, 0, 0));
  *q0 = q_axis_angle(*a0, jq[0]);
  *a1 = qrot(qmul(*q0, q_off), V(0, 1, 0));
  *q1 = q_axis_angle(*a1, jq[1]);
  *a2 = qrot(qmul(*q1, qmul(*q0, q_off)), V(0, 0, 1));
  *q2 = q_axis_angle(*a2, jq[2]);
}

static void fk_one(const RefTemplate *t, const real *jq, const real *jqd, real *body_q, real *body_qd) {
  for (int i = 0; i < t->nb; ++i) {
    int par = t->joint_parent[i], ty = t->joint_type[i];
    v3 p_wp = V(0, 0, 0); qt q_wp = Q(0, 0, 0, 1); v3 w_wp = V(0, 0, 0), v_wp = V(0, 0, 0);
    if (par >= 0) {
      p_wp = ld3(body_q + par * 7); q_wp = ld4(body_q + par * 7 + 3);
      w_wp = ld3(body_qd + par * 6); v_wp = ld3(body_qd + par * 6 + 3);
    }
    const real *q = jq + t->q_start[i], *qd = jqd + t->qd_start[i];
    v3 axis = ld3(t->axis + i * 3);
    v3 p_jc = V(0, 0, 0); qt q_jc = Q(0, 0, 0, 1); v3 w_jc = V(0, 0, 0), v_jc = V(0, 0, 0);
    if (ty == JOINT_REVOLUTE) {
      q_jc = q_axis_angle(axis, q[0]); w_jc = vscale(axis, qd[0]);
    } else if (ty == JOINT_FREE) {
      p_jc = ld3(q); q_jc = ld4(q + 3); w_jc = ld3(qd); v_jc = ld3(qd + 3);
    } else if (ty == JOINT_COMPOUND) {
      v3 a0, a1, a2; qt q0, q1, q2;
      compound_axes(ld4(t->X_c + i * 7 + 3), q, &a0, &a1, &a2, &q0, &q1, &q2);
      q_jc = qmul(q2, qmul(q1, q0));
      w_jc = vadd(vadd(vscale(a0, qd[0]), vscale(a1, qd[1])), vscale(a2, qd[2]));
    } /* FIXED: identity */
    v3 p_pj = ld3(t->X_p + i * 7); qt q_pj = ld4(t->X_p + i * 7 + 3);
    v3 p_wj = vadd(p_wp, qrot(q_wp, p_pj)); qt q_wj = qmul(q_wp, q_pj);
    v3 p_wc = vadd(p_wj, qrot(q_wj, p_jc)); qt q_wc = qmul(q_wj, q_jc);
    v3 ang = qrot(q_wj, w_jc), lin = qrot(q_wj, v_jc);
    v3 com = ld3(t->com + i * 3);
    st3(body_q + i * 7, p_wc); st4(body_q + i * 7 + 3, q_wc);
    st3(body_qd + i * 6, vadd(w_wp, ang));
    st3(body_qd + i * 6 + 3, vadd(v_wp, vadd(lin, vcross(ang, com))));
  }
}

/* adjoint of fk_one: adj_body_q/adj_body_qd are consumed (modified in place: parents receive
 * their children's contributions), adj_jq / adj_jqd are accumulated. */
static void fk_one_adj(const RefTemplate *t, const real *jq, const real *jqd, const real *body_q, real *adj_body_q,
                       real *adj_body_qd, real *adj_jq, real *adj_jqd) {
  for (int i = t->nb - 1; i >= 0; --i) {
    int par = t->joint_parent[i], ty = t->joint_type[i];
    v3 p_wp = V(0, 0, 0); qt q_wp = Q(0, 0, 0, 1);
    if (par >= 0) { p_wp = ld3(body_q + par * 7); q_wp = ld4(body_q + par * 7 + 3); }
    (void)p_wp;
    const real *q = jq + t->q_start[i], *qd = jqd + t->qd_start[i];
    real *aq = adj_jq + t->q_start[i], *aqd = adj_jqd + t->qd_start[i];
    v3 axis = ld3(t->axis + i * 3);
    v3 com = ld3(t->com + i * 3);
    v3 p_pj = ld3(t->X_p + i * 7); qt q_pj = ld4(t->X_p + i * 7 + 3);
    qt q_off = ld4(t->X_c + i * 7 + 3);
    /* recompute forward locals */
    v3 p_jc = V(0, 0, 0); qt q_jc = Q(0, 0, 0, 1); v3 w_jc = V(0, 0, 0), v_jc = V(0, 0, 0);
    v3 a0 = V(0, 0, 0), a1 = a0, a2 = a0; qt q0 = Q(0, 0, 0, 1), q1 = q0, q2 = q0;
    if (ty == JOINT_REVOLUTE) { q_jc = q_axis_angle(axis, q[0]); w_jc = vscale(axis, qd[0]); }
    else if (ty == JOINT_FREE) { p_jc = ld3(q); q_jc = ld4(q + 3); w_jc = ld3(qd); v_jc = ld3(qd + 3); }
    else if (ty == JOINT_COMPOUND) {
      compound_axes(q_off, q, &a0, &a1, &a2, &q0, &q1, &q2);
      q_jc = qmul(q2, qmul(q1, q0));
      w_jc = vadd(vadd(vscale(a0, qd[0]), vscale(a1, qd[1])), vscale(a2, qd[2]));
    }
    qt q_wj = qmul(q_wp, q_pj);
    v3 ang = qrot(q_wj, w_jc);
    /* incoming adjoints */
    v3 g_p = ld3(adj_body_q + i * 7); qt g_q = ld4(adj_body_q + i * 7 + 3);
    v3 g_w = ld3(adj_body_qd + i * 6), g_v = ld3(adj_body_qd + i * 6 + 3);
    /* body_qd = v_wp + (ang, lin + ang x com) */
    v3 adj_w_wp = g_w, adj_v_wp = g_v;
    v3 adj_ang = g_w, adj_lin = g_v;
    adj_vcross(ang, com, &adj_ang, NULL, g_v);
    qt adj_q_wj = Q(0, 0, 0, 0); v3 adj_w_jc = V(0, 0, 0), adj_v_jc = V(0, 0, 0);
    adj_qrot(q_wj, v_jc, &adj_q_wj, &adj_v_jc, adj_lin);
    adj_qrot(q_wj, w_jc, &adj_q_wj, &adj_w_jc, adj_ang);
    /* X_wc = X_wj * X_jc */
    qt adj_q_jc = Q(0, 0, 0, 0); v3 adj_p_jc = V(0, 0, 0);
    adj_qmul(q_wj, q_jc, &adj_q_wj, &adj_q_jc, g_q);
    v3 adj_p_wj = g_p;
    adj_qrot(q_wj, p_jc, &adj_q_wj, &adj_p_jc, g_p);
    /* X_wj = X_wp * X_pj */
    qt adj_q_wp = Q(0, 0, 0, 0);
    adj_qmul(q_wp, q_pj, &adj_q_wp, NULL, adj_q_wj);
    v3 adj_p_wp = adj_p_wj;
    adj_qrot(q_wp, p_pj, &adj_q_wp, NULL, adj_p_wj);
    /* joint coordinates */
    if (ty == JOINT_REVOLUTE) {
      adj_q_axis_angle(axis, q[0], NULL, &aq[0], adj_q_jc);
      aqd[0] += vdot(axis, adj_w_jc);
    } else if (ty == JOINT_FREE) {
      add3(aq, adj_p_jc); add4(aq + 3, adj_q_jc); add3(aqd, adj_w_jc); add3(aqd + 3, adj_v_jc);
    } else if (ty == JOINT_COMPOUND) {
      v3 adj_a0 = vscale(adj_w_jc, qd[0]), adj_a1 = vscale(adj_w_jc, qd[1]), adj_a2 = vscale(adj_w_jc, qd[2]);
      aqd[0] += vdot(a0, adj_w_jc); aqd[1] += vdot(a1, adj_w_jc); aqd[2] += vdot(a2, adj_w_jc);
      /* q_jc = q2 * (q1 * q0) */
      qt q10 = qmul(q1, q0);
      qt adj_q2 = Q(0, 0, 0, 0), adj_q10 = adj_q2, adj_q1 = adj_q2, adj_q0 = adj_q2;
      adj_qmul(q2, q10, &adj_q2, &adj_q10, adj_q_jc);
      /* q2 = aa(a2, q[2]) */
      adj_q_axis_angle(a2, q[2], &adj_a2, &aq[2], adj_q2);
      /* a2 = rot(q1*(q0*q_off), ez) */
      qt q0o = qmul(q0, q_off); qt q10o = qmul(q1, q0o);
      qt adj_q10o = Q(0, 0, 0, 0), adj_q0o = adj_q10o;
      adj_qrot(q10o, V(0, 0, 1), &adj_q10o, NULL, adj_a2);
      adj_qmul(q1, q0o, &adj_q1, &adj_q0o, adj_q10o);
      adj_qmul(q1, q0, &adj_q1, &adj_q0, adj_q10);
      /* q1 = aa(a1, q[1]) */
      adj_q_axis_angle(a1, q[1], &adj_a1, &aq[1], adj_q1);
      /* a1 = rot(q0*q_off, ey) */
      adj_qrot(q0o, V(0, 1, 0), &adj_q0o, NULL, adj_a1);
      adj_qmul(q0, q_off, &adj_q0, NULL, adj_q0o);
      /* q0 = aa(a0, q[0]); a0 = rot(q_off, ex) is constant */
      adj_q_axis_angle(a0, q[0], &adj_a0, &aq[0], adj_q0);
    }
    /* parent */
    if (par >= 0) {
      add3(adj_body_q + par * 7, adj_p_wp); add4(adj_body_q + par * 7 + 3, adj_q_wp);
      add3(adj_body_qd + par * 6, adj_w_wp); add3(adj_body_qd + par * 6 + 3, adj_v_wp);
    }
  }
}

/* ------------------------------------------------------------------- contacts
 * one env: body_q [nb][7], body_qd [nb][6], body_f [nb][6] (accumulated, atomic_sub in the reference) */
static void contacts_fwd(const RefTemplate *t, const real *body_q, const real *body_qd, real *body_f) {
  for (int k = 0; k < t->nc; ++k) {
    int b = t->c_body[k];
    v3 p = ld3(body_q + b * 7); qt q = ld4(body_q + b * 7 + 3);
    v3 w = ld3(body_qd + b * 6), v = ld3(body_qd + b * 6 + 3);
    v3 n = V(0, 1, 0);
    v3 cp = vsub(vadd(p, qrot(q, ld3(t->c_point + k * 3))), vscale(n, t->c_dist[k]));
    v3 r = vsub(cp, vadd(p, qrot(q, ld3(t->com + b * 3))));
    v3 dpdt = vadd(v, vcross(w, r));
    real c = vdot(n, cp);
    if (c > (real)0) continue;
    const real *mat = t->materials + t->c_mat[k] * 4;
    real ke = mat[0], kd = mat[1], kf = mat[2], mu = mat[3];
    real vn = vdot(n, dpdt);
    v3 vt = vsub(dpdt, vscale(n, vn));
    real fn = c * ke;
    real fd = (vn < (real)0 ? vn : (real)0) * kd * (c < (real)0 ? (real)1 : (real)0);
    real a_ = kf * vlen(vt), b_ = (real)0 - mu * (fn + fd);
    v3 ft = vscale(vnormalize(vt), a_ < b_ ? a_ : b_);
    v3 f_total = vadd(vscale(n, fn + fd), ft);
    f_total = V(clampr(f_total.x, -500, 500), clampr(f_total.y, -500, 500), clampr(f_total.z, -500, 500));
    v3 t_total = vcross(r, f_total);
    real *bf = body_f + b * 6;
    bf[0] -= t_total.x; bf[1] -= t_total.y; bf[2] -= t_total.z;
    bf[3] -= f_total.x; bf[4] -= f_total.y; bf[5] -= f_total.z;
  }
}

/* The touch decision of the HIP kernels, bit for bit (ref_rollout_backward_forced, ref_touch_fp32): they evaluate the height of
 * integrator_euler.py:118-121 in fp32 with PINNED roundings -- row 1 of the rotation matrix of q and the dot product as explicit
 * fused multiply-adds, nothing else contracted (ppr-diffphys_amd/csrc/pd_math.h rot_row1 / contact_height) -- so that their forward
 * and adjoint kernels, and this restatement, get the same bits from the same stored (q, p_y) and candidate.  This file is built
 * with -ffp-contract=off; fmaf is the correctly rounded fused operation. */
static int touch_pinned_fp32(const RefTemplate *t, int k, const real *body_q) {
  const int b = t->c_body[k];
  const float py = (float)body_q[b * 7 + 1];
  const float qx = (float)body_q[b * 7 + 3], qy = (float)body_q[b * 7 + 4], qz = (float)body_q[b * 7 + 5], qw = (float)body_q[b * 7 + 6];
  const float s = fmaf(2.0f * qw, qw, -1.0f), tx = 2.0f * qx, ty = 2.0f * qy, tz = 2.0f * qz;
  const float wz = tz * qw, wx = tx * qw;
  const float r0 = fmaf(tx, qy, wz), r1 = fmaf(ty, qy, s), r2 = fmaf(ty, qz, -wx);
  const float Px = (float)t->c_point[k * 3], Py = (float)t->c_point[k * 3 + 1], Pz = (float)t->c_point[k * 3 + 2];
  const float h = fmaf(r2, Pz, fmaf(r1, Py, fmaf(r0, Px, py))) - (float)t->c_dist[k];
  return !(h > 0.0f);
}

/* pinned != 0 (ref_rollout_backward_forced): `touching` is the decision of the fp32 implementation whose trajectory is being
 * differentiated (touch_pinned_fp32 on its stored state), whatever side of 0 this precision's height lands on.
 * list != NULL && list[0] >= 0: exactly the candidates list[1 .. list[0]] touch (a decision taken elsewhere, e.g. by ref_touch_fp32
 * on the unperturbed states when the states passed here carry a deliberate perturbation). */
static void contacts_adj(const RefTemplate *t, const real *body_q, const real *body_qd, const real *adj_body_f,
                         real *adj_body_q, real *adj_body_qd, int pinned, const int *list) {
  const int use_list = list != NULL && list[0] >= 0;
  const int n_it = use_list ? list[0] : t->nc;
  for (int it = 0; it < n_it; ++it) {
    const int k = use_list ? list[1 + it] : it;
    int b = t->c_body[k];
    v3 p = ld3(body_q + b * 7); qt q = ld4(body_q + b * 7 + 3);
    v3 w = ld3(body_qd + b * 6), v = ld3(body_qd + b * 6 + 3);
    v3 n = V(0, 1, 0);
    v3 cpt = ld3(t->c_point + k * 3), com = ld3(t->com + b * 3);
    v3 cp = vsub(vadd(p, qrot(q, cpt)), vscale(n, t->c_dist[k]));
    v3 r = vsub(cp, vadd(p, qrot(q, com)));
    v3 dpdt = vadd(v, vcross(w, r));
    real c = vdot(n, cp);
    if (!use_list && (pinned ? !touch_pinned_fp32(t, k, body_q) : c > (real)0)) continue;
    const real *mat = t->materials + t->c_mat[k] * 4;
    real ke = mat[0], kd = mat[1], kf = mat[2], mu = mat[3];
    real vn = vdot(n, dpdt);
    v3 vt = vsub(dpdt, vscale(n, vn));
    real fn = c * ke;
    real stepc = (c < (real)0 ? (real)1 : (real)0);
    real fd = (vn < (real)0 ? vn : (real)0) * kd * stepc;
    real lvt = vlen(vt);
    v3 nvt = vnormalize(vt);
    real a_ = kf * lvt, b_ = (real)0 - mu * (fn + fd);
    real m = a_ < b_ ? a_ : b_;
    v3 f_raw = vadd(vscale(n, fn + fd), vscale(nvt, m));
    v3 fc = V(clampr(f_raw.x, -500, 500), clampr(f_raw.y, -500, 500), clampr(f_raw.z, -500, 500));
    /* reverse */
    v3 g_t = ld3(adj_body_f + b * 6), g_f = ld3(adj_body_f + b * 6 + 3);
    v3 adj_t_total = vneg(g_t), adj_fc = vneg(g_f);
    v3 adj_r = V(0, 0, 0);
    adj_vcross(r, fc, &adj_r, &adj_fc, adj_t_total);
    v3 adj_f_raw = V(adj_fc.x * clamp_pass(f_raw.x, -500, 500), adj_fc.y * clamp_pass(f_raw.y, -500, 500),
                     adj_fc.z * clamp_pass(f_raw.z, -500, 500));
    real adj_fnfd = vdot(n, adj_f_raw);
    v3 adj_nvt = vscale(adj_f_raw, m);
    real adj_m = vdot(adj_f_raw, nvt);
    real adj_lvt = 0;
    if (a_ < b_) adj_lvt = adj_m * kf; else adj_fnfd += -mu * adj_m;
    v3 adj_vt = V(0, 0, 0);
    adj_vnormalize(vt, &adj_vt, adj_nvt);
    adj_vlen(vt, &adj_vt, adj_lvt);
    real adj_fn = adj_fnfd, adj_fd = adj_fnfd;
    real adj_c = adj_fn * ke;
    real adj_vn = (vn < (real)0 ? (real)1 : (real)0) * kd * stepc * adj_fd;
    v3 adj_dpdt = adj_vt;
    adj_vn += -vdot(n, adj_vt);
    vacc(&adj_dpdt, vscale(n, adj_vn));
    v3 adj_w = V(0, 0, 0);
    adj_vcross(w, r, &adj_w, &adj_r, adj_dpdt);
    v3 adj_cp = vscale(n, adj_c);
    vacc(&adj_cp, adj_r);
    v3 adj_p = vneg(adj_r);
    qt adj_q = Q(0, 0, 0, 0);
    adj_qrot(q, com, &adj_q, NULL, vneg(adj_r));
    vacc(&adj_p, adj_cp);
    adj_qrot(q, cpt, &adj_q, NULL, adj_cp);
    add3(adj_body_q + b * 7, adj_p); add4(adj_body_q + b * 7 + 3, adj_q);
    add3(adj_body_qd + b * 6, adj_w); add3(adj_body_qd + b * 6 + 3, adj_dpdt);
  }
}

/* --------------------------------------------------------------------- joints */
static real joint_force(real q, real qd, real target, real ke, real kd, real act, real lo, real up, real lke, real lkd) {
  real limit_f = 0;
  if (q < lo) limit_f = lke * (lo - q) - lkd * (qd < (real)0 ? qd : (real)0);
  if (q > up) limit_f = lke * (up - q) - lkd * (qd > (real)0 ? qd : (real)0);
  return ke * (q - target) + kd * qd + act - limit_f;
}
/* adjoint of joint_force wrt (q, qd, target, ke, kd, act) given g = adj of the scalar */
static void joint_force_adj(real q, real qd, real target, real ke, real kd, real lo, real up, real lke, real lkd, real g,
                            real *adj_q, real *adj_qd, real *adj_target, real *adj_ke, real *adj_kd, real *adj_act) {
  *adj_ke += g * (q - target); *adj_q += g * ke; *adj_target += -g * ke;
  *adj_kd += g * qd; *adj_qd += g * kd; *adj_act += g;
  real adj_limit = -g;
  if (q > up) { *adj_q += -lke * adj_limit; if (qd > (real)0) *adj_qd += -lkd * adj_limit; }
  else if (q < lo) { *adj_q += -lke * adj_limit; if (qd < (real)0) *adj_qd += -lkd * adj_limit; }
}

static void quat_decompose(qt q, real *ang) {
  v3 c0 = qrot(q, V(1, 0, 0)), c1 = qrot(q, V(0, 1, 0)), c2 = qrot(q, V(0, 0, 1));
  ang[0] = -R_ATAN2(c2.y, c2.z); ang[1] = -asin_c(-c2.x); ang[2] = -R_ATAN2(c1.x, c0.x);
}
static void quat_decompose_adj(qt q, const real *g, qt *adj_q) {
  v3 ex = V(1, 0, 0), ey = V(0, 1, 0), ez = V(0, 0, 1);
  v3 c0 = qrot(q, ex), c1 = qrot(q, ey), c2 = qrot(q, ez);
  v3 a0 = V(0, 0, 0), a1 = a0, a2 = a0;
  real gphi = -g[0], gth = -g[1], gpsi = -g[2];
  { real y = c2.y, x = c2.z, d = x * x + y * y; a2.y += gphi * x / d; a2.z += -gphi * y / d; }
  { real s = -c2.x; a2.x += -gth * inv_sqrt_1mx2(s); }
  { real y = c1.x, x = c0.x, d = x * x + y * y; a1.x += gpsi * x / d; a0.x += -gpsi * y / d; }
  adj_qrot(q, ex, adj_q, NULL, a0); adj_qrot(q, ey, adj_q, NULL, a1); adj_qrot(q, ez, adj_q, NULL, a2);
}

typedef struct {  /* per-joint forward locals shared by fwd and adj */
  int par, ty;
  v3 p_pj; qt q_pj; v3 pp; qt qp; v3 x_p; qt q_p; v3 r_p, w_p, v_p;
  v3 x_c; qt q_c; v3 r_c, w_c, v_c; v3 com_p, com_c;
  v3 x_err, v_err, w_err; qt r_err;
} JointCtx;

static void joint_ctx(const RefTemplate *t, int i, const real *body_q, const real *body_qd, JointCtx *c) {
  c->par = t->joint_parent[i]; c->ty = t->joint_type[i];
  c->p_pj = ld3(t->X_p + i * 7); c->q_pj = ld4(t->X_p + i * 7 + 3);
  c->x_p = c->p_pj; c->q_p = c->q_pj; c->r_p = V(0, 0, 0); c->w_p = V(0, 0, 0); c->v_p = V(0, 0, 0);
  c->pp = V(0, 0, 0); c->qp = Q(0, 0, 0, 1); c->com_p = V(0, 0, 0);
  if (c->par >= 0) {
    int p = c->par;
    c->pp = ld3(body_q + p * 7); c->qp = ld4(body_q + p * 7 + 3); c->com_p = ld3(t->com + p * 3);
    c->x_p = vadd(c->pp, qrot(c->qp, c->p_pj)); c->q_p = qmul(c->qp, c->q_pj);
    c->r_p = vsub(c->x_p, vadd(c->pp, qrot(c->qp, c->com_p)));
    c->w_p = ld3(body_qd + p * 6); c->v_p = ld3(body_qd + p * 6 + 3);
  }
  c->x_c = ld3(body_q + i * 7); c->q_c = ld4(body_q + i * 7 + 3); c->com_c = ld3(t->com + i * 3);
  c->r_c = vsub(c->x_c, vadd(c->x_c, qrot(c->q_c, c->com_c)));
  c->w_c = ld3(body_qd + i * 6); c->v_c = ld3(body_qd + i * 6 + 3);
  c->x_err = vsub(c->x_c, c->x_p); c->r_err = qmul(qconj(c->q_p), c->q_c);
  c->v_err = vsub(c->v_c, c->v_p); c->w_err = vsub(c->w_c, c->w_p);
}

static v3 clamp3(v3 a, real l) { return V(clampr(a.x, -l, l), clampr(a.y, -l, l), clampr(a.z, -l, l)); }
static v3 clamp3_pass(v3 a, v3 g, real l) { return V(g.x * clamp_pass(a.x, -l, l), g.y * clamp_pass(a.y, -l, l), g.z * clamp_pass(a.z, -l, l)); }

static void joints_fwd(const RefTemplate *t, const real *body_q, const real *body_qd, const real *target, const real *act,
                       const real *tke, const real *tkd, real *body_f) {
  const real ake = t->attach_ke, akd = t->attach_kd, ads = (real)0.01;
  for (int i = 0; i < t->nb; ++i) {
    JointCtx c; joint_ctx(t, i, body_q, body_qd, &c);
    if (c.ty == JOINT_FREE) continue;
    int qds = t->qd_start[i];
    v3 t_total = V(0, 0, 0), f_total = V(0, 0, 0);
    if (c.ty == JOINT_FIXED) {
      v3 ang_err = vscale(vnormalize(qv(c.r_err)), acos_c(c.r_err.w) * (real)2);
      if (g_twist_atan2) ang_err = vscale(qv(c.r_err), fixed_ang_h(qv(c.r_err), c.r_err.w, NULL, NULL));
      f_total = vadd(f_total, vadd(vscale(c.x_err, ake), vscale(c.v_err, akd)));
      t_total = vadd(t_total, vadd(vscale(qrot(c.q_p, ang_err), ake), vscale(c.w_err, akd * ads)));
    }
    if (c.ty == JOINT_REVOLUTE) {
      v3 axis = ld3(t->axis + i * 3);
      v3 axis_p = qrot(c.q_p, axis), axis_c = qrot(c.q_c, axis);
      v3 a = vscale(axis, vdot(qv(c.r_err), axis));
      qt twist = qnormalize(Q(a.x, a.y, a.z, c.r_err.w));
      real sgn = vdot(axis, qv(twist)) < (real)0 ? (real)-1 : (real)1;
      real q = acos_c(twist.w) * (real)2 * sgn;
      if (g_twist_atan2) q = twist_angle_atan2(axis, vdot(qv(c.r_err), axis), c.r_err.w, NULL, NULL);
      real qd = vdot(c.w_err, axis_p);
      real jf = joint_force(q, qd, target[qds], tke[qds], tkd[qds], act[qds], t->limit_lower[qds], t->limit_upper[qds],
                            t->limit_ke[qds], t->limit_kd[qds]);
      t_total = vscale(axis_p, jf);
      v3 swing = vcross(axis_p, axis_c);
      f_total = vadd(f_total, vadd(vscale(c.x_err, ake), vscale(c.v_err, akd)));
      t_total = vadd(t_total, vadd(vscale(swing, ake), vscale(vsub(c.w_err, vscale(axis_p, qd)), akd * ads)));
    }
    if (c.ty == JOINT_COMPOUND) {
      qt q_off = ld4(t->X_c + i * 7 + 3);
      qt q_pc = qmul(qmul(qmul(qconj(q_off), qconj(c.q_p)), c.q_c), q_off);
      real ang[3]; quat_decompose(q_pc, ang);
      v3 ax[3]; ax[0] = V(1, 0, 0);
      qt q_0 = q_axis_angle(ax[0], ang[0]);
      ax[1] = qrot(q_0, V(0, 1, 0));
      qt q_1 = q_axis_angle(ax[1], ang[1]);
      ax[2] = qrot(qmul(q_1, q_0), V(0, 0, 1));
      qt q_w = qmul(c.q_p, q_off);
      t_total = V(0, 0, 0);
      for (int k = 0; k < 3; ++k) {
        v3 axw = qrot(q_w, ax[k]);
        int j = qds + k;
        real jf = joint_force(ang[k], vdot(axw, c.w_err), target[j], tke[j], tkd[j], act[j], t->limit_lower[j],
                              t->limit_upper[j], t->limit_ke[j], t->limit_kd[j]);
        t_total = vadd(t_total, vscale(axw, jf));
      }
      t_total = clamp3(t_total, (real)1e4);
      v3 f_sub = clamp3(vadd(vscale(c.x_err, ake), vscale(c.v_err, akd)), (real)1e4);
      f_total = vadd(f_total, f_sub);
    }
    if (c.par >= 0) {
      add3(body_f + c.par * 6, vadd(t_total, vcross(c.r_p, f_total)));
      add3(body_f + c.par * 6 + 3, f_total);
    }
    add3(body_f + i * 6, vneg(vadd(t_total, vcross(c.r_c, f_total))));
    add3(body_f + i * 6 + 3, vneg(f_total));
  }
}

static void joints_adj(const RefTemplate *t, const real *body_q, const real *body_qd, const real *target, const real *act,
                       const real *tke, const real *tkd, const real *adj_body_f, real *adj_body_q, real *adj_body_qd,
                       real *adj_target, real *adj_act, real *adj_tke, real *adj_tkd) {
  const real ake = t->attach_ke, akd = t->attach_kd, ads = (real)0.01;
  (void)act;
  for (int i = t->nb - 1; i >= 0; --i) {
    JointCtx c; joint_ctx(t, i, body_q, body_qd, &c);
    if (c.ty == JOINT_FREE) continue;
    int qds = t->qd_start[i];
    /* ---- recompute forward f_total (needed by the cross-product adjoints) */
    v3 f_total = V(0, 0, 0);
    v3 f_raw = vadd(vscale(c.x_err, ake), vscale(c.v_err, akd));
    if (c.ty == JOINT_COMPOUND) f_total = clamp3(f_raw, (real)1e4); else f_total = f_raw;
    /* ---- adjoint of the scatter */
    v3 gt_c = ld3(adj_body_f + i * 6), gf_c = ld3(adj_body_f + i * 6 + 3);
    v3 adj_t = vneg(gt_c), adj_f = vneg(gf_c);
    v3 adj_r_c = V(0, 0, 0), adj_r_p = V(0, 0, 0);
    adj_vcross(c.r_c, f_total, &adj_r_c, &adj_f, vneg(gt_c));
    if (c.par >= 0) {
      v3 gt_p = ld3(adj_body_f + c.par * 6), gf_p = ld3(adj_body_f + c.par * 6 + 3);
      vacc(&adj_t, gt_p); vacc(&adj_f, gf_p);
      adj_vcross(c.r_p, f_total, &adj_r_p, &adj_f, gt_p);
    }
    v3 adj_x_err = V(0, 0, 0), adj_v_err = V(0, 0, 0), adj_w_err = V(0, 0, 0);
    qt adj_r_err = Q(0, 0, 0, 0), adj_q_p = Q(0, 0, 0, 0), adj_q_c = Q(0, 0, 0, 0);
    if (c.ty == JOINT_FIXED) {
      v3 rv = qv(c.r_err); real ac = acos_c(c.r_err.w) * (real)2;
      v3 nrm = vnormalize(rv); v3 ang_err = vscale(nrm, ac);
      vacc(&adj_x_err, vscale(adj_f, ake)); vacc(&adj_v_err, vscale(adj_f, akd));
      vacc(&adj_w_err, vscale(adj_t, akd * ads));
      real hss = 0, h_w = 0, h = 0;
      if (g_twist_atan2) { h = fixed_ang_h(rv, c.r_err.w, &hss, &h_w); ang_err = vscale(rv, h); }
      v3 adj_ang_err = V(0, 0, 0);
      adj_qrot(c.q_p, ang_err, &adj_q_p, &adj_ang_err, vscale(adj_t, ake));
      if (g_twist_atan2) {
        const real va = vdot(rv, adj_ang_err);
        adj_r_err.x += adj_ang_err.x * h + rv.x * (va * hss); adj_r_err.y += adj_ang_err.y * h + rv.y * (va * hss);
        adj_r_err.z += adj_ang_err.z * h + rv.z * (va * hss);
        adj_r_err.w += va * h_w;
      } else {
        v3 adj_nrm = vscale(adj_ang_err, ac); real adj_ac = vdot(adj_ang_err, nrm);
        v3 adj_rv = V(0, 0, 0);
        adj_vnormalize(rv, &adj_rv, adj_nrm);
        adj_r_err.x += adj_rv.x; adj_r_err.y += adj_rv.y; adj_r_err.z += adj_rv.z;
        adj_r_err.w += -(real)2 * adj_ac * inv_sqrt_1mx2(c.r_err.w);
      }
    }
    if (c.ty == JOINT_REVOLUTE) {
      v3 axis = ld3(t->axis + i * 3);
      v3 axis_p = qrot(c.q_p, axis), axis_c = qrot(c.q_c, axis);
      real da = vdot(qv(c.r_err), axis);
      v3 a = vscale(axis, da);
      qt tq = Q(a.x, a.y, a.z, c.r_err.w);
      qt twist = qnormalize(tq);
      real sgn = vdot(axis, qv(twist)) < (real)0 ? (real)-1 : (real)1;
      real q = acos_c(twist.w) * (real)2 * sgn;
      real dq_dda = 0, dq_dw = 0;
      if (g_twist_atan2) q = twist_angle_atan2(axis, da, c.r_err.w, &dq_dda, &dq_dw);
      real qd = vdot(c.w_err, axis_p);
      real jf = joint_force(q, qd, target[qds], tke[qds], tkd[qds], act[qds], t->limit_lower[qds], t->limit_upper[qds],
                            t->limit_ke[qds], t->limit_kd[qds]);
      vacc(&adj_x_err, vscale(adj_f, ake)); vacc(&adj_v_err, vscale(adj_f, akd));
      /* t = jf*axis_p + swing*ake + (w_err - qd*axis_p)*akd*ads */
      real adj_jf = vdot(adj_t, axis_p);
      v3 adj_axis_p = vscale(adj_t, jf), adj_axis_c = V(0, 0, 0);
      v3 adj_swing = vscale(adj_t, ake);
      vacc(&adj_w_err, vscale(adj_t, akd * ads));
      real adj_qd = -vdot(adj_t, axis_p) * akd * ads;
      vacc(&adj_axis_p, vscale(adj_t, -qd * akd * ads));
      adj_vcross(axis_p, axis_c, &adj_axis_p, &adj_axis_c, adj_swing);
      real adj_q = 0;
      joint_force_adj(q, qd, target[qds], tke[qds], tkd[qds], t->limit_lower[qds], t->limit_upper[qds], t->limit_ke[qds],
                      t->limit_kd[qds], adj_jf, &adj_q, &adj_qd, &adj_target[qds], &adj_tke[qds], &adj_tkd[qds], &adj_act[qds]);
      /* qd = dot(w_err, axis_p) */
      vacc(&adj_w_err, vscale(axis_p, adj_qd)); vacc(&adj_axis_p, vscale(c.w_err, adj_qd));
      /* q = acos(twist.w)*2*sgn */
      qt adj_twist = Q(0, 0, 0, -adj_q * (real)2 * sgn * inv_sqrt_1mx2(twist.w));
      qt adj_tq = Q(0, 0, 0, 0);
      adj_qnormalize(tq, &adj_tq, adj_twist);
      real adj_da = vdot(qv(adj_tq), axis);
      if (g_twist_atan2) { adj_da = adj_q * dq_dda; adj_tq.w = adj_q * dq_dw; }
      adj_r_err.x += axis.x * adj_da; adj_r_err.y += axis.y * adj_da; adj_r_err.z += axis.z * adj_da;
      adj_r_err.w += adj_tq.w;
      adj_qrot(c.q_p, axis, &adj_q_p, NULL, adj_axis_p);
      adj_qrot(c.q_c, axis, &adj_q_c, NULL, adj_axis_c);
    }
    if (c.ty == JOINT_COMPOUND) {
      qt q_off = ld4(t->X_c + i * 7 + 3);
      qt qa = qmul(qconj(q_off), qconj(c.q_p)); qt qb = qmul(qa, c.q_c); qt q_pc = qmul(qb, q_off);
      real ang[3]; quat_decompose(q_pc, ang);
      v3 ax[3]; ax[0] = V(1, 0, 0);
      qt q_0 = q_axis_angle(ax[0], ang[0]);
      ax[1] = qrot(q_0, V(0, 1, 0));
      qt q_1 = q_axis_angle(ax[1], ang[1]);
      qt q10 = qmul(q_1, q_0);
      ax[2] = qrot(q10, V(0, 0, 1));
      qt q_w = qmul(c.q_p, q_off);
      v3 axw[3]; real jf[3], qdk[3]; v3 t_raw = V(0, 0, 0);
      for (int k = 0; k < 3; ++k) {
        axw[k] = qrot(q_w, ax[k]); qdk[k] = vdot(axw[k], c.w_err);
        int j = qds + k;
        jf[k] = joint_force(ang[k], qdk[k], target[j], tke[j], tkd[j], act[j], t->limit_lower[j], t->limit_upper[j],
                            t->limit_ke[j], t->limit_kd[j]);
        t_raw = vadd(t_raw, vscale(axw[k], jf[k]));
      }
      /* f_total += clamp(f_raw) */
      v3 adj_f_raw = clamp3_pass(f_raw, adj_f, (real)1e4);
      vacc(&adj_x_err, vscale(adj_f_raw, ake)); vacc(&adj_v_err, vscale(adj_f_raw, akd));
      v3 adj_t_raw = clamp3_pass(t_raw, adj_t, (real)1e4);
      real adj_ang[3] = {0, 0, 0};
      v3 adj_ax[3] = {V(0, 0, 0), V(0, 0, 0), V(0, 0, 0)};
      qt adj_q_w = Q(0, 0, 0, 0);
      for (int k = 2; k >= 0; --k) {
        int j = qds + k;
        real adj_jf = vdot(adj_t_raw, axw[k]);
        v3 adj_axw = vscale(adj_t_raw, jf[k]);
        real adj_qdk = 0;
        joint_force_adj(ang[k], qdk[k], target[j], tke[j], tkd[j], t->limit_lower[j], t->limit_upper[j], t->limit_ke[j],
                        t->limit_kd[j], adj_jf, &adj_ang[k], &adj_qdk, &adj_target[j], &adj_tke[j], &adj_tkd[j], &adj_act[j]);
        vacc(&adj_axw, vscale(c.w_err, adj_qdk)); vacc(&adj_w_err, vscale(axw[k], adj_qdk));
        adj_qrot(q_w, ax[k], &adj_q_w, &adj_ax[k], adj_axw);
      }
      adj_qmul(c.q_p, q_off, &adj_q_p, NULL, adj_q_w);
      /* ax[2] = rot(q_1*q_0, ez) */
      qt adj_q10 = Q(0, 0, 0, 0), adj_q_1 = adj_q10, adj_q_0 = adj_q10;
      adj_qrot(q10, V(0, 0, 1), &adj_q10, NULL, adj_ax[2]);
      adj_qmul(q_1, q_0, &adj_q_1, &adj_q_0, adj_q10);
      adj_q_axis_angle(ax[1], ang[1], &adj_ax[1], &adj_ang[1], adj_q_1);
      adj_qrot(q_0, V(0, 1, 0), &adj_q_0, NULL, adj_ax[1]);
      adj_q_axis_angle(ax[0], ang[0], NULL, &adj_ang[0], adj_q_0);
      qt adj_q_pc = Q(0, 0, 0, 0);
      quat_decompose_adj(q_pc, adj_ang, &adj_q_pc);
      /* q_pc = ((conj(q_off)*conj(q_p))*q_c)*q_off */
      qt adj_qb = Q(0, 0, 0, 0), adj_qa = adj_qb, adj_cqp = adj_qb;
      adj_qmul(qb, q_off, &adj_qb, NULL, adj_q_pc);
      adj_qmul(qa, c.q_c, &adj_qa, &adj_q_c, adj_qb);
      adj_qmul(qconj(q_off), qconj(c.q_p), NULL, &adj_cqp, adj_qa);
      qacc(&adj_q_p, qconj(adj_cqp));
    }
    /* ---- common tail: errors -> body states */
    {
      qt adj_cqp = Q(0, 0, 0, 0);
      adj_qmul(qconj(c.q_p), c.q_c, &adj_cqp, &adj_q_c, adj_r_err);
      qacc(&adj_q_p, qconj(adj_cqp));
    }
    v3 adj_x_c = adj_x_err, adj_x_p = vneg(adj_x_err);
    v3 adj_w_c = adj_w_err, adj_w_p = vneg(adj_w_err), adj_v_c = adj_v_err, adj_v_p = vneg(adj_v_err);
    /* r_c = x_c - (x_c + rot(q_c, com_c)) */
    adj_qrot(c.q_c, c.com_c, &adj_q_c, NULL, vneg(adj_r_c));
    add3(adj_body_q + i * 7, adj_x_c); add4(adj_body_q + i * 7 + 3, adj_q_c);
    add3(adj_body_qd + i * 6, adj_w_c); add3(adj_body_qd + i * 6 + 3, adj_v_c);
    if (c.par >= 0) {
      int p = c.par;
      qt adj_qp = Q(0, 0, 0, 0); v3 adj_pp = V(0, 0, 0);
      /* r_p = x_p - (pp + rot(qp, com_p)) */
      vacc(&adj_x_p, adj_r_p); vacc(&adj_pp, vneg(adj_r_p));
      adj_qrot(c.qp, c.com_p, &adj_qp, NULL, vneg(adj_r_p));
      /* x_p = pp + rot(qp, p_pj); q_p = qp*q_pj */
      vacc(&adj_pp, adj_x_p);
      adj_qrot(c.qp, c.p_pj, &adj_qp, NULL, adj_x_p);
      adj_qmul(c.qp, c.q_pj, &adj_qp, NULL, adj_q_p);
      add3(adj_body_q + p * 7, adj_pp); add4(adj_body_q + p * 7 + 3, adj_qp);
      add3(adj_body_qd + p * 6, adj_w_p); add3(adj_body_qd + p * 6 + 3, adj_v_p);
    }
  }
}

/* ------------------------------------------------------------------ integrate */
static void integrate_fwd(const RefTemplate *t, const real *body_q, const real *body_qd, const real *body_f, const real *inv_m,
                          const real *I, const real *inv_I, real dt, real *q_new, real *qd_new) {
  for (int i = 0; i < t->nb; ++i) {
    v3 x0 = ld3(body_q + i * 7); qt r0 = ld4(body_q + i * 7 + 3);
    v3 w0 = ld3(body_qd + i * 6), v0 = ld3(body_qd + i * 6 + 3);
    v3 t0 = ld3(body_f + i * 6), f0 = ld3(body_f + i * 6 + 3);
    v3 com = ld3(t->com + i * 3); v3 g = ld3(t->gravity);
    real im = inv_m[i]; real nz = im != (real)0 ? (real)1 : (real)0;
    v3 x_com = vadd(x0, qrot(r0, com));
    v3 v1 = vadd(v0, vscale(vadd(vscale(f0, im), vscale(g, nz)), dt));
    v3 x1 = vadd(x_com, vscale(v1, dt));
    v3 wb = qrot_inv(r0, w0);
    v3 Iwb; mat_vec(I + i * 9, wb, &Iwb);
    v3 tb = vsub(qrot_inv(r0, t0), vcross(wb, Iwb));
    v3 a; mat_vec(inv_I + i * 9, tb, &a);
    v3 w1 = qrot(r0, vadd(wb, vscale(a, dt)));
    qt r1 = qnormalize(qadd(r0, qscale(qmul(Q(w1.x, w1.y, w1.z, 0), r0), (real)0.5 * dt)));
    w1 = vscale(w1, (real)1 - (real)0.1 * dt);
    w1 = clamp3(w1, 10); v1 = clamp3(v1, 10);
    st3(q_new + i * 7, vsub(x1, qrot(r1, com))); st4(q_new + i * 7 + 3, r1);
    st3(qd_new + i * 6, w1); st3(qd_new + i * 6 + 3, v1);
  }
}

static void integrate_adj(const RefTemplate *t, const real *body_q, const real *body_qd, const real *body_f, const real *inv_m,
                          const real *I, const real *inv_I, real dt, const real *adj_q_new, const real *adj_qd_new,
                          real *adj_body_q, real *adj_body_qd, real *adj_body_f, real *adj_inv_m, real *adj_I, real *adj_inv_I,
                          const int *forced_mask) {
  /* forced_mask != NULL (ref_rollout_backward_forced): per body the 6-bit mask (w.x w.y w.z v.x v.y v.z) of the components the
   * forward pass being differentiated clamped (:78-88) -- its decision, not this precision's recomputation */
  for (int i = 0; i < t->nb; ++i) {
    v3 x0 = ld3(body_q + i * 7); qt r0 = ld4(body_q + i * 7 + 3);
    v3 w0 = ld3(body_qd + i * 6), v0 = ld3(body_qd + i * 6 + 3);
    v3 t0 = ld3(body_f + i * 6), f0 = ld3(body_f + i * 6 + 3);
    v3 com = ld3(t->com + i * 3); v3 g = ld3(t->gravity);
    real im = inv_m[i]; real nz = im != (real)0 ? (real)1 : (real)0;
    (void)x0;
    v3 v1 = vadd(v0, vscale(vadd(vscale(f0, im), vscale(g, nz)), dt));
    v3 wb = qrot_inv(r0, w0);
    v3 Iwb; mat_vec(I + i * 9, wb, &Iwb);
    v3 tb = vsub(qrot_inv(r0, t0), vcross(wb, Iwb));
    v3 a; mat_vec(inv_I + i * 9, tb, &a);
    v3 u = vadd(wb, vscale(a, dt));
    v3 w1 = qrot(r0, u);
    qt W = Q(w1.x, w1.y, w1.z, 0);
    qt rq = qadd(r0, qscale(qmul(W, r0), (real)0.5 * dt));
    qt r1 = qnormalize(rq);
    v3 w1d = vscale(w1, (real)1 - (real)0.1 * dt);
    /* reverse */
    v3 g_p = ld3(adj_q_new + i * 7); qt g_r = ld4(adj_q_new + i * 7 + 3);
    v3 g_w = ld3(adj_qd_new + i * 6), g_v = ld3(adj_qd_new + i * 6 + 3);
    v3 adj_x1 = g_p; qt adj_r1 = g_r;
    adj_qrot(r1, com, &adj_r1, NULL, vneg(g_p));
    v3 adj_v1 = clamp3_pass(v1, g_v, 10);
    v3 adj_w1 = clamp3_pass(w1d, g_w, 10);
    if (forced_mask) {
      const int mk = forced_mask[i];
      adj_w1 = V(mk & 1 ? (real)0 : g_w.x, mk & 2 ? (real)0 : g_w.y, mk & 4 ? (real)0 : g_w.z);
      adj_v1 = V(mk & 8 ? (real)0 : g_v.x, mk & 16 ? (real)0 : g_v.y, mk & 32 ? (real)0 : g_v.z);
    }
    adj_w1 = vscale(adj_w1, (real)1 - (real)0.1 * dt);
    qt adj_rq = Q(0, 0, 0, 0);
    adj_qnormalize(rq, &adj_rq, adj_r1);
    qt adj_r0 = adj_rq; qt adj_W = Q(0, 0, 0, 0);
    adj_qmul(W, r0, &adj_W, &adj_r0, qscale(adj_rq, (real)0.5 * dt));
    vacc(&adj_w1, qv(adj_W));
    v3 adj_u = V(0, 0, 0);
    adj_qrot(r0, u, &adj_r0, &adj_u, adj_w1);
    v3 adj_wb = adj_u, adj_a = vscale(adj_u, dt);
    add_outer(adj_inv_I + i * 9, adj_a, tb);
    v3 adj_tb = matT_vec(inv_I + i * 9, adj_a);
    v3 adj_t0 = V(0, 0, 0);
    adj_qrot_inv(r0, t0, &adj_r0, &adj_t0, adj_tb);
    v3 adj_Iwb = V(0, 0, 0);
    adj_vcross(wb, Iwb, &adj_wb, &adj_Iwb, vneg(adj_tb));
    add_outer(adj_I + i * 9, adj_Iwb, wb);
    vacc(&adj_wb, matT_vec(I + i * 9, adj_Iwb));
    v3 adj_w0 = V(0, 0, 0);
    adj_qrot_inv(r0, w0, &adj_r0, &adj_w0, adj_wb);
    v3 adj_x_com = adj_x1; vacc(&adj_v1, vscale(adj_x1, dt));
    v3 adj_v0 = adj_v1; v3 adj_f0 = vscale(adj_v1, im * dt);
    adj_inv_m[i] += vdot(adj_v1, f0) * dt;
    v3 adj_x0 = adj_x_com;
    adj_qrot(r0, com, &adj_r0, NULL, adj_x_com);
    add3(adj_body_q + i * 7, adj_x0); add4(adj_body_q + i * 7 + 3, adj_r0);
    add3(adj_body_qd + i * 6, adj_w0); add3(adj_body_qd + i * 6 + 3, adj_v0);
    add3(adj_body_f + i * 6, adj_t0); add3(adj_body_f + i * 6 + 3, adj_f0);
  }
}

/* -------------------------------------------------------------------- rollout
 * Flat env-major arrays as in ForwardWarp (dp_model.py:1162-1172):
 *   q_init [bs*nq], qd_init [bs*nqd], torques/refs [T][bs*nqd], res_f [T][bs*nb][6], gains [bs*nqd],
 *   inv_mass [bs*nb], inertia / inv_inertia [bs*nb][9].
 * Stored trajectory: states_q [T+1][bs*nb][7], states_qd [T+1][bs*nb][6], states_f [T][bs*nb][6].
 * Outputs at frames: wp_pos [F][bs*nb][7], wp_vel [F][bs*nb][6], grf / jaf [F][bs*nb][6]. */
/* ------------------------------------------------------------------- singularity probe (test diagnostics)
 * Where the step function is not differentiable, an fp32 evaluation one ulp away may differentiate the other branch: the
 * gradient then jumps by a stiffness (ke = 1e4 N/m for a contact) while every value stays within rounding.  For each
 * (step, env) of a stored trajectory this reports how close the step sits to such a set:
 *   out[0]  min over contact candidates of |c|, the height whose sign decides "touching" (integrator_euler.py:130-133), metres
 *   out[1]  min over bodies and components of | |x| - 10 | for the velocities BEFORE the clamp of :78-88 (w1 after damping, v1)
 *   out[2]  min over TOUCHING candidates of |kf |vt| - mu (fn + fd) ... | i.e. |a_ - b_| of the Coulomb switch (:160-165), newtons
 *   out[3]  min over revolute joints of 1 - |twist.w|: the acos / its guarded adjoint at joint angle 0 (:398-400)
 *   out[4]  min over TOUCHING candidates and force components of | |f_k| - 500 |: the contact force clamp (:172-175), newtons
 * layout out[(step * bs + env) * 5 + k].  tests/test_gpu_tight.py uses it to EXPLAIN every env whose fp32 gradient is off. */
void ref_singularity_probe(const RefTemplate *t, int bs, int nsteps, real dt, const real *states_q, const real *states_qd,
                           const real *states_f, const real *inv_mass, const real *inertia, const real *inv_inertia, real *out) {
  const int nb = t->nb;
  const size_t SQ = (size_t)bs * nb * 7, SD = (size_t)bs * nb * 6;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < bs; ++e) {
    size_t oq = (size_t)e * nb * 7, od = (size_t)e * nb * 6;
    for (int s = 0; s < nsteps; ++s) {
      const real *bq = states_q + s * SQ + oq, *bqd = states_qd + s * SD + od, *bf = states_f + s * SD + od;
      real c_min = (real)1e30, cl_min = (real)1e30, fr_min = (real)1e30, ac_min = (real)1e30, fc_min = (real)1e30;
      for (int k = 0; k < t->nc; ++k) {
        int b = t->c_body[k];
        v3 p = ld3(bq + b * 7); qt q = ld4(bq + b * 7 + 3);
        v3 w = ld3(bqd + b * 6), v = ld3(bqd + b * 6 + 3);
        v3 cp = vadd(p, qrot(q, ld3(t->c_point + k * 3)));
        real c = cp.y - t->c_dist[k];
        real ac = c < 0 ? -c : c;
        if (ac < c_min) c_min = ac;
        if (c > (real)0) continue;
        v3 r = vsub(V(cp.x, c, cp.z), vadd(p, qrot(q, ld3(t->com + b * 3))));
        v3 dpdt = vadd(v, vcross(w, r));
        const real *mat = t->materials + t->c_mat[k] * 4;
        real vn = dpdt.y;
        v3 vt = V(dpdt.x, 0, dpdt.z);
        real fn = c * mat[0], fd = (vn < (real)0 ? vn : (real)0) * mat[1] * (c < (real)0 ? (real)1 : (real)0);
        real d = mat[2] * vlen(vt) - ((real)0 - mat[3] * (fn + fd));
        real a_ = mat[2] * vlen(vt), b_ = (real)0 - mat[3] * (fn + fd);
        if (d < 0) d = -d;
        if (d < fr_min) fr_min = d;
        {
          real lvt = vlen(vt), mm = a_ < b_ ? a_ : b_;
          v3 nv = lvt > (real)0 ? vscale(vt, (real)1 / lvt) : V(0, 0, 0);
          real comps[3] = {nv.x * mm, (fn + fd) + nv.y * mm, nv.z * mm};
          for (int kk = 0; kk < 3; ++kk) {
            real dd = (comps[kk] < 0 ? -comps[kk] : comps[kk]) - (real)500;
            if (dd < 0) dd = -dd;
            if (dd < fc_min) fc_min = dd;
          }
        }
      }
      for (int i = 0; i < nb; ++i) {
        v3 x0 = ld3(bq + i * 7); qt r0 = ld4(bq + i * 7 + 3);
        v3 w0 = ld3(bqd + i * 6), v0 = ld3(bqd + i * 6 + 3);
        v3 t0 = ld3(bf + i * 6), f0 = ld3(bf + i * 6 + 3);
        v3 g = ld3(t->gravity);
        real im = inv_mass[(size_t)e * nb + i]; real nz = im != (real)0 ? (real)1 : (real)0;
        (void)x0;
        v3 v1 = vadd(v0, vscale(vadd(vscale(f0, im), vscale(g, nz)), dt));
        v3 wb = qrot_inv(r0, w0);
        v3 Iwb; mat_vec(inertia + ((size_t)e * nb + i) * 9, wb, &Iwb);
        v3 tb = vsub(qrot_inv(r0, t0), vcross(wb, Iwb));
        v3 a; mat_vec(inv_inertia + ((size_t)e * nb + i) * 9, tb, &a);
        v3 w1 = vscale(qrot(r0, vadd(wb, vscale(a, dt))), (real)1 - (real)0.1 * dt);
        real comps[6] = {w1.x, w1.y, w1.z, v1.x, v1.y, v1.z};
        for (int k = 0; k < 6; ++k) {
          real d = (comps[k] < 0 ? -comps[k] : comps[k]) - (real)10;
          if (d < 0) d = -d;
          if (d < cl_min) cl_min = d;
        }
        if (t->joint_type[i] == JOINT_REVOLUTE) {
          int par = t->joint_parent[i];
          qt q_p = ld4(t->X_p + i * 7 + 3);
          if (par >= 0) q_p = qmul(ld4(bq + par * 7 + 3), q_p);
          qt r_err = qmul(qconj(q_p), r0);
          v3 ax = ld3(t->axis + i * 3);
          v3 av = vscale(ax, vdot(qv(r_err), ax));
          qt tw = qnormalize(Q(av.x, av.y, av.z, r_err.w));
          real d = (real)1 - (tw.w < 0 ? -tw.w : tw.w);
          if (d < ac_min) ac_min = d;
        }
      }
      real *o = out + ((size_t)s * bs + e) * 5;
      o[0] = c_min; o[1] = cl_min; o[2] = fr_min; o[3] = ac_min; o[4] = fc_min;
    }
  }
}

/* Branch log (test diagnostics): the discrete decisions of each step of a stored trajectory, per (step, env, body):
 *   touch[..]  number of the body's contact candidates with c <= 0 (integrator_euler.py:130-133)
 *   slide[..]  number of those whose friction is the kf |vt| branch of the min (:160-165)
 *   clamp[..]  6-bit mask of the velocity components that :78-88 clamps (w.x w.y w.z v.x v.y v.z), recomputed from
 *              (state, body_f) of the step
 * Two evaluations of a rollout that agree on all of these differentiate the same smooth function. */
void ref_branch_log(const RefTemplate *t, int bs, int nsteps, real dt, const real *states_q, const real *states_qd,
                    const real *states_f, const real *inv_mass, const real *inertia, const real *inv_inertia, int *touch, int *slide,
                    int *clamp) {
  const int nb = t->nb;
  const size_t SQ = (size_t)bs * nb * 7, SD = (size_t)bs * nb * 6;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < bs; ++e) {
    size_t oq = (size_t)e * nb * 7, od = (size_t)e * nb * 6;
    for (int s = 0; s < nsteps; ++s) {
      const real *bq = states_q + s * SQ + oq, *bqd = states_qd + s * SD + od, *bf = states_f + s * SD + od;
      int *to = touch + ((size_t)s * bs + e) * nb, *so = slide + ((size_t)s * bs + e) * nb, *co = clamp + ((size_t)s * bs + e) * nb;
      for (int i = 0; i < nb; ++i) { to[i] = 0; so[i] = 0; co[i] = 0; }
      for (int k = 0; k < t->nc; ++k) {
        int b = t->c_body[k];
        v3 p = ld3(bq + b * 7); qt q = ld4(bq + b * 7 + 3);
        v3 w = ld3(bqd + b * 6), v = ld3(bqd + b * 6 + 3);
        v3 cp = vadd(p, qrot(q, ld3(t->c_point + k * 3)));
        real c = cp.y - t->c_dist[k];
        if (c > (real)0) continue;
        to[b]++;
        v3 r = vsub(V(cp.x, c, cp.z), vadd(p, qrot(q, ld3(t->com + b * 3))));
        v3 dpdt = vadd(v, vcross(w, r));
        const real *mat = t->materials + t->c_mat[k] * 4;
        real vn = dpdt.y;
        v3 vt = V(dpdt.x, 0, dpdt.z);
        real fn = c * mat[0], fd = (vn < (real)0 ? vn : (real)0) * mat[1] * (c < (real)0 ? (real)1 : (real)0);
        if (mat[2] * vlen(vt) < (real)0 - mat[3] * (fn + fd)) so[b]++;
      }
      for (int i = 0; i < nb; ++i) {
        qt r0 = ld4(bq + i * 7 + 3);
        v3 w0 = ld3(bqd + i * 6), v0 = ld3(bqd + i * 6 + 3);
        v3 t0 = ld3(bf + i * 6), f0 = ld3(bf + i * 6 + 3);
        v3 g = ld3(t->gravity);
        real im = inv_mass[(size_t)e * nb + i]; real nz = im != (real)0 ? (real)1 : (real)0;
        v3 v1 = vadd(v0, vscale(vadd(vscale(f0, im), vscale(g, nz)), dt));
        v3 wb = qrot_inv(r0, w0);
        v3 Iwb; mat_vec(inertia + ((size_t)e * nb + i) * 9, wb, &Iwb);
        v3 tb = vsub(qrot_inv(r0, t0), vcross(wb, Iwb));
        v3 a; mat_vec(inv_inertia + ((size_t)e * nb + i) * 9, tb, &a);
        v3 w1 = vscale(qrot(r0, vadd(wb, vscale(a, dt))), (real)1 - (real)0.1 * dt);
        real comps[6] = {w1.x, w1.y, w1.z, v1.x, v1.y, v1.z};
        for (int k = 0; k < 6; ++k)
          if (comps[k] < (real)-10 || comps[k] > (real)10) co[i] |= 1 << k;
      }
    }
  }
}

void ref_rollout_forward(const RefTemplate *t, int bs, int nsteps, real dt, const real *q_init, const real *qd_init,
                         const real *torques, const real *res_f, const real *refs, const real *target_ke,
                         const real *target_kd, const real *inv_mass, const real *inertia, const real *inv_inertia,
                         int nframes, const int *frame2step, real *states_q, real *states_qd, real *states_f, real *wp_pos,
                         real *wp_vel, real *grf, real *jaf) {
  const int nb = t->nb, nq = t->nq, nqd = t->nqd;
  const size_t SQ = (size_t)bs * nb * 7, SD = (size_t)bs * nb * 6;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < bs; ++e) {
    size_t oq = (size_t)e * nb * 7, od = (size_t)e * nb * 6;
    fk_one(t, q_init + (size_t)e * nq, qd_init + (size_t)e * nqd, states_q + oq, states_qd + od);
    for (int s = 0; s < nsteps; ++s) {
      const real *bq = states_q + s * SQ + oq, *bqd = states_qd + s * SD + od;
      real *bf = states_f + s * SD + od;
      memcpy(bf, res_f + s * SD + od, sizeof(real) * nb * 6); /* clear_forces + wp_add */
      contacts_fwd(t, bq, bqd, bf);
      int fr = -1;
      for (int f = 0; f < nframes; ++f) if (frame2step[f] == s) fr = f;
      if (fr >= 0) {
        memcpy(wp_pos + fr * SQ + oq, bq, sizeof(real) * nb * 7);
        memcpy(wp_vel + fr * SD + od, bqd, sizeof(real) * nb * 6);
        if (grf) memcpy(grf + fr * SD + od, bf, sizeof(real) * nb * 6);
      }
      joints_fwd(t, bq, bqd, refs + (size_t)s * bs * nqd + (size_t)e * nqd, torques + (size_t)s * bs * nqd + (size_t)e * nqd,
                 target_ke + (size_t)e * nqd, target_kd + (size_t)e * nqd, bf);
      if (fr >= 0 && jaf && grf)
        for (int k = 0; k < nb * 6; ++k) jaf[fr * SD + od + k] = bf[k] - grf[fr * SD + od + k];
      integrate_fwd(t, bq, bqd, bf, inv_mass + (size_t)e * nb, inertia + (size_t)e * nb * 9, inv_inertia + (size_t)e * nb * 9,
                    dt, states_q + (s + 1) * SQ + oq, states_qd + (s + 1) * SD + od);
      if (g_round_states) {
        real *nq_ = states_q + (s + 1) * SQ + oq, *nd_ = states_qd + (s + 1) * SD + od;
        /* mode 1: round to nearest; mode 2: the fp32 neighbour on alternating sides of that (a second sample of the same size) */
        for (int k = 0; k < nb * 7; ++k) { float f = (float)nq_[k]; nq_[k] = (real)(g_round_states == 2 ? nextafterf(f, (k & 1) ? -INFINITY : INFINITY) : f); }
        for (int k = 0; k < nb * 6; ++k) { float f = (float)nd_[k]; nd_[k] = (real)(g_round_states == 2 ? nextafterf(f, (k & 1) ? INFINITY : -INFINITY) : f); }
      }
    }
    /* a frame may name state_steps[nsteps] (dp_model.py:396 allocates it, :1241-1246 would read it); no force snapshot
     * exists for that state (:1225-1228), the caller's zero-initialised grf / jaf rows stay zero */
    for (int f = 0; f < nframes; ++f)
      if (frame2step[f] == nsteps) {
        memcpy(wp_pos + f * SQ + oq, states_q + nsteps * SQ + oq, sizeof(real) * nb * 7);
        memcpy(wp_vel + f * SD + od, states_qd + nsteps * SD + od, sizeof(real) * nb * 6);
      }
  }
}

/* Reverse sweep.  All grad outputs must be zero-initialised by the caller; they are accumulated.
 * grad shapes mirror the inputs; g_mass is left untouched (body_mass is loaded but unused, integrator_euler.py:43). */
/* Forced variant (test diagnostics): the reverse sweep over a trajectory ANOTHER implementation saved (the HIP kernels' fp32
 * states and total wrenches, cast to this precision), taking the discrete decisions that implementation recorded instead of
 * re-deciding them here on rounded states:
 *   clamp_mask [nsteps][bs*nb]  6-bit velocity-clamp masks of each step's integration (integrator_euler.py:78-88) as that
 *                               implementation stored them, or NULL (re-decide here)
 *   pinned_touch != 0           "this candidate touches" (c <= 0, :130-133) is decided by touch_pinned_fp32 on the stored state
 *   touch_list [nsteps][bs][touch_cap]  or NULL: per env-step [0] = n, then the n candidates that touch (what ref_touch_fp32
 *                               returns); takes precedence where n < touch_cap, i.e. where the list is complete
 * What these leave undecided (the Coulomb min of :160-165, the +-500 N force clamp of :172-175) is re-decided here in this
 * precision; ref_singularity_probe reports how close to those switches a step sits.  With NULL / 0 this IS ref_rollout_backward. */
static void rollout_backward_impl(const RefTemplate *t, int bs, int nsteps, real dt, const real *q_init, const real *qd_init,
                          const real *torques, const real *refs, const real *target_ke, const real *target_kd,
                          const real *inv_mass, const real *inertia, const real *inv_inertia, int nframes,
                          const int *frame2step, const real *states_q, const real *states_qd, const real *states_f,
                          const real *adj_pos, const real *adj_vel, real *g_q_init, real *g_qd_init, real *g_torques,
                          real *g_res_f, real *g_refs, real *g_ke, real *g_kd, real *g_inv_mass, real *g_inertia,
                          real *g_inv_inertia, const int *clamp_mask, int pinned_touch, const int *touch_list, int touch_cap) {
  const int nb = t->nb, nq = t->nq, nqd = t->nqd;
  const size_t SQ = (size_t)bs * nb * 7, SD = (size_t)bs * nb * 6;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < bs; ++e) {
    size_t oq = (size_t)e * nb * 7, od = (size_t)e * nb * 6;
    real *aq_next = (real *)calloc((size_t)nb * 7, sizeof(real));
    real *aqd_next = (real *)calloc((size_t)nb * 6, sizeof(real));
    real *aq = (real *)calloc((size_t)nb * 7, sizeof(real));
    real *aqd = (real *)calloc((size_t)nb * 6, sizeof(real));
    real *af = (real *)calloc((size_t)nb * 6, sizeof(real));
    for (int s = nsteps; s >= 0; --s) {
      /* seeds written into state s's grads (dp_model.py:1264-1271) */
      for (int f = 0; f < nframes; ++f)
        if (frame2step[f] == s) {
          for (int k = 0; k < nb * 7; ++k) aq_next[k] += adj_pos[f * SQ + oq + k];
          for (int k = 0; k < nb * 6; ++k) aqd_next[k] += adj_vel[f * SD + od + k];
        }
      if (s == 0) break;
      int st = s - 1; /* adjoint of step st: state st -> state st+1 */
      const real *bq = states_q + st * SQ + oq, *bqd = states_qd + st * SD + od, *bf = states_f + st * SD + od;
      memset(aq, 0, sizeof(real) * nb * 7); memset(aqd, 0, sizeof(real) * nb * 6); memset(af, 0, sizeof(real) * nb * 6);
      integrate_adj(t, bq, bqd, bf, inv_mass + (size_t)e * nb, inertia + (size_t)e * nb * 9, inv_inertia + (size_t)e * nb * 9,
                    dt, aq_next, aqd_next, aq, aqd, af, g_inv_mass + (size_t)e * nb, g_inertia + (size_t)e * nb * 9,
                    g_inv_inertia + (size_t)e * nb * 9, clamp_mask ? clamp_mask + ((size_t)st * bs + e) * nb : NULL);
      size_t oc = (size_t)st * bs * nqd + (size_t)e * nqd;
      joints_adj(t, bq, bqd, refs + oc, torques + oc, target_ke + (size_t)e * nqd, target_kd + (size_t)e * nqd, af, aq, aqd,
                 g_refs + oc, g_torques + oc, g_ke + (size_t)e * nqd, g_kd + (size_t)e * nqd);
      contacts_adj(t, bq, bqd, af, aq, aqd, pinned_touch, (touch_list && touch_list[((size_t)st * bs + e) * touch_cap] < touch_cap) ? touch_list + ((size_t)st * bs + e) * touch_cap : NULL);
      for (int k = 0; k < nb * 6; ++k) g_res_f[st * SD + od + k] += af[k]; /* adjoint of wp_add */
      real *tmp;
      tmp = aq_next; aq_next = aq; aq = tmp;
      tmp = aqd_next; aqd_next = aqd; aqd = tmp;
    }
    fk_one_adj(t, q_init + (size_t)e * nq, qd_init + (size_t)e * nqd, states_q + oq, aq_next, aqd_next,
               g_q_init + (size_t)e * nq, g_qd_init + (size_t)e * nqd);
    free(aq_next); free(aqd_next); free(aq); free(aqd); free(af);
  }
}

void ref_rollout_backward(const RefTemplate *t, int bs, int nsteps, real dt, const real *q_init, const real *qd_init,
                          const real *torques, const real *refs, const real *target_ke, const real *target_kd,
                          const real *inv_mass, const real *inertia, const real *inv_inertia, int nframes,
                          const int *frame2step, const real *states_q, const real *states_qd, const real *states_f,
                          const real *adj_pos, const real *adj_vel, real *g_q_init, real *g_qd_init, real *g_torques,
                          real *g_res_f, real *g_refs, real *g_ke, real *g_kd, real *g_inv_mass, real *g_inertia,
                          real *g_inv_inertia) {
  rollout_backward_impl(t, bs, nsteps, dt, q_init, qd_init, torques, refs, target_ke, target_kd, inv_mass, inertia, inv_inertia, nframes,
                        frame2step, states_q, states_qd, states_f, adj_pos, adj_vel, g_q_init, g_qd_init, g_torques, g_res_f, g_refs,
                        g_ke, g_kd, g_inv_mass, g_inertia, g_inv_inertia, NULL, 0, NULL, 0);
}
void ref_rollout_backward_forced(const RefTemplate *t, int bs, int nsteps, real dt, const real *q_init, const real *qd_init,
                          const real *torques, const real *refs, const real *target_ke, const real *target_kd,
                          const real *inv_mass, const real *inertia, const real *inv_inertia, int nframes,
                          const int *frame2step, const real *states_q, const real *states_qd, const real *states_f,
                          const real *adj_pos, const real *adj_vel, real *g_q_init, real *g_qd_init, real *g_torques,
                          real *g_res_f, real *g_refs, real *g_ke, real *g_kd, real *g_inv_mass, real *g_inertia,
                          real *g_inv_inertia, const int *clamp_mask, int pinned_touch, const int *touch_list, int touch_cap) {
  rollout_backward_impl(t, bs, nsteps, dt, q_init, qd_init, torques, refs, target_ke, target_kd, inv_mass, inertia, inv_inertia, nframes,
                        frame2step, states_q, states_qd, states_f, adj_pos, adj_vel, g_q_init, g_qd_init, g_torques, g_res_f, g_refs,
                        g_ke, g_kd, g_inv_mass, g_inertia, g_inv_inertia, clamp_mask, pinned_touch, touch_list, touch_cap);
}

/* Per (step, env): the template indices of the candidates touch_pinned_fp32 says touch, ascending; out [nsteps][bs][cap]
 * ([0] = count, then up to cap - 1 indices, -1 padded).  tests cross-check it against the hit log the kernels wrote. */
void ref_touch_fp32(const RefTemplate *t, int bs, int nsteps, const real *states_q, int *out, int cap) {
  const int nb = t->nb;
  const size_t SQ = (size_t)bs * nb * 7;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < bs; ++e)
    for (int s = 0; s < nsteps; ++s) {
      int *o = out + ((size_t)s * bs + e) * cap, n = 0;
      for (int k = 1; k < cap; ++k) o[k] = -1;
      for (int k = 0; k < t->nc; ++k)
        if (touch_pinned_fp32(t, k, states_q + s * SQ + (size_t)e * nb * 7)) { if (1 + n < cap) o[1 + n] = k; ++n; }
      o[0] = n;
    }
}

/* Batched FK for ForwardKinematics (dp_model.py:1022-1130): n independent articulations. */
void ref_fk_forward(const RefTemplate *t, int n, const real *joint_q, const real *joint_qd, real *body_q, real *body_qd) {
#pragma omp parallel for schedule(static)
  for (int e = 0; e < n; ++e)
    fk_one(t, joint_q + (size_t)e * t->nq, joint_qd + (size_t)e * t->nqd, body_q + (size_t)e * t->nb * 7,
           body_qd + (size_t)e * t->nb * 6);
}
void ref_fk_backward(const RefTemplate *t, int n, const real *joint_q, const real *joint_qd, const real *body_q,
                     const real *adj_body_q, const real *adj_body_qd, real *g_joint_q, real *g_joint_qd) {
#pragma omp parallel for schedule(static)
  for (int e = 0; e < n; ++e) {
    real *aq = (real *)malloc(sizeof(real) * t->nb * 7), *aqd = (real *)malloc(sizeof(real) * t->nb * 6);
    memcpy(aq, adj_body_q + (size_t)e * t->nb * 7, sizeof(real) * t->nb * 7);
    memcpy(aqd, adj_body_qd + (size_t)e * t->nb * 6, sizeof(real) * t->nb * 6);
    fk_one_adj(t, joint_q + (size_t)e * t->nq, joint_qd + (size_t)e * t->nqd, body_q + (size_t)e * t->nb * 7, aq, aqd,
               g_joint_q + (size_t)e * t->nq, g_joint_qd + (size_t)e * t->nqd);
    free(aq); free(aqd);
  }
}
