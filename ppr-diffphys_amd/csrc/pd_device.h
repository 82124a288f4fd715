// Device-side model view + the per-lane physics of one sim step and its adjoint.
//
// Mapping: a 64-lane wavefront is cut into 64/SEGW segments; one
// articulation (env) per segment, lane l of a segment owns body l (l < nb) for the per-body
// passes (joints, integration) and is one of SEGW sweep lanes for the per-contact-point pass.
// Body state lives in registers for the whole rollout; the kinematic chain (what a child needs
// from its parent, what a parent receives from its children) is staged through LDS records
// private to the segment, so no cross-wave synchronisation is ever needed.
//
// What is computed follows /root/reference/diffphys/integrator_euler.py (cited per function).
#pragma once
#include "pd_math.h"

enum { PD_JOINT_REVOLUTE = 1, PD_JOINT_FIXED = 3, PD_JOINT_FREE = 4, PD_JOINT_COMPOUND = 5 };
enum { PD_JT_REVOLUTE = 1, PD_JT_COMPOUND = 2, PD_JT_FIXED = 4 };  // template mask bits

// LDS strides are odd so that lanes (= bodies) hit distinct banks with 4-byte accesses.
#define PD_REC 17  // floats per staged body record: p[0:3] q[3:7] w[7:10] v[10:13] rc[13:16]
#define PD_ADJ 13  // floats of a body-state adjoint: p q w v
#define PD_W6 7    // stride of a 6-float wrench slot

struct PdDevModel {
  int nb, nq, nqd, nc, ntiles, max_children, max_depth;
  const int *jtype, *jparent, *qstart, *qdstart, *depth;  // [nb]
  const unsigned long long *children;                     // [nb] 8 child ids packed, 0xff = none
  const float *X_p, *X_c, *axis, *com;                    // [nb*7] [nb*7] [nb*3] [nb*3]
  const float *lim_lo, *lim_hi, *lim_ke, *lim_kd;         // [nqd]
  // ground-contact candidates, grouped by body and cut into tiles of <= SEGW spatially compact points
  const float4 *pts;                                      // [nc] (x,y,z,dist)
  const unsigned char *pt_mat;                            // [nc padded to 16] material index of each point
  const float4 *materials;                                // [nmat] (ke,kd,kf,mu)
  const float4 *body_sphere;                              // [nb] bounding sphere of the body's points (w < 0: none)
  const float4 *tile_lo, *tile_hi;                        // [ntiles] body-frame AABB: (lo.xyz, max dist), (hi.xyz, safety margin)
  const int *tile_pack;                                   // [ntiles] first point | count << 16 | body << 24
  const int2 *body_tiles;                                 // [nb] (first tile, tile count)
  const int *small_tiles;                                 // [4*64] flat list (tile | body << 16) of the tiles of small bodies, -1 padded
  unsigned long long big_bodies;                          // bodies whose tiles are NOT in small_tiles
  int nmat, n_small;
  int list_cap;                                            // ints reserved for the tile list (>= ntiles and >= 2*nb)
  int has_limits;                                          // any joint_limit_ke / kd != 0 (else the limit force is identically 0)
  float gx, gy, gz, attach_ke, attach_kd;
  float spec_safety, spec_slack;                          // speculative contact cull: allowed sinking per step (pd_kernels.hip sink_margin)
  int env_lds_floats;                                     // per-env LDS scratch
  int cu_count;                                           // compute units of the device (launch heuristics)
  int env_lds_jc;                                         // + joint hand-over records (2-role wave-specialised adjoint only)
  int env_lds_bwd3;                                       // per-env LDS scratch of the 3-role adjoint kernel (k_rollout_bwd3)
  int env_lds_rec2;                                       // quad-lane adjoint (64-lane copy of an eligible model, else 0): PD_QGEN generations of cull vectors + records and of the state-only hand-over
  const float *X_p_env;                                   // [xp_envs][nb][7] per-env joint_X_p bound by the caller, or null (template X_p)
  int xp_envs;
};

#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

struct BodyState { v3 p; qt r; v3 w; v3 v; };
struct BodyAdj { v3 p; qt r; v3 w; v3 v; };

PD_DEV BodyAdj adj_zero() {
  BodyAdj a;
  a.p = V3(0, 0, 0); a.r = Q4(0, 0, 0, 0); a.w = V3(0, 0, 0); a.v = V3(0, 0, 0);
  return a;
}
PD_DEV void adj_store(float *s, const BodyAdj &a) {
  s[0] = a.p.x; s[1] = a.p.y; s[2] = a.p.z; s[3] = a.r.x; s[4] = a.r.y; s[5] = a.r.z; s[6] = a.r.w;
  s[7] = a.w.x; s[8] = a.w.y; s[9] = a.w.z; s[10] = a.v.x; s[11] = a.v.y; s[12] = a.v.z;
}
PD_DEV void adj_add_from(BodyAdj &a, const float *s) {
  a.p.x += s[0]; a.p.y += s[1]; a.p.z += s[2]; a.r.x += s[3]; a.r.y += s[4]; a.r.z += s[5]; a.r.w += s[6];
  a.w.x += s[7]; a.w.y += s[8]; a.w.z += s[9]; a.v.x += s[10]; a.v.y += s[11]; a.v.z += s[12];
}

// adj_add_from for N records in a row, same sums in the same order, in packed fp32 (v_pk_add_f32: two sums per issue slot at the
// price of one -- the adjoint kernels are bound by VALU issue, one wave64 instruction per ~4.2 cycles and SIMD):
// 7 N instead of 13 N adds, all LDS reads in flight together.
typedef float pd_f2 __attribute__((ext_vector_type(2)));
template <int N>
PD_DEV void adj_add_from_n(BodyAdj &a, const float *const (&s)[N]) {
  pd_f2 c[N][6];
  float cz[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int j = 0; j < 6; ++j) { c[k][j].x = s[k][2 * j]; c[k][j].y = s[k][2 * j + 1]; }
    cz[k] = s[k][12];
  }
  pd_f2 acc[6];
  acc[0].x = a.p.x; acc[0].y = a.p.y; acc[1].x = a.p.z; acc[1].y = a.r.x; acc[2].x = a.r.y; acc[2].y = a.r.z;
  acc[3].x = a.r.w; acc[3].y = a.w.x; acc[4].x = a.w.y; acc[4].y = a.w.z; acc[5].x = a.v.x; acc[5].y = a.v.y;
  float az = a.v.z;
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] += c[k][j];
    az += cz[k];
  }
  a.p.x = acc[0].x; a.p.y = acc[0].y; a.p.z = acc[1].x; a.r.x = acc[1].y; a.r.y = acc[2].x; a.r.z = acc[2].y;
  a.r.w = acc[3].x; a.w.x = acc[3].y; a.w.y = acc[4].x; a.w.z = acc[4].y; a.v.x = acc[5].x; a.v.y = acc[5].y;
  a.v.z = az;
}
// the forward pass's counterpart: N six-float wrenches added to (t, f) in order, 3 N packed adds instead of 6 N
template <int N>
PD_DEV void wrench_add_from_n(v3 &t, v3 &f, const float *const (&s)[N]) {
  pd_f2 c[N][3], acc[3];
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { c[k][j].x = s[k][2 * j]; c[k][j].y = s[k][2 * j + 1]; }
  }
  acc[0].x = t.x; acc[0].y = t.y; acc[1].x = t.z; acc[1].y = f.x; acc[2].x = f.y; acc[2].y = f.z;
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] += c[k][j];
  }
  t = V3(acc[0].x, acc[0].y, acc[1].x); f = V3(acc[1].y, acc[2].x, acc[2].y);
}

// ... and with the own joint's wrench SUBTRACTED first: ((((c0 - w) + c1) + c2) + ...) = ((((-w) + c0) + c1) + ...) bit for bit, without the six
// sign flips that forming -w costs (the packed add takes the negation as an operand modifier)
template <int N>
PD_DEV void wrench_sub_add_from_n(v3 &t, v3 &f, v3 wt, v3 wf, const float *const (&s)[N]) {
  pd_f2 c[N][3], acc[3], w[3];
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { c[k][j].x = s[k][2 * j]; c[k][j].y = s[k][2 * j + 1]; }
  }
  w[0].x = wt.x; w[0].y = wt.y; w[1].x = wt.z; w[1].y = wf.x; w[2].x = wf.y; w[2].y = wf.z;
#pragma unroll
  for (int j = 0; j < 3; ++j) acc[j] = c[0][j] - w[j];
#pragma unroll
  for (int k = 1; k < N; ++k) {
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j] += c[k][j];
  }
  t = V3(acc[0].x, acc[0].y, acc[1].x); f = V3(acc[1].y, acc[2].x, acc[2].y);
}

PD_DEV v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
PD_DEV qt ld4(const float *p) { return Q4(p[0], p[1], p[2], p[3]); }

// Per-lane constants of body l (registers for the whole rollout).
struct JointLimit { float lo, up, ke, kd; };
struct BodyConst {
  int type, parent, qstart, qdstart, depth;
  unsigned long long children;
  v3 com, axis, p_pj, com_par;
  qt q_pj, q_off;
  float4 sphere;  // bounding sphere of this body's contact candidates
  float reach;    // >= distance of any contact candidate from the centre of mass (0: no candidates)
  JointLimit lim[3];  // joint limits of this body's (up to three) joint dofs: loop-invariant, loaded once
  int tile_first, tile_count;
  int small_e[4];  // this lane's entries of the small-body tile list (chunk u: entry u*SEGW + lane)
  int child[4];    // first four children (-1 = none); the rest, if any, are walked from `children`
  int pidx;        // parent, or 0 for a body without one (a valid record index for unguarded reads)
  float alen;      // |axis| (1 for every URDF-imported joint)
};

// env: the articulation this lane works for (only read when a per-env joint_X_p is bound: dp_interface.py:465 of the reference)
PD_DEV BodyConst load_body_const(const PdDevModel &m, int b, int env) {
  BodyConst c;
  c.type = m.jtype[b]; c.parent = m.jparent[b]; c.qstart = m.qstart[b]; c.qdstart = m.qdstart[b];
  c.depth = m.depth[b]; c.children = m.children[b];
  c.com = ld3(m.com + b * 3); c.axis = ld3(m.axis + b * 3);
  const float *xp = m.X_p_env ? m.X_p_env + ((size_t)(env % m.xp_envs) * m.nb + b) * 7 : m.X_p + b * 7;
  c.p_pj = ld3(xp); c.q_pj = ld4(xp + 3); c.q_off = ld4(m.X_c + b * 7 + 3);
  c.com_par = c.parent >= 0 ? ld3(m.com + c.parent * 3) : V3(0, 0, 0);
  c.pidx = c.parent >= 0 ? c.parent : 0;
  c.alen = length(c.axis);
  c.sphere = m.body_sphere[b];
  c.reach = c.sphere.w >= 0.0f ? length(V3(c.sphere.x, c.sphere.y, c.sphere.z) - c.com) + c.sphere.w : 0.0f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int cid = (int)((c.children >> (8 * k)) & 0xffull);
    c.child[k] = cid == 0xff ? -1 : cid;
  }
  c.tile_first = m.body_tiles[b].x; c.tile_count = m.body_tiles[b].y;
#pragma unroll
  for (int k = 0; k < 3; ++k) {  // identically-zero limit forces (no joint has limit gains) cost no loads at all
    JointLimit L;
    L.lo = -1.0e30f; L.up = 1.0e30f; L.ke = 0.f; L.kd = 0.f;
    if (m.has_limits) {
      const int dof = c.qdstart + k < m.nqd ? c.qdstart + k : m.nqd - 1;
      L.lo = m.lim_lo[dof]; L.up = m.lim_hi[dof]; L.ke = m.lim_ke[dof]; L.kd = m.lim_kd[dof];
    }
    c.lim[k] = L;
  }
  return c;
}

// Returns the cull vector (p_y, Ry): Ry = second row of R(q), so the world height of a body-frame point x is
// p_y + Ry . x.  It is also stored 16-byte aligned in cull[b] so the sweeps fetch it with one ds_read_b128.
// Rm = rotm(s.r): its row 1 (pd_math.h rot_row1, pinned roundings) is what the exact contact test computes heights from, in the
// forward pass and in the adjoint alike.
PD_DEV float4 stage_record(float *rec, float4 *cull, int b, const BodyState &s, v3 rc, const float *Rm) {
  float *r = rec + b * PD_REC;
  r[0] = s.p.x; r[1] = s.p.y; r[2] = s.p.z; r[3] = s.r.x; r[4] = s.r.y; r[5] = s.r.z; r[6] = s.r.w;
  r[7] = s.w.x; r[8] = s.w.y; r[9] = s.w.z; r[10] = s.v.x; r[11] = s.v.y; r[12] = s.v.z;
  r[13] = rc.x; r[14] = rc.y; r[15] = rc.z;
  float4 cv = make_float4(s.p.y, Rm[3], Rm[4], Rm[5]);
  cull[b] = cv;
  return cv;
}
PD_DEV void stage_record(float *rec, int b, const BodyState &s, v3 com) {  // FK-only kernels: no cull vector
  float *r = rec + b * PD_REC;
  v3 rc = qrot(s.r, com);
  r[0] = s.p.x; r[1] = s.p.y; r[2] = s.p.z; r[3] = s.r.x; r[4] = s.r.y; r[5] = s.r.z; r[6] = s.r.w;
  r[7] = s.w.x; r[8] = s.w.y; r[9] = s.w.z; r[10] = s.v.x; r[11] = s.v.y; r[12] = s.v.z;
  r[13] = rc.x; r[14] = rc.y; r[15] = rc.z;
}

// ---------------------------------------------------------------------------------------------
// Forward kinematics of one joint (warp.sim.articulation.eval_fk).
// jq / jqd point at this joint's coordinates; parent state comes from the LDS record.
struct FkLocals {
  v3 p_jc, w_jc, v_jc, a0, a1, a2;
  qt q_jc, q0, q1, q2, q_wj;
};

template <int JT>
PD_DEV void fk_joint_local(const BodyConst &c, const float *jq, const float *jqd, FkLocals &L) {
  L.p_jc = V3(0, 0, 0); L.w_jc = V3(0, 0, 0); L.v_jc = V3(0, 0, 0); L.q_jc = Q4(0, 0, 0, 1);
  L.a0 = L.a1 = L.a2 = V3(0, 0, 0); L.q0 = L.q1 = L.q2 = Q4(0, 0, 0, 1);
  if ((JT & PD_JT_REVOLUTE) && c.type == PD_JOINT_REVOLUTE) {
    L.q_jc = q_axis_angle(c.axis, jq[0]);
    L.w_jc = c.axis * jqd[0];
  } else if (c.type == PD_JOINT_FREE) {
    L.p_jc = ld3(jq); L.q_jc = ld4(jq + 3); L.w_jc = ld3(jqd); L.v_jc = ld3(jqd + 3);
  } else if ((JT & PD_JT_COMPOUND) && c.type == PD_JOINT_COMPOUND) {
    L.a0 = qrot(c.q_off, V3(1, 0, 0));
    L.q0 = q_axis_angle(L.a0, jq[0]);
    L.a1 = qrot(qmul(L.q0, c.q_off), V3(0, 1, 0));
    L.q1 = q_axis_angle(L.a1, jq[1]);
    L.a2 = qrot(qmul(L.q1, qmul(L.q0, c.q_off)), V3(0, 0, 1));
    L.q2 = q_axis_angle(L.a2, jq[2]);
    L.q_jc = qmul(L.q2, qmul(L.q1, L.q0));
    L.w_jc = L.a0 * jqd[0] + L.a1 * jqd[1] + L.a2 * jqd[2];
  }
}

template <int JT>
PD_DEV BodyState fk_joint(const BodyConst &c, const float *jq, const float *jqd, const float *rec) {
  v3 p_wp = V3(0, 0, 0), w_wp = V3(0, 0, 0), v_wp = V3(0, 0, 0);
  qt q_wp = Q4(0, 0, 0, 1);
  if (c.parent >= 0) {
    const float *r = rec + c.parent * PD_REC;
    p_wp = ld3(r); q_wp = ld4(r + 3); w_wp = ld3(r + 7); v_wp = ld3(r + 10);
  }
  FkLocals L;
  fk_joint_local<JT>(c, jq, jqd, L);
  v3 p_wj = p_wp + qrot(q_wp, c.p_pj);
  qt q_wj = qmul(q_wp, c.q_pj);
  BodyState s;
  s.p = p_wj + qrot(q_wj, L.p_jc);
  s.r = qmul(q_wj, L.q_jc);
  v3 ang = qrot(q_wj, L.w_jc), lin = qrot(q_wj, L.v_jc);
  s.w = w_wp + ang;
  s.v = v_wp + (lin + cross(ang, c.com));
  return s;
}

// Adjoint of fk_joint: g = adjoint of this body's (p,q,w,v); writes the joint-coordinate
// gradients (overwrite) and returns the contribution to the parent's state adjoint.
// POST: post-processing of the stored joint-coordinate gradients (pd_math.h grad_post).
template <int JT, int POST>
PD_DEV BodyAdj fk_joint_adj(const BodyConst &c, const float *jq, const float *jqd, const float *rec, const BodyAdj &g,
                            float *gq, float *gqd) {
  auto P = [](float x) { return grad_post<POST>(x); };
  v3 p_wp = V3(0, 0, 0);
  qt q_wp = Q4(0, 0, 0, 1);
  if (c.parent >= 0) {
    const float *r = rec + c.parent * PD_REC;
    p_wp = ld3(r); q_wp = ld4(r + 3);
  }
  (void)p_wp;
  FkLocals L;
  fk_joint_local<JT>(c, jq, jqd, L);
  qt q_wj = qmul(q_wp, c.q_pj);
  BodyAdj par = adj_zero();
  par.w = g.w; par.v = g.v;
  v3 adj_ang = g.w, adj_lin = g.v;
  adj_cross_a(c.com, adj_ang, g.v);
  qt adj_q_wj = Q4(0, 0, 0, 0);
  v3 adj_w_jc = V3(0, 0, 0), adj_v_jc = V3(0, 0, 0), adj_p_jc = V3(0, 0, 0);
  adj_qrot(q_wj, L.v_jc, adj_q_wj, adj_v_jc, adj_lin);
  adj_qrot(q_wj, L.w_jc, adj_q_wj, adj_w_jc, adj_ang);
  qt adj_q_jc = Q4(0, 0, 0, 0);
  adj_qmul(q_wj, L.q_jc, adj_q_wj, adj_q_jc, g.r);
  adj_qrot(q_wj, L.p_jc, adj_q_wj, adj_p_jc, g.p);
  adj_qmul_a(c.q_pj, par.r, adj_q_wj);
  par.p = g.p;
  adj_qrot_q(q_wp, c.p_pj, par.r, g.p);
  if ((JT & PD_JT_REVOLUTE) && c.type == PD_JOINT_REVOLUTE) {
    float a = 0.f;
    adj_q_axis_angle_ang(c.axis, jq[0], a, adj_q_jc);
    gq[0] = P(a);
    gqd[0] = P(dot(c.axis, adj_w_jc));
  } else if (c.type == PD_JOINT_FREE) {
    gq[0] = P(adj_p_jc.x); gq[1] = P(adj_p_jc.y); gq[2] = P(adj_p_jc.z);
    gq[3] = P(adj_q_jc.x); gq[4] = P(adj_q_jc.y); gq[5] = P(adj_q_jc.z); gq[6] = P(adj_q_jc.w);
    gqd[0] = P(adj_w_jc.x); gqd[1] = P(adj_w_jc.y); gqd[2] = P(adj_w_jc.z);
    gqd[3] = P(adj_v_jc.x); gqd[4] = P(adj_v_jc.y); gqd[5] = P(adj_v_jc.z);
  } else if ((JT & PD_JT_COMPOUND) && c.type == PD_JOINT_COMPOUND) {
    float aq0 = 0.f, aq1 = 0.f, aq2 = 0.f;
    v3 adj_a0 = adj_w_jc * jqd[0], adj_a1 = adj_w_jc * jqd[1], adj_a2 = adj_w_jc * jqd[2];
    gqd[0] = P(dot(L.a0, adj_w_jc)); gqd[1] = P(dot(L.a1, adj_w_jc)); gqd[2] = P(dot(L.a2, adj_w_jc));
    qt q10 = qmul(L.q1, L.q0);
    qt adj_q2 = Q4(0, 0, 0, 0), adj_q10 = adj_q2, adj_q1 = adj_q2, adj_q0 = adj_q2;
    adj_qmul(L.q2, q10, adj_q2, adj_q10, adj_q_jc);
    adj_q_axis_angle(L.a2, jq[2], adj_a2, aq2, adj_q2);
    qt q0o = qmul(L.q0, c.q_off), q10o = qmul(L.q1, q0o);
    qt adj_q10o = Q4(0, 0, 0, 0), adj_q0o = adj_q10o;
    adj_qrot_q(q10o, V3(0, 0, 1), adj_q10o, adj_a2);
    adj_qmul(L.q1, q0o, adj_q1, adj_q0o, adj_q10o);
    adj_qmul(L.q1, L.q0, adj_q1, adj_q0, adj_q10);
    adj_q_axis_angle(L.a1, jq[1], adj_a1, aq1, adj_q1);
    adj_qrot_q(q0o, V3(0, 1, 0), adj_q0o, adj_a1);
    adj_qmul_a(c.q_off, adj_q0, adj_q0o);
    adj_q_axis_angle(L.a0, jq[0], adj_a0, aq0, adj_q0);
    gq[0] = P(aq0); gq[1] = P(aq1); gq[2] = P(aq2);
  }
  return par;
}

// ---------------------------------------------------------------------------------------------
// One ground-contact candidate (integrator_euler.py:93-179).  rec = staged record of its body.
// Returns false when the point is above ground (the kernel's early return, :132-133).
struct ContactOut { v3 t, f; };

// r = the body's staged record, cv = its cull vector (p_y, row 1 of rotm(q)): the height -- and with it "touching" -- comes from
// contact_height (pinned roundings, the adjoint recomputes the same bits), x and z from the reference's quaternion rotation.
// Returns whether the point touches (c <= 0 or NaN: the reference continues past `if c > 0: return` for a NaN too); o is
// computed either way.
PD_DEV bool contact_point_fwd(const float *r, float4 cv, float4 P, float4 mat, ContactOut &o) {
  const v3 p = ld3(r), w = ld3(r + 7), v = ld3(r + 10), rc = ld3(r + 13);
  const qt q = ld4(r + 3);
  const float c = contact_height(cv, P);
  const v3 rp = qrot(q, V3(P.x, P.y, P.z));
  const v3 cp = V3(p.x + rp.x, c, p.z + rp.z);
  v3 rr = cp - (p + rc);
  v3 dpdt = v + cross(w, rr);
  float ke = mat.x, kd = mat.y, kf = mat.z, mu = mat.w;
  float vn = dpdt.y;
  v3 vt = V3(dpdt.x, dpdt.y - vn, dpdt.z);
  float fn = c * ke;
  float fd = fminf(vn, 0.0f) * kd * (c < 0.0f ? 1.0f : 0.0f);
  float a_ = kf * length(vt), b_ = 0.0f - mu * (fn + fd);
  v3 ft = normalize(vt) * (a_ < b_ ? a_ : b_);
  v3 f = clamp3(V3(ft.x, (fn + fd) + ft.y, ft.z), 500.0f);
  o.f = f;
  o.t = cross(rr, f);
  return !(c > 0.0f);
}

// Adjoint: g_t, g_f = adjoint of the body's wrench accumulator; returns the contribution to (p,q,w,v).
// MAT: the four rotations by the body's quaternion (two forward, two adjoint) through one matrix and one matrix adjoint
// (pd_math.h: rotm).  Measured: -2.3 % adjoint time where the contacts run inline on the integrate wave (quad 8192), +3 % on the
// revolute kernel's contact wave (Laikago 4096) -- so the caller chooses.
// cv = the body's cull vector as staged (p_y, row 1 of rotm(q)): the height, and with it "touching", is the forward pass's bit for bit.
template <bool MAT = false>
PD_DEV bool contact_point_adj(const float *r, float4 cv, float4 P, float4 mat, v3 g_t, v3 g_f, BodyAdj &out) {
  v3 p = ld3(r), w = ld3(r + 7), v = ld3(r + 10), rc = ld3(r + 13);
  qt q = ld4(r + 3);
  float M[9];
  v3 cpt = V3(P.x, P.y, P.z), com, cp;
  if (MAT) {
    rotm(q, M);
    com = matT_vec(M, rc);
    cp = (p + mat_vec(M, cpt)) - V3(0.f, P.w, 0.f);
  } else {
    com = qrot_inv(q, rc);  // body-frame COM back from the staged rc = rot(q, com): no table read on the hit path
    cp = (p + qrot(q, cpt)) - V3(0.f, P.w, 0.f);
  }
  cp.y = contact_height(cv, P);  // the height decides "touching": exactly the forward pass's arithmetic
  float c = cp.y;
  if (c > 0.0f) return false;
  v3 rr = cp - (p + rc);
  v3 dpdt = v + cross(w, rr);
  float ke = mat.x, kd = mat.y, kf = mat.z, mu = mat.w;
  float vn = dpdt.y;
  v3 vt = V3(dpdt.x, dpdt.y - vn, dpdt.z);
  float fn = c * ke;
  float stepc = c < 0.0f ? 1.0f : 0.0f;
  float fd = fminf(vn, 0.0f) * kd * stepc;
  float lvt = length(vt);
  v3 nvt = normalize(vt);
  float a_ = kf * lvt, b_ = 0.0f - mu * (fn + fd);
  float mm = a_ < b_ ? a_ : b_;
  v3 f_raw = V3(nvt.x * mm, (fn + fd) + nvt.y * mm, nvt.z * mm);
  v3 fc = clamp3(f_raw, 500.0f);
  // reverse (body_f -= (t, f))
  v3 adj_t = -g_t, adj_fc = -g_f, adj_r = V3(0, 0, 0);
  adj_cross(rr, fc, adj_r, adj_fc, adj_t);
  v3 adj_fr = clamp3_pass(f_raw, adj_fc, 500.0f);
  float adj_fnfd = adj_fr.y;
  v3 adj_nvt = adj_fr * mm;
  float adj_m = dot(adj_fr, nvt);
  float adj_lvt = 0.f;
  if (a_ < b_) adj_lvt = adj_m * kf; else adj_fnfd += -mu * adj_m;
  v3 adj_vt = V3(0, 0, 0);
  adj_normalize(vt, adj_vt, adj_nvt);
  adj_length(vt, adj_vt, adj_lvt);
  float adj_c = adj_fnfd * ke;
  float adj_vn = (vn < 0.0f ? 1.0f : 0.0f) * kd * stepc * adj_fnfd;
  v3 adj_dpdt = adj_vt;
  adj_vn += -adj_vt.y;
  adj_dpdt.y += adj_vn;
  v3 adj_w = V3(0, 0, 0);
  adj_cross(w, rr, adj_w, adj_r, adj_dpdt);
  v3 adj_cp = V3(adj_r.x, adj_r.y + adj_c, adj_r.z);
  qt adj_q = Q4(0, 0, 0, 0);
  if (MAT) {
    float aM[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) aM[k] = 0.f;
    add_outer(aM, -adj_r, com);
    add_outer(aM, adj_cp, cpt);
    rotm_adj(q, aM, adj_q);
  } else {
    adj_qrot_q(q, com, adj_q, -adj_r);
    adj_qrot_q(q, cpt, adj_q, adj_cp);
  }
  out.p = adj_cp - adj_r;
  out.r = adj_q;
  out.w = adj_w;
  out.v = adj_dpdt;
  return true;
}

// contact_point_adj (quaternion form) in two pieces for the adjoint kernels' contact wave (round 6): everything the hit's adjoint needs of
// the STATE -- the forward pass's quantities again -- is computed when the body's record is staged, before the wrench adjoints exist
// (hand-over A); the reverse sweep follows them.  contact_pre_barrier pins the values at the seam so that the same instructions come out
// wherever the two pieces are called (back to back in the generic sweep, around the wait in the fast path).
struct ContactPre { bool touch, a_lt_b; v3 rr, fc, f_raw, nvt, vt, w, com; qt q; float mm, vn, stepc; };
PD_DEV ContactPre contact_point_adj_pre(const float *r, float4 cv, float4 P, float4 mat) {
  ContactPre C;
  const v3 p = ld3(r), v = ld3(r + 10), rc = ld3(r + 13);
  C.w = ld3(r + 7);
  C.q = ld4(r + 3);
  const v3 cpt = V3(P.x, P.y, P.z);
  C.com = qrot_inv(C.q, rc);
  v3 cp = (p + qrot(C.q, cpt)) - V3(0.f, P.w, 0.f);
  cp.y = contact_height(cv, P);
  const float c = cp.y;
  C.touch = !(c > 0.0f);
  C.rr = cp - (p + rc);
  const v3 dpdt = v + cross(C.w, C.rr);
  const float ke = mat.x, kd = mat.y, kf = mat.z, mu = mat.w;
  C.vn = dpdt.y;
  C.vt = V3(dpdt.x, dpdt.y - C.vn, dpdt.z);
  const float fn = c * ke;
  C.stepc = c < 0.0f ? 1.0f : 0.0f;
  const float fd = fminf(C.vn, 0.0f) * kd * C.stepc;
  const float lvt = length(C.vt);
  C.nvt = normalize(C.vt);
  const float a_ = kf * lvt, b_ = 0.0f - mu * (fn + fd);
  C.a_lt_b = a_ < b_;
  C.mm = C.a_lt_b ? a_ : b_;
  C.f_raw = V3(C.nvt.x * C.mm, (fn + fd) + C.nvt.y * C.mm, C.nvt.z * C.mm);
  C.fc = clamp3(C.f_raw, 500.0f);
  return C;
}
#define PD_PIN_(x) asm volatile("" : "+v"(x))
#define PD_PIN3_(a) do { PD_PIN_((a).x); PD_PIN_((a).y); PD_PIN_((a).z); } while (0)
PD_DEV void contact_pre_barrier(ContactPre &C) {
  PD_PIN3_(C.rr); PD_PIN3_(C.fc); PD_PIN3_(C.f_raw); PD_PIN3_(C.nvt); PD_PIN3_(C.vt); PD_PIN3_(C.w); PD_PIN3_(C.com);
  PD_PIN_(C.q.x); PD_PIN_(C.q.y); PD_PIN_(C.q.z); PD_PIN_(C.q.w); PD_PIN_(C.mm); PD_PIN_(C.vn); PD_PIN_(C.stepc);
}
PD_DEV bool contact_point_adj_rest(const ContactPre &C, float4 P, float4 mat, v3 g_t, v3 g_f, BodyAdj &out) {
  if (!C.touch) return false;
  const v3 cpt = V3(P.x, P.y, P.z);
  const float ke = mat.x, kd = mat.y, kf = mat.z, mu = mat.w;
  // reverse (body_f -= (t, f))
  v3 adj_t = -g_t, adj_fc = -g_f, adj_r = V3(0, 0, 0);
  adj_cross(C.rr, C.fc, adj_r, adj_fc, adj_t);
  v3 adj_fr = clamp3_pass(C.f_raw, adj_fc, 500.0f);
  float adj_fnfd = adj_fr.y;
  v3 adj_nvt = adj_fr * C.mm;
  float adj_m = dot(adj_fr, C.nvt);
  float adj_lvt = 0.f;
  if (C.a_lt_b) adj_lvt = adj_m * kf; else adj_fnfd += -mu * adj_m;
  v3 adj_vt = V3(0, 0, 0);
  adj_normalize(C.vt, adj_vt, adj_nvt);
  adj_length(C.vt, adj_vt, adj_lvt);
  float adj_c = adj_fnfd * ke;
  float adj_vn = (C.vn < 0.0f ? 1.0f : 0.0f) * kd * C.stepc * adj_fnfd;
  v3 adj_dpdt = adj_vt;
  adj_vn += -adj_vt.y;
  adj_dpdt.y += adj_vn;
  v3 adj_w = V3(0, 0, 0);
  adj_cross(C.w, C.rr, adj_w, adj_r, adj_dpdt);
  v3 adj_cp = V3(adj_r.x, adj_r.y + adj_c, adj_r.z);
  qt adj_q = Q4(0, 0, 0, 0);
  adj_qrot_q(C.q, C.com, adj_q, -adj_r);
  adj_qrot_q(C.q, cpt, adj_q, adj_cp);
  out.p = adj_cp - adj_r;
  out.r = adj_q;
  out.w = adj_w;
  out.v = adj_dpdt;
  return true;
}

// ---------------------------------------------------------------------------------------------
// Joint PD + attachment forces for joint i == child body i (integrator_euler.py:289-451).
PD_DEV float joint_force(float q, float qd, float target, float ke, float kd, float act, float lo, float up, float lke,
                         float lkd) {
  float limit_f = 0.0f;  // :274-281
  if (q < lo) limit_f = lke * (lo - q) - lkd * fminf(qd, 0.0f);
  if (q > up) limit_f = lke * (up - q) - lkd * fmaxf(qd, 0.0f);
  return ke * (q - target) + kd * qd + act - limit_f;  // :284
}
PD_DEV void joint_force_adj(float q, float qd, float target, float ke, float kd, float lo, float up, float lke, float lkd,
                            float g, float &adj_q, float &adj_qd, float &adj_target, float &adj_ke, float &adj_kd,
                            float &adj_act) {
  adj_ke += g * (q - target); adj_q += g * ke; adj_target += -g * ke;
  adj_kd += g * qd; adj_qd += g * kd; adj_act += g;
  // the limit force's adjoint (:274-281) as selects -- the same sums (x + 0 is x), no divergent region: the nested conditionals compiled to
  // two taken branches per dof on a wave whose instruction stream is its run time (round 6)
  const float adj_limit = -g;
  const bool over = q > up, under = !over && q < lo;
  adj_q += (over || under) ? -lke * adj_limit : 0.0f;
  adj_qd += ((over && qd > 0.0f) || (under && qd < 0.0f)) ? -lkd * adj_limit : 0.0f;
}

// (-fno-signed-zeros, csrc/Makefile: the sign of a zero is not defined in this library.  The one place a value hangs on it is
// atan2f(+-0, negative) = +-pi below: a compound joint angle of EXACTLY 180 degrees about its first or third axis may come out as
// +pi or -pi.  Both name the same rotation; the PD term ke (q - target) differs by 2 pi ke there -- a configuration far outside
// the |second angle| < pi/2 range in which quat_decompose is a chart at all.  Stated in INTEGRATION.md section 4.)
PD_DEV void quat_decompose(qt q, float *ang, v3 &c0, v3 &c1, v3 &c2) {  // :245-258; also returns the rotated basis
  c0 = qrot(q, V3(1, 0, 0)); c1 = qrot(q, V3(0, 1, 0)); c2 = qrot(q, V3(0, 0, 1));
  ang[0] = -atan2_any(c2.y, c2.z); ang[1] = -asin_c(-c2.x); ang[2] = -atan2_any(c1.x, c0.x);
}
PD_DEV void quat_decompose(qt q, float *ang) {
  v3 c0, c1, c2;
  quat_decompose(q, ang, c0, c1, c2);
}
// The adjoint's recomputation of the same: the rotated basis is the three columns of rotm(q) -- qrot's terms without the
// products with the basis vectors' zeros (which the compiler may not drop without fast-math): 22 instructions instead of 3 x 23.
PD_DEV void quat_decompose_cols(qt q, float *ang, v3 &c0, v3 &c1, v3 &c2) {
  const float s = 2.0f * q.w * q.w - 1.0f, tw = 2.0f * q.w, tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  c0 = V3(s + q.x * tx, q.z * tw + q.y * tx, q.z * tx - q.y * tw);
  c1 = V3(q.x * ty - q.z * tw, s + q.y * ty, q.x * tw + q.z * ty);
  c2 = V3(q.y * tw + q.x * tz, q.y * tz - q.x * tw, s + q.z * tz);
  // (round 5: atan2_any, pd_math.h, instead of libdevice's atan2f: the same function to 2e-7 relative, ~20 instructions less per call on the
  // one wave whose instruction stream IS the step of a compound robot's small-batch rollout)
  ang[0] = -atan2_any(c2.y, c2.z); ang[1] = -asin_c(-c2.x); ang[2] = -atan2_any(c1.x, c0.x);
}
PD_DEV void quat_decompose_adj(qt q, v3 c0, v3 c1, v3 c2, const float *g, qt &adj_q) {  // c* = the rotated basis of the forward pass
  float gphi = -g[0], gth = -g[1], gpsi = -g[2];
  // matrix adjoint of rotm(q): only five of its entries are touched (c0.x, c1.x, c2.xyz)
  float A[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  { float y = c2.y, x = c2.z, d = x * x + y * y; const float id = rcp_hw(d); A[5] = gphi * x * id; A[8] = -gphi * y * id; }
  { float s = -c2.x; A[2] = -gth * inv_sqrt_1mx2(s); }
  { float y = c1.x, x = c0.x, d = x * x + y * y; const float id = rcp_hw(d); A[1] = gpsi * x * id; A[0] = -gpsi * y * id; }
  rotm_adj(q, A, adj_q);
}

struct JointCtx {  // locals shared by the forward and the adjoint
  v3 pp, x_p, r_p, w_p, v_p, r_c, x_err, v_err, w_err;
  qt qp, q_p, r_err;
};

// HP (here and in the adjoint functions below): the caller promises a PLAIN model (pd_parented(JT)): every joint that is not FREE
// hangs on a body (c.parent >= 0) and every child joint frame (joint_X_c) has the identity rotation -- true of every robot of the
// reference (Warp's URDF importer only makes such models); the host sends anything else to the generic instantiation.  The
// `if (parent)` regions then cost no exec-mask code and no moves that merge their results with the values of lanes outside them
// (-62 instructions per adjoint step), and the compound joint's six quaternion products with q_off drop out of the adjoint
// (x * identity is x exactly; -5 % adjoint time for human / quad).  The forward pass keeps its products: its arithmetic is frozen.
template <bool HP = false>
PD_DEV void joint_ctx(const BodyConst &c, const BodyState &s, v3 rc_c, const float *rec, JointCtx &j) {
  j.pp = V3(0, 0, 0); j.qp = Q4(0, 0, 0, 1); j.x_p = c.p_pj; j.q_p = c.q_pj;
  j.r_p = V3(0, 0, 0); j.w_p = V3(0, 0, 0); j.v_p = V3(0, 0, 0);
  if (HP || c.parent >= 0) {  // :326-333
    const float *r = rec + (HP ? c.pidx : c.parent) * PD_REC;
    j.pp = ld3(r); j.qp = ld4(r + 3); j.w_p = ld3(r + 7); j.v_p = ld3(r + 10);
    j.x_p = j.pp + qrot(j.qp, c.p_pj);
    j.q_p = qmul(j.qp, c.q_pj);
    j.r_p = j.x_p - (j.pp + ld3(r + 13));
  }
  j.r_c = s.p - (s.p + rc_c);  // :338, rc_c = rot(q_c, com) from the staging
  j.x_err = s.p - j.x_p; j.r_err = qmul(qconj(j.q_p), s.r);  // :369-372
  j.v_err = s.v - j.v_p; j.w_err = s.w - j.w_p;
}

// tgt/act/ke/kd: this joint's dofs (1 for revolute, 3 for compound).  Outputs the wrench pair:
// parent += (t + r_p x f, f), child -= (t + r_c x f, f)   (:448-451)
// PLAINC: compound joints of a PLAIN model (joint_ctx) in the restructured form of the adjoint's joint_adj_prep -- no products
// with the identity child frame, the rotated basis as matrix columns, the axis chain with its zero components taken out, one
// matrix for the three axis rotations: the same terms minus products with exact zeros, ~200 instructions less per joint.
// Only the compound-only instantiation sets it: the revolute forward pass (Laikago) stays bit for bit what round 1 shipped.
// Rm = rotm(s.r) of this body (the integration made it for the staging).  HP: plain model (joint_ctx) -- the call may then run
// for EVERY lane, unguarded: a lane without a joint (the FREE root, idle lanes) computes on the record of body c.pidx = 0 and
// the caller drops its result.
// ALL (round 6, with HP): the model has ONE joint type besides the FREE root (JT is a single bit) and the call runs for every lane without
// a type test -- no divergent region, no exec-mask code, no moves merging its results; the caller drops the result of a lane without a joint.
template <int JT, bool PLAINC = false, bool HP = PLAINC, bool ALL = false>
PD_DEV void joint_fwd(const PdDevModel &m, const BodyConst &c, const BodyState &s, v3 rc_c, const float *Rm, const float *rec, const float *tgt,
                      const float *act, const float *ke, const float *kd, v3 &wp_t, v3 &wp_f, v3 &wc_t, v3 &wc_f) {
  JointCtx j;
  joint_ctx<HP>(c, s, rc_c, rec, j);
  const float ake = m.attach_ke, akd = m.attach_kd, ads = 0.01f;
  v3 t_total = V3(0, 0, 0), f_total = V3(0, 0, 0);
  if ((JT & PD_JT_FIXED) && c.type == PD_JOINT_FIXED) {  // :385-390
    float hss_, hw_;
    const v3 rv = qvec(j.r_err);
    v3 ang_err = rv * fixed_ang_h(dot(rv, rv), j.r_err.w, hss_, hw_);  // = normalize(rv) * 2 acos(w), scale-invariantly (pd_math.h)
    f_total += j.x_err * ake + j.v_err * akd;
    t_total += qrot(j.q_p, ang_err) * ake + j.w_err * (akd * ads);
  }
  static_assert(!ALL || (HP && (JT == PD_JT_REVOLUTE || JT == PD_JT_COMPOUND)), "ALL: a plain model with one joint type");
  if ((JT & PD_JT_REVOLUTE) && (HP || c.type == PD_JOINT_REVOLUTE)) {  // :392-409
    v3 axis_p = qrot(j.q_p, c.axis), axis_c = mat_vec(Rm, c.axis);
    float q = twist_angle(dot(qvec(j.r_err), c.axis), j.r_err.w, c.alen);  // :394-400, see pd_math.h
    float qd = dot(j.w_err, axis_p);
    const JointLimit L = c.lim[0];
    float jf = joint_force(q, qd, tgt[0], ke[0], kd[0], act[0], L.lo, L.up, L.ke, L.kd);
    t_total = axis_p * jf;
    v3 swing = cross(axis_p, axis_c);
    f_total += j.x_err * ake + j.v_err * akd;
    t_total += swing * ake + (j.w_err - axis_p * qd) * (akd * ads);
  }
  if (PLAINC && (JT & PD_JT_COMPOUND) && (ALL || c.type == PD_JOINT_COMPOUND)) {  // :411-445, restructured
    const qt q_pc = qmul(qconj(j.q_p), s.r);
    float ang[3];
    v3 b0, b1, b2;
    quat_decompose_cols(q_pc, ang, b0, b1, b2);
    float s0, c0;
    sincos_half_pi(ang[0] * 0.5f, s0, c0);  // |ang / 2| <= pi / 2: no range reduction needed (round 5; was sincosf, ~25 instructions more each)
    const v3 ax1 = V3(0.f, 2.0f * c0 * c0 - 1.0f, s0 * (2.0f * c0));
    float2 sc1_;
    qt q_1 = q_axis_angle_sc(ax1, ang[1], sc1_);
    q_1.x = 0.f;
    const qt q10 = Q4(q_1.w * s0, c0 * q_1.y + q_1.z * s0, c0 * q_1.z - q_1.y * s0, q_1.w * c0);
    const v3 ax2 = V3(q10.y * (2.0f * q10.w) + q10.x * (2.0f * q10.z), q10.y * (2.0f * q10.z) - q10.x * (2.0f * q10.w),
                      (2.0f * q10.w * q10.w - 1.0f) + q10.z * (2.0f * q10.z));
    float Mw[9];
    rotm(j.q_p, Mw);
    const v3 axw[3] = {V3(Mw[0], Mw[3], Mw[6]), mat_vec(Mw, ax1), mat_vec(Mw, ax2)};
    t_total = V3(0, 0, 0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const JointLimit L = c.lim[k];
      float jf = joint_force(ang[k], dot(axw[k], j.w_err), tgt[k], ke[k], kd[k], act[k], L.lo, L.up, L.ke, L.kd);
      t_total += axw[k] * jf;
    }
    t_total = clamp3(t_total, 1.0e4f);
    f_total += clamp3(j.x_err * ake + j.v_err * akd, 1.0e4f);
  }
  if (!PLAINC && (JT & PD_JT_COMPOUND) && c.type == PD_JOINT_COMPOUND) {  // :411-445
    qt q_pc = qmul(qmul(qmul(qconj(c.q_off), qconj(j.q_p)), s.r), c.q_off);
    float ang[3];
    quat_decompose(q_pc, ang);
    v3 ax[3];
    ax[0] = V3(1, 0, 0);
    float2 sc_;  // (the decomposition's angles lie in [-pi, pi]: half of one needs no range reduction)
    qt q_0 = q_axis_angle_sc(ax[0], ang[0], sc_);
    ax[1] = qrot(q_0, V3(0, 1, 0));
    qt q_1 = q_axis_angle_sc(ax[1], ang[1], sc_);
    ax[2] = qrot(qmul(q_1, q_0), V3(0, 0, 1));
    qt q_w = qmul(j.q_p, c.q_off);
    t_total = V3(0, 0, 0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      v3 axw = qrot(q_w, ax[k]);
      const JointLimit L = c.lim[k];
      float jf = joint_force(ang[k], dot(axw, j.w_err), tgt[k], ke[k], kd[k], act[k], L.lo, L.up, L.ke, L.kd);
      t_total += axw * jf;
    }
    t_total = clamp3(t_total, 1.0e4f);
    f_total += clamp3(j.x_err * ake + j.v_err * akd, 1.0e4f);
  }
  wp_t = t_total + cross(j.r_p, f_total); wp_f = f_total;
  wc_t = t_total + cross(j.r_c, f_total); wc_f = f_total;
}

// Adjoint.  gc_* = adjoint of the child's wrench accumulator, gp_* = of the parent's (zero if none).
// own += d/d(child state); par = d/d(parent state); a_* = per-dof gradients (overwritten).
//
// Two halves: joint_adj_prep recomputes everything that depends on the stored state and the controls only (the joint
// frames and errors, and for a COMPOUND joint the whole forward pass: angle decomposition, axes, PD forces) -- the
// role-split adjoint kernel runs it on the joint wave BEFORE the wrench adjoints exist -- and joint_adj_apply is the part
// that needs them.  joint_adj = prep + apply.
struct JointPrep {
  JointCtx j;
  v3 f_raw, f_total;
  // COMPOUND only
  qt qa, q_pc, q_0, q_1, q10, q_w;
  v3 ax1, ax2, axw[3], t_raw;
  float ang[3], jf[3], qdk[3];
  float Mw[9];  // rotm(q_w), handed from prep to apply
  v3 c0, c1, c2;  // columns of R(q_pc): quat_decompose and its adjoint share them
  float2 sc0, sc1;  // (sin, cos) of ang[0] / 2 and ang[1] / 2, as q_axis_angle computed them
};

template <int JT, bool HP = false, bool ALL = false>  // ALL: see joint_fwd -- one joint type, no type tests
PD_DEV void joint_adj_prep(const PdDevModel &m, const BodyConst &c, const BodyState &s, v3 rc_c, const float *rec, const float *tgt,
                           const float *act, const float *ke, const float *kd, JointPrep &P) {
  joint_ctx<HP>(c, s, rc_c, rec, P.j);
  const JointCtx &j = P.j;
  const float ake = m.attach_ke, akd = m.attach_kd;
  P.f_raw = j.x_err * ake + j.v_err * akd;
  P.f_total = ((JT & PD_JT_COMPOUND) && (ALL || c.type == PD_JOINT_COMPOUND)) ? clamp3(P.f_raw, 1.0e4f) : P.f_raw;
  if ((JT & PD_JT_COMPOUND) && (ALL || c.type == PD_JOINT_COMPOUND)) {
    P.qa = HP ? qconj(j.q_p) : qmul(qconj(c.q_off), qconj(j.q_p));  // (HP: identity child frames, see joint_ctx)
    const qt qb = qmul(P.qa, s.r);
    P.q_pc = HP ? qb : qmul(qb, c.q_off);
    quat_decompose_cols(P.q_pc, P.ang, P.c0, P.c1, P.c2);
    const v3 ax0 = V3(1, 0, 0);
    // the axis chain of integrator_euler.py:418-427 with its structure spelled out -- q_0 = (s0, 0, 0, c0) turns about x, so
    // ax1 = rot(q_0, e_y), q_1 = (ax1 s1, c1) and q_1 q_0 have zero components, and ax2 = rot(q_1 q_0, e_z) is a matrix column.
    // The same terms as the generic qrot / qmul minus the products with exact zeros (which the compiler may not drop without
    // fast-math): same values, ~45 instructions less here and ~70 less in the adjoint below
    float s0, c0;
    sincos_half_pi(P.ang[0] * 0.5f, s0, c0);
    P.sc0 = make_float2(s0, c0);
    P.q_0 = Q4(s0, 0.f, 0.f, c0);
    P.ax1 = V3(0.f, 2.0f * c0 * c0 - 1.0f, s0 * (2.0f * c0));
    P.q_1 = q_axis_angle_sc(P.ax1, P.ang[1], P.sc1);
    P.q_1.x = 0.f;
    P.q10 = Q4(P.q_1.w * s0, c0 * P.q_1.y + P.q_1.z * s0, c0 * P.q_1.z - P.q_1.y * s0, P.q_1.w * c0);
    {
      const qt q = P.q10;
      P.ax2 = V3(q.y * (2.0f * q.w) + q.x * (2.0f * q.z), q.y * (2.0f * q.z) - q.x * (2.0f * q.w), (2.0f * q.w * q.w - 1.0f) + q.z * (2.0f * q.z));
    }
    P.q_w = HP ? j.q_p : qmul(j.q_p, c.q_off);
    const v3 ax[3] = {ax0, P.ax1, P.ax2};
    P.t_raw = V3(0, 0, 0);
    rotm(P.q_w, P.Mw);  // one quaternion rotates the three axes (pd_math.h)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      P.axw[k] = k == 0 ? V3(P.Mw[0], P.Mw[3], P.Mw[6]) : mat_vec(P.Mw, ax[k]); P.qdk[k] = dot(P.axw[k], j.w_err);
      const JointLimit L = c.lim[k];
      P.jf[k] = joint_force(P.ang[k], P.qdk[k], tgt[k], ke[k], kd[k], act[k], L.lo, L.up, L.ke, L.kd);
      P.t_raw += P.axw[k] * P.jf[k];
    }
  }
}

template <int JT, bool HP = false, bool ALL = false>
PD_DEV void joint_adj_apply(const PdDevModel &m, const BodyConst &c, const BodyState &s, const JointPrep &P, const float *tgt,
                            const float *act, const float *ke, const float *kd, v3 gc_t, v3 gc_f, v3 gp_t, v3 gp_f, BodyAdj &own,
                            BodyAdj &par, float *a_tgt, float *a_act, float *a_ke, float *a_kd) {
  const JointCtx &j = P.j;
  const float ake = m.attach_ke, akd = m.attach_kd, ads = 0.01f;
  const v3 f_raw = P.f_raw, f_total = P.f_total;
  v3 adj_t = -gc_t, adj_f = -gc_f, adj_r_c = V3(0, 0, 0), adj_r_p = V3(0, 0, 0);
  adj_cross(j.r_c, f_total, adj_r_c, adj_f, -gc_t);
  if (HP || c.parent >= 0) {
    adj_t += gp_t; adj_f += gp_f;
    adj_cross(j.r_p, f_total, adj_r_p, adj_f, gp_t);
  }
  v3 adj_x_err = V3(0, 0, 0), adj_v_err = V3(0, 0, 0), adj_w_err = V3(0, 0, 0);
  qt adj_r_err = Q4(0, 0, 0, 0), adj_q_p = Q4(0, 0, 0, 0), adj_q_c = Q4(0, 0, 0, 0);
  if ((JT & PD_JT_FIXED) && c.type == PD_JOINT_FIXED) {
    v3 rv = qvec(j.r_err);
    float hss, h_w;
    const float h = fixed_ang_h(dot(rv, rv), j.r_err.w, hss, h_w);
    v3 ang_err = rv * h;
    adj_x_err += adj_f * ake; adj_v_err += adj_f * akd; adj_w_err += adj_t * (akd * ads);
    v3 adj_ang_err = V3(0, 0, 0);
    adj_qrot(j.q_p, ang_err, adj_q_p, adj_ang_err, adj_t * ake);
    const float va = dot(rv, adj_ang_err);
    v3 adj_rv = adj_ang_err * h + rv * (va * hss);
    adj_r_err.x += adj_rv.x; adj_r_err.y += adj_rv.y; adj_r_err.z += adj_rv.z;
    adj_r_err.w += va * h_w;
  }
  if ((JT & PD_JT_REVOLUTE) && c.type == PD_JOINT_REVOLUTE) {
    v3 axis_p = qrot(j.q_p, c.axis), axis_c = qrot(s.r, c.axis);
    float dq_dda, dq_dw;
    float q = twist_angle(dot(qvec(j.r_err), c.axis), j.r_err.w, c.alen, dq_dda, dq_dw);
    float qd = dot(j.w_err, axis_p);
    const JointLimit L = c.lim[0];
    float lo = L.lo, up = L.up, lke = L.ke, lkd = L.kd;
    float jf = joint_force(q, qd, tgt[0], ke[0], kd[0], act[0], lo, up, lke, lkd);
    adj_x_err += adj_f * ake; adj_v_err += adj_f * akd;
    float adj_jf = dot(adj_t, axis_p);
    v3 adj_axis_p = adj_t * jf, adj_axis_c = V3(0, 0, 0);
    adj_w_err += adj_t * (akd * ads);
    float adj_qd = -dot(adj_t, axis_p) * (akd * ads);
    adj_axis_p += adj_t * (-qd * (akd * ads));
    adj_cross(axis_p, axis_c, adj_axis_p, adj_axis_c, adj_t * ake);
    float adj_q = 0.f;
    a_tgt[0] = 0.f; a_act[0] = 0.f; a_ke[0] = 0.f; a_kd[0] = 0.f;
    joint_force_adj(q, qd, tgt[0], ke[0], kd[0], lo, up, lke, lkd, adj_jf, adj_q, adj_qd, a_tgt[0], a_ke[0], a_kd[0], a_act[0]);
    adj_w_err += axis_p * adj_qd; adj_axis_p += j.w_err * adj_qd;
    const float adj_da = adj_q * dq_dda;
    adj_r_err.x += c.axis.x * adj_da; adj_r_err.y += c.axis.y * adj_da; adj_r_err.z += c.axis.z * adj_da;
    adj_r_err.w += adj_q * dq_dw;
    adj_qrot_q(j.q_p, c.axis, adj_q_p, adj_axis_p);
    adj_qrot_q(s.r, c.axis, adj_q_c, adj_axis_c);
  }
  if ((JT & PD_JT_COMPOUND) && (ALL || c.type == PD_JOINT_COMPOUND)) {
    const v3 ax[3] = {V3(1, 0, 0), P.ax1, P.ax2};
    v3 adj_f_raw = clamp3_pass(f_raw, adj_f, 1.0e4f);
    adj_x_err += adj_f_raw * ake; adj_v_err += adj_f_raw * akd;
    v3 adj_t_raw = clamp3_pass(P.t_raw, adj_t, 1.0e4f);
    float adj_ang[3] = {0.f, 0.f, 0.f};
    v3 adj_ax[3] = {V3(0, 0, 0), V3(0, 0, 0), V3(0, 0, 0)};
    qt adj_q_w = Q4(0, 0, 0, 0);
    const float *Mw = P.Mw;
    float aW[9];  // matrix adjoint of rotm(q_w): axw[k] = Mw ax[k]
#pragma unroll
    for (int k = 0; k < 9; ++k) aW[k] = 0.f;
#pragma unroll
    for (int k = 2; k >= 0; --k) {
      float adj_jf = dot(adj_t_raw, P.axw[k]);
      v3 adj_axw = adj_t_raw * P.jf[k];
      float adj_qdk = 0.f;
      a_tgt[k] = 0.f; a_act[k] = 0.f; a_ke[k] = 0.f; a_kd[k] = 0.f;
      const JointLimit L = c.lim[k];
      joint_force_adj(P.ang[k], P.qdk[k], tgt[k], ke[k], kd[k], L.lo, L.up, L.ke, L.kd, adj_jf, adj_ang[k], adj_qdk, a_tgt[k], a_ke[k],
                      a_kd[k], a_act[k]);
      adj_axw += j.w_err * adj_qdk; adj_w_err += P.axw[k] * adj_qdk;
      adj_ax[k] += matT_vec(Mw, adj_axw);
      add_outer(aW, adj_axw, ax[k]);
    }
    rotm_adj(P.q_w, aW, adj_q_w);
    if (HP) adj_q_p += adj_q_w; else adj_qmul_a(c.q_off, adj_q_p, adj_q_w);
    // adjoint of the axis chain, with the zero components of q_0, q_1 taken out like in joint_adj_prep
    const float s0 = P.sc0.x, c0 = P.sc0.y;
    qt G;  // adjoint of q10 = q_1 q_0 through ax2 = rot(q10, e_z)
    {
      const qt q = P.q10;
      const v3 g = adj_ax[2];
      G = Q4(g.x * (2.0f * q.z) - g.y * (2.0f * q.w), g.x * (2.0f * q.w) + g.y * (2.0f * q.z),
             g.x * (2.0f * q.x) + g.y * (2.0f * q.y) + g.z * (4.0f * q.z), g.x * (2.0f * q.y) - g.y * (2.0f * q.x) + g.z * (4.0f * q.w));
    }
    const qt adj_q_1 = Q4(c0 * G.x - s0 * G.w, c0 * G.y - s0 * G.z, c0 * G.z + s0 * G.y, c0 * G.w + s0 * G.x);  // G conj(q_0)
    float adj_q0x = P.q_1.w * G.x - P.q_1.y * G.z + P.q_1.z * G.y;  // conj(q_1) G: only x and w reach the angle
    float adj_q0w = P.q_1.w * G.w + P.q_1.y * G.y + P.q_1.z * G.z;
    adj_q_axis_angle_sc(ax[1], P.sc1.x, P.sc1.y, adj_ax[1], adj_ang[1], adj_q_1);
    adj_q0x += adj_ax[1].z * (2.0f * c0);                                  // ax1 = (0, 2 c0^2 - 1, 2 c0 s0)
    adj_q0w += adj_ax[1].y * (4.0f * c0) + adj_ax[1].z * (2.0f * s0);
    adj_ang[0] += 0.5f * (c0 * adj_q0x - s0 * adj_q0w);                    // q_0 = (sin, 0, 0, cos)(ang[0] / 2)
    qt adj_q_pc = Q4(0, 0, 0, 0);
    quat_decompose_adj(P.q_pc, P.c0, P.c1, P.c2, adj_ang, adj_q_pc);
    qt adj_qb = Q4(0, 0, 0, 0), adj_qa = adj_qb, adj_cqp = adj_qb;
    if (HP) adj_qb += adj_q_pc; else adj_qmul_a(c.q_off, adj_qb, adj_q_pc);
    adj_qmul(P.qa, s.r, adj_qa, adj_q_c, adj_qb);
    if (HP) adj_cqp += adj_qa; else adj_qmul_b(qconj(c.q_off), adj_cqp, adj_qa);
    adj_q_p += qconj(adj_cqp);
  }
  if (!((JT & PD_JT_COMPOUND) && (ALL || c.type == PD_JOINT_COMPOUND))) {  // r_err = conj(q_p) * q_c  (a COMPOUND joint does not use r_err:
    qt adj_cqp = Q4(0, 0, 0, 0);                                  //  its adjoint is exactly zero there)
    adj_qmul(qconj(j.q_p), s.r, adj_cqp, adj_q_c, adj_r_err);
    adj_q_p += qconj(adj_cqp);
  }
  adj_qrot_q(s.r, c.com, adj_q_c, -adj_r_c);  // r_c = x_c - (x_c + rot(q_c, com)); rc_c is a function of q_c
  own.p += adj_x_err; own.r += adj_q_c; own.w += adj_w_err; own.v += adj_v_err;
  par = adj_zero();
  if (HP || c.parent >= 0) {
    v3 adj_x_p = adj_r_p - adj_x_err;
    par.p = adj_x_p - adj_r_p;                           // x_p = pp + ..., r_p = x_p - (pp + rc_par)
    float aP[9];  // matrix adjoint of rotm(qp): rc_par = rotm(qp) com_par, x_p = pp + rotm(qp) p_pj
#pragma unroll
    for (int k = 0; k < 9; ++k) aP[k] = 0.f;
    add_outer(aP, -adj_r_p, c.com_par);
    add_outer(aP, adj_x_p, c.p_pj);
    rotm_adj(j.qp, aP, par.r);
    adj_qmul_a(c.q_pj, par.r, adj_q_p);
    par.w = -adj_w_err; par.v = -adj_v_err;
  }
}

template <int JT, bool HP = false>
PD_DEV void joint_adj(const PdDevModel &m, const BodyConst &c, const BodyState &s, v3 rc_c, const float *rec, const float *tgt,
                      const float *act, const float *ke, const float *kd, v3 gc_t, v3 gc_f, v3 gp_t, v3 gp_f, BodyAdj &own,
                      BodyAdj &par, float *a_tgt, float *a_act, float *a_ke, float *a_kd) {
  JointPrep P;
  joint_adj_prep<JT, HP>(m, c, s, rc_c, rec, tgt, act, ke, kd, P);
  joint_adj_apply<JT, HP>(m, c, s, P, tgt, act, ke, kd, gc_t, gc_f, gp_t, gp_f, own, par, a_tgt, a_act, a_ke, a_kd);
}

// ---------------------------------------------------------------------------------------------
// Revolute joint adjoint in two halves (wave-specialised adjoint kernel): rev_forward recomputes everything that depends
// on the stored state and the controls only -- it runs on the otherwise idle contact wave and is handed over through
// LDS -- and rev_adjoint, on the body wave, is the part that needs the wrench adjoints.  Together they equal the
// revolute branch of joint_adj.
#define PD_JC 23  // floats of the hand-over record (odd stride)
#define PD_QPRE 22  // floats per LANE of the quad-lane adjoint's state-only hand-over (contact wave -> body wave, [field][64 lanes]): s (4), t0, f0,
                    // clamp mask, rc, rotm rows (3) and columns (3), the forward values of integrate_bodies' adjoint (QIntTmp: wb, Iwb, tb, u, w1, il, r1), pad
#define PD_QGEN 3   // generations of it (and of the records / cull vectors the same wave stages): step k - 1 is written while k and k + 1 are read
struct RevCache {
  qt q_p, r_err;
  v3 x_p, axis_p, axis_c;
  float q, qd, jf, dq_dda, dq_dw;  // partial derivatives of the twist angle (pd_math.h twist_angle)
};
PD_DEV void rev_cache_store(float *d, const RevCache &R) {
  d[0] = R.q_p.x; d[1] = R.q_p.y; d[2] = R.q_p.z; d[3] = R.q_p.w; d[4] = R.r_err.x; d[5] = R.r_err.y; d[6] = R.r_err.z; d[7] = R.r_err.w;
  d[8] = R.x_p.x; d[9] = R.x_p.y; d[10] = R.x_p.z;
  d[11] = R.axis_p.x; d[12] = R.axis_p.y; d[13] = R.axis_p.z; d[14] = R.axis_c.x; d[15] = R.axis_c.y; d[16] = R.axis_c.z;
  d[17] = R.q; d[18] = R.qd; d[19] = R.jf; d[20] = R.dq_dda; d[21] = R.dq_dw;
}
PD_DEV RevCache rev_cache_load(const float *d) {
  RevCache R;
  R.q_p = ld4(d); R.r_err = ld4(d + 4); R.x_p = ld3(d + 8); R.axis_p = ld3(d + 11); R.axis_c = ld3(d + 14);
  R.q = d[17]; R.qd = d[18]; R.jf = d[19]; R.dq_dda = d[20]; R.dq_dw = d[21];
  return R;
}

// pp, qp, w_p: pose and angular velocity of the parent body (ignored for a joint to the world)
template <bool HP = false>
PD_DEV RevCache rev_forward(const PdDevModel &m, const BodyConst &c, qt q_c, v3 w_c, v3 pp, qt qp, v3 w_p, float tgt, float act, float ke,
                            float kd) {
  RevCache R;
  R.x_p = c.p_pj; R.q_p = c.q_pj;
  if (HP || c.parent >= 0) {
    R.x_p = pp + qrot(qp, c.p_pj);
    R.q_p = qmul(qp, c.q_pj);
  } else {
    w_p = V3(0, 0, 0);
  }
  R.r_err = qmul(qconj(R.q_p), q_c);
  R.axis_p = qrot(R.q_p, c.axis);
  R.axis_c = qrot(q_c, c.axis);
  R.q = twist_angle(dot(qvec(R.r_err), c.axis), R.r_err.w, c.alen, R.dq_dda, R.dq_dw);
  R.qd = dot(w_c - w_p, R.axis_p);
  const JointLimit L = c.lim[0];
  R.jf = joint_force(R.q, R.qd, tgt, ke, kd, act, L.lo, L.up, L.ke, L.kd);
  return R;
}

// Parent quantities are passed explicitly (pp, qp, w_p, v_p, rc_par = rot(qp, com_par); ignored for a joint to the world).
template <bool HP = false>
PD_DEV void rev_adjoint_core(const PdDevModel &m, const BodyConst &c, const BodyState &s, v3 rc_c, v3 pp, qt qp, v3 w_p, v3 v_p, v3 rc_par,
                             const RevCache &R, float tgt, float ke, float kd, v3 gc_t, v3 gc_f, v3 gp_t, v3 gp_f, BodyAdj &own, BodyAdj &par,
                             float *aR, float &a_tgt, float &a_act, float &a_ke, float &a_kd) {  // aR: matrix adjoint of rotm(s.r), see integrate_adj2
  const float ake = m.attach_ke, akd = m.attach_kd, ads = 0.01f;
  v3 r_p = V3(0, 0, 0);
  if (HP || c.parent >= 0) {
    r_p = R.x_p - (pp + rc_par);
  } else {
    pp = V3(0, 0, 0); w_p = pp; v_p = pp; qp = Q4(0, 0, 0, 1);
  }
  const v3 r_c = s.p - (s.p + rc_c);
  const v3 x_err = s.p - R.x_p, v_err = s.v - v_p, w_err = s.w - w_p;
  const v3 f_total = x_err * ake + v_err * akd;
  v3 adj_t = -gc_t, adj_f = -gc_f, adj_r_c = V3(0, 0, 0), adj_r_p = V3(0, 0, 0);
  adj_cross(r_c, f_total, adj_r_c, adj_f, -gc_t);
  if (HP || c.parent >= 0) {
    adj_t += gp_t; adj_f += gp_f;
    adj_cross(r_p, f_total, adj_r_p, adj_f, gp_t);
  }
  v3 adj_x_err = adj_f * ake, adj_v_err = adj_f * akd, adj_w_err = adj_t * (akd * ads);
  qt adj_r_err = Q4(0, 0, 0, 0), adj_q_p = Q4(0, 0, 0, 0), adj_q_c = Q4(0, 0, 0, 0);
  const float adj_jf = dot(adj_t, R.axis_p);
  v3 adj_axis_p = adj_t * R.jf, adj_axis_c = V3(0, 0, 0);
  float adj_qd = -adj_jf * (akd * ads);
  adj_axis_p += adj_t * (-R.qd * (akd * ads));
  adj_cross(R.axis_p, R.axis_c, adj_axis_p, adj_axis_c, adj_t * ake);
  float adj_q = 0.f;
  a_tgt = 0.f; a_act = 0.f; a_ke = 0.f; a_kd = 0.f;
  const JointLimit L = c.lim[0];
  joint_force_adj(R.q, R.qd, tgt, ke, kd, L.lo, L.up, L.ke, L.kd, adj_jf, adj_q, adj_qd, a_tgt, a_ke, a_kd, a_act);
  adj_w_err += R.axis_p * adj_qd; adj_axis_p += w_err * adj_qd;
  const float adj_da = adj_q * R.dq_dda;
  adj_r_err.x += c.axis.x * adj_da; adj_r_err.y += c.axis.y * adj_da; adj_r_err.z += c.axis.z * adj_da;
  adj_r_err.w += adj_q * R.dq_dw;
  adj_qrot_q(R.q_p, c.axis, adj_q_p, adj_axis_p);
  add_outer(aR, adj_axis_c, c.axis);  // axis_c = rotm(s.r) axis
  {  // r_err = conj(q_p) * q_c
    qt adj_cqp = Q4(0, 0, 0, 0);
    adj_qmul(qconj(R.q_p), s.r, adj_cqp, adj_q_c, adj_r_err);
    adj_q_p += qconj(adj_cqp);
  }
  add_outer(aR, -adj_r_c, c.com);     // rc_c = rotm(s.r) com
  own.p += adj_x_err; own.r += adj_q_c; own.w += adj_w_err; own.v += adj_v_err;
  par = adj_zero();
  if (HP || c.parent >= 0) {
    v3 adj_x_p = adj_r_p - adj_x_err;
    par.p = adj_x_p - adj_r_p;
    float aP[9];  // matrix adjoint of rotm(qp): rc_par = rotm(qp) com_par, x_p = pp + rotm(qp) p_pj
#pragma unroll
    for (int k = 0; k < 9; ++k) aP[k] = 0.f;
    add_outer(aP, -adj_r_p, c.com_par);
    add_outer(aP, adj_x_p, c.p_pj);
    rotm_adj(qp, aP, par.r);
    adj_qmul_a(c.q_pj, par.r, adj_q_p);
    par.w = -adj_w_err; par.v = -adj_v_err;
  }
}

template <bool HP = false>
PD_DEV void rev_adjoint(const PdDevModel &m, const BodyConst &c, const BodyState &s, v3 rc_c, const float *rec, const RevCache &R,
                        float tgt, float ke, float kd, v3 gc_t, v3 gc_f, v3 gp_t, v3 gp_f, BodyAdj &own, BodyAdj &par, float *aR, float &a_tgt,
                        float &a_act, float &a_ke, float &a_kd) {
  v3 pp = V3(0, 0, 0), w_p = pp, v_p = pp, rc_par = pp;
  qt qp = Q4(0, 0, 0, 1);
  if (HP || c.parent >= 0) {
    const float *r = rec + (HP ? c.pidx : c.parent) * PD_REC;  // (pidx: the call may run for a lane without a joint, see the CLONE kernels)
    pp = ld3(r); qp = ld4(r + 3); w_p = ld3(r + 7); v_p = ld3(r + 10); rc_par = ld3(r + 13);
  }
  rev_adjoint_core<HP>(m, c, s, rc_c, pp, qp, w_p, v_p, rc_par, R, tgt, ke, kd, gc_t, gc_f, gp_t, gp_f, own, par, aR, a_tgt, a_act, a_ke, a_kd);
}

// ---------------------------------------------------------------------------------------------
// integrate_bodies (integrator_euler.py:21-91) for one body.
// rc = rot(q, com) of the input state (from the staging); rc_out = the same for the returned state.
// sink_rate: bound on how fast any contact candidate of the body can have lost height over this step,
// |v1_y| + |w1|_1 * reach (the pose update uses the unclamped v1, w1), for the speculative contact cull.
// clamp_mask: bit k set when component k of (w, v) was clamped (:78-88).  The forward kernel stores it beside the step's wrench and
// the adjoint takes the clamp's pass / block decision from it -- recomputing w1 there through rotm(q) instead of qrot can land
// on the other side of +-10 by an ulp, and a rollout that sits on the clamps (a robot dropped into the ground) then differentiates
// a different function (found by the randomised sweep: one env of 3 200 off by 6 % in every gradient).
template <bool LEAN = false>  // LEAN (round 6): qmul_pure for the quaternion update (the wave-specialised forward kernels of plain models)
PD_DEV BodyState integrate_fwd(const PdDevModel &m, const BodyConst &c, const BodyState &s, const float *Rm, v3 rc, v3 t0, v3 f0, float inv_m,
                               const float *I, const float *invI, float dt, float *R1, v3 &rc_out, float &sink_rate, unsigned &clamp_mask) {
  // Rm = rotm(s.r) (pd_math.h): the four rotations by the body's quaternion are matrix products, exactly as the adjoint
  // recomputes them (integrate_adj2); R1 = rotm of the new quaternion, for the staging, the joints and the next step
  v3 g = V3(m.gx, m.gy, m.gz);
  float nz = inv_m != 0.0f ? 1.0f : 0.0f;
  v3 x_com = s.p + rc;                                          // :61
  v3 v1 = s.v + (f0 * inv_m + g * nz) * dt;                     // :64
  v3 x1 = x_com + v1 * dt;                                      // :65
  v3 wb = matT_vec(Rm, s.w);                                    // :68
  v3 tb = matT_vec(Rm, t0) - cross(wb, mat_vec(I, wb));         // :69
  v3 w1 = mat_vec(Rm, wb + mat_vec(invI, tb) * dt);             // :71
  qt r1 = qnormalize(s.r + (LEAN ? qmul_pure(w1, s.r) : qmul(Q4(w1.x, w1.y, w1.z, 0.f), s.r)) * (0.5f * dt));  // :72
  sink_rate = fabsf(v1.y) + (fabsf(w1.x) + fabsf(w1.y) + fabsf(w1.z)) * c.reach;
  w1 = w1 * (1.0f - 0.1f * dt);                                 // :75
  BodyState o;
  if constexpr (LEAN) {
    // the same clamp and the same mask from ONE compare per component: x < -10 || x > 10 is |x| > 10 (false for NaN either way: it passes
    // unclamped), the clamped value copysign(10, x) -- 4 instructions per component instead of 6.5, one condition register live at a time
    // instead of twelve (the scalar registers the compare pairs took were spilled around the loop)
    const float xs[6] = {w1.x, w1.y, w1.z, v1.x, v1.y, v1.z};
    float ys[6];
    clamp_mask = 0u;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const bool hit = fabsf(xs[k]) > 10.0f;
      ys[k] = hit ? copysignf(10.0f, xs[k]) : xs[k];
      clamp_mask |= hit ? (1u << k) : 0u;
    }
    o.w = V3(ys[0], ys[1], ys[2]); o.v = V3(ys[3], ys[4], ys[5]);
  } else {
  o.w = clamp3(w1, 10.0f); o.v = clamp3(v1, 10.0f);             // :78-88
  clamp_mask = (clamp_pass(w1.x, -10.0f, 10.0f) == 0.0f ? 1u : 0u) | (clamp_pass(w1.y, -10.0f, 10.0f) == 0.0f ? 2u : 0u) |
               (clamp_pass(w1.z, -10.0f, 10.0f) == 0.0f ? 4u : 0u) | (clamp_pass(v1.x, -10.0f, 10.0f) == 0.0f ? 8u : 0u) |
               (clamp_pass(v1.y, -10.0f, 10.0f) == 0.0f ? 16u : 0u) | (clamp_pass(v1.z, -10.0f, 10.0f) == 0.0f ? 32u : 0u);
  }
  rotm(r1, R1);
  rc_out = mat_vec(R1, c.com);
  o.r = r1; o.p = x1 - rc_out;                                  // :90
  return o;
}

// Two phases: everything the wrench adjoint (adj_t0, adj_f0) needs comes first and is handed to `wrench_ready` -- the
// role-split adjoint kernel publishes it to the joint and contact waves there -- then the rest (state adjoint, inertia
// and inverse-mass gradients).  Terms and accumulation order are those of the plain reverse sweep.
// g_I / g_invI: 9-float accumulators, either registers (float *) or LDS (LdsAcc9: read - fma - write, same rounding).
#define PD_GACC 37  // LDS floats per body: the two 9-float accumulators + inertia + inverse inertia (odd stride)
struct LdsAcc9 { float *p; };
PD_DEV void add_outer(LdsAcc9 M, v3 a, v3 b) {
  float t[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) t[k] = M.p[k];
  add_outer(t, a, b);
#pragma unroll
  for (int k = 0; k < 9; ++k) M.p[k] = t[k];
}
template <typename ACC, typename F>
PD_DEV void integrate_adj2(const PdDevModel &m, const BodyConst &c, const BodyState &s, const float *Rm, unsigned clamp_mask, v3 t0, v3 f0, float inv_m,
                           const float *I, const float *invI, float dt, const BodyAdj &gn, BodyAdj &a, float *aR, float &g_inv_m, ACC g_I,
                           ACC g_invI, F &&wrench_ready) {
  // Rm = rotm(s.r): the six rotations by the body's own quaternion are matrix products, and their quaternion adjoints are
  // accumulated as ONE matrix adjoint in aR (+=), which the caller converts once with rotm_adj(s.r, aR, ...) -- after the
  // joint adjoint has added its own rotations by s.r where that runs on the same wave (pd_math.h)
  v3 wb = matT_vec(Rm, s.w);
  v3 Iwb = mat_vec(I, wb);
  v3 tb = matT_vec(Rm, t0) - cross(wb, Iwb);
  v3 u = wb + mat_vec(invI, tb) * dt;
  v3 w1 = mat_vec(Rm, u);
  qt W = Q4(w1.x, w1.y, w1.z, 0.f);
  qt rq = s.r + qmul(W, s.r) * (0.5f * dt);
  qt r1 = qnormalize(rq);
  // ---- reverse, phase 1: the path to the wrench adjoint
  qt adj_r1 = gn.r;
  adj_qrot_q(r1, c.com, adj_r1, -gn.p);
  // clamp adjoints: pass or block exactly as the forward pass clamped (its mask, integrate_fwd)
  v3 adj_v1 = V3((clamp_mask & 8u) ? 0.0f * gn.v.x : gn.v.x, (clamp_mask & 16u) ? 0.0f * gn.v.y : gn.v.y, (clamp_mask & 32u) ? 0.0f * gn.v.z : gn.v.z);
  v3 adj_w1 = V3((clamp_mask & 1u) ? 0.0f * gn.w.x : gn.w.x, (clamp_mask & 2u) ? 0.0f * gn.w.y : gn.w.y, (clamp_mask & 4u) ? 0.0f * gn.w.z : gn.w.z) *
              (1.0f - 0.1f * dt);
  qt adj_rq = Q4(0, 0, 0, 0);
  adj_qnormalize(rq, adj_rq, adj_r1);
  const qt gW = adj_rq * (0.5f * dt);
  qt adj_W = Q4(0, 0, 0, 0);
  adj_qmul_a(s.r, adj_W, gW);
  adj_w1 += qvec(adj_W);
  const v3 adj_u = matT_vec(Rm, adj_w1);
  v3 adj_wb = adj_u, adj_a = adj_u * dt;
  v3 adj_tb = matT_vec(invI, adj_a);
  const v3 adj_t0 = mat_vec(Rm, adj_tb);
  adj_v1 += gn.p * dt;
  const v3 adj_f0 = adj_v1 * (inv_m * dt);
  wrench_ready(adj_t0, adj_f0);
  // ---- phase 2
  qt adj_r0 = adj_rq;
  adj_qmul_b(W, adj_r0, gW);
  add_outer(aR, adj_w1, u);    // w1 = Rm u
  add_outer(g_invI, adj_a, tb);
  add_outer(aR, t0, adj_tb);   // Rm^T t0
  v3 adj_Iwb = V3(0, 0, 0);
  adj_cross(wb, Iwb, adj_wb, adj_Iwb, -adj_tb);
  add_outer(g_I, adj_Iwb, wb);
  adj_wb += matT_vec(I, adj_Iwb);
  const v3 adj_w0 = mat_vec(Rm, adj_wb);
  add_outer(aR, s.w, adj_wb);  // wb = Rm^T w
  g_inv_m += dot(adj_v1, f0) * dt;
  add_outer(aR, gn.p, c.com);  // x_com = p + Rm com
  a.p = gn.p; a.r = adj_r0; a.w = adj_w0; a.v = adj_v1;
}

PD_DEV void integrate_adj(const PdDevModel &m, const BodyConst &c, const BodyState &s, const float *Rm, unsigned clamp_mask, v3 t0, v3 f0, float inv_m,
                          const float *I, const float *invI, float dt, const BodyAdj &gn, BodyAdj &a, float *aR, v3 &adj_t0, v3 &adj_f0,
                          float &g_inv_m, float *g_I, float *g_invI) {
  integrate_adj2(m, c, s, Rm, clamp_mask, t0, f0, inv_m, I, invI, dt, gn, a, aR, g_inv_m, g_I, g_invI, [&](v3 t, v3 f) { adj_t0 = t; adj_f0 = f; });
}
