// Four lanes per body ("quad-lane") arithmetic for the small-batch rollout kernels ("latency path").
//
// Below ~2 048 envs a rollout step is ONE wave's instruction stream: the lane-per-body kernels run ~760 (forward) / ~1 360 (adjoint)
// instructions per step on the body wave at 5-6 cycles each while three of four SIMDs idle.  Here a body is FOUR lanes -- lane
// 4 b + c holds component c (x, y, z, w) of every vector / quaternion of body b -- so one env fills a wave (13 Laikago bodies = 52
// lanes) and the stream gets shorter: a vector sum / scale is 1 instruction instead of 3, a 3 x 3 product 3 (v_fmac_f32 with a DPP
// quad_perm broadcast of the vector's component) instead of 9, a cross product 3 instead of 6, a quaternion product 7 instead of 16.
// Scalars of a body (joint angle, gains, masses) are computed redundantly in its four lanes: transcendental chains do not shrink.
// Vectors keep 0 in lane 3.  Semantics: /root/reference/diffphys/integrator_euler.py, cited per function like pd_device.h.
#pragma once
#include "pd_device.h"

#define PD_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
template <int CTRL>
PD_DEV float q_dpp(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true)); }
template <int CTRL>
PD_DEV unsigned q_dppu(unsigned x) { return (unsigned)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xf, 0xf, true); }
#define Q_BC0(x) q_dpp<PD_QP(0, 0, 0, 0)>(x)
#define Q_BC1(x) q_dpp<PD_QP(1, 1, 1, 1)>(x)
#define Q_BC2(x) q_dpp<PD_QP(2, 2, 2, 2)>(x)
#define Q_BC3(x) q_dpp<PD_QP(3, 3, 3, 3)>(x)
#define Q_ROT1(x) q_dpp<PD_QP(1, 2, 0, 3)>(x)  // lane c reads component c + 1 (mod 3); lane 3 itself

// Per-lane selectors / signs of component c (registers for the whole rollout).
struct QLane {
  int c;
  bool isv;                         // c < 3: this lane holds a vector component
  float two, m3, d0, d1, d2;        // 2 (0 in lane 3); 1 in lane 3 only; delta(c, j)
  float sg0, sg1, sg2;              // sign of the w term of R[c][j]
  float s1, s2, s3, sc;             // quaternion product signs (see q_qmul), conjugate
};
PD_DEV QLane q_lane(int c) {
  QLane k;
  k.c = c; k.isv = c < 3;
  k.two = c < 3 ? 2.f : 0.f; k.m3 = c == 3 ? 1.f : 0.f;
  k.d0 = c == 0 ? 1.f : 0.f; k.d1 = c == 1 ? 1.f : 0.f; k.d2 = c == 2 ? 1.f : 0.f;
  // R01 = -wz, R02 = +wy, R10 = +wz, R12 = -wx, R20 = -wy, R21 = +wx; lane 3 holds a zero row
  k.sg0 = c == 1 ? 1.f : (c == 2 ? -1.f : 0.f);
  k.sg1 = c == 0 ? -1.f : (c == 2 ? 1.f : 0.f);
  k.sg2 = c == 0 ? 1.f : (c == 1 ? -1.f : 0.f);
  k.s1 = (c == 0 || c == 2) ? 1.f : -1.f;   // (+, -, +, -)
  k.s2 = c < 2 ? 1.f : -1.f;                // (+, +, -, -)
  k.s3 = (c == 1 || c == 2) ? 1.f : -1.f;   // (-, +, +, -)
  k.sc = c == 3 ? 1.f : -1.f;               // (-, -, -, +)
  return k;
}
PD_DEV float q_pick(const QLane &k, v3 a) { return k.c == 0 ? a.x : (k.c == 1 ? a.y : (k.c == 2 ? a.z : 0.f)); }
PD_DEV float q_pick(const QLane &k, qt a) { return k.c == 0 ? a.x : (k.c == 1 ? a.y : (k.c == 2 ? a.z : a.w)); }

PD_DEV float q_sum3(float p) { return Q_BC0(p) + Q_BC1(p) + Q_BC2(p); }                    // every lane: sum over x, y, z
PD_DEV float q_sum4(float p) { return (Q_BC0(p) + Q_BC1(p)) + (Q_BC2(p) + Q_BC3(p)); }
PD_DEV float q_dot3(float a, float b) { return q_sum3(a * b); }
// (a x b)_c: t_c = a_c b_{c+1} - a_{c+1} b_c, then (a x b)_c = t_{c+1}: three instructions; lane 3 gives a_3 b_3 - a_3 b_3 = 0
PD_DEV float q_cross(float a, float b) { const float t = a * Q_ROT1(b) - Q_ROT1(a) * b; return Q_ROT1(t); }

struct QM3 { float a, b, c; };  // what lane r holds of a 3 x 3 matrix: its row r (or its column r); lane 3: zeros
// Multiply-adds whose one factor is a quad broadcast: the compiler fuses a DPP read into v_mul / v_add but not into a contracted
// fma (v_fma_f32 is VOP3, no DPP operand on gfx9) and leaves  v_mov_b32_dpp + v_fma  per term -- 133 DPP moves in the adjoint's step
// loop.  v_fmac_f32 is VOP2 and takes the DPP read itself; written as asm, with the s_nop 1 the DPP read-after-VALU-write hazard
// wants in front (the hazard recognizer does not look into asm).  Term order = the order the contracted expression had.
#define Q_QP_(a, b, c, d) " quad_perm:[" #a "," #b "," #c "," #d "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define Q_S_BC0 Q_QP_(0, 0, 0, 0)
#define Q_S_BC1 Q_QP_(1, 1, 1, 1)
#define Q_S_BC2 Q_QP_(2, 2, 2, 2)
#define Q_S_BC3 Q_QP_(3, 3, 3, 3)
#ifndef PD_QUAD_NO_ASM
PD_DEV float q_mv(QM3 m, float v) {   // lane r: sum_j m[r][j] v_j
  float r;
  asm("s_nop 1\n\tv_mul_f32_dpp %0, %1, %3" Q_S_BC1 "v_fmac_f32_dpp %0, %1, %2" Q_S_BC0 "v_fmac_f32_dpp %0, %1, %4" Q_S_BC2
      : "=&v"(r) : "v"(v), "v"(m.a), "v"(m.b), "v"(m.c));
  return r;
}
#else
PD_DEV float q_mv(QM3 m, float v) { return m.a * Q_BC0(v) + m.b * Q_BC1(v) + m.c * Q_BC2(v); }           // lane r: sum_j m[r][j] v_j
#endif
PD_DEV float q_mvc(QM3 m, float v0, float v1, float v2) { return m.a * v0 + m.b * v1 + m.c * v2; }      // v: a per-body constant

// The three permuted, signed copies of b that a Hamilton product a * b needs (reusable when b meets several a):
//   c = a.w b + a.x P1 + a.y P2 + a.z P3,  P1 = (b.w, -b.z, b.y, -b.x), P2 = (b.z, b.w, -b.x, -b.y), P3 = (-b.y, b.x, b.w, -b.z)
struct QPerm { float p1, p2, p3; };
PD_DEV QPerm q_perm(const QLane &k, float b) {
  QPerm P;
  P.p1 = q_dpp<PD_QP(3, 2, 1, 0)>(b) * k.s1; P.p2 = q_dpp<PD_QP(2, 3, 0, 1)>(b) * k.s2; P.p3 = q_dpp<PD_QP(1, 0, 3, 2)>(b) * k.s3;
  return P;
}
#ifndef PD_QUAD_NO_ASM
PD_DEV float q_qmul(float a, float b, const QPerm &P) {
  float r;
  asm("s_nop 1\n\tv_mul_f32_dpp %0, %1, %3" Q_S_BC0 "v_fmac_f32_dpp %0, %1, %2" Q_S_BC3 "v_fmac_f32_dpp %0, %1, %4" Q_S_BC1 "v_fmac_f32_dpp %0, %1, %5" Q_S_BC2
      : "=&v"(r) : "v"(a), "v"(b), "v"(P.p1), "v"(P.p2), "v"(P.p3));
  return r;
}
#else
PD_DEV float q_qmul(float a, float b, const QPerm &P) { return Q_BC3(a) * b + Q_BC0(a) * P.p1 + Q_BC1(a) * P.p2 + Q_BC2(a) * P.p3; }
#endif
PD_DEV float q_qmul(const QLane &k, float a, float b) { return q_qmul(a, b, q_perm(k, b)); }
// conj(a) * b = a.w b - (a.x P1 + a.y P2 + a.z P3)
#ifndef PD_QUAD_NO_ASM
PD_DEV float q_qmul_conj(float a, float b, const QPerm &P) {
  float r;
  asm("s_nop 1\n\tv_mul_f32_dpp %0, %1, -%3" Q_S_BC1 "v_fmac_f32_dpp %0, %1, -%2" Q_S_BC0 "v_fmac_f32_dpp %0, %1, -%4" Q_S_BC2 "v_fmac_f32_dpp %0, %1, %5" Q_S_BC3
      : "=&v"(r) : "v"(a), "v"(P.p1), "v"(P.p2), "v"(P.p3), "v"(b));
  return r;
}
#else
PD_DEV float q_qmul_conj(float a, float b, const QPerm &P) { return Q_BC3(a) * b - (Q_BC0(a) * P.p1 + Q_BC1(a) * P.p2 + Q_BC2(a) * P.p3); }
#endif

// quat_rotate (pd_math.h qrot): v (2 w^2 - 1) + 2 w (u x v) + 2 u (u . v); v in lanes (lane 3: 0), result lane 3: 0
PD_DEV float q_qrot(const QLane &k, float q, float v) {
  const float w = Q_BC3(q);
  return v * (2.0f * w * w - 1.0f) + q_cross(q, v) * (2.0f * w) + (k.two * q) * q_dot3(q, v);
}

// Rows AND columns of rotm(q) for lane r:  R[r][j] = 2 q_r q_j + sgn(r, j) 2 w q_k + delta_rj (2 w^2 - 1), columns the same with -w.
// Roundings PINNED (explicit fma, nothing else contracted) so that ROW 1 comes out bit for bit as pd_math.h rot_row1 computes it:
// the contact wave tests heights against the cull vector (p_y, row 1) this kernel stages, and the adjoint kernel recomputes that
// row from the stored quaternion with rot_row1 -- both must land on the same side of 0.
// (2 q_x) q_y and (2 q_y) q_x are the same product exactly, (2 w) q_z and (2 q_z) w round alike: the generic formula IS rot_row1's.)
PD_DEV void q_rotm(const QLane &k, float q, QM3 &row, QM3 &col) {
#pragma clang fp contract(off)
  const float w = Q_BC3(q), w2 = 2.0f * w, s = __builtin_fmaf(w2, w, -1.0f), q2 = k.two * q;
  const float b0 = k.sg0 * (w2 * q_dpp<PD_QP(0, 2, 1, 3)>(q)), b1 = k.sg1 * (w2 * q_dpp<PD_QP(2, 1, 0, 3)>(q)),
              b2 = k.sg2 * (w2 * q_dpp<PD_QP(1, 0, 2, 3)>(q));
  const float x = Q_BC0(q), y = Q_BC1(q), z = Q_BC2(q);
  row.a = __builtin_fmaf(q2, x, __builtin_fmaf(k.d0, s, b0)); row.b = __builtin_fmaf(q2, y, __builtin_fmaf(k.d1, s, b1));
  row.c = __builtin_fmaf(q2, z, __builtin_fmaf(k.d2, s, b2));
  col.a = __builtin_fmaf(q2, x, __builtin_fmaf(k.d0, s, -b0)); col.b = __builtin_fmaf(q2, y, __builtin_fmaf(k.d1, s, -b1));
  col.c = __builtin_fmaf(q2, z, __builtin_fmaf(k.d2, s, -b2));
}

// Per-body constants of lane (b, c).
struct QBody {
  int type, pidx, qdstart;
  bool joint;                       // a REVOLUTE joint hangs this body on body pidx (plain model: pd_parented)
  float com, com0, com1, com2;      // centre of mass: component c / replicated
  float axis, ax0, ax1, ax2, alen;  // joint axis
  float p_pj, q_pj;                 // parent-frame joint transform, component c
  QPerm pj;                         // permuted copies of q_pj: q_p = qp * q_pj costs four instructions
  float g;                          // gravity component
  float reach, sphere_w;
  QM3 I, invI;                      // row c of the inertia / inverse inertia (lane 3: 0)
  float inv_m, ke, kd;
  JointLimit lim;
};

struct QState { float p, r, w, v; };  // component c of position / quaternion / angular / linear velocity (vectors: lane 3 holds 0)

// integrate_bodies (integrator_euler.py:21-91), pd_device.h integrate_fwd in quad-lane form.  Rr / Rc: rows / columns of rotm(s.r);
// rc = R com.  mask: this body's 6-bit clamp mask (all four lanes).  sink: see integrate_fwd.
PD_DEV QState q_integrate(const QLane &k, const QBody &B, QState s, QM3 Rr, QM3 Rc, float rc, float t0, float f0, float dt, QM3 &R1r, QM3 &R1c,
                          float &rc_out, float &sink, unsigned &mask) {
  const float nz = B.inv_m != 0.0f ? 1.0f : 0.0f;
  const float x_com = s.p + rc;                                // :61
  const float v1 = s.v + (f0 * B.inv_m + B.g * nz) * dt;       // :64
  const float x1 = x_com + v1 * dt;                            // :65
  const float wb = q_mv(Rc, s.w);                              // :68  R^T w
  const float tb = q_mv(Rc, t0) - q_cross(wb, q_mv(B.I, wb));  // :69
  const float u = wb + q_mv(B.invI, tb) * dt;
  float w1 = q_mv(Rr, u);                                      // :71  (lane 3: its "row" is zero)
  // :72  quat(w1, 0) * r: xyz = r.w w1 + w1 x r_v, w = -(w1 . r_v)
  const float qm = Q_BC3(s.r) * w1 + q_cross(w1, s.r) - k.m3 * q_sum3(w1 * s.r);
  const float rq = s.r + qm * (0.5f * dt);
  const float r1 = rq * rcp_hw(sqrt_hw(q_sum4(rq * rq)));
  sink = Q_BC1(fabsf(v1)) + q_sum3(fabsf(w1)) * B.reach;
  w1 = w1 * (1.0f - 0.1f * dt);                                // :75
  QState o;
  o.w = clampf(w1, -10.0f, 10.0f); o.v = clampf(v1, -10.0f, 10.0f);   // :78-88
  unsigned mk = (clamp_pass(w1, -10.0f, 10.0f) == 0.0f ? (1u << k.c) : 0u) | (clamp_pass(v1, -10.0f, 10.0f) == 0.0f ? (8u << k.c) : 0u);
  mk = k.isv ? mk : 0u;
  mk |= q_dppu<PD_QP(1, 0, 3, 2)>(mk);
  mk |= q_dppu<PD_QP(2, 3, 0, 1)>(mk);
  mask = mk;
  q_rotm(k, r1, R1r, R1c);
  rc_out = q_mvc(R1r, B.com0, B.com1, B.com2);
  o.r = r1; o.p = x1 - rc_out;                                 // :90
  return o;
}

// eval_body_joints for a REVOLUTE joint of a plain model (integrator_euler.py:289-409; pd_device.h joint_ctx + joint_fwd), quad-lane.
// Parent quantities (component c of each, vectors with 0 in lane 3) come from the staged record.  Outputs the wrench pair
// parent += (wp_t, f), child -= (wc_t, f)   (:448-451).
PD_DEV void q_joint_fwd(const QLane &k, const QBody &B, const QState &s, QM3 Rr, float rc, float pp, float qp, float w_p, float v_p, float rc_par,
                        float tgt, float act, float ake, float akd, float &wp_t, float &wc_t, float &f_out) {
  const float ads = 0.01f;
  const float x_p = pp + q_qrot(k, qp, B.p_pj);                 // :327
  const float q_p = q_qmul(qp, B.q_pj, B.pj);
  const float r_p = x_p - (pp + rc_par);                        // :328
  const float r_c = s.p - (s.p + rc);                           // :338
  const float x_err = s.p - x_p;                                // :369
  const QPerm Pr = q_perm(k, s.r);
  const float r_err = q_qmul_conj(q_p, s.r, Pr);                // :370
  const float v_err = s.v - v_p, w_err = s.w - w_p;             // :371-372
  const float axis_p = q_qrot(k, q_p, B.axis);                  // :392
  const float axis_c = q_mvc(Rr, B.ax0, B.ax1, B.ax2);
  const float da = q_dot3(r_err, B.axis);
  const float q = twist_angle(da, Q_BC3(r_err), B.alen);        // :394-400, pd_math.h
  const float qd = q_dot3(w_err, axis_p);
  const float jf = joint_force(q, qd, tgt, B.ke, B.kd, act, B.lim.lo, B.lim.up, B.lim.ke, B.lim.kd);   // :403
  const float swing = q_cross(axis_p, axis_c);
  const float f = x_err * ake + v_err * akd;                    // :406
  const float t = axis_p * jf + (swing * ake + (w_err - axis_p * qd) * (akd * ads));   // :403-409
  wp_t = t + q_cross(r_p, f);
  wc_t = t + q_cross(r_c, f);
  f_out = f;
}

// ================================================================================================= adjoints, quad-lane
// Body-state adjoint, component c of each part (vectors: lane 3 holds 0).
struct QAdj { float p, r, w, v; };

// adj_q += d <g, qrot(q, v)> / dq   (pd_math.h adj_qrot_q): lanes 0-2 the vector part, lane 3 the scalar part
PD_DEV float q_adj_qrot_q(const QLane &k, float q, float v, float g) {
  const float w = Q_BC3(q);
  const float uv = q_dot3(q, v), ug = q_dot3(q, g), vg = q_dot3(v, g);
  const float au = q_cross(v, g) * (2.0f * w) + (g * uv + v * ug) * 2.0f;
  const float aw = 4.0f * w * vg + 2.0f * q_dot3(q_cross(q, v), g);
  return k.isv ? au : aw;
}

// d <A, rotm(q)> / dq for a matrix adjoint A held by rows (lane r: A[r][0..2]); pd_math.h rotm_adj:
//   adj_u = 2 (A + A^T) u + 2 w K,   adj_w = 4 w tr(A) + 2 u . K,   K = axial(A - A^T) = (A21 - A12, A02 - A20, A10 - A01)
PD_DEV float q_rotm_adj(const QLane &k, float q, QM3 A) {
  const float w = Q_BC3(q);
  const float Au = q_mv(A, q);                                        // lane r: sum_j A[r][j] u_j   (lane 3: zero row)
  // sum_j A[j][r] u_j: all three column sums on every lane (a DPP read must see the whole quad active), the lane's own picked after
  const float c0 = q_sum3(q * A.a), c1 = q_sum3(q * A.b), c2 = q_sum3(q * A.c);
  const float ATu = k.c == 0 ? c0 : (k.c == 1 ? c1 : c2);
  const float m_next = k.c == 0 ? A.b : (k.c == 1 ? A.c : A.a);       // A[r][r+1]
  const float m_prev = k.c == 0 ? A.c : (k.c == 1 ? A.a : A.b);       // A[r][r+2]
  const float K = q_dpp<PD_QP(2, 0, 1, 3)>(m_prev) - Q_ROT1(m_next);  // lane c: A[c+2][c+1] - A[c+1][c+2]
  const float dg = k.c == 0 ? A.a : (k.c == 1 ? A.b : A.c);           // A[r][r]
  const float au = 2.0f * (Au + ATu) + (2.0f * w) * K;
  const float aw = 4.0f * w * q_sum3(dg) + 2.0f * q_dot3(q, K);
  return k.isv ? au : aw;
}

#ifndef PD_QUAD_NO_ASM
PD_DEV void q_add_outer(QM3 &M, float a, float b) {   // M[r][j] += a_r b_j
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %3, %4" Q_S_BC0 "v_fmac_f32_dpp %1, %3, %4" Q_S_BC1 "v_fmac_f32_dpp %2, %3, %4" Q_S_BC2
      : "+v"(M.a), "+v"(M.b), "+v"(M.c) : "v"(b), "v"(a));
}
#else
PD_DEV void q_add_outer(QM3 &M, float a, float b) { M.a += a * Q_BC0(b); M.b += a * Q_BC1(b); M.c += a * Q_BC2(b); }   // M[r][j] += a_r b_j
#endif
PD_DEV void q_add_outer_c(QM3 &M, float a, float b0, float b1, float b2) { M.a += a * b0; M.b += a * b1; M.c += a * b2; }  // b: per-body constant

// Adjoint of integrate_bodies (pd_device.h integrate_adj2), quad-lane.  Rr / Rc rows / columns of rotm(s.r); It / invIt: the
// transposed inertia / inverse inertia (row c of the transpose); mask: the forward pass's clamp mask; gn: adjoint of the next state.
// q_integrate_adj_pre recomputes the forward values, phase 1 (q_integrate_adj_wrench) ends with the wrench adjoint (adj_t0, adj_f0),
// phase 2 (q_integrate_adj_rest) needs nothing from other waves.
struct QIntTmp { float wb, Iwb, tb, u, w1, adj_w1, adj_v1, adj_rq, gW, adj_tb, adj_a, adj_wb, il, r1; };
// The forward values the adjoint needs again (no adjoint enters): the kernel runs this one step AHEAD of the reverse part, while it
// would otherwise wait for the contact adjoints.
PD_DEV void q_integrate_adj_pre(const QLane &k, const QBody &B, const QState &s, QM3 Rr, QM3 Rc, float t0, float dt, QIntTmp &T) {
  T.wb = q_mv(Rc, s.w);
  T.Iwb = q_mv(B.I, T.wb);
  T.tb = q_mv(Rc, t0) - q_cross(T.wb, T.Iwb);
  T.u = T.wb + q_mv(B.invI, T.tb) * dt;
  T.w1 = q_mv(Rr, T.u);
  const float qm = Q_BC3(s.r) * T.w1 + q_cross(T.w1, s.r) - k.m3 * q_sum3(T.w1 * s.r);   // quat(w1, 0) * r
  const float rq = s.r + qm * (0.5f * dt);
  T.il = rcp_hw(sqrt_hw(q_sum4(rq * rq)));
  T.r1 = rq * T.il;
}
PD_DEV void q_integrate_adj_wrench(const QLane &k, const QBody &B, const QState &s, QM3 Rr, QM3 Rc, QM3 invIt, unsigned mask, float dt,
                                   const QAdj &gn, QIntTmp &T, float &adj_t0, float &adj_f0) {
  const float il = T.il, r1 = T.r1;
  // ---- reverse
  const float adj_r1 = gn.r + q_adj_qrot_q(k, r1, B.com, -gn.p);                        // p1 = x1 - rot(r1, com)
  T.adj_v1 = (mask & (8u << k.c)) ? 0.0f * gn.v : gn.v;                                 // clamp adjoints: the forward pass's decisions
  T.adj_w1 = ((mask & (1u << k.c)) ? 0.0f * gn.w : gn.w) * (1.0f - 0.1f * dt);
  T.adj_rq = (adj_r1 - r1 * q_sum4(r1 * adj_r1)) * il;                                  // adj_qnormalize
  T.gW = T.adj_rq * (0.5f * dt);
  {  // adj_W = gW * conj(r); its vector part goes to w1
    const float cr = s.r * k.sc;
    const float aW = q_qmul(T.gW, cr, q_perm(k, cr));
    T.adj_w1 += k.isv ? aW : 0.f;
  }
  const float adj_u = q_mv(Rc, T.adj_w1);
  T.adj_wb = adj_u; T.adj_a = adj_u * dt;
  T.adj_tb = q_mv(invIt, T.adj_a);
  adj_t0 = q_mv(Rr, T.adj_tb);
  T.adj_v1 += gn.p * dt;
  adj_f0 = T.adj_v1 * (B.inv_m * dt);
}
PD_DEV void q_integrate_adj_rest(const QLane &k, const QBody &B, const QState &s, QM3 Rr, QM3 It, float t0, float f0, float dt, const QAdj &gn,
                                 QIntTmp &T, QAdj &a, QM3 &aR, float &g_inv_m, QM3 &g_I, QM3 &g_invI) {
  {  // adj_r0 = adj_rq + conj(W) * gW,  W = (w1, 0)
    a.r = T.adj_rq + q_qmul_conj(T.w1, T.gW, q_perm(k, T.gW));
  }
  q_add_outer(aR, T.adj_w1, T.u);       // w1 = R u
  q_add_outer(g_invI, T.adj_a, T.tb);
  q_add_outer(aR, t0, T.adj_tb);        // R^T t0
  const float g = -T.adj_tb;
  T.adj_wb += q_cross(T.Iwb, g);        // adj_cross(wb, Iwb, adj_wb, adj_Iwb, -adj_tb)
  const float adj_Iwb = q_cross(g, T.wb);
  q_add_outer(g_I, adj_Iwb, T.wb);
  T.adj_wb += q_mv(It, adj_Iwb);
  a.w = q_mv(Rr, T.adj_wb);
  q_add_outer(aR, s.w, T.adj_wb);       // wb = R^T w
  g_inv_m += q_dot3(T.adj_v1, f0) * dt;
  q_add_outer_c(aR, gn.p, B.com0, B.com1, B.com2);   // x_com = p + R com
  a.p = gn.p; a.v = T.adj_v1;
}

// What the contact wave recomputed of the joint's forward pass (pd_device.h RevCache), component c of the vector parts
struct QRev { float q_p, r_err, x_p, axis_p, axis_c, q, qd, jf, dq_dda, dq_dw; };
PD_DEV QRev q_rev_load(const QLane &k, const float *d, int qv) {
  QRev R;
  R.q_p = d[k.c]; R.r_err = d[4 + k.c];
  const float xp = d[8 + qv], ap = d[11 + qv], ac = d[14 + qv];
  R.x_p = k.isv ? xp : 0.f; R.axis_p = k.isv ? ap : 0.f; R.axis_c = k.isv ? ac : 0.f;
  R.q = d[17]; R.qd = d[18]; R.jf = d[19]; R.dq_dda = d[20]; R.dq_dw = d[21];
  return R;
}

// Adjoint of the revolute joint (pd_device.h rev_adjoint_core), plain model.  gc_* / gp_*: adjoints of the child's / the parent's
// wrench accumulators.  own += d / d(child state), par = d / d(parent state), aR += matrix adjoint of rotm(s.r).
PD_DEV void q_rev_adjoint(const QLane &k, const QBody &B, const QState &s, float rc, float pp, float qp, float w_p, float v_p, float rc_par,
                          float com_par0, float com_par1, float com_par2, float p_pj0, float p_pj1, float p_pj2, const QPerm &pjc,
                          const QRev &R, float tgt, float ake, float akd, float gc_t, float gc_f, float gp_t, float gp_f, QAdj &own, QAdj &par,
                          QM3 &aR, float &a_tgt, float &a_act, float &a_ke, float &a_kd) {
  const float ads = 0.01f;
  const float r_p = R.x_p - (pp + rc_par);
  const float r_c = s.p - (s.p + rc);
  const float x_err = s.p - R.x_p, v_err = s.v - v_p, w_err = s.w - w_p;
  const float f_total = x_err * ake + v_err * akd;
  float adj_t = gp_t - gc_t, adj_f = gp_f - gc_f;
  const float ngc = -gc_t;
  const float adj_r_c = q_cross(f_total, ngc);            // adj_cross(r_c, f_total, adj_r_c, adj_f, -gc_t)
  adj_f += q_cross(ngc, r_c);
  const float adj_r_p = q_cross(f_total, gp_t);           // adj_cross(r_p, f_total, adj_r_p, adj_f, gp_t)
  adj_f += q_cross(gp_t, r_p);
  const float adj_x_err = adj_f * ake, adj_v_err = adj_f * akd;
  float adj_w_err = adj_t * (akd * ads);
  const float adj_jf = q_dot3(adj_t, R.axis_p);
  float adj_axis_p = adj_t * R.jf + adj_t * (-R.qd * (akd * ads));
  float adj_qd = -adj_jf * (akd * ads);
  const float gsw = adj_t * ake;                          // adj_cross(axis_p, axis_c, adj_axis_p, adj_axis_c, adj_t * ake)
  adj_axis_p += q_cross(R.axis_c, gsw);
  const float adj_axis_c = q_cross(gsw, R.axis_p);
  float adj_q = 0.f;
  a_tgt = 0.f; a_act = 0.f; a_ke = 0.f; a_kd = 0.f;
  joint_force_adj(R.q, R.qd, tgt, B.ke, B.kd, B.lim.lo, B.lim.up, B.lim.ke, B.lim.kd, adj_jf, adj_q, adj_qd, a_tgt, a_ke, a_kd, a_act);
  adj_w_err += R.axis_p * adj_qd; adj_axis_p += w_err * adj_qd;
  const float adj_r_err = k.isv ? B.axis * (adj_q * R.dq_dda) : adj_q * R.dq_dw;
  float adj_q_p = q_adj_qrot_q(k, R.q_p, B.axis, adj_axis_p);
  q_add_outer_c(aR, adj_axis_c, B.ax0, B.ax1, B.ax2);     // axis_c = rotm(s.r) axis
  {  // r_err = conj(q_p) * q_c:  adj_q_c = q_p * adj_r_err,  adj_conj(q_p) = adj_r_err * conj(q_c)
    own.r += q_qmul(R.q_p, adj_r_err, q_perm(k, adj_r_err));
    const float cr = s.r * k.sc;
    adj_q_p += q_qmul(adj_r_err, cr, q_perm(k, cr)) * k.sc;
  }
  q_add_outer_c(aR, -adj_r_c, B.com0, B.com1, B.com2);    // rc_c = rotm(s.r) com
  own.p += adj_x_err; own.w += adj_w_err; own.v += adj_v_err;
  const float adj_x_p = adj_r_p - adj_x_err;
  par.p = adj_x_p - adj_r_p;
  QM3 aP;  // matrix adjoint of rotm(qp): rc_par = rotm(qp) com_par, x_p = pp + rotm(qp) p_pj
  aP.a = adj_x_p * p_pj0 - adj_r_p * com_par0; aP.b = adj_x_p * p_pj1 - adj_r_p * com_par1; aP.c = adj_x_p * p_pj2 - adj_r_p * com_par2;
  par.r = q_rotm_adj(k, qp, aP) + q_qmul(adj_q_p, B.q_pj * k.sc, pjc);   // q_p = qp * q_pj: adj_qp += adj_q_p * conj(q_pj)
  par.w = -adj_w_err; par.v = -adj_v_err;
}
