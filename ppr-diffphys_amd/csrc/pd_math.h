// Per-lane spatial algebra for the gfx950 rollout kernels: 3-vectors, quaternions
// (x,y,z,w), and the reverse-mode rule of every primitive.  Semantics are the Warp
// built-ins the reference's kernels call: quat_rotate as
// x(2w^2-1) + 2w(q_v x x) + 2 q_v (q_v . x), Hamilton product, conjugate inverse,
// normalize(vec3)=0 at zero length, clamp/min adjoints routed to the selected argument.
// No fast-math (inf/NaN semantics are kept; the boundary scrubs NaN like dp_utils.py:43-57 of the reference).
// acos/asin clamp their argument and drop the adjoint at |x| = 1 (acos_c / inv_sqrt_1mx2 below).
#pragma once
#include <hip/hip_runtime.h>

#define PD_DEV __device__ __forceinline__
// Numeric policy of this translation unit (pd_model_set_numeric_policy, include/ppr_diffphys.h): the kernel file is compiled once per
// policy and the host picks the launchers at run time.  0 = PD_NUM_STABLE (default): the revolute twist angle through atan2 and the
// FIXED joint's angular error scale-invariantly -- the same functions as the reference's, well-conditioned in fp32.  1 = PD_NUM_LITERAL:
// the reference's text, 2 acos(twist.w) sign(..) and normalize(v) 2 acos(w) (integrator_euler.py:385-400), for side-by-side runs
// against Warp.  Nothing else differs between the two.
#ifndef PD_POLICY
#define PD_POLICY 0
#endif

struct v3 { float x, y, z; };
struct qt { float x, y, z, w; };

PD_DEV v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
PD_DEV qt Q4(float x, float y, float z, float w) { qt r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
PD_DEV v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
PD_DEV v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
PD_DEV v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
PD_DEV v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
PD_DEV v3 &operator+=(v3 &a, v3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
PD_DEV v3 &operator-=(v3 &a, v3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
PD_DEV float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PD_DEV v3 cross(v3 a, v3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// v_sqrt_f32 / v_rcp_f32 AS THEY ARE.  clang wraps the 1-ulp sqrtf and 1 / x in a rescue for denormal inputs / results (sqrt: compare, two
// selects, two v_ldexp around the instruction; 1 / x: frexp mantissa + exponent, negate, v_ldexp) -- five / four extra instructions per use,
// ~65 per kernel.  The bare instructions give the same bits for every argument in [2^-96, 2^96]; outside it a denormal argument reads as
// zero and a denormal result is flushed -- lengths, squared norms and 1 - x^2 terms never live there (1 / denormal overflows to inf either way).
PD_DEV float sqrt_hw(float x) { return __builtin_amdgcn_sqrtf(x); }
PD_DEV float rcp_hw(float x) { return __builtin_amdgcn_rcpf(x); }
PD_DEV float length(v3 a) { return sqrt_hw(dot(a, a)); }
PD_DEV v3 normalize(v3 a) { float l = length(a); return l > 0.0f ? a * rcp_hw(l) : V3(0.f, 0.f, 0.f); }

PD_DEV v3 qvec(qt q) { return V3(q.x, q.y, q.z); }
PD_DEV qt operator+(qt a, qt b) { return Q4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
PD_DEV qt operator*(qt a, float s) { return Q4(a.x * s, a.y * s, a.z * s, a.w * s); }
PD_DEV qt &operator+=(qt &a, qt b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; return a; }
PD_DEV float qdot(qt a, qt b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
PD_DEV qt qconj(qt q) { return Q4(-q.x, -q.y, -q.z, q.w); }
PD_DEV qt qmul(qt a, qt b) {
  return Q4(a.w * b.x + b.w * a.x + a.y * b.z - a.z * b.y, a.w * b.y + b.w * a.y + a.z * b.x - a.x * b.z,
            a.w * b.z + b.w * a.z + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
// qmul(Q4(a.x, a.y, a.z, 0), b) without the four products with the zero: the same value for finite b (x + 0 * y is x up to the sign of a
// zero), four instructions less on a wave whose instruction count is its run time (integrate_bodies :72, round 6)
PD_DEV qt qmul_pure(v3 a, qt b) {
  return Q4(b.w * a.x + a.y * b.z - a.z * b.y, b.w * a.y + a.z * b.x - a.x * b.z, b.w * a.z + a.x * b.y - a.y * b.x,
            -(a.x * b.x) - a.y * b.y - a.z * b.z);
}
PD_DEV qt qnormalize(qt q) { return q * rcp_hw(sqrt_hw(qdot(q, q))); }
PD_DEV v3 qrot(qt q, v3 v) {
  v3 u = qvec(q);
  return v * (2.0f * q.w * q.w - 1.0f) + cross(u, v) * (2.0f * q.w) + u * (2.0f * dot(u, v));
}
PD_DEV v3 qrot_inv(qt q, v3 v) {
  v3 u = qvec(q);
  return v * (2.0f * q.w * q.w - 1.0f) - cross(u, v) * (2.0f * q.w) + u * (2.0f * dot(u, v));
}
PD_DEV qt q_axis_angle(v3 axis, float ang) {
  float s, c;
  sincosf(ang * 0.5f, &s, &c);
  return Q4(axis.x * s, axis.y * s, axis.z * s, c);
}
// sin and cos for |x| <= pi / 2 (half of an angle that atan2 / asin returned): no range reduction, Taylor polynomials in x^2
// (absolute error 1.1e-7 / 7.5e-8 over the interval, checked against float64).  For the ADJOINT's recomputation only; the
// forward pass calls sincosf.  ~16 instructions against ~40 executed (120 emitted, with the large-argument path) per call.
PD_DEV void sincos_half_pi(float x, float &s, float &c) {
  const float x2 = x * x;
  const float p = -1.6666667e-1f + x2 * (8.3333338e-3f + x2 * (-1.9841270e-4f + x2 * (2.7557319e-6f + x2 * (-2.5052108e-8f + x2 * 1.6059044e-10f))));
  const float q = 4.1666668e-2f + x2 * (-1.3888889e-3f + x2 * (2.4801588e-5f + x2 * (-2.7557320e-7f + x2 * 2.0876756e-9f)));
  s = x + (x * x2) * p;
  c = (1.0f - 0.5f * x2) + (x2 * x2) * q;
}
PD_DEV qt q_axis_angle_sc(v3 axis, float ang, float2 &sc) {  // also returns (sin, cos) of ang / 2 for the adjoint; |ang| <= pi
  float s, c;
  sincos_half_pi(ang * 0.5f, s, c);
  sc = make_float2(s, c);
  return Q4(axis.x * s, axis.y * s, axis.z * s, c);
}
// 1/sqrt(1-x^2) for the acos/asin adjoints; 0 (contribution dropped, not inf) where sqrt(1-x^2) is not > 0,
// as Warp's builtin adjoints do.  POLICY, see DESIGN.md section 6.
PD_DEV float inv_sqrt_1mx2(float x) { float d = sqrt_hw(1.0f - x * x); return d > 0.0f ? rcp_hw(d) : 0.0f; }
PD_DEV float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
// Warp's acos/asin builtins clamp their argument to [-1, 1] (recall): a unit-quaternion w that rounds to
// 1.0000001f must not turn into NaN.  POLICY, see DESIGN.md section 6.
PD_DEV float acos_c(float x) { return acosf(clampf(x, -1.0f, 1.0f)); }
PD_DEV float asin_c(float x) { return asinf(clampf(x, -1.0f, 1.0f)); }
// Twist angle of a revolute joint (integrator_euler.py:394-400): the reference takes twist = normalize((axis (r.xyz . axis), r.w)) and
// q = 2 acos(twist.w) sign(axis . twist.xyz).  With y = |axis| |r.xyz . axis| and x = r.w that is q = 2 sign(r.xyz . axis) atan2(y, x)
// EXACTLY (acos(x / sqrt(x^2 + y^2)) = atan2(y, x) for y >= 0) -- and this is how it is evaluated here: acos near 1 turns one ulp of
// twist.w into 7e-4 rad of a small joint angle and its derivative -1 / sqrt(1 - w^2) cancels catastrophically, so a literal fp32
// evaluation (the fp32 C oracle does it) is 1e-3 .. 1 off the float64 value of the SAME function in most 100-step Laikago
// rollouts; the atan2 form is accurate to an ulp and has no singular point at angle 0.  da = r.xyz . axis, alen = |axis|.
// dq_dda, dq_dw: the partial derivatives, for the adjoint.  A NAMED DEVIATION in evaluation, not in the function.
// atan2(y, x) for y >= 0, result in [0, pi]: quotient of the smaller by the larger magnitude, one reduction about tan(pi/8) and
// the degree-9 odd polynomial of Cephes' atanf on |t| <= tan(pi/8) (relative error 2e-7; about 30 instructions, two v_rcp_f32,
// where libdevice's atan2f -- signed zeros, infinities, denormals -- is about 60).  atan2(0, 0) = 0; NaN in, NaN out.
PD_DEV float atan2_pos(float y, float x) {
  const float ax = fabsf(x), hi = fmaxf(ax, y), lo = fminf(ax, y);
  float t = hi > 0.0f ? lo * rcp_hw(hi) : 0.0f;                      // in [0, 1]
  const bool red = t > 0.41421356f;                                    // tan(pi/8)
  const float tr = (t - 1.0f) * rcp_hw(t + 1.0f);                   // atan(t) = pi/4 + atan((t - 1) / (t + 1))
  t = red ? tr : t;
  const float z = t * t;
  float r = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * t + t;
  r = red ? r + 0.78539816339f : r;
  r = y > ax ? 1.57079632679f - r : r;                                 // the quotient was |x| / y
  r = x < 0.0f ? 3.14159265359f - r : r;
  return (x != x || y != y) ? x + y : r;
}
// atan2 for either sign of y: +-atan2_pos(|y|, x), in [-pi, pi] (atan2(+-0, negative) = +pi: the sign of a zero is not defined in this
// library, Makefile).  ~34 instructions where libdevice's atan2f executes ~55; used by the compound joint's angle decomposition.
PD_DEV float atan2_any(float y, float x) {
  const float r = atan2_pos(fabsf(y), x);
  return y < 0.0f ? -r : r;
}
PD_DEV float twist_angle(float da, float w, float alen, float &dq_dda, float &dq_dw) {
#if PD_POLICY == 1  // the reference's text: twist = normalize((axis da, w)), q = 2 acos(twist.w) sign(axis . twist.xyz); adjoint through
  {                              // acos' (guarded: 0 at |twist.w| = 1) and the normalisation
    const float y0 = da * alen, n2 = w * w + y0 * y0, il = rcp_hw(sqrt_hw(n2)), tw = w * il, sg = da < 0.0f ? -1.0f : 1.0f;
    const float q0 = acos_c(tw) * 2.0f * sg;
    const float dq = -2.0f * sg * inv_sqrt_1mx2(tw);
    dq_dw = dq * (il - w * w * il * il * il);
    dq_dda = dq * (-w * y0 * alen * il * il * il);
    return q0;
  }
#endif
  const float y = fabsf(da) * alen, sgn = da < 0.0f ? -1.0f : 1.0f;
  const float d = w * w + y * y, id = d > 0.0f ? rcp_hw(d) : 0.0f;
  dq_dda = 2.0f * alen * w * id;
  dq_dw = -2.0f * sgn * y * id;
  return 2.0f * sgn * atan2_pos(y, w);
}
// The FIXED joint's angular error normalize(v) * 2 acos(w) (integrator_euler.py:385-390) as the same function of a unit quaternion in
// its scale-invariant form  v h,  h = 2 atan2(|v|, w) / |v|  (series of atan(x) / x below x = |v| / w = 1e-2): r = conj(q_p) q_c of
// fp32-normalised quaternions has |r| = 1 + O(1e-7), which the literal acos(r.w) turns into +-9e-4 rad of spurious angle at the joint's
// operating point -- NAMED DEVIATION in evaluation, like twist_angle (both C oracles evaluate it this way with
// ref_set_twist_eval(1)).  Partials: d(v h)/dv = h I + hs_over_s v v^T,  d(v h)/dw = v h_w.
PD_DEV float fixed_ang_h(float s2, float w, float &hs_over_s, float &h_w) {
#if PD_POLICY == 1  // the reference's text: normalize(v) * acos(w) * 2 (normalize(0) = 0), adjoint through acos' (guarded) and normalize
  {
    const float sl = sqrt_hw(s2), isl = sl > 0.0f ? rcp_hw(sl) : 0.0f, h0 = 2.0f * acos_c(w) * isl;
    hs_over_s = -h0 * isl * isl; h_w = -2.0f * inv_sqrt_1mx2(w) * isl;
    return h0;
  }
#endif
  const float den = s2 + w * w;
  float h = 0.0f, hss = 0.0f;
  if (w > 0.0f && s2 < 1e-4f * w * w) {
    const float iw = rcp_hw(w), x2 = s2 * iw * iw;
    const float u = 1.0f - x2 * (1.0f / 3.0f - x2 * (1.0f / 5.0f - x2 * (1.0f / 7.0f)));
    const float upx = -2.0f / 3.0f + x2 * (4.0f / 5.0f - x2 * (6.0f / 7.0f));
    h = 2.0f * u * iw; hss = 2.0f * upx * iw * iw * iw;
  } else if (s2 > 0.0f) {
    const float sl = sqrt_hw(s2), phi = atan2_pos(sl, w);
    h = 2.0f * phi * rcp_hw(sl); hss = 2.0f * (w * rcp_hw(den) - phi * rcp_hw(sl)) * rcp_hw(s2);
  }
  hs_over_s = hss; h_w = den > 0.0f ? -2.0f * rcp_hw(den) : 0.0f;
  return h;
}
PD_DEV float twist_angle(float da, float w, float alen) {
#if PD_POLICY == 1
  { const float y0 = da * alen; return acos_c(w * rcp_hw(sqrt_hw(w * w + y0 * y0))) * 2.0f * (da < 0.0f ? -1.0f : 1.0f); }
#endif
  const float y = fabsf(da) * alen;
  return 2.0f * (da < 0.0f ? -1.0f : 1.0f) * atan2_pos(y, w);
}
// Gradient post-processing of the reference's autograd boundaries, applied where a gradient is STORED (one compare + one select
// per value instead of a pass over the tensor afterwards): POST 1 = remove_nan with clip=False (dp_utils.py:43-57 of the
// reference: NaN -> 0, inf kept), what ForwardWarp.backward does to every gradient it returns (dp_model.py:1294-1384);
// POST 2 = ForwardKinematics.backward's (dp_model.py:1109-1110, 1122-1123): NaN -> 0, then values above 1 -> 1.
// NaN -> 0 in ONE instruction: v_med3_f32(x, 0, x) is the median -- x, also for +-inf -- unless an operand is NaN, in which case
// the hardware returns min3 with v_min_f32's "the operand that is not a NaN" rule, i.e. 0.
template <int POST>
PD_DEV float grad_post(float x) {
  if (POST >= 1) x = __builtin_amdgcn_fmed3f(x, 0.0f, x);
  if (POST == 2) x = x > 1.0f ? 1.0f : x;
  return x;
}
PD_DEV float clamp_pass(float x, float lo, float hi) { return (x < lo || x > hi) ? 0.0f : 1.0f; }
PD_DEV v3 clamp3(v3 a, float l) { return V3(clampf(a.x, -l, l), clampf(a.y, -l, l), clampf(a.z, -l, l)); }
PD_DEV v3 clamp3_pass(v3 a, v3 g, float l) {
  return V3(g.x * clamp_pass(a.x, -l, l), g.y * clamp_pass(a.y, -l, l), g.z * clamp_pass(a.z, -l, l));
}

// ---- adjoints: accumulate, like a tape replay ------------------------------------
PD_DEV void adj_cross(v3 a, v3 b, v3 &adj_a, v3 &adj_b, v3 g) { adj_a += cross(b, g); adj_b += cross(g, a); }
PD_DEV void adj_cross_a(v3 b, v3 &adj_a, v3 g) { adj_a += cross(b, g); }
PD_DEV void adj_cross_b(v3 a, v3 &adj_b, v3 g) { adj_b += cross(g, a); }
PD_DEV void adj_qmul(qt a, qt b, qt &adj_a, qt &adj_b, qt g) { adj_a += qmul(g, qconj(b)); adj_b += qmul(qconj(a), g); }
PD_DEV void adj_qmul_a(qt b, qt &adj_a, qt g) { adj_a += qmul(g, qconj(b)); }
PD_DEV void adj_qmul_b(qt a, qt &adj_b, qt g) { adj_b += qmul(qconj(a), g); }
PD_DEV void adj_qrot_q(qt q, v3 v, qt &adj_q, v3 g) {
  v3 u = qvec(q);
  float uv = dot(u, v), ug = dot(u, g);
  v3 au = cross(v, g) * (2.0f * q.w) + (g * uv + v * ug) * 2.0f;
  adj_q.x += au.x; adj_q.y += au.y; adj_q.z += au.z;
  adj_q.w += 4.0f * q.w * dot(v, g) + 2.0f * dot(cross(u, v), g);
}
PD_DEV void adj_qrot(qt q, v3 v, qt &adj_q, v3 &adj_v, v3 g) { adj_v += qrot_inv(q, g); adj_qrot_q(q, v, adj_q, g); }
PD_DEV void adj_qrot_inv_q(qt q, v3 v, qt &adj_q, v3 g) {
  v3 u = qvec(q);
  float uv = dot(u, v), ug = dot(u, g);
  v3 au = cross(v, g) * (-2.0f * q.w) + (g * uv + v * ug) * 2.0f;
  adj_q.x += au.x; adj_q.y += au.y; adj_q.z += au.z;
  adj_q.w += 4.0f * q.w * dot(v, g) - 2.0f * dot(cross(u, v), g);
}
PD_DEV void adj_qrot_inv(qt q, v3 v, qt &adj_q, v3 &adj_v, v3 g) { adj_v += qrot(q, g); adj_qrot_inv_q(q, v, adj_q, g); }
PD_DEV void adj_qnormalize(qt q, qt &adj_q, qt g) {
  float il = rcp_hw(sqrt_hw(qdot(q, q)));
  qt n = q * il;
  float ng = qdot(n, g);
  adj_q += (g + n * (-ng)) * il;
}
PD_DEV void adj_normalize(v3 a, v3 &adj_a, v3 g) {
  float l = length(a);
  if (l > 0.0f) {
    float il = rcp_hw(l);
    v3 n = a * il;
    adj_a += (g - n * dot(n, g)) * il;
  }
}
PD_DEV void adj_length(v3 a, v3 &adj_a, float g) { adj_a += normalize(a) * g; }
PD_DEV void adj_q_axis_angle(v3 axis, float ang, v3 &adj_axis, float &adj_ang, qt g) {
  float s, c;
  sincosf(ang * 0.5f, &s, &c);
  v3 gv = qvec(g);
  adj_axis += gv * s;
  adj_ang += 0.5f * (c * dot(axis, gv) - s * g.w);
}
// the same with (sin, cos) of ang / 2 handed over from the forward pass instead of recomputed
PD_DEV void adj_q_axis_angle_sc(v3 axis, float s, float c, v3 &adj_axis, float &adj_ang, qt g) {
  v3 gv = qvec(g);
  adj_axis += gv * s;
  adj_ang += 0.5f * (c * dot(axis, gv) - s * g.w);
}
PD_DEV void adj_q_axis_angle_ang_sc(v3 axis, float s, float c, float &adj_ang, qt g) {
  adj_ang += 0.5f * (c * dot(axis, qvec(g)) - s * g.w);
}
PD_DEV void adj_q_axis_angle_ang(v3 axis, float ang, float &adj_ang, qt g) {
  float s, c;
  sincosf(ang * 0.5f, &s, &c);
  adj_ang += 0.5f * (c * dot(axis, qvec(g)) - s * g.w);
}

// Rotation by a quaternion as a matrix: rotm(q) v == qrot(q, v) and rotm(q)^T v == qrot_inv(q, v) for ANY q -- it is the linear
// map of the formula above, (2w^2 - 1) I + 2w [u]x + 2 u u^T, not a normalised rotation.  Where one quaternion rotates
// several vectors in a step, 22 instructions for the matrix + 9 per vector replace 23 per vector, and in the adjoint
// 9 (outer product into adj_M) + 9 (transposed product) per vector + one rotm_adj replace ~70 per vector.
// Row 1 of rotm(q) -- the world-up row: the height of a body-frame point x is p_y + row1 . x -- with its roundings PINNED
// (no contraction beyond the explicit fma): the forward pass decides "this contact point touches" on a height computed from
// this row, and the adjoint, which recomputes it from the stored quaternion in another kernel, must land on the same side.
PD_DEV void rot_row1(qt q, float &a, float &b, float &c) {
#pragma clang fp contract(off)
  const float s = __builtin_fmaf(2.0f * q.w, q.w, -1.0f), tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  const float wz = tz * q.w, wx = tx * q.w;
  a = __builtin_fmaf(tx, q.y, wz);
  b = __builtin_fmaf(ty, q.y, s);
  c = __builtin_fmaf(ty, q.z, -wx);
}
// Height above ground of contact candidate P = (x, y, z, dist) of a body with cull vector cv = (p_y, row 1 of rotm(q)):
// integrator_euler.py:118-121 with n = +y.  Pinned like rot_row1: every kernel gets the same bits from the same (cv, P).
PD_DEV float contact_height(float4 cv, float4 P) {
#pragma clang fp contract(off)
  return __builtin_fmaf(cv.w, P.z, __builtin_fmaf(cv.z, P.y, __builtin_fmaf(cv.y, P.x, cv.x))) - P.w;
}
PD_DEV void rotm(qt q, float *M) {
  const float s = 2.0f * q.w * q.w - 1.0f, tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  const float xy = tx * q.y, xz = tx * q.z, yz = ty * q.z, wx = tx * q.w, wy = ty * q.w, wz = tz * q.w;
  M[0] = s + tx * q.x; M[1] = xy - wz; M[2] = xz + wy;
  rot_row1(q, M[3], M[4], M[5]);
  M[6] = xz - wy; M[7] = yz + wx; M[8] = s + tz * q.z;
}
// adj_q += d<A, rotm(q)>/dq for the accumulated matrix adjoint A
PD_DEV void rotm_adj(qt q, const float *A, qt &adj_q) {
  const float s13 = A[1] + A[3], s26 = A[2] + A[6], s57 = A[5] + A[7], d75 = A[7] - A[5], d26 = A[2] - A[6], d31 = A[3] - A[1];
  adj_q.x += 2.0f * (2.0f * q.x * A[0] + q.y * s13 + q.z * s26 + q.w * d75);
  adj_q.y += 2.0f * (2.0f * q.y * A[4] + q.x * s13 + q.z * s57 + q.w * d26);
  adj_q.z += 2.0f * (2.0f * q.z * A[8] + q.x * s26 + q.y * s57 + q.w * d31);
  adj_q.w += 2.0f * (2.0f * q.w * (A[0] + A[4] + A[8]) + q.x * d75 + q.y * d26 + q.z * d31);
}

PD_DEV v3 mat_vec(const float *M, v3 a) {
  return V3(M[0] * a.x + M[1] * a.y + M[2] * a.z, M[3] * a.x + M[4] * a.y + M[5] * a.z, M[6] * a.x + M[7] * a.y + M[8] * a.z);
}
PD_DEV v3 matT_vec(const float *M, v3 a) {
  return V3(M[0] * a.x + M[3] * a.y + M[6] * a.z, M[1] * a.x + M[4] * a.y + M[7] * a.z, M[2] * a.x + M[5] * a.y + M[8] * a.z);
}
PD_DEV void add_outer(float *M, v3 a, v3 b) {
  M[0] += a.x * b.x; M[1] += a.x * b.y; M[2] += a.x * b.z;
  M[3] += a.y * b.x; M[4] += a.y * b.y; M[5] += a.y * b.z;
  M[6] += a.z * b.x; M[7] += a.z * b.y; M[8] += a.z * b.z;
}
