// gfx950 kernels: whole-rollout forward, whole-rollout adjoint, batched FK forward / adjoint.
//
// One launch covers all T steps of all envs (the reference issues 4 launches + 3 aux ops per step
// from Python, /root/reference/diffphys/dp_model.py:1209-1234).  Per step the only HBM traffic is
// the SoA state spill for the adjoint, the step's controls, and frame outputs.  See DESIGN.md.
#include "pd_device.h"
#include "pd_args.h"


template <int SEGW>
struct Seg {
  static constexpr int EPW = 64 / SEGW;
  static constexpr unsigned long long MASK = SEGW == 64 ? ~0ull : ((1ull << SEGW) - 1ull);
};

// Segment-local ballot: bit i = predicate of lane i of my segment.
template <int SEGW>
PD_DEV unsigned long long seg_ballot(bool pred, int seg) {
  unsigned long long b = __ballot(pred);
  return (b >> (seg * SEGW)) & Seg<SEGW>::MASK;
}

// Ground-contact sweep for one segment (= one env).  Three-level cull, all conservative, then the
// exact test of the reference inside on_hit.  Tables live in LDS (copied once per workgroup):
//   L1  per body  : bounding sphere of all its candidate points vs y = 0          (lane = body)
//   L2  per tile  : tile = <= SEGW spatially compact points of ONE body; sphere test (lane = tile of a surviving body)
//   L3  per point : y-row test  c = p_y + Ry . x - dist                            (lane = point of a surviving tile)
// on_hit(point index, body, record pointer, point float4) runs for lanes whose L3 value is <= eps.
struct SweepTables {
  const float4 *pts, *tsphere;
  const int4 *tinfo;
  const int2 *btiles;
};

template <int SEGW, typename F>
PD_DEV void sweep_contacts(const PdDevModel &m, const SweepTables &T, float4 sphere, const float *rec, int *list, bool is_body,
                           int seg, int l, F &&on_hit) {
  if (m.nc == 0) return;
  bool surv = false;
  if (is_body && sphere.w >= 0.0f) {
    const float *r = rec + l * PD_REC;
    float ylow = r[1] + (r[16] * sphere.x + r[17] * sphere.y + r[18] * sphere.z) - sphere.w;
    surv = !(ylow > 1e-4f * (1.0f + sphere.w));
  }
  unsigned long long wave_any = __ballot(surv);
  if (wave_any == 0ull) return;
  unsigned long long mb = (wave_any >> (seg * SEGW)) & Seg<SEGW>::MASK;
  const unsigned long long lt = (1ull << l) - 1ull;
  int nlist = 0;
  while (__ballot(mb != 0ull) != 0ull) {  // one surviving body per segment per iteration
    int bb = 0, nt = 0, t_first = 0;
    if (mb != 0ull) {
      bb = __ffsll((long long)mb) - 1;
      mb &= mb - 1ull;
      int2 bt = T.btiles[bb];
      t_first = bt.x; nt = bt.y;
    }
    const float *r = rec + bb * PD_REC;
    const float py = r[1], ryx = r[16], ryy = r[17], ryz = r[18];
    for (int t0 = 0; __ballot(t0 < nt) != 0ull; t0 += SEGW) {
      int t = t0 + l;
      bool pass = t < nt;
      if (pass) {
        float4 sp = T.tsphere[t_first + t];
        float ylow = py + (ryx * sp.x + ryy * sp.y + ryz * sp.z) - sp.w;
        pass = !(ylow > 1e-4f * (1.0f + sp.w));
      }
      unsigned long long ps = seg_ballot<SEGW>(pass, seg);
      if (pass) list[nlist + __popcll(ps & lt)] = t_first + t;
      nlist += __popcll(ps);
    }
  }
  WAVE_SYNC();
  for (int k = 0; __ballot(k < nlist) != 0ull; ++k) {
    if (k < nlist) {
      int4 ti = T.tinfo[list[k]];
      if (l < ti.y) {
        float4 P = T.pts[ti.x + l];
        const float *r = rec + ti.z * PD_REC;
        float cq = r[1] + (r[16] * P.x + r[17] * P.y + r[18] * P.z) - P.w;
        if (cq <= 1e-4f) on_hit(ti.x + l, ti.z, r, P);
      }
    }
  }
}

// Copies the contact tables into LDS (once per workgroup) and returns the per-env scratch base.
PD_DEV float *lds_setup(const PdDevModel &m, unsigned char *smem, SweepTables &T, int env_slot) {
  const int nc4 = m.nc > 0 ? m.nc : 1, nt4 = m.ntiles > 0 ? m.ntiles : 1, nbp = (m.nb + 1) & ~1;
  float4 *pts = (float4 *)smem;
  float4 *tsp = pts + nc4;
  int4 *tin = (int4 *)(tsp + nt4);
  int2 *btl = (int2 *)(tin + nt4);
  for (int i = threadIdx.x; i < m.nc; i += PD_BLOCK) pts[i] = m.pts[i];
  for (int i = threadIdx.x; i < m.ntiles; i += PD_BLOCK) { tsp[i] = m.tile_sphere[i]; tin[i] = m.tile_info[i]; }
  for (int i = threadIdx.x; i < m.nb; i += PD_BLOCK) btl[i] = m.body_tiles[i];
  __syncthreads();
  T.pts = pts; T.tsphere = tsp; T.tinfo = tin; T.btiles = btl;
  return (float *)(btl + nbp) + (size_t)env_slot * m.env_lds_floats;
}

// =============================================================================================
template <int SEGW, int JT>
__global__ __launch_bounds__(PD_BLOCK) void k_rollout_fwd(PdDevModel m, RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int EPW = Seg<SEGW>::EPW;
  constexpr int ND = (JT & PD_JT_COMPOUND) ? 3 : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = lane / SEGW, l = lane % SEGW;
  const int env = (blockIdx.x * PD_WAVES + wave) * EPW + seg;
  const bool env_ok = env < a.bs;
  const bool is_body = env_ok && l < m.nb;
  const int b = l < m.nb ? l : m.nb - 1;
  const int nb = m.nb, N = a.bs * nb;

  SweepTables tabs;
  float *scratch = lds_setup(m, smem, tabs, wave * EPW + seg);
  float *rec = scratch, *facc = rec + nb * PD_REC, *pcon = facc + nb * PD_W6;
  int *list = (int *)(pcon + nb * PD_W6);

  const BodyConst c = load_body_const(m, b);
  const int ec = env_ok ? env : 0;       // clamped env for safe addressing
  const size_t idx = (size_t)ec * nb + b;  // flat body index (env-major)
  const int ndof = c.type == PD_JOINT_REVOLUTE ? 1 : (c.type == PD_JOINT_COMPOUND ? 3 : 0);

  float inv_m = a.inv_mass[idx], I[9], invI[9], ke[ND], kd[ND];
#pragma unroll
  for (int k = 0; k < 9; ++k) { I[k] = a.inertia[idx * 9 + k]; invI[k] = a.inv_inertia[idx * 9 + k]; }
#pragma unroll
  for (int k = 0; k < ND; ++k) {
    bool on = k < ndof;
    ke[k] = on ? a.target_ke[(size_t)ec * m.nqd + c.qdstart + k] : 0.f;
    kd[k] = on ? a.target_kd[(size_t)ec * m.nqd + c.qdstart + k] : 0.f;
  }
  if (is_body) {
#pragma unroll
    for (int k = 0; k < 6; ++k) facc[b * PD_W6 + k] = 0.f;
  }

  // ---- eval_fk (dp_model.py:1204): level-synchronous walk of the chain through LDS
  BodyState s;
  s.p = V3(0, 0, 0); s.r = Q4(0, 0, 0, 1); s.w = V3(0, 0, 0); s.v = V3(0, 0, 0);
  for (int d = 0; d <= m.max_depth; ++d) {
    if (is_body && c.depth == d) {
      s = fk_joint<JT>(c, a.q_init + (size_t)ec * m.nq + c.qstart, a.qd_init + (size_t)ec * m.nqd + c.qdstart, rec);
      stage_record(rec, b, s, c.com);
    }
    WAVE_SYNC();
  }

  float *traj_q = a.ws, *traj_qd = a.ws + (size_t)a.nsteps * 7 * N, *traj_f = a.ws + (size_t)a.nsteps * 13 * N;
  // Controls are software-prefetched one step ahead: with one wavefront per SIMD there is no other
  // wave to hide the HBM latency of a load issued at its point of use.
  float n_tgt[ND], n_act[ND], n_rf[6];
  auto load_controls = [&](int step) {
    const int sc = step < a.nsteps ? step : a.nsteps - 1;
    const size_t o = (size_t)sc * a.bs * m.nqd + (size_t)ec * m.nqd + c.qdstart;
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      bool on = k < ndof;
      n_tgt[k] = on ? a.refs[o + k] : 0.f;
      n_act[k] = on ? a.torques[o + k] : 0.f;
    }
    const float *rf = a.res_f + ((size_t)sc * N + idx) * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) n_rf[k] = rf[k];
  };
  if (a.nsteps > 0) load_controls(0);
  for (int step = 0; step < a.nsteps; ++step) {
    float tgt[ND], act[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) { tgt[k] = n_tgt[k]; act[k] = n_act[k]; }
    v3 ft = V3(n_rf[0], n_rf[1], n_rf[2]), ff = V3(n_rf[3], n_rf[4], n_rf[5]);  // clear_forces + wp_add
    load_controls(step + 1);
    // spill the state for the adjoint (SoA planes: lanes of a wave write consecutive floats)
    if (is_body) {
      float *tq = traj_q + (size_t)step * 7 * N + idx, *td = traj_qd + (size_t)step * 6 * N + idx;
      tq[0] = s.p.x; tq[(size_t)N] = s.p.y; tq[(size_t)2 * N] = s.p.z;
      tq[(size_t)3 * N] = s.r.x; tq[(size_t)4 * N] = s.r.y; tq[(size_t)5 * N] = s.r.z; tq[(size_t)6 * N] = s.r.w;
      td[0] = s.w.x; td[(size_t)N] = s.w.y; td[(size_t)2 * N] = s.w.z;
      td[(size_t)3 * N] = s.v.x; td[(size_t)4 * N] = s.v.y; td[(size_t)5 * N] = s.v.z;
    }
    // ---- eval_body_contacts
    sweep_contacts<SEGW>(m, tabs, c.sphere, rec, list, is_body, seg, l, [&](int pt, int pb, const float *r, float4 P) {
      ContactOut o;
      if (contact_point_fwd(r, P, m.pt_mat[pt], o)) {
        float *f = facc + pb * PD_W6;
        atomicAdd(f + 0, -o.t.x); atomicAdd(f + 1, -o.t.y); atomicAdd(f + 2, -o.t.z);
        atomicAdd(f + 3, -o.f.x); atomicAdd(f + 4, -o.f.y); atomicAdd(f + 5, -o.f.z);
      }
    });
    WAVE_SYNC();
    if (is_body) {
      float *f = facc + b * PD_W6;
      ft += V3(f[0], f[1], f[2]); ff += V3(f[3], f[4], f[5]);
#pragma unroll
      for (int k = 0; k < 6; ++k) f[k] = 0.f;
    }
    const int fr = a.frame_of_step[step];
    v3 grf_t = ft, grf_f = ff;
    // ---- eval_body_joints
    v3 wp_t = V3(0, 0, 0), wp_f = wp_t, wc_t = wp_t, wc_f = wp_t;
    if (is_body && c.type != PD_JOINT_FREE) joint_fwd<JT>(m, c, s, rec, tgt, act, ke, kd, wp_t, wp_f, wc_t, wc_f);
    if (is_body) {
      float *pc = pcon + b * PD_W6;
      pc[0] = wp_t.x; pc[1] = wp_t.y; pc[2] = wp_t.z; pc[3] = wp_f.x; pc[4] = wp_f.y; pc[5] = wp_f.z;
    }
    WAVE_SYNC();
    ft -= wc_t; ff -= wc_f;
    for (int k = 0; k < m.max_children; ++k) {
      int cid = (int)((c.children >> (8 * k)) & 0xffull);
      if (is_body && cid != 0xff) {
        const float *pc = pcon + cid * PD_W6;
        ft += V3(pc[0], pc[1], pc[2]); ff += V3(pc[3], pc[4], pc[5]);
      }
    }
    if (is_body) {
      float *tf = traj_f + (size_t)step * 6 * N + idx;
      tf[0] = ft.x; tf[(size_t)N] = ft.y; tf[(size_t)2 * N] = ft.z;
      tf[(size_t)3 * N] = ff.x; tf[(size_t)4 * N] = ff.y; tf[(size_t)5 * N] = ff.z;
      if (fr >= 0) {  // frame gather (dp_model.py:1231-1248)
        float *o = a.wp_pos + ((size_t)fr * N + idx) * 7;
        o[0] = s.p.x; o[1] = s.p.y; o[2] = s.p.z; o[3] = s.r.x; o[4] = s.r.y; o[5] = s.r.z; o[6] = s.r.w;
        o = a.wp_vel + ((size_t)fr * N + idx) * 6;
        o[0] = s.w.x; o[1] = s.w.y; o[2] = s.w.z; o[3] = s.v.x; o[4] = s.v.y; o[5] = s.v.z;
        if (a.grf) {
          o = a.grf + ((size_t)fr * N + idx) * 6;
          o[0] = grf_t.x; o[1] = grf_t.y; o[2] = grf_t.z; o[3] = grf_f.x; o[4] = grf_f.y; o[5] = grf_f.z;
        }
        if (a.jaf) {
          o = a.jaf + ((size_t)fr * N + idx) * 6;
          o[0] = ft.x - grf_t.x; o[1] = ft.y - grf_t.y; o[2] = ft.z - grf_t.z;
          o[3] = ff.x - grf_f.x; o[4] = ff.y - grf_f.y; o[5] = ff.z - grf_f.z;
        }
      }
    }
    // ---- integrate_bodies
    s = integrate_fwd(m, c, s, ft, ff, inv_m, I, invI, a.dt);
    WAVE_SYNC();
    if (is_body) stage_record(rec, b, s, c.com);
    WAVE_SYNC();
  }
}

// =============================================================================================
template <int SEGW, int JT>
__global__ __launch_bounds__(PD_BLOCK) void k_rollout_bwd(PdDevModel m, RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int EPW = Seg<SEGW>::EPW;
  constexpr int ND = (JT & PD_JT_COMPOUND) ? 3 : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = lane / SEGW, l = lane % SEGW;
  const int env = (blockIdx.x * PD_WAVES + wave) * EPW + seg;
  const bool env_ok = env < a.bs;
  const bool is_body = env_ok && l < m.nb;
  const int b = l < m.nb ? l : m.nb - 1;
  const int nb = m.nb, N = a.bs * nb;

  SweepTables tabs;
  float *scratch = lds_setup(m, smem, tabs, wave * EPW + seg);
  float *rec = scratch, *adjf = rec + nb * PD_REC, *cslot = adjf + nb * PD_W6, *cacc = cslot + nb * PD_ADJ;
  int *list = (int *)(cacc + nb * PD_ADJ);

  const BodyConst c = load_body_const(m, b);
  const int ec = env_ok ? env : 0;
  const size_t idx = (size_t)ec * nb + b;
  const int ndof = c.type == PD_JOINT_REVOLUTE ? 1 : (c.type == PD_JOINT_COMPOUND ? 3 : 0);

  float inv_m = a.inv_mass[idx], I[9], invI[9], ke[ND], kd[ND];
#pragma unroll
  for (int k = 0; k < 9; ++k) { I[k] = a.inertia[idx * 9 + k]; invI[k] = a.inv_inertia[idx * 9 + k]; }
#pragma unroll
  for (int k = 0; k < ND; ++k) {
    bool on = k < ndof;
    ke[k] = on ? a.target_ke[(size_t)ec * m.nqd + c.qdstart + k] : 0.f;
    kd[k] = on ? a.target_kd[(size_t)ec * m.nqd + c.qdstart + k] : 0.f;
  }
  float g_inv_m = 0.f, g_I[9], g_invI[9], g_ke[ND], g_kd[ND];
#pragma unroll
  for (int k = 0; k < 9; ++k) { g_I[k] = 0.f; g_invI[k] = 0.f; }
#pragma unroll
  for (int k = 0; k < ND; ++k) { g_ke[k] = 0.f; g_kd[k] = 0.f; }
  if (is_body) {
#pragma unroll
    for (int k = 0; k < PD_ADJ; ++k) cacc[b * PD_ADJ + k] = 0.f;
  }

  const float *traj_q = a.ws, *traj_qd = a.ws + (size_t)a.nsteps * 7 * N, *traj_f = a.ws + (size_t)a.nsteps * 13 * N;
  BodyAdj gn = adj_zero();  // adjoint of state step+1
  BodyState s;
  s.p = V3(0, 0, 0); s.r = Q4(0, 0, 0, 1); s.w = V3(0, 0, 0); s.v = V3(0, 0, 0);

  // The stored state, wrench and controls of the NEXT iteration (step - 1) are software-prefetched.
  float n_s[19], n_tgt[ND], n_act[ND];
  auto load_step = [&](int step) {
    const int sc = step >= 0 ? step : 0;
    const float *tq = traj_q + (size_t)sc * 7 * N + idx, *td = traj_qd + (size_t)sc * 6 * N + idx;
    const float *tf = traj_f + (size_t)sc * 6 * N + idx;
#pragma unroll
    for (int k = 0; k < 7; ++k) n_s[k] = tq[(size_t)k * N];
#pragma unroll
    for (int k = 0; k < 6; ++k) { n_s[7 + k] = td[(size_t)k * N]; n_s[13 + k] = tf[(size_t)k * N]; }
    const size_t o = (size_t)sc * a.bs * m.nqd + (size_t)ec * m.nqd + c.qdstart;
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      bool on = k < ndof;
      n_tgt[k] = on ? a.refs[o + k] : 0.f;
      n_act[k] = on ? a.torques[o + k] : 0.f;
    }
  };
  if (a.nsteps > 0) load_step(a.nsteps - 1);
  for (int step = a.nsteps - 1; step >= 0; --step) {
    {  // seeds of state step+1 (dp_model.py:1264-1271)
      int fr = a.frame_of_step[step + 1];
      if (fr >= 0) {
        const float *gp = a.adj_pos + ((size_t)fr * N + idx) * 7, *gv = a.adj_vel + ((size_t)fr * N + idx) * 6;
        gn.p += V3(gp[0], gp[1], gp[2]); gn.r += Q4(gp[3], gp[4], gp[5], gp[6]);
        gn.w += V3(gv[0], gv[1], gv[2]); gn.v += V3(gv[3], gv[4], gv[5]);
      }
    }
    s.p = V3(n_s[0], n_s[1], n_s[2]); s.r = Q4(n_s[3], n_s[4], n_s[5], n_s[6]);
    s.w = V3(n_s[7], n_s[8], n_s[9]); s.v = V3(n_s[10], n_s[11], n_s[12]);
    v3 t0 = V3(n_s[13], n_s[14], n_s[15]), f0 = V3(n_s[16], n_s[17], n_s[18]);
    float tgt[ND], act[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) { tgt[k] = n_tgt[k]; act[k] = n_act[k]; }
    const size_t oc = (size_t)step * a.bs * m.nqd + (size_t)ec * m.nqd + c.qdstart;
    load_step(step - 1);
    if (is_body) stage_record(rec, b, s, c.com);
    // ---- adjoint of integrate_bodies
    BodyAdj ga = adj_zero();
    v3 adj_t0, adj_f0;
    integrate_adj(m, c, s, t0, f0, inv_m, I, invI, a.dt, gn, ga, adj_t0, adj_f0, g_inv_m, g_I, g_invI);
    if (is_body) {
      float *o = a.g_res_f + ((size_t)step * N + idx) * 6;  // adjoint of wp_add
      o[0] = adj_t0.x; o[1] = adj_t0.y; o[2] = adj_t0.z; o[3] = adj_f0.x; o[4] = adj_f0.y; o[5] = adj_f0.z;
      float *f = adjf + b * PD_W6;
      f[0] = adj_t0.x; f[1] = adj_t0.y; f[2] = adj_t0.z; f[3] = adj_f0.x; f[4] = adj_f0.y; f[5] = adj_f0.z;
    }
    WAVE_SYNC();
    // ---- adjoint of eval_body_joints
    BodyAdj par = adj_zero();
    float a_tgt[ND], a_act[ND], a_ke[ND], a_kd[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) { a_tgt[k] = 0.f; a_act[k] = 0.f; a_ke[k] = 0.f; a_kd[k] = 0.f; }
    if (is_body && c.type != PD_JOINT_FREE) {
      v3 gp_t = V3(0, 0, 0), gp_f = gp_t;
      if (c.parent >= 0) { gp_t = ld3(adjf + c.parent * PD_W6); gp_f = ld3(adjf + c.parent * PD_W6 + 3); }
      joint_adj<JT>(m, c, s, rec, tgt, act, ke, kd, adj_t0, adj_f0, gp_t, gp_f, ga, par, a_tgt, a_act, a_ke, a_kd);
    }
    if (is_body) {
      adj_store(cslot + b * PD_ADJ, par);
#pragma unroll
      for (int k = 0; k < ND; ++k) {
        if (k < ndof) { a.g_refs[oc + k] = a_tgt[k]; a.g_torques[oc + k] = a_act[k]; }
        g_ke[k] += a_ke[k]; g_kd[k] += a_kd[k];
      }
      if (c.type == PD_JOINT_FREE) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { a.g_refs[oc + k] = 0.f; a.g_torques[oc + k] = 0.f; }
      }
    }
    // ---- adjoint of eval_body_contacts
    sweep_contacts<SEGW>(m, tabs, c.sphere, rec, list, is_body, seg, l, [&](int pt, int pb, const float *r, float4 P) {
      BodyAdj o;
      if (contact_point_adj(r, P, m.pt_mat[pt], ld3(m.com + pb * 3), ld3(adjf + pb * PD_W6), ld3(adjf + pb * PD_W6 + 3), o)) {
        float *d = cacc + pb * PD_ADJ;
        atomicAdd(d + 0, o.p.x); atomicAdd(d + 1, o.p.y); atomicAdd(d + 2, o.p.z);
        atomicAdd(d + 3, o.r.x); atomicAdd(d + 4, o.r.y); atomicAdd(d + 5, o.r.z); atomicAdd(d + 6, o.r.w);
        atomicAdd(d + 7, o.w.x); atomicAdd(d + 8, o.w.y); atomicAdd(d + 9, o.w.z);
        atomicAdd(d + 10, o.v.x); atomicAdd(d + 11, o.v.y); atomicAdd(d + 12, o.v.z);
      }
    });
    WAVE_SYNC();
    for (int k = 0; k < m.max_children; ++k) {
      int cid = (int)((c.children >> (8 * k)) & 0xffull);
      if (is_body && cid != 0xff) adj_add_from(ga, cslot + cid * PD_ADJ);
    }
    if (is_body) {
      float *d = cacc + b * PD_ADJ;
      adj_add_from(ga, d);
#pragma unroll
      for (int k = 0; k < PD_ADJ; ++k) d[k] = 0.f;
    }
    gn = ga;
    WAVE_SYNC();
  }
  {  // seeds of state 0
    int fr = a.frame_of_step[0];
    if (fr >= 0) {
      const float *gp = a.adj_pos + ((size_t)fr * N + idx) * 7, *gv = a.adj_vel + ((size_t)fr * N + idx) * 6;
      gn.p += V3(gp[0], gp[1], gp[2]); gn.r += Q4(gp[3], gp[4], gp[5], gp[6]);
      gn.w += V3(gv[0], gv[1], gv[2]); gn.v += V3(gv[3], gv[4], gv[5]);
    }
  }
  // ---- adjoint of eval_fk: rec holds state 0 (staged in the last loop iteration)
  if (a.nsteps == 0) {
    for (int d = 0; d <= m.max_depth; ++d) {  // nothing staged yet: rebuild state 0
      if (is_body && c.depth == d) {
        s = fk_joint<JT>(c, a.q_init + (size_t)ec * m.nq + c.qstart, a.qd_init + (size_t)ec * m.nqd + c.qdstart, rec);
        stage_record(rec, b, s, c.com);
      }
      WAVE_SYNC();
    }
  }
  for (int d = m.max_depth; d >= 0; --d) {
    if (is_body && c.depth == d) {
      for (int k = 0; k < m.max_children; ++k) {
        int cid = (int)((c.children >> (8 * k)) & 0xffull);
        if (cid != 0xff) adj_add_from(gn, cslot + cid * PD_ADJ);
      }
      BodyAdj par = fk_joint_adj<JT>(c, a.q_init + (size_t)ec * m.nq + c.qstart, a.qd_init + (size_t)ec * m.nqd + c.qdstart, rec,
                                     gn, a.g_q_init + (size_t)ec * m.nq + c.qstart, a.g_qd_init + (size_t)ec * m.nqd + c.qdstart);
      adj_store(cslot + b * PD_ADJ, par);
    }
    WAVE_SYNC();
  }
  if (is_body) {
    a.g_inv_mass[idx] = g_inv_m;
#pragma unroll
    for (int k = 0; k < 9; ++k) { a.g_inertia[idx * 9 + k] = g_I[k]; a.g_inv_inertia[idx * 9 + k] = g_invI[k]; }
    const size_t og = (size_t)ec * m.nqd + c.qdstart;
#pragma unroll
    for (int k = 0; k < ND; ++k)
      if (k < ndof) { a.g_ke[og + k] = g_ke[k]; a.g_kd[og + k] = g_kd[k]; }
    if (c.type == PD_JOINT_FREE) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { a.g_ke[og + k] = 0.f; a.g_kd[og + k] = 0.f; }
    }
  }
}

// =============================================================================================
// Batched FK (ForwardKinematics, dp_model.py:1022-1130): n articulations, one per segment.
template <int SEGW, int JT, bool BWD>
__global__ __launch_bounds__(PD_BLOCK) void k_fk(PdDevModel m, FkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int EPW = Seg<SEGW>::EPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = lane / SEGW, l = lane % SEGW;
  const int env = (blockIdx.x * PD_WAVES + wave) * EPW + seg;
  const bool is_body = env < a.n && l < m.nb;
  const int b = l < m.nb ? l : m.nb - 1, nb = m.nb;
  const int ec = env < a.n ? env : 0;
  float *rec = (float *)smem + (size_t)(wave * EPW + seg) * (nb * (PD_REC + PD_ADJ));
  float *cslot = rec + nb * PD_REC;
  const BodyConst c = load_body_const(m, b);
  const float *jq = a.joint_q + (size_t)ec * m.nq + c.qstart, *jqd = a.joint_qd + (size_t)ec * m.nqd + c.qdstart;
  BodyState s;
  s.p = V3(0, 0, 0); s.r = Q4(0, 0, 0, 1); s.w = V3(0, 0, 0); s.v = V3(0, 0, 0);
  for (int d = 0; d <= m.max_depth; ++d) {
    if (is_body && c.depth == d) {
      s = fk_joint<JT>(c, jq, jqd, rec);
      stage_record(rec, b, s, c.com);
    }
    WAVE_SYNC();
  }
  const size_t idx = (size_t)ec * nb + b;
  if (!BWD) {
    if (is_body) {
      float *o = a.body_q + idx * 7;
      o[0] = s.p.x; o[1] = s.p.y; o[2] = s.p.z; o[3] = s.r.x; o[4] = s.r.y; o[5] = s.r.z; o[6] = s.r.w;
      o = a.body_qd + idx * 6;
      o[0] = s.w.x; o[1] = s.w.y; o[2] = s.w.z; o[3] = s.v.x; o[4] = s.v.y; o[5] = s.v.z;
    }
    return;
  }
  BodyAdj g = adj_zero();
  if (is_body) {
    const float *gp = a.adj_body_q + idx * 7, *gv = a.adj_body_qd + idx * 6;
    g.p = V3(gp[0], gp[1], gp[2]); g.r = Q4(gp[3], gp[4], gp[5], gp[6]);
    g.w = V3(gv[0], gv[1], gv[2]); g.v = V3(gv[3], gv[4], gv[5]);
  }
  for (int d = m.max_depth; d >= 0; --d) {
    if (is_body && c.depth == d) {
      for (int k = 0; k < m.max_children; ++k) {
        int cid = (int)((c.children >> (8 * k)) & 0xffull);
        if (cid != 0xff) adj_add_from(g, cslot + cid * PD_ADJ);
      }
      BodyAdj par = fk_joint_adj<JT>(c, jq, jqd, rec, g, a.g_joint_q + (size_t)ec * m.nq + c.qstart,
                                     a.g_joint_qd + (size_t)ec * m.nqd + c.qdstart);
      adj_store(cslot + b * PD_ADJ, par);
    }
    WAVE_SYNC();
  }
}

// =============================================================================================
// Launchers: this file is compiled once per segment width (-DPD_SEGW=16|32|64).
#ifndef PD_SEGW
#error "compile with -DPD_SEGW=16, 32 or 64"
#endif
#define PD_CAT2(a, b) a##b
#define PD_CAT(a, b) PD_CAT2(a, b)

template <int JT>
static hipError_t launch_jt(int kind, const PdDevModel &m, const void *args, int nblocks, size_t lds, hipStream_t st) {
  switch (kind) {
    case PD_K_ROLLOUT_FWD:
      hipLaunchKernelGGL((k_rollout_fwd<PD_SEGW, JT>), dim3(nblocks), dim3(PD_BLOCK), lds, st, m, *(const RolloutArgs *)args);
      break;
    case PD_K_ROLLOUT_BWD:
      hipLaunchKernelGGL((k_rollout_bwd<PD_SEGW, JT>), dim3(nblocks), dim3(PD_BLOCK), lds, st, m, *(const RolloutArgs *)args);
      break;
    case PD_K_FK_FWD:
      hipLaunchKernelGGL((k_fk<PD_SEGW, JT, false>), dim3(nblocks), dim3(PD_BLOCK), lds, st, m, *(const FkArgs *)args);
      break;
    case PD_K_FK_BWD:
      hipLaunchKernelGGL((k_fk<PD_SEGW, JT, true>), dim3(nblocks), dim3(PD_BLOCK), lds, st, m, *(const FkArgs *)args);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int JT>
static hipError_t set_lds_jt(int bytes) {
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void *)k_rollout_fwd<PD_SEGW, JT>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
  if ((e = hipFuncSetAttribute((const void *)k_rollout_bwd<PD_SEGW, JT>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
  if ((e = hipFuncSetAttribute((const void *)k_fk<PD_SEGW, JT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
  return hipFuncSetAttribute((const void *)k_fk<PD_SEGW, JT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// jt: PD_JT_REVOLUTE only, PD_JT_COMPOUND only, anything else -> generic (all joint types)
hipError_t PD_CAT(pd_launch_seg, PD_SEGW)(int kind, int jt, const PdDevModel &m, const void *args, int nblocks, size_t lds,
                                          hipStream_t st) {
  if (jt == PD_JT_REVOLUTE) return launch_jt<PD_JT_REVOLUTE>(kind, m, args, nblocks, lds, st);
  if (jt == PD_JT_COMPOUND) return launch_jt<PD_JT_COMPOUND>(kind, m, args, nblocks, lds, st);
  return launch_jt<PD_JT_REVOLUTE | PD_JT_COMPOUND | PD_JT_FIXED>(kind, m, args, nblocks, lds, st);
}
hipError_t PD_CAT(pd_set_lds_seg, PD_SEGW)(int jt, int bytes) {
  if (jt == PD_JT_REVOLUTE) return set_lds_jt<PD_JT_REVOLUTE>(bytes);
  if (jt == PD_JT_COMPOUND) return set_lds_jt<PD_JT_COMPOUND>(bytes);
  return set_lds_jt<PD_JT_REVOLUTE | PD_JT_COMPOUND | PD_JT_FIXED>(bytes);
}
