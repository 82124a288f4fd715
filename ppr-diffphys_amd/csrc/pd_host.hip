// Host side of libpprdiffphys_hip.so: model compiler back end (contact ordering, cull tables,
// chain topology), device upload, launch wrappers and the C ABI of include/ppr_diffphys.h.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ppr_diffphys.h"
#include "pd_args.h"

static thread_local std::string g_err;
static int fail(const std::string &msg) { g_err = msg; return 1; }
static int hip_fail(hipError_t e, const char *what) { return fail(std::string(what) + ": " + hipGetErrorString(e)); }

struct pd_model {
  // host copy of the template
  int nb = 0, nq = 0, nqd = 0, nc = 0, nmat = 0;
  std::vector<int> jtype, jparent, qstart, qdstart, cbody, cmat;
  std::vector<float> X_p, X_c, axis, com, lim_lo, lim_hi, lim_ke, lim_kd, cpoint, cdist, materials;
  float gravity[3] = {0, 0, 0}, attach_ke = 0, attach_kd = 0;
  // derived
  int segw = 0, jt = 0;
  std::vector<int> contact_order;  // device contact-table entry -> template candidate (pd_model_contact_order)
  // second device copy in the 64-lane mapping for the quad-lane (four lanes per body) small-batch kernels: revolute-only plain
  // models with at most 16 bodies; null otherwise
  struct Quad { void *blob; PdDevModel dev; size_t lds_tables; int jt; };
  Quad *quad = nullptr;
  int policy = 0;  // pd_model_set_numeric_policy: PD_NUM_STABLE / PD_NUM_LITERAL (which objects' rollout kernels a launch takes)
  int family = 0;  // pd_model_set_kernel_family: 0 automatic (by batch size), 1 lane per body always, 2 quad-lane wherever eligible
  void *blob = nullptr;
  PdDevModel dev{};
  size_t lds_rollout = 0, lds_rollout_bwd = 0, lds_fk = 0;  // at PD_BWAVES env groups per workgroup (the maximum)
  size_t lds_max = 0;  // what the kernels' dynamic-LDS attribute is at least set to
  size_t lds_tables = 0;                                      // contact tables, for the kernels that copy them into LDS
  // per-env joint_X_p bound by the caller (pd_model_bind_joint_X_p); null = the template's
  const float *xp_env = nullptr;
  int xp_envs = 0;
  // step -> frame tables on the device, one per (nsteps, frame2step) seen so far
  struct FosEntry { int nsteps; std::vector<int> f2s; int *dev; };
  std::vector<FosEntry> fos;
  // timing (per model) and the geometry of the last launch of each kind
  bool timing = false;
  hipEvent_t ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  bool ev_valid[2] = {false, false};
  int last_launch[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
};

// kd-order: recursively split along the longest axis so that runs of `leaf` consecutive points are compact
static void kd_order(std::vector<int> &ids, int lo, int hi, const float *pts, int leaf) {
  int n = hi - lo;
  if (n <= leaf) return;
  float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
  for (int i = lo; i < hi; ++i)
    for (int k = 0; k < 3; ++k) {
      float v = pts[ids[i] * 3 + k];
      mn[k] = std::min(mn[k], v); mx[k] = std::max(mx[k], v);
    }
  int ax = 0;
  for (int k = 1; k < 3; ++k) if (mx[k] - mn[k] > mx[ax] - mn[ax]) ax = k;
  std::stable_sort(ids.begin() + lo, ids.begin() + hi, [&](int a, int b) { return pts[a * 3 + ax] < pts[b * 3 + ax]; });
  int half = ((n / 2 + leaf - 1) / leaf) * leaf;
  if (half >= n) half = n - leaf > 0 ? ((n - 1) / leaf) * leaf : n;
  if (half <= 0 || half >= n) return;
  kd_order(ids, lo, lo + half, pts, leaf);
  kd_order(ids, lo + half, hi, pts, leaf);
}

static float4 bound_sphere(const std::vector<int> &ids, const float *pts, const float *dist) {
  if (ids.empty()) return make_float4(0, 0, 0, -1.0f);
  float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f}, dmax = -1e30f;
  for (int i : ids) {
    for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], pts[i * 3 + k]); mx[k] = std::max(mx[k], pts[i * 3 + k]); }
    dmax = std::max(dmax, dist[i]);
  }
  float c[3] = {0.5f * (mn[0] + mx[0]), 0.5f * (mn[1] + mx[1]), 0.5f * (mn[2] + mx[2])};
  float r = 0.f;
  for (int i : ids) {
    float d2 = 0.f;
    for (int k = 0; k < 3; ++k) { float d = pts[i * 3 + k] - c[k]; d2 += d * d; }
    r = std::max(r, std::sqrt(d2));
  }
  // lower bound of c = p_y + (R x)_y - dist over the set is p_y + (R c)_y - (r + max dist); |dist| margin keeps w >= 0
  return make_float4(c[0], c[1], c[2], r * 1.0001f + std::max(dmax, 0.0f) + 1e-6f);
}

// body-frame AABB of a point set: lo = (min xyz, max dist), hi = (max xyz, safety margin of the cull test)
static void bound_box(const std::vector<int> &ids, const float *pts, const float *dist, float4 &lo, float4 &hi) {
  float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f}, dmax = 0.f, ext = 0.f;
  for (int i : ids) {
    for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], pts[i * 3 + k]); mx[k] = std::max(mx[k], pts[i * 3 + k]); }
    dmax = std::max(dmax, dist[i]);
  }
  for (int k = 0; k < 3; ++k) ext = std::max(ext, std::max(std::fabs(mn[k]), std::fabs(mx[k])));
  lo = make_float4(mn[0], mn[1], mn[2], dmax);
  hi = make_float4(mx[0], mx[1], mx[2], 1e-4f * (1.0f + ext + dmax));
}

static void free_device(pd_model *m) {
  if (m->blob) { (void)hipFree(m->blob); m->blob = nullptr; }
}
static void free_quad(pd_model *m) {
  if (m->quad) { if (m->quad->blob) (void)hipFree(m->quad->blob); delete m->quad; m->quad = nullptr; }
}

template <typename T>
static size_t put(std::vector<unsigned char> &buf, const std::vector<T> &v) {
  size_t off = (buf.size() + 255) & ~(size_t)255;
  buf.resize(off + std::max<size_t>(v.size() * sizeof(T), 16));
  if (!v.empty()) memcpy(buf.data() + off, v.data(), v.size() * sizeof(T));
  return off;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel ON A DEVICE, not of a model: keep the running
// maximum per (device, segment width, joint mix) so that a second model never lowers what an earlier, larger one needs, and
// a process that builds models on several GPUs raises the attribute on each of them.
#define PD_MAX_DEVICES 64
static int g_lds_attr[PD_MAX_DEVICES][3][3];
static int jt_slot(int jt) { return jt == PD_JT_REVOLUTE ? 0 : (jt == PD_JT_COMPOUND ? 1 : 2); }

// Builds the device copy for segment width `segw` into temporaries and commits blob / dev / lds_* / segw / jt only when
// every check has passed: a failed call leaves the model exactly as it was.
static int build_device(pd_model *m, int segw) {
  const int nb = m->nb;
  if (segw == 0) segw = nb <= 16 ? 16 : (nb <= 32 ? 32 : 64);
  if (segw != 16 && segw != 32 && segw != 64) return fail("segment width must be 16, 32 or 64");
  if (nb > segw) return fail("segment width smaller than the number of bodies");
  // ---- chain topology
  std::vector<int> depth(nb, 0);
  std::vector<unsigned long long> children(nb, ~0ull);
  std::vector<int> nchild(nb, 0);
  int max_depth = 0, max_children = 0, jt = 0;
  for (int i = 0; i < nb; ++i) {
    int p = m->jparent[i];
    if (p >= i) return fail("parents must precede children");
    if (p >= 0) {
      depth[i] = depth[p] + 1;
      if (nchild[p] >= 8) return fail("more than 8 children per body is not supported");
      children[p] &= ~(0xffull << (8 * nchild[p]));
      children[p] |= (unsigned long long)i << (8 * nchild[p]);
      nchild[p]++;
      max_children = std::max(max_children, nchild[p]);
    }
    max_depth = std::max(max_depth, depth[i]);
    switch (m->jtype[i]) {
      case PD_JOINT_REVOLUTE: jt |= PD_JT_REVOLUTE; break;
      case PD_JOINT_COMPOUND: jt |= PD_JT_COMPOUND; break;
      case PD_JOINT_FIXED: jt |= PD_JT_FIXED; break;
      case PD_JOINT_FREE: break;
      default: return fail("unsupported joint type " + std::to_string(m->jtype[i]) + " (revolute, fixed, free, compound only)");
    }
  }
  // the two specialised instantiations (revolute-only, compound-only robots) also assume that every joint which is not FREE
  // hangs on a body and that every child joint frame has the identity rotation (pd_parented: a PLAIN model, see
  // pd_device.h joint_ctx); anything else takes the generic one
  bool world_joint = false, turned_child_frame = false;
  for (int i = 0; i < nb; ++i) {
    world_joint |= m->jtype[i] != PD_JOINT_FREE && m->jparent[i] < 0;
    const float *qc = &m->X_c[i * 7 + 3];
    turned_child_frame |= !(qc[0] == 0.f && qc[1] == 0.f && qc[2] == 0.f && qc[3] == 1.f);
  }
  if ((jt != PD_JT_REVOLUTE && jt != PD_JT_COMPOUND) || world_joint || turned_child_frame) jt = PD_JT_REVOLUTE | PD_JT_COMPOUND | PD_JT_FIXED;
  // ---- contact table: grouped by body, kd-ordered inside a body, cut into tiles of <= segw points
  std::vector<float4> pts, tile_lo, tile_hi, mats;
  std::vector<unsigned char> pt_mat;
  std::vector<int> tile_pack;
  std::vector<int2> body_tiles(nb, make_int2(0, 0));
  std::vector<float4> body_sphere(nb, make_float4(0, 0, 0, -1.0f));
  std::vector<int> order;
  for (int b = 0; b < nb; ++b) {
    std::vector<int> ids;
    for (int k = 0; k < m->nc; ++k) if (m->cbody[k] == b) ids.push_back(k);
    body_tiles[b] = make_int2((int)tile_pack.size(), 0);
    if (ids.empty()) continue;
    body_sphere[b] = bound_sphere(ids, m->cpoint.data(), m->cdist.data());
    // the ORDER of a body's points is that of the 16-lane mapping whatever the segment width (its tiles are then cut segw points at a
    // time): hit-log entries are indices into this table, and a forward pass in one lane mapping may be followed by an adjoint in
    // another (the quad-lane forward kernel runs on the 64-lane tables, the lane-per-body adjoint on the 16-lane ones)
    kd_order(ids, 0, (int)ids.size(), m->cpoint.data(), 16);
    for (size_t t0 = 0; t0 < ids.size(); t0 += segw) {
      std::vector<int> tid(ids.begin() + t0, ids.begin() + std::min(ids.size(), t0 + segw));
      float4 blo, bhi;
      bound_box(tid, m->cpoint.data(), m->cdist.data(), blo, bhi);
      tile_lo.push_back(blo); tile_hi.push_back(bhi);
      tile_pack.push_back((int)pts.size() | ((int)tid.size() << 16) | (b << 24));
      body_tiles[b].y++;
      for (int k : tid) {
        pts.push_back(make_float4(m->cpoint[k * 3], m->cpoint[k * 3 + 1], m->cpoint[k * 3 + 2], m->cdist[k]));
        int mi = m->cmat[k];
        if (mi < 0 || mi >= m->nmat) return fail("contact_material out of range");
        pt_mat.push_back((unsigned char)mi);
        order.push_back(k);
      }
    }
  }
  const int nc = (int)pts.size(), ntiles = (int)tile_pack.size();
  // small bodies (<= 8 tiles) contribute their tiles to a static flat list of at most 4*segw entries, laid out as
  // 4 chunks of 64 slots so that lane l of a segment reads entry u*64 + l; the rest are "big" (cooperative L2)
  std::vector<int> small_tiles(4 * 64, -1);
  unsigned long long big_bodies = 0ull;
  int n_small = 0;
  for (int b = 0; b < nb; ++b) {
    int nt = body_tiles[b].y;
    if (nt == 0) continue;
    if (nt <= 8 && n_small + nt <= 4 * segw) {
      for (int t = 0; t < nt; ++t, ++n_small) small_tiles[(n_small / segw) * 64 + n_small % segw] = (body_tiles[b].x + t) | (b << 16);
    } else {
      big_bodies |= 1ull << b;
    }
  }
  if (nc > 65535) return fail("more than 65535 contact candidates per articulation is not supported");
  if (m->nmat > 255) return fail("more than 255 contact materials is not supported");
  for (int i = 0; i < m->nmat; ++i)
    mats.push_back(make_float4(m->materials[i * 4], m->materials[i * 4 + 1], m->materials[i * 4 + 2], m->materials[i * 4 + 3]));
  if (mats.empty()) mats.push_back(make_float4(0, 0, 0, 0));
  if (pts.empty()) pts.push_back(make_float4(0, 0, 0, 0));
  pt_mat.resize(((std::max(nc, 1) + 15) / 16) * 16, 0);
  if (tile_pack.empty()) { tile_pack.push_back(0); tile_lo.push_back(make_float4(0, 0, 0, 0)); tile_hi.push_back(make_float4(0, 0, 0, 0)); }

  // ---- upload
  std::vector<unsigned char> buf;
  size_t o_jtype = put(buf, m->jtype), o_jparent = put(buf, m->jparent), o_qstart = put(buf, m->qstart), o_qdstart = put(buf, m->qdstart);
  size_t o_depth = put(buf, depth), o_children = put(buf, children);
  size_t o_Xp = put(buf, m->X_p), o_Xc = put(buf, m->X_c), o_axis = put(buf, m->axis), o_com = put(buf, m->com);
  size_t o_lo = put(buf, m->lim_lo), o_hi = put(buf, m->lim_hi), o_lke = put(buf, m->lim_ke), o_lkd = put(buf, m->lim_kd);
  size_t o_pts = put(buf, pts), o_ptm = put(buf, pt_mat), o_mats = put(buf, mats);
  size_t o_bs = put(buf, body_sphere), o_ts = put(buf, tile_lo), o_th = put(buf, tile_hi), o_ti = put(buf, tile_pack), o_bt = put(buf, body_tiles), o_st = put(buf, small_tiles);
  // ---- sizes first (nothing is touched if the model does not fit)
  PdDevModel d{};
  d.nb = nb; d.nq = m->nq; d.nqd = m->nqd; d.nc = nc; d.ntiles = ntiles;
  d.max_children = max_children; d.max_depth = max_depth;
  d.nmat = m->nmat;
  d.big_bodies = big_bodies; d.n_small = n_small;
  d.list_cap = std::max(ntiles, 2 * nb);
  d.has_limits = 0;
  for (int i = 0; i < m->nqd; ++i) if (m->lim_ke[i] != 0.f || m->lim_kd[i] != 0.f) d.has_limits = 1;
  d.gx = m->gravity[0]; d.gy = m->gravity[1]; d.gz = m->gravity[2];
  d.attach_ke = m->attach_ke; d.attach_kd = m->attach_kd;
  // the speculative cull's margin (pd_kernels.hip sink_margin): tight where candidates are many (mesh robots), generous where they are few
  d.spec_safety = nc > 512 ? 1.25f : 3.0f; d.spec_slack = nc > 512 ? 1.0e-4f : 1.0e-3f;
  d.X_p_env = m->xp_env; d.xp_envs = m->xp_envs;
  // cull vectors (float4 per body, 16-B aligned) + records + wrench slots + adjoint slots + tile list + hit list (8*segw) + per-hit result slots (13*segw)
  d.env_lds_floats = ((nb * (4 + PD_REC + PD_W6 + 2 * PD_ADJ) + PD_ADJ + std::max(ntiles, 2 * nb) + 8 * segw + PD_ADJ * segw + 3) / 4) * 4 + 4;  // + PD_ADJ: the zero record
  // lanes (env e, body b) of one wave address base_e + f(b): an env stride of 16 mod 32 floats lets the envs of a wave
  // alternate between the two halves of the 32 LDS banks (2-way, the minimum for 64 lanes) instead of piling onto one
  d.env_lds_floats += (16 - d.env_lds_floats % 32 + 32) % 32;
  const int envs_per_block = PD_BWAVES * (64 / segw);
  const size_t lds_tables = (size_t)std::max(nc, 1) * 16 + (size_t)std::max(ntiles, 1) * 32 + (size_t)std::max(m->nmat, 1) * 16 +
                            (size_t)((std::max(ntiles, 1) + 3) & ~3) * 4 + (size_t)((nb + 1) & ~1) * 8 + (size_t)((std::max(nc, 1) + 15) & ~15);
  const size_t lds_rollout = lds_tables + (size_t)envs_per_block * d.env_lds_floats * 4;
  {
    int dev_id = 0, cus = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess) cus = 0;
    d.cu_count = cus;
  }
  // the role-split adjoint (k_rollout_bwd3) keeps the contact tables in global memory; per env: two generations of cull vectors and records,
  // wrench adjoints, (parent, own) joint slots + the zero record, contact sums, inertia-gradient accumulators, tile list, hit list, per-hit slots, signals
  d.env_lds_jc = ((nb * PD_JC + 31) / 32) * 32;
  // quad-lane adjoint (the 64-lane copy of an eligible model, build_quad): PD_QGEN generations of what its state wave stages ahead of the
  // other two -- cull vectors + records, the state-only values of the body wave's step (PD_QPRE floats per lane), joint hand-over records
  d.env_lds_rec2 = (segw == 64 && jt == PD_JT_REVOLUTE && nb <= 16) ? PD_QGEN * (((nb * (4 + PD_REC) + 31) / 32) * 32 + PD_QPRE * 64 + d.env_lds_jc) : 0;
  d.env_lds_bwd3 = ((nb * (2 * (4 + PD_REC) + PD_W6 + 3 * PD_ADJ + PD_GACC) + PD_GACC + PD_ADJ + std::max(ntiles, 2 * nb) + 8 * segw + PD_ADJ * segw + 3) / 4) * 4 + 4;
  d.env_lds_bwd3 += (16 - d.env_lds_bwd3 % 32 + 32) % 32;  // env stride 16 mod 32, as above
  // revolute-only: 2-role kernel (+ joint hand-over records) or the 3-role one, no tables; other joint mixes: the 2-role
  // k_rollout_bwd3 with the contact tables in LDS (or the unsplit kernel, A/B only)
  const size_t lds_rollout_bwd = jt == PD_JT_REVOLUTE
                                     ? (size_t)envs_per_block * std::max(d.env_lds_bwd3, d.env_lds_floats + 2 * d.env_lds_jc + d.env_lds_rec2) * 4
                                     : lds_tables + (size_t)envs_per_block * std::max(d.env_lds_bwd3, d.env_lds_floats) * 4;
  // (the 160 KiB checks below are for PD_BWAVES env groups per workgroup, the most a launch uses)
  const size_t lds_fk = (size_t)envs_per_block * nb * (PD_REC + PD_ADJ) * 4;
  if (lds_rollout_bwd > 160 * 1024) return fail("model needs " + std::to_string(lds_rollout_bwd) + " B of LDS per workgroup (> 160 KiB); use a wider segment");
  if (lds_rollout > 160 * 1024) return fail("model needs " + std::to_string(lds_rollout) + " B of LDS per workgroup (> 160 KiB); use a wider segment");
  // (k_reduce_fk runs FK workgroups of PD_REDUCE_BLOCK / 64 body waves: that many times the 4-wave FK workgroup's records)
  const size_t lds_fk_wide = lds_fk * (PD_REDUCE_BLOCK / 64 / PD_BWAVES);
  if (lds_fk_wide > 160 * 1024) return fail("model needs " + std::to_string(lds_fk_wide) + " B of LDS per FK workgroup (> 160 KiB); use a wider segment");
  const int lds_max = (int)std::max(std::max(std::max(lds_rollout, lds_rollout_bwd), lds_fk), lds_fk_wide);
  int dev_id = 0;
  if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0) dev_id = 0;
  int uncached = 0;  // (a device index beyond the table is simply never cached: the attribute is set on every build)
  int &attr = dev_id < PD_MAX_DEVICES ? g_lds_attr[dev_id][segw == 16 ? 0 : (segw == 32 ? 1 : 2)][jt_slot(jt)] : uncached;
  if (lds_max > attr) {
    hipError_t ea = segw == 16 ? pd_set_lds_seg16(jt, lds_max) : (segw == 32 ? pd_set_lds_seg32(jt, lds_max) : pd_set_lds_seg64(jt, lds_max));
    if (ea == hipSuccess)  // ... and on the PD_NUM_LITERAL kernels, whichever policy the model runs under now
      ea = segw == 16 ? pd_set_lds_seg16_literal(jt, lds_max) : (segw == 32 ? pd_set_lds_seg32_literal(jt, lds_max) : pd_set_lds_seg64_literal(jt, lds_max));
    if (ea != hipSuccess) return hip_fail(ea, "hipFuncSetAttribute(LDS)");
    attr = lds_max;
  }
  // ---- upload into a fresh blob, then commit
  void *blob = nullptr;
  hipError_t e = hipMalloc(&blob, buf.size());
  if (e != hipSuccess) return hip_fail(e, "hipMalloc(model)");
  e = hipMemcpy(blob, buf.data(), buf.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(blob); return hip_fail(e, "hipMemcpy(model)"); }
  unsigned char *base = (unsigned char *)blob;
  d.jtype = (const int *)(base + o_jtype); d.jparent = (const int *)(base + o_jparent);
  d.qstart = (const int *)(base + o_qstart); d.qdstart = (const int *)(base + o_qdstart);
  d.depth = (const int *)(base + o_depth); d.children = (const unsigned long long *)(base + o_children);
  d.X_p = (const float *)(base + o_Xp); d.X_c = (const float *)(base + o_Xc);
  d.axis = (const float *)(base + o_axis); d.com = (const float *)(base + o_com);
  d.lim_lo = (const float *)(base + o_lo); d.lim_hi = (const float *)(base + o_hi);
  d.lim_ke = (const float *)(base + o_lke); d.lim_kd = (const float *)(base + o_lkd);
  d.pts = (const float4 *)(base + o_pts); d.pt_mat = base + o_ptm; d.materials = (const float4 *)(base + o_mats);
  d.body_sphere = (const float4 *)(base + o_bs); d.tile_lo = (const float4 *)(base + o_ts); d.tile_hi = (const float4 *)(base + o_th);
  d.tile_pack = (const int *)(base + o_ti); d.body_tiles = (const int2 *)(base + o_bt);
  d.small_tiles = (const int *)(base + o_st);
  free_device(m);
  m->blob = blob; m->dev = d;
  m->lds_rollout = lds_rollout; m->lds_rollout_bwd = lds_rollout_bwd; m->lds_fk = lds_fk; m->lds_tables = lds_tables; m->lds_max = (size_t)lds_max;
  m->segw = segw; m->jt = jt; m->contact_order = order;
  return 0;
}

// The quad-lane kernels' device copy (64-lane mapping).  Eligible: revolute-only PLAIN models (pd_parented: what the specialised
// instantiation assumes) with at most 16 bodies.  Not an error when the model is not eligible or the copy does not fit: the
// lane-per-body kernels then serve every batch size.
static void build_quad(pd_model *m) {
  free_quad(m);
  if (m->jt != PD_JT_REVOLUTE || m->nb > 16) return;
  pd_model tmp = *m;  // host arrays are copied, the device copy is built into the temporary and moved over
  tmp.blob = nullptr; tmp.quad = nullptr;
  if (build_device(&tmp, 64) != 0 || tmp.jt != PD_JT_REVOLUTE) { if (tmp.blob) (void)hipFree(tmp.blob); return; }
  m->quad = new pd_model::Quad{tmp.blob, tmp.dev, tmp.lds_tables, tmp.jt};
}

static int g_groups = 0;   // -DPD_STAMPS diagnostic builds only (pd_debug_set_groups): env groups per workgroup, 0 = automatic

// Launch geometry of one call: kernel variant, env groups per workgroup (small batches spread over all CUs), LDS bytes.
static PdLaunchCfg launch_cfg(const pd_model *m, int kind, int n_envs, bool loss = false) {
  const PdDevModel &d = m->dev;
  const int epw = 64 / m->segw, n_groups = (n_envs + epw - 1) / epw;
  PdLaunchCfg c{};
  c.kernel = pd_kernel_variant(kind, m->jt, n_groups, d.cu_count);
  // the loss-evaluating forward exists wave-specialised only (round 6): its unsplit instantiation does not survive the register allocator in
  // the branch-free form the other forward kernels of plain models have, and an env must give the same bits whichever kernel runs it
  if (loss && c.kernel == PD_KV_FWD_UNSPLIT) c.kernel = PD_KV_FWD_SPLIT;
  c.roles = pd_variant_roles(c.kernel);
#ifndef PD_NO_CULLW
  // revolute-only robots: a third wave per env group runs the speculative contact cull (k_rollout_fwd CULLW)
  if (c.kernel == PD_KV_FWD_SPLIT && m->jt == PD_JT_REVOLUTE && pd_fwd_cull_cap(d.nb, m->segw, d.list_cap, d.env_lds_floats, PD_REC, PD_W6) >= PD_CULLW_MIN_CAP)
    c.roles = 3;
#endif
  c.groups = kind <= PD_K_ROLLOUT_BWD ? (g_groups ? g_groups : pd_groups_per_wg(n_groups, d.cu_count)) : PD_BWAVES;
  c.nblocks = (n_groups + c.groups - 1) / c.groups;
  c.threads = c.roles * c.groups * 64;
  const size_t envs = (size_t)c.groups * epw;
  switch (c.kernel) {
    case PD_KV_FWD_SPLIT: case PD_KV_FWD_UNSPLIT: case PD_KV_BWD_UNSPLIT: c.lds = m->lds_tables + envs * d.env_lds_floats * 4; break;
    case PD_KV_BWD_2ROLE: case PD_KV_BWD_2ROLE_EARLY: c.lds = envs * (d.env_lds_floats + 2 * d.env_lds_jc) * 4; break;
    case PD_KV_BWD_3ROLE: c.lds = envs * d.env_lds_bwd3 * 4; break;
    case PD_KV_BWD3_2ROLE: c.lds = m->lds_tables + envs * d.env_lds_bwd3 * 4; break;
    default: c.lds = m->lds_fk; break;
  }
  return c;
}

// Small batches of an eligible robot take the quad-lane forward kernel: one env per wave pair, so up to PD_BWAVES x CUs envs fill the
// chip with one workgroup per CU (1 024 on MI355X); beyond that the lane-per-body kernels (four envs per wave) have the throughput.
static bool use_quad(const pd_model *m, int kind, int n_envs, const void *args) {
  if (!m->quad || m->family == 1 || (kind != PD_K_ROLLOUT_FWD && kind != PD_K_ROLLOUT_BWD)) return false;
  // forward: while one workgroup per CU holds the batch (4 x CUs envs); adjoint: while every wave has a SIMD to itself (2 x CUs) --
  // measured: 1 024 envs forward 0.153 ms against 0.184, adjoint 0.273 against 0.244 (two quad pairs per SIMD lose)
  return m->family == 2 || n_envs <= (kind == PD_K_ROLLOUT_FWD ? PD_BWAVES : PD_BWAVES / 2) * m->quad->dev.cu_count;
}

static hipError_t launch(const pd_model *m, int kind, const void *args, int n_envs, hipStream_t st) {
  if (use_quad(m, kind, n_envs, args)) {
    const PdDevModel &d = m->quad->dev;
    PdLaunchCfg c{};
    c.kernel = kind == PD_K_ROLLOUT_FWD ? PD_KV_FWD_QUAD : PD_KV_BWD_QUAD;
    c.roles = 3;   // forward: body, contact and cull wave per env; adjoint: body, contact and state wave
#ifdef PD_NO_CULLW
    if (kind == PD_K_ROLLOUT_FWD) c.roles = 2;
#endif
    if (kind == PD_K_ROLLOUT_FWD && pd_fwd_cull_cap(d.nb, 64, d.list_cap, d.env_lds_floats, PD_REC, PD_W6) < PD_CULLW_MIN_CAP) c.roles = 2;
    c.groups = g_groups ? g_groups : pd_groups_per_wg(n_envs, d.cu_count);
    if (kind == PD_K_ROLLOUT_BWD && c.groups > 2) c.groups = 2;   // (three roles: at most 384 threads, two waves per SIMD -- the body wave needs its 256 VGPRs)
    c.nblocks = (n_envs + c.groups - 1) / c.groups;
    c.threads = c.roles * c.groups * 64;
    c.lds = kind == PD_K_ROLLOUT_FWD ? m->quad->lds_tables + (size_t)c.groups * d.env_lds_floats * 4   // contact tables in LDS
                                     : (size_t)c.groups * (d.env_lds_floats + 2 * d.env_lds_jc + d.env_lds_rec2) * 4;  // adjoint: + joint hand-over records + second generation of records
    if (c.nblocks == 0) return hipSuccess;
    int *ll = const_cast<pd_model *>(m)->last_launch[kind];
    ll[0] = c.nblocks; ll[1] = c.threads; ll[2] = (int)c.lds; ll[3] = c.groups;
    return (m->policy == PD_NUM_LITERAL ? pd_launch_seg64_literal : pd_launch_seg64)(kind, m->quad->jt, d, args, c, st);
  }
  const PdLaunchCfg c = launch_cfg(m, kind, n_envs, kind == PD_K_ROLLOUT_FWD && ((const RolloutArgs *)args)->loss_target != nullptr);
  if (c.nblocks == 0) return hipSuccess;
  if (kind < 2) {
    int *ll = const_cast<pd_model *>(m)->last_launch[kind];
    ll[0] = c.nblocks; ll[1] = c.threads; ll[2] = (int)c.lds; ll[3] = c.groups * (64 / m->segw);
  }
  if (m->policy == PD_NUM_LITERAL && kind <= PD_K_ROLLOUT_BWD) {  // (FK evaluates no joint force: one copy of those kernels)
    if (m->segw == 16) return pd_launch_seg16_literal(kind, m->jt, m->dev, args, c, st);
    if (m->segw == 32) return pd_launch_seg32_literal(kind, m->jt, m->dev, args, c, st);
    return pd_launch_seg64_literal(kind, m->jt, m->dev, args, c, st);
  }
  if (m->segw == 16) return pd_launch_seg16(kind, m->jt, m->dev, args, c, st);
  if (m->segw == 32) return pd_launch_seg32(kind, m->jt, m->dev, args, c, st);
  return pd_launch_seg64(kind, m->jt, m->dev, args, c, st);
}

static void timing_begin(pd_model *m, int kind, hipStream_t st) {
  if (!m->timing) return;
  if (!m->ev[kind][0]) { (void)hipEventCreate(&m->ev[kind][0]); (void)hipEventCreate(&m->ev[kind][1]); }
  (void)hipEventRecord(m->ev[kind][0], st);
}
static void timing_end(pd_model *m, int kind, hipStream_t st) {
  if (!m->timing) return;
  (void)hipEventRecord(m->ev[kind][1], st);
  m->ev_valid[kind] = true;
}

// Validates frame2step (host) and returns the cached device table frame_of_step[nsteps + 1] (frame index of each state, or
// -1).  A new (nsteps, frame2step) costs one allocation and one synchronous upload; a repeated one costs a compare.
static int frame_table(pd_model *m, int nsteps, int nframes, const int *f2s, const int **out, hipStream_t st) {
  if (nframes < 0) return fail("negative frame count");
  if (nframes > 0 && !f2s) return fail("null frame2step");
  std::vector<int> fos((size_t)nsteps + 1, -1);
  for (int f = 0; f < nframes; ++f) {
    const int s = f2s[f];
    if (s < 0 || s > nsteps)
      return fail("frame2step[" + std::to_string(f) + "] = " + std::to_string(s) + " is outside 0.." + std::to_string(nsteps));
    if (fos[s] >= 0)
      return fail("frame2step names step " + std::to_string(s) + " twice (frames " + std::to_string(fos[s]) + " and " + std::to_string(f) + ")");
    fos[s] = f;
  }
  for (const auto &e : m->fos)
    if (e.nsteps == nsteps && (int)e.f2s.size() == nframes && std::equal(e.f2s.begin(), e.f2s.end(), f2s)) { *out = e.dev; return 0; }
  // The tables are a few hundred bytes each and their device pointers may be baked into HIP graphs the caller captured
  // (the header: "warm up, then capture"), so NONE is freed before pd_model_destroy -- no eviction, no synchronisation that
  // would also invalidate a capture in progress.  A caller that keeps inventing frame lists is told so instead.
  if (m->fos.size() >= 4096) return fail("more than 4096 distinct (nsteps, frame2step) tables on one model; destroy and recreate the model");
  {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (st != nullptr && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
      return fail("a new (nsteps, frame2step) needs a synchronous upload, which a stream capture in progress forbids: warm up first");
  }
  int *dev = nullptr;
  hipError_t e = hipMalloc((void **)&dev, fos.size() * sizeof(int));
  if (e != hipSuccess) return hip_fail(e, "hipMalloc(frame table)");
  e = hipMemcpy(dev, fos.data(), fos.size() * sizeof(int), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(dev); return hip_fail(e, "hipMemcpy(frame table)"); }
  m->fos.push_back({nsteps, std::vector<int>(f2s, f2s + nframes), dev});
  *out = dev;
  return 0;
}

static unsigned long long *g_dbg = nullptr;  // diagnostic builds only

extern "C" {

int pd_abi_version(void) { return PD_ABI_VERSION; }
const char *pd_last_error(void) { return g_err.c_str(); }

int pd_model_create(const pd_model_desc *d, pd_model **out) {
  if (!d || !out) return fail("null argument");
  if (d->nb <= 0 || d->nb > 64) return fail("nb must be in 1..64 (one articulation per wavefront segment)");
  pd_model *m = new pd_model();
  m->nb = d->nb; m->nq = d->nq; m->nqd = d->nqd; m->nc = d->nc; m->nmat = d->nmat;
  auto vi = [](const int *p, size_t n) { return std::vector<int>(p, p + n); };
  auto vf = [](const float *p, size_t n) { return std::vector<float>(p, p + n); };
  m->jtype = vi(d->joint_type, d->nb); m->jparent = vi(d->joint_parent, d->nb);
  m->qstart = vi(d->joint_q_start, d->nb); m->qdstart = vi(d->joint_qd_start, d->nb);
  m->X_p = vf(d->joint_X_p, d->nb * 7); m->X_c = vf(d->joint_X_c, d->nb * 7);
  m->axis = vf(d->joint_axis, d->nb * 3); m->com = vf(d->body_com, d->nb * 3);
  m->lim_lo = vf(d->joint_limit_lower, d->nqd); m->lim_hi = vf(d->joint_limit_upper, d->nqd);
  m->lim_ke = vf(d->joint_limit_ke, d->nqd); m->lim_kd = vf(d->joint_limit_kd, d->nqd);
  m->cbody = vi(d->contact_body, d->nc); m->cmat = vi(d->contact_material, d->nc);
  m->cpoint = vf(d->contact_point, (size_t)d->nc * 3); m->cdist = vf(d->contact_dist, d->nc);
  m->materials = vf(d->shape_materials, (size_t)d->nmat * 4);
  memcpy(m->gravity, d->gravity, sizeof(float) * 3);
  m->attach_ke = d->joint_attach_ke; m->attach_kd = d->joint_attach_kd;
  for (int k = 0; k < d->nc; ++k)
    if (m->cbody[k] < 0 || m->cbody[k] >= d->nb) { delete m; return fail("contact_body out of range"); }
  // default segment width: the narrowest that holds the bodies AND whose workgroup (contact tables + 64/width envs per
  // wave) fits the 160 KiB of LDS -- a robot with more contact candidates gets fewer envs per workgroup instead of an error
  {
    int rc = 1;
    for (int w = m->nb <= 16 ? 16 : (m->nb <= 32 ? 32 : 64); w <= 64 && rc; w *= 2) rc = build_device(m, w);
    if (rc) { free_device(m); delete m; return 1; }
  }
  build_quad(m);
  *out = m;
  return 0;
}

void pd_model_destroy(pd_model *m) {
  if (!m) return;
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 2; ++j)
      if (m->ev[k][j]) (void)hipEventDestroy(m->ev[k][j]);
  for (auto &e : m->fos) (void)hipFree(e.dev);
  free_device(m);
  free_quad(m);
  delete m;
}

// Setup-time call: the device copy is rebuilt (one synchronisation, because launches in flight may still read the old one).
int pd_model_set_segment_width(pd_model *m, int lanes) {
  if (!m) return fail("null model");
  (void)hipDeviceSynchronize();
  if (build_device(m, lanes)) return 1;
  build_quad(m);
  return 0;
}

int pd_model_set_kernel_family(pd_model *m, int family) {
  if (!m) return fail("null model");
  if (family < 0 || family > 2) return fail("kernel family: 0 automatic, 1 lane per body, 2 quad-lane where eligible");
  m->family = family;
  return 0;
}
int pd_model_get_kernel_family(const pd_model *m, int *eligible) {
  if (eligible) *eligible = (m && m->quad) ? 1 : 0;
  return m ? m->family : 0;
}
int pd_model_set_numeric_policy(pd_model *m, int policy) {
  if (!m) return fail("null model");
  if (policy != PD_NUM_STABLE && policy != PD_NUM_LITERAL) return fail("numeric policy: PD_NUM_STABLE (0) or PD_NUM_LITERAL (1)");
  m->policy = policy;
  return 0;
}
int pd_model_get_numeric_policy(const pd_model *m) { return m ? m->policy : 0; }
int pd_model_get_segment_width(const pd_model *m) { return m ? m->segw : 0; }

int pd_model_bind_joint_X_p(pd_model *m, const float *joint_X_p_dev, int n_envs) {
  if (!m) return fail("null model");
  if ((joint_X_p_dev == nullptr) != (n_envs == 0) || n_envs < 0) return fail("joint_X_p binding needs a device pointer and n_envs > 0 (or NULL and 0 to unbind)");
  m->xp_env = joint_X_p_dev; m->xp_envs = n_envs;
  m->dev.X_p_env = joint_X_p_dev; m->dev.xp_envs = n_envs;
  if (m->quad) { m->quad->dev.X_p_env = joint_X_p_dev; m->quad->dev.xp_envs = n_envs; }
  return 0;
}

int pd_model_contact_order(const pd_model *m, int *order_host, int capacity) {
  if (!m || !order_host) return fail("null argument");
  if (capacity < (int)m->contact_order.size()) return fail("contact order needs room for " + std::to_string(m->contact_order.size()) + " ints");
  std::copy(m->contact_order.begin(), m->contact_order.end(), order_host);
  return 0;
}

size_t pd_rollout_workspace_floats(const pd_model *m, int bs, int nsteps) {
  return m ? (size_t)nsteps * PD_TRAJ_FLOATS * (size_t)bs * m->nb + (size_t)nsteps * (size_t)bs * PD_HITLOG : 0;  // + hit log (ints)
}

// trajectory-loss extras of the two *_traj_loss entries (null = the plain rollout)
struct TrajLossFwd { const float *target; const unsigned char *outseq; float rot_ratio; float *seed_pos, *seed_gt, *table, *reduced, *scale; };
struct TrajLossBwd { const float *seed_pos, *scale, *gain; float *work; };
extern "C" __attribute__((visibility("hidden"))) int pd_traj_loss_reduce_launch(int bs, int nframes, const float *table, float *reduced, float *scale, hipStream_t st);  // pd_loss.hip; internal
extern "C" __attribute__((visibility("hidden"))) int pd_traj_seeds_launch(int bs, int nb, int nframes, const float *seed_pos, const float *scale, const float *gain, const float *adj_pos,
                                                                         const float *adj_vel, float *work, hipStream_t st);

// The rollout kernels address a lane's element of a step's tensors as a 64-bit step base + a 32-bit lane offset (idx * 16 bytes at most;
// idx * 9 as an int for the inertia gradients; env * 128 bytes into the hit log): bs x bodies < 2^27 and bs < 2^24 keep all of them in
// range -- 10 M Laikago envs, far more than 288 GB holds at any useful horizon (1.2 KB of saved trajectory per env-step).  Refused, not wrapped.
static int batch_too_large(const pd_model *m, int bs) {
  if ((size_t)bs * (size_t)m->nb >= ((size_t)1 << 27) || bs >= (1 << 24))
    return fail("batch too large: " + std::to_string(bs) + " envs x " + std::to_string(m->nb) + " bodies (supported: bs x bodies < 2^27 and bs < 2^24; split the batch)");
  return 0;
}

// Row f4, second half: the FK of the control reference (and its adjoint) as extra workgroups of the launch that sits between the
// rollout launches anyway -- k_reduce_fk = [reduce_loss | FK forward ...], k_seeds_fk = [seeds ... | FK backward ...] (pd_kernels.hip)
static int check_fk_ride(const pd_fk_ride *fk, bool backward) {
  if (!fk) return 0;
  if (fk->n < 0 || fk->bs < 0) return fail("negative size (fk ride)");
  if (fk->n == 0) return 0;
  if (fk->bs == 0 || fk->n % fk->bs != 0) return fail("fk ride: n = " + std::to_string(fk->n) + " articulations is not frames x bs (bs = " + std::to_string(fk->bs) + ")");
  if (!fk->joint_q_dev || !fk->joint_qd_dev) return fail("null device pointer (fk ride)");
  if (!backward && (!fk->body_q_dev || !fk->body_qd_dev)) return fail("null device pointer (fk ride)");
  if (backward && (!fk->adj_body_q_dev || !fk->adj_body_qd_dev || !fk->g_joint_q_dev || !fk->g_joint_qd_dev)) return fail("null device pointer (fk ride)");
  return 0;
}
static FkArgs fk_ride_args(const pd_fk_ride *fk) {
  FkArgs f{};
  f.n = fk->n; f.joint_q = fk->joint_q_dev; f.joint_qd = fk->joint_qd_dev; f.body_q = fk->body_q_dev; f.body_qd = fk->body_qd_dev;
  f.adj_body_q = fk->adj_body_q_dev; f.adj_body_qd = fk->adj_body_qd_dev; f.g_joint_q = fk->g_joint_q_dev; f.g_joint_qd = fk->g_joint_qd_dev;
  f.perm_bs = fk->bs;
  return f;
}
// lead_blocks workgroups of the carrying pass (needing lead_lds bytes), then the FK workgroups of fk_n articulations
static hipError_t launch_ride(const pd_model *m, int kind, const void *args, int fk_n, int lead_blocks, size_t lead_lds, hipStream_t st) {
  const int epw = 64 / m->segw, n_groups = (fk_n + epw - 1) / epw;
  PdLaunchCfg c{};
  c.kernel = PD_KV_FK; c.roles = 1;
  c.groups = kind == PD_K_REDUCE_FK ? PD_REDUCE_BLOCK / 64 : PD_BWAVES;
  c.nblocks = lead_blocks + (n_groups + c.groups - 1) / c.groups;
  c.threads = c.groups * 64;
  c.lds = std::max(lead_lds, m->lds_fk * (size_t)(c.groups / PD_BWAVES));
  if (c.nblocks == 0) return hipSuccess;
  if (m->segw == 16) return pd_launch_seg16(kind, m->jt, m->dev, args, c, st);
  if (m->segw == 32) return pd_launch_seg32(kind, m->jt, m->dev, args, c, st);
  return pd_launch_seg64(kind, m->jt, m->dev, args, c, st);
}
static int reduce_launch(const pd_model *m, int bs, int nframes, const float *table, float *reduced, float *scale, const pd_fk_ride *fk, hipStream_t st) {
  if (!fk || fk->n == 0) return pd_traj_loss_reduce_launch(bs, nframes, table, reduced, scale, st) ? fail("trajectory-loss reduction launch failed") : 0;
  ReduceFkArgs a{};
  a.fk = fk_ride_args(fk);
  a.red = TrajReduceArgs{bs, nframes, table, reduced, scale, 1, nullptr};
  const size_t bytes = (size_t)bs * nframes * sizeof(float);
  a.in_lds = bytes <= std::min((size_t)PD_REDUCE_LDS_BYTES, m->lds_max) ? 1 : 0;
  hipError_t e = launch_ride(m, PD_K_REDUCE_FK, &a, fk->n, 1, a.in_lds ? bytes : 0, st);
  return e == hipSuccess ? 0 : hip_fail(e, "reduce_loss + fk launch");
}

static int rollout_forward_impl(const pd_model *cm, int bs, int nsteps, float dt, const float *q_init, const float *qd_init,
                       const float *torques, const float *res_f, const float *refs, const float *target_ke,
                       const float *target_kd, const float *inv_mass, const float *inertia, const float *inv_inertia, int nframes,
                       const int *frame2step, float *ws, float *wp_pos, float *wp_vel, float *grf, float *jaf, const TrajLossFwd *tl,
                       const pd_fk_ride *fk, void *stream) {
  pd_model *m = const_cast<pd_model *>(cm);
  if (!m) return fail("null model");
  if (bs < 0 || nsteps < 0) return fail("negative size");
  if (batch_too_large(m, bs)) return 1;
  if (check_fk_ride(fk, false)) return 1;
  const int *fos = nullptr;
  if (frame_table(m, nsteps, nframes, frame2step, &fos, (hipStream_t)stream)) return 1;
  if (bs == 0) {  // nothing to roll out; an empty batch still gets a defined reduced loss (0) from the trajectory-loss entry
    if (tl && tl->reduced) return reduce_launch(m, 0, nframes, tl->table, tl->reduced, tl->scale, fk, (hipStream_t)stream);
    return 0;
  }
  if (!q_init || !qd_init || !target_ke || !target_kd || !inv_mass || !inertia || !inv_inertia) return fail("null device pointer");
  if (nsteps > 0 && (!torques || !res_f || !refs || !ws)) return fail("null device pointer");
  if (nframes > 0 && (!wp_pos || !wp_vel)) return fail("null device pointer");
  if (m->xp_env && m->xp_envs != bs) return fail("joint_X_p is bound for " + std::to_string(m->xp_envs) + " envs, rollout has " + std::to_string(bs));
  RolloutArgs a{};
  a.bs = bs; a.nsteps = nsteps; a.nframes = nframes; a.dt = dt;
  a.q_init = q_init; a.qd_init = qd_init; a.torques = torques; a.res_f = res_f; a.refs = refs;
  a.target_ke = target_ke; a.target_kd = target_kd; a.inv_mass = inv_mass; a.inertia = inertia; a.inv_inertia = inv_inertia;
  a.frame_of_step = fos; a.ws = ws; a.wp_pos = wp_pos; a.wp_vel = wp_vel; a.grf = grf; a.jaf = jaf; a.dbg = g_dbg;
  a.hitlog = (int *)(ws + (size_t)nsteps * PD_TRAJ_FLOATS * (size_t)bs * m->nb);
  if (tl) {
    if (nframes > 0 && (!tl->target || !tl->seed_pos || !tl->table || !tl->reduced || !tl->scale)) return fail("null device pointer (trajectory loss)");
    a.loss_target = nframes > 0 ? tl->target : nullptr; a.loss_outseq = tl->outseq; a.loss_rot_ratio = tl->rot_ratio;
    a.loss_seed_pos = tl->seed_pos; a.loss_seed_gt = tl->seed_gt; a.loss_table = tl->table;
  }
  hipStream_t st = (hipStream_t)stream;
  timing_begin(m, 0, st);
  hipError_t e = launch(m, PD_K_ROLLOUT_FWD, &a, bs, st);
  timing_end(m, 0, st);
  if (e != hipSuccess) return hip_fail(e, "rollout_forward launch");
  if (tl) return reduce_launch(m, bs, nframes, tl->table, tl->reduced, tl->scale, fk, st);
  return 0;
}

int pd_rollout_forward(const pd_model *m, int bs, int nsteps, float dt, const float *q_init, const float *qd_init,
                       const float *torques, const float *res_f, const float *refs, const float *target_ke,
                       const float *target_kd, const float *inv_mass, const float *inertia, const float *inv_inertia, int nframes,
                       const int *frame2step, float *ws, float *wp_pos, float *wp_vel, float *grf, float *jaf, void *stream) {
  return rollout_forward_impl(m, bs, nsteps, dt, q_init, qd_init, torques, res_f, refs, target_ke, target_kd, inv_mass, inertia, inv_inertia,
                              nframes, frame2step, ws, wp_pos, wp_vel, grf, jaf, nullptr, nullptr, stream);
}

int pd_rollout_forward_traj_loss(const pd_model *m, int bs, int nsteps, float dt, const float *q_init, const float *qd_init,
                                 const float *torques, const float *res_f, const float *refs, const float *target_ke,
                                 const float *target_kd, const float *inv_mass, const float *inertia, const float *inv_inertia, int nframes,
                                 const int *frame2step, float *ws, float *wp_pos, float *wp_vel, float *grf, float *jaf,
                                 const float *target_pos, const unsigned char *outseq, float rot_ratio, float *seed_pos, float *seed_gt,
                                 float *loss_table, float *reduced, float *scale, void *stream) {
  const TrajLossFwd tl{target_pos, outseq, rot_ratio, seed_pos, seed_gt, loss_table, reduced, scale};
  return rollout_forward_impl(m, bs, nsteps, dt, q_init, qd_init, torques, res_f, refs, target_ke, target_kd, inv_mass, inertia, inv_inertia,
                              nframes, frame2step, ws, wp_pos, wp_vel, grf, jaf, &tl, nullptr, stream);
}

int pd_rollout_forward_traj_loss_fk(const pd_model *m, int bs, int nsteps, float dt, const float *q_init, const float *qd_init,
                                    const float *torques, const float *res_f, const float *refs, const float *target_ke,
                                    const float *target_kd, const float *inv_mass, const float *inertia, const float *inv_inertia, int nframes,
                                    const int *frame2step, float *ws, float *wp_pos, float *wp_vel, float *grf, float *jaf,
                                    const float *target_pos, const unsigned char *outseq, float rot_ratio, float *seed_pos, float *seed_gt,
                                    float *loss_table, float *reduced, float *scale, const pd_fk_ride *fk, void *stream) {
  const TrajLossFwd tl{target_pos, outseq, rot_ratio, seed_pos, seed_gt, loss_table, reduced, scale};
  return rollout_forward_impl(m, bs, nsteps, dt, q_init, qd_init, torques, res_f, refs, target_ke, target_kd, inv_mass, inertia, inv_inertia,
                              nframes, frame2step, ws, wp_pos, wp_vel, grf, jaf, &tl, fk, stream);
}

static int rollout_backward_impl(const pd_model *cm, int bs, int nsteps, float dt, const float *q_init, const float *qd_init,
                        const float *torques, const float *refs, const float *target_ke, const float *target_kd,
                        const float *inv_mass, const float *inertia, const float *inv_inertia, int nframes, const int *frame2step,
                        const float *ws, const float *adj_pos, const float *adj_vel, float *g_q_init, float *g_qd_init,
                        float *g_torques, float *g_res_f, float *g_refs, float *g_ke, float *g_kd, float *g_inv_mass,
                        float *g_inertia, float *g_inv_inertia, const TrajLossBwd *tl, const pd_fk_ride *fk, void *stream) {
  pd_model *m = const_cast<pd_model *>(cm);
  if (!m) return fail("null model");
  if (bs < 0 || nsteps < 0) return fail("negative size");
  if (batch_too_large(m, bs)) return 1;
  if (check_fk_ride(fk, true)) return 1;
  const int *fos = nullptr;
  if (frame_table(m, nsteps, nframes, frame2step, &fos, (hipStream_t)stream)) return 1;
  if (bs == 0) {
    if (fk && fk->n > 0) {  // no rollout, but the FK adjoint that rides along still runs
      SeedsFkArgs sa{};
      sa.fk = fk_ride_args(fk);
      hipError_t e = launch_ride(m, PD_K_SEEDS_FK, &sa, fk->n, 0, 0, (hipStream_t)stream);
      if (e != hipSuccess) return hip_fail(e, "fk backward launch");
    }
    return 0;
  }
  if (!q_init || !qd_init || !target_ke || !target_kd || !inv_mass || !inertia || !inv_inertia || !g_q_init || !g_qd_init || !g_ke ||
      !g_kd || !g_inv_mass || !g_inertia || !g_inv_inertia)
    return fail("null device pointer");
  if (nsteps > 0 && (!torques || !refs || !ws || !g_torques || !g_res_f || !g_refs)) return fail("null device pointer");
  if (!tl && nframes > 0 && (!adj_pos || !adj_vel)) return fail("null device pointer");
  if (tl && ((adj_pos == nullptr) != (adj_vel == nullptr))) return fail("adj_pos and adj_vel come together (both, or neither)");
  if (tl && nframes > 0 && (!tl->seed_pos || !tl->scale || !tl->gain || !tl->work)) return fail("null device pointer (trajectory loss)");
  if (m->xp_env && m->xp_envs != bs) return fail("joint_X_p is bound for " + std::to_string(m->xp_envs) + " envs, rollout has " + std::to_string(bs));
  RolloutArgs a{};
  a.bs = bs; a.nsteps = nsteps; a.nframes = nframes; a.dt = dt;
  if (fk && fk->n > 0) {  // seeds pass (if any) + FK backward workgroups in one launch
    SeedsFkArgs sa{};
    sa.fk = fk_ride_args(fk);
    const bool seeds = tl && nframes > 0;
    if (seeds) sa.seeds = TrajSeedsArgs{bs, m->nb, nframes, tl->seed_pos, tl->scale, tl->gain, adj_pos, adj_vel, tl->work, pd_traj_seeds_blocks(bs, m->nb, nframes)};
    hipError_t e = launch_ride(m, PD_K_SEEDS_FK, &sa, fk->n, sa.seeds.nblocks, 0, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "seeds + fk backward launch");
    if (seeds) { adj_pos = tl->work; adj_vel = tl->work + (size_t)nframes * bs * m->nb * 7; }
  } else if (tl && nframes > 0) {
    // the seeds of this sweep, built on the device: work = [F][bs*nb][7] poses then [F][bs*nb][6] twists (pd_trajloss.h traj_seeds_block)
    if (pd_traj_seeds_launch(bs, m->nb, nframes, tl->seed_pos, tl->scale, tl->gain, adj_pos, adj_vel, tl->work, (hipStream_t)stream)) return fail("seed launch failed");
    adj_pos = tl->work; adj_vel = tl->work + (size_t)nframes * bs * m->nb * 7;
  }
  a.q_init = q_init; a.qd_init = qd_init; a.torques = torques; a.refs = refs;
  a.target_ke = target_ke; a.target_kd = target_kd; a.inv_mass = inv_mass; a.inertia = inertia; a.inv_inertia = inv_inertia;
  a.frame_of_step = fos; a.ws = const_cast<float *>(ws); a.adj_pos = adj_pos; a.adj_vel = adj_vel;
  a.g_q_init = g_q_init; a.g_qd_init = g_qd_init; a.g_torques = g_torques; a.g_res_f = g_res_f; a.g_refs = g_refs;
  a.g_ke = g_ke; a.g_kd = g_kd; a.g_inv_mass = g_inv_mass; a.g_inertia = g_inertia; a.g_inv_inertia = g_inv_inertia; a.dbg = g_dbg;
  a.hitlog = (int *)(const_cast<float *>(ws) + (size_t)nsteps * PD_TRAJ_FLOATS * (size_t)bs * m->nb);
  hipStream_t st = (hipStream_t)stream;
  timing_begin(m, 1, st);
  hipError_t e = launch(m, PD_K_ROLLOUT_BWD, &a, bs, st);
  timing_end(m, 1, st);
  return e == hipSuccess ? 0 : hip_fail(e, "rollout_backward launch");
}

int pd_rollout_backward(const pd_model *m, int bs, int nsteps, float dt, const float *q_init, const float *qd_init,
                        const float *torques, const float *refs, const float *target_ke, const float *target_kd,
                        const float *inv_mass, const float *inertia, const float *inv_inertia, int nframes, const int *frame2step,
                        const float *ws, const float *adj_pos, const float *adj_vel, float *g_q_init, float *g_qd_init,
                        float *g_torques, float *g_res_f, float *g_refs, float *g_ke, float *g_kd, float *g_inv_mass,
                        float *g_inertia, float *g_inv_inertia, void *stream) {
  return rollout_backward_impl(m, bs, nsteps, dt, q_init, qd_init, torques, refs, target_ke, target_kd, inv_mass, inertia, inv_inertia, nframes,
                               frame2step, ws, adj_pos, adj_vel, g_q_init, g_qd_init, g_torques, g_res_f, g_refs, g_ke, g_kd, g_inv_mass,
                               g_inertia, g_inv_inertia, nullptr, nullptr, stream);
}

int pd_rollout_backward_traj_loss(const pd_model *m, int bs, int nsteps, float dt, const float *q_init, const float *qd_init,
                                  const float *torques, const float *refs, const float *target_ke, const float *target_kd,
                                  const float *inv_mass, const float *inertia, const float *inv_inertia, int nframes, const int *frame2step,
                                  const float *ws, const float *adj_pos, const float *adj_vel, const float *seed_pos, const float *scale,
                                  const float *g_loss, float *seed_work, float *g_q_init, float *g_qd_init, float *g_torques, float *g_res_f, float *g_refs,
                                  float *g_ke, float *g_kd, float *g_inv_mass, float *g_inertia, float *g_inv_inertia, void *stream) {
  const TrajLossBwd tl{seed_pos, scale, g_loss, seed_work};
  return rollout_backward_impl(m, bs, nsteps, dt, q_init, qd_init, torques, refs, target_ke, target_kd, inv_mass, inertia, inv_inertia, nframes,
                               frame2step, ws, adj_pos, adj_vel, g_q_init, g_qd_init, g_torques, g_res_f, g_refs, g_ke, g_kd, g_inv_mass,
                               g_inertia, g_inv_inertia, &tl, nullptr, stream);
}

int pd_rollout_backward_traj_loss_fk(const pd_model *m, int bs, int nsteps, float dt, const float *q_init, const float *qd_init,
                                     const float *torques, const float *refs, const float *target_ke, const float *target_kd,
                                     const float *inv_mass, const float *inertia, const float *inv_inertia, int nframes, const int *frame2step,
                                     const float *ws, const float *adj_pos, const float *adj_vel, const float *seed_pos, const float *scale,
                                     const float *g_loss, float *seed_work, float *g_q_init, float *g_qd_init, float *g_torques, float *g_res_f, float *g_refs,
                                     float *g_ke, float *g_kd, float *g_inv_mass, float *g_inertia, float *g_inv_inertia, const pd_fk_ride *fk,
                                     void *stream) {
  const TrajLossBwd tl{seed_pos, scale, g_loss, seed_work};
  return rollout_backward_impl(m, bs, nsteps, dt, q_init, qd_init, torques, refs, target_ke, target_kd, inv_mass, inertia, inv_inertia, nframes,
                               frame2step, ws, adj_pos, adj_vel, g_q_init, g_qd_init, g_torques, g_res_f, g_refs, g_ke, g_kd, g_inv_mass,
                               g_inertia, g_inv_inertia, &tl, fk, stream);
}

int pd_fk_forward(const pd_model *m, int n, const float *joint_q, const float *joint_qd, float *body_q, float *body_qd, void *stream) {
  if (!m) return fail("null model");
  if (n < 0) return fail("negative size");
  if (!joint_q || !joint_qd || !body_q || !body_qd) return fail("null device pointer");
  FkArgs a{};
  a.n = n; a.joint_q = joint_q; a.joint_qd = joint_qd; a.body_q = body_q; a.body_qd = body_qd;
  hipError_t e = launch(m, PD_K_FK_FWD, &a, n, (hipStream_t)stream);
  return e == hipSuccess ? 0 : hip_fail(e, "fk_forward launch");
}

int pd_fk_backward(const pd_model *m, int n, const float *joint_q, const float *joint_qd, const float *adj_body_q,
                   const float *adj_body_qd, float *g_joint_q, float *g_joint_qd, void *stream) {
  if (!m) return fail("null model");
  if (n < 0) return fail("negative size");
  if (!joint_q || !joint_qd || !adj_body_q || !adj_body_qd || !g_joint_q || !g_joint_qd) return fail("null device pointer");
  FkArgs a{};
  a.n = n; a.joint_q = joint_q; a.joint_qd = joint_qd; a.adj_body_q = adj_body_q; a.adj_body_qd = adj_body_qd;
  a.g_joint_q = g_joint_q; a.g_joint_qd = g_joint_qd;
  hipError_t e = launch(m, PD_K_FK_BWD, &a, n, (hipStream_t)stream);
  return e == hipSuccess ? 0 : hip_fail(e, "fk_backward launch");
}

#ifdef PD_STAMPS
// Diagnostic build only (make stamps; never in the shipped library, not in the public header).
// buffer for the in-kernel phase stamps ([blocks*waves][16] u64)
void pd_debug_set_buffer(void *dev) { g_dbg = (unsigned long long *)dev; }
void pd_debug_set_groups(int g) { g_groups = g < 0 ? 0 : (g > PD_BWAVES ? PD_BWAVES : g); }
#endif

// Identity of this build: "<git HEAD or 'unknown'>+<hash of the kernel / host sources>", baked in by the Makefile.  smoke() and the
// tests compare the source hash with the sources that sit beside the library, so a stale .so cannot pass for the tree's.
#ifndef PD_BUILD_ID
#define PD_BUILD_ID "unknown+unknown"
#endif
const char *pd_build_id(void) { return PD_BUILD_ID; }

int pd_model_set_timing(pd_model *m, int on) {
  if (!m) return fail("null model");
  m->timing = on != 0;
  if (!m->timing) m->ev_valid[0] = m->ev_valid[1] = false;
  return 0;
}
float pd_last_kernel_ms(const pd_model *m, int kind) {
  if (!m || kind < 0 || kind > 1 || !m->ev_valid[kind]) return -1.0f;
  float ms = -1.0f;
  if (hipEventSynchronize(m->ev[kind][1]) != hipSuccess) return -1.0f;
  if (hipEventElapsedTime(&ms, m->ev[kind][0], m->ev[kind][1]) != hipSuccess) return -1.0f;
  return ms;
}
int pd_last_launch_info(const pd_model *m, int kind, int out[4]) {
  if (!m || !out || kind < 0 || kind > 1) return fail("bad argument");
  for (int k = 0; k < 4; ++k) out[k] = m->last_launch[kind][k];
  return 0;
}

}  // extern "C"
