// gfx950 kernels: whole-rollout forward, whole-rollout adjoint, batched FK forward / adjoint.
//
// One launch covers all T steps of all envs (the reference issues 4 launches + 3 aux ops per step
// from Python, /root/reference/diffphys/dp_model.py:1209-1234).  Per step the only HBM traffic is
// the SoA state + wrench spill for the adjoint, the step's controls, the contact hit log and frame
// outputs.  Revolute-only robots run wave-specialised (body waves + contact waves, two workgroup
// hand-overs per step through LDS signal words).  See DESIGN.md section 3.
#include "pd_device.h"
#include "pd_args.h"
#include "pd_se3.h"
#include "pd_quad.h"

// This file is compiled once per segment width AND per numeric policy (pd_math.h PD_POLICY; Makefile).  The PD_NUM_LITERAL objects carry
// the rollout kernels only, under their own names (the FK kernels evaluate no joint force: one copy serves both policies).
#if PD_POLICY == 1
#define k_rollout_fwd k_rollout_fwd_literal
#define k_rollout_bwd k_rollout_bwd_literal
#define k_rollout_bwd3 k_rollout_bwd3_literal
#define PD_LAUNCH_NAME(w) PD_CAT(PD_CAT(pd_launch_seg, w), _literal)
#define PD_SET_LDS_NAME(w) PD_CAT(PD_CAT(pd_set_lds_seg, w), _literal)
#else
#define PD_LAUNCH_NAME(w) PD_CAT(pd_launch_seg, w)
#define PD_SET_LDS_NAME(w) PD_CAT(pd_set_lds_seg, w)
#endif

// In-kernel phase stamps (diagnostic build only, never in the shipped library): cdna_hip_programming.md section 7.
#ifdef PD_STAMPS
#define STAMP_DECL unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = pd_memtime()
#define STAMP_ARGS , unsigned long long *st_acc, unsigned long long &st_t
#define STAMP_PASS , st_acc, st_t
#define STAMP(i) do { unsigned long long t2_ = pd_memtime(); st_acc[i] += t2_ - st_t; st_t = t2_; } while (0)
#define STAMP_COUNT(i, n) do { st_acc[i] += (unsigned long long)(n); } while (0)
#define STAMP_FLUSH(a) do { if ((a).dbg && (threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 16; ++i_) (a).dbg[((size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * 16 + i_] = st_acc[i_]; } while (0)
__device__ __forceinline__ unsigned long long pd_memtime() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_COUNT(i, n)
#define STAMP_FLUSH(a)
#define STAMP_ARGS
#define STAMP_PASS
#endif


// The hardware returns vector-memory results in issue order, so waiting for ANY load issued after a batch of prefetches
// waits for the whole batch.  The rollout loops therefore consume nothing they load in the same iteration: every
// global load is a prefetch for the next iteration, and the iteration opens with one explicit "everything older has
// landed" wait (free: those loads are an iteration old), which also tells the compiler that no later wait is needed.
#define PD_WAIT_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)  // vmcnt(0), expcnt / lgkmcnt untouched

// Global accesses of the rollout loops: wave-uniform base (step, plane: scalar registers and scalar arithmetic) plus a
// per-lane 32-bit BYTE offset computed once, i.e. the saddr + voffset addressing mode -- no per-lane 64-bit multiplies
// in the loops.  Offsets stay below 4 GB for any batch that fits the workspace.
typedef const __attribute__((address_space(4))) int *pd_const_int_p;  // constant address space => scalar (s_load) access
// The saddr + voffset mode only comes out of instruction selection when it SEES  (scalar base) + zext(32-bit lane offset)  in the block of the
// access.  Both halves are loop-invariant in the rollout loops, so the optimiser hoists  base + zext(offset)  as a per-lane 64-bit address
// and the loop pays a 64-bit vector add per access plus the scalar chain that feeds it.  PD_OPAQUE_V / _S make a value opaque where they
// stand (no instruction: an empty asm that "updates" the register), which keeps the two halves apart until selection (round 6).
#define PD_OPAQUE_V(x) asm volatile("" : "+v"(x))
#define PD_OPAQUE_S(x) do { if constexpr (CLONE && SPLIT && SEGW < 64) asm volatile("" : "+s"(x)); } while (0)  // (the unsplit instantiations keep these bases in vector registers, and the 64-lane segment's instantiation with all of them forced does not compile -- "illegal VGPR to SGPR copy": not forced there)
PD_DEV int ld_uniform(const int *p, int i) { return ((pd_const_int_p)(unsigned long long)p)[i]; }  // read-only input, wave-uniform index
PD_DEV float ldg(const float *ubase, unsigned boff) { return *(const float *)((const char *)ubase + boff); }
PD_DEV void stg(float *ubase, unsigned boff, float v) { *(float *)((char *)ubase + boff) = v; }
PD_DEV float4 ldg4(const float *ubase, unsigned boff) { return *(const float4 *)((const char *)ubase + boff); }
PD_DEV void stg4(float *ubase, unsigned boff, float4 v) { *(float4 *)((char *)ubase + boff) = v; }
PD_DEV float2 ldg2(const float *ubase, unsigned boff) { return *(const float2 *)((const char *)ubase + boff); }
PD_DEV void stg2(float *ubase, unsigned boff, float2 v) { *(float2 *)((char *)ubase + boff) = v; }

// Saved trajectory (workspace): per step PD_TRAJ_G planes of float4, [step][plane][bs*nb]; consecutive lanes touch consecutive
// 16-byte words, and a step costs 5 vector-memory instructions per lane instead of 19 (their issue rate, not the bytes,
// is what the rollout loops feel).  Planes:  0: q   1: (w, v.x)   2: (p, v.y)   3: (v.z, t)   4: (f, clamp mask)
// where (t, f) is the total body wrench of the step.  The adjoint's contact wave needs planes 0-1 of a body and 0-2 of its parent.
#define PD_TRAJ_G 5

// Every gradient the rollout adjoint stores goes through remove_nan (NaN -> 0, inf kept): the boundary's post-processing
// (dp_model.py:1294-1384 of the reference), done here at ONE instruction per stored value (pd_math.h grad_post: v_med3_f32)
// instead of ten passes over the tensors.
#define NZ(x) grad_post<1>(x)

// Seeds of a frame state in the reverse sweep (dp_model.py:1264-1271).  (pd_rollout_backward_traj_loss builds its seeds into
// adj_pos / adj_vel with a small launch of its own, pd_loss.hip k_traj_seeds: with the scaling here, behind run-time checks, the
// 2-role adjoint went from 252 VGPRs / no spill to 256 / 36 spilled, and as a separate instantiation it still ran 0.314 ms
// against 0.278.)
PD_DEV void add_frame_seeds(const RolloutArgs &a, int fr, int N, size_t idx, BodyAdj &gn) {
  const float *gp = a.adj_pos + ((size_t)fr * N + idx) * 7, *gv = a.adj_vel + ((size_t)fr * N + idx) * 6;
  gn.p += V3(gp[0], gp[1], gp[2]); gn.r += Q4(gp[3], gp[4], gp[5], gp[6]);
  gn.w += V3(gv[0], gv[1], gv[2]); gn.v += V3(gv[3], gv[4], gv[5]);
}

template <int SEGW>
struct Seg {
  static constexpr int EPW = 64 / SEGW;
  static constexpr unsigned long long MASK = SEGW == 64 ? ~0ull : ((1ull << SEGW) - 1ull);
};

// Stream compaction inside a segment: returns this lane's slot among the segment's lanes with pred set
// (v_mbcnt on the ballot masked to the segment: 4 VALU ops) and adds the segment's count to `total`.
struct SegMask { unsigned lo, hi; };
template <int SEGW>
PD_DEV SegMask seg_mask(int seg) {
  unsigned long long mk = Seg<SEGW>::MASK << (seg * SEGW);
  SegMask s; s.lo = (unsigned)mk; s.hi = (unsigned)(mk >> 32);
  return s;
}
PD_DEV int seg_slot(bool pred, SegMask sm, int &total) {
  unsigned long long b = __ballot(pred);
  unsigned lo = (unsigned)b & sm.lo, hi = (unsigned)(b >> 32) & sm.hi;
  int slot = total + (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
  total += __popc(lo) + __popc(hi);
  return slot;
}

// Per-body sums of per-hit results held one hit per lane (lanes 0 .. nh-1 of the segment, hits of a body contiguous), without
// touching LDS.  Iteration i lets every lane whose position p in its body's run is >= i take  z(lane - 1) + x(own)  with z the
// previous iteration's value of the lane below (one DPP lane shift) and x the lane's ORIGINAL term: z_p becomes
// ((x_{p-i} + x_{p-i+1}) + ...) + x_p, so after (run length - 1) iterations the LAST lane of a run holds the plain left-to-right
// sum of the run -- the order of the ds_add_f32 path and of every other path, so results do not depend on which path ran,
// and entries that contribute exactly 0 may sit anywhere.  The loop runs (longest run of the wave - 1) times, typically 1-3
// (a foot's few points); the first version walked the lanes one by one, (hits of the env - 1) iterations, and was 11 % of the
// adjoint kernel at 4096 envs.  2 NV VALU instructions per iteration (v_add_f32_dpp wave_shr:1 + v_cndmask).
template <int NV>
PD_DEV void seg_run_sum(float *acc, int pb, int l, int nh, bool &last) {
  const int pbp = __builtin_amdgcn_update_dpp(-1, pb, 0x138, 0xf, 0xf, false);  // body of the hit one lane down (wave_shr:1)
  const int pbn = __builtin_amdgcn_update_dpp(-1, pb, 0x130, 0xf, 0xf, false);  // ... one lane up (wave_shl:1)
  const bool in = l < nh, start = in && (l == 0 || pbp != pb);
  last = in && (l == nh - 1 || pbn != pb);
  // position in the run = distance to the nearest run start at or below this lane
  const int lane = (int)(threadIdx.x & 63);
  const unsigned long long starts = __ballot(start), below = starts & ((2ull << lane) - 1ull);
  const int pos = in ? lane - (63 - __clzll((long long)below)) : -1;
  float x[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) x[i] = acc[i];
  for (int it = 1; __ballot(pos >= it) != 0ull; ++it) {
    const bool upd = pos >= it;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[i]), 0x138, 0xf, 0xf, true));
      acc[i] = upd ? prev + x[i] : acc[i];
    }
  }
}

// Pairwise hand-over between a body wave and its contact wave through an LDS word (a step counter): the producer
// publishes after a workgroup-scope release, the consumer polls.  Unlike s_barrier it does not tie the four wave pairs of
// a workgroup together.
PD_DEV void pair_signal(int *flag, int value) {
#ifdef PD_SIGNAL_FENCE
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  // Everything handed over lives in LDS, and the LDS executes one wave's instructions in issue order: the records written above
  // land before the flag does, so the producer need not wait for them (the release fence's s_waitcnt lgkmcnt(0), one LDS round
  // trip on every hand-over).  All lanes store the same word: no exec-mask round trip either.
  asm volatile("" ::: "memory");
  __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
#endif
}
// The low 30 bits of the word are the counter; the producer may pass a flag in bit 30.  Returns the word.
#define PD_SIG_FLAG 0x40000000
PD_DEV int pair_wait(int *flag, int value) {
  int v;
  // plain polling: an s_sleep between the reads (64 clocks) costs more in wake-up delay than the issue slots the reads take from
  // the partner wave (measured, sustained timing: -2 % forward, -1 % adjoint at 4096 envs and at 512)
  while (((v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) & (PD_SIG_FLAG - 1)) < value) {
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return v;
}
// Timing experiments only (-DPD_KNOCK=bits, results are WRONG): what one role's own instruction stream costs.
//   1: role-0 waves (body / integrate) run through their waits    2: the other roles' waves run through theirs
//   4: role-0 waves leave at once                                 8: the other roles' waves leave at once
//  16: applies to the forward kernel (default: the adjoint kernels)
// e.g. 9 = body waves alone, 6 = the other role(s) alone, 3 = every wave free of the others (EXPERIMENTS.md round 6).
#ifdef PD_KNOCK
PD_DEV int pair_wait_knock(int *flag, int value, int role) {
  if (PD_KNOCK & (role ? 2 : 1)) return value;
  return pair_wait(flag, value);
}
#define PD_KNOCK_EXIT(fwd, role) do { if ((((PD_KNOCK) & 16) != 0) == (fwd) && ((PD_KNOCK) & ((role) ? 8 : 4))) return; } while (0)
#else
#define PD_KNOCK_EXIT(fwd, role)
#endif

// Ground-contact sweep for one segment (= one env).  Conservative three-level cull, then the exact
// test of the reference inside on_hit.  All tables are in LDS (copied once per workgroup):
//   L1  per body  : bounding sphere of all its candidate points vs y = 0               (lane = body)
//   L2  per tile  : tile = <= SEGW spatially compact points of ONE body; body-frame AABB support test (tighter than a
//                   sphere for the elongated tiles of a limb, fewer tiles reach L3)       (lane = tile of a surviving body)
//   L3  per point : y-row test  c = p_y + Ry . x - dist, PD_UNROLL tiles per iteration so that their LDS reads overlap;
//                   survivors are compacted into a per-env hit list                      (lane = point)
//   hit pass      : lanes = compacted hits (dense), on_hit(record, point, material) does the reference's arithmetic
// With one wavefront per SIMD nothing else hides LDS latency, so the structure minimises DEPENDENT LDS round trips.
#ifndef PD_UNROLL
#define PD_UNROLL 4          // tiles handled per L3 iteration (their LDS reads are issued back to back)
#endif
#define PD_HIT_CAP_TILES 8   // hit-list capacity in units of SEGW (two L3 iterations)

struct SweepTables {
  const float4 *pts, *tlo, *thi, *mats;
  const unsigned char *pmat;
  const int *tpack;
  const int2 *btiles;
};

PD_DEV bool cull_above(float4 cv, float4 sp) {  // true when the whole sphere is provably above y = 0
  float ylow = cv.x + (cv.y * sp.x + cv.z * sp.y + cv.w * sp.z) - sp.w;
  return ylow > 1e-4f * (1.0f + sp.w);
}
// Same for a body-frame box [lo, hi]: the lowest world height over the box is p_y + sum_i min(Ry_i lo_i, Ry_i hi_i).
PD_DEV bool cull_above_box(float4 cv, float4 lo, float4 hi) {
  float ylow = cv.x + fminf(cv.y * lo.x, cv.y * hi.x) + fminf(cv.z * lo.y, cv.z * hi.y) + fminf(cv.w * lo.z, cv.w * hi.z) - lo.w;
  return ylow > hi.w;
}

// Hit pass for one batch [j0, j0 + SEGW) of the env's compacted hit list.  Lane j computes hit j0 + j (compute() does
// the reference's arithmetic and returns NV floats, zeros when the exact test says "above ground") and parks the
// result in slot[j]; then lane b (= body b) sums the contiguous run of hits that belong to body b -- hits are appended
// tile by tile and tiles are grouped by body -- and adds it to dst[b].  No LDS float atomics: 13 ds_add_f32 per hit
// cost ~2400 cycles per step and stalled the partner wave's LDS traffic; this is plain stores plus a short
// lane-per-body loop, and the summation order is fixed.
//
// ATOMIC = true keeps ds_add_f32 (measured faster for the 6-float forward wrench: 0.358 vs 0.415 ms per rollout;
// slower for the 13-float adjoint: 0.646 vs 0.617 ms).  Both orders are fixed, so results are reproducible either way.
template <int SEGW, int NV, int DSTRIDE, bool ATOMIC, typename F>
PD_DEV void sweep_flush_batch(const SweepTables &T, const float *rec, const int *hits, float *slot, float *dst, int j0, int nh,
                              int run_start, int run_cnt, int l, int nb, F &&compute) {
  const int j = j0 + l;
  if (j < nh) {
    int e = hits[j];  // point | material << 16 | body << 24
    int pt = e & 0xffff, pb = (e >> 24) & 0x3f;
    float out[NV];
    compute(rec + pb * PD_REC, T.pts[pt], T.mats[(e >> 16) & 0xff], out);
    if (ATOMIC) {
#pragma unroll
      for (int i = 0; i < NV; ++i) atomicAdd(dst + pb * DSTRIDE + i, out[i]);
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) slot[l * NV + i] = out[i];
    }
  }
  if (ATOMIC) return;
  WAVE_SYNC();
  const int hi_all = nh < j0 + SEGW ? nh : j0 + SEGW;
  const int t_lo = (run_start > j0 ? run_start : j0) - j0;
  const int t_hi = (run_cnt > 0 ? (run_start + run_cnt < hi_all ? run_start + run_cnt : hi_all) : j0) - j0;
  // the running sum continues from dst: the result is the plain left-to-right sum over the body's hits whatever the
  // batch boundaries are (they move with entries that contribute exactly 0, and those depend on which path logged them)
  float acc[NV];
  const bool mine = l < nb && t_lo < t_hi;
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = mine ? dst[l * DSTRIDE + i] : 0.f;
  for (int t = t_lo; __ballot(t < t_hi) != 0ull; ++t) {
    if (t < t_hi) {
#pragma unroll
      for (int i = 0; i < NV; ++i) acc[i] += slot[t * NV + i];
    }
  }
  if (mine) {
#pragma unroll
    for (int i = 0; i < NV; ++i) dst[l * DSTRIDE + i] = acc[i];
  }
  WAVE_SYNC();
}

// L1 + L2 of the sweep: fills list[] with the packed tiles that may hold a hit and returns their count (per env).
// cv / cull may be the exact cull vectors of the state, or -- speculative pre-cull of the NEXT state, wave-specialised
// forward kernel -- vectors whose height is lowered by a verified bound on the body's motion (see k_rollout_fwd).
// cap: capacity of list[] per env (the cull wave's short list; the count returned may exceed it: the caller then drops the cull)
template <int SEGW>
PD_DEV int sweep_cull(const PdDevModel &m, const SweepTables &T, const BodyConst &c, float4 cv, const float4 *cull, int *list,
                      bool is_body, int seg, int l STAMP_ARGS, int cap = 0x7fffffff) {
  if (m.nc == 0) return 0;
  const bool surv = is_body && c.sphere.w >= 0.0f && !cull_above(cv, c.sphere);
  const unsigned long long wave_any = __ballot(surv);
  STAMP(8);
  if (wave_any == 0ull) return 0;
  const unsigned long long M = (wave_any >> (seg * SEGW)) & Seg<SEGW>::MASK;  // surviving bodies of my env
  const SegMask sm = seg_mask<SEGW>(seg);
  int nlist = 0;
  // ---- L2, small bodies: lane = entry of the static flat list of their tiles (held in registers)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (u * SEGW < m.n_small) {  // uniform
      const int e = c.small_e[u];
      bool pass = e >= 0 && ((M >> (e >> 16)) & 1ull) != 0ull;
      int pk = 0;
      if (pass) {
        pk = T.tpack[e & 0xffff];
        pass = !cull_above_box(cull[e >> 16], T.tlo[e & 0xffff], T.thi[e & 0xffff]);
      }
      int s = seg_slot(pass, sm, nlist);
      if (pass && s < cap) list[s] = pk;
    }
  }
  // ---- L2, big bodies (rarely survive L1): the segment's lanes share one body's tiles
  unsigned long long mb = M & m.big_bodies;
  while (__ballot(mb != 0ull) != 0ull) {
    int bb = 0, nt = 0, t_first = 0;
    if (mb != 0ull) {
      bb = __ffsll((long long)mb) - 1;
      mb &= mb - 1ull;
      int2 bt = T.btiles[bb];
      t_first = bt.x; nt = bt.y;
    }
    const float4 cb = cull[bb];
    for (int t0 = 0; __ballot(t0 < nt) != 0ull; t0 += SEGW) {
      int t = t0 + l;
      bool pass = t < nt;
      int pk = 0;
      if (pass) {
        pk = T.tpack[t_first + t];
        pass = !cull_above_box(cb, T.tlo[t_first + t], T.thi[t_first + t]);
      }
      int s = seg_slot(pass, sm, nlist);
      if (pass && s < cap) list[s] = pk;
    }
  }
  STAMP(9);
  WAVE_SYNC();
  return nlist;
}

// L3 + hit pass over list[0, nlist): exact point test, compaction, on_hit arithmetic.  log_n: see sweep_contacts.
template <int SEGW, int NV, int DSTRIDE, bool ATOMIC, typename F>
PD_DEV void sweep_points(const PdDevModel &m, const SweepTables &T, const float *rec, const float4 *cull, const int *list, int nlist,
                         int *hits, float *slot, float *dst, int seg, int l, int &log_n, F &&compute STAMP_ARGS) {
  const SegMask sm = seg_mask<SEGW>(seg);
  // ---- L3: point cull, PD_UNROLL tiles per iteration (their LDS reads overlap).  Hits are appended tile by tile, so the
  // hits of one body form one contiguous run [run_start, run_start + run_cnt) of the hit list, tracked by lane == body.
  int nh = 0, run_start = 0, run_cnt = 0;
  bool flushed = false;
  auto flush_all = [&]() {
    WAVE_SYNC();
    for (int j0 = 0; __ballot(j0 < nh) != 0ull; j0 += SEGW)
      sweep_flush_batch<SEGW, NV, DSTRIDE, ATOMIC>(T, rec, hits, slot, dst, j0, nh, run_start, run_cnt, l, m.nb, compute);
    nh = 0; run_cnt = 0;
  };
  for (int k0 = 0; __ballot(k0 < nlist) != 0ull; k0 += PD_UNROLL) {
    int e[PD_UNROLL], mi[PD_UNROLL];
    bool hit[PD_UNROLL];
#pragma unroll
    for (int u = 0; u < PD_UNROLL; ++u) e[u] = (k0 + u < nlist) ? list[k0 + u] : 0;  // count field 0 => no lane is valid
#pragma unroll
    for (int u = 0; u < PD_UNROLL; ++u) {
      int pt0 = e[u] & 0xffff, n = (e[u] >> 16) & 0xff, pb = (e[u] >> 24) & 0x3f;
      hit[u] = false; mi[u] = 0;
      if (l < n) {
        float4 P = T.pts[pt0 + l];
        float4 cb = cull[pb];
        mi[u] = T.pmat[pt0 + l];  // carried in the hit entry: the hit pass then needs no dependent material lookup
        hit[u] = cb.x + (cb.y * P.x + cb.z * P.y + cb.w * P.z) - P.w <= 1e-4f;
      }
    }
#pragma unroll
    for (int u = 0; u < PD_UNROLL; ++u) {
      const int before = nh;
      int s = seg_slot(hit[u], sm, nh);
      if (hit[u]) hits[s] = ((e[u] & 0xffff) + l) | (mi[u] << 16) | (e[u] & 0x3f000000);
      if (l == ((e[u] >> 24) & 0x3f) && nh > before) {
        if (run_cnt == 0) run_start = before;
        run_cnt += nh - before;
      }
    }
    if (__ballot(nh > (PD_HIT_CAP_TILES - PD_UNROLL) * SEGW) != 0ull) { flush_all(); flushed = true; }  // rare
  }
  STAMP(10);
  log_n = (!flushed && nh < PD_HITLOG) ? nh : -1;
  flush_all();
  STAMP(11);
}

// L3 alone, for the speculative sweep of the wave-specialised forward kernel: candidates of list[0, nlist) whose height
// under the (lowered) vectors `cull` passes the point test are compacted into hits[]; nh = their count.  Nothing is
// evaluated here -- the state the contacts belong to does not exist yet.  Returns false when some env's candidates do
// not fit hits[] (the caller then redoes the sweep the exact way).
template <int SEGW>
PD_DEV bool sweep_l3_spec(const SweepTables &T, const float4 *cull, const int *list, int nlist, int *hits, int seg, int l, int &nh) {
  const SegMask sm = seg_mask<SEGW>(seg);
  nh = 0;
  for (int k0 = 0; __ballot(k0 < nlist) != 0ull; k0 += PD_UNROLL) {
    if (__ballot(nh > (PD_HIT_CAP_TILES - PD_UNROLL) * SEGW) != 0ull) return false;
    int e[PD_UNROLL], mi[PD_UNROLL];
    bool hit[PD_UNROLL];
#pragma unroll
    for (int u = 0; u < PD_UNROLL; ++u) e[u] = (k0 + u < nlist) ? list[k0 + u] : 0;
#pragma unroll
    for (int u = 0; u < PD_UNROLL; ++u) {
      int pt0 = e[u] & 0xffff, n = (e[u] >> 16) & 0xff, pb = (e[u] >> 24) & 0x3f;
      hit[u] = false; mi[u] = 0;
      if (l < n) {
        float4 P = T.pts[pt0 + l];
        float4 cb = cull[pb];
        mi[u] = T.pmat[pt0 + l];
        hit[u] = cb.x + (cb.y * P.x + cb.z * P.y + cb.w * P.z) - P.w <= 1e-4f;
      }
    }
#pragma unroll
    for (int u = 0; u < PD_UNROLL; ++u) {
      int s = seg_slot(hit[u], sm, nh);
      if (hit[u]) hits[s] = ((e[u] & 0xffff) + l) | (mi[u] << 16) | (e[u] & 0x3f000000);
    }
  }
  WAVE_SYNC();
  return true;
}

// cv = this lane's own cull vector (registers), cull = the segment's cull vectors in LDS.
// dst: per-body accumulators [nb][DSTRIDE] (zeroed by their owner before the sweep); slot: SEGW*NV floats of scratch.
// replay_cnt == PD_NO_REPLAY : full sweep; log_n returns what to log (write_hit_log, done by the caller off the critical path)
// replay_cnt >= 0                            : adjoint sweep, the first replay_cnt entries of log ARE the hit list
#define PD_NO_REPLAY (-2)
template <int SEGW>
PD_DEV void write_hit_log(int *log, const int *hits, int log_n, bool env_ok, int l) {
  if (!env_ok) return;
  if (l == 0) log[0] = log_n;
  for (int j = l; j < log_n; j += SEGW) log[1 + j] = hits[j];
}
template <int SEGW, int NV, int DSTRIDE, bool ATOMIC, typename F>
PD_DEV void sweep_contacts(const PdDevModel &m, const SweepTables &T, const BodyConst &c, float4 cv, const float *rec,
                           const float4 *cull, int *list, int *hits, float *slot, float *dst, bool is_body, bool env_ok, int seg,
                           int l, int *log, int replay_cnt, int &log_n, F &&compute STAMP_ARGS, bool have_pre = false, int pre_e = 0) {
  if (replay_cnt >= 0) {  // wave-uniform: every env of this wave has a usable log entry
    const int nh_r = replay_cnt;
    if (ATOMIC) {
      // sums by ds_add_f32 in hit order: no per-hit slots, no run bounds, and an entry the caller already holds goes straight
      // from its register into the hit arithmetic (round 3: the detour through hits[] / rs / re cost the compound robots'
      // integrate wave two wave syncs and ~20 LDS instructions per step)
      for (int j = l; __ballot(j < nh_r) != 0ull; j += SEGW) {
        if (j < nh_r) {
          const int e = have_pre ? pre_e : log[1 + j];
          const int pt = e & 0xffff, pb = (e >> 24) & 0x3f;
          float out[NV];
          compute(rec + pb * PD_REC, T.pts[pt], T.mats[(e >> 16) & 0xff], out);
#pragma unroll
          for (int i = 0; i < NV; ++i) atomicAdd(dst + pb * DSTRIDE + i, out[i]);
        }
      }
      STAMP(11);
      return;
    }
    if (have_pre) {  // the caller fetched this lane's entry ahead of time (a list has at most PD_HITLOG - 1 <= SEGW entries then)
      if (l < nh_r) hits[l] = pre_e;
    } else {
      for (int j = l; j < nh_r; j += SEGW) hits[j] = log[1 + j];
    }
    int *rs = list, *re = list + m.nb;  // run bounds per body (the tile list is not used in a replay)
    if (l < m.nb) { rs[l] = 0; re[l] = 0; }
    WAVE_SYNC();
    for (int j = l; j < nh_r; j += SEGW) {
      const int pb = (hits[j] >> 24) & 0x3f;
      const int prev = j > 0 ? (hits[j - 1] >> 24) & 0x3f : -1, next = j + 1 < nh_r ? (hits[j + 1] >> 24) & 0x3f : -1;
      if (prev != pb) rs[pb] = j;
      if (next != pb) re[pb] = j + 1;
    }
    WAVE_SYNC();
    int rstart = 0, rcnt = 0;
    if (l < m.nb) { rstart = rs[l]; rcnt = re[l] - rstart; }
    STAMP(10);
    for (int j0 = 0; __ballot(j0 < nh_r) != 0ull; j0 += SEGW)
      sweep_flush_batch<SEGW, NV, DSTRIDE, ATOMIC>(T, rec, hits, slot, dst, j0, nh_r, rstart, rcnt, l, m.nb, compute);
    STAMP(11);
    return;
  }
  log_n = 0;  // what the forward caller should log for this env: >= 0 hit count (entries are in hits[]), -1 = did not fit
  const int nlist = sweep_cull<SEGW>(m, T, c, cv, cull, list, is_body, seg, l STAMP_PASS);
  sweep_points<SEGW, NV, DSTRIDE, ATOMIC>(m, T, rec, cull, list, nlist, hits, slot, dst, seg, l, log_n, compute STAMP_PASS);
}

// Speculative contact cull (k_rollout_fwd): the height any contact candidate of the body is allowed to lose over the
// next PD_SPEC_K steps -- per step 1.5 x what its present velocity would give plus 0.2 mm.  The value only steers how
// often the exact sweep has to be redone, never the result.
#ifndef PD_SPEC_K
#define PD_SPEC_K 4  // steps served by one speculative cull
#endif
// allowed sinking per step = m.spec_safety x what the present velocity gives + m.spec_slack metres: per MODEL since round 6 (pd_host.hip).
// Laikago (3 838 mesh points): 1.25 x + 0.1 mm -- every candidate costs a lane of the hit pass (round 3's sweep: 1.5 / 2e-4 before).  Box
// robots (human, quad: 8 points per body): 3 x + 1 mm -- their candidates are few whatever the margin, and the tight one made the exact
// sweep run again in 10 % of human's wave-steps (quad 2048 forward 0.219 -> 0.186 ms, human 1024 0.179 -> 0.175; 4 x the same, 6 x worse).
// (the revolute-only instantiations -- Laikago: mesh contacts -- keep the tight pair as IMMEDIATES: as kernel arguments the two values cost
// the headline forward kernel two scalar registers in a loop that spills them, 0.202 -> 0.205 ms; a revolute-only box robot gets the tight
// margin then: more exact sweeps, the same results)
#define PD_SPEC_SAFETY_TIGHT 1.25f
#define PD_SPEC_SLACK_TIGHT 1.0e-4f
// STEPS: how many steps the margin must hold for (PD_SPEC_K; one more when a cull wave delivers the candidates a step later)
template <int JT, int STEPS = PD_SPEC_K>
PD_DEV float sink_margin(const PdDevModel &m, const BodyConst &c, const BodyState &s, float dt) {
  const float safety = JT == PD_JT_REVOLUTE ? PD_SPEC_SAFETY_TIGHT : m.spec_safety, slack = JT == PD_JT_REVOLUTE ? PD_SPEC_SLACK_TIGHT : m.spec_slack;
  return (float)STEPS * (safety * dt * (fabsf(s.v.y) + (fabsf(s.w.x) + fabsf(s.w.y) + fabsf(s.w.z)) * c.reach) + slack);
}

// Copies the contact tables into LDS (once per workgroup) and returns the per-env scratch base.  COPY = false leaves
// the tables in global memory (wave-specialised adjoint: it replays the forward's hit log and fetches the few points it
// needs a step ahead, so the ~75 KB of LDS go to the joint hand-over records instead).
template <bool COPY>
PD_DEV float *lds_setup(const PdDevModel &m, unsigned char *smem, SweepTables &T, int env_slot, int env_floats) {  // all threads of the workgroup
  const int NT = blockDim.x;
  if (!COPY) {
    T.pts = m.pts; T.tlo = m.tile_lo; T.thi = m.tile_hi; T.mats = m.materials; T.tpack = m.tile_pack; T.btiles = m.body_tiles; T.pmat = m.pt_mat;
    return (float *)smem + (size_t)env_slot * env_floats;
  }
  const int nc4 = m.nc > 0 ? m.nc : 1, nt4 = m.ntiles > 0 ? m.ntiles : 1, nm4 = m.nmat > 0 ? m.nmat : 1;
  const int nbp = (m.nb + 1) & ~1, ncb = (nc4 + 15) & ~15, ntp = (nt4 + 3) & ~3;
  float4 *pts = (float4 *)smem;
  float4 *tlo = pts + nc4;
  float4 *thi = tlo + nt4;
  float4 *mat = thi + nt4;
  int *tpk = (int *)(mat + nm4);
  int2 *btl = (int2 *)(tpk + ntp);
  unsigned char *pmt = (unsigned char *)(btl + nbp);
  for (int i = threadIdx.x; i < m.nc; i += NT) pts[i] = m.pts[i];
  for (int i = threadIdx.x; i < m.ntiles; i += NT) { tlo[i] = m.tile_lo[i]; thi[i] = m.tile_hi[i]; tpk[i] = m.tile_pack[i]; }
  for (int i = threadIdx.x; i < m.nmat; i += NT) mat[i] = m.materials[i];
  for (int i = threadIdx.x; i < m.nb; i += NT) btl[i] = m.body_tiles[i];
  for (int i = threadIdx.x; i < ncb / 4; i += NT) ((unsigned int *)pmt)[i] = ((const unsigned int *)m.pt_mat)[i];
  __syncthreads();
  T.pts = pts; T.tlo = tlo; T.thi = thi; T.mats = mat; T.tpack = tpk; T.btiles = btl; T.pmat = pmt;
  return (float *)(pmt + ncb) + (size_t)env_slot * env_floats;  // env_floats is a multiple of 4: 16-B aligned
}

// =============================================================================================
// SPLIT = wave specialisation (8 waves, contact waves beside body waves).  Kernels whose body wave needs more than
// 256 VGPRs (compound joints) run unsplit: 4 waves per workgroup, one per SIMD, sweeps inline.
// (second launch-bound argument = minimum waves per SIMD: the unsplit kernel's 4-wave workgroups must stay within 256 VGPRs so
// that two of them are resident per CU)
// A body wave and its contact wave share a SIMD (a workgroup's waves are dealt to the four SIMDs cyclically), and a SIMD issues one
// fp32 vector instruction per ~4.2 cycles whatever the waves -- so WHO gets the issue slots matters.  The
// hardware prefers the older wave (the body wave); the contact wave raises its priority (s_setprio) for the windows in which the
// pair waits for IT: forward, hand-over A .. B (the hit pass the body wave then waits for) and on through the speculative cull
// that the next hit pass needs (0.226 -> 0.214 ms at 4096 envs, 0.442 -> 0.418 at 8192); adjoint, A .. B (0.293 -> 0.282,
// 0.577 -> 0.562).  Raised while it recomputes rev_forward, or for the whole kernel (round 2), the adjoint is slower (0.298).
// Serving the body wave of the NEXT group instead (contact wave beside an unrelated body wave) was 1 % faster in the forward pass
// without priorities, is 1 % slower with them, and is 11 % slower in the adjoint: pairs stay on one SIMD.
#define PD_PRIO_CRITICAL 3
// LOSS (pd_rollout_forward_traj_loss): at every frame state the body wave also evaluates se3_loss of the
// pose it is about to store against the frame's target pose (dp_model.py:777, dp_utils.py:113-138), stores both unscaled gradients
// and the mean over the env's bodies -- the [bs][F] table reduce_loss works on -- so that no pose makes the round trip through a loss
// launch and torch before the adjoint can be seeded.  A separate instantiation: the plain rollout kernel is untouched.
// QUAD (small batches of revolute-only robots; pd_quad.h): the body wave gives every body FOUR lanes (component-parallel arithmetic),
// one env per wave, SEGW = 64 so that the contact wave and every LDS table are those of the 64-lane mapping.  Same records, same
// hand-overs, same trajectory layout: the adjoint kernels run on what it saves.
#if defined(PD_KNOCK) && (PD_KNOCK & 16)
#define pair_wait(f, v) pair_wait_knock(f, v, knock_role)
#endif
// RUNSUM (with CULLW, lane per body): the hit pass sums per body in registers -- launched when four env groups fill the workgroup (see there)
template <int SEGW, int JT, bool SPLIT, bool LOSS = false, bool QUAD = false, bool CULLW = false, bool RUNSUM = false>
__global__ __launch_bounds__(CULLW ? PD_BLOCK3 : (SPLIT ? PD_BLOCK : PD_FK_BLOCK), CULLW ? 3 : 2) void k_rollout_fwd(PdDevModel m, RolloutArgs a) {
  static_assert(!RUNSUM || (CULLW && !QUAD), "run sums in the hit pass: the lane-per-body kernels with the cull wave");
  // TRAJC: the contact wave stores planes 0-2 of the trajectory out of the staged records (round 3 measured this a loss, when that wave's idle
  // window held the cull; with the cull on its own wave: Laikago 2 048 / 4 096 / 8 192 envs forward 0.170 / 0.188 / 0.377 -> 0.166 / 0.186 / 0.371 ms)
  constexpr bool TRAJC = CULLW && !QUAD;
  static_assert(!QUAD || (SEGW == 64 && SPLIT && JT == PD_JT_REVOLUTE), "quad-lane body wave: one env per wave, revolute-only plain models");
  static_assert(!CULLW || (SPLIT && JT == PD_JT_REVOLUTE), "cull wave: wave-specialised kernels of revolute-only robots (<= 168 VGPRs: three waves per SIMD)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int EPW = Seg<SEGW>::EPW;
  constexpr int ND = (JT & PD_JT_COMPOUND) ? 3 : 1;
  // env groups (= body waves) per workgroup: chosen by the host per launch (1 .. PD_BWAVES) so that small batches spread
  // over all compute units instead of filling a few
  const int bw = (int)blockDim.x / (CULLW ? 192 : (SPLIT ? 128 : 64));
  const int role = SPLIT ? (int)(threadIdx.x >> 6) / bw : 0;  // wave-uniform: 0 body wave, 1 contact wave, 2 (CULLW) cull wave
  const bool contact_wave = role != 0;
  [[maybe_unused]] const int knock_role = role;
  const int lane = threadIdx.x & 63, wave = (int)(threadIdx.x >> 6) % bw;
  const int seg = lane / SEGW, l = lane % SEGW;
  const int env = (blockIdx.x * bw + wave) * EPW + seg;
  const bool env_ok = env < a.bs;
  const int ec = env_ok ? env : 0;  // clamped env for safe addressing
  const bool is_body = env_ok && l < m.nb;
  const int b = l < m.nb ? l : m.nb - 1;
  const int nb = m.nb, N = a.bs * nb;

  SweepTables tabs;
  float *scratch = lds_setup<true>(m, smem, tabs, wave * EPW + seg, m.env_lds_floats);
  float4 *cull = (float4 *)scratch;
  // pcon: nb + 1 records, the last one stays zero and stands in for "no child" (the gather then needs no predicates)
  float *rec = scratch + 4 * nb, *facc = rec + nb * PD_REC, *pcon = facc + nb * PD_W6;
  // speculative cull vectors of the two latest states (by step parity) + the env's "speculation failed" flag; they live
  // in the room the adjoint kernel's wider per-body slots leave in the shared per-env size
  const int spec_off = ((4 + PD_REC + 2 * PD_W6) * nb + PD_W6 + 3) & ~3;  // 16-byte aligned like cull
  float4 *spec = (float4 *)(scratch + spec_off);
  int *spec_bad = (int *)(scratch + spec_off + 8 * nb);  // 4 words: [0] unused, [1..2] of the wave's first env = pair signals
  int *sig = (int *)(scratch - (size_t)seg * m.env_lds_floats + spec_off + 8 * nb) + 1;  // pair signals: words 1, 2 (, 3) after the first env's flag
  int *list = spec_bad + 4, *hits = list + m.list_cap;
  float *slot = (float *)(hits + PD_HIT_CAP_TILES * SEGW);
  // CULLW: the cull wave's own candidate list and (short) tile list, in the room the adjoint kernel's wider per-hit slots leave in the
  // shared per-env size (the forward pass sums six floats per hit, the adjoint thirteen); spec_bad[0] of an env = its candidate count
  int *hits2 = (int *)(slot + 6 * SEGW), *list2 = hits2 + PD_HIT_CAP_TILES * SEGW;
  const int cap2 = m.env_lds_floats - (int)((float *)list2 - scratch);
  if (SPLIT) {
    if (lane == 0) { sig[0] = 0; sig[1] = 0; sig[2] = 0; }
    __syncthreads();
  }
  PD_KNOCK_EXIT(true, knock_role);
  if constexpr (QUAD) {
    if (!env_ok) return;  // one env per wave pair: a pair past the batch has nothing to do (no workgroup barrier follows)
  }

  BodyConst c = load_body_const(m, b, ec);
#pragma unroll
  for (int u = 0; u < 4; ++u) c.small_e[u] = m.small_tiles[u * 64 + (l < 64 ? l : 0)];
  auto contact_hit = [&](const float *r, float4 P, float4 mat, float *out) {  // body_f -= (t, f)   (:179)
    ContactOut o;
    const bool touching = contact_point_fwd(r, cull[(int)(r - rec) / PD_REC], P, mat, o);
    out[0] = touching ? -o.t.x : 0.f; out[1] = touching ? -o.t.y : 0.f; out[2] = touching ? -o.t.z : 0.f;
    out[3] = touching ? -o.f.x : 0.f; out[4] = touching ? -o.f.y : 0.f; out[5] = touching ? -o.f.z : 0.f;
    return touching;
  };
  if constexpr (CULLW) {
    if (role == 2) {
      // ---- cull wave (round 6).  The speculative cull of an epoch costs 3 600-4 200 cycles; on the contact wave it began after hand-over
      // B of the epoch's first step and overran the next hand-over A by ~2 400 cycles (Laikago 4096: the hit pass of the epoch's second
      // step started that late; with the culls taken out -- a timing build, wrong results -- the forward pass took 0.162-0.172 ms against
      // 0.203, the quad-lane one 0.115 against 0.135).  Here it has a wave of its own and TWO steps: it culls with the vectors of state
      // e K (staged before hand-over A of that step, margins for PD_SPEC_K + 1 steps) and the contact wave takes the candidates over
      // after hand-over B of step e K + 1, for the hit passes of steps e K + 2 .. e K + K + 1.  One generation of hits2 / list2: the next
      // cull starts behind hand-over A of step (e + 1) K, which the body wave signals after B of step e K + 2 at the earliest.
      STAMP_DECL;
      for (int e = 0; e * PD_SPEC_K + 2 < a.nsteps; ++e) {
        pair_wait(sig, e * PD_SPEC_K + 1);
        STAMP(7);
        const float4 *sp = spec + (e & 1) * nb;
        float4 cv = make_float4(0.f, 0.f, 1.f, 0.f);
        if (is_body) cv = sp[b];
        const int nlist = sweep_cull<SEGW>(m, tabs, c, cv, sp, list2, is_body, seg, l STAMP_PASS, cap2);
        int nh2 = 0;
        bool ok = __ballot(nlist > cap2) == 0ull;   // (a tile list that did not fit: no candidates this epoch, the contact wave sweeps exactly)
        if (ok) ok = sweep_l3_spec<SEGW>(tabs, sp, list2, nlist, hits2, seg, l, nh2);
        if (l == 0) spec_bad[0] = nh2;
        STAMP(10);
        STAMP_COUNT(13, ok ? 0 : 1);
        STAMP_COUNT(14, __shfl(nlist, 0));
        pair_signal(sig + 2, (e + 1) | (ok ? 0 : PD_SIG_FLAG));  // C: the candidates of epoch e
      }
      STAMP_FLUSH(a);
      return;
    }
  }
  if (SPLIT && contact_wave) {
    // ---- contact wave: eval_body_contacts for the partner body wave's envs, between hand-overs A and B of each step.
    // The cull (L1-L3) is SPECULATED, once per epoch of PD_SPEC_K steps, in the wait for the body wave's integration:
    // after hand-over B of an epoch's first step s this wave culls with the vectors of state s lowered by margin_b(s), a
    // guess of how far body b can sink over the epoch.  The body wave checks the guess against the motion it then
    // integrates (integrate_fwd: sink_rate, summed over the epoch) and raises the env's flag when a body sank further;
    // a raised flag makes this wave redo the exact sweep.  The candidates are a superset, in the same order, of what
    // the exact sweep finds in any state of the epoch; contact_hit applies the reference's exact test to each, so the
    // wrench sums are bit-identical to the unspeculated sweep.
    STAMP_DECL;
    const SegMask sm = seg_mask<SEGW>(seg);
    int nh = 0;         // speculated candidates of my env, hits[0, nh)
    bool have = false;  // wave-uniform: the candidates are there and nothing overwrote them
    // When no env of the wave has more candidates than the segment has lanes (the usual case) lane j keeps candidate j --
    // entry, point, material -- in registers for the whole epoch: the per-step hit pass then starts with the record read.
    bool lane_owns = false;  // wave-uniform
    bool two = false;        // wave-uniform: some env has more candidates than lanes (<= 2 SEGW): lane j also keeps candidate SEGW + j
    int c_e = 0, c_e2 = 0;
    float4 c_P = make_float4(0.f, 0.f, 0.f, 0.f), c_M = c_P, c_P2 = c_P, c_M2 = c_P;
    // (The cull in TWO pieces -- tile levels behind B(s), point level behind B(s + 1), candidates serving steps s + 2 .. s + K + 1 -- was
    // built again in round 6 on the branch-free body wave: 0.214-0.216 ms against 0.202 at 4096 envs.  At normal priority beside the
    // body wave's busy phases the point level takes 4 000 cycles instead of 2 200 and still runs into the next hand-over A, the margins
    // for K + 1 steps make 8.0 candidates per env-step instead of 7.0: EXPERIMENTS.md, profiles/r06_fwd_stamps.txt.)
    for (int step = 0; step < a.nsteps; ++step) {
      int *lg = a.hitlog + ((size_t)step * a.bs + ec) * PD_HITLOG;
      // A: records + cull vectors of this step are staged, wrench accumulators are zero; bit 30: a body of one of my envs
      // outran its margin
      const int sigA = pair_wait(sig, step + 1);
      __builtin_amdgcn_s_setprio(PD_PRIO_CRITICAL);  // the body wave will wait for this hit pass
      STAMP(7);
      // TRAJC: this step's state out of the staged record of this lane's body (the body wave restages it only behind hand-over B): this wave,
      // idle 58 % of the step since the cull has a wave of its own, stores planes 0-2 of the trajectory behind B
      [[maybe_unused]] v3 ts_p = V3(0, 0, 0), ts_w = ts_p, ts_v = ts_p;
      [[maybe_unused]] qt ts_q = Q4(0, 0, 0, 1);
      if constexpr (TRAJC) {
        const float *r = rec + b * PD_REC;
        ts_p = ld3(r); ts_q = ld4(r + 3); ts_w = ld3(r + 7); ts_v = ld3(r + 10);
      }
#ifdef PD_ALWAYS_REDO   // (checking build: the exact sweep every step -- the speculated passes must give the same bits)
      const bool redo = true;
#else
      const bool redo = !have || (sigA & PD_SIG_FLAG) != 0;  // wave-uniform
#endif
      STAMP_COUNT(13, redo ? 1 : 0);
      STAMP_COUNT(14, __shfl(nh, 0));
      int log_n = 0;
      bool touching = false, touching2 = false;  // lane_owns path: do my candidates touch (logged after hand-over B)
      if (redo) {
        float4 cv = make_float4(0.f, 0.f, 1.f, 0.f);
        if (is_body) cv = cull[b];
        const int nlist = sweep_cull<SEGW>(m, tabs, c, cv, cull, list, is_body, seg, l STAMP_PASS);
        sweep_points<SEGW, 6, PD_W6, true>(m, tabs, rec, cull, list, nlist, hits, slot, facc, seg, l, log_n, contact_hit STAMP_PASS);
        have = false;  // hits[] now holds this step's exact hits
      } else if (lane_owns) {
        // evaluate the candidates on the state that now exists: contact_hit applies the reference's exact test
        // Per-body sums of the candidates' wrenches: ds_add_f32 per touching lane -- or (RUNSUM: the instantiation launched when four env groups
        // fill the workgroup, >= 4 x CUs groups: every SIMD holds a body, a contact and a cull wave and the LDS atomics of 16 envs queue up) DPP
        // run sums in the registers and one plain store per body: the same left-to-right order, the same bits (Laikago 4096 forward 0.198 ->
        // 0.189 ms, 8192 0.388 -> 0.372; at 2 048 envs the run sums lose 3 %, in the quad-lane kernel -- one env's candidates in one 64-lane
        // segment, runs of up to seven -- 37 %)
        if (RUNSUM && !two) {
          float out[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          const int pbr = l < nh ? (c_e >> 24) & 0x3f : -2;
          if (l < nh) touching = contact_hit(rec + pbr * PD_REC, c_P, c_M, out);
          bool last;
          seg_run_sum<6>(out, pbr, l, nh, last);
          if (last) {
#pragma unroll
            for (int i = 0; i < 6; ++i) facc[pbr * PD_W6 + i] = out[i];
          }
        } else if (l < nh) {
          float out[6];
          touching = contact_hit(rec + ((c_e >> 24) & 0x3f) * PD_REC, c_P, c_M, out);
          if (touching) {
#pragma unroll
            for (int i = 0; i < 6; ++i) atomicAdd(facc + ((c_e >> 24) & 0x3f) * PD_W6 + i, out[i]);
          }
        }
        if (two) {  // (a body's candidates are contiguous in the list and this pass comes second: the sums keep the list's order)
          if (SEGW + l < nh) {
            float out[6];
            touching2 = contact_hit(rec + ((c_e2 >> 24) & 0x3f) * PD_REC, c_P2, c_M2, out);
            if (touching2) {
#pragma unroll
              for (int i = 0; i < 6; ++i) atomicAdd(facc + ((c_e2 >> 24) & 0x3f) * PD_W6 + i, out[i]);
            }
          }
        }
        STAMP(11);
      } else {
        // more candidates than lanes somewhere in the wave: batches out of the LDS list; the ones that touch go to the
        // adjoint's log straight away
        // (per-body sums in registers, seg_run_sum, were measured here too: slower than the 6 ds_add_f32 of the few
        // candidates that touch -- the chain runs over ALL candidates)
        for (int j0 = 0; __ballot(j0 < nh) != 0ull; j0 += SEGW) {
          const int j = j0 + l;
          bool tch = false;
          int e = 0;
          if (j < nh) {
            e = hits[j];
            float out[6];
            tch = contact_hit(rec + ((e >> 24) & 0x3f) * PD_REC, tabs.pts[e & 0xffff], tabs.mats[(e >> 16) & 0xff], out);
            if (tch) {
#pragma unroll
              for (int i = 0; i < 6; ++i) atomicAdd(facc + ((e >> 24) & 0x3f) * PD_W6 + i, out[i]);
            }
          }
          const int s = seg_slot(tch, sm, log_n);
          if (tch && env_ok && s < PD_HITLOG - 1) lg[1 + s] = e;
        }
        if (l == 0 && env_ok) lg[0] = log_n < PD_HITLOG ? log_n : -1;
        STAMP(11);
      }
      STAMP(12);
      pair_signal(sig + 1, step + 1);  // B: contact wrenches are complete
      if constexpr (TRAJC) {
        if (is_body) {
          float *tj = a.ws + (size_t)step * (PD_TRAJ_G * 4) * N;
          const unsigned boff16c = (unsigned)((size_t)ec * nb + b) * 16u;
          stg4(tj, boff16c, make_float4(ts_q.x, ts_q.y, ts_q.z, ts_q.w));
          stg4(tj + (size_t)4 * N, boff16c, make_float4(ts_w.x, ts_w.y, ts_w.z, ts_v.x));
          stg4(tj + (size_t)8 * N, boff16c, make_float4(ts_p.x, ts_p.y, ts_p.z, ts_v.y));
        }
      }
      const bool cull_now = !CULLW && step % PD_SPEC_K == 0 && step + 1 < a.nsteps;  // state `step` opened an epoch: cull for the steps it serves
      if (!cull_now) __builtin_amdgcn_s_setprio(0);  // (the cull stays urgent: the next hit pass needs its candidates)
      // the adjoint's log is written off the critical path
      if (redo) {
        write_hit_log<SEGW>(lg, hits, log_n, env_ok, l);
      } else if (lane_owns) {
        const int s = seg_slot(touching, sm, log_n);
        if (touching && env_ok && s < PD_HITLOG - 1) lg[1 + s] = c_e;
        if (two) {
          const int s2 = seg_slot(touching2, sm, log_n);
          if (touching2 && env_ok && s2 < PD_HITLOG - 1) lg[1 + s2] = c_e2;
        }
        if (l == 0 && env_ok) lg[0] = log_n < PD_HITLOG ? log_n : -1;
      }
      if (redo) lane_owns = false;
      if (CULLW && step % PD_SPEC_K == 1 && step + 1 < a.nsteps) {
        // the cull wave's candidates of the epoch that opened with state step - 1: they serve the hit passes of steps step + 1 .. step + K
        WAVE_SYNC();  // the log is read out of hits[] before candidates may overwrite it
        const int sigC = pair_wait(sig + 2, step / PD_SPEC_K + 1);
        have = (sigC & PD_SIG_FLAG) == 0;
        nh = have ? spec_bad[0] : 0;
        lane_owns = have && __ballot(nh > 2 * SEGW) == 0ull;
        two = lane_owns && __ballot(nh > SEGW) != 0ull;
        if (lane_owns) {
          c_e = l < nh ? hits2[l] : 0;
          c_P = tabs.pts[c_e & 0xffff]; c_M = tabs.mats[(c_e >> 16) & 0xff];
          if (two) {
            c_e2 = SEGW + l < nh ? hits2[SEGW + l] : 0;
            c_P2 = tabs.pts[c_e2 & 0xffff]; c_M2 = tabs.mats[(c_e2 >> 16) & 0xff];
          }
        } else if (have) {  // more candidates than two per lane somewhere: they stay a list, in this wave's own buffer
          for (int j = l; j < nh; j += SEGW) hits[j] = hits2[j];
          WAVE_SYNC();
        }
      }
      if (cull_now) {
        WAVE_SYNC();  // the log is read out of hits[] before the candidates overwrite it
        const float4 *sp = spec + ((step / PD_SPEC_K) & 1) * nb;
        float4 cv = make_float4(0.f, 0.f, 1.f, 0.f);
        if (is_body) cv = sp[b];
        const int nlist = sweep_cull<SEGW>(m, tabs, c, cv, sp, list, is_body, seg, l STAMP_PASS);
        have = sweep_l3_spec<SEGW>(tabs, sp, list, nlist, hits, seg, l, nh);
        lane_owns = have && __ballot(nh > 2 * SEGW) == 0ull;
        two = lane_owns && __ballot(nh > SEGW) != 0ull;
        if (lane_owns) {
          c_e = l < nh ? hits[l] : 0;
          c_P = tabs.pts[c_e & 0xffff]; c_M = tabs.mats[(c_e >> 16) & 0xff];
          if (two) {
            c_e2 = SEGW + l < nh ? hits[SEGW + l] : 0;
            c_P2 = tabs.pts[c_e2 & 0xffff]; c_M2 = tabs.mats[(c_e2 >> 16) & 0xff];
          }
        }
        STAMP(10);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    STAMP_FLUSH(a);
    return;
  }
  if constexpr (QUAD) {
    // ================= body wave, four lanes per body: lane 4 qb + qc holds component qc of body qb of this wave's ONE env
    // Round 6, as in the lane-per-body kernels (CLONE): no divergent region in the step loop.  A wave whose env lies past the batch has
    // left (below the hand-over words' initialisation: the contact wave too); an idle quad (qb >= nb) is a CLONE of the env's last body and
    // writes what that body's quad writes; lane 3 of a quad -- it holds the w of quaternions and nothing of vectors -- sends its "vector"
    // stores to the PAD float of its body's record / wrench slot instead of sitting them out, and reads component 2 where the others read
    // their own (a select zeroes it).  The loop had 27 branches and 16 exec-mask regions per step.
    const int qb = lane >> 2, qc = lane & 3, bb = qb < nb ? qb : nb - 1, qv = qc < 3 ? qc : 2;
    const bool qbody = true;
    const QLane k = q_lane(qc);
    const size_t qidx = (size_t)ec * nb + bb;
    QBody B;
    {
      const BodyConst cb = load_body_const(m, bb, ec);
      B.type = cb.type; B.pidx = cb.pidx; B.qdstart = cb.qdstart;
      B.joint = qbody && cb.type == PD_JOINT_REVOLUTE;
      B.com = q_pick(k, cb.com); B.com0 = cb.com.x; B.com1 = cb.com.y; B.com2 = cb.com.z;
      B.axis = q_pick(k, cb.axis); B.ax0 = cb.axis.x; B.ax1 = cb.axis.y; B.ax2 = cb.axis.z; B.alen = cb.alen;
      B.p_pj = q_pick(k, cb.p_pj); B.q_pj = q_pick(k, cb.q_pj); B.pj = q_perm(k, B.q_pj);
      B.g = q_pick(k, V3(m.gx, m.gy, m.gz));
      B.reach = cb.reach; B.sphere_w = cb.sphere.w; B.lim = cb.lim[0];
      B.inv_m = a.inv_mass[qidx];
      const bool on = cb.type == PD_JOINT_REVOLUTE;
      B.ke = on ? a.target_ke[(size_t)ec * m.nqd + cb.qdstart] : 0.f;
      B.kd = on ? a.target_kd[(size_t)ec * m.nqd + cb.qdstart] : 0.f;
      const float *Ib = a.inertia + qidx * 9 + qv * 3, *Jb = a.inv_inertia + qidx * 9 + qv * 3;
      B.I.a = k.isv ? Ib[0] : 0.f; B.I.b = k.isv ? Ib[1] : 0.f; B.I.c = k.isv ? Ib[2] : 0.f;
      B.invI.a = k.isv ? Jb[0] : 0.f; B.invI.b = k.isv ? Jb[1] : 0.f; B.invI.c = k.isv ? Jb[2] : 0.f;
    }
    int qcz[4];  // first four children of body bb, the zero record for a missing one
    {
      const unsigned long long ch = m.children[bb];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const int cid = (int)((ch >> (8 * j)) & 0xffull); qcz[j] = (qbody && cid != 0xff) ? cid : nb; }
    }
    const unsigned long long q_children = m.children[bb];
    if (lane < nb) {
#pragma unroll
      for (int j = 0; j < 6; ++j) facc[lane * PD_W6 + j] = 0.f;
    }
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 6; ++j) pcon[nb * PD_W6 + j] = 0.f;
    }
    // ---- eval_fk (dp_model.py:1204) in the lane-per-body layout (lanes 0 .. nb-1), once; the records it stages are then read back
    // component-wise
    for (int d = 0; d <= m.max_depth; ++d) {
      if (is_body && c.depth == d) {
        BodyState s0 = fk_joint<JT>(c, a.q_init + (size_t)ec * m.nq + c.qstart, a.qd_init + (size_t)ec * m.nqd + c.qdstart, rec);
        float Rm0[9];
        rotm(s0.r, Rm0);
        stage_record(rec, cull, b, s0, mat_vec(Rm0, c.com), Rm0);
      }
      WAVE_SYNC();
    }
    QState s;
    {
      const float *r = rec + bb * PD_REC;
      s.p = k.isv ? r[qv] : 0.f; s.r = r[3 + qc]; s.w = k.isv ? r[7 + qv] : 0.f; s.v = k.isv ? r[10 + qv] : 0.f;
    }
    QM3 Rr, Rc;
    q_rotm(k, s.r, Rr, Rc);
    float rc = q_mvc(Rr, B.com0, B.com1, B.com2);
    auto q_margin = [&](const QState &x) {  // sink_margin (the speculative contact cull), all four lanes
      return (float)(CULLW ? PD_SPEC_K + 1 : PD_SPEC_K) * (PD_SPEC_SAFETY_TIGHT * a.dt * (Q_BC1(fabsf(x.v)) + q_sum3(fabsf(x.w)) * B.reach) + PD_SPEC_SLACK_TIGHT);  // (quad-lane: revolute-only)
    };
    // staging: the record fields are contiguous vectors, lane c writes component c of each; the cull vector (p_y, row 1 of rotm) is
    // lane 1's: its own p and its row
    auto q_stage = [&](const QState &x, float rcx, const QM3 &R, float lower, float4 *spec_dst) {
      {
        float *r = rec + bb * PD_REC;
        r[3 + qc] = x.r;
        r[k.isv ? qc : 16] = x.p; r[k.isv ? 7 + qc : 16] = x.w; r[k.isv ? 10 + qc : 16] = x.v; r[k.isv ? 13 + qc : 16] = rcx;  // (lane 3: the record's pad float)
        if (qc == 1) {
          cull[bb] = make_float4(x.p, R.a, R.b, R.c);
          if (spec_dst) spec_dst[bb] = make_float4(x.p - lower, R.a, R.b, R.c);
        }
      }
    };
    float margin = q_margin(s), sunk = 0.f;
    WAVE_SYNC();  // every lane has read the FK records
    q_stage(s, rc, Rr, margin, spec);  // epoch 0
    // (CULLW, see the lane-per-body form: the pair in use and the pair staged; nothing is in use before state 2)
    [[maybe_unused]] float margin_n = margin, sunk_n = 0.f;
    if (CULLW) margin = __builtin_inff();
    bool spec_failed = true;
    // controls one step ahead; the trajectory planes are float4 per body: lane c owns float c of each
    const unsigned boff_qd = (unsigned)((size_t)ec * m.nqd + B.qdstart) * 4u, boff_rf = (unsigned)(qidx * 6 + qv) * 4u;
    const unsigned boff_tj = (unsigned)(qidx * 4 + qc) * 4u;
    float n_tgt = 0.f, n_act = 0.f, n_rft = 0.f, n_rff = 0.f;
    int n_fr = -1;
    auto load_controls = [&](int step) {
      const int sc = __builtin_amdgcn_readfirstlane(step < a.nsteps ? step : a.nsteps - 1);
      n_fr = ld_uniform(a.frame_of_step, sc);
      const size_t o = (size_t)sc * a.bs * m.nqd;
      n_tgt = ldg(a.refs + o, boff_qd);   // (unconditional: the root reads the first of its own six dofs, its joint result is dropped)
      n_act = ldg(a.torques + o, boff_qd);
      const float *rf = a.res_f + (size_t)sc * N * 6;
      n_rft = ldg(rf, boff_rf); n_rff = ldg(rf + 3, boff_rf);
    };
    auto q_frame_out = [&](int cfr, const QState &x) {  // frame gather (dp_model.py:1231-1248)
      if (qbody) {
        float *o = a.wp_pos + ((size_t)cfr * N + qidx) * 7;
        o[3 + qc] = x.r;
        float *ov = a.wp_vel + ((size_t)cfr * N + qidx) * 6;
        if (k.isv) { o[qc] = x.p; ov[qc] = x.w; ov[3 + qc] = x.v; }
      }
    };
    auto q_frame_loss = [&](int cfr, float x_p, float x_r) {  // after the step loop, on the poses read back (see the lane-per-body form)
      if constexpr (LOSS) {
        // trajectory loss of this frame (see frame_loss of the lane-per-body form): the body's pose is gathered into every lane of
        // its quad, se3_loss runs redundantly in the four lanes, lane c stores components c and 3 + c of the two gradients
        const float pose[7] = {Q_BC0(x_p), Q_BC1(x_p), Q_BC2(x_p), Q_BC0(x_r), Q_BC1(x_r), Q_BC2(x_r), Q_BC3(x_r)};
        float lb = 0.f;
        const size_t ot = (((size_t)ec * a.nframes + cfr) * nb + bb) * 7;
        float tg[7], gp[7], gg[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) tg[j] = a.loss_target[ot + j];
        lb = pd_se3::se3_loss_eval<7>(pose, tg, a.loss_rot_ratio, true, gp, gg);
        const float gp_v = qc == 0 ? gp[0] : (qc == 1 ? gp[1] : gp[2]), gp_q = qc == 0 ? gp[3] : (qc == 1 ? gp[4] : (qc == 2 ? gp[5] : gp[6]));
        const float gg_v = qc == 0 ? gg[0] : (qc == 1 ? gg[1] : gg[2]), gg_q = qc == 0 ? gg[3] : (qc == 1 ? gg[4] : (qc == 2 ? gg[5] : gg[6]));
        if (qbody) {
          float *o = a.loss_seed_pos + ((size_t)cfr * N + qidx) * 7;
          o[3 + qc] = gp_q;
          if (k.isv) o[qc] = gp_v;
          if (a.loss_seed_gt) {
            a.loss_seed_gt[ot + 3 + qc] = gg_q;
            if (k.isv) a.loss_seed_gt[ot + qc] = gg_v;
          }
        }
        lb = (qb < nb && qc == 0) ? lb : 0.f;   // one lane per REAL body (not its clones) carries the body's loss into the mean over bodies (dp_model.py:777)
#pragma unroll
        for (int w = 32; w >= 1; w >>= 1) lb += __shfl_xor(lb, w, 64);
        if (lane == 0 && env_ok) {
          const size_t oe = (size_t)ec * a.nframes + cfr;
          a.loss_table[oe] = (a.loss_outseq && a.loss_outseq[oe]) ? 0.f : lb / (float)nb;
        }
      }
    };
    float o_p3 = 0.f, o_p4 = 0.f;  // planes 3 / 4 of the previous step (its total wrench and clamp mask), stored one step late
    if (a.nsteps > 0) load_controls(0);
    const float ake = m.attach_ke, akd = m.attach_kd;
    STAMP_DECL;
    for (int step = 0; step < a.nsteps; ++step) {
      pair_signal(sig, (step + 1) | (spec_failed ? PD_SIG_FLAG : 0));  // A: this step's records are staged
      STAMP(0);
      PD_WAIT_VMEM();
      const float tgt = n_tgt, act = n_act;
      float ft = k.isv ? n_rft : 0.f, ff = k.isv ? n_rff : 0.f;  // clear_forces + wp_add
      const int fr = n_fr;
      load_controls(step + 1);
      STAMP(1);
      // ---- eval_body_joints (while the contact wave sweeps)
      float wp_t, wc_t, jf_;
      {
        const float *pr = rec + B.pidx * PD_REC;
        const float l_pp = pr[qv], qp = pr[3 + qc], l_wp = pr[7 + qv], l_vp = pr[10 + qv], l_rc = pr[13 + qv];  // (unguarded reads, then selects)
        const float pp = k.isv ? l_pp : 0.f, w_p = k.isv ? l_wp : 0.f, v_p = k.isv ? l_vp : 0.f, rc_par = k.isv ? l_rc : 0.f;
        q_joint_fwd(k, B, s, Rr, rc, pp, qp, w_p, v_p, rc_par, tgt, act, ake, akd, wp_t, wc_t, jf_);
      }
      wp_t = B.joint ? wp_t : 0.f; wc_t = B.joint ? wc_t : 0.f; jf_ = B.joint ? jf_ : 0.f;
      pcon[bb * PD_W6 + (k.isv ? qc : 6)] = wp_t; pcon[bb * PD_W6 + (k.isv ? 3 + qc : 6)] = jf_;  // (lane 3: the slot's pad float)
      STAMP(2);
      WAVE_SYNC();
      float jt = -wc_t, jf = -jf_;  // joint wrench on this body: own joint first, then children in index order
      {
        float ct[4], cf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { ct[j] = pcon[qcz[j] * PD_W6 + qv]; cf[j] = pcon[qcz[j] * PD_W6 + 3 + qv]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { jt += ct[j]; jf += cf[j]; }
      }
      for (int j = 4; j < m.max_children; ++j) {
        const int cid = (int)((q_children >> (8 * j)) & 0xffull);
        if (cid != 0xff) { jt += pcon[cid * PD_W6 + qv]; jf += pcon[cid * PD_W6 + 3 + qv]; }
      }
      jt = k.isv ? jt : 0.f; jf = k.isv ? jf : 0.f;
      STAMP(6);
      // ---- trajectory planes 0-2 of this state, planes 3-4 of the previous step, frame pose: issued where this wave is about to wait
      // (every DPP read happens with the whole quad active: values are formed first, selected after, stored under the body mask last)
      const float vx_all = Q_BC0(s.v), vy_all = Q_BC1(s.v);
      const float pl1 = k.isv ? s.w : vx_all, pl2 = k.isv ? s.p : vy_all;
      {
        float *tj = a.ws + (size_t)step * (PD_TRAJ_G * 4) * N;
        stg(tj, boff_tj, s.r);
        stg(tj + (size_t)4 * N, boff_tj, pl1);
        stg(tj + (size_t)8 * N, boff_tj, pl2);
        // (no `step > 0` region: step 0 writes the zeros o_p3 / o_p4 hold into step 0's planes 3-4, step 1 overwrites them)
        float *tp = a.ws + (size_t)(step > 0 ? step - 1 : 0) * (PD_TRAJ_G * 4) * N;
        stg(tp + (size_t)12 * N, boff_tj, o_p3); stg(tp + (size_t)16 * N, boff_tj, o_p4);
      }
      if (fr >= 0) q_frame_out(fr, s);
      STAMP(8);
      pair_wait(sig + 1, step + 1);  // B: contact wrenches are complete
      STAMP(9);
      {  // (lane 3 reads floats 3 and 6 of the slot -- dropped -- and zeroes them: float 3 is zeroed by lane 0 as well, 6 is the pad)
        float *f = facc + bb * PD_W6;
        const float f_t = f[qc], f_f = f[3 + qc];
        ft += k.isv ? f_t : 0.f; ff += k.isv ? f_f : 0.f;
        f[qc] = 0.f; f[3 + qc] = 0.f;
      }
      const float grf_t = ft, grf_f = ff;  // res_f + contacts (integrator_euler.py:510)
      ft += jt; ff += jf;
      if (fr >= 0) if (k.isv) {  // force snapshots of a frame step (the wave-uniform test first)
        if (a.grf) { float *o = a.grf + ((size_t)fr * N + qidx) * 6; o[qc] = grf_t; o[3 + qc] = grf_f; }
        if (a.jaf) { float *o = a.jaf + ((size_t)fr * N + qidx) * 6; o[qc] = ft - grf_t; o[3 + qc] = ff - grf_f; }
      }
      {  // plane 3: (v.z, t)
        const float vz_all = Q_BC2(s.v), t_sh = q_dpp<PD_QP(0, 0, 1, 2)>(ft);
        o_p3 = qc == 0 ? vz_all : t_sh;
      }
      STAMP(3);
      // ---- integrate_bodies
      float sink;
      unsigned mask;
      {
        QM3 R1r, R1c;
        s = q_integrate(k, B, s, Rr, Rc, rc, ft, ff, a.dt, R1r, R1c, rc, sink, mask);
        Rr = R1r; Rc = R1c;
      }
      o_p4 = k.isv ? ff : __uint_as_float(mask);  // plane 4: (f, clamp mask)
      STAMP(4);
      {  // did every body stay inside the margin the cull speculated with?  (NaN counts as "no")
        sunk += sink * a.dt;
        if (CULLW) sunk_n += sink * a.dt;
        const bool bad = B.sphere_w >= 0.0f && !(sunk <= 0.98f * margin);
        spec_failed = __ballot(bad) != 0ull;
        if constexpr (CULLW) {
          const bool rot = step % PD_SPEC_K == 0;
          margin = rot ? margin_n : margin; sunk = rot ? sunk_n : sunk;
        }
      }
      WAVE_SYNC();
      {
        const bool epoch = (step + 1) % PD_SPEC_K == 0;  // state step+1 opens a speculation epoch
        float lower = margin;
        if (epoch) {
          lower = q_margin(s);
          if (CULLW) { margin_n = lower; sunk_n = 0.f; } else { margin = lower; sunk = 0.f; }
        }
        q_stage(s, rc, Rr, lower, epoch ? spec + (((step + 1) / PD_SPEC_K) & 1) * nb : nullptr);
      }
      STAMP(5);
    }
    STAMP_FLUSH(a);
    if (a.nsteps > 0 && qbody) {
      float *tp = a.ws + (size_t)(a.nsteps - 1) * (PD_TRAJ_G * 4) * N;
      stg(tp + (size_t)12 * N, boff_tj, o_p3); stg(tp + (size_t)16 * N, boff_tj, o_p4);
    }
    {  // a frame may name the state after the last step: pose / twist, zero force rows
      const int fr_last = ld_uniform(a.frame_of_step, a.nsteps);
      if (fr_last >= 0) {
        q_frame_out(fr_last, s);
        if (qbody && k.isv) {
          if (a.grf) { a.grf[((size_t)fr_last * N + qidx) * 6 + qc] = 0.f; a.grf[((size_t)fr_last * N + qidx) * 6 + 3 + qc] = 0.f; }
          if (a.jaf) { a.jaf[((size_t)fr_last * N + qidx) * 6 + qc] = 0.f; a.jaf[((size_t)fr_last * N + qidx) * 6 + 3 + qc] = 0.f; }
        }
      }
    }
    if constexpr (LOSS) {  // every lane reads back the pose components it stored itself; the quad gathers them by DPP
      for (int f = 0; f < a.nframes; ++f) {
        const float *o = a.wp_pos + ((size_t)f * N + qidx) * 7;
        const float l_p = o[qv], l_r = o[3 + qc];
        q_frame_loss(f, k.isv ? l_p : 0.f, l_r);
      }
    }
    return;
  }
  // CLONE (round 6; wave-specialised kernels of PLAIN models, pd_parented(JT)): the body wave's step is BRANCH-FREE.  An idle lane
  // (l >= nb) is a full clone of its env's last body -- same constants, same state, same loads -- and writes what that body's lane
  // writes to the same LDS / global addresses (same values, same instruction); a lane without a joint (the FREE root) runs the joint
  // pass on the record of body 0 and drops the result in a select.  A lone wave pays ~9 cycles per branch instruction taken or not,
  // ~2.6 per exec-mask instruction and ~4.5 per v_mov that merges a divergent region's results (scripts/micro/issue_mix.hip,
  // profiles/r06_issue_mix.txt): the guarded form of this loop carried 22 branches, 13 exec-mask regions and 67 moves per step.
  // Only GLOBAL stores stay guarded, by env_ok alone (an env slot past the batch clones env 0 without its contacts).
  // The unsplit instantiations of plain models keep their guards -- two busy waves per SIMD: in the branch-free form human 4096 ran 0.408
  // ms against 0.349, quad 8192 0.807 against 0.731 -- but take the same ARITHMETIC (LEANA: the quaternion update without its zero
  // products, the own joint's wrench subtracted inside the packed sums): a compound robot's env must give the same bits on either side of
  // the batch size that switches kernels (test_unsplit_forward_speculates_and_redoes_its_cull).
  static_assert(SPLIT || !LOSS, "the loss-evaluating forward is wave-specialised only");
  constexpr bool CLONE = SPLIT && pd_parented(JT);
  constexpr bool LEANA = pd_parented(JT);
  const bool wr = CLONE || is_body;            // LDS writes / per-body work
  const bool gw = CLONE ? env_ok : is_body;    // global stores
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
  if (wr) {
#pragma unroll
    for (int k = 0; k < 6; ++k) facc[b * PD_W6 + k] = 0.f;
  }
  if (l == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) pcon[nb * PD_W6 + k] = 0.f;
  }
  int cz[4];  // first four children, the zero record for a missing one
#pragma unroll
  for (int k = 0; k < 4; ++k) cz[k] = wr && c.child[k] >= 0 ? c.child[k] : nb;

  // ---- eval_fk (dp_model.py:1204): level-synchronous walk of the chain through LDS
  BodyState s;
  s.p = V3(0, 0, 0); s.r = Q4(0, 0, 0, 1); s.w = V3(0, 0, 0); s.v = V3(0, 0, 0);
  float Rm[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};  // rotm(s.r) of the current state: integration, staging and the joint's axis share it
  v3 rc = V3(0, 0, 0);  // Rm com of the current state, shared by staging, joints and integration
  float margin = 0.f, sunk = 0.f;  // speculative contact cull: allowed / integrated loss of height since the epoch's state
  float margin98 = __builtin_inff();
  // CULLW: the candidates of the epoch that opens with state S serve states S + 2 .. S + K + 1 (the cull wave has two steps), so an epoch's
  // bound and sum run beside its predecessor's for one step: (margin98, sunk) is the pair in use, (margin98_n, sunk_n) the one staged
  constexpr int SPEC_STEPS = CULLW ? PD_SPEC_K + 1 : PD_SPEC_K;
  [[maybe_unused]] float margin98_n = __builtin_inff(), sunk_n = 0.f;
  for (int d = 0; d <= m.max_depth; ++d) {
    if (wr && c.depth == d) {
      s = fk_joint<JT>(c, a.q_init + (size_t)ec * m.nq + c.qstart, a.qd_init + (size_t)ec * m.nqd + c.qdstart, rec);
      rotm(s.r, Rm);
      rc = mat_vec(Rm, c.com);
      float4 cv = stage_record(rec, cull, b, s, rc, Rm);
      margin = sink_margin<JT, SPEC_STEPS>(m, c, s, a.dt);
      if (CULLW) margin98_n = c.sphere.w >= 0.0f ? 0.98f * margin : __builtin_inff();
      else margin98 = c.sphere.w >= 0.0f ? 0.98f * margin : __builtin_inff();
      cv.x -= margin;
      if (SPLIT) spec[b] = cv;  // epoch 0
    }
    WAVE_SYNC();
  }
  bool spec_failed = true;  // wave-uniform: some body of the wave outran its margin (nothing is speculated for step 0)

  // Controls are software-prefetched one step ahead: with one wavefront per SIMD there is no other
  // wave to hide the HBM latency of a load issued at its point of use.
  unsigned boff = (unsigned)idx * 4u, boff_qd = (unsigned)((size_t)ec * m.nqd + c.qdstart) * 4u;  // per-lane byte offsets
  unsigned boff16 = boff * 4u, boff24 = boff * 6u;  // ... of this body's float4 (trajectory planes) and of its six floats (res_f)
  float n_tgt[ND], n_act[ND], n_rf[6];
  int n_fr = -1;  // frame that state `step` is gathered into (or -1), fetched with the controls
  size_t ow_prev = (size_t)12 * N;  // CLONE: offset of planes 3-4 of the previous step (of step 0 at step 0)

  auto load_controls = [&](int step) {
    const int sc = __builtin_amdgcn_readfirstlane(step < a.nsteps ? step : a.nsteps - 1);  // keeps the address arithmetic scalar
    n_fr = ld_uniform(a.frame_of_step, sc);  // scalar load: no vector-memory instruction for a wave-uniform value
    // (the LOADS keep the compiler's addressing -- hoisted per-lane 64-bit addresses plus one scalar step offset shared by refs and
    // torques: the saddr form costs more scalar instructions per array than the vector adds it saves, and running offsets instead of the
    // step x stride products overflow the scalar register file into v_readlane / v_writelane inside the loop: both built and counted)
    const size_t o = (size_t)sc * a.bs * m.nqd;
    const float *rb = a.refs + o, *tb = a.torques + o, *rf = a.res_f + (size_t)sc * N * 6;
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      // (CLONE: unconditional -- the FREE root reads the first ND of its own six dofs, valid addresses, and its joint result is dropped)
      bool on = LEANA || k < ndof;
      n_tgt[k] = on ? ldg(rb + k, boff_qd) : 0.f;
      n_act[k] = on ? ldg(tb + k, boff_qd) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float2 v = ldg2(rf + 2 * k, boff24);
      n_rf[2 * k] = v.x; n_rf[2 * k + 1] = v.y;
    }
  };
  // The trajectory record of a step is written in two parts, both where this wave is about to wait for the contact wave:
  // the state (planes 0-2, frame pose / twist) straight from the live registers of the step itself, the total wrench and
  // the clamp mask of the step's integration (planes 3-4) one step late from eight copies -- instead of one late record
  // from copies of everything (26 moves per step on this wave's chain)
  float o_vz = 0.f;
  v3 o_ft = V3(0, 0, 0), o_ff = o_ft;
  unsigned o_mask = 0u, clamp_mask = 0u;  // which velocity components the step's integration clamped (stored for the adjoint)
  auto spill_state = [&](int step, const BodyState &cs, int cfr) {
    size_t oj = (size_t)step * (PD_TRAJ_G * 4) * N;
    PD_OPAQUE_S(oj);  // (the OFFSET: a pointer that went through the asm is a generic one, its accesses flat_*; and in front of the guard:
    //                    inside the divergent region the forced scalar does not compile)
    float *tj = a.ws + oj;
    if (!gw) return;
    if constexpr (!TRAJC) {  // (TRAJC: the contact wave stores these three planes)
      stg4(tj, boff16, make_float4(cs.r.x, cs.r.y, cs.r.z, cs.r.w));
      stg4(tj + (size_t)4 * N, boff16, make_float4(cs.w.x, cs.w.y, cs.w.z, cs.v.x));
      stg4(tj + (size_t)8 * N, boff16, make_float4(cs.p.x, cs.p.y, cs.p.z, cs.v.y));
    }
    if (cfr >= 0) {  // frame gather (dp_model.py:1231-1248)
      float *o = a.wp_pos + ((size_t)cfr * N + idx) * 7;
      o[0] = cs.p.x; o[1] = cs.p.y; o[2] = cs.p.z; o[3] = cs.r.x; o[4] = cs.r.y; o[5] = cs.r.z; o[6] = cs.r.w;
      o = a.wp_vel + ((size_t)cfr * N + idx) * 6;
      o[0] = cs.w.x; o[1] = cs.w.y; o[2] = cs.w.z; o[3] = cs.v.x; o[4] = cs.v.y; o[5] = cs.v.z;
    }
  };
  auto frame_loss = [&](int cfr, const BodyState &cs) {  // cfr >= 0, wave-uniform
    if constexpr (LOSS) {
      float lb = 0.f;
      if (is_body) {
        const float pose[7] = {cs.p.x, cs.p.y, cs.p.z, cs.r.x, cs.r.y, cs.r.z, cs.r.w};
        const size_t ot = (((size_t)ec * a.nframes + cfr) * nb + b) * 7;
        float tg[7], gp[7], gg[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) tg[k] = a.loss_target[ot + k];
        lb = pd_se3::se3_loss_eval<7>(pose, tg, a.loss_rot_ratio, true, gp, gg);
        float *o = a.loss_seed_pos + ((size_t)cfr * N + idx) * 7;
#pragma unroll
        for (int k = 0; k < 7; ++k) o[k] = gp[k];
        if (a.loss_seed_gt) {
#pragma unroll
          for (int k = 0; k < 7; ++k) a.loss_seed_gt[ot + k] = gg[k];
        }
      }
      // mean over the env's bodies (dp_model.py:777 .mean(-1)): butterfly over the segment's lanes, idle lanes carry 0
#pragma unroll
      for (int w = SEGW / 2; w >= 1; w >>= 1) lb += __shfl_xor(lb, w, SEGW);
      if (l == 0 && env_ok) {
        const size_t oe = (size_t)ec * a.nframes + cfr;
        a.loss_table[oe] = (a.loss_outseq && a.loss_outseq[oe]) ? 0.f : lb / (float)nb;  // loss_traj[outseq_idx] = 0 (:778)
      }
    }
  };
  auto spill_wrench_at = [&](float *tj) {  // what the o_* registers hold for a step, to planes 3-4 of that step at tj
    if (!gw) return;
    stg4(tj, boff16, make_float4(o_vz, o_ft.x, o_ft.y, o_ft.z));
    stg4(tj + (size_t)4 * N, boff16, make_float4(o_ff.x, o_ff.y, o_ff.z, __uint_as_float(o_mask)));
  };
  auto spill_wrench = [&](int step) { spill_wrench_at(a.ws + (size_t)step * (PD_TRAJ_G * 4) * N + (size_t)12 * N); };
  if (a.nsteps > 0) load_controls(0);

  // unsplit kernel: the speculated candidates of this wave's envs (see the sweep in the loop)
  int u_since = PD_SPEC_K, u_nh = 0, u_e = 0;
  bool u_have = false, u_owns = false;
  float4 u_P = make_float4(0.f, 0.f, 0.f, 0.f), u_M = u_P;
  STAMP_DECL;
  for (int step = 0; step < a.nsteps; ++step) {
    // hand-over A first: the records of this step were staged at the end of the previous iteration (or by FK), so the
    // contact wave starts sweeping while this wave still unpacks controls and spills the state
    if (SPLIT) pair_signal(sig, (step + 1) | (spec_failed ? PD_SIG_FLAG : 0));  // A: hand this step's records to the contact wave
    PD_OPAQUE_V(boff16);  // (see PD_OPAQUE_V: the trajectory stores' lane offset stays a 32-bit operand)
    STAMP(0);
    PD_WAIT_VMEM();
    float tgt[ND], act[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) { tgt[k] = n_tgt[k]; act[k] = n_act[k]; }
    v3 ft = V3(n_rf[0], n_rf[1], n_rf[2]), ff = V3(n_rf[3], n_rf[4], n_rf[5]);  // clear_forces + wp_add
    const int fr = n_fr;
    load_controls(step + 1);
    if (!SPLIT) {
      // Unsplit kernel (compound / generic robots at large batches): the same speculation as the contact wave's, inline (round 3;
      // rounds 1-2 ran the exact three-level sweep here EVERY step: ~38 % of quad's forward step).  Every PD_SPEC_K steps -- or at
      // once when a body outran its margin -- this wave culls with the vectors of the CURRENT state lowered by the margin and keeps
      // the candidates (lane j = candidate j); the steps in between evaluate them on the state that then exists.  contact_hit
      // applies the reference's exact test to each, the candidates are a superset of the exact sweep's hits in the same order, so
      // the wrench sums are those of the exact sweep.
      WAVE_SYNC();
      int *lg = a.hitlog + ((size_t)step * a.bs + ec) * PD_HITLOG;
      if (u_since >= PD_SPEC_K || spec_failed || !u_have) {  // wave-uniform
        if (is_body) {
          margin = sink_margin<JT>(m, c, s, a.dt); sunk = 0.f;
          float4 cv = cull[b];
          cv.x -= margin;
          spec[b] = cv;
        }
        WAVE_SYNC();
        const int nlist = sweep_cull<SEGW>(m, tabs, c, is_body ? spec[b] : make_float4(0.f, 0.f, 1.f, 0.f), spec, list, is_body, seg, l STAMP_PASS);
        u_have = sweep_l3_spec<SEGW>(tabs, spec, list, nlist, hits, seg, l, u_nh);
        u_owns = u_have && __ballot(u_nh > SEGW) == 0ull;
        if (u_owns) {
          u_e = l < u_nh ? hits[l] : 0;
          u_P = tabs.pts[u_e & 0xffff]; u_M = tabs.mats[(u_e >> 16) & 0xff];
        }
        u_since = 0;
      }
      ++u_since;
      if (!u_have) {  // the candidates did not fit the list: the exact sweep, which flushes in pieces
        int log_n;
        sweep_contacts<SEGW, 6, PD_W6, true>(m, tabs, c, is_body ? cull[b] : make_float4(0.f, 0.f, 1.f, 0.f), rec, cull, list, hits, slot, facc,
                                             is_body, env_ok, seg, l, nullptr, PD_NO_REPLAY, log_n, contact_hit STAMP_PASS);
        write_hit_log<SEGW>(lg, hits, log_n, env_ok, l);
      } else {
        const SegMask usm = seg_mask<SEGW>(seg);
        int log_n = 0;
        if (u_owns) {
          bool tch = false;
          if (l < u_nh) {
            float out[6];
            tch = contact_hit(rec + ((u_e >> 24) & 0x3f) * PD_REC, u_P, u_M, out);
            if (tch) {
#pragma unroll
              for (int i = 0; i < 6; ++i) atomicAdd(facc + ((u_e >> 24) & 0x3f) * PD_W6 + i, out[i]);
            }
          }
          const int sl = seg_slot(tch, usm, log_n);
          if (tch && env_ok && sl < PD_HITLOG - 1) lg[1 + sl] = u_e;
        } else {  // more candidates than lanes: batches out of the LDS list
          for (int j0 = 0; __ballot(j0 < u_nh) != 0ull; j0 += SEGW) {
            const int j = j0 + l;
            bool tch = false;
            int e = 0;
            if (j < u_nh) {
              e = hits[j];
              float out[6];
              tch = contact_hit(rec + ((e >> 24) & 0x3f) * PD_REC, tabs.pts[e & 0xffff], tabs.mats[(e >> 16) & 0xff], out);
              if (tch) {
#pragma unroll
                for (int i = 0; i < 6; ++i) atomicAdd(facc + ((e >> 24) & 0x3f) * PD_W6 + i, out[i]);
              }
            }
            const int sl = seg_slot(tch, usm, log_n);
            if (tch && env_ok && sl < PD_HITLOG - 1) lg[1 + sl] = e;
          }
        }
        if (l == 0 && env_ok) lg[0] = log_n < PD_HITLOG ? log_n : -1;
      }
    }
    STAMP(1);
    // ---- eval_body_joints (runs while the contact wave sweeps)
    v3 wp_t = V3(0, 0, 0), wp_f = wp_t, wc_t = wp_t, wc_f = wp_t;
    if constexpr (LEANA) {  // (the unsplit kernels too: where the joint pass sits among the blocks decides which products the compiler fuses)
      joint_fwd<JT, JT == PD_JT_COMPOUND, true, true>(m, c, s, rc, Rm, rec, tgt, act, ke, kd, wp_t, wp_f, wc_t, wc_f);
      const bool jointed = c.type != PD_JOINT_FREE;
      wp_t = jointed ? wp_t : V3(0, 0, 0); wp_f = jointed ? wp_f : V3(0, 0, 0);
      wc_t = jointed ? wc_t : V3(0, 0, 0); wc_f = jointed ? wc_f : V3(0, 0, 0);
    } else {
      if (is_body && c.type != PD_JOINT_FREE) joint_fwd<JT, JT == PD_JT_COMPOUND>(m, c, s, rc, Rm, rec, tgt, act, ke, kd, wp_t, wp_f, wc_t, wc_f);
    }
    if (wr) {
      float *pc = pcon + b * PD_W6;
      pc[0] = wp_t.x; pc[1] = wp_t.y; pc[2] = wp_t.z; pc[3] = wp_f.x; pc[4] = wp_f.y; pc[5] = wp_f.z;
    }
    STAMP(2);
    WAVE_SYNC();
    v3 jt = -wc_t, jf = -wc_f;  // joint wrench on this body: own joint first, then children in index order
    {  // first four children: all LDS reads are issued back to back (one exposed latency instead of one per child), packed sums
      const float *const src[4] = {pcon + cz[0] * PD_W6, pcon + cz[1] * PD_W6, pcon + cz[2] * PD_W6, pcon + cz[3] * PD_W6};
      if constexpr (LEANA) wrench_sub_add_from_n(jt, jf, wc_t, wc_f, src);  // (same sums, the negation as an operand modifier)
      else wrench_add_from_n(jt, jf, src);
    }
    for (int k = 4; k < m.max_children; ++k) {
      int cid = (int)((c.children >> (8 * k)) & 0xffull);
      if (wr && cid != 0xff) {
        const float *pc = pcon + cid * PD_W6;
        jt += V3(pc[0], pc[1], pc[2]); jf += V3(pc[3], pc[4], pc[5]);
      }
    }
    STAMP(6);
    if (SPLIT) {
      // the previous step's trajectory record and frame outputs are issued where this wave is about to wait anyway (measured
      // against right after hand-over A, and against after the vmcnt wait: -2 % / -0.5 % forward time at 4096 envs)
      spill_state(step, s, fr);
      // CLONE: no `step > 0` region of its own -- step 0 writes the zeros o_* hold into step 0's planes 3-4, which step 1 (or the store
      // behind the loop) overwrites: same lane, same address, program order
      if constexpr (CLONE) {
        size_t ow = ow_prev;  // planes 3-4 of step max(step - 1, 0)
        PD_OPAQUE_S(ow);
        spill_wrench_at(a.ws + ow);
        ow_prev = (size_t)step * (PD_TRAJ_G * 4) * N + (size_t)12 * N;
      } else if (step > 0) {
        spill_wrench(step - 1);
      }
      STAMP(8);
      pair_wait(sig + 1, step + 1);  // B: contact wrenches are complete
      STAMP(9);
    } else {
      spill_state(step, s, fr);
      if (step > 0) spill_wrench(step - 1);
      WAVE_SYNC();
    }
    if (wr) {
      float *f = facc + b * PD_W6;
      ft += V3(f[0], f[1], f[2]); ff += V3(f[3], f[4], f[5]);
#pragma unroll
      for (int k = 0; k < 6; ++k) f[k] = 0.f;
    }
    const v3 grf_t = ft, grf_f = ff;  // res_f + contacts (integrator_euler.py:510)
    ft += jt; ff += jf;
    if (fr >= 0) if (gw) {  // force snapshots of a frame step (the wave-uniform test first: a scalar branch around the region) (4 of 100 steps): written here, no copies carried for them
      if (a.grf) {
        float *o = a.grf + ((size_t)fr * N + idx) * 6;
        o[0] = grf_t.x; o[1] = grf_t.y; o[2] = grf_t.z; o[3] = grf_f.x; o[4] = grf_f.y; o[5] = grf_f.z;
      }
      if (a.jaf) {
        float *o = a.jaf + ((size_t)fr * N + idx) * 6;
        o[0] = ft.x - grf_t.x; o[1] = ft.y - grf_t.y; o[2] = ft.z - grf_t.z;
        o[3] = ff.x - grf_f.x; o[4] = ff.y - grf_f.y; o[5] = ff.z - grf_f.z;
      }
    }
    o_vz = s.v.z; o_ft = ft; o_ff = ff;  // the total wrench goes to the trajectory with the clamp mask, one step late
    STAMP(3);
    // ---- integrate_bodies
    float sink_rate;
    {
      float R1[9];
      s = integrate_fwd<LEANA>(m, c, s, Rm, rc, ft, ff, inv_m, I, invI, a.dt, R1, rc, sink_rate, clamp_mask);
#pragma unroll
      for (int k = 0; k < 9; ++k) Rm[k] = R1[k];
    }
    o_mask = clamp_mask;
    STAMP(4);
    {  // did every body stay inside the margin the cull speculated with?  (NaN counts as "no")
      sunk += sink_rate * a.dt;
      if (CULLW) sunk_n += sink_rate * a.dt;
      // CLONE: one compare -- the bound carries "has candidates at all" (+inf otherwise) and the factor, set where the margin is
      const bool bad = CLONE ? !(sunk <= margin98) : (wr && c.sphere.w >= 0.0f && !(sunk <= 0.98f * margin));
      spec_failed = __ballot(bad) != 0ull;
      if constexpr (CULLW) {  // state step + 1 was the last one of the previous epoch's (selects: a branch here would end the basic block the
        const bool rot = step % PD_SPEC_K == 0;  // compiler contracts multiply-adds in, and the poses' last bits with it)
        margin98 = rot ? margin98_n : margin98; sunk = rot ? sunk_n : sunk;
      }
    }
    WAVE_SYNC();
    if (wr) {
      float4 cv = stage_record(rec, cull, b, s, rc, Rm);
      if (SPLIT && (step + 1) % PD_SPEC_K == 0) {  // state step+1 opens a speculation epoch
        margin = sink_margin<JT, SPEC_STEPS>(m, c, s, a.dt);
        if (CULLW) { sunk_n = 0.f; margin98_n = c.sphere.w >= 0.0f ? 0.98f * margin : __builtin_inff(); }
        else { sunk = 0.f; margin98 = c.sphere.w >= 0.0f ? 0.98f * margin : __builtin_inff(); }
        cv.x -= margin;
        spec[(((step + 1) / PD_SPEC_K) & 1) * nb + b] = cv;
      }
    }
    if (!SPLIT) WAVE_SYNC();
    STAMP(5);
  }
  if (a.nsteps > 0) spill_wrench(a.nsteps - 1);
  {  // a frame may name the state after the last step (state_steps[nsteps], dp_model.py:396,1241-1246); no force
     // snapshot exists for it (the reference appends grf / jaf for step in steps_idx only, :1225-1228): zero rows
    const int fr_last = ld_uniform(a.frame_of_step, a.nsteps);
    if (fr_last >= 0 && is_body) {
      float *o = a.wp_pos + ((size_t)fr_last * N + idx) * 7;
      o[0] = s.p.x; o[1] = s.p.y; o[2] = s.p.z; o[3] = s.r.x; o[4] = s.r.y; o[5] = s.r.z; o[6] = s.r.w;
      o = a.wp_vel + ((size_t)fr_last * N + idx) * 6;
      o[0] = s.w.x; o[1] = s.w.y; o[2] = s.w.z; o[3] = s.v.x; o[4] = s.v.y; o[5] = s.v.z;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        if (a.grf) a.grf[((size_t)fr_last * N + idx) * 6 + k] = 0.f;
        if (a.jaf) a.jaf[((size_t)fr_last * N + idx) * 6 + k] = 0.f;
      }
    }
  }
  if constexpr (LOSS) {
    // The trajectory loss, AFTER the step loop: every lane reads back the frame poses it stored and evaluates se3_loss on them.  Inside
    // the loop (at the 4 frame steps of 100) the loss code cost the loop its register allocation on EVERY step: 188 VGPRs / 55 spilled
    // SGPRs against 159 / 28, 0.222-0.224 ms against 0.211.
    for (int f = 0; f < a.nframes; ++f) {
      BodyState cs = s;
      if (is_body) {
        const float *o = a.wp_pos + ((size_t)f * N + idx) * 7;
        cs.p = V3(o[0], o[1], o[2]); cs.r = Q4(o[3], o[4], o[5], o[6]);
      }
      frame_loss(f, cs);
    }
  }
  STAMP_FLUSH(a);
}

// =============================================================================================
// EARLY (SPLIT only): hand-over A is signalled from inside the adjoint of integrate_bodies, as soon as the wrench adjoint
// exists (integrate_adj2), instead of after it.
#ifdef PD_KNOCK
#undef pair_wait
#if !(PD_KNOCK & 16)
#define pair_wait(f, v) pair_wait_knock(f, v, knock_role)
#endif
#endif
// QUAD: the body wave in the four-lanes-per-body form (pd_quad.h), one env per wave, 64-lane mapping -- see k_rollout_fwd.
template <int SEGW, int JT, bool SPLIT, bool EARLY = false, bool QUAD = false>
__global__ __launch_bounds__(SPLIT ? PD_BLOCK : PD_FK_BLOCK) void k_rollout_bwd(PdDevModel m, RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int EPW = Seg<SEGW>::EPW;
  constexpr int ND = (JT & PD_JT_COMPOUND) ? 3 : 1;
  const int bw = (int)blockDim.x / (QUAD ? 192 : (SPLIT ? 128 : 64));  // env groups per workgroup (host's choice per launch)
  const int role = SPLIT ? (int)(threadIdx.x >> 6) / bw : 0;  // wave-uniform: 0 body wave, 1 contact wave, 2 (QUAD) state wave
  const bool contact_wave = role != 0;
  [[maybe_unused]] const int knock_role = role;
  const int lane = threadIdx.x & 63, wave = (int)(threadIdx.x >> 6) % bw;
  const int seg = lane / SEGW, l = lane % SEGW;
  const int env = (blockIdx.x * bw + wave) * EPW + seg;
  const bool env_ok = env < a.bs;
  const int ec = env_ok ? env : 0;  // clamped env for safe addressing
  const bool is_body = env_ok && l < m.nb;
  const int b = l < m.nb ? l : m.nb - 1;
  const int nb = m.nb, N = a.bs * nb;
  // (the state-only half of a revolute joint's adjoint, rev_forward, comes from the contact wave through LDS; the body wave recomputing it
  // itself was measured slower at every batch size -- EXPERIMENTS.md round 3 -- and is gone from the sources since round 5)

  SweepTables tabs;
  const int env_stride = m.env_lds_floats + (SPLIT ? 2 * m.env_lds_jc : 0) + (QUAD ? m.env_lds_rec2 : 0);
  float *scratch = lds_setup<!SPLIT>(m, smem, tabs, wave * EPW + seg, env_stride);
  float4 *cull = (float4 *)scratch;
  // cslot: nb + 1 records, the last one stays zero and stands in for "no child" (the gather then needs no predicates)
  float *rec = scratch + 4 * nb, *adjf = rec + nb * PD_REC, *cslot = adjf + nb * PD_W6, *cacc = cslot + (nb + 1) * PD_ADJ;
  int *list = (int *)(cacc + nb * PD_ADJ), *hits = list + m.list_cap;
  float *slot = (float *)(hits + PD_HIT_CAP_TILES * SEGW);
  // SPLIT: revolute joint hand-over records, PD_JC floats per body, two generations (by step parity: the contact wave
  // writes step k - 1's while the body wave may still read step k's)
  float *jc = scratch + m.env_lds_floats;
  // pair signals: the spare words at the end of the first env's area
  int *sig = (int *)(scratch - (size_t)seg * env_stride + m.env_lds_floats - 4);
  if (SPLIT) {
    if (lane == 0 && !contact_wave) { sig[0] = 0; sig[1] = 0; sig[2] = 0; sig[3] = 0; }
    __syncthreads();
  }
  PD_KNOCK_EXIT(false, knock_role);
  if constexpr (QUAD) {
    if (!env_ok) return;  // one env per wave pair: a pair past the batch has nothing to do (no workgroup barrier follows)
  }

  BodyConst c = load_body_const(m, b, ec);
#pragma unroll
  for (int u = 0; u < 4; ++u) c.small_e[u] = m.small_tiles[u * 64 + (l < 64 ? l : 0)];
  if (l == 0) {
#pragma unroll
    for (int k = 0; k < PD_ADJ; ++k) cslot[nb * PD_ADJ + k] = 0.f;
  }
  int cz[4];  // first four children, the zero record for a missing one
#pragma unroll
  for (int k = 0; k < 4; ++k) cz[k] = is_body && c.child[k] >= 0 ? c.child[k] : nb;
  // (quad-lane adjoint: records and cull vectors in two generations by step parity, see its body wave; rec_s / cull_s = this step's)
  const float *rec_s = rec;
  const float4 *cull_s = cull;
  auto contact_hit = [&](const float *r, float4 P, float4 mat, float *out) {
    const int pb = (int)(r - rec_s) / PD_REC;
    BodyAdj o = adj_zero();
    if constexpr (QUAD) {  // the two pieces of contact_point_adj, as the quad-lane kernel's contact wave runs them on its fast path
      ContactPre C = contact_point_adj_pre(r, cull_s[pb], P, mat);
      contact_pre_barrier(C);
      contact_point_adj_rest(C, P, mat, ld3(adjf + pb * PD_W6), ld3(adjf + pb * PD_W6 + 3), o);
    } else {
      contact_point_adj(r, cull_s[pb], P, mat, ld3(adjf + pb * PD_W6), ld3(adjf + pb * PD_W6 + 3), o);
    }
    adj_store(out, o);
  };
  if (SPLIT && contact_wave) {
    // ---- contact wave.  Per step:
    //   .. A  : (idle time of the old design) nothing here waits for the body wave: fetch the hit list the forward
    //           sweep logged for the NEXT iteration, the points / materials of this iteration's hits, and the stored
    //           pose of this lane's body and its parent; recompute the state-only half of the revolute joint's
    //           adjoint (rev_forward) and hand it to the body wave through LDS
    //   A..B  : wrench adjoints are staged: adjoint of eval_body_contacts for the logged hits, one lane per hit,
    //           summed per body in hit order by a lane-per-component pass
    // A log that did not fit (count -1) or holds more hits than the segment has lanes takes the generic sweep.
    STAMP_DECL;
    const bool rev = is_body && c.type == PD_JOINT_REVOLUTE;
    const size_t qd_off = (size_t)ec * m.nqd + c.qdstart;
    const float ke1 = rev ? a.target_ke[qd_off] : 0.f, kd1 = rev ? a.target_kd[qd_off] : 0.f;
    const int lq = l < PD_HITLOG - 1 ? l : PD_HITLOG - 2;
    const unsigned boff_c = (unsigned)((size_t)ec * nb + b) * 4u, boff_p = (unsigned)((size_t)ec * nb + (c.parent >= 0 ? c.parent : b)) * 4u;
    const unsigned boff_qd = (unsigned)qd_off * 4u, boff_lg = (unsigned)ec * (PD_HITLOG * 4u);
    auto load_log = [&](int step, int &cnt, int &e) {
      const int *lg = a.hitlog + (size_t)__builtin_amdgcn_readfirstlane(step > 0 ? step : 0) * a.bs * PD_HITLOG;
      cnt = __float_as_int(ldg((const float *)lg, boff_lg)); e = __float_as_int(ldg((const float *)lg + 1, boff_lg + (unsigned)lq * 4u));
    };
    auto load_ctrl = [&](int step, float &tgt, float &act) {
      const size_t o = (size_t)__builtin_amdgcn_readfirstlane(step > 0 ? step : 0) * a.bs * m.nqd;
      tgt = rev ? ldg(a.refs + o, boff_qd) : 0.f; act = rev ? ldg(a.torques + o, boff_qd) : 0.f;
    };
    auto fetch_point = [&](int cnt, int &e, float4 &P, float4 &M) {  // entries past the count are uninitialised memory
      if (!(env_ok && l < cnt && l < PD_HITLOG - 1)) e = 0;
      P = m.pts[e & 0xffff]; M = m.materials[(e >> 16) & 0xff];
    };
    // stored pose of this lane's body (q, w) and of its parent (p, q, w), one iteration ahead like everything else here
    auto load_pose = [&](int step, float4 *o) {
      const float *tj = a.ws + (size_t)__builtin_amdgcn_readfirstlane(step > 0 ? step : 0) * (PD_TRAJ_G * 4) * N;
      if (rev) {
        o[0] = ldg4(tj, boff_c * 4u); o[1] = ldg4(tj + (size_t)4 * N, boff_c * 4u);
        o[2] = ldg4(tj, boff_p * 4u); o[3] = ldg4(tj + (size_t)4 * N, boff_p * 4u); o[4] = ldg4(tj + (size_t)8 * N, boff_p * 4u);
      }
    };
    int cnt_c = 0, e_c = 0, cnt_n = 0, e_n = 0;
    float4 P_c, M_c;
    float tgt_c = 0.f, act_c = 0.f;
    float4 pose[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) pose[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- QUAD (round 6): a THIRD role, the state wave.  Everything of a step that no adjoint enters runs there, up to PD_QGEN - 1 steps
    // ahead of the other two: the state-only half of the revolute joints' adjoint (rev_forward, lane-per-body layout: it was this wave's)
    // and PRE(step) of the body wave's stream -- unpack the stored state, rotm, the staged record and cull vector, the forward values
    // integrate_bodies' adjoint needs again -- in the body wave's four-lanes-per-body layout, handed over through LDS (PD_QPRE floats
    // per lane).  Measured at 512 envs with one role's waves alone in the kernel: body wave 0.174 ms, contact wave 0.181 ms, the pair
    // 0.198 ms -- each wave's own instruction stream is the step.  PRE inside the contact wave: body 0.153, contact 0.224, pair 0.233.
    // Generations by step modulo PD_QGEN: the state wave writes step k's when the body wave has signalled A of step k + PD_QGEN - 1.
    const int qb = lane >> 2, qc = lane & 3, bbq = qb < nb ? qb : nb - 1;
    const QLane kq = q_lane(qc);
    const int qgen_floats = m.env_lds_rec2 / PD_QGEN - PD_QPRE * 64 - m.env_lds_jc;   // cull vectors + records of one generation
    float *const qgen = jc + 2 * m.env_lds_jc, *const qpre = qgen + PD_QGEN * qgen_floats;
    QBody Bq;
    float npl_c[PD_TRAJ_G];
    const unsigned boff_tjq = (unsigned)(((size_t)ec * nb + bbq) * 4 + qc) * 4u;
    auto load_planes = [&](int step, float *pl) {
      const float *tj = a.ws + (size_t)__builtin_amdgcn_readfirstlane(step > 0 ? step : 0) * (PD_TRAJ_G * 4) * N;
#pragma unroll
      for (int g = 0; g < PD_TRAJ_G; ++g) pl[g] = ldg(tj + (size_t)(4 * g) * N, boff_tjq);
    };
    auto q_pre = [&](const float *pl, int g) {   // PRE of the step whose planes are pl, into generation g
      // ---- unpack the stored step: planes q | (w, v.x) | (p, v.y) | (v.z, t) | (f, clamp mask)
      QState s;
      const float vx = Q_BC3(pl[1]), vy = Q_BC3(pl[2]), vz = Q_BC0(pl[3]);
      const float tsh = q_dpp<PD_QP(1, 2, 3, 3)>(pl[3]);
      const float maskf = Q_BC3(pl[4]);
      s.r = pl[0];
      s.w = kq.isv ? pl[1] : 0.f; s.p = kq.isv ? pl[2] : 0.f;
      s.v = qc == 0 ? vx : (qc == 1 ? vy : (qc == 2 ? vz : 0.f));
      const float t0 = kq.isv ? tsh : 0.f, f0 = kq.isv ? pl[4] : 0.f;
      QM3 Rr, Rc;
      q_rotm(kq, s.r, Rr, Rc);
      const float rc = q_mvc(Rr, Bq.com0, Bq.com1, Bq.com2);
      {  // staging (stage_record): this wave's contact adjoint and the body wave's joint adjoint read records and cull vectors
        float *G = qgen + g * qgen_floats, *r = G + 4 * nb + bbq * PD_REC;
        r[3 + qc] = s.r;
        r[kq.isv ? qc : 16] = s.p; r[kq.isv ? 7 + qc : 16] = s.w; r[kq.isv ? 10 + qc : 16] = s.v; r[kq.isv ? 13 + qc : 16] = rc;  // (lane 3: the record's pad float)
        if (qc == 1) ((float4 *)G)[bbq] = make_float4(s.p, Rr.a, Rr.b, Rr.c);
      }
      QIntTmp T;
      q_integrate_adj_pre(kq, Bq, s, Rr, Rc, t0, a.dt, T);
      float *P = qpre + g * (PD_QPRE * 64) + lane;
      P[0 * 64] = s.p; P[1 * 64] = s.r; P[2 * 64] = s.w; P[3 * 64] = s.v; P[4 * 64] = t0; P[5 * 64] = f0; P[6 * 64] = maskf; P[7 * 64] = rc;
      P[8 * 64] = Rr.a; P[9 * 64] = Rr.b; P[10 * 64] = Rr.c; P[11 * 64] = Rc.a; P[12 * 64] = Rc.b; P[13 * 64] = Rc.c;
      P[14 * 64] = T.wb; P[15 * 64] = T.Iwb; P[16 * 64] = T.tb; P[17 * 64] = T.u; P[18 * 64] = T.w1; P[19 * 64] = T.il; P[20 * 64] = T.r1;
    };
    int g3 = a.nsteps > 0 ? (a.nsteps - 1) % PD_QGEN : 0;   // generation of the step at hand (step % PD_QGEN)
    if constexpr (QUAD) {
      if (role == 2) {
        // ================= state wave
        const BodyConst cb = load_body_const(m, bbq, ec);
        const size_t qidx = (size_t)ec * nb + bbq;
        const int qv = qc < 3 ? qc : 2;
        Bq.com0 = cb.com.x; Bq.com1 = cb.com.y; Bq.com2 = cb.com.z;
        const float *Ib = a.inertia + qidx * 9, *Jb = a.inv_inertia + qidx * 9;
        Bq.I.a = kq.isv ? Ib[qv * 3] : 0.f; Bq.I.b = kq.isv ? Ib[qv * 3 + 1] : 0.f; Bq.I.c = kq.isv ? Ib[qv * 3 + 2] : 0.f;
        Bq.invI.a = kq.isv ? Jb[qv * 3] : 0.f; Bq.invI.b = kq.isv ? Jb[qv * 3 + 1] : 0.f; Bq.invI.c = kq.isv ? Jb[qv * 3 + 2] : 0.f;
        float *const jcq = qpre + PD_QGEN * (PD_QPRE * 64);   // the joint hand-over records, PD_QGEN generations
        if (a.nsteps > 0) { load_ctrl(a.nsteps - 1, tgt_c, act_c); load_pose(a.nsteps - 1, pose); load_planes(a.nsteps - 1, npl_c); }
        for (int step = a.nsteps - 1; step >= 0; --step) {
          PD_WAIT_VMEM();
          // generation g3 is free once the body wave has left step + PD_QGEN behind: its hand-over A of step + PD_QGEN - 1 (<= 0: at once)
          pair_wait(sig, a.nsteps - step - (PD_QGEN - 1));
          if (rev) {
            const qt q_c = Q4(pose[0].x, pose[0].y, pose[0].z, pose[0].w), qp = Q4(pose[2].x, pose[2].y, pose[2].z, pose[2].w);
            const v3 w_c = V3(pose[1].x, pose[1].y, pose[1].z), pp = V3(pose[4].x, pose[4].y, pose[4].z), w_p = V3(pose[3].x, pose[3].y, pose[3].z);
            rev_cache_store(jcq + g3 * m.env_lds_jc + b * PD_JC, rev_forward<pd_parented(JT)>(m, c, q_c, w_c, pp, qp, w_p, tgt_c, act_c, ke1, kd1));
          }
          q_pre(npl_c, g3);
          load_ctrl(step - 1, tgt_c, act_c); load_pose(step - 1, pose); load_planes(step - 1, npl_c);
          pair_signal(sig + 1, a.nsteps - step);   // S: this step's joint hand-over records, records, cull vectors and PRE are staged
          g3 = g3 == 0 ? PD_QGEN - 1 : g3 - 1;
        }
        return;
      }
    }
    if (a.nsteps > 0) {
      load_log(a.nsteps - 1, cnt_c, e_c);
      if constexpr (!QUAD) { load_ctrl(a.nsteps - 1, tgt_c, act_c); load_pose(a.nsteps - 1, pose); }
      load_log(a.nsteps - 2, cnt_n, e_n);
      fetch_point(cnt_c, e_c, P_c, M_c);
    }
    for (int step = a.nsteps - 1; step >= 0; --step) {
      PD_WAIT_VMEM();
      // (the hand-over records are double-buffered by step parity: the body wave may still be reading the previous ones)
      if (!QUAD && rev)
      {
        const qt q_c = Q4(pose[0].x, pose[0].y, pose[0].z, pose[0].w), qp = Q4(pose[2].x, pose[2].y, pose[2].z, pose[2].w);
        const v3 w_c = V3(pose[1].x, pose[1].y, pose[1].z), pp = V3(pose[4].x, pose[4].y, pose[4].z), w_p = V3(pose[3].x, pose[3].y, pose[3].z);
        float *dst = jc + (step & 1) * m.env_lds_jc + b * PD_JC;
        rev_cache_store(dst, rev_forward<pd_parented(JT)>(m, c, q_c, w_c, pp, qp, w_p, tgt_c, act_c, ke1, kd1));
      }
      STAMP(8);
      // request everything the next iteration needs; nothing loaded here is touched before the next PD_WAIT_VMEM
      float tgt_n, act_n;
      float4 P_n, M_n;
      int cnt_n2, e_n2;
      fetch_point(cnt_n, e_n, P_n, M_n);
      if constexpr (!QUAD) {
        load_ctrl(step - 1, tgt_n, act_n);
        load_pose(step - 1, pose);
      } else {
        tgt_n = 0.f; act_n = 0.f;
      }
      load_log(step - 2, cnt_n2, e_n2);
      const bool fast = __ballot(env_ok && (cnt_c < 0 || cnt_c > SEGW)) == 0ull;  // wave-uniform
      const int nh = fast && env_ok ? cnt_c : 0;
      STAMP(7);
      if constexpr (!QUAD) pair_signal(sig + 1, a.nsteps - step);   // J: this step's joint hand-over records
      // Round 6, quad-lane kernel: the STATE half of the logged hits' adjoints (contact_point_adj_pre: the forward pass's quantities again) runs
      // as soon as the state wave has staged the step's records (hand-over S) instead of behind the wrench adjoints (hand-over A): that much
      // less stands between A and B (adjoint 0.177 -> 0.168 ms at 512 envs).  The lane-per-body kernel keeps both halves behind A: with a
      // hand-over R of its body wave right behind the staging, the early half took issue slots from that wave's phase 1 (4096 envs: 0.268 -> 0.273)
      const int pb = l < nh ? (e_c >> 24) & 0x3f : -2;
      ContactPre cpre;
      cpre.touch = false;
      if constexpr (QUAD) {
        pair_wait(sig + 1, a.nsteps - step);
        rec_s = qgen + g3 * qgen_floats + 4 * nb;
        cull_s = (const float4 *)(qgen + g3 * qgen_floats);
        if (fast && l < nh) cpre = contact_point_adj_pre(rec_s + pb * PD_REC, cull_s[pb], P_c, M_c);
      }
      // A: wait for the wrench adjoints (adjf; lane-per-body kernel: and the records, cull vectors)
      pair_wait(sig, a.nsteps - step);
      __builtin_amdgcn_s_setprio(PD_PRIO_CRITICAL);  // the body wave will wait for these contact adjoints
      STAMP(9);
      if (fast) {
        float out[PD_ADJ];
#pragma unroll
        for (int i = 0; i < PD_ADJ; ++i) out[i] = 0.f;
        if constexpr (QUAD) {
          if (l < nh) {
            BodyAdj o = adj_zero();
            contact_pre_barrier(cpre);
            contact_point_adj_rest(cpre, P_c, M_c, ld3(adjf + pb * PD_W6), ld3(adjf + pb * PD_W6 + 3), o);
            adj_store(out, o);
          }
        } else {
          if (l < nh) contact_hit(rec_s + pb * PD_REC, P_c, M_c, out);
        }
        STAMP(10);
        bool last;
        seg_run_sum<PD_ADJ>(out, pb, l, nh, last);
        if (last) {  // cacc is zero (its owner clears it after reading) and a body has one run
#pragma unroll
          for (int i = 0; i < PD_ADJ; ++i) cacc[pb * PD_ADJ + i] = out[i];
        }
        STAMP(11);
      } else {
        float4 cv = make_float4(0.f, 0.f, 1.f, 0.f);
        if (is_body) cv = cull_s[b];
        int *lg = a.hitlog + ((size_t)step * a.bs + ec) * PD_HITLOG;
        const bool replay = __ballot(env_ok && cnt_c < 0) == 0ull;  // -1: the list did not fit the log, cull again (whole wave)
        int log_n_unused;
        sweep_contacts<SEGW, PD_ADJ, PD_ADJ, false>(m, tabs, c, cv, rec_s, cull_s, list, hits, slot, cacc, is_body, env_ok, seg, l,
                                                    replay ? lg : nullptr, replay ? (env_ok ? cnt_c : 0) : PD_NO_REPLAY, log_n_unused,
                                                    contact_hit STAMP_PASS);
      }
      STAMP(12);
      pair_signal(sig + 2, a.nsteps - step);  // B: contact adjoints are complete
      __builtin_amdgcn_s_setprio(0);
      cnt_c = cnt_n; e_c = e_n; P_c = P_n; M_c = M_n; tgt_c = tgt_n; act_c = act_n;
      cnt_n = cnt_n2; e_n = e_n2;
      if constexpr (QUAD) g3 = g3 == 0 ? PD_QGEN - 1 : g3 - 1;
    }
    STAMP_FLUSH(a);
    return;
  }
  if constexpr (QUAD) {
    static_assert(SEGW == 64 && SPLIT && !EARLY && JT == PD_JT_REVOLUTE, "quad-lane body wave: one env per wave, revolute-only plain models");
    // ================= body wave, four lanes per body (see k_rollout_fwd): the reverse sweep of dp_model.py:1251-1400
    const int qb = lane >> 2, qc = lane & 3, bb = qb < nb ? qb : nb - 1, qv = qc < 3 ? qc : 2;
    // round 6 (see the forward kernel's quad-lane loop): waves past the batch have left, idle quads clone the env's last body, lane 3 of a quad
    // sends the vector fields it has no part in to a PAD float (of its body's record, or of its wrench-adjoint slot: the 13-float adjoint
    // records have none) and reads component 2 where the others read their own
    const bool qbody = true;
    float *const dummy = adjf + bb * PD_W6 + 6;
    const QLane k = q_lane(qc);
    const size_t qidx = (size_t)ec * nb + bb;
    QBody B;
    QM3 It, invIt;
    float com_par0, com_par1, com_par2, p_pj0, p_pj1, p_pj2;
    QPerm pjc;
    {
      const BodyConst cb = load_body_const(m, bb, ec);
      B.type = cb.type; B.pidx = cb.pidx; B.qdstart = cb.qdstart;
      B.joint = qbody && cb.type == PD_JOINT_REVOLUTE;
      B.com = q_pick(k, cb.com); B.com0 = cb.com.x; B.com1 = cb.com.y; B.com2 = cb.com.z;
      B.axis = q_pick(k, cb.axis); B.ax0 = cb.axis.x; B.ax1 = cb.axis.y; B.ax2 = cb.axis.z; B.alen = cb.alen;
      B.p_pj = q_pick(k, cb.p_pj); B.q_pj = q_pick(k, cb.q_pj); B.pj = q_perm(k, B.q_pj);
      pjc = q_perm(k, B.q_pj * k.sc);
      com_par0 = cb.com_par.x; com_par1 = cb.com_par.y; com_par2 = cb.com_par.z;
      p_pj0 = cb.p_pj.x; p_pj1 = cb.p_pj.y; p_pj2 = cb.p_pj.z;
      B.g = q_pick(k, V3(m.gx, m.gy, m.gz));
      B.reach = cb.reach; B.sphere_w = cb.sphere.w; B.lim = cb.lim[0];
      B.inv_m = a.inv_mass[qidx];
      const bool on = cb.type == PD_JOINT_REVOLUTE;
      B.ke = on ? a.target_ke[(size_t)ec * m.nqd + cb.qdstart] : 0.f;
      B.kd = on ? a.target_kd[(size_t)ec * m.nqd + cb.qdstart] : 0.f;
      const float *Ib = a.inertia + qidx * 9, *Jb = a.inv_inertia + qidx * 9;
      B.I.a = k.isv ? Ib[qv * 3] : 0.f; B.I.b = k.isv ? Ib[qv * 3 + 1] : 0.f; B.I.c = k.isv ? Ib[qv * 3 + 2] : 0.f;
      B.invI.a = k.isv ? Jb[qv * 3] : 0.f; B.invI.b = k.isv ? Jb[qv * 3 + 1] : 0.f; B.invI.c = k.isv ? Jb[qv * 3 + 2] : 0.f;
      It.a = k.isv ? Ib[qv] : 0.f; It.b = k.isv ? Ib[3 + qv] : 0.f; It.c = k.isv ? Ib[6 + qv] : 0.f;
      invIt.a = k.isv ? Jb[qv] : 0.f; invIt.b = k.isv ? Jb[3 + qv] : 0.f; invIt.c = k.isv ? Jb[6 + qv] : 0.f;
    }
    int qcz[4];
    const unsigned long long q_children = m.children[bb];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int cid = (int)((q_children >> (8 * j)) & 0xffull); qcz[j] = (qbody && cid != 0xff) ? cid : nb; }
    if (lane < PD_ADJ) cslot[nb * PD_ADJ + lane] = 0.f;   // the zero record
    // cacc / cslot record of this lane: the quaternion float, and the three vector floats (lane 3: the pad)
    float *const ca_r = cacc + bb * PD_ADJ + 3 + qc, *const ca_p = k.isv ? cacc + bb * PD_ADJ + qc : dummy,
                *const ca_w = k.isv ? cacc + bb * PD_ADJ + 7 + qc : dummy, *const ca_v = k.isv ? cacc + bb * PD_ADJ + 10 + qc : dummy;
    float *const cs_r = cslot + bb * PD_ADJ + 3 + qc, *const cs_p = k.isv ? cslot + bb * PD_ADJ + qc : dummy,
                *const cs_w = k.isv ? cslot + bb * PD_ADJ + 7 + qc : dummy, *const cs_v = k.isv ? cslot + bb * PD_ADJ + 10 + qc : dummy;
    *ca_r = 0.f; *ca_p = 0.f; *ca_w = 0.f; *ca_v = 0.f;
    QM3 g_I, g_invI;
    g_I.a = g_I.b = g_I.c = 0.f; g_invI = g_I;
    float g_inv_m = 0.f, g_ke = 0.f, g_kd = 0.f;
    QAdj gn;
    gn.p = gn.r = gn.w = gn.v = 0.f;
    // stored state / wrench / controls of the NEXT iteration are prefetched: lane c owns float c of each trajectory plane
    const unsigned boff_tj = (unsigned)(qidx * 4 + qc) * 4u, boff_qd = (unsigned)((size_t)ec * m.nqd + B.qdstart) * 4u;
    const unsigned boff_rf = (unsigned)(qidx * 6 + qv) * 4u;
    float n_tgt = 0.f;
    int n_fr = -1;
    auto load_step = [&](int step) {
      const int sc = __builtin_amdgcn_readfirstlane(step >= 0 ? step : 0);
      n_fr = a.frame_of_step[sc + 1];
      const size_t o = (size_t)sc * a.bs * m.nqd;
      n_tgt = ldg(a.refs + o, boff_qd);   // (unconditional; the applied torque enters the adjoint through the contact wave's jf only)
    };
    const float ake = m.attach_ke, akd = m.attach_kd;
    // Round 6: everything of a step that no adjoint enters -- unpacking the stored state, rotm, the staged record and cull vector, the
    // forward values integrate_bodies' adjoint needs again -- is PRE(step).  It first moved one step ahead inside this wave, into the wait
    // for the contact adjoints (0.208 -> 0.199 ms at 512 envs); then to the CONTACT wave, whose stream was less than half of this one's
    // (see there): this wave takes PD_QPRE floats per lane from LDS.  What stays here of the look-ahead: the target angle and the
    // frame's seeds (global loads, requested one step early).
    const int qgen_floats = m.env_lds_rec2 / PD_QGEN - PD_QPRE * 64 - m.env_lds_jc;
    float *const qgen = jc + 2 * m.env_lds_jc, *const qpre = qgen + PD_QGEN * qgen_floats;
    QState s;
    float t0 = 0.f, f0 = 0.f, tgt = 0.f, rc = 0.f, sd_p = 0.f, sd_r = 0.f, sd_w = 0.f, sd_v = 0.f;
    unsigned mask = 0u;
    QM3 Rr, Rc;
    QIntTmp T;
    auto q_ahead = [&](int step) {
      // ---- seeds of this state (dp_model.py:1264-1271): requested here, added when the running adjoint reaches the step
      const int fr = n_fr;
      sd_p = 0.f; sd_r = 0.f; sd_w = 0.f; sd_v = 0.f;
      if (fr >= 0) {
        const float *gp = a.adj_pos + ((size_t)fr * N + qidx) * 7, *gv = a.adj_vel + ((size_t)fr * N + qidx) * 6;
        sd_p = gp[qv]; sd_r = gp[3 + qc]; sd_w = gv[qv]; sd_v = gv[3 + qv];
      }
      tgt = n_tgt;
      load_step(step - 1);
    };
    auto q_take = [&](int g) {   // the contact wave's PRE of generation g
      const float *P = qpre + g * (PD_QPRE * 64) + lane;
      s.p = P[0 * 64]; s.r = P[1 * 64]; s.w = P[2 * 64]; s.v = P[3 * 64]; t0 = P[4 * 64]; f0 = P[5 * 64]; mask = __float_as_uint(P[6 * 64]); rc = P[7 * 64];
      Rr.a = P[8 * 64]; Rr.b = P[9 * 64]; Rr.c = P[10 * 64]; Rc.a = P[11 * 64]; Rc.b = P[12 * 64]; Rc.c = P[13 * 64];
      T.wb = P[14 * 64]; T.Iwb = P[15 * 64]; T.tb = P[16 * 64]; T.u = P[17 * 64]; T.w1 = P[18 * 64]; T.il = P[19 * 64]; T.r1 = P[20 * 64];
    };
    int g3 = a.nsteps > 0 ? (a.nsteps - 1) % PD_QGEN : 0;   // generation of the step at hand (step % PD_QGEN)
    if (a.nsteps > 0) {
      load_step(a.nsteps - 1);
      PD_WAIT_VMEM();
      q_ahead(a.nsteps - 1);
      pair_wait(sig + 1, 1);   // S: the state wave's step nsteps - 1
      q_take(g3);
    }
    STAMP_DECL;
    for (int step = a.nsteps - 1; step >= 0; --step) {
      gn.p += k.isv ? sd_p : 0.f; gn.r += sd_r; gn.w += k.isv ? sd_w : 0.f; gn.v += k.isv ? sd_v : 0.f;   // (zeros off a frame step)
      const size_t oc = (size_t)step * a.bs * m.nqd;
      const float *const rec_g = qgen + g3 * qgen_floats + 4 * nb, *const jc_g = qpre + PD_QGEN * (PD_QPRE * 64) + g3 * m.env_lds_jc;
      STAMP(0);
      // ---- adjoint of integrate_bodies, phase 1: the wrench adjoint
      QM3 aR;
      aR.a = aR.b = aR.c = 0.f;
      float adj_t0, adj_f0;
      q_integrate_adj_wrench(k, B, s, Rr, Rc, invIt, mask, a.dt, gn, T, adj_t0, adj_f0);
      adjf[bb * PD_W6 + (k.isv ? qc : 6)] = adj_t0; adjf[bb * PD_W6 + (k.isv ? 3 + qc : 6)] = adj_f0;  // (lane 3: the slot's pad float)
      pair_signal(sig, a.nsteps - step);  // A: records (staged one step ago) + wrench adjoints
      STAMP(1);
      if (k.isv) {
        float *o = a.g_res_f + (size_t)step * N * 6;  // adjoint of wp_add
        stg(o, boff_rf, NZ(adj_t0)); stg(o + 3, boff_rf, NZ(adj_f0));
      }
      // ---- phase 2 (needs nothing from the other wave)
      QAdj ga;
      q_integrate_adj_rest(k, B, s, Rr, It, t0, f0, a.dt, gn, T, ga, aR, g_inv_m, g_I, g_invI);
      STAMP(5);
      STAMP(6);   // (the joint hand-over records came with hand-over S of this step, taken one iteration ago)
      // ---- adjoint of eval_body_joints
      QAdj par;
      par.p = par.r = par.w = par.v = 0.f;
      float a_tgt = 0.f, a_act = 0.f, a_ke = 0.f, a_kd = 0.f;
      {
        const float *pr = rec_g + B.pidx * PD_REC, *pa = adjf + B.pidx * PD_W6;
        const float l_pp = pr[qv], qp = pr[3 + qc], l_wp = pr[7 + qv], l_vp = pr[10 + qv], l_rcp = pr[13 + qv], l_gt = pa[qv], l_gf = pa[3 + qv];
        const float pp = k.isv ? l_pp : 0.f, w_p = k.isv ? l_wp : 0.f, v_p = k.isv ? l_vp : 0.f, rc_par = k.isv ? l_rcp : 0.f;
        const float gp_t = k.isv ? l_gt : 0.f, gp_f = k.isv ? l_gf : 0.f;
        const QRev R = q_rev_load(k, jc_g + bb * PD_JC, qv);
        QAdj own, pj_;
        own.p = own.r = own.w = own.v = 0.f;
        QM3 aRj;
        aRj.a = aRj.b = aRj.c = 0.f;
        q_rev_adjoint(k, B, s, rc, pp, qp, w_p, v_p, rc_par, com_par0, com_par1, com_par2, p_pj0, p_pj1, p_pj2, pjc, R, tgt, ake, akd,
                      adj_t0, adj_f0, gp_t, gp_f, own, pj_, aRj, a_tgt, a_act, a_ke, a_kd);
        // a body without a joint (the FREE root, idle lanes) computed on the record of body pidx: dropped here
        if (B.joint) {
          ga.p += own.p; ga.r += own.r; ga.w += own.w; ga.v += own.v;
          par = pj_;
          aR.a += aRj.a; aR.b += aRj.b; aR.c += aRj.c;
        } else {
          a_tgt = 0.f; a_act = 0.f; a_ke = 0.f; a_kd = 0.f;
        }
      }
      STAMP(2);
      ga.r += q_rotm_adj(k, s.r, aR);
      {
        *cs_r = par.r; *cs_p = par.p; *cs_w = par.w; *cs_v = par.v;
        if (qc == 0) {
          if (B.type == PD_JOINT_REVOLUTE) { stg(a.g_refs + oc, boff_qd, NZ(a_tgt)); stg(a.g_torques + oc, boff_qd, NZ(a_act)); }
        }
        // the root's six dof columns are zero (a FREE joint reads no dof, integrator_euler.py:382): lanes 0-5 of the root's quad
        // and its neighbour write one each
        if (B.type == PD_JOINT_FREE) {
          stg(a.g_refs + oc + qc, boff_qd, 0.f); stg(a.g_torques + oc + qc, boff_qd, 0.f);
          if (qc < 2) { stg(a.g_refs + oc + 4 + qc, boff_qd, 0.f); stg(a.g_torques + oc + 4 + qc, boff_qd, 0.f); }
        }
      }
      g_ke += a_ke; g_kd += a_kd;
      STAMP(7);
      WAVE_SYNC();
      {  // children (own joint's contribution went into ga above): all LDS reads in flight together
        float cp[4], cr[4], cw[4], cv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float *src = cslot + qcz[j] * PD_ADJ;
          cp[j] = src[qv]; cr[j] = src[3 + qc]; cw[j] = src[7 + qv]; cv[j] = src[10 + qv];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { ga.p += k.isv ? cp[j] : 0.f; ga.r += cr[j]; ga.w += k.isv ? cw[j] : 0.f; ga.v += k.isv ? cv[j] : 0.f; }
      }
      for (int j = 4; j < m.max_children; ++j) {
        const int cid = (int)((q_children >> (8 * j)) & 0xffull);
        if (cid != 0xff) {
          const float *src = cslot + cid * PD_ADJ;
          ga.r += src[3 + qc];
          if (k.isv) { ga.p += src[qc]; ga.w += src[7 + qc]; ga.v += src[10 + qc]; }
        }
      }
      STAMP(3);
      // ---- the NEXT iteration's look-ahead: seeds / target requested, the contact wave's PRE(step - 1) taken (its LDS reads fly while this
      // wave waits for B; at step 0: PRE of state 0 again, unused)
      q_ahead(step - 1);
      g3 = g3 == 0 ? PD_QGEN - 1 : g3 - 1;
      pair_wait(sig + 1, step > 0 ? a.nsteps - step + 1 : a.nsteps);  // S: the state wave's step - 1 (at step 0 there is none: what is taken is not used)
      q_take(g3);
      STAMP(9);
      pair_wait(sig + 2, a.nsteps - step);  // B: contact adjoints are complete
      STAMP(8);
      {
        const float c_r = *ca_r, c_p = *ca_p, c_w = *ca_w, c_v = *ca_v;
        ga.r += c_r; ga.p += k.isv ? c_p : 0.f; ga.w += k.isv ? c_w : 0.f; ga.v += k.isv ? c_v : 0.f;
        *ca_r = 0.f; *ca_p = 0.f; *ca_w = 0.f; *ca_v = 0.f;
      }
      gn = ga;
      STAMP(4);
    }
    STAMP_FLUSH(a);
    auto q_seeds = [&](int fr) {  // dp_model.py:1264-1271
      const float *gp = a.adj_pos + ((size_t)fr * N + qidx) * 7, *gv = a.adj_vel + ((size_t)fr * N + qidx) * 6;
      const float sp = gp[qv], sr = gp[3 + qc], sw = gv[qv], sv = gv[3 + qv];
      gn.p += k.isv ? sp : 0.f; gn.r += sr; gn.w += k.isv ? sw : 0.f; gn.v += k.isv ? sv : 0.f;
    };
    if (a.frame_of_step[0] >= 0) q_seeds(a.frame_of_step[0]);  // seeds of state 0
    // ---- adjoint of eval_fk, in the lane-per-body layout: the running adjoint is transposed through LDS; state 0's records are the
    // contact wave's generation 0 (taken above: hand-over P of step 0)
    float *const rec0 = qgen + 4 * nb;
    WAVE_SYNC();
    if (qbody) {
      float *d = cacc + bb * PD_ADJ;
      d[3 + qc] = gn.r;
      if (k.isv) { d[qc] = gn.p; d[7 + qc] = gn.w; d[10 + qc] = gn.v; }
    }
    WAVE_SYNC();
    {
      BodyAdj gs = adj_zero();
      if (is_body) adj_add_from(gs, cacc + b * PD_ADJ);
      if (a.nsteps == 0) {
        for (int d = 0; d <= m.max_depth; ++d) {  // nothing staged yet: rebuild state 0
          if (is_body && c.depth == d) {
            BodyState s0 = fk_joint<JT>(c, a.q_init + (size_t)ec * m.nq + c.qstart, a.qd_init + (size_t)ec * m.nqd + c.qdstart, rec0);
            stage_record(rec0, b, s0, c.com);
          }
          WAVE_SYNC();
        }
      }
      for (int d = m.max_depth; d >= 0; --d) {
        if (is_body && c.depth == d) {
          for (int j = 0; j < m.max_children; ++j) {
            int cid = (int)((c.children >> (8 * j)) & 0xffull);
            if (cid != 0xff) adj_add_from(gs, cslot + cid * PD_ADJ);
          }
          BodyAdj pr_ = fk_joint_adj<JT, 1>(c, a.q_init + (size_t)ec * m.nq + c.qstart, a.qd_init + (size_t)ec * m.nqd + c.qdstart, rec0,
                                          gs, a.g_q_init + (size_t)ec * m.nq + c.qstart, a.g_qd_init + (size_t)ec * m.nqd + c.qdstart);
          adj_store(cslot + b * PD_ADJ, pr_);
        }
        WAVE_SYNC();
      }
    }
    if (qbody) {
      if (qc == 0) a.g_inv_mass[qidx] = NZ(g_inv_m);
      if (k.isv) {
        float *gi = a.g_inertia + qidx * 9 + qc * 3, *gj = a.g_inv_inertia + qidx * 9 + qc * 3;
        gi[0] = NZ(g_I.a); gi[1] = NZ(g_I.b); gi[2] = NZ(g_I.c);
        gj[0] = NZ(g_invI.a); gj[1] = NZ(g_invI.b); gj[2] = NZ(g_invI.c);
      }
      const size_t og = (size_t)ec * m.nqd + B.qdstart;
      if (B.type == PD_JOINT_REVOLUTE && qc == 0) { a.g_ke[og] = NZ(g_ke); a.g_kd[og] = NZ(g_kd); }
      if (B.type == PD_JOINT_FREE) {
        a.g_ke[og + qc] = 0.f; a.g_kd[og + qc] = 0.f;
        if (qc < 2) { a.g_ke[og + 4 + qc] = 0.f; a.g_kd[og + 4 + qc] = 0.f; }
      }
    }
    return;
  }
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

  BodyAdj gn = adj_zero();  // adjoint of state step+1
  BodyState s;
  s.p = V3(0, 0, 0); s.r = Q4(0, 0, 0, 1); s.w = V3(0, 0, 0); s.v = V3(0, 0, 0);

  // The stored state, wrench and controls of the NEXT iteration (step - 1) are software-prefetched.
  const unsigned boff = (unsigned)idx * 4u, boff_qd = (unsigned)((size_t)ec * m.nqd + c.qdstart) * 4u;  // per-lane byte offsets
  const bool zero_by_lanes = m.jtype[0] == PD_JOINT_FREE && nb >= 7;  // see the control-gradient stores below
  const unsigned boff_zero = (unsigned)((size_t)ec * m.nqd + m.qdstart[0] + (b >= 1 && b <= 6 ? b - 1 : 0)) * 4u;
  float4 n_s[PD_TRAJ_G];
  float n_tgt[ND], n_act[ND];
  int n_fr = -1;  // frame seeded into state step + 1 (or -1), fetched with the state
  auto load_step = [&](int step) {
    const int sc = __builtin_amdgcn_readfirstlane(step >= 0 ? step : 0);  // keeps the address arithmetic scalar
    n_fr = a.frame_of_step[sc + 1];  // (a scalar load here couples the LDS waits to it through lgkmcnt: +2 %)
    const float *tj = a.ws + (size_t)sc * (PD_TRAJ_G * 4) * N;
#pragma unroll
    for (int g = 0; g < PD_TRAJ_G; ++g) n_s[g] = ldg4(tj + (size_t)(4 * g) * N, boff * 4u);
    const size_t o = (size_t)sc * a.bs * m.nqd;
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      bool on = k < ndof;
      n_tgt[k] = on ? ldg(a.refs + o + k, boff_qd) : 0.f;
      n_act[k] = on ? ldg(a.torques + o + k, boff_qd) : 0.f;
    }
  };
  if (a.nsteps > 0) load_step(a.nsteps - 1);
  STAMP_DECL;
  for (int step = a.nsteps - 1; step >= 0; --step) {
    PD_WAIT_VMEM();
    {  // seeds of state step+1 (dp_model.py:1264-1271)
      const int fr = n_fr;
      if (fr >= 0) {
        add_frame_seeds(a, fr, N, idx, gn);
      }
    }
    s.r = Q4(n_s[0].x, n_s[0].y, n_s[0].z, n_s[0].w); s.w = V3(n_s[1].x, n_s[1].y, n_s[1].z);
    s.p = V3(n_s[2].x, n_s[2].y, n_s[2].z); s.v = V3(n_s[1].w, n_s[2].w, n_s[3].x);
    v3 t0 = V3(n_s[3].y, n_s[3].z, n_s[3].w), f0 = V3(n_s[4].x, n_s[4].y, n_s[4].z);
    const unsigned clamp_mask = __float_as_uint(n_s[4].w);  // which of (w, v) the forward pass clamped in this step
    float tgt[ND], act[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) { tgt[k] = n_tgt[k]; act[k] = n_act[k]; }
    const size_t oc = (size_t)step * a.bs * m.nqd;  // uniform part; the lane part is boff_qd
    load_step(step - 1);
    // unsplit kernels replay the forward hit list inline further down: fetch its length now, far ahead of its use
    int *lg = a.hitlog + ((size_t)step * a.bs + ec) * PD_HITLOG;
    const int log_cnt = (!SPLIT && env_ok) ? lg[0] : 0;
    float Rm[9];
    rotm(s.r, Rm);  // the body's rotation as a matrix: shared by the staging and the adjoint of integrate_bodies
    const v3 rc = mat_vec(Rm, c.com);
    if (is_body) stage_record(rec, cull, b, s, rc, Rm);
    STAMP(0);
    // ---- adjoint of integrate_bodies
    BodyAdj ga = adj_zero();
    v3 adj_t0 = V3(0, 0, 0), adj_f0 = adj_t0;
    float aR[9];  // matrix adjoint of Rm, summed over integrate_bodies and (revolute, split) the joint; converted after both
#pragma unroll
    for (int k = 0; k < 9; ++k) aR[k] = 0.f;
    if (SPLIT && EARLY) {
      integrate_adj2(m, c, s, Rm, clamp_mask, t0, f0, inv_m, I, invI, a.dt, gn, ga, aR, g_inv_m, g_I, g_invI, [&](v3 t, v3 f) {
        adj_t0 = t; adj_f0 = f;
        if (is_body) {
          float *o = adjf + b * PD_W6;
          o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = f.x; o[4] = f.y; o[5] = f.z;
        }
        pair_signal(sig, a.nsteps - step);  // A: records + wrench adjoints are staged
      });
      if (is_body) {
        float *o = a.g_res_f + (size_t)step * N * 6;  // adjoint of wp_add
        stg2(o, boff * 6u, make_float2(NZ(adj_t0.x), NZ(adj_t0.y))); stg2(o + 2, boff * 6u, make_float2(NZ(adj_t0.z), NZ(adj_f0.x)));
        stg2(o + 4, boff * 6u, make_float2(NZ(adj_f0.y), NZ(adj_f0.z)));
      }
      STAMP(1);
      pair_wait(sig + 1, a.nsteps - step);  // the contact wave's joint hand-over records
    } else {
    integrate_adj(m, c, s, Rm, clamp_mask, t0, f0, inv_m, I, invI, a.dt, gn, ga, aR, adj_t0, adj_f0, g_inv_m, g_I, g_invI);
    if (is_body) {
      float *f = adjf + b * PD_W6;
      f[0] = adj_t0.x; f[1] = adj_t0.y; f[2] = adj_t0.z; f[3] = adj_f0.x; f[4] = adj_f0.y; f[5] = adj_f0.z;
    }
    STAMP(1);
    // A: hand records + wrench adjoints to the contact wave FIRST -- at 4096 envs it is the later wave of the pair, and whatever
    // this wave does ahead of the signal delays it: with the remove_nan selects and the g_res_f stores before the signal the
    // adjoint took 0.324 ms, behind it 0.312 (same-box A/B) -- then the stores, then take its joint hand-over records
    if (SPLIT) pair_signal(sig, a.nsteps - step);
    if (is_body) {
      float *o = a.g_res_f + (size_t)step * N * 6;  // adjoint of wp_add
      stg2(o, boff * 6u, make_float2(NZ(adj_t0.x), NZ(adj_t0.y))); stg2(o + 2, boff * 6u, make_float2(NZ(adj_t0.z), NZ(adj_f0.x)));
      stg2(o + 4, boff * 6u, make_float2(NZ(adj_f0.y), NZ(adj_f0.z)));
    }
    if (SPLIT) {
      pair_wait(sig + 1, a.nsteps - step);  // the contact wave's joint hand-over records
    } else {
      WAVE_SYNC();
    }
    }
    // ---- adjoint of eval_body_joints (runs while the contact wave sweeps)
    BodyAdj par = adj_zero();
    float a_tgt[ND], a_act[ND], a_ke[ND], a_kd[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) { a_tgt[k] = 0.f; a_act[k] = 0.f; a_ke[k] = 0.f; a_kd[k] = 0.f; }
    if (is_body && c.type != PD_JOINT_FREE) {
      v3 gp_t = V3(0, 0, 0), gp_f = gp_t;
      if ((SPLIT && pd_parented(JT)) || c.parent >= 0) { gp_t = ld3(adjf + c.parent * PD_W6); gp_f = ld3(adjf + c.parent * PD_W6 + 3); }
      if (SPLIT)  // revolute only: the state-only half comes from the contact wave
      {
        const RevCache R = rev_cache_load(jc + (step & 1) * m.env_lds_jc + b * PD_JC);
        rev_adjoint<pd_parented(JT)>(m, c, s, rc, rec, R, tgt[0], ke[0], kd[0], adj_t0, adj_f0, gp_t, gp_f, ga, par, aR, a_tgt[0], a_act[0], a_ke[0], a_kd[0]);
      }
      else
        joint_adj<JT>(m, c, s, rc, rec, tgt, act, ke, kd, adj_t0, adj_f0, gp_t, gp_f, ga, par, a_tgt, a_act, a_ke, a_kd);
    }
    rotm_adj(s.r, aR, ga.r);
    if (is_body) {
      adj_store(cslot + b * PD_ADJ, par);
#pragma unroll
      for (int k = 0; k < ND; ++k) {
        if (k < ndof) { stg(a.g_refs + oc + k, boff_qd, NZ(a_tgt[k])); stg(a.g_torques + oc + k, boff_qd, NZ(a_act[k])); }
        g_ke[k] += a_ke[k]; g_kd[k] += a_kd[k];
      }
      // the root's six dof columns are zero (a FREE joint reads no dof, integrator_euler.py:382).  Lanes 1..6 write one each
      // beside their own entry -- two vector-memory instructions per array and step instead of seven: with eight waves
      // on a CU it is the NUMBER of such instructions (one per ~27 cycles across the CU), not their bytes, that the loop feels
      if (zero_by_lanes) {
        if (b >= 1 && b <= 6) { stg(a.g_refs + oc, boff_zero, 0.f); stg(a.g_torques + oc, boff_zero, 0.f); }
      } else if (c.type == PD_JOINT_FREE) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { stg(a.g_refs + oc + k, boff_qd, 0.f); stg(a.g_torques + oc + k, boff_qd, 0.f); }
      }
    }
    STAMP(2);
    WAVE_SYNC();
    {  // first four children with all LDS reads in flight together (packed sums), then any further ones
      const float *const src[4] = {cslot + cz[0] * PD_ADJ, cslot + cz[1] * PD_ADJ, cslot + cz[2] * PD_ADJ, cslot + cz[3] * PD_ADJ};
      adj_add_from_n(ga, src);
    }
    for (int k = 4; k < m.max_children; ++k) {
      int cid = (int)((c.children >> (8 * k)) & 0xffull);
      if (is_body && cid != 0xff) adj_add_from(ga, cslot + cid * PD_ADJ);
    }
    STAMP(3);
    if (SPLIT) {
      pair_wait(sig + 2, a.nsteps - step);  // B: contact adjoints are complete
    } else {
      const bool replay = __ballot(log_cnt < 0) == 0ull;
      int log_n_unused;
      const int cnt = log_cnt;
      sweep_contacts<SEGW, PD_ADJ, PD_ADJ, !SPLIT>(m, tabs, c, is_body ? cull[b] : make_float4(0.f, 0.f, 1.f, 0.f), rec, cull, list, hits, slot,
                                                   cacc, is_body, env_ok, seg, l, replay ? lg : nullptr, replay ? cnt : PD_NO_REPLAY,
                                                   log_n_unused, contact_hit STAMP_PASS);
      WAVE_SYNC();
    }
    if (is_body) {
      float *d = cacc + b * PD_ADJ;
      const float *const src[1] = {d};
      adj_add_from_n(ga, src);
#pragma unroll
      for (int k = 0; k < PD_ADJ; ++k) d[k] = 0.f;
    }
    gn = ga;
    if (!SPLIT) WAVE_SYNC();
    STAMP(4);
  }
  STAMP_FLUSH(a);
  {  // seeds of state 0
    int fr = a.frame_of_step[0];
    if (fr >= 0) {
      add_frame_seeds(a, fr, N, idx, gn);
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
      BodyAdj par = fk_joint_adj<JT, 1>(c, a.q_init + (size_t)ec * m.nq + c.qstart, a.qd_init + (size_t)ec * m.nqd + c.qdstart, rec,
                                     gn, a.g_q_init + (size_t)ec * m.nq + c.qstart, a.g_qd_init + (size_t)ec * m.nqd + c.qdstart);
      adj_store(cslot + b * PD_ADJ, par);
    }
    WAVE_SYNC();
  }
  if (is_body) {
    a.g_inv_mass[idx] = NZ(g_inv_m);
#pragma unroll
    for (int k = 0; k < 9; ++k) { a.g_inertia[idx * 9 + k] = NZ(g_I[k]); a.g_inv_inertia[idx * 9 + k] = NZ(g_invI[k]); }
    const size_t og = (size_t)ec * m.nqd + c.qdstart;
#pragma unroll
    for (int k = 0; k < ND; ++k)
      if (k < ndof) { a.g_ke[og + k] = NZ(g_ke[k]); a.g_kd[og + k] = NZ(g_kd[k]); }
    if (c.type == PD_JOINT_FREE) {
#pragma unroll
      for (int k = 0; k < 6; ++k) { a.g_ke[og + k] = 0.f; a.g_kd[og + k] = 0.f; }
    }
  }
}

// =============================================================================================
// Role-split adjoint rollout (revolute-only robots, i.e. the headline Laikago path): THREE waves per group of 64/SEGW envs,
// one of each per SIMD (12 waves per workgroup; <= 168 VGPRs each so that three fit a SIMD):
//   I  integrate wave : owns the running state adjoint gn.  Per step: seeds, adjoint of integrate_bodies in two phases --
//                       the short path to the wrench adjoint (adj_t0, adj_f0) first, published to LDS (hand-over A), then
//                       the rest (state adjoint, inertia / inverse-mass gradients) while the other two waves work; finally
//                       it gathers the joint wave's results (own + children, hand-over J) and the contact sums (hand-over C)
//   J  joint wave     : the whole adjoint of eval_body_joints.  Before A it recomputes the state-only half of its joint
//                       (rev_forward) from the stored trajectory, prefetched a step ahead, and keeps it in registers; after A
//                       it reads the wrench adjoints and the staged records, runs rev_adjoint, publishes (own, parent)
//                       contributions, then writes the control gradients off the critical path
//   C  contact wave   : adjoint of eval_body_contacts for the logged hits, one lane per hit (as in the 2-role kernel)
// Per step the critical chain is  phase 1 (I) -> max(rev_adjoint (J), contacts (C), phase 2 (I)) -> gather (I)  instead of
// integrate_adj + rev_adjoint + gather on one wave.  Arithmetic and summation order are those of the 2-role kernel.
#define PD_BWD3_BOUNDS(roles) ((roles) * PD_BWAVES * 64)
// ROLES = 3: I, C, J waves (<= 168 VGPRs each).  ROLES = 2: the integrate wave also replays the contacts (between its phase 2
// and the wait for the joint wave) -- compound-joint robots, whose joint adjoint needs more than 168 registers but whose
// box contacts are a handful of points: two waves per env group, <= 256 VGPRs each.
template <int SEGW, int JT, int ROLES>
__global__ __launch_bounds__(PD_BWD3_BOUNDS(ROLES)) void k_rollout_bwd3(PdDevModel m, RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int EPW = Seg<SEGW>::EPW;
  constexpr int ND = (JT & PD_JT_COMPOUND) ? 3 : 1;
  static_assert(ROLES == 2 || ROLES == 3, "two or three roles");
  const int bw = (int)blockDim.x / (64 * ROLES);  // env groups per workgroup (host's choice per launch)
  const int lane = threadIdx.x & 63, wave_id = threadIdx.x >> 6;
  // 0: I (+ contacts when ROLES == 2), 1: C, 2: J   (wave-uniform)
  const int role = ROLES == 3 ? wave_id / bw : (wave_id / bw ? 2 : 0), wave = wave_id % bw;
  [[maybe_unused]] const int knock_role = role;
  const int seg = lane / SEGW, l = lane % SEGW;
  const int env = (blockIdx.x * bw + wave) * EPW + seg;
  const bool env_ok = env < a.bs;
  const int ec = env_ok ? env : 0;  // clamped env for safe addressing
  const bool is_body = env_ok && l < m.nb;
  const int b = l < m.nb ? l : m.nb - 1;
  const int nb = m.nb, N = a.bs * nb;

  SweepTables tabs;
  // ROLES == 2: the (small) contact tables are copied into LDS, the inline replay then has no exposed global loads
  float *scratch = lds_setup<ROLES == 2>(m, smem, tabs, wave * EPW + seg, m.env_lds_bwd3);
  // staged records and cull vectors: TWO generations, by step parity -- the integrate wave stages step - 1 while the other
  // waves still work on step (the joint wave then starts the state-only half of step - 1 without waiting for anybody)
  float4 *const cull0 = (float4 *)scratch;
  float *const rec0 = scratch + 8 * nb;
  float4 *cull = cull0;
  float *rec = rec0;
  auto select_step = [&](int step) { rec = rec0 + (step & 1) * nb * PD_REC; cull = cull0 + (step & 1) * nb; };
  // cslot: nb + 1 records, the last one stays zero and stands in for "no child" (the gather then needs no predicates)
  float *adjf = rec0 + 2 * nb * PD_REC, *cslot = adjf + nb * PD_W6, *oslot = cslot + (nb + 1) * PD_ADJ;
  float *cacc = oslot + nb * PD_ADJ;
  // gacc: the integrate wave's running gradients of body_inertia / body_inv_inertia (2 x 9 per body) followed by the body's
  // inertia and inverse inertia (2 x 9); they live here, not in registers, so that the wave stays within the 168 VGPRs
  // three waves per SIMD allow (stride 37: odd, conflict-free)
  float *gacc = cacc + nb * PD_ADJ;
  int *list = (int *)(gacc + (nb + 1) * PD_GACC), *hits = list + m.list_cap;
  float *slot = (float *)(hits + PD_HIT_CAP_TILES * SEGW);
  // hand-over words (step counters) at the end of the wave's first env: [0] A (I -> J, C)   [1] J (J -> I)   [2] C (C -> I)
  // [3] S (I -> J: records staged; joint mixes other than revolute-only)
  int *sig = (int *)(scratch - (size_t)seg * m.env_lds_bwd3 + m.env_lds_bwd3 - 4);
  if (role == 0) {
    if (lane == 0) { sig[0] = 0; sig[1] = 0; sig[2] = 0; sig[3] = 0; }
    if (is_body) {
#pragma unroll
      for (int k = 0; k < PD_ADJ; ++k) { cacc[b * PD_ADJ + k] = 0.f; oslot[b * PD_ADJ + k] = 0.f; cslot[b * PD_ADJ + k] = 0.f; }
    }
    if (l == 0) {
#pragma unroll
      for (int k = 0; k < PD_ADJ; ++k) cslot[nb * PD_ADJ + k] = 0.f;
    }
  }
  __syncthreads();
  PD_KNOCK_EXIT(false, knock_role);

  const size_t idx = (size_t)ec * nb + b;
  const unsigned boff = (unsigned)idx * 4u;  // per-lane byte offset of this body's float

  // (each role loads the per-body constants itself: what one role needs is dead in the others, and a shared load ahead
  // of the role branch kept all of them live -- 68 spilled VGPRs at the 168 three waves per SIMD allow)
  if (ROLES == 3 && role == 1) {
    // ---- C: contact wave
    BodyConst c = load_body_const(m, b, ec);
#pragma unroll
    for (int u = 0; u < 4; ++u) c.small_e[u] = m.small_tiles[u * 64 + (l < 64 ? l : 0)];
    auto contact_hit = [&](const float *r, float4 P, float4 mat, float *out) {
      const int pb = (int)(r - rec) / PD_REC;
      BodyAdj o = adj_zero();
      contact_point_adj(r, cull[pb], P, mat, ld3(adjf + pb * PD_W6), ld3(adjf + pb * PD_W6 + 3), o);
      adj_store(out, o);
    };
    STAMP_DECL;
    const int lq = l < PD_HITLOG - 1 ? l : PD_HITLOG - 2;
    const unsigned boff_lg = (unsigned)ec * (PD_HITLOG * 4u);
    auto load_log = [&](int step, int &cnt, int &e) {
      const int *lg = a.hitlog + (size_t)__builtin_amdgcn_readfirstlane(step > 0 ? step : 0) * a.bs * PD_HITLOG;
      cnt = __float_as_int(ldg((const float *)lg, boff_lg)); e = __float_as_int(ldg((const float *)lg + 1, boff_lg + (unsigned)lq * 4u));
    };
    auto fetch_point = [&](int cnt, int &e, float4 &P, float4 &M) {  // entries past the count are uninitialised memory
      if (!(env_ok && l < cnt && l < PD_HITLOG - 1)) e = 0;
      P = m.pts[e & 0xffff]; M = m.materials[(e >> 16) & 0xff];
    };
    int cnt_c = 0, e_c = 0, cnt_n = 0, e_n = 0;
    float4 P_c, M_c;
    if (a.nsteps > 0) {
      load_log(a.nsteps - 1, cnt_c, e_c);
      load_log(a.nsteps - 2, cnt_n, e_n);
      fetch_point(cnt_c, e_c, P_c, M_c);
    }
    for (int step = a.nsteps - 1; step >= 0; --step) {
      PD_WAIT_VMEM();
      select_step(step);
      float4 P_n, M_n;
      int cnt_n2, e_n2;
      fetch_point(cnt_n, e_n, P_n, M_n);
      load_log(step - 2, cnt_n2, e_n2);
      const bool fast = __ballot(env_ok && (cnt_c < 0 || cnt_c > SEGW)) == 0ull;  // wave-uniform
      const int nh = fast && env_ok ? cnt_c : 0;
      STAMP(7);
      pair_wait(sig, a.nsteps - step);  // A: records, cull vectors and wrench adjoints of this step are staged
      STAMP(9);
      if (fast) {
        float out[PD_ADJ];
#pragma unroll
        for (int i = 0; i < PD_ADJ; ++i) out[i] = 0.f;
        const int pb = l < nh ? (e_c >> 24) & 0x3f : -2;
        if (l < nh) contact_hit(rec + pb * PD_REC, P_c, M_c, out);
        STAMP(10);
        bool last;
        seg_run_sum<PD_ADJ>(out, pb, l, nh, last);
        if (last) {  // cacc is zero (its owner clears it after reading) and a body has one run
#pragma unroll
          for (int i = 0; i < PD_ADJ; ++i) cacc[pb * PD_ADJ + i] = out[i];
        }
        STAMP(11);
      } else {
        float4 cv = make_float4(0.f, 0.f, 1.f, 0.f);
        if (is_body) cv = cull[b];
        int *lg = a.hitlog + ((size_t)step * a.bs + ec) * PD_HITLOG;
        const bool replay = __ballot(env_ok && cnt_c < 0) == 0ull;  // -1: the list did not fit the log, cull again (whole wave)
        int log_n_unused;
        sweep_contacts<SEGW, PD_ADJ, PD_ADJ, false>(m, tabs, c, cv, rec, cull, list, hits, slot, cacc, is_body, env_ok, seg, l,
                                                    replay ? lg : nullptr, replay ? (env_ok ? cnt_c : 0) : PD_NO_REPLAY, log_n_unused,
                                                    contact_hit STAMP_PASS);
      }
      STAMP(12);
      pair_signal(sig + 2, a.nsteps - step);  // C: contact adjoints are complete
      cnt_c = cnt_n; e_c = e_n; P_c = P_n; M_c = M_n;
      cnt_n = cnt_n2; e_n = e_n2;
    }
    STAMP_FLUSH(a);
    return;
  }

  if (role == 2) {
    // ---- J: joint wave
    STAMP_DECL;
    const BodyConst c = load_body_const(m, b, ec);
    const unsigned boff_qd = (unsigned)((size_t)ec * m.nqd + c.qdstart) * 4u;
    const bool has_par = c.parent >= 0;
    // CLONE3 (round 6; compound-only plain models, the any-joint-mix loop below): an idle lane clones its env's last body -- same loads,
    // same arithmetic, the same values to the same LDS / global addresses -- so the loop has no idle-lane regions; the joint adjoint keeps
    // ONE divergent region per half (the FREE root sits it out) without type or parent tests inside (joint_adj_prep / _apply: ALL).
    constexpr bool CLONE3 = JT == PD_JT_COMPOUND;
    const bool wr = CLONE3 || is_body, gw = CLONE3 ? env_ok : is_body;
    const int ndof = c.type == PD_JOINT_REVOLUTE ? 1 : (c.type == PD_JOINT_COMPOUND ? 3 : 0);
    float ke[ND], kd[ND], g_ke[ND], g_kd[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      const bool on = wr && k < ndof;
      ke[k] = on ? a.target_ke[(size_t)ec * m.nqd + c.qdstart + k] : 0.f;
      kd[k] = on ? a.target_kd[(size_t)ec * m.nqd + c.qdstart + k] : 0.f;
      g_ke[k] = 0.f; g_kd[k] = 0.f;
    }
    // the root's six dof columns of g_refs / g_torques are zero (a FREE joint reads no dof, integrator_euler.py:382): lanes
    // 1..6 write one each beside their own entries -- two store instructions per array and step instead of seven
    const bool root_free = m.jtype[0] == PD_JOINT_FREE;
    const bool zero_by_lanes = root_free && nb >= 7;
    const unsigned boff_zero = (unsigned)((size_t)ec * m.nqd + m.qdstart[0] + (b >= 1 && b <= 6 ? b - 1 : 0)) * 4u;
    auto store_controls = [&](int step, const float *a_tgt, const float *a_act, const float *a_ke, const float *a_kd) {
      const size_t oc = (size_t)__builtin_amdgcn_readfirstlane(step) * a.bs * m.nqd;
      if constexpr (CLONE3) {  // every non-FREE joint has ND dofs: one region for the six stores
        if (gw && ndof > 0) {
#pragma unroll
          for (int k = 0; k < ND; ++k) { stg(a.g_refs + oc + k, boff_qd, NZ(a_tgt[k])); stg(a.g_torques + oc + k, boff_qd, NZ(a_act[k])); }
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) { g_ke[k] += a_ke[k]; g_kd[k] += a_kd[k]; }
      } else {
#pragma unroll
      for (int k = 0; k < ND; ++k) {
        if (is_body && k < ndof) { stg(a.g_refs + oc + k, boff_qd, NZ(a_tgt[k])); stg(a.g_torques + oc + k, boff_qd, NZ(a_act[k])); }
        g_ke[k] += a_ke[k]; g_kd[k] += a_kd[k];
      }
      }
      if (zero_by_lanes) {
        if (gw && b >= 1 && b <= 6 && l < nb) { stg(a.g_refs + oc, boff_zero, 0.f); stg(a.g_torques + oc, boff_zero, 0.f); }
      } else if (is_body && c.type == PD_JOINT_FREE) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { stg(a.g_refs + oc + k, boff_qd, 0.f); stg(a.g_torques + oc + k, boff_qd, 0.f); }
      }
    };
    if constexpr (JT == PD_JT_REVOLUTE) {
      const bool rev = is_body && c.type == PD_JOINT_REVOLUTE;
      const unsigned boff_p = (unsigned)((size_t)ec * nb + (has_par ? c.parent : b)) * 4u;
      // stored pose of this lane's body (q, w) and of its parent (p, q, w), and the controls, one iteration ahead
      float4 pose[5];
      float tgt_n = 0.f, act_n = 0.f;
#pragma unroll
      for (int k = 0; k < 5; ++k) pose[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      auto load_next = [&](int step) {
        const int sc = __builtin_amdgcn_readfirstlane(step > 0 ? step : 0);
        const float *tj = a.ws + (size_t)sc * (PD_TRAJ_G * 4) * N;
        if (rev) {
          pose[0] = ldg4(tj, boff * 4u); pose[1] = ldg4(tj + (size_t)4 * N, boff * 4u);
          pose[2] = ldg4(tj, boff_p * 4u); pose[3] = ldg4(tj + (size_t)4 * N, boff_p * 4u); pose[4] = ldg4(tj + (size_t)8 * N, boff_p * 4u);
          const size_t o = (size_t)sc * a.bs * m.nqd;
          tgt_n = ldg(a.refs + o, boff_qd); act_n = ldg(a.torques + o, boff_qd);
        }
      };
      if (a.nsteps > 0) load_next(a.nsteps - 1);
      for (int step = a.nsteps - 1; step >= 0; --step) {
        PD_WAIT_VMEM();
        const qt q_c = Q4(pose[0].x, pose[0].y, pose[0].z, pose[0].w), qp = Q4(pose[2].x, pose[2].y, pose[2].z, pose[2].w);
        const v3 w_c = V3(pose[1].x, pose[1].y, pose[1].z), pp = V3(pose[4].x, pose[4].y, pose[4].z), w_p = V3(pose[3].x, pose[3].y, pose[3].z);
        const float tgt = tgt_n, act = act_n;
        RevCache R = rev_forward<pd_parented(JT)>(m, c, q_c, w_c, pp, qp, w_p, tgt, act, ke[0], kd[0]);
        STAMP(8);
        load_next(step - 1);  // nothing loaded here is touched before the next PD_WAIT_VMEM
        STAMP(7);
        pair_wait(sig, a.nsteps - step);  // A: wrench adjoints and records of this step are staged
        STAMP(9);
        select_step(step);
        BodyAdj own = adj_zero(), par = adj_zero();
        float a_tgt[1] = {0.f}, a_act[1] = {0.f}, a_ke[1] = {0.f}, a_kd[1] = {0.f};
        if (rev) {
          // one batch of LDS reads: own wrench adjoint, the parent's, and what rev_adjoint needs beyond the prefetched pose
          const float *r = rec + b * PD_REC, *prec = rec + (has_par ? c.parent : b) * PD_REC;
          const v3 gc_t = ld3(adjf + b * PD_W6), gc_f = ld3(adjf + b * PD_W6 + 3);
          v3 gp_t = V3(0, 0, 0), gp_f = gp_t;
          if (has_par) { gp_t = ld3(adjf + c.parent * PD_W6); gp_f = ld3(adjf + c.parent * PD_W6 + 3); }
          BodyState s;
          s.p = ld3(r); s.r = q_c; s.w = w_c; s.v = ld3(r + 10);
          const v3 rc_c = ld3(r + 13), v_p = ld3(prec + 10), rc_par = ld3(prec + 13);
          float aR[9];
#pragma unroll
          for (int k = 0; k < 9; ++k) aR[k] = 0.f;
          rev_adjoint_core<pd_parented(JT)>(m, c, s, rc_c, pp, qp, w_p, v_p, rc_par, R, tgt, ke[0], kd[0], gc_t, gc_f, gp_t, gp_f, own, par, aR, a_tgt[0], a_act[0],
                                            a_ke[0], a_kd[0]);
          rotm_adj(s.r, aR, own.r);
        }
        if (is_body) { adj_store(cslot + b * PD_ADJ, par); adj_store(oslot + b * PD_ADJ, own); }
        STAMP(10);
        pair_signal(sig + 1, a.nsteps - step);  // J: (own, parent) contributions are complete
        store_controls(step, a_tgt, a_act, a_ke, a_kd);  // control gradients, off the critical path
        STAMP(11);
      }
    } else {
      // any joint mix: the whole joint adjoint after hand-over A, state from the staged records; controls a step ahead
      float n_tgt[ND], n_act[ND];
      auto load_next = [&](int step) {
        const size_t o = (size_t)__builtin_amdgcn_readfirstlane(step > 0 ? step : 0) * a.bs * m.nqd;
#pragma unroll
        for (int k = 0; k < ND; ++k) {
          const bool on = CLONE3 || (is_body && k < ndof);  // (CLONE3: unconditional; the root reads the first of its own dofs and never uses them)
          n_tgt[k] = on ? ldg(a.refs + o + k, boff_qd) : 0.f;
          n_act[k] = on ? ldg(a.torques + o + k, boff_qd) : 0.f;
        }
      };
      if (a.nsteps > 0) load_next(a.nsteps - 1);  // (a zero-step rollout has no controls: null pointers)
      for (int step = a.nsteps - 1; step >= 0; --step) {
        PD_WAIT_VMEM();
        float tgt[ND], act[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) { tgt[k] = n_tgt[k]; act[k] = n_act[k]; }
        load_next(step - 1);
        STAMP(7);
        // S: the records of this step are staged -- the integrate wave does that during the PREVIOUS step, before it waits
        // for this wave -- so the state-only half of the joint adjoint starts right after the previous step's hand-over J
        pair_wait(sig + 3, a.nsteps - step);
        select_step(step);
        BodyState s;
        s.p = V3(0, 0, 0); s.r = Q4(0, 0, 0, 1); s.w = V3(0, 0, 0); s.v = V3(0, 0, 0);
        JointPrep P;
        const bool jointed = wr && c.type != PD_JOINT_FREE;
        if (jointed) {
          const float *r = rec + b * PD_REC;
          s.p = ld3(r); s.r = ld4(r + 3); s.w = ld3(r + 7); s.v = ld3(r + 10);
          joint_adj_prep<JT, pd_parented(JT), CLONE3>(m, c, s, ld3(r + 13), rec, tgt, act, ke, kd, P);
        }
        STAMP(8);
        pair_wait(sig, a.nsteps - step);  // A: the wrench adjoints of this step are staged
        __builtin_amdgcn_s_setprio(PD_PRIO_CRITICAL);  // the integrate wave will wait for these contributions (quad 8192: 1.43 -> 1.40 ms)
        STAMP(9);
        BodyAdj own = adj_zero(), par = adj_zero();
        float a_tgt[ND], a_act[ND], a_ke[ND], a_kd[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) { a_tgt[k] = 0.f; a_act[k] = 0.f; a_ke[k] = 0.f; a_kd[k] = 0.f; }
        if (jointed) {
          const v3 gc_t = ld3(adjf + b * PD_W6), gc_f = ld3(adjf + b * PD_W6 + 3);
          v3 gp_t = V3(0, 0, 0), gp_f = gp_t;
          if (CLONE3 || has_par) { gp_t = ld3(adjf + c.pidx * PD_W6); gp_f = ld3(adjf + c.pidx * PD_W6 + 3); }  // (plain model: every jointed body hangs on one)
          joint_adj_apply<JT, pd_parented(JT), CLONE3>(m, c, s, P, tgt, act, ke, kd, gc_t, gc_f, gp_t, gp_f, own, par, a_tgt, a_act, a_ke, a_kd);
        }
        if (wr) { adj_store(cslot + b * PD_ADJ, par); adj_store(oslot + b * PD_ADJ, own); }
        STAMP(10);
        pair_signal(sig + 1, a.nsteps - step);  // J: (own, parent) contributions are complete
        __builtin_amdgcn_s_setprio(0);
        store_controls(step, a_tgt, a_act, a_ke, a_kd);
        STAMP(11);
      }
    }
    if (is_body) {
      const size_t og = (size_t)ec * m.nqd + c.qdstart;
#pragma unroll
      for (int k = 0; k < ND; ++k)
        if (k < ndof) { a.g_ke[og + k] = NZ(g_ke[k]); a.g_kd[og + k] = NZ(g_kd[k]); }
      if (c.type == PD_JOINT_FREE) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { a.g_ke[og + k] = 0.f; a.g_kd[og + k] = 0.f; }
      }
    }
    STAMP_FLUSH(a);
    return;
  }

  // ---- I: integrate wave.  The loop needs the centre of mass and the child list only; everything else of the per-body
  // constants is loaded after the loop, for the adjoint of eval_fk (nothing of it is live across the steps)
  BodyConst c;
  if (ROLES == 2) {  // the inline contact replay (and its re-cull fallback) needs the cull constants too
    c = load_body_const(m, b, ec);
#pragma unroll
    for (int u = 0; u < 4; ++u) c.small_e[u] = m.small_tiles[u * 64 + (l < 64 ? l : 0)];
  } else {
    c.com = ld3(m.com + b * 3); c.children = m.children[b];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int cid = (int)((c.children >> (8 * k)) & 0xffull);
      c.child[k] = cid == 0xff ? -1 : cid;
    }
  }
  auto contact_hit = [&](const float *r, float4 P, float4 mat, float *out) {  // ROLES == 2 only
    const int pb = (int)(r - rec) / PD_REC;
    BodyAdj o = adj_zero();
    contact_point_adj<true>(r, cull[pb], P, mat, ld3(adjf + pb * PD_W6), ld3(adjf + pb * PD_W6 + 3), o);
    adj_store(out, o);
  };
  // inertia and inverse inertia are read from LDS where they are used (9 + 9 registers less across the step)
  float inv_m = a.inv_mass[idx];
  float g_inv_m = 0.f;
  // CLONE3 (see the joint wave): idle lanes clone the env's last body -- also its LDS accumulators: every lane of a read-fma-write reads
  // before any writes, so clones add the same value to the same old one and write the same sum
  constexpr bool CLONE3 = JT == PD_JT_COMPOUND;
  const bool wr = CLONE3 || is_body, gw = CLONE3 ? env_ok : is_body;
  float *ga_lds = gacc + (CLONE3 ? b : (l < nb ? l : nb)) * PD_GACC;  // (else idle lanes share a dummy slot: they must not alias body nb - 1's sums)
  float *I = ga_lds + 18, *invI = I + 9;
#pragma unroll
  for (int k = 0; k < 18; ++k) ga_lds[k] = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) { I[k] = a.inertia[idx * 9 + k]; invI[k] = a.inv_inertia[idx * 9 + k]; }
  int cz[4];  // first four children, the zero record for a missing one
#pragma unroll
  for (int k = 0; k < 4; ++k) cz[k] = wr && c.child[k] >= 0 ? c.child[k] : nb;

  BodyAdj gn = adj_zero();  // adjoint of state step+1
  BodyState s;
  s.p = V3(0, 0, 0); s.r = Q4(0, 0, 0, 1); s.w = V3(0, 0, 0); s.v = V3(0, 0, 0);
  v3 t0 = V3(0, 0, 0), f0 = t0;
  unsigned clamp_mask = 0u;  // which of (w, v) the forward pass clamped in the step (stored beside its wrench)
  float4 n_s[PD_TRAJ_G];
  int n_fr = -1, fr = -1;  // frame seeded into state step + 1 (or -1), fetched with the state
  auto load_step = [&](int step) {
    const int sc = __builtin_amdgcn_readfirstlane(step >= 0 ? step : 0);  // keeps the address arithmetic scalar
    n_fr = a.frame_of_step[sc + 1];
    const float *tj = a.ws + (size_t)sc * (PD_TRAJ_G * 4) * N;
#pragma unroll
    for (int g = 0; g < PD_TRAJ_G; ++g) n_s[g] = ldg4(tj + (size_t)(4 * g) * N, boff * 4u);
  };
  // unpacks the fetched state of `step`, stages its record and cull vector (generation step & 1) and tells the joint wave (S)
  auto stage_step = [&](int step) {
    PD_WAIT_VMEM();
    s.r = Q4(n_s[0].x, n_s[0].y, n_s[0].z, n_s[0].w); s.w = V3(n_s[1].x, n_s[1].y, n_s[1].z);
    s.p = V3(n_s[2].x, n_s[2].y, n_s[2].z); s.v = V3(n_s[1].w, n_s[2].w, n_s[3].x);
    t0 = V3(n_s[3].y, n_s[3].z, n_s[3].w); f0 = V3(n_s[4].x, n_s[4].y, n_s[4].z); clamp_mask = __float_as_uint(n_s[4].w);
    fr = n_fr;
    select_step(step);
    float Rm[9];
    rotm(s.r, Rm);  // the body's rotation as a matrix: shared by the staging and the adjoint of integrate_bodies
    const v3 rc = mat_vec(Rm, c.com);
    if (wr) stage_record(rec, cull, b, s, rc, Rm);
    if (JT != PD_JT_REVOLUTE) pair_signal(sig + 3, a.nsteps - step);  // S
  };
  // (ROLES == 2) the forward hit list of a step is replayed inline: its length and this lane's entry are fetched a step ahead
  // (with the state), so that the replay does not open with a load at its point of use
  const bool pre_ok = ROLES == 2 && SEGW >= PD_HITLOG - 1;  // one entry per lane covers every list the log can hold
  const int lq = l < PD_HITLOG - 1 ? l : PD_HITLOG - 2;
  const unsigned boff_lg = (unsigned)ec * (PD_HITLOG * 4u);
  int cnt_c = 0, e_c = 0, cnt_n = 0, e_n = 0;
  auto load_log = [&](int step, int &cnt, int &e) {
    const int *lgp = a.hitlog + (size_t)__builtin_amdgcn_readfirstlane(step > 0 ? step : 0) * a.bs * PD_HITLOG;
    cnt = __float_as_int(ldg((const float *)lgp, boff_lg)); e = __float_as_int(ldg((const float *)lgp + 1, boff_lg + (unsigned)lq * 4u));
  };
  if (pre_ok && a.nsteps > 0) load_log(a.nsteps - 1, cnt_c, e_c);
  if (a.nsteps > 0) { load_step(a.nsteps - 1); stage_step(a.nsteps - 1); }
  STAMP_DECL;
  for (int step = a.nsteps - 1; step >= 0; --step) {
    // s, t0, f0, fr and the staged record of `step` are in place (stage_step ran in the previous iteration)
    if (fr >= 0) {  // seeds of state step+1 (dp_model.py:1264-1271)
      add_frame_seeds(a, fr, N, idx, gn);
    }
    // (ROLES == 2) the forward hit list of this step is replayed inline further down: fetch its length now, far ahead
    int *lg = a.hitlog + ((size_t)step * a.bs + ec) * PD_HITLOG;
    const int log_cnt = (ROLES == 2 && env_ok) ? (pre_ok ? cnt_c : lg[0]) : 0;
    STAMP(0);
    // ---- adjoint of integrate_bodies; hand-over A as soon as the wrench adjoint exists
    BodyAdj ga = adj_zero();
    v3 adj_t0 = V3(0, 0, 0), adj_f0 = adj_t0;
    float Rm[9];
    rotm(s.r, Rm);  // (recomputed rather than carried from stage_step: nine registers across the loop cost more than 22 instructions)
    float aR[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) aR[k] = 0.f;
    integrate_adj2(m, c, s, Rm, clamp_mask, t0, f0, inv_m, I, invI, a.dt, gn, ga, aR, g_inv_m, LdsAcc9{ga_lds}, LdsAcc9{ga_lds + 9}, [&](v3 t, v3 f) {
      adj_t0 = t; adj_f0 = f;
      if (wr) {
        float *o = adjf + b * PD_W6;
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = f.x; o[4] = f.y; o[5] = f.z;
      }
      pair_signal(sig, a.nsteps - step);  // A
      // the stored state of the NEXT iteration is requested here: its 20 registers are free during phase 1, and phase 2
      // (plus the contacts) covers the HBM latency before stage_step consumes it
      load_step(step - 1);
      if (pre_ok) load_log(step - 1, cnt_n, e_n);
    });
    rotm_adj(s.r, aR, ga.r);
    if (gw) {
      float *o = a.g_res_f + (size_t)__builtin_amdgcn_readfirstlane(step) * N * 6;  // adjoint of wp_add
      stg2(o, boff * 6u, make_float2(NZ(adj_t0.x), NZ(adj_t0.y))); stg2(o + 2, boff * 6u, make_float2(NZ(adj_t0.z), NZ(adj_f0.x)));
      stg2(o + 4, boff * 6u, make_float2(NZ(adj_f0.y), NZ(adj_f0.z)));
    }
    STAMP(1);
    if (ROLES == 2) {  // adjoint of eval_body_contacts, while the joint wave works
      WAVE_SYNC();     // records and wrench adjoints were written by this wave's own lanes
      const bool replay = __ballot(log_cnt < 0) == 0ull;
      int log_n_unused;
      sweep_contacts<SEGW, PD_ADJ, PD_ADJ, true>(m, tabs, c, is_body ? cull[b] : make_float4(0.f, 0.f, 1.f, 0.f), rec, cull, list, hits, slot,
                                                 cacc, is_body, env_ok, seg, l, replay ? lg : nullptr, replay ? log_cnt : PD_NO_REPLAY,
                                                 log_n_unused, contact_hit STAMP_PASS, pre_ok, e_c);
      WAVE_SYNC();
      STAMP(5);
    }
    // the next step's records, staged before this step's results are awaited (the other generation: nobody reads it now)
    if (step > 0) stage_step(step - 1);
    STAMP(6);
    pair_wait(sig + 1, a.nsteps - step);  // J: joint contributions are complete
    STAMP(2);
    {  // own joint first, then the first four children with all LDS reads in flight together (packed sums), then any further ones
      const float *const src[5] = {oslot + b * PD_ADJ, cslot + cz[0] * PD_ADJ, cslot + cz[1] * PD_ADJ, cslot + cz[2] * PD_ADJ, cslot + cz[3] * PD_ADJ};
      adj_add_from_n(ga, src);
    }
    for (int k = 4; k < m.max_children; ++k) {
      int cid = (int)((c.children >> (8 * k)) & 0xffull);
      if (wr && cid != 0xff) adj_add_from(ga, cslot + cid * PD_ADJ);
    }
    STAMP(3);
    if (ROLES == 3) pair_wait(sig + 2, a.nsteps - step);  // C: contact adjoints are complete
    if (wr) {
      float *d = cacc + b * PD_ADJ;
      const float *const src[1] = {d};
      adj_add_from_n(ga, src);
#pragma unroll
      for (int k = 0; k < PD_ADJ; ++k) d[k] = 0.f;
    }
    gn = ga;
    cnt_c = cnt_n; e_c = e_n;
    STAMP(4);
  }
  select_step(0);
  STAMP_FLUSH(a);
  {  // seeds of state 0
    int fr = a.frame_of_step[0];
    if (fr >= 0) {
      add_frame_seeds(a, fr, N, idx, gn);
    }
  }
  // ---- adjoint of eval_fk: rec holds state 0 (staged in the last loop iteration); the J wave is past its last use of cslot
  c = load_body_const(m, b, ec);
  if (a.nsteps == 0) {
    for (int d = 0; d <= m.max_depth; ++d) {  // nothing staged yet: rebuild state 0
      if (is_body && c.depth == d) {
        s = fk_joint<JT>(c, a.q_init + (size_t)ec * m.nq + c.qstart, a.qd_init + (size_t)ec * m.nqd + c.qdstart, rec);
        stage_record(rec, b, s, c.com);
      }
      WAVE_SYNC();
    }
  }
  WAVE_SYNC();
  for (int d = m.max_depth; d >= 0; --d) {
    if (is_body && c.depth == d) {
      for (int k = 0; k < m.max_children; ++k) {
        int cid = (int)((c.children >> (8 * k)) & 0xffull);
        if (cid != 0xff) adj_add_from(gn, cslot + cid * PD_ADJ);
      }
      BodyAdj par = fk_joint_adj<JT, 1>(c, a.q_init + (size_t)ec * m.nq + c.qstart, a.qd_init + (size_t)ec * m.nqd + c.qdstart, rec,
                                     gn, a.g_q_init + (size_t)ec * m.nq + c.qstart, a.g_qd_init + (size_t)ec * m.nqd + c.qdstart);
      adj_store(cslot + b * PD_ADJ, par);
    }
    WAVE_SYNC();
  }
  if (is_body) {
    a.g_inv_mass[idx] = NZ(g_inv_m);
#pragma unroll
    for (int k = 0; k < 9; ++k) { a.g_inertia[idx * 9 + k] = NZ(ga_lds[k]); a.g_inv_inertia[idx * 9 + k] = NZ(ga_lds[9 + k]); }
  }
}

// =============================================================================================
// Batched FK (ForwardKinematics, dp_model.py:1022-1130): n articulations, one per segment.
// the articulations of workgroup `block` of an FK launch (blockDim.x / 64 body waves per workgroup)
template <int SEGW, int JT, bool BWD>
PD_DEV void fk_chain(const PdDevModel &m, const FkArgs &a, unsigned char *smem, int block) {
  constexpr int EPW = Seg<SEGW>::EPW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = lane / SEGW, l = lane % SEGW;
  const int env = (block * (int)(blockDim.x >> 6) + wave) * EPW + seg;
  const bool is_body = env < a.n && l < m.nb;
  const int b = l < m.nb ? l : m.nb - 1, nb = m.nb;
  const int ec = env < a.n ? env : 0;
  float *rec = (float *)smem + (size_t)(wave * EPW + seg) * (nb * (PD_REC + PD_ADJ));
  float *cslot = rec + nb * PD_REC;
  const BodyConst c = load_body_const(m, b, ec);
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
  // body rows: articulation order, or env-major when the articulations come frame-major (FkArgs::perm_bs)
  const size_t row = a.perm_bs > 0 ? (size_t)(ec % a.perm_bs) * (size_t)(a.n / a.perm_bs) + (size_t)(ec / a.perm_bs) : (size_t)ec;
  const size_t idx = row * nb + b;
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
      BodyAdj par = fk_joint_adj<JT, 2>(c, jq, jqd, rec, g, a.g_joint_q + (size_t)ec * m.nq + c.qstart,
                                     a.g_joint_qd + (size_t)ec * m.nqd + c.qdstart);
      adj_store(cslot + b * PD_ADJ, par);
    }
    WAVE_SYNC();
  }
}

template <int SEGW, int JT, bool BWD>
__global__ __launch_bounds__(PD_FK_BLOCK) void k_fk(PdDevModel m, FkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  fk_chain<SEGW, JT, BWD>(m, a, smem, blockIdx.x);
}

// Row f4: reduce_loss of the trajectory loss (workgroup 0) and the FK of the control reference (dp_model.py:758 of the reference; the
// other workgroups, 16 body waves each) in ONE launch between the two rollout launches -- the FK has no launch of its own any more.
template <int SEGW, int JT>
__global__ __launch_bounds__(PD_REDUCE_BLOCK) void k_reduce_fk(PdDevModel m, ReduceFkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (blockIdx.x == 0) {
    if (a.in_lds) traj_loss_reduce_block<true>(a.red, (float *)smem);
    else traj_loss_reduce_block<false>(a.red, (float *)smem);
    return;
  }
  fk_chain<SEGW, JT, false>(m, a.fk, smem, blockIdx.x - 1);
}
// ... and its adjoint rides on the launch that builds the adjoint rollout's seeds (workgroups seeds.nblocks .. are FK backward)
template <int SEGW, int JT>
__global__ __launch_bounds__(PD_FK_BLOCK) void k_seeds_fk(PdDevModel m, SeedsFkArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((int)blockIdx.x < a.seeds.nblocks) { traj_seeds_block(a.seeds, blockIdx.x); return; }
  fk_chain<SEGW, JT, true>(m, a.fk, smem, blockIdx.x - a.seeds.nblocks);
}

// =============================================================================================
// Launchers: this file is compiled once per segment width (-DPD_SEGW=16|32|64).
#ifndef PD_SEGW
#error "compile with -DPD_SEGW=16, 32 or 64"
#endif
#define PD_CAT2(a, b) a##b
#define PD_CAT(a, b) PD_CAT2(a, b)

// Wave specialisation (pd_split / pd_split_launch in pd_args.h).  Adjoint: revolute-only robots (compound joints need
// > 256 VGPRs, no room for a partner wave).  Forward: every joint mix fits, and it pays while a CU holds at most one
// workgroup (the latency regime: human at 1024 envs -32 %); with several workgroups per CU the unsplit kernel's 4-wave
// workgroups pack twice as many body waves per SIMD (quad at 8192 envs: split +22 %), so the launcher picks per launch.

// cfg: the host's choice for this launch (pd_args.h: pd_launch_cfg) -- kernel variant, workgroups, threads, LDS bytes
template <int JT>
static hipError_t launch_jt(int kind, const PdDevModel &m, const void *args, const PdLaunchCfg &cfg, hipStream_t st) {
  const dim3 g(cfg.nblocks), t(cfg.threads);
  const size_t lds = cfg.lds;
  switch (kind) {
    case PD_K_ROLLOUT_FWD:
      if (cfg.kernel == PD_KV_FWD_QUAD) {
        if constexpr (PD_SEGW == 64 && JT == PD_JT_REVOLUTE) {
          const bool loss = ((const RolloutArgs *)args)->loss_target != nullptr;
          if (cfg.roles == 3) {  // with the cull wave
            if (loss) hipLaunchKernelGGL((k_rollout_fwd<PD_SEGW, JT, true, true, true, true>), g, t, lds, st, m, *(const RolloutArgs *)args);
            else hipLaunchKernelGGL((k_rollout_fwd<PD_SEGW, JT, true, false, true, true>), g, t, lds, st, m, *(const RolloutArgs *)args);
          } else {
            if (loss) hipLaunchKernelGGL((k_rollout_fwd<PD_SEGW, JT, true, true, true>), g, t, lds, st, m, *(const RolloutArgs *)args);
            else hipLaunchKernelGGL((k_rollout_fwd<PD_SEGW, JT, true, false, true>), g, t, lds, st, m, *(const RolloutArgs *)args);
          }
          break;
        } else {
          return hipErrorInvalidValue;
        }
      }
      if (cfg.roles == 3) {  // wave-specialised forward with the cull wave (revolute-only robots)
        if constexpr (JT == PD_JT_REVOLUTE) {
          const bool loss = ((const RolloutArgs *)args)->loss_target != nullptr;
          if (cfg.groups >= PD_BWAVES) {  // full workgroups: per-body sums in registers (RUNSUM)
            if (loss) hipLaunchKernelGGL((k_rollout_fwd<PD_SEGW, JT, true, true, false, true, true>), g, t, lds, st, m, *(const RolloutArgs *)args);
            else hipLaunchKernelGGL((k_rollout_fwd<PD_SEGW, JT, true, false, false, true, true>), g, t, lds, st, m, *(const RolloutArgs *)args);
          } else {
            if (loss) hipLaunchKernelGGL((k_rollout_fwd<PD_SEGW, JT, true, true, false, true>), g, t, lds, st, m, *(const RolloutArgs *)args);
            else hipLaunchKernelGGL((k_rollout_fwd<PD_SEGW, JT, true, false, false, true>), g, t, lds, st, m, *(const RolloutArgs *)args);
          }
          break;
        } else {
          return hipErrorInvalidValue;
        }
      }
      // (unsplit: compound-only robots above 4 x CUs env groups -- pd_kernel_variant; no other joint mix has that instantiation)
      if (((const RolloutArgs *)args)->loss_target) {  // trajectory loss at the frame states (pd_rollout_forward_traj_loss)
        if (cfg.kernel != PD_KV_FWD_SPLIT) return hipErrorInvalidValue;  // (the host sends every loss-evaluating launch to the split kernel: pd_host.hip launch_cfg)
        hipLaunchKernelGGL((k_rollout_fwd<PD_SEGW, JT, true, true>), g, t, lds, st, m, *(const RolloutArgs *)args);
      } else if (cfg.kernel == PD_KV_FWD_SPLIT || JT != PD_JT_COMPOUND)
        hipLaunchKernelGGL((k_rollout_fwd<PD_SEGW, JT, true>), g, t, lds, st, m, *(const RolloutArgs *)args);
      else
        hipLaunchKernelGGL((k_rollout_fwd<PD_SEGW, JT, JT != PD_JT_COMPOUND>), g, t, lds, st, m, *(const RolloutArgs *)args);
      break;
    case PD_K_ROLLOUT_BWD:
      if constexpr (pd_split(JT)) {
        if (cfg.kernel == PD_KV_BWD_QUAD) {
          if constexpr (PD_SEGW == 64 && JT == PD_JT_REVOLUTE) {
            hipLaunchKernelGGL((k_rollout_bwd<PD_SEGW, JT, true, false, true>), g, t, lds, st, m, *(const RolloutArgs *)args);
            break;
          } else {
            return hipErrorInvalidValue;
          }
        }
        hipLaunchKernelGGL((k_rollout_bwd<PD_SEGW, JT, true, false>), g, t, lds, st, m, *(const RolloutArgs *)args);
      } else {
        hipLaunchKernelGGL((k_rollout_bwd3<PD_SEGW, JT, 2>), g, t, lds, st, m, *(const RolloutArgs *)args);
      }
      break;
#if PD_POLICY == 0  // (the host routes the FK kinds to these launchers whatever the model's policy)
    case PD_K_FK_FWD:
      hipLaunchKernelGGL((k_fk<PD_SEGW, JT, false>), g, t, lds, st, m, *(const FkArgs *)args);
      break;
    case PD_K_FK_BWD:
      hipLaunchKernelGGL((k_fk<PD_SEGW, JT, true>), g, t, lds, st, m, *(const FkArgs *)args);
      break;
    case PD_K_REDUCE_FK:
      hipLaunchKernelGGL((k_reduce_fk<PD_SEGW, JT>), g, t, lds, st, m, *(const ReduceFkArgs *)args);
      break;
    case PD_K_SEEDS_FK:
      hipLaunchKernelGGL((k_seeds_fk<PD_SEGW, JT>), g, t, lds, st, m, *(const SeedsFkArgs *)args);
      break;
#endif
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

template <int JT>
static hipError_t set_lds_jt(int bytes) {
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void *)k_rollout_fwd<PD_SEGW, JT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
  if ((e = hipFuncSetAttribute((const void *)k_rollout_fwd<PD_SEGW, JT, JT != PD_JT_COMPOUND>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
  if constexpr (PD_SEGW == 64 && JT == PD_JT_REVOLUTE) {
    if ((e = hipFuncSetAttribute((const void *)k_rollout_fwd<PD_SEGW, JT, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
    if ((e = hipFuncSetAttribute((const void *)k_rollout_fwd<PD_SEGW, JT, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
  }
  if constexpr (JT == PD_JT_REVOLUTE) {  // ... with the cull wave
    if ((e = hipFuncSetAttribute((const void *)k_rollout_fwd<PD_SEGW, JT, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
    if ((e = hipFuncSetAttribute((const void *)k_rollout_fwd<PD_SEGW, JT, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
    if ((e = hipFuncSetAttribute((const void *)k_rollout_fwd<PD_SEGW, JT, true, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
    if ((e = hipFuncSetAttribute((const void *)k_rollout_fwd<PD_SEGW, JT, true, true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
    if constexpr (PD_SEGW == 64) {
      if ((e = hipFuncSetAttribute((const void *)k_rollout_fwd<PD_SEGW, JT, true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
      if ((e = hipFuncSetAttribute((const void *)k_rollout_fwd<PD_SEGW, JT, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
    }
  }
  if ((e = hipFuncSetAttribute((const void *)k_rollout_fwd<PD_SEGW, JT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
  if constexpr (pd_split(JT)) {
    if ((e = hipFuncSetAttribute((const void *)k_rollout_bwd<PD_SEGW, JT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
    if constexpr (PD_SEGW == 64 && JT == PD_JT_REVOLUTE) {
      if ((e = hipFuncSetAttribute((const void *)k_rollout_bwd<PD_SEGW, JT, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
    }
  } else {
    if ((e = hipFuncSetAttribute((const void *)k_rollout_bwd3<PD_SEGW, JT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
  }
#if PD_POLICY == 0
  if ((e = hipFuncSetAttribute((const void *)k_fk<PD_SEGW, JT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
  if ((e = hipFuncSetAttribute((const void *)k_reduce_fk<PD_SEGW, JT>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
  if ((e = hipFuncSetAttribute((const void *)k_seeds_fk<PD_SEGW, JT>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e;
  return hipFuncSetAttribute((const void *)k_fk<PD_SEGW, JT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
#else
  return hipSuccess;
#endif
}

// jt: PD_JT_REVOLUTE only, PD_JT_COMPOUND only, anything else -> generic (all joint types)
hipError_t PD_LAUNCH_NAME(PD_SEGW)(int kind, int jt, const PdDevModel &m, const void *args, const PdLaunchCfg &cfg, hipStream_t st) {
  if (jt == PD_JT_REVOLUTE) return launch_jt<PD_JT_REVOLUTE>(kind, m, args, cfg, st);
  if (jt == PD_JT_COMPOUND) return launch_jt<PD_JT_COMPOUND>(kind, m, args, cfg, st);
  return launch_jt<PD_JT_REVOLUTE | PD_JT_COMPOUND | PD_JT_FIXED>(kind, m, args, cfg, st);
}
hipError_t PD_SET_LDS_NAME(PD_SEGW)(int jt, int bytes) {
  if (jt == PD_JT_REVOLUTE) return set_lds_jt<PD_JT_REVOLUTE>(bytes);
  if (jt == PD_JT_COMPOUND) return set_lds_jt<PD_JT_COMPOUND>(bytes);
  return set_lds_jt<PD_JT_REVOLUTE | PD_JT_COMPOUND | PD_JT_FIXED>(bytes);
}
