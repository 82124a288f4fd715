// Microbenchmark (round 6): what one LDS hand-over between two waves of a workgroup costs -- the pair_signal / pair_wait of the rollout
// kernels (csrc/pd_kernels.hip).  Two waves ping-pong: wave P writes NREC floats per lane to LDS, publishes a step counter; wave C polls
// it, reads the floats, writes its own, publishes; P polls.  Reported: s_memtime ticks and wall clocks per ROUND TRIP (two hand-overs),
// with WORK dependent FMAs per side per round subtracted by running WORK = 0 and WORK = 64 (the difference tells whether the FMAs hide
// any of the latency).
//   MODE 0: release fence (s_waitcnt lgkmcnt(0)) + lane-0 store, plain polling                      -- rounds 2-5
//   MODE 1: no fence (the LDS executes a wave's instructions in order), all-lane store, plain polling -- round 6
//   MODE 2: as 1, polling with two reads in flight
//   MODE 3: as 1, s_sleep 1 between polls
// PLACE 0: the two waves on different SIMDs (waves 0 and 1 of the workgroup); PLACE 1: on the same SIMD (waves 0 and 4).
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ inline void sig_fence(int *flag, int v) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ inline void sig_plain(int *flag, int v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
template <int MODE>
__device__ inline void wait_for(int *flag, int v) {
  if (MODE == 2) {
    int a = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
      const int b = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (__builtin_amdgcn_readfirstlane(a) >= v) break;
      a = b;
    }
  } else {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < v) {
      if (MODE == 3) __builtin_amdgcn_s_sleep(1);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int MODE, int PLACE, int WORK>
__global__ void k(float *out, long long *ticks, int rounds) {
  __shared__ float rec[2][64 * 4];
  __shared__ int flag[2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int partner = PLACE ? 4 : 1;
  if (threadIdx.x == 0) { flag[0] = 0; flag[1] = 0; }
  __syncthreads();
  if (wave != 0 && wave != partner) return;
  const int me = wave == 0 ? 0 : 1;
  float x = lane * 0.001f, acc = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 1; r <= rounds; ++r) {
    if (me == 1) { wait_for<MODE>(&flag[0], r); x += rec[0][lane] + rec[0][64 + lane] + rec[0][128 + lane] + rec[0][192 + lane]; }
#pragma unroll
    for (int i = 0; i < WORK; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    rec[me][lane] = x; rec[me][64 + lane] = x + 1.f; rec[me][128 + lane] = x + 2.f; rec[me][192 + lane] = x + 3.f;
    if (MODE == 0) sig_fence(&flag[me], r); else sig_plain(&flag[me], r);
    if (me == 0) { wait_for<MODE>(&flag[1], r); x += rec[1][lane] + rec[1][64 + lane] + rec[1][128 + lane] + rec[1][192 + lane]; }
    acc += x;
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && me == 0) ticks[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 128 + me * 64 + lane] = acc;
}

template <int MODE, int PLACE, int WORK>
double run(float *d, long long *t, double &wall_ns) {
  const int rounds = 20000, blocks = 64;
  hipLaunchKernelGGL((k<MODE, PLACE, WORK>), dim3(blocks), dim3(512), 0, 0, d, t, rounds);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, PLACE, WORK>), dim3(blocks), dim3(512), 0, 0, d, t, rounds);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[64];
  (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < blocks; ++i) s += (double)h[i];
  wall_ns = ms * 1e6 / rounds;
  return s / blocks / rounds;
}

template <int MODE, int PLACE>
void row(float *d, long long *t, const char *what) {
  double w0, w64;
  const double t0 = run<MODE, PLACE, 0>(d, t, w0), t64 = run<MODE, PLACE, 64>(d, t, w64);
  printf("%-58s %s  round trip %6.1f ticks %6.1f ns | with 64 dependent FMAs per side %6.1f ticks %6.1f ns\n", what,
         PLACE ? "same SIMD     " : "different SIMDs", t0, w0, t64, w64);
}

int main() {
  float *d; long long *t;
  (void)hipMalloc(&d, 64 * 128 * sizeof(float)); (void)hipMalloc(&t, 64 * sizeof(long long));
  printf("# one round trip = two hand-overs (P -> C, C -> P); 4 floats per lane written and read per hand-over; s_memtime ticks as the device counts them\n");
  row<0, 0>(d, t, "fence + lane-0 store, plain poll (rounds 2-5)");
  row<1, 0>(d, t, "no fence, all-lane store, plain poll (round 6)");
  row<2, 0>(d, t, "no fence, poll with two reads in flight");
  row<3, 0>(d, t, "no fence, s_sleep 1 between polls");
  row<0, 1>(d, t, "fence + lane-0 store, plain poll (rounds 2-5)");
  row<1, 1>(d, t, "no fence, all-lane store, plain poll (round 6)");
  row<2, 1>(d, t, "no fence, poll with two reads in flight");
  row<3, 1>(d, t, "no fence, s_sleep 1 between polls");
  return 0;
}
