// sdc_reward.hip -- RING PATH of the history-normalised rewards: one workgroup (4 wavefronts) per queued env.
//
// utils/reward_creator.py:16-45 normalises the step's energy with the 25th / 75th percentiles, the IQR-clipped
// mean and the clipped population std of a 10 000-entry sliding history.  sdc_trackers.hpp maintains all of that
// incrementally; the dynamics kernel answers most env-steps from the trackers alone (no history read) and queues
// an env here only when a wanted rank or clip bound has moved past the keys its tracker lists.  This kernel then
// streams the env's ring ONCE (40 KB as 10 x dwordx4 per lane, coalesced, all loads in flight together) into LDS
// and re-anchors the trackers with one primitive, the REBUILD SWEEP around two anchor keys g:
//   per key and anchor  ds = x - (g+1)  borrows <=> x <= g,   dp = (g-1) - x  borrows <=> x >= g
// (v_sub_co_u32 + v_addc_co_u32: the borrow counts the predicate, and in wrap-around arithmetic a key on the wrong
// side of g gets a distance above every legitimate one, so the 4 smallest ds / dp -- v_med3_u32 insertion network --
// are the 4 nearest keys above / below g).  It serves
//   * the quartile trackers: new anchor = the last listed key on the side the wanted rank left by;
//   * the tail trackers: new anchors exactly at the clip bounds, plus one fp64 summation pass;
//   * the bootstrap (first steps, injected state): anchors from an exact bisection on the key space.
// Histories shorter than SMALL_N are computed directly.
//
// Round-1 measurements that shaped this: streaming all 4096 rings every step is HBM-bound at >= 23 us (ablation,
// tools/ablate) and was 32 us in practice; with the trackers ~10 % of the envs need their ring on a given step.
// Those few workgroups per CU run on a COLD instruction cache (the dynamics kernel ran in between; the cache is
// 64 KB per CU pair): in-kernel stamps showed ~0.45 us per KB of straight-line code executed, i.e. the first
// versions (40-way unrolled register sweeps, separate slide / rebuild code: 68 KB, later 45 KB) spent 13 us per
// workgroup mostly fetching instructions.  Hence: keys in LDS, short rolled loops, ONE non-inlined sweep routine,
// so that this kernel and the dynamics kernel fit the instruction cache together.
#include "sdc_trackers.hpp"

namespace {

using namespace sdc_rw;

struct RewardShared {
  unsigned red_u[2][4];
  unsigned red_v[2][4];
  unsigned sweep[4][2][2 + 2 * QW];  // per wave, per anchor: count <=, count >=, QW pred distances, QW succ distances
  unsigned fin[2][2 + 2 * QW];       // the same, merged over the workgroup
  double red_d[4];
  double red_e[4];
  double red_f[4][4];
};

__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// ------------------------------------------------------------------------------------------------
// Wave reductions on the DPP data path: xor 1, xor 2 (quad_perm), row_half_mirror, row_mirror reduce within each row
// of 16 lanes; row_bcast15 / row_bcast31 carry the rows into lane 63, which holds the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned identity, unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}
#define SDC_DPP_STAGES(STAGE)                      \
  STAGE(0xB1, 0xF)  /* quad_perm [1,0,3,2] */      \
  STAGE(0x4E, 0xF)  /* quad_perm [2,3,0,1] */      \
  STAGE(0x141, 0xF) /* row_half_mirror */          \
  STAGE(0x140, 0xF) /* row_mirror */               \
  STAGE(0x142, 0xA) /* row_bcast15 -> rows 1, 3 */ \
  STAGE(0x143, 0xC) /* row_bcast31 -> rows 2, 3 */
__device__ __forceinline__ unsigned wave_sum_to63(unsigned v) {
#define STAGE(C, M) v += dpp_u32<C, M>(0u, v);
  SDC_DPP_STAGES(STAGE)
#undef STAGE
  return v;
}
__device__ __forceinline__ unsigned wave_min_to63(unsigned v) {
#define STAGE(C, M) v = min(v, dpp_u32<C, M>(KEY_NONE, v));
  SDC_DPP_STAGES(STAGE)
#undef STAGE
  return v;
}
__device__ __forceinline__ unsigned wave_max_to63(unsigned v) {
#define STAGE(C, M) v = max(v, dpp_u32<C, M>(0u, v));
  SDC_DPP_STAGES(STAGE)
#undef STAGE
  return v;
}
// the 4 smallest of the wave's 64 ascending 4-lists, into lane 63
__device__ __forceinline__ void wave_merge_to63(L4& A) {
#define STAGE(C, M)                                                                         \
  {                                                                                         \
    const unsigned b0 = dpp_u32<C, M>(KEY_NONE, A.e0), b1 = dpp_u32<C, M>(KEY_NONE, A.e1);  \
    const unsigned b2 = dpp_u32<C, M>(KEY_NONE, A.e2), b3 = dpp_u32<C, M>(KEY_NONE, A.e3);  \
    asc_insert(A, b0);                                                                      \
    asc_insert(A, b1);                                                                      \
    asc_insert(A, b2);                                                                      \
    asc_insert(A, b3);                                                                      \
  }
  SDC_DPP_STAGES(STAGE)
#undef STAGE
}
__device__ __forceinline__ void l4_sweep_insert(L4& L, const unsigned d) {
  L.e3 = umed3(L.e2, d, L.e3);
  L.e2 = umed3(L.e1, d, L.e2);
  L.e1 = umed3(L.e0, d, L.e1);
  L.e0 = min(L.e0, d);
}
// d = a - b, cnt += borrow
#define SDC_SUB_COUNT(d, cnt, a, b) \
  asm("v_sub_co_u32 %0, vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "=&v"(d), "+v"(cnt) : "v"(a), "v"(b) : "vcc")

// block-wide reductions over 4 wavefronts (lane 63 of each wave holds its partial); one barrier per call, `par`
// alternates the LDS slot
__device__ __forceinline__ unsigned block_sum_u32(unsigned v, unsigned (*red)[4], int par, int wave, int lane) {
  v = wave_sum_to63(v);
  if (lane == 63) red[par][wave] = v;
  __syncthreads();
  return red[par][0] + red[par][1] + red[par][2] + red[par][3];
}
__device__ __forceinline__ unsigned block_min_u32(unsigned v, unsigned (*red)[4], int par, int wave, int lane) {
  v = wave_min_to63(v);
  if (lane == 63) red[par][wave] = v;
  __syncthreads();
  return min(min(red[par][0], red[par][1]), min(red[par][2], red[par][3]));
}
__device__ __forceinline__ unsigned block_max_u32(unsigned v, unsigned (*red)[4], int par, int wave, int lane) {
  v = wave_max_to63(v);
  if (lane == 63) red[par][wave] = v;
  __syncthreads();
  return max(max(red[par][0], red[par][1]), max(red[par][2], red[par][3]));
}

// ------------------------------------------------------------------------------------------------
// Exact order statistics at ranks k1, k1+1, k3, k3+1 by bisection on the key space with block-wide counts
// (bootstrap, tiny histories, verify mode).  Returns {a1, b1, a3, b3}.  Block-uniform control flow.
__device__ __noinline__ uint4 quartiles_by_bisection(const uint4* __restrict__ lk, const int k1, const int k3,
                                                     RewardShared* shp, const int lane, const int wave) {
  RewardShared& sh = *shp;
  int par = 0;
  unsigned kmin = KEY_NONE, kmax = 0u;
#pragma unroll 1
  for (int q = 0; q < SDC_HIST_PER_THREAD / 4; q++) {
    const uint4 v = lk[q * SDC_BLOCK];
    const unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      kmin = min(kmin, x[c]);
      kmax = max(kmax, x[c] == KEY_NONE ? 0u : x[c]);
    }
  }
  kmin = block_min_u32(kmin, sh.red_u, par, wave, lane);
  kmax = block_max_u32(kmax, sh.red_v, par, wave, lane);
  par ^= 1;
  unsigned lo1 = kmin, hi1 = kmax, lo3 = kmin, hi3 = kmax;
  while (lo1 < hi1 || lo3 < hi3) {
    const unsigned m1 = lo1 + ((hi1 - lo1) >> 1);
    const unsigned m3 = lo3 + ((hi3 - lo3) >> 1);
    unsigned cnt = 0;  // packed: count(key <= m1) << 16 | count(key <= m3); each <= 10240
#pragma unroll 1
    for (int q = 0; q < SDC_HIST_PER_THREAD / 4; q++) {
      const uint4 v = lk[q * SDC_BLOCK];
      const unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int c = 0; c < 4; c++) cnt += ((x[c] <= m1) ? 0x10000u : 0u) + ((x[c] <= m3) ? 1u : 0u);
    }
    cnt = block_sum_u32(cnt, sh.red_u, par, wave, lane);
    par ^= 1;
    const int c1 = (int)(cnt >> 16), c3 = (int)(cnt & 0xFFFFu);
    if (lo1 < hi1) {
      if (c1 >= k1 + 1) hi1 = m1; else lo1 = m1 + 1;
    }
    if (lo3 < hi3) {
      if (c3 >= k3 + 1) hi3 = m3; else lo3 = m3 + 1;
    }
  }
  // successors: value at rank k+1 = same value if count(<= v_k) >= k+2, else min{key > v_k}
  unsigned cnt = 0, s1 = KEY_NONE, s3 = KEY_NONE;
#pragma unroll 1
  for (int q = 0; q < SDC_HIST_PER_THREAD / 4; q++) {
    const uint4 v = lk[q * SDC_BLOCK];
    const unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      cnt += ((x[c] <= lo1) ? 0x10000u : 0u) + ((x[c] <= lo3) ? 1u : 0u);
      if (x[c] > lo1) s1 = min(s1, x[c]);
      if (x[c] > lo3) s3 = min(s3, x[c]);
    }
  }
  cnt = block_sum_u32(cnt, sh.red_u, par, wave, lane);
  s1 = block_min_u32(s1, sh.red_v, par, wave, lane);
  par ^= 1;
  s3 = block_min_u32(s3, sh.red_u, par, wave, lane);
  uint4 r;
  r.x = lo1;
  r.z = lo3;
  r.y = ((int)(cnt >> 16) >= k1 + 2 || s1 == KEY_NONE) ? lo1 : s1;
  r.w = ((int)(cnt & 0xFFFFu) >= k3 + 2 || s3 == KEY_NONE) ? lo3 : s3;
  __syncthreads();
  return r;
}

// ------------------------------------------------------------------------------------------------
// REBUILD SWEEP around anchors g1, g3 (any keys in (0, KEY_NONE), present in the ring or not).  Leaves, per anchor,
// {#{x <= g}, #{x >= g} (empty slots included), 4 smallest predecessor distances, 4 smallest successor distances}
// merged over the workgroup in sh.fin.  ONE copy of this code serves every re-anchoring (see the file header).
__device__ __noinline__ void rebuild_sweep(const uint4* __restrict__ lk, const unsigned g1, const unsigned g3,
                                           RewardShared* shp, const int lane, const int wave) {
  RewardShared& sh = *shp;
  unsigned cle1 = 0u, cge1 = 0u, cle3 = 0u, cge3 = 0u;
  L4 pd1 = {KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE}, sd1 = pd1, pd3 = pd1, sd3 = pd1;
  const unsigned g1p = g1 + 1u, g1m = g1 - 1u, g3p = g3 + 1u, g3m = g3 - 1u;
#pragma unroll 1
  for (int q4 = 0; q4 < SDC_HIST_PER_THREAD / 4; q4++) {
    const uint4 v4 = lk[q4 * SDC_BLOCK];
    const unsigned xs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const unsigned x = xs[c4];
      unsigned ds1, dp1, ds3, dp3;
      SDC_SUB_COUNT(ds1, cle1, x, g1p);
      SDC_SUB_COUNT(dp1, cge1, g1m, x);
      SDC_SUB_COUNT(ds3, cle3, x, g3p);
      SDC_SUB_COUNT(dp3, cge3, g3m, x);
      l4_sweep_insert(sd1, ds1);
      l4_sweep_insert(pd1, dp1);
      l4_sweep_insert(sd3, ds3);
      l4_sweep_insert(pd3, dp3);
    }
  }
  // across lanes (one copy of the merge code: the four lists rotate through it)
  cle1 = wave_sum_to63(cle1);
  cge1 = wave_sum_to63(cge1);
  cle3 = wave_sum_to63(cle3);
  cge3 = wave_sum_to63(cge3);
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    wave_merge_to63(pd1);
    const L4 t = pd1;
    pd1 = sd1;
    sd1 = pd3;
    pd3 = sd3;
    sd3 = t;
  }
  if (lane == 63) {
    unsigned* w = sh.sweep[wave][0];
    w[0] = cle1; w[1] = cge1;
    w[2] = pd1.e0; w[3] = pd1.e1; w[4] = pd1.e2; w[5] = pd1.e3;
    w[6] = sd1.e0; w[7] = sd1.e1; w[8] = sd1.e2; w[9] = sd1.e3;
    w = sh.sweep[wave][1];
    w[0] = cle3; w[1] = cge3;
    w[2] = pd3.e0; w[3] = pd3.e1; w[4] = pd3.e2; w[5] = pd3.e3;
    w[6] = sd3.e0; w[7] = sd3.e1; w[8] = sd3.e2; w[9] = sd3.e3;
  }
  __syncthreads();
  // across the 4 wavefronts: thread t of wave 0 merges anchor t
  if (wave == 0 && lane < 2) {
    unsigned c_le = 0u, c_ge = 0u;
    L4 P = {KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE}, Sx = P;
#pragma unroll 1
    for (int w = 0; w < 4; w++) {
      const unsigned* v = sh.sweep[w][lane];
      c_le += v[0];
      c_ge += v[1];
#pragma unroll
      for (int i = 0; i < QW; i++) {
        asc_insert(P, v[2 + i]);
        asc_insert(Sx, v[2 + QW + i]);
      }
    }
    unsigned* f = sh.fin[lane];
    f[0] = c_le; f[1] = c_ge;
    f[2] = P.e0; f[3] = P.e1; f[4] = P.e2; f[5] = P.e3;
    f[6] = Sx.e0; f[7] = Sx.e1; f[8] = Sx.e2; f[9] = Sx.e3;
  }
  __syncthreads();
}

// tracker t of the last rebuild_sweep, anchored at g (wave-uniform scalars)
__device__ __forceinline__ QTrack rebuild_finish(const RewardShared& sh, const int t, const unsigned g, const int n) {
  const unsigned* f = sh.fin[t];
  const unsigned c_le = sfl(f[0]), c_ge = sfl(f[1]);
  const unsigned p0 = sfl(f[2]), p1 = sfl(f[3]), p2 = sfl(f[4]), p3 = sfl(f[5]);
  const unsigned s0 = sfl(f[6]), s1 = sfl(f[7]), s2 = sfl(f[8]), s3 = sfl(f[9]);
  QTrack q;
  q.g = g;
  // empty slots (KEY_NONE) satisfy x >= g: take them out of the >= count
  const int n_empty = SDC_HIST_STRIDE - n;
  q.c_le = (int)c_le;
  q.c_lt = n - ((int)c_ge - n_empty);
  // a legitimate predecessor distance is < g; a legitimate successor distance is < KEY_NONE - g - 1
  const unsigned smax = KEY_NONE - g - 1u;
  q.np = (p0 < g) + (p1 < g) + (p2 < g) + (p3 < g);
  q.ns = (s0 < smax) + (s1 < smax) + (s2 < smax) + (s3 < smax);
  q.P.e0 = p0 < g ? g - 1u - p0 : 0u;
  q.P.e1 = p1 < g ? g - 1u - p1 : 0u;
  q.P.e2 = p2 < g ? g - 1u - p2 : 0u;
  q.P.e3 = p3 < g ? g - 1u - p3 : 0u;
  q.S.e0 = s0 < smax ? g + 1u + s0 : KEY_NONE;
  q.S.e1 = s1 < smax ? g + 1u + s1 : KEY_NONE;
  q.S.e2 = s2 < smax ? g + 1u + s2 : KEY_NONE;
  q.S.e3 = s3 < smax ? g + 1u + s3 : KEY_NONE;
  return q;
}

// The key tracker q should re-anchor on so that ranks k, k+1 come inside its window: its own anchor if they already
// are, else the last listed key on the side they left by; 0 if it cannot tell (invalid tracker, empty list).
__device__ __forceinline__ unsigned slide_anchor(const QTrack& q, const int k, const int n) {
  unsigned a, b;
  if (!qt_valid(q)) return 0u;
  if (qt_resolve(q, k, n, a, b)) return q.g;
  const int hi_rank = (k + 1 > n - 1) ? k : k + 1;
  if (hi_rank >= q.c_le + q.ns) return q.ns >= 1 ? lget(q.S, q.ns - 1) : 0u;
  return q.np >= 1 ? lget(q.P, q.np - 1) : 0u;
}

// fp64 sums of v, v^2 over the keys <= gl and over the keys <= gh (gl <= gh < KEY_NONE), block-wide
__device__ __forceinline__ void tail_sums(const uint4* __restrict__ lk, const unsigned gl, const unsigned gh,
                                          RewardShared& sh, const int lane, const int wave, TTrack& tl, TTrack& th) {
  double a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
#pragma unroll 1
  for (int q4 = 0; q4 < SDC_HIST_PER_THREAD / 4; q4++) {
    const uint4 v4 = lk[q4 * SDC_BLOCK];
    const unsigned xs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const unsigned x = xs[c4];
      if (x <= gh) {
        const double v = key_f64(x), vv = v * v;
        b1 += v;
        b2 += vv;
        if (x <= gl) {
          a1 += v;
          a2 += vv;
        }
      }
    }
  }
#pragma unroll 1
  for (int r = 0; r < 4; r++) {   // one copy of the fp64 wave reduction: the four sums rotate through it
    a1 = wave_sum_f64(a1);
    const double t = a1;
    a1 = a2;
    a2 = b1;
    b1 = b2;
    b2 = t;
  }
  if (lane == 0) {
    sh.red_f[wave][0] = a1;
    sh.red_f[wave][1] = a2;
    sh.red_f[wave][2] = b1;
    sh.red_f[wave][3] = b2;
  }
  __syncthreads();
  tl.s1 = (sh.red_f[0][0] + sh.red_f[1][0]) + (sh.red_f[2][0] + sh.red_f[3][0]);
  tl.s2 = (sh.red_f[0][1] + sh.red_f[1][1]) + (sh.red_f[2][1] + sh.red_f[3][1]);
  th.s1 = (sh.red_f[0][2] + sh.red_f[1][2]) + (sh.red_f[2][2] + sh.red_f[3][2]);
  th.s2 = (sh.red_f[0][3] + sh.red_f[1][3]) + (sh.red_f[2][3] + sh.red_f[3][3]);
  __syncthreads();
}

// Clipped mean / population std straight from the ring, fp64, centred on `ctr`: tiny histories and the verify mode.
// Returns {mean, sd}.
__device__ __noinline__ double2 direct_moments(const uint4* __restrict__ lk, const int n, const double lb, const double ub,
                                               const double ctr, RewardShared* shp, const int lane, const int wave) {
  RewardShared& sh = *shp;
  double s = 0.0, s2 = 0.0;
#pragma unroll 1
  for (int q = 0; q < SDC_HIST_PER_THREAD / 4; q++) {
    const uint4 v = lk[q * SDC_BLOCK];
    const unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (x[c] != KEY_NONE) {
        double f = key_f64(x[c]);
        f = f < lb ? lb : (f > ub ? ub : f);  // np.clip
        f -= ctr;
        s += f;
        s2 += f * f;
      }
    }
  }
  s = wave_sum_f64(s);
  s2 = wave_sum_f64(s2);
  if (lane == 0) {
    sh.red_d[wave] = s;
    sh.red_e[wave] = s2;
  }
  __syncthreads();
  const double S1 = (sh.red_d[0] + sh.red_d[1]) + (sh.red_d[2] + sh.red_d[3]);
  const double S2 = (sh.red_e[0] + sh.red_e[1]) + (sh.red_e[2] + sh.red_e[3]);
  __syncthreads();
  const double m0 = S1 / (double)n;
  const double var = S2 / (double)n - m0 * m0;
  double2 r;
  r.x = ctr + m0;
  r.y = var > 0 ? sqrt(var) : 0.0;
  return r;
}

// stage one env's ring in LDS: every load of the workgroup is in flight before the first use; each lane only ever
// reads back its own 10 x 16 bytes (no barrier needed for them)
__device__ __forceinline__ void stage_ring(const unsigned* __restrict__ ring, uint4* __restrict__ keys, const int tid) {
  const uint4* hp = reinterpret_cast<const uint4*>(ring);
  uint4 v[SDC_HIST_PER_THREAD / 4];
#pragma unroll
  for (int q = 0; q < SDC_HIST_PER_THREAD / 4; q++) v[q] = hp[q * SDC_BLOCK + tid];
#pragma unroll
  for (int q = 0; q < SDC_HIST_PER_THREAD / 4; q++) keys[q * SDC_BLOCK + tid] = v[q];
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// The trackers in the header are already post-update (this step's insertion / eviction applied by the dynamics
// kernel).  Grid: min(N, 1024) workgroups, each looping over the queue; the queue length is only known on the
// device, surplus workgroups exit at once.
extern "C" __global__ __launch_bounds__(SDC_BLOCK, 2) void sdc_reward_kernel(SdcDev S, float* __restrict__ rew,
                                                                           float* __restrict__ info) {
  __shared__ RewardShared sh;
  __shared__ uint4 keys[SDC_HIST_STRIDE / 4];   // 40 KB: the env's history ring
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int par = S.step_parity;
  const unsigned env0 = S.work_list[blockIdx.x];   // speculative: issued together with the queue length
  const int cnt = (int)sfl(S.work_cnt[par]);
  if (blockIdx.x == 0 && tid == 0) S.work_cnt[par ^ 1] = 0u;   // the next step's queue
  if ((int)blockIdx.x >= cnt) return;
  if (tid == 0) prof_stamp(S, SDC_PROF_REWARD, blockIdx.x, 0);
  const uint4* lk = keys + tid;

  for (int w = blockIdx.x; w < cnt; w += gridDim.x) {
    const int env = (int)sfl(w == (int)blockIdx.x ? env0 : S.work_list[w]);
    unsigned* hw = S.hdr + (size_t)env * SDC_HDR_DWORDS;
    const unsigned hd0 = hw[lane], hd1 = hw[64 + lane];
    stage_ring(S.hist + (size_t)env * SDC_HIST_STRIDE, keys, tid);
    const int n = rec_i32(hd0, H_N);            // includes this step's energy (appended by the dynamics kernel)
    const double energy = rec_f64(hd0, H_EOFF); // energy - hist_ref, fp64
    Trackers T = trackers_load(hd0, hd1);
    int k1, k3;
    quartile_ranks(n, k1, k3);
    int path = 1;  // diagnostics: 1 ring read (slide / tail re-anchor), 2 bisection + rebuild
    double mean, sd;
    if (n < SMALL_N) {
      // tiny history: everything directly from the ring
      const uint4 qa = quartiles_by_bisection(lk, k1, k3, &sh, lane, wave);
      const Bounds b = clip_bounds(n, qa.x, qa.y, qa.z, qa.w);
      const double2 m = direct_moments(lk, n, b.lb, b.ub, b.ctr, &sh, lane, wave);
      mean = m.x;
      sd = m.y;
      T.q1.g = T.q3.g = T.tl.q.g = T.th.q.g = 0u;
      path = 2;
    } else {
      // quartiles: re-anchor each tracker on the last listed key towards the wanted rank (or keep its anchor)
      unsigned a1 = 0, b1 = 0, a3 = 0, b3 = 0;
      const unsigned g1 = slide_anchor(T.q1, k1, n), g3 = slide_anchor(T.q3, k3, n);
      bool okq = false;
      if (g1 != 0u && g3 != 0u) {
        if (g1 != T.q1.g || g3 != T.q3.g) {
          rebuild_sweep(lk, g1, g3, &sh, lane, wave);
          T.q1 = rebuild_finish(sh, 0, g1, n);
          T.q3 = rebuild_finish(sh, 1, g3, n);
        }
        okq = qt_resolve(T.q1, k1, n, a1, b1) && qt_resolve(T.q3, k3, n, a3, b3);
      }
      if (!okq) {
        // bootstrap (or a tracker that lost its window): exact bisection for the anchors, then the rebuild sweep
        const uint4 qa = quartiles_by_bisection(lk, k1, k3, &sh, lane, wave);
        a1 = sfl(qa.x); b1 = sfl(qa.y); a3 = sfl(qa.z); b3 = sfl(qa.w);
        rebuild_sweep(lk, a1, a3, &sh, lane, wave);
        T.q1 = rebuild_finish(sh, 0, a1, n);
        T.q3 = rebuild_finish(sh, 1, a3, n);
        path = 2;
      }
      const Bounds b = clip_bounds(n, a1, b1, a3, b3);
      int cl = 0, ch = 0;
      double l1 = 0.0, l2 = 0.0, h1 = 0.0, h2 = 0.0;
      if (!(tt_below(T.tl, b.klb, n, cl, l1, l2) && tt_below(T.th, b.kub, n, ch, h1, h2))) {
        // a clip bound has moved past the listed keys (or bootstrap): re-anchor both tail trackers at the bounds
        const unsigned gl = sfl(b.klb - 1u), gh = sfl(b.kub - 1u);
        rebuild_sweep(lk, gl, gh, &sh, lane, wave);
        T.tl.q = rebuild_finish(sh, 0, gl, n);
        T.th.q = rebuild_finish(sh, 1, gh, n);
        tail_sums(lk, gl, gh, sh, lane, wave, T.tl, T.th);
        cl = T.tl.q.c_le; l1 = T.tl.s1; l2 = T.tl.s2;
        ch = T.th.q.c_le; h1 = T.th.s1; h2 = T.th.s2;
      }
      clipped_moments(n, b, cl, l1, l2, ch, h1, h2, mean, sd);
    }
    const double z = (energy - mean) / (sd > 0 ? sd : 1.0);
    if (wave == 0) {
      const Rewards r = step_rewards(z, rec_f64(hd0, H_NORM_CI), rec_f64(hd0, H_OLDEST), (double)rec_i32(hd0, H_OVERDUE), hd0);
      unsigned o0 = hd0, o1 = hd1;
      put_f64(o0, H_RET, r.ret0);
      put_f64(o0, H_RET + 2, r.ret1);
      put_f64(o0, H_RET + 4, r.ret2);
      trackers_put(o0, o1, T);
      if (lane >= H_RET) hw[lane] = o0;
      hw[64 + lane] = o1;
      if (lane == 0) store_rewards(r, z, path, env, rew, info ? info + (size_t)env * SDC_INFO_DIM : nullptr);
    }
    __syncthreads();
  }
  if (tid == 0) prof_stamp(S, SDC_PROF_REWARD, blockIdx.x, 1);
}

// ------------------------------------------------------------------------------------------------
// Verify mode (debug_flags bit 0), after every step and for every env: what the stored (post-step) trackers say --
// quartile keys, clipped mean / std, the z-score that was reported -- against an exact bisection and a direct fp64
// pass over the ring.  A mismatch sets SDC_FAULT_ORDER_STAT in info[fault] and the sticky header bit.
extern "C" __global__ __launch_bounds__(SDC_BLOCK, 2) void sdc_reward_verify_kernel(SdcDev S, float* __restrict__ info) {
  __shared__ RewardShared sh;
  __shared__ uint4 keys[SDC_HIST_STRIDE / 4];
  const int env = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  unsigned* hw = S.hdr + (size_t)env * SDC_HDR_DWORDS;
  float* inf = info + (size_t)env * SDC_INFO_DIM;
  const unsigned hd0 = hw[lane], hd1 = hw[64 + lane];
  const int n = rec_i32(hd0, H_N);
  if (n < SMALL_N) return;
  stage_ring(S.hist + (size_t)env * SDC_HIST_STRIDE, keys, tid);
  const uint4* lk = keys + tid;
  Trackers T = trackers_load(hd0, hd1);
  int k1, k3;
  quartile_ranks(n, k1, k3);
  unsigned a1 = 0, b1 = 0, a3 = 0, b3 = 0;
  bool bad = false;
  if (!qt_resolve(T.q1, k1, n, a1, b1) || !qt_resolve(T.q3, k3, n, a3, b3)) bad = true;
  const uint4 qa = quartiles_by_bisection(lk, k1, k3, &sh, lane, wave);
  if (qa.x != a1 || qa.y != b1 || qa.z != a3 || qa.w != b3) bad = true;
  const Bounds b = clip_bounds(n, qa.x, qa.y, qa.z, qa.w);
  int cl = 0, ch = 0;
  double l1 = 0, l2 = 0, h1 = 0, h2 = 0, mean = 0, sd = 0;
  if (!tt_below(T.tl, b.klb, n, cl, l1, l2) || !tt_below(T.th, b.kub, n, ch, h1, h2)) bad = true;
  clipped_moments(n, b, cl, l1, l2, ch, h1, h2, mean, sd);
  const double2 m = direct_moments(lk, n, b.lb, b.ub, b.ctr, &sh, lane, wave);
  if (!(fabs(m.x - mean) <= 1e-7 * m.y + 1e-9) || !(fabs(m.y - sd) <= 1e-6 * m.y + 1e-9)) bad = true;
  const double z = (rec_f64(hd0, H_EOFF) - m.x) / (m.y > 0 ? m.y : 1.0);
  if (!(fabs((double)inf[SDC_INFO_ENERGY_Z] - z) <= 1e-5 * fabs(z) + 1e-6)) bad = true;
  if (tid == 0 && bad) {
    inf[SDC_INFO_FAULT] = (float)((unsigned)inf[SDC_INFO_FAULT] | SDC_FAULT_ORDER_STAT);
    hw[H_STICKY] = (unsigned)rec_i32(hd0, H_STICKY) | 1u;
  }
}
