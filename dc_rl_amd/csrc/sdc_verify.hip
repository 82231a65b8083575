// sdc_verify.hip -- verify mode of the reward normalisation (sdc_config.debug_flags bit 0; every GPU parity test
// runs with it): after each step, one workgroup per env recomputes the order statistics by exact bisection on the
// key space and the clipped mean / std by a direct fp64 pass over the env's history ring (staged in LDS), and
// compares them with the env's rank windows and running sums (sdc_trackers.hpp) and with the
// z-score that was reported.
// Measurement / test infrastructure only: never launched when debug_flags is 0.
#include "sdc_trackers.hpp"

namespace {

using namespace sdc_rw;

struct RewardShared {
  unsigned red_u[2][4];
  unsigned red_v[2][4];
  double red_d[4];
  double red_e[4];
};

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
  const unsigned hd0 = hw[lane];
  const int n = rec_i32(hd0, H_N);
  if (n < SMALL_N) return;
  stage_ring(S.hist + (size_t)env * SDC_HIST_STRIDE, keys, tid);
  const uint4* lk = keys + tid;
  const uint4 qw = reinterpret_cast<const uint4*>(S.qwin)[(size_t)env * SDC_WIN + lane];
  const QTrack q1 = qt_load(hd0, H_Q1, qw.x), q3 = qt_load(hd0, H_Q3, qw.y);
  const QTrack bu = qt_load(hd0, H_BU, qw.z), bl = qt_load(hd0, H_BL, qw.w);
  int k1, k3;
  quartile_ranks(n, k1, k3);
  unsigned a1 = 0, b1 = 0, a3 = 0, b3 = 0;
  bool bad = rec_i32(hd0, H_VALID) != 1;
  // 1. the quartile windows against an exact bisection ...
  if (!qt_resolve(q1, k1, n, a1, b1) || !qt_resolve(q3, k3, n, a3, b3)) bad = true;
  const uint4 qa = quartiles_by_bisection(lk, k1, k3, &sh, lane, wave);
  if (qa.x != a1 || qa.y != b1 || qa.z != a3 || qa.w != b3) bad = true;
  // ... and every key of all four windows against its rank: lane i's key v must satisfy #{x < v} <= r0 + i < #{x <= v}
  // (one window per wavefront, one window key per lane, all ring keys from LDS; the lower clip bound's window lives on
  // complemented keys)
  {
    const QTrack& q = wave == 0 ? q1 : (wave == 1 ? q3 : (wave == 2 ? bu : bl));
    const unsigned flip = wave == 3 ? KEY_NONE : 0u;
    if (q.hi <= 0 || q.hi > SDC_WIN || q.r0 < 0 || q.r0 + q.hi > n) bad = true;
    else if (lane_key(q.w, 0) != (unsigned)rec_i32(hd0, H_WFIRST + wave) || lane_key(q.w, q.hi - 1) != (unsigned)rec_i32(hd0, H_WLAST + wave))
      bad = true;      // the cached first / last key of the window (what the step's outside-the-window test reads)
    else if (lane < q.hi) {
      int clt = 0, cle = 0;
      for (int j = 0; j < SDC_HIST_STRIDE / 4; j++) {
        const uint4 v = keys[j];
        const unsigned x[4] = {v.x, v.y, v.z, v.w};
        for (int i = 0; i < 4; i++) {
          if (x[i] == KEY_NONE) continue;   // empty slot
          clt += (x[i] ^ flip) < q.w ? 1 : 0;
          cle += (x[i] ^ flip) <= q.w ? 1 : 0;
        }
      }
      if (!(clt <= q.r0 + lane && q.r0 + lane < cle)) bad = true;
    } else if (q.w != KEY_NONE) bad = true;
  }
  bad = __syncthreads_or(bad ? 1 : 0) != 0;
  // 2. the reported z-score against a direct fp64 pass over the ring
  const Bounds b = clip_bounds(n, qa.x, qa.y, qa.z, qa.w);
  const double2 m = direct_moments(lk, n, b.lb, b.ub, b.ctr, &sh, lane, wave);
  const double z = (rec_f64(hd0, H_EOFF) - m.x) / (m.y > 0 ? m.y : 1.0);
  if (!(fabs((double)inf[SDC_INFO_ENERGY_Z] - z) <= 2e-6 * fabs(z) + 1e-6)) bad = true;
  // 3. the running tail counts against the ring (the sums that go with them are covered by the z check above), and
  //    the running total against the direct one
  {
    unsigned cq = 0;                  // packed counts (hi << 16 | lo) of the keys at or beyond the stored clip bounds
    const unsigned kbs0 = (unsigned)rec_i32(hd0, H_KB), kbs1 = (unsigned)rec_i32(hd0, H_KB + 1);
    double s1 = 0.0;
    for (int q = 0; q < SDC_HIST_PER_THREAD / 4; q++) {
      const uint4 v = lk[q * SDC_BLOCK];
      const unsigned x[4] = {v.x, v.y, v.z, v.w};
      for (int i = 0; i < 4; i++) {
        if (x[i] == KEY_NONE) continue;
        if (x[i] >= kbs0) cq += 0x10000u;
        if (~x[i] >= kbs1) cq += 1u;
        s1 += key_f64(x[i]);
      }
    }
    cq = block_sum_u32(cq, sh.red_v, 0, wave, lane);
    if (cq != (((unsigned)rec_i32(hd0, H_QC) << 16) | ((unsigned)rec_i32(hd0, H_QC + 1) & 0xFFFFu))) bad = true;
    s1 = wave_sum_f64(s1);
    __syncthreads();
    if (lane == 0) sh.red_d[wave] = s1;
    __syncthreads();
    s1 = (sh.red_d[0] + sh.red_d[1]) + (sh.red_d[2] + sh.red_d[3]);
    if (!(fabs(s1 - rec_f64(hd0, H_A1)) <= 1e-9 * (fabs(s1) + (double)n))) bad = true;
  }
  if (tid == 0 && bad) {
    inf[SDC_INFO_FAULT] = (float)((unsigned)inf[SDC_INFO_FAULT] | SDC_FAULT_ORDER_STAT);
    hw[H_STICKY] = (unsigned)rec_i32(hd0, H_STICKY) | 1u;
  }
}
