// sdc_trackers.hpp -- incremental reward normalisation (utils/reward_creator.py:16-45): the state kept per env so
// that a step normally needs NO pass over the 10 000-entry energy history.
//
// normalize_energy needs the 25th / 75th percentiles of the history (np.percentile, linear: order statistics k and
// k+1 each) and the mean / population std of the history clipped to [lb, ub] = [q1 - 1.5 iqr, q3 + 1.5 iqr].  A
// step inserts one value and evicts at most one:
//   * QUARTILE TRACKERS (QTrack, one per quartile, in the env's 256-byte header): an anchor key G, the exact counts
//     #{x < G}, #{x <= G}, the (up to) 4 largest keys below and 4 smallest keys above G -- a window of ~9
//     consecutive order statistics.  O(1) scalar update per step; the wanted rank moves by at most one per step, so
//     the window is re-centred AHEAD of need by one sweep over the ring (sdc_ringpath.hpp);
//   * TOTAL SUMS A1 = sum v, A2 = sum v^2 over the whole history (fp64, O(1) update);
//   * TAIL SETS: every key above a threshold tau_hi (resp. below tau_lo), unordered, 512 slots each, in global
//     memory; thresholds sit well inside [lb, ub], so the clipped sums are
//        sum clip(v)   = A1 - sum_{v > ub} (v - ub)     - sum_{v < lb} (v - lb)
//        sum clip(v)^2 = A2 - sum_{v > ub} (v^2 - ub^2) - sum_{v < lb} (v^2 - lb^2)
//     with the two correction sums taken over the tail sets only (8 keys per lane and side, one coalesced load).
//     The clip bounds may jump by many keys per step (they move 2.5 x the local spacing at the quartiles), which
//     an ordered window cannot follow; a set does not care.  The lower set is stored COMPLEMENTED (~key), so both
//     sides run the same "keys above a threshold" code.
// Everything here is wave-uniform scalar work or one-wavefront vector work.
#pragma once
#include "sdc_device.hpp"

namespace sdc_rw {

constexpr unsigned KEY_NONE = 0xFFFFFFFFu;  // empty ring slot; also "+infinity" in ascending neighbour lists
constexpr int QW = SDC_QW;
constexpr int SMALL_N = 32;                 // below this the step computes the normalisation directly from the ring

__device__ __forceinline__ unsigned f32_key(float f) {
  const unsigned b = __float_as_uint(f);
  return b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {
  // top bit set (was >= +0): clear it; else (was negative): flip all bits
  const unsigned m = (unsigned)((int)k >> 31);
  return __uint_as_float(k ^ (~m | 0x80000000u));
}
__device__ __forceinline__ double key_f64(unsigned k) { return (double)key_f32(k); }
__device__ __forceinline__ unsigned sfl(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

// ------------------------------------------------------------------------------------------------
// tracker: O(1) maintenance (wave-uniform scalar code)

// 4-entry sorted lists kept as four named scalars (not arrays: LLVM turns unrolled select chains over an array
// back into a dynamically indexed load, which would push the whole tracker into scratch memory).
struct L4 {
  unsigned e0, e1, e2, e3;
};
static_assert(QW == 4, "the tracker lists are written out for 4 entries");

// ascending list of the 4 smallest: insert x (keeps the 4 smallest of list + x)
__device__ __forceinline__ void asc_insert(L4& L, unsigned x) {
  L.e3 = min(max(L.e2, x), L.e3);  // clamp x into [e2, e3] (old values)
  L.e2 = min(max(L.e1, x), L.e2);
  L.e1 = min(max(L.e0, x), L.e1);
  L.e0 = min(L.e0, x);
}
// descending list of the 4 largest
__device__ __forceinline__ void desc_insert(L4& L, unsigned x) {
  L.e3 = max(min(L.e2, x), L.e3);
  L.e2 = max(min(L.e1, x), L.e2);
  L.e1 = max(min(L.e0, x), L.e1);
  L.e0 = max(L.e0, x);
}
__device__ __forceinline__ unsigned lget(const L4& L, int j) {
  return j == 0 ? L.e0 : (j == 1 ? L.e1 : (j == 2 ? L.e2 : L.e3));
}
// remove one occurrence of x from the first `cnt` entries; `fill` pads the tail.  Returns false if absent.
__device__ __forceinline__ bool list_remove(L4& L, int& cnt, unsigned x, unsigned fill) {
  int j = -1;
  if (3 < cnt && L.e3 == x) j = 3;
  if (2 < cnt && L.e2 == x) j = 2;
  if (1 < cnt && L.e1 == x) j = 1;
  if (0 < cnt && L.e0 == x) j = 0;
  if (j < 0) return false;
  if (j <= 0) L.e0 = L.e1;
  if (j <= 1) L.e1 = L.e2;
  if (j <= 2) L.e2 = L.e3;
  L.e3 = fill;
  cnt -= 1;
  return true;
}

struct QTrack {
  unsigned g;          // anchor key; 0 = invalid (no tracker)
  int c_lt, c_le;      // #{x < g}, #{x <= g} over the current ring
  int np, ns;          // valid entries of P / S
  L4 P;                // the np largest keys below g, descending; unused entries 0
  L4 S;                // the ns smallest keys above g, ascending; unused entries KEY_NONE
};

__device__ __forceinline__ QTrack qt_load(unsigned hd, int base) {
  QTrack q;
  q.g = (unsigned)rec_i32(hd, base + T_G);
  q.c_lt = rec_i32(hd, base + T_CLT);
  q.c_le = rec_i32(hd, base + T_CLE);
  q.np = rec_i32(hd, base + T_NP);
  q.ns = rec_i32(hd, base + T_NS);
  q.P.e0 = (unsigned)rec_i32(hd, base + T_P + 0);
  q.P.e1 = (unsigned)rec_i32(hd, base + T_P + 1);
  q.P.e2 = (unsigned)rec_i32(hd, base + T_P + 2);
  q.P.e3 = (unsigned)rec_i32(hd, base + T_P + 3);
  q.S.e0 = (unsigned)rec_i32(hd, base + T_S + 0);
  q.S.e1 = (unsigned)rec_i32(hd, base + T_S + 1);
  q.S.e2 = (unsigned)rec_i32(hd, base + T_S + 2);
  q.S.e3 = (unsigned)rec_i32(hd, base + T_S + 3);
  return q;
}
__device__ __forceinline__ void qt_store(const QTrack& q, unsigned* w) {
  w[T_G] = q.g;
  w[T_CLT] = (unsigned)q.c_lt;
  w[T_CLE] = (unsigned)q.c_le;
  w[T_NP] = (unsigned)q.np;
  w[T_NS] = (unsigned)q.ns;
  w[T_P + 0] = q.P.e0;
  w[T_P + 1] = q.P.e1;
  w[T_P + 2] = q.P.e2;
  w[T_P + 3] = q.P.e3;
  w[T_S + 0] = q.S.e0;
  w[T_S + 1] = q.S.e1;
  w[T_S + 2] = q.S.e2;
  w[T_S + 3] = q.S.e3;
}

// Remove one occurrence of x_old from a tracker.  Sets q.g = 0 if the tracker turns out to be inconsistent.
__device__ __forceinline__ void qt_evict(QTrack& q, unsigned x_old) {
  if (x_old < q.g) {
    q.c_lt -= 1;
    q.c_le -= 1;
    // the list holds exactly the np largest keys below g: the evicted key is in it iff it is >= the smallest listed
    if (q.np > 0 && x_old >= lget(q.P, q.np - 1)) {
      if (!list_remove(q.P, q.np, x_old, 0u)) q.g = 0u;
    }
  } else if (x_old == q.g) {
    q.c_le -= 1;
  } else {
    if (q.ns > 0 && x_old <= lget(q.S, q.ns - 1)) {
      if (!list_remove(q.S, q.ns, x_old, KEY_NONE)) q.g = 0u;
    }
  }
}
// Add x_new to a tracker that describes m keys.
__device__ __forceinline__ void qt_insert(QTrack& q, unsigned x_new, int m) {
  if (x_new < q.g) {
    const bool complete = q.np == q.c_lt;  // every key below g is listed
    q.c_lt += 1;
    q.c_le += 1;
    if (complete || (q.np > 0 && x_new > lget(q.P, q.np - 1))) {
      desc_insert(q.P, x_new);
      q.np = min(QW, q.np + 1);
    }
  } else if (x_new == q.g) {
    q.c_le += 1;
  } else {
    const bool complete = q.ns == m - q.c_le;  // every key above g is listed
    if (complete || (q.ns > 0 && x_new < lget(q.S, q.ns - 1))) {
      asc_insert(q.S, x_new);
      q.ns = min(QW, q.ns + 1);
    }
  }
}
// Apply this step's eviction (x_old, if has_old) and insertion (x_new) to a tracker that described the ring of
// the previous step, which held n_prev keys.
__device__ __forceinline__ void qt_update(QTrack& q, unsigned x_new, unsigned x_old, bool has_old, int n_prev) {
  if (has_old) qt_evict(q, x_old);
  qt_insert(q, x_new, has_old ? n_prev - 1 : n_prev);
}

// key at rank r, if the window covers it
__device__ __forceinline__ bool qt_value_at(const QTrack& q, int r, unsigned& out) {
  if (r >= q.c_lt && r < q.c_le) {
    out = q.g;
    return true;
  }
  if (r < q.c_lt) {
    const int j = q.c_lt - 1 - r;
    if (j >= q.np) return false;
    out = lget(q.P, j);
    return true;
  }
  const int j = r - q.c_le;
  if (j >= q.ns) return false;
  out = lget(q.S, j);
  return true;
}
// ranks k and k+1 (the second only if it exists)
__device__ __forceinline__ bool qt_resolve(const QTrack& q, int k, int n, unsigned& a, unsigned& b) {
  if (q.g == 0u || q.g == KEY_NONE) return false;
  if (!qt_value_at(q, k, a)) return false;
  if (k + 1 > n - 1) {
    b = a;
    return true;
  }
  return qt_value_at(q, k + 1, b);
}
// header write-back: lane i of `o` holds dword i
__device__ __forceinline__ void put_u32(unsigned& o, int idx, unsigned v) {
  const unsigned sv = sfl(v);
  asm("v_writelane_b32 %0, %1, %2" : "+v"(o) : "s"(sv), "n"(idx));
}
__device__ __forceinline__ void put_f64(unsigned& o, int idx, double v) {
  put_u32(o, idx, (unsigned)__double2loint(v));
  put_u32(o, idx + 1, (unsigned)__double2hiint(v));
}
__device__ __forceinline__ void qt_put(unsigned& o, int base, const QTrack& q) {
  put_u32(o, base + T_G, q.g);
  put_u32(o, base + T_CLT, (unsigned)q.c_lt);
  put_u32(o, base + T_CLE, (unsigned)q.c_le);
  put_u32(o, base + T_NP, (unsigned)q.np);
  put_u32(o, base + T_NS, (unsigned)q.ns);
  put_u32(o, base + T_P + 0, q.P.e0);
  put_u32(o, base + T_P + 1, q.P.e1);
  put_u32(o, base + T_P + 2, q.P.e2);
  put_u32(o, base + T_P + 3, q.P.e3);
  put_u32(o, base + T_S + 0, q.S.e0);
  put_u32(o, base + T_S + 1, q.S.e1);
  put_u32(o, base + T_S + 2, q.S.e2);
  put_u32(o, base + T_S + 3, q.S.e3);
}

// ------------------------------------------------------------------------------------------------
// quartile values (numpy _lerp), clip bounds, and the bounds in key space
struct Bounds {
  double lb, ub, ctr;
  unsigned klb, kub;  // klb = smallest key whose value is >= lb, kub = smallest key whose value is > ub
};
__device__ __forceinline__ void quartile_ranks(const int n, int& k1, int& k3) {
  k1 = (n - 1) >> 2;        // floor((n-1) * 0.25), np.percentile 'linear'
  k3 = (3 * (n - 1)) >> 2;  // floor((n-1) * 0.75)
}
__device__ __forceinline__ Bounds clip_bounds(const int n, unsigned a1, unsigned b1, unsigned a3, unsigned b3) {
  const double t1 = (double)((n - 1) & 3) * 0.25, t3 = (double)((3 * (n - 1)) & 3) * 0.25;
  const double fa1 = key_f64(a1), fb1 = key_f64(b1);
  const double fa3 = key_f64(a3), fb3 = key_f64(b3);
  // numpy _lerp: a + (b-a)*t, and b - (b-a)*(1-t) where t >= 0.5
  const double d1 = fb1 - fa1, d3 = fb3 - fa3;
  const double qv1 = (t1 == 0.0) ? fa1 : ((t1 >= 0.5) ? fb1 - d1 * (1.0 - t1) : fa1 + d1 * t1);
  const double qv3 = (t3 == 0.0) ? fa3 : ((t3 >= 0.5) ? fb3 - d3 * (1.0 - t3) : fa3 + d3 * t3);
  const double iqr = qv3 - qv1;
  Bounds b;
  b.lb = qv1 - 1.5 * iqr;
  b.ub = qv3 + 1.5 * iqr;
  b.ctr = 0.5 * (qv1 + qv3);
  const float lbf = (float)b.lb, ubf = (float)b.ub;
  unsigned klb = f32_key(lbf) + (((double)lbf < b.lb) ? 1u : 0u);
  unsigned kub = f32_key(ubf) + (((double)ubf <= b.ub) ? 1u : 0u);
  klb = min(max(klb, 2u), KEY_NONE - 2u);
  b.klb = klb;
  b.kub = min(max(kub, klb), KEY_NONE - 2u);
  return b;
}
// clipped mean / std from the total sums and the tail corrections T1 = sum_{tails} (v - bound), T2 = sum (v^2 - bound^2)
// (n_full, rc_full: the history capacity and its reciprocal -- once the ring is full, n is that constant and the two
// divisions take the 3-instruction form)
__device__ __forceinline__ void clipped_moments(const int n, const Bounds& b, const double A1, const double A2, const double T1,
                                                const double T2, double& mean, double& sd, const int n_full = 0,
                                                const double rc_full = 0.0) {
  const double C1 = A1 - T1, C2 = A2 - T2;
  double m2;
  if (n == n_full) {
    mean = sdc_div_const(C1, (double)n, rc_full);
    m2 = sdc_div_const(C2, (double)n, rc_full);
  } else {
    mean = C1 / (double)n;
    m2 = C2 / (double)n;
  }
  const double var = m2 - mean * mean;
  sd = (var > 0 && b.ub > b.lb) ? sqrt(var) : 0.0;
}

__device__ __forceinline__ bool qt_valid(const QTrack& q) { return q.g != 0u && q.g != KEY_NONE; }

// ------------------------------------------------------------------------------------------------
// TAIL SETS.  Per env and side 512 slots (SDC_TAIL_CAP) in global memory, lane l owning slots [8 l, 8 l + 8) as
// two uint4.  Side 0 = upper tail, keys as they are; side 1 = lower tail, keys complemented.  In this "flipped"
// space both sides hold every key > tau; an empty slot is 0.
struct TailSet {
  unsigned k[8];   // this lane's 8 slots
};
constexpr unsigned TAIL_EMPTY = 0u;
__device__ __forceinline__ unsigned tail_flip(int side) { return side ? KEY_NONE : 0u; }
__device__ __forceinline__ TailSet tail_load(const uint4* __restrict__ p, const int lane) {
  const uint4 a = p[2 * lane], b = p[2 * lane + 1];
  TailSet s;
  s.k[0] = a.x; s.k[1] = a.y; s.k[2] = a.z; s.k[3] = a.w;
  s.k[4] = b.x; s.k[5] = b.y; s.k[6] = b.z; s.k[7] = b.w;
  return s;
}
__device__ __forceinline__ void tail_store(uint4* __restrict__ p, const int lane, const TailSet& s) {
  p[2 * lane] = make_uint4(s.k[0], s.k[1], s.k[2], s.k[3]);
  p[2 * lane + 1] = make_uint4(s.k[4], s.k[5], s.k[6], s.k[7]);
}

// remove one occurrence of key x (flipped space) from the set; returns false if it is not there
__device__ __forceinline__ bool tail_remove(TailSet& s, const unsigned x, const int lane) {
  int c = -1;
#pragma unroll
  for (int i = 7; i >= 0; i--) c = (s.k[i] == x) ? i : c;
  const unsigned long long m = __ballot(c >= 0);
  if (m == 0ull) return false;
  const int owner = __ffsll((long long)m) - 1;
#pragma unroll
  for (int i = 0; i < 8; i++) s.k[i] = (lane == owner && i == c) ? TAIL_EMPTY : s.k[i];
  return true;
}
// put key x (flipped space) into an empty slot; returns false if the set is full
__device__ __forceinline__ bool tail_insert(TailSet& s, const unsigned x, const int lane) {
  int c = -1;
#pragma unroll
  for (int i = 7; i >= 0; i--) c = (s.k[i] == TAIL_EMPTY) ? i : c;
  const unsigned long long m = __ballot(c >= 0);
  if (m == 0ull) return false;
  const int owner = __ffsll((long long)m) - 1;
#pragma unroll
  for (int i = 0; i < 8; i++) s.k[i] = (lane == owner && i == c) ? x : s.k[i];
  return true;
}
// this lane's share of sum (v - bound), sum (v^2 - bound^2) over the set keys >= kb (kb, keys in flipped space)
__device__ __forceinline__ void tail_scan(const TailSet& s, const unsigned kb, const unsigned flip, const double bound,
                                          double& t1, double& t2) {
  const double b2 = bound * bound;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (s.k[i] >= kb) {   // kb >= 1 > TAIL_EMPTY
      const double v = key_f64(s.k[i] ^ flip);
      t1 += v - bound;
      t2 += v * v - b2;
    }
  }
}
// this lane's share of (count, sum v, sum v^2) over the set keys in [lo, hi) (flipped space): the keys a clip bound
// crosses when it moves from one to the other.  Returns false (and leaves c / s1 / s2 alone) if no lane of the
// wavefront has such a key -- the common case, one compare pair per slot and no fp64 work.
__device__ __forceinline__ bool tail_crossing(const TailSet& s, const unsigned lo, const unsigned hi, const unsigned flip, int& c,
                                              double& s1, double& s2) {
  bool any = false;
#pragma unroll
  for (int i = 0; i < 8; i++) any = any || (s.k[i] >= lo && s.k[i] < hi);
  if (__ballot(any) == 0ull) return false;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (s.k[i] >= lo && s.k[i] < hi) {
      const double v = key_f64(s.k[i] ^ flip);
      c += 1;
      s1 += v;
      s2 += v * v;
    }
  }
  return true;
}
// header record of the running tail sums (H_QC / H_QS1 / H_QS2_*)
__device__ __forceinline__ void put_running_tails(unsigned& o, const int c_hi, const int c_lo, const double s1_hi,
                                                  const double s1_lo, const double s2_hi, const double s2_lo) {
  put_u32(o, H_QC, (unsigned)c_hi);
  put_u32(o, H_QC + 1, (unsigned)c_lo);
  put_f64(o, H_QS1, s1_hi);
  put_f64(o, H_QS1 + 2, s1_lo);
  put_f64(o, H_QS2_HI, s2_hi);
  put_f64(o, H_QS2_LO, s2_lo);
}
// this lane's number of set keys below kb (the slack between the threshold and the clip bound)
__device__ __forceinline__ unsigned tail_count_below(const TailSet& s, const unsigned kb) {
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) c += (s.k[i] != TAIL_EMPTY && s.k[i] < kb) ? 1u : 0u;
  return c;
}
// step of a threshold move: the key distance that holds ~128 keys, from `keys` keys found within `dist`
__device__ __forceinline__ unsigned band_estimate(const unsigned dist, const int keys) {
  const unsigned long long b = ((unsigned long long)dist * 128ull) / (unsigned long long)max(keys, 16);
  return (unsigned)min(b, (unsigned long long)max(dist, 1u));
}
// raise the threshold to tau2 (flipped space): drop the keys <= tau2; returns the new count
__device__ __forceinline__ int tail_raise(TailSet& s, const unsigned tau2) {
  int c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    s.k[i] = (s.k[i] <= tau2) ? TAIL_EMPTY : s.k[i];
    c += (s.k[i] != TAIL_EMPTY) ? 1 : 0;
  }
  return wave_sum_i32(c);
}

// rewards (utils/reward_creator.py:48-334) from the z-score and the step's physical quantities; running returns
struct RewardIn {
  double z;             // normalize_energy(bat_total_energy_with_battery_KWh)
  double norm_ci_next;  // norm_CI
  double oldest_norm;   // ls_oldest_task_age
  double overdue;       // ls_overdue_penalty
  double energy_kwh;    // bat_total_energy_with_battery_KWh
  double hour;          // hour of the day after the step (reward_params["hour"])
  double ite_kw, total_kw, water;   // dc_ITE_total_power_kW, dc_total_power_kW, dc_water_usage
};
struct Rewards {
  double r[3], ret[3];
};
__device__ __forceinline__ double tou_price(const int h) {   // reward_creator.py:166-189
  return h < 6 ? 0.25 : (h < 11 ? 0.41 : (h < 16 ? 0.30 : (h < 22 ? 0.27 : 0.25)));
}
__device__ __forceinline__ Rewards step_rewards(const RewardIn& in, const int (&method)[3], const unsigned hd0) {
  const double foot = -1.0 * (in.norm_ci_next * in.z / 0.50);
  const double overdue_pen = -0.3 * sqrt(in.overdue) + 0.3;
  const double age_pen = -0.1 * in.oldest_norm;
  double rls = foot + overdue_pen + age_pen;
  rls = rls < -10 ? -10 : (rls > 10 ? 10 : rls);
  Rewards o;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    double r;
    switch (method[a]) {   // wave-uniform
      case SDC_REWARD_DEFAULT: r = a == 0 ? rls : foot; break;
      case SDC_REWARD_FOOTPRINT: r = foot; break;
      case SDC_REWARD_TOU: r = -1.0 * in.energy_kwh * tou_price((int)in.hour % 24); break;
      case SDC_REWARD_ENERGY_EFFICIENCY: r = in.ite_kw / in.total_kw; break;
      case SDC_REWARD_PUE: r = -fabs((in.ite_kw != 0 ? in.total_kw / in.ite_kw : (double)INFINITY) - 1); break;
      case SDC_REWARD_WATER: r = -0.01 * in.water; break;
      default: r = 0.0;   // SDC_REWARD_CUSTOM: custom_agent_reward returns 0
    }
    o.r[a] = r;
    o.ret[a] = rec_f64(hd0, H_RET + 2 * a) + r;
  }
  return o;
}
// lane 0 writes the step's rewards and the reward-side info columns
__device__ __forceinline__ void store_rewards(const Rewards& r, const double z, const int path, const int env,
                                              float* __restrict__ rew, float* __restrict__ inf_row) {
  rew[env * 3 + 0] = (float)r.r[0];
  rew[env * 3 + 1] = (float)r.r[1];
  rew[env * 3 + 2] = (float)r.r[2];
  if (inf_row) {
    inf_row[SDC_INFO_ENERGY_Z] = (float)z;
    inf_row[SDC_INFO_RESERVED] = (float)path;   // diagnostic: 0 no ring read, 1 slid ahead of need, 2 tail set re-collected, 3 rebuilt
    inf_row[SDC_INFO_EP_RETURN_LS] = (float)r.ret[0];
    inf_row[SDC_INFO_EP_RETURN_DC] = (float)r.ret[1];
    inf_row[SDC_INFO_EP_RETURN_BAT] = (float)r.ret[2];
  }
}

}  // namespace sdc_rw
