// sdc_trackers.hpp -- incremental reward normalisation (utils/reward_creator.py:16-45): the state kept per env so
// that a step normally needs NO pass over the 10 000-entry energy history.
//
// normalize_energy needs the 25th / 75th percentiles of the history (np.percentile, linear: order statistics k and
// k+1 each) and the mean / population std of the history clipped to [lb, ub] = [q1 - 1.5 iqr, q3 + 1.5 iqr].  A
// step inserts one value and evicts at most one:
//   * RANK WINDOWS (QTrack): 64 consecutive order statistics around a wanted rank, one key per lane (four windows =
//     1 KB per env next to the header).  O(1) update per step across the lanes; the wanted rank moves by at most
//     one per step, so a window is re-centred AHEAD of need, every ~600 steps, by one sweep over the ring
//     (sdc_ringpath.hpp).  Two of them sit on the quartile ranks;
//   * TOTAL SUMS A1 = sum v, A2 = sum v^2 over the whole history (fp64, O(1) update);
//   * RUNNING TAIL SUMS: per side (count, sum v, sum v^2) over the keys beyond the clip bound, so that
//        sum clip(v)   = A1 - sum_{v > ub} (v - ub)     - sum_{v < lb} (v - lb)
//        sum clip(v)^2 = A2 - sum_{v > ub} (v^2 - ub^2) - sum_{v < lb} (v^2 - lb^2)
//     cost a few multiply-adds; and two more rank windows, one around each clip bound, which list the keys a bound
//     moves across from one step to the next (the bounds move with the quartiles, a fraction of a key per step out
//     in the tails).  However heavy a tail is, it is only ever a count and two sums.
// Everything here is wave-uniform scalar work or one-wavefront vector work.
#pragma once
#include "sdc_device.hpp"

namespace sdc_rw {

constexpr unsigned KEY_NONE = 0xFFFFFFFFu;  // empty ring slot; also "+infinity" in ascending neighbour lists
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
// DPP lane moves (one VALU instruction each; a lane without a source keeps `identity`)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned identity, unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xF, false);
}
// lane `src`'s value of v (any lane pattern: ds_bpermute).  Not __shfl: that adds the caller's own lane id (for widths below
// 64), and inside the multi-step kernels that loop-invariant lane id was hoisted out of the step loop and SPILLED (round 4).
__device__ __forceinline__ unsigned lane_gather(unsigned v, int src) { return (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)v); }
// A double constant of a COLD path, materialised where it is used: a plain literal is hoisted out of the multi-step kernels'
// step loop (two VGPRs held -- or spilled -- across the whole hot path for a branch almost never taken).
template <unsigned HI, unsigned LO>
__device__ __forceinline__ double cold_f64() {
  unsigned lo, hi;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(lo), "=v"(hi) : "i"(LO), "i"(HI));
  return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ unsigned lane_key(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, (int)sfl((unsigned)l)); }

// ------------------------------------------------------------------------------------------------
// RANK WINDOW: SDC_WIN = 64 CONSECUTIVE order statistics of the env's history, one key per lane.
// Lane i < hi holds the key of rank r0 + i (ascending); lanes >= hi hold KEY_NONE.  hi == 0: no window.
// A step removes one key from the history and adds one; each lands below the window (the ranks move: r0 -/+ 1),
// above it (nothing changes) or inside it (one compare across the lanes, one DPP shift).  The wanted rank
// random-walks through the window by about +-0.6 per step (more under autocorrelated energies), so a window
// re-centred on it (32 ranks of room on both sides) lasts ~600 steps before the ring has to be read again
// (qt_refill in sdc_ringpath.hpp).
constexpr int WIN = SDC_WIN;
struct QTrack {
  unsigned w;   // this lane's key
  int r0;       // rank of lane 0's key in the sorted history
  int hi;       // valid lanes
};
__device__ __forceinline__ bool qt_valid(const QTrack& q) { return q.hi > 0; }
__device__ __forceinline__ QTrack qt_load(unsigned hd, int base, unsigned w) {
  QTrack q;
  q.w = w;
  q.r0 = rec_i32(hd, base + T_R0);
  q.hi = rec_i32(hd, base + T_HI);
  return q;
}
// Remove one occurrence of y from the history a tracker describes.  Returns true if the window's lanes changed.
// An inconsistent tracker (y lies inside the window's span but is not in it) is dropped: hi = 0.
__device__ __forceinline__ bool qt_evict(QTrack& q, const unsigned y, const int lane) {
  const int m = __popcll(__ballot(q.w < y));   // valid keys below y (KEY_NONE lanes never count)
  if (m >= q.hi) return false;                 // above the window: no rank inside it moves
  if (__builtin_expect(lane_key(q.w, m) != y, 1)) {
    if (m == 0) q.r0 -= 1;                     // below the window: every rank inside it moves down
    else q.hi = 0;
    return false;
  }
  // inside (equal keys are interchangeable: take the first): close the gap from above
  const unsigned down = dpp_u32<0x130, 0xF>(KEY_NONE, q.w);   // wave_shl:1 -- lane i <- lane i + 1
  q.w = lane >= m ? down : q.w;
  q.hi -= 1;
  return true;
}
// Add x to a history of m keys that the tracker describes.  Returns true if the window's lanes changed.
// A window that starts (ends) the history keeps doing so: a key below (above) all of it still enters, and the key at
// the far end drops out if the window is full.
__device__ __forceinline__ bool qt_insert(QTrack& q, const unsigned x, const int m, const int lane) {
  const int p = __popcll(__ballot(q.w <= x));  // valid keys <= x: x belongs at lane p
  if (p == 0 && q.r0 != 0) {                   // below the window: every rank inside it moves up
    q.r0 += 1;
    return false;
  }
  const bool ends = q.r0 + q.hi == m;          // the window lists the history's last key
  if (p == q.hi && !ends) return false;        // above the window and of keys it does not list: no rank inside it moves
  if (q.hi == WIN && ends) {
    // full, and it must go on ending the history: x enters at lane p - 1, the keys below it move down, the first drops out
    const unsigned down = dpp_u32<0x130, 0xF>(KEY_NONE, q.w);   // wave_shl:1 -- lane i <- lane i + 1
    q.w = lane < p - 1 ? down : (lane == p - 1 ? x : q.w);
    q.r0 += 1;
    return true;
  }
  const unsigned up = dpp_u32<0x138, 0xF>(0u, q.w);             // wave_shr:1 -- lane i <- lane i - 1; lane 63's key drops out
  q.w = lane < p ? q.w : (lane == p ? x : up);
  q.hi = min(WIN, q.hi + 1);
  return true;
}
// this step's eviction (x_old, if has_old) and insertion (x_new) on a tracker of the previous step's n_prev keys
__device__ __forceinline__ bool qt_update(QTrack& q, unsigned x_new, unsigned x_old, bool has_old, int n_prev, int lane) {
  bool ch = false;
  if (has_old) ch = qt_evict(q, x_old, lane);
  if (q.hi > 0) ch = qt_insert(q, x_new, has_old ? n_prev - 1 : n_prev, lane) || ch;
  return ch;
}
// keys at ranks k and k+1 (the second only if it exists)
__device__ __forceinline__ bool qt_resolve(const QTrack& q, int k, int n, unsigned& a, unsigned& b) {
  const int t = k - q.r0;
  const int tb = (k + 1 > n - 1) ? t : t + 1;
  if (q.hi <= 0 || t < 0 || tb >= q.hi) return false;
  a = lane_key(q.w, t);
  b = lane_key(q.w, tb);
  return true;
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
  put_u32(o, base + T_R0, (unsigned)q.r0);
  put_u32(o, base + T_HI, (unsigned)q.hi);
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
  // (both forms evaluated, then selected: written as nested conditionals this was two divergent branches per quartile)
  const double up1 = fb1 - d1 * (1.0 - t1), lo1 = fa1 + d1 * t1, up3 = fb3 - d3 * (1.0 - t3), lo3 = fa3 + d3 * t3;
  const double m1 = (t1 >= 0.5) ? up1 : lo1, m3 = (t3 >= 0.5) ? up3 : lo3;
  const double qv1 = (t1 == 0.0) ? fa1 : m1;
  const double qv3 = (t3 == 0.0) ? fa3 : m3;
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
// 1 / sqrt(v) for v > 0 (fp32-representable magnitude): hardware v_rsq_f32 + two Newton steps in fp64 (<= 3e-16 relative)
// -- 10 instructions where sqrt() and the division by it are 40
__device__ __forceinline__ double inv_sqrt_pos(const double v) {
  double y = (double)__builtin_amdgcn_rsqf((float)v);
  const double hv = 0.5 * v;
  y = y * fma(-hv * y, y, 1.5);
  y = y * fma(-hv * y, y, 1.5);
  return y;
}
// sqrt of a small non-negative integer count, to fp32 accuracy (the overdue penalty: -0.3 sqrt(n) + 0.3)
__device__ __forceinline__ double sqrt_count(const double n) { return (double)__builtin_sqrtf((float)n); }

// clipped mean / std from the total sums and the tail corrections T1 = sum_{tails} (v - bound), T2 = sum (v^2 - bound^2)
// (n_full, rc_full: the history capacity and its reciprocal -- once the ring is full, n is that constant and the two
// divisions take the 3-instruction form).  inv_sd = 1 / sd, or 1 when sd is 0 (the z-score's divisor, reward_creator.py:44)
__device__ __forceinline__ void clipped_moments(const int n, const Bounds& b, const double A1, const double A2, const double T1,
                                                const double T2, double& mean, double& sd, double& inv_sd, const int n_full = 0,
                                                const double rc_full = 0.0, const double n_full_d = 0.0) {
  const double C1 = A1 - T1, C2 = A2 - T2;
  double m2;
  if (n == n_full) {
    mean = sdc_div_const(C1, n_full_d, rc_full);      // (n_full_d == (double)n here, handed in as a scalar)
    m2 = sdc_div_const(C2, n_full_d, rc_full);
  } else {
    mean = C1 / (double)n;
    m2 = C2 / (double)n;
  }
  const double var = m2 - mean * mean;
  const bool pos = var > 1e-30 && b.ub > b.lb;
  const double y = inv_sqrt_pos(pos ? var : 1.0);
  sd = pos ? var * y : 0.0;
  inv_sd = pos ? y : 1.0;
}


// ------------------------------------------------------------------------------------------------
// CLIP-BOUND WINDOWS and the running tail sums.  The clipped sums need, per side, (count, sum v, sum v^2) over the
// keys at or beyond the clip bound; they are kept as running values (header H_QC / H_QS1 / H_QS2_*).  A step adjusts
// them for the value appended and the one evicted, and then for the keys the bound has moved across -- which the
// side's rank window (a QTrack centred on the rank of the first key beyond the bound) lists, as long as both the
// old and the new bound lie inside its span.  The lower side works on complemented keys, so both sides run the same
// "keys at or above a bound" code.

// does the window list EVERY history key in [lo, hi)?  (it does if lo lies above its first key -- or the window starts
// the history -- and hi does not lie above its last key -- or the window ends the history)
__device__ __forceinline__ bool qt_spans(const QTrack& q, const unsigned lo, const unsigned hi, const int n) {
  if (q.hi <= 0) return false;
  const bool lo_ok = q.r0 == 0 || lane_key(q.w, 0) < lo;
  const bool hi_ok = q.r0 + q.hi >= n || hi <= lane_key(q.w, q.hi - 1);
  return lo_ok && hi_ok;
}
// this lane's share of (count, sum v, sum v^2) over the window keys in [lo, hi) (flipped space): the keys a clip bound
// crosses when it moves from one to the other.  Returns false (and leaves c / s1 / s2 alone) if no lane of the
// wavefront has such a key -- the common case: two compares, a ballot and no fp64 work.
__device__ __forceinline__ bool win_crossing(const QTrack& q, const unsigned lo, const unsigned hi, const unsigned flip, int& c,
                                             double& s1, double& s2) {
  const bool in = q.w >= lo && q.w < hi;   // (KEY_NONE lanes: hi <= KEY_NONE)
  if (__builtin_expect(__ballot(in) == 0ull, 1)) return false;
  if (in) {
    const double v = key_f64(q.w ^ flip);
    c = 1;
    s1 = v;
    s2 = v * v;
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

// ------------------------------------------------------------------------------------------------
// per-agent rewards
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
// The agents' rewards, once for every mapping of the step (utils/reward_creator.py): the footprint term and default_ls_reward (:48-130),
// then one agent slot's reward by its configured method (:154-334).  COLD: the whole-wavefront fallback's copy -- its two non-inline
// literals are built where they are used instead of being held in registers across the step.
template <bool COLD = false>
__device__ __forceinline__ void reward_terms(const double z, const double norm_ci_next, const double overdue, const double oldest_norm,
                                             double& foot, double& rls) {
  foot = -1.0 * (norm_ci_next * z / 0.50);
  const double overdue_pen = -0.3 * sqrt_count(overdue) + 0.3;
  double age_pen;
  if constexpr (COLD) age_pen = cold_f64<0xBFB99999u, 0x9999999Au>() * oldest_norm;   // -0.1
  else age_pen = -0.1 * oldest_norm;
  rls = foot + overdue_pen + age_pen;
  rls = rls < -10 ? -10 : (rls > 10 ? 10 : rls);
}
template <bool COLD = false>
__device__ __forceinline__ double agent_reward(const int method, const bool ls_slot, const double rls, const double foot, const double energy_kwh,
                                               const double hour, const double ite_kw, const double total_kw, const double water) {
  double r;
  switch (method) {   // wave-uniform
    case SDC_REWARD_DEFAULT: r = ls_slot ? rls : foot; break;
    case SDC_REWARD_FOOTPRINT: r = foot; break;
    case SDC_REWARD_TOU: r = -1.0 * energy_kwh * tou_price((int)hour % 24); break;
    case SDC_REWARD_ENERGY_EFFICIENCY: r = ite_kw / total_kw; break;
    case SDC_REWARD_PUE: r = -fabs((ite_kw != 0 ? total_kw / ite_kw : (double)INFINITY) - 1); break;
    case SDC_REWARD_WATER:
      if constexpr (COLD) r = cold_f64<0xBF847AE1u, 0x47AE147Bu>() * water;   // -0.01
      else r = -0.01 * water;
      break;
    default: r = 0.0;   // SDC_REWARD_CUSTOM: custom_agent_reward returns 0
  }
  return r;
}
__device__ __forceinline__ Rewards step_rewards(const RewardIn& in, const int (&method)[3], const unsigned hd0) {
  double foot, rls;
  reward_terms<true>(in.z, in.norm_ci_next, in.overdue, in.oldest_norm, foot, rls);
  Rewards o;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double r = agent_reward<true>(method[a], a == 0, rls, foot, in.energy_kwh, in.hour, in.ite_kw, in.total_kw, in.water);
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
    inf_row[SDC_INFO_RESERVED] = (float)path;   // diagnostic: 0 no ring read, 1 a window re-centred ahead of need, 3 rebuilt
    inf_row[SDC_INFO_EP_RETURN_LS] = (float)r.ret[0];
    inf_row[SDC_INFO_EP_RETURN_DC] = (float)r.ret[1];
    inf_row[SDC_INFO_EP_RETURN_BAT] = (float)r.ret[2];
  }
}

}  // namespace sdc_rw
