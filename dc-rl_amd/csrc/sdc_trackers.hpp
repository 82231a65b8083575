// sdc_trackers.hpp -- O(1) maintenance of reward normalisation (utils/reward_creator.py:16-45) shared by the
// dynamics kernel (fast path: no history read) and the reward kernel (ring path: slides / re-anchors).
//
// normalize_energy needs, over a 10 000-entry sliding history, the 25th / 75th percentiles (np.percentile, linear:
// order statistics k and k+1 each) and the mean / population std of the history clipped to
// [q1 - 1.5 iqr, q3 + 1.5 iqr].  A step inserts one value and evicts at most one, so both are maintained
// incrementally by four TRACKERS kept in the env's 512-byte header:
//   * QTrack (quartiles): anchor key G present or not in the ring, exact counts #{x < G}, #{x <= G}, the (up to) 4
//     largest keys below and 4 smallest keys above G -- a window of ~9 consecutive order statistics;
//   * TTrack (clip bounds): a QTrack around an arbitrary anchor plus running fp64 sums of v, v^2 over {x <= G}.
// Everything here is wave-uniform scalar work.  When a wanted rank or clip bound has moved past the listed keys
// the env is queued for sdc_reward_kernel, which re-reads its ring.
#pragma once
#include "sdc_device.hpp"

namespace sdc_rw {

constexpr unsigned KEY_NONE = 0xFFFFFFFFu;  // empty ring slot; also "+infinity" in ascending neighbour lists
constexpr int QW = SDC_QW;
constexpr int SMALL_N = 32;                 // below this the reward kernel computes directly from the ring

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

// Apply this step's eviction (x_old, if has_old) and insertion (x_new) to a tracker that described the ring of
// the previous step, which held n_prev keys.  Sets q.g = 0 if the tracker turns out to be inconsistent.
__device__ __forceinline__ void qt_update(QTrack& q, unsigned x_new, unsigned x_old, bool has_old, int n_prev) {
  int m = n_prev;
  if (has_old) {
    m -= 1;
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
// ------------------------------------------------------------------------------------------------
// TAIL trackers.  normalize_energy clips the history to [lb, ub] = [q1 - 1.5 iqr, q3 + 1.5 iqr] and takes mean / std
// of the clipped values.  With F(k) = (count, sum v, sum v^2) over the keys < k, the clipped sums are
//   sum clip(v)   = F(kub).s1 - F(klb).s1 + n_lo lb   + n_hi ub,
//   sum clip(v)^2 = F(kub).s2 - F(klb).s2 + n_lo lb^2 + n_hi ub^2,     n_lo = F(klb).c,  n_hi = n - F(kub).c,
// where klb = smallest key whose value is >= lb and kub = smallest key whose value is > ub.  A tail tracker is a
// QTrack around an ARBITRARY anchor key g plus running fp64 sums over {x <= g}: the step's insertion / eviction
// update it in O(1), and F(k) for a bound k near g is read off the window (the listed keys between g and k are
// added or removed).  Only when the bound has moved past the listed keys is the ring needed: the tracker is then
// re-anchored exactly at the bound with one sweep + one summation pass over the VGPR-resident ring.
struct TTrack {
  QTrack q;
  double s1, s2;  // sum of v, v^2 over the keys <= q.g
};

__device__ __forceinline__ TTrack tt_load(unsigned hd, int base) {
  TTrack t;
  t.q = qt_load(hd, base);
  t.s1 = rec_f64(hd, base + T_SUM1);
  t.s2 = rec_f64(hd, base + T_SUM2);
  return t;
}
__device__ __forceinline__ void tt_update(TTrack& t, unsigned x_new, unsigned x_old, bool has_old, int n_prev) {
  if (has_old && x_old <= t.q.g) {
    const double v = key_f64(x_old);
    t.s1 -= v;
    t.s2 -= v * v;
  }
  if (x_new <= t.q.g) {
    const double v = key_f64(x_new);
    t.s1 += v;
    t.s2 += v * v;
  }
  qt_update(t.q, x_new, x_old, has_old, n_prev);
}
// (count, sum, sum of squares) over the keys < kb, if the window covers the span between the anchor and kb
__device__ __forceinline__ bool tt_below(const TTrack& t, const unsigned kb, const int n, int& c, double& s1, double& s2) {
  const QTrack& q = t.q;
  if (q.g == 0u || q.g == KEY_NONE) return false;
  bool covered;
  if (kb > q.g) {  // add the listed keys in (g, kb)
    c = q.c_le;
    s1 = t.s1;
    s2 = t.s2;
    covered = q.ns == n - q.c_le;  // every key above g is listed
    auto add = [&](int i, unsigned e) {
      if (i < q.ns) {
        if (e < kb) {
          const double v = key_f64(e);
          c += 1;
          s1 += v;
          s2 += v * v;
        } else {
          covered = true;
        }
      }
    };
    add(0, q.S.e0);
    add(1, q.S.e1);
    add(2, q.S.e2);
    add(3, q.S.e3);
  } else {         // remove the copies of g and the listed keys in [kb, g)
    const double vg = key_f64(q.g), ceq = (double)(q.c_le - q.c_lt);
    c = q.c_lt;
    s1 = t.s1 - ceq * vg;
    s2 = t.s2 - ceq * (vg * vg);
    covered = q.np == q.c_lt;      // every key below g is listed
    auto sub = [&](int i, unsigned e) {
      if (i < q.np) {
        if (e >= kb) {
          const double v = key_f64(e);
          c -= 1;
          s1 -= v;
          s2 -= v * v;
        } else {
          covered = true;
        }
      }
    };
    sub(0, q.P.e0);
    sub(1, q.P.e1);
    sub(2, q.P.e2);
    sub(3, q.P.e3);
  }
  return covered;
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
__device__ __forceinline__ void tt_put(unsigned& o, int base, const TTrack& t) {
  qt_put(o, base, t.q);
  put_f64(o, base + T_SUM1, t.s1);
  put_f64(o, base + T_SUM2, t.s2);
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
// clipped mean / std from F(klb) = (cl, l1, l2) and F(kub) = (ch, h1, h2)
__device__ __forceinline__ void clipped_moments(const int n, const Bounds& b, int cl, double l1, double l2, int ch, double h1,
                                                double h2, double& mean, double& sd) {
  const double n_lo = (double)cl, n_hi = (double)(n - ch);
  const double C1 = (h1 - l1) + n_lo * b.lb + n_hi * b.ub;
  const double C2 = (h2 - l2) + n_lo * (b.lb * b.lb) + n_hi * (b.ub * b.ub);
  mean = C1 / (double)n;
  const double var = C2 / (double)n - mean * mean;
  sd = (var > 0 && b.ub > b.lb) ? sqrt(var) : 0.0;
}

// the four trackers of one env
struct Trackers {
  QTrack q1, q3;
  TTrack tl, th;
};
__device__ __forceinline__ Trackers trackers_load(unsigned hd0, unsigned hd1) {
  Trackers T;
  T.q1 = qt_load(hd0, H_Q1);
  T.q3 = qt_load(hd0, H_Q3);
  T.tl = tt_load(hd1, H_LO - 64);
  T.th = tt_load(hd1, H_HI - 64);
  return T;
}
__device__ __forceinline__ void trackers_put(unsigned& o0, unsigned& o1, const Trackers& T) {
  qt_put(o0, H_Q1, T.q1);
  qt_put(o0, H_Q3, T.q3);
  tt_put(o1, H_LO - 64, T.tl);
  tt_put(o1, H_HI - 64, T.th);
}
__device__ __forceinline__ bool qt_valid(const QTrack& q) { return q.g != 0u && q.g != KEY_NONE; }

// FAST PATH (dynamics kernel): apply this step's insertion / eviction to the trackers and, if every window still
// covers what is asked of it, produce the clipped mean / std without touching the ring.  n already includes x_new.
// Returns false when the ring is needed (the trackers are then left post-update for sdc_reward_kernel).
__device__ __forceinline__ bool reward_fast(const int n, const unsigned x_new, const unsigned x_old, Trackers& T, double& mean,
                                            double& sd) {
  if (n < SMALL_N) {
    T.q1.g = T.q3.g = T.tl.q.g = T.th.q.g = 0u;
    mean = 0.0;
    sd = 0.0;
    return n < 2;   // a single value: z = 0 (no ring needed)
  }
  const bool has_old = x_old != KEY_NONE;
  const int n_prev = has_old ? n : n - 1;
  bool ok = true;
  if (qt_valid(T.q1)) qt_update(T.q1, x_new, x_old, has_old, n_prev); else ok = false;
  if (qt_valid(T.q3)) qt_update(T.q3, x_new, x_old, has_old, n_prev); else ok = false;
  if (qt_valid(T.tl.q)) tt_update(T.tl, x_new, x_old, has_old, n_prev); else ok = false;
  if (qt_valid(T.th.q)) tt_update(T.th, x_new, x_old, has_old, n_prev); else ok = false;
  if (!ok) return false;
  int k1, k3;
  quartile_ranks(n, k1, k3);
  unsigned a1, b1, a3, b3;
  if (!qt_resolve(T.q1, k1, n, a1, b1) || !qt_resolve(T.q3, k3, n, a3, b3)) return false;
  const Bounds b = clip_bounds(n, a1, b1, a3, b3);
  int cl, ch;
  double l1, l2, h1, h2;
  if (!tt_below(T.tl, b.klb, n, cl, l1, l2) || !tt_below(T.th, b.kub, n, ch, h1, h2)) return false;
  clipped_moments(n, b, cl, l1, l2, ch, h1, h2, mean, sd);
  return true;
}

// rewards (utils/reward_creator.py:48-130) from the z-score, running episode returns
struct Rewards {
  double ls, foot, ret0, ret1, ret2;
};
__device__ __forceinline__ Rewards step_rewards(const double z, const double norm_ci_next, const double oldest_norm,
                                                const double overdue, const unsigned hd0) {
  Rewards r;
  r.foot = -1.0 * (norm_ci_next * z / 0.50);
  const double overdue_pen = -0.3 * sqrt(overdue) + 0.3;
  const double age_pen = -0.1 * oldest_norm;
  double rls = r.foot + overdue_pen + age_pen;
  r.ls = rls < -10 ? -10 : (rls > 10 ? 10 : rls);
  r.ret0 = rec_f64(hd0, H_RET) + r.ls;
  r.ret1 = rec_f64(hd0, H_RET + 2) + r.foot;
  r.ret2 = rec_f64(hd0, H_RET + 4) + r.foot;
  return r;
}
// lane 0 writes the step's rewards and the reward-side info columns
__device__ __forceinline__ void store_rewards(const Rewards& r, const double z, const int path, const int env,
                                              float* __restrict__ rew, float* __restrict__ inf_row) {
  rew[env * 3 + 0] = (float)r.ls;
  rew[env * 3 + 1] = (float)r.foot;
  rew[env * 3 + 2] = (float)r.foot;
  if (inf_row) {
    inf_row[SDC_INFO_ENERGY_Z] = (float)z;
    inf_row[SDC_INFO_RESERVED] = (float)path;   // diagnostic: 0 no ring read, 1 ring read, 2 bisection + rebuild
    inf_row[SDC_INFO_EP_RETURN_LS] = (float)r.ret0;
    inf_row[SDC_INFO_EP_RETURN_DC] = (float)r.ret1;
    inf_row[SDC_INFO_EP_RETURN_BAT] = (float)r.ret2;
  }
}

}  // namespace sdc_rw
