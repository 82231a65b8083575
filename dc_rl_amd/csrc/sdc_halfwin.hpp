// sdc_halfwin.hpp -- the O(1) part of the reward normalisation (sdc_trackers.hpp) for TWO envs at once: each half of the
// wavefront (32 lanes) carries one env, and an env's 64-key rank window lives in TWO registers of its half:
//     lane l of the half holds the window's keys 2l (a) and 2l + 1 (b).
// With that interleaving a one-position shift of the window needs a single cross-lane move (DPP wave_shl:1 / wave_shr:1
// of one register), and the pair-wise lane reductions visit the keys in the same tree as the whole-wavefront code does
// (in-lane pair first = its xor-1 stage), so sums round identically on both paths.
// Everything here is per-lane code whose values are uniform inside a half; "ballots" are split per half.
//
// The whole-wavefront forms (QTrack, qt_refill, rebuild_state) remain the slow path: an env whose step needs the ring
// (a window to re-centre, a rebuild) or meets anything unusual is redone by env_reward() from its untouched state.
#pragma once
#include "sdc_trackers.hpp"

namespace sdc_hw {
using sdc_rw::KEY_NONE;
using sdc_rw::key_f64;
constexpr int WIN = SDC_WIN;

struct HWin {
  unsigned a, b;   // keys 2l, 2l + 1 of the window (KEY_NONE beyond hi)
  int r0, hi;      // rank of key 0 in the sorted history; valid keys (0: no window)
};

// number of true flags among the half's 64 keys
__device__ __forceinline__ int half_count(const bool pa, const bool pb, const int h) {
  const unsigned long long ma = __ballot(pa), mb = __ballot(pb);
  const unsigned xa = h ? (unsigned)(ma >> 32) : (unsigned)ma, xb = h ? (unsigned)(mb >> 32) : (unsigned)mb;
  return __popc(xa) + __popc(xb);
}
// key at window position p (0..63) of this lane's env.  Must be called with every lane active (ds_bpermute returns 0
// for a disabled source lane); p is clamped.
// (keys by value: a select between two struct members turns into an indexed load from a stack copy of the struct)
__device__ __forceinline__ unsigned key_at(const unsigned wa, const unsigned wb, int p, const int lane_base) {
  p = p < 0 ? 0 : (p > WIN - 1 ? WIN - 1 : p);
  const unsigned src = (p & 1) ? wb : wa;   // (the source lane is in this lane's half: same p)
  return (unsigned)__builtin_amdgcn_ds_bpermute((lane_base + (p >> 1)) << 2, (int)src);
}
__device__ __forceinline__ unsigned next_lane(const unsigned v, const int l) {   // lane l <- lane l + 1 of the half
  const unsigned t = sdc_rw::dpp_u32<0x130, 0xF>(KEY_NONE, v);                   // wave_shl:1
  return l == 31 ? KEY_NONE : t;
}
__device__ __forceinline__ unsigned prev_lane(const unsigned v, const int l) {   // lane l <- lane l - 1 of the half
  const unsigned t = sdc_rw::dpp_u32<0x138, 0xF>(0u, v);                         // wave_shr:1
  return l == 0 ? 0u : t;
}

// qt_evict: remove one occurrence of y (lanes with `on`); returns whether the lane's keys changed.  Call with all lanes
// active.
__device__ __forceinline__ bool hw_evict(HWin& q, const unsigned y, const bool on, const int h, const int l) {
  const int m = half_count(q.a < y, q.b < y, h);      // valid keys below y (KEY_NONE never counts)
  const unsigned km = key_at(q.a, q.b, m, h << 5);
  const unsigned na = next_lane(q.a, l);
  bool ch = false;
  if (on && m < q.hi) {
    if (km != y) {
      if (m == 0) q.r0 -= 1;     // below the window: every rank inside it moves down
      else q.hi = 0;             // inconsistent: drop the window
    } else {
      // inside (equal keys are interchangeable: take the first): close the gap from above
      const unsigned a2 = 2 * l >= m ? q.b : q.a;
      const unsigned b2 = 2 * l + 1 >= m ? na : q.b;
      q.a = a2;
      q.b = b2;
      q.hi -= 1;
      ch = true;
    }
  }
  return ch;
}
// qt_insert: add x to a history of m_hist keys (lanes with `on`)
__device__ __forceinline__ bool hw_insert(HWin& q, const unsigned x, const int m_hist, const bool on, const int h, const int l) {
  const int p = half_count(q.a <= x, q.b <= x, h);    // valid keys <= x: x belongs at position p
  const unsigned na = next_lane(q.a, l), pb = prev_lane(q.b, l);
  bool ch = false;
  if (on && q.hi > 0) {
    if (p == 0 && q.r0 != 0) {
      q.r0 += 1;                                       // below the window: every rank inside it moves up
    } else {
      const bool ends = q.r0 + q.hi == m_hist;         // the window lists the history's last key
      if (!(p == q.hi && !ends)) {
        if (q.hi == WIN && ends) {
          // full, and it must go on ending the history: x enters at p - 1, the keys below it move down, the first drops out
          const unsigned a2 = 2 * l < p - 1 ? q.b : (2 * l == p - 1 ? x : q.a);
          const unsigned b2 = 2 * l + 1 < p - 1 ? na : (2 * l + 1 == p - 1 ? x : q.b);
          q.a = a2;
          q.b = b2;
          q.r0 += 1;
        } else {
          // the keys from p on move up by one; the key at position 63 drops out
          const unsigned a2 = 2 * l < p ? q.a : (2 * l == p ? x : pb);
          const unsigned b2 = 2 * l + 1 < p ? q.b : (2 * l + 1 == p ? x : q.a);
          q.a = a2;
          q.b = b2;
          q.hi = min(WIN, q.hi + 1);
        }
        ch = true;
      }
    }
  }
  return ch;
}
__device__ __forceinline__ bool hw_update(HWin& q, const unsigned x_new, const unsigned x_old, const bool has_old, const int n_prev,
                                          const bool on, const int h, const int l) {
  const bool c1 = hw_evict(q, x_old, on && has_old, h, l);
  const bool c2 = hw_insert(q, x_new, has_old ? n_prev - 1 : n_prev, on, h, l);
  return c1 || c2;
}
// keys at ranks k and k + 1 (the second only if it exists); all lanes active
__device__ __forceinline__ bool hw_resolve(const HWin& q, const int k, const int n, const int h, unsigned& a, unsigned& b) {
  const int t = k - q.r0;
  const int tb = (k + 1 > n - 1) ? t : t + 1;
  a = key_at(q.a, q.b, t, h << 5);
  b = key_at(q.a, q.b, tb, h << 5);
  return q.hi > 0 && t >= 0 && tb < q.hi;
}
// does the window list EVERY history key in [lo, hi)?  (qt_spans)  all lanes active
__device__ __forceinline__ bool hw_spans(const HWin& q, const unsigned lo, const unsigned hi, const int n, const int h) {
  const unsigned first = key_at(q.a, q.b, 0, h << 5), last = key_at(q.a, q.b, q.hi - 1, h << 5);
  const bool lo_ok = q.r0 == 0 || first < lo;
  const bool hi_ok = q.r0 + q.hi >= n || hi <= last;
  return q.hi > 0 && lo_ok && hi_ok;
}

// half-wave integer sum (every lane of the half gets it)
__device__ __forceinline__ unsigned half_sum_u32(unsigned v) {
  v += sdc_rw::dpp_u32<0xB1, 0xF>(0u, v);
  v += sdc_rw::dpp_u32<0x4E, 0xF>(0u, v);
  v += sdc_rw::dpp_u32<0x141, 0xF>(0u, v);
  v += sdc_rw::dpp_u32<0x140, 0xF>(0u, v);
  const auto s = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return s[0] + s[1];
}

}  // namespace sdc_hw
