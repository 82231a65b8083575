// sdc_quadwin.hpp -- the O(1) part of the reward normalisation for FOUR envs per wavefront: each DPP row (16 lanes) carries
// one env, and an env's 64-key rank window lives in FOUR registers of its row:
//     lane l of the row holds the window's keys 4l, 4l + 1, 4l + 2, 4l + 3.
// A one-position shift of the window is an in-lane move of three registers plus ONE cross-lane move (DPP row_shl:1 /
// row_shr:1 -- the row is the env, so the shift never crosses an env boundary), and the in-lane sum ((k0 + k1) + (k2 + k3))
// followed by the row's butterfly visits the keys in the same tree as the half-wave form's (sdc_halfwin.hpp: in-lane pair,
// then five stages), so sums round identically on both mappings.
// The functions carry the same names as the half-wave ones, overloaded on the window type: pair_reward_fast
// (sdc_step.hip) is written once for both.
#pragma once
#include "sdc_halfwin.hpp"

namespace sdc_hw {

constexpr int QROW = 16;      // lanes per env

struct QWin {
  unsigned k0, k1, k2, k3;   // keys 4l .. 4l + 3 of the window (KEY_NONE beyond hi)
  int r0, hi;                // rank of key 0 in the sorted history; valid keys (0: no window)
};

// the 16 ballot bits of this lane's row
__device__ __forceinline__ unsigned row_ballot(const bool p, const int row) { return (unsigned)(__ballot(p) >> (row * QROW)) & 0xFFFFu; }
// number of true flags among the row's 64 keys
__device__ __forceinline__ int row_count(const bool p0, const bool p1, const bool p2, const bool p3, const int row) {
  return __popc(row_ballot(p0, row)) + __popc(row_ballot(p1, row)) + __popc(row_ballot(p2, row)) + __popc(row_ballot(p3, row));
}
// key at window position p (0..63) of this lane's env; every lane active; p (uniform in the row) is clamped
__device__ __forceinline__ unsigned key_at(const QWin& q, int p, const int lane_base) {
  p = p < 0 ? 0 : (p > WIN - 1 ? WIN - 1 : p);
  const int j = p & 3;
  const unsigned src = j == 0 ? q.k0 : (j == 1 ? q.k1 : (j == 2 ? q.k2 : q.k3));   // (the source lane is in this lane's row: same p)
  return (unsigned)__builtin_amdgcn_ds_bpermute((lane_base + (p >> 2)) << 2, (int)src);
}
__device__ __forceinline__ unsigned next_in_row(const unsigned v) { return sdc_rw::dpp_u32<0x101, 0xF>(KEY_NONE, v); }   // row_shl:1: lane l <- l + 1
__device__ __forceinline__ unsigned prev_in_row(const unsigned v) { return sdc_rw::dpp_u32<0x111, 0xF>(0u, v); }         // row_shr:1: lane l <- l - 1

__device__ __forceinline__ bool hw_evict(QWin& q, const unsigned y, const bool on, const int row, const int l) {
  const int m = row_count(q.k0 < y, q.k1 < y, q.k2 < y, q.k3 < y, row);      // valid keys below y (KEY_NONE never counts)
  const unsigned km = key_at(q, m, row * QROW);
  const unsigned n0 = next_in_row(q.k0);
  bool ch = false;
  if (on && m < q.hi) {
    if (km != y) {
      if (m == 0) q.r0 -= 1;     // below the window: every rank inside it moves down
      else q.hi = 0;             // inconsistent: drop the window
    } else {
      // inside (equal keys are interchangeable: take the first): close the gap from above
      const int p = 4 * l;
      const unsigned a0 = p >= m ? q.k1 : q.k0, a1 = p + 1 >= m ? q.k2 : q.k1;
      const unsigned a2 = p + 2 >= m ? q.k3 : q.k2, a3 = p + 3 >= m ? n0 : q.k3;
      q.k0 = a0; q.k1 = a1; q.k2 = a2; q.k3 = a3;
      q.hi -= 1;
      ch = true;
    }
  }
  return ch;
}
__device__ __forceinline__ bool hw_insert(QWin& q, const unsigned x, const int m_hist, const bool on, const int row, const int l) {
  const int p = row_count(q.k0 <= x, q.k1 <= x, q.k2 <= x, q.k3 <= x, row);   // valid keys <= x: x belongs at position p
  const unsigned n0 = next_in_row(q.k0), p3 = prev_in_row(q.k3);
  bool ch = false;
  if (on && q.hi > 0) {
    if (p == 0 && q.r0 != 0) {
      q.r0 += 1;                                       // below the window: every rank inside it moves up
    } else {
      const bool ends = q.r0 + q.hi == m_hist;         // the window lists the history's last key
      if (!(p == q.hi && !ends)) {
        const int s = 4 * l;
        if (q.hi >= WIN && ends) {   // (not `== WIN`: the compiler then rebuilds {r0, hi} with a hoisted constant 64 -- a register pair held across the multi-step kernels' loop)
          // full, and it must go on ending the history: x enters at p - 1, the keys below it move down, the first drops out
          const int t = p - 1;
          const unsigned a0 = s < t ? q.k1 : (s == t ? x : q.k0), a1 = s + 1 < t ? q.k2 : (s + 1 == t ? x : q.k1);
          const unsigned a2 = s + 2 < t ? q.k3 : (s + 2 == t ? x : q.k2), a3 = s + 3 < t ? n0 : (s + 3 == t ? x : q.k3);
          q.k0 = a0; q.k1 = a1; q.k2 = a2; q.k3 = a3;
          q.r0 += 1;
        } else {
          // the keys from p on move up by one; the key at position 63 drops out
          const unsigned a0 = s < p ? q.k0 : (s == p ? x : p3), a1 = s + 1 < p ? q.k1 : (s + 1 == p ? x : q.k0);
          const unsigned a2 = s + 2 < p ? q.k2 : (s + 2 == p ? x : q.k1), a3 = s + 3 < p ? q.k3 : (s + 3 == p ? x : q.k2);
          q.k0 = a0; q.k1 = a1; q.k2 = a2; q.k3 = a3;
          q.hi += q.hi < WIN ? 1 : 0;
        }
        ch = true;
      }
    }
  }
  return ch;
}
__device__ __forceinline__ bool hw_update(QWin& q, const unsigned x_new, const unsigned x_old, const bool has_old, const int n_prev,
                                          const bool on, const int row, const int l) {
  const bool c1 = hw_evict(q, x_old, on && has_old, row, l);
  const bool c2 = hw_insert(q, x_new, has_old ? n_prev - 1 : n_prev, on, row, l);
  return c1 || c2;
}
// keys at ranks k and k + 1 (the second only if it exists); all lanes active
__device__ __forceinline__ bool hw_resolve(const QWin& q, const int k, const int n, const int row, unsigned& a, unsigned& b) {
  const int t = k - q.r0;
  const int tb = (k + 1 > n - 1) ? t : t + 1;
  a = key_at(q, t, row * QROW);
  b = key_at(q, tb, row * QROW);
  return q.hi > 0 && t >= 0 && tb < q.hi;
}

// sums over the 16 lanes of each row (every lane of the row gets it): the first four stages of half_sum_f64 / half_sum_u32
__device__ __forceinline__ double row_sum_f64(double v) {
  v += dpp_f64<SDC_DPP_XOR1>(v);
  v += dpp_f64<SDC_DPP_XOR2>(v);
  v += dpp_f64<SDC_DPP_HALF_MIRROR>(v);
  v += dpp_f64<SDC_DPP_MIRROR>(v);
  return v;
}
__device__ __forceinline__ unsigned row_sum_u32(unsigned v) {
  v += sdc_rw::dpp_u32<0xB1, 0xF>(0u, v);
  v += sdc_rw::dpp_u32<0x4E, 0xF>(0u, v);
  v += sdc_rw::dpp_u32<0x141, 0xF>(0u, v);
  v += sdc_rw::dpp_u32<0x140, 0xF>(0u, v);
  return v;
}

// ---- what pair_reward_fast asks of a window type, for both mappings ---------------------------------------------------------
// first / last valid key (cached in the header)
__device__ __forceinline__ unsigned win_first(const HWin& q, const int h) { return key_at(q.a, q.b, 0, h << 5); }
__device__ __forceinline__ unsigned win_last(const HWin& q, const int h) { return key_at(q.a, q.b, q.hi - 1, h << 5); }
__device__ __forceinline__ unsigned win_first(const QWin& q, const int row) { return key_at(q, 0, row * QROW); }
__device__ __forceinline__ unsigned win_last(const QWin& q, const int row) { return key_at(q, q.hi - 1, row * QROW); }
// this lane's keys in [lo, hi) (the keys a clip bound has crossed): their number, and with `flip` undone their sum and sum
// of squares, each in the in-lane order the trees above continue
__device__ __forceinline__ bool win_any_in(const HWin& q, const unsigned lo, const unsigned hi) {
  return (q.a >= lo && q.a < hi) || (q.b >= lo && q.b < hi);
}
__device__ __forceinline__ bool win_any_in(const QWin& q, const unsigned lo, const unsigned hi) {
  return (q.k0 >= lo && q.k0 < hi) || (q.k1 >= lo && q.k1 < hi) || (q.k2 >= lo && q.k2 < hi) || (q.k3 >= lo && q.k3 < hi);
}
__device__ __forceinline__ void win_crossed(const HWin& q, const unsigned lo, const unsigned hi, const unsigned flip, const bool en,
                                            unsigned& c, double& s1, double& s2) {
  const bool xa = en && q.a >= lo && q.a < hi, xb = en && q.b >= lo && q.b < hi;
  const double va = xa ? key_f64(q.a ^ flip) : 0.0, vb = xb ? key_f64(q.b ^ flip) : 0.0;
  c = (xa ? 1u : 0u) + (xb ? 1u : 0u);
  s1 = va + vb;
  s2 = va * va + vb * vb;
}
__device__ __forceinline__ void win_crossed(const QWin& q, const unsigned lo, const unsigned hi, const unsigned flip, const bool en,
                                            unsigned& c, double& s1, double& s2) {
  const bool x0 = en && q.k0 >= lo && q.k0 < hi, x1 = en && q.k1 >= lo && q.k1 < hi;
  const bool x2 = en && q.k2 >= lo && q.k2 < hi, x3 = en && q.k3 >= lo && q.k3 < hi;
  const double v0 = x0 ? key_f64(q.k0 ^ flip) : 0.0, v1 = x1 ? key_f64(q.k1 ^ flip) : 0.0;
  const double v2 = x2 ? key_f64(q.k2 ^ flip) : 0.0, v3 = x3 ? key_f64(q.k3 ^ flip) : 0.0;
  c = ((x0 ? 1u : 0u) + (x1 ? 1u : 0u)) + ((x2 ? 1u : 0u) + (x3 ? 1u : 0u));
  s1 = (v0 + v1) + (v2 + v3);                             // (the half-wave form: in-lane pair, then the xor-1 stage)
  s2 = (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
}

}  // namespace sdc_hw
