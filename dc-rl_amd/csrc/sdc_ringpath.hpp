// sdc_ringpath.hpp -- the rare moments when an env's history ring must be read: in-wave primitives.
//
// sdc_trackers.hpp answers the reward normalisation from two quartile trackers, two tail sets and running sums.
// A tracker's window of listed keys runs out every few steps (the wanted rank random-walks through it); the env's
// own wavefront then SLIDES the tracker: it moves the anchor to the last listed key on the side that ran out and sweeps the ring once
// for what lies beyond it -- 10 240 keys, 160 per lane, read from L2 / HBM as coalesced dwordx4 loads:
//   up:   d = x - (pivot+1): borrows <=> x <= pivot; a legitimate d is the distance of a key above the pivot
//   down: d = (pivot-1) - x: borrows <=> x >= pivot -- the same formula on complemented keys (~a = -a-1)
// One v_sub_co_u32 / v_addc_co_u32 pair per key gives the distance and counts the predicate; the 4 smallest
// distances (v_med3_u32 insertion network per lane, DPP merge across the wave) are the new neighbours; everything
// on the near side of the new anchor is already known from the old window.
//
// The dynamics kernel slides AHEAD of need, at its start (slide_trackers is called when a window would run out on
// this step in the worst case), so the sweep overlaps with the other resident wavefronts instead of extending the
// kernel's tail.  Far more rarely a tail set's threshold must move down (one sweep that re-collects both sets).  A
// full REBUILD (bisections on the key space + a two-sided sweep + collection + fp64 sums) bootstraps everything on
// the first steps and after state injection.
//
// Why in-wave and not a separate reward kernel (round-1 measurements, MI355X, 4096 envs): a kernel that streams
// every env's ring each step is HBM-bound at >= 23 us (32 us in practice); a kernel that only serves the ~10 % of
// envs that need their ring still took 16-22 us, because its few workgroups per CU are latency-bound single waves
// (~6 ns per instruction with nothing to overlap, cold instruction cache).  Inside the dynamics kernel the same
// work hides behind 15 other resident wavefronts per CU.
#pragma once
#include "sdc_trackers.hpp"

namespace sdc_rw {

constexpr int RING_VECS = SDC_HIST_STRIDE / 4 / SDC_WAVE;   // dwordx4 loads per lane for one pass over the ring (40)

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
__device__ __forceinline__ unsigned from63(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#define STAGE(C, M) v += dpp_u32<C, M>(0u, v);
  SDC_DPP_STAGES(STAGE)
#undef STAGE
  return from63(v);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define STAGE(C, M) v = min(v, dpp_u32<C, M>(KEY_NONE, v));
  SDC_DPP_STAGES(STAGE)
#undef STAGE
  return from63(v);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define STAGE(C, M) v = max(v, dpp_u32<C, M>(0u, v));
  SDC_DPP_STAGES(STAGE)
#undef STAGE
  return from63(v);
}
// the 4 smallest of the wave's 64 ascending 4-lists (wave-uniform result)
__device__ __forceinline__ void wave_merge_l4(L4& A) {
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
  A.e0 = from63(A.e0);
  A.e1 = from63(A.e1);
  A.e2 = from63(A.e2);
  A.e3 = from63(A.e3);
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

// ------------------------------------------------------------------------------------------------
// One env's ring as this wavefront reads it: lane l fetches dwordx4 number q * 64 + l for q = 0 .. 39.  `patch_slot`
// (-1: none) names a slot whose content is `patch_x` regardless of memory: the key this very wavefront has just
// stored there (the store may not be visible to its own later loads through the vector L1).
struct RingView {
  const uint4* hp;
  int patch_slot;
  unsigned patch_x;
};
__device__ __forceinline__ uint4 ring_fetch(const RingView& R, const int q, const int lane) {
  uint4 v = R.hp[q * SDC_WAVE + lane];
  if ((R.patch_slot >> 2) == q * SDC_WAVE + lane) {
    const int c = R.patch_slot & 3;
    if (c == 0) v.x = R.patch_x;
    if (c == 1) v.y = R.patch_x;
    if (c == 2) v.z = R.patch_x;
    if (c == 3) v.w = R.patch_x;
  }
  return v;
}

// ------------------------------------------------------------------------------------------------
// one-sided sweep beyond a pivot
struct SlideOut {
  unsigned count;  // up: #{x <= pivot}; down: #{x >= pivot} (empty slots included)
  L4 dist;         // the 4 smallest legitimate distances beyond the pivot (then KEY_NONE-ish garbage)
};
// (before this step's append: the ring in memory is the truth, no patch)
template <bool UP>
__device__ __forceinline__ void slide_sweep(const uint4* __restrict__ hp, const int lane, const unsigned pivot, SlideOut& o) {
  const unsigned pp = (UP ? pivot : ~pivot) + 1u;
  unsigned c = 0u;
  L4 d = {KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE};
  // batches of dwordx4 loads per lane, each batch fully in flight before its first use
  constexpr int BATCH = RING_VECS / 8;   // (20 in flight was measured no faster: the sweep is VALU-bound; 5 keeps the code small)
#pragma unroll 1
  for (int hf = 0; hf < RING_VECS / BATCH; hf++) {
    uint4 v[BATCH];
#pragma unroll
    for (int q = 0; q < BATCH; q++) v[q] = hp[(hf * BATCH + q) * SDC_WAVE + lane];
#pragma unroll
    for (int q = 0; q < BATCH; q++) {
      const unsigned xs[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
      for (int c4 = 0; c4 < 4; c4++) {
        const unsigned x = UP ? xs[c4] : ~xs[c4];
        unsigned e;
        SDC_SUB_COUNT(e, c, x, pp);
        l4_sweep_insert(d, e);
      }
    }
  }
  o.count = wave_sum_u32(c);
  wave_merge_l4(d);
  o.dist = d;
}

// first / last index of `x` in a sorted 4-list of which the first `cnt` entries are valid (x is present)
__device__ __forceinline__ int first_index(const L4& L, unsigned x) {
  return L.e0 == x ? 0 : (L.e1 == x ? 1 : (L.e2 == x ? 2 : 3));
}
__device__ __forceinline__ void l4_push(L4& L, int& cnt, unsigned v) {
  if (cnt == 0) L.e0 = v;
  if (cnt == 1) L.e1 = v;
  if (cnt == 2) L.e2 = v;
  if (cnt == 3) L.e3 = v;
  cnt += 1;
}

// Move the anchor up to the largest listed key above it (q.ns >= 1), given the sweep beyond that key.
__device__ __forceinline__ void qt_slide_up(QTrack& q, const SlideOut& o) {
  const unsigned g2 = lget(q.S, q.ns - 1);
  const int e0 = first_index(q.S, g2);          // keys S[0..e0) lie strictly between the old and the new anchor
  const int c_eq = q.c_le - q.c_lt;
  // new lower list (descending): S[e0-1] .. S[0], then the old anchor c_eq times, then the old lower list
  L4 P2 = {0u, 0u, 0u, 0u};
  int cnt = 0;
  if (e0 >= 3) l4_push(P2, cnt, q.S.e2);
  if (e0 >= 2) l4_push(P2, cnt, q.S.e1);
  if (e0 >= 1) l4_push(P2, cnt, q.S.e0);
#pragma unroll
  for (int r = 0; r < QW; r++)
    if (r < c_eq) l4_push(P2, cnt, q.g);
  if (q.np > 0) l4_push(P2, cnt, q.P.e0);
  if (q.np > 1) l4_push(P2, cnt, q.P.e1);
  if (q.np > 2) l4_push(P2, cnt, q.P.e2);
  if (q.np > 3) l4_push(P2, cnt, q.P.e3);
  const unsigned smax = KEY_NONE - g2 - 1u;
  const L4& dist = o.dist;
  q.c_lt = q.c_le + e0;
  q.c_le = (int)o.count;
  q.g = g2;
  q.P = P2;
  q.np = min(QW, cnt);
  q.ns = (dist.e0 < smax) + (dist.e1 < smax) + (dist.e2 < smax) + (dist.e3 < smax);
  q.S.e0 = dist.e0 < smax ? g2 + 1u + dist.e0 : KEY_NONE;
  q.S.e1 = dist.e1 < smax ? g2 + 1u + dist.e1 : KEY_NONE;
  q.S.e2 = dist.e2 < smax ? g2 + 1u + dist.e2 : KEY_NONE;
  q.S.e3 = dist.e3 < smax ? g2 + 1u + dist.e3 : KEY_NONE;
}
// Move the anchor down to the smallest listed key below it (q.np >= 1), given the sweep beyond that key.
__device__ __forceinline__ void qt_slide_down(QTrack& q, const int n, const SlideOut& o) {
  const unsigned g2 = lget(q.P, q.np - 1);
  const int e0 = first_index(q.P, g2);          // keys P[0..e0) lie strictly between the new and the old anchor
  const int c_eq = q.c_le - q.c_lt;
  // new upper list (ascending): P[e0-1] .. P[0], then the old anchor c_eq times, then the old upper list
  L4 S2 = {KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE};
  int cnt = 0;
  if (e0 >= 3) l4_push(S2, cnt, q.P.e2);
  if (e0 >= 2) l4_push(S2, cnt, q.P.e1);
  if (e0 >= 1) l4_push(S2, cnt, q.P.e0);
#pragma unroll
  for (int r = 0; r < QW; r++)
    if (r < c_eq) l4_push(S2, cnt, q.g);
  if (q.ns > 0) l4_push(S2, cnt, q.S.e0);
  if (q.ns > 1) l4_push(S2, cnt, q.S.e1);
  if (q.ns > 2) l4_push(S2, cnt, q.S.e2);
  if (q.ns > 3) l4_push(S2, cnt, q.S.e3);
  const int n_empty = SDC_HIST_STRIDE - n;      // empty slots (KEY_NONE) satisfy x >= pivot
  const L4& dist = o.dist;
  q.c_le = q.c_lt - e0;
  q.c_lt = n - ((int)o.count - n_empty);
  q.g = g2;
  q.S = S2;
  q.ns = min(QW, cnt);
  q.np = (dist.e0 < g2) + (dist.e1 < g2) + (dist.e2 < g2) + (dist.e3 < g2);
  q.P.e0 = dist.e0 < g2 ? g2 - 1u - dist.e0 : 0u;
  q.P.e1 = dist.e1 < g2 ? g2 - 1u - dist.e1 : 0u;
  q.P.e2 = dist.e2 < g2 ? g2 - 1u - dist.e2 : 0u;
  q.P.e3 = dist.e3 < g2 ? g2 - 1u - dist.e3 : 0u;
}

// tracker access with a run-time header offset (v_readlane / v_writelane take the lane from an SGPR / M0)
__device__ __forceinline__ void put_dyn(unsigned& o, int idx, unsigned v) {
  const unsigned sv = sfl(v);
  asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(o) : "s"(sv), "s"(idx) : "m0");
}
__device__ __forceinline__ void qt_put_dyn(unsigned& o, const int base, const QTrack& q) {
  put_dyn(o, base + T_G, q.g);
  put_dyn(o, base + T_CLT, (unsigned)q.c_lt);
  put_dyn(o, base + T_CLE, (unsigned)q.c_le);
  put_dyn(o, base + T_NP, (unsigned)q.np);
  put_dyn(o, base + T_NS, (unsigned)q.ns);
  put_dyn(o, base + T_P + 0, q.P.e0);
  put_dyn(o, base + T_P + 1, q.P.e1);
  put_dyn(o, base + T_P + 2, q.P.e2);
  put_dyn(o, base + T_P + 3, q.P.e3);
  put_dyn(o, base + T_S + 0, q.S.e0);
  put_dyn(o, base + T_S + 1, q.S.e1);
  put_dyn(o, base + T_S + 2, q.S.e2);
  put_dyn(o, base + T_S + 3, q.S.e3);
}

// slide requests: 2 bits per quartile tracker (Q1, Q3): 0 none, 1 up, 2 down
enum { SLIDE_NONE = 0, SLIDE_UP = 1, SLIDE_DOWN = 2 };
__device__ __forceinline__ int slide_req(int d1, int d3) { return d1 | (d3 << 2); }

// SLIDE the requested quartile trackers of one env (header dwords in hd, one per lane) over its ring, which holds n
// keys.  One copy of the sweep / surgery code: the trackers take turns through it.
__device__ __forceinline__ unsigned slide_trackers(unsigned hd, const RingView& R, const int lane, const int n, const int req) {
#pragma unroll 1
  for (int t = 0; t < 2; t++) {
    const int d = (req >> (2 * t)) & 3;
    if (d == SLIDE_NONE) continue;
    const int base = t == 0 ? H_Q1 : H_Q3;
    QTrack A = qt_load(hd, base);
    const unsigned pivot = d == SLIDE_UP ? lget(A.S, A.ns - 1) : lget(A.P, A.np - 1);
    SlideOut o;
    if (d == SLIDE_UP) {
      slide_sweep<true>(R.hp, lane, pivot, o);
      qt_slide_up(A, o);
    } else {
      slide_sweep<false>(R.hp, lane, pivot, o);
      qt_slide_down(A, n, o);
    }
    qt_put_dyn(hd, base, A);
  }
  return hd;
}

// ------------------------------------------------------------------------------------------------
// AHEAD-OF-NEED tests on the pre-step trackers (ring of n_prev keys; x_old is the key this step will evict, if any;
// the step's new key is not known yet).

// quartile tracker: would ranks k_next, k_next+1 still be inside the window after this step in the worst case?
// (an insertion below the window shifts every covered rank up by one; one above leaves them in place)
__device__ __forceinline__ int quartile_slide_ahead(const QTrack& q0, const unsigned x_old, const bool has_old,
                                                    const int k_next, const int n_next) {
  if (!qt_valid(q0)) return SLIDE_NONE;   // nothing to slide: the end-of-step rebuild will create it
  // common case first: at least two spare ranks on both sides survive any eviction + insertion
  if (k_next - (q0.c_lt - q0.np) >= 3 && (q0.c_le + q0.ns - 1) - (k_next + 1) >= 3) return SLIDE_NONE;
  QTrack q = q0;
  int m = n_next - 1;                     // keys after the eviction, before the insertion
  if (has_old) qt_evict(q, x_old);
  if (!qt_valid(q)) return SLIDE_NONE;
  const int lo = q.c_lt - q.np, hi = q.c_le + q.ns - 1;   // covered ranks
  const bool complete_lo = q.np == q.c_lt, complete_hi = q.ns == m - q.c_le;
  const int hi_rank = (k_next + 1 > n_next - 1) ? k_next : k_next + 1;
  if (!complete_hi && hi_rank > hi) return q0.ns >= 1 ? SLIDE_UP : SLIDE_NONE;
  if (!complete_lo && k_next < lo + 1) return q0.np >= 1 ? SLIDE_DOWN : SLIDE_NONE;
  return SLIDE_NONE;
}
// ------------------------------------------------------------------------------------------------
// TAIL SETS: collection from the ring.  Every valid key whose flipped image exceeds the side's threshold is appended
// (LDS atomic counter) to that side's 512-slot array in LDS; the caller then takes the arrays into registers.
struct TailLds {
  unsigned keys[2][SDC_TAIL_CAP];
  unsigned cnt[2];
};
__device__ __forceinline__ void tails_collect(const RingView& R, const int lane, const unsigned tau_hi, const unsigned tau_lo,
                                              TailLds& L, double* sums /* nullptr, or out: sum v, sum v^2 over the ring */) {
#pragma unroll 1
  for (int i = lane; i < 2 * SDC_TAIL_CAP; i += SDC_WAVE) (&L.keys[0][0])[i] = TAIL_EMPTY;
  if (lane < 2) L.cnt[lane] = 0u;
  double a1 = 0.0, a2 = 0.0;
#pragma unroll 1
  for (int q = 0; q < RING_VECS; q++) {
    const uint4 v4 = ring_fetch(R, q, lane);
    const unsigned xs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const unsigned x = xs[c4];
      if (x != KEY_NONE) {
        if (x > tau_hi) {
          const unsigned pos = atomicAdd(&L.cnt[0], 1u);
          if (pos < SDC_TAIL_CAP) L.keys[0][pos] = x;
        }
        if (~x > tau_lo) {
          const unsigned pos = atomicAdd(&L.cnt[1], 1u);
          if (pos < SDC_TAIL_CAP) L.keys[1][pos] = ~x;
        }
        if (sums) {
          const double v = key_f64(x);
          a1 += v;
          a2 += v * v;
        }
      }
    }
  }
  if (sums) {
    sums[0] = wave_sum_f64(a1);
    sums[1] = wave_sum_f64(a2);
  }
}
// LDS slot i * 64 + lane <-> register slot i of lane (bank-conflict free; which key sits in which slot is immaterial)
__device__ __forceinline__ TailSet tail_from_lds(const TailLds& L, const int side, const int lane) {
  TailSet s;
#pragma unroll
  for (int i = 0; i < 8; i++) s.k[i] = L.keys[side][i * SDC_WAVE + lane];
  return s;
}
__device__ __forceinline__ void tail_to_lds(TailLds& L, const int side, const int lane, const TailSet& s) {
#pragma unroll
  for (int i = 0; i < 8; i++) L.keys[side][i * SDC_WAVE + lane] = s.k[i];
}

// ------------------------------------------------------------------------------------------------
// END-OF-STEP evaluation of the (post-update) quartile trackers.  quartile_slide_now: which way must tracker q slide
// so that ranks k, k+1 come inside its window?  (3 = it cannot: rebuild)
__device__ __forceinline__ int quartile_slide_now(const QTrack& q, const int k, const int n) {
  unsigned a, b;
  if (!qt_valid(q)) return 3;
  if (qt_resolve(q, k, n, a, b)) return SLIDE_NONE;
  const int hi_rank = (k + 1 > n - 1) ? k : k + 1;
  if (hi_rank >= q.c_le + q.ns) return q.ns >= 1 ? SLIDE_UP : 3;
  return q.np >= 1 ? SLIDE_DOWN : 3;
}

// ------------------------------------------------------------------------------------------------
// REBUILD (bootstrap, injected state, a tracker that lost its window): everything from the ring, one wavefront.

// exact order statistics at ranks k1, k1+1, k3, k3+1 by bisection on the key space: {a1, b1, a3, b3}
__device__ __forceinline__ uint4 wave_bisection(const RingView& R, const int lane, const int k1, const int k3) {
  unsigned kmin = KEY_NONE, kmax = 0u;
#pragma unroll 1
  for (int q = 0; q < RING_VECS; q++) {
    const uint4 v = ring_fetch(R, q, lane);
    const unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      kmin = min(kmin, x[c]);
      kmax = max(kmax, x[c] == KEY_NONE ? 0u : x[c]);
    }
  }
  kmin = wave_min_u32(kmin);
  kmax = wave_max_u32(kmax);
  unsigned lo1 = kmin, hi1 = kmax, lo3 = kmin, hi3 = kmax;
  while (lo1 < hi1 || lo3 < hi3) {
    const unsigned m1 = lo1 + ((hi1 - lo1) >> 1);
    const unsigned m3 = lo3 + ((hi3 - lo3) >> 1);
    unsigned cnt = 0;  // packed: count(key <= m1) << 16 | count(key <= m3); each <= 10240
#pragma unroll 1
    for (int q = 0; q < RING_VECS; q++) {
      const uint4 v = ring_fetch(R, q, lane);
      const unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int c = 0; c < 4; c++) cnt += ((x[c] <= m1) ? 0x10000u : 0u) + ((x[c] <= m3) ? 1u : 0u);
    }
    cnt = wave_sum_u32(cnt);
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
  for (int q = 0; q < RING_VECS; q++) {
    const uint4 v = ring_fetch(R, q, lane);
    const unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      cnt += ((x[c] <= lo1) ? 0x10000u : 0u) + ((x[c] <= lo3) ? 1u : 0u);
      if (x[c] > lo1) s1 = min(s1, x[c]);
      if (x[c] > lo3) s3 = min(s3, x[c]);
    }
  }
  cnt = wave_sum_u32(cnt);
  s1 = wave_min_u32(s1);
  s3 = wave_min_u32(s3);
  uint4 r;
  r.x = lo1;
  r.z = lo3;
  r.y = ((int)(cnt >> 16) >= k1 + 2 || s1 == KEY_NONE) ? lo1 : s1;
  r.w = ((int)(cnt & 0xFFFFu) >= k3 + 2 || s3 == KEY_NONE) ? lo3 : s3;
  return r;
}

// two-sided sweep around anchors gA, gB (any keys in (0, KEY_NONE), present or not) -> two fresh trackers
__device__ __forceinline__ void wave_rebuild_pair(const RingView& R, const int lane, const unsigned gA, const unsigned gB,
                                                  const int n, QTrack& A, QTrack& B) {
  unsigned cleA = 0u, cgeA = 0u, cleB = 0u, cgeB = 0u;
  L4 pdA = {KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE}, sdA = pdA, pdB = pdA, sdB = pdA;
  const unsigned gAp = gA + 1u, gAm = gA - 1u, gBp = gB + 1u, gBm = gB - 1u;
#pragma unroll 1
  for (int q = 0; q < RING_VECS; q++) {
    const uint4 v4 = ring_fetch(R, q, lane);
    const unsigned xs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const unsigned x = xs[c4];
      unsigned dsA, dpA, dsB, dpB;
      SDC_SUB_COUNT(dsA, cleA, x, gAp);   // x - (g+1) borrows <=> x <= g
      SDC_SUB_COUNT(dpA, cgeA, gAm, x);   // (g-1) - x borrows <=> x >= g
      SDC_SUB_COUNT(dsB, cleB, x, gBp);
      SDC_SUB_COUNT(dpB, cgeB, gBm, x);
      l4_sweep_insert(sdA, dsA);
      l4_sweep_insert(pdA, dpA);
      l4_sweep_insert(sdB, dsB);
      l4_sweep_insert(pdB, dpB);
    }
  }
  // across lanes: one copy of the merge code, the four lists rotate through it
#pragma unroll 1
  for (int r = 0; r < 4; r++) {
    wave_merge_l4(pdA);
    const L4 t = pdA;
    pdA = sdA;
    sdA = pdB;
    pdB = sdB;
    sdB = t;
  }
  auto finish = [&](unsigned g, unsigned cle, unsigned cge, const L4& pd, const L4& sd, QTrack& q) {
    cle = wave_sum_u32(cle);
    cge = wave_sum_u32(cge);
    q.g = g;
    const int n_empty = SDC_HIST_STRIDE - n;   // empty slots (KEY_NONE) satisfy x >= g
    q.c_le = (int)cle;
    q.c_lt = n - ((int)cge - n_empty);
    // a legitimate predecessor distance is < g; a legitimate successor distance is < KEY_NONE - g - 1
    const unsigned smax = KEY_NONE - g - 1u;
    q.np = (pd.e0 < g) + (pd.e1 < g) + (pd.e2 < g) + (pd.e3 < g);
    q.ns = (sd.e0 < smax) + (sd.e1 < smax) + (sd.e2 < smax) + (sd.e3 < smax);
    q.P.e0 = pd.e0 < g ? g - 1u - pd.e0 : 0u;
    q.P.e1 = pd.e1 < g ? g - 1u - pd.e1 : 0u;
    q.P.e2 = pd.e2 < g ? g - 1u - pd.e2 : 0u;
    q.P.e3 = pd.e3 < g ? g - 1u - pd.e3 : 0u;
    q.S.e0 = sd.e0 < smax ? g + 1u + sd.e0 : KEY_NONE;
    q.S.e1 = sd.e1 < smax ? g + 1u + sd.e1 : KEY_NONE;
    q.S.e2 = sd.e2 < smax ? g + 1u + sd.e2 : KEY_NONE;
    q.S.e3 = sd.e3 < smax ? g + 1u + sd.e3 : KEY_NONE;
  };
  finish(gA, cleA, cgeA, pdA, sdA, A);
  finish(gB, cleB, cgeB, pdB, sdB, B);
}

// clipped mean / population std straight from the ring, fp64, centred on `ctr` (tiny histories)
__device__ __forceinline__ void wave_direct_moments(const RingView& R, const int lane, const int n, const double lb,
                                                    const double ub, const double ctr, double& mean, double& sd) {
  double s = 0.0, s2 = 0.0;
#pragma unroll 1
  for (int q = 0; q < RING_VECS; q++) {
    const uint4 v = ring_fetch(R, q, lane);
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
  const double m0 = s / (double)n;
  mean = ctr + m0;
  const double var = s2 / (double)n - m0 * m0;
  sd = var > 0 ? sqrt(var) : 0.0;
}

// Tail corrections straight from the ring (an env whose tails do not fit the sets: more than ~480 keys beyond a clip
// bound): sum (v - bound), sum (v^2 - bound^2) over the keys >= kub and over the keys < klb, and how many there are.
__device__ __forceinline__ void tails_direct(const RingView& R, const int lane, const Bounds& b, double& t1, double& t2,
                                             int& n_hi, int& n_lo) {
  double a1 = 0.0, a2 = 0.0;
  unsigned c = 0u;   // packed: n_hi << 16 | n_lo
  const double ub2 = b.ub * b.ub, lb2 = b.lb * b.lb;
#pragma unroll 1
  for (int q = 0; q < RING_VECS; q++) {
    const uint4 v4 = ring_fetch(R, q, lane);
    const unsigned xs[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const unsigned x = xs[c4];
      if (x != KEY_NONE && (x >= b.kub || x < b.klb)) {
        const double v = key_f64(x);
        if (x >= b.kub) {
          a1 += v - b.ub;
          a2 += v * v - ub2;
          c += 0x10000u;
        } else {
          a1 += v - b.lb;
          a2 += v * v - lb2;
          c += 1u;
        }
      }
    }
  }
  t1 = wave_sum_f64(a1);
  t2 = wave_sum_f64(a2);
  c = wave_sum_u32(c);
  n_hi = (int)(c >> 16);
  n_lo = (int)(c & 0xFFFFu);
}

// Full rebuild of one env's reward state from its ring (n >= 2 keys, this step's key included).
//   n < SMALL_N: only this step's clipped mean / std, directly.
//   else: fresh quartile trackers, total sums, this step's tail corrections T1 / T2 by a direct sweep, and -- if the
//   tails fit -- tail sets (into L) whose thresholds leave ~128 keys (at least a quarter of a set) of slack inside
//   the clip bounds; `direct` is set when a tail has more keys than a set can hold.
struct Rebuilt {
  QTrack q1, q3;
  unsigned tau[2];
  double A1, A2, T1, T2;
  double mean, sd;   // n < SMALL_N only
  Bounds b;
  bool direct;
};
__device__ __forceinline__ Rebuilt rebuild_state(const RingView& R, const int lane, const int n, TailLds& L, double* sums2) {
  Rebuilt o;
  const bool tiny = n < SMALL_N;
  o.direct = false;
  o.tau[0] = o.tau[1] = 0u;   // everything: the history still fits a set
  o.A1 = o.A2 = o.T1 = o.T2 = o.mean = o.sd = 0.0;
  int ra, rb;
  quartile_ranks(n, ra, rb);
  // one copy of the bisection: first the quartile ranks, then (if needed) the ranks of the tail thresholds
#pragma unroll 1
  for (int ph = 0; ph < 2; ph++) {
    const uint4 r = wave_bisection(R, lane, ra, rb);
    if (ph == 1) {
      o.tau[0] = sfl(r.z);    // keys >  key at rank n-1-off_hi
      o.tau[1] = sfl(~r.x);   // keys <  key at rank off_lo
      break;
    }
    o.b = clip_bounds(n, r.x, r.y, r.z, r.w);
    if (tiny) {
      wave_direct_moments(R, lane, n, o.b.lb, o.b.ub, o.b.ctr, o.mean, o.sd);
      return o;
    }
    wave_rebuild_pair(R, lane, sfl(r.x), sfl(r.z), n, o.q1, o.q3);
    int n_hi, n_lo;
    tails_direct(R, lane, o.b, o.T1, o.T2, n_hi, n_lo);
    o.direct = n_hi > SDC_TAIL_CAP - 96 || n_lo > SDC_TAIL_CAP - 96;   // sets need >= 64 keys of slack to be stable
    if (o.direct || n <= SDC_TAIL_CAP) break;
    // thresholds by rank: the tail itself plus 128 keys of slack (at least half a set, at most all but 32 slots)
    ra = min(min(max(n_lo + 128, SDC_TAIL_CAP / 2), SDC_TAIL_CAP - 32), n - 1);
    rb = max(n - 1 - min(max(n_hi + 128, SDC_TAIL_CAP / 2), SDC_TAIL_CAP - 32), 0);
  }
  tails_collect(R, lane, o.direct ? KEY_NONE : o.tau[0], o.direct ? KEY_NONE : o.tau[1], L, sums2);
  o.A1 = sums2[0];
  o.A2 = sums2[1];
  return o;
}

}  // namespace sdc_rw
