// sdc_ringpath.hpp -- the rare moments when an env's history ring must be read: in-wave primitives.
//
// sdc_trackers.hpp answers the reward normalisation from four 64-key rank windows and running sums.  The rank a
// window is centred on random-walks through it and gets near an edge every ~600 steps; the env's own wavefront then
// REFILLS the window (qt_refill): it drops the keys on the far side and sweeps the ring once -- 10 240 keys, 160
// per lane, coalesced dwordx4 loads -- for the keys just beyond the window's last key:
//   d = x - (pivot+1) borrows <=> x <= pivot (counted: how many copies of the pivot lie beyond the window);
//   otherwise d is the key's distance above the pivot; every lane keeps the 4 smallest it sees, and below the
//   smallest 4th-smallest of any lane those lists are complete: compacted, ranked by counting, placed.
// Extending downwards is the same code on complemented keys with the window reversed.
//
// The dynamics kernel refills AHEAD of need and at the END of a step (when a window could run out on the next step
// in the worst case): the memory system is quiet then, and a wavefront that is one step away from a refill has had
// issue priority and an L2 prefetch of its ring since the start of the launch.  A full REBUILD (4-rank bisections on
// the key space + a collecting sweep per window pair + one fp64 sweep for the sums) bootstraps everything on the
// first steps and after state injection, and is the fallback whenever a window turns out not to cover.
//
// Why in-wave and not a separate reward kernel (round-1 measurements, MI355X, 4096 envs): a kernel that streams
// every env's ring each step is HBM-bound at >= 23 us (32 us in practice); a kernel that only serves the envs that
// need their ring still took 16-22 us, because its few workgroups per CU are latency-bound single waves
// (~6 ns per instruction with nothing to overlap, cold instruction cache).  Inside the dynamics kernel the same
// work hides behind 15 other resident wavefronts per CU.
#pragma once
#include "sdc_trackers.hpp"

namespace sdc_rw {

constexpr int RING_VECS = SDC_HIST_STRIDE / 4 / SDC_WAVE;   // dwordx4 loads per lane for one pass over the ring (40)

// ------------------------------------------------------------------------------------------------
// Wave reductions on the DPP data path: xor 1, xor 2 (quad_perm), row_half_mirror, row_mirror reduce within each row
// of 16 lanes; row_bcast15 / row_bcast31 carry the rows into lane 63, which holds the result.
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
// One pass over the ring: f(x0, x1, x2, x3) for every dwordx4 of this lane.
//   ring_sweep: loads in batches of 5, each batch fully in flight before its first use (the rebuild's sweeps: they
//   run with most of the step's state live, so registers are short);
//   ring_sweep_pipelined: the same batches double-buffered -- while one is consumed the next is in flight (the
//   refill, whose sweep sits on the kernel's critical path).
template <class F>
__device__ __forceinline__ void ring_sweep(const RingView& R, const int lane, F&& f) {
  constexpr int BATCH = RING_VECS / 8;
#pragma unroll 1
  for (int hf = 0; hf < RING_VECS / BATCH; hf++) {
    uint4 v[BATCH];
#pragma unroll
    for (int q = 0; q < BATCH; q++) v[q] = ring_fetch(R, hf * BATCH + q, lane);
#pragma unroll
    for (int q = 0; q < BATCH; q++) f(v[q].x, v[q].y, v[q].z, v[q].w);
  }
}
// (BATCH: 5 where the sweep runs inside an env pair's wavefront, whose registers are short; 10 in the spare wavefronts
// that do nothing else -- 13.13 -> 13.00 us per step; 20 no longer fits their registers either: 28 us)
template <int BATCH, class F>
__device__ __forceinline__ void ring_sweep_pipelined(const RingView& R, const int lane, F&& f) {
  constexpr int NB = RING_VECS / BATCH;   // batches of loads in flight, taken in pairs
  static_assert(NB * BATCH == RING_VECS && (NB & 1) == 0, "batching assumes an even number of full batches");
  uint4 a[BATCH], b[BATCH];
#pragma unroll
  for (int q = 0; q < BATCH; q++) a[q] = ring_fetch(R, q, lane);
#pragma unroll 1
  for (int hb = 0; hb < NB; hb += 2) {
#pragma unroll
    for (int q = 0; q < BATCH; q++) b[q] = ring_fetch(R, (hb + 1) * BATCH + q, lane);
#pragma unroll
    for (int q = 0; q < BATCH; q++) f(a[q].x, a[q].y, a[q].z, a[q].w);
    if (hb + 2 < NB) {
#pragma unroll
      for (int q = 0; q < BATCH; q++) a[q] = ring_fetch(R, (hb + 2) * BATCH + q, lane);
    }
#pragma unroll
    for (int q = 0; q < BATCH; q++) f(b[q].x, b[q].y, b[q].z, b[q].w);
  }
}

// ------------------------------------------------------------------------------------------------
// LDS scratch of the ring paths (one wavefront = one workgroup: LDS operations of a wavefront complete in order)
struct TailLds {
  unsigned keys[2][256];
  unsigned cnt[2];
};

// ------------------------------------------------------------------------------------------------
// QUARTILE TRACKER REFILL.
enum { REFILL_NONE = 0, REFILL_UP = 1, REFILL_DOWN = 2 };

// AHEAD-OF-NEED test on a tracker of the ring's n keys: the next step moves the wanted rank by at most one inside
// the window and takes at most one key out of it.
// (quartile windows: margins 3 / 6; a clip bound may cross a few keys in one step: margins 10 / 10)
__device__ __forceinline__ int qt_refill_ahead(const QTrack& q, const int k_next, const int n, const int m_lo, const int m_hi) {
  if (q.hi <= 0) return REFILL_NONE;        // nothing to extend: the end-of-step rebuild will create it
  const int t = k_next - q.r0;
  if (t > q.hi - m_hi && q.r0 + q.hi < n) return REFILL_UP;
  if (t < m_lo && q.r0 > 0) return REFILL_DOWN;
  return REFILL_NONE;
}

__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// Re-centre a tracker's window on rank k of the n keys in the ring: keep the part of the window that stays, fetch
// what lies beyond its last (UP) / first (DOWN) key with ONE sweep:
//   d = x - (pivot+1) borrows <=> x <= pivot (counted: the copies of the pivot beyond the window); otherwise d is
//   the key's distance above the pivot, and every lane keeps the 4 smallest it sees (v_med3_u32 insertion network).
// Below D = the smallest 4th-smallest of any lane the lanes' lists are COMPLETE (a key a lane dropped is >= that
// lane's 4th); those ~45 keys (64 lanes x 4 slots: a birthday bound) are compacted, ranked by counting and placed.
// A run of equal keys can stop the complete part short; then another sweep continues from that key.
// `side` = KEY_NONE for a window over complemented keys (the lower clip bound's), else 0.
template <int BATCH = 5>
__device__ __forceinline__ void qt_refill(QTrack& q, const int dir, const int k, const int n, const RingView& R, const int lane,
                                          TailLds& L, const unsigned side) {
  // Work in a space where the window grows upwards: for DOWN complement the keys, reverse the window and count ranks
  // from the top.
  const unsigned fd = dir == REFILL_DOWN ? KEY_NONE : 0u;
  const unsigned f = fd ^ side;               // what turns a ring key into a key of that space
  const int hi = q.hi;
  unsigned w = q.w;
  int r0 = q.r0, kk = k;
  if (fd) {
    const unsigned rv = lane_gather(q.w, (hi - 1 - lane) & 63);
    w = lane < hi ? ~rv : KEY_NONE;
    r0 = n - (q.r0 + hi);
    kk = n - 1 - k;
  }
  const int n_empty = f ? SDC_HIST_STRIDE - n : 0;       // complemented empty slots (0) count as <= pivot
  int top = r0 + hi;                                     // first rank above what the window holds so far
  const int s = min(max(kk - WIN / 2 - r0, 0), hi - 1);  // lanes to drop at the bottom
  const int kept = hi - s;
  unsigned pivot = lane_key(w, hi - 1);
  // the new window is assembled in LDS: first the kept keys
  unsigned* win = &L.keys[1][0];
  win[lane] = KEY_NONE;
  wave_sync();   // (one wavefront: orders the compiler's view of the cross-lane LDS traffic)
  if (lane >= s && lane < hi) win[lane - s] = w;
  int filled = kept;
  // Rounds: normally one.  A run of equal keys that overflows some lane's list stops the complete part short of it;
  // the next round then starts from that key (its copies are what the borrow count measures).
#pragma unroll 1
  for (int round = 0; round < 8 && filled < WIN && top < n; round++) {
    const unsigned pp = pivot + 1u;
    const unsigned smax = KEY_NONE - pp;                 // a legitimate distance is below this
    unsigned c = 0u;
    unsigned e0 = KEY_NONE, e1 = KEY_NONE, e2 = KEY_NONE, e3 = KEY_NONE;   // this lane's 4 smallest distances, ascending
    ring_sweep_pipelined<BATCH>(R, lane, [&](unsigned x0, unsigned x1, unsigned x2, unsigned x3) {
      const unsigned xs[4] = {x0 ^ f, x1 ^ f, x2 ^ f, x3 ^ f};
#pragma unroll
      for (int i = 0; i < 4; i++) {
        unsigned d;
        SDC_SUB_COUNT(d, c, xs[i], pp);
        e3 = umed3(e2, d, e3);
        e2 = umed3(e1, d, e2);
        e1 = umed3(e0, d, e1);
        e0 = min(e0, d);
      }
    });
    const int extra = (int)wave_sum_u32(c) - n_empty - top;   // copies of the pivot not in the window yet
    const unsigned D = min(wave_min_u32(e3), smax);
    if (extra < 0) {
      filled = -1;   // the tracker does not describe this ring
      break;
    }
    // compact the complete part of the lists into LDS: slot j of every lane in turn (ballot + prefix count)
    const unsigned mine3[3] = {e0, e1, e2};                   // (e3 >= D always)
    int m = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const unsigned long long mk = __ballot(mine3[j] < D);
      if (mine3[j] < D) L.keys[0][m + (int)__popcll(mk & ((1ull << lane) - 1ull))] = mine3[j];
      m += (int)__popcll(mk);
    }
    if (lane < 4) L.keys[0][m + lane] = KEY_NONE;             // pad the list to a multiple of 4 (ranks below nothing)
    // copies of the pivot first
    if (lane < extra && filled + lane < WIN) win[filled + lane] = pivot;
    wave_sync();
    // then the caught keys, each at the position its rank gives it: rank = how many caught distances are smaller
    // (one borrow-count per list entry, one key per lane and pass).  Equal keys get the same rank and land on one
    // slot; the slots they leave empty are filled from the left after the last round (the window is ascending).
#pragma unroll 1
    for (int c0 = 0; c0 < m; c0 += SDC_WAVE) {
      const unsigned mine = c0 + lane < m ? L.keys[0][c0 + lane] : 0u;
      unsigned rank = 0u;
#pragma unroll 2
      for (int i = 0; i < m; i += 4) {
        const uint4 v4 = *reinterpret_cast<const uint4*>(&L.keys[0][i]);
        unsigned t;
        SDC_SUB_COUNT(t, rank, v4.x, mine);
        SDC_SUB_COUNT(t, rank, v4.y, mine);
        SDC_SUB_COUNT(t, rank, v4.z, mine);
        SDC_SUB_COUNT(t, rank, v4.w, mine);
        (void)t;
      }
      const int pos = filled + extra + (int)rank;
      if (c0 + lane < m && pos < WIN) win[pos] = pp + mine;
    }
    wave_sync();
    filled += extra + m;
    top += extra + m;
    if (D >= smax) break;           // everything above the pivot has been seen
    if (filled - kept >= 12) break; // a usable extension: the rest can wait for the next refill
    pivot = pp + D;                 // the key that overflowed a lane: the next round counts its copies
  }
  if (filled <= kept) {             // (also filled == -1)
    q.hi = 0;
    return;
  }
  const int hi2 = min(WIN, filled);
  const int r2 = r0 + s;
  // slots that equal keys left empty take the key to their left: an inclusive prefix maximum over the lanes (DPP scan)
  unsigned v = win[lane];
  v = lane < hi2 && v != KEY_NONE ? v : 0u;
  v = max(v, dpp_u32<0x111, 0xF>(0u, v));   // row_shr:1
  v = max(v, dpp_u32<0x112, 0xF>(0u, v));   // row_shr:2
  v = max(v, dpp_u32<0x114, 0xF>(0u, v));   // row_shr:4
  v = max(v, dpp_u32<0x118, 0xF>(0u, v));   // row_shr:8
  v = max(v, dpp_u32<0x142, 0xA>(0u, v));   // row_bcast:15 -> rows 1, 3
  v = max(v, dpp_u32<0x143, 0xC>(0u, v));   // row_bcast:31 -> rows 2, 3
  if (fd) {
    const unsigned rv = lane_gather(v, (hi2 - 1 - lane) & 63);
    q.w = lane < hi2 ? ~rv : KEY_NONE;
    q.r0 = n - (r2 + hi2);
  } else {
    q.w = lane < hi2 ? v : KEY_NONE;
    q.r0 = r2;
  }
  q.hi = hi2;
}

// ------------------------------------------------------------------------------------------------
// THE SAME REFILL BY A WHOLE WORKGROUP (the spare sweep workgroups of a step launch): its NW wavefronts take a quarter of
// the ring each -- 10 loads per lane, all in flight at once -- keep their lanes' 4 smallest distances and borrow counts
// as above, and meet in LDS: the counts add up, D is the smallest 4th-smallest of ANY lane of the workgroup, the complete
// parts of all lists are compacted into one list (~140 keys instead of ~45: four times the lists, each seeing a quarter
// of the keys), ranked by counting (the entries dealt round the wavefronts) and placed.  One sweep then costs each of
// the four SIMDs ~0.5 us of issue instead of one SIMD ~2 us plus eight memory round trips -- and the two env-pair
// wavefronts that share a SIMD with a sweeping wavefront are what ends a step launch (tools/wave_tail.py).
// Every wavefront of the workgroup calls this with the same arguments (q: the same window in each); all return the same q.
constexpr int COOP_NW = 4;
constexpr int COOP_LIST = COOP_NW * 3 * SDC_WAVE + 4;   // every lane's three complete slots, worst case
struct CoopLds {
  unsigned list[COOP_LIST];
  unsigned win[WIN];
  unsigned cnt[COOP_NW], dmin[COOP_NW], m[COOP_NW];
};
__device__ __forceinline__ void qt_refill_coop(QTrack& q, const int dir, const int k, const int n, const RingView& R, const int lane,
                                               const int wave, CoopLds& C, const unsigned side) {
  static_assert(RING_VECS % COOP_NW == 0, "the ring splits evenly over the wavefronts");
  constexpr int PER = RING_VECS / COOP_NW;
  const unsigned fd = dir == REFILL_DOWN ? KEY_NONE : 0u;
  const unsigned f = fd ^ side;
  const int hi = q.hi;
  unsigned w = q.w;
  int r0 = q.r0, kk = k;
  if (fd) {
    const unsigned rv = lane_gather(q.w, (hi - 1 - lane) & 63);
    w = lane < hi ? ~rv : KEY_NONE;
    r0 = n - (q.r0 + hi);
    kk = n - 1 - k;
  }
  const int n_empty = f ? SDC_HIST_STRIDE - n : 0;
  int top = r0 + hi;
  const int s = min(max(kk - WIN / 2 - r0, 0), hi - 1);
  const int kept = hi - s;
  unsigned pivot = lane_key(w, hi - 1);
  if (wave == 0) {
    C.win[lane] = KEY_NONE;
    wave_sync();
    if (lane >= s && lane < hi) C.win[lane - s] = w;
  }
  __syncthreads();
  int filled = kept;
#pragma unroll 1
  for (int round = 0; round < 8 && filled < WIN && top < n; round++) {     // (every condition is the same in all wavefronts)
    const unsigned pp = pivot + 1u;
    const unsigned smax = KEY_NONE - pp;
    unsigned c = 0u;
    unsigned e0 = KEY_NONE, e1 = KEY_NONE, e2 = KEY_NONE, e3 = KEY_NONE;
    {
      uint4 v[PER];
#pragma unroll
      for (int i = 0; i < PER; i++) v[i] = ring_fetch(R, wave * PER + i, lane);
#pragma unroll
      for (int i = 0; i < PER; i++) {
        const unsigned xs[4] = {v[i].x ^ f, v[i].y ^ f, v[i].z ^ f, v[i].w ^ f};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          unsigned d;
          SDC_SUB_COUNT(d, c, xs[j], pp);
          e3 = umed3(e2, d, e3);
          e2 = umed3(e1, d, e2);
          e1 = umed3(e0, d, e1);
          e0 = min(e0, d);
        }
      }
    }
    const unsigned cw = wave_sum_u32(c), dw = wave_min_u32(e3);
    if (lane == 0) {
      C.cnt[wave] = cw;
      C.dmin[wave] = dw;
    }
    __syncthreads();
    const int extra = (int)(C.cnt[0] + C.cnt[1] + C.cnt[2] + C.cnt[3]) - n_empty - top;
    const unsigned D = min(min(min(C.dmin[0], C.dmin[1]), min(C.dmin[2], C.dmin[3])), smax);
    if (extra < 0) {
      filled = -1;
      break;
    }
    // the complete part of every lane's list (e3 >= D always), compacted: this wavefront's share behind the others'
    const unsigned mine3[3] = {e0, e1, e2};
    unsigned long long mk[3];
    int mw = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      mk[j] = __ballot(mine3[j] < D);
      mw += (int)__popcll(mk[j]);
    }
    if (lane == 0) C.m[wave] = (unsigned)mw;
    __syncthreads();
    int off = 0, m = 0;
#pragma unroll
    for (int v = 0; v < COOP_NW; v++) {
      if (v < wave) off += (int)C.m[v];
      m += (int)C.m[v];
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (mine3[j] < D) C.list[off + (int)__popcll(mk[j] & ((1ull << lane) - 1ull))] = mine3[j];
      off += (int)__popcll(mk[j]);
    }
    if (wave == 0 && lane < 4) C.list[m + lane] = KEY_NONE;      // pad to a multiple of 4 (ranks below nothing)
    if (wave == 0 && lane < extra && filled + lane < WIN) C.win[filled + lane] = pivot;   // copies of the pivot first
    __syncthreads();
    // rank = how many caught distances are smaller; the entries are dealt round the wavefronts, 64 per pass
#pragma unroll 1
    for (int c0 = wave * SDC_WAVE; c0 < m; c0 += COOP_NW * SDC_WAVE) {
      const unsigned mine = c0 + lane < m ? C.list[c0 + lane] : 0u;
      unsigned rank = 0u;
#pragma unroll 2
      for (int i = 0; i < m; i += 4) {
        const uint4 v4 = *reinterpret_cast<const uint4*>(&C.list[i]);
        unsigned t;
        SDC_SUB_COUNT(t, rank, v4.x, mine);
        SDC_SUB_COUNT(t, rank, v4.y, mine);
        SDC_SUB_COUNT(t, rank, v4.z, mine);
        SDC_SUB_COUNT(t, rank, v4.w, mine);
        (void)t;
      }
      const int pos = filled + extra + (int)rank;
      if (c0 + lane < m && pos < WIN) C.win[pos] = pp + mine;
    }
    __syncthreads();
    filled += extra + m;
    top += extra + m;
    if (D >= smax) break;
    if (filled - kept >= 12) break;
    pivot = pp + D;
  }
  if (filled <= kept) {
    q.hi = 0;
    return;
  }
  const int hi2 = min(WIN, filled);
  const int r2 = r0 + s;
  unsigned v = C.win[lane];      // (every wavefront reads the finished window: all return the same q)
  v = lane < hi2 && v != KEY_NONE ? v : 0u;
  v = max(v, dpp_u32<0x111, 0xF>(0u, v));
  v = max(v, dpp_u32<0x112, 0xF>(0u, v));
  v = max(v, dpp_u32<0x114, 0xF>(0u, v));
  v = max(v, dpp_u32<0x118, 0xF>(0u, v));
  v = max(v, dpp_u32<0x142, 0xA>(0u, v));
  v = max(v, dpp_u32<0x143, 0xC>(0u, v));
  if (fd) {
    const unsigned rv = lane_gather(v, (hi2 - 1 - lane) & 63);
    q.w = lane < hi2 ? ~rv : KEY_NONE;
    q.r0 = n - (r2 + hi2);
  } else {
    q.w = lane < hi2 ? v : KEY_NONE;
    q.r0 = r2;
  }
  q.hi = hi2;
}

// ------------------------------------------------------------------------------------------------
// REBUILD (bootstrap, injected state, a tracker or set that did not cover): everything from the ring, one wavefront.

// exact order statistics at four ranks by bisection on the key space (robust against any number of equal keys)
struct Keys4 {
  unsigned v[4];
};
__device__ __forceinline__ Keys4 wave_bisection4(const RingView& R, const int lane, const int r0, const int r1, const int r2,
                                                 const int r3) {
  unsigned kmin = KEY_NONE, kmax = 0u;
  ring_sweep(R, lane, [&](unsigned x0, unsigned x1, unsigned x2, unsigned x3) {
    kmin = min(min(kmin, x0), min(min(x1, x2), x3));
    kmax = max(max(kmax, x0 == KEY_NONE ? 0u : x0), max(max(x1 == KEY_NONE ? 0u : x1, x2 == KEY_NONE ? 0u : x2), x3 == KEY_NONE ? 0u : x3));
  });
  kmin = wave_min_u32(kmin);
  kmax = wave_max_u32(kmax);
  const int rk[4] = {r0, r1, r2, r3};
  unsigned lo[4] = {kmin, kmin, kmin, kmin}, hi[4] = {kmax, kmax, kmax, kmax};
#pragma unroll 1
  while (lo[0] < hi[0] || lo[1] < hi[1] || lo[2] < hi[2] || lo[3] < hi[3]) {
    unsigned mp[4], c[4] = {0u, 0u, 0u, 0u};   // count(key <= mid) = borrows of key - (mid + 1); mid < KEY_NONE - 1
#pragma unroll
    for (int j = 0; j < 4; j++) mp[j] = lo[j] + ((hi[j] - lo[j]) >> 1) + 1u;
    ring_sweep(R, lane, [&](unsigned x0, unsigned x1, unsigned x2, unsigned x3) {
      const unsigned xs[4] = {x0, x1, x2, x3};
#pragma unroll
      for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          unsigned d;
          SDC_SUB_COUNT(d, c[j], xs[i], mp[j]);
          (void)d;
        }
      }
    });
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int cj = (int)wave_sum_u32(c[j]);
      if (lo[j] < hi[j]) {
        if (cj >= rk[j] + 1) hi[j] = mp[j] - 1u; else lo[j] = mp[j];
      }
    }
  }
  Keys4 o;
#pragma unroll
  for (int j = 0; j < 4; j++) o.v[j] = sfl(lo[j]);
  return o;
}

// first rank / last rank of the window a tracker of rank k gets when it is built from scratch over n keys
__device__ __forceinline__ void window_ranks(const int k, const int n, int& a, int& b) {
  a = max(0, min(k - WIN / 2, n - WIN));
  b = min(a + WIN - 1, n - 1);
}
// Both trackers' windows from the keys vA <= vB at their first and last ranks (A, B): one sweep counts the keys up to
// vA and catches the keys strictly between vA and vB (fewer than 64) in LDS; they are ranked by counting and placed
// between the copies of vA below and of vB above.
__device__ __forceinline__ void windows_build(const RingView& R, const int lane, const Keys4& kv, const int A1, const int B1,
                                              const int A3, const int B3, TailLds& L, QTrack& q1, QTrack& q3) {
  if (lane < 2) L.cnt[lane] = 0u;
  unsigned c1 = 0u, c3 = 0u;                                  // count(key <= vA)
  const unsigned p1 = kv.v[0] + 1u, p3 = kv.v[2] + 1u;
  // a key strictly between: d = x - (vA + 1) < vB - vA - 1   (empty when vB <= vA + 1; then the compare never holds)
  const unsigned g1 = kv.v[1] - kv.v[0], g3 = kv.v[3] - kv.v[2];
  const unsigned w1 = g1 >= 2u ? g1 - 1u : 0u, w3 = g3 >= 2u ? g3 - 1u : 0u;
  ring_sweep(R, lane, [&](unsigned x0, unsigned x1, unsigned x2, unsigned x3) {
    const unsigned xs[4] = {x0, x1, x2, x3};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      unsigned d1, d3;
      SDC_SUB_COUNT(d1, c1, xs[i], p1);
      SDC_SUB_COUNT(d3, c3, xs[i], p3);
      if (d1 < w1) {
        const unsigned pos = atomicAdd(&L.cnt[0], 1u);
        if (pos < (unsigned)WIN) L.keys[0][pos] = xs[i];
      }
      if (d3 < w3) {
        const unsigned pos = atomicAdd(&L.cnt[1], 1u);
        if (pos < (unsigned)WIN) L.keys[1][pos] = xs[i];
      }
    }
  });
  wave_sync();
  auto finish = [&](const int side, const unsigned cle, const unsigned vA, const unsigned vB, const int A, const int B) {
    QTrack q;
    const int len = B - A + 1;
    const int nA = min((int)wave_sum_u32(cle) - A, len);       // copies of vA from rank A on
    const int mb = min((int)sfl(L.cnt[side]), WIN);            // keys strictly between
    const unsigned mine = lane < mb ? L.keys[side][lane] : KEY_NONE;
    int rank = 0;
#pragma unroll 1
    for (int i = 0; i < mb; i++) {
      const unsigned v = L.keys[side][i];
      rank += (v < mine || (v == mine && i < lane)) ? 1 : 0;
    }
    unsigned* win = &L.keys[side][WIN];
    win[lane] = lane < nA ? vA : (lane < len ? vB : KEY_NONE);
    wave_sync();   // (one wavefront: orders the compiler's view of the cross-lane LDS traffic)
    if (lane < mb && nA + rank < len) win[nA + rank] = mine;
    wave_sync();
    q.w = win[lane];
    q.r0 = A;
    q.hi = (nA >= 1 && nA + mb <= len) ? len : 0;             // (inconsistent counts: no tracker)
    return q;
  };
  q1 = finish(0, c1, kv.v[0], kv.v[1], A1, B1);
  q3 = finish(1, c3, kv.v[2], kv.v[3], A3, B3);
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

// Totals and tail sums straight from the ring: A1 / A2 = sum v, sum v^2 over all keys; per side (count, sum v,
// sum v^2) over the keys >= kub (side 0) and over the keys < klb (side 1).
struct RingSums {
  double A1, A2;
  int qc[2];
  double qs1[2], qs2[2];
};
__device__ __forceinline__ RingSums ring_sums(const RingView& R, const int lane, const Bounds& b) {
  double a1 = 0.0, a2 = 0.0, h1 = 0.0, h2 = 0.0, l1 = 0.0, l2 = 0.0;
  unsigned c = 0u;   // packed: n_hi << 16 | n_lo
  ring_sweep(R, lane, [&](unsigned x0, unsigned x1, unsigned x2, unsigned x3) {
    const unsigned xs[4] = {x0, x1, x2, x3};
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const unsigned x = xs[c4];
      if (x != KEY_NONE) {
        const double v = key_f64(x), v2 = v * v;
        const bool hi = x >= b.kub, lo = x < b.klb;
        a1 += v;
        a2 += v2;
        h1 += hi ? v : 0.0;
        h2 += hi ? v2 : 0.0;
        l1 += lo ? v : 0.0;
        l2 += lo ? v2 : 0.0;
        c += (hi ? 0x10000u : 0u) + (lo ? 1u : 0u);
      }
    }
  });
  RingSums o;
  o.A1 = wave_sum_f64(a1);
  o.A2 = wave_sum_f64(a2);
  o.qs1[0] = wave_sum_f64(h1);
  o.qs2[0] = wave_sum_f64(h2);
  o.qs1[1] = wave_sum_f64(l1);
  o.qs2[1] = wave_sum_f64(l2);
  c = wave_sum_u32(c);
  o.qc[0] = (int)(c >> 16);
  o.qc[1] = (int)(c & 0xFFFFu);
  return o;
}

// Full rebuild of one env's reward state from its ring (n >= 2 keys, this step's key included).
//   n < SMALL_N: only this step's clipped mean / std, directly.
//   else: the four rank windows, the total sums and the running tail sums.
struct Rebuilt {
  QTrack q1, q3, bu, bl;
  double A1, A2;
  int qc[2];
  double qs1[2], qs2[2];
  double mean, sd;   // n < SMALL_N only
  Bounds b;
  bool ok;
};
__device__ __forceinline__ Rebuilt rebuild_state(const RingView& R, const int lane, const int n, TailLds& L) {
  Rebuilt o;
  const bool tiny = n < SMALL_N;
  o.ok = false;
  o.A1 = o.A2 = o.mean = o.sd = 0.0;
  o.qc[0] = o.qc[1] = 0;
  o.qs1[0] = o.qs1[1] = o.qs2[0] = o.qs2[1] = 0.0;
  o.b = Bounds{0.0, 0.0, 0.0, 2u, 2u};
  o.q1.hi = o.q3.hi = o.bu.hi = o.bl.hi = 0;
  int k1, k3, ra, rb, rc, rd;
  quartile_ranks(n, k1, k3);
  window_ranks(k1, n, ra, rb);
  window_ranks(k3, n, rc, rd);
  // one copy of the bisection and of the window builder: first the quartile windows, then the clip-bound windows
#pragma unroll 1
  for (int ph = 0; ph < 2; ph++) {
    const Keys4 kv = wave_bisection4(R, lane, ra, rb, rc, rd);
    QTrack wa, wb;
    windows_build(R, lane, kv, ra, rb, rc, rd, L, wa, wb);
    if (!qt_valid(wa) || !qt_valid(wb)) return o;   // cannot happen with a consistent ring
    if (ph == 1) {
      o.bu = wa;
      // the lower bound's window lives on complemented keys: reverse it, count ranks from the top
      const unsigned rv = lane_gather(wb.w, (wb.hi - 1 - lane) & 63);
      o.bl.w = lane < wb.hi ? ~rv : KEY_NONE;
      o.bl.r0 = n - (wb.r0 + wb.hi);
      o.bl.hi = wb.hi;
      o.ok = true;
      break;
    }
    o.q1 = wa;
    o.q3 = wb;
    unsigned a1 = 0u, b1 = 0u, a3 = 0u, b3 = 0u;
    if (!qt_resolve(o.q1, k1, n, a1, b1) || !qt_resolve(o.q3, k3, n, a3, b3)) return o;
    o.b = clip_bounds(n, a1, b1, a3, b3);
    if (tiny) {
      wave_direct_moments(R, lane, n, o.b.lb, o.b.ub, o.b.ctr, o.mean, o.sd);
      return o;
    }
    const RingSums rs = ring_sums(R, lane, o.b);
    o.A1 = rs.A1;
    o.A2 = rs.A2;
    o.qc[0] = rs.qc[0]; o.qc[1] = rs.qc[1];
    o.qs1[0] = rs.qs1[0]; o.qs1[1] = rs.qs1[1];
    o.qs2[0] = rs.qs2[0]; o.qs2[1] = rs.qs2[1];
    // the first key at or above the upper bound has rank n - n_hi; the first key not below the lower bound rank n_lo
    window_ranks(n - rs.qc[0], n, ra, rb);
    window_ranks(rs.qc[1], n, rc, rd);
  }
  return o;
}

}  // namespace sdc_rw
