// sdc_sweep.hpp -- what the step kernels' launches share around the env wavefronts: the block -> env mapping, the kernel-argument
// touch, the service of the previous step's window re-centring requests by spare wavefronts, the issue priority by dispatch round.
#pragma once
#include "sdc_ringpath.hpp"
#include "sdc_tuning.hpp"


// block -> first env pair of the block.  Workgroup b runs on XCD b % 8 (the dispatcher deals workgroups round-robin to
// the 8 XCDs, each with its own L2): give every XCD a CONTIGUOUS range of envs, so that output lines shared by
// neighbouring envs (rew, done, the unaligned obs rows) are assembled in one L2 instead of being written back in pieces
// from several.
__device__ __forceinline__ int first_pair_of_block(const int bi, const int nb, const int wpb = SDC_STEP_WPB) {
  const int vb = (nb % 8 == 0) ? (bi % 8) * (nb / 8) + bi / 8 : bi;
  return vb * wpb;
}

// The spare wavefronts of a step launch (32 workgroups, first in the grid): they serve the re-centring requests of the
// previous step (see SdcRefillReq): one sweep over that env's ring as the previous step left it, the re-centred window out
// as a result.  serve_recentring_request: one wavefront serves one request alone (the lane-per-env kernel's one-wavefront workgroups: sdc_wide.hip);
// serve_recentring_requests_coop: the four wavefronts of a workgroup share each sweep (round 3, the default).
// The kernel arguments (SdcDev by value + the output pointers: ten 64-byte lines) are read by scalar loads wherever the
// compiler first needs a field -- several dependent batches, each a miss in the scalar cache at the start of a launch.
// One load per line up front brings them all in with a single round trip; what follows hits.
struct KernargTouch { unsigned t[8]; };
// (the last touched dword, 0x1c0, must lie inside the kernel-argument segment: SdcDev by value, rel_hint, eight pointers,
// then the 256 bytes of implicit arguments of code object v5 -- the grid size the kernel reads is among them)
static_assert(sizeof(SdcDev) + 8 + 8 * sizeof(void*) + 256 >= 0x1c4, "kernarg_touch reads past the kernel-argument segment");
__device__ __forceinline__ KernargTouch kernarg_touch() {
  KernargTouch k;
  const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile(
      "s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x40\n\ts_load_dword %2, %8, 0x80\n\ts_load_dword %3, %8, 0xc0\n\t"
      "s_load_dword %4, %8, 0x100\n\ts_load_dword %5, %8, 0x140\n\ts_load_dword %6, %8, 0x180\n\ts_load_dword %7, %8, 0x1c0"
      : "=&s"(k.t[0]), "=&s"(k.t[1]), "=&s"(k.t[2]), "=&s"(k.t[3]), "=&s"(k.t[4]), "=&s"(k.t[5]), "=&s"(k.t[6]), "=&s"(k.t[7])
      : "s"(ka));
  return k;
}
__device__ __forceinline__ void kernarg_touch_done(const KernargTouch& k) {     // (the registers stay reserved until here)
  asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(k.t[0]), "s"(k.t[1]), "s"(k.t[2]), "s"(k.t[3]), "s"(k.t[4]), "s"(k.t[5]),
               "s"(k.t[6]), "s"(k.t[7]));
}
// request j of set `set`, by ONE wavefront (the lane-per-env kernel's one-wavefront sweep workgroups: sdc_wide.hip)
__device__ __forceinline__ void serve_recentring_request(const SdcDev& S, const int set, const int j, const int lane, sdc_rw::TailLds& tl) {
  using namespace sdc_rw;
  const SdcRefillReq* rq = S.rq + set * S.rq_max + j;
  if (rq->step != S.step_no - 1) return;                            // stale (a multi-step launch came in between)
  const int env = rq->env, w = rq->win, n = rq->n;
  QTrack A = {rq->keys[lane], rq->r0, rq->hi};
  const RingView R = {reinterpret_cast<const uint4*>(S.hist + (size_t)env * SDC_HIST_STRIDE), rq->patch_slot, rq->patch_x};
  qt_refill<10>(A, rq->dir, rq->kt, n, R, lane, tl, w == 3 ? KEY_NONE : 0u);
  SdcRefillRes* rs = S.rs + set * S.rq_max + j;
  rs->keys[lane] = A.w;
  if (lane == 0) {
    rs->r0 = A.r0;
    rs->hi = A.hi;
    rs->step = S.step_no;
    rs->env_win = env * 4 + w;
  }
}

// The same service by the whole sweep workgroup: workgroup b takes requests b, b + 32, ... one after the other, its four
// wavefronts sharing each sweep (qt_refill_coop).  Every wavefront of the workgroup must call this (barriers inside).
__device__ __forceinline__ void serve_recentring_requests_coop(const SdcDev& S, const int wg, const int wave, const int lane,
                                                               sdc_rw::CoopLds& C) {
  using namespace sdc_rw;
  static_assert(SDC_STEP_WPB == COOP_NW, "one quarter of the ring per wavefront of the workgroup");
  const int set = S.step_no % 3;
  if (wg == 0 && wave == 0 && lane == 0) S.rq_count[(S.step_no + 2) % 3] = 0;     // the set the NEXT step's requests go to
  const int cnt = min(S.rq_count[set], S.rq_max);
  if (wg >= cnt) return;
  __builtin_amdgcn_s_setprio(SDC_SWEEP_PRIO);
#pragma unroll 1
  for (int j = wg; j < cnt; j += S.sweep_blocks) {
    const SdcRefillReq* rq = S.rq + set * S.rq_max + j;
    if (rq->step != S.step_no - 1) continue;                          // stale (a multi-step launch came in between)
    const int env = rq->env, w = rq->win, n = rq->n;
    QTrack A = {rq->keys[lane], rq->r0, rq->hi};
    const RingView R = {reinterpret_cast<const uint4*>(S.hist + (size_t)env * SDC_HIST_STRIDE), rq->patch_slot, rq->patch_x};
    qt_refill_coop(A, rq->dir, rq->kt, n, R, lane, wave, C, w == 3 ? KEY_NONE : 0u);
    if (wave == 0) {
      SdcRefillRes* rs = S.rs + set * S.rq_max + j;
      rs->keys[lane] = A.w;
      if (lane == 0) {
        rs->r0 = A.r0;
        rs->hi = A.hi;
        rs->step = S.step_no;
        rs->env_win = env * 4 + w;
      }
    }
    __syncthreads();      // (the LDS meeting point is reused by the next request)
  }
}

// Issue priority of an env workgroup by the dispatch round it arrives in (a round = 256 workgroups, one per CU: one more wavefront
// on every SIMD).  A SIMD issues from its OLDEST wavefront first, so of the wavefronts that share a SIMD the one from a later round
// would finish last by as much as the others took: up to ~2.75 rounds (all resident: three wavefronts per SIMD at most) the later
// rounds run the dynamics at raised priority, which evens them out (4096 envs: 1.8 us).  Beyond three rounds the wavefronts of the
// fourth, fifth ... round start when an older one ENDS: there the older ones should end early, and only the LAST round -- the
// wavefronts that end the launch -- is raised (round 4, four-env kernel: 16 384 envs 26.4 -> 24.9 us, 13 312 envs 23.9 -> 22.2,
// 24 576 envs 36.2 -> 34.1); with exactly three full rounds (12 288 envs: the spare sweep wavefronts push 128 env wavefronts into
// a fourth round of their own) no raise is best (22.4 -> 21.5).
__device__ __forceinline__ void set_round_priority(const int pb, const int n_blocks) {
  bool late;
  if (4 * n_blocks <= 11 * SDC_CUS) late = pb >= SDC_CUS;
  else if (n_blocks <= 3 * SDC_CUS) late = false;
  else late = pb >= ((n_blocks - 1) / SDC_CUS) * SDC_CUS;
  if (late)
    __builtin_amdgcn_s_setprio(SDC_LATE_PRIO);
  else
    __builtin_amdgcn_s_setprio(SDC_BASE_PRIO);
}
