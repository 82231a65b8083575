// sdc_step.hip -- the single-step kernels: ONE launch = one env-step of all N environments.
//   sdc_dynamics_fast_kernel   the common case, two envs per wavefront (sdc_pairstep.hpp pair_step<true>)
//   sdc_dynamics_quad_kernel   the common case, four envs per wavefront (quad_step): batches above 5 632 envs
//   sdc_dynamics_kernel        everything else (several episode steps in one batch, rule-based policies, alternate rewards, ...)
// The multi-step kernels (sdc_rollout*, the closed loop) are in sdc_rollout.hip, the lane-per-env kernel for the largest batches
// in sdc_wide.hip.
#include "sdc_pairstep.hpp"
#include "sdc_sweep.hpp"

// One launch of this kernel is one env-step of all N environments.
template <bool FAST>
__device__ __forceinline__ void dynamics_launch(const SdcDev& S, PairShared* shs, double* kt, const int rel_hint, const int32_t* __restrict__ actions,
                                                float* __restrict__ obs, float* __restrict__ share_obs,
                                                unsigned char* __restrict__ done, float* __restrict__ info,
                                                float* __restrict__ final_obs, float* __restrict__ rew) {
  const KernargTouch ktouch = kernarg_touch();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / SDC_WAVE));   // wave-uniform, in an SGPR
  const int lane = threadIdx.x % SDC_WAVE;
  kernarg_touch_done(ktouch);
  const int pair_blocks = (int)gridDim.x - S.sweep_blocks;
  // the sweep workgroups sit FIRST in the grid (see below)
  const int sweep_first = 0;
  const int bx = (int)blockIdx.x;
  if (bx >= sweep_first && bx < sweep_first + S.sweep_blocks) {
    // WHERE the 32 sweep workgroups sit in the grid decides how the env pairs' workgroups land on the CUs (round 3,
    // tools/wave_tail.py: every pair wavefront stamps the SIMD it ran on).  Inserted after the first 320 pair workgroups
    // (round 2) they pushed the dispatcher off its stride: 128 SIMDs received THREE pair wavefronts and 128 only one, and
    // the last wavefront of every launch (300 of 300) was one of a trio, ~1.2 us behind the rest.  First or last in the
    // grid every SIMD gets exactly two.  Last, the sweeps -- dispatched last, the youngest wavefronts of their SIMDs --
    // end the launch (15.4 us per step); FIRST and at raised issue priority they are done while the pairs still run:
    // 13.1 us per step against 13.6 with the insertion point at 320 (uniform-random actions; all-idle actions, four
    // times the requests: 14.1 against 15.1).
    static_assert(sizeof(sdc_rw::CoopLds) <= sizeof(PairShared) * SDC_STEP_WPB, "the sweep workgroup's LDS");
    serve_recentring_requests_coop(S, bx - sweep_first, wave, lane, *reinterpret_cast<sdc_rw::CoopLds*>(shs));
    return;
  }
  const int pb = bx < sweep_first ? bx : bx - S.sweep_blocks;          // index among the pair workgroups
  const int env0 = (first_pair_of_block(pb, pair_blocks) + wave) * EPW;
  if (env0 >= S.n_envs) return;
  // (of the two env pairs that share a SIMD at 4096 envs, the one whose workgroup arrived in the second round of 256 -- one per
  // CU -- would finish ~1.8 us after the other: set_round_priority)
  set_round_priority(pb, pair_blocks);
  if (!FAST && lane == 0) prof_stamp(S, SDC_PROF_DYNAMICS, env0, 0);
  pair_step<FAST>(S, shs[wave], env0, lane, rel_hint, actions, obs, share_obs, done, info, final_obs, rew,
                  FAST ? nullptr : S.actions_out, S.step_no, true, kt, true);
  if (!FAST && lane == 0) prof_stamp(S, SDC_PROF_DYNAMICS, env0, 1);
}
// (three resident wavefronts per SIMD: two env pairs + room for a spare one, <= 168 VGPRs)
// the general kernel, and the one for the common case (see pair_dynamics; the host picks: sdc_capi.hip fast_case)
extern "C" __global__ SDC_STEP_BOUNDS void sdc_dynamics_kernel(
    SdcDev S, const int rel_hint, const int32_t* __restrict__ actions, float* __restrict__ obs, float* __restrict__ share_obs,
    unsigned char* __restrict__ done, float* __restrict__ info, float* __restrict__ final_obs, float* __restrict__ rew) {
  __shared__ PairShared shs[SDC_STEP_WPB];
  __shared__ double ktab[SDC_K_LDS];
  dynamics_launch<false>(S, shs, ktab, rel_hint, actions, obs, share_obs, done, info, final_obs, rew);
}
extern "C" __global__ SDC_STEP_BOUNDS void sdc_dynamics_fast_kernel(
    SdcDev S, const int rel_hint, const int32_t* __restrict__ actions, float* __restrict__ obs, float* __restrict__ share_obs,
    unsigned char* __restrict__ done, float* __restrict__ info, float* __restrict__ final_obs, float* __restrict__ rew) {
  __shared__ PairShared shs[SDC_STEP_WPB];
  __shared__ double ktab[SDC_K_LDS];
  dynamics_launch<true>(S, shs, ktab, rel_hint, actions, obs, share_obs, done, info, final_obs, rew);
}

// the common case, four envs per wavefront (quad_step): workgroup = 4 wavefronts = 16 envs; the sweep workgroups as above
__device__ __forceinline__ void quad_launch(const SdcDev& S, QuadShared* shs, double* kt, const int rel_hint,
                                            const int32_t* __restrict__ actions, float* __restrict__ obs,
                                            float* __restrict__ share_obs, unsigned char* __restrict__ done,
                                            float* __restrict__ info, float* __restrict__ final_obs, float* __restrict__ rew) {
  const KernargTouch ktouch = kernarg_touch();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / SDC_WAVE));
  const int lane = threadIdx.x % SDC_WAVE;
  kernarg_touch_done(ktouch);
  const int env_blocks = (int)gridDim.x - S.sweep_blocks;
  const int bx = (int)blockIdx.x;
  if (bx < S.sweep_blocks) {     // (the sweep workgroups first in the grid: see dynamics_launch)
    static_assert(sizeof(sdc_rw::CoopLds) <= sizeof(QuadShared) * SDC_STEP_WPB, "the sweep workgroup's LDS");
    serve_recentring_requests_coop(S, bx, wave, lane, *reinterpret_cast<sdc_rw::CoopLds*>(shs));
    return;
  }
  const int pb = bx - S.sweep_blocks;
  const int env0 = (first_pair_of_block(pb, env_blocks) + wave) * QE;
  if (env0 >= S.n_envs) return;
  set_round_priority(pb, env_blocks);
  quad_step<false>(S, shs[wave], env0, lane, rel_hint, actions, obs, share_obs, done, info, final_obs, rew, S.step_no, true, kt, true);
}
// (three resident wavefronts per SIMD = 12 envs: with the reward-side loads behind the dynamics (quad_step) the
// kernel fits 168 VGPRs without spilling; measured at 12 288 envs: 22.6 us per step against 30.4 with two)

extern "C" __global__ SDC_QUAD_BOUNDS void sdc_dynamics_quad_kernel(
    SdcDev S, const int rel_hint, const int32_t* __restrict__ actions, float* __restrict__ obs, float* __restrict__ share_obs,
    unsigned char* __restrict__ done, float* __restrict__ info, float* __restrict__ final_obs, float* __restrict__ rew) {
  __shared__ QuadShared shs[SDC_STEP_WPB];
  __shared__ double ktab[SDC_K_LDS];
  quad_launch(S, shs, ktab, rel_hint, actions, obs, share_obs, done, info, final_obs, rew);
}
