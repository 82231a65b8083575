// sdc_rollout.hip -- the multi-step kernels: K env-steps per launch (sdc_rollout: action sequences known up front or chosen by the
// built-in rule-based policies; sdc_rollout_actor: the closed loop with the reference's actor networks inside the kernel).
#include "sdc_pairstep.hpp"
#include "sdc_sweep.hpp"
#include "sdc_actor.hpp"

// K env-steps per launch for action sequences that are known up front or chosen by the built-in rule-based policies
// (scripted evaluation, the reference's RBC / do-nothing baselines): every wavefront advances its own two envs K times
// -- envs do not interact, so there is nothing to wait for between steps; the dispatch ramp, the launch gap and the
// tail of a launch are paid once per K steps.  actions [K][N][3] (or null when every agent slot has a policy);
// obs [K][N][3][26], share_obs [K][N][29] (or null), rew [K][N][3], done [K][N], info [K][N][44] (or null) hold every
// step's outputs.  The host keeps K within the episode (sdc_rollout).
template <bool FAST>
__device__ __forceinline__ void rollout_launch(const SdcDev& S, PairShared* shs, double* kt, const int K, const int rel_hint,
                                               const int32_t* __restrict__ actions, float* __restrict__ obs,
                                               float* __restrict__ share_obs, unsigned char* __restrict__ done,
                                               float* __restrict__ info, float* __restrict__ final_obs, float* __restrict__ rew) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / SDC_WAVE));
  const int env0 = (first_pair_of_block((int)blockIdx.x, (int)gridDim.x) + wave) * EPW;
  const int lane = threadIdx.x % SDC_WAVE;
  const size_t N = (size_t)S.n_envs;
  if (env0 >= S.n_envs) return;
  if (!FAST && lane == 0) prof_stamp(S, SDC_PROF_DYNAMICS, env0, 0);
  int32_t* const aout = FAST ? nullptr : S.actions_out;
#pragma unroll 1
  for (int k = 0; k < K; k++) {
    if ((int)blockIdx.x >= SDC_CUS) __builtin_amdgcn_s_setprio(SDC_LATE_PRIO);    // (see sdc_dynamics_kernel)
    // (opaque copies: otherwise every per-env / per-lane address of the step is hoisted out of the loop and held in
    // registers across it)
    int env_k = env0;
    asm volatile("" : "+s"(env_k));
    const int lane_k = lane_fresh();   // (recomputed every step: two instructions instead of a register held across the loop)
    pair_step<FAST>(S, shs[wave], env_k, lane_k, (FAST || rel_hint >= 0) ? rel_hint + k : -1,
                    (FAST || actions) ? actions + (size_t)k * N * 3 : nullptr, obs + (size_t)k * N * SDC_OBS_OUT,
                    (FAST || share_obs) ? share_obs + (size_t)k * N * SDC_SHARE_OBS_DIM : nullptr, done + (size_t)k * N,
                    (FAST || info) ? info + (size_t)k * N * SDC_INFO_DIM : nullptr, k == K - 1 ? final_obs : nullptr,
                    rew + (size_t)k * N * 3, aout ? aout + (size_t)k * N * 3 : nullptr, S.step_no + k, false, kt, k == 0);
    // this wavefront's stores of step k are the loads of its step k + 1: complete them and drop stale lines of the
    // CU's vector L1 (workgroup scope: the L2 behind it is the same for both)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    wave_sync();
  }
  if (!FAST && lane == 0) prof_stamp(S, SDC_PROF_DYNAMICS, env0, 1);
}
extern "C" __global__ __launch_bounds__(SDC_WAVE * SDC_STEP_WPB, 8 / SDC_STEP_WPB) void sdc_rollout_kernel(
    SdcDev S, const int K, const int rel_hint, const int32_t* __restrict__ actions, float* __restrict__ obs,
    float* __restrict__ share_obs, unsigned char* __restrict__ done, float* __restrict__ info, float* __restrict__ final_obs,
    float* __restrict__ rew) {
  __shared__ PairShared shs[SDC_STEP_WPB];
  __shared__ double ktab[SDC_K_LDS];
  rollout_launch<false>(S, shs, ktab, K, rel_hint, actions, obs, share_obs, done, info, final_obs, rew);
}
extern "C" __global__ __launch_bounds__(SDC_WAVE * SDC_STEP_WPB, 8 / SDC_STEP_WPB) void sdc_rollout_fast_kernel(
    SdcDev S, const int K, const int rel_hint, const int32_t* __restrict__ actions, float* __restrict__ obs,
    float* __restrict__ share_obs, unsigned char* __restrict__ done, float* __restrict__ info, float* __restrict__ final_obs,
    float* __restrict__ rew) {
  __shared__ PairShared shs[SDC_STEP_WPB];
  __shared__ double ktab[SDC_K_LDS];
  rollout_launch<true>(S, shs, ktab, K, rel_hint, actions, obs, share_obs, done, info, final_obs, rew);
}

// the common case with four envs per wavefront (large batches: see quad_step)
extern "C" __global__ SDC_QUAD_BOUNDS void sdc_rollout_quad_kernel(
    SdcDev S, const int K, const int rel_hint, const int32_t* __restrict__ actions, float* __restrict__ obs,
    float* __restrict__ share_obs, unsigned char* __restrict__ done, float* __restrict__ info, float* __restrict__ final_obs,
    float* __restrict__ rew) {
  __shared__ QuadShared shs[SDC_STEP_WPB];
  __shared__ double ktab[SDC_K_LDS];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / SDC_WAVE));
  const int env0 = (first_pair_of_block((int)blockIdx.x, (int)gridDim.x) + wave) * QE;
  const int lane = threadIdx.x % SDC_WAVE;
  const size_t N = (size_t)S.n_envs;
  if (env0 >= S.n_envs) return;
#pragma unroll 1
  for (int k = 0; k < K; k++) {
    if ((int)blockIdx.x >= SDC_CUS) __builtin_amdgcn_s_setprio(SDC_LATE_PRIO);
    int env_k = env0;
    asm volatile("" : "+s"(env_k));
    const int lane_k = lane_fresh();   // (recomputed every step: two instructions instead of a register held across the loop)
    quad_step<false>(S, shs[wave], env_k, lane_k, rel_hint + k, actions + (size_t)k * N * 3, obs + (size_t)k * N * SDC_OBS_OUT,
                     share_obs + (size_t)k * N * SDC_SHARE_OBS_DIM, done + (size_t)k * N, info + (size_t)k * N * SDC_INFO_DIM,
                     k == K - 1 ? final_obs : nullptr, rew + (size_t)k * N * 3, S.step_no + k, false, ktab, k == 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    wave_sync();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// CLOSED LOOP in one launch: K env-steps with the three agents' ACTOR NETWORKS (sdc_actor.hpp: the reference's
// StochasticPolicy, 26 -> 64 -> 64 -> 3, fp32) evaluated inside the kernel between the steps -- observation -> actor ->
// action -> step never leaves the wavefront that owns the env pair, and there is no launch, no dispatch ramp and no
// host round trip per step.  The common case only (see pair_dynamics FAST).
// Workgroup = 8 wavefronts (16 envs) sharing ONE copy of the three networks in LDS (76 KB; with the wavefronts' own
// 5.5 KB each and the constant table ~122 KB of the CU's 160 KB: one workgroup per CU, two wavefronts per SIMD at 4096
// envs).  obs_in [N][3][26]: the observations the first actions are chosen from (the engine's latest).  actions_out
// [K][N][3] receives what the actors chose, logits_out [K][N][3][3] (or null) their logits.
#define SDC_ACTOR_WPB 8
struct ActorLds {
  PairShared shs[SDC_ACTOR_WPB];
  double ktab[SDC_K_LDS];
  SdcActorDev net[3];
};
static_assert(offsetof(ActorLds, net) % 16 == 0, "the weights are read as ds_read_b128");
extern "C" __global__ __launch_bounds__(SDC_WAVE * SDC_ACTOR_WPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void
sdc_rollout_actor_kernel(SdcDev S, const int K, const int rel_hint, const SdcActorDev* __restrict__ nets,
                         const float* __restrict__ obs_in, const int sample, float* __restrict__ obs,
                         float* __restrict__ share_obs, unsigned char* __restrict__ done, float* __restrict__ info,
                         float* __restrict__ final_obs, float* __restrict__ rew, int32_t* __restrict__ actions_out,
                         float* __restrict__ logits_out, float* __restrict__ obs_latch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  ActorLds& L = *reinterpret_cast<ActorLds*>(lds_raw);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / SDC_WAVE));
  const int lane = threadIdx.x % SDC_WAVE;
  // the three networks: global -> LDS, the whole workgroup copying (coalesced uint4), once per launch
  {
    const uint4* src = reinterpret_cast<const uint4*>(nets);
    uint4* dst = reinterpret_cast<uint4*>(L.net);
    for (int i = (int)threadIdx.x; i < (int)(3 * sizeof(SdcActorDev) / 16); i += SDC_WAVE * SDC_ACTOR_WPB) dst[i] = src[i];
  }
  __syncthreads();
  const int env0 = (first_pair_of_block((int)blockIdx.x, (int)gridDim.x, SDC_ACTOR_WPB) + wave) * EPW;
  if (env0 >= S.n_envs) return;
  PairShared& sh = L.shs[wave];
  const size_t N = (size_t)S.n_envs;
  const int h = lane >> 5, l = lane & (HL - 1);
  const int envc = env0 + h;
  // the observation pool of both envs from the latest observations (inverse of obs_padded_at / share layout:
  // pool[0..25] = agent_ls, [26] = agent_dc[11], [27] = agent_dc[13], [28] = agent_bat[12])
  if (l < SDC_POOL_DIM) {
    const float* o = obs_in + (size_t)envc * SDC_OBS_OUT;
    sh.pool[h][l] = l < SDC_OBS_PAD ? o[l] : (l == SDC_P_WNEXT ? o[SDC_OBS_PAD + 11] : (l == SDC_P_NTNEXT ? o[SDC_OBS_PAD + 13] : o[2 * SDC_OBS_PAD + 12]));
  }
  // the sampler's key: (seed, global env index, EPISODE NUMBER, episode step) -- all of it env state or configuration, so that a
  // checkpoint restored later, or the same steps asked for in launches of other lengths, draw the same actions (rounds 3 keyed
  // on the library's launch counter).  A launch never crosses an episode end: one load per launch.
  const unsigned ep_key = S.rec[(size_t)envc * SDC_REC_DWORDS + R_EPISODE];
  wave_sync();
#ifdef SDC_ACTOR_CLOCK      // (measurement build: shader-clock cycles per phase, summed over the K steps, into info slots 38..40 of the last step)
  unsigned long long ck[3] = {0, 0, 0}, c0, c1;
#define SDC_CK(i) c1 = __builtin_amdgcn_s_memtime(); ck[i] += c1 - c0; c0 = c1;
#else
#define SDC_CK(i)
#endif
#pragma unroll 1
  for (int k = 0; k < K; k++) {
    int env_k = env0;
    asm volatile("" : "+s"(env_k));
    const int lane_k = lane_fresh();   // (recomputed every step: two instructions instead of a register held across the loop)
#ifdef SDC_ACTOR_CLOCK
    c0 = __builtin_amdgcn_s_memtime();
#endif
    // ---- the three actors on the current observations (in the LDS pool) ------------------------------------------------
    int act[3];
    const int lk = lane_k & (HL - 1), hk = lane_k >> 5;
    const int rel_now = rel_hint + k;
    float lg[3][3];
    {
      float x[3];
#pragma unroll
      for (int a = 0; a < 3; a++) x[a] = lk < SDC_ACT_IN ? obs_padded_at(sh.pool[hk], a * SDC_OBS_PAD + lk) : 0.0f;
#ifdef SDC_ACTOR_SKIP
      // (measurement: the kernel without the networks -- pseudo-random actions from a hash)
#pragma unroll
      for (int a = 0; a < 3; a++) {
        unsigned hsh = (unsigned)(env_k + hk) * 2654435761u + (unsigned)(rel_now * 3 + a) * 40503u;
        hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
        const int pick = (int)(hsh % 3u);
        lg[a][0] = pick == 0 ? 1.f : 0.f; lg[a][1] = pick == 1 ? 1.f : 0.f; lg[a][2] = pick == 2 ? 1.f : 0.f;
      }
      (void)x;
#else
      sdc_act::forward3(L.net, x, lane_k, lg);
#endif
    }
    SDC_CK(0)
    // one uniform per (env, episode step, agent): ONE Philox block per (env, episode step), keyed on the GLOBAL env index like
    // the resets, its words x, y, z for agent_ls, agent_dc, agent_bat
    float u3[3] = {0.0f, 0.0f, 0.0f};
    if (sample) {
      const Philox4 r = philox4x32_10((unsigned)rel_now, (unsigned)(S.env_base + env_k + hk), ep_key, 0xAC70u, (unsigned)S.seed,
                                      (unsigned)(S.seed >> 32));
      u3[0] = (float)(r.x >> 8) * (1.0f / 16777216.0f);
      u3[1] = (float)(r.y >> 8) * (1.0f / 16777216.0f);
      u3[2] = (float)(r.z >> 8) * (1.0f / 16777216.0f);
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float u = u3[a];
      const float l0 = lg[a][0], l1 = lg[a][1], l2 = lg[a][2];
      act[a] = sdc_act::pick_action(l0, l1, l2, sample != 0, u);
      if (logits_out && lk < SDC_ACT_OUT)
        logits_out[(((size_t)k * N + (size_t)(env_k + hk)) * 3 + a) * 3 + lk] = lk == 0 ? l0 : (lk == 1 ? l1 : l2);
    }
    if (lk == 0) {
      int32_t* ao = actions_out + ((size_t)k * N + (size_t)(env_k + hk)) * 3;
      ao[0] = act[0];
      ao[1] = act[1];
      ao[2] = act[2];
    }
    SDC_CK(1)
    // ---- the env step on those actions ------------------------------------------------------------------------------------
    pair_step<true, true>(S, sh, env_k, lane_k, rel_now, nullptr, obs + (size_t)k * N * SDC_OBS_OUT,
                          share_obs + (size_t)k * N * SDC_SHARE_OBS_DIM, done + (size_t)k * N, info + (size_t)k * N * SDC_INFO_DIM,
                          k == K - 1 ? final_obs : nullptr, rew + (size_t)k * N * 3, nullptr, S.step_no + k, false, L.ktab, k == 0,
                          act[0], act[1], act[2]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    wave_sync();
    SDC_CK(2)
  }
#ifdef SDC_ACTOR_CLOCK
  if (lane == 0)
    for (int i = 0; i < 3; i++) info[((size_t)(K - 1) * N + (size_t)env0) * SDC_INFO_DIM + 38 + i] = (float)ck[i];
#endif
  // the observations the NEXT launch's first actions are chosen from (unless the episode ended: the host then copies the
  // reset observations in)
  if (obs_latch) {
#pragma unroll
    for (int q = 0; q < (EPW * SDC_OBS_OUT + SDC_WAVE - 1) / SDC_WAVE; q++) {
      const int idx = q * SDC_WAVE + lane;
      if (idx < EPW * SDC_OBS_OUT) {
        const int e = idx >= SDC_OBS_OUT ? 1 : 0, j = idx - e * SDC_OBS_OUT;
        obs_latch[(size_t)env0 * SDC_OBS_OUT + idx] = obs_padded_at(sh.pool[e], j);
      }
    }
  }
}
// (host side: the dynamic LDS the closed-loop kernel is launched with)
size_t sdc_rollout_actor_lds_bytes() { return sizeof(ActorLds); }

// The closed loop with FOUR envs per wavefront (large batches): workgroup = 8 wavefronts = 32 envs sharing the LDS copy of
// the networks; every MFMA row carries an env (forward3_quad).
struct ActorQuadLds {
  QuadShared shs[SDC_ACTOR_WPB];
  double ktab[SDC_K_LDS];
  SdcActorDev net[3];
};
static_assert(sizeof(ActorQuadLds) <= 160 * 1024, "one workgroup per CU");
static_assert(offsetof(ActorQuadLds, net) % 16 == 0, "the weights are read as ds_read_b128");
extern "C" __global__ __launch_bounds__(SDC_WAVE * SDC_ACTOR_WPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void
sdc_rollout_actor_quad_kernel(SdcDev S, const int K, const int rel_hint, const SdcActorDev* __restrict__ nets,
                              const float* __restrict__ obs_in, const int sample, float* __restrict__ obs,
                              float* __restrict__ share_obs, unsigned char* __restrict__ done, float* __restrict__ info,
                              float* __restrict__ final_obs, float* __restrict__ rew, int32_t* __restrict__ actions_out,
                              float* __restrict__ logits_out, float* __restrict__ obs_latch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  ActorQuadLds& L = *reinterpret_cast<ActorQuadLds*>(lds_raw);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / SDC_WAVE));
  const int lane = threadIdx.x % SDC_WAVE;
  {
    const uint4* src = reinterpret_cast<const uint4*>(nets);
    uint4* dst = reinterpret_cast<uint4*>(L.net);
    for (int i = (int)threadIdx.x; i < (int)(3 * sizeof(SdcActorDev) / 16); i += SDC_WAVE * SDC_ACTOR_WPB) dst[i] = src[i];
  }
  __syncthreads();
  const int env0 = (first_pair_of_block((int)blockIdx.x, (int)gridDim.x, SDC_ACTOR_WPB) + wave) * QE;
  if (env0 >= S.n_envs) return;
  QuadShared& sh = L.shs[wave];
  const size_t N = (size_t)S.n_envs;
  {
    // the observation pools of the four envs from the latest observations (see sdc_rollout_actor_kernel): two entries per lane
    const int r = lane >> 4, l = lane & (QL - 1);
    const float* o = obs_in + (size_t)(env0 + r) * SDC_OBS_OUT;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int j = 2 * l + t;
      if (j < SDC_POOL_DIM)
        sh.pool[r][j] = j < SDC_OBS_PAD ? o[j] : (j == SDC_P_WNEXT ? o[SDC_OBS_PAD + 11] : (j == SDC_P_NTNEXT ? o[SDC_OBS_PAD + 13] : o[2 * SDC_OBS_PAD + 12]));
    }
  }
  const unsigned ep_key = S.rec[(size_t)(env0 + (lane >> 4)) * SDC_REC_DWORDS + R_EPISODE];   // (see sdc_rollout_actor_kernel)
  wave_sync();
#pragma unroll 1
  for (int k = 0; k < K; k++) {
    int env_k = env0;
    asm volatile("" : "+s"(env_k));
    const int lane_k = lane_fresh();   // (recomputed every step: two instructions instead of a register held across the loop)
    const int row = lane_k >> 4, lk = lane_k & (QL - 1);
    const int half = lane_k >> 5, kk = lane_k & 31;
    const int rel_now = rel_hint + k;
    // ---- the three actors on the current observations of the four envs ---------------------------------------------------
    float lg[3][3];
    {
      float x[3][2];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const int pi = obs_pool_index(a * SDC_OBS_PAD + (kk < SDC_ACT_IN ? kk : 0));
        const float v0 = sh.pool[2 * half][pi < 0 ? 0 : pi], v1 = sh.pool[2 * half + 1][pi < 0 ? 0 : pi];
        x[a][0] = (kk < SDC_ACT_IN && pi >= 0) ? v0 : 0.0f;
        x[a][1] = (kk < SDC_ACT_IN && pi >= 0) ? v1 : 0.0f;
      }
      sdc_act::forward3_quad(L.net, x, lane_k, lg);
    }
    float u3[3] = {0.0f, 0.0f, 0.0f};
    if (sample) {
      const Philox4 r = philox4x32_10((unsigned)rel_now, (unsigned)(S.env_base + env_k + row), ep_key, 0xAC70u, (unsigned)S.seed,
                                      (unsigned)(S.seed >> 32));
      u3[0] = (float)(r.x >> 8) * (1.0f / 16777216.0f);
      u3[1] = (float)(r.y >> 8) * (1.0f / 16777216.0f);
      u3[2] = (float)(r.z >> 8) * (1.0f / 16777216.0f);
    }
    int act[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float l0 = lg[a][0], l1 = lg[a][1], l2 = lg[a][2];
      act[a] = sdc_act::pick_action(l0, l1, l2, sample != 0, u3[a]);
      if (logits_out && lk < SDC_ACT_OUT)
        logits_out[(((size_t)k * N + (size_t)(env_k + row)) * 3 + a) * 3 + lk] = lk == 0 ? l0 : (lk == 1 ? l1 : l2);
    }
    if (lk == 0) {
      int32_t* ao = actions_out + ((size_t)k * N + (size_t)(env_k + row)) * 3;
      ao[0] = act[0];
      ao[1] = act[1];
      ao[2] = act[2];
    }
    quad_step<true>(S, sh, env_k, lane_k, rel_now, nullptr, obs + (size_t)k * N * SDC_OBS_OUT,
                    share_obs + (size_t)k * N * SDC_SHARE_OBS_DIM, done + (size_t)k * N, info + (size_t)k * N * SDC_INFO_DIM,
                    k == K - 1 ? final_obs : nullptr, rew + (size_t)k * N * 3, S.step_no + k, false, L.ktab, k == 0, act[0], act[1],
                    act[2]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    wave_sync();
  }
  if (obs_latch) {
#pragma unroll
    for (int q = 0; q < (QE * SDC_OBS_OUT + SDC_WAVE - 1) / SDC_WAVE; q++) {
      const int idx = q * SDC_WAVE + lane;
      if (idx < QE * SDC_OBS_OUT) {
        const int e = (idx >= SDC_OBS_OUT ? 1 : 0) + (idx >= 2 * SDC_OBS_OUT ? 1 : 0) + (idx >= 3 * SDC_OBS_OUT ? 1 : 0);
        obs_latch[(size_t)env0 * SDC_OBS_OUT + idx] = obs_padded_at(sh.pool[e], idx - e * SDC_OBS_OUT);
      }
    }
  }
}
size_t sdc_rollout_actor_quad_lds_bytes() { return sizeof(ActorQuadLds); }
