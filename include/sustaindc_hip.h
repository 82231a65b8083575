/*
 * sustaindc_hip.h -- C-ABI of the MI355X-native vectorised SustainDC step.
 *
 * One handle = N environment instances resident on ONE GPU.  All obs / action / reward / info
 * buffers are device memory owned by the caller (PyTorch-ROCm tensors in the Python host); the
 * library borrows the raw pointers for the duration of a call and owns only its internal
 * struct-of-arrays state, trace tables and history rings.  Launches are asynchronous on the
 * caller's HIP stream.  No torch types, no C++ types: plain pointers and sizes.
 *
 * The reference (HewlettPackard/dc-rl) is pure Python and has no FFI for this path; each entry
 * point below names the reference interface it replaces (file:line under /root/reference).
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Return value: 0 on success, negative on error; message via sdc_last_error().
 * Threading: one host thread per handle; one handle per GPU.
 */
#ifndef SUSTAINDC_HIP_H
#define SUSTAINDC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDC_MAX_RACKS 64
#define SDC_N_AGENTS 3
#define SDC_OBS_PAD 26       /* per-agent obs padded to 26 (harl/envs/sustaindc/harlsustaindc_env.py:25-26) */
#define SDC_SHARE_OBS_DIM 29 /* harlsustaindc_env.py:78-80 */
#define SDC_INFO_DIM 44
#define SDC_TABLE_LEN 35040  /* 365 d x 96 steps (utils/managers.py:184) */

/* info[N][SDC_INFO_DIM] columns; names are the reference's info keys
 * (envs/carbon_ls.py:291-308, envs/dc_gym.py:213-229, envs/bat_env_fwd_view.py:111-122,
 *  sustaindc_env.py:676-683) */
enum sdc_info_col {
  SDC_INFO_LS_ORIGINAL_WORKLOAD = 0,
  SDC_INFO_LS_SHIFTED_WORKLOAD,
  SDC_INFO_LS_TASKS_IN_QUEUE,
  SDC_INFO_LS_NORM_TASKS_IN_QUEUE,
  SDC_INFO_LS_TASKS_DROPPED,
  SDC_INFO_LS_TASKS_PROCESSED,
  SDC_INFO_LS_OLDEST_TASK_AGE,
  SDC_INFO_LS_AVERAGE_TASK_AGE,
  SDC_INFO_LS_OVERDUE_PENALTY,
  SDC_INFO_LS_COMPUTED_TASKS,
  SDC_INFO_LS_CURRENT_HOUR,
  SDC_INFO_LS_AGE_HIST0, SDC_INFO_LS_AGE_HIST1, SDC_INFO_LS_AGE_HIST2, SDC_INFO_LS_AGE_HIST3, SDC_INFO_LS_AGE_HIST4,
  SDC_INFO_DC_ITE_TOTAL_POWER_KW,
  SDC_INFO_DC_CT_TOTAL_POWER_KW,
  SDC_INFO_DC_COMPRESSOR_TOTAL_POWER_KW,
  SDC_INFO_DC_HVAC_TOTAL_POWER_KW,
  SDC_INFO_DC_TOTAL_POWER_KW,
  SDC_INFO_DC_CRAC_SETPOINT_DELTA,
  SDC_INFO_DC_CRAC_SETPOINT,
  SDC_INFO_DC_CPU_WORKLOAD_FRACTION,
  SDC_INFO_DC_INT_TEMPERATURE,
  SDC_INFO_DC_EXTERIOR_AMBIENT_TEMP,
  SDC_INFO_DC_WATER_USAGE,
  SDC_INFO_BAT_ACTION,
  SDC_INFO_BAT_SOC,
  SDC_INFO_BAT_CO2_FOOTPRINT,
  SDC_INFO_BAT_AVG_CI,
  SDC_INFO_BAT_TOTAL_ENERGY_WITHOUT_BATTERY_KWH,
  SDC_INFO_BAT_TOTAL_ENERGY_WITH_BATTERY_KWH,
  SDC_INFO_NORM_CI,
  SDC_INFO_OUTSIDE_TEMP,
  SDC_INFO_DAY,
  SDC_INFO_HOUR,
  SDC_INFO_FAULT,    /* bit mask, see SDC_FAULT_* (the reference raises / asserts instead) */
  SDC_INFO_ENERGY_Z, /* normalize_energy() output shared by the three rewards */
  SDC_INFO_RESERVED, /* diagnostic: how this step's reward normalisation was served: 0 incremental state only (no
                        history read), 1 a rank window was re-centred over the history inline, 2 incremental state
                        only and a window re-centred by a spare wavefront of the previous launch was taken over,
                        3 the state was rebuilt from the history, 5 a clip bound had left its window and the bounds' side
                        (tail sums, that window) was redone from the history (and, only with debug_flags bit 3: 4 incremental
                        state only and a re-centring request was filed).  Scheduling-dependent (which requests find
                        a free slot), unlike every other output */
  /* running return of the current episode INCLUDING this step (== the episode return on the done step);
   * feeds the return statistics the runners log (harl/common/base_logger.py:75-88) without host sums */
  SDC_INFO_EP_RETURN_LS,
  SDC_INFO_EP_RETURN_DC,
  SDC_INFO_EP_RETURN_BAT,
  SDC_INFO_EPISODE_STEP /* steps taken in the current episode, including this one */
};

#define SDC_FAULT_OUTLET_DELTA 1u  /* envs/datacenter.py:295-300 raises */
#define SDC_FAULT_CPU_LOAD 2u      /* envs/dc_gym.py:288-290 asserts */
#define SDC_FAULT_BAT_DISCHARGE 4u /* envs/bat_env_fwd_view.py:237 asserts */
#define SDC_FAULT_WORKLOAD 8u      /* envs/carbon_ls.py:333-336 raises */
#define SDC_FAULT_TABLE_RANGE 16u  /* cursor would leave the year table (reference: IndexError) */
#define SDC_FAULT_ACTION 64u       /* agent_dc / agent_bat action outside {0,1,2} (the reference's action_mapping /
                                      _action_to_direction lookups raise KeyError: envs/dc_gym.py:160, bat_env_fwd_view.py:99);
                                      the step treats it as "no change" / "idle".  agent_ls: any other value is "do
                                      nothing" in the reference too (envs/carbon_ls.py:266), flagged all the same */
#define SDC_FAULT_ORDER_STAT 32u   /* debug_flags bit 0: the incremental reward state disagreed with the exact recomputation */

typedef struct sdc_handle sdc_handle;

/* replaces: EnvConfig / SustainDC.__init__ wiring (sustaindc_env.py:34-160) for N envs */
typedef struct {
  int32_t n_envs;
  int32_t device;          /* HIP device ordinal */
  int32_t episode_steps;   /* days_per_episode * 96 (utils/managers.py:113) */
  int32_t hist_cap;        /* 10000 (utils/reward_creator.py:5) */
  int32_t queue_max_len;   /* 1000 (sustaindc_env.py:149) */
  int32_t n_locations;     /* number of trace-table sets */
  int32_t n_dc_configs;    /* number of data-centre parameter sets */
  int32_t auto_reset;      /* 1: reset finished envs inside sdc_step (harl/envs/env_wrappers.py:176-190) */
  uint64_t seed;           /* counter-based RNG seed for device-side resets */
  double weather_noise_std;   /* 0.75 (utils/managers.py:504) ; 0 disables the noise */
  double weather_noise_weight;/* 0.02 (utils/managers.py:504) */
  int32_t max_roll_days;   /* 14: roll in [0, 14) days (utils/managers.py:601) */
  int32_t debug_flags;     /* bit 0: VERIFY MODE -- after every step check the incremental reward state (the four rank
                              windows, running sums) and the reported z-score against an exact bisection
                              and a direct fp64 pass over each env's history (slow; a mismatch sets
                              SDC_FAULT_ORDER_STAT).  Bits 1, 3, 4: diagnostics in info[reserved] / info[40..43]
                              (why a rebuild happened; per-wavefront phase durations; absolute wavefront start /
                              end stamps) -- measurement only, they overwrite the episode-return columns.
                              Bit 6 (64): TEST HOOK -- sdc_create reads the environment variable SDC_TEST_STEP_NO and
                              starts the launch counter there (tests of the counter's wrap); ignored otherwise.
                              Bit 7 (128): always launch the GENERAL step / rollout kernels, never the ones specialised
                              for the common case (same results to the bit: tests compare the two).
                              Bits 9 / 10 (512 / 1024): the common-case kernels with two / four envs per wavefront
                              whatever the batch size (by default four when the batch is a multiple of four and
                              large: single steps above 5632 envs, the multi-step entry points above 4096; same
                              results to the bit).
                              Bit 13 (8192): TEST HOOK -- every 61st (env + launch) redoes its clip bounds' side from the
                              history as if a bound had left its window (a path ~4e-8 of the env-steps take by themselves) */
  int32_t reward_method[3]; /* reward function per agent slot (ls, dc, bat), utils/reward_creator.py:322-334:
                               SDC_REWARD_DEFAULT the slot's own default_*_reward, SDC_REWARD_FOOTPRINT
                               default_dc_reward = default_bat_reward, SDC_REWARD_CUSTOM custom_agent_reward (0),
                               SDC_REWARD_TOU, SDC_REWARD_ENERGY_EFFICIENCY, SDC_REWARD_PUE, SDC_REWARD_WATER.
                               As in the reference only default_ls_reward appends to the energy history
                               (reward_creator.py:63): the history grows iff reward_method[0] == SDC_REWARD_DEFAULT */
  int32_t env_index_base;  /* global index of this batch's env 0 when the batch is one shard of a multi-GPU job: the
                              counter-based RNG of device-side resets is keyed on (seed, env_index_base + env, episode),
                              so a job draws the same start day / hour / roll / weather noise for global env i whatever
                              the number of GPUs it is sharded over (harl/utils/envs_tools.py:56-65 keys months and
                              seeds on the global rank the same way) */
  int32_t policy[3];       /* who chooses each agent slot's action (ls, dc, bat): SDC_POLICY_EXTERNAL the caller's
                              actions array; SDC_POLICY_DO_NOTHING the reference's base agents (utils/base_agents.py:
                              ls 1, dc 1, bat 2 -- what SustainDC plays for agents that are not trained,
                              sustaindc_env.py:172-191, :623-655); SDC_POLICY_RBC (bat slot) RBCBatteryAgent
                              (utils/rbc_agents.py:21-47, look_ahead 3, smooth_window 1); SDC_POLICY_TRIM_AND_RESPOND
                              (dc slot) trim_and_respond_ctrl (utils/trim_and_respond.py:8-38).  The policies run inside
                              the step kernel, so sdc_rollout can run closed-loop episodes without an action array. */
  int32_t reserved2;
  double trim_and_respond_limit; /* TandR_monitor_limit (27 in the reference), compared with the room temperature
                                    (dc_int_temperature) the previous step reported */
} sdc_config;

enum sdc_policy { SDC_POLICY_EXTERNAL = 0, SDC_POLICY_DO_NOTHING = 1, SDC_POLICY_RBC = 2, SDC_POLICY_TRIM_AND_RESPOND = 3 };

enum sdc_reward_method {
  SDC_REWARD_DEFAULT = 0,
  SDC_REWARD_FOOTPRINT = 1,
  SDC_REWARD_CUSTOM = 2,
  SDC_REWARD_TOU = 3,               /* DEVIATION: the reference indexes its price table with the float hour and raises
                                       KeyError off the full hour (reward_creator.py:191); here the hour is truncated */
  SDC_REWARD_ENERGY_EFFICIENCY = 4,
  SDC_REWARD_PUE = 5,
  SDC_REWARD_WATER = 6
};

/* replaces: DC_Config + Rack/CPU constants + sized HVAC values
 * (utils/dc_config_reader.py:39-145, envs/datacenter.py:31-135, utils/make_envs_pyenv.py:139-197) */
typedef struct {
  int32_t n_racks;
  int32_t reserved;
  double rack_n[SDC_MAX_RACKS];      /* CPUs per rack after the MAX_W_PER_RACK cap (datacenter.py:67-74) */
  double rack_full[SDC_MAX_RACKS];   /* full-load W per CPU */
  double rack_idle[SDC_MAX_RACKS];   /* idle W per CPU */
  double rack_supply[SDC_MAX_RACKS]; /* supply approach temperature, unclamped */
  double rack_return[SDC_MAX_RACKS]; /* return approach temperature */
  double m_cpu, c_cpu, rs_cpu;       /* datacenter.py:31-39 */
  double m_fan, c_fan, rs_fan;       /* datacenter.py:41-49 */
  double itfan_ref_p, itfan_ref_v_ratio, it_fan_full_load_v;
  double c_air, rho_air, crac_supply_pu;
  double ct_fan_ref_p, ctafr;        /* SIZED (make_envs_pyenv.py:159-161) */
  double min_temp, max_temp;         /* CRAC set-point clamp (make_envs_pyenv.py:125-126) */
  double init_setpoint;              /* 18 (make_envs_pyenv.py:124) */
  double bat_capacity_mwh;           /* sized battery capacity (sustaindc_env.py:152) */
} sdc_dc_params;

/* replaces: the (day, hour, roll, noise) draws of SustainDC.reset / Weather_Manager.reset
 * (sustaindc_env.py:454-461, utils/managers.py:581-628) when the caller wants to inject them
 * (parity tests).  All pointers are HOST memory, indexed by env. */
typedef struct {
  const int32_t* day;   /* [N] */
  const int32_t* hour;  /* [N] 0..23 */
  const double* ci_min; /* [N] min of C over [cursor, cursor+2880) (managers.py:435-437) */
  const double* ci_max;
  const double* t_min;  /* [N] same for the noised/rolled/clipped temperature (managers.py:606-608) */
  const double* t_max;
  const double* t_win;  /* [N][weather_window_len]: T[cursor0 + k] after noise+roll+clip */
  const double* wb_win; /* [N][weather_window_len]: wet bulb likewise */
  /* Alternative injection one level earlier (noise != NULL; ci_min .. wb_win are then ignored and may be NULL): the
   * draws of Weather_Manager.reset themselves -- the year's coherent-noise array as CoherentNoise.generate returned it
   * (managers.py:35-48) and the roll in days (:601) -- next to day / hour.  The device then does what it does after
   * its own draws: add the noise to the location's T / WB tables, roll, clip to [0, 45], take the 30-day min / max
   * from the cursor (managers.py:596-613) and the CI bounds (:435-437). */
  const double* noise;     /* [N][SDC_TABLE_LEN] or NULL */
  const int32_t* roll_days; /* [N], with noise */
} sdc_reset_override;

const char* sdc_last_error(void);
/* ABI version of this header: bumped whenever a struct layout or an entry point's arguments change.  sdc_version()
 * returns the value the library was BUILT with; a caller compares the two before anything else (dc_rl_amd/_lib.py
 * does) -- the .so is shipped out of band, so a stale one must fail loudly, not corrupt silently.
 *   100  round 1        300  sdc_config: env_index_base, policy[3], trim_and_respond_limit; sdc_reset_override: noise,
 *                            roll_days; sdc_rollout: actions_out; debug_flags bit 6
 *   310  sdc_set_actor, sdc_rollout_actor (closed loop with the actor networks inside the kernel); debug_flags bit 7
 *        (debug_flags bits 9 / 10 came later without a bump: no layout or argument list changed) */
#define SDC_ABI_VERSION 311
int sdc_version(void);

int sdc_create(const sdc_config* cfg, sdc_handle** out);
int sdc_destroy(sdc_handle* h);

/* replaces: SustainDC.seed (sustaindc_env.py:241-251): re-key the counter-based RNG of device-side resets */
int sdc_set_seed(sdc_handle* h, uint64_t seed);

/* episode_steps + 18: samples of per-env weather the step can touch */
int sdc_weather_window_len(const sdc_handle* h);

/* replaces: Workload_Manager / CI_Manager / Weather_Manager table construction
 * (utils/managers.py:152-197, :318-388, :488-569).  Host arrays of length n (= SDC_TABLE_LEN):
 * W = cpu_smooth after scale_array + 16-tap smoothing (managers.py:268-271), C = carbon_smooth,
 * T / WB = interpolated dry / wet bulb BEFORE noise. */
int sdc_set_tables(sdc_handle* h, int loc_id, const double* W, const double* C, const double* T, const double* WB,
                   int n);

int sdc_set_dc_params(sdc_handle* h, int cfg_id, const sdc_dc_params* p);

/* per-env assignment (host arrays [N]): trace set, DC parameter set, and the inclusive range the
 * random start day is drawn from (sustaindc_env.py:198, :454) */
int sdc_assign_envs(sdc_handle* h, const int32_t* loc_id, const int32_t* cfg_id, const int32_t* day_lo,
                    const int32_t* day_hi);

/* replaces: SustainDC.reset (sustaindc_env.py:436-531) / ShareVecEnv.reset (env_wrappers.py:275-280).
 * mask_host: NULL = all envs, else [N] bytes (host).  ovr: NULL = draw on device.
 * obs [N][3][26] f32, share_obs [N][29] f32 (device; may be NULL).  A masked reset writes the masked envs' rows only.
 * Closed loop (sdc_set_actor): the library keeps its own copy of the latest observations; a reset with obs == NULL
 * invalidates it (sdc_rollout_actor then refuses until a reset / step has delivered observations), a masked reset
 * takes over the masked rows only (the other rows of the caller's buffer are not read). */
int sdc_reset(sdc_handle* h, const uint8_t* mask_host, const sdc_reset_override* ovr, float* obs, float* share_obs,
              void* stream);

/* replaces: SustainDC.step (sustaindc_env.py:533-621) + HARL adaptation + auto-reset
 * (harlsustaindc_env.py:106-131, env_wrappers.py:168-192) for all N envs.
 * actions [N][3] int32 (ls, dc, bat) in {0,1,2}; obs [N][3][26]; share_obs [N][29]; rew [N][3];
 * done [N] u8; info [N][SDC_INFO_DIM] f32; final_obs [N][3][26] receives the pre-reset observation of
 * envs that finished ("original_obs"); info / final_obs / share_obs may be NULL. */
int sdc_step(sdc_handle* h, const int32_t* actions, float* obs, float* share_obs, float* rew, uint8_t* done,
             float* info, float* final_obs, void* stream);

/* n_steps env-steps in ONE launch for action sequences known up front (scripted / rule-based policies -- the
 * reference's utils/rbc_agents.py, utils/base_agents.py --, open-loop evaluation): every env advances n_steps times
 * without waiting for the others.  actions [n_steps][N][3]; obs [n_steps][N][3][26], share_obs [n_steps][N][29],
 * rew [n_steps][N][3], done [n_steps][N], info [n_steps][N][SDC_INFO_DIM] receive every step's outputs (share_obs /
 * info / final_obs may be NULL).  n_steps must not exceed sdc_steps_to_episode_end(); if it reaches the episode's
 * end and auto_reset is on, the finished envs are reset as in sdc_step (the last step's obs slice holds the reset
 * observation, final_obs [N][3][26] the pre-reset one).  Same results as n_steps calls of sdc_step.
 * Agent slots with a built-in policy (sdc_config.policy) ignore `actions`, which may be NULL when all three have one;
 * actions_out [n_steps][N][3] (device, or NULL) receives the actions every step applied. */
int sdc_rollout(sdc_handle* h, int n_steps, const int32_t* actions, float* obs, float* share_obs, float* rew,
                uint8_t* done, float* info, float* final_obs, int32_t* actions_out, void* stream);
/* CLOSED LOOP.  replaces: the actor forward pass of the reference's rollout loop (harl/runners/on_policy_base_runner.py
 * collect -> harl/algorithms/actors/on_policy_base.py get_actions -> StochasticPolicy.forward,
 * harl/models/policy_models/stochastic_policy.py:11-60) between two env steps: one agent's network
 *     LayerNorm(26) -> Linear(26, 64) -> act -> LayerNorm(64) -> Linear(64, 64) -> act -> LayerNorm(64) -> Linear(64, 3)
 * (harl/models/base/mlp.py:8-72, act.py:45-84; hidden_sizes [64, 64], happo.yaml:58) with the weights in torch's layout
 * ([out][in], row-major), fp32.  sdc_set_actor copies them to the device (host pointers). */
typedef struct {
  float ln0_gamma[26], ln0_beta[26];                 /* feature_norm (use_feature_normalization) */
  float w1[64 * 26], b1[64], ln1_gamma[64], ln1_beta[64];
  float w2[64 * 64], b2[64], ln2_gamma[64], ln2_beta[64];
  float w3[3 * 64], b3[3];                           /* act.action_out.linear */
  int32_t use_feature_normalization;                 /* 1: LayerNorm over the 26 inputs first */
  int32_t activation;                                /* 0 tanh, 1 relu */
} sdc_actor_params;
int sdc_set_actor(sdc_handle* h, int agent_slot, const sdc_actor_params* p);
/* n_steps env-steps in ONE launch, every step's three actions chosen INSIDE the kernel by the three actors from the
 * step's own observations (the first from the observations the last sdc_reset / sdc_step / sdc_rollout* call returned,
 * of which the library keeps a copy once an actor is set): observation -> actor -> action -> step without a launch or a
 * host round trip per step.  sample = 0: the distributions' mode (deterministic = True in the reference), 1: a draw
 * (counter-based RNG keyed on seed, global env index, episode step, agent).  Outputs as sdc_rollout (all required but
 * final_obs); actions_out [n_steps][N][3] receives the actions, logits_out [n_steps][N][3][3] (may be NULL) the actors'
 * logits.  Only for the common case the specialised kernels serve (lock-step batch with feature rows, one data-centre
 * config of <= 32 racks, default rewards, an even number of envs) and three actors with the same activation; anything
 * else is refused. */
int sdc_rollout_actor(sdc_handle* h, int n_steps, int sample, float* obs, float* share_obs, float* rew, uint8_t* done,
                      float* info, float* final_obs, int32_t* actions_out, float* logits_out, void* stream);
/* steps until the first env finishes its episode (0: a reset is due) */
int sdc_steps_to_episode_end(const sdc_handle* h);
/* which envs finished their episode in the last sdc_step / sdc_rollout call -- the `done` output, but from the host's
 * mirror of the step counters (episodes have a fixed length), so a caller that keeps everything on the device learns
 * about episode boundaries (harl/envs/env_wrappers.py:176-190: "original_obs" bookkeeping) without a device->host
 * read.  Returns the number of finished envs; done_host [N] (host, may be NULL) is filled only when it is > 0. */
int sdc_last_done(const sdc_handle* h, uint8_t* done_host);
/* name of the kernel the last sdc_step / sdc_rollout launched ("" before the first): the host picks by batch size and configuration
 * between the general kernel, the common-case kernels with two / four envs per wavefront and the lane-per-env kernel of the
 * largest batches (sdc_capi.hip fast_case / quad_case / wide_case) -- all give the same results; tests and benchmarks name
 * what they measured with this. */
const char* sdc_last_step_kernel(const sdc_handle* h);

/* parity injection + env checkpoint: copy one named state field to / from HOST memory, dense per env.
 * int32[N]:  cursor t_rel day hourq q_popped q_cum q_cumT q_head q_cum_hm1 q_cumT_hm1 last_delta consecutive
 *            scale hist_len hist_pos episode fault loc_id cfg_id day_lo day_hi hist_n
 * double[N]: stpt bat_load ci_min ci_den t_min t_den hist_ref
 * record (uint32[N][64], the raw 256-byte state records);  header (uint32[N][64], step hand-off + reward state);
 * qwin (uint32[N][64][4], the reward state's rank windows: per lane the keys of {Q1, Q3, upper bound, lower bound});
 * ep_return (double[N][3]);
 * hist (float[N][hist_stride], energy minus hist_ref, NaN = empty slot: every slot >= hist_len must be NaN);
 * t_win wb_win (double[N][weather_window_len]);  qtab (uint32[N][queue_stride][2]). */
int sdc_get_state(sdc_handle* h, const char* field, void* host_buf, size_t bytes);
int sdc_set_state(sdc_handle* h, const char* field, const void* host_buf, size_t bytes);
int sdc_hist_stride(const sdc_handle* h);
int sdc_queue_stride(const sdc_handle* h);

/* Per-kernel timing (measurement only; off by default).  enable = k > 0 samples every k-th sdc_step, 0 switches it
 * off.  In a sampled step one lane per workgroup of each kernel stamps the device's constant-rate wall clock at
 * entry and exit; sdc_profile_read synchronises the device and accumulates, per sampled launch,
 * max(exit) - min(entry) over the workgroups -- the launch's duration on the GPU, with no host-event overhead.
 *   out[0] = sdc_dynamics_kernel (the step kernel) total ms, out[1] = 0 (no separate reward kernel any more),
 *   out[2] = sdc_reset_kernel (auto-reset) total ms, out[3] = steps sampled, out[4] = auto-resets sampled. */
int sdc_profile_enable(sdc_handle* h, int enable);
int sdc_profile_read(sdc_handle* h, double* out5, int reset);

#ifdef __cplusplus
}
#endif
#endif
