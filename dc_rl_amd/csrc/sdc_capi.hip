// sdc_capi.hip -- host side of the C-ABI declared in include/sustaindc_hip.h.
//
// Owns the device-resident state of N environments on one GPU and launches the kernels on the caller's stream:
// sdc_dynamics_kernel (one launch = one env-step of all N envs, rewards included), sdc_reset_kernel at episode
// boundaries, sdc_reward_verify_kernel only in verify mode.  No CPU fallback: every entry point fails with an
// error code when HIP reports one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <memory>
#include <vector>

#include "sdc_device.hpp"
#include "sdc_actor.hpp"

extern "C" __global__ void sdc_dynamics_kernel(SdcDev S, int rel_hint, const int32_t* actions, float* obs, float* share_obs,
                                               unsigned char* done, float* info, float* final_obs, float* rew);
extern "C" __global__ void sdc_dynamics_fast_kernel(SdcDev S, int rel_hint, const int32_t* actions, float* obs, float* share_obs,
                                                    unsigned char* done, float* info, float* final_obs, float* rew);
extern "C" __global__ void sdc_dynamics_quad_kernel(SdcDev S, int rel_hint, const int32_t* actions, float* obs, float* share_obs,
                                                    unsigned char* done, float* info, float* final_obs, float* rew);
extern "C" __global__ void sdc_dynamics_wide_kernel(SdcDev S, int rel_hint, const int32_t* actions, float* obs, float* share_obs,
                                                    unsigned char* done, float* info, float* final_obs, float* rew);
extern "C" __global__ void sdc_dynamics_wide_gen_kernel(SdcDev S, int rel_hint, const int32_t* actions, float* obs, float* share_obs,
                                                        unsigned char* done, float* info, float* final_obs, float* rew);
extern "C" __global__ void sdc_rollout_quad_kernel(SdcDev S, int K, int rel_hint, const int32_t* actions, float* obs,
                                                   float* share_obs, unsigned char* done, float* info, float* final_obs, float* rew);
extern "C" __global__ void sdc_rollout_fast_kernel(SdcDev S, int K, int rel_hint, const int32_t* actions, float* obs,
                                                   float* share_obs, unsigned char* done, float* info, float* final_obs, float* rew);
extern "C" __global__ void sdc_rollout_actor_kernel(SdcDev S, int K, int rel_hint, const SdcActorDev* nets, const float* obs_in,
                                                    int sample, float* obs, float* share_obs, unsigned char* done, float* info,
                                                    float* final_obs, float* rew, int32_t* actions_out, float* logits_out,
                                                    float* obs_latch);
size_t sdc_rollout_actor_lds_bytes();
extern "C" __global__ void sdc_rollout_actor_quad_kernel(SdcDev S, int K, int rel_hint, const SdcActorDev* nets, const float* obs_in,
                                                         int sample, float* obs, float* share_obs, unsigned char* done, float* info,
                                                         float* final_obs, float* rew, int32_t* actions_out, float* logits_out,
                                                         float* obs_latch);
size_t sdc_rollout_actor_quad_lds_bytes();
extern "C" __global__ void sdc_reward_verify_kernel(SdcDev S, float* info);
extern "C" __global__ void sdc_features_kernel(SdcDev S, int use_sma);
extern "C" __global__ void sdc_rollout_kernel(SdcDev S, int K, int rel_hint, const int32_t* actions, float* obs, float* share_obs,
                                              unsigned char* done, float* info, float* final_obs, float* rew);

extern "C" __global__ void sdc_reset_kernel(SdcDev S, int use_override, const int* ovr_day, const int* ovr_hour,
                                            const double* ovr_ci_min, const double* ovr_ci_max, const double* ovr_t_min,
                                            const double* ovr_t_max, int only_done, float* obs, float* share_obs,
                                            const double* inj_noise, const int* inj_roll);

namespace {

thread_local std::string g_err;

int fail(const char* what, hipError_t e) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return -1;
}
int fail_msg(const std::string& m) {
  g_err = m;
  return -2;
}

#define HIP_TRY(expr)                          \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) return fail(#expr, _e); \
  } while (0)

struct Field {
  const char* name;
  void** ptr;      // plain per-env array (ptr != nullptr) ...
  size_t elem;     // ... of `elem` bytes per env,
  int rec_idx;     // or a field of a strided per-env record: first dword,
  int rec_dwords;  // width in dwords,
  int in_hdr = 0;  // 0: the 256-byte state record, 1: the 256-byte header
};

}  // namespace

struct sdc_handle {
  sdc_config cfg;
  SdcDev d;
  int rel_hint = -1;   // the episode step all envs are at, if they are in lock-step (else -1)
  int step_no = 3;           // steps launched (stamps the deferred window re-centrings; starts above the stamps of zeroed memory)
  int device;
  std::vector<void*> allocs;
  std::vector<Field> fields;
  // override staging (device)
  int* ovr_day = nullptr;
  int* ovr_hour = nullptr;
  double* ovr_ci_min = nullptr;
  double* ovr_ci_max = nullptr;
  double* ovr_t_min = nullptr;
  double* ovr_t_max = nullptr;
  // host mirror for auto-reset scheduling: steps left until the earliest env finishes
  std::vector<int> host_t_rel;  // exact at the last sync point
  int pending = 0;              // steps launched since then (every env advances by one per step)
  int steps_to_terminal = 0;
  std::vector<unsigned char> feat_host;   // host mirror of R_FEAT_OK: the env's episode has valid observation feature rows
  int n_feat_host = 0;                    // how many envs have
  // closed loop (sdc_set_actor / sdc_rollout_actor): the three actor networks and the library's copy of the latest
  // observations (what the first actions of a launch are chosen from); allocated when the first actor is set
  SdcActorDev* actor_dev = nullptr;
  bool actor_set[3] = {false, false, false};
  int actor_activation[3] = {0, 0, 0};
  bool actor_lds_set = false;   // the closed-loop kernels' dynamic-LDS limit has been raised on this handle's device
  float* obs_latch = nullptr;
  bool latch_valid = false;
  int racks_cfg0 = 0;                     // racks of data-centre config 0 (the specialised kernels take <= 32: one pass)
  const char* last_step_kernel = "";      // sdc_last_step_kernel
  int rack_cls_cfg0 = 0;                  // ... and its rack classes (SdcRackClasses; 0: more than the lane-per-env kernel's tables hold)
  // several data-centre configs: host copies of the configs and of the assignment, from which every env's own copy of its
  // config's scalars is built (SdcDev::prm_env) -- the common-case kernels then serve the batch as they serve one config
  std::vector<SdcDcDev> dc_host;
  std::vector<int> cfg_host;
  std::vector<unsigned char> dc_set;
  double* prm_env_dev = nullptr;
  bool prm_env_ok = false;
  int racks_max = 0;
  // the lane-per-env kernel's general form (sdc_wide.hip GEN): one SdcWideCfg per config, when the batch's configs qualify
  SdcWideCfg* wcfg_dev = nullptr;
  bool wide_gen_ok = false;
  std::vector<unsigned char> last_done;   // which envs finished in the last sdc_step / sdc_rollout call (host mirror)
  int n_last_done = 0;
  bool tables_set = false, assigned = false, started = false;
  // optional per-kernel timing: the kernels stamp the device wall clock per workgroup into one slot per sampled step
  int prof = 0;       // sample every `prof`-th step (0 = off)
  long prof_tick = 0;
  unsigned long long* prof_buf = nullptr;  // [PROF_SLOTS][3][N][2]
  int prof_used = 0;
  std::vector<unsigned char> prof_has_reset;
  double wall_clock_khz = 100000.0;
  double acc_ms[5] = {0, 0, 0, 0, 0};      // dynamics, reward, reset, steps, resets
};

namespace {

constexpr int PROF_SLOTS = 256;
#ifndef SDC_STEP_WPB
#define SDC_STEP_WPB 4
#endif
constexpr int STEP_WPB = SDC_STEP_WPB;   // csrc/sdc_step.hip SDC_STEP_WPB: env pairs (wavefronts) per workgroup of the step kernel
int step_blocks(int n_envs) { return ((n_envs + 1) / 2 + STEP_WPB - 1) / STEP_WPB; }
// The step counter that stamps re-centring requests wraps at 3 * 2^22: a multiple of the 3 rotating request sets and of
// the 2^22 the header stamps are taken modulo, so set rotation and stamp ages stay continuous across the wrap (the one
// request in flight at the wrap misses its full-width result stamp and falls back to the inline sweep).
constexpr int STEP_WRAP = 3 << 22;
int next_step_no(int s, int by) {
  if (by == 0) return s > STEP_WRAP - 4096 ? s % 3 + 3 : s;     // a multi-step launch must not straddle the wrap
  s += by;
  return s >= STEP_WRAP ? s - STEP_WRAP : s;
}
bool all_policies(const sdc_handle* h) {
  return h->d.policy[0] != SDC_POLICY_EXTERNAL && h->d.policy[1] != SDC_POLICY_EXTERNAL && h->d.policy[2] != SDC_POLICY_EXTERNAL;
}

template <typename T>
int dev_alloc(sdc_handle* h, T** p, size_t count, bool zero = true) {
  void* q = nullptr;
  HIP_TRY(hipMalloc(&q, count * sizeof(T)));
  if (zero) HIP_TRY(hipMemset(q, 0, count * sizeof(T)));
  h->allocs.push_back(q);
  *p = reinterpret_cast<T*>(q);
  return 0;
}

// fold the steps launched since the last sync point into the per-env mirror
void sync_mirror(sdc_handle* h) {
  if (h->pending) {
    for (int e = 0; e < h->cfg.n_envs; e++) h->host_t_rel[e] += h->pending;
    h->pending = 0;
  }
}

// one field of every env's record (256-byte state record, or 256-byte header) <-> a dense host array
// the episode's observation feature rows of the envs a reset kernel has just reset (sdc_features.hip); episodes too long
// for the kernel's LDS windows go without (the step then computes the features itself)
void launch_features(sdc_handle* h, const SdcDev& d, hipStream_t st) {
  if (!d.feat) return;
  const size_t win = sizeof(double) * (size_t)(d.episode_steps + 25 + d.lw), sma = sizeof(double) * (size_t)(d.episode_steps + 22);
  const size_t tile = sizeof(float) * SDC_WAVE * (SDC_FEAT_ROW + 1), cap = 64 * 1024;
  // four wavefronts share an env's windows (and the window's moving averages, computed once) where four tiles fit beside them
  const int waves = win + sma + 4 * tile <= cap ? 4 : 1;
  const int use_sma = win + sma + waves * tile <= cap ? 1 : 0;
  hipLaunchKernelGGL(sdc_features_kernel, dim3(d.n_envs), dim3(SDC_WAVE * waves), win + (use_sma ? sma : 0) + waves * tile, st, d, use_sma);
  (void)h;
}

int rec_put(sdc_handle* h, int idx, int dwords, const void* host, int in_hdr = 0) {
  unsigned* base = in_hdr ? h->d.hdr : h->d.rec;
  const size_t pitch = sizeof(unsigned) * (in_hdr ? SDC_HDR_DWORDS : SDC_REC_DWORDS);
  HIP_TRY(hipMemcpy2D(base + idx, pitch, host, sizeof(unsigned) * dwords, sizeof(unsigned) * dwords,
                      (size_t)h->cfg.n_envs, hipMemcpyHostToDevice));
  return 0;
}
int rec_get(sdc_handle* h, int idx, int dwords, void* host, int in_hdr = 0) {
  const unsigned* base = in_hdr ? h->d.hdr : h->d.rec;
  const size_t pitch = sizeof(unsigned) * (in_hdr ? SDC_HDR_DWORDS : SDC_REC_DWORDS);
  HIP_TRY(hipMemcpy2D(host, sizeof(unsigned) * dwords, base + idx, pitch, sizeof(unsigned) * dwords,
                      (size_t)h->cfg.n_envs, hipMemcpyDeviceToHost));
  return 0;
}

// the episode's precomputed observation rows follow the traces, the env's location and its weather windows
int invalidate_features(sdc_handle* h) {
  std::vector<unsigned> z((size_t)h->cfg.n_envs, 0u);
  h->feat_host.assign((size_t)h->cfg.n_envs, 0);
  h->n_feat_host = 0;
  return rec_put(h, R_FEAT_OK, 1, z.data());
}
// the envs a reset has just given fresh feature rows (launch_features: every env the reset kernel has reset)
void note_features(sdc_handle* h, int e) {
  if (h->d.feat && !h->feat_host[e]) {
    h->feat_host[e] = 1;
    h->n_feat_host += 1;
  }
}
// THE COMMON CASE, for which the step / rollout kernels exist in a specialised form (sdc_step.hip, template FAST):
// every env in lock-step with valid feature rows, one data-centre config, the caller's actions on all three slots, the
// default reward functions, no diagnostics or profiling, an even number of envs, every output array present.
// debug_flags bit 0 (the verify kernel, a separate launch) and bit 6 (test hook of sdc_create) do not touch the step;
// bit 7 forces the general kernel (tests compare the two bit for bit).  The common case runs FOUR envs per wavefront
// (sdc_*_quad_kernel) when the batch is a multiple of four envs and large enough for that mapping to pay: a SIMD has to hold
// more than one such wavefront, or nothing overlaps its waits.  Measured (tools/step_scan.py, MI355X: 1024 SIMDs): single
// steps are faster with four envs per wavefront from ~6 700 envs on (6144: 16.3 us with two, 16.7 with four; 7168: 21.7 /
// 18.3); the multi-step kernels (sdc_rollout, sdc_rollout_actor), whose two-env form keeps two wavefronts per SIMD and
// needs a second round above 4096 envs, from any batch above 4096 (5120 envs: rollout 17.7 / 14.0 us per step, closed loop
// 28.2 / 19.7).  debug_flags bit 9 keeps two envs per wavefront, bit 10 picks four whatever the size (the tests compare
// all of them bit for bit).
#ifndef SDC_FAST_DEBUG
#define SDC_FAST_DEBUG 0
#endif
constexpr int FAST_DEBUG_FLAGS = SDC_FAST_DEBUG ? (8 | 16 | 32 | 256) : 0;   // (measurement builds: see sdc_step.hip)
#ifndef SDC_QUAD_MIN_ENVS_STEP
// (round 4: 5 632 envs = 704 env-pair workgroups = 2.75 dispatch rounds is the last size at which two envs per wavefront win --
// 12.6 against 13.1 us per step; 6 144 envs: 16.5 against 13.3, the third round full and the spare sweep wavefronts pushing 128 env
// wavefronts into a fourth)
#define SDC_QUAD_MIN_ENVS_STEP 5636
#endif
#ifndef SDC_QUAD_MIN_ENVS_LOOP
#define SDC_QUAD_MIN_ENVS_LOOP 4100
#endif
bool quad_case(const sdc_handle* h, const bool multi_step) {
  return (h->cfg.n_envs & 3) == 0 && h->d.n_cfg == 1 && (h->d.debug_flags & (512 | FAST_DEBUG_FLAGS)) == 0 &&
         (h->cfg.n_envs >= (multi_step ? SDC_QUAD_MIN_ENVS_LOOP : SDC_QUAD_MIN_ENVS_STEP) || (h->d.debug_flags & 1024));
}
int quad_blocks(int n_envs) { return (n_envs / 4 + STEP_WPB - 1) / STEP_WPB; }
// ONE LANE PER ENV (sdc_wide.hip): single steps of the largest lock-step batches -- a multiple of 64 envs, one config of <= 31
// racks, 16-byte aligned output rows (whole-line stores through the wavefront's staging block); debug_flags bit 11 forces it
// for any such batch, bit 12 keeps it off
#ifndef SDC_WIDE_MIN_ENVS
// (measured, us per step with the episode boundary inside, lane per env / four per wavefront, round 6 -- after the record's one-line
// layout and the kernel-argument touch: 6 144 envs 13.74 / 13.40, 7 168: 14.02 / 13.69, 7 680: 14.16 / 14.34, 8 192: 14.25 / 14.47,
// 8 704: 14.63 / 15.52, 12 288: 15.6 / 21.4, 16 384: 16.5 / 24.8, 32 768: 23.6 / 40.5, 65 536: 42.7 / 73.4; round 5's crossover was 9 216)
#define SDC_WIDE_MIN_ENVS 7680
#endif
#ifndef SDC_WIDE_ROLLOUT_MIN_ENVS
#define SDC_WIDE_ROLLOUT_MIN_ENVS 12288      // sdc_rollout: K single-step launches of the lane-per-env kernel from here (below: one K-step launch)
#endif
// the structural conditions of the lane-per-env kernel (either form): a multiple of 64 envs, the queue table's time-major mirror,
// whole-line stores through the workgroup's staging block (16-byte aligned output rows)
bool wide_shape(const sdc_handle* h, const float* obs, const float* share_obs, const float* info, const float* final_obs) {
  const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
  return (h->cfg.n_envs & 63) == 0 && h->d.qcum_t != nullptr && (h->d.debug_flags & (512 | 1024 | 4096 | FAST_DEBUG_FLAGS)) == 0 &&
         (h->cfg.n_envs >= SDC_WIDE_MIN_ENVS || (h->d.debug_flags & 2048)) && al16(obs) && al16(share_obs) && al16(info) && al16(final_obs);
}
// ... its common-case form (one config of <= 32 racks in <= 8 classes; fast_case holds as well: the caller checks both)
bool wide_case(const sdc_handle* h, const float* obs, const float* share_obs, const float* info, const float* final_obs) {
  return wide_shape(h, obs, share_obs, info, final_obs) && h->d.n_cfg == 1 && h->racks_cfg0 <= 32 && h->rack_cls_cfg0 > 0;
}
int wide_sweep_blocks(const sdc_handle* h) { return std::min(h->d.rq_max, 256) / 2; }     // (two wavefronts each, a request per wavefront)
// what every specialised kernel needs: all envs in lock-step with valid feature rows, every output array present, no profiling
bool lockstep_case(const sdc_handle* h, const float* share_obs, const float* info, bool timed) {
  const SdcDev& d = h->d;
  return h->rel_hint >= 0 && d.feat != nullptr && h->n_feat_host == h->cfg.n_envs && share_obs && info && !timed &&
         (h->cfg.n_envs & 1) == 0 && (d.debug_flags & ~(1 | 64 | 512 | 1024 | 2048 | 4096 | FAST_DEBUG_FLAGS)) == 0;
}
bool fast_case(const sdc_handle* h, const int32_t* actions, const float* share_obs, const float* info, bool timed) {
  const SdcDev& d = h->d;
  return lockstep_case(h, share_obs, info, timed) &&
         (d.n_cfg == 1 ? (h->racks_cfg0 > 0 && h->racks_cfg0 <= 32) : (h->prm_env_ok && h->racks_max <= 32)) && actions &&
         d.policy[0] == SDC_POLICY_EXTERNAL && d.policy[1] == SDC_POLICY_EXTERNAL && d.policy[2] == SDC_POLICY_EXTERNAL &&
         d.reward_method[0] == SDC_REWARD_DEFAULT && d.reward_method[1] == SDC_REWARD_DEFAULT &&
         d.reward_method[2] == SDC_REWARD_DEFAULT;
}
// ... and the lane-per-env kernel's GENERAL form (sdc_wide.hip GEN): several configs (SdcWideCfg: rebuild_wide_cfg), rule-based
// policies on any slot, any reward function for the dc / battery agents.  The ls agent keeps default_ls_reward -- with another one
// the history is not appended to (utils/reward_creator.py:63), a mode the per-lane reward path does not have.
bool wide_gen_case(const sdc_handle* h, const int32_t* actions, const float* obs, const float* share_obs, const float* info,
                   const float* final_obs, bool timed) {
  const SdcDev& d = h->d;
  const bool acts = actions != nullptr || (d.policy[0] != SDC_POLICY_EXTERNAL && d.policy[1] != SDC_POLICY_EXTERNAL &&
                                           d.policy[2] != SDC_POLICY_EXTERNAL);
  return lockstep_case(h, share_obs, info, timed) && wide_shape(h, obs, share_obs, info, final_obs) && h->wide_gen_ok && acts &&
         d.reward_method[0] == SDC_REWARD_DEFAULT;
}

// several configs: (re)build every env's copy of its config's scalars once all configs and the assignment are known
int rebuild_prm_env(sdc_handle* h) {
  h->prm_env_ok = false;
  h->d.prm_env = nullptr;
  const int N = h->cfg.n_envs, C = h->cfg.n_dc_configs;
  if (C <= 1 || (int)h->cfg_host.size() != N) return 0;
  for (int c = 0; c < C; c++)
    if (!h->dc_set[c]) return 0;
  // (the config's scalars lie contiguously from sdc_dc_params::m_cpu to SdcDcDev::ret_sum: sdc_step.hip's P_* enum, asserted there)
  constexpr size_t P_COUNT_HOST = (offsetof(SdcDcDev, ret_sum) - offsetof(SdcDcDev, p.m_cpu)) / sizeof(double) + 1;
  static_assert(P_COUNT_HOST <= 32, "prm_env rows are 32 doubles");
  std::vector<double> tab((size_t)N * 32, 0.0);
  h->racks_max = 0;
  for (int e = 0; e < N; e++) {
    const SdcDcDev& c = h->dc_host[h->cfg_host[e]];
    std::memcpy(&tab[(size_t)e * 32], &c.p.m_cpu, sizeof(double) * P_COUNT_HOST);
    h->racks_max = std::max(h->racks_max, c.p.n_racks);
  }
  if (!h->prm_env_dev && dev_alloc(h, &h->prm_env_dev, (size_t)N * 32) != 0) return -1;
  HIP_TRY(hipMemcpy(h->prm_env_dev, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
  h->d.prm_env = h->prm_env_dev;
  h->prm_env_ok = true;
  return 0;
}

// the lane-per-env kernel's general form: one SdcWideCfg per config, if every config is set, has <= 32 racks in <= SDC_WIDE_MAX_CLS
// classes, and the scalars the kernel keeps wave-uniform are the same bits in all of them
int rebuild_wide_cfg(sdc_handle* h) {
  h->wide_gen_ok = false;
  h->d.wcfg = nullptr;
  const int C = h->cfg.n_dc_configs;
  if (C > SDC_WIDE_MAX_CFG || (int)h->dc_set.size() != C || (int)h->dc_host.size() != C) return 0;
  for (int c = 0; c < C; c++)
    if (!h->dc_set[c]) return 0;
  std::vector<SdcWideCfg> tab((size_t)C);
  std::memset(tab.data(), 0, sizeof(SdcWideCfg) * tab.size());
  int max_cls = 0, max_racks = 0;
  for (int c = 0; c < C; c++) {
    const SdcDcDev& e = h->dc_host[c];
    const sdc_dc_params& p = e.p;
    const SdcDcDev& e0 = h->dc_host[0];
    // wave-uniform in the kernel (read from config 0): must not differ
    const double shared_c[] = {p.m_cpu, p.c_cpu, p.rs_cpu, p.m_fan, p.c_fan, p.rs_fan, p.itfan_ref_p, p.itfan_ref_v_ratio, p.it_fan_full_load_v,
                               p.c_air, p.rho_air, p.crac_supply_pu, p.min_temp, p.max_temp, e.rc_itfan_ref_v_ratio, e.rc_rho_air, e.k_outlet};
    const sdc_dc_params& p0 = e0.p;
    const double shared_0[] = {p0.m_cpu, p0.c_cpu, p0.rs_cpu, p0.m_fan, p0.c_fan, p0.rs_fan, p0.itfan_ref_p, p0.itfan_ref_v_ratio, p0.it_fan_full_load_v,
                               p0.c_air, p0.rho_air, p0.crac_supply_pu, p0.min_temp, p0.max_temp, e0.rc_itfan_ref_v_ratio, e0.rc_rho_air, e0.k_outlet};
    if (std::memcmp(shared_c, shared_0, sizeof(shared_c)) != 0) return 0;
    if (p.n_racks > 32) return 0;
    SdcWideCfg& w = tab[(size_t)c];
    int n = 0;
    for (int r = 0; r < p.n_racks; r++) {
      int k = -1;
      for (int j = 0; j < n; j++)
        if (w.cls[j][0] == p.rack_n[r] && w.cls[j][1] == p.rack_supply[r] && w.cls[j][2] == p.rack_full[r] && w.cls[j][3] == p.rack_idle[r]) k = j;
      if (k < 0) {
        if (n == SDC_WIDE_MAX_CLS) return 0;
        k = n++;
        w.cls[k][0] = p.rack_n[r]; w.cls[k][1] = p.rack_supply[r]; w.cls[k][2] = p.rack_full[r]; w.cls[k][3] = p.rack_idle[r];
      }
      w.map[r >> 3] |= (unsigned)k << (4 * (r & 7));
    }
    w.n_cls = n;
    w.n_racks = p.n_racks;
    w.scal[WC_RET_SUM] = e.ret_sum; w.scal[WC_RC_N_RACKS] = e.rc_n_racks; w.scal[WC_CT_FAN_REF_P] = p.ct_fan_ref_p;
    w.scal[WC_RC_CTAFR] = e.rc_ctafr; w.scal[WC_BAT_CAP] = p.bat_capacity_mwh; w.scal[WC_RC_BAT_CAP] = e.rc_bat_capacity;
    max_cls = std::max(max_cls, n);
    max_racks = std::max(max_racks, p.n_racks);
  }
  if (!h->wcfg_dev && dev_alloc(h, &h->wcfg_dev, (size_t)SDC_WIDE_MAX_CFG) != 0) return -1;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h->wcfg_dev, tab.data(), sizeof(SdcWideCfg) * tab.size(), hipMemcpyHostToDevice));
  h->d.wcfg = h->wcfg_dev;
  h->d.wide_max_cls = max_cls;
  h->d.wide_max_racks4 = (max_racks + 3) / 4 * 4;
  h->wide_gen_ok = true;
  return 0;
}

// the reward state (rank windows, running sums) describes the ring contents: drop it when the ring is injected
int invalidate_trackers(sdc_handle* h) {
  std::vector<unsigned> z((size_t)h->cfg.n_envs, 0u);
  if (rec_put(h, H_VALID, 1, z.data(), 1)) return -1;
  return 0;
}

// keep the library's copy of the latest observations (closed loop only: no actor set, no copy)
int latch_obs(sdc_handle* h, const float* obs, hipStream_t st) {
  if (!h->obs_latch || !obs) return 0;
  HIP_TRY(hipMemcpyAsync(h->obs_latch, obs, sizeof(float) * (size_t)h->cfg.n_envs * SDC_OBS_OUT, hipMemcpyDeviceToDevice, st));
  h->latch_valid = true;
  return 0;
}

// the envs whose episode ended with the step(s) just launched, from the host mirror of the step counters
void note_done(sdc_handle* h) {
  const int N = h->cfg.n_envs;
  h->last_done.assign((size_t)N, 0);
  h->n_last_done = 0;
  for (int e = 0; e < N; e++)
    if (h->host_t_rel[e] >= h->cfg.episode_steps) {
      h->last_done[e] = 1;
      h->n_last_done += 1;
    }
}

void recompute_steps_to_terminal(sdc_handle* h) {
  sync_mirror(h);
  int m = 1 << 30;
  for (int e = 0; e < h->cfg.n_envs; e++) {
    const int left = h->cfg.episode_steps - h->host_t_rel[e];
    if (left < m) m = left;
  }
  h->steps_to_terminal = m;
  // envs in lock-step: the kernels are told the episode step up front (see env_step)
  h->rel_hint = h->host_t_rel.empty() ? -1 : h->host_t_rel[0];
  for (int e = 1; e < h->cfg.n_envs && h->rel_hint >= 0; e++)
    if (h->host_t_rel[e] != h->rel_hint) h->rel_hint = -1;
}

}  // namespace

extern "C" {

const char* sdc_last_error(void) { return g_err.c_str(); }
int sdc_version(void) { return SDC_ABI_VERSION; }

int sdc_create(const sdc_config* cfg, sdc_handle** out) {
  if (!cfg || !out) return fail_msg("sdc_create: null argument");
  if (cfg->n_envs <= 0) return fail_msg("sdc_create: n_envs must be > 0");
  if (cfg->episode_steps <= 0) return fail_msg("sdc_create: episode_steps must be > 0");
  if (cfg->hist_cap < 2 || cfg->hist_cap > SDC_HIST_STRIDE)
    return fail_msg("sdc_create: hist_cap must be in [2, 10240]");
  if (cfg->n_locations <= 0 || cfg->n_dc_configs <= 0) return fail_msg("sdc_create: need >= 1 location and dc config");
  if (cfg->env_index_base < 0) return fail_msg("sdc_create: env_index_base must be >= 0");
  if (cfg->queue_max_len <= 0 || cfg->queue_max_len > 65535) return fail_msg("sdc_create: bad queue_max_len");
  if ((long long)cfg->episode_steps * 20 > 0x7FFFFFFFLL / cfg->episode_steps)
    return fail_msg("sdc_create: episode too long for the 32-bit queue prefix sums");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (cfg->device < 0 || cfg->device >= ndev) return fail_msg("sdc_create: no such HIP device");
  HIP_TRY(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail_msg(std::string("sdc_create: built for gfx950 (MI355X) only, device is ") + prop.gcnArchName);

  sdc_handle* h = new sdc_handle();
  h->cfg = *cfg;
  if (cfg->debug_flags & 64)   // test hook (tests of the launch counter's wrap): start the counter where the environment says
    if (const char* t = std::getenv("SDC_TEST_STEP_NO")) h->step_no = std::atoi(t) % STEP_WRAP;
  h->device = cfg->device;
  SdcDev& d = h->d;
  std::memset(&d, 0, sizeof(d));
  const int N = cfg->n_envs;
  d.n_envs = N;
  d.episode_steps = cfg->episode_steps;
  d.hist_cap = cfg->hist_cap;
  d.queue_max = cfg->queue_max_len;
  d.rc_queue_max = 1.0 / (double)cfg->queue_max_len;
  d.rc_hist_cap = 1.0 / (double)cfg->hist_cap;
  d.queue_max_d = (double)cfg->queue_max_len;
  d.hist_cap_d = (double)cfg->hist_cap;
  d.table_len = SDC_TABLE_LEN;
  static_assert(SDC_TABLE_LEN % 8 == 0, "sdc_reset_kernel: a lane's 8 samples of the year's walk are all inside the table or all outside");
  d.lw = cfg->episode_steps + 18;
  d.qstride = (cfg->episode_steps + 63) / 64 * 64;
  d.max_roll_days = cfg->max_roll_days;
  d.debug_flags = cfg->debug_flags;
  d.env_base = cfg->env_index_base;
  for (int a = 0; a < 3; a++) {
    if (cfg->reward_method[a] < 0 || cfg->reward_method[a] > SDC_REWARD_WATER) {
      sdc_destroy(h);
      return fail_msg("sdc_create: unknown reward_method");
    }
    d.reward_method[a] = cfg->reward_method[a];
  }
  for (int a = 0; a < 3; a++) {
    const int pol = cfg->policy[a];
    const bool ok = pol == SDC_POLICY_EXTERNAL || pol == SDC_POLICY_DO_NOTHING || (a == 2 && pol == SDC_POLICY_RBC) ||
                    (a == 1 && pol == SDC_POLICY_TRIM_AND_RESPOND);
    if (!ok) {
      sdc_destroy(h);
      return fail_msg("sdc_create: policy must be EXTERNAL or DO_NOTHING, RBC for the battery slot, TRIM_AND_RESPOND for the dc slot");
    }
    d.policy[a] = pol;
  }
  d.tr_limit = cfg->trim_and_respond_limit;
  d.actions_out = nullptr;
  d.seed = cfg->seed;
  d.noise_std = cfg->weather_noise_std;
  d.noise_weight = cfg->weather_noise_weight;

#define A(ptr, count)                                              \
  do {                                                             \
    if (dev_alloc(h, &(ptr), (size_t)(count)) != 0) {              \
      sdc_destroy(h);                                              \
      return -1;                                                   \
    }                                                              \
  } while (0)
  double *tabW, *tabC, *tabT, *tabWB, *hour_lut;
  SdcDcDev* dcp;
  A(tabW, (size_t)cfg->n_locations * SDC_TABLE_LEN);
  A(tabC, (size_t)cfg->n_locations * SDC_TABLE_LEN);
  A(tabT, (size_t)cfg->n_locations * SDC_TABLE_LEN);
  A(tabWB, (size_t)cfg->n_locations * SDC_TABLE_LEN);
  A(hour_lut, 96 * 2);
  A(dcp, cfg->n_dc_configs);
  d.tabW = tabW; d.tabC = tabC; d.tabT = tabT; d.tabWB = tabWB; d.hour_lut = hour_lut; d.dc = dcp; d.n_cfg = cfg->n_dc_configs;
  A(d.rec, (size_t)N * SDC_REC_DWORDS);
  A(d.qtab, (size_t)N * d.qstride);
  d.qcum_t = nullptr;
  d.hist_t = nullptr;
  if ((N & 63) == 0 && (N >= SDC_WIDE_MIN_ENVS || (cfg->debug_flags & 2048))) {      // (batches the lane-per-env kernel can serve: wide_case)
    // ... and BEHIND it, in the same allocation, the history ring's slot-major mirror for the batches that get one (rows qstride ..
    // qstride + hist_cap of the same [row][N] array: the lane-per-env kernel addresses it from the pointer and the strides it holds anyway)
    const bool mirror = N >= SDC_HIST_MIRROR_MIN_ENVS;
    A(d.qcum_t, (size_t)N * ((size_t)d.qstride + (mirror ? (size_t)d.hist_cap : 0)));      // (zeroed by the allocation)
    if (mirror) {
      d.hist_t = d.qcum_t + (size_t)N * d.qstride;
      if (hipMemset(d.hist_t, 0xFF, sizeof(unsigned) * (size_t)N * d.hist_cap) != hipSuccess) {      // every slot empty
        sdc_destroy(h);
        return fail_msg("sdc_create: clearing the history rings' mirror failed");
      }
    }
  }
  A(d.t_win, (size_t)N * d.lw);
  A(d.wb_win, (size_t)N * d.lw);
  A(d.hist, (size_t)N * SDC_HIST_STRIDE);
  if (hipMemset(d.hist, 0xFF, sizeof(unsigned) * (size_t)N * SDC_HIST_STRIDE) != hipSuccess) {  // every slot empty
    sdc_destroy(h);
    return fail_msg("sdc_create: clearing the history rings failed");
  }
  A(d.hdr, (size_t)N * SDC_HDR_DWORDS);
  A(d.qwin, (size_t)N * (4 * SDC_WIN));
  d.feat = nullptr;
  if (sizeof(double) * (size_t)(cfg->episode_steps + 25 + d.lw) <= 50 * 1024)   // the features kernel's LDS windows (+ 8.4 KB tile per wavefront)
    A(d.feat, (size_t)N * (size_t)(cfg->episode_steps + 1) * SDC_FEAT_ROW);
  A(d.rq_count, 4);
  // deferred re-centring capacity by batch size (sdc_device.hpp): ~26 requests per step and 4096 envs on average
  d.rq_max = std::min((int)SDC_RQ_LIMIT, std::max((int)SDC_RQ_MIN, (N / 32 + 127) / 128 * 128));
  d.sweep_blocks = std::min(128, std::max(32, N / 128));   // four wavefronts each; a workgroup serves requests b, b + sweep_blocks, ...
  A(d.rq, 3 * (size_t)d.rq_max);
  A(d.rs, 3 * (size_t)d.rq_max);
  A(d.reset_mask, N);
  A(h->ovr_day, N); A(h->ovr_hour, N);
  A(h->ovr_ci_min, N); A(h->ovr_ci_max, N); A(h->ovr_t_min, N); A(h->ovr_t_max, N);
#undef A

  // hour LUT: utils/managers.py:66-88 sc_obs -- round(hour/24, 3) * 2pi -> cos/sin * 0.5 + 0.5
  {
    double lut[192];
    const double two_pi = 3.141592653589793 * 2;
    for (int q = 0; q < 96; q++) {
      const double hour = q * 0.25;
      const double nh = (std::rint((hour / 24) * 1000.0) / 1000.0) * two_pi;
      lut[2 * q] = std::cos(nh) * 0.5 + 0.5;
      lut[2 * q + 1] = std::sin(nh) * 0.5 + 0.5;
    }
    if (hipMemcpy(hour_lut, lut, sizeof(lut), hipMemcpyHostToDevice) != hipSuccess) {
      sdc_destroy(h);
      return fail_msg("sdc_create: hour LUT upload failed");
    }
  }
  h->host_t_rel.assign(N, cfg->episode_steps);  // "finished": a reset is required before stepping
  h->feat_host.assign(N, 0);
  h->fields = {
      {"cursor", nullptr, 4, R_CURSOR, 1}, {"t_rel", nullptr, 4, R_TREL, 1}, {"day", nullptr, 4, R_DAY, 1},
      {"hourq", nullptr, 4, R_HOURQ, 1}, {"q_popped", nullptr, 4, R_QPOPPED, 1}, {"q_cum", nullptr, 4, R_QCUM, 1},
      {"q_cumT", nullptr, 4, R_QCUMT, 1}, {"q_head", nullptr, 4, R_QHEAD, 1}, {"q_cum_hm1", nullptr, 4, R_QCUM_HM1, 1},
      {"q_cumT_hm1", nullptr, 4, R_QCUMT_HM1, 1}, {"last_delta", nullptr, 4, R_LAST_DELTA, 1},
      {"consecutive", nullptr, 4, R_CONSEC, 1}, {"scale", nullptr, 4, R_SCALE, 1}, {"hist_len", nullptr, 4, R_HIST_LEN, 1},
      {"hist_pos", nullptr, 4, R_HIST_POS, 1}, {"episode", nullptr, 4, R_EPISODE, 1}, {"fault", nullptr, 4, R_FAULT, 1},
      {"loc_id", nullptr, 4, R_LOC, 1}, {"cfg_id", nullptr, 4, R_CFG, 1}, {"day_lo", nullptr, 4, R_DAY_LO, 1},
      {"day_hi", nullptr, 4, R_DAY_HI, 1},
      {"stpt", nullptr, 8, R_STPT, 2}, {"bat_load", nullptr, 8, R_BAT, 2}, {"ci_min", nullptr, 8, R_CI_MIN, 2},
      {"ci_den", nullptr, 8, R_CI_DEN, 2}, {"t_min", nullptr, 8, R_T_MIN, 2}, {"t_den", nullptr, 8, R_T_DEN, 2},
      {"hist_ref", nullptr, 8, R_HIST_REF, 2},
      {"record", (void**)&d.rec, 4 * SDC_REC_DWORDS, 0, 0},
      {"hist", (void**)&d.hist, sizeof(unsigned) * SDC_HIST_STRIDE, 0, 0},
      {"hist_n", nullptr, 4, H_N, 1, 1}, {"ep_return", nullptr, 24, H_RET, 6, 1},
      {"order_stat_sticky", nullptr, 4, H_STICKY, 1, 1},
      {"header", (void**)&d.hdr, 4 * SDC_HDR_DWORDS, 0, 0},
      {"qwin", (void**)&d.qwin, 4 * 4 * SDC_WIN, 0, 0},
      {"t_win", (void**)&d.t_win, sizeof(double) * (size_t)d.lw, 0, 0},
      {"wb_win", (void**)&d.wb_win, sizeof(double) * (size_t)d.lw, 0, 0},
      {"qtab", (void**)&d.qtab, sizeof(uint2) * (size_t)d.qstride, 0, 0},
  };
  // scale starts at 1, last_delta = None (envs/dc_gym.py:81-83)
  {
    std::vector<int> ones(N, 1), none(N, -2);
    if (rec_put(h, R_SCALE, 1, ones.data()) != 0 || rec_put(h, R_LAST_DELTA, 1, none.data()) != 0) {
      sdc_destroy(h);
      return -1;
    }
  }
  *out = h;
  return 0;
}

int sdc_destroy(sdc_handle* h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
  return 0;
}

int sdc_set_seed(sdc_handle* h, uint64_t seed) {
  if (!h) return fail_msg("sdc_set_seed: null handle");
  h->cfg.seed = seed;
  h->d.seed = seed;
  return 0;
}

int sdc_weather_window_len(const sdc_handle* h) { return h ? h->d.lw : -1; }
int sdc_hist_stride(const sdc_handle* h) { return h ? SDC_HIST_STRIDE : -1; }
int sdc_queue_stride(const sdc_handle* h) { return h ? h->d.qstride : -1; }

int sdc_set_tables(sdc_handle* h, int loc_id, const double* W, const double* C, const double* T, const double* WB,
                   int n) {
  if (!h || !W || !C || !T || !WB) return fail_msg("sdc_set_tables: null argument");
  if (loc_id < 0 || loc_id >= h->cfg.n_locations) return fail_msg("sdc_set_tables: loc_id out of range");
  if (n != SDC_TABLE_LEN) return fail_msg("sdc_set_tables: tables must hold 35040 samples");
  HIP_TRY(hipSetDevice(h->device));
  const size_t off = (size_t)loc_id * SDC_TABLE_LEN, bytes = sizeof(double) * SDC_TABLE_LEN;
  HIP_TRY(hipMemcpy(const_cast<double*>(h->d.tabW) + off, W, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(const_cast<double*>(h->d.tabC) + off, C, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(const_cast<double*>(h->d.tabT) + off, T, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(const_cast<double*>(h->d.tabWB) + off, WB, bytes, hipMemcpyHostToDevice));
  if (invalidate_features(h)) return -1;
  h->tables_set = true;
  return 0;
}

int sdc_set_dc_params(sdc_handle* h, int cfg_id, const sdc_dc_params* p) {
  if (!h || !p) return fail_msg("sdc_set_dc_params: null argument");
  if (cfg_id < 0 || cfg_id >= h->cfg.n_dc_configs) return fail_msg("sdc_set_dc_params: cfg_id out of range");
  if (p->n_racks <= 0 || p->n_racks > SDC_MAX_RACKS) return fail_msg("sdc_set_dc_params: n_racks must be in [1, 64]");
  HIP_TRY(hipSetDevice(h->device));
  // reciprocals for the kernels' 3-instruction divisions: exact unless a divisor's significand is all ones
  SdcDcDev e;
  e.p = *p;
  const double divisors[5] = {(double)p->n_racks, p->itfan_ref_v_ratio, p->rho_air, p->ctafr, p->bat_capacity_mwh};
  double* rcs[5] = {&e.rc_n_racks, &e.rc_itfan_ref_v_ratio, &e.rc_rho_air, &e.rc_ctafr, &e.rc_bat_capacity};
  for (int i = 0; i < 5; i++) {
    unsigned long long bits;
    std::memcpy(&bits, &divisors[i], 8);
    if (!(divisors[i] > 0) || !std::isfinite(divisors[i]) || (bits & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull)
      return fail_msg("sdc_set_dc_params: n_racks, itfan_ref_v_ratio, rho_air, ctafr and bat_capacity_mwh must be "
                      "positive, finite, and not have an all-ones significand");
    *rcs[i] = 1.0 / divisors[i];
  }
  e.k_outlet = 1.918 / (p->c_air * p->rho_air * 0.526);
  e.n_racks_f = (double)p->n_racks;
  e.ret_sum = 0.0;
  for (int r = 0; r < p->n_racks; r++) e.ret_sum += p->rack_return[r];
  // rack classes: distinct (cpus, full, idle, supply) tuples in order of first appearance, grouped by (cpus, supply)
  {
    SdcRackClasses& rc = e.rc;
    std::memset(&rc, 0, sizeof(rc));
    struct Cls { double n, full, idle, supply; int grp; };
    std::vector<Cls> cls;
    std::vector<std::pair<double, double>> grp;
    std::vector<int> of_rack((size_t)p->n_racks, 0);
    bool fits = p->n_racks <= 32;
    for (int r = 0; r < p->n_racks && fits; r++) {
      const Cls c = {p->rack_n[r], p->rack_full[r], p->rack_idle[r], p->rack_supply[r], 0};
      int k = -1;
      for (size_t j = 0; j < cls.size(); j++)
        if (cls[j].n == c.n && cls[j].full == c.full && cls[j].idle == c.idle && cls[j].supply == c.supply) k = (int)j;
      if (k < 0) {
        int g = -1;
        for (size_t j = 0; j < grp.size(); j++)
          if (grp[j].first == c.n && grp[j].second == c.supply) g = (int)j;
        if (g < 0) { g = (int)grp.size(); grp.push_back({c.n, c.supply}); }
        k = (int)cls.size();
        cls.push_back(c);
        cls.back().grp = g;
      }
      of_rack[(size_t)r] = k;
      if (cls.size() > SDC_MAX_RACK_CLS) fits = false;
    }
    if (fits) {
      // renumber the classes group by group
      std::vector<int> renum(cls.size(), 0);
      int next = 0;
      rc.n_grp = (int)grp.size();
      for (int g = 0; g < rc.n_grp; g++) {
        rc.grp_begin[g] = next;
        rc.grp_n[g] = grp[(size_t)g].first;
        rc.grp_supply[g] = grp[(size_t)g].second;
        for (size_t j = 0; j < cls.size(); j++)
          if (cls[j].grp == g) {
            renum[j] = next;
            rc.cls_full[next] = cls[j].full;
            rc.cls_idle[next] = cls[j].idle;
            next++;
          }
      }
      rc.grp_begin[rc.n_grp] = next;
      rc.n_cls = next;
      for (int r = 0; r < p->n_racks; r++) rc.cls_of_rack[r] = renum[(size_t)of_rack[(size_t)r]];
    }
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(const_cast<SdcDcDev*>(h->d.dc) + cfg_id, &e, sizeof(e), hipMemcpyHostToDevice));
  if (cfg_id == 0) {
    h->racks_cfg0 = p->n_racks;
    h->rack_cls_cfg0 = e.rc.n_cls;
  }
  h->dc_host.resize((size_t)h->cfg.n_dc_configs);
  h->dc_set.resize((size_t)h->cfg.n_dc_configs, 0);
  h->dc_host[cfg_id] = e;
  h->dc_set[cfg_id] = 1;
  if (rebuild_prm_env(h)) return -1;
  return rebuild_wide_cfg(h);
}

int sdc_assign_envs(sdc_handle* h, const int32_t* loc_id, const int32_t* cfg_id, const int32_t* day_lo,
                    const int32_t* day_hi) {
  if (!h || !loc_id || !cfg_id || !day_lo || !day_hi) return fail_msg("sdc_assign_envs: null argument");
  const int N = h->cfg.n_envs;
  for (int e = 0; e < N; e++) {
    if (loc_id[e] < 0 || loc_id[e] >= h->cfg.n_locations) return fail_msg("sdc_assign_envs: loc_id out of range");
    if (cfg_id[e] < 0 || cfg_id[e] >= h->cfg.n_dc_configs) return fail_msg("sdc_assign_envs: cfg_id out of range");
    if (day_lo[e] < 0 || day_hi[e] > 364 || day_lo[e] > day_hi[e]) return fail_msg("sdc_assign_envs: bad day range");
  }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  if (rec_put(h, R_LOC, 1, loc_id) || rec_put(h, R_CFG, 1, cfg_id) || rec_put(h, R_DAY_LO, 1, day_lo) ||
      rec_put(h, R_DAY_HI, 1, day_hi) || invalidate_features(h))
    return -1;
  // the CRAC set-point starts at the config's initial value (make_envs_pyenv.py:124) and is never reset
  if (!h->started) {
    std::vector<SdcDcDev> ps(h->cfg.n_dc_configs);
    HIP_TRY(hipMemcpy(ps.data(), h->d.dc, sizeof(SdcDcDev) * ps.size(), hipMemcpyDeviceToHost));
    std::vector<double> st(N);
    for (int e = 0; e < N; e++) st[e] = ps[cfg_id[e]].p.init_setpoint;
    if (rec_put(h, R_STPT, 2, st.data())) return -1;
  }
  h->assigned = true;
  h->cfg_host.assign(cfg_id, cfg_id + N);
  h->dc_set.resize((size_t)h->cfg.n_dc_configs, 0);
  if (rebuild_prm_env(h)) return -1;
  return rebuild_wide_cfg(h);
}

int sdc_reset(sdc_handle* h, const uint8_t* mask_host, const sdc_reset_override* ovr, float* obs, float* share_obs,
              void* stream) {
  if (!h) return fail_msg("sdc_reset: null handle");
  if (!h->tables_set || !h->assigned) return fail_msg("sdc_reset: call sdc_set_tables / sdc_set_dc_params / sdc_assign_envs first");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int N = h->cfg.n_envs;
  SdcDev d = h->d;
  if (mask_host) {
    HIP_TRY(hipMemcpyAsync(d.reset_mask, mask_host, (size_t)N, hipMemcpyHostToDevice, st));
  } else {
    d.reset_mask = nullptr;
  }
  double* inj_noise = nullptr;
  int* inj_roll = nullptr;
  const bool inject_noise = ovr && ovr->noise;
  if (inject_noise) {
    // the reference's own draws, the arithmetic on the device: day / hour / roll + the year's noise array per env
    if (!ovr->day || !ovr->hour || !ovr->roll_days) return fail_msg("sdc_reset: noise injection needs day, hour and roll_days");
    for (int e = 0; e < N; e++) {
      if (mask_host && !mask_host[e]) continue;
      if (ovr->day[e] < 0 || ovr->day[e] > 364 || ovr->hour[e] < 0 || ovr->hour[e] > 23 || ovr->roll_days[e] < 0 ||
          ovr->roll_days[e] > 364)
        return fail_msg("sdc_reset: injected day / hour / roll_days out of range");
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&inj_noise), sizeof(double) * (size_t)N * SDC_TABLE_LEN));
    if (hipMalloc(reinterpret_cast<void**>(&inj_roll), sizeof(int) * (size_t)N) != hipSuccess) {
      (void)hipFree(inj_noise);
      return fail_msg("sdc_reset: allocation for the injected roll failed");
    }
    hipError_t e1 = hipMemcpy(inj_noise, ovr->noise, sizeof(double) * (size_t)N * SDC_TABLE_LEN, hipMemcpyHostToDevice);
    hipError_t e2 = hipMemcpy(inj_roll, ovr->roll_days, sizeof(int) * (size_t)N, hipMemcpyHostToDevice);
    hipError_t e3 = hipMemcpy(h->ovr_day, ovr->day, sizeof(int) * (size_t)N, hipMemcpyHostToDevice);
    hipError_t e4 = hipMemcpy(h->ovr_hour, ovr->hour, sizeof(int) * (size_t)N, hipMemcpyHostToDevice);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
      (void)hipFree(inj_noise);
      (void)hipFree(inj_roll);
      return fail_msg("sdc_reset: upload of the injected noise failed");
    }
  } else if (ovr) {
    if (!ovr->day || !ovr->hour || !ovr->ci_min || !ovr->ci_max || !ovr->t_min || !ovr->t_max || !ovr->t_win ||
        !ovr->wb_win)
      return fail_msg("sdc_reset: incomplete override");
    for (int e = 0; e < N; e++) {
      if (mask_host && !mask_host[e]) continue;
      if (ovr->day[e] < 0 || ovr->day[e] > 364 || ovr->hour[e] < 0 || ovr->hour[e] > 23)
        return fail_msg("sdc_reset: override day/hour out of range");
    }
    HIP_TRY(hipMemcpyAsync(h->ovr_day, ovr->day, sizeof(int) * (size_t)N, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->ovr_hour, ovr->hour, sizeof(int) * (size_t)N, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->ovr_ci_min, ovr->ci_min, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->ovr_ci_max, ovr->ci_max, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->ovr_t_min, ovr->t_min, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->ovr_t_max, ovr->t_max, sizeof(double) * (size_t)N, hipMemcpyHostToDevice, st));
    const size_t row = sizeof(double) * (size_t)d.lw;
    if (!mask_host) {
      HIP_TRY(hipMemcpyAsync(d.t_win, ovr->t_win, row * N, hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(d.wb_win, ovr->wb_win, row * N, hipMemcpyHostToDevice, st));
    } else {
      for (int e = 0; e < N; e++) {
        if (!mask_host[e]) continue;
        HIP_TRY(hipMemcpyAsync(d.t_win + (size_t)e * d.lw, ovr->t_win + (size_t)e * d.lw, row, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d.wb_win + (size_t)e * d.lw, ovr->wb_win + (size_t)e * d.lw, row, hipMemcpyHostToDevice, st));
      }
    }
    // the host buffers may be pageable: the copies above must have consumed them before we return
    HIP_TRY(hipStreamSynchronize(st));
  }
  hipLaunchKernelGGL(sdc_reset_kernel, dim3(N), dim3(SDC_WAVE), 0, st, d, inject_noise ? 2 : (ovr ? 1 : 0), h->ovr_day,
                     h->ovr_hour, h->ovr_ci_min, h->ovr_ci_max, h->ovr_t_min, h->ovr_t_max, 0, obs, share_obs, inj_noise,
                     inj_roll);
  launch_features(h, d, st);
  if (inject_noise) {
    const hipError_t es = hipStreamSynchronize(st);
    (void)hipFree(inj_noise);
    (void)hipFree(inj_roll);
    if (es != hipSuccess) return fail("sdc_reset: injected reset", es);
  }
  HIP_TRY(hipGetLastError());
  // the closed loop's copy of the latest observations: a reset without an observation buffer leaves it stale (the next
  // sdc_rollout_actor refuses until a reset / step has delivered observations), and a MASKED reset only wrote the masked
  // envs' rows of `obs` -- the other rows of the caller's buffer are whatever it held, so only those rows are taken over
  if (h->obs_latch && !obs) h->latch_valid = false;
  if (h->obs_latch && obs && mask_host) {
    const size_t row = sizeof(float) * SDC_OBS_OUT;
    for (int e = 0; e < N;) {
      if (!mask_host[e]) { e++; continue; }
      int e1 = e;
      while (e1 < N && mask_host[e1]) e1++;
      HIP_TRY(hipMemcpyAsync(h->obs_latch + (size_t)e * SDC_OBS_OUT, obs + (size_t)e * SDC_OBS_OUT, row * (size_t)(e1 - e),
                             hipMemcpyDeviceToDevice, st));
      e = e1;
    }
  } else if (latch_obs(h, obs, st)) {
    return -1;
  }
  if (mask_host) HIP_TRY(hipStreamSynchronize(st));  // mask staging buffer is reused by the next call
  sync_mirror(h);
  for (int e = 0; e < N; e++)
    if (!mask_host || mask_host[e]) {
      h->host_t_rel[e] = 0;
      note_features(h, e);
    }
  recompute_steps_to_terminal(h);
  h->started = true;
  return 0;
}

int sdc_step(sdc_handle* h, const int32_t* actions, float* obs, float* share_obs, float* rew, uint8_t* done,
             float* info, float* final_obs, void* stream) {
  if (!h || !obs || !rew || !done) return fail_msg("sdc_step: null argument");
  if (!actions && !all_policies(h)) return fail_msg("sdc_step: actions may only be NULL when every agent slot has a policy");
  if (!h->started) return fail_msg("sdc_step: sdc_reset must be called first");
  if (h->steps_to_terminal <= 0)
    return fail_msg("sdc_step: an environment has finished its episode; call sdc_reset (auto_reset is off)");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int N = h->cfg.n_envs;
  const bool timed = h->prof > 0 && (h->prof_tick++ % h->prof) == 0 && h->prof_used < PROF_SLOTS;
  SdcDev d = h->d;
  if (timed) {
    d.prof_ts = h->prof_buf + (size_t)h->prof_used * 3 * N * 2;
    h->prof_has_reset[h->prof_used] = 0;
  }
  d.step_no = h->step_no;
  h->step_no = next_step_no(h->step_no, 1);
  if (fast_case(h, actions, share_obs, info, timed) && wide_case(h, obs, share_obs, info, final_obs)) {
    d.sweep_blocks = wide_sweep_blocks(h);
    h->last_step_kernel = "sdc_dynamics_wide_kernel";
    hipLaunchKernelGGL(sdc_dynamics_wide_kernel, dim3(d.sweep_blocks + N / SDC_WAVE), dim3(2 * SDC_WAVE), 0, st, d, h->rel_hint, actions,
                       obs, share_obs, done, info, final_obs, rew);
  } else if (wide_gen_case(h, actions, obs, share_obs, info, final_obs, timed)) {
    // a large batch of SEVERAL configs, or with rule-based policies / other reward functions: the lane-per-env kernel's general form
    d.sweep_blocks = wide_sweep_blocks(h);
    h->last_step_kernel = "sdc_dynamics_wide_gen_kernel";
    hipLaunchKernelGGL(sdc_dynamics_wide_gen_kernel, dim3(d.sweep_blocks + N / SDC_WAVE), dim3(2 * SDC_WAVE), 0, st, d, h->rel_hint, actions,
                       obs, share_obs, done, info, final_obs, rew);
  } else if (fast_case(h, actions, share_obs, info, timed) && quad_case(h, false)) {
    h->last_step_kernel = "sdc_dynamics_quad_kernel";
    hipLaunchKernelGGL(sdc_dynamics_quad_kernel, dim3(d.sweep_blocks + quad_blocks(N)), dim3(SDC_WAVE * STEP_WPB), 0, st, d,
                       h->rel_hint, actions, obs, share_obs, done, info, final_obs, rew);
  } else if (fast_case(h, actions, share_obs, info, timed)) {
    h->last_step_kernel = "sdc_dynamics_fast_kernel";
    hipLaunchKernelGGL(sdc_dynamics_fast_kernel, dim3(d.sweep_blocks + step_blocks(N)), dim3(SDC_WAVE * STEP_WPB), 0, st, d,
                       h->rel_hint, actions, obs, share_obs, done, info, final_obs, rew);
  } else {
    h->last_step_kernel = "sdc_dynamics_kernel";
    hipLaunchKernelGGL(sdc_dynamics_kernel, dim3(d.sweep_blocks + step_blocks(N)), dim3(SDC_WAVE * STEP_WPB), 0, st, d, h->rel_hint,
                       actions, obs, share_obs, done, info, final_obs, rew);
  }
  if (h->cfg.debug_flags & 1) hipLaunchKernelGGL(sdc_reward_verify_kernel, dim3(N), dim3(SDC_BLOCK), 0, st, d, info);
  HIP_TRY(hipGetLastError());
  h->n_last_done = 0;
  h->steps_to_terminal -= 1;
  h->pending += 1;
  if (h->rel_hint >= 0) h->rel_hint += 1;
  if (h->steps_to_terminal == 0) {
    // At least one env just finished.  Episodes have a fixed length and every env advances one step per
    // launch, so the host knows this from its mirror of the step counters -- no device read-back.
    sync_mirror(h);
    note_done(h);
    if (h->cfg.auto_reset) {
      // harl/envs/env_wrappers.py:176-190: reset inside the same step call and return the reset obs
      d.reset_mask = nullptr;
      if (timed) h->prof_has_reset[h->prof_used] = 1;
      hipLaunchKernelGGL(sdc_reset_kernel, dim3(N), dim3(SDC_WAVE), 0, st, d, 0, h->ovr_day, h->ovr_hour, h->ovr_ci_min,
                         h->ovr_ci_max, h->ovr_t_min, h->ovr_t_max, 1, obs, share_obs, nullptr, nullptr);
      launch_features(h, d, st);
      HIP_TRY(hipGetLastError());
      for (int e = 0; e < N; e++)
        if (h->host_t_rel[e] >= h->cfg.episode_steps) {
          h->host_t_rel[e] = 0;
          note_features(h, e);
        }
      recompute_steps_to_terminal(h);
    }
  }
  if (timed) h->prof_used += 1;
  if (latch_obs(h, obs, st)) return -1;
  return 0;
}

int sdc_rollout(sdc_handle* h, int n_steps, const int32_t* actions, float* obs, float* share_obs, float* rew,
                uint8_t* done, float* info, float* final_obs, int32_t* actions_out, void* stream) {
  if (!h || !obs || !rew || !done) return fail_msg("sdc_rollout: null argument");
  if (!actions && !all_policies(h)) return fail_msg("sdc_rollout: actions may only be NULL when every agent slot has a policy");
  if (!h->started) return fail_msg("sdc_rollout: sdc_reset must be called first");
  if (n_steps <= 0) return fail_msg("sdc_rollout: n_steps must be positive");
  if (n_steps > h->steps_to_terminal)
    return fail_msg("sdc_rollout: the rollout would run past the end of an episode (" +
                    std::to_string(h->steps_to_terminal) + " steps left); split it there");
  if (h->cfg.debug_flags & 1) return fail_msg("sdc_rollout: verify mode checks single steps; use sdc_step");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int N = h->cfg.n_envs;
  SdcDev d = h->d;
  d.actions_out = actions_out;
  // (a MULTI-STEP launch of four envs per wavefront has no launch boundary between its steps: it stays ahead of K lane-per-env
  // launches up to SDC_WIDE_ROLLOUT_MIN_ENVS envs -- 8 192 envs: 11.6 against 14.3 us per step, 12 288: 15.4 / 15.6, 16 384: 23.4 / 17.4)
  const bool roll_wide = N >= SDC_WIDE_ROLLOUT_MIN_ENVS || (h->d.debug_flags & 2048);
  const bool w_common = roll_wide && actions && !actions_out && fast_case(h, actions, share_obs, info, false) &&
                        wide_case(h, obs, share_obs, info, final_obs);
  // (the slices of step k start k * N rows in: aligned like the arrays themselves for the batches this kernel takes, N % 64 == 0)
  const bool w_gen = roll_wide && !w_common && wide_gen_case(h, actions, obs, share_obs, info, final_obs, false) &&
                     (!actions_out || (reinterpret_cast<uintptr_t>(actions_out) & 3u) == 0);
  if (w_common || w_gen) {
    // A batch the lane-per-env kernel serves (sdc_wide.hip), from SDC_WIDE_ROLLOUT_MIN_ENVS envs: n_steps single-step launches of it, the
    // deferred re-centrings running between them as in sdc_step -- faster than one n_steps launch of four envs per wavefront
    // (16 384 envs: 17.7 against 23.4 us per step), the same outputs to the bit.  Its general form likewise: several configs,
    // rule-based policies (a step's policy reads the state the previous launch left), other reward functions.
    d.sweep_blocks = wide_sweep_blocks(h);
    h->last_step_kernel = w_common ? "sdc_dynamics_wide_kernel" : "sdc_dynamics_wide_gen_kernel";
    for (int k = 0; k < n_steps; k++) {
      d.step_no = h->step_no;
      h->step_no = next_step_no(h->step_no, 1);
      const size_t o = (size_t)k * N;
      d.actions_out = actions_out ? actions_out + o * 3 : nullptr;
      const int rel_k = h->rel_hint >= 0 ? h->rel_hint + k : h->rel_hint;
      if (w_common)
        hipLaunchKernelGGL(sdc_dynamics_wide_kernel, dim3(d.sweep_blocks + N / SDC_WAVE), dim3(2 * SDC_WAVE), 0, st, d, rel_k,
                           actions + o * 3, obs + o * SDC_OBS_OUT, share_obs + o * SDC_SHARE_OBS_DIM, done + o,
                           info + o * SDC_INFO_DIM, final_obs, rew + o * 3);
      else
        hipLaunchKernelGGL(sdc_dynamics_wide_gen_kernel, dim3(d.sweep_blocks + N / SDC_WAVE), dim3(2 * SDC_WAVE), 0, st, d, rel_k,
                           actions ? actions + o * 3 : nullptr, obs + o * SDC_OBS_OUT, share_obs + o * SDC_SHARE_OBS_DIM, done + o,
                           info + o * SDC_INFO_DIM, final_obs, rew + o * 3);
    }
  } else {
    // (a multi-step launch has no spare wavefronts between its steps: it re-centres inline, and requests left by the
    // step before it are dropped -- their results would describe a ring several steps old)
    h->step_no = next_step_no(h->step_no, 0);        // (room for the launch's n_steps stamps below the wrap)
    d.step_no = h->step_no;
    h->step_no = next_step_no(h->step_no, n_steps + 3);
    HIP_TRY(hipMemsetAsync(d.rq_count, 0, sizeof(int) * 4, st));
    const bool r_fast = fast_case(h, actions, share_obs, info, false) && !actions_out;
    h->last_step_kernel = r_fast ? (quad_case(h, true) ? "sdc_rollout_quad_kernel" : "sdc_rollout_fast_kernel") : "sdc_rollout_kernel";
    if (fast_case(h, actions, share_obs, info, false) && !actions_out && quad_case(h, true))
      hipLaunchKernelGGL(sdc_rollout_quad_kernel, dim3(quad_blocks(N)), dim3(SDC_WAVE * STEP_WPB), 0, st, d, n_steps, h->rel_hint,
                         actions, obs, share_obs, done, info, final_obs, rew);
    else if (fast_case(h, actions, share_obs, info, false) && !actions_out)
      hipLaunchKernelGGL(sdc_rollout_fast_kernel, dim3(step_blocks(N)), dim3(SDC_WAVE * STEP_WPB), 0, st, d, n_steps, h->rel_hint,
                         actions, obs, share_obs, done, info, final_obs, rew);
    else
      hipLaunchKernelGGL(sdc_rollout_kernel, dim3(step_blocks(N)), dim3(SDC_WAVE * STEP_WPB), 0, st, d, n_steps, h->rel_hint, actions,
                         obs, share_obs, done, info, final_obs, rew);
  }
  HIP_TRY(hipGetLastError());
  h->n_last_done = 0;
  h->steps_to_terminal -= n_steps;
  h->pending += n_steps;
  if (h->rel_hint >= 0) h->rel_hint += n_steps;
  if (h->steps_to_terminal == 0) {
    sync_mirror(h);
    note_done(h);
    if (h->cfg.auto_reset) {
      // as in sdc_step: the finished envs are reset inside the call; the LAST step's obs / share_obs slices receive
      // the reset observation, final_obs the pre-reset one
      d.reset_mask = nullptr;
      const size_t last = (size_t)(n_steps - 1) * N;
      hipLaunchKernelGGL(sdc_reset_kernel, dim3(N), dim3(SDC_WAVE), 0, st, d, 0, h->ovr_day, h->ovr_hour, h->ovr_ci_min,
                         h->ovr_ci_max, h->ovr_t_min, h->ovr_t_max, 1, obs + last * SDC_OBS_OUT,
                         share_obs ? share_obs + last * SDC_SHARE_OBS_DIM : nullptr, nullptr, nullptr);
      launch_features(h, d, st);
      HIP_TRY(hipGetLastError());
      for (int e = 0; e < N; e++)
        if (h->host_t_rel[e] >= h->cfg.episode_steps) {
          h->host_t_rel[e] = 0;
          note_features(h, e);
        }
      recompute_steps_to_terminal(h);
    }
  }
  if (latch_obs(h, obs + (size_t)(n_steps - 1) * N * SDC_OBS_OUT, st)) return -1;
  return 0;
}

int sdc_set_actor(sdc_handle* h, int slot, const sdc_actor_params* p) {
  if (!h || !p) return fail_msg("sdc_set_actor: null argument");
  if (slot < 0 || slot > 2) return fail_msg("sdc_set_actor: agent_slot must be 0 (ls), 1 (dc) or 2 (bat)");
  if (p->activation < 0 || p->activation > 1) return fail_msg("sdc_set_actor: activation must be 0 (tanh) or 1 (relu)");
  HIP_TRY(hipSetDevice(h->device));
  if (!h->actor_dev) {
    if (dev_alloc(h, &h->actor_dev, 3) != 0) return -1;
    if (dev_alloc(h, &h->obs_latch, (size_t)h->cfg.n_envs * SDC_OBS_OUT) != 0) return -1;
    h->latch_valid = false;      // (filled by the next reset / step / rollout)
  }
  // torch's [out][in] rows -> the kernel's k-major layout, four consecutive k per lane (sdc_actor.hpp)
  // (per call, on the heap: 26 KB is too large for the stack, and a function-static buffer would be shared by engines on
  // other host threads -- ctypes releases the GIL during this call)
  std::unique_ptr<SdcActorDev> ap(new SdcActorDev);
  SdcActorDev& a = *ap;
  std::memset(&a, 0, sizeof(a));
  for (int k = 0; k < SDC_ACT_IN; k++) {
    a.ln0_g[k] = p->ln0_gamma[k];
    a.ln0_b[k] = p->ln0_beta[k];
  }
  for (int j = 0; j < SDC_ACT_H; j++) {
    for (int k = 0; k < SDC_ACT_IN; k++) a.w1[k / 4][j][k & 3] = p->w1[j * SDC_ACT_IN + k];
    for (int k = 0; k < SDC_ACT_H; k++) a.w2[k / 4][j][k & 3] = p->w2[j * SDC_ACT_H + k];
    a.b1[j] = p->b1[j]; a.ln1_g[j] = p->ln1_gamma[j]; a.ln1_b[j] = p->ln1_beta[j];
    a.b2[j] = p->b2[j]; a.ln2_g[j] = p->ln2_gamma[j]; a.ln2_b[j] = p->ln2_beta[j];
    for (int c = 0; c < SDC_ACT_OUT; c++) a.w3[c][j] = p->w3[c * SDC_ACT_H + j];
  }
  for (int c = 0; c < SDC_ACT_OUT; c++) a.b3[c] = p->b3[c];
  a.flags = (p->use_feature_normalization ? 1 : 0) | (p->activation << 1);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h->actor_dev + slot, &a, sizeof(a), hipMemcpyHostToDevice));
  h->actor_set[slot] = true;
  h->actor_activation[slot] = p->activation;
  return 0;
}

int sdc_rollout_actor(sdc_handle* h, int n_steps, int sample, float* obs, float* share_obs, float* rew, uint8_t* done,
                      float* info, float* final_obs, int32_t* actions_out, float* logits_out, void* stream) {
  if (!h || !obs || !share_obs || !rew || !done || !info || !actions_out) return fail_msg("sdc_rollout_actor: null argument");
  if (!h->actor_set[0] || !h->actor_set[1] || !h->actor_set[2]) return fail_msg("sdc_rollout_actor: sdc_set_actor all three agents first");
  if (h->actor_activation[0] != h->actor_activation[1] || h->actor_activation[0] != h->actor_activation[2])
    return fail_msg("sdc_rollout_actor: the three actors must share one activation (the reference builds them from one "
                    "model config: happo.yaml activation_func)");
  if (!h->started) return fail_msg("sdc_rollout_actor: sdc_reset must be called first");
  if (!h->latch_valid) return fail_msg("sdc_rollout_actor: no observations yet (the actors were set after the last reset / step: reset or step once)");
  if (n_steps <= 0) return fail_msg("sdc_rollout_actor: n_steps must be positive");
  if (n_steps > h->steps_to_terminal)
    return fail_msg("sdc_rollout_actor: the rollout would run past the end of an episode (" + std::to_string(h->steps_to_terminal) +
                    " steps left); split it there");
  // the common case only: what fast_case checks, with the actions coming from the actors instead of the caller
  static const int32_t some_actions = 0;
  if (!fast_case(h, &some_actions, share_obs, info, false) || (h->cfg.debug_flags & 1))
    return fail_msg("sdc_rollout_actor: needs the common case (lock-step batch with feature rows, one data-centre config of <= 32 "
                    "racks, external-action slots, default rewards, an even number of envs, no debug flags)");
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int N = h->cfg.n_envs;
  SdcDev d = h->d;
  d.actions_out = nullptr;
  h->step_no = next_step_no(h->step_no, 0);
  d.step_no = h->step_no;
  h->step_no = next_step_no(h->step_no, n_steps + 3);
  HIP_TRY(hipMemsetAsync(d.rq_count, 0, sizeof(int) * 4, st));
  constexpr int AWPB = 8;     // sdc_step.hip SDC_ACTOR_WPB
  if (!h->actor_lds_set) {     // (a per-device attribute: once per handle, not once per process)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sdc_rollout_actor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sdc_rollout_actor_lds_bytes()));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sdc_rollout_actor_quad_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sdc_rollout_actor_quad_lds_bytes()));
    h->actor_lds_set = true;
  }
  if (quad_case(h, true)) {   // four envs per wavefront (batches above 4096 envs)
    const int blocks = (N / 4 + AWPB - 1) / AWPB;
    hipLaunchKernelGGL(sdc_rollout_actor_quad_kernel, dim3(blocks), dim3(SDC_WAVE * AWPB), sdc_rollout_actor_quad_lds_bytes(), st, d,
                       n_steps, h->rel_hint, h->actor_dev, h->obs_latch, sample ? 1 : 0, obs, share_obs, done, info, final_obs, rew,
                       actions_out, logits_out, h->obs_latch);
  } else {
    const int blocks = ((N + 1) / 2 + AWPB - 1) / AWPB;
    hipLaunchKernelGGL(sdc_rollout_actor_kernel, dim3(blocks), dim3(SDC_WAVE * AWPB), sdc_rollout_actor_lds_bytes(), st, d, n_steps,
                       h->rel_hint, h->actor_dev, h->obs_latch, sample ? 1 : 0, obs, share_obs, done, info, final_obs, rew, actions_out,
                       logits_out, h->obs_latch);
  }
  HIP_TRY(hipGetLastError());
  h->n_last_done = 0;
  h->steps_to_terminal -= n_steps;
  h->pending += n_steps;
  if (h->rel_hint >= 0) h->rel_hint += n_steps;
  if (h->steps_to_terminal == 0) {
    sync_mirror(h);
    note_done(h);
    if (h->cfg.auto_reset) {
      d.reset_mask = nullptr;
      const size_t last = (size_t)(n_steps - 1) * N;
      hipLaunchKernelGGL(sdc_reset_kernel, dim3(N), dim3(SDC_WAVE), 0, st, d, 0, h->ovr_day, h->ovr_hour, h->ovr_ci_min,
                         h->ovr_ci_max, h->ovr_t_min, h->ovr_t_max, 1, obs + last * SDC_OBS_OUT,
                         share_obs + last * SDC_SHARE_OBS_DIM, nullptr, nullptr);
      launch_features(h, d, st);
      HIP_TRY(hipGetLastError());
      for (int e = 0; e < N; e++)
        if (h->host_t_rel[e] >= h->cfg.episode_steps) {
          h->host_t_rel[e] = 0;
          note_features(h, e);
        }
      recompute_steps_to_terminal(h);
      if (latch_obs(h, obs + last * SDC_OBS_OUT, st)) return -1;     // the next launch starts from the reset observations
    }
  }
  return 0;
}

int sdc_steps_to_episode_end(const sdc_handle* h) { return h ? h->steps_to_terminal : -1; }

const char* sdc_last_step_kernel(const sdc_handle* h) { return h ? h->last_step_kernel : ""; }

int sdc_last_done(const sdc_handle* h, uint8_t* done_host) {
  if (!h) return -1;
  if (h->n_last_done > 0 && done_host) std::memcpy(done_host, h->last_done.data(), (size_t)h->cfg.n_envs);
  return h->n_last_done;
}

int sdc_profile_enable(sdc_handle* h, int enable) {
  if (!h) return fail_msg("sdc_profile_enable: null handle");
  HIP_TRY(hipSetDevice(h->device));
  if (enable > 0 && !h->prof_buf) {
    if (dev_alloc(h, &h->prof_buf, (size_t)PROF_SLOTS * 3 * h->cfg.n_envs * 2) != 0) return -1;
    h->prof_has_reset.assign(PROF_SLOTS, 0);
    HIP_TRY(hipMemset(h->prof_buf, 0, sizeof(unsigned long long) * PROF_SLOTS * 3 * h->cfg.n_envs * 2));
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device) == hipSuccess && khz > 0)
      h->wall_clock_khz = khz;
  }
  h->prof = enable > 0 ? enable : 0;
  h->prof_tick = 0;
  return 0;
}

int sdc_profile_read(sdc_handle* h, double* out5, int reset) {
  if (!h || !out5) return fail_msg("sdc_profile_read: null argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t N = (size_t)h->cfg.n_envs;
  if (h->prof_used > 0) {
    std::vector<unsigned long long> ts((size_t)h->prof_used * 3 * N * 2);
    HIP_TRY(hipMemcpy(ts.data(), h->prof_buf, ts.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int s = 0; s < h->prof_used; s++) {
      for (int k = 0; k < 3; k++) {
        if (k == SDC_PROF_RESET && !h->prof_has_reset[s]) continue;
        const unsigned long long* p = ts.data() + ((size_t)s * 3 + k) * N * 2;
        unsigned long long t0 = ~0ull, t1 = 0ull;
        for (size_t e = 0; e < N; e++) {
          if (p[2 * e] == 0ull) continue;   // this workgroup did not run (ring path: only the queued envs)
          t0 = std::min(t0, p[2 * e]);
          t1 = std::max(t1, p[2 * e + 1]);
        }
        if (t1 > t0) h->acc_ms[k] += (double)(t1 - t0) / h->wall_clock_khz;   // first workgroup in -> last workgroup out
      }
      h->acc_ms[3] += 1;
      h->acc_ms[4] += h->prof_has_reset[s];
    }
    h->prof_used = 0;
    HIP_TRY(hipMemset(h->prof_buf, 0, sizeof(unsigned long long) * PROF_SLOTS * 3 * N * 2));
  }
  for (int i = 0; i < 5; i++) out5[i] = h->acc_ms[i];
  if (reset)
    for (int i = 0; i < 5; i++) h->acc_ms[i] = 0;
  return 0;
}

static const Field* find_field(sdc_handle* h, const char* name) {
  for (const Field& f : h->fields)
    if (std::strcmp(f.name, name) == 0) return &f;
  return nullptr;
}

int sdc_get_state(sdc_handle* h, const char* field, void* host_buf, size_t bytes) {
  if (!h || !field || !host_buf) return fail_msg("sdc_get_state: null argument");
  const Field* f = find_field(h, field);
  if (!f) return fail_msg(std::string("sdc_get_state: unknown field ") + field);
  const size_t need = f->elem * (size_t)h->cfg.n_envs;
  if (bytes != need) return fail_msg(std::string("sdc_get_state: size mismatch for ") + field);
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  if (!f->ptr) return rec_get(h, f->rec_idx, f->rec_dwords, host_buf, f->in_hdr);
  HIP_TRY(hipMemcpy(host_buf, *f->ptr, need, hipMemcpyDeviceToHost));
  if (std::strcmp(field, "hist") == 0) {  // device keys -> fp32 offsets (empty slot -> NaN)
    unsigned* u = static_cast<unsigned*>(host_buf);
    for (size_t i = 0; i < need / 4; i++) u[i] = sdc_key_f32(u[i]);
  }
  return 0;
}

// the queue table's time-major mirror rebuilt from the table (after a host write to it)
__global__ void sdc_qcum_mirror_kernel(SdcDev S) {
  const int t = (int)blockIdx.x, env = (int)(blockIdx.y * blockDim.x + threadIdx.x);
  if (env < S.n_envs) S.qcum_t[(size_t)t * S.n_envs + env] = S.qtab[(size_t)env * S.qstride + t].x;
}

// ... and the history ring's slot-major mirror from the rings
__global__ void sdc_hist_mirror_kernel(SdcDev S) {
  const int slot = (int)blockIdx.x, env = (int)(blockIdx.y * blockDim.x + threadIdx.x);
  if (env < S.n_envs) S.hist_t[(size_t)slot * S.n_envs + env] = S.hist[(size_t)env * SDC_HIST_STRIDE + slot];
}

int sdc_set_state(sdc_handle* h, const char* field, const void* host_buf, size_t bytes) {
  if (!h || !field || !host_buf) return fail_msg("sdc_set_state: null argument");
  const Field* f = find_field(h, field);
  if (!f) return fail_msg(std::string("sdc_set_state: unknown field ") + field);
  const size_t need = f->elem * (size_t)h->cfg.n_envs;
  if (bytes != need) return fail_msg(std::string("sdc_set_state: size mismatch for ") + field);
  // what the kernels index with is validated BEFORE anything reaches the device: a config id out of range would index
  // S.dc[] / the per-env config scalars out of bounds on the next step (with one config the kernels never read the id)
  const bool is_cfg = std::strcmp(field, "cfg_id") == 0, is_record = std::strcmp(field, "record") == 0;
  if (is_cfg || is_record) {
    const int* c = static_cast<const int*>(host_buf);
    for (int e = 0; e < h->cfg.n_envs; e++) {
      const int id = is_cfg ? c[e] : (int)static_cast<const unsigned*>(host_buf)[(size_t)e * SDC_REC_DWORDS + R_CFG];
      if (id < 0 || id >= h->cfg.n_dc_configs)
        return fail_msg(is_cfg ? "sdc_set_state: cfg_id out of range" : "sdc_set_state: record with a cfg_id out of range");
    }
  }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  if (!f->ptr) {
    if (rec_put(h, f->rec_idx, f->rec_dwords, host_buf, f->in_hdr)) return -1;
    if (std::strcmp(field, "hist_len") == 0 || std::strcmp(field, "hist_pos") == 0) {
      if (std::strcmp(field, "hist_len") == 0 && rec_put(h, H_N, 1, host_buf, 1)) return -1;
      if (invalidate_trackers(h)) return -1;
    }
  } else if (std::strcmp(field, "hist") == 0) {  // fp32 offsets -> device keys (NaN -> empty slot)
    if (invalidate_trackers(h)) return -1;
    std::vector<unsigned> k(need / 4);
    const unsigned* u = static_cast<const unsigned*>(host_buf);
    for (size_t i = 0; i < k.size(); i++) k[i] = ((u[i] & 0x7FFFFFFFu) > 0x7F800000u) ? 0xFFFFFFFFu : sdc_f32_key(u[i]);
    HIP_TRY(hipMemcpy(*f->ptr, k.data(), need, hipMemcpyHostToDevice));
  } else {
    HIP_TRY(hipMemcpy(*f->ptr, host_buf, need, hipMemcpyHostToDevice));
  }
  if (h->cfg.n_dc_configs > 1) {             // the envs' own copies of their configs' scalars follow the assignment
    if (is_cfg) {
      const int* c = static_cast<const int*>(host_buf);
      h->cfg_host.assign(c, c + h->cfg.n_envs);
      if (rebuild_prm_env(h)) return -1;
    } else if (is_record) {
      const unsigned* r = static_cast<const unsigned*>(host_buf);
      h->cfg_host.resize((size_t)h->cfg.n_envs);
      for (int e = 0; e < h->cfg.n_envs; e++) h->cfg_host[e] = (int)r[(size_t)e * SDC_REC_DWORDS + R_CFG];
      if (rebuild_prm_env(h)) return -1;
    }
  }
  if (h->d.qcum_t && std::strcmp(field, "qtab") == 0) {
    hipLaunchKernelGGL(sdc_qcum_mirror_kernel, dim3(h->d.qstride, (h->cfg.n_envs + 255) / 256), dim3(256), 0, 0, h->d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
  }
  if (h->d.hist_t && std::strcmp(field, "hist") == 0) {
    hipLaunchKernelGGL(sdc_hist_mirror_kernel, dim3(h->d.hist_cap, (h->cfg.n_envs + 255) / 256), dim3(256), 0, 0, h->d);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
  }
  h->latch_valid = false;                  // (closed loop: the library's copy of the latest observations describes the state before this write)
  if (invalidate_features(h)) return -1;   // whatever was written, the precomputed observation rows may no longer match it
  // Deferred window re-centrings in flight belong to the state that has just been overwritten: a restored header may
  // carry request stamps (H_PEND) that the NEXT step would find "two steps old" again and take a swept window over --
  // one that already contains the restored step's own insertion, which the take-over would then replay a second time.
  // Moving the launch counter past every stamp (3 steps: requests are served at +1 and taken over at +2) makes all of
  // them stale, in the headers and in the request / result sets alike; the windows concerned are re-requested.
  h->step_no = next_step_no(h->step_no, 3);
  HIP_TRY(hipMemset(h->d.rq_count, 0, sizeof(int) * 4));
  if (std::strcmp(field, "t_rel") == 0 || std::strcmp(field, "record") == 0) {
    std::vector<int> tr(h->cfg.n_envs);
    if (rec_get(h, R_TREL, 1, tr.data())) return -1;
    h->pending = 0;
    h->host_t_rel = tr;
    recompute_steps_to_terminal(h);
  }
  return 0;
}

}  // extern "C"
