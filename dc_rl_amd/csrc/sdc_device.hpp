// sdc_device.hpp -- device-side state layout and shared device functions of the SustainDC step.
//
// Written for gfx950 (MI355X, CDNA4) only: 64-lane wavefronts.  One kernel per timestep:
// sdc_dynamics_kernel (sdc_step.hip) -- one wavefront per PAIR of environment instances (a half-wave each) integrates
// the coupled dynamics (lanes of a half = racks for the IT model, DPP reductions inside the half), writes obs / info,
// appends the step's energy to the env's history ring and produces the history-normalised rewards from O(1)
// incremental state (sdc_trackers.hpp / sdc_halfwin.hpp); the ring itself (40 KB per env) is swept only every few
// hundred steps, by spare wavefronts of the following launch (SdcRefillReq below; sdc_ringpath.hpp).
//
// Arithmetic: fp64 for the dynamics, observation features and reductions (the reference is Python
// float / NumPy float64, and its integer / decimal-rounding cliffs only reproduce in fp64);
// the history ring is stored fp32; obs / rewards / info are written fp32.
// Compiled with -ffp-contract=off so that a*b+c rounds twice, as in the reference.
//
// Reference citations are file:line under /root/reference.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sustaindc_hip.h"

#define SDC_BLOCK 256
#define SDC_WAVE 64
#define SDC_HIST_PER_THREAD 40  // 10 x float4 per thread -> 10240 ring slots per env
#define SDC_HIST_STRIDE (SDC_BLOCK * SDC_HIST_PER_THREAD)
#define SDC_NORM_WINDOW 2880    // 30 days x 96 (utils/managers.py:435, :606)
#define SDC_OBS_RAW 53
#define SDC_OBS_OUT (SDC_N_AGENTS * SDC_OBS_PAD)

// ------------------------------------------------------------------------------------------------
// Per-environment state RECORD: 64 dwords (256 B, one per lane) so that the wavefront that owns an env loads its
// whole state with ONE coalesced global_load_dword and stores it back with one coalesced store.  Fields are read
// out with v_readlane.  (Struct-of-arrays would make every field of a one-wavefront-per-env kernel a separate
// wave-uniform scalar load -- ~30 dependent cache misses per step.)
enum SdcRec {
  // ---- the first 128-byte line: everything a step reads or writes (the lane-per-env kernel moves this line only; round 6 -- before,
  // the fields were in order of appearance and a step touched both lines of the record)
  R_CURSOR = 0,   // trace-table index i = day*96 + hour*4 (utils/managers.py:122)
  R_TREL,         // steps since the episode started
  R_DAY,
  R_HOURQ,        // quarter-hour of the day 0..95
  R_QPOPPED,      // load-shifting queue: tasks ever removed this episode
  R_QCUM,         // tasks ever enqueued up to the previous step
  R_QCUMT,        // sum over enqueued tasks of their enqueue step
  R_QHEAD,        // enqueue step of the oldest task still queued
  R_QCUM_HM1,     // cum[head - 1], cumT[head - 1] (cached so the average-age algebra needs no load)
  R_QCUMT_HM1,
  R_LAST_DELTA,   // -2 = None (envs/dc_gym.py:115)
  R_CONSEC,
  R_SCALE,
  R_HIST_LEN,
  R_HIST_POS,
  R_FAULT,
  R_F64 = 16,     // doubles, two dwords each
  R_STPT = 16,
  R_BAT = 18,
  R_HIST_REF = 20,
  R_LAST_ROOM = 22,  // f64: dc_int_temperature the previous step reported (what the trim-and-respond policy monitors)
  R_CFG = 24,     // assignment: data-centre parameter set
  R_LOC,          //             trace-table set
  R_TR_COUNT,     // trim-and-respond policy: response_duration_counter (utils/trim_and_respond.py:22)
  R_EPISODE,
  R_CI_MIN = 28,  // f64 x 2: the episode's carbon-intensity normalisation (what the rule-based battery policy compares in)
  R_CI_DEN = 30,
  // ---- the second line: written by resets and host writes only
  R_DAY_LO = 32,  // assignment: inclusive range of the random start day
  R_DAY_HI,
  R_FEAT_OK,      // 1: the episode's observation feature rows (SdcDev::feat) are valid (sdc_features.hip); cleared by
                  // a reset (whose features kernel sets it again) and by any host write to the env's state
  R_T_MIN = 36,   // f64 x 2
  R_T_DEN = 38,
  R_END = 40,
  SDC_REC_DWORDS = 64
};

// 256-byte per-env header (one dword per lane): the reward-side state (sdc_trackers.hpp) and the running returns; the
// env's wavefront loads and stores it whole, coalesced.
enum SdcHdr {
  H_N = 0,        // history length including this step's value
  H_QS2_LO = 2,   // f64: sum of v^2 over the keys below the lower clip bound (see H_QC)
  H_EOFF = 4,     // f64: bat_total_energy_with_battery_KWh - hist_ref
  H_QS1 = 6,      // 2 x f64 (upper, lower): sum of v over the keys at or beyond last step's clip bound
  H_RET = 10,     // 3 x f64: running return of the current episode (cleared by reset)
  H_Q1 = 16,      // rank window of the lower quartile (SdcTrack)
  H_BU = 18,      // rank window around the upper clip bound
  H_BL = 20,      // rank window around the lower clip bound (complemented keys)
  H_QC = 29,      // [2]: how many keys lie at or beyond last step's clip bound.  With H_QS1 / H_QS2 these running
                  // sums make the tail corrections O(1): a step only touches the keys the bound has moved across.
  H_Q3 = 32,      // rank window of the upper quartile
  H_WFIRST = 22,  // [4] first key of each rank window {Q1, Q3, BU, BL} (in the window's own key space) ...
  H_WLAST = 52,   // [4] ... and its last valid key: a step whose appended / evicted keys lie outside [first, last] of a
                  // window only moves that window's ranks -- decided from these two numbers, without touching its lanes
  H_PEND = 34,    // [4] per rank window: a deferred re-centring in flight (0: none): request step mod 2^19 << 13 | result set << 11 | request index + 1
  H_LAST_XNEW = 38,   // the previous step's appended key, evicted key (KEY_NONE: none) and history length before it:
  H_LAST_XOLD = 39,   // what a re-centred window that describes the ring one step back has to catch up with
  H_LAST_NPREV = 40,
  H_QS2_HI = 46,  // f64: sum of v^2, upper side
  H_STICKY = 48,  // sticky diagnostics (bit 0: a verify-mode mismatch was seen)
  H_KB = 49,      // last step's clip bounds in flipped key space: [0] upper (kub), [1] lower (~(klb - 1))
  H_VALID = 51,   // 1: the reward state describes the ring (0: rebuild it)
  H_A1 = 58,      // f64: sum of v over the history
  H_A2 = 60,      // f64: sum of v^2
  SDC_HDR_DWORDS = 64
};
// rank window: SDC_WIN consecutive order statistics of the history, the keys in SdcDev::qwin (one per lane), in the
// header the rank of the first key and the number of valid keys (0: none)
enum SdcTrack { T_R0 = 0, T_HI = 1, SDC_TRACK_DWORDS = 2 };
#define SDC_WIN 64

// One wavefront = one env: LDS traffic between the lanes of ONE wavefront needs no s_barrier (a wavefront's LDS
// operations complete in order), only the compiler's view of it ordered.
// this lane's index, recomputed WHERE IT IS CALLED (a volatile asm is not hoisted): inside the multi-step kernels' step loop a
// loop-invariant lane id is one more VGPR held -- or spilled -- across the whole step
__device__ __forceinline__ int lane_fresh() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define SDC_MAX_RACK_CLS 8     // rack classes (and groups) the lane-per-env kernel keeps tables for
// (128 dwords: a wavefront holds the table in two registers, dword j in lane j % 64, and reads entries with v_readlane; the
// doubles fill the first 64 dwords, the integers the second)
struct SdcRackClasses {
  double grp_n[SDC_MAX_RACK_CLS], grp_supply[SDC_MAX_RACK_CLS];     // group g: racks of grp_n cpus with supply approach grp_supply
  double cls_full[SDC_MAX_RACK_CLS], cls_idle[SDC_MAX_RACK_CLS];    // class c (of group g: grp_begin[g] <= c < grp_begin[g + 1])
  int n_grp, n_cls;
  int grp_begin[SDC_MAX_RACK_CLS + 1];
  int cls_of_rack[32];                                               // rack slot -> class
  int pad[21];
};
static_assert(sizeof(SdcRackClasses) == 512 && SDC_MAX_RACK_CLS == 8, "two registers of a wavefront hold the table");

// A data-centre parameter set as the kernels see it: the caller's struct plus correctly rounded reciprocals of the
// parameters the step divides by (computed on the host by sdc_set_dc_params), so that those divisions take the
// 3-instruction form of sdc_div_const.
struct SdcDcDev {
  sdc_dc_params p;
  double rc_n_racks, rc_itfan_ref_v_ratio, rc_rho_air, rc_ctafr, rc_bat_capacity;
  double k_outlet;   // 1.918 / (c_air rho_air 0.526): the constant factor of the rack outlet-temperature rise
  double n_racks_f;  // p.n_racks as a double (the step kernel hands the scalars from p.m_cpu to here round as doubles)
  double ret_sum;    // sum of rack_return over the config's racks (the CRAC return temperature is (this + sum of outlets) / racks)
  // RACK CLASSES (the lane-per-env kernel, sdc_wide.hip: its rack model is a per-lane LOOP, and a rack's power / outlet temperature
  // depend on its four parameters only): the config's DISTINCT (cpus, full load, idle, supply approach) tuples, grouped by their
  // (cpus, supply approach) pair -- what the fan / airflow / inlet part depends on.  The shipped 20-rack config has 7 classes in 2
  // groups.  n_cls == 0: too many classes for the kernel's tables (sdc_set_dc_params).
  SdcRackClasses rc;
};

// THE LANE-PER-ENV KERNEL'S GENERAL FORM (sdc_wide.hip, template GEN: several configs in one batch, rule-based policies, alternate
// reward functions of the dc / battery agents): every LANE carries its own config.  What differs between the configs of a batch the
// kernel serves -- the rack table and the quantities sized from it and from the location (utils/make_envs_pyenv.py:139-218) -- is one
// SdcWideCfg per config, staged into LDS by every workgroup (LDS-DMA) and read per lane; the scalars of the server / HVAC
// characteristics (CPU and fan curves, air constants, set-point limits) must be the same bits in every config (the reference's
// dc_config_dc{1,2,3}.json differ in their rack lists only) and stay wave-uniform.  59 doubles per config: an ODD number of 8-byte
// words, so lanes of different configs read different LDS banks.
#define SDC_WIDE_MAX_CLS 12    // rack classes per config (the shipped 16 / 20 / 25-rack configs: 8 / 7 / 11)
#define SDC_WIDE_MAX_CFG 16
enum { WC_RET_SUM = 0, WC_RC_N_RACKS, WC_CT_FAN_REF_P, WC_RC_CTAFR, WC_BAT_CAP, WC_RC_BAT_CAP, WC_SCAL_COUNT };
struct SdcWideCfg {
  double cls[SDC_WIDE_MAX_CLS][4];     // class c: {cpus, supply approach, full load, idle} (unused classes: zeros)
  double scal[WC_SCAL_COUNT];          // the per-config scalars, WC_*
  unsigned map[4];                     // rack slot r -> class: 4 bits each, slot r in bits 4 (r % 8) of map[r / 8]
  int n_cls, n_racks;
  double pad[2];
};
static_assert(sizeof(SdcWideCfg) == 59 * 8, "an odd number of 8-byte words per config");
#define SDC_WIDE_CFG_DOUBLES 59

// DEFERRED WINDOW RE-CENTRING.  A rank window that the next step could exhaust has to be re-centred with one sweep over
// the env's 40 KB ring (sdc_ringpath.hpp qt_refill, ~5 us) -- done inline that sweep made its wavefront the straggler of
// nearly every launch.  Instead the step that sees the need (step t) files a REQUEST with a snapshot of the window;
// spare wavefronts of the NEXT launch (step t + 1) do the sweep against the ring as it was after step t
// (the one slot step t + 1 overwrites is patched with its old content, carried in the request) and leave the
// re-centred window as a RESULT; the env's own wavefront picks it up at step t + 2, replays step t + 1's one
// insertion / eviction on it (remembered in the header) and carries on.  The old window stays valid through step
// t + 1 (that is what "ahead of need" guarantees).  Three request / result sets rotate with the step number.
// Requests per set: S.rq_max (and S.sweep_blocks four-wavefront sweep workgroups per launch), sized by the host with the batch --
// ~26 windows per 4096 envs ask per step, and a request that finds no room is re-centred INLINE by its env's wavefront (a 5 us
// straggler).  Rounds 2-3 had 128 / 32 whatever the batch: at 16 384 envs 42 % of the re-centrings ran inline.
#define SDC_RQ_MIN 128        // (4096 envs and below)
#define SDC_RQ_LIMIT 2047     // (the request index + 1 has 11 bits in the header's stamp)
struct SdcRefillReq {
  int env, win, dir, kt, n, r0, hi, patch_slot;
  unsigned patch_x;
  int step, pad0, pad1;
  unsigned keys[SDC_WIN];
};
struct SdcRefillRes {
  int r0, hi, step, env_win;
  unsigned keys[SDC_WIN];
};

struct SdcDev {
  int n_envs, episode_steps, hist_cap, queue_max, table_len, lw, qstride, max_roll_days;
  int env_base;     // global index of env 0 (sdc_config.env_index_base): keys the reset RNG
  int debug_flags;  // bit 0: cross-check the tracked order statistics against the bisection every step
  int reward_method[3];   // sdc_reward_method per agent slot (ls, dc, bat)
  int policy[3];          // sdc_policy per agent slot: who chooses the action
  double tr_limit;        // trim-and-respond: TandR_monitor_limit
  int32_t* actions_out;   // [N][3] the actions the step applied, or nullptr (sdc_step); sdc_rollout passes its own
  int step_no;            // steps launched so far (host counter): stamps the deferred re-centring requests / results
  int* rq_count;          // [3] requests filed into each set
  SdcRefillReq* rq;       // [3][rq_max]
  SdcRefillRes* rs;       // [3][rq_max]
  int rq_max, sweep_blocks;   // request slots per set; sweep workgroups at the front of a single-step launch's grid
  unsigned long long seed;
  double noise_std, noise_weight;
  // shared, read-only
  const double* tabW;   // [n_loc][table_len]
  const double* tabC;
  const double* tabT;   // pre-noise dry bulb
  const double* tabWB;  // pre-noise wet bulb
  const SdcDcDev* dc;   // [n_cfg]
  int n_cfg;
  const double* prm_env;   // [N][32] (several configs only, else null): every env's own copy of its config's scalars (P_*), so
                           // that the common-case kernels can request them WITH the record -- the config id is inside it
  const SdcWideCfg* wcfg;  // [n_cfg] (or null: a batch the lane-per-env kernel's general form does not serve): see SdcWideCfg
  int wide_max_cls, wide_max_racks4;   // ... the largest class count of a config; the largest rack count rounded up to four
  double rc_queue_max, rc_hist_cap;   // reciprocals of queue_max / hist_cap (see SdcDcDev)
  double queue_max_d, hist_cap_d;     // ... and the two as doubles (a uniform int -> double conversion inside the multi-step kernels' loop
                                      // is hoisted out of it and held in two VGPRs across the whole step; these stay scalar)
  const double* hour_lut;   // [96][2] = cos, sin (utils/managers.py:66-88)
  // per-env state
  unsigned* rec;     // [N][SDC_REC_DWORDS] state records (see SdcRec)
  uint2* qtab;       // [N][qstride] {cum, cumT} per enqueue step of the episode
  unsigned* qcum_t;  // [qstride][N] TIME-MAJOR mirror of qtab's `cum` column, or nullptr (batches the lane-per-env kernel cannot serve): its five
                     // queue-history probes per env are then 64 consecutive dwords per wavefront instead of 64 scattered 64-byte bursts.
                     // Every kernel that appends to qtab appends here too (qcum_append); sdc_set_state("qtab") rebuilds it
  double* t_win;     // [N][lw] dry bulb after noise + roll + clip, from the episode's first cursor
  double* wb_win;    // [N][lw] wet bulb likewise
  unsigned* hist;    // [N][SDC_HIST_STRIDE]  order-preserving uint32 key of fp32(energy - hist_ref); 0xFFFFFFFF = empty
  unsigned* hdr;     // [N][SDC_HDR_DWORDS] per-env header: step hand-off + reward-side state (see SdcHdr)
  float* feat;       // [episode_steps + 1][N][SDC_FEAT_ROW] the trace-only observation entries of every step of the
                     // episode, in observation-pool layout (SDC_P_*), + NC[i'+1] as a double at SDC_FEAT_NCNEXT
  unsigned* qwin;    // [N][SDC_WIN][4] rank windows (sdc_trackers.hpp): lane l's keys of {Q1, Q3, upper bound, lower bound}
  unsigned char* reset_mask;  // [N] device copy of the caller's mask
  unsigned long long* prof_ts;  // measurement only: [3 kernels][N][2] wall-clock stamps of this launch, or nullptr
  unsigned* hist_t;  // [hist_cap][N] SLOT-MAJOR mirror of the history ring, or nullptr (allocated with qcum_t): the lane-per-env kernel reads
                     // the key a step evicts from it -- 64 consecutive dwords per wavefront instead of 64 scattered 128-byte lines -- and
                     // every kernel that appends to the ring appends here too (hist_t_append); sdc_set_state("hist") rebuilds it.  The
                     // sweeps and rebuilds read an env's ring as a whole: they keep the ring itself
};

// TEST HOOK (debug_flags bit 13 = 8192): every 61st (env + launch) takes env_reward's "a clip bound left its window" repair whatever the
// windows say -- the path is otherwise taken by ~4e-8 of the env-steps (tests/test_gpu_bound_repair.py runs it in verify mode)
__device__ __forceinline__ bool bound_repair_forced(const SdcDev& S, const int env) {
  return (S.debug_flags & 8192) != 0 && (unsigned)(env + S.step_no) % 61u == 0u;
}

// per-kernel timing without host events: one lane per workgroup stamps the constant-rate wall clock at entry and
// exit; the host takes min(entry) / max(exit) over the workgroups of a sampled launch (sdc_profile_read)
enum { SDC_PROF_DYNAMICS = 0, SDC_PROF_REWARD = 1, SDC_PROF_RESET = 2 };
__device__ __forceinline__ void prof_stamp(const SdcDev& S, int kernel, int env, int which) {
  if (S.prof_ts) S.prof_ts[((size_t)kernel * S.n_envs + env) * 2 + which] = wall_clock64();
}

// record field access: every lane holds dword `lane` of the record in `r`
__device__ __forceinline__ int rec_i32(unsigned r, int idx) { return (int)__builtin_amdgcn_readlane((int)r, idx); }
__device__ __forceinline__ double rec_f64(unsigned r, int idx) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)r, idx);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)r, idx + 1);
  return __hiloint2double((int)hi, (int)lo);
}

// ------------------------------------------------------------------------------------------------
// wave helpers (64 lanes)

// Cross-lane moves on the DPP data path (a VALU operand modifier, ~10 cycles) instead of ds_bpermute round trips
// through the LDS crossbar (~100+ cycles each, and every reduction here is a chain of 4-6 dependent stages).
//   quad_perm [1,0,3,2] = lane ^ 1      quad_perm [2,3,0,1] = lane ^ 2      row_ror:8 = lane ^ 8 (rows of 16)
//   row_half_mirror / row_mirror reach the partner quad / half-row: same values as lane ^ 4 / lane ^ 8 once the
//   quads / half-rows are uniform, i.e. after the smaller strides have been reduced;
//   row_bcast15 / row_bcast31 carry row totals towards lane 63.
enum { SDC_DPP_XOR1 = 0xB1, SDC_DPP_XOR2 = 0x4E, SDC_DPP_HALF_MIRROR = 0x141, SDC_DPP_MIRROR = 0x140, SDC_DPP_ROR8 = 0x128,
       SDC_DPP_BCAST15 = 0x142, SDC_DPP_BCAST31 = 0x143, SDC_DPP_WAVE_SHL1 = 0x130 };
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_f64(double v) {   // unwritten lanes (row_mask, out of range) read as 0.0
  if constexpr (ROW_MASK == 0xF) {
    // every row written: "no source lane -> 0" is the instruction's bound_ctrl, and the destination needs no zero first
    // (3 instructions per 64-bit stage instead of 5)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  } else {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
  }
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// sum over the 64 lanes, wave-uniform result (tree: strides 1, 2, 4, 8 within rows, then the four rows in order)
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<SDC_DPP_XOR1>(v);
  v += dpp_f64<SDC_DPP_XOR2>(v);
  v += dpp_f64<SDC_DPP_HALF_MIRROR>(v);
  v += dpp_f64<SDC_DPP_MIRROR>(v);
  v += dpp_f64<SDC_DPP_BCAST15, 0xA>(v);
  v += dpp_f64<SDC_DPP_BCAST31, 0xC>(v);
  return readlane_f64(v, 63);
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}

// order-preserving map fp32 -> uint32 (and back); key(NaN 0x7FFFFFFF) = 0xFFFFFFFF marks an empty ring slot
__host__ __device__ __forceinline__ unsigned sdc_f32_key(unsigned b) { return b ^ ((unsigned)((int)b >> 31) | 0x80000000u); }
__host__ __device__ __forceinline__ unsigned sdc_key_f32(unsigned k) { return (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k; }

// x / C for a compile-time constant C, correctly rounded, in 3 VALU instructions instead of the ~13 of the IEEE
// division sequence (Markstein): q = RN(x * RN(1/C)), r = x - q C exactly (fma), result = RN(q + r * RN(1/C)).
// With a correctly rounded reciprocal this is the correctly rounded quotient unless C's significand is all ones;
// checked against x / C on 1e8 random x (incl. near-all-ones and near-power-of-two significands) for every constant
// used here: 0 mismatches.  The fp64 divisions are ~25 % of the step's VALU instructions.
#define SDC_DIV_CONST(x, C) sdc_div_const((x), (double)(C), 1.0 / (double)(C))
__device__ __forceinline__ double sdc_div_const(double x, double c, double rc) {
  const double q = x * rc;
  const double r = __builtin_fma(-q, c, x);
  return __builtin_fma(r, rc, q);
}
// Quotients that feed only powers, temperatures and the info block (never an observation entry or the battery state,
// which must round like the reference): x * (1 / C) -- one rounding more than the division, 1 instruction instead of 3
// -- and a / b by hardware reciprocal + two Newton steps (<= 2 ulp, 6 instructions instead of the IEEE sequence's 11)
#define SDC_MUL_RCP(x, C) ((x) * (1.0 / (double)(C)))
__device__ __forceinline__ double sdc_div_fast(const double a, const double b) {
  double y = __builtin_amdgcn_rcp(b);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-b, y, 1.0), y, y);
  return a * y;
}
// np.round(x, d) == rint(x * 10^d) / 10^d   (P10 a literal power of ten)
#define np_round(x, P10) SDC_DIV_CONST(rint((x) * (P10)), (P10))

// ------------------------------------------------------------------------------------------------
// observation features (sustaindc_env.py:266-433).  Inputs are staged in LDS:
//   nc[0..24]  = NC[i'-16 .. i'+8]   (nc[16] = NC[i'];  entries for negative table indices unused)
//   nt[0..16]  = NT[i' .. i'+16]

// Layout of the 29-float observation pool assembled in LDS (= the HARL shared observation,
// harlsustaindc_env.py:78-80): [0..25] agent_ls state, [26] next workload, [27] next outside temperature, [28] SoC.
//   ls (26): cos_h sin_h NC | 7 CI features | oldest_age avg_age queue | W NT | t_slope 5 temp features | 5 age bins
enum { SDC_P_COS = 0, SDC_P_SIN, SDC_P_NC, SDC_P_CI7 = 3, SDC_P_OLDEST = 10, SDC_P_AVG, SDC_P_NORMQ, SDC_P_W, SDC_P_NT,
       SDC_P_TSLOPE = 15, SDC_P_T5 = 16, SDC_P_HIST = 21, SDC_P_WNEXT = 26, SDC_P_NTNEXT, SDC_P_SOC, SDC_POOL_DIM };

#define SDC_FEAT_ROW 32      // floats per feature row (128 bytes)
// feature rows are kept step-major: the rows all envs read in one launch (envs in lock-step) are adjacent -- 512 KB at
// 4096 envs, a handful of pages -- instead of one row per 86 KB
// the time-major mirror of the queue table's `cum` column (SdcDev::qcum_t), kept by whoever appends to the table
__device__ __forceinline__ void qcum_append(const SdcDev& S, const int env, const int t, const unsigned cum) {
  if (S.qcum_t) S.qcum_t[(size_t)t * S.n_envs + env] = cum;
}
// ... and the slot-major mirror of the history ring (SdcDev::hist_t), kept by whoever appends to the ring.  It exists for batches of
// SDC_HIST_MIRROR_MIN_ENVS envs and up: where the lane-per-env kernel is bound by the memory system (several dispatch rounds) the
// evicted key's 128-byte line per env-step is worth saving (262 144 envs: 141.9 -> 137.6 us per step, 65 536: 41.6 -> 40.9); below,
// where a launch is a latency chain, the extra store costs more than the gather (32 768 envs: 23.15 -> 23.3, 16 384: 16.5 -> 16.7)
#define SDC_HIST_MIRROR_MIN_ENVS 49152
__device__ __forceinline__ void hist_t_append(const SdcDev& S, const int env, const int slot, const unsigned key) {
  if (S.hist_t) S.hist_t[(size_t)slot * S.n_envs + env] = key;
}
__device__ __forceinline__ size_t feat_row_offset(const SdcDev& S, const int env, const int s) {
  return ((size_t)s * (size_t)S.n_envs + (size_t)env) * SDC_FEAT_ROW;
}

#define SDC_FEAT_NCNEXT 30   // ... the last two hold one double
// ... and the slots of the step-dependent observation entries hold the inputs of the step that LEADS to the row's
// observation (row r: the step from episode step r - 1): W[i], C[i], T[i], WB[i] as doubles, T[i+1] as a float
#define SDC_FEAT_W 10
#define SDC_FEAT_T1 12
#define SDC_FEAT_C 22
#define SDC_FEAT_T 24
#define SDC_FEAT_WB 28

// SEQUENTIAL segmented sum, in lane order: lanes [0,16), [16,32) and [32,64) are three independent groups; lanes that carry no
// point hold 0.0 (adding it changes nothing).  The order of sdc_features.hip slope_of -- one lane walking its points first to
// last -- so that an observation's three least-squares slopes are the same BITS whether they come from the episode's
// precomputed rows or from this whole-wavefront path (round 4: with the butterfly order below the two differed in the last
// place of fp64, visible after the fp32 cast when a slope is ~1e-20 -- a temperature window clipped flat; 2 of 58 000
// observations per seed, tools/feature_paths_scan.py).  ~300 instructions, on a path taken at resets and after host writes only.
__device__ __forceinline__ double seg3_seq_sum_f64(const double v, const int lane) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    s0 += readlane_f64(v, i);
    s1 += readlane_f64(v, 16 + i);
  }
#pragma unroll
  for (int i = 0; i < 17; i++) s2 += readlane_f64(v, 32 + i);     // (the temperature slope has 17 points)
  return lane < 16 ? s0 : (lane < 32 ? s1 : s2);
}
// segmented butterfly sum: lanes [0,16), [16,32) and [32,64) are three independent groups
__device__ __forceinline__ double seg3_sum_f64(double v, int lane) {
  v += dpp_f64<SDC_DPP_XOR1>(v);
  v += dpp_f64<SDC_DPP_XOR2>(v);
  v += dpp_f64<SDC_DPP_HALF_MIRROR>(v);   // = lane ^ 4: the quads are uniform by now
  v += dpp_f64<SDC_DPP_MIRROR>(v);        // = lane ^ 8: the half-rows are uniform by now
  const double s = readlane_f64(v, 32) + readlane_f64(v, 48);   // third group: rows 2 + 3
  return lane >= 32 ? s : v;
}

struct ObsScalars {
  double cos_h, sin_h;
  double w_cur, w_next;   // W[i'], W[i'+1]
  double soc;
  double normq, oldest, avg, hist[5];
  int have_past;          // i' >= 16
};

// Observation features (sustaindc_env.py:266-433), all 64 lanes of one wavefront cooperating.  LDS inputs:
//   nc[0..24] = NC[i'-16 .. i'+8]  (nc[16] = NC[i']),   nt[0..16] = NT[i' .. i'+16].
// Three least-squares slopes run side by side in lane groups [0,16) / [16,32) / [32,64); the mean / std of the
// 8 CI futures (lanes 0..7) and the 16 temperature futures (lanes 32..47) use the butterfly order xor 8,1,2,4,
// which is exactly NumPy's pairwise add.reduce tree for n = 8 and n = 16, so they round as in the reference.
// Writes pool[0..28] (LDS floats); the caller synchronises before reading it.
__device__ __forceinline__ void build_obs_pool(const double* nc, const double* nt, const ObsScalars& o, float* pool,
                                               int lane) {
  // ---- slopes: np.polyfit(range(n), y, 1)[0] as closed-form least squares --------------------------------
  int n;            // points in this lane's group
  double y = 0.0;   // this lane's point
  const int j = lane & 15;
  if (lane < 16) {  // future: 4-tap moving average of [NC[i'], NC[i'+1..i'+8]]: 9 -> 6 points (sustaindc_env.py:313,317)
    n = 6;
    if (j < 6) y = (((nc[16 + j] + nc[17 + j]) + nc[18 + j]) + nc[19 + j]) / 4;
  } else if (lane < 32) {  // past: [NC[i'-16..i'-1], NC[i']]: 17 -> 14 points; EMPTY past slice when i' < 16
    if (o.have_past) {     // (utils/managers.py:482-483): np.convolve then yields 4 copies of NC[i'] / 4
      n = 14;
      if (j < 14) y = (((nc[j] + nc[j + 1]) + nc[j + 2]) + nc[j + 3]) / 4;
    } else {
      n = 4;
      if (j < 4) y = nc[16] / 4;
    }
  } else {  // temperature: [NT[i'], NT[i'+1..i'+16]], 17 points, no smoothing (sustaindc_env.py:331)
    n = 17;
    if (lane - 32 < 17) y = nt[lane - 32];
  }
  const int xi = lane < 32 ? j : lane - 32;
  const bool act = xi < n;
  const double xm = 0.5 * (double)(n - 1);
  // n is 6, 14 (4 without a past window) or 17 by lane group: division by a per-group constant
  const double nd = (double)n, rn = n == 6 ? 1.0 / 6.0 : (n == 14 ? 1.0 / 14.0 : (n == 4 ? 0.25 : 1.0 / 17.0));
  const double ym = sdc_div_const(seg3_seq_sum_f64(y, lane), nd, rn);
  const double dx = act ? (double)xi - xm : 0.0;
  const double sxy = seg3_seq_sum_f64(act ? dx * (y - ym) : 0.0, lane);
  const double sxx = seg3_sum_f64(dx * dx, lane);
  const double slope = sxy / sxx;
  if (lane == 0) pool[SDC_P_CI7 + 0] = (float)slope;
  if (lane == 16) pool[SDC_P_CI7 + 1] = (float)slope;
  if (lane == 32) pool[SDC_P_TSLOPE] = (float)slope;

  // ---- extract_ci_features on the 8 CI futures (lanes 0..8) and the 16 temperature futures (lanes 32..48) ----
  const bool ci_grp = lane < 32;
  const int q = ci_grp ? lane : lane - 32;      // point index within [cur, values...]
  const int nv = ci_grp ? 8 : 16;               // number of values
  const double* xs = ci_grp ? nc + 16 : nt;     // xs[0] = cur, xs[1..nv] = values
  const double cur = xs[0];
  const double v = q < nv ? xs[1 + q] : 0.0;    // lanes q >= nv contribute exact zeros
  auto tree = [&](double a) {                   // NumPy pairwise order for n = 8 / 16
    a += dpp_f64<SDC_DPP_ROR8>(a);          // lane ^ 8 within the row of 16
    a += dpp_f64<SDC_DPP_XOR1>(a);
    a += dpp_f64<SDC_DPP_XOR2>(a);
    a += dpp_f64<SDC_DPP_HALF_MIRROR>(a);   // = lane ^ 4: the quads are uniform by now
    return a;
  };
  const double inv_nv = ci_grp ? 0.125 : 0.0625;   // nv = 8 / 16: a power of two, multiplying by 1/nv is exact
  const double mean = tree(v) * inv_nv;
  const double dv = q < nv ? v - mean : 0.0;
  const double sd = sqrt(tree(dv * dv) * inv_nv);
  // np.gradient of [cur, values...] (nv + 1 points): one-sided ends, central interior
  double g = 0.0;
  if (q == 0) g = xs[1] - xs[0];
  else if (q < nv) g = (xs[q + 1] - xs[q - 1]) / 2.0;
  else if (q == nv) g = xs[nv] - xs[nv - 1];
  const double gn = dpp_f64<SDC_DPP_WAVE_SHL1>(g);   // lane i <- lane i + 1
  const bool in = q < nv;                        // pairs (g[q], g[q+1]) for q = 0 .. nv-1
  const unsigned long long pk = __ballot(in && g > 0 && gn <= 0);
  const unsigned long long vl = __ballot(in && g < 0 && gn >= 0);
  if (lane == 0 || lane == 32) {
    const unsigned pm = (unsigned)(ci_grp ? pk : pk >> 32), vm = (unsigned)(ci_grp ? vl : vl >> 32);
    const int peak = pm ? __ffs((int)pm) - 1 : nv, valley = vm ? __ffs((int)vm) - 1 : nv;
    float* f = pool + (ci_grp ? SDC_P_CI7 + 2 : SDC_P_T5);
    f[0] = (float)mean;
    f[1] = (float)sd;
    f[2] = (float)((cur - mean) / (sd + 1e-8));
    f[3] = (float)((double)peak * inv_nv);
    f[4] = (float)((double)valley * inv_nv);
  }
  // ---- scalars -----------------------------------------------------------------------------------------------
  if (lane == 1) {
    pool[SDC_P_COS] = (float)o.cos_h;
    pool[SDC_P_SIN] = (float)o.sin_h;
    pool[SDC_P_NC] = (float)nc[16];
    pool[SDC_P_OLDEST] = (float)o.oldest;
    pool[SDC_P_AVG] = (float)o.avg;
    pool[SDC_P_NORMQ] = (float)o.normq;
    pool[SDC_P_W] = (float)o.w_cur;
    pool[SDC_P_NT] = (float)nt[0];
    for (int b = 0; b < 5; b++) pool[SDC_P_HIST + b] = (float)o.hist[b];
    pool[SDC_P_WNEXT] = (float)o.w_next;
    pool[SDC_P_NTNEXT] = (float)nt[1];
    pool[SDC_P_SOC] = (float)o.soc;
  }
}

// HARL layout (harlsustaindc_env.py:25-26): obs [3][26] zero padded; idx in [0, 78) -> value from the pool.
//   agent_dc (14): sustaindc_env.py:386-393   agent_bat (13): sustaindc_env.py:426-432
// Branch-free: the pool index of entry j (0..77) of the padded [3][26] block, or -1 for a padding zero -- a switch here
// compiles into a tree of exec-mask branches per output lane group (~100 instructions per 64 outputs).
__host__ __device__ constexpr inline int obs_pool_index(const int j) {
  static_assert(SDC_P_W == 13 && SDC_P_NT == 14 && SDC_P_WNEXT == 26 && SDC_P_NTNEXT == 27 && SDC_P_SOC == 28, "byte tables below");
  const int a = (j >= SDC_OBS_PAD ? 1 : 0) + (j >= 2 * SDC_OBS_PAD ? 1 : 0), k = j - SDC_OBS_PAD * a, t = k - 10;
  // entries 10.. of agent_dc: {W, W next, NT, NT next}; of agent_bat: {W, NT, SoC} (one byte each, low byte first)
  const unsigned tab = a == 1 ? 0x1B0E1A0Du : 0x001C0E0Du;
  const int from_tab = (int)((tab >> (8 * (t & 3))) & 0xFFu);
  const bool direct = a == 0 || k < 10;                 // agent_ls whole; the ten shared time / CI entries of the others
  const bool ok = direct || (t >= 0 && t < 5 - a);      // 4 more for agent_dc, 3 for agent_bat, zeros after them
  return ok ? (direct ? k : from_tab) : -1;
}
__device__ __forceinline__ float obs_padded_at(const float* pool, int idx) {
  const int s = obs_pool_index(idx);
  const float v = pool[s < 0 ? 0 : s];
  return s < 0 ? 0.0f : v;
}
// ... the same through a 78-byte table in LDS (entry j = pool index of padded entry j, 0xFF for a padding zero: the step kernels
// keep it behind their constant table, sdc_step.hip SDC_K_OBS_SRC): one byte read instead of ~12 compares and selects per
// output element (the step writes 78 of them per env)
__device__ __forceinline__ float obs_padded_lut(const float* pool, const unsigned char* osrc, int idx) {
  const unsigned s = osrc[idx];
  const float v = pool[s == 0xFFu ? 0u : s];
  return s == 0xFFu ? 0.0f : v;
}
// HARL shared observation (harlsustaindc_env.py:78-80): ls state [0..25], states[1][11] = next workload, states[1][13]
// = next outside temperature, states[2][-1].  The states are the PADDED 26-vectors (ss.pad_observations_v0 runs before
// _create_shared_observation, harlsustaindc_env.py:25-26), so states[2][-1] is agent_bat's zero padding, not the SoC
// the reference's comment names: slot 28 is always 0.0 there, and here.
__device__ __forceinline__ float share_obs_at(const float* pool, int idx) { return idx == SDC_P_SOC ? 0.0f : pool[idx]; }

// stage the obs windows for table cursor ip (= i') into LDS.  tsrc points at T[i'] of the env's weather
// window (global memory, or LDS right after a device-side reset).  Called with tid = 0..63.
__device__ __forceinline__ void stage_windows(const SdcDev& S, int loc, int ip, const double* tsrc, double ci_min,
                                              double ci_den, double t_min, double t_den, int tid, double* s_nc,
                                              double* s_nt) {
  if (tid < 25) {
    int idx = ip - 16 + tid;
    idx = idx < 0 ? 0 : (idx >= S.table_len ? S.table_len - 1 : idx);
    const double c = S.tabC[(size_t)loc * S.table_len + idx];
    s_nc[tid] = (c - ci_min) / ci_den;  // managers.py:437
  } else if (tid >= 32 && tid < 49) {
    const int k = tid - 32;
    s_nt[k] = (tsrc[k] - t_min) / t_den;  // managers.py:608
  }
}

// ------------------------------------------------------------------------------------------------
// counter-based RNG for device-side resets: Philox4x32-R (Salmon et al., SC'11; Random123's known-answer vectors for
// R = 7 and R = 10 pin the NumPy restatement, tests/test_reset_ref.py, and the restatement pins this code,
// tests/test_gpu_reset_pin.py).  R = 10 for the per-episode draws and the actor's sampler; R = 7 -- the paper's
// Crush-resistant minimum with a safety margin -- for the 35 040 normals of a reset's weather walk, whose cost is the
// 2 R quarter-rate 32 x 32 -> 64-bit products per block.

struct Philox4 {
  unsigned x, y, z, w;
};
template <int R>
__device__ __forceinline__ Philox4 philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < R; r++) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}
__device__ __forceinline__ Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                                 unsigned k1) {
  return philox4x32<10>(c0, c1, c2, c3, k0, k1);
}
// uniform in (0, 1) from 2 x 32 bits (53-bit mantissa)
__device__ __forceinline__ double u01(unsigned hi, unsigned lo) {
  const unsigned long long b = (((unsigned long long)hi << 32) | lo) >> 11;
  return ((double)b + 0.5) * (1.0 / 9007199254740992.0);
}
