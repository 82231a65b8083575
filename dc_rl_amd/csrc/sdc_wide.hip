// sdc_wide.hip -- the step kernel for the LARGEST batches: ONE LANE PER ENVIRONMENT, 64 envs = a workgroup of TWO wavefronts.
//
// The pair / quad kernels (sdc_pairstep.hpp) carry an env on 32 / 16 lanes: the rack model runs lane = rack, and every per-env
// scalar instruction -- most of the step -- is issued once per 2 / 4 envs.  That is the right trade while the batch is small
// enough that a launch is a latency chain per wavefront (4096 envs).  At 32 768 envs the quad kernel issues 410 VALU instructions
// per env-step and the SIMDs are the bound.  Here an env is ONE lane: the scalar physics is issued once per 64 envs (57 VALU
// instructions per env-step, counters), the rack model runs once per distinct rack (its parameters wave-uniform: scalar operands),
// and nothing in the dynamics crosses lanes.
//
// A lone wavefront runs its instruction stream at ~6 cycles per instruction whatever the dependences, so a workgroup's time is the
// number of instructions on its critical path: wavefront 0 (DYNAMICS) computes the step's energy and writes observations and state,
// wavefront 1 (REWARDS) meanwhile reads the reward-side state, serves arriving re-centred windows, applies the step's EVICTION
// to the rank windows (the ring key the step overwrites is known from the start), finds the oldest queued task for the other
// wavefront -- and does the insertion, the moments and the rewards once the energy is handed over.  Three barriers.
//
// The arithmetic is pair_dynamics' / pair_reward_fast's, expression for expression in the same order (the rack sums in the
// half-wave reduction's tree order), so the outputs and the state are the same bits whichever kernel steps a batch.
//
// Reward normalisation (utils/reward_creator.py:16-45): the O(1) path per lane -- the rank windows' first / last keys and ranks
// from the header, the few keys around the wanted ranks gathered from the windows in memory (sdc_trackers.hpp: a window lists 64
// of the 10 000 keys, a step's keys land inside one on ~5 % of the env-steps).  What needs a window's 64 keys side by side -- a key
// landing inside it, a re-centred window arriving, a re-centring request to file -- is done by the WHOLE wavefront for that one
// env (lane = key, the sdc_trackers.hpp primitives), all such windows of a phase fetched together; anything unusual falls back to
// env_reward() (sdc_pairstep.hpp).
//
// The host picks this kernel for a lock-step batch of a multiple of 64 envs from SDC_WIDE_MIN_ENVS: its COMMON-CASE form (one
// config, the caller's actions, default rewards: sdc_capi.hip wide_case) or its GENERAL form (template GEN, wide_gen_case): every
// lane carries its own config (SdcWideCfg, staged in LDS: BASELINE configs[3]'s 16 / 20 / 25-rack mixes), the rule-based policies of
// utils/rbc_agents.py:3-47, utils/trim_and_respond.py:8-38 and utils/base_agents.py choose actions inside the step, the dc / battery
// agents take any of utils/reward_creator.py:154-334.  Same expressions either way: the same bits as the pair kernels.
// Reference: sustaindc_env.py:533-737 (per-block citations: sdc_pairstep.hpp).
// Measurements behind every choice: profiles/r5_wide_experiments.txt, r5_wide_timeline.txt.
#include <type_traits>
#include "sdc_pairstep.hpp"
#include "sdc_sweep.hpp"

namespace {

constexpr int WE = SDC_WAVE;     // envs per wavefront: lane = env

// (a value the optimiser may not look through: a chain of selects between elements of a small local array is otherwise merged into
// ONE load with a computed index -- and the array then lives in scratch memory)
__device__ __forceinline__ unsigned opq(unsigned x) {
  asm volatile("" : "+v"(x));
  return x;
}
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4v nt_load4(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p)); }
__device__ __forceinline__ void nt_store4(float* p, const f4v v) { __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(p)); }

// LDS of an env workgroup (64 envs, two wavefronts), 40 KB -- four workgroups per CU, two wavefronts per SIMD:
//   rec  16 KB  the 64 state records (dynamics wavefront), from the first loads until the patched block has gone back to memory (12 of
//               every record's 16 chunks hold data: two of the others carry the reward wavefront's hand-BACK); then its output
//               staging: the obs rows (20 KB, running into `row`), the share_obs rows, the info rows
//   row   8 KB  the 64 feature rows on the way in; during the rack model the rack classes' results {power, outlet} per lane; from the
//               end of the dynamics its upper half holds the HAND-OVER to the reward wavefront (WideHand)
//   hdr  16 KB  the 64 headers (reward wavefront), likewise: in its registers from its first read to the commit, the LDS meanwhile the
//               landing place of the window-update tasks' windows; behind the commit the whole-wavefront fallback's scratch (WideLate)
// The GENERAL form (GEN): `row` holds the results of up to 12 rack classes (12 KB), and behind the three blocks sits the batch's
// config table (SdcWideCfg x n_cfg <= 16: 7.4 KB) -- 51.4 KB, three workgroups per CU (49 152 envs in one dispatch round).
template <bool GEN>
struct WideSharedT {
  unsigned rec[WE * 64];
  unsigned row[WE * 32];
  unsigned hdr[WE * 64];
};
template <>
struct WideSharedT<true> {
  unsigned rec[WE * 64];
  unsigned row[WE * 4 * SDC_WIDE_MAX_CLS];
  unsigned hdr[WE * 64];
  double cfg[SDC_WIDE_MAX_CFG * SDC_WIDE_CFG_DOUBLES];
};
static_assert(sizeof(WideSharedT<true>) * 3 <= 160 * 1024 && sizeof(WideSharedT<false>) * 4 == 160 * 1024, "workgroups per CU");
static_assert(sizeof(float) * WE * SDC_OBS_OUT <= sizeof(unsigned) * (WE * 64 + WE * 16), "the obs rows are staged across `rec` and the lower half of `row`");
// what the dynamics hand to the reward part, per env (lane): [field][lane]
struct WideHand {
  double e_off[WE], energy[WE], norm_ci[WE];
  unsigned x_new[WE];
  int hl[WE], slot[WE];
};
// ... and what the reward wavefront hands BACK -- the oldest queued task's step, the cached prefix counts before it, the two normalised
// ages -- sits in two chunks of the lane's record image that hold no record data (the records' padding is neither loaded nor stored):
// written before barrier 2, read by the dynamics wavefront right behind it, rewritten by nobody
static_assert(sizeof(WideHand) <= sizeof(unsigned) * WE * 16, "the hand-over sits in the upper half of `row`");
// ... the general form's extension, behind it (its `row` is 12 KB)
struct WideHandGen {
  double p_it[WE], total_kw[WE], water[WE];
};
static_assert(sizeof(WideHandGen) <= sizeof(unsigned) * WE * 16 && SDC_WIDE_MAX_CLS >= 12, "dwords 32 WE .. 48 WE of the general form's `row`");
// (measurement build -DSDC_WIDE_STAMPS: lane 0 of both wavefronts stamps the wall clock (100 MHz) at the marks WST(i); the reward wavefront
// leaves them in columns 0..23 of its first env's info row -- tools/dev/wide_timeline.py, wide_entry.py, wide_tail.py; stamp 16 = the
// workgroup's first instruction, 18 / 17 / 19 = window keys requested / arriving windows served / the oldest task found)
#ifdef SDC_WIDE_STAMPS
#define WST(i) do { if (lane == 0) reinterpret_cast<unsigned long long*>(sh.row + WE * 16 + 896)[i] = wall_clock64(); } while (0)
// (the first stamps are taken while the feature rows are still streaming into `row`: kept in a register, written later)
#define WST_HOLD(v) const unsigned long long v = wall_clock64()
#define WST_PUT(i, v) do { if (lane == 0) reinterpret_cast<unsigned long long*>(sh.row + WE * 16 + 896)[i] = (v); } while (0)
#else
#define WST(i)
#define WST_HOLD(v)
#define WST_PUT(i, v)
#endif
static_assert(sizeof(WideHand) <= 896 * 4, "the stamps sit behind the hand-over");
struct WideLate {                  // (aliases a 16 KB block that has gone back to memory; the sweep wavefronts: from the start)
  sdc_rw::TailLds tl;              // scratch of the ring paths (env_reward: window refill, rebuild; the sweep wavefronts)
  float back[WE][8];               // what a whole-wavefront reward step hands back to its env's lane {z, path, ret[3]}
};
static_assert(sizeof(WideLate) <= sizeof(unsigned) * WE * 64, "");

// ---- BLOCK I/O: an env's 256-byte record / header as 64 x 4-byte accesses per lane would be 64 L2 requests per instruction (every
// lane another line: measured 2.6 M requests per launch at 32 768 envs, the kernel's bound).  The wavefront's 64 records ARE one
// contiguous 16 KB block: it comes in with 16 whole-line LDS-DMA loads (global_load_lds_dwordx4: no staging registers), each lane
// reads its env's 16-byte chunks from LDS, patches them, and the block goes back out with 16 whole-line stores.
// LDS image: chunk c of record e sits in 16-byte slot e * CPR + ((c + e) mod CPR) -- the DMA writes LDS linearly (base + lane x 16),
// so the rotation is applied to the SOURCE address; with it the lanes' reads of "their chunk c" spread over all banks.
typedef __attribute__((address_space(1))) const void* sdc_gptr;
typedef __attribute__((address_space(3))) void* sdc_lptr;
template <int CPR, int NC = CPR>     // 16-byte chunks per record: 16 (256-byte records / headers), 8 (128-byte feature rows); the first NC of them
__device__ __forceinline__ void block_load(const void* gbase, unsigned* lds, const int lane) {
  static_assert(CPR == 16 || CPR == 8, "");
#pragma unroll
  for (int k = 0; k < CPR; k++) {
    const int sl = k * WE + lane;
    const int e = sl / CPR, c = ((sl % CPR) - e) & (CPR - 1);
    const char* g = reinterpret_cast<const char*>(gbase) + (size_t)(e * CPR + c) * 16;
    if (NC == CPR || c < NC) __builtin_amdgcn_global_load_lds((sdc_gptr)g, (sdc_lptr)(lds + k * WE * 4), 16, 0, 0);
  }
}
// A state record's FIRST 128-byte line (chunks 0..7) holds everything a step reads or writes (sdc_device.hpp SdcRec): that line comes
// in; the chunks the step changes go back -- 0..5 (96 bytes: cursor .. last room temperature), the general form 0..7 (its
// trim-and-respond counter sits in chunk 6).  The record's second line is the reset's.  Chunks 12, 13 of the LDS image: the hand-back.
#define WIDE_REC_LOAD 8
#define WIDE_REC_CHUNKS 12
// window-update tasks of a step: their windows in the header block's LDS (the headers are in registers by then: wide_rewards)
#define WIDE_TASK_OFF 512
#define WIDE_TASK_SLOTS 8
#ifndef WIDE_ROOMY_WGS
#define WIDE_ROOMY_WGS SDC_CUS      // env workgroups up to which the sweeps run BELOW the env wavefronts (one workgroup per CU: see the kernel)
#endif
static_assert(R_CI_DEN + 2 == WIDE_REC_LOAD * 4 && WIDE_REC_LOAD <= WIDE_REC_CHUNKS && WIDE_REC_CHUNKS + 2 <= 16, "the record's first line, then two chunks for the hand-back");
template <int CPR>
__device__ __forceinline__ uint4 block_get(const unsigned* lds, const int e, const int c) {
  return reinterpret_cast<const uint4*>(lds)[e * CPR + ((c + e) & (CPR - 1))];
}
template <int CPR>
__device__ __forceinline__ void block_put(unsigned* lds, const int e, const int c, const uint4 v) {
  reinterpret_cast<uint4*>(lds)[e * CPR + ((c + e) & (CPR - 1))] = v;
}
// the block back to memory in whole lines; records whose bit is set in `skip` are left alone (their env's state was written by
// the whole-wavefront fallback)
typedef unsigned u4v __attribute__((ext_vector_type(4)));
template <int NC = 16, bool NT = false>      // (the first NC chunks of every record; NT: non-temporal)
__device__ __forceinline__ void block_store16(void* gbase, const unsigned* lds, const int lane, const unsigned long long skip) {
  // all sixteen LDS reads in flight together (a read inside each store's condition would be waited for one at a time)
  uint4 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = reinterpret_cast<const uint4*>(lds)[k * WE + lane];
  // slot k * 64 + lane holds chunk c of record e = 4 k + lane / 16; e's skip bit is bit 4 k of `skip >> (lane / 16)`
  const int e0 = lane >> 4, l15 = lane & 15;
  const unsigned long long s = skip >> e0;
  uint4* const g = reinterpret_cast<uint4*>(gbase);
  if (skip == 0ull) {      // (wave-uniform: the usual case)
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int e = 4 * k + e0, c = (l15 - e) & 15;
      if (NC == 16 || c < NC) {
        if (NT) {
          const u4v t = {v[k].x, v[k].y, v[k].z, v[k].w};
          __builtin_nontemporal_store(t, reinterpret_cast<u4v*>(g + e * 16 + c));
        } else {
          g[e * 16 + c] = v[k];
        }
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int e = 4 * k + e0, c = (l15 - e) & 15;
      if (!((s >> (4 * k)) & 1ull) && (NC == 16 || c < NC)) g[e * 16 + c] = v[k];
    }
  }
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// pairwise (binary-tree) sum over rack slots 0..31 in slot order, streamed four slots at a time; total() when done.  The tree
// is half_sum_f64's (strides 1, 2, 4, 8 inside the rows, then the two rows), so the sums round as in the other mappings;
// slots without a rack hold zeros there, which change nothing here.
struct TreeSum32 {
  double a4, a8, a16, a32;
  int k;
  __device__ __forceinline__ void init() { k = 0; a4 = a8 = a16 = a32 = 0.0; }
  // four slots at once (k a multiple of 4, wave-uniform: the branches below are scalar): their two bottom levels are plain adds,
  // (v0 + v1) + (v2 + v3).  Level a_m holds the sum of a complete group of m slots that still waits for its right-hand neighbour.
  __device__ __forceinline__ void push4(const double v0, const double v1, const double v2, const double v3) {
    double v = (v0 + v1) + (v2 + v3);
    if (k & 4) {
      v = a4 + v;
      if (k & 8) {
        v = a8 + v;
        if (k & 16) a32 = a16 + v;      // (slots 0..31: the two rows of the half-wave reduction)
        else a16 = v;
      } else {
        a8 = v;
      }
    } else {
      a4 = v;
    }
    k += 4;
  }
  // the tree's upper levels: a count that is not a power of two leaves partial sums on several levels (lower level = later slots)
  __device__ __forceinline__ double total() const {
    if (k & 32) return a32;
    double t = 0.0;
    bool have = false;
    if (k & 4) { t = a4; have = true; }
    if (k & 8) { t = have ? a8 + t : a8; have = true; }
    if (k & 16) { t = have ? a16 + t : a16; }
    return t;
  }
};

// ---- load shifting (sdc_physics.hpp ls_algebra: envs/carbon_ls.py:172-324).  Both wavefronts of an env workgroup evaluate it (integer
// algebra on the step's workload, the action and five queue probes): the dynamics wavefront for the utilisation, the reward wavefront
// for the oldest-task search -- which needs a second, dependent memory round trip (the queue table ahead of the oldest task) that would
// otherwise sit on the path to the step's energy.
constexpr int WIDE_QA = 16;        // table entries ahead of the oldest task's step requested up front (two per dwordx4)

// ---- the DYNAMICS wavefront of an env workgroup: lane = env ---------------------------------------------------------------------------
template <bool GEN>
__device__ __forceinline__ void wide_dynamics(const SdcDev& S, WideSharedT<GEN>& sh, const int env0, const int lane, const int rel_hint,
                                              const int32_t* __restrict__ actions, float* __restrict__ obs, float* __restrict__ share_obs,
                                              unsigned char* __restrict__ done, float* __restrict__ final_obs,
                                              const unsigned long long wst_top = 0ull) {
  using namespace sdc_rw;
  const int env = env0 + lane;
  KLit kt{};
  WST_HOLD(wst0);
  // ---- loads: actions, state record, feature row, queue-history probes ----------------------------------------------------
  int a_ls = 1, a_dc = 1, a_bat = 2;       // (GEN: rule-based slots never read the caller's array, which may be null)
  if (!GEN || actions) {
    const int32_t* ap = actions + (size_t)env * 3;
    const int t0 = ap[0], t1 = ap[1], t2 = ap[2];
    if (!GEN || S.policy[0] == SDC_POLICY_EXTERNAL) a_ls = t0;
    if (!GEN || S.policy[1] == SDC_POLICY_EXTERNAL) a_dc = t1;
    if (!GEN || S.policy[2] == SDC_POLICY_EXTERNAL) a_bat = t2;
  }
  // the config's scalars (P_*: double j in lane j) and its rack-class table (dword j in lane j % 64 of two registers), requested with
  // the first loads and read with v_readlane where they are used: wave-uniform operands, no load in the middle of the step
  const SdcDcDev& D = S.dc[0];
  const double prm_l = lane < P_COUNT ? reinterpret_cast<const double*>(&D.p.m_cpu)[lane] : 0.0;
  const unsigned tab0 = reinterpret_cast<const unsigned*>(&D.rc)[lane], tab1 = reinterpret_cast<const unsigned*>(&D.rc)[lane + 64];
  auto PRM = [&](const int j) { return readlane_f64(prm_l, j); };
  // the wavefront's 64 records and 64 feature rows: two contiguous blocks, in through the LDS (block I/O above)
  block_load<16, WIDE_REC_LOAD>(S.rec + (size_t)env0 * SDC_REC_DWORDS, sh.rec, lane);
  block_load<8>(S.feat + feat_row_offset(S, env0, rel_hint + 1), sh.row, lane);
  if constexpr (GEN) {
    // the batch's config table (SdcWideCfg x n_cfg, L2-resident) into LDS: read per lane, by the lane's config id, from barrier 1 on
    const int bytes = S.n_cfg * (int)sizeof(SdcWideCfg);
#pragma unroll 1
    for (int o = 0; o < bytes; o += WE * 16)
      if (o + lane * 16 < bytes)
        __builtin_amdgcn_global_load_lds((sdc_gptr)(reinterpret_cast<const char*>(S.wcfg) + o + lane * 16),
                                         (sdc_lptr)(reinterpret_cast<char*>(sh.cfg) + o), 16, 0, 0);
  }
  static_assert(R_CURSOR == 0 && R_TREL == 1 && R_DAY == 2 && R_HOURQ == 3 && R_QPOPPED == 4 && R_QCUM == 5 && R_QCUMT == 6 &&
                R_QHEAD == 7 && R_QCUM_HM1 == 8 && R_QCUMT_HM1 == 9 && R_LAST_DELTA == 10 && R_CONSEC == 11 && R_SCALE == 12 &&
                R_HIST_LEN == 13 && R_HIST_POS == 14 && R_FAULT == 15 && R_STPT == 16 && R_BAT == 18 && R_HIST_REF == 20 &&
                R_LAST_ROOM == 22, "record layout the chunk reads assume");
  int cumq[5];     // cum[now - 97], cum[now - 24], cum[now - 48], cum[now - 72], cum[now - 96] (0 before the episode's start)
#pragma unroll
  for (int s = 0; s < 5; s++) {
    const int back = s == 0 ? 97 : 24 * s;
    const int t = rel_hint - back;
    cumq[s] = t >= 0 ? (int)S.qcum_t[(size_t)t * S.n_envs + env] : 0;      // (the table's time-major mirror: 64 consecutive dwords)
  }

  WST_HOLD(wst1);
  dma_wait();
  __syncthreads();      // (1) records, feature rows and headers are in LDS
  const uint4 r0 = block_get<16>(sh.rec, lane, 0), r1 = block_get<16>(sh.rec, lane, 1), r2 = block_get<16>(sh.rec, lane, 2);
  const uint4 r3 = block_get<16>(sh.rec, lane, 3), r4 = block_get<16>(sh.rec, lane, 4), r5 = block_get<16>(sh.rec, lane, 5);
  // GEN: config, location, trim-and-respond counter (chunk 6), the carbon-intensity normalisation (7)
  uint4 r6 = make_uint4(0u, 0u, 0u, 0u), r7 = r6;
  if constexpr (GEN) {
    r6 = block_get<16>(sh.rec, lane, 6);
    r7 = block_get<16>(sh.rec, lane, 7);
  }
  static_assert(R_CFG == 24 && R_LOC == 25 && R_TR_COUNT == 26 && R_CI_MIN == 28 && R_CI_DEN == 30, "record layout the GEN chunk reads assume");
  float row[SDC_FEAT_ROW];
#pragma unroll
  for (int q = 0; q < SDC_FEAT_ROW / 4; q++) {
    const uint4 v = block_get<8>(sh.row, lane, q);
    row[4 * q] = __uint_as_float(v.x); row[4 * q + 1] = __uint_as_float(v.y); row[4 * q + 2] = __uint_as_float(v.z); row[4 * q + 3] = __uint_as_float(v.w);
  }
  wave_sync();
  const int i = (int)r0.x, rel = (int)r0.y, day = (int)r0.z, hourq = (int)r0.w;
  const int popped0 = (int)r1.x, cum_prev = (int)r1.y;
  const unsigned cumT_prev = r1.z;
  int last_delta = (int)r2.z, consecutive = (int)r2.w, scale = (int)r3.x;
  int hl = (int)r3.y, hpos = (int)r3.z;
  const unsigned fault0 = r3.w;
  const double stpt0 = __hiloint2double((int)r4.y, (int)r4.x);
  double bat_load = __hiloint2double((int)r4.w, (int)r4.z);
  const double href0 = __hiloint2double((int)r5.y, (int)r5.x);
  // GEN: this lane's config (SdcWideCfg in LDS: 59 doubles per config -- an odd stride, lanes of different configs read different banks)
  const double* wc = nullptr;
  if constexpr (GEN) wc = &sh.cfg[0] + (int)r6.x * SDC_WIDE_CFG_DOUBLES;
  constexpr int WC_SCAL = 4 * SDC_WIDE_MAX_CLS, WC_MAP = WC_SCAL + WC_SCAL_COUNT;
  static_assert(offsetof(SdcWideCfg, scal) == 8 * WC_SCAL && offsetof(SdcWideCfg, map) == 8 * WC_MAP && offsetof(SdcWideCfg, n_cls) == 8 * (WC_MAP + 2), "");
  // a per-config scalar: the lane's own (GEN) or config 0's, wave-uniform
  auto PCFG = [&](const int wc_j, const int p_j) __attribute__((always_inline)) {
    if constexpr (GEN) return wc[WC_SCAL + wc_j];
    else return PRM(p_j);
  };

  unsigned fault = 0;
  if (i + 9 > S.table_len - 1) fault |= SDC_FAULT_TABLE_RANGE;
  if ((unsigned)a_ls > 2u || (unsigned)a_dc > 2u || (unsigned)a_bat > 2u) {
    fault |= SDC_FAULT_ACTION;
    if ((unsigned)a_ls > 2u) a_ls = 1;
    if ((unsigned)a_dc > 2u) a_dc = 1;
    if ((unsigned)a_bat > 2u) a_bat = 2;
  }
  static_assert(SDC_FEAT_W == 10 && SDC_FEAT_T1 == 12 && SDC_FEAT_C == 22 && SDC_FEAT_T == 24 && SDC_FEAT_WB == 28 &&
                SDC_FEAT_NCNEXT == 30, "feature-row slots of the step's inputs");
  auto row_f64 = [&](const int j) { return __hiloint2double(__float_as_int(row[j + 1]), __float_as_int(row[j])); };
  const double wl = row_f64(SDC_FEAT_W), ci_i = row_f64(SDC_FEAT_C), amb = row_f64(SDC_FEAT_T), wet_bulb = row_f64(SDC_FEAT_WB);
  const double norm_ci = row_f64(SDC_FEAT_NCNEXT);
  const double amb_next = (double)row[SDC_FEAT_T1];
  const double hour = (double)hourq * 0.25;

  // ---- load shifting: envs/carbon_ls.py:172-324 (pair_dynamics, same expressions) ---------------------------------------------
  if (wl < 0 || wl > 1) fault |= SDC_FAULT_WORKLOAD;
  const int now = rel;
  const LsStep ls = ls_algebra(kt, wl, a_ls, popped0, cum_prev, cumT_prev, now, S.queue_max, cumq[0], cumq[1], cumq[2], cumq[3], cumq[4]);
  const int overdue = ls.overdue, popped = ls.popped, dropped = ls.dropped, processed = ls.processed;
  const int cum_now = ls.cum_now, total = ls.total;
  const unsigned cumT_now = ls.cumT_now;
  const double util = ls_utilisation(kt, ls);
  double hist[5];
  const double den = (double)max(total, 1), rden = 1.0 / den;
  ls_age_hist(ls, den, rden, hist);
  const double normq = sdc_div_const((double)total, S.queue_max_d, S.rc_queue_max);
  // (the oldest task's step, the cached prefix counts before it and the two ages: from the reward wavefront, behind barrier 2)

  WST_PUT(0, wst0);
  WST_PUT(1, wst1);
  WST_PUT(16, wst_top);      // (the workgroup's first instruction, before it has read a kernel argument)
  WST(2);
  // ---- GEN: rule-based policies choose the dc / battery actions (pair_dynamics, same expressions) ------------------------------------
  int tr_count = (int)r6.z;
  if constexpr (GEN) {
    // (the previous step's room temperature: dc_int_temperature; the carbon intensity three steps ahead from the location's table)
    if (S.policy[1] == SDC_POLICY_TRIM_AND_RESPOND) a_dc = trim_and_respond_action(S.tr_limit, __hiloint2double((int)r5.w, (int)r5.z), tr_count);
    if (S.policy[2] == SDC_POLICY_RBC) {
      const int i3 = min(max(i + 3, 0), S.table_len - 1);
      const double c3 = S.tabC[(size_t)(int)r6.y * S.table_len + i3];
      a_bat = rbc_battery_action(c3, ci_i, __hiloint2double((int)r7.y, (int)r7.x), __hiloint2double((int)r7.w, (int)r7.z));
    }
  }

  // ---- CRAC set-point integrator: envs/dc_gym.py:160-174 --------------------------------------------------------------------
  if (util < 0.0 || util > 1.0) fault |= SDC_FAULT_CPU_LOAD;
  const int delta = a_dc - 1;
  const double stpt = setpoint_step(a_dc, last_delta, consecutive, scale, stpt0, PRM(P_MAX_TEMP), PRM(P_MIN_TEMP));

  // ---- rack model: envs/datacenter.py:250-317, :157-181 -------------------------------------------------------------------------
  // A rack's power and outlet temperature depend on its four parameters (and the env's set-point and load) only, and configs
  // repeat them: the model is evaluated once per rack CLASS (SdcRackClasses: 7 classes in 2 groups for the shipped 20 racks) --
  // what depends on (cpus, supply approach) once per group -- the results parked per lane in LDS, and the racks' sums then take
  // every slot's class value in slot order (the half-wave reduction's tree).  Same expressions on the same inputs as a pass over
  // all racks: the same bits.
  const double load_pct = util * 100;
  const RackEnv E = {PRM(P_M_CPU), PRM(P_C_CPU), PRM(P_M_FAN), PRM(P_C_FAN), PRM(P_RS_CPU) * KDIV(load_pct, 100), PRM(P_RS_FAN) * KDIV(load_pct, 20),
                     PRM(P_ITFAN_REF_P), PRM(P_RC_ITFAN_REF_V_RATIO), PRM(P_IT_FAN_FULL_LOAD_V), PRM(P_K_OUTLET)};
  bool bad_delta = false;
  TreeSum32 s_out, s_pw;
  s_out.init();
  s_pw.init();
  if constexpr (GEN) {
    // every lane its own config: class j of the lane's config (its four parameters from the config table in LDS), for j up to the
    // largest class count of the batch's configs; then the lane's racks in slot order, the class of slot r from the config's map
    const uint2 ncr = *reinterpret_cast<const uint2*>(wc + WC_MAP + 2);
    const int n_cls = (int)ncr.x, R = (int)ncr.y;
    double* cls_pw = reinterpret_cast<double*>(sh.row);                 // [class][lane]
    double* cls_out = cls_pw + SDC_WIDE_MAX_CLS * WE;
    const int max_cls = S.wide_max_cls;
#pragma unroll 2
    for (int c = 0; c < max_cls; c++) {
      double inlet;
      const RackOut ro = rack_point(kt, E, wc[4 * c], wc[4 * c + 1], wc[4 * c + 2], wc[4 * c + 3], stpt, inlet);
      if (c < n_cls && (ro.out - inlet < 2 || !ro.plain)) bad_delta = true;
      cls_pw[c * WE + lane] = ro.pc + ro.pf;
      cls_out[c * WE + lane] = ro.out;
    }
    wave_sync();
    const unsigned* wmap = reinterpret_cast<const unsigned*>(wc + WC_MAP);
    const int r_end = S.wide_max_racks4;
#pragma unroll 2
    for (int rk = 0; rk < r_end; rk += 4) {
      const unsigned mw = wmap[rk >> 3] >> (4 * (rk & 7));      // (slots rk .. rk + 3: four nibbles of one map word)
      double vp[4], vo[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool in = rk + j < R;      // (a slot beyond the lane's last rack adds the 0.0 the other mappings' idle lanes add)
        const int c = (int)((mw >> (4 * j)) & 15u);
        vp[j] = in ? cls_pw[c * WE + lane] : 0.0;
        vo[j] = in ? cls_out[c * WE + lane] : 0.0;
      }
      s_pw.push4(vp[0], vp[1], vp[2], vp[3]);
      s_out.push4(vo[0], vo[1], vo[2], vo[3]);
    }
  } else {
    const int R = (int)PRM(P_N_RACKS);
    // the class table: dword j in lane j % 64 of two registers, entries read with v_readlane (wave-uniform operands)
    auto tab_f64 = [&](const int j) { return __hiloint2double((int)lane_key(tab0, 2 * j + 1), (int)lane_key(tab0, 2 * j)); };   // double j of the first 32
    auto tab_i32 = [&](const int j) { return (int)lane_key(tab1, j); };                                                        // int j of the second half
    static_assert(offsetof(SdcRackClasses, grp_n) == 0 && offsetof(SdcRackClasses, grp_supply) == 64 && offsetof(SdcRackClasses, cls_full) == 128 &&
                  offsetof(SdcRackClasses, cls_idle) == 192 && offsetof(SdcRackClasses, n_grp) == 256 && offsetof(SdcRackClasses, grp_begin) == 264 &&
                  offsetof(SdcRackClasses, cls_of_rack) == 300, "table entries by index");
    const int n_grp = tab_i32(0);
    double* cls_pw = reinterpret_cast<double*>(sh.row);                 // [class][lane]
    double* cls_out = cls_pw + SDC_MAX_RACK_CLS * WE;
#pragma unroll 1
    for (int g = 0; g < n_grp; g++) {
      // what depends on (number of CPUs, supply approach) once per group (sdc_physics.hpp rack_air), the airflow's logarithm with it
      const double r_n = tab_f64(g);
      const RackAir ra = rack_air(kt, E, r_n, tab_f64(8 + g), stpt);
      const bool plain_v = rack_plain(kt, ra.vtot);
      const double l2v = log2_pos_normal(plain_v ? ra.vtot : 1.0, kt);
      const int c0 = tab_i32(2 + g), c1 = tab_i32(3 + g);
#pragma unroll 2
      for (int c = c0; c < c1; c++) {
        const double pw = rack_cpu_power(ra, r_n, tab_f64(16 + c), tab_f64(24 + c)) + ra.pf;
        const bool plain = rack_plain(kt, pw) && plain_v;
        // (log2 of 1.0 is exactly 0.0: what rack_point evaluates for the airflow when the power is not a plain number)
        const double out = rack_outlet(kt, E, ra.inlet, log2_pos_normal(plain ? pw : 1.0, kt), plain ? l2v : 0.0);
        if (out - ra.inlet < 2 || !plain) bad_delta = true;
        cls_pw[c * WE + lane] = pw;
        cls_out[c * WE + lane] = out;
      }
    }
    wave_sync();
    // (four slots per trip: the values come from LDS together, the tree's two bottom levels need no bookkeeping; a slot beyond the
    // last rack adds the 0.0 the other mappings' idle lanes add)
    int rk = 0;
#pragma unroll 2
    for (; rk + 4 <= R; rk += 4) {      // (whole trips: no slot to blank)
      double vp[4], vo[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int c = tab_i32(11 + rk + j);
        vp[j] = cls_pw[c * WE + lane];
        vo[j] = cls_out[c * WE + lane];
      }
      s_pw.push4(vp[0], vp[1], vp[2], vp[3]);
      s_out.push4(vo[0], vo[1], vo[2], vo[3]);
    }
    if (rk < R) {
      double vp[4], vo[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool in = rk + j < R;
        const int c = in ? tab_i32(11 + rk + j) : 0;
        vp[j] = in ? cls_pw[c * WE + lane] : 0.0;
        vo[j] = in ? cls_out[c * WE + lane] : 0.0;
      }
      s_pw.push4(vp[0], vp[1], vp[2], vp[3]);
      s_out.push4(vo[0], vo[1], vo[2], vo[3]);
    }
  }
  WST(3);
  if (bad_delta) fault |= SDC_FAULT_OUTLET_DELTA;
  const double sum_outlet = s_out.total();
  const double p_it = s_pw.total();
  const double rc_n_racks = PCFG(WC_RC_N_RACKS, P_RC_N_RACKS);
  const double avg_ret = (PCFG(WC_RET_SUM, P_RET_SUM) + sum_outlet) * rc_n_racks;
  const double mean_outlet = sum_outlet * rc_n_racks;

  // ---- HVAC: envs/datacenter.py:432-474 ; water :325-353 (sdc_physics.hpp hvac_water) ------------------------------------------------------
  const HvacPrm HP = {PRM(P_C_AIR), PRM(P_RHO_AIR), PCFG(WC_CT_FAN_REF_P, P_CT_FAN_REF_P), PRM(P_CRAC_SUPPLY_PU), PRM(P_RC_RHO_AIR),
                      PCFG(WC_RC_CTAFR, P_RC_CTAFR)};
  const HvacOut hv = hvac_water(kt, HP, p_it, avg_ret, stpt, amb, wet_bulb);
  const double comp = hv.comp, ct = hv.ct, water = hv.water, total_kw = hv.total_kw;

  // ---- battery: envs/bat_env_fwd_view.py:84-245, envs/battery_model.py:94-132 (sdc_physics.hpp battery_step) -----------------------------
  const BatOut bo = battery_step(kt, a_bat, bat_load, PCFG(WC_BAT_CAP, P_BAT_CAP), PCFG(WC_RC_BAT_CAP, P_RC_BAT_CAP), total_kw, ci_i, fault);
  const double e_nobat = bo.e_nobat, energy = bo.energy, co2 = bo.co2, soc_after = bo.soc_after;

  // ---- time: utils/managers.py:127-147 -------------------------------------------------------------------------------------------
  int hourq_n = hourq + 1, day_n = day;
  if (hourq_n >= 96) {
    hourq_n = 0;
    day_n += 1;
  }
  const int ip = i + 1;

  // ---- history append (utils/reward_creator.py:7-14) -------------------------------------------------------------------------------
  const double href = hl == 0 ? energy : href0;
  const double e_off = energy - href;
  const int slot = hist_append_slot(hl, hpos, S.hist_cap);
  const unsigned x_new = sdc_f32_key(__float_as_uint((float)e_off));
  const unsigned f_all = fault0 | fault;

  WST(4);
  // ---- hand-over to the reward wavefront --------------------------------------------------------------------------------------------
  {
    WideHand& H = *reinterpret_cast<WideHand*>(sh.row + WE * 16);
    H.e_off[lane] = e_off; H.energy[lane] = energy; H.norm_ci[lane] = norm_ci;
    H.x_new[lane] = x_new; H.hl[lane] = hl; H.slot[lane] = slot;
    if constexpr (GEN) {      // (inputs of the other reward functions: utils/reward_creator.py:154-334)
      WideHandGen& G = *reinterpret_cast<WideHandGen*>(sh.row + WE * 32);
      G.p_it[lane] = p_it; G.total_kw[lane] = total_kw; G.water[lane] = water;
    }
  }
  __syncthreads();      // (2) the step's energy is known; so are the oldest task and the ages
  const uint4 hb0 = block_get<16>(sh.rec, lane, WIDE_REC_CHUNKS), hb1 = block_get<16>(sh.rec, lane, WIDE_REC_CHUNKS + 1);
  const int head = (int)hb1.x, cum_hm1 = (int)hb1.y;
  const unsigned cumT_hm1 = hb1.z;
  const double oldest_norm = __hiloint2double((int)hb0.y, (int)hb0.x), avg_norm = __hiloint2double((int)hb0.w, (int)hb0.z);

  WST(5);
  // ---- new state: this step's key into the ring, its prefix counts into the queue table, the record's changed chunks into the block and
  // the block out.  (All of the step's stores sit HERE, behind the reward part: this hardware counts loads and stores in one
  // counter and retires them in order, so a wait for a late load -- a window's keys for the whole-wavefront steps above -- is also
  // a wait for every store issued before it.)
  S.hist[(size_t)env * SDC_HIST_STRIDE + slot] = x_new;
  S.qtab[(size_t)env * S.qstride + now] = make_uint2((unsigned)cum_now, cumT_now);
  S.qcum_t[(size_t)now * S.n_envs + env] = (unsigned)cum_now;
  // (the ring's slot-major mirror, SdcDev::hist_t, of the batches that have one: it lies behind qcum_t)
  if (S.n_envs >= SDC_HIST_MIRROR_MIN_ENVS) S.qcum_t[(size_t)(S.qstride + slot) * S.n_envs + env] = x_new;
  block_put<16>(sh.rec, lane, 0, make_uint4((unsigned)ip, (unsigned)(rel + 1), (unsigned)day_n, (unsigned)hourq_n));
  block_put<16>(sh.rec, lane, 1, make_uint4((unsigned)popped, (unsigned)cum_now, cumT_now, (unsigned)head));
  block_put<16>(sh.rec, lane, 2, make_uint4((unsigned)cum_hm1, cumT_hm1, (unsigned)delta, (unsigned)consecutive));
  block_put<16>(sh.rec, lane, 3, make_uint4((unsigned)scale, (unsigned)hl, (unsigned)hpos, f_all));
  if constexpr (GEN) {
    block_put<16>(sh.rec, lane, 6, make_uint4(r6.x, r6.y, (unsigned)tr_count, r6.w));
    if (S.actions_out) {     // the actions the step applied (rule-based policies: what they chose)
      int32_t* ao = S.actions_out + (size_t)env * 3;
      ao[0] = a_ls; ao[1] = a_dc; ao[2] = a_bat;
    }
  }
  block_put<16>(sh.rec, lane, 4, make_uint4((unsigned)__double2loint(stpt), (unsigned)__double2hiint(stpt), (unsigned)__double2loint(bat_load),
                                               (unsigned)__double2hiint(bat_load)));
  block_put<16>(sh.rec, lane, 5, make_uint4((unsigned)__double2loint(href), (unsigned)__double2hiint(href), (unsigned)__double2loint(mean_outlet),
                                               (unsigned)__double2hiint(mean_outlet)));
  wave_sync();
  block_store16<GEN ? 8 : 6>(S.rec + (size_t)env0 * SDC_REC_DWORDS, sh.rec, lane, 0ull);

  WST(6);
  // ---- outputs ----------------------------------------------------------------------------------------------------------------------
  // the observation pool (sdc_device.hpp SDC_P_*): the trace-only entries are the feature row's, nine depend on the step
  float pool[SDC_POOL_DIM];
#pragma unroll
  for (int j = 0; j < SDC_POOL_DIM; j++) pool[j] = row[j];
  pool[SDC_P_OLDEST] = (float)oldest_norm;
  pool[SDC_P_AVG] = (float)avg_norm;
  pool[SDC_P_NORMQ] = (float)normq;
#pragma unroll
  for (int b = 0; b < 5; b++) pool[SDC_P_HIST + b] = (float)hist[b];
  pool[SDC_P_SOC] = (float)soc_after;
  const bool terminal = rel + 1 >= S.episode_steps;
  {
    // obs [3][26] of this lane's env into its row of the staging block, then the block out in whole lines
    float* const stage = reinterpret_cast<float*>(&sh.rec[0]);
    wave_sync();
    float* srow = stage + lane * SDC_OBS_OUT;
#pragma unroll
    for (int j = 0; j < SDC_OBS_OUT; j += 2) {
      float2 v;
      v.x = obs_pool_index(j) < 0 ? 0.0f : pool[obs_pool_index(j) < 0 ? 0 : obs_pool_index(j)];
      v.y = obs_pool_index(j + 1) < 0 ? 0.0f : pool[obs_pool_index(j + 1) < 0 ? 0 : obs_pool_index(j + 1)];
      *reinterpret_cast<float2*>(srow + j) = v;
    }
    wave_sync();
    const f4v* s4 = reinterpret_cast<const f4v*>(stage);
    float* o4 = obs + (size_t)env0 * SDC_OBS_OUT;
    f4v* f4 = reinterpret_cast<f4v*>(final_obs + (size_t)env0 * SDC_OBS_OUT);
    constexpr int NV = WE * SDC_OBS_OUT / 4;
#pragma unroll
    for (int k = 0; k < (NV + WE - 1) / WE; k++) {
      const int q = k * WE + lane;
      if (q < NV) {
        const f4v v = s4[q];
        nt_store4(o4 + 4 * q, v);
        if (final_obs && terminal) f4[q] = v;
      }
    }
    wave_sync();
    float* hrow = stage + lane * SDC_SHARE_OBS_DIM;
#pragma unroll
    for (int j = 0; j < SDC_SHARE_OBS_DIM; j++) hrow[j] = j == SDC_P_SOC ? 0.0f : pool[j];
    wave_sync();
    float* h4 = share_obs + (size_t)env0 * SDC_SHARE_OBS_DIM;
    constexpr int NH = WE * SDC_SHARE_OBS_DIM / 4;
#pragma unroll
    for (int k = 0; k < (NH + WE - 1) / WE; k++) {
      const int q = k * WE + lane;
      if (q < NH) nt_store4(h4 + 4 * q, s4[q]);
    }
  }
  {
    float inf[SDC_INFO_DIM];
    const InfoLs il = {wl, util, normq, oldest_norm, avg_norm, hour, total, dropped, processed, overdue};
    info_put_ls(inf, il, hist);
    const InfoDc id = {p_it, ct, comp, total_kw, stpt, mean_outlet, amb, water, soc_after, co2, ci_i, e_nobat, energy, norm_ci, amb_next,
                       delta, a_bat, day_n, hourq_n, rel + 1, f_all};
    info_put_dc(kt, inf, id);      // (the five reward-side columns: zeros here, filled in by the reward wavefront, which sends the block out)
    // ... through the staging block too: the wavefront's 64 info rows are 11 KB of whole lines
    float* const stage = reinterpret_cast<float*>(&sh.rec[0]);
    wave_sync();
    f4v* irow = reinterpret_cast<f4v*>(stage + lane * SDC_INFO_DIM);
#pragma unroll
    for (int q = 0; q < SDC_INFO_DIM / 4; q++) {
      f4v v;
      v.x = inf[4 * q]; v.y = inf[4 * q + 1]; v.z = inf[4 * q + 2]; v.w = inf[4 * q + 3];
      irow[q] = v;
    }
  }
  done[env] = (unsigned char)(terminal ? 1 : 0);
  WST(7);
  __syncthreads();      // (3) the info rows are staged (minus the reward-side columns)
}

// ---- the REWARD wavefront of an env workgroup: lane = env; whole-wavefront steps (lane = key) for what needs a window's keys ----------
template <bool GEN>
__device__ __forceinline__ void wide_rewards(const SdcDev& S, WideSharedT<GEN>& sh, const int env0, const int lane, const int rel_hint,
                                             const int32_t* __restrict__ actions, float* __restrict__ info, float* __restrict__ rew) {
  using namespace sdc_rw;
  const int env = env0 + lane;
  WideLate& late = *reinterpret_cast<WideLate*>(sh.hdr);
  KLit kt{};
  // history length and ring position BEFORE the step, straight from the record in memory (the record BLOCK is on its way into LDS
  // for the dynamics wavefront): they address the ring key this step evicts, and that load starts one round trip earlier this way
  const uint4 r3g = *reinterpret_cast<const uint4*>(S.rec + (size_t)env * SDC_REC_DWORDS + 12);
  static_assert(R_HIST_LEN == 13 && R_HIST_POS == 14, "dwords 1 and 2 of the record's fourth chunk");
  block_load<16>(S.hdr + (size_t)env0 * SDC_HDR_DWORDS, sh.hdr, lane);
  int a_ls = 1;      // (GEN: a do-nothing ls slot never reads the caller's array, which may be null)
  if (!GEN || (actions && S.policy[0] == SDC_POLICY_EXTERNAL)) a_ls = actions[(size_t)env * 3];
  // the step's workload straight from its feature row: the row BLOCK in LDS is the dynamics wavefront's, which parks its rack-class
  // results over it once it has read it
  const float2 wl2 = *reinterpret_cast<const float2*>(S.feat + feat_row_offset(S, env0, rel_hint + 1) + (size_t)lane * SDC_FEAT_ROW + SDC_FEAT_W);
  const uint2* qt = S.qtab + (size_t)env * S.qstride;
  int cumq[5];     // cum[now - 97], cum[now - 24], cum[now - 48], cum[now - 72], cum[now - 96] (0 before the episode's start)
#pragma unroll
  for (int s = 0; s < 5; s++) {
    const int back = s == 0 ? 97 : 24 * s;
    const int t = rel_hint - back;
    cumq[s] = t >= 0 ? (int)S.qcum_t[(size_t)t * S.n_envs + env] : 0;      // (the table's time-major mirror: 64 consecutive dwords)
  }
  // the evicted ring key (read BEFORE this step's key goes into that slot: the dynamics wavefront stores it behind barrier 2)
  int hl = (int)r3g.y;
  const int hpos = (int)r3g.z;
  const int slot0 = hl < S.hist_cap ? hl : hpos;
  unsigned x_old = 0xFFFFFFFFu;
  // (batches with the ring's slot-major mirror read it there -- consecutive dwords instead of a 128-byte line per env; SdcDev::hist_t =
  // rows qstride .. of the array qcum_t points to)
  if (hl >= S.hist_cap) {
    if (S.n_envs >= SDC_HIST_MIRROR_MIN_ENVS) x_old = S.qcum_t[(size_t)(S.qstride + slot0) * S.n_envs + env];
    else x_old = S.hist[(size_t)env * SDC_HIST_STRIDE + slot0];
  }
  dma_wait();
  __syncthreads();      // (1) records, feature rows and headers are in LDS
  // ---- the oldest queued task and the ages (pair_dynamics, same expressions): the load-shifting algebra once more, then the queue
  // table ahead of the oldest task -- requested now, looked at after the header work below
  const uint4 q0 = block_get<16>(sh.rec, lane, 0), q1 = block_get<16>(sh.rec, lane, 1), q2 = block_get<16>(sh.rec, lane, 2);
  const int rel = (int)q0.y, hourq = (int)q0.w, popped0 = (int)q1.x, cum_prev = (int)q1.y;
  const unsigned cumT_prev = q1.z;
  int head = (int)q1.w, cum_hm1 = (int)q2.x;
  unsigned cumT_hm1 = q2.y;
  const int now = rel;
  const double wl = __hiloint2double(__float_as_int(wl2.y), __float_as_int(wl2.x));
  if ((unsigned)a_ls > 2u) a_ls = 1;
  const LsStep ls = ls_algebra(kt, wl, a_ls, popped0, cum_prev, cumT_prev, now, S.queue_max, cumq[0], cumq[1], cumq[2], cumq[3], cumq[4]);
  const int overdue = ls.overdue, hourq_n = hourq + 1 >= 96 ? 0 : hourq + 1;
  constexpr int QA = WIDE_QA;
  uint4 qa[QA / 2];
#pragma unroll
  for (int q = 0; q < QA / 2; q++) qa[q] = make_uint4(0u, 0u, 0u, 0u);
  if (a_ls == 2 || cumq[0] - popped0 > 0) {       // (drain, or overdue tasks to run: the oldest task may move)
#pragma unroll
    for (int q = 0; q < QA / 2; q++) {
      const int t = head + 2 * q;
      if (t + 1 < rel) qa[q] = *reinterpret_cast<const uint4*>(qt + t);
      else if (t < rel) { const uint2 e = qt[t]; qa[q] = make_uint4(e.x, e.y, 0u, 0u); }
    }
  }
  WST_HOLD(wst8);

  // ---- reward-side state: the env's header (its 64 dwords in this lane's registers), then the few window keys a step usually needs
  unsigned hd[SDC_HDR_DWORDS];
#pragma unroll
  for (int q = 0; q < SDC_HDR_DWORDS / 4; q++) {
    const uint4 v = block_get<16>(sh.hdr, lane, q);
    hd[4 * q] = v.x; hd[4 * q + 1] = v.y; hd[4 * q + 2] = v.z; hd[4 * q + 3] = v.w;
  }
  auto hd_f64 = [&](const int j) { return __hiloint2double((int)hd[j + 1], (int)hd[j]); };
  // window w of this lane's env: rank of its first key, valid keys, cached first / last key (window order: Q1, Q3, BU, BL)
  constexpr int HW[4] = {H_Q1, H_Q3, H_BU, H_BL};
  int wr0[4], whi[4];
  unsigned wf[4], wlast[4], pend[4];
#pragma unroll
  for (int w = 0; w < 4; w++) {
    wr0[w] = (int)hd[HW[w] + T_R0];
    whi[w] = (int)hd[HW[w] + T_HI];
    wf[w] = hd[H_WFIRST + w];
    wlast[w] = hd[H_WLAST + w];
    pend[w] = hd[H_PEND + w];
  }
  // history length after this step's append, the quartile ranks it asks for
  const int n_pre = hl;
  const int n_step = hl < S.hist_cap ? hl + 1 : hl;
  int k1, k3;
  quartile_ranks(n_step, k1, k3);
  // FOUR keys of every window, gathered now: around the wanted rank (quartile windows: the rank moves by at most one position
  // when both of the step's keys land outside the window) and around last step's clip bound (bound windows: the two keys
  // either side of it).  ck[w][j] = key at window position cb[w] + j.
  unsigned ck[4][4];
  int cb[4];
  cb[0] = k1 - wr0[0] - 1;
  cb[1] = k3 - wr0[1] - 1;
  cb[2] = n_pre - (int)hd[H_QC] - wr0[2] - 2;
  cb[3] = n_pre - (int)hd[H_QC + 1] - wr0[3] - 2;
  {
    const unsigned* qw = S.qwin + (size_t)env * SDC_WIN * 4;
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int p = min(max(cb[w] + j, 0), SDC_WIN - 1);
        ck[w][j] = qw[p * 4 + w];
      }
  }

  WST_HOLD(wst18);
  // ---- rewards + reward-state upkeep (utils/reward_creator.py:16-130): pair_reward_fast, lane = env ---------------------------------
  const int n = n_step;         // history length after this step's append
  const bool has_old = x_old != KEY_NONE;
  const int n_prev = has_old ? n : n - 1;
  const int m_hist = has_old ? n_prev - 1 : n_prev;
  double A1 = hd_f64(H_A1), A2 = hd_f64(H_A2);
  bool ok = (n >= SMALL_N) & (whi[0] > 0) & (whi[1] > 0) & (whi[2] > 0) & (whi[3] > 0) & ((int)hd[H_VALID] == 1);
  bool touched = false;     // a window of this env has been rewritten in memory by this step (a fallback must then rebuild)
  bool arrived = false;
  const unsigned kbl[2] = {hd[H_KB], hd[H_KB + 1]};
  unsigned* qwin_env0 = S.qwin + (size_t)env0 * SDC_WIN * 4;

  // what a whole-wavefront step on window w of env e (lane = key) hands to e's lane: the window's header values and the four keys
  // around the position the lane will look at
  auto deliver = [&](const int e, const int w, const QTrack& q, const int base) __attribute__((always_inline)) {
    const unsigned first = lane_key(q.w, 0), last = lane_key(q.w, max(q.hi - 1, 0));
    unsigned c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) c[j] = lane_key(q.w, min(max(base + j, 0), SDC_WIN - 1));
    if (lane == e) {
#pragma unroll
      for (int x = 0; x < 4; x++)
        if (x == w) {
          wr0[x] = q.r0; whi[x] = q.hi; wf[x] = first; wlast[x] = last; cb[x] = base;
#pragma unroll
          for (int j = 0; j < 4; j++) ck[x][j] = c[j];
        }
    }
  };
  // where lane e will look in window w: the wanted rank's position - 1 (quartiles), the old bound's position - 2 (bounds)
  auto look_base = [&](const int w, const QTrack& q, const int k_w, const unsigned kbl_w) __attribute__((always_inline)) {
    if (w < 2) return k_w - q.r0 - 1;
    return (int)__popcll(__ballot(q.w < kbl_w)) - 2;      // (valid keys below the bound; KEY_NONE never counts)
  };

  // ---- deferred re-centrings (SdcRefillReq / SdcRefillRes): a window requested two steps ago arrives now ---------------------------
  {
    bool due[4];
    unsigned age[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
      age[w] = ((unsigned)S.step_no - (pend[w] >> 13)) & 0x7FFFFu;
      due[w] = pend[w] != 0u && age[w] >= 2u;     // (1: being swept right now; anything else but 2: stale)
    }
    unsigned long long dm = __ballot(due[0] | due[1] | due[2] | due[3]);
    if (__builtin_expect(dm != 0ull, 0)) {
      const unsigned lx_new = hd[H_LAST_XNEW], lx_old = hd[H_LAST_XOLD];
      const int l_nprev = (int)hd[H_LAST_NPREV];
      while (dm) {
        const int e = __ffsll((long long)dm) - 1;
        dm &= dm - 1;
        const unsigned e_new = lane_key(lx_new, e), e_old = lane_key(lx_old, e);
        const int e_nprev = (int)lane_key((unsigned)l_nprev, e);
        const bool e_ok = lane_key(ok ? 1u : 0u, e) != 0u;
#pragma unroll
        for (int w = 0; w < 4; w++) {
          if (!lane_key(due[w] ? 1u : 0u, e)) continue;
          const unsigned pd = lane_key(pend[w], e);
          const int idx = (int)(pd & 0x7FFu) - 1, set = (int)((pd >> 11) & 3u);
          const SdcRefillRes* rs = S.rs + (set > 2 ? 0 : set) * S.rq_max + (idx < 0 ? 0 : idx);
          const int4 rh = *reinterpret_cast<const int4*>(rs);
          QTrack r = {rs->keys[lane], rh.x, rh.y};
          const unsigned flip = w == 3 ? KEY_NONE : 0u;
          const bool good = lane_key(age[w], e) == 2u && rh.z == S.step_no - 1 && rh.w == (env0 + e) * 4 + w && r.hi > 0 && e_ok;
          // the result describes the ring as the request's step left it: replay the previous step's insertion / eviction
          if (good) qt_update(r, e_new ^ flip, e_old ^ flip, e_old != KEY_NONE, e_nprev, lane);
          if (good && r.hi > 0) {
            qwin_env0[((size_t)e * SDC_WIN + lane) * 4 + w] = r.w;
            const int kw = w == 0 ? (int)lane_key((unsigned)k1, e) : (int)lane_key((unsigned)k3, e);
            deliver(e, w, r, look_base(w, r, kw, lane_key(kbl[w == 3 ? 1 : 0], e)));
            if (lane == e) { touched = true; arrived = true; }
          }
          if (lane == e) pend[w] = 0u;
        }
      }
    }
  }

  WST_HOLD(wst17);
  // ---- O(1) window updates, the two halves of a step's qt_update on every window: the EVICTION (the ring key this step overwrites:
  // known now) before barrier 2, in the time this wavefront would wait for the energy; the INSERTION behind it.  Per lane: a key that
  // lands below / above a window only moves its ranks.  A key that lands INSIDE a window of some env (or where the window starts /
  // ends the history): the whole wavefront updates that window, lane = key -- ~3 such (env, window) tasks per wavefront and step,
  // 8-10 in the busiest workgroup of a launch, which is the one the launch waits for: ALL the windows of a phase are requested at
  // once, by LDS-DMA into the header block's free part (one memory round trip), then taken one after the other.
  unsigned x_new_late = 0u;      // (this step's key: known behind barrier 2)
  auto window_tasks = [&](const bool (&upd)[4], auto EVICT) __attribute__((always_inline)) {
    constexpr bool evict = decltype(EVICT)::value;
    unsigned long long um[4];
#pragma unroll
    for (int w = 0; w < 4; w++) um[w] = __ballot(upd[w]);
    unsigned* const wbuf = sh.hdr + WIDE_TASK_OFF;      // [slot][key]
    while (__builtin_expect((um[0] | um[1] | um[2] | um[3]) != 0ull, 0)) {
      {
        int k = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
          unsigned long long m = um[w];
          while (m != 0ull && k < WIDE_TASK_SLOTS) {
            const int e = __ffsll((long long)m) - 1;
            m &= m - 1;
            __builtin_amdgcn_global_load_lds((sdc_gptr)(qwin_env0 + ((size_t)e * SDC_WIN + lane) * 4 + w), (sdc_lptr)(wbuf + k * SDC_WIN), 4, 0, 0);
            k++;
          }
        }
      }
      dma_wait();
      int k = 0;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const unsigned flip = w == 3 ? KEY_NONE : 0u;
        while (um[w] != 0ull && k < WIDE_TASK_SLOTS) {
          const int e = __ffsll((long long)um[w]) - 1;
          um[w] &= um[w] - 1;
          QTrack q = {wbuf[k * SDC_WIN + lane], (int)lane_key((unsigned)wr0[w], e), (int)lane_key((unsigned)whi[w], e)};
          k++;
          bool wd;
          if (evict) {
            wd = qt_evict(q, lane_key(x_old, e) ^ flip, lane);
          } else {
            const bool e_has_old = lane_key(x_old, e) != KEY_NONE;
            const int e_nprev = (int)lane_key((unsigned)n_prev, e);
            wd = qt_insert(q, lane_key(x_new_late, e) ^ flip, e_has_old ? e_nprev - 1 : e_nprev, lane);
          }
          if (wd) qwin_env0[((size_t)e * SDC_WIN + lane) * 4 + w] = q.w;
          const int kw = w == 0 ? (int)lane_key((unsigned)k1, e) : (int)lane_key((unsigned)k3, e);
          deliver(e, w, q, look_base(w, q, kw, lane_key(kbl[w == 3 ? 1 : 0], e)));
          if (lane == e && wd) touched = true;
        }
      }
    }
  };
  {
    bool upd[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const unsigned y = x_old ^ (w == 3 ? KEY_NONE : 0u);
      const bool e_below = has_old && y < wf[w];
      const bool e_ok = !has_old || y > wlast[w] || e_below;          // (evicted key above the window / below it)
      upd[w] = ok && !e_ok;
      if (e_ok && e_below) wr0[w] -= 1;
    }
    window_tasks(upd, std::true_type{});
  }

  WST_PUT(8, wst8);
  WST_PUT(17, wst17);
  WST_PUT(18, wst18);
  WST(9);

  // ---- oldest task: smallest step hd in [head, now] with cum[hd] > popped; it only moves when tasks were popped ----------------------
  double oldest_norm, avg_norm;
  {
    const int popped = ls.popped, cum_now = ls.cum_now, total = ls.total;
    const unsigned cumT_now = ls.cumT_now;
    const double den = (double)max(total, 1), rden = 1.0 / den;
    double oldest = 0.0, avg = 0.0;
    const bool was_empty = (cum_prev - popped0) == 0;
    bool need = total > 0 && !was_empty && popped != popped0;
    if (need) {
      // among the QA entries requested up front (entry j = step head + j)?
      int f = -1;
      unsigned cm1 = 0u, ctm1 = 0u;     // cum / cumT of the entry before the first hit
#pragma unroll
      for (int j = QA - 1; j >= 0; j--) {
        const int t = head + j;
        const unsigned cx = (j & 1) ? qa[j / 2].z : qa[j / 2].x;
        const int c = (t == now) ? cum_now : (int)cx;
        if (t <= now && c > popped) f = j;
      }
#pragma unroll
      for (int j = 0; j < QA - 1; j++) {
        if (f == j + 1) {
          cm1 = (j & 1) ? qa[j / 2].z : qa[j / 2].x;
          ctm1 = (j & 1) ? qa[j / 2].w : qa[j / 2].y;
        }
      }
      if (f >= 0) {
        need = false;
        if (f > 0) {
          head += f;
          cum_hm1 = (int)cm1;
          cumT_hm1 = ctm1;
        }
      }
    }
    if (need) {
      // further ahead than that (rare): walk the table
      int t = head + QA;
      while (t < now && (int)qt[t].x <= popped) t++;      // (cum[now] > popped: the walk ends at `now` at the latest)
      head = t;
      if (head == now) {
        cum_hm1 = cum_prev;
        cumT_hm1 = cumT_prev;
      } else {
        const uint2 e = qt[head - 1];
        cum_hm1 = (int)e.x;
        cumT_hm1 = e.y;
      }
    }
    ls_ages(ls, now, cum_prev, cumT_prev, was_empty, den, rden, head, cum_hm1, cumT_hm1, oldest, avg);
    oldest_norm = KDIV(oldest, 24);
    avg_norm = KDIV(avg, 24);
    block_put<16>(sh.rec, lane, WIDE_REC_CHUNKS, make_uint4((unsigned)__double2loint(oldest_norm), (unsigned)__double2hiint(oldest_norm),
                                                          (unsigned)__double2loint(avg_norm), (unsigned)__double2hiint(avg_norm)));
    block_put<16>(sh.rec, lane, WIDE_REC_CHUNKS + 1, make_uint4((unsigned)head, (unsigned)cum_hm1, cumT_hm1, 0u));
  }
  WST(19);
  __syncthreads();      // (2) the step's energy is known
  const WideHand& H = *reinterpret_cast<const WideHand*>(sh.row + WE * 16);
  const double e_off = H.e_off[lane], energy = H.energy[lane], norm_ci = H.norm_ci[lane];
  const unsigned x_new = H.x_new[lane];
  x_new_late = x_new;
  const int slot = H.slot[lane];
  hl = H.hl[lane];      // (after the append: == n)

  WST(10);
  // ---- O(1) updates: running sums, the four windows ------------------------------------------------------------------------------------
  const double vn = key_f64(x_new), vo = has_old ? key_f64(x_old) : 0.0;
  A1 += vn - vo;
  A2 += vn * vn - vo * vo;
  {
    // the appended key against every window as the eviction left it (qt_update = qt_evict, then qt_insert: the eviction half ran
    // before barrier 2)
    bool upd[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const unsigned x = x_new ^ (w == 3 ? KEY_NONE : 0u);
      const bool ends = wr0[w] + whi[w] == m_hist;
      const bool i_below = x < wf[w] && wr0[w] != 0;
      const bool i_ok = i_below || (x >= wlast[w] && !ends);          // (appended key below the window / above it)
      upd[w] = ok && !i_ok && whi[w] > 0;
      if (i_ok && i_below) wr0[w] += 1;      // (the window's keys stay where they are, its ranks move)
    }
    window_tasks(upd, std::false_type{});
  }
  ok = ok && whi[0] > 0 && whi[1] > 0 && whi[2] > 0 && whi[3] > 0;
  // key at position p of window w, if it is one of the four this lane holds
  auto key_at_w = [&](const int w, const int p, bool& have) __attribute__((always_inline)) {
    const int j = p - cb[w];
    have = have && j >= 0 && j < 4;
    const unsigned c0 = opq(ck[w][0]), c1 = opq(ck[w][1]), c2 = opq(ck[w][2]), c3 = opq(ck[w][3]);
    return j <= 0 ? c0 : (j == 1 ? c1 : (j == 2 ? c2 : c3));
  };
  WST(11);
  unsigned a1, b1, a3, b3;
  {
    // keys at ranks k and k + 1 (the second only if it exists): hw_resolve
    bool have = true;
    const int t1 = k1 - wr0[0], tb1 = (k1 + 1 > n - 1) ? t1 : t1 + 1;
    const int t3 = k3 - wr0[1], tb3 = (k3 + 1 > n - 1) ? t3 : t3 + 1;
    a1 = key_at_w(0, t1, have); b1 = key_at_w(0, tb1, have);
    a3 = key_at_w(1, t3, have); b3 = key_at_w(1, tb3, have);
    const bool r1 = whi[0] > 0 && t1 >= 0 && tb1 < whi[0], r3 = whi[1] > 0 && t3 >= 0 && tb3 < whi[1];
    ok = ok && r1 && r3 && have;
  }
  const Bounds b = clip_bounds(n, a1, b1, a3, b3);
  const unsigned kb0 = b.kub;               // upper tail: keys >= kub
  const unsigned kb1 = ~(b.klb - 1u);       // lower tail, flipped: ~x >= ~(klb-1)  <=>  x < klb
  int qc0 = (int)hd[H_QC], qc1 = (int)hd[H_QC + 1];
  double qs1_0 = hd_f64(H_QS1), qs1_1 = hd_f64(H_QS1 + 2);
  double qs2_0 = hd_f64(H_QS2_HI), qs2_1 = hd_f64(H_QS2_LO);
  {
    const double vo2 = key_f64(x_old);
    if (has_old && x_old >= kbl[0]) { qc0 -= 1; qs1_0 -= vo2; qs2_0 -= vo2 * vo2; }
    if (has_old && ~x_old >= kbl[1]) { qc1 -= 1; qs1_1 -= vo2; qs2_1 -= vo2 * vo2; }
    if (x_new >= kbl[0]) { qc0 += 1; qs1_0 += vn; qs2_0 += vn * vn; }
    if (~x_new >= kbl[1]) { qc1 += 1; qs1_1 += vn; qs2_1 += vn * vn; }
  }
  {
    const unsigned lo0 = min(kb0, kbl[0]), hi0 = max(kb0, kbl[0]), lo1 = min(kb1, kbl[1]), hi1 = max(kb1, kbl[1]);
    const bool sp0 = whi[2] > 0 && (wr0[2] == 0 || wf[2] < lo0) && (wr0[2] + whi[2] >= n || hi0 <= wlast[2]);
    const bool sp1 = whi[3] > 0 && (wr0[3] == 0 || wf[3] < lo1) && (wr0[3] + whi[3] >= n || hi1 <= wlast[3]);
    ok = ok && (kb0 == kbl[0] || sp0) && (kb1 == kbl[1] || sp1);
    // the keys a bound has crossed since last step (the window lists them: sp0 / sp1): this lane holds the two keys either side
    // of the OLD bound (positions pos - 2 .. pos + 1, pos = the first key at or beyond it).  None or ONE crossed key is settled
    // here; more -- or a picture of the neighbourhood that does not check out -- by the whole wavefront on the window's 64 keys.
    auto crossing = [&](const int w, const unsigned kb, const unsigned kbo, const unsigned flip, int& qc, double& qs1, double& qs2) __attribute__((always_inline)) {
      if (kb == kbo) return false;
      const int pos = cb[w] + 2;
      const bool xA2 = pos - 2 >= 0 && pos - 2 < whi[w], xA = pos - 1 >= 0 && pos - 1 < whi[w];
      const bool xB = pos >= 0 && pos < whi[w], xB2 = pos + 1 >= 0 && pos + 1 < whi[w];
      const unsigned kA2 = opq(ck[w][0]), kA = opq(ck[w][1]), kB = opq(ck[w][2]), kB2 = opq(ck[w][3]);
      // the lane's picture must be the window's: `pos` keys lie below the old bound
      if (pos < 0 || pos > whi[w] || (xA && !(kA < kbo)) || (xB && !(kB >= kbo))) return true;
      if (kb > kbo) {
        // bound moved out: the keys in [kbo, kb) leave the tail
        if (!xB || kB >= kb) return false;
        if (xB2 && kB2 < kb) return true;
        const double v = key_f64(kB ^ flip);
        qc += -1; qs1 += -1.0 * v; qs2 += -1.0 * (v * v);
      } else {
        // bound moved in: the keys in [kb, kbo) enter the tail
        if (!xA || kA < kb) return false;
        if (xA2 && kA2 >= kb) return true;
        const double v = key_f64(kA ^ flip);
        qc += 1; qs1 += 1.0 * v; qs2 += 1.0 * (v * v);
      }
      return false;
    };
    const bool cx0 = ok && crossing(2, kb0, kbl[0], 0u, qc0, qs1_0, qs2_0);
    const bool cx1 = ok && crossing(3, kb1, kbl[1], KEY_NONE, qc1, qs1_1, qs2_1);
    unsigned long long cm = __ballot(cx0 | cx1);
    if (__builtin_expect(cm != 0ull, 0)) {
      if (__ballot(touched) != 0ull) {     // (the windows' keys come from memory: this wavefront's own stores to them must have landed)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      while (cm) {
        const int e = __ffsll((long long)cm) - 1;
        cm &= cm - 1;
#pragma unroll
        for (int w = 2; w < 4; w++) {
          if (!lane_key(((w == 2 ? cx0 : cx1)) ? 1u : 0u, e)) continue;
          const unsigned flip = w == 3 ? KEY_NONE : 0u;
          const unsigned e_lo = lane_key(w == 2 ? lo0 : lo1, e), e_hi = lane_key(w == 2 ? hi0 : hi1, e);
          const unsigned key = qwin_env0[((size_t)e * SDC_WIN + lane) * 4 + w];
          const bool in = key >= e_lo && key < e_hi;        // (KEY_NONE lanes: hi <= KEY_NONE)
          const int c = (int)__popcll(__ballot(in));
          const double v = in ? key_f64(key ^ flip) : 0.0;
          const double s1 = wave_sum_f64(v), s2 = wave_sum_f64(v * v);
          if (lane == e) {
            const bool outw = (w == 2 ? kb0 > kbl[0] : kb1 > kbl[1]);       // bound moved out: the keys in between leave the tail
            const double sg = outw ? -1.0 : 1.0;
            if (w == 2) { qc0 += (outw ? -1 : 1) * c; qs1_0 += sg * s1; qs2_0 += sg * s2; }
            else { qc1 += (outw ? -1 : 1) * c; qs1_1 += sg * s1; qs2_1 += sg * s2; }
          }
        }
      }
    }
  }
  // sum (v - bound), sum (v^2 - bound^2) over the keys beyond the bounds
  const double tt1 = (qs1_0 - (double)qc0 * b.ub) + (qs1_1 - (double)qc1 * b.lb);
  const double tt2 = (qs2_0 - (double)qc0 * (b.ub * b.ub)) + (qs2_1 - (double)qc1 * (b.lb * b.lb));
  double mean, sd, inv_sd;
  clipped_moments(n, b, A1, A2, tt1, tt2, mean, sd, inv_sd, S.hist_cap, S.rc_hist_cap, S.hist_cap_d);
  WST(12);
  // ---- a window that the next step could exhaust: file a re-centring request (served by the NEXT launch's sweep wavefronts) -----------
  {
    int k1n, k3n;
    quartile_ranks(n < S.hist_cap ? n + 1 : n, k1n, k3n);
    const int ktw[4] = {k1n, k3n, n - qc0, n - qc1};
    bool want[4];
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const int m_lo = w < 2 ? 3 : 10, m_hi = w < 2 ? 6 : 10;
      const int t = ktw[w] - wr0[w];
      want[w] = ok & (pend[w] == 0u) & (((t > whi[w] - m_hi) & (wr0[w] + whi[w] < n)) | ((t < m_lo) & (wr0[w] > 0)));
    }
    unsigned long long rm = __ballot(want[0] | want[1] | want[2] | want[3]);
    if (__builtin_expect(rm != 0ull, 0)) {
      // (the windows' keys come from memory: this wavefront's own stores to them above must have landed)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int set = (S.step_no + 1) % 3;
      while (rm) {
        const int e = __ffsll((long long)rm) - 1;
        rm &= rm - 1;
        const int e_n = (int)lane_key((unsigned)n, e), e_slot = (int)lane_key((unsigned)slot, e);
        const int slot_next = e_n < S.hist_cap ? e_n : (e_slot + 1 == S.hist_cap ? 0 : e_slot + 1);
        const unsigned patch_x = S.hist[(size_t)(env0 + e) * SDC_HIST_STRIDE + slot_next];
#pragma unroll
        for (int w = 0; w < 4; w++) {
          if (!lane_key((want[w] && ok) ? 1u : 0u, e)) continue;
          int idx = -1;
          if (lane == 0) idx = atomicAdd(&S.rq_count[set], 1);
          idx = (int)sfl((unsigned)idx);
          if (idx < 0 || idx >= S.rq_max) {
            if (lane == e) ok = false;          // no room: re-centre inline on the fallback path
            continue;
          }
          SdcRefillReq* rq = S.rq + set * S.rq_max + idx;
          rq->keys[lane] = qwin_env0[((size_t)e * SDC_WIN + lane) * 4 + w];
          const int q_r0 = (int)lane_key((unsigned)wr0[w], e), q_hi = (int)lane_key((unsigned)whi[w], e);
          const int kt_w = (int)lane_key((unsigned)ktw[w], e);
          if (lane == 0) {
            const int t = kt_w - q_r0;
            rq->env = env0 + e; rq->win = w;
            rq->dir = (t > q_hi - (w < 2 ? 6 : 10) && q_r0 + q_hi < e_n) ? REFILL_UP : REFILL_DOWN;
            rq->kt = kt_w; rq->n = e_n; rq->r0 = q_r0; rq->hi = q_hi;
            rq->patch_slot = slot_next; rq->patch_x = patch_x; rq->step = S.step_no;
          }
          if (lane == e) pend[w] = (((unsigned)S.step_no & 0x7FFFFu) << 13) | ((unsigned)set << 11) | (unsigned)(idx + 1);
        }
      }
    }
  }
  const double z = (e_off - mean) * inv_sd;     // (n >= SMALL_N >= 2 here)
  double r_a[3], ret_a[3];
  {
    double foot, rls;
    reward_terms(z, norm_ci, (double)overdue, oldest_norm, foot, rls);
    r_a[0] = rls; r_a[1] = foot; r_a[2] = foot;
    if constexpr (GEN) {
      // the dc / battery agents' reward functions (utils/reward_creator.py:154-334: sdc_trackers.hpp agent_reward); the ls agent keeps
      // default_ls_reward here (wide_gen_case)
      const WideHandGen& G = *reinterpret_cast<const WideHandGen*>(sh.row + WE * 32);
      const double ite_kw = SDC_DIV_CONST(G.p_it[lane], 1e3), hour = (double)hourq_n * 0.25;
#pragma unroll
      for (int a = 1; a < 3; a++) r_a[a] = agent_reward(S.reward_method[a], false, rls, foot, energy, hour, ite_kw, G.total_kw[lane], G.water[lane]);
    }
#pragma unroll
    for (int a = 0; a < 3; a++) ret_a[a] = hd_f64(H_RET + 2 * a) + r_a[a];
  }
  float z_f = (float)z, path_f = arrived ? 2.0f : 0.0f, ret_f[3] = {(float)ret_a[0], (float)ret_a[1], (float)ret_a[2]};
  if (ok) {
    auto put64 = [&](const int j, const double v) { hd[j] = (unsigned)__double2loint(v); hd[j + 1] = (unsigned)__double2hiint(v); };
    hd[H_KB] = kb0;
    hd[H_KB + 1] = kb1;
    hd[H_VALID] = 1u;
    put64(H_A1, A1);
    put64(H_A2, A2);
    hd[H_QC] = (unsigned)qc0;
    hd[H_QC + 1] = (unsigned)qc1;
    put64(H_QS1, qs1_0);
    put64(H_QS1 + 2, qs1_1);
    put64(H_QS2_HI, qs2_0);
    put64(H_QS2_LO, qs2_1);
#pragma unroll
    for (int w = 0; w < 4; w++) {
      hd[HW[w] + T_R0] = (unsigned)wr0[w];
      hd[HW[w] + T_HI] = (unsigned)whi[w];
      hd[H_PEND + w] = pend[w];
      hd[H_WFIRST + w] = wf[w];
      hd[H_WLAST + w] = wlast[w];
    }
    hd[H_N] = (unsigned)n;
    hd[H_LAST_XNEW] = x_new;
    hd[H_LAST_XOLD] = x_old;
    hd[H_LAST_NPREV] = (unsigned)n_prev;
    put64(H_EOFF, e_off);
    put64(H_RET, ret_a[0]);
    put64(H_RET + 2, ret_a[1]);
    put64(H_RET + 4, ret_a[2]);
#pragma unroll
    for (int q = 0; q < SDC_HDR_DWORDS / 4; q++)
      block_put<16>(sh.hdr, lane, q, make_uint4(hd[4 * q], hd[4 * q + 1], hd[4 * q + 2], hd[4 * q + 3]));
    rew[(size_t)env * 3 + 0] = (float)r_a[0];
    rew[(size_t)env * 3 + 1] = (float)r_a[1];
    rew[(size_t)env * 3 + 2] = (float)r_a[2];
  }
  WST(13);
  // ---- an env whose step needs anything else (no reward state yet, a window that does not cover, several keys across a bound, no
  // room for a request): env_reward() redoes it whole-wavefront from its state in memory -- which this step has not touched, or
  // else is told to rebuild ------------------------------------------------------------------------------------------------------
  {
    unsigned long long fm = __ballot(!ok);
    wave_sync();
    block_store16(S.hdr + (size_t)env0 * SDC_HDR_DWORDS, sh.hdr, lane, fm);     // (the fallback's envs: their headers come from env_reward)
    if (__builtin_expect(fm != 0ull, 0)) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      while (fm) {
        const int e = __ffsll((long long)fm) - 1;
        fm &= fm - 1;
        const int env_e = env0 + e;
        auto pi = [&](const int v) { return __builtin_amdgcn_readlane(v, e); };
        auto pf = [&](const double v) { return readlane_f64(v, e); };
        unsigned hd_e = S.hdr[(size_t)env_e * SDC_HDR_DWORDS + lane];
        if (pi(touched ? 1 : 0)) put_u32(hd_e, H_VALID, 0u);
        const uint4 qw_e = reinterpret_cast<const uint4*>(S.qwin)[(size_t)env_e * SDC_WIN + lane];
        // (ITE / total power, water: inputs of the other reward functions, which only the general form is launched with)
        double e_pit = 0.0, e_tkw = 0.0, e_wat = 0.0;
        if constexpr (GEN) {
          const WideHandGen& G = *reinterpret_cast<const WideHandGen*>(sh.row + WE * 32);
          e_pit = G.p_it[e]; e_tkw = G.total_kw[e]; e_wat = G.water[e];
        }
        env_reward(S, env_e, lane, hd_e, qw_e, pi(hl), pi(slot), (unsigned)pi((int)x_new), (unsigned)pi((int)x_old), pf(e_off), pf(energy),
                   pf(norm_ci), pf(oldest_norm), pi(overdue), pi(hourq_n), e_pit, e_tkw, e_wat, rew,
                   &late.back[0][0] + e * 8 - SDC_INFO_ENERGY_Z, late.tl);
      }
      wave_sync();
      if (!ok) {
        z_f = late.back[lane][0]; path_f = late.back[lane][1]; ret_f[0] = late.back[lane][2]; ret_f[1] = late.back[lane][3]; ret_f[2] = late.back[lane][4];
      }
    }
  }

  WST(14);
  // ---- the five reward-side info columns into the staged rows, the wavefronts' 64 info rows out as 11 KB of whole lines ------------------
  __syncthreads();      // (3) the info rows are staged (minus these columns)
  {
    float* const stage = reinterpret_cast<float*>(&sh.rec[0]);
    float* irow = stage + lane * SDC_INFO_DIM;
    irow[SDC_INFO_ENERGY_Z] = z_f;
    irow[SDC_INFO_RESERVED] = path_f;
    irow[SDC_INFO_EP_RETURN_LS] = ret_f[0];
    irow[SDC_INFO_EP_RETURN_DC] = ret_f[1];
    irow[SDC_INFO_EP_RETURN_BAT] = ret_f[2];
    wave_sync();
    const f4v* s4 = reinterpret_cast<const f4v*>(stage);
    float* i4 = info + (size_t)env0 * SDC_INFO_DIM;
#pragma unroll
    for (int k = 0; k < SDC_INFO_DIM / 4; k++) nt_store4(i4 + 4 * (k * WE + lane), s4[k * WE + lane]);
  }
#ifdef SDC_WIDE_STAMPS
  WST(15);
  __builtin_amdgcn_s_waitcnt(0);
  if (lane < 24) info[(size_t)env0 * SDC_INFO_DIM + lane] = (float)((reinterpret_cast<unsigned long long*>(sh.row + WE * 16 + 896)[lane]) & 0xFFFFFFull);
#endif
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------------
// One launch = one env-step of all N environments.  Grid: S.sweep_blocks sweep workgroups (first: their two wavefronts each serve
// re-centring requests of the previous step, one wavefront per request, while the envs step), then N / 64 env workgroups of TWO
// wavefronts: wavefront 0 integrates the 64 envs' dynamics, wavefront 1 keeps their reward state -- its loads, the windows that
// arrive, everything that does not need the step's energy run BESIDE the dynamics on another SIMD, and once the energy is handed
// over (LDS, one barrier) the dynamics wavefront's outputs and the reward wavefront's normalisation run side by side again.
template <bool GEN>
__device__ __forceinline__ void wide_kernel_body(const SdcDev& S, WideSharedT<GEN>& sh, const int rel_hint, const int32_t* __restrict__ actions,
                                                 float* __restrict__ obs, float* __restrict__ share_obs, unsigned char* __restrict__ done,
                                                 float* __restrict__ info, float* __restrict__ final_obs, float* __restrict__ rew) {
#ifdef SDC_WIDE_STAMPS
  const unsigned long long wst_top = wall_clock64();
#else
  constexpr unsigned long long wst_top = 0ull;
#endif
#ifndef WIDE_KTOUCH
#define WIDE_KTOUCH 1
#endif
#if WIDE_KTOUCH
  const KernargTouch ktouch = kernarg_touch();      // (every line of the kernel arguments behind ONE scalar-cache round trip: sdc_sweep.hpp)
#endif
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / SDC_WAVE));
  const int lane = threadIdx.x % SDC_WAVE;
  const int bx = (int)blockIdx.x;
#if WIDE_KTOUCH
  kernarg_touch_done(ktouch);
#endif
  // Issue priorities.  An env wavefront placed beside an ACTIVE sweep wavefront loses ~2 us of its ~12 us at the sweeps' raised
  // priority (the other kernels' choice; measured at 16 384 envs: the rack-model segment 5.6 instead of 3.7 us in the same 8 % of
  // the workgroups every launch -- and the launch ends with its slowest workgroup).  Up to one env workgroup per CU a sweep is ~5 us
  // of a launch that lasts ~18 us anyway: there it runs BELOW the env wavefronts (12 288 / 16 384 envs: -0.7 us per step).  With
  // more workgroups than that it made no difference or cost a little (24 576 envs +0.2, 65 536 +1.5 us at <= 1024): raised, as before.
  if (bx < S.sweep_blocks) {
    // one wavefront per request: a sweep is 40 KB through one wavefront, ~5 us -- shorter than the step it runs beside
    sdc_rw::TailLds& tl = reinterpret_cast<WideLate*>(wave == 0 ? sh.rec : sh.hdr)->tl;
    const int set = S.step_no % 3;
    const int cnt = min(S.rq_count[set], S.rq_max);
    if (bx == 0 && wave == 0 && lane == 0) S.rq_count[(S.step_no + 2) % 3] = 0;     // the set the NEXT step's requests go to
    if (2 * bx + wave < cnt && (int)gridDim.x - S.sweep_blocks > WIDE_ROOMY_WGS) __builtin_amdgcn_s_setprio(SDC_SWEEP_PRIO);
#pragma unroll 1
    for (int j = 2 * bx + wave; j < cnt; j += 2 * S.sweep_blocks) serve_recentring_request(S, set, j, lane, tl);
    return;
  }
  const int nb = (int)gridDim.x - S.sweep_blocks;
  const int env0 = first_pair_of_block(bx - S.sweep_blocks, nb, 1) * WE;     // (every XCD a contiguous range of envs)
  if (env0 >= S.n_envs) return;
  if (nb <= WIDE_ROOMY_WGS) __builtin_amdgcn_s_setprio(2);
  if (wave == 0) wide_dynamics<GEN>(S, sh, env0, lane, rel_hint, actions, obs, share_obs, done, final_obs, wst_top);
  else wide_rewards<GEN>(S, sh, env0, lane, rel_hint, actions, info, rew);
}

extern "C" __global__ __launch_bounds__(2 * SDC_WAVE) __attribute__((amdgpu_waves_per_eu(1, 2))) void sdc_dynamics_wide_kernel(
    SdcDev S, const int rel_hint, const int32_t* __restrict__ actions, float* __restrict__ obs, float* __restrict__ share_obs,
    unsigned char* __restrict__ done, float* __restrict__ info, float* __restrict__ final_obs, float* __restrict__ rew) {
  __shared__ WideSharedT<false> sh;
  wide_kernel_body<false>(S, sh, rel_hint, actions, obs, share_obs, done, info, final_obs, rew);
}

// ... and its GENERAL form: every lane its own config, rule-based policies, any reward function for the dc / battery agents
// (sdc_capi.hip wide_gen_case).  51 KB of LDS: three workgroups per CU.
extern "C" __global__ __launch_bounds__(2 * SDC_WAVE) __attribute__((amdgpu_waves_per_eu(1, 2))) void sdc_dynamics_wide_gen_kernel(
    SdcDev S, const int rel_hint, const int32_t* __restrict__ actions, float* __restrict__ obs, float* __restrict__ share_obs,
    unsigned char* __restrict__ done, float* __restrict__ info, float* __restrict__ final_obs, float* __restrict__ rew) {
  __shared__ WideSharedT<true> sh;
  wide_kernel_body<true>(S, sh, rel_hint, actions, obs, share_obs, done, info, final_obs, rew);
}
