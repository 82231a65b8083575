// sdc_pairstep.hpp -- the coupled per-timestep dynamics, TWO ENVIRONMENTS PER WAVEFRONT (block = 4 wavefronts = 8 envs).
//
// Most of a step is "scalar" physics: the load-shifting queue algebra, the set-point integrator, chiller / cooling tower /
// water, the battery, the reward arithmetic -- one value per ENV, not per rack.  With one wavefront per env those
// instructions ran on 64 lanes that all held the same number (round 1: 1 400 VALU instructions per env-step, the SIMDs'
// VALU issue was the bound).  Here lanes 0..31 of a wavefront carry env 2w and lanes 32..63 env 2w + 1: every
// per-env instruction is issued ONCE for two envs, the rack model runs lane = rack inside each half (<= 32 racks per
// pass), half-wave reductions stay on the DPP path (+ one v_permlane16_swap to join the two rows of a half), and a
// launch needs half the wavefronts (half the dispatch ramp, two instead of four resident waves per SIMD).
//
// Per-env values live in LDS as the wavefront's "scalar register file": each half stages its env's 256-byte state
// record, the 25 scalars of its data-centre config and its step inputs there with coalesced loads, and every lane reads
// the field it needs (a broadcast read inside its half).
//
// Memory plan per wavefront:
//   level 0  the 2 x 3 actions (hand-issued first), the pair's two records (512 contiguous bytes, one dwordx2 per lane),
//            the config scalars of a single-config job and -- when the host knows the episode step (lock-step batch) --
//            each env's feature row + queue probes: ONE round trip before the dynamics in the usual case;
//   level 1  what level 0 could not address (several configs; a batch that is not in lock-step);
//   behind the staging, consumed late: the queue table ahead of the oldest task (actions that can pop), the evicted ring
//            key, both headers, the rank windows' keys.
// The history-normalised rewards run for both envs at once on the O(1) path (pair_reward_fast: four 64-key rank windows,
// two keys per lane of the half, sdc_halfwin.hpp); an env that needs its ring falls back to the whole-wavefront form
// (env_reward: sdc_trackers.hpp / sdc_ringpath.hpp).
//
// Reference: sustaindc_env.py:533-737 and the sub-environment steps it drives (see per-block citations).
#pragma once
#include "sdc_ringpath.hpp"
#include "sdc_halfwin.hpp"
#include "sdc_quadwin.hpp"
#include "sdc_physics.hpp"

namespace {

constexpr int EPW = 2;    // envs per wavefront

// gather slots (8 bytes each) of one env
enum {
  G_W0 = 0, G_W1, G_W2,   // W[i], W[i+1], W[i+2]
  G_C0,                   // C[i]
  G_T0, G_WB0, G_T1,      // T[i], WB[i], T[i+1] from the env's weather window
  G_LUT,                  // hour LUT {cos, sin} is 16 bytes: two slots
  G_LUT2,
  G_Q97, G_Q24, G_Q48, G_Q72, G_Q96,   // queue prefix counts cum[now - a]
  G_NCN = 14,             // NC[i'+1] (norm_CI of the reward) when the episode has feature rows
  G_C3 = 15,              // C[i+3]: the rule-based battery policy's forecast sample
  G_NC = 16,              // 25 slots: C[i'-16 .. i'+8]
  G_NT = 41,              // 17 slots: T[i' .. i'+16]
  G_END = 58
};

struct PairShared {
  double g[EPW][64];                   // gathered step inputs; g[G_NC..] / g[G_NT..] are normalised in place to NC / NT
  double prm[EPW][HL];                 // config scalars (P_*)
  double osc[EPW][16];                 // step-dependent observation scalars, for the in-step feature path only
  unsigned rec[EPW][SDC_REC_DWORDS];   // state records: loaded, read field by field, patched, stored
  unsigned hdr[EPW][SDC_HDR_DWORDS];   // reward-side headers, likewise (fast path; the slow path works on registers)
  float pool[EPW][32];                 // observation pool (see build_obs_pool)
  float info[EPW][SDC_INFO_DIM];
  unsigned long long dbg_t[2];
  unsigned long long dbg_s[2];
  unsigned dbg_bits;                   // (diagnostics, debug_flags bit 3) which rare paths this wavefront's step took
  sdc_rw::TailLds tl;                  // scratch of the ring paths (window refill, rebuild): one env at a time
};

// sum over the 32 lanes of each half; every lane gets its half's sum.  Same tree as the round-1 64-lane reduction
// restricted to a half (strides 1, 2, 4, 8 inside the rows, then the two rows), so the rack sums round identically.
__device__ __forceinline__ double half_sum_f64(double v) {
  v += dpp_f64<SDC_DPP_XOR1>(v);
  v += dpp_f64<SDC_DPP_XOR2>(v);
  v += dpp_f64<SDC_DPP_HALF_MIRROR>(v);
  v += dpp_f64<SDC_DPP_MIRROR>(v);          // every lane of a row: the row's total
  // v_permlane16_swap exchanges the odd rows of its first operand with the even rows of the second: from two copies
  // of v, a = {r0, r0, r2, r2} and b = {r1, r1, r3, r3}
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto slo = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto shi = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double a = __hiloint2double((int)shi[0], (int)slo[0]), b = __hiloint2double((int)shi[1], (int)slo[1]);
  return b + a;   // (row 1 + row 0, as row_bcast15 added them)
}
// the 32 ballot bits of this lane's half
__device__ __forceinline__ unsigned half_ballot(const bool p, const int h) {
  const unsigned long long m = __ballot(p);
  return h ? (unsigned)(m >> 32) : (unsigned)m;
}

// ---- FOUR ENVS PER WAVEFRONT (the common-case kernels; MapQuad below) ---------------------------------------------------------
// The same step with a DPP row of 16 lanes per env instead of a half: every per-env instruction is issued once for FOUR
// envs, a launch needs a quarter of the wavefronts of one-per-env (half of the pair mapping's: half the dispatch ramp, one
// wavefront per SIMD at 4096 envs), the reductions over an env stay inside a row (no permlane stage), and the rack model
// takes two passes for configs of more than 16 racks.  pair_dynamics and pair_reward_fast are written once for both
// mappings (template parameter M); what differs is spelled `if constexpr (M::LPE == 16)` there.
constexpr int QE = 4;     // envs per wavefront
constexpr int QL = 16;    // lanes per env
struct QuadShared {
  double g[QE][16];                    // gathered step inputs (the slots below G_NC: the common case has feature rows)
  double prm[HL];                      // config scalars (P_*): ONE config in the common case
  double rk[4][HL];                    // ... and its per-rack parameters {supply, idle, full, n} of racks 0..31
  unsigned rec[QE][SDC_REC_DWORDS];
  unsigned hdr[QE][SDC_HDR_DWORDS];
  float pool[QE][32];
  float info[QE][SDC_INFO_DIM];
  unsigned long long dbg_t[2];
  unsigned long long dbg_s[2];
  unsigned dbg_bits;
  sdc_rw::TailLds tl;
};
struct MapPair {
  static constexpr int LPE = HL, ENVS = EPW, KPL = 2;
  using Shared = PairShared;
  using Win = sdc_hw::HWin;
};
struct MapQuad {
  static constexpr int LPE = QL, ENVS = QE, KPL = 4;
  using Shared = QuadShared;
  using Win = sdc_hw::QWin;
};
__device__ __forceinline__ const double* prm_of(const PairShared& sh, const int h) { return sh.prm[h]; }
__device__ __forceinline__ const double* prm_of(const QuadShared& sh, const int) { return sh.prm; }
template <int LPE>
__device__ __forceinline__ unsigned env_ballot(const bool p, const int h) {
  if constexpr (LPE == HL) return half_ballot(p, h);
  else return sdc_hw::row_ballot(p, h);
}

// what the per-env reward part needs from the dynamics, one value per lane (uniform inside a half)
struct DynOut {
  double energy, e_off, norm_ci, oldest_norm, p_it, total_kw, water;
  int overdue, hourq_n, hl, slot;
  unsigned x_new;
};

// LDS record access: this lane's env record
__device__ __forceinline__ int lrec_i32(const unsigned* rp, int idx) { return (int)rp[idx]; }
__device__ __forceinline__ double lrec_f64(const unsigned* rp, int idx) { return *reinterpret_cast<const double*>(rp + idx); }

// ------------------------------------------------------------------------------------------------
// the coupled dynamics at cursor i and the observation at i' = i + 1 of BOTH envs of the wavefront: lane = (half h, l)
//
// FAST (here and in pair_reward_fast / pair_step): the launch is the COMMON CASE, which the host checks before it picks
// the kernel (sdc_capi.hip fast_case) -- every env in lock-step with valid feature rows, one data-centre config, the
// caller's actions on all three slots, the default reward functions, no diagnostics, an even number of envs, all output
// arrays present.  What the general code decides at run time is then a compile-time constant: the same source, the
// same arithmetic in the same order (so both kernels give the same bits), minus the tests, the exec-mask bookkeeping
// around them and the kernel arguments only the other cases read.
// M (MapPair / MapQuad): lanes per env.  With 16 a lane carries TWO entries of the feature row (frow = entry 2l, frow_b =
// 2l + 1) and of the queue table ahead of the oldest task (q_ahead = step head + 2l, q_ahead_b = head + 2l + 1).
template <bool FAST, class M = MapPair>
__device__ __forceinline__ DynOut pair_dynamics(const SdcDev& S, const int envc, const int h, const int l, const int a_ls,
                                                const int a_dc_in, const int a_bat_in, unsigned fault, const bool feat_ok_in,
                                                const float frow, const uint2 q_ahead, const bool q_ahead_ok,
                                                int32_t* __restrict__ actions_out_in, typename M::Shared& sh, const double* kt_lds,
                                                const float frow_b = 0.0f, const uint2 q_ahead_b = make_uint2(0u, 0u)) {
  static_assert(M::LPE == HL || FAST, "four envs per wavefront: the common case only");
  constexpr int LPE = M::LPE;
  typename KSel<FAST>::type kt{};
  if constexpr (FAST) kt.t = kt_lds;
  const bool feat_ok = FAST ? true : feat_ok_in;
  int32_t* const actions_out = FAST ? nullptr : actions_out_in;
  const unsigned* rp = sh.rec[h];
  const double* g = sh.g[h];
  const double* pr = prm_of(sh, h);
  const int i = lrec_i32(rp, R_CURSOR);
  const int rel = lrec_i32(rp, R_TREL);
  const int day = lrec_i32(rp, R_DAY);
  const int hourq = lrec_i32(rp, R_HOURQ);
  const double hour = (double)hourq * 0.25;
  const double wl = g[G_W0], w_ip = g[G_W1], w_ip1 = g[G_W2];
  const double ci_i = g[G_C0];
  const double amb = g[G_T0], wet_bulb = g[G_WB0], amb_next = g[G_T1];
  // norm_CI = NC[i'+1] (sustaindc_env.py:681): from the episode's feature row, or from the window gathered by this step
  const double norm_ci = feat_ok ? g[G_NCN] : g[G_NC + 17];

  const bool lane0 = h == 0 && l == 0;
  SDC_AT(1, sh, lane0);
  // ---- load shifting: envs/carbon_ls.py:172-324 (sdc_physics.hpp ls_algebra) ------------------------------------------------
  if (wl < 0 || wl > 1) fault |= SDC_FAULT_WORKLOAD;
  const uint2* qt = S.qtab + (size_t)envc * S.qstride;
  const int now = rel;
  const int popped0 = lrec_i32(rp, R_QPOPPED);
  const int cum_prev = lrec_i32(rp, R_QCUM);
  const unsigned cumT_prev = (unsigned)lrec_i32(rp, R_QCUMT);
  auto cum_g = [&](int slot) -> int { return (int)(unsigned)__double2loint(g[slot]); };  // .x of the gathered uint2 (0 if t < 0)
  const LsStep ls = ls_algebra(kt, wl, a_ls, popped0, cum_prev, cumT_prev, now, S.queue_max, cum_g(G_Q97), cum_g(G_Q24), cum_g(G_Q48),
                               cum_g(G_Q72), cum_g(G_Q96));
  const int overdue = ls.overdue, popped = ls.popped, dropped = ls.dropped, processed = ls.processed;
  const int cum_now = ls.cum_now, total = ls.total;
  const unsigned cumT_now = ls.cumT_now;
  const double util = ls_utilisation(kt, ls);
  double hist[5];
  const double den = (double)max(total, 1), rden = 1.0 / den;   // (an integer <= 1000: significand never all ones)
  ls_age_hist(ls, den, rden, hist);
  SDC_AT(2, sh, lane0);
  // oldest task: smallest step hd in [head, now] with cum[hd] > popped.  It only moves when tasks were popped
  // (or the queue was empty): then a 32-ary search over the half's lanes (<= 2 rounds) finds it and cum/cumT[hd-1]
  // are cached.
  int head = lrec_i32(rp, R_QHEAD);
  int cum_hm1 = lrec_i32(rp, R_QCUM_HM1);
  unsigned cumT_hm1 = (unsigned)lrec_i32(rp, R_QCUMT_HM1);
  double oldest = 0.0, avg = 0.0;
  {
    const bool was_empty = (cum_prev - popped0) == 0;
    const bool need = total > 0 && !was_empty && popped != popped0;   // this half searches
    // The usual case needs no memory round trip in the middle of the step: the 32 table entries from the old head on
    // were requested with the step's inputs (q_ahead: lane l holds cum / cumT of step head + l) whenever the action can
    // pop tasks; the new head is almost always among them (it moves past <= 90 tasks, and a step's defer adds ~5-15).
    bool need_search = need;
    if constexpr (LPE == HL) {
    if (q_ahead_ok) {
      const int t = head + l;
      const int c = (t == now) ? cum_now : (int)q_ahead.x;
      const unsigned m = half_ballot(need && t <= now && c > popped, h);
      const int f = __ffs((int)m) - 1;
      const int src = ((h << 5) + max(f - 1, 0)) << 2;      // (byte address of the source lane, inside this half)
      const int c_m1 = __builtin_amdgcn_ds_bpermute(src, (int)q_ahead.x), ct_m1 = __builtin_amdgcn_ds_bpermute(src, (int)q_ahead.y);
      if (need && m != 0u) {
        need_search = false;
        if (f > 0) {            // (f == 0: the head stays, and so do the cached cum / cumT of the step before it)
          head += f;
          cum_hm1 = c_m1;
          cumT_hm1 = (unsigned)ct_m1;
        }
      }
    }
    } else {
    if (q_ahead_ok) {
      // the same 32 entries, two per lane of the row: entry index 2 l + {0, 1} = step head + that
      const int t0 = head + 2 * l, t1 = t0 + 1;
      const int c0 = (t0 == now) ? cum_now : (int)q_ahead.x, c1 = (t1 == now) ? cum_now : (int)q_ahead_b.x;
      const unsigned m0 = sdc_hw::row_ballot(need && t0 <= now && c0 > popped, h);
      const unsigned m1 = sdc_hw::row_ballot(need && t1 <= now && c1 > popped, h);
      const bool found = (m0 | m1) != 0u;
      const int f0 = m0 != 0u ? 2 * (__ffs((int)m0) - 1) : 64, f1 = m1 != 0u ? 2 * (__ffs((int)m1) - 1) + 1 : 64;
      const int f = min(f0, f1);
      const int fm1 = found ? max(f - 1, 0) : 0;
      const bool od = (fm1 & 1) != 0;                          // (uniform in the row: every lane offers the entry of that parity)
      const int src = ((h << 4) + (fm1 >> 1)) << 2;
      const int c_m1 = __builtin_amdgcn_ds_bpermute(src, od ? (int)q_ahead_b.x : (int)q_ahead.x);
      const int ct_m1 = __builtin_amdgcn_ds_bpermute(src, od ? (int)q_ahead_b.y : (int)q_ahead.y);
      if (need && found) {
        need_search = false;
        if (f > 0) {
          head += f;
          cum_hm1 = c_m1;
          cumT_hm1 = (unsigned)ct_m1;
        }
      }
    }
    }
    if (__builtin_expect(__ballot(need_search) != 0ull, 0)) {
      SDC_DBG_BIT(FAST, sh, 1u);
      const bool need = need_search;
      int lo = head, hi = now;
      while (__ballot(need && hi - lo + 1 > LPE) != 0ull) {
        const bool act = need && hi - lo + 1 > LPE;
        const int len = hi - lo + 1;
        const int stride = (len + LPE - 1) / LPE;
        const int t = min(lo + (l + 1) * stride - 1, hi);
        int c = 0;
        if (act) c = (t == now) ? cum_now : (int)qt[t].x;
        const unsigned m = env_ballot<LPE>(act && c > popped, h);
        const int f = __ffs((int)m) - 1;  // exists: cum[now] > popped
        if (act) {
          const int nlo = lo + f * stride;
          hi = min(lo + (f + 1) * stride - 1, hi);
          lo = nlo;
        }
      }
      {
        const int t = lo + l;
        int c = 0;
        if (need && t <= hi) c = (t == now) ? cum_now : (int)qt[t].x;
        const unsigned m = env_ballot<LPE>(need && t <= hi && c > popped, h);
        if (need) head = lo + (__ffs((int)m) - 1);
      }
      if (need) {
        if (head == 0) {
          cum_hm1 = 0;
          cumT_hm1 = 0;
        } else if (head == now) {
          cum_hm1 = cum_prev;
          cumT_hm1 = cumT_prev;
        } else {
          const uint2 e = qt[head - 1];
          cum_hm1 = (int)e.x;
          cumT_hm1 = e.y;
        }
      }
    }
    ls_ages(ls, now, cum_prev, cumT_prev, was_empty, den, rden, head, cum_hm1, cumT_hm1, oldest, avg);
  }
  const double normq = sdc_div_const((double)total, S.queue_max_d, S.rc_queue_max);
  const double oldest_norm = KDIV(oldest, 24), avg_norm = KDIV(avg, 24);
  // the load-shifting entries of the observation pool and of the info block leave for LDS here (lane 1 / lane 0 of the
  // half), so that none of them stays in registers across the rack model below
  if (feat_ok) {
    if (l == 1) {
      float* pool = sh.pool[h];
      pool[SDC_P_OLDEST] = (float)oldest_norm;
      pool[SDC_P_AVG] = (float)avg_norm;
      pool[SDC_P_NORMQ] = (float)normq;
      for (int b = 0; b < 5; b++) pool[SDC_P_HIST + b] = (float)hist[b];
    }
  } else if (l == 0) {
    if constexpr (!FAST) {
      double* o = sh.osc[h];
      o[5] = normq; o[6] = oldest_norm; o[7] = avg_norm;
      for (int b = 0; b < 5; b++) o[8 + b] = hist[b];
    }
  }
  if (l == 0) {
    const InfoLs il = {wl, util, normq, oldest_norm, avg_norm, hour, total, dropped, processed, overdue};
    info_put_ls(sh.info[h], il, hist);
  }

  SDC_AT(3, sh, lane0);
  // ---- rule-based policies for agent_dc / agent_bat (sdc_config.policy; 0 = the caller's action) ----------------------
  int a_dc = a_dc_in, a_bat = a_bat_in;
  int tr_count = lrec_i32(rp, R_TR_COUNT);
  if (!FAST && S.policy[1] == SDC_POLICY_TRIM_AND_RESPOND) a_dc = trim_and_respond_action(S.tr_limit, lrec_f64(rp, R_LAST_ROOM), tr_count);
  if (!FAST && S.policy[2] == SDC_POLICY_RBC) a_bat = rbc_battery_action(g[G_C3], ci_i, lrec_f64(rp, R_CI_MIN), lrec_f64(rp, R_CI_DEN));

  // ---- CRAC set-point integrator: envs/dc_gym.py:160-174 ----------------------------------------
  if (util < 0.0 || util > 1.0) fault |= SDC_FAULT_CPU_LOAD;
  const int delta = a_dc - 1;  // make_envs_pyenv.py:127-131
  int consecutive = lrec_i32(rp, R_CONSEC), scale = lrec_i32(rp, R_SCALE);
  const double stpt = setpoint_step(a_dc, lrec_i32(rp, R_LAST_DELTA), consecutive, scale, lrec_f64(rp, R_STPT), pr[P_MAX_TEMP], pr[P_MIN_TEMP]);

  SDC_AT(4, sh, lane0);
  // ---- rack model, lane = rack inside the half: envs/datacenter.py:250-317, :157-181 ------------------
  // (the common case with four envs per wavefront is one config; with two it may be several: the config id is in the record)
  const sdc_dc_params& P = S.dc[((FAST && LPE != HL) || (FAST && S.n_cfg == 1)) ? 0 : lrec_i32(rp, R_CFG)].p;
  const int R = (int)pr[P_N_RACKS];
  const double load_pct = util * 100;
  double pcpu = 0.0, pfan = 0.0, outlet = 0.0;
  double outlet_a = 0.0, pw_a = 0.0;       // (16 lanes per env: the first rack pass)
  bool bad_delta = false;
  {
    // (the rack's expressions: sdc_physics.hpp rack_point -- x^y as exp2(y log2 x), no library-function fallback: its ~50 constants would
    // be materialised in front of this loop on every step)
    const RackEnv E = {pr[P_M_CPU], pr[P_C_CPU], pr[P_M_FAN], pr[P_C_FAN], pr[P_RS_CPU] * KDIV(load_pct, 100), pr[P_RS_FAN] * KDIV(load_pct, 20),
                       pr[P_ITFAN_REF_P], pr[P_RC_ITFAN_REF_V_RATIO], pr[P_IT_FAN_FULL_LOAD_V], pr[P_K_OUTLET]};
    auto rack = [&](const int rk, const bool valid) __attribute__((always_inline)) {
      // (four envs per wavefront: the rack's four parameters come from LDS, where the step's FIRST loads left them -- read
      // from the config in memory here they are a memory round trip in the middle of the dynamics, which a wavefront
      // without much company on its SIMD waits out: sdc_rollout at 8 192 envs 15.8 -> 15.1 us per step.  Two envs per
      // wavefront: left where they are used -- ahead of the dynamics they join a burst of loads, and the step launch of
      // 4096 envs measured 11.9-12.0 us against 11.75 with the loads in place, same box.)
      double r_supply, r_idle, r_full, r_n;
      if constexpr (FAST && LPE != HL) {
        r_supply = sh.rk[0][rk]; r_idle = sh.rk[1][rk]; r_full = sh.rk[2][rk]; r_n = sh.rk[3][rk];
      } else {
        r_supply = P.rack_supply[rk]; r_idle = P.rack_idle[rk]; r_full = P.rack_full[rk]; r_n = P.rack_n[rk];
      }
      double inlet;
      const RackOut ro = rack_point(kt, E, r_n, r_supply, r_full, r_idle, stpt, inlet);
      if (valid && (ro.out - inlet < 2 || !ro.plain)) bad_delta = true;
      pcpu += valid ? ro.pc : 0.0;
      pfan += valid ? ro.pf : 0.0;
      outlet += valid ? ro.out : 0.0;
    };
    if constexpr (FAST && LPE == QL) {
      // 16 lanes per env: racks 0..15, then 16..31 (the sums of the two passes are reduced apart and added -- rows 1 + 0
      // of the half-wave tree -- so that both mappings round alike)
      rack(l, l < R);
      outlet_a = outlet;
      pw_a = pcpu + pfan;
      outlet = 0.0; pcpu = 0.0; pfan = 0.0;
      rack(l + QL, l + QL < R);
    } else if constexpr (FAST) {
      // the host has checked that the config has <= 32 racks: ONE pass, every lane of the half computing (lanes without
      // a rack on the table's unused entries: their results are dropped by selects) -- straight-line code.  A loop here
      // makes the compiler fetch all of the body's ~25 constants in front of it and hold them in registers across it.
      rack(l, l < R);
    } else {
#pragma unroll 1
      for (int rk = l; rk < R; rk += HL) rack(rk, true);      // (one pass for the shipped 16 / 20 / 25-rack configs)
    }
  }
  SDC_AT(5, sh, lane0);
  if (env_ballot<LPE>(bad_delta, h) != 0u) fault |= SDC_FAULT_OUTLET_DELTA;
  // (ONE reduction for CPU + fan power: only their total is used.  The reference sums the two lists separately and adds
  // the totals; the difference is a rounding of the last place)
  // CRAC return temperature (datacenter.py:531-541: the mean of return approach + outlet over the racks): the approach
  // temperatures are constants of the config, their sum comes from the host
  const double sum_outlet = LPE == HL ? half_sum_f64(outlet) : sdc_hw::row_sum_f64(outlet) + sdc_hw::row_sum_f64(outlet_a);
  const double avg_ret = (pr[P_RET_SUM] + sum_outlet) * pr[P_RC_N_RACKS];
  const double mean_outlet = sum_outlet * pr[P_RC_N_RACKS];
  const double p_it = LPE == HL ? half_sum_f64(pcpu + pfan) : sdc_hw::row_sum_f64(pcpu + pfan) + sdc_hw::row_sum_f64(pw_a);

  SDC_AT(6, sh, lane0);
  // ---- HVAC: envs/datacenter.py:432-474 ; water :325-353 (sdc_physics.hpp hvac_water) ------------------------------------------
  const HvacPrm HP = {pr[P_C_AIR], pr[P_RHO_AIR], pr[P_CT_FAN_REF_P], pr[P_CRAC_SUPPLY_PU], pr[P_RC_RHO_AIR], pr[P_RC_CTAFR]};
  const HvacOut hv = hvac_water(kt, HP, p_it, avg_ret, stpt, amb, wet_bulb);
  const double comp = hv.comp, ct = hv.ct, water = hv.water, total_kw = hv.total_kw;

  SDC_AT(7, sh, lane0);
  // ---- battery: envs/bat_env_fwd_view.py:84-245, envs/battery_model.py:94-132 (sdc_physics.hpp battery_step) ----------------------
  double bat_load = lrec_f64(rp, R_BAT);
  const BatOut bo = battery_step(kt, a_bat, bat_load, pr[P_BAT_CAP], pr[P_RC_BAT_CAP], total_kw, ci_i, fault);
  const double e_nobat = bo.e_nobat, energy = bo.energy, co2 = bo.co2, soc_after = bo.soc_after;

  SDC_AT(8, sh, lane0);
  // ---- time: utils/managers.py:127-147 -------------------------------------------------------------
  int hourq_n = hourq + 1, day_n = day;
  if (hourq_n >= 96) {
    hourq_n = 0;
    day_n += 1;
  }
  const int ip = i + 1;

  // ---- observations at i' (sustaindc_env.py:565-585) ----------------------------------------------------------------
  if (feat_ok) {
    // the trace-only entries come from the episode's feature row (sdc_features.hip), one float per lane of the half;
    // lane 1 adds the nine entries that depend on the step
    constexpr unsigned TRACE_ONLY = 0x7u | (0x7Fu << SDC_P_CI7) | (1u << SDC_P_W) | (1u << SDC_P_NT) | (1u << SDC_P_TSLOPE) |
                                    (0x1Fu << SDC_P_T5) | (1u << SDC_P_WNEXT) | (1u << SDC_P_NTNEXT);
    float* pool = sh.pool[h];
    if constexpr (LPE == HL) {
      if (l < SDC_POOL_DIM && ((TRACE_ONLY >> l) & 1u)) pool[l] = frow;
    }      // (four envs per wavefront: quad_step has put them there with the staging -- two registers fewer across the dynamics)
    if (l == 1) pool[SDC_P_SOC] = (float)soc_after;
  } else if (l == 0) {
    // no feature rows for this episode: the wavefront computes the features of this env below (whole-wave, per env)
    if constexpr (!FAST) {
      double* o = sh.osc[h];
      o[0] = g[G_LUT]; o[1] = g[G_LUT2]; o[2] = w_ip; o[3] = w_ip1; o[4] = soc_after;
      o[13] = ip >= 16 ? 1.0 : 0.0;
    }
  }

  // ---- history append (utils/reward_creator.py:7-14) --------------------------------------------------------
  // The ring holds fp32 OFFSETS from the env's first energy value (kept in fp64): normalize_energy is
  // shift-invariant, and offsets keep the fp32 rounding error proportional to the spread of the history
  // instead of to the ~300 kWh magnitude (two nearly equal energies would otherwise lose the z-score).
  // Stored as order-preserving keys for the order-statistic trackers.
  // As in the reference, only default_ls_reward appends (reward_creator.py:63): with another ls reward method the
  // history stays as it is and the other agents' footprint rewards are normalised against it.
  const bool append = FAST ? true : S.reward_method[0] == SDC_REWARD_DEFAULT;
  int hl = lrec_i32(rp, R_HIST_LEN), hpos = lrec_i32(rp, R_HIST_POS);
  const double href = hl == 0 ? energy : lrec_f64(rp, R_HIST_REF);
  const double e_off = energy - href;
  const int slot = append ? hist_append_slot(hl, hpos, S.hist_cap) : -1;
  const unsigned x_new = sdc_f32_key(__float_as_uint((float)e_off));

  SDC_AT(9, sh, lane0);
  wave_sync();     // every lane has read what it needs from the records: lane 0 of each half may now patch its record
  if (l == 0) {
    // ---- info block --------------------------------------------------------------------------------
    const unsigned f_all = (unsigned)lrec_i32(rp, R_FAULT) | fault;
    const InfoDc id = {p_it, ct, comp, total_kw, stpt, mean_outlet, amb, water, soc_after, co2, ci_i, e_nobat, energy, norm_ci, amb_next,
                       delta, a_bat, day_n, hourq_n, rel + 1, f_all};
    info_put_dc(kt, sh.info[h], id);

    // ---- history append: the ring slot gets this step's key; the queue table this step's prefix counts ----------
    if (append) {
      S.hist[(size_t)envc * SDC_HIST_STRIDE + slot] = x_new;
      hist_t_append(S, envc, slot, x_new);
    }
    S.qtab[(size_t)envc * S.qstride + now] = make_uint2((unsigned)cum_now, cumT_now);
    qcum_append(S, envc, now, (unsigned)cum_now);

    // ---- new state record ------------------------------------------------------------------------------------
    unsigned* o = sh.rec[h];
    o[R_CURSOR] = (unsigned)ip;
    o[R_TREL] = (unsigned)(rel + 1);
    o[R_DAY] = (unsigned)day_n;
    o[R_HOURQ] = (unsigned)hourq_n;
    o[R_QPOPPED] = (unsigned)popped;
    o[R_QCUM] = (unsigned)cum_now;
    o[R_QCUMT] = cumT_now;
    o[R_QHEAD] = (unsigned)head;
    o[R_QCUM_HM1] = (unsigned)cum_hm1;
    o[R_QCUMT_HM1] = cumT_hm1;
    o[R_LAST_DELTA] = (unsigned)delta;
    o[R_CONSEC] = (unsigned)consecutive;
    o[R_SCALE] = (unsigned)scale;
    o[R_HIST_LEN] = (unsigned)hl;
    o[R_HIST_POS] = (unsigned)hpos;
    o[R_FAULT] = f_all;
    o[R_TR_COUNT] = (unsigned)tr_count;
    *reinterpret_cast<double*>(o + R_STPT) = stpt;
    *reinterpret_cast<double*>(o + R_BAT) = bat_load;
    *reinterpret_cast<double*>(o + R_HIST_REF) = href;
    *reinterpret_cast<double*>(o + R_LAST_ROOM) = mean_outlet;
    if (actions_out) {     // the actions the step applied (rule-based policies: what they chose)
      actions_out[(size_t)envc * 3 + 0] = a_ls;
      actions_out[(size_t)envc * 3 + 1] = a_dc;
      actions_out[(size_t)envc * 3 + 2] = a_bat;
    }
  }
  SDC_AT(10, sh, lane0);
  DynOut o;
  o.energy = energy; o.e_off = e_off; o.norm_ci = norm_ci; o.oldest_norm = oldest_norm; o.p_it = p_it;
  o.total_kw = total_kw; o.water = water; o.overdue = overdue; o.hourq_n = hourq_n; o.hl = hl; o.slot = slot;
  o.x_new = x_new;
  return o;
}

// per-env scalars out of the halves: lane 32 * e holds env e's value
__device__ __forceinline__ int pick_i32(int v, int e) { return __builtin_amdgcn_readlane(v, e * HL); }
__device__ __forceinline__ double pick_f64(double v, int e) { return readlane_f64(v, e * HL); }

// ------------------------------------------------------------------------------------------------
// rewards of ONE env (utils/reward_creator.py:16-130), whole wavefront, wave-uniform control flow.  Four rank windows and
// running sums (sdc_trackers.hpp) normally answer without reading the history ring; a miss rebuilds them from the ring
// right here.  hd0: the env's header (lane i = dword i), qw: its rank windows (lane i = key i of each).
__device__ __forceinline__ void env_reward(const SdcDev& S, const int env, const int lane, const unsigned hd0, const uint4 qw,
                                           const int hl, const int slot, const unsigned x_new_v, const unsigned x_old,
                                           const double e_off, const double energy, const double norm_ci,
                                           const double oldest_norm, const int overdue, const int hourq_n, const double p_it,
                                           const double total_kw, const double water, float* __restrict__ rew,
                                           float* __restrict__ inf_row, sdc_rw::TailLds& tl) {
  using namespace sdc_rw;
  const bool append = S.reward_method[0] == SDC_REWARD_DEFAULT;
  const unsigned x_new = sfl(x_new_v);
  const int n = (int)sfl((unsigned)hl);
  const bool has_old = append && x_old != KEY_NONE;
  unsigned o0 = hd0;
  double mean = 0.0, sd = 0.0, inv_sd = 1.0;
  int path = 0;   // diagnostics: 0 no ring read, 1 a window re-centred ahead of need, 3 rebuilt
  const RingView R = {reinterpret_cast<const uint4*>(S.hist + (size_t)env * SDC_HIST_STRIDE), slot, x_new};
  if (__builtin_expect(n >= 2, 1)) {
    int k1, k3;
    quartile_ranks(n, k1, k3);
    // quartile windows q1 / q3; clip-bound windows bu (upper bound, keys as they are) / bl (lower bound, keys
    // complemented, so that on both sides "beyond the bound" means "at or above it")
    QTrack q1 = qt_load(hd0, H_Q1, qw.x), q3 = qt_load(hd0, H_Q3, qw.y);
    QTrack bu = qt_load(hd0, H_BU, qw.z), bl = qt_load(hd0, H_BL, qw.w);
    bool wd1 = false, wd3 = false, wdu = false, wdl = false;   // a window goes back to memory only if its lanes changed
    double A1 = rec_f64(hd0, H_A1), A2 = rec_f64(hd0, H_A2);
    bool ok = n >= SMALL_N && qt_valid(q1) && qt_valid(q3) && qt_valid(bu) && qt_valid(bl) && rec_i32(hd0, H_VALID) == 1;
    int why = ok ? 0 : 1;                            // diagnostics (debug_flags bit 1): why a rebuild was needed
    if (__builtin_expect(ok && append, 1)) {
      // O(1) updates: running sums, the four windows
      const double vn = key_f64(x_new), vo = has_old ? key_f64(x_old) : 0.0;
      const int n_prev = has_old ? n : n - 1;
      A1 += vn - vo;
      A2 += vn * vn - vo * vo;
      wd1 = qt_update(q1, x_new, x_old, has_old, n_prev, lane);
      wd3 = qt_update(q3, x_new, x_old, has_old, n_prev, lane);
      wdu = qt_update(bu, x_new, x_old, has_old, n_prev, lane);
      wdl = qt_update(bl, ~x_new, ~x_old, has_old, n_prev, lane);
      if (!(qt_valid(q1) && qt_valid(q3) && qt_valid(bu) && qt_valid(bl))) { ok = false; why = 2; }
    }
    unsigned kb0 = 0u, kb1 = 0u;
    // running (count, sum v, sum v^2) over the keys at or beyond each clip bound
    int qc0 = 0, qc1 = 0;
    double qs1_0 = 0.0, qs1_1 = 0.0, qs2_0 = 0.0, qs2_1 = 0.0;
    bool done_eval = false;
    if (__builtin_expect(ok, 1)) {
      unsigned a1, b1, a3, b3;
      if (__builtin_expect(qt_resolve(q1, k1, n, a1, b1) && qt_resolve(q3, k3, n, a3, b3), 1)) {
        const Bounds b = clip_bounds(n, a1, b1, a3, b3);
        kb0 = b.kub;               // upper tail: keys >= kub
        kb1 = ~(b.klb - 1u);       // lower tail, flipped: ~x >= ~(klb-1)  <=>  x < klb
  // first the value that came and the one that went against last step's bounds kbl, then the keys the bounds
        // have moved across since -- which a bound's window lists, as long as both the old and the new bound lie
        // inside its span
        const unsigned kbl0 = (unsigned)rec_i32(hd0, H_KB), kbl1 = (unsigned)rec_i32(hd0, H_KB + 1);
        qc0 = rec_i32(hd0, H_QC);
        qc1 = rec_i32(hd0, H_QC + 1);
        qs1_0 = rec_f64(hd0, H_QS1);
        qs1_1 = rec_f64(hd0, H_QS1 + 2);
        qs2_0 = rec_f64(hd0, H_QS2_HI);
        qs2_1 = rec_f64(hd0, H_QS2_LO);
        if (append) {
          const double vn = key_f64(x_new), vo = key_f64(x_old);
          if (has_old && x_old >= kbl0) { qc0 -= 1; qs1_0 -= vo; qs2_0 -= vo * vo; }
          if (has_old && ~x_old >= kbl1) { qc1 -= 1; qs1_1 -= vo; qs2_1 -= vo * vo; }
          if (x_new >= kbl0) { qc0 += 1; qs1_0 += vn; qs2_0 += vn * vn; }
          if (~x_new >= kbl1) { qc1 += 1; qs1_1 += vn; qs2_1 += vn * vn; }
        }
        const unsigned lo0 = min(kb0, kbl0), hi0 = max(kb0, kbl0), lo1 = min(kb1, kbl1), hi1 = max(kb1, kbl1);
        const bool forced = bound_repair_forced(S, env);
        const bool cov0 = (kb0 == kbl0 || qt_spans(bu, lo0, hi0, n)) && !forced, cov1 = (kb1 == kbl1 || qt_spans(bl, lo1, hi1, n)) && !forced;
        if (__builtin_expect(cov0 && cov1, 1)) {
          int dc0 = 0, dc1 = 0;
          double d1_0 = 0.0, d2_0 = 0.0, d1_1 = 0.0, d2_1 = 0.0;
          const bool x0 = kb0 != kbl0 && win_crossing(bu, lo0, hi0, 0u, dc0, d1_0, d2_0);
          const bool x1 = kb1 != kbl1 && win_crossing(bl, lo1, hi1, KEY_NONE, dc1, d1_1, d2_1);
          if (__builtin_expect(x0, 0)) {
            const double sg = kb0 > kbl0 ? -1.0 : 1.0;   // bound moved out: the keys in between leave the tail
            qc0 += (kb0 > kbl0 ? -1 : 1) * (int)wave_sum_u32((unsigned)dc0);
            qs1_0 += sg * wave_sum_f64(d1_0);
            qs2_0 += sg * wave_sum_f64(d2_0);
          }
          if (__builtin_expect(x1, 0)) {
            const double sg = kb1 > kbl1 ? -1.0 : 1.0;
            qc1 += (kb1 > kbl1 ? -1 : 1) * (int)wave_sum_u32((unsigned)dc1);
            qs1_1 += sg * wave_sum_f64(d1_1);
            qs2_1 += sg * wave_sum_f64(d2_1);
          }
  // sum (v - bound), sum (v^2 - bound^2) over the keys beyond the bounds
          const double t1 = (qs1_0 - (double)qc0 * b.ub) + (qs1_1 - (double)qc1 * b.lb);
          const double t2 = (qs2_0 - (double)qc0 * (b.ub * b.ub)) + (qs2_1 - (double)qc1 * (b.lb * b.lb));
          clipped_moments(n, b, A1, A2, t1, t2, mean, sd, inv_sd, S.hist_cap, S.rc_hist_cap, S.hist_cap_d);
          done_eval = true;
        } else {
          // A CLIP BOUND LEFT ITS WINDOW (the quartiles moved it further in one step than the window reaches, usually while the window's
          // deferred re-centring is still in flight): ~4e-8 of the env-steps.  Everything else the state holds is good -- the quartile
          // windows, the bounds they give -- and this step's moments need the bounds' TAIL SUMS, not their windows: the sums of the NEW
          // bounds come from the ring in one pass (ring_sums; the totals fresh with them), ~10 us instead of the full rebuild's 0.4 ms
          // (two 32-pass bisections).  The window is then moved towards the rank of the first key beyond its bound by the
          // ahead-of-need refill below (12-45 ranks per sweep); should the bound have jumped further, the next step lands here again.
          why = cov0 ? 7 : 6;
          const RingSums rs = ring_sums(R, lane, b);
          A1 = rs.A1; A2 = rs.A2;
          qc0 = rs.qc[0]; qc1 = rs.qc[1];
          qs1_0 = rs.qs1[0]; qs1_1 = rs.qs1[1];
          qs2_0 = rs.qs2[0]; qs2_1 = rs.qs2[1];
          const double t1 = (qs1_0 - (double)qc0 * b.ub) + (qs1_1 - (double)qc1 * b.lb);
          const double t2 = (qs2_0 - (double)qc0 * (b.ub * b.ub)) + (qs2_1 - (double)qc1 * (b.lb * b.lb));
          clipped_moments(n, b, A1, A2, t1, t2, mean, sd, inv_sd, S.hist_cap, S.rc_hist_cap, S.hist_cap_d);
          done_eval = true;
          path = 5 + ((S.debug_flags & 2) ? why : 0);      // 5: the clip bounds' tail sums redone from the ring
        }
      } else {
        why = 4;
      }
    }
    bool valid = true;
    if (__builtin_expect(!done_eval, 0)) {
      // miss (no state yet, a window that did not cover, an inconsistency): rebuild everything from the ring
      const Rebuilt rb = rebuild_state(R, lane, n, tl);
      if (n < SMALL_N || !rb.ok) {   // tiny history (nothing to keep), or a ring no window can describe
        mean = rb.mean;
        sd = rb.sd;
        inv_sd = sd > 0 ? 1.0 / sd : 1.0;
        q1.hi = q3.hi = bu.hi = bl.hi = 0;
        valid = false;
      } else {
        q1 = rb.q1;
        q3 = rb.q3;
        bu = rb.bu;
        bl = rb.bl;
        wd1 = wd3 = wdu = wdl = true;
        A1 = rb.A1;
        A2 = rb.A2;
        kb0 = rb.b.kub;
        kb1 = ~(rb.b.klb - 1u);
        qc0 = rb.qc[0]; qc1 = rb.qc[1];
        qs1_0 = rb.qs1[0]; qs1_1 = rb.qs1[1];
        qs2_0 = rb.qs2[0]; qs2_1 = rb.qs2[1];
        const double t1 = (qs1_0 - (double)qc0 * rb.b.ub) + (qs1_1 - (double)qc1 * rb.b.lb);
        const double t2 = (qs2_0 - (double)qc0 * (rb.b.ub * rb.b.ub)) + (qs2_1 - (double)qc1 * (rb.b.lb * rb.b.lb));
        clipped_moments(n, rb.b, A1, A2, t1, t2, mean, sd, inv_sd);
      }
      path = 3 + ((S.debug_flags & 2) ? why : 0);
    }
    put_u32(o0, H_KB, kb0);
    put_u32(o0, H_KB + 1, kb1);
    put_u32(o0, H_VALID, valid ? 1u : 0u);
    put_f64(o0, H_A1, A1);
    put_f64(o0, H_A2, A2);
    put_running_tails(o0, qc0, qc1, qs1_0, qs1_1, qs2_0, qs2_1);
    // Windows AHEAD of need: if, in the worst case for the keys the next step removes and adds, a window would no
    // longer cover what is asked of it, re-centre it now -- at the end of this wavefront's life, when the memory
    // system is quiet and the other wavefronts of its SIMD are finishing -- instead of at the start of the next
    // launch, where the sweep's loads would queue behind every env's start-of-step traffic.
    if (__builtin_expect(valid && n >= SMALL_N, 1)) {
      int k1n, k3n;
      quartile_ranks((append && n < S.hist_cap) ? n + 1 : n, k1n, k3n);
      // a bound's window is centred on the rank of the first key beyond the bound
      const int req = qt_refill_ahead(q1, k1n, n, 3, 6) | (qt_refill_ahead(q3, k3n, n, 3, 6) << 2) |
                      (qt_refill_ahead(bu, n - qc0, n, 10, 10) << 4) | (qt_refill_ahead(bl, n - qc1, n, 10, 10) << 6);
      if (__builtin_expect(req != 0, 0)) {
        __builtin_amdgcn_s_setprio(3);   // the step ends when the slowest wavefront does: let this one issue first
        // one copy of the refill code: the windows take turns through it
#pragma unroll 1
        for (int t = 0; t < 4; t++) {
          const int d = (req >> (2 * t)) & 3;
          if (d == REFILL_NONE) continue;
          QTrack A = t == 0 ? q1 : (t == 1 ? q3 : (t == 2 ? bu : bl));
          const int kt = t == 0 ? k1n : (t == 1 ? k3n : (t == 2 ? n - qc0 : n - qc1));
          qt_refill(A, d, kt, n, R, lane, tl, t == 3 ? KEY_NONE : 0u);
          if (t == 0) { q1 = A; wd1 = true; }
          if (t == 1) { q3 = A; wd3 = true; }
          if (t == 2) { bu = A; wdu = true; }
          if (t == 3) { bl = A; wdl = true; }
        }
        path = max(path, 1);
      }
    }
    qt_put(o0, H_Q1, q1);
    qt_put(o0, H_Q3, q3);
    qt_put(o0, H_BU, bu);
    qt_put(o0, H_BL, bl);
    // first / last key of every window (what the O(1) path's outside-the-window test reads)
    put_u32(o0, H_WFIRST + 0, lane_key(q1.w, 0)); put_u32(o0, H_WLAST + 0, lane_key(q1.w, max(q1.hi - 1, 0)));
    put_u32(o0, H_WFIRST + 1, lane_key(q3.w, 0)); put_u32(o0, H_WLAST + 1, lane_key(q3.w, max(q3.hi - 1, 0)));
    put_u32(o0, H_WFIRST + 2, lane_key(bu.w, 0)); put_u32(o0, H_WLAST + 2, lane_key(bu.w, max(bu.hi - 1, 0)));
    put_u32(o0, H_WFIRST + 3, lane_key(bl.w, 0)); put_u32(o0, H_WLAST + 3, lane_key(bl.w, max(bl.hi - 1, 0)));
    if (wd1 || wd3 || wdu || wdl)
      reinterpret_cast<uint4*>(S.qwin)[(size_t)env * SDC_WIN + lane] = make_uint4(q1.w, q3.w, bu.w, bl.w);
  } else {
    put_u32(o0, H_VALID, 0u);
  }
  put_u32(o0, H_N, (unsigned)n);
  // (this path re-centres inline and rebuilds: deferred re-centrings in flight for this env are dropped)
  put_u32(o0, H_PEND, 0u); put_u32(o0, H_PEND + 1, 0u); put_u32(o0, H_PEND + 2, 0u); put_u32(o0, H_PEND + 3, 0u);
  put_f64(o0, H_EOFF, e_off);                                 // bat_total_energy_with_battery_KWh - hist_ref
  const double z = n < 2 ? 0.0 : (e_off - mean) * inv_sd;
  const RewardIn rin = {z, norm_ci, oldest_norm, (double)overdue, energy, (double)hourq_n * 0.25, SDC_DIV_CONST(p_it, 1e3), total_kw, water};
  const Rewards rr = step_rewards(rin, S.reward_method, hd0);
  put_f64(o0, H_RET, rr.ret[0]);
  put_f64(o0, H_RET + 2, rr.ret[1]);
  put_f64(o0, H_RET + 4, rr.ret[2]);
  if (lane == 0) store_rewards(rr, z, path, env, rew, inf_row);
  __builtin_nontemporal_store(o0, &S.hdr[(size_t)env * SDC_HDR_DWORDS + lane]);
}

// ------------------------------------------------------------------------------------------------
// rewards of BOTH envs at once (utils/reward_creator.py:16-130): the O(1) path of env_reward() below in half-wave form
// (sdc_halfwin.hpp) -- running sums, the four rank windows, clip bounds, tail sums, moments, z, the three rewards, the new
// header -- same arithmetic in the same order, so both paths give the same bits.  An env whose step needs anything else
// (no valid reward state yet, a window that does not cover, a window to re-centre ahead of need, fewer than SMALL_N
// keys, a non-appending reward configuration) is left untouched and reported in the returned mask: env_reward()
// then redoes it from its unmodified state.  wa / wb: the lane's keys 2l / 2l + 1 of {Q1, Q3, BU, BL}.
// Returns the ballot of lanes whose env was completed here.
// a window of either mapping from / to its KPL keys per lane
__device__ __forceinline__ sdc_hw::HWin win_from(const unsigned (&k)[2], const int r0, const int hi) { return sdc_hw::HWin{k[0], k[1], r0, hi}; }
__device__ __forceinline__ sdc_hw::QWin win_from(const unsigned (&k)[4], const int r0, const int hi) {
  return sdc_hw::QWin{k[0], k[1], k[2], k[3], r0, hi};
}
__device__ __forceinline__ void win_keys(const sdc_hw::HWin& q, unsigned (&k)[2]) { k[0] = q.a; k[1] = q.b; }
__device__ __forceinline__ void win_keys(const sdc_hw::QWin& q, unsigned (&k)[4]) { k[0] = q.k0; k[1] = q.k1; k[2] = q.k2; k[3] = q.k3; }
template <int LPE>
__device__ __forceinline__ double env_sum_f64(const double v) {
  if constexpr (LPE == HL) return half_sum_f64(v);
  else return sdc_hw::row_sum_f64(v);
}
template <int LPE>
__device__ __forceinline__ unsigned env_sum_u32(const unsigned v) {
  if constexpr (LPE == HL) return sdc_hw::half_sum_u32(v);
  else return sdc_hw::row_sum_u32(v);
}

// wk[i]: keys KPL l + i of the four windows {Q1, Q3, BU, BL} (one uint4 per key position, as they lie in SdcDev::qwin)
template <bool FAST, class M = MapPair>
__device__ __forceinline__ unsigned long long pair_reward_fast(const SdcDev& S, const int envc, const bool active, const int h,
                                                               const int l, const uint4 (&wk)[M::KPL], const DynOut& d,
                                                               const unsigned x_old, float* __restrict__ rew,
                                                               typename M::Shared& sh, const int step_no, const bool defer) {
  using namespace sdc_rw;
  using namespace sdc_hw;
  using Win = typename M::Win;
  constexpr int LPE = M::LPE, KPL = M::KPL;
  const unsigned* hp = sh.hdr[h];
  const bool lane0 = h == 0 && l == 0;
  SDC_AT(11, sh, lane0);
  const int n = d.hl;
  const bool append = FAST ? true : S.reward_method[0] == SDC_REWARD_DEFAULT;
  const bool has_old = x_old != KEY_NONE;
  const unsigned x_new = d.x_new;
  int k1, k3;
  quartile_ranks(n, k1, k3);
  unsigned kx[KPL], ky[KPL], kz[KPL], kw[KPL];
#pragma unroll
  for (int i = 0; i < KPL; i++) { kx[i] = wk[i].x; ky[i] = wk[i].y; kz[i] = wk[i].z; kw[i] = wk[i].w; }
  Win q1 = win_from(kx, (int)hp[H_Q1 + T_R0], (int)hp[H_Q1 + T_HI]);
  Win q3 = win_from(ky, (int)hp[H_Q3 + T_R0], (int)hp[H_Q3 + T_HI]);
  Win bu = win_from(kz, (int)hp[H_BU + T_R0], (int)hp[H_BU + T_HI]);
  Win bl = win_from(kw, (int)hp[H_BL + T_R0], (int)hp[H_BL + T_HI]);
  double A1 = lrec_f64(hp, H_A1), A2 = lrec_f64(hp, H_A2);
  bool ok = append & (n >= SMALL_N) & (q1.hi > 0) & (q3.hi > 0) & (bu.hi > 0) & (bl.hi > 0) & ((int)hp[H_VALID] == 1);
  if constexpr (!FAST) ok = ok & !bound_repair_forced(S, envc);     // (test hook, debug_flags bit 13: env_reward takes the step)
  // ---- deferred re-centrings (SdcRefillReq / SdcRefillRes): a window requested two steps ago arrives now --------------
  unsigned pend0 = hp[H_PEND], pend1 = hp[H_PEND + 1], pend2 = hp[H_PEND + 2], pend3 = hp[H_PEND + 3];
  // cached first / last key of every window (see below)
  unsigned wf0 = hp[H_WFIRST], wf1 = hp[H_WFIRST + 1], wf2 = hp[H_WFIRST + 2], wf3 = hp[H_WFIRST + 3];
  unsigned wl0 = hp[H_WLAST], wl1 = hp[H_WLAST + 1], wl2 = hp[H_WLAST + 2], wl3 = hp[H_WLAST + 3];
  bool wdc = false;    // a window was replaced
  bool filed = false;  // (diagnostics) a re-centring request was filed
  if (__builtin_expect(__ballot((pend0 | pend1 | pend2 | pend3) != 0u) != 0ull, 0)) {
    SDC_DBG_BIT(FAST, sh, 8u);
    const unsigned lx_new = hp[H_LAST_XNEW], lx_old = hp[H_LAST_XOLD];
    const int l_nprev = (int)hp[H_LAST_NPREV];
    // pd = (request step mod 2^19) << 13 | result set << 11 | request index + 1.  All due results are requested first (ONE
    // memory round trip whatever the number of windows), then each is replayed and installed; windows with nothing due
    // in either env are skipped as a whole.
    struct Arrival { int4 hd; unsigned k[KPL]; bool due; unsigned age; };
    auto fetch = [&](const unsigned pd) __attribute__((always_inline)) {
      Arrival a;
      const int idx = (int)(pd & 0x7FFu) - 1, set = (int)((pd >> 11) & 3u);
      a.age = ((unsigned)step_no - (pd >> 13)) & 0x7FFFFu;      // steps since the request (mod 2^19)
      a.due = pd != 0u && a.age >= 2u;   // (1: being swept right now; anything else but 2: stale -- a multi-step launch, restored state)
      const SdcRefillRes* rs = S.rs + (set > 2 ? 0 : set) * S.rq_max + (idx < 0 ? 0 : idx);
      a.hd = make_int4(0, 0, -1, -1);
#pragma unroll
      for (int i = 0; i < KPL; i++) a.k[i] = KEY_NONE;
      if (a.due) {
        a.hd = *reinterpret_cast<const int4*>(rs);
#pragma unroll
        for (int i = 0; i < KPL; i++) a.k[i] = rs->keys[KPL * l + i];
      }
      return a;
    };
    auto install = [&](const Arrival& a, Win& q, unsigned& pd, unsigned& wf, unsigned& wl, const int w,
                       const unsigned flip) __attribute__((always_inline)) {
      if (__ballot(a.due) == 0ull) return;
      Win r = win_from(a.k, a.hd.x, a.hd.y);
      const bool good = a.due && a.age == 2u && a.hd.z == step_no - 1 && a.hd.w == envc * 4 + w && r.hi > 0 && ok;
      // the result describes the ring as the request's step left it: replay the previous step's insertion / eviction
      hw_update(r, lx_new ^ flip, lx_old ^ flip, lx_old != KEY_NONE, l_nprev, good, h, l);
      const unsigned rf = win_first(r, h), rl = win_last(r, h);
      if (good && r.hi > 0) {
        q = r;
        wf = rf;
        wl = rl;
        wdc = true;
      }
      if (a.due) pd = 0u;
    };
    const Arrival a0 = fetch(pend0), a1 = fetch(pend1), a2 = fetch(pend2), a3 = fetch(pend3);
    install(a0, q1, pend0, wf0, wl0, 0, 0u);
    install(a1, q3, pend1, wf1, wl1, 1, 0u);
    install(a2, bu, pend2, wf2, wl2, 2, 0u);
    install(a3, bl, pend3, wf3, wl3, 3, KEY_NONE);
  }
  SDC_AT(12, sh, lane0);
  // O(1) updates: running sums, the four windows
  const double vn = key_f64(x_new), vo = has_old ? key_f64(x_old) : 0.0;
  const int n_prev = has_old ? n : n - 1;
  A1 += vn - vo;
  A2 += vn * vn - vo * vo;
  // Most steps neither key lands INSIDE a window (a window lists 64 of the 10 000 keys): it lies below (the window's
  // ranks move by one) or above (nothing moves).  That is decided per window from its cached first / last key --
  // per-lane arithmetic on values that are uniform in the half, no ballots, no LDS permutes; only when some window of
  // either env does have a key inside (or starts / ends the history where the key would land) do the lanes run the
  // general update (hw_update), which also refreshes the cache.
  const int m_hist = has_old ? n_prev - 1 : n_prev;
  auto outside = [&](const Win& q, const unsigned first, const unsigned last, const unsigned flip, int& r0n) __attribute__((always_inline)) {
    const unsigned y = x_old ^ flip, x = x_new ^ flip;
    const bool e_below = has_old && y < first;
    const bool e_ok = !has_old || y > last || e_below;          // (evicted key above the window / below it)
    const int r0e = q.r0 - (e_below ? 1 : 0);
    const bool ends = r0e + q.hi == m_hist;
    const bool i_below = x < first && r0e != 0;
    const bool i_ok = i_below || (x >= last && !ends);          // (appended key below the window / above it)
    r0n = r0e + (i_below ? 1 : 0);
    return e_ok && i_ok;
  };
  int r0n1, r0n3, r0nu, r0nl;
  const bool out1 = outside(q1, wf0, wl0, 0u, r0n1), out3 = outside(q3, wf1, wl1, 0u, r0n3);
  const bool outu = outside(bu, wf2, wl2, 0u, r0nu), outl = outside(bl, wf3, wl3, KEY_NONE, r0nl);
  // (per window: about one wavefront in ten has a key inside SOME window of one of its envs, almost never inside two)
  auto update = [&](Win& q, const bool out, const int r0n, unsigned& wf, unsigned& wl, const unsigned flip) __attribute__((always_inline)) {
    if (__builtin_expect(__ballot(ok && !out) == 0ull, 1)) {
      q.r0 = r0n;
      return false;
    }
    SDC_DBG_BIT(FAST, sh, 2u);
    const bool wd = hw_update(q, x_new ^ flip, x_old ^ flip, has_old, n_prev, ok, h, l);
    wf = win_first(q, h);
    wl = win_last(q, h);
    return wd;
  };
  const bool wd1 = update(q1, out1, r0n1, wf0, wl0, 0u), wd3 = update(q3, out3, r0n3, wf1, wl1, 0u);
  const bool wdu = update(bu, outu, r0nu, wf2, wl2, 0u), wdl = update(bl, outl, r0nl, wf3, wl3, KEY_NONE);
  SDC_AT(13, sh, lane0);
  ok = ok && q1.hi > 0 && q3.hi > 0 && bu.hi > 0 && bl.hi > 0;
  unsigned a1, b1, a3, b3;
  const bool r1 = hw_resolve(q1, k1, n, h, a1, b1), r3 = hw_resolve(q3, k3, n, h, a3, b3);
  ok = ok && r1 && r3;
  const Bounds b = clip_bounds(n, a1, b1, a3, b3);
  const unsigned kb0 = b.kub;               // upper tail: keys >= kub
  const unsigned kb1 = ~(b.klb - 1u);       // lower tail, flipped: ~x >= ~(klb-1)  <=>  x < klb
  SDC_AT(14, sh, lane0);
  // first the value that came and the one that went against last step's bounds kbl, then the keys the bounds have
  // moved across since -- which a bound's window lists, as long as both the old and the new bound lie inside its span
  const unsigned kbl0 = hp[H_KB], kbl1 = hp[H_KB + 1];
  int qc0 = (int)hp[H_QC], qc1 = (int)hp[H_QC + 1];
  double qs1_0 = lrec_f64(hp, H_QS1), qs1_1 = lrec_f64(hp, H_QS1 + 2);
  double qs2_0 = lrec_f64(hp, H_QS2_HI), qs2_1 = lrec_f64(hp, H_QS2_LO);
  {
    const double vo2 = key_f64(x_old);
    if (has_old && x_old >= kbl0) { qc0 -= 1; qs1_0 -= vo2; qs2_0 -= vo2 * vo2; }
    if (has_old && ~x_old >= kbl1) { qc1 -= 1; qs1_1 -= vo2; qs2_1 -= vo2 * vo2; }
    if (x_new >= kbl0) { qc0 += 1; qs1_0 += vn; qs2_0 += vn * vn; }
    if (~x_new >= kbl1) { qc1 += 1; qs1_1 += vn; qs2_1 += vn * vn; }
  }
  const unsigned lo0 = min(kb0, kbl0), hi0 = max(kb0, kbl0), lo1 = min(kb1, kbl1), hi1 = max(kb1, kbl1);
  const bool sp0 = bu.hi > 0 && (bu.r0 == 0 || wf2 < lo0) && (bu.r0 + bu.hi >= n || hi0 <= wl2);   // (hw_spans on the cached keys)
  const bool sp1 = bl.hi > 0 && (bl.r0 == 0 || wf3 < lo1) && (bl.r0 + bl.hi >= n || hi1 <= wl3);
  ok = ok && (kb0 == kbl0 || sp0) && (kb1 == kbl1 || sp1);
  {
    // the keys a bound has crossed: this lane's share over its two keys, then the half's total (rare: skipped as a
    // whole when no lane of the wavefront has one)
    const bool en0 = ok && kb0 != kbl0, en1 = ok && kb1 != kbl1;
    const bool x0 = en0 && win_any_in(bu, lo0, hi0), x1 = en1 && win_any_in(bl, lo1, hi1);
    if (__builtin_expect(__ballot(x0 || x1) != 0ull, 0)) SDC_DBG_BIT(FAST, sh, 4u);
    if (__builtin_expect(__ballot(x0) != 0ull, 0)) {
      unsigned cl;
      double s1l, s2l;
      win_crossed(bu, lo0, hi0, 0u, en0, cl, s1l, s2l);
      const unsigned c = env_sum_u32<LPE>(cl);
      const double s1 = env_sum_f64<LPE>(s1l), s2 = env_sum_f64<LPE>(s2l);
      const double sg = kb0 > kbl0 ? -1.0 : 1.0;   // bound moved out: the keys in between leave the tail
      qc0 += (kb0 > kbl0 ? -1 : 1) * (int)c;
      qs1_0 += sg * s1;
      qs2_0 += sg * s2;
    }
    if (__builtin_expect(__ballot(x1) != 0ull, 0)) {
      unsigned cl;
      double s1l, s2l;
      win_crossed(bl, lo1, hi1, KEY_NONE, en1, cl, s1l, s2l);
      const unsigned c = env_sum_u32<LPE>(cl);
      const double s1 = env_sum_f64<LPE>(s1l), s2 = env_sum_f64<LPE>(s2l);
      const double sg = kb1 > kbl1 ? -1.0 : 1.0;
      qc1 += (kb1 > kbl1 ? -1 : 1) * (int)c;
      qs1_1 += sg * s1;
      qs2_1 += sg * s2;
    }
  }
  SDC_AT(15, sh, lane0);
  // sum (v - bound), sum (v^2 - bound^2) over the keys beyond the bounds
  const double t1 = (qs1_0 - (double)qc0 * b.ub) + (qs1_1 - (double)qc1 * b.lb);
  const double t2 = (qs2_0 - (double)qc0 * (b.ub * b.ub)) + (qs2_1 - (double)qc1 * (b.lb * b.lb));
  double mean, sd, inv_sd;
  clipped_moments(n, b, A1, A2, t1, t2, mean, sd, inv_sd, S.hist_cap, S.rc_hist_cap, S.hist_cap_d);
  // a window that the next step could exhaust is re-centred by the slow path (which then redoes this step)
  {
    int k1n, k3n;
    quartile_ranks(n < S.hist_cap ? n + 1 : n, k1n, k3n);
    auto ahead = [&](const Win& q, const int k_next, const int m_lo, const int m_hi) __attribute__((always_inline)) {
      const int t = k_next - q.r0;
      return ((t > q.hi - m_hi) & (q.r0 + q.hi < n)) | ((t < m_lo) & (q.r0 > 0));
    };
    // (`&`, not `&&`: a short-circuit here is a divergent branch per window -- four of them, each with its exec-mask save
    // and restore, in the middle of the step's straight-line code)
    const bool w0 = (pend0 == 0u) & ahead(q1, k1n, 3, 6), w1 = (pend1 == 0u) & ahead(q3, k3n, 3, 6);
    const bool w2 = (pend2 == 0u) & ahead(bu, n - qc0, 10, 10), w3 = (pend3 == 0u) & ahead(bl, n - qc1, 10, 10);
    if (__builtin_expect(__ballot(ok && (w0 || w1 || w2 || w3)) != 0ull, 0)) {
      SDC_DBG_BIT(FAST, sh, 16u);
      if (!defer) {
        ok = ok && !(w0 || w1 || w2 || w3);      // (multi-step launches re-centre inline: the slow path redoes this step)
      } else {
        // file a request per window: a snapshot of the window as this step leaves it, the rank it should be centred on,
        // and the content of the ring slot the NEXT step overwrites (the sweep must see the ring as it is now)
        const int set = (step_no + 1) % 3;
        const int slot_next = n < S.hist_cap ? n : (d.slot + 1 == S.hist_cap ? 0 : d.slot + 1);
        unsigned patch_x = KEY_NONE;
        if (ok && (w0 || w1 || w2 || w3)) patch_x = S.hist[(size_t)envc * SDC_HIST_STRIDE + slot_next];
        auto request = [&](const Win& q, unsigned& pd, const bool want, const int w, const int kt) __attribute__((always_inline)) {
          int idx = -1;
          if (want && ok && active && l == 0) idx = atomicAdd(&S.rq_count[set], 1);
          idx = __builtin_amdgcn_ds_bpermute((h * LPE) << 2, idx);       // lane 0 of the env's lanes tells the others
          if (want && ok) {
            if (idx < 0 || idx >= S.rq_max) {
              ok = false;                          // no room (or a missing env): re-centre inline on the slow path
            } else {
              SdcRefillReq* rq = S.rq + set * S.rq_max + idx;
              unsigned qk[KPL];
              win_keys(q, qk);
#pragma unroll
              for (int i = 0; i < KPL; i++) rq->keys[KPL * l + i] = qk[i];
              if (l == 0) {
                const int t = kt - q.r0;
                rq->env = envc; rq->win = w;
                rq->dir = (t > q.hi - (w < 2 ? 6 : 10) && q.r0 + q.hi < n) ? REFILL_UP : REFILL_DOWN;
                rq->kt = kt; rq->n = n; rq->r0 = q.r0; rq->hi = q.hi;
                rq->patch_slot = slot_next; rq->patch_x = patch_x; rq->step = step_no;
              }
              pd = (((unsigned)step_no & 0x7FFFFu) << 13) | ((unsigned)set << 11) | (unsigned)(idx + 1);
              filed = true;
            }
          }
        };
        request(q1, pend0, w0, 0, k1n);
        request(q3, pend1, w1, 1, k3n);
        request(bu, pend2, w2, 2, n - qc0);
        request(bl, pend3, w3, 3, n - qc1);
      }
    }
  }
  SDC_AT(16, sh, lane0);
  const double z = (d.e_off - mean) * inv_sd;     // (n >= SMALL_N >= 2 here)
  // rewards (step_rewards), per lane
  double r[3], ret[3];
  {
    double foot, rls;
    sdc_rw::reward_terms(z, d.norm_ci, (double)d.overdue, d.oldest_norm, foot, rls);
    const double ite_kw = SDC_DIV_CONST(d.p_it, 1e3), hour = (double)d.hourq_n * 0.25;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const double v = sdc_rw::agent_reward(FAST ? (int)SDC_REWARD_DEFAULT : S.reward_method[a], a == 0, rls, foot, d.energy, hour, ite_kw,
                                            d.total_kw, d.water);
      r[a] = v;
      ret[a] = lrec_f64(hp, H_RET + 2 * a) + v;
    }
  }
  SDC_AT(17, sh, lane0);
  wave_sync();   // every lane has read the header fields it needs
  const bool commit = ok && active;
  if (commit) {
    if (wd1 || wd3 || wdu || wdl || wdc) {
      uint4* qw = reinterpret_cast<uint4*>(S.qwin) + (size_t)envc * SDC_WIN + KPL * l;
      unsigned o1[KPL], o3[KPL], ou[KPL], ol[KPL];
      win_keys(q1, o1); win_keys(q3, o3); win_keys(bu, ou); win_keys(bl, ol);
#pragma unroll
      for (int i = 0; i < KPL; i++) qw[i] = make_uint4(o1[i], o3[i], ou[i], ol[i]);
    }
    if (l == 0) {
      unsigned* o = sh.hdr[h];
      auto put64 = [&](int idx, double v) { *reinterpret_cast<double*>(o + idx) = v; };
      o[H_KB] = kb0;
      o[H_KB + 1] = kb1;
      o[H_VALID] = 1u;
      put64(H_A1, A1);
      put64(H_A2, A2);
      o[H_QC] = (unsigned)qc0;
      o[H_QC + 1] = (unsigned)qc1;
      put64(H_QS1, qs1_0);
      put64(H_QS1 + 2, qs1_1);
      put64(H_QS2_HI, qs2_0);
      put64(H_QS2_LO, qs2_1);
      o[H_Q1 + T_R0] = (unsigned)q1.r0; o[H_Q1 + T_HI] = (unsigned)q1.hi;
      o[H_Q3 + T_R0] = (unsigned)q3.r0; o[H_Q3 + T_HI] = (unsigned)q3.hi;
      o[H_BU + T_R0] = (unsigned)bu.r0; o[H_BU + T_HI] = (unsigned)bu.hi;
      o[H_BL + T_R0] = (unsigned)bl.r0; o[H_BL + T_HI] = (unsigned)bl.hi;
      o[H_N] = (unsigned)n;
      o[H_PEND] = pend0; o[H_PEND + 1] = pend1; o[H_PEND + 2] = pend2; o[H_PEND + 3] = pend3;
      o[H_WFIRST] = wf0; o[H_WFIRST + 1] = wf1; o[H_WFIRST + 2] = wf2; o[H_WFIRST + 3] = wf3;
      o[H_WLAST] = wl0; o[H_WLAST + 1] = wl1; o[H_WLAST + 2] = wl2; o[H_WLAST + 3] = wl3;
      o[H_LAST_XNEW] = x_new;
      o[H_LAST_XOLD] = x_old;
      o[H_LAST_NPREV] = (unsigned)n_prev;
      put64(H_EOFF, d.e_off);
      put64(H_RET, ret[0]);
      put64(H_RET + 2, ret[1]);
      put64(H_RET + 4, ret[2]);
      rew[envc * 3 + 0] = (float)r[0];
      rew[envc * 3 + 1] = (float)r[1];
      rew[envc * 3 + 2] = (float)r[2];
      float* inf = sh.info[h];
      inf[SDC_INFO_ENERGY_Z] = (float)z;
      inf[SDC_INFO_RESERVED] = wdc ? 2.0f : (SDC_DBG_OK(FAST) && (S.debug_flags & 8) && filed) ? 4.0f : 0.0f;   // no ring read by this wavefront (2: a deferred re-centred window arrived; 4, with the timing diagnostics on: a request was filed)
      inf[SDC_INFO_EP_RETURN_LS] = (float)ret[0];
      inf[SDC_INFO_EP_RETURN_DC] = (float)ret[1];
      inf[SDC_INFO_EP_RETURN_BAT] = (float)ret[2];
    }
  }
  wave_sync();
  if (commit) {
    if constexpr (LPE == HL)
      (reinterpret_cast<unsigned long long*>(S.hdr + (size_t)envc * SDC_HDR_DWORDS))[l] = reinterpret_cast<const unsigned long long*>(sh.hdr[h])[l];
    else
      (reinterpret_cast<uint4*>(S.hdr + (size_t)envc * SDC_HDR_DWORDS))[l] = reinterpret_cast<const uint4*>(sh.hdr[h])[l];
  }
  SDC_AT(18, sh, lane0);
  return __ballot(ok);
}

// One env-step of the env pair (env0, env0 + 1) by its wavefront: loads the state, runs the dynamics of both, the rewards
// and the reward-state upkeep of each, stores the new state and the outputs.
// ACTOR: the three actions of this lane's env come in registers (act_reg: the in-kernel neural policy of
// sdc_rollout_actor_kernel has just computed them) instead of from the caller's array.
template <bool FAST, bool ACTOR = false>
__device__ __forceinline__ void pair_step(const SdcDev& S, PairShared& sh, const int env0, const int lane, const int rel_hint,
                                          const int32_t* __restrict__ actions, float* __restrict__ obs,
                                          float* __restrict__ share_obs, unsigned char* __restrict__ done,
                                          float* __restrict__ info, float* __restrict__ final_obs,
                                          float* __restrict__ rew, int32_t* __restrict__ actions_out, const int step_no,
                                          const bool defer, double* kt, const bool kt_fill, const int act_reg0 = 1,
                                          const int act_reg1 = 1, const int act_reg2 = 2) {
  const int TL = S.table_len;
  double kt0 = 0.0, kt1 = 0.0;
  if (kt_fill) ktab_fetch(lane, kt0, kt1);      // (the constant table: requested first, stored with the record)
  const int h = lane >> 5, l = lane & (HL - 1);
  const int n_here = FAST ? EPW : min(EPW, S.n_envs - env0);   // envs of this pair that exist (1 for the last pair of an odd batch)
  const int envc = env0 + min(h, n_here - 1);             // this lane's env (the missing one mirrors the last)
  const bool active = h < n_here;                         // lanes of a missing env compute, but store nothing
  const int env1c = env0 + n_here - 1;

  // The env's three actions, requested FIRST and by hand: left to the compiler, these loads are scheduled behind the wait
  // for the state record (their pointer and the policy flags arrive with a later batch of kernel arguments) and cost the
  // step a second memory round trip.  Memory returns loads in order, so once the record below has arrived these have too.
  typedef int int3v __attribute__((ext_vector_type(3)));
  int3v act_v = {1, 1, 2};
  if constexpr (ACTOR) {
    act_v.x = act_reg0;
    act_v.y = act_reg1;
    act_v.z = act_reg2;
  } else if (FAST || actions != nullptr) {
    const int32_t* ap = actions + (size_t)envc * 3;
    asm volatile("global_load_dwordx3 %0, %1, off nt" : "=v"(act_v) : "v"(ap) : "memory");
  }
  // When the host knows the episode step every env is at (envs in lock-step: rel_hint >= 0), the step's feature row
  // -- which also holds its trace inputs -- and its queue-history probes are requested together with the state
  // record: ONE memory round trip before the dynamics start instead of two (record, then what it points to).
  const bool pre = FAST ? true : (rel_hint >= 0 && S.feat != nullptr);
  float frow_pre = 0.0f;
  double q_pre = 0.0;
  if (pre) {
    // (non-temporal: a row is read once, 672 steps after it was written -- 11.32 -> 11.20 us per step at 4096 envs; the same hint on
    // the record, the header, the rank windows, the queue probes or the evicted key: nothing, or slower)
    frow_pre = __builtin_nontemporal_load(&S.feat[feat_row_offset(S, envc, rel_hint + 1) + l]);
    if (l >= G_Q97 && l <= G_Q96) {
      const int back = l == G_Q97 ? 97 : 24 * (l - G_Q97);   // 97, 24, 48, 72, 96
      const int t = rel_hint - back;
      if (t >= 0) q_pre = *reinterpret_cast<const double*>(S.qtab + (size_t)envc * S.qstride + t);
    }
  }
  // with a single data-centre configuration (the usual job) its scalars do not wait for the record either
  const bool one_cfg = FAST ? true : S.n_cfg == 1;
  double prm_pre = 0.0;
  if (one_cfg && l < P_COUNT)
    prm_pre = (!FAST || S.n_cfg == 1) ? reinterpret_cast<const double*>(&S.dc[0].p.m_cpu)[l]
                                      : S.prm_env[(size_t)envc * 32 + l];   // (several configs: the env's own copy, see SdcDev)
  const unsigned long long dbg_entry = (SDC_DBG_OK(FAST) && (S.debug_flags & 16)) ? wall_clock64() : 0ull;
  if (SDC_DBG_OK(FAST) && (S.debug_flags & 8) && lane == 0) sh.dbg_bits = 0u;

  // ---- level 0: the two state records (one dwordx2 per lane, 512 contiguous bytes), headers, actions ----------------
  uint2* recp = reinterpret_cast<uint2*>(S.rec + (size_t)envc * SDC_REC_DWORDS) + l;
  const uint2 rr = *recp;
  {
    unsigned r0 = rr.x, r1 = rr.y;
    // (the record is here: so are the actions, requested before it -- memory returns loads in order.  The compiler has
    // waited for r0 / r1 to pass them in; the explicit wait costs nothing then -- no load has been issued since the
    // record's -- and keeps the actions' arrival independent of how the compiler counts the loads it knows about)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(act_v), "+v"(r0), "+v"(r1));
  }
  int a_ls = 1, a_dc = 1, a_bat = 2;       // (rule-based slots never read the caller's array, which may be null)
  if (FAST || S.policy[0] == SDC_POLICY_EXTERNAL) a_ls = act_v.x;
  if (FAST || S.policy[1] == SDC_POLICY_EXTERNAL) a_dc = act_v.y;
  if (FAST || S.policy[2] == SDC_POLICY_EXTERNAL) a_bat = act_v.z;
  unsigned long long dbg_rec = 0ull;
  if (SDC_DBG_OK(FAST) && __builtin_expect((S.debug_flags & 32) != 0, 0)) {
    unsigned tmp = rr.x;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(tmp)::"memory");
    dbg_rec = wall_clock64() + (tmp & 0u);
  }
  reinterpret_cast<uint2*>(sh.rec[h])[l] = rr;
  if (kt_fill) ktab_store(kt, lane, kt0, kt1);
  if constexpr (FAST) {
    // the common case: nothing of what the dynamics read from LDS depends on the record -- the config scalars, the feature
    // row's input slots and the queue probes go to LDS with it, behind ONE synchronisation
    static_assert(SDC_FEAT_W == 10 && SDC_FEAT_C == 22 && SDC_FEAT_T == 24 && SDC_FEAT_WB == 28 && SDC_FEAT_NCNEXT == 30 &&
                  G_W0 == 0 && G_C0 == 3 && G_T0 == 4 && G_WB0 == 5 && G_NCN == 14, "slot table below");
    static_assert(SDC_FEAT_T - 2 * G_T0 == SDC_FEAT_C - 2 * G_C0, "C and T share an offset");
    if (l < P_COUNT) sh.prm[h][l] = prm_pre;
    double* gh = sh.g[h];
    unsigned* g32 = reinterpret_cast<unsigned*>(gh);
    const int off = l < 12 ? SDC_FEAT_W - 2 * G_W0 : l < 26 ? SDC_FEAT_C - 2 * G_C0 : l < 30 ? SDC_FEAT_WB - 2 * G_WB0
                                                                                            : SDC_FEAT_NCNEXT - 2 * G_NCN;
    if ((0xF3C00C00u >> l) & 1u) g32[l - off] = (unsigned)__float_as_int(frow_pre);
    if (l >= G_Q97 && l <= G_Q96) gh[l] = q_pre;
    if (l == SDC_FEAT_T1) gh[G_T1] = (double)frow_pre;
  }
  wave_sync();
  const unsigned* rp = sh.rec[h];
  const int i = lrec_i32(rp, R_CURSOR), rel = lrec_i32(rp, R_TREL);
  const int loc = lrec_i32(rp, R_LOC);
  const SdcDcDev* PD = &S.dc[FAST ? 0 : lrec_i32(rp, R_CFG)];
  const int hourq = lrec_i32(rp, R_HOURQ);
  const int hourq_n = hourq + 1 >= 96 ? 0 : hourq + 1;
  unsigned fault = 0;
  if (i + 9 > TL - 1) fault |= SDC_FAULT_TABLE_RANGE;
  // actions outside {0,1,2}: the reference's dict lookups raise (envs/dc_gym.py:160, bat_env_fwd_view.py:99); here
  // the step flags SDC_FAULT_ACTION and treats the action as "do nothing" / "no change" / "idle"
  if (__builtin_expect((unsigned)a_ls > 2u || (unsigned)a_dc > 2u || (unsigned)a_bat > 2u, 0)) {
    fault |= SDC_FAULT_ACTION;
    if ((unsigned)a_ls > 2u) a_ls = 1;
    if ((unsigned)a_dc > 2u) a_dc = 1;
    if ((unsigned)a_bat > 2u) a_bat = 2;
  }
  // ---- level 1 ----------------------------------------------------------------------------------------------------------
  // config scalars: lane j of the half fetches scalar j of its env's config (one coalesced 8-byte load), LDS hands them round
  if constexpr (!FAST)
    if (l < P_COUNT) sh.prm[h][l] = one_cfg ? prm_pre : reinterpret_cast<const double*>(&PD->p.m_cpu)[l];
  // the ring slot this step's energy will overwrite: its current key is the evicted value the reward state's
  // order-statistic trackers need (0xFFFFFFFF while the ring is still filling)
  const int hl0 = lrec_i32(rp, R_HIST_LEN);
  const int slot0 = hl0 < S.hist_cap ? hl0 : lrec_i32(rp, R_HIST_POS);
  const bool append = FAST ? true : S.reward_method[0] == SDC_REWARD_DEFAULT;   // else the history does not change this step
  // The trace-only observation entries of this step come precomputed (sdc_features.hip), unless the episode has no
  // feature rows (a host write since the reset, an episode too long for that kernel): then the CI / temperature
  // windows are gathered and the features computed here.
  const bool feat_ok = FAST ? true : (S.feat != nullptr && lrec_i32(rp, R_FEAT_OK) == 1);
  const bool fast = FAST ? true : (pre && feat_ok && rel == rel_hint);   // what was requested up front is what this step needs
  float frow = frow_pre;
  if (feat_ok && !fast) frow = S.feat[feat_row_offset(S, envc, rel + 1) + l];
  const bool want_c3 = FAST ? false : S.policy[2] == SDC_POLICY_RBC;
  if constexpr (!FAST) {
    const double ci_min = lrec_f64(rp, R_CI_MIN), ci_den = lrec_f64(rp, R_CI_DEN);
    const double t_min = lrec_f64(rp, R_T_MIN), t_den = lrec_f64(rp, R_T_DEN);
    auto tix = [&](int idx) { return idx < 0 ? 0 : (idx > TL - 1 ? TL - 1 : idx); };
    const double* tW = S.tabW + (size_t)loc * TL;
    const double* tC = S.tabC + (size_t)loc * TL;
    const double* tw = S.t_win + (size_t)envc * S.lw + rel;
    const double* wbw = S.wb_win + (size_t)envc * S.lw + rel;
    const uint2* qt = S.qtab + (size_t)envc * S.qstride;
    auto gather = [&](const int s) -> double {
      const double* src = nullptr;
      if (s <= (feat_ok ? G_W0 : G_W2)) src = tW + tix(i + s);   // (W[i+1], W[i+2], the hour LUT: observations only)
      else if (s == G_C0) src = tC + tix(i);
      else if (s == G_T0) src = tw;
      else if (s == G_WB0) src = wbw;
      else if (s == G_T1) src = tw + 1;
      else if (!feat_ok && s == G_LUT) src = S.hour_lut + 2 * hourq_n;
      else if (!feat_ok && s == G_LUT2) src = S.hour_lut + 2 * hourq_n + 1;
      else if (s >= G_Q97 && s <= G_Q96) {
        const int back = s == G_Q97 ? 97 : 24 * (s - G_Q97);   // 97, 24, 48, 72, 96
        const int t = rel - back;
        if (t >= 0) src = reinterpret_cast<const double*>(qt + t);
      } else if (s == G_C3 && want_c3) src = tC + tix(i + 3);
      else if (!feat_ok && s >= G_NC && s < G_NC + 25) src = tC + tix(i + 1 - 16 + (s - G_NC));
      else if (!feat_ok && s >= G_NT && s < G_NT + 17) src = tw + 1 + (s - G_NT);
      double v = 0.0;
      if (src) v = *src;
      if (!feat_ok) {
        // NC = (C - min) / (max - min) (utils/managers.py:437), NT likewise (:608): ONE division sequence for both windows
        const bool is_nc = s >= G_NC && s < G_NC + 25, is_nt = s >= G_NT && s < G_NT + 17;
        if (is_nc || is_nt) v = (v - (is_nc ? ci_min : t_min)) / (is_nc ? ci_den : t_den);
      }
      return v;
    };
    double* gh = sh.g[h];
    if (__builtin_expect(fast, 1)) {
      // the row's input slots and the probes go to the places the gather would have put them: one predicated 4-byte
      // write (the doubles W, C, T, WB, NC[i'+1] arrive as float pairs of the row), two 8-byte ones -- no per-slot branches
      static_assert(SDC_FEAT_W == 10 && SDC_FEAT_C == 22 && SDC_FEAT_T == 24 && SDC_FEAT_WB == 28 && SDC_FEAT_NCNEXT == 30 &&
                    G_W0 == 0 && G_C0 == 3 && G_T0 == 4 && G_WB0 == 5 && G_NCN == 14, "slot table below");
      unsigned* g32 = reinterpret_cast<unsigned*>(gh);
      const int off = l < 12 ? SDC_FEAT_W - 2 * G_W0 : l < 26 ? SDC_FEAT_C - 2 * G_C0 : l < 30 ? SDC_FEAT_WB - 2 * G_WB0
                                                                                              : SDC_FEAT_NCNEXT - 2 * G_NCN;
      static_assert(SDC_FEAT_T - 2 * G_T0 == SDC_FEAT_C - 2 * G_C0, "C and T share an offset");
      if ((0xF3C00C00u >> l) & 1u) g32[l - off] = (unsigned)__float_as_int(frow);
      if (l >= G_Q97 && l <= G_Q96) gh[l] = q_pre;
      if (l == SDC_FEAT_T1) gh[G_T1] = (double)frow;
      if (l == G_C3 && want_c3) gh[G_C3] = gather(G_C3);
    } else {
      if (l != G_NCN) gh[l] = gather(l);
      if (feat_ok) {
        unsigned* g32 = reinterpret_cast<unsigned*>(gh);
        if (l == SDC_FEAT_NCNEXT || l == SDC_FEAT_NCNEXT + 1) g32[2 * G_NCN + (l - SDC_FEAT_NCNEXT)] = (unsigned)__float_as_int(frow);
      } else {
        gh[HL + l] = gather(HL + l);
      }
    }
    wave_sync();
  }

  unsigned long long dbg_a0 = 0ull;
  if (SDC_DBG_OK(FAST) && __builtin_expect((S.debug_flags & 8) != 0, 0)) dbg_a0 = wall_clock64();
  // the rank windows of both envs, one key each per lane: wanted at the end of the step, so the loads are issued here --
  // after the start-of-launch burst of every env's record / header / gather loads -- and ride along in 8 registers
  // reward-side state (headers: returns, trackers, sums; the rank windows' keys; the evicted ring key): wanted at the end of
  // the step, so these loads are issued here -- after the start-of-launch burst of every env's record / gather loads
  // (measured: whatever joins that burst makes every wavefront's start slower) -- and ride along in registers
  // (loads that depend on a condition go FIRST: the hardware's counter of outstanding loads can only express "all but the
  // last k", so a wait for a value that is followed by a load which may or may not have been issued waits for everything)
  // the queue table from the oldest task's step on (pair_dynamics: where the new oldest task is after tasks were popped)
  const bool q_ahead_ok = a_ls == 2;
  uint2 q_ahead = make_uint2(0u, 0u);
  if (q_ahead_ok) {
    const int t = lrec_i32(rp, R_QHEAD) + l;
    if (t < rel) q_ahead = (S.qtab + (size_t)envc * S.qstride)[t];
  }
  unsigned x_old_l = 0xFFFFFFFFu;
  if (hl0 >= S.hist_cap && append) x_old_l = S.hist[(size_t)envc * SDC_HIST_STRIDE + slot0];   // (one address per half)
  const unsigned hdA = S.hdr[(size_t)env0 * SDC_HDR_DWORDS + lane];
  const unsigned hdB = S.hdr[(size_t)env1c * SDC_HDR_DWORDS + lane];
  const uint4 wka = reinterpret_cast<const uint4*>(S.qwin)[(size_t)envc * SDC_WIN + 2 * l];       // keys 2l of the 4 windows
  const uint4 wkb = reinterpret_cast<const uint4*>(S.qwin)[(size_t)envc * SDC_WIN + 2 * l + 1];   // keys 2l + 1
  const DynOut d = pair_dynamics<FAST>(S, envc, h, l, a_ls, a_dc, a_bat, fault, feat_ok, frow, q_ahead, q_ahead_ok, actions_out, sh, kt);
  wave_sync();

  // ---- episodes without feature rows: the observation features of such an env, all 64 lanes cooperating ---------------
  if (!FAST && __builtin_expect(__ballot(!feat_ok) != 0ull, 0)) {
    for (int e = 0; e < n_here; e++) {
      if (pick_i32(feat_ok ? 1 : 0, e)) continue;
      const double* os = sh.osc[e];
      ObsScalars o;
      o.cos_h = os[0]; o.sin_h = os[1]; o.w_cur = os[2]; o.w_next = os[3]; o.soc = os[4]; o.normq = os[5];
      o.oldest = os[6]; o.avg = os[7];
      for (int b = 0; b < 5; b++) o.hist[b] = os[8 + b];
      o.have_past = os[13] != 0.0;
      build_obs_pool(sh.g[e] + G_NC, sh.g[e] + G_NT, o, sh.pool[e], lane);
    }
    wave_sync();
  }
  __builtin_amdgcn_s_setprio(SDC_BASE_PRIO);
  if (SDC_DBG_OK(FAST) && __builtin_expect((S.debug_flags & 8) != 0, 0) && lane == 0) sh.dbg_t[0] = wall_clock64();

  // ---- rewards + reward-state upkeep: both envs at once on the O(1) path; an env that needs its ring (or anything
  // unusual) is redone whole-wavefront from its untouched state ------------------------------------------------------------
  sh.hdr[0][lane] = hdA;
  sh.hdr[1][lane] = hdB;
  wave_sync();
  const uint4 wk2[2] = {wka, wkb};
  const unsigned long long fast_m = pair_reward_fast<FAST>(S, envc, active, h, l, wk2, d, x_old_l, rew, sh, step_no, defer);
  // (the loop sits behind its own unlikely test: otherwise the ~50 constants of env_reward, hoisted into the loop's
  // preheader, are materialised on every step)
  const bool all_fast = ((fast_m & 1ull) != 0ull) && (n_here < 2 || ((fast_m >> HL) & 1ull) != 0ull);
  if (__builtin_expect(!all_fast, 0))
#pragma unroll 1
  for (int e = 0; e < n_here; e++) {
    if (__builtin_expect((fast_m >> (e * HL)) & 1ull, 1)) continue;
    const unsigned x_old = (unsigned)__builtin_amdgcn_readlane((int)x_old_l, e * HL);
    const uint4 qw_e = reinterpret_cast<const uint4*>(S.qwin)[(size_t)(env0 + e) * SDC_WIN + lane];
    env_reward(S, env0 + e, lane, e == 0 ? hdA : hdB, qw_e, pick_i32(d.hl, e), pick_i32(d.slot, e),
               (unsigned)pick_i32((int)d.x_new, e), x_old, pick_f64(d.e_off, e), pick_f64(d.energy, e),
               pick_f64(d.norm_ci, e), pick_f64(d.oldest_norm, e), pick_i32(d.overdue, e), pick_i32(d.hourq_n, e),
               pick_f64(d.p_it, e), pick_f64(d.total_kw, e), pick_f64(d.water, e), rew, sh.info[e], sh.tl);
  }
  if (SDC_DBG_OK(FAST) && __builtin_expect((S.debug_flags & 8) != 0, 0)) {
    wave_sync();
    if (lane == 0) {
      const unsigned long long dbg_a3 = wall_clock64();
      for (int e = 0; e < n_here; e++) {
        float* inf = sh.info[e];
        inf[40] = (S.debug_flags & 32) ? (float)(dbg_rec - dbg_entry) : (S.debug_flags & 16) ? (float)(dbg_a0 & 0xFFFFFull) : 0.0f;
        inf[41] = (S.debug_flags & 16) ? (float)(dbg_a0 - dbg_entry) : (float)(sh.dbg_t[0] - dbg_a0);
        inf[42] = (float)(dbg_a3 - sh.dbg_t[0]);
        if (SDC_STAMP_A != 0 && SDC_STAMP_B != 0) inf[42] = (float)(sh.dbg_s[1] - sh.dbg_s[0]);
        inf[SDC_INFO_RESERVED] += (float)(8u * sh.dbg_bits);
        if (S.debug_flags & 256)     // where the wavefront ran: XCC id << 16 | HW_ID (wave, SIMD, CU, SH, SE)
          inf[40] = (float)(((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu) << 16) | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFFu));
        inf[43] = (S.debug_flags & 16) ? (float)(dbg_a3 & 0xFFFFFull) : (float)(dbg_a3 - dbg_a0);
      }
    }
  }
  wave_sync();

  // ---- coalesced stores: records, obs [3][26] (78 floats per env), share_obs [29], info [44]: the pair's rows are adjacent --
  // (outputs non-temporal: nothing in this launch reads them again, and whole lines that have already left the L2 shorten
  // the write-back at the end of the launch; partial-line stores -- rew, done, the ring slot -- must NOT be: they turn into
  // read-modify-writes at the memory and add 10 us.  The state records / headers: plain stores, measured the same.)
  if (active) *reinterpret_cast<unsigned long long*>(recp) = reinterpret_cast<const unsigned long long*>(sh.rec[h])[l];
  const int rel_now = lrec_i32(sh.rec[h], R_TREL);     // (patched: rel + 1)
  const bool terminal = rel_now >= S.episode_steps;
  const unsigned long long term_m = __ballot(terminal && active);
#pragma unroll
  for (int k = 0; k < (EPW * SDC_OBS_OUT + SDC_WAVE - 1) / SDC_WAVE; k++) {
    const int idx = k * SDC_WAVE + lane;
    if (idx < n_here * SDC_OBS_OUT) {
      const int e = idx >= SDC_OBS_OUT ? 1 : 0, j = idx - e * SDC_OBS_OUT;
      const float v = obs_padded_lut(sh.pool[e], reinterpret_cast<const unsigned char*>(kt + SDC_K_OBS_SRC), j);
      SDC_OUT_STORE(v, &obs[(size_t)env0 * SDC_OBS_OUT + idx]);
      if (final_obs && ((term_m >> (e * HL)) & 1ull)) final_obs[(size_t)env0 * SDC_OBS_OUT + idx] = v;
    }
  }
  if ((FAST || share_obs) && lane < n_here * SDC_SHARE_OBS_DIM) {
    const int e = lane >= SDC_SHARE_OBS_DIM ? 1 : 0, j = lane - e * SDC_SHARE_OBS_DIM;
    SDC_OUT_STORE(share_obs_at(sh.pool[e], j), &share_obs[(size_t)env0 * SDC_SHARE_OBS_DIM + lane]);
  }
  if (FAST || info) {
#pragma unroll
    for (int k = 0; k < (EPW * SDC_INFO_DIM + SDC_WAVE - 1) / SDC_WAVE; k++) {
      const int idx = k * SDC_WAVE + lane;
      if (idx < n_here * SDC_INFO_DIM) {
        const int e = idx >= SDC_INFO_DIM ? 1 : 0, j = idx - e * SDC_INFO_DIM;
        SDC_OUT_STORE(sh.info[e][j], &info[(size_t)env0 * SDC_INFO_DIM + idx]);
      }
    }
  }
  if (l == 0 && active) done[envc] = (unsigned char)(terminal ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// One env-step of FOUR envs (env0 .. env0 + 3) by their wavefront, a DPP row of 16 lanes each: the common case only (see
// pair_dynamics FAST; the host also checks that the batch is a multiple of four envs).  Same memory plan as pair_step,
// the per-lane shares twice as wide: the state record and the header as one dwordx4 per lane, the feature row as one
// dwordx2, the rank windows as four dwordx4 (keys 4l .. 4l + 3 of the four windows), the queue table ahead of the oldest
// task as two dwordx2.
// ACTOR: the three actions of this lane's env come in registers.
template <bool ACTOR = false>
__device__ __forceinline__ void quad_step(const SdcDev& S, QuadShared& sh, const int env0, const int lane, const int rel_hint,
                                          const int32_t* __restrict__ actions, float* __restrict__ obs,
                                          float* __restrict__ share_obs, unsigned char* __restrict__ done,
                                          float* __restrict__ info, float* __restrict__ final_obs, float* __restrict__ rew,
                                          const int step_no, const bool defer, double* kt, const bool kt_fill,
                                          const int act_reg0 = 1, const int act_reg1 = 1, const int act_reg2 = 2) {
  const int TL = S.table_len;
  double kt0 = 0.0, kt1 = 0.0;
  if (kt_fill) ktab_fetch(lane, kt0, kt1);
  const int h = lane >> 4, l = lane & (QL - 1);     // h: the env's row
  const int envc = env0 + h;
  typedef int int3v __attribute__((ext_vector_type(3)));
  int3v act_v = {1, 1, 2};
  if constexpr (ACTOR) {
    act_v.x = act_reg0;
    act_v.y = act_reg1;
    act_v.z = act_reg2;
  } else {
    const int32_t* ap = actions + (size_t)envc * 3;
    asm volatile("global_load_dwordx3 %0, %1, off nt" : "=v"(act_v) : "v"(ap) : "memory");
  }
  // the step's feature row (entries 2l, 2l + 1) and queue-history probes, with the record: one round trip
  const float* frp = S.feat + feat_row_offset(S, envc, rel_hint + 1) + 2 * l;   // (non-temporal: see pair_step)
  const float2 frow2 = make_float2(__builtin_nontemporal_load(frp), __builtin_nontemporal_load(frp + 1));
  double q_pre = 0.0;
  if (l >= G_Q97 && l <= G_Q96) {
    const int back = l == G_Q97 ? 97 : 24 * (l - G_Q97);   // 97, 24, 48, 72, 96
    const int t = rel_hint - back;
    if (t >= 0) q_pre = *reinterpret_cast<const double*>(S.qtab + (size_t)envc * S.qstride + t);
  }
  double prm_pre = 0.0;
  if (lane < P_COUNT) prm_pre = reinterpret_cast<const double*>(&S.dc[0].p.m_cpu)[lane];
  // the per-rack parameters of racks 0..31: lanes 0..31 fetch {supply, idle}, lanes 32..63 {full, n} (see pair_step)
  const sdc_dc_params& P0 = S.dc[0].p;
  const int rl = lane & 31, rh = lane >> 5;
  const double rk_pre0 = rh == 0 ? P0.rack_supply[rl] : P0.rack_full[rl];
  const double rk_pre1 = rh == 0 ? P0.rack_idle[rl] : P0.rack_n[rl];
  uint4* recp = reinterpret_cast<uint4*>(S.rec + (size_t)envc * SDC_REC_DWORDS) + l;
  const uint4 rr = *recp;
  {
    unsigned r0 = rr.x, r1 = rr.y, r2 = rr.z, r3 = rr.w;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(act_v), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
  }
  int a_ls = act_v.x, a_dc = act_v.y, a_bat = act_v.z;
  reinterpret_cast<uint4*>(sh.rec[h])[l] = rr;
  if (kt_fill) ktab_store(kt, lane, kt0, kt1);
  if (lane < P_COUNT) sh.prm[lane] = prm_pre;
  sh.rk[2 * rh][rl] = rk_pre0;
  sh.rk[2 * rh + 1][rl] = rk_pre1;
  {
    // the row's input slots (doubles W, C, T, WB, NC[i'+1] as float pairs of the row; T[i+1] as a float) and the probes
    static_assert(SDC_FEAT_W == 10 && SDC_FEAT_C == 22 && SDC_FEAT_T == 24 && SDC_FEAT_WB == 28 && SDC_FEAT_NCNEXT == 30 &&
                  SDC_FEAT_T1 == 12 && G_W0 == 0 && G_C0 == 3 && G_T0 == 4 && G_WB0 == 5 && G_NCN == 14, "slot table below");
    double* gh = sh.g[h];
    const int slot = l == 5 ? G_W0 : (l == 11 ? G_C0 : (l == 12 ? G_T0 : (l == 14 ? G_WB0 : G_NCN)));
    if (l == 5 || l == 11 || l == 12 || l == 14 || l == 15) reinterpret_cast<float2*>(gh)[slot] = frow2;
    if (l >= G_Q97 && l <= G_Q96) gh[l] = q_pre;
    if (l == 6) gh[G_T1] = (double)frow2.x;
    // ... and the trace-only entries of the NEXT observation (pair_dynamics: TRACE_ONLY) straight into the pool
    constexpr unsigned TRACE_ONLY = 0x7u | (0x7Fu << SDC_P_CI7) | (1u << SDC_P_W) | (1u << SDC_P_NT) | (1u << SDC_P_TSLOPE) |
                                    (0x1Fu << SDC_P_T5) | (1u << SDC_P_WNEXT) | (1u << SDC_P_NTNEXT);
    float* pool = sh.pool[h];
    if (2 * l < SDC_POOL_DIM && ((TRACE_ONLY >> (2 * l)) & 1u)) pool[2 * l] = frow2.x;
    if (2 * l + 1 < SDC_POOL_DIM && ((TRACE_ONLY >> (2 * l + 1)) & 1u)) pool[2 * l + 1] = frow2.y;
  }
  wave_sync();
  const unsigned* rp = sh.rec[h];
  const int i = lrec_i32(rp, R_CURSOR), rel = lrec_i32(rp, R_TREL);
  unsigned fault = 0;
  if (i + 9 > TL - 1) fault |= SDC_FAULT_TABLE_RANGE;
  if (__builtin_expect((unsigned)a_ls > 2u || (unsigned)a_dc > 2u || (unsigned)a_bat > 2u, 0)) {
    fault |= SDC_FAULT_ACTION;
    if ((unsigned)a_ls > 2u) a_ls = 1;
    if ((unsigned)a_dc > 2u) a_dc = 1;
    if ((unsigned)a_bat > 2u) a_bat = 2;
  }
  const int hl0 = lrec_i32(rp, R_HIST_LEN);
  const int slot0 = hl0 < S.hist_cap ? hl0 : lrec_i32(rp, R_HIST_POS);

  // reward-side state, consumed at the end of the step: requested here, behind the staging
  const bool q_ahead_ok = a_ls == 2;
  uint2 q_ahead = make_uint2(0u, 0u), q_ahead_b = make_uint2(0u, 0u);
  if (q_ahead_ok) {
    const int t = lrec_i32(rp, R_QHEAD) + 2 * l;
    const uint2* qt = S.qtab + (size_t)envc * S.qstride;
    if (t < rel) q_ahead = qt[t];
    if (t + 1 < rel) q_ahead_b = qt[t + 1];
  }
  // (the evicted ring key must be read BEFORE the dynamics: they store this step's key into that slot)
  unsigned x_old_l = 0xFFFFFFFFu;
  if (hl0 >= S.hist_cap) x_old_l = S.hist[(size_t)envc * SDC_HIST_STRIDE + slot0];   // (one address per row)
  const DynOut d = pair_dynamics<true, MapQuad>(S, envc, h, l, a_ls, a_dc, a_bat, fault, true, frow2.x, q_ahead, q_ahead_ok, nullptr,
                                                sh, kt, frow2.y, q_ahead_b);
  // (the reward-side loads AFTER the dynamics: 20 registers fewer across them; their latency is the other resident
  // wavefronts' time)
  const uint4 hd4 = reinterpret_cast<const uint4*>(S.hdr + (size_t)envc * SDC_HDR_DWORDS)[l];
  uint4 wk[4];
#pragma unroll
  for (int j = 0; j < 4; j++) wk[j] = reinterpret_cast<const uint4*>(S.qwin)[(size_t)envc * SDC_WIN + 4 * l + j];
  wave_sync();
  __builtin_amdgcn_s_setprio(SDC_BASE_PRIO);
  reinterpret_cast<uint4*>(sh.hdr[h])[l] = hd4;
  wave_sync();
  const unsigned long long fast_m = pair_reward_fast<true, MapQuad>(S, envc, true, h, l, wk, d, x_old_l, rew, sh, step_no, defer);
  // an env that needs its ring (or anything unusual) is redone whole-wavefront from its untouched state
  const bool all_fast = ((fast_m & 1ull) != 0ull) && (((fast_m >> QL) & 1ull) != 0ull) && (((fast_m >> (2 * QL)) & 1ull) != 0ull) &&
                        (((fast_m >> (3 * QL)) & 1ull) != 0ull);
  if (__builtin_expect(!all_fast, 0))
#pragma unroll 1
    for (int e = 0; e < QE; e++) {
      if (__builtin_expect((fast_m >> (e * QL)) & 1ull, 1)) continue;
      const unsigned x_old = (unsigned)__builtin_amdgcn_readlane((int)x_old_l, e * QL);
      const unsigned hd_e = sh.hdr[e][lane];       // (untouched: the O(1) path commits nothing for an env it gives up on)
      const uint4 qw_e = reinterpret_cast<const uint4*>(S.qwin)[(size_t)(env0 + e) * SDC_WIN + lane];
      auto pi = [&](const int v) { return __builtin_amdgcn_readlane(v, e * QL); };
      auto pf = [&](const double v) { return readlane_f64(v, e * QL); };
      env_reward(S, env0 + e, lane, hd_e, qw_e, pi(d.hl), pi(d.slot), (unsigned)pi((int)d.x_new), x_old, pf(d.e_off), pf(d.energy),
                 pf(d.norm_ci), pf(d.oldest_norm), pi(d.overdue), pi(d.hourq_n), pf(d.p_it), pf(d.total_kw), pf(d.water), rew,
                 sh.info[e], sh.tl);
    }
  wave_sync();

  // ---- coalesced stores: the four envs' records, obs rows (4 x 78 floats), share_obs (4 x 29), info (4 x 44) are adjacent ----
  *recp = reinterpret_cast<const uint4*>(sh.rec[h])[l];
  const int rel_now = lrec_i32(sh.rec[h], R_TREL);     // (patched: rel + 1)
  const bool terminal = rel_now >= S.episode_steps;
  const unsigned long long term_m = __ballot(terminal);
#pragma unroll
  for (int k = 0; k < (QE * SDC_OBS_OUT + SDC_WAVE - 1) / SDC_WAVE; k++) {
    const int idx = k * SDC_WAVE + lane;
    if (idx < QE * SDC_OBS_OUT) {
      const int e = (idx >= SDC_OBS_OUT ? 1 : 0) + (idx >= 2 * SDC_OBS_OUT ? 1 : 0) + (idx >= 3 * SDC_OBS_OUT ? 1 : 0);
      const int j = idx - e * SDC_OBS_OUT;
      const float v = obs_padded_lut(sh.pool[e], reinterpret_cast<const unsigned char*>(kt + SDC_K_OBS_SRC), j);
      SDC_OUT_STORE(v, &obs[(size_t)env0 * SDC_OBS_OUT + idx]);
      if (final_obs && ((term_m >> (e * QL)) & 1ull)) final_obs[(size_t)env0 * SDC_OBS_OUT + idx] = v;
    }
  }
#pragma unroll
  for (int k = 0; k < (QE * SDC_SHARE_OBS_DIM + SDC_WAVE - 1) / SDC_WAVE; k++) {
    const int idx = k * SDC_WAVE + lane;
    if (idx < QE * SDC_SHARE_OBS_DIM) {
      const int e = (idx >= SDC_SHARE_OBS_DIM ? 1 : 0) + (idx >= 2 * SDC_SHARE_OBS_DIM ? 1 : 0) + (idx >= 3 * SDC_SHARE_OBS_DIM ? 1 : 0);
      SDC_OUT_STORE(share_obs_at(sh.pool[e], idx - e * SDC_SHARE_OBS_DIM), &share_obs[(size_t)env0 * SDC_SHARE_OBS_DIM + idx]);
    }
  }
#pragma unroll
  for (int k = 0; k < (QE * SDC_INFO_DIM + SDC_WAVE - 1) / SDC_WAVE; k++) {
    const int idx = k * SDC_WAVE + lane;
    if (idx < QE * SDC_INFO_DIM) {
      const int e = (idx >= SDC_INFO_DIM ? 1 : 0) + (idx >= 2 * SDC_INFO_DIM ? 1 : 0) + (idx >= 3 * SDC_INFO_DIM ? 1 : 0);
      SDC_OUT_STORE(sh.info[e][idx - e * SDC_INFO_DIM], &info[(size_t)env0 * SDC_INFO_DIM + idx]);
    }
  }
  if (l == 0) done[envc] = (unsigned char)(terminal ? 1 : 0);
}

}  // namespace
