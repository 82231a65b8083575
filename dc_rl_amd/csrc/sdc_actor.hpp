// sdc_actor.hpp -- a neural policy INSIDE the rollout kernel: the reference's actor network evaluated by the wavefront
// that owns the env pair, between two of its env-steps, so that a closed loop (observation -> actor -> action -> step)
// never leaves the kernel.
//
// The network is HARL's StochasticPolicy for a discrete action space (harl/models/policy_models/stochastic_policy.py:11-60
// over harl/models/base/mlp.py:8-72 and harl/models/base/act.py:45-84 / distributions.py:38-53):
//     x (26, the agent's zero-padded observation)
//       -> LayerNorm(26)                                   (use_feature_normalization, happo.yaml:62)
//       -> Linear(26, 64) -> act -> LayerNorm(64)          (hidden_sizes [64, 64], activation tanh, happo.yaml:58-60)
//       -> Linear(64, 64) -> act -> LayerNorm(64)
//       -> Linear(64, 3)  -> Categorical(logits): mode() = argmax (deterministic) or sample()
// one network per agent (agent_ls, agent_dc, agent_bat), fp32 like the reference.
//
// Mapping: the two hidden layers are contractions and run on the MATRIX CORES, wavefront-locally -- no barrier, no
// partner wavefront.  The instruction is v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 outer products, K = 1,
//     D_b[i][j] += A_b[i] * B_b[j]        b = 0..15, lane 4 b + j holds column j of block b, register i = row i
// read as ONE 4 x 64 product: rows i = envs (2 of the 4 are real: the wavefront's pair), columns 4 b + j = hidden unit
// = LANE.  B: lane l supplies W[k][unit l] -- a row of the k-major weight matrix, straight from LDS.  A: with the
// broadcast modifiers (cbsz = 4: all 16 blocks take block `abid`'s A), lanes 4 m + e of ONE register supply x[env e][k]
// for the instruction that names abid = m; a register P_v with P_v[4 m + e] = x_e[4 m + v] therefore serves the 16
// instructions k = 4 m + v, m = 0..15 -- and because the layer's output sits as (lane = unit, register = env), P_v is
// a quad permutation (DPP) of the activations: the layer-to-layer hand-over never touches LDS.
// Cost per wavefront and env-step: 3 x (28 + 64) MFMAs of 8 cycles, 69 ds_read_b128 of weights, ~100 VALU.
// (The first form of this, lane-per-unit packed FMAs with the inputs broadcast through LDS, spent 8.5 us per step on the
// three networks, LDS-bandwidth bound; the second, 16x16x4 tiles pooling a workgroup's 16 envs between two barriers, 4.3 us
// plus what a barrier per step costs a workgroup whose wavefronts' step times differ)
#pragma once
#include "sdc_device.hpp"

#define SDC_ACT_IN SDC_OBS_PAD     // 26
#define SDC_ACT_M1 7               // K-quads of layer 1 (26 inputs padded to 28)
#define SDC_ACT_H 64
#define SDC_ACT_M2 (SDC_ACT_H / 4)
#define SDC_ACT_OUT 3

// one agent's actor as the kernel reads it (device memory, then LDS); filled by sdc_set_actor (sdc_capi.hip)
struct SdcActorDev {
  float ln0_g[32], ln0_b[32];                 // feature LayerNorm over the 26 inputs (entries 26.. unused)
  float w1[SDC_ACT_M1][SDC_ACT_H][4];         // w1[m][j][v] = W1[j][4 m + v]   (W as torch stores it: [out][in]; 0 beyond k = 25)
  float w2[SDC_ACT_M2][SDC_ACT_H][4];         // w2[m][j][v] = W2[j][4 m + v]: one ds_read_b128 per lane = four B operands
  float w3[4][SDC_ACT_H];                     // w3[c][j] = W3[c][j], c < 3
  float b1[SDC_ACT_H], ln1_g[SDC_ACT_H], ln1_b[SDC_ACT_H];
  float b2[SDC_ACT_H], ln2_g[SDC_ACT_H], ln2_b[SDC_ACT_H];
  float b3[4];
  int flags;                                  // bit 0: feature LayerNorm on; bits 1-2: activation (0 tanh, 1 relu)
  int pad[3];
};
static_assert(sizeof(SdcActorDev) % 16 == 0, "copied to LDS as uint4");

namespace sdc_act {

typedef float f4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_f32(const float v) {   // (every row written, no source lane -> 0)
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// REDUCTIONS over the wavefront.  Within the 16 lanes of a DPP row: four butterfly stages, N chains interleaved (each
// stage waits out the previous one).
template <int N>
__device__ __forceinline__ void row_sums(float (&v)[N]) {
#define SDC_ACT_STAGE(C) \
  _Pragma("unroll") for (int i = 0; i < N; i++) v[i] += dpp_f32<C>(v[i]);
  SDC_ACT_STAGE(SDC_DPP_XOR1)
  SDC_ACT_STAGE(SDC_DPP_XOR2)
  SDC_ACT_STAGE(SDC_DPP_HALF_MIRROR)
  SDC_ACT_STAGE(SDC_DPP_MIRROR)
#undef SDC_ACT_STAGE
}
// ACROSS the four rows, two (then four) values per register: v_permlane16_swap exchanges the odd rows of its first
// operand with the even rows of its second, so swapping two DIFFERENT row-sum registers a, b and adding the results
// gives rows [a0+a1, b0+b1, a2+a3, b2+b3] -- both pair sums in one register, two instructions, no copies; one
// v_permlane32_swap + add of two such registers gives rows [A, B, C, D], the four totals.  (Reducing each value on its
// own -- swap with itself -- took a copy, a swap and an add per value and stage: the networks are VALU-issue bound, and
// a third of their instructions were these.)
__device__ __forceinline__ float swap16_sum(const float a, const float b) {
  const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}
__device__ __forceinline__ float swap32_sum(const float r, const float q) {
  const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(r), __float_as_uint(q), false, false);
  return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}
// rows [T0, T1, T2, T3] -> even = [T0, T0, T2, T2], odd = [T1, T1, T3, T3]: each half of the wavefront its own pair
__device__ __forceinline__ void spread16(const float t, float& even, float& odd) {
  const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  even = __uint_as_float(s[0]);
  odd = __uint_as_float(s[1]);
}
__device__ __forceinline__ float row_value(const float t, const int row) {      // (wave-uniform: an SGPR)
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), 16 * row));
}
template <int KIND>
__device__ __forceinline__ float activate(const float x) {
  if constexpr (KIND == 1) return x > 0.0f ? x : 0.0f;
  // tanh x = 1 - 2 / (e^(2x) + 1): hardware exp2 / rcp (~1 ulp each), exact limits at +-inf
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// The A operands of a layer from its input as (lane = entry k, x0 = env 0's value, x1 = env 1's):
// P[v][lane 4 m + e] = x_e[4 m + v] -- a quad broadcast of x0 and of x1, merged on the lane's parity (lanes 4 m + 2, + 3
// repeat envs 0, 1: rows 2, 3 of the product, never read).
template <int V>
__device__ __forceinline__ float quad_bcast(const float x) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), V * 0x55, 0xF, 0xF, true));
}
__device__ __forceinline__ void a_operands(float (&P)[4], const float x0, const float x1, const bool odd) {
  // (both broadcasts FIRST, by every lane, then the select: written as `odd ? bcast(x1) : bcast(x0)` each DPP ran under half
  // the exec mask -- its source lanes switched off)
  const float e0 = quad_bcast<0>(x0), e1 = quad_bcast<1>(x0), e2 = quad_bcast<2>(x0), e3 = quad_bcast<3>(x0);
  const float o0 = quad_bcast<0>(x1), o1 = quad_bcast<1>(x1), o2 = quad_bcast<2>(x1), o3 = quad_bcast<3>(x1);
  P[0] = odd ? o0 : e0;
  P[1] = odd ? o1 : e1;
  P[2] = odd ? o2 : e2;
  P[3] = odd ? o3 : e3;
}
// bias, activation, LayerNorm(64) of the three agents' layer outputs at once (torch: biased variance, eps 1e-5).  One pass over
// SHIFTED values d = v - v[unit 0]: E[d^2] - E[d]^2 is the same variance, and with a ReLU's unbounded outputs (activations of
// ~30 with a small spread) the unshifted form's cancellation error passed eps; shifted, it is the spread that is squared.
// In: acc[a][e] = env e's pre-activation; out: h[a][e], lane = unit.
template <int KIND>
__device__ __forceinline__ void epilogue(const f4 (&acc)[3], const float (&bias)[3], const float (&g)[3], const float (&b)[3],
                                         float (&h)[3][2]) {
  float r[12];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const float v = activate<KIND>(acc[a][e] + bias[a]);
      const float d = v - __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
      h[a][e] = d;
      r[4 * a + e] = d;
      r[4 * a + 2 + e] = d * d;
    }
  row_sums<12>(r);
#pragma unroll
  for (int a = 0; a < 3; a++) {
    // rows: [sum v (env 0), sum v (env 1), sum v^2 (env 0), sum v^2 (env 1)]
    const float t = swap32_sum(swap16_sum(r[4 * a], r[4 * a + 1]), swap16_sum(r[4 * a + 2], r[4 * a + 3]));
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const float mean = row_value(t, e) * (1.0f / SDC_ACT_H);
      const float var = fmaxf(row_value(t, 2 + e) * (1.0f / SDC_ACT_H) - mean * mean, 0.0f);
      h[a][e] = (h[a][e] - mean) * __builtin_amdgcn_rsqf(var + 1e-5f) * g[a] + b[a];
    }
  }
}

// A layer's K-quads in chunks of SDC_ACT_CH, double-buffered: the weights of chunk c + 1 (or, after the last, the first
// chunk of the NEXT layer: they depend on nothing) are fetched -- ds_read_b128: four B operands per lane -- while chunk c
// multiplies.  The scheduling barriers keep that order (left alone, the scheduler sinks every fetch next to its use and
// each group of MFMAs waits out an LDS round trip).
#define SDC_ACT_CH 2
struct WBuf { f4 w[2][3][SDC_ACT_CH]; };
template <int NM>
__device__ __forceinline__ void fetch3(f4 (&B)[3][SDC_ACT_CH], const SdcActorDev* A, const bool layer2, const int m0, const int lane) {
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int i = 0; i < NM; i++)
      B[a][i] = layer2 ? *reinterpret_cast<const f4*>(A[a].w2[m0 + i][lane]) : *reinterpret_cast<const f4*>(A[a].w1[m0 + i][lane]);
}
template <int M, int MEND>       // the MFMAs of K-quads M .. MEND-1, the three agents interleaved (an accumulator is touched every
struct KQuads {                  // third instruction: no MFMA waits for the one before it)
  static __device__ __forceinline__ void run(f4 (&acc)[3], const float (&P)[3][4], const f4 (&B)[3][SDC_ACT_CH]) {
#define SDC_ACT_MFMA(V) \
  _Pragma("unroll") for (int a = 0; a < 3; a++) \
    acc[a] = __builtin_amdgcn_mfma_f32_4x4x1f32(P[a][V], B[a][M % SDC_ACT_CH][V], acc[a], 4, M, 0);
    SDC_ACT_MFMA(0) SDC_ACT_MFMA(1) SDC_ACT_MFMA(2) SDC_ACT_MFMA(3)
#undef SDC_ACT_MFMA
    if constexpr (M + 1 < MEND) KQuads<M + 1, MEND>::run(acc, P, B);
  }
};
// chunk C of a layer of MQ K-quads (buffer C % 2 holds it; PAR: the buffer parity the layer started on)
template <int C, int MQ, bool LAYER2, int PAR>
struct Chunks {
  static constexpr int NCH = (MQ + SDC_ACT_CH - 1) / SDC_ACT_CH;
  static __device__ __forceinline__ void run(f4 (&acc)[3], const float (&P)[3][4], WBuf& W, const SdcActorDev* A, const int lane) {
    constexpr int M0 = C * SDC_ACT_CH, M1 = M0 + SDC_ACT_CH < MQ ? M0 + SDC_ACT_CH : MQ;
    if constexpr (C + 1 < NCH) {
      constexpr int N1 = (M1 + SDC_ACT_CH < MQ ? M1 + SDC_ACT_CH : MQ) - M1;
      fetch3<N1>(W.w[(PAR + C + 1) & 1], A, LAYER2, M1, lane);
    } else if constexpr (!LAYER2) {
      fetch3<SDC_ACT_CH>(W.w[(PAR + C + 1) & 1], A, true, 0, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    KQuads<M0, M1>::run(acc, P, W.w[(PAR + C) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (C + 1 < NCH) Chunks<C + 1, MQ, LAYER2, PAR>::run(acc, P, W, A, lane);
  }
};
// (KQuads indexes a chunk's buffer by M % SDC_ACT_CH: chunks start at multiples of SDC_ACT_CH)

// THE THREE AGENTS' NETWORKS FOR THE WAVEFRONT'S TWO ENVS, interleaved (three independent chains fill each other's
// latencies).  x[a]: this lane's input of agent a (lane (h, k): entry k < 26 of env h's padded observation of that agent,
// 0 beyond).  lg[a][0..2] = agent a's logits for THE LANE'S OWN env.
template <int KIND>
__device__ __forceinline__ void forward3_kind(const SdcActorDev* A, const float (&x)[3], const int lane, float (&lg)[3][3]) {
  const int k = lane & 31;
  const bool odd = (lane & 1) != 0;
  WBuf W;
  fetch3<SDC_ACT_CH>(W.w[0], A, false, 0, lane);
  // feature LayerNorm (observation entries are of order one -- normalised by construction -- so one pass is safe here too)
  float xn[3] = {x[0], x[1], x[2]};
  {
    float r[6] = {x[0], x[0] * x[0], x[1], x[1] * x[1], x[2], x[2] * x[2]};
    row_sums<6>(r);
#pragma unroll
    for (int a = 0; a < 3; a++)
      if (A[a].flags & 1) {
        // rows [sum x (env 0), sum x^2 (env 0), sum x (env 1), sum x^2 (env 1)]: an env is a half = two rows
        float sx, sxx;
        spread16(swap16_sum(r[2 * a], r[2 * a + 1]), sx, sxx);
        const float mean = sx * (1.0f / SDC_ACT_IN);
        const float var = fmaxf(sxx * (1.0f / SDC_ACT_IN) - mean * mean, 0.0f);
        xn[a] = (x[a] - mean) * __builtin_amdgcn_rsqf(var + 1e-5f) * A[a].ln0_g[k] + A[a].ln0_b[k];
      }
  }
  float P[3][4];
  f4 acc[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    // env 0's inputs sit in lanes 0..25, env 1's in lanes 32..57: both to lanes 0..25 (the lane = the entry k)
    const float xa = k < SDC_ACT_IN ? xn[a] : 0.0f;
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(xa), __float_as_uint(xa), false, false);
    a_operands(P[a], __uint_as_float(s[0]), __uint_as_float(s[1]), odd);
    acc[a] = f4{0.f, 0.f, 0.f, 0.f};
  }
  // ---- layer 1: 26 (28) -> 64 ------------------------------------------------------------------------------------------
  Chunks<0, SDC_ACT_M1, false, 0>::run(acc, P, W, A, lane);
  constexpr int PAR2 = ((SDC_ACT_M1 + SDC_ACT_CH - 1) / SDC_ACT_CH) & 1;      // (the buffer layer 2's first chunk went to)
  float h[3][2];
  {
    const float bias[3] = {A[0].b1[lane], A[1].b1[lane], A[2].b1[lane]};
    const float g[3] = {A[0].ln1_g[lane], A[1].ln1_g[lane], A[2].ln1_g[lane]};
    const float b[3] = {A[0].ln1_b[lane], A[1].ln1_b[lane], A[2].ln1_b[lane]};
    epilogue<KIND>(acc, bias, g, b, h);
  }
  // ---- layer 2: 64 -> 64 -----------------------------------------------------------------------------------------------
#pragma unroll
  for (int a = 0; a < 3; a++) {
    a_operands(P[a], h[a][0], h[a][1], odd);
    acc[a] = f4{0.f, 0.f, 0.f, 0.f};
  }
  Chunks<0, SDC_ACT_M2, true, PAR2>::run(acc, P, W, A, lane);
  {
    const float bias[3] = {A[0].b2[lane], A[1].b2[lane], A[2].b2[lane]};
    const float g[3] = {A[0].ln2_g[lane], A[1].ln2_g[lane], A[2].ln2_g[lane]};
    const float b[3] = {A[0].ln2_b[lane], A[1].ln2_b[lane], A[2].ln2_b[lane]};
    epilogue<KIND>(acc, bias, g, b, h);
  }
  // ---- the three logits per env: 64 -> 3, a product per lane-resident unit and a reduction; every lane ends with ITS env's ---
  float p[10][2];                       // [agent a, action c -> 3 a + c][env]; entry 9 pads the pairs
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int c = 0; c < SDC_ACT_OUT; c++) {
      const float w3 = A[a].w3[c][lane];
      p[3 * a + c][0] = w3 * h[a][0];
      p[3 * a + c][1] = w3 * h[a][1];
    }
  p[9][0] = p[9][1] = 0.0f;
  {
    float r[18];
#pragma unroll
    for (int i = 0; i < 9; i++) { r[2 * i] = p[i][0]; r[2 * i + 1] = p[i][1]; }
    row_sums<18>(r);
#pragma unroll
    for (int i = 0; i < 9; i++) { p[i][0] = r[2 * i]; p[i][1] = r[2 * i + 1]; }
  }
  float own[10];
#pragma unroll
  for (int i = 0; i < 10; i += 2)       // rows [X (env 0), Y (env 0), X (env 1), Y (env 1)], X = entry i, Y = entry i + 1
    spread16(swap32_sum(swap16_sum(p[i][0], p[i + 1][0]), swap16_sum(p[i][1], p[i + 1][1])), own[i], own[i + 1]);
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int c = 0; c < SDC_ACT_OUT; c++) lg[a][c] = own[3 * a + c] + A[a].b3[c];
}
// (the activation is a template parameter: chosen per element at run time it became a branch around every exp -> rcp chain;
// the three agents share it -- sdc_set_actor refuses a mix)
__device__ __forceinline__ void forward3(const SdcActorDev* A, const float (&x)[3], const int lane, float (&lg)[3][3]) {
  if (__builtin_amdgcn_readfirstlane((A[0].flags >> 1) & 3) == 1) forward3_kind<1>(A, x, lane, lg);
  else forward3_kind<0>(A, x, lane, lg);
}

// ---- FOUR ENVS PER WAVEFRONT (sdc_rollout_actor_quad_kernel: large batches) -----------------------------------------------
// The same networks with all four rows of the 4 x 64 product in use: the 276 MFMAs per wavefront-step serve four envs.
// The A operands are quad broadcasts of FOUR activation registers (lane = entry k, register = env) merged on lane % 4.
__device__ __forceinline__ void a_operands4(float (&P)[4], const float x0, const float x1, const float x2, const float x3,
                                            const int lane) {
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
#define SDC_ACT_P(V)                                                                              \
  {                                                                                               \
    const float e0 = quad_bcast<V>(x0), e1 = quad_bcast<V>(x1), e2 = quad_bcast<V>(x2), e3 = quad_bcast<V>(x3); \
    const float lo = b0 ? e1 : e0, hi = b0 ? e3 : e2;                                             \
    P[V] = b1 ? hi : lo;                                                                          \
  }
  SDC_ACT_P(0) SDC_ACT_P(1) SDC_ACT_P(2) SDC_ACT_P(3)
#undef SDC_ACT_P
}
// bias, activation, LayerNorm(64) of the three agents' layer outputs, four envs: acc[a][e] -> h[a][e] (lane = unit).  The
// 24 sums: four DPP stages inside the rows, then per agent and moment the four envs' row sums merged into one register
// (rows [env 0, 1, 2, 3] totals) and read back as wave-uniform values.
template <int KIND>
__device__ __forceinline__ void epilogue4(const f4 (&acc)[3], const float (&bias)[3], const float (&g)[3], const float (&b)[3],
                                          float (&h)[3][4]) {
  float r[24];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float v = activate<KIND>(acc[a][e] + bias[a]);
      const float d = v - __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));   // (shifted moments: see epilogue)
      h[a][e] = d;
      r[8 * a + e] = d;
      r[8 * a + 4 + e] = d * d;
    }
  row_sums<24>(r);
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float t1 = swap32_sum(swap16_sum(r[8 * a], r[8 * a + 1]), swap16_sum(r[8 * a + 2], r[8 * a + 3]));
    const float t2 = swap32_sum(swap16_sum(r[8 * a + 4], r[8 * a + 5]), swap16_sum(r[8 * a + 6], r[8 * a + 7]));
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float mean = row_value(t1, e) * (1.0f / SDC_ACT_H);
      const float var = fmaxf(row_value(t2, e) * (1.0f / SDC_ACT_H) - mean * mean, 0.0f);
      h[a][e] = (h[a][e] - mean) * __builtin_amdgcn_rsqf(var + 1e-5f) * g[a] + b[a];
    }
  }
}
// x[a][j]: lane (half H, k) holds entry k < 26 (0 beyond) of agent a's padded observation of env 2 H + j.
// lg[a][0..2] = agent a's logits for the env of THE LANE'S ROW (row r of the wavefront = env r).
template <int KIND>
__device__ __forceinline__ void forward3_quad_kind(const SdcActorDev* A, const float (&x)[3][2], const int lane, float (&lg)[3][3]) {
  const int k = lane & 31;
  WBuf W;
  fetch3<SDC_ACT_CH>(W.w[0], A, false, 0, lane);
  // feature LayerNorm: sums over a half (two rows); rows of the merged registers = envs 0..3
  float xn[3][2];
  {
    float r[12];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      r[4 * a] = x[a][0]; r[4 * a + 1] = x[a][1];
      r[4 * a + 2] = x[a][0] * x[a][0]; r[4 * a + 3] = x[a][1] * x[a][1];
    }
    row_sums<12>(r);
#pragma unroll
    for (int a = 0; a < 3; a++) {
      float s0, s1, q0, q1;         // this lane's half: sum x / sum x^2 of its env j = 0 and j = 1
      spread16(swap16_sum(r[4 * a], r[4 * a + 1]), s0, s1);
      spread16(swap16_sum(r[4 * a + 2], r[4 * a + 3]), q0, q1);
      xn[a][0] = x[a][0];
      xn[a][1] = x[a][1];
      if (A[a].flags & 1) {
        const float g0 = A[a].ln0_g[k], b0 = A[a].ln0_b[k];
        const float m0 = s0 * (1.0f / SDC_ACT_IN), m1 = s1 * (1.0f / SDC_ACT_IN);
        const float v0 = fmaxf(q0 * (1.0f / SDC_ACT_IN) - m0 * m0, 0.0f), v1 = fmaxf(q1 * (1.0f / SDC_ACT_IN) - m1 * m1, 0.0f);
        xn[a][0] = (x[a][0] - m0) * __builtin_amdgcn_rsqf(v0 + 1e-5f) * g0 + b0;
        xn[a][1] = (x[a][1] - m1) * __builtin_amdgcn_rsqf(v1 + 1e-5f) * g0 + b0;
      }
    }
  }
  float P[3][4];
  f4 acc[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    // envs 0, 1 sit in lanes 0..25, envs 2, 3 in lanes 32..57: all four to lanes 0..25 (the lane = the entry k)
    const float xa = k < SDC_ACT_IN ? xn[a][0] : 0.0f, xb = k < SDC_ACT_IN ? xn[a][1] : 0.0f;
    const auto sa = __builtin_amdgcn_permlane32_swap(__float_as_uint(xa), __float_as_uint(xa), false, false);
    const auto sb = __builtin_amdgcn_permlane32_swap(__float_as_uint(xb), __float_as_uint(xb), false, false);
    a_operands4(P[a], __uint_as_float(sa[0]), __uint_as_float(sb[0]), __uint_as_float(sa[1]), __uint_as_float(sb[1]), lane);
    acc[a] = f4{0.f, 0.f, 0.f, 0.f};
  }
  Chunks<0, SDC_ACT_M1, false, 0>::run(acc, P, W, A, lane);
  constexpr int PAR2 = ((SDC_ACT_M1 + SDC_ACT_CH - 1) / SDC_ACT_CH) & 1;
  float h[3][4];
  {
    const float bias[3] = {A[0].b1[lane], A[1].b1[lane], A[2].b1[lane]};
    const float g[3] = {A[0].ln1_g[lane], A[1].ln1_g[lane], A[2].ln1_g[lane]};
    const float b[3] = {A[0].ln1_b[lane], A[1].ln1_b[lane], A[2].ln1_b[lane]};
    epilogue4<KIND>(acc, bias, g, b, h);
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    a_operands4(P[a], h[a][0], h[a][1], h[a][2], h[a][3], lane);
    acc[a] = f4{0.f, 0.f, 0.f, 0.f};
  }
  Chunks<0, SDC_ACT_M2, true, PAR2>::run(acc, P, W, A, lane);
  {
    const float bias[3] = {A[0].b2[lane], A[1].b2[lane], A[2].b2[lane]};
    const float g[3] = {A[0].ln2_g[lane], A[1].ln2_g[lane], A[2].ln2_g[lane]};
    const float b[3] = {A[0].ln2_b[lane], A[1].ln2_b[lane], A[2].ln2_b[lane]};
    epilogue4<KIND>(acc, bias, g, b, h);
  }
  // the logits: per (agent, action) the four envs' products reduced into one register whose ROW e holds env e's total
#pragma unroll
  for (int a = 0; a < 3; a++) {
    float r[12];
#pragma unroll
    for (int c = 0; c < SDC_ACT_OUT; c++) {
      const float w3 = A[a].w3[c][lane];
#pragma unroll
      for (int e = 0; e < 4; e++) r[4 * c + e] = w3 * h[a][e];
    }
    row_sums<12>(r);
#pragma unroll
    for (int c = 0; c < SDC_ACT_OUT; c++)
      lg[a][c] = swap32_sum(swap16_sum(r[4 * c], r[4 * c + 1]), swap16_sum(r[4 * c + 2], r[4 * c + 3])) + A[a].b3[c];
  }
}
__device__ __forceinline__ void forward3_quad(const SdcActorDev* A, const float (&x)[3][2], const int lane, float (&lg)[3][3]) {
  if (__builtin_amdgcn_readfirstlane((A[0].flags >> 1) & 3) == 1) forward3_quad_kind<1>(A, x, lane, lg);
  else forward3_quad_kind<0>(A, x, lane, lg);
}

// the action of one env from its three logits: mode() = first maximum (torch argmax), or a draw from
// softmax(logits) by inverse CDF with the uniform u in [0, 1)
__device__ __forceinline__ int pick_action(const float l0, const float l1, const float l2, const bool sample, const float u) {
  if (!sample) return (l1 > l0 && l1 >= l2) ? 1 : ((l2 > l0 && l2 > l1) ? 2 : 0);
  const float m = fmaxf(l0, fmaxf(l1, l2));
  const float e0 = __builtin_amdgcn_exp2f((l0 - m) * 1.4426950408889634f), e1 = __builtin_amdgcn_exp2f((l1 - m) * 1.4426950408889634f),
              e2 = __builtin_amdgcn_exp2f((l2 - m) * 1.4426950408889634f);
  const float t = u * (e0 + e1 + e2);
  return (t >= e0 ? 1 : 0) + (t >= e0 + e1 ? 1 : 0);
}

}  // namespace sdc_act
