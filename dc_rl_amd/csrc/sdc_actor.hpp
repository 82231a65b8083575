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
// Mapping: lane = hidden unit (64 lanes = 64 units), BOTH envs of the wavefront at once (two accumulators per lane share
// every weight read); the layer input is handed round as float2 {env 0, env 1} through LDS (broadcast reads); the three
// agents' weights sit in LDS once per workgroup (76 KB), k-major and interleaved in pairs so that a lane reads the two
// weights of inputs 2i, 2i + 1 with one ds_read_b64.  LayerNorm / logits: DPP reductions.  This is a contraction, but
// at 2 envs per wavefront an MFMA tile (>= 16 rows) would be 7/8 padding; the fp32 VALU form costs ~470 instructions
// per agent and pair.
#pragma once
#include "sdc_device.hpp"

#define SDC_ACT_IN SDC_OBS_PAD     // 26
#define SDC_ACT_H 64
#define SDC_ACT_OUT 3

// one agent's actor as the kernel reads it (device memory, then LDS); filled by sdc_set_actor (sdc_capi.hip)
struct SdcActorDev {
  float ln0_g[32], ln0_b[32];                 // feature LayerNorm over the 26 inputs (entries 26.. unused)
  float w1[SDC_ACT_IN / 2][SDC_ACT_H][2];     // w1[i][j] = {W1[j][2i], W1[j][2i+1]}   (W as torch stores it: [out][in])
  float b1[SDC_ACT_H], ln1_g[SDC_ACT_H], ln1_b[SDC_ACT_H];
  float w2[SDC_ACT_H / 2][SDC_ACT_H][2];
  float b2[SDC_ACT_H], ln2_g[SDC_ACT_H], ln2_b[SDC_ACT_H];
  float w3[4][SDC_ACT_H];                     // w3[c][j] = W3[c][j], c < 3
  float b3[4];
  int flags;                                  // bit 0: feature LayerNorm on; bits 1-2: activation (0 tanh, 1 relu)
  int pad[3];
};
static_assert(sizeof(SdcActorDev) % 16 == 0, "copied to LDS as uint4");

namespace sdc_act {

typedef float v2f __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp_f32(const float v) {   // (every row written, no source lane -> 0)
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 32 lanes of each half (every lane gets its half's sum): the tree of half_sum_f64
__device__ __forceinline__ float half_sum_f32(float v) {
  v += dpp_f32<SDC_DPP_XOR1>(v);
  v += dpp_f32<SDC_DPP_XOR2>(v);
  v += dpp_f32<SDC_DPP_HALF_MIRROR>(v);
  v += dpp_f32<SDC_DPP_MIRROR>(v);
  const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(s[1]) + __uint_as_float(s[0]);
}
// sum over all 64 lanes, every lane gets it
__device__ __forceinline__ float wave_sum_f32(float v) {
  v = half_sum_f32(v);
  const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(s[1]) + __uint_as_float(s[0]);
}
// N sums at once, stage by stage: a reduction is a chain of seven dependent steps, each a DPP / permlane operation that
// must wait out the previous one (the compiler pads a lone chain with s_nop); N independent chains fill each other's gaps
template <int N, bool HALF>
__device__ __forceinline__ void sums(float (&v)[N]) {
#define SDC_ACT_STAGE(C)  \
  _Pragma("unroll") for (int i = 0; i < N; i++) v[i] += dpp_f32<C>(v[i]);
  SDC_ACT_STAGE(SDC_DPP_XOR1)
  SDC_ACT_STAGE(SDC_DPP_XOR2)
  SDC_ACT_STAGE(SDC_DPP_HALF_MIRROR)
  SDC_ACT_STAGE(SDC_DPP_MIRROR)
#undef SDC_ACT_STAGE
#pragma unroll
  for (int i = 0; i < N; i++) {
    const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false);
    v[i] = __uint_as_float(s[1]) + __uint_as_float(s[0]);
  }
  if constexpr (!HALF) {
#pragma unroll
    for (int i = 0; i < N; i++) {
      const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false);
      v[i] = __uint_as_float(s[1]) + __uint_as_float(s[0]);
    }
  }
}
__device__ __forceinline__ float activate(const float x, const int kind) {
  if (kind == 1) return x > 0.0f ? x : 0.0f;
  // tanh x = 1 - 2 / (e^(2x) + 1): hardware exp2 / rcp (~1 ulp each), exact limits at +-inf
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
// LayerNorm over the 64 lanes (torch: biased variance, eps 1e-5), both envs
// (one pass: sum and sum of squares of both envs as four interleaved reductions, variance = E[x^2] - mean^2; the inputs
// are activations in [-1, 1] or a ReLU's outputs of order one: the cancellation costs ~1e-7 of the value)
__device__ __forceinline__ void layer_norm64(float& a0, float& a1, const float g, const float b) {
  float r[4] = {a0, a1, a0 * a0, a1 * a1};
  sums<4, false>(r);
  const float m0 = r[0] * (1.0f / SDC_ACT_H), m1 = r[1] * (1.0f / SDC_ACT_H);
  const float v0 = fmaxf(r[2] * (1.0f / SDC_ACT_H) - m0 * m0, 0.0f), v1 = fmaxf(r[3] * (1.0f / SDC_ACT_H) - m1 * m1, 0.0f);
  a0 = (a0 - m0) * __builtin_amdgcn_rsqf(v0 + 1e-5f) * g + b;
  a1 = (a1 - m1) * __builtin_amdgcn_rsqf(v1 + 1e-5f) * g + b;
}

// One agent's forward pass for both envs of the wavefront.  x: this lane's input (lane (h, k): entry k < 26 of env h's
// padded observation of this agent, 0 beyond); xs: the wavefront's LDS hand-round buffer, float2[64].
// Returns the three logits of env 0 in lg[0..2], of env 1 in lg[3..5] (every lane).
__device__ __forceinline__ void forward(const SdcActorDev& A, const float x, const int lane, float2* xs, float (&lg)[6]) {
  const int h = lane >> 5, k = lane & 31;
  const int kind = (A.flags >> 1) & 3;
  float xn = x;
  if (A.flags & 1) {
    // (observation entries are of order one -- normalised by construction -- so the one-pass variance is safe here too)
    float r[2] = {x, x * x};
    sums<2, true>(r);
    const float mean = r[0] * (1.0f / SDC_ACT_IN);
    const float var = fmaxf(r[1] * (1.0f / SDC_ACT_IN) - mean * mean, 0.0f);
    xn = (x - mean) * __builtin_amdgcn_rsqf(var + 1e-5f) * A.ln0_g[k] + A.ln0_b[k];
  }
  wave_sync();                                     // (the previous layer's readers are done with xs)
  if (k < SDC_ACT_IN) reinterpret_cast<float*>(xs)[2 * k + h] = xn;
  wave_sync();
  // (both envs in one packed instruction: v_pk_fma_f32 {acc0, acc1} += {w, w} * {x0, x1}, the weight broadcast by op_sel)
  v2f acc = {A.b1[lane], A.b1[lane]};
#pragma unroll
  for (int i = 0; i < SDC_ACT_IN / 2; i++) {
    const float4 xx = reinterpret_cast<const float4*>(xs)[i];          // {x0[2i], x1[2i], x0[2i+1], x1[2i+1]}, broadcast
    const float2 w = *reinterpret_cast<const float2*>(A.w1[i][lane]);
    acc = __builtin_elementwise_fma(v2f{w.x, w.x}, v2f{xx.x, xx.y}, acc);
    acc = __builtin_elementwise_fma(v2f{w.y, w.y}, v2f{xx.z, xx.w}, acc);
  }
  float a0 = activate(acc.x, kind);
  float a1 = activate(acc.y, kind);
  layer_norm64(a0, a1, A.ln1_g[lane], A.ln1_b[lane]);
  wave_sync();
  xs[lane] = make_float2(a0, a1);
  wave_sync();
  v2f acc2 = {A.b2[lane], A.b2[lane]};
#pragma unroll 16
  for (int i = 0; i < SDC_ACT_H / 2; i++) {
    const float4 xx = reinterpret_cast<const float4*>(xs)[i];
    const float2 w = *reinterpret_cast<const float2*>(A.w2[i][lane]);
    acc2 = __builtin_elementwise_fma(v2f{w.x, w.x}, v2f{xx.x, xx.y}, acc2);
    acc2 = __builtin_elementwise_fma(v2f{w.y, w.y}, v2f{xx.z, xx.w}, acc2);
  }
  float c0 = activate(acc2.x, kind);
  float c1 = activate(acc2.y, kind);
  layer_norm64(c0, c1, A.ln2_g[lane], A.ln2_b[lane]);
#pragma unroll
  for (int c = 0; c < SDC_ACT_OUT; c++) {
    const float w = A.w3[c][lane];
    lg[c] = w * c0;
    lg[3 + c] = w * c1;
  }
  sums<6, false>(lg);
#pragma unroll
  for (int c = 0; c < SDC_ACT_OUT; c++) {
    lg[c] += A.b3[c];
    lg[3 + c] += A.b3[c];
  }
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
