// Pieces of the fused CTC prefix search (ctc_search.hip): tuning constants, the softmax numerator,
// and the LDS ring slot that producer and consumer waves of an utterance share.
#pragma once
#include "ctc_frame.hpp"

#ifndef PDT_SHORT_MIN  // window and target of the short lists' survivor count
#define PDT_SHORT_MIN 8
#define PDT_SHORT_LO 14
#define PDT_SHORT_HI 26
#define PDT_SHORT_PROBE 18
#endif
#ifndef PDT_SPIN_SLEEP
#define PDT_SPIN_SLEEP 2
#endif
#ifndef PDT_UTT_PER_WG  // one-producer form: utterances per workgroup and ring depth (LDS permitting)
#define PDT_UTT_PER_WG 2
#define PDT_RING_STAGES 4
#endif
#ifndef PDT_CONSUMER_PRIO
#define PDT_CONSUMER_PRIO 3
#endif

namespace pdt {

// exp(x) for x <= 0 (softmax numerators): OCML's expf without its overflow / underflow guards --
// the same hi/lo range reduction around v_exp_f32; v_ldexp_f32 flushes what underflows.
// Arguments below -200 (masked vocabulary entries: -inf logits) are clamped first: the result
// underflows to 0 either way, but -inf would turn the rounding-error term into inf - inf.
__device__ __forceinline__ float exp_nonpos(float x) {
  x = fmaxf(x, -200.0f);
  const float t = x * 0x1.715476p+0f;                       // x * log2(e), rounded
  const float lo = __builtin_fmaf(x, 0x1.715476p+0f, -t);   // its rounding error
  const float n = __builtin_rintf(t);
  const float f = (t - n) + __builtin_fmaf(x, 0x1.4ae0bep-26f, lo);  // + x * (log2(e) - fl(log2(e)))
  return __builtin_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// max of two floats without the input canonicalisation fmaxf() has to add for values that come
// straight from memory (NaN logits are outside the contract either way)
__device__ __forceinline__ float fmax_raw(float a, float b) {
  float r;
  asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ float fmax3_raw(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// two at a time: the multiplies / fmas / adds of the range reduction as packed fp32 (one
// instruction for both); identical arithmetic per element, so identical results
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 exp_nonpos2(f32x2 x) {
  const f32x2 L = {0x1.715476p+0f, 0x1.715476p+0f}, L2 = {0x1.4ae0bep-26f, 0x1.4ae0bep-26f};
  x = __builtin_elementwise_max(x, f32x2{-200.0f, -200.0f});
  const f32x2 t = x * L;
  const f32x2 lo = __builtin_elementwise_fma(x, L, -t);
  const f32x2 n = {__builtin_rintf(t.x), __builtin_rintf(t.y)};
  const f32x2 f = (t - n) + __builtin_elementwise_fma(x, L2, lo);
  return f32x2{__builtin_ldexpf(__builtin_amdgcn_exp2f(f.x), (int)n.x),
               __builtin_ldexpf(__builtin_amdgcn_exp2f(f.y), (int)n.y)};
}

__device__ __forceinline__ float fmin_raw(float a, float b) {
  float r;
  asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fmin3_raw(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// exp_nonpos2 for arguments in [-86, 0] (every result a NORMAL float32), cheaper by a third and the
// same bits: no clamp; the rounding to the nearest integer by adding and subtracting 1.5 * 2^23 (round
// to nearest even, as v_rndne_f32: two 2.5-cycle adds, packed, instead of a 4.3-cycle rndne + cvt per
// element); and 2^n applied by adding n to the exponent field -- n sits in the low mantissa bits of
// t + 1.5 * 2^23, shifted up 23 bits everything else falls off the word -- instead of v_ldexp_f32, which
// differs only where the result would be a denormal (n < -126 + the exponent of exp2(f): excluded).
__device__ __forceinline__ f32x2 exp_tame2(const f32x2 x) {
  const f32x2 L = {0x1.715476p+0f, 0x1.715476p+0f}, L2 = {0x1.4ae0bep-26f, 0x1.4ae0bep-26f};
  const f32x2 MG = {12582912.0f, 12582912.0f};
  const f32x2 t = x * L;
  const f32x2 lo = __builtin_elementwise_fma(x, L, -t);
  const f32x2 tm = t + MG;
  const f32x2 n = tm - MG;
  const f32x2 f = (t - n) + __builtin_elementwise_fma(x, L2, lo);
  const float y0 = __builtin_amdgcn_exp2f(f.x), y1 = __builtin_amdgcn_exp2f(f.y);
  return f32x2{__uint_as_float(__float_as_uint(y0) + (__float_as_uint(tm.x) << 23)),
               __uint_as_float(__float_as_uint(y1) + (__float_as_uint(tm.y) << 23))};
}

// Reductions over each 32-lane HALF of a wave (the two-frames-per-pass producer of ctc_search.hip).
// Maximum of floats that are not NaN, left in EVERY lane of the half: rotations inside the 16-lane rows
// (max is idempotent: after ror 1, 2, 4, 8 every lane holds its row's maximum), then the two rows of a
// half are exchanged by v_permlane16_swap (first result: rows 0, 0, 2, 2; second: rows 1, 1, 3, 3).
__device__ __forceinline__ float half_max_all_f(float x) {
  float v = x;
  asm volatile(
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmax_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// Sum over a half, left in ITS LAST lane (31 / 63): wave_sum_f's DPP steps without the last one, so the
// two 16-lane row totals R0, R1 of a half come out as R0 + R1 in exactly that routine's association.
__device__ __forceinline__ float half_sum_at31(float x) {
  float v = x;
  asm volatile(
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}

// LDS ring slot shared by the producer and consumer waves of one utterance.
struct RingLayout {
  int row_floats;   // V + 1 padded to 4
  int pos_bytes;    // V padded to 16
  int slot_bytes;
  int nstage;
  int utt_bytes;    // ring + consumer scratch + producer scratch + flags
  int utt_per_wg;
  int producers;    // producer waves per utterance
  // rows too long for any LDS ring: the row of probabilities and the token -> list-position table
  // of every slot live in the HBM workspace instead (g_slot_bytes per slot, L2-resident); the LDS
  // slot keeps the list and the header
  int row_global, g_slot_bytes;
};

__host__ __device__ inline RingLayout ring_layout(int V, int W, int nstage, int utt_per_wg,
                                                  int producers, bool row_global = false) {
  RingLayout r;
  r.row_floats = (V + 1 + 3) & ~3;
  r.pos_bytes = (V + 15) & ~15;
  r.row_global = row_global ? 1 : 0;
  r.g_slot_bytes = row_global ? r.row_floats * 4 + r.pos_bytes : 0;
  // (8 KiB per slot at least in that form: the freed ring holds the checkpoint table of the
  // output walk)
  r.slot_bytes = row_global ? 8192 : r.row_floats * 4 + PDT_WAVE * 8 + r.pos_bytes + 16;
  r.nstage = nstage;
  const int consumer = consumer_scratch_bytes(W);  // nxt tables + chm + info
  r.utt_bytes = (r.slot_bytes * nstage + consumer + producers * PDT_SURV_CAP * 8 + 32 + 15) & ~15;
  r.utt_per_wg = utt_per_wg;
  r.producers = producers;
  return r;
}

}  // namespace pdt
