// Wave-level (64-lane) ordering primitives used by the beam-search kernels:
// order-preserving float keys, bitonic sort of one key per lane, top-64 merge, and
// "sorted top-M of a V-vector" selection.  No LDS memory traffic except the small survivor list.
#pragma once
#include "pdt_common.hpp"

namespace pdt {

typedef unsigned long long u64;

// monotone float -> uint map (any finite value, +-inf): a < b  <=>  fkey(a) < fkey(b).
// fkey(x) > 0 for every x, so 0 can serve as "no candidate".
__device__ __forceinline__ unsigned fkey(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// the same for values known to be >= +0 (probabilities): the IEEE bit pattern is already
// monotone there; +1 keeps 0 free as "no candidate".  One instruction instead of three.
__device__ __forceinline__ unsigned fkey_nonneg(float f) { return __float_as_uint(f) + 1u; }
__device__ __forceinline__ float fkey_nonneg_inv(unsigned k) { return __uint_as_float(k - 1u); }
// (value key, index) -> one u64 whose max is "largest value, then lowest index"
__device__ __forceinline__ u64 pack_key(unsigned key, unsigned idx) {
  return ((u64)key << 32) | (u64)(0xffffffffu - idx);
}
__device__ __forceinline__ unsigned key_of(u64 k) { return (unsigned)(k >> 32); }
__device__ __forceinline__ unsigned idx_of(u64 k) { return 0xffffffffu - (unsigned)k; }

// value of lane `src` (0..63; taken mod 64): the bare LDS-crossbar permute.  HIP's __shfl adds
// two VALU instructions of width arithmetic that a 64-wide wave does not need.
__device__ __forceinline__ int shfl_i(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
__device__ __forceinline__ u64 shfl_u64(u64 v, int src) {
  const unsigned lo = (unsigned)shfl_i((int)(unsigned)v, src);
  const unsigned hi = (unsigned)shfl_i((int)(unsigned)(v >> 32), src);
  return ((u64)hi << 32) | lo;
}

// lanes whose index has bit B clear: the lower lane of every pair (l, l ^ X) when B is the top
// set bit of X
constexpr u64 lanes_with_bit_clear(int B) {
  u64 m = 0;
  for (int l = 0; l < 64; ++l)
    if ((l & B) == 0) m |= 1ull << l;
  return m;
}

// The mask as a lane predicate, materialised where it is used by two s_mov_b32 the compiler
// may not hoist (hoisted masks filled the scalar registers of the frame loops, and their spills
// are v_readlane / v_writelane in a VALU-bound kernel).
template <u64 MASK>
__device__ __forceinline__ bool lane_predicate() {
  return __builtin_amdgcn_inverse_ballot_w64(MASK);
}

// value of lane (lane ^ X).  One DPP move where one exists: quad_perm for X = 1, 2, 3,
// row_half_mirror / row_mirror for X = 7 / 15, row_ror:8 for X = 8.  The others (4, 16, 31, 32,
// 63) would take two or three VALU instructions each; the kernels that sort are VALU-issue-bound
// while their LDS pipe is nearly idle, so those go through the LDS crossbar (ds_bpermute: no LDS
// memory is touched) -- measured 3.88 -> 3.80 ms on the CTC search; routing EVERY stage through
// the crossbar instead lengthens the dependency chain too much (4.23 ms).
// (mov_dpp: every lane has a source, so there is no "old" value to copy.)
template <int X>
constexpr bool xor_is_dpp() { return X == 1 || X == 2 || X == 3 || X == 7 || X == 8 || X == 15; }
template <int X>
__device__ __forceinline__ unsigned xor_shfl(unsigned v) {
  if constexpr (X == 1) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  } else if constexpr (X == 2) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  } else if constexpr (X == 3) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x1B, 0xf, 0xf, true);   // quad_perm [3,2,1,0]
  } else if constexpr (X == 7) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);  // row_half_mirror
  } else if constexpr (X == 15) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true);  // row_mirror
  } else if constexpr (X == 8) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, true);  // row_ror:8
  } else {
    static_assert(X == 4 || X == 16 || X == 31 || X == 32 || X == 63, "xor_shfl: unsupported pattern");
#ifdef PDT_SORT_VALU
    // Register-only forms (gfx950): the crossbar answers after ~70 cycles alone and ~190 with
    // every wave of the CU using it, a dependent VALU instruction after 8-30 -- and the
    // consumer of the CTC search is one dependency chain.
    if constexpr (X == 4) {
      // lanes 0-3 / 8-11 of a row read 4 lanes up, lanes 4-7 / 12-15 read 4 lanes down
      int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xf, 0x5, false);  // row_shl:4, banks 0 and 2
      t = __builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xf, 0xA, false);            // row_shr:4, banks 1 and 3
      return (unsigned)t;
    } else if constexpr (X == 16) {
      // v_permlane16_swap: odd rows of the first operand <-> even rows of the second; with both
      // = v the first becomes rows (0, 0, 2, 2) and the second rows (1, 1, 3, 3)
      const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
      return lane_predicate<lanes_with_bit_clear(16)>() ? r[1] : r[0];
    } else if constexpr (X == 32) {
      // v_permlane32_swap: upper half of the first operand <-> lower half of the second
      const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
      return lane_predicate<lanes_with_bit_clear(32)>() ? r[1] : r[0];
    } else if constexpr (X == 31) {
      return xor_shfl<16>(xor_shfl<15>(v));
    } else {
      return xor_shfl<32>(xor_shfl<16>(xor_shfl<15>(v)));
    }
#else
    return (unsigned)__builtin_amdgcn_ds_bpermute((int)((lane_id() ^ X) << 2), (int)v);
#endif
  }
}
template <int X>
__device__ __forceinline__ u64 xor_shfl(u64 v) {
  return ((u64)xor_shfl<X>((unsigned)(v >> 32)) << 32) | xor_shfl<X>((unsigned)v);
}

// One compare-exchange stage of a DESCENDING sort: partners (l, l ^ X), the lower lane keeps the
// larger key.  B = top set bit of X.
template <int X, int B, typename T>
__device__ __forceinline__ T cmpx_stage(T key) {
#ifndef PDT_NO_BANK_STAGES
  if constexpr (sizeof(T) == 4 && (X == 4 || X == 7 || X == 8 || X == 15)) {
    // partners a DPP row operation reaches AND "keep the larger" lanes that are whole 4-lane banks
    // (B = 4: banks 0 and 2; B = 8: banks 0 and 1): the bank mask of the DPP operand does the
    // selection -- max over the lower banks, min over the upper ones, both from the stage's input:
    // 2 VALU, no lane predicate (l ^ 4 went through the LDS crossbar before: 4 instructions)
    constexpr int LO = B == 4 ? 0x5 : 0x3, HI = 0xf & ~LO;
    constexpr int C_LO = X == 4 ? 0x104 /* row_shl:4 */ : (X == 7 ? 0x141 : (X == 8 ? 0x128 : 0x140));
    constexpr int C_HI = X == 4 ? 0x114 /* row_shr:4 */ : C_LO;
    const unsigned k = (unsigned)key;
    const unsigned t = max(k, (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, C_LO, 0xf, LO, false));
    return (T)min(t, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)k, C_HI, 0xf, HI, false));
  }
#endif
  const T other = xor_shfl<X>(key);
  const bool lower = lane_predicate<lanes_with_bit_clear(B)>();
  if constexpr (sizeof(T) == 4 && xor_is_dpp<X>()) {
    // max / min fold the DPP move into themselves: 3 VALU, no scalar work
    return lower ? max(key, other) : min(key, other);
  } else {
    return ((key > other) == lower) ? key : other;
  }
}
// half-cleaners of a bitonic sequence of 2 * J lanes: l ^ J, l ^ J/2, ..., l ^ 1
template <int J, typename T>
__device__ __forceinline__ T bitonic_merge(T key) {
  key = cmpx_stage<J, J>(key);
  if constexpr (J > 1) key = bitonic_merge<J / 2>(key);
  return key;
}
// sorts every aligned group of S lanes, given sorted groups of S / 2: a "flip" (l ^ (S - 1):
// both halves descending become one bitonic exchange) and the half-cleaners below it.  All
// groups come out descending, so the lane predicates are the six single-bit masks.
template <int S, typename T>
__device__ __forceinline__ T sort_groups(T key) {
  key = cmpx_stage<S - 1, S / 2>(key);
  if constexpr (S > 2) key = bitonic_merge<S / 4>(key);
  return key;
}

// bitonic sort of one key per lane, DESCENDING (lane 0 ends with the maximum); 21
// compare-exchange stages, 6 of them through the LDS crossbar
template <typename T>
__device__ __forceinline__ T wave_sort_desc(T key) {
  key = sort_groups<2>(key);
  key = sort_groups<4>(key);
  key = sort_groups<8>(key);
  key = sort_groups<16>(key);
  key = sort_groups<32>(key);
  key = sort_groups<64>(key);
  return key;
}

// the same network stopped after 15 stages: each half of the wave holds ITS 32 keys sorted
// descending
template <typename T>
__device__ __forceinline__ T half_wave_sort_desc(T key) {
  key = sort_groups<2>(key);
  key = sort_groups<4>(key);
  key = sort_groups<8>(key);
  key = sort_groups<16>(key);
  key = sort_groups<32>(key);
  return key;
}

// ... and after 10 stages: every aligned group of 16 lanes sorted descending
template <typename T>
__device__ __forceinline__ T row_sort_desc(T key) {
  key = sort_groups<2>(key);
  key = sort_groups<4>(key);
  key = sort_groups<8>(key);
  key = sort_groups<16>(key);
  return key;
}

// cur: sorted descending; add: arbitrary.  Returns the 64 largest of the union, sorted descending.
__device__ __forceinline__ u64 wave_merge_top64(u64 cur, u64 add) {
  const int lane = lane_id();
  add = wave_sort_desc<u64>(add);
  const u64 rev = shfl_u64(add, PDT_WAVE - 1 - lane);
  const u64 key = cur > rev ? cur : rev;  // bitonic
  return bitonic_merge<32>(key);
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
  unsigned v = x;
  v = max(v, (unsigned)dpp_or<PDT_DPP_ROW_SHR(1)>((int)v, 0));
  v = max(v, (unsigned)dpp_or<PDT_DPP_ROW_SHR(2)>((int)v, 0));
  v = max(v, (unsigned)dpp_or<PDT_DPP_ROW_SHR(4), 0xf, 0xe>((int)v, 0));
  v = max(v, (unsigned)dpp_or<PDT_DPP_ROW_SHR(8), 0xf, 0xc>((int)v, 0));
  v = max(v, (unsigned)dpp_or<PDT_DPP_ROW_BCAST15, 0xa>((int)v, 0));
  v = max(v, (unsigned)dpp_or<PDT_DPP_ROW_BCAST31, 0xc>((int)v, 0));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// max over lanes 0..15 only (one DPP row): 4 steps instead of 6
__device__ __forceinline__ unsigned row0_max_u32(unsigned x) {
  unsigned v = x;
  v = max(v, (unsigned)dpp_or<PDT_DPP_ROW_SHR(1)>((int)v, 0));
  v = max(v, (unsigned)dpp_or<PDT_DPP_ROW_SHR(2)>((int)v, 0));
  v = max(v, (unsigned)dpp_or<PDT_DPP_ROW_SHR(4), 0xf, 0xe>((int)v, 0));
  v = max(v, (unsigned)dpp_or<PDT_DPP_ROW_SHR(8), 0xf, 0xc>((int)v, 0));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}
// maximum over the wave of floats that are not NaN (same written-out DPP chain as wave_sum_f)
__device__ __forceinline__ float wave_max_f(float x) {
  float v = x;
  asm volatile(
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_sum_f(float x) {
  // Written out: the compiler folds a masked DPP move into the add only for integer identities
  // (three instructions per step otherwise).  Lanes a step leaves out keep their value; the
  // s_nop covers the VALU-write -> DPP-read hazard the assembler does not track.
  float v = x;
  asm volatile(
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Sorted (descending by value, ascending by index on ties) top-64 of x[0..V): lane i returns
// the i-th largest as pack_key(fkey(x[v]), v); lanes beyond V return 0.  x may live in LDS or
// global memory.  Only the first M <= 64 entries are guaranteed exact (M = how many the
// caller will consume); smaller M prunes harder.
//   1. per-lane maximum over a strided slice; the M-th largest lane maximum tau is a lower
//      bound of the M-th largest element;
//   2. survivors (>= tau) are compacted into `surv` (LDS, capacity PDT_SURV_CAP);
//   3. <= 64 survivors: one bitonic sort.  Otherwise (heavy ties / clustered values) a
//      chunked top-64 merge over the whole vector.
#define PDT_SURV_CAP 64
// LONG: rows of thousands of elements read by a wave that has its SIMD almost to itself --
// eight loads are put in flight before any is used, so the passes run at LDS / L2 throughput
// instead of one round trip per 64 elements (needs the registers of a low-occupancy kernel).
// NONNEG: every x[v] >= +0 (the keys inside the result are then fkey_nonneg keys; callers that
// only read the indices do not care).
// lmax_in: the caller already holds max over its lane's slice of the ordering keys (it had
// the values in registers), so the first pass over x is skipped.
// probe / probe_rank: if given (and V > 64), *probe receives the probe_rank-th largest per-lane
// maximum (as a key) -- a by-product callers use to guess the next row's threshold.
// BIAS: the ranked values are bias + x[v] (one float32 add, as the caller's own arithmetic forms
// them; + 0.0f turns a -0.0 sum into +0.0 so the two zeros tie as in a float compare): elements whose sums round to the same float tie and come out lowest index first, even
// when their x differ.
// SOFT (with BIAS): the ranked values are bias + ((x[v] - soft_max) - soft_lse) -- a row of raw
// scores seen through its log-softmax, the three float32 operations in the order
// log_softmax(-1) followed by the caller's addition performs them.
template <bool LONG = false, bool NONNEG = false, bool BIAS = false, bool SOFT = false>
__device__ __forceinline__ u64 wave_top_sorted_strided(const float *xb, const int64_t sx, int V,
                                                       int M, u64 *surv,
                                                       const unsigned *lmax_in = nullptr,
                                                       unsigned *probe = nullptr,
                                                       int probe_rank = 1, float bias = 0.0f,
                                                       float soft_max = 0.0f, float soft_lse = 0.0f) {
  int lane = lane_id();
  asm volatile("" : "+v"(lane));  // nothing lane-derived is hoisted out of the caller's frame loop
  auto X = [&](int v) {
    if (SOFT) return (bias + ((xb[(int64_t)v * sx] - soft_max) - soft_lse)) + 0.0f;
    return BIAS ? (bias + xb[(int64_t)v * sx]) + 0.0f : xb[(int64_t)v * sx];
  };
  auto fkey = [](float f) { return NONNEG ? fkey_nonneg(f) : pdt::fkey(f); };
  if (V <= PDT_WAVE) {
    const u64 k = lane < V ? pack_key(fkey(X(lane)), (unsigned)lane) : 0ull;
    return wave_sort_desc<u64>(k);
  }
  constexpr int B = LONG ? 8 : 1;  // 64-element chunks per batch of loads
  unsigned lmax = 0u;
  if (lmax_in) {
    lmax = *lmax_in;
  } else {
    int v = lane;
    if constexpr (LONG) {
      for (; v + (B - 1) * PDT_WAVE < V; v += B * PDT_WAVE) {
        float x[B];
#pragma unroll
        for (int i = 0; i < B; ++i) x[i] = X(v + i * PDT_WAVE);
#pragma unroll
        for (int i = 0; i < B; ++i) lmax = max(lmax, fkey(x[i]));
      }
    }
    for (; v < V; v += PDT_WAVE) lmax = max(lmax, fkey(X(v)));
  }
  const unsigned sorted_max = wave_sort_desc<unsigned>(lmax);
  const unsigned tau = (unsigned)__builtin_amdgcn_readlane((int)sorted_max, M - 1);
  if (probe) *probe = (unsigned)__builtin_amdgcn_readlane((int)sorted_max, probe_rank - 1);
  int count = 0;
  for (int v0 = 0; v0 < V; v0 += B * PDT_WAVE) {
    unsigned keys[B];
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int v = v0 + i * PDT_WAVE + lane;
      keys[i] = v < V ? fkey(X(v)) : 0u;  // 0 < tau: never a survivor
    }
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int v = v0 + i * PDT_WAVE + lane;
      const bool pred = keys[i] >= tau;
      const u64 b = __ballot(pred);
      if (b) {
        const int pos = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
        if (pred && pos < PDT_SURV_CAP) surv[pos] = pack_key(keys[i], (unsigned)v);
        count += __popcll(b);
      }
    }
  }
  wave_sync();
  if (count <= PDT_SURV_CAP) return wave_sort_desc<u64>(lane < count ? surv[lane] : 0ull);
  // slow path: too many survivors
  u64 cur = 0ull;
  for (int v0 = 0; v0 < V; v0 += PDT_WAVE) {
    const int v = v0 + lane;
    const unsigned key = v < V ? fkey(X(v)) : 0u;
    const bool pred = v < V && key >= tau;
    if (__ballot(pred)) cur = wave_merge_top64(cur, pred ? pack_key(key, (unsigned)v) : 0ull);
  }
  return cur;
}
// The same selection for a row in global memory of at most 64 * NR elements, read ONCE: the row
// stays in registers (all loads in flight) between the per-lane maxima and the survivor pass.
// Longer rows fall through to wave_top_sorted_strided<LONG>.
template <int NR, bool BIAS = false, bool SOFT = false>
__device__ __forceinline__ u64 wave_top_sorted_regs(const float *xb, const int64_t sx, int V, int M,
                                                    u64 *surv, float bias = 0.0f, float soft_max = 0.0f,
                                                    float soft_lse = 0.0f) {
  if (V > NR * PDT_WAVE || V <= PDT_WAVE)
    return wave_top_sorted_strided<true, false, BIAS, SOFT>(xb, sx, V, M, surv, nullptr, nullptr, 1, bias, soft_max,
                                                            soft_lse);
  int lane = lane_id();
  asm volatile("" : "+v"(lane));
  unsigned keys[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int v = lane + i * PDT_WAVE;
    float x = 0.0f;
    if (i * PDT_WAVE < V && v < V) {
      const float raw = xb[(int64_t)v * sx];
      x = SOFT ? (bias + ((raw - soft_max) - soft_lse)) + 0.0f : (BIAS ? (bias + raw) + 0.0f : raw);
    }
    keys[i] = (i * PDT_WAVE < V && v < V) ? fkey(x) : 0u;  // 0 < every key: never a survivor
  }
  unsigned lmax = 0u;
#pragma unroll
  for (int i = 0; i < NR; ++i) lmax = max(lmax, keys[i]);
  const unsigned sorted_max = wave_sort_desc<unsigned>(lmax);
  const unsigned tau = (unsigned)__builtin_amdgcn_readlane((int)sorted_max, M - 1);
  int count = 0;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    if (i * PDT_WAVE < V) {
      const int v = lane + i * PDT_WAVE;
      const bool pred = keys[i] >= tau && keys[i] != 0u;
      const u64 b = __ballot(pred);
      if (b) {
        const int pos = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
        if (pred && pos < PDT_SURV_CAP) surv[pos] = pack_key(keys[i], (unsigned)v);
        count += __popcll(b);
      }
    }
  }
  wave_sync();
  if (count <= PDT_SURV_CAP) return wave_sort_desc<u64>(lane < count ? surv[lane] : 0ull);
  // too many survivors (heavy ties): the chunked merge of the general form
  return wave_top_sorted_strided<true, false, BIAS, SOFT>(xb, sx, V, M, surv, nullptr, nullptr, 1, bias, soft_max,
                                                          soft_lse);
}

// The same selection over a row a wave already HOLDS as ordering keys: keys[i] belongs to element
// lane + 64 i, 0 where there is none.  Returns false when the survivor buffer overflows (heavy ties): the
// caller falls back to a form that can walk the row again.
template <int NR>
__device__ __forceinline__ bool wave_top_sorted_keys(const unsigned (&keys)[NR], const int M, u64 *surv, u64 &out) {
  int lane = lane_id();
  asm volatile("" : "+v"(lane));
  unsigned lmax = 0u;
#pragma unroll
  for (int i = 0; i < NR; ++i) lmax = max(lmax, keys[i]);
  const unsigned sorted_max = wave_sort_desc<unsigned>(lmax);
  const unsigned tau = (unsigned)__builtin_amdgcn_readlane((int)sorted_max, M - 1);
  int count = 0;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const bool pred = keys[i] >= tau && keys[i] != 0u;
    const u64 b = __ballot(pred);
    if (b) {
      const int pos = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
      if (pred && pos < PDT_SURV_CAP) surv[pos] = pack_key(keys[i], (unsigned)(lane + i * PDT_WAVE));
      count += __popcll(b);
    }
  }
  wave_sync();
  if (count > PDT_SURV_CAP) return false;
  out = wave_sort_desc<u64>(lane < count ? surv[lane] : 0ull);
  return true;
}

template <bool LONG = false, bool NONNEG = false>
__device__ __forceinline__ u64 wave_top_sorted(const float *x, int V, int M, u64 *surv,
                                               const unsigned *lmax_in = nullptr,
                                               unsigned *probe = nullptr, int probe_rank = 1) {
  return wave_top_sorted_strided<LONG, NONNEG>(x, 1, V, M, surv, lmax_in, probe, probe_rank);
}

}  // namespace pdt
