// Shared device helpers for the gfx950 kernels (wave64, DPP cross-lane moves).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pdt_amd.h"

#define PDT_WAVE 64
#define PDT_INF __builtin_huge_valf()

// DPP control words (GFX9 encoding; gfx950 keeps the wave_* and row_bcast forms).
#define PDT_DPP_ROW_SHR(n) (0x110 + (n))
#define PDT_DPP_WAVE_SHR1 0x138
#define PDT_DPP_ROW_BCAST15 0x142
#define PDT_DPP_ROW_BCAST31 0x143

namespace pdt {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// value of lane-1 (lane 0 receives `first`).  One v_mov_b32_dpp wave_shr:1.
__device__ __forceinline__ float shr1(float v, float first) {
  return __int_as_float(__builtin_amdgcn_update_dpp(
      __float_as_int(first), __float_as_int(v), PDT_DPP_WAVE_SHR1, 0xf, 0xf, false));
}
__device__ __forceinline__ int shr1(int v, int first) {
  return __builtin_amdgcn_update_dpp(first, v, PDT_DPP_WAVE_SHR1, 0xf, 0xf, false);
}

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_or(float v, float ident) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v),
                                                    CTRL, ROW_MASK, BANK_MASK, false));
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_or(int v, int ident) {
  return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, BANK_MASK, false);
}

// inclusive wave-wide min scan of floats that are not NaN: six v_min_f32_dpp.  Written out --
// left to the compiler every step is a move of the identity, a DPP move, a canonicalising
// v_max and the v_min (the row loops that scan once per DP row are VALU-bound).  A lane a step
// does not reach (no source lane, or masked out) keeps its value; the s_nop covers the
// VALU-write -> DPP-read hazard the assembler does not track.
__device__ __forceinline__ float wave_incl_scan_min(float x) {
  float v = x;
  asm volatile(
      "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xe\n\t"
      "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}

// inclusive wave-wide integer add scan
__device__ __forceinline__ int wave_incl_scan_add(int x) {
  int v = x;
  v += dpp_or<PDT_DPP_ROW_SHR(1)>(x, 0);
  v += dpp_or<PDT_DPP_ROW_SHR(2)>(x, 0);
  v += dpp_or<PDT_DPP_ROW_SHR(3)>(x, 0);
  v += dpp_or<PDT_DPP_ROW_SHR(4), 0xf, 0xe>(v, 0);
  v += dpp_or<PDT_DPP_ROW_SHR(8), 0xf, 0xc>(v, 0);
  v += dpp_or<PDT_DPP_ROW_BCAST15, 0xa>(v, 0);
  v += dpp_or<PDT_DPP_ROW_BCAST31, 0xc>(v, 0);
  return v;
}

__device__ __forceinline__ float wave_min(float x) {
  const float s = wave_incl_scan_min(x);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 63));
}
// sum over lanes 0..15 only (one DPP row): 4 steps
__device__ __forceinline__ int row0_sum(int x) {
  int v = x;
  v += dpp_or<PDT_DPP_ROW_SHR(1)>(v, 0);
  v += dpp_or<PDT_DPP_ROW_SHR(2)>(v, 0);
  v += dpp_or<PDT_DPP_ROW_SHR(4), 0xf, 0xe>(v, 0);
  v += dpp_or<PDT_DPP_ROW_SHR(8), 0xf, 0xc>(v, 0);
  return __builtin_amdgcn_readlane(v, 15);
}
__device__ __forceinline__ int wave_sum(int x) {
  return __builtin_amdgcn_readlane(wave_incl_scan_add(x), 63);
}

// order LDS traffic between lanes of ONE wave (no workgroup barrier: waves are independent)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup -> work-item remap so that the workgroups that land on one XCD (blockIdx % 8
// under round-robin dispatch) own one contiguous range of items and share L2 lines.
// Bijective for any grid size (cdna_hip_programming.md T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = b & 7u, idx = b >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
  return base + idx;
}

}  // namespace pdt
