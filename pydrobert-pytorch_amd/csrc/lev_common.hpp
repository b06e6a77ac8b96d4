// Pieces shared by the two Levenshtein kernels: argument block, token staging
// (int64 global -> int32 LDS ids + length detection) and the host-side cost analysis.
#pragma once
#include "pdt_common.hpp"

namespace pdt {

struct LevArgs {
  const int64_t *ref, *hyp;
  int64_t ref_st, ref_sn, hyp_st, hyp_sn;
  int R, H, N;
  int has_eos, include_eos;
  int64_t eos;
  float ins, del, sub, mult;
  int norm, mode, exclude_last, count;
  float padding;
  float *out;
  int64_t out_sh, out_sn;
  int64_t *ref_lens_out, *hyp_lens_out;
  int32_t *status;
  int waves_per_wg, lds_per_wave;  // bytes
  // optimal-completion extras (lev_rowsync.hip)
  uint32_t *bitmask;
  int64_t *class_tokens;
  int32_t *max_count;
  int W;
};

// Geometry of the bit-parallel kernels (lev_bitpar.hip) for one call: lanes per utterance,
// utterances per wave, LDS slices and the offsets of the workspace pieces.  ok = 0: the shape is
// not served (bit-vector sequence longer than 1024 tokens, or lookups that do not fit the LDS).
struct BitparPlan {
  int ok, lgL, upw;
  size_t lds_classify, lds_sub;  // bytes: per wave / per utterance
  size_t lds_tail;               // bytes per workgroup after the utterances' tables (ring of match words, then distances)
  size_t off_lens, off_yh, off_msk, total;  // workspace
};
BitparPlan plan_bitpar(int64_t X, int64_t Y, int64_t N);

// Stage one utterance's tokens tok[t*st + off], t < T, into LDS as int32 and return the
// sequence length (reference _lens_from_eos, _string.py:137-143, plus the include_eos
// fix-up of :198-218).  `fits` tells whether every staged token survives the int64->int32
// narrowing; `missing` whether include_eos was requested but no eos was found.
__device__ __forceinline__ int stage_tokens(const int64_t *tok, int T, int64_t st, int64_t off,
                                            int has_eos, int64_t eos, int include_eos,
                                            int *dst, bool &fits, bool &missing) {
  const int lane = lane_id();
  int len = T;
  bool ok = true;
  for (int t0 = 0; t0 < T; t0 += PDT_WAVE) {
    const int t = t0 + lane;
    const int64_t v = t < T ? tok[(int64_t)t * st + off] : 0;
    ok = ok && ((int64_t)(int32_t)v == v);
    if (t < T) dst[t] = (int32_t)v;
    if (has_eos) {
      const unsigned long long b = __ballot(t < T && v == eos);
      if (b != 0ull) {
        len = t0 + (int)__builtin_ctzll(b);
        break;
      }
    }
  }
  fits = __all(ok) != 0;
  missing = false;
  if (has_eos && include_eos) {
    if (len == T)
      missing = true;
    else
      len += 1;
  }
  return len;
}

// Slow path for token values outside int32: replace every token by the index of its first
// occurrence in ref (hyp tokens absent from ref become -1).  Equality is preserved, which is
// all the DP needs (the reference only ever evaluates ref != hyp, _string.py:291).
__device__ __forceinline__ void remap_tokens_by_first_occurrence(const LevArgs &a, int64_t n,
                                                                 int ref_len, int hyp_len,
                                                                 int *ref_l, int *hyp_l) {
  const int lane = lane_id();
  const int64_t roff = n * a.ref_sn, hoff = n * a.hyp_sn;
  for (int r = lane; r < ref_len; r += PDT_WAVE) {
    const int64_t v = a.ref[(int64_t)r * a.ref_st + roff];
    int id = r;
    for (int k = 0; k < r; ++k)
      if (a.ref[(int64_t)k * a.ref_st + roff] == v) {
        id = k;
        break;
      }
    ref_l[r] = id;
  }
  for (int h = lane; h < hyp_len; h += PDT_WAVE) {
    const int64_t v = a.hyp[(int64_t)h * a.hyp_st + hoff];
    int id = -1;
    for (int k = 0; k < ref_len; ++k)
      if (a.ref[(int64_t)k * a.ref_st + roff] == v) {
        id = k;
        break;
      }
    hyp_l[h] = id;
  }
}

// post-processing of one DP value (reference _string.py:357-378, :394-405)
__device__ __forceinline__ float lev_finish(float v, float mult, int norm, int ref_len,
                                            float if_empty_ref) {
  v = v * mult;
  if (norm) {
    v = __fdiv_rn(v, (float)ref_len);
    if (ref_len == 0) v = if_empty_ref;
  }
  return v;
}

}  // namespace pdt
