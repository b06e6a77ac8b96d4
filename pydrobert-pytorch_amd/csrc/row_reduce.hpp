// One row of logits through a wave: maximum, arg-max, sum of exponentials -- with the row read
// ONCE, every load in flight, into registers (seq_ops.hip, ocd_loss.hip).
#pragma once
#include "wave_select.hpp"

namespace pdt {

// Rows of up to 64 * NR elements are held in registers (one pass over memory); NR = 8 keeps the
// kernels at eight waves per SIMD for the common vocabularies, 16 serves rows up to 1024.
// One row x[0..V) (element stride sv) through a wave.  Returns the wave-wide maximum and, if asked,
// the packed (value, lowest index) arg-max key and sum_v exp(x[v] - max).  Rows beyond 64 * NR
// elements are streamed twice, eight loads in flight.
struct RowStats {
  float mx, sum;
  u64 best;
};
template <int NR>
__device__ __forceinline__ void row_load(const float *x, const int64_t sv, const int V, float (&r)[NR]) {
  const int lane = lane_id();
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int v = lane + i * PDT_WAVE;
    r[i] = (i * PDT_WAVE < V && v < V) ? x[(int64_t)v * sv] : -PDT_INF;
  }
}
template <bool WANT_ARG, bool WANT_SUM, int NR>
__device__ __forceinline__ RowStats row_stats(const float *x, const int64_t sv, const int V, float (&r)[NR]) {
  const int lane = lane_id();
  RowStats st;
  st.sum = 0.0f;
  st.best = 0ull;
  float mx = -PDT_INF;
  u64 best = 0ull;
  const bool in_regs = V <= NR * PDT_WAVE;
  if (in_regs) {
    row_load<NR>(x, sv, V, r);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (i * PDT_WAVE < V) {
        mx = fmaxf(mx, r[i]);
        if (WANT_ARG && lane + i * PDT_WAVE < V) {
          const u64 k = pack_key(fkey(r[i]), (unsigned)(lane + i * PDT_WAVE));
          best = k > best ? k : best;
        }
      }
    }
  } else {
    for (int v0 = 0; v0 < V; v0 += 8 * PDT_WAVE) {
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = v0 + i * PDT_WAVE + lane < V ? x[(int64_t)(v0 + i * PDT_WAVE + lane) * sv] : -PDT_INF;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        mx = fmaxf(mx, t[i]);
        if (WANT_ARG && v0 + i * PDT_WAVE + lane < V) {
          const u64 k = pack_key(fkey(t[i]), (unsigned)(v0 + i * PDT_WAVE + lane));
          best = k > best ? k : best;
        }
      }
    }
  }
  if (WANT_ARG) {  // wave arg-max of the packed keys (highest value, lowest index)
    const unsigned hi = (unsigned)(best >> 32);
    const unsigned hmax = wave_max_u32(hi);
    const unsigned lo = hi == hmax ? (unsigned)best : 0u;
    const unsigned lmax = wave_max_u32(lo);
    st.best = ((u64)hmax << 32) | lmax;
    st.mx = fkey_inv(hmax);
  } else {
    st.mx = wave_max_f(mx);
  }
  if (WANT_SUM) {
    float s = 0.0f;
    if (in_regs) {
#pragma unroll
      for (int i = 0; i < NR; ++i)
        if (i * PDT_WAVE < V) s += expf(r[i] - st.mx);  // (exp(-inf) = 0 beyond V)
    } else {
      for (int v0 = 0; v0 < V; v0 += 8 * PDT_WAVE) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = v0 + i * PDT_WAVE + lane < V ? x[(int64_t)(v0 + i * PDT_WAVE + lane) * sv] : -PDT_INF;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += expf(t[i] - st.mx);
      }
    }
    st.sum = wave_sum_f(s);
  }
  return st;
}

}  // namespace pdt
