// The per-frame extension probabilities of CTCPrefixSearch with a language model (reference
// _decoding.py:1110-1135), fused: the reference (and this package's differentiable path) forms them
// with a log_softmax / softmax over the LM scores, a scale, an exp and one or two broadcast
// multiplies -- four passes over (N, K', V) per frame, 87 us of the 208 a frame took at
// N = 1024, K' = 16, V = 1000.  Here one wave reads a row of LM scores once (registers), reduces it
// and writes the mixed row:
//   shallow fusion   ext[n, k, v] = p_ctc[n, v] * exp(beta * log_softmax(lm[n, k])[v])    (:1130-1135)
//   valid mixture    ext[n, k, v] = (1 - beta) * p_ctc[n, v]
//                                   + beta * softmax(lm[n, k])[v] * (1 - p_blank[n])        (:1120-1128)
// HBM-bound: 4 V bytes in, 4 V out per row.
#include "row_reduce.hpp"

namespace pdt {

struct FusionArgs {
  const float *lm;      // (rows, V) contiguous, rows = N * Kp
  const float *nonext;  // (N, V) through element strides
  int64_t ne_sn, ne_sv;
  const float *blank;   // (N,) through an element stride
  int64_t bl_sn;
  int64_t rows;
  int Kp, V, valid_mixture;
  float beta;
  float *out;           // (rows, V) contiguous
};

template <int NR>
__global__ void __launch_bounds__(256) fusion_ext_kernel(const FusionArgs a) {
  const int lane = lane_id();
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const int64_t n = row / a.Kp;
  const int V = a.V;
  const float *x = a.lm + row * (int64_t)V;
  const float *pc = a.nonext + n * a.ne_sn;
  float *o = a.out + row * (int64_t)V;
  float r[NR];
  const RowStats st = row_stats<false, true, NR>(x, 1, V, r);
  const float log_sum = logf(st.sum), keep = 1.0f - a.beta;
  const float scale = a.valid_mixture ? 1.0f - a.blank[n * a.bl_sn] : 0.0f;
  auto mix = [&](const int v, const float xv) {
    const float p = pc[(int64_t)v * a.ne_sv];
    if (a.valid_mixture) {
      const float lm_p = (expf(xv - st.mx) / st.sum) * scale;
      o[v] = keep * p + a.beta * lm_p;
    } else {
      o[v] = p * expf(a.beta * ((xv - st.mx) - log_sum));
    }
  };
  if (V <= NR * PDT_WAVE) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (i * PDT_WAVE < V && lane + i * PDT_WAVE < V) mix(lane + i * PDT_WAVE, r[i]);
  } else {
    for (int v = lane; v < V; v += PDT_WAVE) mix(v, x[v]);
  }
}

// The language-model factor of the mix alone -- what multiplies (shallow fusion) or is blended with
// (valid mixture) the frame's probabilities: F[v] = exp(beta * log_softmax(lm)[v]), or softmax(lm)[v].
// The same expressions on the same row statistics as fusion_ext_kernel, so
// fusion: p * F[v], valid mixture: keep * p + beta * (G[v] * (1 - p_blank)) reproduce its output to the
// bit.  For a bigram model the factor depends on the context token only: the search of
// ctc_lm_table.hip reads rows of this table instead of scoring the model in every frame.
template <int NR>
__global__ void __launch_bounds__(256) lm_factor_kernel(const float *lm, int64_t rows, int V, float beta, int valid_mixture,
                                                        float *out, int64_t out_stride) {
  const int lane = lane_id();
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *x = lm + row * (int64_t)V;
  float *o = out + row * out_stride;
  float r[NR];
  const RowStats st = row_stats<false, true, NR>(x, 1, V, r);
  const float log_sum = logf(st.sum);
  auto fac = [&](const int v, const float xv) {
    o[v] = valid_mixture ? expf(xv - st.mx) / st.sum : expf(beta * ((xv - st.mx) - log_sum));
  };
  if (V <= NR * PDT_WAVE) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (i * PDT_WAVE < V && lane + i * PDT_WAVE < V) fac(lane + i * PDT_WAVE, r[i]);
  } else {
    for (int v = lane; v < V; v += PDT_WAVE) fac(v, x[v]);
  }
}

}  // namespace pdt

extern "C" int pdt_lm_factor_table(const float *lm_log_probs, int64_t rows, int64_t V, float beta, int valid_mixture,
                                   float *out, int64_t out_stride, void *stream) {
  using namespace pdt;
  if (rows < 0 || V < 1 || out_stride < V) return PDT_E_ARG;
  if (rows == 0) return PDT_OK;
  if (!lm_log_probs || !out) return PDT_E_ARG;
  if (rows >= (1ll << 31) * 4 || V >= (1ll << 31)) return PDT_E_TOO_LONG;
  auto kern = V <= 8 * PDT_WAVE ? lm_factor_kernel<8> : lm_factor_kernel<16>;
  hipLaunchKernelGGL(kern, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, lm_log_probs, rows,
                     (int)V, beta, valid_mixture, out, out_stride);
  return (int)hipGetLastError();
}

extern "C" int pdt_fusion_ext(const float *lm_log_probs, int64_t N, int64_t Kp, int64_t V,
                              const float *nonext, int64_t ne_sn, int64_t ne_sv, const float *blank,
                              int64_t bl_sn, float beta, int valid_mixture, float *out, void *stream) {
  using namespace pdt;
  if (N < 0 || Kp < 1 || V < 1) return PDT_E_ARG;
  if (N == 0) return PDT_OK;
  if (!lm_log_probs || !nonext || !out || (valid_mixture && !blank)) return PDT_E_ARG;
  const int64_t rows = N * Kp;
  if (rows >= (1ll << 31) * 4 || V >= (1ll << 31)) return PDT_E_TOO_LONG;
  FusionArgs a{lm_log_probs, nonext, ne_sn, ne_sv, blank, bl_sn, rows, (int)Kp, (int)V, valid_mixture, beta, out};
  auto kern = V <= 8 * PDT_WAVE ? fusion_ext_kernel<8> : fusion_ext_kernel<16>;
  hipLaunchKernelGGL(kern, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
