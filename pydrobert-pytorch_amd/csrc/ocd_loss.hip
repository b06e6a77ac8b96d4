// Hard optimal-completion distillation loss, fused (reference _string.py:1188-1251).
//
// The reference expands the logits to (H, N, C, V), builds the (H, N, C) int64 target tensor
// with optimal_completion and calls cross_entropy on H*N*C rows.  Here the completion sets stay
// in their compact form -- the per-prefix CLASS BITMASKS produced by pdt_oc_mask plus the
// per-utterance class-token table -- and one wave per (h, n) row computes
//     loss[h, n] = 1/|S| * sum_{t in S} w_t * (logsumexp(logits[h, n, :]) - logits[h, n, t])
// reading the logits row once.  Backward: d loss / d logits[v] =
//     g * (Wsum / |S| * softmax[v] - [v in S] * w_v / |S|),   Wsum = sum_{t in S} w_t,
// with set membership kept as a V-bit map in LDS.  Both passes are HBM-bound on the logits.
#include "row_reduce.hpp"

namespace pdt {

struct OcdArgs {
  const float *logits; int64_t lg_sh, lg_sn, lg_sv;  // (H, N, V) through element strides
  const uint32_t *bitmask;    // (H, N, W)
  const int64_t *class_tokens;  // (N, R)
  const float *weight;        // (V,) or null
  int H, N, V, R, W;
  int64_t ignore_index;
  float *loss;                // (H, N)
  int *count;                 // (H, N) number of targets that are not ignore_index
  const float *grad_loss;     // backward: (H, N)
  float *grad_logits;         // backward: (H, N, V) contiguous
  int *status;                // bit 0: a target outside [0, V)
};

// Visits the tokens of the completion set of row (h, n): f(token) for every set bit -- four table
// look-ups in flight at a time (one at a time each is a round trip to L2 / HBM, and a lane of a
// 13-token set has several).
// `w`: this lane's word of the row's class bitmask (loaded by the caller together with the logits);
// rows wider than 64 words (references beyond 2048 tokens) take the remaining words 64 at a time.
template <typename F>
__device__ __forceinline__ void for_each_target(const OcdArgs &a, unsigned w, int64_t row, int64_t n, F &&f) {
  const int lane = lane_id();
  for (int w0 = 0; w0 < a.W; w0 += PDT_WAVE) {
    if (w0 > 0) w = w0 + lane < a.W ? a.bitmask[row * a.W + w0 + lane] : 0u;
    const int64_t *tab = a.class_tokens + n * (int64_t)a.R + (int64_t)(w0 + lane) * 32;
    while (w) {
      int64_t tok[4];
      int cnt = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (w) {
          const int b = __builtin_ctz(w);
          w &= w - 1u;
          tok[q] = tab[b];
          cnt = q + 1;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < cnt) f(tok[q]);
    }
  }
}

template <bool BACKWARD, int NR>
__global__ void __launch_bounds__(256) ocd_loss_kernel(const OcdArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;  // row = h * N + n
  if (row >= (int64_t)a.H * a.N) return;
  const int64_t h = row / a.N, n = row - h * a.N;
  const int V = a.V;
  const float *x = a.logits + h * a.lg_sh + n * a.lg_sn;
  const unsigned my_word = lane < a.W ? a.bitmask[row * a.W + lane] : 0u;  // (in flight with the row)
  // log-sum-exp of the row: read once into registers when it has at most 64 NR elements
  float r[NR];
  const RowStats st = row_stats<false, true, NR>(x, a.lg_sv, V, r);
  const float lse = st.mx + logf(st.sum);
  // the row again in LDS (rows that fit the registers): x[token] of the targets without a global load
  const bool staged = V <= NR * PDT_WAVE;
  float *xl = reinterpret_cast<float *>(smem) + (size_t)wave * (NR * PDT_WAVE);
  if (staged) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (i * PDT_WAVE < V) xl[lane + i * PDT_WAVE] = r[i];
    wave_sync();
  }
  unsigned *member = reinterpret_cast<unsigned *>(smem + (size_t)4 * NR * PDT_WAVE * 4) + (size_t)wave * ((V + 31) / 32);
  if (BACKWARD) {
    for (int i = lane; i < (V + 31) / 32; i += PDT_WAVE) member[i] = 0u;
    wave_sync();
  }
  float acc = 0.0f, wsum = 0.0f;
  int cnt = 0;
  bool bad = false;
  for_each_target(a, my_word, row, n, [&](int64_t tok) {
    if (tok == a.ignore_index) return;
    if (tok < 0 || tok >= V) {
      bad = true;
      return;
    }
    const float w = a.weight ? a.weight[tok] : 1.0f;
    acc += w * (lse - (staged ? xl[tok] : x[tok * a.lg_sv]));
    wsum += w;
    ++cnt;
    if (BACKWARD) atomicOr(&member[tok >> 5], 1u << (tok & 31));
  });
  acc = wave_sum_f(acc);
  wsum = wave_sum_f(wsum);
  cnt = wave_sum(cnt);
  if (__ballot(bad) && lane == 0 && a.status) atomicOr(a.status, 1);
  const float denom = (float)(cnt > 0 ? cnt : 1);  // clamp_min(1), :1241
  if (!BACKWARD) {
    if (lane == 0) {
      a.loss[row] = acc / denom;
      a.count[row] = cnt;
    }
    return;
  }
  wave_sync();
  const float g = a.grad_loss[row] / denom;
  float *go = a.grad_logits + row * (int64_t)V;
  auto grad_of = [&](const int v, const float xv) {
    float gv = g * wsum * expf(xv - lse);
    if ((member[v >> 5] >> (v & 31)) & 1u) gv -= g * (a.weight ? a.weight[v] : 1.0f);
    go[v] = gv;
  };
  if (V <= NR * PDT_WAVE) {
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (i * PDT_WAVE < V && lane + i * PDT_WAVE < V) grad_of(lane + i * PDT_WAVE, r[i]);
  } else {
    for (int v = lane; v < V; v += PDT_WAVE) grad_of(v, x[(int64_t)v * a.lg_sv]);
  }
}

}  // namespace pdt

extern "C" {

static int ocd_launch(pdt::OcdArgs &a, bool backward, void *stream) {
  using namespace pdt;
  const int64_t rows = (int64_t)a.H * a.N;
  const bool small = a.V <= 8 * PDT_WAVE;
  // staged rows of the four waves (64 NR floats each), then the membership maps of the backward pass
  const size_t smem = (size_t)4 * (small ? 8 : 16) * PDT_WAVE * 4 + (backward ? (size_t)4 * ((a.V + 31) / 32) * 4 : 0);
  if (smem > 64 * 1024) return PDT_E_TOO_LONG;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  auto kern = backward ? (small ? ocd_loss_kernel<true, 8> : ocd_loss_kernel<true, 16>)
                       : (small ? ocd_loss_kernel<false, 8> : ocd_loss_kernel<false, 16>);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int pdt_ocd_loss_forward(const float *logits, int64_t H, int64_t N, int64_t V, int64_t lg_sh,
                         int64_t lg_sn, int64_t lg_sv, const uint32_t *bitmask,
                         const int64_t *class_tokens, int64_t R, const float *weight,
                         int64_t ignore_index, float *loss, int32_t *count, int32_t *status,
                         void *stream) {
  using namespace pdt;
  if (H < 0 || N < 0 || V < 1 || R < 0) return PDT_E_ARG;
  if (H == 0 || N == 0) return PDT_OK;
  if (!logits || !bitmask || !class_tokens || !loss || !count) return PDT_E_ARG;
  if (H * N >= (1ll << 31) * 4) return PDT_E_TOO_LONG;
  OcdArgs a{};
  a.logits = logits; a.lg_sh = lg_sh; a.lg_sn = lg_sn; a.lg_sv = lg_sv;
  a.bitmask = bitmask; a.class_tokens = class_tokens; a.weight = weight;
  a.H = (int)H; a.N = (int)N; a.V = (int)V; a.R = (int)R; a.W = (int)pdt_oc_mask_words(R);
  a.ignore_index = ignore_index; a.loss = loss; a.count = count; a.status = status;
  return ocd_launch(a, false, stream);
}

int pdt_ocd_loss_backward(const float *logits, int64_t H, int64_t N, int64_t V, int64_t lg_sh,
                          int64_t lg_sn, int64_t lg_sv, const uint32_t *bitmask,
                          const int64_t *class_tokens, int64_t R, const float *weight,
                          int64_t ignore_index, const float *grad_loss, float *grad_logits,
                          void *stream) {
  using namespace pdt;
  if (H < 0 || N < 0 || V < 1 || R < 0) return PDT_E_ARG;
  if (H == 0 || N == 0) return PDT_OK;
  if (!logits || !bitmask || !class_tokens || !grad_loss || !grad_logits) return PDT_E_ARG;
  if (H * N >= (1ll << 31) * 4) return PDT_E_TOO_LONG;
  OcdArgs a{};
  a.logits = logits; a.lg_sh = lg_sh; a.lg_sn = lg_sn; a.lg_sv = lg_sv;
  a.bitmask = bitmask; a.class_tokens = class_tokens; a.weight = weight;
  a.H = (int)H; a.N = (int)N; a.V = (int)V; a.R = (int)R; a.W = (int)pdt_oc_mask_words(R);
  a.ignore_index = ignore_index; a.grad_loss = grad_loss; a.grad_logits = grad_logits;
  return ocd_launch(a, true, stream);
}

}  // extern "C"
