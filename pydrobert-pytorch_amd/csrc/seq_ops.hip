// Two "next" decoding operators on the same data layouts as the
// searches:
//   pdt_ctc_greedy_search   -- ctc_greedy_search (reference _decoding.py:507-558): per frame
//       log-softmax max + argmax, then blank removal / repeat collapsing / compaction, one wave
//       per utterance, ONE pass over the logits;
//   pdt_sequence_log_probs_{forward,backward} -- sequence_log_probs on tensors
//       (_decoding.py:1516-1551): log_softmax + gather + masked sum over the step axis, fused;
//       the backward writes softmax-minus-onehot rows scaled by the upstream gradient.
// Both are HBM-bound on the logits (4 * V bytes per frame).
#include "wave_select.hpp"

namespace pdt {

struct GreedyArgs {
  const float *logits; int64_t lg_st, lg_sn, lg_sv;  // (T, N, V) through element strides
  const int64_t *in_lens;   // (N,) or null
  int T, N, V, blank, is_probs;
  float *max_out;           // (N,)
  int64_t *paths;           // (T, N) through strides
  int64_t pa_st, pa_sn;
  int64_t *out_lens;        // (N,)
  int lds_per_wave;
};

__global__ void __launch_bounds__(256) ctc_greedy_kernel(const GreedyArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int64_t n = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (n >= a.N) return;
  int *arg = reinterpret_cast<int *>(smem + (size_t)wave * a.lds_per_wave);  // [T]
  const int T = a.T, V = a.V;
  const int in_len = a.in_lens ? (int)min((int64_t)T, max((int64_t)0, a.in_lens[n])) : T;
  float total = a.is_probs ? 1.0f : 0.0f;  // lane-uniform
  for (int t = 0; t < T; ++t) {
    const float *x = a.logits + (int64_t)t * a.lg_st + n * a.lg_sn;
    // running (value, index) maximum with lowest-index ties, and sum-exp for the normaliser
    u64 best = 0ull;
    float mx = -PDT_INF;
    for (int v = lane; v < V; v += PDT_WAVE) {
      const float xv = x[(int64_t)v * a.lg_sv];
      const u64 k = pack_key(fkey(xv), (unsigned)v);
      best = k > best ? k : best;
      mx = fmaxf(mx, xv);
    }
    // wave arg-max of the packed keys
    {
      unsigned hi = (unsigned)(best >> 32);
      const unsigned hmax = wave_max_u32(hi);
      const unsigned lo = hi == hmax ? (unsigned)best : 0u;
      const unsigned lmax = wave_max_u32(lo);
      best = ((u64)hmax << 32) | lmax;
    }
    const int am = (int)idx_of(best);
    const float xmax = fkey_inv(key_of(best));
    if (lane == 0) arg[t] = am;
    if (t < in_len) {  // frames beyond the length contribute 0 (log) or 1 (prob): :540-543
      if (a.is_probs) {
        total *= xmax;
      } else {
        float s = 0.0f;
        for (int v = lane; v < V; v += PDT_WAVE) s += expf(x[(int64_t)v * a.lg_sv] - xmax);
        s = wave_sum_f(s);
        total += -logf(s);  // log_softmax(x)[argmax] = -log sum exp(x - max)
      }
    }
  }
  wave_sync();
  // keep mask, compaction (:531-552): out[j] = j-th kept token; positions >= out_len keep
  // the raw arg-max, as masked_scatter_ leaves them in the reference
  int count = 0;
  for (int t0 = 0; t0 < T; t0 += PDT_WAVE) {
    const int t = t0 + lane;
    bool keep = false;
    int tok = 0;
    if (t < T) {
      tok = arg[t];
      keep = tok != a.blank && (t == 0 || tok != arg[t - 1]) && t < in_len;
    }
    const u64 b = __ballot(keep);
    const int pos = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
    if (keep) a.paths[(int64_t)pos * a.pa_st + n * a.pa_sn] = tok;
    count += __popcll(b);
  }
  for (int t = count + lane; t < T; t += PDT_WAVE) a.paths[(int64_t)t * a.pa_st + n * a.pa_sn] = arg[t];
  if (lane == 0) {
    a.max_out[n] = total;
    a.out_lens[n] = count;
  }
}

// ---- sequence_log_probs -------------------------------------------------------------------
// hyp viewed as (A, S, B): A = dims before the step axis, S = steps, B = dims after it.
struct SlpArgs {
  const float *logits;      // hyp.shape + (V,), contiguous
  const int64_t *hyp;       // (A, S, B) contiguous
  int A, S, B, V;
  int has_eos;
  int64_t eos;
  float *out;               // (A, B)
  const float *grad_out;    // backward (A, B)
  float *grad_logits;       // backward, same layout as logits
};

// length of sequence (a, b) = index of first eos + 1, else S (:1533-1546)
__device__ __forceinline__ int slp_len(const SlpArgs &a, int ai, int bi) {
  if (!a.has_eos) return a.S;
  for (int s = 0; s < a.S; ++s)
    if (a.hyp[((int64_t)ai * a.S + s) * a.B + bi] == a.eos) return s + 1;
  return a.S + 1;  // no eos: _lens_from_eos gives S, plus one -> nothing masked
}

template <bool BACKWARD>
__global__ void __launch_bounds__(256) slp_kernel(const SlpArgs a) {
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;  // row = (ai * S + s) * B + bi
  const int64_t rows = (int64_t)a.A * a.S * a.B;
  if (row >= rows) return;
  const int bi = (int)(row % a.B);
  const int s = (int)((row / a.B) % a.S);
  const int ai = (int)(row / ((int64_t)a.B * a.S));
  const int V = a.V;
  const int64_t tok = a.hyp[row];
  bool masked = tok < 0 || tok >= V;
  if (a.has_eos && !masked) masked = s >= slp_len(a, ai, bi);
  float *go = BACKWARD ? a.grad_logits + row * (int64_t)V : nullptr;
  if (masked) {
    if (BACKWARD)
      for (int v = lane; v < V; v += PDT_WAVE) go[v] = 0.0f;
    return;
  }
  const float *x = a.logits + row * (int64_t)V;
  float mx = -PDT_INF;
  for (int v = lane; v < V; v += PDT_WAVE) mx = fmaxf(mx, x[v]);
  mx = wave_max_f(mx);
  float sum = 0.0f;
  for (int v = lane; v < V; v += PDT_WAVE) sum += expf(x[v] - mx);
  sum = wave_sum_f(sum);
  const float lse = mx + logf(sum);
  if (!BACKWARD) {
    if (lane == 0) atomicAdd(&a.out[(int64_t)ai * a.B + bi], x[tok] - lse);
    return;
  }
  const float g = a.grad_out[(int64_t)ai * a.B + bi];
  for (int v = lane; v < V; v += PDT_WAVE) go[v] = g * ((v == tok ? 1.0f : 0.0f) - expf(x[v] - lse));
}

// deterministic forward: one wave per output element sums its S steps in order
__global__ void __launch_bounds__(256) slp_forward_kernel(const SlpArgs a) {
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int64_t o = (int64_t)blockIdx.x * 4 + wave;  // o = ai * B + bi
  if (o >= (int64_t)a.A * a.B) return;
  const int ai = (int)(o / a.B), bi = (int)(o % a.B);
  const int V = a.V;
  const int len = slp_len(a, ai, bi);
  float acc = 0.0f;
  for (int s = 0; s < a.S; ++s) {
    const int64_t row = ((int64_t)ai * a.S + s) * a.B + bi;
    const int64_t tok = a.hyp[row];
    if (tok < 0 || tok >= V || s >= len) continue;
    const float *x = a.logits + row * (int64_t)V;
    float mx = -PDT_INF;
    for (int v = lane; v < V; v += PDT_WAVE) mx = fmaxf(mx, x[v]);
    mx = wave_max_f(mx);
    float sum = 0.0f;
    for (int v = lane; v < V; v += PDT_WAVE) sum += expf(x[v] - mx);
    sum = wave_sum_f(sum);
    acc += (x[tok] - mx) - logf(sum);
  }
  if (lane == 0) a.out[o] = acc;
}

}  // namespace pdt

extern "C" {

int pdt_ctc_greedy_search(const float *logits, int64_t T, int64_t N, int64_t V, int64_t lg_st,
                          int64_t lg_sn, int64_t lg_sv, const int64_t *in_lens, int64_t blank_idx,
                          int is_probs, float *max_out, int64_t *paths, int64_t pa_st,
                          int64_t pa_sn, int64_t *out_lens, void *stream) {
  using namespace pdt;
  if (T < 0 || N < 0 || V < 1 || blank_idx < 0 || blank_idx >= V) return PDT_E_ARG;
  if (N == 0) return PDT_OK;
  if ((T > 0 && (!logits || !paths)) || !max_out || !out_lens) return PDT_E_ARG;
  const size_t per_wave = ((size_t)(T > 0 ? T : 1) * 4 + 15) & ~(size_t)15;
  if (per_wave * 4 > 160 * 1024) return PDT_E_TOO_LONG;
  GreedyArgs a{};
  a.logits = logits; a.lg_st = lg_st; a.lg_sn = lg_sn; a.lg_sv = lg_sv;
  a.in_lens = in_lens; a.T = (int)T; a.N = (int)N; a.V = (int)V; a.blank = (int)blank_idx;
  a.is_probs = is_probs; a.max_out = max_out; a.paths = paths; a.pa_st = pa_st; a.pa_sn = pa_sn;
  a.out_lens = out_lens; a.lds_per_wave = (int)per_wave;
  const size_t smem = per_wave * 4;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_greedy_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(ctc_greedy_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), smem,
                     (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int pdt_sequence_log_probs_forward(const float *logits, const int64_t *hyp, int64_t A, int64_t S,
                                   int64_t B, int64_t V, int has_eos, int64_t eos, float *out,
                                   void *stream) {
  using namespace pdt;
  if (A < 0 || S < 0 || B < 0 || V < 1) return PDT_E_ARG;
  if (A * B == 0) return PDT_OK;
  if ((S > 0 && (!logits || !hyp)) || !out) return PDT_E_ARG;
  if (A * S * B >= (1ll << 31) * 4) return PDT_E_TOO_LONG;
  SlpArgs a{};
  a.logits = logits; a.hyp = hyp; a.A = (int)A; a.S = (int)S; a.B = (int)B; a.V = (int)V;
  a.has_eos = has_eos; a.eos = eos; a.out = out;
  hipLaunchKernelGGL(slp_forward_kernel, dim3((unsigned)((A * B + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int pdt_sequence_log_probs_backward(const float *logits, const int64_t *hyp, int64_t A, int64_t S,
                                    int64_t B, int64_t V, int has_eos, int64_t eos,
                                    const float *grad_out, float *grad_logits, void *stream) {
  using namespace pdt;
  if (A < 0 || S < 0 || B < 0 || V < 1) return PDT_E_ARG;
  if (A * S * B == 0) return PDT_OK;
  if (!logits || !hyp || !grad_out || !grad_logits) return PDT_E_ARG;
  if (A * S * B >= (1ll << 31) * 4) return PDT_E_TOO_LONG;
  SlpArgs a{};
  a.logits = logits; a.hyp = hyp; a.A = (int)A; a.S = (int)S; a.B = (int)B; a.V = (int)V;
  a.has_eos = has_eos; a.eos = eos; a.grad_out = grad_out; a.grad_logits = grad_logits;
  hipLaunchKernelGGL(slp_kernel<true>, dim3((unsigned)((A * S * B + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // extern "C"
