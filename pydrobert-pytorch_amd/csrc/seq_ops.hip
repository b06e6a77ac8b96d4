// Two "next" decoding operators on the same data layouts as the
// searches:
//   pdt_ctc_greedy_search   -- ctc_greedy_search (reference _decoding.py:507-558): per frame
//       log-softmax max + argmax, then blank removal / repeat collapsing / compaction, one wave
//       per utterance, ONE pass over the logits;
//   pdt_sequence_log_probs_{forward,backward} -- sequence_log_probs on tensors
//       (_decoding.py:1516-1551): log_softmax + gather + masked sum over the step axis, fused;
//       the backward writes softmax-minus-onehot rows scaled by the upstream gradient.
// Both are HBM-bound on the logits (4 * V bytes per frame) -- once every row is read with all its
// loads in flight and the rows of one sequence are spread over the waves of a workgroup: frames
// are independent, only the reductions over them are ordered.  (Round 1 walked the frames of a
// sequence with one wave, one 64-element load at a time, and sequence_log_probs re-derived the
// sequence length with a serial scan in every row's wave: 2.4 / 2.2 / 33 ms at N = 4096, T = 512,
// V = 257, where the logits are 0.4 ms of HBM time.)
#include "row_reduce.hpp"

namespace pdt {

struct GreedyArgs {
  const float *logits; int64_t lg_st, lg_sn, lg_sv;  // (T, N, V) through element strides
  const int64_t *in_lens;   // (N,) or null
  int T, N, V, blank, is_probs;
  float *max_out;           // (N,)
  int64_t *paths;           // (T, N) through strides
  int64_t pa_st, pa_sn;
  int64_t *out_lens;        // (N,)
  int nw;                   // waves per utterance
};

// One workgroup per utterance; its waves take the frames t = w, w + NW, ...
template <int NR>
__global__ void __launch_bounds__(1024) ctc_greedy_kernel(const GreedyArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6), NW = a.nw;
  const int64_t n = blockIdx.x;
  int *arg = reinterpret_cast<int *>(smem);                   // [T]
  float *val = reinterpret_cast<float *>(arg + (a.T > 0 ? a.T : 1));  // [T] per-frame log-prob / prob of the arg-max
  const int T = a.T, V = a.V;
  const int in_len = a.in_lens ? (int)min((int64_t)T, max((int64_t)0, a.in_lens[n])) : T;
  for (int t = wave; t < T; t += NW) {
    const float *x = a.logits + (int64_t)t * a.lg_st + n * a.lg_sn;
    float r[NR];
    // frames beyond the length contribute 0 (log) or 1 (prob), :540-543 -- but their arg-max is
    // still reported in the tail of `paths`
    RowStats st;
    if (t < in_len && !a.is_probs)
      st = row_stats<true, true, NR>(x, a.lg_sv, V, r);
    else
      st = row_stats<true, false, NR>(x, a.lg_sv, V, r);
    if (lane == 0) {
      arg[t] = (int)idx_of(st.best);
      // log_softmax(x)[argmax] = -log sum exp(x - max)
      val[t] = t < in_len ? (a.is_probs ? st.mx : -logf(st.sum)) : (a.is_probs ? 1.0f : 0.0f);
    }
  }
  __syncthreads();
  if (wave != 0) return;
  // the joint (log-)probability: frames in order, 64 partial accumulators combined in lane order
  float total = a.is_probs ? 1.0f : 0.0f;
  for (int t = lane; t < T; t += PDT_WAVE) total = a.is_probs ? total * val[t] : total + val[t];
  for (int off = 1; off < PDT_WAVE; off <<= 1) {
    const float o = __shfl_xor(total, off);
    total = a.is_probs ? total * o : total + o;
  }
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
  int nw;                   // waves per sequence
};

// One workgroup per sequence (a, b); its waves take the steps s = w, w + NW, ...  Wave 0 first
// finds the length (index of the first eos + 1, else S + 1: nothing masked, :1533-1546) with
// ballots over 64 steps at a time.  Forward: per-wave partial sums in step order, combined in wave
// order (deterministic).  Backward: each row is read once into registers and its gradient row
// written once.
template <bool BACKWARD, int NR>
__global__ void __launch_bounds__(1024) slp_kernel(const SlpArgs a) {
  __shared__ int s_len;
  __shared__ float s_part[16];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6), NW = a.nw;
  const int64_t o = blockIdx.x;  // o = ai * B + bi
  const int ai = (int)(o / a.B), bi = (int)(o % a.B);
  const int V = a.V, S = a.S;
  const int64_t *hseq = a.hyp + (int64_t)ai * S * a.B + bi;  // step s at hseq[s * B]
  if (wave == 0) {
    int len = S + 1;
    if (a.has_eos) {
      for (int s0 = 0; s0 < S && len == S + 1; s0 += 8 * PDT_WAVE) {
        int64_t t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = s0 + q * PDT_WAVE + lane < S ? hseq[(int64_t)(s0 + q * PDT_WAVE + lane) * a.B] : a.eos + 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const u64 hit = __ballot(t[q] == a.eos);
          if (hit && len == S + 1) len = s0 + q * PDT_WAVE + (int)__builtin_ctzll(hit) + 1;
        }
      }
    }
    if (lane == 0) s_len = len;
  }
  __syncthreads();
  const int len = s_len;
  const float g = BACKWARD ? a.grad_out[o] : 0.0f;
  float acc = 0.0f;
  // this wave's tokens, 64 steps at a time (lane l: step wave + NW * (j0 + l))
  // (backward: blockIdx.y selects a block of 64 NW steps -- gradient rows are independent, so long
  // sequences spread over several workgroups; forward: one workgroup sums all the steps in order)
  const int j_begin = BACKWARD ? (int)blockIdx.y * PDT_WAVE : 0;
  const int j_end = BACKWARD ? j_begin + PDT_WAVE : 0x7fffffff;
  for (int j0 = j_begin; j0 < j_end && wave + (int64_t)NW * j0 < S; j0 += PDT_WAVE) {
    const int s_mine = wave + NW * (j0 + lane);
    const int64_t tok_mine = s_mine < S ? hseq[(int64_t)s_mine * a.B] : -1;
    for (int j = 0; j < PDT_WAVE; ++j) {
      const int s = wave + NW * (j0 + j);
      if (s >= S) break;
      const int64_t tok = __shfl(tok_mine, j);
      const int64_t row = ((int64_t)ai * S + s) * a.B + bi;
      const bool masked = tok < 0 || tok >= V || s >= len;
      float *go = BACKWARD ? a.grad_logits + row * (int64_t)V : nullptr;
      if (masked) {
        if (BACKWARD)
          for (int v = lane; v < V; v += PDT_WAVE) go[v] = 0.0f;
        continue;
      }
      const float *x = a.logits + row * (int64_t)V;
      float r[NR];
      const RowStats st = row_stats<false, true, NR>(x, 1, V, r);
      const float lse = st.mx + logf(st.sum);
      if (!BACKWARD) {
        acc += (x[tok] - st.mx) - logf(st.sum);
      } else if (V <= NR * PDT_WAVE) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int v = lane + i * PDT_WAVE;
          if (i * PDT_WAVE < V && v < V) go[v] = g * ((v == tok ? 1.0f : 0.0f) - expf(r[i] - lse));
        }
      } else {
        for (int v = lane; v < V; v += PDT_WAVE) go[v] = g * ((v == tok ? 1.0f : 0.0f) - expf(x[v] - lse));
      }
    }
  }
  if (BACKWARD) return;
  if (lane == 0) s_part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float total = 0.0f;
    for (int w = 0; w < NW; ++w) total += s_part[w];
    a.out[o] = total;
  }
}

// waves per sequence: a power of two <= 16, no more than the steps give work to
static int seq_waves(int64_t steps) {
  int nw = 1;
  while (nw < 16 && nw * 4 <= steps) nw *= 2;
  return nw;
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
  const size_t smem = ((size_t)(T > 0 ? T : 1) * 8 + 15) & ~(size_t)15;  // arg-max + value per frame
  if (smem > 160 * 1024) return PDT_E_TOO_LONG;
  if (N > 0x7fffffffll) return PDT_E_TOO_LONG;
  GreedyArgs a{};
  a.logits = logits; a.lg_st = lg_st; a.lg_sn = lg_sn; a.lg_sv = lg_sv;
  a.in_lens = in_lens; a.T = (int)T; a.N = (int)N; a.V = (int)V; a.blank = (int)blank_idx;
  a.is_probs = is_probs; a.max_out = max_out; a.paths = paths; a.pa_st = pa_st; a.pa_sn = pa_sn;
  a.out_lens = out_lens; a.nw = seq_waves(T);
  auto kern = V <= 8 * PDT_WAVE ? ctc_greedy_kernel<8> : ctc_greedy_kernel<16>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)N), dim3(64 * a.nw), smem, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int pdt_sequence_log_probs_forward(const float *logits, const int64_t *hyp, int64_t A, int64_t S,
                                   int64_t B, int64_t V, int has_eos, int64_t eos, float *out,
                                   void *stream) {
  using namespace pdt;
  if (A < 0 || S < 0 || B < 0 || V < 1) return PDT_E_ARG;
  if (A * B == 0) return PDT_OK;
  if ((S > 0 && (!logits || !hyp)) || !out) return PDT_E_ARG;
  if (A * B >= (1ll << 31) || A >= (1ll << 31) || S >= (1ll << 31) || B >= (1ll << 31)) return PDT_E_TOO_LONG;
  SlpArgs a{};
  a.logits = logits; a.hyp = hyp; a.A = (int)A; a.S = (int)S; a.B = (int)B; a.V = (int)V;
  a.has_eos = has_eos; a.eos = eos; a.out = out; a.nw = seq_waves(S);
  auto kern = V <= 8 * PDT_WAVE ? slp_kernel<false, 8> : slp_kernel<false, 16>;
  hipLaunchKernelGGL(kern, dim3((unsigned)(A * B)), dim3(64 * a.nw), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int pdt_sequence_log_probs_backward(const float *logits, const int64_t *hyp, int64_t A, int64_t S,
                                    int64_t B, int64_t V, int has_eos, int64_t eos,
                                    const float *grad_out, float *grad_logits, void *stream) {
  using namespace pdt;
  if (A < 0 || S < 0 || B < 0 || V < 1) return PDT_E_ARG;
  if (A * S * B == 0) return PDT_OK;
  if (!logits || !hyp || !grad_out || !grad_logits) return PDT_E_ARG;
  if (A * B >= (1ll << 31) || A >= (1ll << 31) || S >= (1ll << 31) || B >= (1ll << 31)) return PDT_E_TOO_LONG;
  SlpArgs a{};
  a.logits = logits; a.hyp = hyp; a.A = (int)A; a.S = (int)S; a.B = (int)B; a.V = (int)V;
  a.has_eos = has_eos; a.eos = eos; a.grad_out = grad_out; a.grad_logits = grad_logits; a.nw = seq_waves(S);
  auto kern = V <= 8 * PDT_WAVE ? slp_kernel<true, 8> : slp_kernel<true, 16>;
  const unsigned ny = (unsigned)((S + (int64_t)a.nw * PDT_WAVE - 1) / ((int64_t)a.nw * PDT_WAVE));  // blocks of 64 nw steps
  if (ny > 65535u) return PDT_E_TOO_LONG;
  hipLaunchKernelGGL(kern, dim3((unsigned)(A * B), ny ? ny : 1u), dim3(64 * a.nw), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // extern "C"
