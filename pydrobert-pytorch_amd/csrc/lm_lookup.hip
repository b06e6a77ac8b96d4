// Back-off n-gram scoring over the reference's reverse-trie buffers (reference _lm.py:403-515,
// trie layout :609-677).
//
// For every query row (one history position of one batch element) and every vocabulary entry v
// the reference walks two paths per step of a Python loop over the n-gram order, holding
// (M + B, S) index / mask tensors (M = rows * V, S = max direct descendants) and scanning all S
// children of every node.  Here one thread owns one (row, v) pair and keeps the walk in
// registers: a binary search among the node's children per order (children are sorted by id,
// invariant 3 at :628), the back-off chain of the row's context is walked alongside.  The only
// HBM traffic that scales is the dense (rows, V) float32 result; the trie tables (a few MB) stay
// in L2 / MALL.  With the forward index (LmArgs::succ_*, built by the host from the same buffers)
// a row no longer searches the children of EVERY vocabulary entry for its last context token: all
// entries get the no-match value (two adds), and only the successors the model lists for that
// token -- a hundredth of the vocabulary, typically -- walk on from their known node.
//
// Arithmetic is the reference's, in the reference's order (float32,
// (last_logp + cur_backoff) + last_backoff, :504-506), so results are bit-identical.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pdt_amd.h"

namespace pdt {

struct LmArgs {
  const int64_t *hist;  // (S, B) through element strides
  int64_t h_ss, h_sb;
  const int64_t *idx;   // per-row position through idx_stride (0: one value), or null: full mode
  int64_t idx_stride;
  int S, B;
  int64_t rows;         // B, or (S + 1) * B in full mode (row = t * B + b, idx = t)
  const float *logps, *logbs;
  const int *child_start;  // [O] absolute index of a node's first child; end = child_start[i + 1]
  const int *ids;          // labels of nodes >= U, indexed node - U
  // forward index of the second level (optional): for a context token c the (last token v, node)
  // pairs of the bigrams "c v" the model holds, v ascending: entries succ_start[c] .. succ_start[c + 1]
  const int *succ_start, *succ_tok, *succ_node;
  int V, N, U, shift;
  int64_t sos;
  float *out;           // (rows, V)
  int *status;          // bit 0: a position outside [0, S]
};

// child of `node` labelled `tok`, or -1
__device__ __forceinline__ int find_child(const LmArgs &a, int node, int tok) {
  int lo = a.child_start[node];
  const int end = a.child_start[node + 1];
  int hi = end;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a.ids[mid - a.U] < tok) lo = mid + 1; else hi = mid;
  }
  return (lo < end && a.ids[lo - a.U] == tok) ? lo : -1;
}

constexpr int kMaxOrder = 16;

// One workgroup per query row: the walk of the row's CONTEXT (its back-off chain) does not depend
// on the vocabulary entry, so a thread does it once and then takes v = tid, tid + 128, ...
// (one thread per (row, v) repeated it V times: a third of a trigram model's searches).
#ifndef PDT_LM_THREADS
#define PDT_LM_THREADS 128  // (more rows per CU at once: a row's context walk is a chain of ~10 loads)
#endif
constexpr int kLmThreads = PDT_LM_THREADS;
__global__ void __launch_bounds__(kLmThreads) lm_lookup_kernel(const LmArgs a) {
  const int64_t r = blockIdx.x;
  int64_t pos;
  int b;
  if (a.idx) {
    b = (int)r;
    pos = a.idx[r * a.idx_stride];
    if (pos < 0 || pos > a.S) {
      if (threadIdx.x == 0) atomicOr(a.status, 1);
      pos = pos < 0 ? 0 : a.S;
    }
  } else {
    const int64_t t = r / a.B;
    b = (int)(r - t * a.B);
    pos = t;
  }
  // context token at distance n >= 1 behind the queried position; sos beyond the start (:452-461)
  auto ctx = [&](int n) -> int {
    const int64_t p = pos - n;
    int64_t tok = p >= 0 ? a.hist[p * a.h_ss + (int64_t)b * a.h_sb] : a.sos;
    if (a.shift && tok == a.sos) tok = a.V;  // :471-472
    return (tok >= 0 && tok < a.U - 1) ? (int)tok : -1;
  };
  const int N = a.N;
  int ct[kMaxOrder];  // the context tokens, read once
  for (int n = 1; n <= N - 1; ++n) ct[n] = ctx(n);
  // back-off chain of the context: bo[n] = log-backoff of the length-n context, 0 once the
  // context is no longer in the trie (:491-497)
  float bo[kMaxOrder];
  {
    int node = ct[1];
    bo[1] = node >= 0 ? a.logbs[node] : 0.0f;
    for (int n = 2; n <= N - 1; ++n) {
      if (node >= 0) {
        const int tok = ct[n];
        node = tok >= 0 ? find_child(a, node, tok) : -1;
      }
      bo[n] = node >= 0 ? a.logbs[node] : 0.0f;
    }
  }
  // the walk of one vocabulary entry from order n0 on, given its node at order n0 - 1
  auto walk = [&](float lp, float last_b, int node, const int n0) {
    for (int n = n0; n <= N - 1; ++n) {
      if (node >= 0) {
        const int tok = ct[n];
        node = tok >= 0 ? find_child(a, node, tok) : -1;
      }
      const float cur_b = n == N - 1 ? 0.0f : bo[n + 1];
      const float lpd = node >= 0 ? a.logps[node] : 0.0f;
      // an infinite entry marks a node that only exists for its children (:499-503)
      const bool clobber = node >= 0 && isfinite(lpd);
      lp = clobber ? lpd : (lp + cur_b) + last_b;
      last_b = clobber ? cur_b : 0.0f;
    }
    return lp;
  };
  if (!a.succ_start) {
    for (int v = (int)threadIdx.x; v < a.V; v += kLmThreads) a.out[r * a.V + v] = walk(a.logps[v], bo[1], v, 1);
    return;
  }
  // every entry as if no bigram "c1 v" existed (node = -1 from the first order on) ...
  for (int v0 = (int)threadIdx.x; v0 < a.V; v0 += 4 * kLmThreads) {  // (four loads in flight)
    float lp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) lp[q] = v0 + q * kLmThreads < a.V ? a.logps[v0 + q * kLmThreads] : 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (v0 + q * kLmThreads < a.V) a.out[r * a.V + v0 + q * kLmThreads] = walk(lp[q], bo[1], -1, 1);
  }
  __syncthreads();  // (a workgroup's own global writes are visible to it after the barrier)
  // ... then the listed successors of c1 again, from their bigram node
  const int c1 = ct[1];
  if (c1 < 0) return;
  for (int e = a.succ_start[c1] + (int)threadIdx.x; e < a.succ_start[c1 + 1]; e += kLmThreads) {
    const int v = a.succ_tok[e];
    if (v >= a.V) continue;
    const int node = a.succ_node[e];
    // order 1 with the node known, then orders 2 .. N - 1 as usual
    const float cur_b = 1 == N - 1 ? 0.0f : bo[2];
    const float lpd = a.logps[node];
    const bool clobber = isfinite(lpd);
    const float lp = clobber ? lpd : (a.logps[v] + cur_b) + bo[1];
    a.out[r * a.V + v] = walk(lp, clobber ? cur_b : 0.0f, node, 2);
  }
}

}  // namespace pdt

extern "C" {

int pdt_lookup_lm_log_probs(const int64_t *hist, int64_t S, int64_t B, int64_t h_ss, int64_t h_sb,
                            const int64_t *idx, int64_t idx_stride, int64_t rows,
                            const float *logps, const float *logbs, const int32_t *child_start,
                            const int32_t *ids, const int32_t *succ_start, const int32_t *succ_tok,
                            const int32_t *succ_node, int64_t V, int64_t N, int64_t U, int64_t sos,
                            float *out, int32_t *status, void *stream) {
  using namespace pdt;
  if (S < 0 || B < 0 || rows < 0 || V < 1 || N < 2 || U < V + 1 || U > V + 2) return PDT_E_ARG;
  if (rows == 0) return PDT_OK;
  if (N > kMaxOrder) return PDT_E_TOO_LONG;
  if (rows * V >= (1ll << 31) * 256 || S >= (1ll << 31) || B >= (1ll << 31)) return PDT_E_TOO_LONG;
  if ((S > 0 && !hist) || !logps || !logbs || !child_start || !ids || !out || !status)
    return PDT_E_ARG;
  if (!idx && rows != (S + 1) * B) return PDT_E_ARG;
  if (idx && rows != B) return PDT_E_ARG;
  LmArgs a{};
  a.hist = hist; a.h_ss = h_ss; a.h_sb = h_sb; a.idx = idx; a.idx_stride = idx_stride;
  a.S = (int)S; a.B = (int)B; a.rows = rows;
  a.logps = logps; a.logbs = logbs; a.child_start = child_start; a.ids = ids;
  if (succ_start && succ_tok && succ_node) {
    a.succ_start = succ_start; a.succ_tok = succ_tok; a.succ_node = succ_node;
  }
  a.V = (int)V; a.N = (int)N; a.U = (int)U; a.shift = (int)(U - V - 1); a.sos = sos;
  a.out = out; a.status = status;
  if (rows >= (1ll << 31)) return PDT_E_TOO_LONG;
  hipLaunchKernelGGL(lm_lookup_kernel, dim3((unsigned)rows), dim3(kLmThreads), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // extern "C"
