// One frame of CTCPrefixSearch with a LookupLanguageModel in the loop, as ONE kernel (reference
// _decoding.py:1110-1163 around :636-934, with _lm.py:403-515 for the scores):
//   back-off n-gram scores of every prefix's context  ->  shallow fusion / valid mixture with the
//   frame's CTC probabilities  ->  per-prefix sorted lists  ->  the prefix step (ctc_frame).
// The host's frame loop ran three kernels and as many (N K', V) tensors per frame for this (the
// model's scores written, read and rewritten as extension probabilities, read again by the step:
// 0.145 ms per frame at N = 1024, K' = 16, V = 1000); here a row of scores lives in the LDS of the
// wave that forms it, and what the step keeps of it is its sorted list and the K' x K' table of
// the extension probabilities at the prefixes' last tokens (DenseCtx::etab).
// One workgroup per batch element; the waves take the prefixes in turn; wave 0 runs the frame; all
// waves copy the histories.  Scores follow lm_lookup.hip operation for operation, the mix follows
// fusion_ext.hip: the same bits as the three-kernel route.
#include <cstdlib>

#include "advance_args.hpp"
#include "ctc_frame.hpp"
#include "row_reduce.hpp"

namespace pdt {

struct LmTrie {
  const float *logps, *logbs;
  const int *child_start;  // [O] absolute index of a node's first child; end = child_start[i + 1]
  const int *ids;          // labels of nodes >= U, indexed node - U
  const int *succ_start, *succ_tok, *succ_node;  // forward index of the second level (lm_lookup.hip)
  int V, N, U, shift;
  int64_t sos;
};

struct CtcLmAdvArgs {
  CtcAdvArgs s;  // the step's own arguments (s.ext unused; HT = int16_t: s.y_prev / s.y_next point at 16-bit tokens)
  LmTrie lm;
  float beta;
  int valid_mixture;
  int row_floats;  // floats of one wave's row buffer
  // utterances whose frames have run out keep their beam (_decoding.py:1165-1181): frame_lens[n] <=
  // frame (frame_lens may be null: every utterance has this frame)
  const int64_t *frame_lens;
  int64_t frame;
  int64_t yn_ss, yn_sn, yn_sk;  // element strides of y_next (the step functions' own: N * W, W, 1)
};

__device__ __forceinline__ int lm_find_child(const LmTrie &a, int node, int tok) {
  int lo = a.child_start[node];
  const int end = a.child_start[node + 1];
  int hi = end;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a.ids[mid - a.U] < tok) lo = mid + 1; else hi = mid;
  }
  return (lo < end && a.ids[lo - a.U] == tok) ? lo : -1;
}

constexpr int kLmMaxOrder = 16;

// log-probabilities of every vocabulary entry after the context ct[1 .. N - 1] (ct[j]: node of the
// token j positions back, -1: none) into row[0 .. V), by one wave: lm_lookup_kernel's arithmetic
// ((last_logp + cur_backoff) + last_backoff, _lm.py:504-506) with lanes for threads.
__device__ __forceinline__ void lm_score_row(const LmTrie &a, const int (&ct)[kLmMaxOrder], float *row) {
  const int lane = lane_id();
  const int N = a.N;
  float bo[kLmMaxOrder];
  {
    int node = ct[1];
    bo[1] = node >= 0 ? a.logbs[node] : 0.0f;
    for (int n = 2; n <= N - 1; ++n) {
      if (node >= 0) {
        const int tok = ct[n];
        node = tok >= 0 ? lm_find_child(a, node, tok) : -1;
      }
      bo[n] = node >= 0 ? a.logbs[node] : 0.0f;
    }
  }
  auto walk = [&](float lp, float last_b, int node, const int n0) {
    for (int n = n0; n <= N - 1; ++n) {
      if (node >= 0) {
        const int tok = ct[n];
        node = tok >= 0 ? lm_find_child(a, node, tok) : -1;
      }
      const float cur_b = n == N - 1 ? 0.0f : bo[n + 1];
      const float lpd = node >= 0 ? a.logps[node] : 0.0f;
      const bool clobber = node >= 0 && isfinite(lpd);  // (an infinite entry: a node that only exists for its children)
      lp = clobber ? lpd : (lp + cur_b) + last_b;
      last_b = clobber ? cur_b : 0.0f;
    }
    return lp;
  };
  // every entry as if no bigram "c1 v" existed ...
  for (int v0 = lane; v0 < a.V; v0 += 4 * PDT_WAVE) {
    float lp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) lp[q] = v0 + q * PDT_WAVE < a.V ? a.logps[v0 + q * PDT_WAVE] : 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (v0 + q * PDT_WAVE < a.V) row[v0 + q * PDT_WAVE] = walk(lp[q], bo[1], -1, 1);
  }
  wave_sync();
  // ... then the listed successors of c1 again, from their bigram node
  const int c1 = ct[1];
  if (c1 >= 0) {
    for (int e = a.succ_start[c1] + lane; e < a.succ_start[c1 + 1]; e += PDT_WAVE) {
      const int v = a.succ_tok[e];
      if (v >= a.V) continue;
      const int node = a.succ_node[e];
      const float cur_b = 1 == N - 1 ? 0.0f : bo[2];
      const float lpd = a.logps[node];
      const bool clobber = isfinite(lpd);
      const float lp = clobber ? lpd : (a.logps[v] + cur_b) + bo[1];
      row[v] = walk(lp, clobber ? cur_b : 0.0f, node, 2);
    }
  }
  wave_sync();
}

// HT: the element type of the histories y_prev / y_next -- int64_t as the step functions exchange
// them, or int16_t: the host's frame loop keeps its own narrow copy between frames (the copy of the
// (t, N, K) history is what a frame costs beyond ~30 us: a quarter of the bytes)
template <typename HT>
__global__ void __launch_bounds__(512) ctc_lm_advance_kernel(const CtcLmAdvArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  const CtcAdvArgs &a = A.s;
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6), NW = a.waves_per_wg;
  const int64_t n = blockIdx.x;
  const int V = a.V, W = a.W, Kp = a.Kp, S = a.S;
  float *p = reinterpret_cast<float *>(smem);
  FrameLds L;
  L.carve(smem + (size_t)((V + 1 + 3) & ~3) * 4, V, W, Kp, true);
  int *srcs = reinterpret_cast<int *>(L.surv);  // reused after the frame
  unsigned char *tail = smem + a.frame_bytes;
  u64 *my_surv = reinterpret_cast<u64 *>(tail) + (size_t)wave * PDT_SURV_CAP;
  int *newtok = reinterpret_cast<int *>(tail);  // (wave 0's survivor scratch, free once the lists stand)
  float *etab = reinterpret_cast<float *>(tail + (size_t)NW * PDT_SURV_CAP * 8);  // [Kp x Kp]
  float *row_base = etab + ((Kp * Kp + 3) & ~3);
  float *row = row_base + (size_t)wave * A.row_floats;

  const HT *y_prev = reinterpret_cast<const HT *>(a.y_prev);
  HT *y_next = reinterpret_cast<HT *>(a.y_next);
  if (A.frame_lens && A.frame_lens[n] <= A.frame) {
    // no such frame: the beam as it was, brought to the full width (absent entries: -inf, length 0),
    // one more row of zeros (what the host's where() over y / lens / nb / b amounts to; the last
    // tokens and the is-prefix relation of such an utterance are never looked at again)
    for (int i = (int)threadIdx.x; i < W; i += NW * PDT_WAVE) {
      const bool has = i < Kp;
      a.nb_next[n * W + i] = has ? a.nb_prev[n * a.pb_sn + i * a.pb_sk] : -PDT_INF;
      a.b_next[n * W + i] = has ? a.b_prev[n * a.pbb_sn + i * a.pbb_sk] : -PDT_INF;
      a.y_next_lens[n * W + i] = has ? a.lens[n * a.le_sn + i * a.le_sk] : a.lens[n * a.le_sn];
      a.y_next_last[n * W + i] = has ? a.last[n * a.la_sn + i * a.la_sk] : 0;
      a.next_src[n * W + i] = has ? i : 0;
      a.next_nonext[n * W + i] = 1;
      for (int bq = 0; bq < W; ++bq) a.next_isp[(n * W + i) * W + bq] = (uint8_t)(bq == i);
    }
    for (int idx = (int)threadIdx.x; idx < (S + 1) * W; idx += NW * PDT_WAVE) {
      const int i = idx / (S + 1), s = idx - i * (S + 1);
      y_next[(int64_t)s * A.yn_ss + n * A.yn_sn + i * A.yn_sk] =
          (s < S) ? y_prev[(int64_t)s * a.yp_ss + n * a.yp_sn + (i < Kp ? i : 0) * a.yp_sk] : (HT)0;
    }
    return;
  }
  for (int v = (int)threadIdx.x; v < V; v += NW * PDT_WAVE) p[v] = a.nonext[n * a.ne_sn + v * a.ne_sv];
  if (threadIdx.x == 0) p[V] = a.blank[n * a.bl_sn];
  __syncthreads();

  const int M = ctc_list_len(V, W, Kp);
  const float keep = 1.0f - A.beta;
  const float scale = A.valid_mixture ? 1.0f - p[V] : 0.0f;
  // Prefixes with the same context (their last N - 1 tokens) have the same scores, the same mixed
  // row and the same list: one of each set -- its lowest entry, the leader -- is worked out, the
  // others copy (a beam of 16 usually ends in 4-8 different contexts).
  int *ctab = reinterpret_cast<int *>(row_base + (size_t)NW * A.row_floats);  // [Kp x (N - 1)] context nodes
  int *leader = ctab + Kp * (kLmMaxOrder - 1);                                 // [Kp]
  const int NC = A.lm.N - 1;
  if ((int)threadIdx.x < Kp) {
    const int k = (int)threadIdx.x;
    const int64_t pos = a.lens[n * a.le_sn + k * a.le_sk];
    for (int j = 1; j <= NC; ++j) {  // (_lm.py:452-472: sos before the start of the prefix)
      const int64_t q = pos - j;
      int64_t tok = q >= 0 ? (int64_t)y_prev[q * a.yp_ss + n * a.yp_sn + k * a.yp_sk] : A.lm.sos;
      if (A.lm.shift && tok == A.lm.sos) tok = V;
      ctab[k * NC + (j - 1)] = (tok >= 0 && tok < A.lm.U - 1) ? (int)tok : -1;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < Kp) {
    const int k = (int)threadIdx.x;
    int lead = k;
    for (int k2 = k - 1; k2 >= 0; --k2) {
      bool same = true;
      for (int j = 0; j < NC; ++j) same = same && ctab[k2 * NC + j] == ctab[k * NC + j];
      if (same) lead = k2;
    }
    leader[k] = lead;
  }
  __syncthreads();
  int rank = 0;  // leaders before k
  for (int k = 0; k < Kp; ++k) {
    if (leader[k] != k) continue;
    const bool mine = rank % NW == wave;
    ++rank;
    if (!mine) continue;
    int ct[kLmMaxOrder];
    for (int j = 1; j <= NC; ++j) ct[j] = ctab[k * NC + (j - 1)];
    lm_score_row(A.lm, ct, row);
    // the mix with the frame's probabilities (fusion_ext.hip)
    float r[16];
    const RowStats st = row_stats<false, true, 16>(row, 1, V, r);
    const float log_sum = logf(st.sum);
    for (int v = lane; v < V; v += PDT_WAVE) {
      const float xv = row[v];
      float o;
      if (A.valid_mixture) {
        const float lm_p = (expf(xv - st.mx) / st.sum) * scale;
        o = keep * p[v] + A.beta * lm_p;
      } else {
        o = p[v] * expf(A.beta * ((xv - st.mx) - log_sum));
      }
      row[v] = o;
    }
    wave_sync();
    // what the frame reads of this row besides its list: the entries at the prefixes' last tokens
    if (lane < Kp) {
      const int lj = (int)min(max(a.last[n * a.la_sn + lane * a.la_sk], (int64_t)0), (int64_t)(V - 1));
      etab[k * Kp + lane] = row[lj];
    }
    const u64 tk = wave_top_sorted<false, false>(row, V, M, my_surv);
    if (lane < M) {
      L.tl_tok[k * PDT_WAVE + lane] = (int)idx_of(tk);
      L.tl_p[k * PDT_WAVE + lane] = fkey_inv(key_of(tk));
    }
    wave_sync();
  }
  __syncthreads();
  for (int k = wave; k < Kp; k += NW) {  // the others: their leader's list and table row
    const int lead = leader[k];
    if (lead == k) continue;
    if (lane < M) {
      L.tl_tok[k * PDT_WAVE + lane] = L.tl_tok[lead * PDT_WAVE + lane];
      L.tl_p[k * PDT_WAVE + lane] = L.tl_p[lead * PDT_WAVE + lane];
    }
    if (lane < Kp) etab[k * Kp + lane] = etab[lead * Kp + lane];
  }
  __syncthreads();

  DenseCtx dc;
  dc.ext = nullptr;
  dc.ext_sk = 0;
  dc.ext_sv = 0;
  dc.etab = etab;
  dc.etab_stride = Kp;
  dc.y_prev = sizeof(HT) == 8 ? reinterpret_cast<const int64_t *>(y_prev + n * a.yp_sn) : nullptr;
  dc.y_prev16 = sizeof(HT) == 2 ? reinterpret_cast<const int16_t *>(y_prev + n * a.yp_sn) : nullptr;
  dc.yp_ss = a.yp_ss;
  dc.yp_sk = a.yp_sk;
  dc.S = S;
  dc.lists_ready = 1;
  if (wave == 0) {
    Beam bm;
    bm.nb = lane < Kp ? a.nb_prev[n * a.pb_sn + lane * a.pb_sk] : -PDT_INF;
    bm.b = lane < Kp ? a.b_prev[n * a.pbb_sn + lane * a.pbb_sk] : -PDT_INF;
    bm.last = lane < Kp ? (int)min(max(a.last[n * a.la_sn + lane * a.la_sk], (int64_t)-1), (int64_t)V) : 0;
    bm.len = lane < Kp ? (int)a.lens[n * a.le_sn + lane * a.le_sk] : 0;
    bm.node = -1;
    unsigned m = 0u;
    if (lane < Kp)
      for (int b = 0; b < Kp; ++b)
        if (a.isp[n * a.ip_sn + lane * a.ip_sa + b * a.ip_sb]) m |= 1u << b;
    bm.isp = m;
    CtcArgs dummy{};
    dummy.N = a.N;
    int new_src, new_tok, new_kind;
#ifdef PDT_STAMPS
    unsigned pdt_stamp_acc[14] = {0};
#endif
    ctc_frame<true>(bm, p, 1.0f, V, W, Kp, 0, n, dummy, dc, L, new_src, new_tok, new_kind PDT_STAMP_ARG);

    // ---- outputs (:855-934) --------------------------------------------------------------
    if (lane < W) {
      const bool valid = new_kind >= 0;
      a.y_next_last[n * W + lane] = bm.last;
      a.y_next_lens[n * W + lane] = bm.len;
      a.nb_next[n * W + lane] = bm.nb;
      a.b_next[n * W + lane] = bm.b;
      a.next_src[n * W + lane] = valid ? new_src : 0;
      a.next_nonext[n * W + lane] = (uint8_t)(new_kind == 2);
      for (int b = 0; b < W; ++b) a.next_isp[(n * W + lane) * W + b] = (uint8_t)((bm.isp >> b) & 1u);
      srcs[lane] = valid ? new_src : -1;
      L.info[lane] = bm.len;
      L.info[W + lane] = new_kind;
      newtok[lane] = new_tok;
    }
  }
  __syncthreads();
  // histories of the source prefixes, the new token behind them.  Token-contiguous histories (the
  // host's frame loop keeps them as (N, K, S) int16: yp_ss = yn_ss = 1) move 16 bytes at a time --
  // a column of the new beam is a plain copy of its source's column; the (S, N, K) layout of the
  // step functions is a permutation inside every row and goes token by token.
  const HT *yp_n = y_prev + n * a.yp_sn;
  HT *yn_n = y_next + n * A.yn_sn;
  const int threads = NW * PDT_WAVE;
  if (sizeof(HT) == 2 && a.yp_ss == 1 && A.yn_ss == 1 && (a.yp_sk & 7) == 0 && (A.yn_sk & 7) == 0 &&
      (a.yp_sn & 7) == 0 && (A.yn_sn & 7) == 0 && (reinterpret_cast<uintptr_t>(y_prev) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(y_next) & 15) == 0) {
    const int chunks = (S + 1 + 7) >> 3;  // 8 tokens per 16 bytes
    for (int idx = (int)threadIdx.x; idx < W * chunks; idx += threads) {
      const int i = idx / chunks, c = idx - i * chunks;
      const int src = srcs[i];
      const int len_i = L.info[i], kind_i = L.info[W + i];
      const bool ext_i = kind_i == 0 || kind_i == 1;
      union { uint4 q; int16_t t[8]; } v;
      v.q = make_uint4(0u, 0u, 0u, 0u);
      if (src >= 0 && c * 8 < S) v.q = *reinterpret_cast<const uint4 *>(yp_n + src * a.yp_sk + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int spos = c * 8 + e;
        if (spos >= S || src < 0) v.t[e] = 0;
        if (src >= 0 && ext_i && spos == len_i - 1) v.t[e] = (int16_t)newtok[i];
      }
      *reinterpret_cast<uint4 *>(yn_n + i * A.yn_sk + c * 8) = v.q;
    }
  } else {
    for (int idx = (int)threadIdx.x; idx < (S + 1) * W; idx += threads) {
      const int s = idx / W, i = idx - s * W;
      const int src = srcs[i];
      const int len_i = L.info[i], kind_i = L.info[W + i];
      const bool ext_i = kind_i == 0 || kind_i == 1;
      HT v = 0;
      if (src >= 0) {
        if (ext_i && s == len_i - 1) v = (HT)newtok[i];
        else if (s < S) v = yp_n[(int64_t)s * a.yp_ss + src * a.yp_sk];
      }
      yn_n[(int64_t)s * A.yn_ss + i * A.yn_sk] = v;
    }
  }
}

}  // namespace pdt

extern "C" int pdt_ctc_lookup_lm_advance(
    const float *nonext, int64_t ne_sn, int64_t ne_sv, const float *blank, int64_t bl_sn, int64_t N, int64_t Kp,
    int64_t V, int64_t width, const float *nb_prev, int64_t nb_sn, int64_t nb_sk, const float *b_prev,
    int64_t b_sn, int64_t b_sk, const void *y_prev, int64_t S, int64_t yp_ss, int64_t yp_sn, int64_t yp_sk,
    const int64_t *y_prev_last, int64_t la_sn, int64_t la_sk, const int64_t *y_prev_lens, int64_t le_sn,
    int64_t le_sk, const uint8_t *prev_is_prefix, int64_t ip_sn, int64_t ip_sa, int64_t ip_sb,
    const float *logps, const float *logbs, const int32_t *child_start, const int32_t *ids,
    const int32_t *succ_start, const int32_t *succ_tok, const int32_t *succ_node, int64_t max_ngram, int64_t U,
    int64_t sos, float beta, int valid_mixture, void *y_next, int64_t *y_next_last, int64_t *y_next_lens,
    float *nb_next, float *b_next, uint8_t *next_is_prefix, int64_t *next_src, uint8_t *next_is_nonext,
    int history_bytes, const int64_t *frame_lens, int64_t frame_index, int64_t yn_ss, int64_t yn_sn,
    int64_t yn_sk, void *stream) {
  using namespace pdt;
  if (N < 0 || Kp < 1 || V < 1 || width < 1 || S < 0 || max_ngram < 2 || U < V + 1 || U > V + 2) return PDT_E_ARG;
  if (history_bytes != 8 && !(history_bytes == 2 && V <= 32767)) return PDT_E_ARG;
  if (N == 0) return PDT_OK;
  if (!nonext || !blank || !nb_prev || !b_prev || !y_prev_last || !y_prev_lens || !prev_is_prefix ||
      (S > 0 && !y_prev) || !y_next || !y_next_last || !y_next_lens || !nb_next || !b_next || !next_is_prefix ||
      !next_src || !next_is_nonext || !logps || !logbs || !child_start || !ids || !succ_start || !succ_tok ||
      !succ_node)
    return PDT_E_ARG;
  if (V >= (1 << 30) || S >= (1 << 26) || N >= (1ll << 31)) return PDT_E_TOO_LONG;
  if (width > kMaxWidth || Kp > kMaxWidth || max_ngram > kLmMaxOrder) return PDT_E_TOO_LONG;
  CtcLmAdvArgs A{};
  CtcAdvArgs &a = A.s;
  a.nonext = nonext; a.ne_sn = ne_sn; a.ne_sv = ne_sv;
  a.blank = blank; a.bl_sn = bl_sn;
  a.nb_prev = nb_prev; a.pb_sn = nb_sn; a.pb_sk = nb_sk;
  a.b_prev = b_prev; a.pbb_sn = b_sn; a.pbb_sk = b_sk;
  a.y_prev = reinterpret_cast<const int64_t *>(y_prev); a.yp_ss = yp_ss; a.yp_sn = yp_sn; a.yp_sk = yp_sk;
  a.last = y_prev_last; a.la_sn = la_sn; a.la_sk = la_sk;
  a.lens = y_prev_lens; a.le_sn = le_sn; a.le_sk = le_sk;
  a.isp = prev_is_prefix; a.ip_sn = ip_sn; a.ip_sa = ip_sa; a.ip_sb = ip_sb;
  a.N = (int)N; a.Kp = (int)Kp; a.V = (int)V; a.W = (int)width; a.S = (int)S;
  a.y_next = reinterpret_cast<int64_t *>(y_next); a.y_next_last = y_next_last; a.y_next_lens = y_next_lens;
  a.next_src = next_src; a.nb_next = nb_next; a.b_next = b_next;
  a.next_isp = next_is_prefix; a.next_nonext = next_is_nonext;
  A.lm = LmTrie{logps, logbs, child_start, ids, succ_start, succ_tok, succ_node, (int)V, (int)max_ngram, (int)U,
                (int)(U - V - 1), sos};
  A.beta = beta;
  A.valid_mixture = valid_mixture;
  A.frame_lens = frame_lens;
  A.frame = frame_index;
  A.yn_ss = yn_ss; A.yn_sn = yn_sn; A.yn_sk = yn_sk;
  A.row_floats = (int)((V + 3) & ~(int64_t)3);
  // waves per element: as many (a power of two <= min(Kp, 8)) as still let four workgroups share a
  // CU's LDS -- every wave carries a row of V floats
  size_t frame = (size_t)((a.V + 1 + 3) & ~3) * 4 + FrameLds::bytes(a.V, a.W, a.Kp, true);
  frame = (frame + 15) & ~(size_t)15;
  auto lds_of = [&](int nw) {
    return frame + (size_t)nw * PDT_SURV_CAP * 8 + (size_t)((Kp * Kp + 3) & ~3) * 4 + (size_t)nw * A.row_floats * 4 +
           (size_t)Kp * kLmMaxOrder * 4;  // + the context table and the leaders
  };
  int nw = 1;
  while (nw < 8 && nw * 2 <= a.Kp) nw *= 2;
  while (nw > 1 && lds_of(nw) > 40 * 1024) nw >>= 1;
  if (const char *e = getenv("PDT_LM_STEP_WAVES")) {  // (experiments)
    const int f = atoi(e);
    if (f == 1 || f == 2 || f == 4 || f == 8) nw = f;
  }
  const size_t smem = lds_of(nw);
  if (smem > 160 * 1024) return PDT_E_TOO_LONG;
  a.waves_per_wg = nw;
  a.frame_bytes = (int)frame;
  auto kern = history_bytes == 2 ? ctc_lm_advance_kernel<int16_t> : ctc_lm_advance_kernel<int64_t>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)a.N), dim3(64 * nw), smem, (hipStream_t)stream, A);
  return (int)hipGetLastError();
}
