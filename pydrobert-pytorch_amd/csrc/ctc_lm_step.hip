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
// (The timing experiments of round 3 -- a phase run twice with identical results, phases skipped --
// are described in EXPERIMENTS.md section 9.16; their macro families left the file in round 4.  What
// stays is the per-phase stamp build, -DPDT_LM_STAMPS.)
#include <cstdlib>

#include "advance_args.hpp"
#include "ctc_frame.hpp"
#include "row_reduce.hpp"
#include "switches.hpp"

namespace pdt {

#ifdef PDT_LM_STAMPS
// per-phase cycles summed over all waves (s_memtime, 100 MHz): 0 setup, 1 LM rows, 2 statistics + mix,
// 3 list selection, 4 waiting at the barriers, 5 the frame (wave 0), 6 slots + histories, 7 launches
__device__ unsigned long long pdt_lm_stamp_acc[8];
#define LM_STAMP(i)                                                          \
  do {                                                                       \
    const unsigned long long now_ = __builtin_readcyclecounter();            \
    if (lane == 0) atomicAdd(&pdt_lm_stamp_acc[i], now_ - stamp_t);          \
    stamp_t = now_;                                                          \
  } while (0)
#else
#define LM_STAMP(i)
#endif

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
  // Slot mode (pdt_ctc_lookup_lm_search): histories live in 2 W slots per utterance, s.y_prev is the
  // slot array (token-contiguous: yp_ss = 1, yp_sk = slot length, yp_sn = 2 W slots) and beam entry k
  // owns slot slot_prev[n * W + k].  A surviving prefix keeps its slot; an extended one gets a slot that
  // was free before the frame, a copy of its source's tokens and the new token -- nothing else moves.
  const int32_t *slot_prev;
  int32_t *slot_next;
  // factor rows of a bigram model by context token, [U][row_floats], and their state (2: ready); or null
  float *cache;
  int32_t *cache_flag;
  int cache_stride;  // floats per row: V rounded up to a 128-byte line
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
__device__ __forceinline__ void lm_frame(const CtcLmAdvArgs &A, unsigned char *smem) {
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
  const bool slots = A.slot_prev != nullptr;
#ifdef PDT_LM_STAMPS
  unsigned long long stamp_t = __builtin_readcyclecounter();
  if (threadIdx.x == 0) atomicAdd(&pdt_lm_stamp_acc[7], 1ull);
#endif
  if (slots && A.frame_lens && A.frame_lens[n] <= A.frame) {
    // no such frame, slot mode: the beam as it was at the full width; the histories stay where they are
    for (int i = (int)threadIdx.x; i < W; i += NW * PDT_WAVE) {
      const bool has = i < Kp;
      a.nb_next[n * W + i] = has ? a.nb_prev[n * a.pb_sn + i * a.pb_sk] : -PDT_INF;
      a.b_next[n * W + i] = has ? a.b_prev[n * a.pbb_sn + i * a.pbb_sk] : -PDT_INF;
      a.y_next_lens[n * W + i] = has ? a.lens[n * a.le_sn + i * a.le_sk] : a.lens[n * a.le_sn];
      a.y_next_last[n * W + i] = has ? a.last[n * a.la_sn + i * a.la_sk] : 0;
      a.next_src[n * W + i] = has ? i : 0;
      a.next_nonext[n * W + i] = 1;
      A.slot_next[n * W + i] = has ? A.slot_prev[n * W + i] : -1;
      for (int bq = 0; bq < W; ++bq) a.next_isp[(n * W + i) * W + bq] = (uint8_t)(bq == i);
    }
    return;
  }
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
  int *slot_l = leader + Kp;                                                   // [W] slots of the old beam; then of the new
  int *dst_l = slot_l + W;                                                     // [W] slots of the new beam
  const int NC = A.lm.N - 1;
  if ((int)threadIdx.x < Kp) {
    const int k = (int)threadIdx.x;
    const int sl = slots ? A.slot_prev[n * W + k] : k;
    if (slots) slot_l[k] = sl;
    const int64_t pos = a.lens[n * a.le_sn + k * a.le_sk];
    for (int j = 1; j <= NC; ++j) {  // (_lm.py:452-472: sos before the start of the prefix)
      const int64_t q = pos - j;
      int64_t tok = (q >= 0 && sl >= 0) ? (int64_t)y_prev[q * a.yp_ss + n * a.yp_sn + sl * a.yp_sk] : A.lm.sos;
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
  // state of every leader's cached row, one load for all of them (lane = beam entry)
  int row_state = 0;
  if (A.cache && lane < Kp && leader[lane] == lane && ctab[lane * NC] >= 0)
    row_state = __hip_atomic_load(&A.cache_flag[ctab[lane * NC]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int rank = 0;  // leaders before k
  for (int k = 0; k < Kp; ++k) {
    if (leader[k] != k) continue;
    const bool mine = rank % NW == wave;
    ++rank;
    if (!mine) continue;
    LM_STAMP(0);
    int ct[kLmMaxOrder];
    for (int j = 1; j <= NC; ++j) ct[j] = ctab[k * NC + (j - 1)];
    // The model's factor of the mix depends on the context alone: exp(x - max) / sum for the valid
    // mixture, exp(beta (log_softmax x)) for shallow fusion.  A whole search (slot mode, bigram model)
    // keeps one row of factors per context token in its workspace: computed by whoever needs it
    // first (several workgroups at once write the same bits), read by everybody afterwards.
    float *crow = nullptr;
    bool hit = false;
    if (A.cache && ct[1] >= 0) {
      // (rows start on 128-byte lines and a compute unit touches a row only after it has seen the
      // row's flag: nothing stale can sit in its L1, so the flag is read relaxed -- an acquire here
      // is an L1 invalidation, ~1.7 us per look-up)
      crow = A.cache + (size_t)ct[1] * A.cache_stride;
      hit = __builtin_amdgcn_readlane(row_state, k) == 2;
    }
    if (hit) {
      for (int v0 = lane; v0 < V; v0 += 8 * PDT_WAVE) {  // (eight loads in flight)
        float f[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = v0 + q * PDT_WAVE < V ? crow[v0 + q * PDT_WAVE] : 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (v0 + q * PDT_WAVE < V) row[v0 + q * PDT_WAVE] = f[q];
      }
      LM_STAMP(1);
    } else {
      lm_score_row(A.lm, ct, row);
      LM_STAMP(1);
      float r[16];
      const RowStats st = row_stats<false, true, 16>(row, 1, V, r);
      const float log_sum = logf(st.sum);
      for (int v = lane; v < V; v += PDT_WAVE) {
        const float xv = row[v];
        const float f = A.valid_mixture ? expf(xv - st.mx) / st.sum : expf(A.beta * ((xv - st.mx) - log_sum));
        row[v] = f;
        if (crow) crow[v] = f;
      }
      if (crow) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(&A.cache_flag[ct[1]], 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    wave_sync();
    // the mix with the frame's probabilities (fusion_ext.hip)
    unsigned lmax = 0u;  // this lane's largest ordering key (every mixed value is >= +0): the selection's first pass
    for (int v = lane; v < V; v += PDT_WAVE) {
      const float f = row[v];
      float o;
      if (A.valid_mixture) {
        const float lm_p = f * scale;
        o = keep * p[v] + A.beta * lm_p;
      } else {
        o = p[v] * f;
      }
      row[v] = o;
      lmax = max(lmax, fkey_nonneg(o));
    }
    wave_sync();
    LM_STAMP(2);
    // what the frame reads of this row besides its list: the entries at the prefixes' last tokens
    if (lane < Kp) {
      const int lj = (int)min(max(a.last[n * a.la_sn + lane * a.la_sk], (int64_t)0), (int64_t)(V - 1));
      etab[k * Kp + lane] = row[lj];
    }
    const u64 tk = wave_top_sorted<false, true>(row, V, M, my_surv, V > PDT_WAVE ? &lmax : nullptr);
    if (lane < M) {
      L.tl_tok[k * PDT_WAVE + lane] = (int)idx_of(tk);
      L.tl_p[k * PDT_WAVE + lane] = fkey_nonneg_inv(key_of(tk));
    }
    wave_sync();
    LM_STAMP(3);
  }
  LM_STAMP(0);
  __syncthreads();
  LM_STAMP(4);
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
  dc.slot = slots ? slot_l : nullptr;
  dc.lists_ready = 1;
  LM_STAMP(0);
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
    LM_STAMP(5);
  }
  __syncthreads();
  LM_STAMP(4);
  // histories of the source prefixes, the new token behind them.  Token-contiguous histories (the
  // host's frame loop keeps them as (N, K, S) int16: yp_ss = yn_ss = 1) move 16 bytes at a time --
  // a column of the new beam is a plain copy of its source's column; the (S, N, K) layout of the
  // step functions is a permutation inside every row and goes token by token.
  if (slots) {
    // the new beam's slots: survivors keep theirs, extensions take the slots that were free before the
    // frame in rank order (2 W slots, at most W of them in use: there are always enough)
    if (wave == 0) {
      const bool valid = lane < W && srcs[lane] >= 0;
      const int kind = lane < W ? L.info[W + lane] : -1;
      const bool ext = valid && (kind == 0 || kind == 1);
      u64 used = 0ull;
      for (int k = 0; k < Kp; ++k)
        if (slot_l[k] >= 0) used |= 1ull << slot_l[k];
      u64 free_slots = ~used & (2 * W >= 64 ? ~0ull : (1ull << (2 * W)) - 1ull);
      const u64 extm = __ballot(ext);
      const int rank = __popcll(extm & ((1ull << lane) - 1ull));
      int ns = -1;
      if (ext) {
        for (int j = 0; j < rank; ++j) free_slots &= free_slots - 1ull;
        ns = __builtin_ctzll(free_slots);
      } else if (valid) {
        ns = slot_l[srcs[lane]];
      }
      if (lane < W) {
        dst_l[lane] = ns;
        A.slot_next[n * W + lane] = ns;
      }
    }
    __syncthreads();
    const HT *hp = y_prev + n * a.yp_sn;
    HT *hn = const_cast<HT *>(hp);
    const int threads = NW * PDT_WAVE;
    constexpr int PER = 16 / (int)sizeof(HT);  // tokens per 16 bytes
    const int chunks = (S + 1 + PER - 1) / PER;
    for (int idx = (int)threadIdx.x; idx < W * chunks; idx += threads) {
      const int i = idx / chunks, c = idx - i * chunks;
      const int src = srcs[i], kind_i = L.info[W + i], len_i = L.info[i];
      if (src < 0 || !(kind_i == 0 || kind_i == 1) || c * PER >= len_i) continue;
      union { uint4 q; HT t[PER]; } v;
      v.q = *reinterpret_cast<const uint4 *>(hp + (int64_t)slot_l[src] * a.yp_sk + c * PER);
#pragma unroll
      for (int e = 0; e < PER; ++e)
        if (c * PER + e == len_i - 1) v.t[e] = (HT)newtok[i];
      *reinterpret_cast<uint4 *>(hn + (int64_t)dst_l[i] * a.yp_sk + c * PER) = v.q;
    }
    LM_STAMP(6);
    return;
  }
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

template <typename HT>
__global__ void __launch_bounds__(512, 4) ctc_lm_advance_kernel(const CtcLmAdvArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  lm_frame<HT>(A, smem);
}

// state of one side of the search's ping-pong: everything a frame reads / writes besides the histories
struct LmSearchState {
  float *nb, *b;
  int64_t *last, *lens;
  uint8_t *isp;
  int32_t *slot;
};

// Every frame of the search in ONE launch: a workgroup stays with its utterance and runs the frames
// one after the other (no utterance waits for the slowest one of every frame: the frames' costs
// differ with the number of distinct contexts in the beam, and a launch per frame takes the
// maximum over the batch a thousand times).  The state goes through the same global buffers as in
// the launch-per-frame form -- written and read by this workgroup only, a barrier in between.
template <typename HT>
__global__ void __launch_bounds__(512, 4)
ctc_lm_search_kernel(const CtcLmAdvArgs A0, const LmSearchState st0, const LmSearchState st1, const float *probs,
                     const int64_t p_st, const int64_t p_sn, const int64_t p_sv, const int n_frames) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int W = A0.s.W;
  for (int t = 0; t < n_frames; ++t) {
    CtcLmAdvArgs A = A0;
    CtcAdvArgs &a = A.s;
    const LmSearchState &prev = (t & 1) ? st1 : st0, &next = (t & 1) ? st0 : st1;
    a.Kp = t == 0 ? 1 : W;
    a.S = t;
    a.nonext = probs + t * p_st; a.ne_sn = p_sn; a.ne_sv = p_sv;
    a.blank = probs + t * p_st + (int64_t)a.V * p_sv; a.bl_sn = p_sn;
    a.nb_prev = prev.nb; a.pb_sn = W; a.pb_sk = 1;
    a.b_prev = prev.b; a.pbb_sn = W; a.pbb_sk = 1;
    a.last = prev.last; a.la_sn = W; a.la_sk = 1;
    a.lens = prev.lens; a.le_sn = W; a.le_sk = 1;
    a.isp = prev.isp; a.ip_sn = (int64_t)W * W; a.ip_sa = W; a.ip_sb = 1;
    a.y_next_last = next.last; a.y_next_lens = next.lens;
    a.nb_next = next.nb; a.b_next = next.b; a.next_isp = next.isp;
    A.slot_prev = prev.slot; A.slot_next = next.slot;
    A.frame = t;
    lm_frame<HT>(A, smem);
    // this utterance's state and histories for the next frame: written by this workgroup, read by it --
    // workgroup scope: the waves of a workgroup share their CU's L1, which the stores go through, so a
    // wait for them and the barrier are all it takes.  (Agent scope here is an L2 write-back and an L1
    // invalidation per frame and workgroup: 143 ms instead of 41.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

}  // namespace pdt

namespace pdt {

// LDS plan of the frame kernel for (V, W, Kp); fills A.row_floats and the step's frame_bytes /
// waves_per_wg.  Returns the dynamic LDS bytes, 0 when the shape does not fit.
static size_t plan_lm_frame(CtcLmAdvArgs &A) {
  CtcAdvArgs &a = A.s;
  const int64_t V = a.V, Kp = a.Kp, width = a.W;
  A.row_floats = (int)((V + 3) & ~(int64_t)3);
  // waves per element: as many (a power of two <= min(Kp, 8)) as still let four workgroups share a
  // CU's LDS -- every wave carries a row of V floats
  size_t frame = (size_t)((a.V + 1 + 3) & ~3) * 4 + FrameLds::bytes(a.V, a.W, a.Kp, true);
  frame = (frame + 15) & ~(size_t)15;
  auto lds_of = [&](int nw) {
    return frame + (size_t)nw * PDT_SURV_CAP * 8 + (size_t)((Kp * Kp + 3) & ~3) * 4 + (size_t)nw * A.row_floats * 4 +
           (size_t)Kp * kLmMaxOrder * 4 + (size_t)width * 8;  // + the context table, the leaders, two slot lists
  };
  int nw = 1;
  while (nw < 8 && nw * 2 <= a.Kp) nw *= 2;
  while (nw > 1 && lds_of(nw) > 40 * 1024) nw >>= 1;
  {  // (experiments)
    const int f = switches().lm_step_waves;
    if (f == 1 || f == 2 || f == 4 || f == 8) nw = f;
  }
  const size_t smem = lds_of(nw);
  if (smem > 160 * 1024) return 0;
  a.waves_per_wg = nw;
  a.frame_bytes = (int)frame;
  return smem;
}

static int launch_lm_frame(CtcLmAdvArgs &A, const int history_bytes, hipStream_t stream) {
  const size_t smem = plan_lm_frame(A);
  if (smem == 0) return PDT_E_TOO_LONG;
  auto kern = history_bytes == 2 ? ctc_lm_advance_kernel<int16_t> : ctc_lm_advance_kernel<int64_t>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)A.s.N), dim3(64 * A.s.waves_per_wg), smem, stream, A);
  return (int)hipGetLastError();
}

// ---- the whole search (pdt_ctc_lookup_lm_search) ------------------------------------------------
__global__ void lm_search_init_kernel(LmSearchState st, const int N, const int W) {
  const int n = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (n >= N) return;
  // one empty prefix per utterance, all of its mass on "ends in blank" (_decoding.py:1083-1097)
  st.nb[(int64_t)n * W] = 0.0f;
  st.b[(int64_t)n * W] = 1.0f;
  st.last[(int64_t)n * W] = 0;
  st.lens[(int64_t)n * W] = 0;
  st.isp[(int64_t)n * W * W] = 1;
  st.slot[(int64_t)n * W] = 0;
}

// y[s][n][i] = token s of beam entry i (0 beyond its length / for absent entries), lens and masses out
template <typename HT>
__global__ void __launch_bounds__(256)
lm_search_gather_kernel(const HT *hist, const int64_t h_sn, const int64_t h_sk, const LmSearchState st, const int N,
                        const int W, const int S_out, int64_t *y, int64_t *y_lens, float *nb, float *b) {
  const int64_t n = blockIdx.x;
  for (int i = (int)threadIdx.x; i < W; i += (int)blockDim.x) {
    y_lens[n * W + i] = st.lens[n * W + i];
    nb[n * W + i] = st.nb[n * W + i];
    b[n * W + i] = st.b[n * W + i];
  }
  for (int idx = (int)threadIdx.x; idx < S_out * W; idx += (int)blockDim.x) {
    const int s = idx / W, i = idx - s * W;
    const int slot = st.slot[n * W + i];
    const int64_t len = st.lens[n * W + i];
    y[((int64_t)s * N + n) * W + i] = (slot >= 0 && s < len) ? (int64_t)hist[n * h_sn + (int64_t)slot * h_sk + s] : 0;
  }
}

static size_t lm_align(size_t v) { return (v + 255) & ~(size_t)255; }

struct LmSearchPlan {
  size_t hist, side[2][6], nonext_flags, src, cache, cache_flag, total;
  int64_t smax;
  int cached;  // factor rows by context token kept (bigram model, table within kLmCacheBytes)
};
constexpr size_t kLmCacheBytes = (size_t)1 << 30;
static LmSearchPlan plan_lm_search(int64_t n_frames, int64_t N, int64_t W, int history_bytes, int64_t V, int64_t U,
                                   int64_t max_ngram) {
  LmSearchPlan p{};
  p.smax = (n_frames + 1 + 7) / 8 * 8;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = lm_align(off + bytes); return o; };
  p.hist = take((size_t)N * 2 * W * p.smax * history_bytes);
  for (int sd = 0; sd < 2; ++sd) {
    p.side[sd][0] = take((size_t)N * W * 4);      // nb
    p.side[sd][1] = take((size_t)N * W * 4);      // b
    p.side[sd][2] = take((size_t)N * W * 8);      // last
    p.side[sd][3] = take((size_t)N * W * 8);      // lens
    p.side[sd][4] = take((size_t)N * W * W);      // is-prefix
    p.side[sd][5] = take((size_t)N * W * 4);      // slots
  }
  p.nonext_flags = take((size_t)N * W);
  p.src = take((size_t)N * W * 8);
  const size_t rf = (size_t)((V + 31) & ~(int64_t)31);
  p.cached = max_ngram == 2 && (size_t)U * rf * 4 <= kLmCacheBytes;
  if (!switches().lm_cache) p.cached = 0;  // (comparisons)
  if (p.cached) {
    p.cache = take((size_t)U * rf * 4);
    p.cache_flag = take((size_t)U * 4);
  }
  p.total = off;
  return p;
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
  return launch_lm_frame(A, history_bytes, (hipStream_t)stream);
}

#ifdef PDT_LM_STAMPS
extern "C" int pdt_debug_lm_stamps(unsigned long long *out, int reset) {
  (void)hipDeviceSynchronize();
  if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pdt::pdt_lm_stamp_acc), 64);
  if (reset) {
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(pdt::pdt_lm_stamp_acc), z, 64);
  }
  return 0;
}
#endif

extern "C" int64_t pdt_ctc_lookup_lm_search_workspace_bytes(int64_t n_frames, int64_t N, int64_t V, int64_t width,
                                                            int64_t max_ngram, int64_t U) {
  if (n_frames < 0 || N < 0 || V < 1 || width < 1 || U < V + 1) return 0;
  return (int64_t)pdt::plan_lm_search(n_frames, N, width, V <= 32767 ? 2 : 8, V, U, max_ngram).total;
}

extern "C" int pdt_ctc_lookup_lm_search(
    const float *probs, int64_t p_st, int64_t p_sn, int64_t p_sv, const int64_t *frame_lens, int64_t n_frames,
    int64_t N, int64_t V, int64_t width, const float *logps, const float *logbs, const int32_t *child_start,
    const int32_t *ids, const int32_t *succ_start, const int32_t *succ_tok, const int32_t *succ_node,
    int64_t max_ngram, int64_t U, int64_t sos, float beta, int valid_mixture, int64_t *y, int64_t *y_lens,
    float *nb, float *b, void *workspace, int64_t workspace_bytes, void *stream) {
  using namespace pdt;
  if (N < 0 || V < 1 || width < 1 || n_frames < 1 || max_ngram < 2 || U < V + 1 || U > V + 2) return PDT_E_ARG;
  if (N == 0) return PDT_OK;
  if (!probs || !y || !y_lens || !nb || !b || !logps || !logbs || !child_start || !ids || !succ_start || !succ_tok ||
      !succ_node || !workspace)
    return PDT_E_ARG;
  if (V >= (1 << 30) || n_frames >= (1 << 26) || N >= (1ll << 31)) return PDT_E_TOO_LONG;
  if (width > kMaxWidth || max_ngram > kLmMaxOrder) return PDT_E_TOO_LONG;
  const int hb = V <= 32767 ? 2 : 8;
  const LmSearchPlan p = plan_lm_search(n_frames, N, width, hb, V, U, max_ngram);
  if ((int64_t)p.total > workspace_bytes) return PDT_E_ARG;
  unsigned char *w = reinterpret_cast<unsigned char *>(workspace);
  LmSearchState st[2];
  for (int sd = 0; sd < 2; ++sd) {
    st[sd].nb = reinterpret_cast<float *>(w + p.side[sd][0]);
    st[sd].b = reinterpret_cast<float *>(w + p.side[sd][1]);
    st[sd].last = reinterpret_cast<int64_t *>(w + p.side[sd][2]);
    st[sd].lens = reinterpret_cast<int64_t *>(w + p.side[sd][3]);
    st[sd].isp = reinterpret_cast<uint8_t *>(w + p.side[sd][4]);
    st[sd].slot = reinterpret_cast<int32_t *>(w + p.side[sd][5]);
  }
  hipStream_t hs = (hipStream_t)stream;
  hipLaunchKernelGGL(lm_search_init_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, hs, st[0], (int)N,
                     (int)width);
  CtcLmAdvArgs A{};
  CtcAdvArgs &a = A.s;
  a.N = (int)N; a.V = (int)V; a.W = (int)width;
  A.lm = LmTrie{logps, logbs, child_start, ids, succ_start, succ_tok, succ_node, (int)V, (int)max_ngram, (int)U,
                (int)(U - V - 1), sos};
  A.beta = beta;
  A.valid_mixture = valid_mixture;
  A.frame_lens = frame_lens;
  if (p.cached) {
    A.cache = reinterpret_cast<float *>(w + p.cache);
    A.cache_flag = reinterpret_cast<int32_t *>(w + p.cache_flag);
    A.cache_stride = (int)((V + 31) & ~(int64_t)31);
    hipError_t e = hipMemsetAsync(A.cache_flag, 0, (size_t)U * 4, hs);
    if (e != hipSuccess) return (int)e;
  }
  // histories: 2 W slots of smax tokens per utterance, token-contiguous
  a.y_prev = reinterpret_cast<const int64_t *>(w + p.hist);
  a.y_next = reinterpret_cast<int64_t *>(w + p.hist);
  a.yp_ss = 1; a.yp_sk = p.smax; a.yp_sn = 2 * width * p.smax;
  A.yn_ss = 1; A.yn_sk = p.smax; A.yn_sn = 2 * width * p.smax;
  a.next_src = reinterpret_cast<int64_t *>(w + p.src);
  a.next_nonext = w + p.nonext_flags;
  // one launch for every frame (ctc_lm_search_kernel); PDT_LM_PERSISTENT=0: a launch per frame (comparisons)
  const bool persistent = switches().lm_persistent != 0;
  if (persistent) {
    a.Kp = (int)width;  // (the LDS plan of the widest frame; the first frame's single prefix fits inside it)
    const size_t smem = plan_lm_frame(A);
    if (smem == 0) return PDT_E_TOO_LONG;
    auto kern = hb == 2 ? ctc_lm_search_kernel<int16_t> : ctc_lm_search_kernel<int64_t>;
    if (smem > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)N), dim3(64 * a.waves_per_wg), smem, hs, A, st[0], st[1], probs, p_st,
                       p_sn, p_sv, (int)n_frames);
    const int rc = (int)hipGetLastError();
    if (rc != PDT_OK) return rc;
  } else {
    for (int64_t t = 0; t < n_frames; ++t) {
      const LmSearchState &prev = st[t & 1], &next = st[(t + 1) & 1];
      const int64_t Kp = t == 0 ? 1 : width;
      a.Kp = (int)Kp;
      a.S = (int)t;
      a.nonext = probs + t * p_st; a.ne_sn = p_sn; a.ne_sv = p_sv;
      a.blank = probs + t * p_st + V * p_sv; a.bl_sn = p_sn;
      a.nb_prev = prev.nb; a.pb_sn = width; a.pb_sk = 1;
      a.b_prev = prev.b; a.pbb_sn = width; a.pbb_sk = 1;
      a.last = prev.last; a.la_sn = width; a.la_sk = 1;
      a.lens = prev.lens; a.le_sn = width; a.le_sk = 1;
      a.isp = prev.isp; a.ip_sn = width * width; a.ip_sa = width; a.ip_sb = 1;
      a.y_next_last = next.last; a.y_next_lens = next.lens;
      a.nb_next = next.nb; a.b_next = next.b; a.next_isp = next.isp;
      A.slot_prev = prev.slot; A.slot_next = next.slot;
      A.frame = t;
      const int rc = launch_lm_frame(A, hb, hs);
      if (rc != PDT_OK) return rc;
    }
  }
  const LmSearchState &fin = st[n_frames & 1];
  if (hb == 2)
    hipLaunchKernelGGL(lm_search_gather_kernel<int16_t>, dim3((unsigned)N), dim3(256), 0, hs,
                       reinterpret_cast<const int16_t *>(w + p.hist), 2 * width * p.smax, p.smax, fin, (int)N,
                       (int)width, (int)n_frames, y, y_lens, nb, b);
  else
    hipLaunchKernelGGL(lm_search_gather_kernel<int64_t>, dim3((unsigned)N), dim3(256), 0, hs,
                       reinterpret_cast<const int64_t *>(w + p.hist), 2 * width * p.smax, p.smax, fin, (int)N,
                       (int)width, (int)n_frames, y, y_lens, nb, b);
  return (int)hipGetLastError();
}
