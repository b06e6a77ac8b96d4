// One frame of CTC prefix beam search at wave level (shared by the fused search kernel in
// ctc_search.hip and the step-function kernel in beam_advance.hip).  See ctc_search.hip for
// the design notes.
#pragma once
#include "wave_select.hpp"

namespace pdt {

// Diagnostic build only (-DPDT_STAMPS): per-phase cycle totals of lane 0 of every wave.
#ifdef PDT_STAMPS
__device__ unsigned long long g_stamps[16];
__device__ unsigned long long *g_stamp_lds_dummy;
#define PDT_STAMP(ph)                                                        \
  do {                                                                       \
    const unsigned long long now_ = __builtin_readcyclecounter();           \
    pdt_stamp_acc[ph] += (unsigned)(now_ - stamp_last_);                     \
    stamp_last_ = now_;                                                      \
  } while (0)
#define PDT_STAMP_BEGIN unsigned long long stamp_last_ = __builtin_readcyclecounter()
#else
#define PDT_STAMP(ph) do {} while (0)
#define PDT_STAMP_BEGIN do {} while (0)
#endif

// Diagnostic build only (-DPDT_STATS): event counters (lane 0 of every wave)
#ifdef PDT_STATS
__device__ unsigned long long g_stats[16];
#define PDT_STAT(i) do { if (lane_id() == 0) atomicAdd(&g_stats[i], 1ull); } while (0)
#else
#define PDT_STAT(i) do {} while (0)
#endif

// Diagnostic build only (-DPDT_UTT_STATS): per utterance (first 8192 of a launch), accumulated in the
// consumer wave's registers and stored once at the end -- [0] when its loop ended (16-cycle units
// since the wave started), [1] frames that left the lean tier, [2] short lists it completed,
// [3] time it waited for its producer  (with -DPDT_UTT_REASONS as well: [2] frames with a rounded tie
// among the best K + 1, [3] those whose tie is among masses that underflowed to 0)
#ifdef PDT_UTT_STATS
__device__ unsigned g_utt_stats[8192 * 4];
#define PDT_UTT(k, v) do { pdt_utt_acc[k] += (unsigned)(v); } while (0)
#else
#define PDT_UTT(k, v) do {} while (0)
#endif

constexpr int kMaxWidth = 32;  // K + K' <= 64 tokens fit one per lane

struct CtcArgs {
  const float *logits;  // (T, N, V + 1), element strides below
  int64_t lg_st, lg_sn, lg_sv;
  const int64_t *lens;  // (N,) or null
  int T, N, V, W, S;    // S = rows of y
  int64_t *y;           // (S, N, W) contiguous
  int64_t *y_lens;      // (N, W)
  float *y_probs;       // (N, W)
  int2 *trie;           // (N, T, W) records (parent node, token); node id = t * W + i
  // Checkpoints for the output walk: every 2^ckpt_shift frames each beam entry records
  // (node, length | origin << 24) -- origin = the entry of the previous checkpoint it descends
  // from -- so the prefixes are read off the trie in parallel segments, not one 'T-hop chain.
  int2 *ckpt;           // (N, ckpt_count, W)
  unsigned char *grow;  // rows of the ring for vocabularies beyond the LDS (ctc_search.hip: RingLayout)
  int ckpt_shift, ckpt_count;
  int lds_per_wave, waves_per_wg;
  // PDT_CTC_EXACT_DIV=1: probabilities as the IEEE quotient e / sum (a division per element of
  // the row) instead of e * (1 / sum) -- slower; lets a caller see which disagreements with a
  // reference are the reciprocal's ("Near ties")
  int exact_div;
  // PDT_CTC_LEAN_EXTRA=0: frames the lean tier's mid / exact paths would decide go to the full tiers
  // (same results; the tests compare the two)
  int no_lean_extra;
};

struct Beam {
  float nb, b;
  int last, len, node;
  unsigned isp;  // bit k' set <=> this prefix is a prefix of beam entry k' (beam width <= 32)
  int origin = 0;  // trie form: the beam entry of the last checkpoint this prefix descends from
};

__device__ __forceinline__ u64 readlane_u64(u64 v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float shfl_f(float v, int l) { return __int_as_float(shfl_i(__float_as_int(v), l)); }
// a zero the compiler cannot hoist out of the frame loop (it kept hoisted zero registers live
// across the whole loop, spilled them, and reloaded them from scratch before every LDS clear)
__device__ __forceinline__ unsigned fresh_zero() {
  unsigned z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}

// Extra inputs of the step-function form (ctc_prefix_search_advance): per-prefix extension
// probabilities (language-model fusion) and a dense (S, N, K') history instead of the trie.
struct DenseCtx {
  const float *ext;      // ext_probs_t[n] : element (k, v) at ext[k * ext_sk + v * ext_sv]
  int64_t ext_sk, ext_sv;
  const int64_t *y_prev; // y_prev[:, n, :] : element (s, k) at y_prev[s * yp_ss + k * yp_sk]
  int64_t yp_ss, yp_sk;
  int S;
  int lists_ready;       // the per-prefix lists are in L.tl_tok / L.tl_p already (built by the caller's waves)
  // Instead of whole rows of extension probabilities: the K' x K' entries the frame reads besides
  // the lists -- etab[k * etab_stride + j] = ext[k, clamp(last token of prefix j)] (a caller that
  // forms each row on the fly and keeps only its list: beam_advance.hip, the n-gram fused step)
  const float *etab = nullptr;
  int etab_stride = 0;
  // the same history as 16-bit tokens (a caller that keeps its own, narrower copy between frames)
  const int16_t *y_prev16 = nullptr;
  const int *slot = nullptr;  // history column of old entry k (null: k itself)
  // prefixes that share their language-model context share ONE list and one etab row: list_id[k] =
  // the list of prefix k (LDS, [Kp]; null: prefix k has list k).  etab row r then belongs to list r.
  const int *list_id = nullptr;
  // bit k: no extension of prefix k can be among the frame's winners (the caller has a strict bound:
  // every extension mass of k lies below a lower bound of the K-th winner) -- its extension streams
  // are closed and its list is not read for candidates (it need not have been built)
  unsigned closed = 0u;
  // lpos[r * lpos_stride + j] = position of prefix j's (clamped) last token in list r, -1 if it is
  // not among the list's entries -- built with the lists by a caller that knows the prefixes' last
  // tokens (null: the frame searches the lists, M reads per look-up)
  const int *lpos = nullptr;
  int lpos_stride = 0;
};

// ints per next-token table: W * W, but at least 128 so that the table not in use (nxt_new
// before the is-prefix update writes it) can double as the 64 x u64 scratch of the candidate
// selection
__host__ __device__ inline int nxt_stride(int W) { return W * W < 128 ? 128 : W * W; }
// bytes of a consumer wave's LDS scratch (two `nxt` tables, chm, info, ...: 3 ints per prefix), rounded
// to 8: what follows it is read and written as 64-bit words (odd widths would leave it 4-byte aligned)
__host__ __device__ inline int consumer_scratch_bytes(int W) {
  return (2 * nxt_stride(W) * 4 + (W > 0 ? W : 1) * 4 * 3 + 7) & ~7;
}

// Per-wave LDS scratch of one frame.
struct FrameLds {
  u64 *surv;           // [PDT_SURV_CAP] survivors of the top-M selection
  int *tl_tok;         // [L * 64] sorted token list(s): L = 1 (shared) or Kp (DENSE)
  float *tl_p;         // [L * 64] extension probability of each list entry
  unsigned char *pos;  // [V] list index of a token or 0xFF (shared list only)
  unsigned *chm;       // [max(W, Kp)] new beam entries that descend from old entry j
  int *info;           // [2 * W] packed description of every new beam entry
  int *nxt_old, *nxt_new;  // [nxt_stride(W)] token of prefix b right after prefix a (trie form only)
  // shared list only: entries the list holds (a producer may hand over a SHORT exact top-c list,
  // c < K + K'; ctc_frame completes it when a frame needs more) and the slot header to record that
  int list_len;
  float *hdr;
  int2 *trie_u = nullptr;  // trie form: this utterance's records, trie + n * T * W (wave-uniform)
  // ROWLESS form (ctc_rowreg.hip: the row never leaves the producer's registers): the two
  // probabilities a frame reads besides the list -- this lane's last token's and the blank's
  float pl_in = 0.0f, pblank_in = 0.0f;
  static __host__ __device__ size_t bytes(int V, int W, int Kp, bool dense) {
    const int RS = W > Kp ? W : Kp;
    size_t b = (size_t)PDT_SURV_CAP * 8 + (size_t)(dense ? Kp : 1) * PDT_WAVE * 8;
    b += dense ? 0 : (size_t)((V + 15) & ~15);
    b += (size_t)RS * 4 + (size_t)2 * W * 4 + (dense ? 0 : (size_t)2 * W * W * 4);
    return (b + 15) & ~(size_t)15;
  }
  __device__ void carve(unsigned char *base, int V, int W, int Kp, bool dense) {
    const int RS = W > Kp ? W : Kp, L = dense ? Kp : 1;
    surv = reinterpret_cast<u64 *>(base);
    tl_tok = reinterpret_cast<int *>(surv + PDT_SURV_CAP);
    tl_p = reinterpret_cast<float *>(tl_tok + L * PDT_WAVE);
    chm = reinterpret_cast<unsigned *>(tl_p + L * PDT_WAVE);
    info = reinterpret_cast<int *>(chm + RS);
    nxt_old = info + 2 * W;
    nxt_new = nxt_old + (dense ? 0 : W * W);
    pos = reinterpret_cast<unsigned char *>(nxt_new + (dense ? 0 : W * W));
  }
};

// Number of list entries a frame needs: K + K' (at most K wins + K'-1 merged-away tokens + the
// prefix's own last token per stream), see ctc_search.hip.
__device__ __forceinline__ int ctc_list_len(int V, int W, int Kp) {
  const int K = min(W, Kp * (V + 1));
  return min(V, K + Kp);
}

// Shared sorted token list of one frame from the unnormalised row p[0..V) (inv = 1 / normaliser):
// tl_tok / tl_p (normalised) / pos (inverse index; entries of the previous list must be 0xFF).
template <bool LONG = false>
__device__ __forceinline__ void build_shared_list(const float *p, float inv, int V, int M, u64 *surv,
                                                  int *tl_tok, float *tl_p, unsigned char *pos,
                                                  const unsigned *lmax_in = nullptr,
                                                  unsigned *probe = nullptr, int probe_rank = 1) {
  const int lane = lane_id();
  const u64 tk = wave_top_sorted<LONG, true>(p, V, M, surv, lmax_in, probe, probe_rank);  // p >= 0
  if (lane < M) {
    const int tok = (int)idx_of(tk);
    tl_tok[lane] = tok;
    tl_p[lane] = p[tok] * inv;
    pos[tok] = (unsigned char)lane;
  }
}

// One frame of the search.  `p` holds the (unnormalised) non-extension probabilities of
// v in [0, V] (index V = blank) and `inv` the reciprocal of their normaliser (1 when already
// normalised): probabilities are p * inv, one correctly rounded reciprocal per frame instead
// of a division per use (<= 1 ulp from the quotient, far inside the 1e-5 parity tolerance).
// Kp = number of live lanes (1 at t = 0, then W).
// On return new_src / new_tok / new_kind describe where lane i's new prefix came from
// (kind: 0/1 extension, 2 non-extension, -1 invalid).
#if defined(PDT_STAMPS)
#define PDT_STAMP_PARAM , unsigned *pdt_stamp_acc
#define PDT_STAMP_ARG , pdt_stamp_acc
#elif defined(PDT_UTT_STATS)
#define PDT_STAMP_PARAM , unsigned *pdt_utt_acc
#define PDT_STAMP_ARG , pdt_utt_acc
#else
#define PDT_STAMP_PARAM
#define PDT_STAMP_ARG
#endif

// Returns whether the list as handed over was enough (false: this wave had to complete a short
// list -- the search kernel's producers send short lists only while that stays rare).
// ROWLESS (shared-list form only): there is no row `p` to read -- the list is complete (M entries)
// and FrameLds::pl_in / pblank_in carry the two other probabilities of the frame.
// TRIE: prefix histories are the trie in HBM + the next-token tables in LDS (the one-kernel searches);
// otherwise the dense (S, N, K') history of the step functions.  Per-prefix lists with a trie: the
// search with a bigram model's factor table (ctc_lm_table.hip).
template <bool DENSE, bool ROWLESS = false, bool TRIE = !DENSE>
__device__ __forceinline__ bool ctc_frame(Beam &bm, const float *p, const float inv, const int V,
                                          const int W, const int Kp, const int t,
                                          const int64_t n, const CtcArgs &a, const DenseCtx &dc,
                                          const FrameLds &L, int &new_src, int &new_tok,
                                          int &new_kind PDT_STAMP_PARAM) {
  // (laundered: inside a frame loop nothing derived from the lane index is loop-invariant to the
  // compiler, so it is recomputed where used instead of hoisted, kept live across the whole
  // loop and spilled -- scratch reloads sat on the critical path of every frame)
  int lane = lane_id();
  asm volatile("" : "+v"(lane));
  const bool live = lane < Kp;
  const int K = min(W, Kp * (V + 1));  // _decoding.py:775
  const int M = min(V, K + Kp);
  const int me = live ? lane : 0;
  PDT_STAMP_BEGIN;

  // ---- sorted token list(s): tokens by descending extension probability ----------------
  // shared form: L.tl_tok / L.tl_p / L.pos were filled by build_shared_list (possibly by
  // another wave); dense form: one list per prefix, built here
  if (DENSE && !dc.lists_ready) {
    for (int k = 0; k < Kp; ++k) {
      const u64 tk = wave_top_sorted_strided<true>(dc.ext + k * dc.ext_sk, dc.ext_sv, V, M, L.surv);  // (rows in HBM: eight loads in flight)
      if (lane < M) {
        L.tl_tok[k * PDT_WAVE + lane] = (int)idx_of(tk);
        L.tl_p[k * PDT_WAVE + lane] = fkey_inv(key_of(tk));
      }
      wave_sync();
    }
  }
  PDT_STAMP(1);
  // the list of prefix k (a wave-uniform or per-lane k alike: an LDS read when lists are shared)
  auto list_of = [&](const int k) -> int { return DENSE ? (dc.list_id ? dc.list_id[k] : k) : 0; };
  const int my_list = list_of(me);
  const int *mt = L.tl_tok + my_list * PDT_WAVE;

  // ---- candidate masses that do not depend on the token (:777-794) ----------------------
  static_assert(!(DENSE && ROWLESS), "the row-less form reads the shared list");
  const float p_blank = ROWLESS ? L.pblank_in : p[V] * inv;
  const int lastc = min(max(bm.last, 0), V - 1);
  const float pl = ROWLESS ? L.pl_in : p[lastc] * inv;  // non-extension probability of my last token
  const float e_last = DENSE ? (dc.etab ? dc.etab[my_list * dc.etab_stride + me] : dc.ext[me * dc.ext_sk + lastc * dc.ext_sv]) : pl;
  const float tot = bm.nb + bm.b;
  const float B = tot * p_blank;
  float NB = bm.nb * pl;
  const bool valid_beam = live && tot > -PDT_INF;
  wave_sync();

  // index of my last token in my list (-1: not among its entries), the list entries still
  // available to this prefix (stream 0; my own last token goes to stream 1), and the
  // ---- merge: an extension of kk that equals an existing prefix feeds that prefix --------
  // (:804-837); only prefixes that have descendants in the beam are visited.
  // `c` = entries the list holds.  Run once per frame, and a second time (accumulate = false:
  // the masses are already in `add`) if a short list had to be completed.
  const int c_list = (DENSE || ROWLESS) ? M : min(L.list_len, M);
  const bool full_list = DENSE || ROWLESS || c_list >= M;
  int jl = -1;
  u64 avail = 0ull;
  bool s1_open = valid_beam, s2_open = valid_beam;
  float add = 0.0f;
  auto index_pass = [&](const int c, const bool accumulate) {
    jl = -1;
    if (!DENSE) {
      const int q = L.pos[lastc];
      jl = q == 0xFF ? -1 : q;
    } else if (dc.lpos) {
      jl = dc.lpos[my_list * dc.lpos_stride + me];
    } else {
      for (int j = 0; j < M; ++j) jl = mt[j] == lastc ? j : jl;
    }
    if (!DENSE && M <= 32) {
      // (32 entries at most: the position byte shifts the bit in directly -- 0xFF, "not listed",
      // lands in bit 63 and falls off the low word -- instead of a 64-bit mask, a compare and
      // two selects per frame)
      const unsigned q = L.pos[lastc];
      avail = (c >= 32 ? ~0u : ((1u << c) - 1u)) & ~(unsigned)(1ull << (q & 63u));
    } else {
      avail = c >= 64 ? ~0ull : ((1ull << c) - 1ull);
      if (jl >= 0) avail &= ~(1ull << jl);
    }
    u64 par = __ballot(live && (bm.isp & ~(1u << lane)) != 0u);
    while (par) {
      const int kk = (int)__builtin_ctzll(par);
      par &= par - 1ull;
      const unsigned isp_kk = (unsigned)__builtin_amdgcn_readlane((int)bm.isp, kk);
      const int len_kk = __builtin_amdgcn_readlane(bm.len, kk);
      const bool child = live && ((isp_kk >> lane) & 1u) && (len_kk + 1 == bm.len);
      u64 cm = __ballot(child);
      if (cm == 0ull) continue;
      const int last_kk = min(max(__builtin_amdgcn_readlane(bm.last, kk), 0), V - 1);
      int jc = jl;  // index of my last token in kk's list
      const int list_kk = list_of(kk);
      if (DENSE) {
        jc = -1;
        if (dc.lpos) jc = dc.lpos[list_kk * dc.lpos_stride + me];
        else for (int j = 0; j < M; ++j) jc = L.tl_tok[list_kk * PDT_WAVE + j] == lastc ? j : jc;
      }
      if (accumulate) {
        const float nb_kk = readlane_f(bm.nb, kk), b_kk = readlane_f(bm.b, kk);
        if (child) {
          // to_match = the last token of the child (whose length is len_kk + 1)
          const float w = (lastc == last_kk ? 0.0f : nb_kk) + b_kk;
          const float e = DENSE ? (dc.etab ? dc.etab[list_kk * dc.etab_stride + me] : dc.ext[kk * dc.ext_sk + lastc * dc.ext_sv]) : pl;
          add += w * e;
        }
      }
      u64 rm = 0ull;
      bool close1 = false;
      while (cm) {
        const int c2 = (int)__builtin_ctzll(cm);
        cm &= cm - 1ull;
        const int jcc = __builtin_amdgcn_readlane(jc, c2);
        if (jcc >= 0) rm |= 1ull << jcc;
        close1 = close1 || (__builtin_amdgcn_readlane(lastc, c2) == last_kk);
      }
      if (lane == kk) {
        avail &= ~rm;
        if (close1) s1_open = false;
      }
    }
  };
  index_pass(c_list, true);
  if (DENSE && ((dc.closed >> lane) & 1u)) {
    avail = 0ull;
    s1_open = false;
  }
  NB = NB + add;
  PDT_STAMP(2);

  // (candidate masses are probabilities >= +0: one-instruction keys, see fkey_nonneg)
  // ---- K-way merge of the per-prefix candidate streams ----------------------------------
  // stream 0: available list entries in order, mass (nb + b) * ext[v]
  // stream 1: my last token, mass b * ext[last]                (:784-789)
  // stream 2: not extending, mass NB + B                        (:842-845)
  const float m1 = bm.b * e_last;
  const float m2 = NB + B;
  new_src = 0, new_tok = 0, new_kind = -1;
  float new_mass = -PDT_INF;
  bool selected = false;
  // a lower bound of the K-th winner's key the lean tier leaves behind for the full tiers
  // (0: none): its candidates are a subset of theirs, so its K-th largest cannot exceed theirs
  unsigned tau_hint = 0u;
  {  // scope of the lean tier's per-lane layout values
  // All 64 lanes hold candidates: lane = G * r + k carries, for prefix k, three of its
  // candidates (slots s = 0..2, "entry" e = r + R * s): entries 0 .. 3R-3 are the first
  // available list entries of stream 0 in order, entry 3R-2 is stream 1, entry 3R-1 stream 2.
  // A prefix's resident entries are consumed in order (they are sorted), so a round is just
  // a wave max + clearing the winning slot; only when a prefix uses up ALL its resident
  // stream-0 entries are its slots refilled with the next 3R-2.
  const int G = Kp <= 16 ? 16 : 32;
  const int kb = lane & (G - 1), rr = lane >> (G == 16 ? 4 : 5);  // (a shift, not a division by a runtime G)
  const int ksrc = kb < Kp ? kb : 0;
  const bool kvalid = kb < Kp && (shfl_i((int)valid_beam, ksrc) != 0);
  const float tot_k = shfl_f(tot, ksrc);
  const int lastc_k = shfl_i(lastc, ksrc);

  // Lean tier (K' <= 16, K + K' <= 32): in 94 % of the frames of the bench input every winner
  // is a prefix's BEST available token, its last-token stream or its non-extension, so the four
  // rows hold just those -- main entries 0 and 1, stream 1, stream 2: one candidate per lane --
  // and ONE 64-key sort ranks them all.  If a winner is a main entry 1, the prefix's entry 2
  // (not resident here) could matter: the full tiers below decide the frame instead (5.6 %).
  if (G == 16 && M <= 32) {
    // this row's candidate of prefix kb: rows 0 / 1 read the list, rows 2 / 3 a stream mass
    // A short list (c_list < M) may end before a prefix's entry 0 / 1: the row then holds an
    // UPPER BOUND of the hidden entry (the mass with the list's last probability, key + 1 so it
    // wins ties); if a bound ranks among the winners the frame needs the complete list.
    const unsigned avk = (unsigned)shfl_i((int)(unsigned)avail, ksrc);
    const unsigned av = rr == 1 ? (avk & (avk - 1u)) : avk;
    const bool hidden = !full_list && rr < 2 && av == 0u;
    // (a shuffle moves the SOURCE lane's operand: fetch both streams, then pick by row)
    const float ms1 = shfl_f(m1, ksrc), ms2 = shfl_f(m2, ksrc);
    const int open12 = shfl_i((int)s1_open | ((int)s2_open << 1), ksrc);
    const float ms = rr == 2 ? ms1 : ms2;
    const bool os = ((open12 >> (rr == 2 ? 0 : 1)) & 1) != 0;
    const int j = av ? __builtin_ctz(av) : (hidden ? c_list - 1 : 0);
    const int list_k = list_of(ksrc);
    const int tokj = L.tl_tok[list_k * PDT_WAVE + j];
    const float pj = L.tl_p[list_k * PDT_WAVE + j];
    const bool has = kvalid && (rr < 2 ? (av != 0u || hidden) : os);
    const unsigned keyL = has ? fkey_nonneg(rr < 2 ? tot_k * pj : ms) + (hidden ? 1u : 0u) : 0u;
    const int tokL = rr < 2 ? tokj : lastc_k;
    // ONE 64-key sort of 32-bit keys: the mass key rounded up to a multiple of 64 with the lane
    // (inverted: lowest lane first) in the freed bits -- 3 VALU per stage instead of the 5-6 of
    // a (key, lane) pair, on the critical path of the utterance.  The order of the top K + 1 is
    // exact unless two of them agree in the upper 26 bits; such a frame (and one where a row-1
    // entry or a bound wins) goes to the full tiers.  No candidate = 0 (valid keys are >= 1, so
    // their rounded keys are >= 64).
    PDT_STAMP(7);
    const unsigned kt = has ? (((keyL + 63u) & ~63u) | (63u - (unsigned)lane)) : (63u - (unsigned)lane);
    const unsigned st = wave_sort_desc<unsigned>(kt);
    PDT_STAMP(8);
    int wl = 63 - (int)(st & 63u);
    unsigned wkey = (unsigned)shfl_i((int)keyL, wl);
    const unsigned st_next = (unsigned)shfl_i((int)st, lane + 1);
    bool tie = lane < K && (st >> 6) != 0u && (st >> 6) == (st_next >> 6);
#ifndef PDT_NO_EXACT_LEAN
    // Two of the first K + 1 agree in the upper 26 bits.  Between prefixes that came by equal masses
    // once (two tokens with the same logit in one frame is all it takes) this repeats in every later
    // frame, and a launch ends with its slowest utterance: rank the same 64 candidates again by their
    // EXACT keys with the full tiers' tie order in the low word (lowest flat candidate index of the
    // reference's layout first: extension (k, list position) -- the last-token stream just before
    // the entry at its own position --, every non-extension after every extension) instead of
    // handing the frame over.  Ties among masses that have underflowed to 0 stay with the full tiers.
    if (__builtin_expect(__ballot(tie) != 0ull && __ballot(tie && (st >> 6) <= 1u) == 0ull && !a.no_lean_extra, 0)) {
      const int jl_k = shfl_i(jl, ksrc);
      const unsigned trank = rr < 2 ? (unsigned)kb * 128u + 2u * (unsigned)j + 1u
                                    : (rr == 2 ? (unsigned)kb * 128u + 2u * (unsigned)(jl_k >= 0 ? jl_k : 63) : (unsigned)(32 + kb) * 128u);
      const bool tie_prev = shfl_i((int)tie, lane - 1) != 0;  // this rank ties with the one before it
      // (a pair at ranks K - 1 and K whose bucket goes on at rank K + 1 is a longer run: the K-th winner
      // may be any of them)
      const bool tie_on = lane <= K && (st >> 6) != 0u && (st >> 6) == (st_next >> 6);
      if (__ballot(tie_on && tie_prev) == 0ull) {
        // runs of two (the usual case: a pair of prefixes with equal masses): the rounded sort has
        // everything else in order, so each pair is put in exact order by itself -- three crossbar
        // fetches and a compare instead of a 64-bit sort
        const unsigned wtr = (unsigned)shfl_i((int)trank, wl);
        const int partner = tie ? lane + 1 : (tie_prev ? lane - 1 : lane);
        const unsigned pkey = (unsigned)shfl_i((int)wkey, partner), ptr = (unsigned)shfl_i((int)wtr, partner);
        const int pwl = shfl_i(wl, partner);
        const bool mine_first = wkey > pkey || (wkey == pkey && wtr < ptr);
        const bool take = tie ? !mine_first : (tie_prev && mine_first);
        wl = take ? pwl : wl;
        wkey = take ? pkey : wkey;
      } else {
        const u64 s64 = wave_sort_desc<u64>(pack_key(keyL, (trank << 6) | (unsigned)lane));
        wl = (int)(idx_of(s64) & 63u);
        wkey = key_of(s64);
      }
      tie = false;
    }
#endif
    const int wth = shfl_i(tokL | (hidden ? (int)0x80000000u : 0), wl);
    const int wtok = wth & 0x7fffffff;
    const bool isw = lane < K && wkey != 0u;
    const int rw = wl >> 4;
    const int wid = wth < 0 ? 64 : 0;
    PDT_STAMP(9);
    bool lean_ok = __ballot((isw && wid >= 64) || tie) == 0ull;
#ifdef PDT_UTT_REASONS
    if (__ballot(tie) != 0ull) PDT_UTT(2, 1);
    if (__ballot(tie && (st >> 6) == 1u) != 0ull) PDT_UTT(3, 1);
#endif
    if (__ballot(isw && wid >= 64) == 0ull) {
      // (not when an upper bound ranks among the first K: that is no real candidate.)  The keys
      // of a bucket of the rounded sort lie in (r - 64, r].
      // (from the K-th winner's exact key: a lower bound of the same kind, a little lower)
      const unsigned kk = (unsigned)__builtin_amdgcn_readlane((int)wkey, K - 1);
      tau_hint = kk > 64u ? ((kk + 63u) & ~63u) - 63u : 1u;
    }
    if (lean_ok && __ballot(isw && rw == 1) != 0ull) {
      // A prefix's entry 1 is among the winners: its entry 2 (not resident here) matters only
      // if it, too, beats the K-th winner.  Entries are sorted, so usually it does not: look the
      // one mass up (or bound it by the list's last probability when the list is short) instead
      // of handing the whole frame to the full tiers.
      const unsigned kth = (unsigned)__builtin_amdgcn_readlane((int)wkey, K - 1);  // 0: fewer than K candidates
      const int kw = wl & 15;
      unsigned a2 = (unsigned)shfl_i((int)(unsigned)avail, kw);
      const float tot2 = shfl_f(tot, kw);
      a2 &= a2 - 1u;
      a2 &= a2 - 1u;  // entries 0 and 1 gone
      const int j2 = a2 ? __builtin_ctz(a2) : c_list - 1;
      const float p2 = L.tl_p[list_of(kw) * PDT_WAVE + j2];
      unsigned key2 = 0u;
      if (a2 != 0u) key2 = fkey_nonneg(tot2 * p2);
      else if (!full_list) key2 = fkey_nonneg(tot2 * p2) + 1u;  // upper bound of a hidden entry
      const bool third_wins = isw && rw == 1 && key2 != 0u && key2 >= kth;
      const u64 tw = __ballot(third_wins);
      lean_ok = tw == 0ull;
#ifdef PDT_STATS
      if (!lean_ok) {
        if (__popcll(tw) == 1) PDT_STAT(6); else PDT_STAT(7);
      }
#endif
#ifndef PDT_NO_MID_TIER
      // Mid tier: ONE prefix's deeper entries reach the winners (97 % of the frames that leave the lean
      // tier; a beam that has collapsed onto one prefix does it frame after frame, and the launch ends
      // with its slowest utterance).  Every other prefix is settled by the lean sort -- its third
      // entry loses against the lean K-th winner, and the final K-th can only be larger -- so the
      // frame's winners are the top K of (lean winners, in rank order in lanes 0..15) and (that
      // prefix's next 16 available list entries, in list order in lanes 16..31): two sorted runs, one
      // 5-stage merge of rounded keys.  Anything unusual -- a rounded tie among the first K + 1, the
      // prefix's last entry here still winning while the list has more (or is short) -- goes to the
      // full tiers as before.
      if (__builtin_expect(tw != 0ull && (tw & (tw - 1ull)) == 0ull && !a.no_lean_extra, 0)) {
        const int kw1 = __builtin_amdgcn_readlane(wl, (int)__builtin_ctzll(tw)) & 15;
        unsigned a2 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)avail, kw1);
        a2 &= a2 - 1u;
        a2 &= a2 - 1u;  // entries 0 and 1 are among the lean candidates
        const float tot1 = readlane_f(tot, kw1);
        const int list1 = list_of(kw1);
        const int n_more = __popc(a2);
        // lane j < 32 with bit j of a2 set owns the list entry j; its rank among the set bits is the
        // lane (16 + rank) that takes it: one forward permute (lanes without an entry send to lane 63)
        const unsigned below = a2 & ((1u << (lane & 31)) - 1u);
        const int rank = __popc(below);
        const bool owner = lane < 32 && ((a2 >> (lane & 31)) & 1u) && rank < 16;
        const int jj = __builtin_amdgcn_ds_permute((owner ? 16 + rank : 63) << 2, owner ? lane + 1 : 0) - 1;
        const bool extra = lane >= 16 && lane < 32 && jj >= 0;
        const int jx = extra ? jj : 0;
        const unsigned key_e = extra ? fkey_nonneg(tot1 * L.tl_p[list1 * PDT_WAVE + jx]) : 0u;
        const int tok_e = L.tl_tok[list1 * PDT_WAVE + jx];
        // payload of lane l < 32: the exact key, the token, (source prefix | kind << 8)
        const unsigned pk = lane < 16 ? (isw ? wkey : 0u) : key_e;
        const int ptok = lane < 16 ? wtok : tok_e;
        const int pinfo = lane < 16 ? ((wl & 15) | ((rw == 2 ? 1 : (rw == 3 ? 2 : 0)) << 8)) : kw1;
        const unsigned km = (lane < 32 && pk != 0u) ? (((pk + 63u) & ~63u) | (63u - (unsigned)lane)) : (63u - (unsigned)lane);
        const unsigned sm = sort_groups<32>(lane < 32 ? km : 0u);
        const unsigned sm_next = (unsigned)shfl_i((int)sm, lane + 1);
        const int from = 63 - (int)(sm & 63u);
        const bool win = lane < K && (sm >> 6) != 0u;
        const bool tie_m = lane < K && (sm >> 6) != 0u && (sm >> 6) == (sm_next >> 6);
        // the last entry taken here (16 + min(n_more, 16) - 1) must lose unless the list holds no more
        const int last_lane = 15 + min(n_more, 16);
        const bool open_end = n_more > 16 || !full_list;
        const bool runs_out = win && open_end && from == last_lane;
        if (n_more > 0 && __ballot(tie_m || runs_out) == 0ull) {
          const unsigned k_w = (unsigned)shfl_i((int)pk, from);
          const int t_w = shfl_i(ptok, from), i_w = shfl_i(pinfo, from);
          if (win) {
            new_src = i_w & 255;
            new_tok = t_w;
            new_kind = i_w >> 8;
            new_mass = fkey_nonneg_inv(k_w);
          }
          selected = true;
        }
      }
#endif
    }
    if (lean_ok) {
      if (isw) {
        new_src = wl & 15;
        new_tok = wtok;
        new_kind = rw == 2 ? 1 : (rw == 3 ? 2 : 0);
        new_mass = fkey_nonneg_inv(wkey);
      }
      selected = true;
    }
  }

  }
  bool list_sufficed = true;
  if (__builtin_expect(!selected, 0)) {  // the full tiers own their layout values: nothing of them is live above
  PDT_STAT(5);
  PDT_UTT(1, 1);
  // All 64 lanes hold candidates: lane = G * r + k carries, for prefix k, three of its
  // candidates (slots s = 0..2, "entry" e = r + R * s): entries 0 .. 3R-3 are the first
  // available list entries of stream 0 in order, entry 3R-2 is stream 1, entry 3R-1 stream 2.
  // A prefix's resident entries are consumed in order (they are sorted), so a round is just
  // a wave max + clearing the winning slot; only when a prefix uses up ALL its resident
  // stream-0 entries are its slots refilled with the next 3R-2.
  // (a lambda, expanded twice: on the list as handed over and, if a bound won, on the completed
  // list.  As a loop the second pass made the first one's values loop-carried, and the spills
  // landed in every frame's state update.)
  auto full_tiers = [&](const bool list_done) -> bool {
  // (laundered: otherwise these are the lean tier's values kept alive -- four scalar spills
  // written in every frame for the 3 % that get here)
  int Kp_f = Kp;
  asm volatile("" : "+s"(Kp_f));
  const int G = Kp_f <= 16 ? 16 : 32, GS = Kp_f <= 16 ? 4 : 5, R = PDT_WAVE >> GS;
  const int M = min(V, K + Kp_f);  // (shadows the frame's M for the same reason)
  const int kb = lane & (G - 1), rr = lane >> (G == 16 ? 4 : 5);  // (a shift, not a division by a runtime G)
  const int ksrc = kb < Kp ? kb : 0;
  const bool kvalid = kb < Kp && (shfl_i((int)valid_beam, ksrc) != 0);
  const float tot_k = shfl_f(tot, ksrc);
  const int lastc_k = shfl_i(lastc, ksrc);
  unsigned key0 = 0u, key1 = 0u, key2 = 0u;
  int tk0 = 0, tk1 = 0, tk2 = 0;
  const int n_main = 3 * R - 2;
  // fills the stream-0 slots of every lane whose prefix is in `mask_k` from that prefix's
  // current `avail`
  // While the list is short (list_done = false) the first entry a prefix does NOT have is an
  // UPPER BOUND of whatever the complete list holds there (token -1, the mass with the list's
  // last probability, key + 1): if a bound is ever taken as a winner the list is completed and
  // the tiers run again; otherwise the short list was all this frame needed.
  auto fill_main = [&](bool mine) {
    const int list_k = list_of(ksrc);
    const int *lt = L.tl_tok + list_k * PDT_WAVE;
    const float *lp = L.tl_p + list_k * PDT_WAVE;
    if (M <= 32) {  // the usual case (K + K' <= 32): half the work per bit operation
      unsigned av = (unsigned)shfl_i((int)(unsigned)avail, ksrc);
      const int n_av = __popc(av);
      for (int i = 0; i < rr; ++i) av &= av - 1u;
#pragma unroll
      for (int sl = 0; sl < 3; ++sl) {
        const int e = rr + R * sl;
        unsigned key = 0u;
        int tok = 0;
        if (e < n_main && av != 0u && kvalid) {
          const int j = __builtin_ctz(av);
          tok = lt[j];
          key = fkey_nonneg(tot_k * lp[j]);
        } else if (!DENSE && !list_done && e == n_av && e < n_main && kvalid) {
          tok = -1;
          key = fkey_nonneg(tot_k * lp[c_list - 1]) + 1u;
        }
        if (mine && e < n_main) {
          if (sl == 0) { key0 = key; tk0 = tok; }
          if (sl == 1) { key1 = key; tk1 = tok; }
          if (sl == 2) { key2 = key; tk2 = tok; }
        }
        for (int i = 0; i < R; ++i) av &= av - 1u;
      }
      return;
    }
    u64 av = shfl_u64(avail, ksrc);
    for (int i = 0; i < rr; ++i) av &= av - 1ull;  // skip to entry rr
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      const int e = rr + R * sl;
      unsigned key = 0u;
      int tok = 0;
      if (e < n_main && av != 0ull && kvalid) {
        const int j = (int)__builtin_ctzll(av);
        tok = lt[j];
        key = fkey_nonneg(tot_k * lp[j]);
      }
      if (mine && e < n_main) {
        if (sl == 0) { key0 = key; tk0 = tok; }
        if (sl == 1) { key1 = key; tk1 = tok; }
        if (sl == 2) { key2 = key; tk2 = tok; }
      }
      for (int i = 0; i < R; ++i) av &= av - 1ull;  // next entry of this lane: e + R
    }
  };
  bool bound_won = false;
  {
    fill_main(true);
    // stream 1 / stream 2 live in slot 2 of rows R-2 / R-1
    const float m1_k = shfl_f(m1, ksrc), m2_k = shfl_f(m2, ksrc);
    const bool o1 = shfl_i((int)s1_open, ksrc) != 0, o2 = shfl_i((int)s2_open, ksrc) != 0;
    if (rr == R - 2) { key2 = (kvalid && o1) ? fkey_nonneg(m1_k) : 0u; tk2 = lastc_k; }
    if (rr == R - 1) { key2 = (kvalid && o2) ? fkey_nonneg(m2_k) : 0u; tk2 = lastc_k; }
  }
  // Exact ties (rare; the reference's torch.topk leaves them unspecified): equal masses go to the
  // lowest flat candidate index of the reference's layout -- extension (k, v) at k * V + v, the
  // non-extension of k at K' * V + k.  Rank of a resident candidate among equals: (prefix, list
  // position) for extensions -- the list is (value, token) ordered; the last-token stream ranks
  // just before the entry at its own position -- and every non-extension after every extension.
  // Only evaluated when a tie has been seen.
  auto tie_rank = [&](const int slot, const int tk) -> unsigned {
    const int e = rr + R * slot;
    if (e == n_main + 1) return (unsigned)(32 + kb) * 128u;
    int j = 63;  // (a bound, or a last token outside the list: after every entry)
    if (e == n_main) {
      const int jl_k = shfl_i(jl, ksrc);
      j = jl_k >= 0 ? jl_k : 63;
    } else if (tk >= 0 && e < n_main) {
      if (DENSE) {
        const int *lt = L.tl_tok + list_of(ksrc) * PDT_WAVE;
        for (int q = 0; q < M; ++q) j = lt[q] == tk ? q : j;
      } else {
        const int q = L.pos[tk];
        j = q == 0xFF ? 63 : q;
      }
    }
    return (unsigned)kb * 128u + (unsigned)(2 * j + (e == n_main ? 0 : 1));
  };
  // Fast path: the K winners all at once.  A lane holding a winner has a local maximum >= the
  // K-th best candidate, and at most K lanes do, so tau = K-th largest local maximum bounds
  // the winners from below; the (>= K, usually ~K) resident candidates >= tau are compacted
  // and sorted once.  Ties resolve as in the serial rounds below (lowest lane, then slot).
  // The serial rounds remain for the frames where the result could depend on entries that are
  // not resident: more than 64 survivors, or a winner that is the last resident stream-0
  // entry of its prefix (the rounds would refill that prefix's slots).
  if (!selected) {
    u64 *sel = TRIE ? reinterpret_cast<u64 *>(L.nxt_new) : L.surv;
    unsigned tau = tau_hint;
    if (tau == 0u) {
      const unsigned lk = max(max(key0, key1), key2);
      const unsigned slk = wave_sort_desc<unsigned>(lk);
      tau = max((unsigned)__builtin_amdgcn_readlane((int)slk, K - 1), 1u);
    }
    const bool p0 = key0 >= tau, p1 = key1 >= tau, p2 = key2 >= tau;
    const u64 b0 = __ballot(p0), b1 = __ballot(p1), b2 = __ballot(p2);
    const int c0 = __popcll(b0), c1 = __popcll(b1), c2 = __popcll(b2);
    const int count = c0 + c1 + c2;
    if (count <= PDT_WAVE) {
      auto below = [&](u64 b) {
        return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
      };
      // low word of a packed key: [tie rank <<] (lane, slot) locator.  Ranking runs on the bare
      // locators; only if two of the first K + 1 survivors have EQUAL masses is it repeated with
      // tie ranks (see tie_rank: lowest flat candidate index first, as the oracle orders them)
      auto rank_survivors = [&](const bool with_ties) -> u64 {
        sel[lane] = (u64)fresh_zero();
        wave_sync();
        const unsigned r0 = with_ties ? tie_rank(0, tk0) << 8 : 0u, r1 = with_ties ? tie_rank(1, tk1) << 8 : 0u,
                       r2 = with_ties ? tie_rank(2, tk2) << 8 : 0u;
        if (p0) sel[below(b0)] = pack_key(key0, r0 | (unsigned)(lane * 4 + 0));
        if (p1) sel[c0 + below(b1)] = pack_key(key1, r1 | (unsigned)(lane * 4 + 1));
        if (p2) sel[c0 + c1 + below(b2)] = pack_key(key2, r2 | (unsigned)(lane * 4 + 2));
        wave_sync();
        // rank by counting: the packed keys are distinct, so the number of larger survivors is
        // a survivor's position in the sorted order (2 VALU per survivor instead of a 21-stage
        // bitonic network; the LDS reads are same-address broadcasts)
        const u64 mine = sel[lane];
        int rank = 0;
        for (int j = 0; j < count; j += 4) {
          const u64 o0 = sel[j], o1 = sel[j + 1], o2 = sel[j + 2], o3 = sel[j + 3];
          rank += (o0 > mine) + (o1 > mine) + (o2 > mine) + (o3 > mine);
        }
        wave_sync();
        if (lane < count && rank <= K) sel[rank] = mine;  // (one more than the winners: the tie check)
        wave_sync();
        const u64 top = (lane <= K && lane < count) ? sel[lane] : 0ull;
        wave_sync();  // sel is nxt_new: the is-prefix update below writes it
        return top;
      };
      u64 s = rank_survivors(false);
      {
        const unsigned k_next = (unsigned)__builtin_amdgcn_mov_dpp((int)key_of(s), 0x130, 0xf, 0xf, true);  // wave_shl:1
        // (masses that have underflowed to 0 all tie: nothing meaningful is left to order there)
        if (__ballot(lane < K && key_of(s) > 1u && key_of(s) == k_next) != 0ull) s = rank_survivors(true);
      }
      if (lane >= K) s = 0ull;
      const unsigned wkey = key_of(s);
      const bool isw = lane < K && wkey != 0u;
      const int id = isw ? (int)(idx_of(s) & 255u) : 0;
      const int wl = id >> 2, sw = id & 3;
      const int e = (wl >> GS) + R * sw;
      if (__ballot(isw && e == n_main - 1) == 0ull) {
        const int t0 = shfl_i(tk0, wl), t1 = shfl_i(tk1, wl), t2 = shfl_i(tk2, wl);
        const int tw = sw == 0 ? t0 : (sw == 1 ? t1 : t2);
        if (__ballot(isw && tw < 0) != 0ull) {
          bound_won = true;
        } else {
          if (isw) {
            new_src = wl & (G - 1);
            new_tok = tw;
            new_kind = e == n_main ? 1 : (e == n_main + 1 ? 2 : 0);
            new_mass = fkey_nonneg_inv(wkey);
          }
          selected = true;
        }
      }
    }
  }
  if (!selected && !bound_won)
  for (int i = 0; i < K; ++i) {
    const unsigned lk = max(max(key0, key1), key2);
    const unsigned mx = wave_max_u32(lk);
    if (mx == 0u) break;  // fewer valid candidates than K: the rest stay invalid (:902-924)
    const int sw_l = key0 == mx ? 0 : (key1 == mx ? 1 : 2);
    const int tw_l = key0 == mx ? tk0 : (key1 == mx ? tk1 : tk2);
    const u64 at_max = __ballot(lk == mx);
    int win = (int)__builtin_ctzll(at_max);
    if ((at_max & (at_max - 1ull)) && mx > 1u) {  // several lanes hold the (non-zero) maximum: lowest flat index first
      const unsigned inv = lk == mx ? ~((tie_rank(sw_l, tw_l) << 6) | (unsigned)lane) : 0u;
      win = (int)(~wave_max_u32(inv) & 63u);
    }
    const int sw = __builtin_amdgcn_readlane(sw_l, win);
    const int wtok = __builtin_amdgcn_readlane(tw_l, win);
    if (wtok < 0) {  // a bound: the short list is not enough for this frame
      bound_won = true;
      break;
    }
    const int wbeam = win & (G - 1);
    const int e = (win >> GS) + R * sw;
    const int wkind = e == n_main ? 1 : (e == n_main + 1 ? 2 : 0);
    const bool rec = lane == i;
    new_src = rec ? wbeam : new_src;
    new_tok = rec ? wtok : new_tok;
    new_kind = rec ? wkind : new_kind;
    new_mass = rec ? fkey_nonneg_inv(mx) : new_mass;
    const bool me_win = lane == win;
    key0 = (me_win & (sw == 0)) ? 0u : key0;
    key1 = (me_win & (sw == 1)) ? 0u : key1;
    key2 = (me_win & (sw == 2)) ? 0u : key2;
    if (e == n_main - 1) {
      // the prefix has used all its resident stream-0 entries: drop them from `avail` and
      // bring in the next ones (rare: one prefix taking more than 3R-2 of the K winners)
      if (lane == wbeam)
        for (int q = 0; q < n_main; ++q) avail &= avail - 1ull;
      fill_main(kb == wbeam);
    }
  }
  return bound_won;
  };
  if (full_tiers(full_list)) {
    if constexpr (!ROWLESS) {
    list_sufficed = false;
    // complete the list here (the short list is a prefix of the complete one, so the entries
    // already indexed stay valid), redo the bookkeeping that depends on list positions, and
    // run the tiers again from a clean slate
    PDT_STAT(4);
#ifndef PDT_UTT_REASONS
    PDT_UTT(2, 1);
#endif
    build_shared_list<false>(p, inv, V, M, reinterpret_cast<u64 *>(L.nxt_new), L.tl_tok, L.tl_p, L.pos);
    if (lane == 0) L.hdr[2] = __int_as_float(M);
    wave_sync();
    index_pass(M, false);
    selected = false;
    new_src = 0, new_tok = 0, new_kind = -1;
    new_mass = -PDT_INF;
    full_tiers(true);
    }
  }
  }
  PDT_STAMP(3);
  // ---- new beam state of lane i (:868-880) ---------------------------------------------
  const int srcl = new_kind >= 0 ? new_src : lane;
  const float NB_s = shfl_f(NB, srcl), B_s = shfl_f(B, srcl);
  const int last_s = shfl_i(lastc, srcl), len_s = shfl_i(bm.len, srcl), node_s = shfl_i(bm.node, srcl);
  const unsigned isp_s = (unsigned)shfl_i((int)bm.isp, srcl);
  const bool is_ext = new_kind == 0 || new_kind == 1;
  const bool is_valid = new_kind >= 0;
  Beam nw;
  nw.nb = !is_valid ? -PDT_INF : (is_ext ? new_mass : NB_s);
  nw.b = !is_valid ? -PDT_INF : (is_ext ? 0.0f : B_s);
  nw.last = !is_valid ? 0 : (is_ext ? new_tok : last_s);
  nw.len = !is_valid ? 0 : len_s + (is_ext ? 1 : 0);
  nw.node = !is_valid ? -1 : (is_ext ? t * W + lane : node_s);
  if (TRIE) nw.origin = shfl_i(bm.origin, srcl);
  if (TRIE && is_valid && is_ext)
    // (uniform base + 32-bit byte offset: T * W * 8 < 2^32 is checked on the host.  The full
    // 64-bit index was ~20 scalar instructions and five reloads of spilled scalars per frame.)
    *reinterpret_cast<int2 *>(reinterpret_cast<char *>(L.trie_u) + (unsigned)(t * W + lane) * 8u) = make_int2(node_s, new_tok);

  PDT_STAMP(4);
  // ---- is-prefix relation and next-token table of the new beam (:883-898) ---------------
  // Only pairs (a, b) whose sources were related need work: chm[j] = new entries descending
  // from old entry j, so lane a visits  U_{j in isp(src_a)} chm[j]  (usually 1-3 entries).
  // nxt[a * W + b] = token of prefix b at position len(a), defined when a is a strict prefix.
  const int RS = W > Kp ? W : Kp;
  if (lane < RS) L.chm[lane] = fresh_zero();
  wave_sync();
  if (is_valid) {
    atomicOr(&L.chm[new_src], 1u << lane);
    L.info[2 * lane] = new_tok;
    L.info[2 * lane + 1] = len_s | (new_src << 20) | ((is_ext ? 1 : 0) << 28);
  }
  wave_sync();
  PDT_STAMP(10);
  unsigned isp_new = 0u;
  bool need_walk = false;
  bool pairs_done = false;
#ifndef PDT_NO_PAIR_PREFIX
#ifndef PDT_PAIR_DENSE
#define PDT_PAIR_DENSE 8  // old entries a source is a prefix of, beyond which all pairs are evaluated
#endif
  // Dense relations (blank-dominated rows: short prefixes are prefixes of most of the beam) make the
  // per-descendant loops below run 10-15 times a frame.  Then ALL pairs (a, b) are evaluated instead,
  // four per lane -- lane = a + 16 q takes b = 4 r + q in round r -- with the same tests, the masks
  // assembled from four ballots: ~140 instructions whatever the density (the loops: ~20 per
  // descendant of the busiest lane).  Widths up to 16, trie histories.
  if (__builtin_expect(TRIE && W <= 16 && !a.no_lean_extra && __ballot(is_valid && __popc(isp_s) > PDT_PAIR_DENSE) != 0ull, 0)) {
    const int pa = lane & 15, pq = lane >> 4;
    const u64 vmask = __ballot(is_valid);
    const int a_tok = shfl_i(new_tok, pa);
    const int a_w = shfl_i(len_s | (new_src << 20) | ((is_ext ? 1 : 0) << 28) | ((is_valid ? 1 : 0) << 29), pa);
    const unsigned a_isp = (unsigned)shfl_i((int)isp_s, pa);
    const int a_len_s = a_w & 0xFFFFF, a_src = (a_w >> 20) & 0xFF;
    const bool a_ext = (a_w >> 28) & 1, a_valid = (a_w >> 29) & 1;
    const int a_len = a_len_s + (a_ext ? 1 : 0);
    unsigned bits = 0u, walks = 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int b = 4 * r + pq;
      const int tok_b = L.info[2 * b], w1 = L.info[2 * b + 1];  // (entries of invalid b are stale: vmask)
      const int lenB = w1 & 0xFFFFF, src_b = (w1 >> 20) & 0xFF;
      const bool ext_b = (w1 >> 28) & 1;
      const int len_b = lenB + (ext_b ? 1 : 0);
      bool pref = a_valid && ((vmask >> b) & 1ull) && b != pa && ((a_isp >> src_b) & 1u) && !(a_len > len_b);
      int tok_at = -1;
      if (pref) tok_at = lenB > a_len_s ? L.nxt_old[a_src * W + src_b] : (ext_b ? tok_b : -1);
      pref = pref && !(a_ext && tok_at != a_tok);
      bool walk = false;
      if (pref && a_len < len_b) {  // strict prefix: the token that follows a inside b
        int nx;
        if (!a_ext) {
          nx = tok_at;
        } else if (lenB == a_len_s + 1) {
          nx = tok_b;
        } else {
          nx = -(2 + b);
          walk = true;
        }
        L.nxt_new[pa * W + b] = nx;
      }
      const u64 m = __ballot(pref), wm = __ballot(walk);
      const unsigned lo = (unsigned)m >> pa, hi = (unsigned)(m >> 32) >> pa;
      bits |= ((lo & 1u) | ((lo >> 15) & 2u) | ((hi & 1u) << 2) | ((hi >> 13) & 8u)) << (4 * r);
      walks |= (unsigned)(wm | (wm >> 16) | (wm >> 32) | (wm >> 48));
    }
    if (is_valid) {  // (lanes 0 .. W - 1: pa is the lane itself)
      isp_new = bits | (1u << lane);
      need_walk = (walks >> lane) & 1u;
    }
    pairs_done = true;
    wave_sync();
  }
#endif
  if (is_valid && !pairs_done) {
    unsigned cand = 0u;
    for (unsigned m = isp_s; m; m &= m - 1u) cand |= L.chm[__builtin_ctz(m)];
    // (an entry is a prefix of itself and never its own strict prefix: one iteration less for every
    // lane, i.e. for the wave -- ~26 instructions of the ~300 a lean frame takes)
    cand &= ~(1u << lane);
    isp_new = 1u << lane;
    while (cand) {
      const int b = __builtin_ctz(cand);
      cand &= cand - 1u;
      const int tok_b = L.info[2 * b], w1 = L.info[2 * b + 1];
      const int lenB = w1 & 0xFFFFF, src_b = (w1 >> 20) & 0xFF;
      const bool ext_b = (w1 >> 28) & 1;
      const int len_b = lenB + (ext_b ? 1 : 0);
      if (nw.len > len_b) continue;
      int tok_at;  // token of new prefix b at position len_s (the length of my source prefix)
      if (lenB > len_s)
        tok_at = !TRIE ? (dc.y_prev16 ? (int)dc.y_prev16[(int64_t)len_s * dc.yp_ss + (dc.slot ? dc.slot[src_b] : src_b) * dc.yp_sk]
                                      : (int)dc.y_prev[(int64_t)len_s * dc.yp_ss + (dc.slot ? dc.slot[src_b] : src_b) * dc.yp_sk])
                       : L.nxt_old[new_src * W + src_b];
      else
        tok_at = ext_b ? tok_b : -1;  // lenB == len_s
      if (is_ext && tok_at != new_tok) continue;
      isp_new |= 1u << b;
      if (TRIE && nw.len < len_b) {  // strict prefix: the token that follows me inside b
        int nx;
        if (!is_ext) {
          nx = tok_at;
        } else if (lenB == len_s + 1) {
          nx = tok_b;  // b = (my new prefix) + tok_b
        } else {
          nx = -(2 + b);  // deeper than the table reaches: resolved below by a trie walk
          need_walk = true;
        }
        L.nxt_new[lane * W + b] = nx;
      }
    }
  }
  if (TRIE && __ballot(need_walk)) {
    // rare: a re-created intermediate prefix.  Token of b at position nw.len = token of the
    // ancestor of b's source node at depth nw.len + 1.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    for (int b = 0; b < K; ++b) {
      const int node_b = __builtin_amdgcn_readlane(node_s, b);
      const int lenB = __builtin_amdgcn_readlane(len_s, b);
      if (need_walk && ((isp_new >> b) & 1u) && L.nxt_new[lane * W + b] == -(2 + b)) {
        int node = node_b, depth = lenB, tok = -1;
        while (node >= 0) {
          const int2 *rec = a.trie + ((int64_t)n * a.T * W + node);
          const int par_ = __hip_atomic_load(&rec->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tok = __hip_atomic_load(&rec->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (depth == nw.len + 1) break;
          node = par_;
          --depth;
        }
        L.nxt_new[lane * W + b] = tok;
      }
    }
  }
  nw.isp = isp_new;
  bm = nw;
  wave_sync();
  PDT_STAMP(5);
  return list_sufficed;
}

// Output walk of the one-kernel searches, second step: the (C + 1) x W segments between consecutive
// checkpoints (tab[c * W + k] = (node, length) of prefix k's ancestor at checkpoint c; the last row
// is the final beam) are read off the trie by all 64 lanes.  A segment is a chain of DEPENDENT loads,
// one per token, and every utterance of a launch gets here at about the same time with nothing else
// left to overlap: a lane therefore walks JW segments AT ONCE -- JW loads in flight per lane instead
// of one (the walk was 9 % of the headline launch: 160 load latencies per lane; now ~40).
template <int JW = 4>
__device__ __forceinline__ void walk_trie_segments(const int2 *tab, const int nseg, const int W, const int2 *trie_u,
                                                   int64_t *y_n, const int64_t row_stride) {
  const int lane = lane_id();
  for (int base = 0; base < nseg; base += JW * PDT_WAVE) {
    int node[JW], ps[JW], stop[JW], col[JW];
#pragma unroll
    for (int j = 0; j < JW; ++j) {
      const int sg = base + j * PDT_WAVE + lane;
      node[j] = -1, ps[j] = -1, stop[j] = 0, col[j] = 0;
      if (sg < nseg) {
        const int c = sg / W;
        col[j] = sg - c * W;
        const int2 top = tab[sg];
        stop[j] = c > 0 ? tab[sg - W].y : 0;
        node[j] = top.x;
        ps[j] = top.y - 1;
      }
    }
    bool any = true;
    while (any) {
      any = false;
      u64 rec[JW];
      bool act[JW];
#pragma unroll
      for (int j = 0; j < JW; ++j) {
        act[j] = ps[j] >= stop[j] && node[j] >= 0;
        rec[j] = 0ull;
        // (a record = (parent, token), written by this wave earlier in the launch and made visible by
        // the caller's fences: one ordinary 8-byte load -- the prefixes of a beam share most of their
        // ancestors, so the L1 serves most of them)
        if (act[j]) rec[j] = *reinterpret_cast<const u64 *>(trie_u + node[j]);
      }
#pragma unroll
      for (int j = 0; j < JW; ++j) {
        if (act[j]) {
          y_n[(int64_t)ps[j] * row_stride + col[j]] = (int64_t)(int)(unsigned)(rec[j] >> 32);
          node[j] = (int)(unsigned)rec[j];
          --ps[j];
          any = true;
        }
      }
    }
  }
}

}  // namespace pdt
