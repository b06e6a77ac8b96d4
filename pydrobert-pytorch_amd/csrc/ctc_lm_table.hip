// CTC prefix beam search with a BIGRAM language model in the loop, the whole search in one launch.
//
// Replaces CTCPrefixSearch.forward with shallow fusion / valid mixture (reference
// _decoding.py:1064-1202; the mix :1113-1135) when the model's factor of the mix depends on the
// prefix's last token only -- a bigram LookupLanguageModel (_lm.py:403-515): then
//     shallow fusion   ext[k, v] = p[v] * F[c_k][v],   F[c] = exp(beta * log_softmax(lm(. | c)))
//     valid mixture    ext[k, v] = (1 - beta) * p[v] + beta * (G[c_k][v] * (1 - p_blank)),  G[c] = softmax(lm(. | c))
// with c_k the last token of prefix k (the start-of-sequence row for the empty prefix).  The rows
// F / G are a (contexts, V) table built ONCE per model and mix (fusion_ext.hip: pdt_lm_factor_table,
// the expressions of pdt_fusion_ext) -- nothing is published between workgroups inside the search.
//
// A workgroup is one utterance: a consumer wave and three worker waves.
//   * Acoustic rows: a worker loads frame t's logits into registers, forms the softmax there (the
//     arithmetic of the searches without a model: per-lane strided sums, DPP reduction, reciprocal
//     with one Newton step) and leaves the NORMALISED row in a three-slot LDS ring, two frames ahead.
//   * Per frame, the consumer publishes the beam's distinct contexts (prefixes that share their last
//     token share everything below); the four waves take the contexts in turn: read the context's
//     factor row (L2: the table is a few MB), mix it with the LDS row, select the sorted list of
//     the K + K' best tokens (threshold = M-th largest per-lane maximum, compaction, one sort) and
//     the etab row (the mixed probability at every prefix's last token).
//   * The consumer then runs the frame routine of ctc_frame.hpp on the per-context lists with the
//     beam in its registers, prefix histories in the trie (no history copies, no state through
//     memory), and publishes the next frame's contexts.
// Per frame and utterance: one frame routine + ceil(D / 4) list selections + one row pass off the
// critical path, against K' row-sized selections on one wave plus the beam's round trip through a
// workspace in ctc_lm_step.hip (which keeps serving n-gram orders above two and tables that do not
// fit).
#include "ctc_ring.hpp"
#include "switches.hpp"

namespace pdt {

struct LmTabArgs {
  CtcArgs c;
  const float *factors;  // (contexts, V) rows, f_stride floats apart
  const float *fmax;     // (contexts,) the largest factor of every row
  int64_t f_stride;
  int sos_row;           // the row of the empty prefix's context
  int contexts;
  // contexts of more than one token (n-gram orders above two): a prefix's row is its last order - 1 tokens
  // read as digits in base ctx_base (start-of-sequence padding in front), so an extension by token v moves
  // row r to (r mod ctx_mod) * ctx_base + v, ctx_mod = ctx_base^(order - 2)  (a bigram model: base U, mod 1)
  int ctx_base, ctx_mod;
  float beta;
  int valid_mixture;
};

struct LmTabLayout {
  int row_floats;  // V + 1 padded to 4
  int rows_bytes;  // three ring slots
  int fmax_floats; // the contexts' largest factors, staged once (0: read from the table's side array)
  int utt_bytes;
};

constexpr int kLmTabRows = 3, kLmTabWaves = 4;

// Diagnostic build only (-DPDT_LMTAB_STAMPS): shader cycles of the consumer wave by segment, summed over
// frames and utterances (profiles/tools/stamps_lm_table.py): 0 own lists, 1 waiting for the workers'
// lists, 2 the frame routine, 3 publishing the next frame's contexts (its wait for the row included)
#ifdef PDT_LMTAB_STAMPS
__device__ unsigned long long g_lmtab_stamps[8];
#define LMTAB_STAMP(i)                                            \
  do {                                                            \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    acc_[i] += now_ - last_;                                      \
    last_ = now_;                                                 \
  } while (0)
// inside the list builder (every wave; summed): 4 waiting for contexts / the row, 5 etab + the mixed
// row (factor row from L2), 6 threshold + survivors, 7 sort, list, positions
#ifdef PDT_LMTAB_LSTAMPS  // (an atomic per stamp, wave and frame: slows the search several times over)
#define LMTAB_LSTAMP(i)                                                      \
  do {                                                                       \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();            \
    if (lane_id() == 0) atomicAdd(&g_lmtab_stamps[i], now_ - llast_);        \
    llast_ = now_;                                                           \
  } while (0)
#define LMTAB_LSTAMP_BEGIN unsigned long long llast_ = __builtin_amdgcn_s_memtime()
#else
#define LMTAB_LSTAMP(i) do {} while (0)
#define LMTAB_LSTAMP_BEGIN do {} while (0)
#endif
#else
#define LMTAB_STAMP(i) do {} while (0)
#define LMTAB_LSTAMP(i) do {} while (0)
#define LMTAB_LSTAMP_BEGIN do {} while (0)
#endif

// ints of the per-frame tables between the lists and the consumer's scratch, padded to 16 bytes: the
// scratch behind them is read and written as 64-bit words (odd widths would leave it 4-byte aligned)
// (etab, lpos: W x W each; ctx_tok, list_id, lastc_pub, ctx_lead, build_ctx: W each; fpair: W x W)
__host__ __device__ inline int lmtab_small_ints(int W) { return (3 * W * W + 5 * W + 3) & ~3; }

__host__ __device__ inline LmTabLayout lmtab_layout(int V, int W, int contexts) {
  LmTabLayout l;
  l.fmax_floats = 0;  // (the rows' largest factors are fetched with the pairs, after every frame: no staging)
  l.row_floats = (V + 1 + 3) & ~3;
  l.rows_bytes = l.row_floats * 4 * kLmTabRows;
  const int lists = W * PDT_WAVE * 8, small = lmtab_small_ints(W) * 4;
  const int consumer = consumer_scratch_bytes(W);
  l.utt_bytes = (l.rows_bytes + lists + small + consumer + kLmTabWaves * PDT_SURV_CAP * 8 + 128 + l.fmax_floats * 4 + 15) & ~15;  // (128: flags, row statistics)
  return l;
}

// NR: 64-element chunks a row takes in registers (V + 1 <= 64 * NR)
// (WC > 0: the beam width as a compile-time constant, as in ctc_search.hip -- the default 16)
template <int NR, int WC = -1>
__global__ void __launch_bounds__(256, NR <= 16 ? 4 : 2)
ctc_lm_table_kernel(const LmTabArgs A, const LmTabLayout ly) {
  extern __shared__ __align__(16) unsigned char smem[];
  const CtcArgs &a = A.c;
  const int lane = lane_id();
  // (the consumers of a CU's four workgroups sit on four different SIMDs: measured, profiles/tools/lm_hwid.py)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t n = (int64_t)xcd_remap(blockIdx.x, gridDim.x);
  const int V = a.V, W = WC > 0 ? WC : a.W;
  float *rows = reinterpret_cast<float *>(smem);
  int *tl_tok = reinterpret_cast<int *>(smem + ly.rows_bytes);       // [W lists x 64]
  float *tl_p = reinterpret_cast<float *>(tl_tok + W * PDT_WAVE);     // [W lists x 64]
  float *etab = tl_p + W * PDT_WAVE;                                  // [W lists x W prefixes]
  int *lpos = reinterpret_cast<int *>(etab + W * W);                  // [W lists x W prefixes] see DenseCtx::lpos
  int *ctx_tok = lpos + W * W;                                        // [W] the frame's contexts: factor row | open << 30
  int *list_id = ctx_tok + W;                                         // [W] the list of every prefix
  int *lastc_pub = list_id + W;                                       // [W] clamped last token of every prefix
  int *ctx_lead = lastc_pub + W;                                      // [W] the first prefix of every context
  int *build_ctx = ctx_lead + W;                                      // [W] the contexts that need a list, in order
  // fpair[k * W + j] = the factor of prefix k's context at prefix j's last token: what the etab rows are
  // mixed from, fetched from the table by the CONSUMER as soon as a frame has decided the new prefixes
  // -- the trip to the table's cache (the streaming logits evict it from the L2: ~3 000 cycles) then runs
  // under the rest of the frame and publish() instead of standing in every frame's chain
  float *fpair = reinterpret_cast<float *>(build_ctx + W);            // [W x W]
  unsigned char *cs = reinterpret_cast<unsigned char *>(etab) + lmtab_small_ints(W) * 4;  // consumer scratch (16-byte aligned)
  u64 *surv0 = reinterpret_cast<u64 *>(cs + consumer_scratch_bytes(W));
  int *flags = reinterpret_cast<int *>(surv0 + kLmTabWaves * PDT_SURV_CAP);
  int *row_ready = flags;      // [3] frame + 1 held by a ring slot
  int *ctx_pub = flags + 3;    // frames whose contexts are published
  int *ctx_count = flags + 4;  // contexts of the published frame (0: the beam is dead)
  int *kp_pub = flags + 5;     // prefixes of the published frame
  int *done = flags + 6;       // [4] frame + 1 up to which a wave's lists are finished (frames WITH lists only)
  // [3 x 3] of a slot's row: the largest token probability, the second largest, the token of the largest
  float *row_stat = reinterpret_cast<float *>(flags + 16);
  int *build_count = flags + 13;  // frame << 6 | lists the published frame needs (0 in nine frames of ten: the workers stay out)
  const int Tn = min(a.S, a.lens ? (int)min((int64_t)a.T, max((int64_t)0, a.lens[n])) : a.T);
  const float keep = 1.0f - A.beta;
  constexpr int kCtxRow = (1 << 30) - 1;

  if (wave == 0 && lane < 32) __hip_atomic_store(&flags[lane], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  __syncthreads();  // flags initialised (the only workgroup barrier)

  auto ld_flag = [&](int *f) { return __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto st_flag = [&](int *f, int v) {
    wave_sync();
    if (lane_id() == 0) __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto wait_above = [&](int *f, int v) {
    while (ld_flag(f) <= v) __builtin_amdgcn_s_sleep(1);
  };

  // ---- a frame's softmax, normalised, into its ring slot (:1093-1095) ------------------------
  auto produce_row = [&](const int t) {
    int lp = lane;
    asm volatile("" : "+v"(lp));
    const float *row = a.logits + (int64_t)t * a.lg_st + n * a.lg_sn + (int64_t)lp * a.lg_sv;
    float x[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (i * PDT_WAVE <= V) x[i] = lp + i * PDT_WAVE <= V ? row[(int64_t)(i * PDT_WAVE) * a.lg_sv] : -PDT_INF;
    float mx = -PDT_INF;
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if (i * PDT_WAVE <= V) mx = fmax_raw(mx, x[i]);
    mx = wave_max_f(mx);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (i * PDT_WAVE <= V) {
        const bool in = lp + i * PDT_WAVE <= V;
        x[i] = in ? exp_nonpos(x[i] - mx) : 0.0f;
        s += x[i];  // (lanes beyond the row add +0: no change)
      }
    }
    s = wave_sum_f(s);
    const float inv0 = __builtin_amdgcn_rcpf(s);
    const float inv = __builtin_fmaf(__builtin_fmaf(-s, inv0, 1.0f), inv0, inv0);
    float *p = rows + (t % kLmTabRows) * ly.row_floats;
    // the two largest token probabilities AS STORED and the token of the largest (the bound of publish()
    // rests on them): per lane, then over the wave -- the runner-up is the best of the other lanes' maxima
    // and the winning lane's own second (equal maxima: the runner-up equals the maximum)
    float p1 = 0.0f, p2 = 0.0f;
    int v1 = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (i * PDT_WAVE <= V && lp + i * PDT_WAVE <= V) {
        const float pv = a.exact_div ? x[i] / s : x[i] * inv;
        p[lp + i * PDT_WAVE] = pv;
        if (lp + i * PDT_WAVE < V) {
          const bool better = pv > p1;
          p2 = better ? p1 : fmax_raw(p2, pv);
          v1 = better ? lp + i * PDT_WAVE : v1;
          p1 = better ? pv : p1;
        }
      }
    }
    const float w1 = wave_max_f(p1);
    const u64 at = __ballot(p1 == w1);
    const int wl = (int)__builtin_ctzll(at);  // (the lowest lane holding the maximum)
    const float w2 = wave_max_f(lp == wl ? p2 : p1);
    const int wv1 = __builtin_amdgcn_readlane(v1, wl);
    if (lp == 0) {
      float *st = row_stat + (t % kLmTabRows) * 3;
      st[0] = w1;
      st[1] = w2;
      st[2] = __int_as_float(wv1);
    }
    st_flag(&row_ready[t % kLmTabRows], t + 1);
  };

  // ---- one list: the sorted top-M tokens of context row `frow` mixed with the frame's row `p`, into `slot` --
  // (returns the lane's list token, -1 beyond the list)
  auto make_list = [&](const float *p, const float *frow, const int M, const int slot, u64 *surv) __attribute__((always_inline)) -> int {
    int lp = lane;
    asm volatile("" : "+v"(lp));
    const float scale = 1.0f - p[V];  // (valid mixture: the mass the blank leaves)
    auto mix = [&](const float pv, const float fv) {
      return A.valid_mixture ? keep * pv + A.beta * (fv * scale) : pv * fv;
    };
    unsigned key[NR];
    unsigned lmax = 0u;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (i * PDT_WAVE < V) {
        const int v = lp + i * PDT_WAVE;
        key[i] = v < V ? fkey_nonneg(mix(p[v], frow[v])) : 0u;
        lmax = max(lmax, key[i]);
      }
    }
    // sorted top-M of the mixed row (wave_top_sorted's selection on registers)
    u64 tk;
    if (V <= PDT_WAVE) {
      tk = wave_sort_desc<u64>(lp < V ? pack_key(key[0], (unsigned)lp) : 0ull);
    } else {
      // Threshold: any value that at least M per-lane maxima reach bounds the M-th best from
      // below.  Cheap form (M <= 32): every DPP row of 16 lanes sorted by itself -- ten stages,
      // none through the crossbar -- and the smallest of the four rows' ceil(M / 4)-th largest:
      // at least 4 * ceil(M / 4) lanes reach it.  It lies at or below the exact M-th largest, so a
      // few more survive; if more than 64 do (or M > 32), the exact one from the 64-key sort.
      auto collect = [&](const unsigned tau) {
        int count = 0;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          if (i * PDT_WAVE < V) {
            const bool pred = key[i] >= tau && key[i] != 0u;
            const u64 bal = __ballot(pred);
            if (bal) {
              const int at = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
              if (pred && at < PDT_SURV_CAP) surv[at] = pack_key(key[i], (unsigned)(lp + i * PDT_WAVE));
              count += __popcll(bal);
            }
          }
        }
        wave_sync();
        return count;
      };
      unsigned tau = 0u;
      int count = PDT_SURV_CAP + 1;
      if (M <= 32) {
        const unsigned rs = row_sort_desc<unsigned>(lmax);
        const int q = (M + 3) >> 2;
        tau = min(min((unsigned)__builtin_amdgcn_readlane((int)rs, q - 1), (unsigned)__builtin_amdgcn_readlane((int)rs, 16 + q - 1)),
                  min((unsigned)__builtin_amdgcn_readlane((int)rs, 32 + q - 1), (unsigned)__builtin_amdgcn_readlane((int)rs, 48 + q - 1)));
        count = collect(max(tau, 1u));
      }
      if (count > PDT_SURV_CAP) {
        const unsigned sorted_max = wave_sort_desc<unsigned>(lmax);
        tau = (unsigned)__builtin_amdgcn_readlane((int)sorted_max, M - 1);
        count = collect(tau);
      }
      if (count <= PDT_SURV_CAP) {
        // one sort of 32-bit keys: the mass key rounded up to a multiple of 64 with the survivor's
        // slot in the freed bits (the lean tier's trick); exact unless two of the first M + 1 agree
        // in the upper 26 bits -- then the (key, token) pairs themselves are sorted
        const u64 mine = lp < count ? surv[lp] : 0ull;
        const unsigned k32 = lp < count ? (((key_of(mine) + 63u) & ~63u) | (63u - (unsigned)lp)) : 0u;
        const unsigned st = wave_sort_desc<unsigned>(k32);
        const unsigned st_next = (unsigned)__builtin_amdgcn_mov_dpp((int)st, 0x130, 0xf, 0xf, true);  // wave_shl:1
        if (__ballot(lp < M && lp + 1 < count && (st >> 6) == (st_next >> 6)) == 0ull) {
          tk = lp < count ? surv[63 - (int)(st & 63u)] : 0ull;
        } else {
          tk = wave_sort_desc<u64>(mine);
        }
      } else {  // heavy ties: chunked top-64 merge, the row formed again
        tk = 0ull;
        for (int v0 = 0; v0 < V; v0 += PDT_WAVE) {
          const int v = v0 + lp;
          const unsigned k = v < V ? fkey_nonneg(mix(p[v], frow[v])) : 0u;
          const bool pred = k >= tau && k != 0u;
          if (__ballot(pred)) tk = wave_merge_top64(tk, pred ? pack_key(k, (unsigned)v) : 0ull);
        }
      }
      wave_sync();
    }
    const int my_tok = lp < M ? (int)idx_of(tk) : -1;
    if (lp < M) {
      tl_tok[slot * PDT_WAVE + lp] = my_tok;
      tl_p[slot * PDT_WAVE + lp] = fkey_nonneg_inv(key_of(tk));
    }
    return my_tok;
  };

  // where every prefix's last token sits in a list (the frame's index look-ups); my_tok: the lane's list token
  auto positions = [&](const int my_tok, const int my_last, const int Kp, const int slot) {
    int lp = lane;
    asm volatile("" : "+v"(lp));
    int where = -1;
    for (int j = 0; j < Kp; ++j) {
      const u64 hit = __ballot(my_tok == __builtin_amdgcn_readlane(my_last, j));
      where = lp == j ? (hit ? (int)__builtin_ctzll(hit) : -1) : where;
    }
    if (lp < Kp) lpos[slot * W + lp] = where;
  };
  // ---- the lists the published frame needs, this wave's share (list r goes to wave (r + 1) mod 4: the
  // consumer takes one only when there are four or more) ----
  auto build_lists = [&](const int t, const int nb, const int Kp) {
    wait_above(&row_ready[t % kLmTabRows], t);
    const float *p = rows + (t % kLmTabRows) * ly.row_floats;
    const int M = ctc_list_len(V, W, Kp);
    int lp = lane;
    asm volatile("" : "+v"(lp));
    const int my_last = lp < Kp ? lastc_pub[lp] : 0;
    for (int r = (wave + kLmTabWaves - 1) % kLmTabWaves; r < nb; r += kLmTabWaves) {
      const int d = __builtin_amdgcn_readfirstlane(build_ctx[r]);
      const int cw = __builtin_amdgcn_readfirstlane(ctx_tok[d]);
      const int my_tok = make_list(p, A.factors + (int64_t)(cw & kCtxRow) * A.f_stride, M, d, surv0 + wave * PDT_SURV_CAP);
      positions(my_tok, my_last, Kp, d);
    }
    st_flag(&done[wave], t + 1);
  };

  if (wave != 0) {
    // ---- workers: acoustic rows two frames ahead; lists in the frames that need any ---------------
    if (wave - 1 < min(2, Tn)) produce_row(wave - 1);
    for (int t = 0; t < Tn; ++t) {
      wait_above(ctx_pub, t);
      // (the word carries its frame: a consumer that needs no lists does not wait for the workers, and a
      // worker that comes late to frame t may find frame t + 1's word here -- then frame t needed none)
      const int bw = __builtin_amdgcn_readfirstlane(__hip_atomic_load(build_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      const int nb = (bw >> 6) == t ? (bw & 63) : 0;
      if (nb > 0) build_lists(t, nb, __builtin_amdgcn_readfirstlane(*kp_pub));
      if (t + 2 < Tn && 1 + (t + 2) % 3 == wave) produce_row(t + 2);
    }
    return;
  }

  // ---- consumer: the sequential beam update ----------------------------------------------------
  FrameLds L;
  L.surv = surv0;
  L.tl_tok = tl_tok;
  L.tl_p = tl_p;
  L.trie_u = a.trie + (int64_t)n * a.T * W;
  L.nxt_old = reinterpret_cast<int *>(cs);
  L.nxt_new = L.nxt_old + nxt_stride(W);
  L.chm = reinterpret_cast<unsigned *>(L.nxt_new + nxt_stride(W));
  L.info = reinterpret_cast<int *>(L.chm + W);
  DenseCtx dc{};
  dc.lists_ready = 1;
  dc.etab = etab;
  dc.etab_stride = W;
  dc.list_id = list_id;
  dc.lpos = lpos;
  dc.lpos_stride = W;
  Beam bm;  // :1097-1105: one empty prefix with all the mass on "ends in blank"
  bm.nb = lane == 0 ? 0.0f : -PDT_INF;
  bm.b = lane == 0 ? 1.0f : -PDT_INF;
  bm.last = 0;
  bm.len = 0;
  bm.node = -1;
  bm.isp = lane == 0 ? 1u : 0u;
  bm.origin = lane;
  int Kp = 1;
  // The contexts of a beam: prefixes with the same last token (or none) share a list.  A list is only
  // built for prefixes whose extensions CAN be among the frame's winners: with every beam entry valid,
  // the K non-extension candidates are K candidates, so the K-th winner is at least the smallest of
  // them (tau, formed from the masses without the merges' additions: a lower bound of what the frame
  // computes, float addition and multiplication being monotone); every extension mass of prefix k is at
  // most tot_k * (largest token probability of the frame mixed with the largest factor of k's row) --
  // the same float operations on larger operands.  ub_k < tau closes prefix k's extension streams
  // (DenseCtx::closed).  In blank-dominated frames that leaves a list or two instead of one per context.
  unsigned closed = 0u;
  int D_pub = 0, nb_pub = 0;  // contexts / lists of the published frame (wave-uniform)
#ifdef PDT_UTT_STATS
  int open_lists_dbg_v = 0, *open_lists_dbg = &open_lists_dbg_v;
#endif
  constexpr int kPairLoads = (WC > 0 ? WC * WC : kMaxWidth * kMaxWidth) / PDT_WAVE;
  float fpv[kPairLoads], fv1 = 0.0f, fm = 0.0f;
  int ctx = A.sos_row;  // this prefix's row of the table (its last order - 1 tokens)
  // (issued right after a frame has decided the new prefixes; read in publish(t1))
  auto fetch_pairs = [&](const int t1) {
    int lp = lane;
    asm volatile("" : "+v"(lp));
    const int lastc = min(max(bm.last, 0), V - 1);
    const int c = min(max(ctx, 0), A.contexts - 1);
    fm = A.fmax[c];  // the row's largest factor (publish()'s bound)
    // the factor of this prefix's context at frame t1's most probable token (publish()'s bound)
    wait_above(&row_ready[t1 % kLmTabRows], t1);
    fv1 = A.factors[(int64_t)c * A.f_stride + __float_as_int(row_stat[(t1 % kLmTabRows) * 3 + 2])];
#pragma unroll
    for (int q = 0; q < kPairLoads; ++q) {
      const int idx = lp + q * PDT_WAVE;
      const int k = idx / W, j = idx - k * W;  // (W a constant in the width-16 instance: shifts)
      const int ck = shfl_i(c, k & (PDT_WAVE - 1)), tj = shfl_i(lastc, j);
      fpv[q] = idx < W * W ? A.factors[(int64_t)ck * A.f_stride + tj] : 0.0f;
    }
  };
  auto publish = [&](const int t) {
    int lp = lane;
    asm volatile("" : "+v"(lp));
    wait_above(&row_ready[t % kLmTabRows], t);
    const float *p = rows + (t % kLmTabRows) * ly.row_floats;
    const float tot = bm.nb + bm.b;
    const bool valid = lp < Kp && tot > -PDT_INF;
    const int lastc = min(max(bm.last, 0), V - 1);
    const int c = min(max(ctx, 0), A.contexts - 1);
    const bool dead = readlane_f(tot, 0) == 0.0f;  // every mass underflowed: nothing left to decide
    const float p_blank = p[V], p1 = row_stat[(t % kLmTabRows) * 3], p2 = row_stat[(t % kLmTabRows) * 3 + 1];
    const float m2_lb = bm.nb * p[lastc] + tot * p_blank;
    const int n_valid = __popcll(__ballot(valid));
    const float tau = n_valid >= min(W, Kp * (V + 1)) ? wave_min(valid ? m2_lb : PDT_INF) : 0.0f;
    // the largest extension probability of this prefix, bounded: the frame's most probable token with
    // ITS factor (fv1), every other token at most the runner-up with the row's largest factor -- the mix
    // is monotone in both operands (before round 5: the largest probability with the largest factor,
    // which opened a list in a third of the frames; now one in ...)
    const float scale_b = 1.0f - p_blank;
    const float ext_max = A.valid_mixture ? fmaxf(keep * p1 + A.beta * (fv1 * scale_b), keep * p2 + A.beta * (fm * scale_b))
                                          : fmaxf(p1 * fv1, p2 * fm);
    const bool open = valid && !(tot * ext_max < tau);
    closed = (unsigned)__ballot(valid && !open);
    // (every valid prefix keeps its context's etab row -- a closed prefix still feeds the merges with
    // its extension masses at its children's tokens; only the LIST is skipped when no prefix of the
    // context is open: bit 30 of the context word)
    int leader = lp;
    bool group_open = false;
    for (int j = W - 1; j >= 0; --j) {
      const int cj = __builtin_amdgcn_readlane(c, j);
      const int fj = __builtin_amdgcn_readlane((int)valid | ((int)open << 1), j);
      if ((fj & 1) && c == cj) leader = j;
      group_open = group_open || ((fj & 2) && c == cj);
    }
    const bool is_leader = valid && leader == lp;
    const u64 leaders = __ballot(is_leader);
    const int my_rank = __popcll(leaders & ((1ull << lp) - 1ull));
    const int id = shfl_i(my_rank, leader);
    const u64 open_leaders = dead ? 0ull : __ballot(is_leader && group_open);
    if (lp < W) {
      list_id[lp] = valid ? id : 0;
      lastc_pub[lp] = lastc;
      if (is_leader) {
        ctx_tok[my_rank] = c | (group_open ? (1 << 30) : 0);
        ctx_lead[my_rank] = lp;
        if (group_open) build_ctx[__popcll(open_leaders & ((1ull << lp) - 1ull))] = my_rank;
      }
    }
    // the factors fetched after the previous frame (or just now, for the first): (context of prefix k,
    // last token of prefix j) for every pair
    for (int q = 0; q < kPairLoads; ++q) {
      const int idx = lp + q * PDT_WAVE;
      if (idx < W * W) fpair[idx] = fpv[q];
    }
    D_pub = dead ? 0 : __popcll(leaders);
    nb_pub = __popcll(open_leaders);
    if (lp == 0) {
      *ctx_count = D_pub;
      *kp_pub = Kp;
      *build_count = (t << 6) | nb_pub;
    }
#ifdef PDT_UTT_STATS
    *open_lists_dbg = nb_pub;
#endif
    st_flag(ctx_pub, t + 1);
  };
  if (Tn > 0) {
    fetch_pairs(0);
    publish(0);
  }
#ifdef PDT_LMTAB_STAMPS
  unsigned long long acc_[4] = {0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
#ifdef PDT_UTT_STATS
  const unsigned long long utt_t0_ = __builtin_readcyclecounter();
  unsigned pdt_utt_acc[4] = {0, 0, 0, 0};
#endif
  for (int t = 0; t < Tn; ++t) {
    // the frame's tables.  etab rows (the mixed probability at every prefix's last token: merges, last-token
    // streams) of EVERY context and the position rows of contexts without a list are a few LDS round trips
    // from fpair: the consumer's own.  Lists -- one frame in ten has any -- wake the workers.
    if (D_pub > 0) {
      int lp = lane;
      asm volatile("" : "+v"(lp));
      const float *p = rows + (t % kLmTabRows) * ly.row_floats;
      const float scale = 1.0f - p[V];
      const int my_last = lp < Kp ? lastc_pub[lp] : 0;
      const float pl = p[my_last];
      for (int d = 0; d < D_pub; ++d) {
        const int cw = __builtin_amdgcn_readfirstlane(ctx_tok[d]);
        const int lead = __builtin_amdgcn_readfirstlane(ctx_lead[d]);
        if (lp < Kp) {
          const float fv = fpair[lead * W + lp];
          etab[d * W + lp] = A.valid_mixture ? keep * pl + A.beta * (fv * scale) : pl * fv;
          if (!((cw >> 30) & 1)) lpos[d * W + lp] = -1;  // no prefix of this context can extend into the winners: no list
        }
      }
      if (nb_pub > 0) build_lists(t, nb_pub, Kp);
    }
    LMTAB_STAMP(0);
#ifdef PDT_UTT_STATS
    if (lane == 0) pdt_utt_acc[3] += (unsigned)D_pub;  // contexts of this frame (an etab row each)
    pdt_utt_acc[2] += (unsigned)open_lists_dbg_v;      // ... of which need a list
#ifdef PDT_UTT_LISTFRAMES
    pdt_utt_acc[1] += open_lists_dbg_v > 0 ? 1u : 0u;  // (instead of the lean-tier exits: frames with any list)
#endif
#endif
    if (D_pub > 0 && nb_pub > 0)
      for (int w = 1; w < kLmTabWaves; ++w) wait_above(&done[w], t);
    wave_sync();
    LMTAB_STAMP(1);
    const float *p = rows + (t % kLmTabRows) * ly.row_floats;
    int ns, nt_, nk;
    if (!(readlane_f(bm.nb + bm.b, 0) == 0.0f)) {
      dc.closed = closed;
      ctc_frame<true, false, true>(bm, p, 1.0f, V, W, Kp, t, n, a, dc, L, ns, nt_, nk PDT_STAMP_ARG);
      int *tmp = L.nxt_old;
      L.nxt_old = L.nxt_new;
      L.nxt_new = tmp;
      Kp = W;
      // the new prefixes' rows: an extension shifts its token into the source's context
      const int ctx_s = shfl_i(ctx, nk >= 0 ? ns : lane);
      ctx = nk < 0 ? A.sos_row : ((nk == 0 || nk == 1) ? (int)((unsigned)ctx_s % (unsigned)A.ctx_mod) * A.ctx_base + nt_ : ctx_s);
    }
    if (t + 1 < Tn) fetch_pairs(t + 1);
    if (((t + 1) & ((1 << a.ckpt_shift) - 1)) == 0) {  // checkpoint (see CtcArgs::ckpt)
      const int c = ((t + 1) >> a.ckpt_shift) - 1;
      if (lane < W)
        a.ckpt[((int64_t)n * a.ckpt_count + c) * W + lane] = make_int2(bm.node, bm.len | (bm.origin << 24));
      bm.origin = lane;
    }
    LMTAB_STAMP(2);
    if (t + 1 < Tn) publish(t + 1);
    LMTAB_STAMP(3);
  }
#ifdef PDT_LMTAB_STAMPS
  if (lane == 0)
    for (int i = 0; i < 4; ++i) atomicAdd(&g_lmtab_stamps[i], acc_[i]);
#endif
#ifdef PDT_UTT_STATS
  if (lane == 0 && n < 8192) {
    g_utt_stats[n * 4 + 0] = (unsigned)((__builtin_readcyclecounter() - utt_t0_) >> 4);
    for (int k = 1; k < 4; ++k) g_utt_stats[n * 4 + k] = pdt_utt_acc[k];
#ifdef PDT_UTT_HWID  // where the consumer ran: HW_ID (SIMD bits 5:4, CU 11:8, SH 12, SE 15:13) | XCC_ID << 16
    g_utt_stats[n * 4 + 1] = (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFFu) | (__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16);
#endif
  }
#endif

  // ---- outputs (:1188-1200): probabilities, lengths, and the prefixes read off the trie ------
  if (lane < W) {
    a.y_probs[n * W + lane] = bm.nb + bm.b;
    a.y_lens[n * W + lane] = bm.len;
  }
  // (records stored by THIS wave, read back below with ordinary loads: its stores complete -- release,
  // workgroup scope: no L2 write-back -- and its CU's L1 holds nothing stale -- acquire; ctc_search.hip)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  {
    // the walk of ctc_search.hip: checkpoint table in the freed row ring (the workers have left)
    const int C = Tn >> a.ckpt_shift;
    int2 *tab = reinterpret_cast<int2 *>(smem);  // [(C + 1) x W]: fits, see launch_ctc_lm_table
    wave_sync();
    if (lane < W) {
      const bool ok = bm.node >= 0;
      tab[C * W + lane] = make_int2(bm.node, bm.len);
      int cur = bm.origin;
      for (int c = C - 1; c >= 0; --c) {
        const int2 *rec = a.ckpt + (((int64_t)n * a.ckpt_count + c) * W + cur);
        const int nd = rec->x, lo = rec->y;
        tab[c * W + lane] = ok ? make_int2(nd, lo & 0xFFFFFF) : make_int2(-1, 0);
        cur = lo >> 24;
      }
    }
    wave_sync();
    walk_trie_segments(tab, (C + 1) * W, W, a.trie + (int64_t)n * a.T * W, a.y + n * W, (int64_t)a.N * W);
    int lmin = lane < W ? bm.len : 0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) lmin = min(lmin, shfl_i(lmin, lane ^ off));
    for (int f = lmin * W + lane; f < a.S * W; f += PDT_WAVE) {
      const int ps = f / W, k = f - ps * W;
      if (ps >= tab[C * W + k].y) a.y[((int64_t)ps * a.N + n) * W + k] = 0;
    }
  }
}

template <int NR, int WC = -1>
static int launch_lm_table(const LmTabArgs &A, const LmTabLayout &ly, hipStream_t stream) {
  const size_t smem = (size_t)ly.utt_bytes;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_lm_table_kernel<NR, WC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((ctc_lm_table_kernel<NR, WC>), dim3((unsigned)A.c.N), dim3(256), smem, stream, A, ly);
  return (int)hipGetLastError();
}

}  // namespace pdt

extern "C" {

static int64_t lm_table_trie_bytes(int64_t T, int64_t N, int64_t width) {
  return ((T + T / 32 + 1) * N * width * (int64_t)sizeof(int2) + 255) & ~(int64_t)255;
}

int64_t pdt_ctc_lm_table_search_workspace_bytes(int64_t T, int64_t N, int64_t V, int64_t width) {
  if (T < 0 || N < 0 || V < 1 || width < 1) return 0;
  return lm_table_trie_bytes(T, N, width) + 16;
}

int pdt_ctc_lm_table_search(const float *logits, int64_t T, int64_t N, int64_t V, int64_t lg_st, int64_t lg_sn,
                            int64_t lg_sv, const int64_t *lens, int64_t width, int64_t S, const float *factors,
                            const float *factor_max, int64_t contexts, int64_t f_stride, int64_t sos_row,
                            int64_t ctx_base, int64_t ctx_mod, float beta, int valid_mixture,
                            int64_t *y, int64_t *y_lens, float *y_probs, void *workspace, void *stream) {
  using namespace pdt;
  if (T < 0 || N < 0 || V < 1 || width < 1 || S < 0 || contexts < 1 || f_stride < V) return PDT_E_ARG;
  if (sos_row < 0 || sos_row >= contexts || contexts < V || ctx_base < V || ctx_mod < 1 || ctx_base * ctx_mod != contexts)
    return PDT_E_ARG;
  if (contexts >= (1 << 30)) return PDT_E_TOO_LONG;  // (a context word keeps the row in 30 bits)
  if (N == 0) return PDT_OK;
  if (!y_lens || !y_probs || !factors || !factor_max || (T > 0 && (!logits || !workspace)) || (S > 0 && !y)) return PDT_E_ARG;
  if (width > kMaxWidth || V + 1 > 80 * PDT_WAVE) return PDT_E_TOO_LONG;
  if (T * width >= (1ll << 31) || N >= (1ll << 31) || T >= (1 << 24)) return PDT_E_TOO_LONG;
  LmTabArgs A{};
  CtcArgs &a = A.c;
  a.logits = logits; a.lg_st = lg_st; a.lg_sn = lg_sn; a.lg_sv = lg_sv;
  a.lens = lens;
  a.T = (int)T; a.N = (int)N; a.V = (int)V; a.W = (int)width; a.S = (int)S;
  a.y = y; a.y_lens = y_lens; a.y_probs = y_probs;
  a.trie = reinterpret_cast<int2 *>(workspace);
  a.ckpt = a.trie + T * N * width;
  a.exact_div = switches().ctc_exact_div == 1 ? 1 : 0;
  a.no_lean_extra = switches().ctc_lean_extra == 0 ? 1 : 0;
  A.ctx_base = (int)ctx_base; A.ctx_mod = (int)ctx_mod;
  A.factors = factors; A.fmax = factor_max; A.f_stride = f_stride; A.contexts = (int)contexts; A.sos_row = (int)sos_row; A.beta = beta; A.valid_mixture = valid_mixture;
  const LmTabLayout ly = lmtab_layout((int)V, (int)width, (int)contexts);
  if ((size_t)ly.utt_bytes > 160 * 1024) return PDT_E_TOO_LONG;
  int sh = 5;  // checkpoint spacing: the (C + 1) x W table of the output walk overlays the row ring
  while (((size_t)(T >> sh) + 1) * width * sizeof(int2) > (size_t)ly.rows_bytes) ++sh;
  a.ckpt_shift = sh;
  a.ckpt_count = (int)(T >> sh) + 1;
  const int chunks = (int)((V + 1 + PDT_WAVE - 1) / PDT_WAVE);
  if (a.W == 16) {
    if (chunks <= 8) return launch_lm_table<8, 16>(A, ly, (hipStream_t)stream);
    if (chunks <= 16) return launch_lm_table<16, 16>(A, ly, (hipStream_t)stream);
    if (chunks <= 32) return launch_lm_table<32, 16>(A, ly, (hipStream_t)stream);
    if (chunks <= 48) return launch_lm_table<48, 16>(A, ly, (hipStream_t)stream);
    return launch_lm_table<80, 16>(A, ly, (hipStream_t)stream);
  }
  if (chunks <= 8) return launch_lm_table<8>(A, ly, (hipStream_t)stream);
  if (chunks <= 16) return launch_lm_table<16>(A, ly, (hipStream_t)stream);
  if (chunks <= 32) return launch_lm_table<32>(A, ly, (hipStream_t)stream);
  if (chunks <= 48) return launch_lm_table<48>(A, ly, (hipStream_t)stream);
  return launch_lm_table<80>(A, ly, (hipStream_t)stream);
}

}  // extern "C"

#ifdef PDT_LMTAB_STAMPS
extern "C" int pdt_debug_read_lmtab_stamps(unsigned long long *host8, int reset) {
  hipError_t e = hipMemcpyFromSymbol(host8, HIP_SYMBOL(pdt::g_lmtab_stamps), sizeof(unsigned long long) * 8);
  if (e == hipSuccess && reset) {
    unsigned long long z[8] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(pdt::g_lmtab_stamps), z, sizeof(z));
  }
  return (int)e;
}
#endif

#ifdef PDT_UTT_STATS
extern "C" int pdt_debug_read_utt_stats_lm(unsigned *host, int count, int reset) {
  hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(pdt::g_utt_stats), sizeof(unsigned) * 4 * (size_t)count);
  if (e == hipSuccess && reset) {
    static unsigned z[8192 * 4];
    e = hipMemcpyToSymbol(HIP_SYMBOL(pdt::g_utt_stats), z, sizeof(z));
  }
  return (int)e;
}
#endif
