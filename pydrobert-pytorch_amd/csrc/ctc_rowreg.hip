// CTC prefix beam search over LONG rows (320 .. 16 447 tokens): the row of a frame never leaves
// the registers of the producer wave that read it.
//
// Replaces CTCPrefixSearch.forward without a language model (reference _decoding.py:1064-1202;
// the softmax of :1093-1095 and the per-frame step :1110-1152) for vocabularies where the form of
// ctc_search.hip -- whole rows of probabilities in an LDS ring -- leaves room for two utterances
// per CU (V = 5000: 3 slots x 25 KB): there the bytes in flight are capped by the residency, not by
// the loads (0.22 of the HBM rate).  What the consumer reads of a frame is small -- the sorted list
// of the K + K' best tokens, the blank's probability, and the probabilities of its prefixes' last
// tokens -- so here
//   * a producer wave loads the whole row into registers (NR x 64 elements, every load in flight at
//     once), takes maximum, exponentials and sum from them, selects the list from them (threshold =
//     M-th largest per-lane maximum, survivors compacted through LDS, one sort), and hands over
//     ONLY the list + a header (reciprocal normaliser, row maximum, the blank's numerator): 0.5 KB
//     per frame instead of the row.  Same arithmetic in the same order as the other forms (per-lane
//     strided sums, then the DPP reduction): identical bits;
//   * the consumer wave (the frame routine of ctc_frame.hpp, ROWLESS) indexes the list in its own
//     token -> position table (V bytes of LDS, set and cleared per frame), and fetches the logit of
//     each prefix's last token itself -- one scattered load per frame, issued as soon as the frame
//     before has decided the prefixes, turned into a probability with the header's maximum and
//     reciprocal (a token on the list reads its probability there).
// An utterance takes ~11 KB of LDS at V = 5000, so the residency is set by the registers: 128 per
// lane (80 of them the row) = four waves per SIMD, every CU holding four utterances x (three
// producers + consumer).
#include <type_traits>

#include "ctc_ring.hpp"
#include "switches.hpp"

// The rows are read once: long ones (more than 16 chunks of 64 per lane) as NON-TEMPORAL loads, so that they
// do not push one another through the L2 -- C5 (V = 5000) 9.1 -> 8.65 ms on the box that measured both;
// C3's 16-chunk rows lost 1.7 % that way and keep plain loads.
#define PDT_ROW_LOAD(p) (NR > 16 ? __builtin_nontemporal_load(p) : *(p))
#ifndef PDT_ROWREG_GUESS_MARGIN  // (a variant build with a huge one sends every guessed row down the
#define PDT_ROWREG_GUESS_MARGIN 0x1p-16f  // margin-failure path: profiles/tools/dump_rowreg.py compares)
#endif

namespace pdt {

struct RowregLayout {
  int nstage;      // ring slots per utterance
  int slot_bytes;  // [64 tokens | 64 probabilities | header: 1/sum, row max, list length, blank, sum]
  int pos_bytes;   // V padded to 16: token -> list position (0xFF: not listed), owned by the consumer
  int utt_bytes;   // ring + pos + consumer scratch + one survivor buffer per producer + flags
  int utt_per_wg, producers;
};

__host__ __device__ inline RowregLayout rowreg_layout(int V, int W, int nstage, int utt_per_wg, int producers) {
  RowregLayout r;
  r.nstage = nstage;
  r.slot_bytes = PDT_WAVE * 8 + 32;
  r.pos_bytes = (V + 15) & ~15;
  const int consumer = consumer_scratch_bytes(W);  // nxt tables + chm + info
  r.utt_bytes = (r.slot_bytes * nstage + r.pos_bytes + consumer + producers * PDT_SURV_CAP * 8 + 32 + 15) & ~15;
  r.utt_per_wg = utt_per_wg;
  r.producers = producers;
  return r;
}

// waves per SIMD the registers of an instantiation allow (the row + ~36 working registers per lane)
constexpr int rowreg_waves(int NR) {
  return NR <= 16 ? 8 : NR <= 24 ? 7 : NR <= 32 ? 6 : NR <= 48 ? 5 : NR <= 80 ? 4 : NR <= 128 ? 3 : NR <= 208 ? 2 : 1;
}

// NR: 64-token chunks of a row the producer's registers hold; NF: how many of them every row of
// this instantiation has (the launcher picks the instantiation with NF < V / 64 <= NR, so the first
// NF need no guard -- a guard per chunk is a scalar branch and, worse, a merge of the two versions
// of everything the chunk touches).  P producers per utterance take the frames t = p, p + P, ... in
// turn; a workgroup is one utterance: three producers and the consumer.
// WC > 0: the beam width as a compile-time constant (the default, 16, has instantiations of its own:
// the consumer's tier choices and table strides fold away, as in ctc_search.hip).
template <int NR, int NF, int P, int WC = -1>
__global__ void __launch_bounds__(256, rowreg_waves(NR))
ctc_rowreg_kernel(const CtcArgs a, const RowregLayout rl) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int u = wave / (P + 1);
  const int role = wave - u * (P + 1);  // 0 .. P-1: producer, P: consumer
  const bool producer = role < P;
  const int64_t n_raw = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * rl.utt_per_wg + u;
  const bool idle = u >= rl.utt_per_wg || n_raw >= a.N;
  const int64_t n = idle ? 0 : n_raw;
  const int V = a.V, W = WC > 0 ? WC : a.W, NS = rl.nstage;
  unsigned char *ub = smem + (size_t)u * rl.utt_bytes;
  unsigned char *ring = ub;
  unsigned char *pos = ub + (size_t)rl.slot_bytes * NS;
  unsigned char *cs = pos + rl.pos_bytes;  // consumer scratch
  u64 *surv0 = reinterpret_cast<u64 *>(cs + consumer_scratch_bytes(W));
  int *consumed = reinterpret_cast<int *>(surv0 + P * PDT_SURV_CAP);  // frames the consumer finished
  int *ready = consumed + 1;                                          // [nstage <= 4] frame + 1 held by a slot
  int *cursor = ready + 4;                                            // [P <= 3] a producer's cursor into its survivor buffer
  static_assert(P <= 3, "three cursors fit the flag words");
  auto slot_tok = [&](int sl) { return reinterpret_cast<int *>(ring + (size_t)sl * rl.slot_bytes); };
  auto slot_p = [&](int sl) { return reinterpret_cast<float *>(slot_tok(sl) + PDT_WAVE); };
  auto slot_hdr = [&](int sl) { return slot_p(sl) + PDT_WAVE; };
  const int Tn = min(a.S, a.lens ? (int)min((int64_t)a.T, max((int64_t)0, a.lens[n])) : a.T);

  if (!producer && !idle) {
    for (int v = lane; v < rl.pos_bytes; v += PDT_WAVE) pos[v] = 0xFF;
    if (lane <= 4) __hip_atomic_store(&consumed[lane], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();  // flags / position table initialised (the only workgroup barrier)
  if (idle) return;

  if (producer) {
    const int pr = role;
    u64 *surv = surv0 + pr * PDT_SURV_CAP;
    const int nt_ = V >> 6, rem_ = V & 63;  // full token chunks; lane of the blank in the chunk after them
    // the row's logits: r[i], i < nt: token chunks with every lane in use; rt: the chunk that ends
    // with the blank (lanes 0 .. rem; tokens below rem)
    float r[NR], rt = 0.0f;
    auto row_of = [&](int t) {
      // (rows are contiguous here -- the launcher sends strided logits to the LDS form -- so a
      // chunk is base + lane * 4 + an immediate)
      int lq = lane;
      asm volatile("" : "+v"(lq));
      return a.logits + (int64_t)t * a.lg_st + n * a.lg_sn + lq;
    };
    if (pr < Tn) {
      const float *row = row_of(pr);
#pragma unroll
      for (int i = 0; i < NR; ++i)
        if (i < NF || i < nt_) r[i] = row[i * PDT_WAVE];
      if (lane <= rem_) rt = row[nt_ * PDT_WAVE];
    }
    float thr_off = PDT_INF;  // the guessed survivor threshold: logit offset from the mean per-lane maximum (none yet)
    int sl = pr % NS;
    for (int t = pr; t < Tn; t += P, sl = sl + P >= NS ? sl + P - NS : sl + P) {
      // (laundered: nothing derived from the lane index or the chunk count is loop-invariant to the
      // compiler -- hoisted, the per-chunk predicates and index words are NR live values that spill)
      int lp = lane, nt = nt_, rem = rem_;
      asm volatile("" : "+v"(lp), "+s"(nt), "+s"(rem));
      const bool in_row = lp <= rem, is_tok = lp < rem;
      const int M = ctc_list_len(V, W, (t == 0 && WC <= 0) ? 1 : W);  // (WC: W entries from frame 0, see the consumer)
      // ---- pass A: per-lane maximum over the tokens; with the blank, the row maximum ---------
      float lmx = is_tok ? rt : -PDT_INF, lmn = in_row ? rt : PDT_INF;
#pragma unroll
      for (int i = 0; i < NR; i += 2) {
        if (i + 1 < NF || i + 1 < nt) {
          lmx = fmax3_raw(lmx, r[i], r[i + 1]);
          lmn = fmin3_raw(lmn, r[i], r[i + 1]);
        } else if (i < nt) {
          lmx = fmax_raw(lmx, r[i]);
          lmn = fmin3_raw(lmn, r[i], r[i]);
        }
      }
      const float mx = wave_max_f(lp == rem ? fmax_raw(lmx, rt) : lmx);
      // A TAME row -- every element within 86 of the maximum, i.e. every numerator a normal float32 --
      // takes exp_tame2 below; rows with masked (-inf) or far-off elements the general routine
      const bool tame = wave_min(lmn) - mx >= -86.0f;
      // ---- pass B: the tokens that can be among the M best -----------------------------------
      // The list is ordered by (numerator, token), and exp() is monotone: a logit threshold that at least
      // M tokens reach -- with a margin of 2^-16 in the logit, 128 ulps of the numerator, beyond anything
      // rounding can reorder -- selects every token whose numerator ties with or exceeds the M-th best,
      // and ranking the survivors by their numerators gives the list the other forms build from a row of
      // numerators.  The threshold is GUESSED (round 5): the mean of the per-lane maxima plus an offset that
      // follows the survivor count from row to row (the exact one -- the M-th largest per-lane maximum, a
      // 64-key sort -- seeds it and takes over whenever a guess leaves fewer than M survivors, more than
      // the buffer holds, or the M-th survivor inside the margin: checked below, after the ranking).
      // Survivors are appended through an LDS cursor (ds_add_rtn: a slot per lane, no ballot / mbcnt
      // arithmetic; their order does not matter, they are sorted).
      typedef __attribute__((address_space(3))) unsigned lds_u32r;
      typedef __attribute__((address_space(3))) u64 lds_u64r;
      const unsigned surv_at = (unsigned)(uintptr_t)(lds_u64r *)surv, surv_lim = surv_at + (PDT_SURV_CAP - 1) * 8u;
      unsigned cur_at = (unsigned)(uintptr_t)(lds_u32r *)(cursor + pr);
      asm volatile("" : "+v"(cur_at));
      auto collect = [&](const float thr) -> int {
        if (lp == 0) *(lds_u32r *)(uintptr_t)cur_at = surv_at;
        wave_sync();
        auto survivors = [&](const float x, const int v, const bool pred) {
          if (pred) {
            const unsigned at = __hip_atomic_fetch_add((lds_u32r *)(uintptr_t)cur_at, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            *(lds_u64r *)(uintptr_t)min(at, surv_lim) = ((u64)__float_as_uint(x) << 32) | (unsigned)v;
          }
        };
#pragma unroll
        for (int i = 0; i < NR; ++i)
          if (i < NF || i < nt) survivors(r[i], lp + i * PDT_WAVE, r[i] >= thr);
        survivors(rt, lp + nt * PDT_WAVE, is_tok && rt >= thr);
        wave_sync();
        return (int)((*(lds_u32r *)(uintptr_t)cur_at - surv_at) >> 3);
      };
      const float margin_of = 0x1p-16f;
      bool guessed = thr_off < PDT_INF;
      float tq = 0.0f, lmx_mean = 0.0f;
      int count = 0;
      if (guessed) {
        lmx_mean = wave_sum_f(lmx > -PDT_INF ? lmx : mx) * (1.0f / PDT_WAVE);
        tq = lmx_mean + thr_off;
        count = collect(tq);
        guessed = count >= M && count <= PDT_SURV_CAP;
        // keep the survivor count around three quarters of the buffer
        if (guessed) thr_off += count > 52 ? 0.05f : (count < M + 8 ? -0.05f : 0.0f);
      }
      if (!guessed) {
        const unsigned sorted_max = wave_sort_desc<unsigned>(fkey(lmx));
        const float tau_x = fkey_inv((unsigned)__builtin_amdgcn_readlane((int)sorted_max, M - 1));
        tq = tau_x - fmaxf(margin_of, fabsf(tau_x) * 0x1p-20f);
        count = collect(tq);
        // (seed the next rows' guesses a little below this row's exact threshold)
        if (lmx_mean == 0.0f) lmx_mean = wave_sum_f(lmx > -PDT_INF ? lmx : mx) * (1.0f / PDT_WAVE);
        thr_off = (tq - lmx_mean) - 0.15f;
      }
      // ---- pass C: softmax numerators and their sum (:1093), e[v] = exp(x[v] - max) ----------
      // per-lane sums over v = lane, lane + 64, ... in order, then the DPP reduction: the other
      // forms' arithmetic.  A chunk's register is free once its numerator is in the sum: the next
      // row of this wave moves in behind it (the last rows of an utterance re-read their own).
      const float *nrow = row_of(t + P < Tn ? t + P : t);
      float s = 0.0f;
      // (ONE version of the unrolled pass, the tame one: a branch around two copies -- or a branch per
      // pair -- makes every chunk's register, reloaded behind its use, a merge of two versions, and the
      // row spills.  A row that is not tame gets its sum again below; what this pass made of it --
      // anything, NaN included -- is dropped.)
#pragma unroll
      for (int i = 0; i < NR; i += 2) {
        if (i + 1 < NF || i + 1 < nt) {  // two chunks: the range reduction in packed fp32, same bits
          const f32x2 e2 = exp_tame2(f32x2{r[i], r[i + 1]} - f32x2{mx, mx});
          s += e2.x;
          s += e2.y;
          r[i] = PDT_ROW_LOAD(&nrow[i * PDT_WAVE]);
          r[i + 1] = PDT_ROW_LOAD(&nrow[(i + 1) * PDT_WAVE]);
        } else if (i < nt) {
          s += exp_tame2(f32x2{r[i] - mx, 0.0f}).x;
          r[i] = PDT_ROW_LOAD(&nrow[i * PDT_WAVE]);
        }
      }
      if (!tame) {
        // masked (-inf) or far-off elements: the general routine over the row read again (L2), the
        // same per-lane order of additions
        const float *row = a.logits + (int64_t)t * a.lg_st + n * a.lg_sn;
        s = 0.0f;
        for (int v = lp; v < nt * PDT_WAVE; v += PDT_WAVE) s += exp_nonpos(row[v] - mx);
      }
      const float et = in_row ? exp_nonpos(rt - mx) : 0.0f;  // (lanes beyond the blank add +0: no change)
      s += et;
      const float eb = readlane_f(et, rem);
      if (in_row) rt = nrow[nt * PDT_WAVE];
      s = wave_sum_f(s);
      const float inv0 = __builtin_amdgcn_rcpf(s);
      const float inv = __builtin_fmaf(__builtin_fmaf(-s, inv0, 1.0f), inv0, inv0);
      // ---- the sorted list: survivors ranked by (numerator key, token) -----------------------
      wave_sync();
      u64 tk = 0ull;
      if (count <= PDT_SURV_CAP && guessed) {
        // M survivors must clear the guessed threshold by the margin (a token just below the threshold
        // could otherwise tie with the M-th best in the numerator).  Checked on the survivor RECORDS --
        // the registers hold the next row by now (a round-5 fuzz run caught a redo of pass B here
        // reading them) -- and a failure, rare as it is, takes the path that reads the row again:
        // every token, the chunked merge below
        const float xm = lp < count ? __uint_as_float((unsigned)(surv[lp] >> 32)) : -PDT_INF;
        if (__popcll(__ballot(xm - tq >= fmaxf(PDT_ROWREG_GUESS_MARGIN, fabsf(xm) * 0x1p-20f))) < M) {
          count = PDT_SURV_CAP + 1;
          tq = -PDT_INF;
          thr_off -= 0.1f;
        }
      }
      if (count <= PDT_SURV_CAP) {
        const u64 rec = lp < count ? surv[lp] : 0ull;
        const unsigned key = fkey_nonneg(exp_nonpos(__uint_as_float((unsigned)(rec >> 32)) - mx));
        tk = wave_sort_desc<u64>(lp < count ? pack_key(key, (unsigned)rec) : 0ull);
      } else {
        // heavy ties / clustered values: chunked top-64 merge.  Rare, and a register file cannot be
        // indexed by a loop counter: the row is read again (L2)
        const float *row = a.logits + (int64_t)t * a.lg_st + n * a.lg_sn;
        for (int v0 = 0; v0 < V; v0 += PDT_WAVE) {
          const int v = v0 + lp;
          const float x = v < V ? row[v] : -PDT_INF;
          const bool pred = v < V && x >= tq;
          if (__ballot(pred))
            tk = wave_merge_top64(tk, pred ? pack_key(fkey_nonneg(exp_nonpos(x - mx)), (unsigned)v) : 0ull);
        }
      }
      // wait for the slot to be free: at most NS frames in flight
      while (t - __hip_atomic_load(consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= NS)
        __builtin_amdgcn_s_sleep(2);
      if (lp < M) {
        const float e = fkey_nonneg_inv(key_of(tk));
        slot_tok(sl)[lp] = (int)idx_of(tk);
        slot_p(sl)[lp] = a.exact_div ? e / s : e * inv;
      }
      if (lp == 0) {
        float *hdr = slot_hdr(sl);
        hdr[0] = inv;
        hdr[1] = mx;
        hdr[2] = a.exact_div ? eb / s : eb * inv;
        hdr[3] = s;
      }
      wave_sync();
      if (lp == 0) __hip_atomic_store(&ready[sl], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return;
  }

  // ---- consumer: the sequential beam update --------------------------------------------------
  __builtin_amdgcn_s_setprio(PDT_CONSUMER_PRIO);
  FrameLds L;
  L.surv = surv0;  // unused by the shared-list form
  L.trie_u = a.trie + (int64_t)n * a.T * W;
  L.nxt_old = reinterpret_cast<int *>(cs);
  L.nxt_new = L.nxt_old + nxt_stride(W);
  L.chm = reinterpret_cast<unsigned *>(L.nxt_new + nxt_stride(W));
  L.info = reinterpret_cast<int *>(L.chm + W);
  L.pos = pos;
  Beam bm;  // :1097-1105: one empty prefix with all the mass on "ends in blank"
  bm.nb = lane == 0 ? 0.0f : -PDT_INF;
  bm.b = lane == 0 ? 1.0f : -PDT_INF;
  bm.last = 0;
  bm.len = 0;
  bm.node = -1;
  bm.isp = lane == 0 ? 1u : 0u;
  bm.origin = lane;
  // (with the width a constant the first frame runs with W entries too, all but the first invalid --
  // ctc_search.hip, kFullFromStart; rows here have at least 128 tokens)
  int Kp = WC > 0 ? WC : 1;
  const float *lg_n = a.logits + n * a.lg_sn;
  // the logit of every prefix's last token in the coming frame (lanes beyond the beam read token 0)
  float xg = Tn > 0 ? lg_n[0] : 0.0f;
  int sl = 0;
  for (int t = 0; t < Tn; ++t, sl = sl + 1 == NS ? 0 : sl + 1) {
    if (__hip_atomic_load(&ready[sl], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= t) {
      __builtin_amdgcn_s_setprio(0);
      do {
        __builtin_amdgcn_s_sleep(PDT_SPIN_SLEEP);
      } while (__hip_atomic_load(&ready[sl], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= t);
      __builtin_amdgcn_s_setprio(PDT_CONSUMER_PRIO);
    }
    const int M = ctc_list_len(V, W, Kp);
    L.tl_tok = slot_tok(sl);
    L.tl_p = slot_p(sl);
    L.hdr = slot_hdr(sl);
    L.list_len = M;
    int ns, nt_, nk;
    // every mass has underflowed to 0: nothing is left to decide (see ctc_search.hip)
    if (!(readlane_f(bm.nb + bm.b, 0) == 0.0f)) {
      int lane_l = lane;
      asm volatile("" : "+v"(lane_l));
      const int tok_l = lane_l < M ? L.tl_tok[lane_l] : 0;
      if (lane_l < M) pos[tok_l] = (unsigned char)lane_l;
      const float inv = L.hdr[0], mx = L.hdr[1], s = L.hdr[3];
      L.pblank_in = L.hdr[2];
      wave_sync();
      const int lastc = min(max(bm.last, 0), V - 1);
      const unsigned q = pos[lastc];
      const float e = exp_nonpos(xg - mx);
      const float pl_row = a.exact_div ? e / s : e * inv;
      L.pl_in = q == 0xFFu ? pl_row : L.tl_p[q & 63u];
      ctc_frame<false, true>(bm, nullptr, inv, V, W, Kp, t, n, a, DenseCtx{}, L, ns, nt_, nk PDT_STAMP_ARG);
      if (lane_l < M) pos[tok_l] = 0xFF;  // (ordered behind the frame's reads by its closing wave_sync)
      int *tmp = L.nxt_old;
      L.nxt_old = L.nxt_new;
      L.nxt_new = tmp;
      if (WC <= 0) Kp = W;
    }
    if (lane == 0) __hip_atomic_store(consumed, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (t + 1 < Tn)
      xg = lg_n[(int64_t)(t + 1) * a.lg_st + (int64_t)min(max(bm.last, 0), V - 1) * a.lg_sv];
    if (((t + 1) & ((1 << a.ckpt_shift) - 1)) == 0) {  // checkpoint (see CtcArgs::ckpt)
      const int c = ((t + 1) >> a.ckpt_shift) - 1;
      if (lane < W)
        a.ckpt[((int64_t)n * a.ckpt_count + c) * W + lane] = make_int2(bm.node, bm.len | (bm.origin << 24));
      bm.origin = lane;
    }
  }

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
    // the walk of ctc_search.hip: checkpoint table in the freed ring + position table
    const int C = Tn >> a.ckpt_shift;
    int2 *tab = reinterpret_cast<int2 *>(ub);  // [(C + 1) x W]: fits, see launch_ctc_rowreg
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

constexpr int kRowregMaxChunks = 256;

// rows of 320 .. 16 447 tokens, beams the one-kernel search holds; PDT_CTC_ROWREG=0 keeps the LDS
// rows of ctc_search.hip (comparisons)
bool ctc_rowreg_applies(int V, int W) {
  const int mode = switches().ctc_rowreg;
  // From five token chunks on (V >= 320): below, the one-producer form of ctc_search.hip -- short exact
  // lists the consumer completes on demand, position tables built by the producer, the chunk count a
  // compile-time constant for V = 256 .. 319 -- is faster (V = 256: 2.13 ms against 2.69; V = 320: 2.71
  // against 2.62; V = 511: 3.26 against 2.74; N = 4096, T = 512).  =2: from two chunks on (comparisons).
  const int lo = mode >= 2 ? 2 : 5;
  return mode != 0 && W >= 1 && W <= kMaxWidth && V / PDT_WAVE >= lo && V / PDT_WAVE <= kRowregMaxChunks;
}

// (one producer per utterance, two utterances per workgroup, was 35 % slower at V = 5000 and 2x at
// V = 1000: a lone producer cannot keep a frame's latency off its consumer)
// (... and for rows of 200-384 tokens: 2.38-2.67 ms against the three-producer form's 2.56-2.74 and the
// LDS-row form's 2.11 at V = 200-256, N = 4096, T = 512)
RowregLayout plan_ctc_rowreg(int V, int W) { return rowreg_layout(V, W, 4, 1, 3); }

// register chunks of the instantiation that serves rows of V tokens (launch_rowreg_nr's table)
static int rowreg_chunks(int V) {
  const int c = V / PDT_WAVE;
  if (c <= 80) return (c + 7) / 8 * 8;
  if (c <= 128) return (c + 15) / 16 * 16;
  return (c + 31) / 32 * 32;
}

void ctc_rowreg_plan(int V, int W, int32_t *plan5) {
  const RowregLayout rl = plan_ctc_rowreg(V, W);
  plan5[0] = rl.producers; plan5[1] = rl.nstage; plan5[2] = rl.utt_per_wg; plan5[3] = 3;
  plan5[4] = rowreg_chunks(V);
}

template <int NR, int NF, int P, int WC = -1>
static int launch_rowreg(const CtcArgs &a, const RowregLayout &rl, hipStream_t stream) {
  const size_t smem = (size_t)rl.utt_bytes * rl.utt_per_wg;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_rowreg_kernel<NR, NF, P, WC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  const unsigned grid = (unsigned)((a.N + rl.utt_per_wg - 1) / rl.utt_per_wg);
  hipLaunchKernelGGL((ctc_rowreg_kernel<NR, NF, P, WC>), dim3(grid), dim3(256), smem, stream, a, rl);
  return (int)hipGetLastError();
}

// full token chunks (the chunk with the blank has a register of its own): instantiations in steps of
// eight chunks up to 80 (four waves per SIMD and more), of 16 up to 128 (three), of 32 up to 256 (two,
// then one: the whole register file of a SIMD holds one row)
template <int P, int WC = -1>
static int launch_rowreg_nr(const CtcArgs &a, const RowregLayout &rl, hipStream_t stream) {
  const int c = a.V / PDT_WAVE;
  if (c <= 80) {
    switch ((c + 7) / 8) {
      case 1: return launch_rowreg<8, 0, P, WC>(a, rl, stream);
      case 2: return launch_rowreg<16, 8, P, WC>(a, rl, stream);
      case 3: return launch_rowreg<24, 16, P, WC>(a, rl, stream);
      case 4: return launch_rowreg<32, 24, P, WC>(a, rl, stream);
      case 5: return launch_rowreg<40, 32, P, WC>(a, rl, stream);
      case 6: return launch_rowreg<48, 40, P, WC>(a, rl, stream);
      case 7: return launch_rowreg<56, 48, P, WC>(a, rl, stream);
      case 8: return launch_rowreg<64, 56, P, WC>(a, rl, stream);
      case 9: return launch_rowreg<72, 64, P, WC>(a, rl, stream);
      default: return launch_rowreg<80, 72, P, WC>(a, rl, stream);
    }
  }
  if (c <= 96) return launch_rowreg<96, 80, P, WC>(a, rl, stream);
  if (c <= 112) return launch_rowreg<112, 96, P, WC>(a, rl, stream);
  if (c <= 128) return launch_rowreg<128, 112, P, WC>(a, rl, stream);
  if (c <= 160) return launch_rowreg<160, 128, P, WC>(a, rl, stream);
  if (c <= 192) return launch_rowreg<192, 160, P, WC>(a, rl, stream);
  if (c <= 224) return launch_rowreg<224, 192, P, WC>(a, rl, stream);
  return launch_rowreg<256, 224, P, WC>(a, rl, stream);
}

int launch_ctc_rowreg(CtcArgs a, hipStream_t stream) {
  const RowregLayout rl = plan_ctc_rowreg(a.V, a.W);
  // checkpoint spacing: the (C + 1) x W table of the output walk overlays the ring and the position table
  const size_t room = (size_t)rl.slot_bytes * rl.nstage + rl.pos_bytes;
  int sh = 5;
  while (((size_t)(a.T >> sh) + 1) * a.W * sizeof(int2) > room) ++sh;
  a.ckpt_shift = sh;
  a.ckpt_count = (a.T >> sh) + 1;
  if (a.W == 16) return launch_rowreg_nr<3, 16>(a, rl, stream);
  return launch_rowreg_nr<3>(a, rl, stream);
}

}  // namespace pdt
