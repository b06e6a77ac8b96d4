// CTC prefix beam search for gfx950 -- producer / consumer waves per utterance, one LANE per
// beam entry.
//
// Replaces CTCPrefixSearch.forward without a language model (reference
// _decoding.py:1064-1202) and its step function ctc_prefix_search_advance (:636-934),
// which the reference evaluates as ~45 dense tensor ops per frame (a (N,K,K,V) one-hot and a
// gather over the whole (t,N,K) history among them).  Here the whole T-frame search of one
// utterance runs inside one workgroup slice:
//   * producer wave(s): stage a frame's logits in an LDS ring slot, softmax statistics by DPP
//     reductions, and the sorted list of the top tokens: all K + K' that can matter
//     (wave_top_sorted), or -- rows of up to 511 tokens, K <= 16 -- only the exact top c for
//     a guessed threshold, which the consumer completes in the rare frame that needs more;
//   * consumer wave: beam state (nb, b, last token, length, trie node, is-prefix row as a
//     32-bit mask) lives in the registers of lane k; candidates are never materialised: the
//     mass of extending prefix k with token v is w_k(v) * p[v], monotone in p[v] for fixed k,
//     so every prefix walks the ONE shared sorted list, and the exact global top-K comes from
//     the tiers in ctc_frame.hpp (lean 64-key sort / threshold + counting rank / serial
//     wave-max rounds with refills);
//   * prefixes are a trie in HBM ((parent, token) per created node) instead of dense
//     (t, N, K) histories, so a frame writes K records instead of gathering t*K tokens; the
//     only history look-ups the algorithm needs (token of prefix b at the position where
//     prefix a ends) are served from a K x K table in LDS maintained incrementally.  The
//     prefixes are read off the trie at the end through checkpoints (CtcArgs::ckpt), in
//     parallel segments.
//
// K + K' list entries suffice: a prefix can take at most K winners, lose at most K' - 1 list
// entries to extensions that merge into existing beam prefixes, and its own last token is a
// separate stream.
//
// Tie policy: equal-mass candidates are taken lowest lane first (the reference's torch.topk
// leaves ties unspecified); parity is defined on tie-free inputs.
#include "ctc_ring.hpp"
#include "switches.hpp"

#include <cstdlib>

namespace pdt {

// Workgroup = utt_per_wg x (P producer waves + one consumer wave).  A producer streams the
// logits: softmax statistics + sorted top-M token list of frame t go into ring slot t % nstage
// while the consumer runs the (sequential) beam update of earlier frames -- the dependency
// chains overlap instead of adding up.  The producer side has no dependence between frames,
// so for long rows (large V, where one wave per frame is the bottleneck) P > 1 producers take
// frames t = p, p + P, ... in turn.  Hand-off through LDS words per utterance
// (workgroup-scope release / acquire): ready[slot] = t + 1 once frame t is in its slot,
// consumed = number of frames the consumer has finished.
// NT >= 0 (P = 1 only): the number of full 64-token chunks of a row, V / 64, as a compile-time
// constant -- the row pass then has no chunk predicates to evaluate (instantiated for the
// byte-sized vocabularies V = 256..319; NT = -1: any V).
// INREG (P = 1 only): the whole row, V + 1 <= 512 elements, sits in the eight prefetch registers
// of the producer's lanes; longer rows take the generic LDS-staged row pass of the P > 1 forms.
// GROW (generic row pass): rows no LDS ring can hold -- see RingLayout::row_global.
// (No amdgpu_num_sgpr: eight waves per SIMD cap the register-resident form at 80 scalar registers
// by themselves -- 96 admit seven waves, measured 2.65 against 2.14 ms -- and the long-row forms,
// four waves per SIMD, may take the 102 there are: an explicit 80 cost them 92-118 scalar spills.)
// WC > 0: the beam width as a compile-time constant (a.W must equal it): the tier choices, list
// lengths and table strides that depend on it fold away -- scalar work and scalar registers the
// register-resident form has none to spare of (instantiated for the default width 16).
// PAIR (the constant-shape instance, V = 256, W = 16): the producer takes TWO frames per pass, frame t in
// lanes 0-31 and frame t + 1 in lanes 32-63 (32 lanes x 8 chunks + the blank): reductions, the 32-key
// sort, the list and the header are paid once per pair of rows -- see the producer below.
template <int P, int NT = -1, bool INREG = (P == 1), bool GROW = false, int WC = -1, int VC = -1, bool PAIR = false>
__global__ void __launch_bounds__(256, INREG ? 8 : 4)
ctc_search_kernel(const CtcArgs a, const RingLayout rl_arg) {
  static_assert(!INREG || P == 1, "the register-resident row pass is a one-producer form");
  static_assert((WC > 0 && VC > 0) ? (P == 1 && INREG && !GROW) : true, "compile-time shapes: the register-resident form");
  // (both known: the LDS layout is a set of constants too -- the launcher checks that it is the
  // layout the plan chose)
  const RingLayout rl = (WC > 0 && VC > 0) ? ring_layout(VC, WC, PDT_RING_STAGES, PDT_UTT_PER_WG, 1) : rl_arg;
  static_assert(!GROW || !INREG, "rows in the workspace are a form of the generic row pass");
  static_assert(NT < 0 || INREG, "compile-time chunk counts belong to the register-resident row pass");
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  // (wave-uniform by construction; telling the compiler keeps the utterance index and every
  // address derived from it in scalar registers)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int u = wave / (P + 1);
  const int role = wave - u * (P + 1);  // 0 .. P-1: producer, P: consumer
  const bool producer = role < P;
  // (clamped: the odd utterance of the last workgroup has no work, but its waves still reach the
  // workgroup barrier below)
  const int64_t n_raw = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * rl.utt_per_wg + u;
  const bool idle = u >= rl.utt_per_wg || n_raw >= a.N;
  const int64_t n = idle ? 0 : n_raw;
  const int V = VC > 0 ? VC : a.V, W = WC > 0 ? WC : a.W;  // (VC: the vocabulary size likewise)
  // (... and with it contiguous rows: the launcher checks the strides)
  const int64_t lg_sv = VC > 0 ? 1 : a.lg_sv, lg_sn = VC > 0 ? VC + 1 : a.lg_sn;
  unsigned char *ub = smem + (size_t)u * rl.utt_bytes;
  unsigned char *ring = ub;
  unsigned char *cs = ub + (size_t)rl.slot_bytes * rl.nstage;      // consumer scratch
  u64 *surv0 = reinterpret_cast<u64 *>(cs + consumer_scratch_bytes(W));
  const int pr = (P == 1 || !producer) ? 0 : role;                   // producer index
  u64 *surv = surv0 + pr * PDT_SURV_CAP;                             // one scratch per producer
  unsigned *surv32 = reinterpret_cast<unsigned *>(surv);             // (short lists: 32-bit sort keys)
  int *consumed = reinterpret_cast<int *>(surv0 + P * PDT_SURV_CAP);  // frames the consumer finished
  int *ready = consumed + 1;                                          // [nstage] frame + 1 held by a slot
  int *want_full = ready + 4;  // consumer -> producers: the lean tier keeps failing, send complete lists
  // slot: [row of probabilities | list tokens | list probabilities | token -> position | header];
  // GROW keeps the first and the fourth in the workspace
  unsigned char *gring = GROW ? a.grow + (size_t)n * rl.nstage * rl.g_slot_bytes : nullptr;
  auto slot_row = [&](int sl) {
    return reinterpret_cast<float *>(GROW ? gring + (size_t)sl * rl.g_slot_bytes : ring + (size_t)sl * rl.slot_bytes);
  };
  auto slot_tok = [&](int sl) {
    return reinterpret_cast<int *>(ring + (size_t)sl * rl.slot_bytes + (GROW ? 0 : (size_t)rl.row_floats * 4));
  };
  auto slot_p = [&](int sl) { return reinterpret_cast<float *>(slot_tok(sl) + PDT_WAVE); };
  auto slot_pos = [&](int sl) {
    return GROW ? reinterpret_cast<unsigned char *>(slot_row(sl) + rl.row_floats)
                : reinterpret_cast<unsigned char *>(slot_p(sl) + PDT_WAVE);
  };
  auto slot_hdr = [&](int sl) {
    return GROW ? slot_p(sl) + PDT_WAVE : reinterpret_cast<float *>(slot_pos(sl) + rl.pos_bytes);
  };
  // frames of this utterance; never more than the S rows of y the caller allocated
  const int Tn = min(a.S, a.lens ? (int)min((int64_t)a.T, max((int64_t)0, a.lens[n])) : a.T);
  const int NS = rl.nstage;

#ifdef PDT_STATS
  {  // which SIMD hosts which role (HW_REG_HW_ID bits 5:4 = SIMD id)
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    PDT_STAT(8 + (int)((hw >> 4) & 3u) * 2 + (producer ? 0 : 1));
  }
#endif
  if (producer && !idle) {
    for (int sl = pr; sl < NS; sl += P)
      for (int v = lane; v < rl.pos_bytes; v += PDT_WAVE) slot_pos(sl)[v] = 0xFF;
    if (pr == 0 && lane <= 5)  // consumed, ready[0 .. 4), want_full
      __hip_atomic_store(&consumed[lane], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();  // flags / pos tables initialised (the only workgroup barrier)
  if (idle) return;

  if (producer) {
    if constexpr (PAIR) {
      // ---- two frames per pass ---------------------------------------------------------------------
      // Lane h = lane & 31 of a half holds the tokens h + 32 i, i = 0 .. 7, of ITS frame (lanes 0-31:
      // frame t, lanes 32-63: frame t + 1; past the last frame the upper half repeats it into a slot
      // nobody reads), lane h = 0 the blank as well.  Same bits as the one-frame form: the numerators
      // are element-wise, and the normaliser is summed in that form's association -- its lane l
      // accumulates the tokens l + 64 j, so a lane here keeps TWO partial sums (even chunks = its lane
      // h, odd chunks = its lane h + 32), both go through the same DPP row steps, and the four row
      // totals are combined as (R0 + R1) + (R2 + R3) like wave_sum_f's last two steps.
      // Survivors of the guessed threshold are appended through an LDS cursor per half (ds_add_rtn: a
      // unique slot per lane, no ballot / mbcnt arithmetic; their order does not matter, they are
      // sorted), and a half whose count misses the window takes the complete selection of the
      // one-frame form on its row in LDS, the whole wave working on it.
      static_assert(!PAIR || (P == 1 && INREG && !GROW && WC == 16 && VC == 256 && NT == 4), "two frames per pass: V = 256, W = 16");
      constexpr int kM = 32;  // ctc_list_len(256, 16, 16)
      constexpr int kShortMin = PDT_SHORT_MIN, kShortMax = 32, kShortLo = PDT_SHORT_LO, kShortHi = PDT_SHORT_HI, kProbeRank = PDT_SHORT_PROBE;
      constexpr u64 kFirstLanes = 0x0000000100000001ull;  // lane 0 of each half
      int lp = lane;
      asm volatile("" : "+v"(lp));
      const int hl = lp & 31, half = lp >> 5;
      const int row_bytes = rl.row_floats * 4;
      unsigned char *const my_ring = ring + half * rl.slot_bytes;  // slot sl + half of the pass
      int *cnt = want_full + 1;  // [2] cursors into the survivor buffers: LDS addresses
      typedef __attribute__((address_space(3))) unsigned lds_u32;
      const unsigned surv_at = (unsigned)(uintptr_t)(lds_u32 *)surv32 + (unsigned)half * 256u;  // 64 words per half
      const unsigned surv_lim = surv_at + 63u * 4u;
      const unsigned ntok = 511u - (unsigned)hl;  // (the token, inverted, of chunk 0)
      float thr = PDT_INF;  // logit offset of the guessed threshold from the row mean (none yet: one survivor, a miss)
      float pre[8], preb = -PDT_INF;  // (preb: the blank's logit in the first lane of a half, -inf elsewhere)
      // rows of the pass after this one: a running pointer; the row past the last frame is the last frame again
      const float *nrow = a.logits + n * (VC + 1) + hl + (int64_t)min(half, Tn - 1) * a.lg_st;
      if (Tn > 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) pre[i] = nrow[i * 32];
        if (lane_predicate<kFirstLanes>()) {
          preb = nrow[VC];
          __hip_atomic_store(&cnt[half], (int)surv_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      for (int t = 0, sl = 0; t < Tn; t += 2, sl ^= 2) {
        // both slots free: at most NS frames in flight (NS = 4: the consumer holds the previous pair)
        while (t + 1 - __hip_atomic_load(consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= NS)
          __builtin_amdgcn_s_sleep(2);
        unsigned char *sb = my_ring + sl * rl.slot_bytes;
        float *p = reinterpret_cast<float *>(sb);
        int *tl_tok = reinterpret_cast<int *>(sb + row_bytes);
        float *tl_pp = reinterpret_cast<float *>(tl_tok + PDT_WAVE);
        unsigned char *pos = reinterpret_cast<unsigned char *>(tl_pp + PDT_WAVE);
        float *hdr = reinterpret_cast<float *>(pos + rl.pos_bytes);
        if (t >= NS) {  // un-index the list this slot held NS frames ago
          const int Mprev = __float_as_int(hdr[2]);
          if (hl < Mprev) pos[tl_tok[hl]] = 0xFF;
        }
        const bool short_now = __hip_atomic_load(want_full, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;
        // row maximum (every lane of the half) and the mean of the tokens (for the threshold only)
        float mxl = fmax3_raw(pre[0], pre[1], pre[2]);
        mxl = fmax3_raw(mxl, pre[3], pre[4]);
        mxl = fmax3_raw(mxl, pre[5], pre[6]);
        mxl = fmax3_raw(mxl, pre[7], preb);
        const float mx = half_max_all_f(mxl);
        // TAME pass: every token within 86 of the maximum (all numerators normal floats): the cheaper exp
        // of the same bits (exp_tame2: no clamp, rounding by the 1.5 * 2^23 trick, 2^n by an integer add)
        float mnl = fmin3_raw(pre[0], pre[1], pre[2]);
        mnl = fmin3_raw(mnl, pre[3], pre[4]);
        mnl = fmin3_raw(mnl, pre[5], pre[6]);
        mnl = fmin_raw(mnl, pre[7]);
        const bool tame = __ballot(!(mnl - mx >= -86.0f)) == 0ull;
        // (the mean of half the tokens: the threshold only has to track the level of the row)
        const f32x2 sx2 = f32x2{pre[0], pre[1]} + f32x2{pre[2], pre[3]};
        const int last_of_half = lp | 31;
        const float mean = shfl_f(half_sum_at31(sx2.x + sx2.y), last_of_half) * (1.0f / 128.0f);
        // survivors: numerator >= the numerator of the guessed threshold (v_exp_f32's accuracy is plenty:
        // any threshold value gives an exact upper set of the list order; +inf: none)
        float te = PDT_INF;
        if (short_now) te = __builtin_amdgcn_exp2f(fminf(mean + thr - mx, 0.0f) * 0x1.715476p+0f);
        // (the cursor's address: once per pass, not rematerialised in every chunk)
        unsigned cnt_at = (unsigned)(uintptr_t)(lds_u32 *)(cnt + half);
        asm volatile("" : "+v"(cnt_at));
        float sA, sB;
        auto token_chunk = [&](const int i, const float e, float &acc, const bool first) {
          p[hl + i * 32] = e;
          acc = first ? e : acc + e;
          if (e >= te) {
            const unsigned at = __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)cnt_at, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // 32-bit sort key: the numerator's upper 23 bits (it is >= +0: its bit pattern is monotone),
            // the token (inverted: lowest first) below -- never 0, the empty word
            *(lds_u32 *)(uintptr_t)min(at, surv_lim) = (__float_as_uint(e) & ~511u) | ((ntok - 32u * (unsigned)i) & 511u);
          }
        };
        if (tame) {
#pragma unroll
          for (int i = 0; i < 8; i += 2) {
            const f32x2 e2 = exp_tame2(f32x2{pre[i], pre[i + 1]} - f32x2{mx, mx});
            token_chunk(i, e2.x, sA, i == 0);
            token_chunk(i + 1, e2.y, sB, i == 0);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; i += 2) {
            const f32x2 e2 = exp_nonpos2(f32x2{pre[i], pre[i + 1]} - f32x2{mx, mx});
            token_chunk(i, e2.x, sA, i == 0);
            token_chunk(i + 1, e2.y, sB, i == 0);
          }
        }
        const float eb = exp_nonpos(preb - mx);  // (+0 where there is no blank)
        if (lane_predicate<kFirstLanes>()) p[VC] = eb;
        sA += eb;
        if (t + 2 < Tn) {
          // (two frames on; at the odd tail the upper half takes the frame the lower half arrives at)
          nrow += (t + 3 < Tn || half == 0) ? 2 * a.lg_st : a.lg_st;
#pragma unroll
          for (int i = 0; i < 8; ++i) pre[i] = nrow[i * 32];
          if (lane_predicate<kFirstLanes>()) preb = nrow[VC];
        }
        const float s = shfl_f(half_sum_at31(sA) + half_sum_at31(sB), last_of_half);
        const float inv0 = __builtin_amdgcn_rcpf(s);
        const float inv = __builtin_fmaf(__builtin_fmaf(-s, inv0, 1.0f), inv0, inv0);
        wave_sync();
        const int used = __hip_atomic_load(&cnt[half], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - (int)surv_at;  // bytes
        if (lane_predicate<kFirstLanes>()) __hip_atomic_store(&cnt[half], (int)surv_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const bool okv = used >= 4 * kShortMin && used <= 4 * kShortMax;
        const int nshort = used >> 2;
        // ONE 32-key sort per half; exact unless two survivors agree in the upper 23 bits (then the
        // (value, token) pairs are sorted instead)
        const unsigned sk = (okv && hl < nshort) ? ((const lds_u32 *)(uintptr_t)surv_at)[hl] : 0u;
        // (both lists within a 16-lane row: 10 stages instead of 15)
        const unsigned st = __ballot(okv && nshort > 16) == 0ull ? row_sort_desc<unsigned>(sk) : half_wave_sort_desc<unsigned>(sk);
        const unsigned st_next = (unsigned)__builtin_amdgcn_mov_dpp((int)st, 0x130, 0xf, 0xf, true);  // wave_shl:1
        int tok = 511 - (int)(st & 511u);
        if (__ballot(okv && hl + 1 < nshort && (st >> 9) == (st_next >> 9)) != 0ull) {
          const int tk0 = 511 - (int)(sk & 511u);
          const u64 tk = half_wave_sort_desc<u64>((okv && hl < nshort) ? pack_key(fkey_nonneg(p[tk0]), (unsigned)tk0) : 0ull);
          tok = (int)idx_of(tk);
        }
        if (okv && hl < nshort) {
          tl_tok[hl] = tok;
          tl_pp[hl] = p[tok] * inv;
          pos[tok] = (unsigned char)hl;
        }
        {
          const float step = fmaxf(fabsf(thr) * 0.03125f, 1e-3f);
          thr += !okv ? 0.0f : (nshort > kShortHi ? step : (nshort < kShortLo ? -step : 0.0f));
        }
        if (lane_predicate<kFirstLanes>()) {
          hdr[0] = inv;
          hdr[2] = __int_as_float(okv ? nshort : kM);
        }
        const u64 bad = __ballot(!okv);
        if (bad != 0ull) {
          // a miss (or the consumer asked for complete lists): the complete selection, the whole wave on
          // the half's row in LDS; its probe seeds the next guess of both halves
          for (int h = 0; h < 2; ++h) {
            if (((bad >> (32 * h)) & 1ull) == 0ull) continue;
            unsigned char *sbh = ring + (sl + h) * rl.slot_bytes;
            float *ph = reinterpret_cast<float *>(sbh);
            int *tokh = reinterpret_cast<int *>(sbh + row_bytes);
            float *pph = reinterpret_cast<float *>(tokh + PDT_WAVE);
            unsigned char *posh = reinterpret_cast<unsigned char *>(pph + PDT_WAVE);
            unsigned probe = 0u;
            wave_sync();
            build_shared_list<false>(ph, readlane_f(inv, 32 * h), VC, kM, surv, tokh, pph, posh, nullptr, &probe, kProbeRank);
            thr = readlane_f(mx, 32 * h) + __logf(fkey_nonneg_inv(probe)) - readlane_f(mean, 32 * h);
          }
        }
        wave_sync();
        if (lane == 0) __hip_atomic_store(&ready[0], t + 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      return;
    }
    // elements of the NEXT row of this wave held in registers while the current one is processed
    constexpr int kPrefetch = 8;
#ifndef PDT_LONG_BATCH
#define PDT_LONG_BATCH 32
#endif
    constexpr int kBatch = PDT_LONG_BATCH;  // long rows: loads in flight beyond the prefetched part
    float pre[kPrefetch];
    // Short lists (P == 1, K <= 16).  94 % of the frames are decided by each prefix's first one or
    // two list entries (lean tier, ctc_frame.hpp), so the producer normally hands over only the
    // exact top-c tokens for some c in [kShortMin, 32]: the tokens whose numerator is >= that of
    // a GUESSED logit threshold  mean + thr_off  (counted and compacted while the exponentials
    // are computed; one 32-key sort).  The guess comes from the previous frames -- seeded by
    // the probe of a complete selection, then nudged to keep c near 20 -- and a miss (c outside
    // the window) just takes the complete selection.  The consumer completes a short list itself
    // in the frames that turn out to need more.
    constexpr int kShortMin = PDT_SHORT_MIN, kShortMax = 32, kShortLo = PDT_SHORT_LO, kShortHi = PDT_SHORT_HI, kProbeRank = PDT_SHORT_PROBE;
    const bool short_ok = INREG && W <= 16 && V > PDT_WAVE;
    float thr_off = PDT_INF;  // no guess yet
    const int nt_ = NT >= 0 ? NT : V / PDT_WAVE, rem_ = V - nt_ * PDT_WAVE;  // full token chunks; lane of the blank
    const float inv_ntok = 1.0f / (float)(nt_ > 0 ? nt_ * PDT_WAVE : 1);
    if (pr < Tn) {
      const float *row0 = a.logits + (int64_t)pr * a.lg_st + n * lg_sn + (int64_t)lane * lg_sv;
#pragma unroll
      for (int i = 0; i < kPrefetch; ++i) {
        const int v = lane + i * PDT_WAVE;
        pre[i] = v <= V ? row0[(int64_t)(i * PDT_WAVE) * lg_sv] : 0.0f;
      }
    }
    int sl = pr % NS;  // t % NS, kept by add / compare: no division in the loop
    for (int t = pr; t < Tn; t += P, sl = sl + P >= NS ? sl + P - NS : sl + P) {
      // wait for the slot to be free: at most NS frames in flight
      while (t - __hip_atomic_load(consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= NS)
        __builtin_amdgcn_s_sleep(2);
      float *p = slot_row(sl);
      int *tl_tok = slot_tok(sl);
      unsigned char *pos = slot_pos(sl);
      float *hdr = slot_hdr(sl);
      // un-index the list this slot held NS frames ago
      if (t >= NS) {
        const int Mprev = __float_as_int(hdr[2]);
        if (lane < Mprev) pos[tl_tok[lane]] = 0xFF;
      }
      // softmax statistics of frame t (:1093): p[v] = exp(x[v] - max), sum over v in [0, V]
      // short lists only while the consumer's lean tier mostly decides the frames (its own
      // completion of a short list is far dearer than a complete selection here)
      const bool short_now = short_ok &&
          __hip_atomic_load(want_full, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;
      float s = 0.0f;
      unsigned lmax = 0u;  // per-lane maximum ordering key over the tokens (not the blank)
      unsigned tkey = 0xFFFFFFFFu;  // key of the guessed threshold (none: nothing survives)
      int nshort = 0;               // tokens at or above it
      float mean = 0.0f, mx_of_row = 0.0f;
      if constexpr (INREG) {
        // the whole row (V + 1 <= 512) sits in the prefetch registers: maximum, exponentials
        // and ordering keys come straight from them -- one LDS store per element instead of
        // store + load + store + load.  Chunks i < nt hold tokens in every lane (wave-uniform
        // branches, no lane masks); chunk nt ends with the blank in lane `rem`.
        // (laundered: chunk predicates and the two masks of chunk nt are recomputed by a scalar
        // compare where used, not hoisted out of the frame loop into scalar registers that spill)
        int lp = lane, nt = nt_, rem = rem_;
        if constexpr (VC > 0) {  // (the lane of the blank is a constant too)
          asm volatile("" : "+v"(lp));
          nt = NT;
          rem = VC - NT * PDT_WAVE;
        } else if constexpr (NT >= 0) {
          asm volatile("" : "+v"(lp), "+s"(rem));
          nt = NT;
        } else {
          asm volatile("" : "+v"(lp), "+s"(nt), "+s"(rem));
        }
        const bool in_row = lp <= rem, is_tok = lp < rem;
        float mx = -PDT_INF, sx = 0.0f;
#pragma unroll
        for (int i = 0; i < kPrefetch; ++i) {
          if (i > nt) break;  // (one scalar compare + branch; the chunks beyond are not visited)
          if (i < nt) {
            mx = fmax_raw(mx, pre[i]);
            sx += pre[i];
          } else if (i == nt) {
            mx = in_row ? fmax_raw(mx, pre[i]) : mx;
          }
        }
        mx = wave_max_f(mx);
        mx_of_row = mx;
        if (short_ok) {
          mean = wave_sum_f(sx) * inv_ntok;  // over the tokens of the full chunks
          // survivors: numerator >= the numerator of the guessed threshold (same exp routine, so
          // the set is an upper set of the list order)
          // (any threshold value gives an exact top-c list: v_exp_f32 accuracy is plenty)
          if (short_now && t > 0 && thr_off < PDT_INF)
            tkey = fkey_nonneg(__builtin_amdgcn_exp2f(fminf(mean + thr_off - mx, 0.0f) * 0x1.715476p+0f));
        }
        auto survivors = [&](const unsigned key, const bool pred, const int v) {
          const u64 bal = __ballot(pred);
          if (bal) {
            const int at = nshort + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            // 32-bit sort key: the value key rounded up to a multiple of 512, token (inverted:
            // lowest first) in the freed bits; V <= 511 here
            if (pred && at < kShortMax) surv32[at] = ((key + 511u) & ~511u) | (511u - (unsigned)v);
            nshort += __popcll(bal);
          }
        };
        auto token_chunk = [&](const int i, const float e) {
          const int v = lp + i * PDT_WAVE;
          p[v] = e;
          s += e;
          const unsigned key = fkey_nonneg(e);
          lmax = max(lmax, key);
          survivors(key, key >= tkey, v);
        };
#pragma unroll
        for (int i = 0; i < kPrefetch; i += 2) {
          if (i > nt) break;
          if (i + 1 < nt) {
            // two full chunks: the range reduction in packed fp32 (v_pk_mul / fma / add)
            const f32x2 e2 = exp_nonpos2(f32x2{pre[i], pre[i + 1]} - f32x2{mx, mx});
            token_chunk(i, e2.x);
            token_chunk(i + 1, e2.y);
            continue;
          }
#pragma unroll
          for (int j = i; j < i + 2; ++j) {
            if (j < nt) {
              token_chunk(j, exp_nonpos(pre[j] - mx));
            } else if (j == nt) {
              const int v = lp + j * PDT_WAVE;
              unsigned key = 0u;
              if (in_row) {
                const float e = exp_nonpos(pre[j] - mx);
                p[v] = e;
                s += e;
                if (is_tok) key = fkey_nonneg(e);
              }
              lmax = max(lmax, key);
              survivors(key, key >= tkey, v);
            }
          }
        }
        if (t + P < Tn) {
          // (lane term first: the per-element offsets are then wave-uniform scalars, not eight
          // hoisted 64-bit vector products)
          const float *nrow = a.logits + (int64_t)(t + P) * a.lg_st + n * lg_sn + (int64_t)lp * lg_sv;
#pragma unroll
          for (int i = 0; i < kPrefetch; ++i) {
            if (i > nt) break;
            if (i < nt) {
              pre[i] = nrow[(int64_t)(i * PDT_WAVE) * lg_sv];
            } else if (i == nt) {
              if (in_row) pre[i] = nrow[(int64_t)(i * PDT_WAVE) * lg_sv];
            }
          }
        }
      } else {
        float mx = -PDT_INF;
#pragma unroll
        for (int i = 0; i < kPrefetch; ++i) {
          const int v = lane + i * PDT_WAVE;
          if (v <= V) {
            p[v] = pre[i];
            mx = fmaxf(mx, pre[i]);
          }
        }
        {
          const float *row = a.logits + (int64_t)t * a.lg_st + n * lg_sn;
          int v = lane + kPrefetch * PDT_WAVE;
          if constexpr (!INREG) {
            // long rows: kBatch loads in flight, then the rest in guarded groups of 8 (a load that
            // waits for the one before it is a round trip to HBM: V = 5000 took nine per frame
            // with batches of 8 and a one-by-one tail)
            for (; v + (kBatch - 1) * PDT_WAVE <= V; v += kBatch * PDT_WAVE) {
              float x[kBatch];
#pragma unroll
              for (int i = 0; i < kBatch; ++i) x[i] = row[(int64_t)(v + i * PDT_WAVE) * lg_sv];
#pragma unroll
              for (int i = 0; i < kBatch; ++i) {
                p[v + i * PDT_WAVE] = x[i];
                mx = fmaxf(mx, x[i]);
              }
            }
            for (; v <= V; v += 8 * PDT_WAVE) {
              float x[8];
#pragma unroll
              for (int i = 0; i < 8; ++i)
                x[i] = v + i * PDT_WAVE <= V ? row[(int64_t)(v + i * PDT_WAVE) * lg_sv] : -PDT_INF;
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                if (v + i * PDT_WAVE <= V) p[v + i * PDT_WAVE] = x[i];
                mx = fmaxf(mx, x[i]);
              }
            }
          }
          for (; v <= V; v += PDT_WAVE) {
            const float x = row[(int64_t)v * lg_sv];
            p[v] = x;
            mx = fmaxf(mx, x);
          }
        }
        if (t + P < Tn) {
          // (lane term first: the per-element offsets are then wave-uniform scalars, not eight
          // hoisted 64-bit vector products)
          const float *nrow = a.logits + (int64_t)(t + P) * a.lg_st + n * lg_sn + (int64_t)lane * lg_sv;
#pragma unroll
          for (int i = 0; i < kPrefetch; ++i) {
            const int v = lane + i * PDT_WAVE;
            if (v <= V) pre[i] = nrow[(int64_t)(i * PDT_WAVE) * lg_sv];
          }
        }
        mx = wave_max_f(mx);
        {
          int v = lane;
          if constexpr (!INREG) {
            for (; v + 7 * PDT_WAVE <= V; v += 8 * PDT_WAVE) {
              float x[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) x[i] = p[v + i * PDT_WAVE];
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const float e = exp_nonpos(x[i] - mx);
                p[v + i * PDT_WAVE] = e;
                s += e;
              }
            }
          }
          for (; v <= V; v += PDT_WAVE) {
            const float e = exp_nonpos(p[v] - mx);
            p[v] = e;
            s += e;
          }
        }
      }
      s = wave_sum_f(s);
      wave_sync();
      const int M = ctc_list_len(V, W, (t == 0 && !(WC > 0 && VC > 0 && VC + 1 >= WC)) ? 1 : W);
      // reciprocal of the normaliser: v_rcp_f32 + one Newton step (within 1 ulp of the quotient;
      // the IEEE division sequence is 12 VALU instructions)
      const float inv0 = __builtin_amdgcn_rcpf(s);
      float inv = __builtin_fmaf(__builtin_fmaf(-s, inv0, 1.0f), inv0, inv0);
      if (VC <= 0 && a.exact_div) {  // the quotient itself, element by element (CtcArgs::exact_div; general instances)
        for (int v = lane; v <= V; v += PDT_WAVE) p[v] = p[v] / s;
        wave_sync();
        inv = 1.0f;
      }
      int Ml = M;
      if (short_ok && nshort >= kShortMin && nshort <= kShortMax) {
        PDT_STAT(1);
        // one 32-key sort of 32-bit keys; exact unless two survivors agree in the upper 23 bits
        // (then the (value, token) pairs are sorted instead)
        const unsigned sk = lane < nshort ? surv32[lane] : 0u;
        unsigned st = nshort <= 16 ? row_sort_desc<unsigned>(sk) : half_wave_sort_desc<unsigned>(sk);
        const unsigned st_next = (unsigned)__builtin_amdgcn_mov_dpp((int)st, 0x130, 0xf, 0xf, true);  // wave_shl:1
        int tok = 511 - (int)(st & 511u);
        if (__ballot(lane + 1 < nshort && (st >> 9) == (st_next >> 9)) != 0ull) {
          const int tk0 = 511 - (int)((lane < nshort ? surv32[lane] : 0u) & 511u);
          const u64 tk = half_wave_sort_desc<u64>(lane < nshort ? pack_key(fkey_nonneg(p[tk0]), (unsigned)tk0) : 0ull);
          tok = (int)idx_of(tk);
        }
        Ml = min(nshort, M);
        if (lane < Ml) {
          tl_tok[lane] = tok;
          slot_p(sl)[lane] = p[tok] * inv;
          pos[tok] = (unsigned char)lane;
        }
        const float step = fmaxf(fabsf(thr_off) * 0.03125f, 1e-3f);
        thr_off += nshort > kShortHi ? step : (nshort < kShortLo ? -step : 0.0f);
      } else {
        PDT_STAT(nshort > kShortMax ? 3 : 2);
        unsigned probe = 0u;
        build_shared_list<!INREG>(p, inv, V, M, surv, tl_tok, slot_p(sl), pos, INREG ? &lmax : nullptr,
                                  &probe, kProbeRank);
        // logit offset (from the row mean) of the kProbeRank-th largest per-lane maximum
        if (short_ok) thr_off = mx_of_row + __logf(fkey_nonneg_inv(probe)) - mean;
      }
      PDT_STAT(0);
      if (lane == 0) {
        hdr[0] = inv;
        hdr[2] = __int_as_float(Ml);
      }
      wave_sync();
      // one producer: frames arrive in order and ready[0] is a plain frame counter
      if (lane == 0)
        __hip_atomic_store(&ready[P == 1 ? 0 : sl], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return;
  }

  // ---- consumer: the sequential beam update ----------------------------------------------
#ifndef PDT_NO_PRIO
  // the consumer's dependency chain is the critical path of the utterance (the producers run
  // ahead by up to nstage frames): its instructions issue first, the producers fill the gaps
  __builtin_amdgcn_s_setprio(PDT_CONSUMER_PRIO);
#endif
  FrameLds L;
  L.surv = surv;  // unused by the shared-list form
  L.trie_u = a.trie + (int64_t)n * a.T * W;
  L.nxt_old = reinterpret_cast<int *>(cs);  // 8-byte aligned: nxt_new doubles as u64 scratch
  L.nxt_new = L.nxt_old + nxt_stride(W);
  L.chm = reinterpret_cast<unsigned *>(L.nxt_new + nxt_stride(W));
  L.info = reinterpret_cast<int *>(L.chm + W);
  Beam bm;  // :1097-1105: one empty prefix with all the mass on "ends in blank"
  bm.nb = lane == 0 ? 0.0f : -PDT_INF;
  bm.b = lane == 0 ? 1.0f : -PDT_INF;
  bm.last = 0;
  bm.len = 0;
  bm.node = -1;
  bm.isp = lane == 0 ? 1u : 0u;
  bm.origin = lane;
  // Live beam entries: 1 at t = 0, then W.  With both shapes known (and V + 1 >= W) the first frame
  // runs with W entries too, all but the first invalid (mass -inf: no candidates, no merges) -- the
  // state a frame leaves behind when it finds fewer than W candidates -- so K, M and every tier
  // choice derived from them are constants of the loop.  The producer sends the longer list (M for
  // W entries) in frame 0 as well.
  constexpr bool kFullFromStart = WC > 0 && VC > 0 && VC + 1 >= WC;
  int Kp = kFullFromStart ? WC : 1;
  int fail_score = 0, full_mode = 0;
  // (32-frame checkpoints in the constant-shape instance: the launcher checks)
  const int ckpt_shift = (WC > 0 && VC > 0) ? 5 : a.ckpt_shift;
#ifdef PDT_STAMPS
  unsigned pdt_stamp_acc[14] = {0};  // wave-uniform: scalar registers
#endif
  int sl = 0;  // t % NS
#ifdef PDT_UTT_STATS
  const unsigned long long utt_t0_ = __builtin_readcyclecounter();
  unsigned pdt_utt_acc[4] = {0, 0, 0, 0};
#endif
  for (int t = 0; t < Tn; ++t, sl = sl + 1 == NS ? 0 : sl + 1) {
    PDT_STAMP_BEGIN;
#ifdef PDT_UTT_STATS
    const unsigned long long w0_ = __builtin_readcyclecounter();
#endif
    if (__hip_atomic_load(&ready[P == 1 ? 0 : sl], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= t) {
      // the producer is behind: wait at low priority (the polling loop itself issues VALU
      // instructions that would otherwise outrank the producer this wave is waiting for)
#ifndef PDT_NO_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      do {
        __builtin_amdgcn_s_sleep(PDT_SPIN_SLEEP);
      } while (__hip_atomic_load(&ready[P == 1 ? 0 : sl], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= t);
#ifndef PDT_NO_PRIO
      __builtin_amdgcn_s_setprio(PDT_CONSUMER_PRIO);
#endif
    }
    PDT_STAMP(0);
#if defined(PDT_UTT_STATS) && !defined(PDT_UTT_REASONS)
    PDT_UTT(3, (__builtin_readcyclecounter() - w0_) >> 4);
#endif
    L.tl_tok = slot_tok(sl);
    L.tl_p = slot_p(sl);
    L.pos = slot_pos(sl);
    L.hdr = slot_hdr(sl);
    L.list_len = __float_as_int(slot_hdr(sl)[2]);
    const float s = slot_hdr(sl)[0];  // reciprocal of the frame's softmax normaliser
    int ns, nt, nk;
    // (a fresh copy of the lane index for the end of the frame: the masks of `lane == 0` / `lane < W`
    // are then a compare where used, not scalar pairs kept across the frame and spilled to lanes)
    int lane_t = lane;
#ifndef PDT_SKIP_CONSUMER  // diagnostic build: producer-side cost alone
    // Every mass has underflowed to 0 (the beam is ordered, so entry 0 holds the largest): every
    // candidate of this and all later frames has mass 0 too, the reference's top-k over them is
    // a tie artefact, and nothing is left to decide -- the beam stays as it is.  (float32
    // probability-space masses get there after a few hundred frames of p_max ~ 0.5, or ~1000 of
    // p_max ~ 0.9; without this exit each such frame runs the full tiers on a beam of zeros.)
    // (one branch, everything that depends on it inside: kept across the frame as a flag it is a
    // scalar pair spilled to lanes and read back every frame)
    bool enough = true;
    if (!(readlane_f(bm.nb + bm.b, 0) == 0.0f)) {
      enough = ctc_frame<false>(bm, slot_row(sl), s, V, W, Kp, t, n, a, DenseCtx{}, L, ns, nt, nk PDT_STAMP_ARG);
      int *tmp = L.nxt_old;  // (a skipped frame leaves the next-token tables where they are)
      L.nxt_old = L.nxt_new;
      L.nxt_new = tmp;
      if (!kFullFromStart) Kp = W;
    }
    // feedback to the producers: a complete selection costs the producer about what two list
    // completions cost this wave, so the balance is at one completed list in two frames: +1
    // per frame this wave had to complete a short list, -1 per frame it did not (0 .. 32);
    // complete lists above 16, short ones again below 4
    // (through a scalar: `enough` is wave-uniform but arrives as a lane mask, and the counter and
    // mode below would live in vector registers -- a dozen vector instructions per frame)
    const int enough_s = __builtin_amdgcn_readfirstlane(enough ? 1 : 0);
    fail_score = enough_s ? max(fail_score - 1, 0) : min(fail_score + 1, 32);
    const int wf = fail_score > 16 ? 1 : (fail_score < 4 ? 0 : full_mode);
    if (wf != full_mode) {
      full_mode = wf;
      asm volatile("" : "+v"(lane_t));
      if (lane_t == 0) __hip_atomic_store(want_full, wf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#else
    ns = nt = nk = 0; (void)s;
    if (!kFullFromStart) Kp = W;
#endif
    asm volatile("" : "+v"(lane_t));
    if (lane_t == 0) __hip_atomic_store(consumed, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (((t + 1) & ((1 << ckpt_shift) - 1)) == 0) {  // checkpoint (see CtcArgs::ckpt)
      const int c = ((t + 1) >> ckpt_shift) - 1;
      if (lane_t < W)
        a.ckpt[((int64_t)n * a.ckpt_count + c) * W + lane_t] = make_int2(bm.node, bm.len | (bm.origin << 24));
      bm.origin = lane_t;
    }
  }

#ifdef PDT_UTT_STATS
  if (lane == 0 && n < 8192) {
    g_utt_stats[n * 4 + 0] = (unsigned)((__builtin_readcyclecounter() - utt_t0_) >> 4);
    for (int k = 1; k < 4; ++k) g_utt_stats[n * 4 + k] = pdt_utt_acc[k];
  }
#endif
#ifdef PDT_STAMPS
  if (lane == 0)
    for (int i = 0; i < 14; ++i) atomicAdd(&g_stamps[i], (unsigned long long)pdt_stamp_acc[i]);
  unsigned long long stamp_last_ = __builtin_readcyclecounter();
#endif
  // ---- outputs (:1188-1200): probabilities, lengths, and the prefixes read off the trie --
  if (lane < W) {
    a.y_probs[n * W + lane] = bm.nb + bm.b;
    a.y_lens[n * W + lane] = bm.len;
  }
  // (the trie and checkpoint records below were stored by THIS wave: once its stores have completed
  // -- the release, workgroup scope: no write-back of the L2 as an agent-scope release costs every
  // wave of the launch -- and this CU's L1 holds nothing stale -- the acquire -- ordinary loads see them)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#ifndef PDT_SKIP_WALK  // diagnostic build: cost of the output walk
  {
    // One chain of bm.len dependent loads per prefix would be ~T global-memory latencies with
    // nothing to overlap (every utterance of the launch ends at about the same time).  Instead:
    // (1) each prefix follows its `origin` links back through the C checkpoints -- C hops --
    // leaving (node, length) of its ancestor at every checkpoint in LDS (the ring is free now);
    // (2) the (C + 1) x W segments between consecutive checkpoints are walked by all 64 lanes.
    const int C = Tn >> ckpt_shift;
    int2 *tab = reinterpret_cast<int2 *>(ring);  // [(C + 1) x W]: fits, see launch_ctc_search
    if (lane < W) {
      const bool ok = bm.node >= 0;
      tab[C * W + lane] = make_int2(bm.node, bm.len);
      int cur = bm.origin;
      for (int c = C - 1; c >= 0; --c) {
        const int2 rec = a.ckpt[((int64_t)n * a.ckpt_count + c) * W + cur];
        const int nd = rec.x, lo = rec.y;
        tab[c * W + lane] = ok ? make_int2(nd, lo & 0xFFFFFF) : make_int2(-1, 0);
        cur = lo >> 24;
      }
    }
    wave_sync();
#ifdef PDT_STAMPS
    unsigned long long st1_ = __builtin_readcyclecounter();
    if (lane == 0) atomicAdd(&g_stamps[11], st1_ - stamp_last_);
#endif
    walk_trie_segments(tab, (C + 1) * W, W, a.trie + (int64_t)n * a.T * W, a.y + n * W, (int64_t)a.N * W);
#ifdef PDT_STAMPS
    unsigned long long st2_ = __builtin_readcyclecounter();
    if (lane == 0) atomicAdd(&g_stamps[12], st2_ - st1_);
#endif
    // rows beyond a prefix's length are 0: the kernel writes every element of y (a separate
    // fill of the whole tensor costs 2 % of the launch; here it is a few stores per prefix)
    int lmin = lane < W ? bm.len : 0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) lmin = min(lmin, shfl_i(lmin, lane ^ off));
    for (int f = lmin * W + lane; f < a.S * W; f += PDT_WAVE) {
      const int pos = f / W, k = f - pos * W;
      if (pos >= tab[C * W + k].y) a.y[((int64_t)pos * a.N + n) * W + k] = 0;
    }
  }
#endif
#ifdef PDT_STAMPS
  if (lane == 0) atomicAdd(&g_stamps[6], __builtin_readcyclecounter() - stamp_last_);
#endif
}

template <int P, int NT = -1, bool INREG = (P == 1), bool GROW = false, int WC = -1, int VC = -1, bool PAIR = false>
static int launch_ctc_search_p(const CtcArgs &a, const RingLayout &rl, hipStream_t stream) {
  const size_t smem = (size_t)rl.utt_bytes * rl.utt_per_wg;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_search_kernel<P, NT, INREG, GROW, WC, VC, PAIR>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  const unsigned grid = (unsigned)((a.N + rl.utt_per_wg - 1) / rl.utt_per_wg);
  hipLaunchKernelGGL((ctc_search_kernel<P, NT, INREG, GROW, WC, VC, PAIR>), dim3(grid), dim3(64 * (P + 1) * rl.utt_per_wg), smem,
                     stream, a, rl);
  return (int)hipGetLastError();
}

// checkpoint spacing: 32 frames, doubled until the (C + 1) x W table of the output walk fits in
// the ring of the launch configuration that will run (the table overlays the freed ring, and the
// ring of the workspace-row form is only nstage x 8 KiB)
__host__ inline int ckpt_shift_for(int T, int W, const RingLayout &rl) {
  const size_t ring = (size_t)rl.slot_bytes * rl.nstage;
  int sh = 5;
  while (((size_t)(T >> sh) + 1) * W * sizeof(int2) > ring) ++sh;
  return sh;
}

// Launch configuration for rows of V tokens and beam width W: producer waves per utterance
// (P), ring depth, utterances per workgroup, whether the row lives in the producer's registers.
struct CtcPlan {
  int producers, nstage, utt_per_wg, inreg;
};
__host__ inline int plan_ctc_search(int V, int W, CtcPlan *plan, RingLayout *rl) {
  if (W < 1 || W > kMaxWidth) return PDT_E_TOO_LONG;
  const size_t soft_cap = 64 * 1024, hard_cap = 160 * 1024;
  // Long rows: one producer wave per frame is the bottleneck and LDS (not registers) bounds
  // the occupancy, so three producers share the frames of an utterance (one utterance per
  // workgroup).
  if (V + 1 > 8 * PDT_WAVE) {
    // Four ring slots while four workgroups still fit a CU (rows up to ~1.7 K tokens), three
    // beyond: with long rows the occupancy is worth more than the fourth slot (measured, N = 4096,
    // three producers: V = 1500 3.73 ms with four slots / 3.92 with three; V = 2000 5.64 / 4.94;
    // V = 3000 8.14 / 6.93; two producers on three slots, the round-1 choice beyond ~3.9 K
    // tokens: V = 5000 32.4 ms against 24.5).
    *rl = ring_layout(V, W, 4, 1, 3);
    if ((size_t)rl->utt_bytes * 4 <= hard_cap) return *plan = CtcPlan{3, 4, 1, 0}, PDT_OK;

    *rl = ring_layout(V, W, 3, 1, 3);
    if ((size_t)rl->utt_bytes <= hard_cap) return *plan = CtcPlan{3, 3, 1, 0}, PDT_OK;
    // beyond: the rows stay in the HBM workspace, still three producers (two producers on a
    // two-slot LDS ring: V = 12 000 8.3 ms against 6.9 this way; below ~10 K tokens the LDS ring
    // wins: V = 8000 16.7 against 19.1, V = 5000 24.6 against 49.6)
    *rl = ring_layout(V, W, 4, 1, 3, true);
    return *plan = CtcPlan{3, 4, 1, 2}, PDT_OK;
  }
  // ring depth and utterances per workgroup from the LDS budget
  int nstage = PDT_RING_STAGES, upw = PDT_UTT_PER_WG;
  *rl = ring_layout(V, W, nstage, upw, 1);
  while ((size_t)rl->utt_bytes * upw > soft_cap && (upw > 1 || nstage > 2)) {
    if (upw > 1) upw = 1; else nstage = 2;
    *rl = ring_layout(V, W, nstage, upw, 1);
  }
  if ((size_t)rl->utt_bytes * upw > hard_cap) return PDT_E_TOO_LONG;
  return *plan = CtcPlan{1, nstage, upw, 1}, PDT_OK;
}

// rows held in the producers' registers (ctc_rowreg.hip)
bool ctc_rowreg_applies(int V, int W);
void ctc_rowreg_plan(int V, int W, int32_t *plan5);
int launch_ctc_rowreg(CtcArgs a, hipStream_t stream);

int launch_ctc_search(const CtcArgs &a, const CtcPlan &plan, const RingLayout &rl, hipStream_t stream) {
  if (plan.inreg == 2) return launch_ctc_search_p<3, -1, false, true>(a, rl, stream);
  if (plan.producers == 3) return launch_ctc_search_p<3>(a, rl, stream);
#ifndef PDT_NO_V256  // (diagnostic builds compare against the run-time shapes)
  {
    const RingLayout c = ring_layout(256, 16, PDT_RING_STAGES, PDT_UTT_PER_WG, 1);
    if (a.V == 256 && a.W == 16 && a.ckpt_shift == 5 && !a.exact_div && !a.no_lean_extra && a.lg_sv == 1 && a.lg_sn == 257 && c.nstage == rl.nstage && c.utt_per_wg == rl.utt_per_wg && c.utt_bytes == rl.utt_bytes &&
        c.slot_bytes == rl.slot_bytes) {
      // (two frames per producer pass: four ring slots, a pair of them per pass)
      if (switches().ctc_pair != 0 && rl.nstage == 4) return launch_ctc_search_p<1, 4, true, false, 16, 256, true>(a, rl, stream);
      return launch_ctc_search_p<1, 4, true, false, 16, 256>(a, rl, stream);
    }
  }
#endif
#ifndef PDT_NO_WIDTH16
  if (a.V / PDT_WAVE == 4 && a.W == 16) return launch_ctc_search_p<1, 4, true, false, 16>(a, rl, stream);
#endif
  if (a.V / PDT_WAVE == 4) return launch_ctc_search_p<1, 4>(a, rl, stream);
  return launch_ctc_search_p<1>(a, rl, stream);
}

}  // namespace pdt

extern "C" {

static int64_t ctc_trie_bytes(int64_t T, int64_t N, int64_t width) {
  // trie records + checkpoints (at most one per 32 frames)
  return ((T + T / 32 + 1) * N * width * (int64_t)sizeof(int2) + 255) & ~(int64_t)255;
}

int64_t pdt_ctc_prefix_search_workspace_bytes(int64_t T, int64_t N, int64_t V, int64_t width) {
  if (T < 0 || N < 0 || V < 1 || width < 1 || V >= (1 << 30)) return 0;
  int64_t bytes = ctc_trie_bytes(T, N, width) + 16;
  pdt::CtcPlan plan;
  pdt::RingLayout rl;
  if (pdt::plan_ctc_search((int)V, (int)(width > pdt::kMaxWidth ? pdt::kMaxWidth : width), &plan, &rl) == PDT_OK &&
      rl.row_global)
    bytes += N * (int64_t)rl.nstage * rl.g_slot_bytes;  // the rows of the ring
  return bytes;
}

int pdt_ctc_prefix_search_plan(int64_t V, int64_t width, int32_t *plan5) {
  int32_t *plan4 = plan5;
  if (V < 1 || width < 1 || !plan5) return PDT_E_ARG;
  if (V >= (1 << 30)) return PDT_E_TOO_LONG;
  pdt::CtcPlan plan;
  pdt::RingLayout rl;
  const int rc = pdt::plan_ctc_search((int)V, (int)width, &plan, &rl);
  if (rc != PDT_OK) return rc;
  plan4[0] = plan.producers; plan4[1] = plan.nstage; plan4[2] = plan.utt_per_wg; plan4[3] = plan.inreg;
  plan5[4] = 0;
  if (pdt::ctc_rowreg_applies((int)V, (int)width)) pdt::ctc_rowreg_plan((int)V, (int)width, plan5);  // (3: long rows in registers)
  return PDT_OK;
}

int pdt_ctc_prefix_search(const float *logits, int64_t T, int64_t N, int64_t V, int64_t lg_st,
                          int64_t lg_sn, int64_t lg_sv, const int64_t *lens, int64_t width,
                          int64_t S, int64_t *y, int64_t *y_lens, float *y_probs,
                          void *workspace, void *stream) {
  using namespace pdt;
  if (T < 0 || N < 0 || V < 1 || width < 1 || S < 0) return PDT_E_ARG;
  if (N == 0) return PDT_OK;
  if (!y_lens || !y_probs || (T > 0 && (!logits || !workspace)) || (S > 0 && !y)) return PDT_E_ARG;
  if (width > kMaxWidth) return PDT_E_TOO_LONG;
  if (T * width >= (1ll << 31) || V >= (1 << 30) || N >= (1ll << 31) || T >= (1 << 24)) return PDT_E_TOO_LONG;
  CtcArgs a{};
  a.logits = logits; a.lg_st = lg_st; a.lg_sn = lg_sn; a.lg_sv = lg_sv;
  a.lens = lens;
  a.T = (int)T; a.N = (int)N; a.V = (int)V; a.W = (int)width; a.S = (int)S;
  a.y = y; a.y_lens = y_lens; a.y_probs = y_probs;
  a.trie = reinterpret_cast<int2 *>(workspace);
  a.ckpt = a.trie + T * N * width;
  a.grow = reinterpret_cast<unsigned char *>(workspace) + ctc_trie_bytes(T, N, width);
  a.exact_div = switches().ctc_exact_div == 1 ? 1 : 0;
  a.no_lean_extra = switches().ctc_lean_extra == 0 ? 1 : 0;
  // (contiguous rows: the register form addresses a row as base + immediates)
  if (ctc_rowreg_applies(a.V, a.W) && lg_sv == 1) return launch_ctc_rowreg(a, (hipStream_t)stream);
  CtcPlan plan;
  RingLayout rl;
  const int rc = plan_ctc_search(a.V, a.W, &plan, &rl);
  if (rc != PDT_OK) return rc;
  a.ckpt_shift = ckpt_shift_for((int)T, (int)width, rl);
  a.ckpt_count = (int)(T >> a.ckpt_shift) + 1;
  return launch_ctc_search(a, plan, rl, (hipStream_t)stream);
}

}  // extern "C"

#ifdef PDT_STATS
extern "C" int pdt_debug_read_stats(unsigned long long *host16, int reset) {
  hipError_t e = hipMemcpyFromSymbol(host16, HIP_SYMBOL(pdt::g_stats), sizeof(unsigned long long) * 16);
  if (e == hipSuccess && reset) {
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(pdt::g_stats), z, sizeof(z));
  }
  return (int)e;
}
#endif

#ifdef PDT_UTT_STATS
extern "C" int pdt_debug_read_utt_stats(unsigned *host, int count, int reset) {
  hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(pdt::g_utt_stats), sizeof(unsigned) * 4 * (size_t)count);
  if (e == hipSuccess && reset) {
    static unsigned z[8192 * 4];
    e = hipMemcpyToSymbol(HIP_SYMBOL(pdt::g_utt_stats), z, sizeof(z));
  }
  return (int)e;
}
#endif

#ifdef PDT_STAMPS
extern "C" int pdt_debug_read_stamps(unsigned long long *host16, int reset) {
  hipError_t e = hipMemcpyFromSymbol(host16, HIP_SYMBOL(pdt::g_stamps), sizeof(unsigned long long) * 16);
  if (e == hipSuccess && reset) {
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(pdt::g_stamps), z, sizeof(z));
  }
  return (int)e;
}
#endif
