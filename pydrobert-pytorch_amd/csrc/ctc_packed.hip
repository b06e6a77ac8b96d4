// CTC prefix beam search for gfx950, beams of up to 16 prefixes over rows of up to 511 tokens:
// FOUR utterances per consumer wave.
//
// Same algorithm, same producer side and same per-utterance state as ctc_search.hip (reference
// _decoding.py:1064-1202 with lm = None, step function :636-934), re-laid-out because the launch
// time of that form follows the consumer's instruction count: there a whole 64-lane wave decides
// one utterance's 16-entry beam.  Here
//   * a workgroup is four utterances: four producer waves (one per utterance: softmax numerators,
//     normaliser, the sorted short list of top tokens into an LDS ring slot per frame) and ONE
//     consumer wave;
//   * the consumer gives every utterance a DPP row: lane 16 q + k holds beam entry k of utterance
//     q (nb, b, last token, length, trie node, is-prefix row, direct parent, child tokens);
//   * the lean tier's four candidates of a prefix -- its two best available list entries, its
//     last-token stream, its non-extension -- sit in four REGISTERS of its lane.  The best 16 of a
//     row's 64 (+ the 17th, for the tie test) come from a bitonic network whose element index is
//     4 * lane + register: eleven of its stages are register-to-register min / max pairs, the
//     others DPP moves inside a row; four utterances per instruction;
//   * a row whose frame the lean tier cannot decide (a third-best extension, a short list's bound
//     or a rounded-key tie among the winners; t = 0) is moved into lanes 0-15 of a scratch beam and
//     run through ctc_frame() -- the complete per-utterance routine of ctc_search.hip -- on that
//     utterance's own LDS tables, then moved back (about 3 % of the frames of an utterance);
//   * after the last frame the four producer waves read the prefixes off the trie, one utterance
//     each (the checkpointed walk of ctc_search.hip).
// A lone wave issues one instruction of ANY kind per ~5 cycles, so the consumer's frame is written
// for a short instruction stream and few dependent LDS round trips: everything a frame reads first
// is requested in one batch together with the producer's flag; the selects of the sort take their
// lane masks from vector registers; the merge of extensions into beam prefixes works from the
// child tokens each prefix keeps in registers.
#include "ctc_ring.hpp"

#include <cstdlib>

namespace pdt {

// ---- LDS of one utterance ----------------------------------------------------------------
//   [ring: nstage slots of (row of probabilities | list tokens | list probabilities | token ->
//    list position | header) | nxt table A | nxt table B | chm | info | sorted keys | producer
//    scratch | flags]
// The next-token tables are 16 x 16 whatever the width, because the table not in use doubles as
// the consumer's per-frame scratch (candidate records 64 x 8 B + source records 16 x 32 B).
struct PackedLayout {
  int row_floats, pos_bytes, slot_bytes, nstage;
  int nxt_a, nxt_b, chm, info, tbuf, surv, flags, utt_bytes;
};
constexpr int kPackNxtBytes = 16 * 16 * 4;
constexpr int kPackUtts = 4;
// (measured: inlined 2.40 ms, as a real call 2.51; six waves per SIMD -- 80 registers, all four
// workgroups of a CU resident -- 2.40, five -- 96 registers, no spills, but the fourth workgroup
// of a CU waits for a second round -- 2.80)
#ifndef PDT_FALLBACK_INLINE
#define PDT_FALLBACK_INLINE __forceinline__
#endif
#ifndef PDT_PACK_WAVES  // waves per SIMD the register allocation leaves room for
#define PDT_PACK_WAVES 6
#endif

__host__ __device__ constexpr PackedLayout make_packed_layout(int row_floats, int pos_bytes, int nstage) {
  PackedLayout p{};
  p.row_floats = row_floats;
  p.pos_bytes = pos_bytes;
  p.slot_bytes = row_floats * 4 + PDT_WAVE * 8 + pos_bytes + 16;
  p.nstage = nstage;
  int off = (p.slot_bytes * nstage + 15) & ~15;
  p.nxt_a = off; off += kPackNxtBytes;
  p.nxt_b = off; off += kPackNxtBytes;
  p.chm = off; off += 16 * 4;
  p.info = off; off += 16 * 16;                 // (token, packed source word, source node, direct parent)
  p.tbuf = off; off += 64 * 4;                  // the sorted keys on their way from (lane, register) to lane
  p.surv = off; off += PDT_SURV_CAP * 8;        // producer scratch (short-list keys / selection survivors)
  p.flags = off; off += 32;                     // consumed, ready[4], want_full
  p.utt_bytes = (off + 15) & ~15;
  return p;
}
// rows of any length up to 511 tokens ...
__host__ __device__ inline PackedLayout packed_layout(int V, int nstage) {
  return make_packed_layout((V + 1 + 3) & ~3, (V + 15) & ~15, nstage);
}
// ... and the layout of an instantiation whose V / 64 is a compile-time constant NT: sized for
// the longest row of the class, three slots, every offset an immediate
constexpr int kPackFixedStages = 3;
template <int NT>
__host__ __device__ constexpr PackedLayout packed_layout_fixed() {
  return make_packed_layout(64 * (NT + 1), 64 * (NT + 1), kPackFixedStages);
}

// ---- DPP row helpers (every operation stays inside the 16 lanes of one utterance) ------------
template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ unsigned dpp_u(unsigned v, unsigned ident) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)ident, (int)v, CTRL, 0xf, BANK, false);
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov(unsigned v) {
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
constexpr int kQuadId = 0xE4, kQuadX1 = 0xB1, kQuadX2 = 0x4E, kQuadX3 = 0x1B, kHalfMirror = 0x141, kMirror = 0x140;

// (mask & a) | (~mask & b) in one instruction (the compiler splits the expression in two when the
// mask's complement folds into another select)
__device__ __forceinline__ unsigned bfi(unsigned mask, unsigned a, unsigned b) {
  unsigned r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
  return r;
}

// (a, b) <- (larger, smaller)
#define PDT_CX(a, b)                                  \
  do {                                                \
    const unsigned hi_ = max(a, b), lo_ = min(a, b);  \
    a = hi_;                                          \
    b = lo_;                                          \
  } while (0)

// compare-exchange with the lane CTRL names, register r against register r (or 3 - r: FLIP); lanes
// whose `lower` is all ones keep the larger key.  The mask is a VECTOR register: a v_bfi per
// register instead of a v_cndmask on a scalar pair the allocator spills.
template <int CTRL, bool FLIP>
__device__ __forceinline__ void lane_stage(unsigned &r0, unsigned &r1, unsigned &r2, unsigned &r3, const unsigned lower) {
  const unsigned a0 = FLIP ? r3 : r0, a1 = FLIP ? r2 : r1, a2 = FLIP ? r1 : r2, a3 = FLIP ? r0 : r3;
  const unsigned h0 = max(r0, dpp_mov<CTRL>(a0)), l0 = min(r0, dpp_mov<CTRL>(a0));
  const unsigned h1 = max(r1, dpp_mov<CTRL>(a1)), l1 = min(r1, dpp_mov<CTRL>(a1));
  const unsigned h2 = max(r2, dpp_mov<CTRL>(a2)), l2 = min(r2, dpp_mov<CTRL>(a2));
  const unsigned h3 = max(r3, dpp_mov<CTRL>(a3)), l3 = min(r3, dpp_mov<CTRL>(a3));
  r0 = bfi(lower, h0, l0);
  r1 = bfi(lower, h1, l1);
  r2 = bfi(lower, h2, l2);
  r3 = bfi(lower, h3, l3);
}
// the 16 largest of two sorted quads (element index 4 * (lane & 3) + register, descending) into the
// quads BANKS names, as a bitonic sequence; returns the largest of the 16 that did not make it
template <int CTRL, int BANKS>
__device__ __forceinline__ unsigned quad_flip(unsigned &r0, unsigned &r1, unsigned &r2, unsigned &r3) {
  const unsigned h0 = max(r0, dpp_u<CTRL, BANKS>(r3, 0u)), l0 = min(r0, dpp_u<CTRL, BANKS>(r3, 0xFFFFFFFFu));
  const unsigned h1 = max(r1, dpp_u<CTRL, BANKS>(r2, 0u)), l1 = min(r1, dpp_u<CTRL, BANKS>(r2, 0xFFFFFFFFu));
  const unsigned h2 = max(r2, dpp_u<CTRL, BANKS>(r1, 0u)), l2 = min(r2, dpp_u<CTRL, BANKS>(r1, 0xFFFFFFFFu));
  const unsigned h3 = max(r3, dpp_u<CTRL, BANKS>(r0, 0u)), l3 = min(r3, dpp_u<CTRL, BANKS>(r0, 0xFFFFFFFFu));
  r0 = h0; r1 = h1; r2 = h2; r3 = h3;
  return max(max(l0, l1), max(l2, l3));  // (meaningful in the quads BANKS names)
}
// sorts a bitonic sequence of 16 (one quad: 4 lanes x 4 registers) descending
__device__ __forceinline__ void quad_merge(unsigned &r0, unsigned &r1, unsigned &r2, unsigned &r3,
                                           const unsigned lo1, const unsigned lo2) {
  lane_stage<kQuadX2, false>(r0, r1, r2, r3, lo2);
  lane_stage<kQuadX1, false>(r0, r1, r2, r3, lo1);
  PDT_CX(r0, r2); PDT_CX(r1, r3);
  PDT_CX(r0, r1); PDT_CX(r2, r3);
}
// Per row of 16 lanes: the 16 largest of the 64 keys (four per lane), descending, as element
// 4 * lane + register of lanes 0-3; returns the 17th largest (in lanes 0-3).  lo1 / lo2: all
// ones in the lanes whose bit 0 / bit 1 is clear.
__device__ __forceinline__ unsigned row_top16(unsigned &r0, unsigned &r1, unsigned &r2, unsigned &r3,
                                              const unsigned lo1, const unsigned lo2) {
  // one lane: 4 keys
  PDT_CX(r0, r1); PDT_CX(r2, r3);
  PDT_CX(r0, r3); PDT_CX(r1, r2);
  PDT_CX(r0, r1); PDT_CX(r2, r3);
  // two lanes: 8
  lane_stage<kQuadX1, true>(r0, r1, r2, r3, lo1);
  PDT_CX(r0, r2); PDT_CX(r1, r3);
  PDT_CX(r0, r1); PDT_CX(r2, r3);
  // a quad: 16
  lane_stage<kQuadX3, true>(r0, r1, r2, r3, lo2);
  lane_stage<kQuadX1, false>(r0, r1, r2, r3, lo1);
  PDT_CX(r0, r2); PDT_CX(r1, r3);
  PDT_CX(r0, r1); PDT_CX(r2, r3);
  // quads 0 | 1 -> quad 0, quads 2 | 3 -> quad 3 (partner: lane ^ 7, register 3 - r)
  const unsigned e1 = quad_flip<kHalfMirror, 0x9>(r0, r1, r2, r3);
  quad_merge(r0, r1, r2, r3, lo1, lo2);
  // quad 0 | quad 3 -> quad 0 (partner: lane ^ 15, register 3 - r)
  const unsigned e2 = quad_flip<kMirror, 0x1>(r0, r1, r2, r3);
  quad_merge(r0, r1, r2, r3, lo1, lo2);
  // the largest key that was dropped on the way
  unsigned f = max(dpp_u<kQuadId, 0x9>(e1, 0u), dpp_u<kQuadId, 0x1>(e2, 0u));
  f = max(f, dpp_mov<kQuadX1>(f));
  f = max(f, dpp_mov<kQuadX2>(f));
  return max(f, dpp_u<kMirror, 0x1>(f, 0u));
}

#ifdef PDT_STATS
#define PDT_STATN(i, n) do { const unsigned long long n_ = (n); if (lane_id() == 0) atomicAdd(&g_stats[i], n_); } while (0)
#else
#define PDT_STATN(i, n) do {} while (0)
#endif

// ---- producer: one wave per utterance, the register-resident row pass of ctc_search.hip ------
// (V + 1 <= 512: the row sits in eight prefetch registers; short exact top-c lists from a guessed
// threshold while the consumer's lean tier decides most frames.)
template <int NT>
__device__ __forceinline__ void packed_producer(const CtcArgs &a, const PackedLayout &pl, unsigned char *ub,
                                                const int64_t n, const int Tn, const int W) {
  const PackedLayout &rl = pl;
  const int lane = lane_id();
  const int V = a.V, NS = rl.nstage;
  u64 *surv = reinterpret_cast<u64 *>(ub + pl.surv);
  unsigned *surv32 = reinterpret_cast<unsigned *>(surv);
  int *consumed = reinterpret_cast<int *>(ub + pl.flags);
  int *ready = consumed + 1;
  int *want_full = ready + 4;
  auto slot_row = [&](int sl) { return reinterpret_cast<float *>(ub + (size_t)sl * rl.slot_bytes); };
  auto slot_tok = [&](int sl) { return reinterpret_cast<int *>(ub + (size_t)sl * rl.slot_bytes + (size_t)rl.row_floats * 4); };
  auto slot_p = [&](int sl) { return reinterpret_cast<float *>(slot_tok(sl) + PDT_WAVE); };
  auto slot_pos = [&](int sl) { return reinterpret_cast<unsigned char *>(slot_p(sl) + PDT_WAVE); };
  auto slot_hdr = [&](int sl) { return reinterpret_cast<float *>(slot_pos(sl) + rl.pos_bytes); };

  constexpr int kPrefetch = 8;
  float pre[kPrefetch];
  constexpr int kShortMin = PDT_SHORT_MIN, kShortMax = 32, kShortLo = PDT_SHORT_LO, kShortHi = PDT_SHORT_HI,
                kProbeRank = PDT_SHORT_PROBE;
  const bool short_ok = V > PDT_WAVE;
  float thr_off = PDT_INF;  // no guess yet
  const int nt_ = NT >= 0 ? NT : V / PDT_WAVE, rem_ = V - nt_ * PDT_WAVE;  // full token chunks; lane of the blank
  const float inv_ntok = 1.0f / (float)(nt_ > 0 ? nt_ * PDT_WAVE : 1);
  if (0 < Tn) {
    const float *row0 = a.logits + n * a.lg_sn + (int64_t)lane * a.lg_sv;
#pragma unroll
    for (int i = 0; i < kPrefetch; ++i) {
      const int v = lane + i * PDT_WAVE;
      pre[i] = v <= V ? row0[(int64_t)(i * PDT_WAVE) * a.lg_sv] : 0.0f;
    }
  }
  int sl = 0;  // t % NS
  for (int t = 0; t < Tn; ++t, sl = sl + 1 == NS ? 0 : sl + 1) {
    // wait for the slot to be free: at most NS frames in flight
    while (t - __hip_atomic_load(consumed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= NS)
      __builtin_amdgcn_s_sleep(2);
#ifdef PDT_SKIP_PRODUCER  // diagnostic build: consumer-side cost alone (the slots keep their first frames)
    if (t >= NS) {
      if (lane == 0) __hip_atomic_store(&ready[0], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      continue;
    }
#endif
    float *p = slot_row(sl);
    int *tl_tok = slot_tok(sl);
    unsigned char *pos = slot_pos(sl);
    float *hdr = slot_hdr(sl);
    if (t >= NS) {  // un-index the list this slot held NS frames ago
      const int Mprev = __float_as_int(hdr[2]);
      if (lane < Mprev) pos[tl_tok[lane]] = 0xFF;
    }
    const bool short_now = short_ok &&
        __hip_atomic_load(want_full, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;
    float s = 0.0f;
    unsigned lmax = 0u;           // per-lane maximum ordering key over the tokens (not the blank)
    unsigned tkey = 0xFFFFFFFFu;  // key of the guessed threshold (none: nothing survives)
    int nshort = 0;               // tokens at or above it
    float mean = 0.0f, mx_of_row = 0.0f;
    {
      // (laundered: chunk predicates and the two masks of chunk nt are recomputed where used, not
      // hoisted out of the frame loop)
      int lp = lane, nt = nt_, rem = rem_;
      if constexpr (NT >= 0) {
        asm volatile("" : "+v"(lp), "+s"(rem));
        nt = NT;
      } else {
        asm volatile("" : "+v"(lp), "+s"(nt), "+s"(rem));
      }
      const bool in_row = lp <= rem, is_tok = lp < rem;
      float mx = -PDT_INF, sx = 0.0f;
#pragma unroll
      for (int i = 0; i < kPrefetch; ++i) {
        if (i > nt) break;
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
      if (t + 1 < Tn) {
        const float *nrow = a.logits + (int64_t)(t + 1) * a.lg_st + n * a.lg_sn + (int64_t)lp * a.lg_sv;
#pragma unroll
        for (int i = 0; i < kPrefetch; ++i) {
          if (i > nt) break;
          if (i < nt) {
            pre[i] = nrow[(int64_t)(i * PDT_WAVE) * a.lg_sv];
          } else if (i == nt) {
            if (in_row) pre[i] = nrow[(int64_t)(i * PDT_WAVE) * a.lg_sv];
          }
        }
      }
    }
    s = wave_sum_f(s);
    wave_sync();
    const int M = ctc_list_len(V, W, t == 0 ? 1 : W);
    const float inv0 = __builtin_amdgcn_rcpf(s);
    float inv = __builtin_fmaf(__builtin_fmaf(-s, inv0, 1.0f), inv0, inv0);
    if (a.exact_div) {  // the quotient itself, element by element (CtcArgs::exact_div)
      for (int v = lane; v <= V; v += PDT_WAVE) p[v] = p[v] / s;
      wave_sync();
      inv = 1.0f;
    }
    int Ml = M;
    if (short_ok && nshort >= kShortMin && nshort <= kShortMax) {
      PDT_STAT(1);
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
      build_shared_list<false>(p, inv, V, M, surv, tl_tok, slot_p(sl), pos, &lmax, &probe, kProbeRank);
      if (short_ok) thr_off = mx_of_row + __logf(fkey_nonneg_inv(probe)) - mean;
    }
    PDT_STAT(0);
    if (lane == 0) {
      hdr[0] = inv;
      hdr[2] = __int_as_float(Ml);
    }
    wave_sync();
    if (lane == 0) __hip_atomic_store(&ready[0], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// ---- one utterance through the complete per-utterance frame routine ---------------------------
// (PDT_FALLBACK_INLINE: as a real call the routine's registers stay out of the frame loop's
// allocation, but the call's own save / restore traffic costs more than the three spills the
// inlined form leaves.)  In: the row's beam in lanes 0-15.  The routine's own lean tier is skipped (the caller's has just
// failed on this frame; `tau` = its K-th key when that is a usable bound).
struct FallbackOut {
  float nb, b;
  int last, len, node;
  unsigned isp;
  int origin;
  unsigned dch;
  int dpar, enough;
};
__device__ PDT_FALLBACK_INLINE FallbackOut packed_fallback(
    float nb, float b, int last, int len, int node, unsigned isp, int origin,
    // wave-uniform (arguments of a real call travel in vector registers all the same):
    int ub_, int slot_, int nxo_, int nxn_,          // byte offsets into the workgroup's LDS
    int tl_off_, int pos_bytes_, int chm_, int info_, int surv_, int V_, int W_, int Kp_, int t_, int T_,
    unsigned tau_, unsigned trie_lo, unsigned trie_hi, int n_) {
  extern __shared__ __align__(16) unsigned char smem[];
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  struct {
    int ub, slot, nxo, nxn, tl_off, pos_bytes, chm, info, surv, V, W, Kp, t, T;
    unsigned tau;
    int2 *trie;
    int64_t n;
  } fa;
  fa.ub = uni(ub_); fa.slot = uni(slot_); fa.nxo = uni(nxo_); fa.nxn = uni(nxn_);
  fa.tl_off = uni(tl_off_); fa.pos_bytes = uni(pos_bytes_); fa.chm = uni(chm_); fa.info = uni(info_);
  fa.surv = uni(surv_); fa.V = uni(V_); fa.W = uni(W_); fa.Kp = uni(Kp_); fa.t = uni(t_); fa.T = uni(T_);
  fa.tau = (unsigned)uni((int)tau_);
  fa.trie = reinterpret_cast<int2 *>(((u64)(unsigned)uni((int)trie_hi) << 32) | (unsigned)uni((int)trie_lo));
  fa.n = uni(n_);
  Beam fb;
  fb.nb = nb; fb.b = b; fb.last = last; fb.len = len; fb.node = node; fb.isp = isp; fb.origin = origin;
  unsigned char *ub = smem + fa.ub, *sb = smem + fa.slot;
  FrameLds L;
  L.surv = reinterpret_cast<u64 *>(ub + fa.surv);
  L.tl_tok = reinterpret_cast<int *>(sb + fa.tl_off);
  L.tl_p = reinterpret_cast<float *>(L.tl_tok + PDT_WAVE);
  L.pos = reinterpret_cast<unsigned char *>(L.tl_p + PDT_WAVE);
  L.hdr = reinterpret_cast<float *>(L.pos + fa.pos_bytes);
  L.list_len = __float_as_int(L.hdr[2]);
  L.chm = reinterpret_cast<unsigned *>(ub + fa.chm);
  L.info = reinterpret_cast<int *>(ub + fa.info);
  L.dpar_tab = L.info + 32;  // (the routine's info records are 2 x 16 ints of the 64 the region holds)
  L.nxt_old = reinterpret_cast<int *>(smem + fa.nxo);
  L.nxt_new = reinterpret_cast<int *>(smem + fa.nxn);
  L.trie_u = fa.trie + fa.n * (int64_t)fa.T * fa.W;
  L.tau_in = fa.tau;
  CtcArgs a{};
  a.trie = fa.trie;
  a.T = fa.T;
  int ns_, nt_, nk_;
#ifdef PDT_STAMPS
  unsigned inner_acc[14];  // (the routine's own phases are not the packed build's subject)
#define PDT_INNER_ACC , inner_acc
#else
#define PDT_INNER_ACC
#endif
  bool enough;
  if (fa.Kp == 1)  // t = 0: the routine as it is (one live prefix; nothing was tried here)
    enough = ctc_frame<false, true, false>(fb, reinterpret_cast<const float *>(sb), L.hdr[0], fa.V, fa.W, 1, fa.t, fa.n,
                                           a, DenseCtx{}, L, ns_, nt_, nk_ PDT_INNER_ACC);
  else
    enough = ctc_frame<false, true, true>(fb, reinterpret_cast<const float *>(sb), L.hdr[0], fa.V, fa.W, fa.Kp, fa.t,
                                          fa.n, a, DenseCtx{}, L, ns_, nt_, nk_ PDT_INNER_ACC);
  FallbackOut o;
  o.nb = fb.nb; o.b = fb.b; o.last = fb.last; o.len = fb.len; o.node = fb.node; o.isp = fb.isp;
  o.origin = fb.origin; o.dch = fb.dch; o.dpar = fb.dpar; o.enough = enough ? 1 : 0;
  return o;
}

// ---- the kernel -------------------------------------------------------------------------------
// NT >= 0: V / 64 as a compile-time constant (the producer's row pass has no chunk predicates, the
// LDS layout is packed_layout_fixed<NT>: immediates, not scalar registers); WC: the width when it is
// 16 (K, M and the beam tests become constants), 0: a.W.
template <int NT, int WC>
__global__ void __launch_bounds__(320, PDT_PACK_WAVES) ctc_search_packed_kernel(const CtcArgs a, const PackedLayout pl_arg) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr PackedLayout pl_fixed = packed_layout_fixed<(NT >= 0 ? NT : 0)>();
  const PackedLayout pl = NT >= 0 ? pl_fixed : pl_arg;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // 0-3 producers, 4 consumer
  const int64_t n0 = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * kPackUtts;
  const int V = a.V, W = WC ? WC : a.W, NS = pl.nstage;
  auto frames_of = [&](int64_t n) {
    return min(a.S, a.lens ? (int)min((int64_t)a.T, max((int64_t)0, a.lens[n])) : a.T);
  };

  if (wave < kPackUtts) {
    unsigned char *ub = smem + (size_t)wave * pl.utt_bytes;
    for (int sl = 0; sl < NS; ++sl) {
      unsigned char *pos = ub + (size_t)sl * pl.slot_bytes + (size_t)pl.row_floats * 4 + PDT_WAVE * 8;
      for (int v = lane; v < pl.pos_bytes; v += PDT_WAVE) pos[v] = 0xFF;
    }
    if (lane <= 5)  // consumed, ready[0 .. 4), want_full
      __hip_atomic_store(reinterpret_cast<int *>(ub + pl.flags) + lane, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();

  if (wave < kPackUtts) {
    const int64_t n = n0 + wave;
    if (n < a.N) packed_producer<NT>(a, pl, smem + (size_t)wave * pl.utt_bytes, n, frames_of(n), W);
  } else {
    // ---- consumer: lane 16 q + k = beam entry k of utterance q --------------------------------
    __builtin_amdgcn_s_setprio(PDT_CONSUMER_PRIO);
    int Tn_q;
    {
      const int q = lane >> 4;
      Tn_q = n0 + q < a.N ? frames_of(n0 + q) : 0;
    }
    const int Tmax = max(max(__builtin_amdgcn_readlane(Tn_q, 0), __builtin_amdgcn_readlane(Tn_q, 16)),
                         max(__builtin_amdgcn_readlane(Tn_q, 32), __builtin_amdgcn_readlane(Tn_q, 48)));
    // :1097-1105: one empty prefix with all the mass on "ends in blank"
    float nb = (lane & 15) == 0 ? 0.0f : -PDT_INF, b = (lane & 15) == 0 ? 1.0f : -PDT_INF;
    int last = 0, len = 0, node = -1, origin = lane & 15;
    unsigned isp = (lane & 15) == 0 ? 1u : 0u;
    int dpar = -1;            // beam entry that is my prefix minus its last token, if the beam holds it
    unsigned dch = 0u;        // beam entries that are my prefix plus one token ...
    int ct0 = -1, ct1 = -1;   // ... and the tokens of the first two of them (-1: none)
    int fail_score = 0, full_mode = 0;  // (row-uniform) the short-list feedback of ctc_search.hip
    int nx_swapped = 0;                 // (row-uniform) which next-token table is the current one
    const int K = W;                    // t >= 1: K' = W, K = min(W, W (V + 1)) (_decoding.py:775)
    const int M = min(V, 2 * W);

#ifdef PDT_STAMPS
    unsigned pdt_stamp_acc[14] = {0};  // wave-uniform: scalar registers
#endif
    int sl = 0;
    for (int t = 0; t < Tmax; ++t, sl = sl + 1 == NS ? 0 : sl + 1) {
      PDT_STAMP_BEGIN;
      // (laundered: inside the frame loop nothing derived from the lane index is loop-invariant to
      // the compiler, so addresses and masks are recomputed where used instead of hoisted, kept
      // live across the whole loop and spilled -- ctc_frame.hpp has the same device)
      int lq = lane;
      asm volatile("" : "+v"(lq));
      const int q = lq >> 4, k = lq & 15, rowbase = lq & 48;
      const int64_t nq = min(n0 + q, (int64_t)a.N - 1);
      const int ubq = q * pl.utt_bytes;  // byte offset of my utterance's LDS
      const int nxo = ubq + (nx_swapped ? pl.nxt_b : pl.nxt_a), nxn = ubq + (nx_swapped ? pl.nxt_a : pl.nxt_b);
      const bool on = t < Tn_q;
      const int slot = ubq + sl * pl.slot_bytes;
      const float *p = reinterpret_cast<const float *>(smem + slot);
      const int *tl_tok = reinterpret_cast<const int *>(smem + slot + pl.row_floats * 4);
      const float *tl_p = reinterpret_cast<const float *>(tl_tok + PDT_WAVE);
      const unsigned char *pos = reinterpret_cast<const unsigned char *>(tl_p + PDT_WAVE);
      const float *hdr = reinterpret_cast<const float *>(pos + pl.pos_bytes);
      const int lastc = min(max(last, 0), V - 1);
      const float tot = nb + b;
      const bool valid = k < W && tot > -PDT_INF;

      // ---- everything the frame reads first, in ONE batch behind the producer's flag -------------
      // (the LDS serves a wave's requests in order, and the producer's stores precede its flag: if
      // the flag read says "ready" the reads issued after it saw the frame.  The compiler barrier
      // keeps that order without the full wait an acquire would put between them.)
      const int *rdy = reinterpret_cast<const int *>(smem) + ((ubq + pl.flags) >> 2) + 1;
      float inv, p_blank_raw, p_last_raw, nb_p, b_p;
      int list_len, last_p;
      unsigned qpos, qc0, qc1;
      const int par = rowbase + max(dpar, 0);
      bool slept = false;
      for (;;) {
        const int have = __hip_atomic_load(rdy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        inv = hdr[0];
        list_len = __float_as_int(hdr[2]);
        p_blank_raw = p[V];
        p_last_raw = p[lastc];
        qpos = pos[lastc];
        qc0 = pos[max(ct0, 0)];
        qc1 = pos[max(ct1, 0)];
        nb_p = shfl_f(nb, par);
        b_p = shfl_f(b, par);
        last_p = shfl_i(lastc, par);
        if (__ballot(on && have <= t) == 0ull) break;
        // the producer is behind: wait at low priority
        if (!slept) __builtin_amdgcn_s_setprio(0);
        slept = true;
        __builtin_amdgcn_s_sleep(PDT_SPIN_SLEEP);
      }
      if (slept) __builtin_amdgcn_s_setprio(PDT_CONSUMER_PRIO);
      PDT_STAMP(0);

      // a beam whose largest mass (entry 0) has underflowed to 0 stays as it is (ctc_search.hip)
#ifdef PDT_SKIP_CONSUMER  // diagnostic build: producer-side cost alone
      const bool run = false;
#else
      const u64 dead_rows = __ballot(k == 0 && !(tot > 0.0f));
      const bool run = on && ((unsigned)(dead_rows >> rowbase) & 1u) == 0u;
#endif
      const bool lean = run && t > 0;
      bool commit_row = false;
      unsigned tau_row = 0u;  // my row's K-th lean key as a lower bound for the full tiers (0: none)

      if (__ballot(lean)) {
        const float p_blank = p_blank_raw * inv;
        const float pl_ = p_last_raw * inv;  // non-extension probability of my last token
        const float B = tot * p_blank;
        float NB = nb * pl_;
        const int c_list = min(max(list_len, 1), M);  // (>= 1 for every produced frame; rows without one only idle along)
        const bool full_list = c_list >= M;
        unsigned avail = (c_list >= 32 ? ~0u : ((1u << c_list) - 1u)) & ~(unsigned)(1ull << (qpos & 63u));
        bool s1_open = valid;
        // ---- merge (:804-837): an extension that equals a beam prefix feeds that prefix ... ------
        {
          const float w = (lastc == last_p ? 0.0f : nb_p) + b_p;
          const float add = w * pl_;
          NB = dpar >= 0 ? NB + add : NB;
        }
        // ... and its parent loses that token: the list entry, or its last-token stream
        if (ct0 >= 0) {
          avail &= ~(unsigned)(1ull << (qc0 & 63u));  // (0xFF, not listed: falls off the low word)
          if (ct0 == lastc) s1_open = false;
        }
        if (ct1 >= 0) {
          avail &= ~(unsigned)(1ull << (qc1 & 63u));
          if (ct1 == lastc) s1_open = false;
        }
        if (__ballot(lean && (dch & (dch - 1u) & ((dch & (dch - 1u)) - 1u)) != 0u)) {
          // three or more children somewhere: the ones beyond the two in registers, one by one
          const int mine = (int)qpos | (lastc << 8);
          unsigned d = dch & (dch - 1u);
          d &= d - 1u;
          if (!lean) d = 0u;
          unsigned rm = 0u;
          while (__ballot(d != 0u)) {
            const int c = d ? __builtin_ctz(d) : 0;
            const int pk = shfl_i(mine, rowbase + c);
            if (d) {
              rm |= (unsigned)(1ull << ((unsigned)pk & 63u));  // (0xFF, not listed: bit 63 falls off the low word)
              if ((pk >> 8) == lastc) s1_open = false;
            }
            d &= d - 1u;
          }
          avail &= ~rm;
        }
        PDT_STAMP(1);
        // ---- the four lean candidates of my prefix ----------------------------------------------
        const float m1 = b * pl_;  // stream 1: my last token (:784-789)
        const float m2 = NB + B;   // stream 2: not extending (:842-845)
        const unsigned av0 = avail, av1 = av0 & (av0 - 1u), av2 = av1 & (av1 - 1u);
        // a short list may end before an entry: the candidate is then an UPPER BOUND of the hidden
        // entry (the list's last probability, key + 1); a bound among the winners fails the row
        const bool hid0 = !full_list && av0 == 0u, hid1 = !full_list && av1 == 0u;
        const int j0 = av0 ? __builtin_ctz(av0) : (hid0 ? c_list - 1 : 0);
        const int j1 = av1 ? __builtin_ctz(av1) : (hid1 ? c_list - 1 : 0);
        const int j2 = av2 ? __builtin_ctz(av2) : c_list - 1;
        const int tok0 = tl_tok[j0], tok1 = tl_tok[j1];
        const float p0 = tl_p[j0], p1 = tl_p[j1], p2 = tl_p[j2];
        const bool has0 = valid && (av0 != 0u || hid0), has1 = valid && (av1 != 0u || hid1);
        const unsigned key0 = has0 ? fkey_nonneg(tot * p0) + (hid0 ? 1u : 0u) : 0u;
        const unsigned key1 = has1 ? fkey_nonneg(tot * p1) + (hid1 ? 1u : 0u) : 0u;
        const unsigned key2 = (valid && s1_open) ? fkey_nonneg(m1) : 0u;
        const unsigned key3 = valid ? fkey_nonneg(m2) : 0u;
        // my third entry (not a candidate here): if my second wins, it must not beat the K-th winner
        unsigned key_e2 = 0u;
        if (valid) key_e2 = av2 != 0u ? fkey_nonneg(tot * p2) : (!full_list ? fkey_nonneg(tot * p2) + 1u : 0u);
        // records the winners read back after the sort (in the next-token table not in use)
        {
          int2 *cand = reinterpret_cast<int2 *>(smem + nxn);
          cand[k] = make_int2((int)key0, tok0 | (hid0 ? (int)0x80000000u : 0));
          cand[16 + k] = make_int2((int)key1, tok1 | (hid1 ? (int)0x80000000u : 0));
          cand[32 + k] = make_int2((int)key2, lastc);
          cand[48 + k] = make_int2((int)key3, lastc);
          int4 *srec = reinterpret_cast<int4 *>(smem + nxn + 512);
          srec[2 * k] = make_int4(__float_as_int(NB), __float_as_int(B), lastc, len | (origin << 24));
          srec[2 * k + 1] = make_int4(node, (int)isp, (int)key_e2, 0);
        }
        PDT_STAMP(2);
        // ---- best 16 of the 64: rounded 32-bit keys with the candidate's id in the freed bits ----
        auto rounded = [&](const unsigned key, const int id) {
          return key ? (((key + 63u) & ~63u) | (unsigned)(63 - id)) : (unsigned)(63 - id);
        };
        unsigned r0 = rounded(key0, k), r1 = rounded(key1, 16 + k), r2 = rounded(key2, 32 + k), r3 = rounded(key3, 48 + k);
        const unsigned lo1 = (lq & 1) ? 0u : 0xFFFFFFFFu, lo2 = (lq & 2) ? 0u : 0xFFFFFFFFu;
        const unsigned f17 = row_top16(r0, r1, r2, r3, lo1, lo2);
        // rank i sits in lane i / 4, register i % 4: through LDS to lane i (+ the 17th behind them)
        unsigned *tb = reinterpret_cast<unsigned *>(smem + ubq + pl.tbuf);
        reinterpret_cast<uint4 *>(tb)[k] = make_uint4(r0, r1, r2, r3);
        tb[16 + k] = f17;  // (lane 0 last: tb[16]; the other lanes write what nobody reads)
        const unsigned st = tb[k], st_next = tb[k + 1], kth_st = tb[K - 1];
        PDT_STAMP(3);
        // ---- winners ---------------------------------------------------------------------------------
        const int id = 63 - (int)(st & 63u);
        const int srck = id & 15, reg = id >> 4;
        const bool isw = k < K && (st >> 6) != 0u;
        const int2 wc = reinterpret_cast<const int2 *>(smem + nxn)[id];
        const int4 sa = reinterpret_cast<const int4 *>(smem + nxn + 512)[2 * srck];
        const int4 sb = reinterpret_cast<const int4 *>(smem + nxn + 512)[2 * srck + 1];
        const bool tie = isw && (st >> 6) == (st_next >> 6);
        const bool bound = isw && wc.y < 0;
        // the K-th winner's bucket holds the keys (r - 64, r]: a third entry at or above r - 63 may win
        const unsigned kth = kth_st >> 6;
        const unsigned kth_low = kth ? (kth << 6) - 63u : 0u;
        const bool third = isw && reg == 1 && sb.z != 0 && (unsigned)sb.z >= kth_low;
        {  // (not when a bound ranks among the first K: that is no real candidate)
          const u64 bounds = __ballot(bound);
          tau_row = (((unsigned)(bounds >> rowbase) & 0xFFFFu) == 0u) ? (kth_low ? kth_low : 1u) : 0u;
        }
        const u64 failing = __ballot(lean && (tie || bound || third));
        const bool row_fails = ((unsigned)(failing >> rowbase) & 0xFFFFu) != 0u;
        commit_row = lean && !row_fails;
        PDT_STATN(5, __popcll(__ballot(lean && row_fails && k == 0)));
        PDT_STATN(6, __popcll(__ballot(lean && k == 0)));
        // ---- new beam entry k (:868-880) -----------------------------------------------------------
        const bool is_ext = reg != 3;
        const int new_tok = wc.y & 0x7fffffff;
        const int len_s = sa.w & 0xFFFFFF, node_s = sb.x;
        const float nw_nb = !isw ? -PDT_INF : (is_ext ? fkey_nonneg_inv((unsigned)wc.x) : __int_as_float(sa.x));
        const float nw_b = !isw ? -PDT_INF : (is_ext ? 0.0f : __int_as_float(sa.y));
        const int nw_last = !isw ? 0 : (is_ext ? new_tok : sa.z);
        const int nw_len = !isw ? 0 : len_s + (is_ext ? 1 : 0);
        const int nw_node = !isw ? -1 : (is_ext ? t * W + k : node_s);
        const int nw_origin = !isw ? origin : (int)((unsigned)sa.w >> 24);
        const bool upd = commit_row && isw;
        int2 *trie_q = a.trie + nq * (int64_t)a.T * W;
        if (upd && is_ext) trie_q[t * W + k] = make_int2(node_s, new_tok);
        PDT_STAMP(4);
        // ---- is-prefix relation and next-token table of the new beam (:883-898) ------------------
        // chm[j] = new entries that descend from old entry j; entry a visits the union over the old
        // entries its source was a prefix of
        unsigned *chm = reinterpret_cast<unsigned *>(smem + ubq + pl.chm);
        int4 *info = reinterpret_cast<int4 *>(smem + ubq + pl.info);
        chm[k] = fresh_zero();
        if (upd) {
          atomicOr(&chm[srck], 1u << k);
          info[k] = make_int4(new_tok, len_s | (srck << 20) | ((is_ext ? 1 : 0) << 28), node_s, -1);
        }
        unsigned nw_isp = 0u, nw_dch = 0u;
        int nw_ct0 = -1, nw_ct1 = -1;
        bool need_walk = false;
        int *nxt_new = reinterpret_cast<int *>(smem + nxn);
        const int *nxt_old = reinterpret_cast<const int *>(smem + nxo);
        if (upd) {
          unsigned cnd = 0u;
          for (unsigned m = (unsigned)sb.y; m; m &= m - 1u) cnd |= chm[__builtin_ctz(m)];
          cnd &= ~(1u << k);
          nw_isp = 1u << k;
          while (cnd) {
            const int bb = __builtin_ctz(cnd);
            cnd &= cnd - 1u;
            const int4 ib = info[bb];
            const int tok_b = ib.x, lenB = ib.y & 0xFFFFF, src_b = (ib.y >> 20) & 0xFF;
            const bool ext_b = (ib.y >> 28) & 1;
            const int len_b = lenB + (ext_b ? 1 : 0);
            if (nw_len > len_b) continue;
            int tok_at;  // token of new prefix bb at position len_s (the length of my source prefix)
            if (lenB > len_s)
              tok_at = nxt_old[srck * W + src_b];
            else
              tok_at = ext_b ? tok_b : -1;  // lenB == len_s
            if (is_ext && tok_at != new_tok) continue;
            nw_isp |= 1u << bb;
            if (nw_len < len_b) {  // strict prefix: the token that follows me inside bb
              int nx;
              if (!is_ext) {
                nx = tok_at;
              } else if (lenB == len_s + 1) {
                nx = tok_b;
              } else {
                nx = -(2 + bb);  // deeper than the table reaches: resolved below by a trie walk
                need_walk = true;
              }
              nxt_new[k * W + bb] = nx;
              if (nw_len + 1 == len_b) {  // my direct child (its last token is nx); I am its direct parent
                if (nw_dch == 0u) nw_ct0 = nx;
                else if ((nw_dch & (nw_dch - 1u)) == 0u) nw_ct1 = nx;
                nw_dch |= 1u << bb;
                reinterpret_cast<int *>(info)[4 * bb + 3] = k;
              }
            }
          }
        }
        if (__ballot(need_walk)) {
          // rare: a re-created intermediate prefix.  Token of bb at position nw_len = token of the
          // ancestor of bb's source node at depth nw_len + 1.
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#pragma unroll 1
          for (int bb = 0; bb < K; ++bb) {
            if (need_walk && ((nw_isp >> bb) & 1u) && nxt_new[k * W + bb] == -(2 + bb)) {
              const int4 ib = info[bb];
              int nd = ib.z, depth = ib.y & 0xFFFFF, tok = -1;
              while (nd >= 0) {
                const int2 *rec = trie_q + nd;
                const int par_ = __hip_atomic_load(&rec->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tok = __hip_atomic_load(&rec->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (depth == nw_len + 1) break;
                nd = par_;
                --depth;
              }
              nxt_new[k * W + bb] = tok;
              // (a direct child's token was recorded as the placeholder: put the real one)
              if (nw_ct0 == -(2 + bb)) nw_ct0 = tok;
              if (nw_ct1 == -(2 + bb)) nw_ct1 = tok;
            }
          }
        }
        const int nw_dpar = upd ? reinterpret_cast<const int *>(info)[4 * k + 3] : -1;
        if (commit_row) {
          nb = nw_nb; b = nw_b; last = nw_last; len = nw_len; node = nw_node; origin = nw_origin;
          isp = nw_isp; dpar = nw_dpar; dch = nw_dch; ct0 = nw_ct0; ct1 = nw_ct1;
          nx_swapped ^= 1;
        }
      }
      PDT_STAMP(5);
      // rows the lean tier could not decide (and every row at t = 0, K' = 1): one bit per row, at
      // its lane 0
      u64 fbm = __ballot(run && k == 0 && !commit_row);
      bool enough_row = true;  // the list as handed over was enough for my row's frame
      const bool any_fallback = fbm != 0ull;
      while (fbm) {
        const int fl = (int)__builtin_ctzll(fbm);  // lane 0 of the row
        fbm &= fbm - 1ull;
        const int fq = fl >> 4;
        PDT_STAT(7);
        const int from = fl + (lane & 15);
        float i_nb = shfl_f(nb, from), i_b = shfl_f(b, from);
        int i_last = shfl_i(last, from), i_len = shfl_i(len, from), i_node = shfl_i(node, from);
        unsigned i_isp = (unsigned)shfl_i((int)isp, from);
        int i_origin = shfl_i(origin, from);
        if (lane >= 16) { i_nb = -PDT_INF; i_b = -PDT_INF; i_last = 0; i_len = 0; i_node = -1; i_isp = 0u; i_origin = lane; }
        const u64 trie_bits = (u64)reinterpret_cast<uintptr_t>(a.trie);
        const int f_ub = fq * pl.utt_bytes;
        const FallbackOut o = packed_fallback(
            i_nb, i_b, i_last, i_len, i_node, i_isp, i_origin, f_ub, f_ub + sl * pl.slot_bytes,
            __builtin_amdgcn_readlane(nxo, fl), __builtin_amdgcn_readlane(nxn, fl), pl.row_floats * 4, pl.pos_bytes,
            pl.chm, pl.info, pl.surv, V, W, t == 0 ? 1 : W, t, a.T,
            (unsigned)__builtin_amdgcn_readlane((int)tau_row, fl), (unsigned)trie_bits, (unsigned)(trie_bits >> 32),
            (int)(n0 + fq));
        // back into the row
        const int back = lane & 15;
        const float r_nb = shfl_f(o.nb, back), r_b = shfl_f(o.b, back);
        const int r_last = shfl_i(o.last, back), r_len = shfl_i(o.len, back), r_node = shfl_i(o.node, back);
        const unsigned r_isp = (unsigned)shfl_i((int)o.isp, back), r_dch = (unsigned)shfl_i((int)o.dch, back);
        const int r_origin = shfl_i(o.origin, back), r_dpar = shfl_i(o.dpar, back);
        // the tokens of the first two direct children: their last tokens
        const unsigned d2 = r_dch & (r_dch - 1u);
        const int c0l = shfl_i(o.last, r_dch ? __builtin_ctz(r_dch) : 0), c1l = shfl_i(o.last, d2 ? __builtin_ctz(d2) : 0);
        if (q == fq) {
          nb = r_nb; b = r_b; last = r_last; len = r_len; node = r_node; isp = r_isp; origin = r_origin;
          dch = r_dch; dpar = r_dpar;
          ct0 = r_dch ? c0l : -1;
          ct1 = d2 ? c1l : -1;
          nx_swapped ^= 1;
          enough_row = o.enough != 0;
        }
      }
      PDT_STAMP(6);
      // ---- hand the slot back, feedback to the producers, checkpoints -----------------------------
      // (every read of the slot has returned; rows whose frames are over have no producer left)
      reinterpret_cast<volatile int *>(smem)[(ubq + pl.flags) >> 2] = t + 1;
      if (any_fallback || __ballot(fail_score != 0 || full_mode != 0)) {
        if (run) {
          // +1 per frame my row had to complete a short list, -1 per frame it did not (0 .. 32);
          // complete lists above 16, short ones again below 4 (ctc_search.hip)
          fail_score = enough_row ? max(fail_score - 1, 0) : min(fail_score + 1, 32);
          const int wf = fail_score > 16 ? 1 : (fail_score < 4 ? 0 : full_mode);
          if (wf != full_mode) {
            full_mode = wf;
            if (k == 0)
              __hip_atomic_store(reinterpret_cast<int *>(smem) + ((ubq + pl.flags) >> 2) + 5, wf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
      if (((t + 1) & ((1 << a.ckpt_shift) - 1)) == 0) {  // checkpoint (CtcArgs::ckpt)
        const int c = ((t + 1) >> a.ckpt_shift) - 1;
        if (on && k < W) {
          a.ckpt[(nq * a.ckpt_count + c) * W + k] = make_int2(node, len | (origin << 24));
          origin = k;
        }
      }
      PDT_STAMP(7);
    }
#ifdef PDT_STAMPS
    if (lane == 0)
      for (int i = 0; i < 14; ++i) atomicAdd(&g_stamps[i], (unsigned long long)pdt_stamp_acc[i]);
#endif
    // ---- outputs (:1188-1200); the final beam goes to LDS for the waves that walk the trie ------
    const int q = lane >> 4, k = lane & 15;
    const int64_t nq = n0 + q;
    const int ubq = q * pl.utt_bytes;
    if (nq < a.N && k < W) {
      a.y_probs[nq * W + k] = nb + b;
      a.y_lens[nq * W + k] = len;
      reinterpret_cast<int4 *>(smem + ubq + pl.nxt_a)[k] = make_int4(node, len, origin, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_s_setprio(0);
  }
  __syncthreads();  // every frame is done: the rings are free, the tries complete

  if (wave < kPackUtts && n0 + wave < a.N) {
    // One chain of len dependent loads per prefix would be ~T global-memory latencies.  Instead
    // (ctc_search.hip): each prefix follows its `origin` links back through the C checkpoints,
    // leaving (node, length) of its ancestor at every checkpoint in LDS (the ring is free now);
    // the (C + 1) x W segments between consecutive checkpoints are walked by all 64 lanes.
    const int64_t n = n0 + wave;
    unsigned char *ub = smem + (size_t)wave * pl.utt_bytes;
    const int Tn = frames_of(n);
    const int C = Tn >> a.ckpt_shift;
    int2 *tab = reinterpret_cast<int2 *>(ub);  // [(C + 1) x W]: fits, see launch_ctc_search_packed
    const int4 fin = reinterpret_cast<const int4 *>(ub + pl.nxt_a)[lane < W ? lane : 0];
    if (lane < W) {
      const bool ok = fin.x >= 0;
      tab[C * W + lane] = make_int2(fin.x, fin.y);
      int cur = fin.z;
      for (int c = C - 1; c >= 0; --c) {
        const int2 *rec = a.ckpt + ((n * a.ckpt_count + c) * W + cur);
        const int nd = __hip_atomic_load(&rec->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int lo = __hip_atomic_load(&rec->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tab[c * W + lane] = ok ? make_int2(nd, lo & 0xFFFFFF) : make_int2(-1, 0);
        cur = (int)((unsigned)lo >> 24);
      }
    }
    wave_sync();
    for (int sg = lane; sg < (C + 1) * W; sg += PDT_WAVE) {
      const int c = sg / W, kk = sg - c * W;
      const int2 top = tab[sg];
      const int stop = c > 0 ? tab[sg - W].y : 0;
      int nd = top.x;
      for (int ps = top.y - 1; ps >= stop && nd >= 0; --ps) {
        const int2 *rec = a.trie + (n * (int64_t)a.T * W + nd);
        const int par = __hip_atomic_load(&rec->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int tok = __hip_atomic_load(&rec->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.y[((int64_t)ps * a.N + n) * W + kk] = tok;
        nd = par;
      }
    }
    // rows beyond a prefix's length are 0: the kernel writes every element of y
    int lmin = lane < W ? fin.y : 0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) lmin = min(lmin, shfl_i(lmin, lane ^ off));
    for (int f = lmin * W + lane; f < a.S * W; f += PDT_WAVE) {
      const int ps = f / W, kk = f - ps * W;
      if (ps >= tab[C * W + kk].y) a.y[((int64_t)ps * a.N + n) * W + kk] = 0;
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------
// The packed form serves beams of up to 16 prefixes over rows of up to 511 tokens; PDT_CTC_PACKED=0
// keeps the one-utterance-per-consumer form (comparisons, tests of both).
// Measured (MI355X, N = 4096, T = 512, V = 256, K = 16): 2.35-2.40 ms against 1.93 for the
// one-utterance form; one workgroup per CU (N = 1024) 1.42 against 1.2.  The consumer's frame is
// ~800 instructions for four utterances against ~550 for one, but a lone wave issues one
// instruction per ~5 cycles and waits out ~9 dependent LDS round trips per frame, and four
// producers on its SIMD stretch both -- so the form is opt-in (PDT_CTC_PACKED=1) and the default
// stays the one-utterance form.  DESIGN.md, section 4.3b, has the numbers.
bool ctc_packed_applies(int V, int W) {
  if (W > 16 || V + 1 > 8 * PDT_WAVE) return false;
  const char *e = std::getenv("PDT_CTC_PACKED");
  return e && e[0] == '1';
}

// which instantiation serves (V, W): the fixed-layout one for the byte-sized vocabularies at the
// full width, the general one otherwise
static bool packed_fixed(int V, int W) { return V / PDT_WAVE == 4 && W == 16; }

// ring depth of the general form: the deepest of 4 / 3 / 2 slots that still lets four workgroups
// share a CU's LDS (all 1024 workgroups of a 4096-utterance launch resident at once); else three
PackedLayout plan_ctc_packed(int V, int W) {
  if (packed_fixed(V, W)) return packed_layout_fixed<4>();
  if (const char *e = std::getenv("PDT_CTC_STAGES")) {  // (experiments)
    const int ns = std::atoi(e);
    if (ns >= 2 && ns <= 4) return packed_layout(V, ns);
  }
  for (int ns = 4; ns >= 2; --ns) {
    const PackedLayout p = packed_layout(V, ns);
    if ((size_t)p.utt_bytes * kPackUtts * 4 <= 160 * 1024) return p;
  }
  return packed_layout(V, 3);
}

int ctc_packed_ring_slots(int V, int W) { return plan_ctc_packed(V, W).nstage; }

template <int NT, int WC>
static int launch_packed(const CtcArgs &a, const PackedLayout &pl, hipStream_t stream) {
  const size_t smem = (size_t)pl.utt_bytes * kPackUtts;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_search_packed_kernel<NT, WC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  const unsigned grid = (unsigned)((a.N + kPackUtts - 1) / kPackUtts);
  hipLaunchKernelGGL((ctc_search_packed_kernel<NT, WC>), dim3(grid), dim3(64 * (kPackUtts + 1)), smem, stream, a, pl);
  return (int)hipGetLastError();
}

int launch_ctc_search_packed(CtcArgs a, hipStream_t stream) {
  const PackedLayout pl = plan_ctc_packed(a.V, a.W);
  // checkpoint spacing from this form's ring (the table of the output walk overlays it)
  const size_t ring = (size_t)pl.slot_bytes * pl.nstage;
  int sh = 5;
  while (((size_t)(a.T >> sh) + 1) * a.W * sizeof(int2) > ring) ++sh;
  a.ckpt_shift = sh;
  a.ckpt_count = (a.T >> sh) + 1;
  if (packed_fixed(a.V, a.W)) return launch_packed<4, 16>(a, pl, stream);
  return launch_packed<-1, 0>(a, pl, stream);
}

}  // namespace pdt

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
