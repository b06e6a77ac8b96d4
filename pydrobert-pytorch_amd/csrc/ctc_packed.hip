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
//     q (nb, b, last token, length, trie node, is-prefix row, direct parent / children);
//   * the lean tier's four candidates of a prefix -- its two best available list entries, its
//     last-token stream, its non-extension -- sit in four REGISTERS of its lane; the best 16
//     (+ the 17th, for the tie test) come from four 16-lane sorts and three top-16 merges, all
//     DPP row operations, four utterances per instruction;
//   * a row whose frame the lean tier cannot decide (a third-best extension, a short list's bound
//     or a rounded-key tie among the winners; t = 0) is moved into lanes 0-15 of a scratch beam and
//     run through ctc_frame() -- the complete per-utterance routine of ctc_search.hip -- on that
//     utterance's own LDS tables, then moved back (about 3 % of the frames of an utterance);
//   * after the last frame the four producer waves read the prefixes off the trie, one utterance
//     each (the checkpointed walk of ctc_search.hip).
#include "ctc_ring.hpp"

#include <cstdlib>

namespace pdt {

// ---- LDS of one utterance ----------------------------------------------------------------
//   [ring: nstage slots, RingLayout geometry | nxt table A | nxt table B | chm | info | producer
//    scratch | flags]
// The next-token tables are 16 x 16 whatever the width, because the table not in use doubles as
// the consumer's per-frame scratch (candidate records 64 x 8 B + source records 16 x 32 B).
struct PackedLayout {
  RingLayout rl;
  int nxt_a, nxt_b, chm, info, surv, flags, utt_bytes;
};
constexpr int kPackNxtBytes = 16 * 16 * 4;
constexpr int kPackUtts = 4;

__host__ __device__ inline PackedLayout packed_layout(int V, int nstage) {
  PackedLayout p;
  p.rl = ring_layout(V, 16, nstage, kPackUtts, 1);
  int off = p.rl.slot_bytes * nstage;
  off = (off + 15) & ~15;
  p.nxt_a = off; off += kPackNxtBytes;
  p.nxt_b = off; off += kPackNxtBytes;
  p.chm = off; off += 16 * 4;
  p.info = off; off += 16 * 16;                 // (token, packed source word, source node, direct parent)
  p.surv = off; off += PDT_SURV_CAP * 8;        // producer scratch (short-list keys / selection survivors)
  p.flags = off; off += 32;                     // consumed, ready[4], want_full
  p.utt_bytes = (off + 15) & ~15;
  return p;
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
constexpr int kQuadX1 = 0xB1, kQuadX2 = 0x4E, kHalfMirror = 0x141, kMirror = 0x140,
              kRor8 = 0x128, kShl4 = 0x104, kShr4 = 0x114, kShl1 = 0x101;

// compare-exchange of a descending sort whose "keep the larger" lanes are whole 4-lane banks:
// the bank mask of the DPP operand does the selection, 2 VALU (max over the lower banks, min over
// the upper ones, both reading the stage's input)
template <int CTRL_LO, int CTRL_HI, int BANK_LO>
__device__ __forceinline__ unsigned cx_bank(unsigned key) {
  const unsigned t = max(key, dpp_u<CTRL_LO, BANK_LO>(key, 0u));
  return min(t, dpp_u<CTRL_HI, 0xf & ~BANK_LO>(key, 0xFFFFFFFFu));
}
__device__ __forceinline__ unsigned cx4(unsigned k) { return cx_bank<kShl4, kShr4, 0x5>(k); }          // l ^ 4
__device__ __forceinline__ unsigned cx8(unsigned k) { return cx_bank<kRor8, kRor8, 0x3>(k); }          // l ^ 8
__device__ __forceinline__ unsigned cx7(unsigned k) { return cx_bank<kHalfMirror, kHalfMirror, 0x5>(k); }  // l ^ 7
__device__ __forceinline__ unsigned cx15(unsigned k) { return cx_bank<kMirror, kMirror, 0x3>(k); }      // l ^ 15

// every row of 16 lanes sorted descending: the bitonic network with flips of wave_select.hpp, its
// bank-granular stages in the 2-VALU form
__device__ __forceinline__ unsigned row_sort16(unsigned k) {
  k = cmpx_stage<1, 1>(k);
  k = cmpx_stage<3, 2>(k);
  k = cmpx_stage<1, 1>(k);
  k = cx7(k);
  k = cmpx_stage<2, 2>(k);
  k = cmpx_stage<1, 1>(k);
  k = cx15(k);
  k = cx4(k);
  k = cmpx_stage<2, 2>(k);
  k = cmpx_stage<1, 1>(k);
  return k;
}
// a, b: rows sorted descending.  Returns the 16 largest of the 32 sorted descending; `dropped`
// takes the elementwise minima (the 16 that did not make it, unordered).
__device__ __forceinline__ unsigned row_merge_top16(unsigned a, unsigned b, unsigned &dropped) {
  const unsigned br = dpp_mov<kMirror>(b);
  unsigned k = max(a, br);  // bitonic
  dropped = min(a, br);
  k = cx8(k);
  k = cx4(k);
  k = cmpx_stage<2, 2>(k);
  k = cmpx_stage<1, 1>(k);
  return k;
}
__device__ __forceinline__ unsigned row_allmax(unsigned v) {
  v = max(v, dpp_mov<kQuadX1>(v));
  v = max(v, dpp_mov<kQuadX2>(v));
  v = max(v, dpp_mov<kHalfMirror>(v));
  return max(v, dpp_mov<kMirror>(v));
}
__device__ __forceinline__ unsigned row_allmin(unsigned v) {
  v = min(v, dpp_mov<kQuadX1>(v));
  v = min(v, dpp_mov<kQuadX2>(v));
  v = min(v, dpp_mov<kHalfMirror>(v));
  return min(v, dpp_mov<kMirror>(v));
}

#ifdef PDT_STATS
#define PDT_STATN(i, n) do { if (lane_id() == 0) atomicAdd(&g_stats[i], (unsigned long long)(n)); } while (0)
#else
#define PDT_STATN(i, n) do {} while (0)
#endif

// ---- producer: one wave per utterance, the register-resident row pass of ctc_search.hip ------
// (V + 1 <= 512: the row sits in eight prefetch registers; short exact top-c lists from a guessed
// threshold while the consumer's lean tier decides most frames.)
template <int NT>
__device__ __forceinline__ void packed_producer(const CtcArgs &a, const PackedLayout &pl, unsigned char *ub,
                                                const int64_t n, const int Tn) {
  const RingLayout &rl = pl.rl;
  const int lane = lane_id();
  const int V = a.V, W = a.W, NS = rl.nstage;
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
    const float inv = __builtin_fmaf(__builtin_fmaf(-s, inv0, 1.0f), inv0, inv0);
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

// ---- the kernel -------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(320, 5) ctc_search_packed_kernel(const CtcArgs a, const PackedLayout pl) {
  extern __shared__ __align__(16) unsigned char smem[];
  const RingLayout &rl = pl.rl;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // 0-3 producers, 4 consumer
  const int64_t n0 = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * kPackUtts;
  const int V = a.V, W = a.W, NS = rl.nstage;
  auto frames_of = [&](int64_t n) {
    return min(a.S, a.lens ? (int)min((int64_t)a.T, max((int64_t)0, a.lens[n])) : a.T);
  };

  if (wave < kPackUtts) {
    unsigned char *ub = smem + (size_t)wave * pl.utt_bytes;
    for (int sl = 0; sl < NS; ++sl) {
      unsigned char *pos = ub + (size_t)sl * rl.slot_bytes + (size_t)rl.row_floats * 4 + PDT_WAVE * 8;
      for (int v = lane; v < rl.pos_bytes; v += PDT_WAVE) pos[v] = 0xFF;
    }
    if (lane <= 5)  // consumed, ready[0 .. 4), want_full
      __hip_atomic_store(reinterpret_cast<int *>(ub + pl.flags) + lane, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();

  if (wave < kPackUtts) {
    const int64_t n = n0 + wave;
    if (n < a.N) packed_producer<NT>(a, pl, smem + (size_t)wave * pl.utt_bytes, n, frames_of(n));
  } else {
    // ---- consumer: lane 16 q + k = beam entry k of utterance q --------------------------------
    __builtin_amdgcn_s_setprio(PDT_CONSUMER_PRIO);
    const int q = lane >> 4, k = lane & 15, rowbase = lane & 48;
    const bool row_exists = n0 + q < a.N;
    const int64_t nq = row_exists ? n0 + q : 0;
    const int Tn_q = row_exists ? frames_of(nq) : 0;
    const int ubq = q * pl.utt_bytes;  // byte offset of my utterance's LDS
    int nxo = ubq + pl.nxt_a, nxn = ubq + pl.nxt_b;
    const int Tmax = max(max(__builtin_amdgcn_readlane(Tn_q, 0), __builtin_amdgcn_readlane(Tn_q, 16)),
                         max(__builtin_amdgcn_readlane(Tn_q, 32), __builtin_amdgcn_readlane(Tn_q, 48)));
    // :1097-1105: one empty prefix with all the mass on "ends in blank"
    float nb = k == 0 ? 0.0f : -PDT_INF, b = k == 0 ? 1.0f : -PDT_INF;
    int last = 0, len = 0, node = -1, origin = k;
    unsigned isp = k == 0 ? 1u : 0u;
    int dpar = -1;       // beam entry that is my prefix minus its last token, if the beam holds it
    unsigned dch = 0u;   // beam entries that are my prefix plus one token
    int fail_score = 0, full_mode = 0;  // (row-uniform) the short-list feedback of ctc_search.hip
    const int K = W;                    // t >= 1: K' = W, K = min(W, W (V + 1)) (_decoding.py:775)
    const int M = min(V, 2 * W);
    int2 *trie_q = a.trie + nq * (int64_t)a.T * W;

    int sl = 0;
    for (int t = 0; t < Tmax; ++t, sl = sl + 1 == NS ? 0 : sl + 1) {
      const bool on = t < Tn_q;
      {
        const int *rdy = reinterpret_cast<const int *>(smem) + ((ubq + pl.flags) >> 2) + 1;
        if (__ballot(on && __hip_atomic_load(rdy, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= t)) {
          __builtin_amdgcn_s_setprio(0);
          do {
            __builtin_amdgcn_s_sleep(PDT_SPIN_SLEEP);
          } while (__ballot(on && __hip_atomic_load(rdy, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= t));
          __builtin_amdgcn_s_setprio(PDT_CONSUMER_PRIO);
        }
      }
      const int slot = ubq + sl * rl.slot_bytes;
      const float *p = reinterpret_cast<const float *>(smem + slot);
      const int *tl_tok = reinterpret_cast<const int *>(smem + slot + rl.row_floats * 4);
      const float *tl_p = reinterpret_cast<const float *>(tl_tok + PDT_WAVE);
      const unsigned char *pos = reinterpret_cast<const unsigned char *>(tl_p + PDT_WAVE);
      const float *hdr = reinterpret_cast<const float *>(pos + rl.pos_bytes);

      const float tot = nb + b;
      const bool valid = k < W && tot > -PDT_INF;
      // a beam whose largest mass has underflowed to 0 stays as it is (ctc_search.hip)
      const unsigned alive = row_allmax(valid ? __float_as_uint(tot) : 0u);
      const bool run = on && alive != 0u;
      const bool lean = run && t > 0;
      bool commit_row = false;
      float nw_nb = nb, nw_b = b;
      int nw_last = last, nw_len = len, nw_node = node, nw_origin = origin, nw_dpar = dpar;
      unsigned nw_isp = isp, nw_dch = dch;

      if (__ballot(lean)) {
        const float inv = hdr[0];
        const int list_len = __float_as_int(hdr[2]);
        const int lastc = min(max(last, 0), V - 1);
        const float p_blank = p[V] * inv;
        const float pl_ = p[lastc] * inv;  // non-extension probability of my last token
        const unsigned qpos = pos[lastc];
        const float B = tot * p_blank;
        float NB = nb * pl_;
        const int c_list = min(max(list_len, 1), M);  // (>= 1 for every produced frame; rows without one only idle along)
        const bool full_list = c_list >= M;
        unsigned avail = (c_list >= 32 ? ~0u : ((1u << c_list) - 1u)) & ~(unsigned)(1ull << (qpos & 63u));
        bool s1_open = valid;
        // ---- merge (:804-837): an extension that equals a beam prefix feeds that prefix --------
        if (__ballot(lean && dpar >= 0)) {
          const int par = rowbase + max(dpar, 0);
          const float nb_p = shfl_f(nb, par), b_p = shfl_f(b, par);
          const int last_p = shfl_i(lastc, par);
          if (dpar >= 0) NB = NB + ((lastc == last_p ? 0.0f : nb_p) + b_p) * pl_;
          // ... and its parent loses that token: the list entry, or its last-token stream
          const int mine = (int)qpos | (lastc << 8);
          unsigned d = lean ? dch : 0u, rm = 0u;
          while (__ballot(d != 0u)) {
            const int c = d ? __builtin_ctz(d) : 0;
            const int pk = shfl_i(mine, rowbase + c);
            if (d) {
              const unsigned jc = (unsigned)pk & 0xFFu;
              rm |= (unsigned)(1ull << (jc & 63u));  // (0xFF, not listed: falls off the low word)
              if ((pk >> 8) == lastc) s1_open = false;
            }
            d &= d - 1u;
          }
          avail &= ~rm;
        }
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
        // ---- best 16 of the 64: rounded 32-bit keys with the candidate's id in the freed bits ----
        auto rounded = [&](const unsigned key, const int id) {
          return key ? (((key + 63u) & ~63u) | (unsigned)(63 - id)) : (unsigned)(63 - id);
        };
        unsigned d01, d23, d03;
        const unsigned s0 = row_sort16(rounded(key0, k)), s1 = row_sort16(rounded(key1, 16 + k));
        const unsigned s2 = row_sort16(rounded(key2, 32 + k)), s3 = row_sort16(rounded(key3, 48 + k));
        const unsigned s01 = row_merge_top16(s0, s1, d01), s23 = row_merge_top16(s2, s3, d23);
        const unsigned st = row_merge_top16(s01, s23, d03);
        const unsigned seventeenth = row_allmax(max(max(d01, d23), d03));
        // ---- winners ---------------------------------------------------------------------------------
        const int id = 63 - (int)(st & 63u);
        const int srck = id & 15, reg = id >> 4;
        const bool isw = k < K && (st >> 6) != 0u;
        const int2 wc = reinterpret_cast<const int2 *>(smem + nxn)[id];
        const int4 sa = reinterpret_cast<const int4 *>(smem + nxn + 512)[2 * srck];
        const int4 sb = reinterpret_cast<const int4 *>(smem + nxn + 512)[2 * srck + 1];
        const unsigned st_next = k == 15 ? seventeenth : dpp_u<kShl1>(st, 0u);
        const bool tie = isw && (st >> 6) == (st_next >> 6);
        const bool bound = isw && wc.y < 0;
        // the K-th winner's bucket holds the keys (r - 64, r]: a third entry at or above r - 63 may win
        const unsigned kth = row_allmin(k < K ? st : 0xFFFFFFFFu) >> 6;
        const unsigned kth_low = kth ? (kth << 6) - 63u : 0u;
        const bool third = isw && reg == 1 && sb.z != 0 && (unsigned)sb.z >= kth_low;
        const u64 failing = __ballot(lean && (tie || bound || third));
        const bool row_fails = ((unsigned)(failing >> rowbase) & 0xFFFFu) != 0u;
        commit_row = lean && !row_fails;
        PDT_STATN(5, __popcll(__ballot(lean && row_fails && k == 0)));
        PDT_STATN(6, __popcll(__ballot(lean && k == 0)));
        // ---- new beam entry k (:868-880) -----------------------------------------------------------
        const bool is_ext = reg != 3;
        const int new_tok = wc.y & 0x7fffffff;
        const int len_s = sa.w & 0xFFFFFF, node_s = sb.x;
        nw_nb = !isw ? -PDT_INF : (is_ext ? fkey_nonneg_inv((unsigned)wc.x) : __int_as_float(sa.x));
        nw_b = !isw ? -PDT_INF : (is_ext ? 0.0f : __int_as_float(sa.y));
        nw_last = !isw ? 0 : (is_ext ? new_tok : sa.z);
        nw_len = !isw ? 0 : len_s + (is_ext ? 1 : 0);
        nw_node = !isw ? -1 : (is_ext ? t * W + k : node_s);
        nw_origin = !isw ? origin : (int)((unsigned)sa.w >> 24);
        const bool upd = commit_row && isw;
        if (upd && is_ext) trie_q[t * W + k] = make_int2(node_s, new_tok);
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
        nw_isp = 0u;
        nw_dch = 0u;
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
              if (nw_len + 1 == len_b) {  // my direct child; I am its direct parent
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
            }
          }
        }
        nw_dpar = upd ? reinterpret_cast<const int *>(info)[4 * k + 3] : -1;
      }
      if (commit_row) {
        nb = nw_nb; b = nw_b; last = nw_last; len = nw_len; node = nw_node; origin = nw_origin;
        isp = nw_isp; dpar = nw_dpar; dch = nw_dch;
        const int tmp = nxo; nxo = nxn; nxn = tmp;
      }
      // rows the lean tier could not decide (and every row at t = 0, K' = 1): one bit per row, at
      // its lane 0
      u64 fbm = __ballot(run && k == 0 && !commit_row);
      bool enough_row = true;  // the list as handed over was enough for my row's frame
      while (fbm) {
        const int fl = (int)__builtin_ctzll(fbm);  // lane 0 of the row
        fbm &= fbm - 1ull;
        const int fq = fl >> 4;
        PDT_STAT(7);
        // ---- one utterance through the complete per-utterance frame routine -------------------------
        unsigned char *ub = smem + (size_t)fq * pl.utt_bytes;
        unsigned char *sb_ = ub + (size_t)sl * rl.slot_bytes;
        const int from = fl + (lane & 15);
        Beam fb;
        fb.nb = shfl_f(nb, from);
        fb.b = shfl_f(b, from);
        fb.last = shfl_i(last, from);
        fb.len = shfl_i(len, from);
        fb.node = shfl_i(node, from);
        fb.isp = (unsigned)shfl_i((int)isp, from);
        fb.origin = shfl_i(origin, from);
        if (lane >= 16) { fb.nb = -PDT_INF; fb.b = -PDT_INF; fb.last = 0; fb.len = 0; fb.node = -1; fb.isp = 0u; fb.origin = lane; }
        FrameLds L;
        L.surv = reinterpret_cast<u64 *>(ub + pl.surv);
        L.tl_tok = reinterpret_cast<int *>(sb_ + (size_t)rl.row_floats * 4);
        L.tl_p = reinterpret_cast<float *>(L.tl_tok + PDT_WAVE);
        L.pos = reinterpret_cast<unsigned char *>(L.tl_p + PDT_WAVE);
        L.hdr = reinterpret_cast<float *>(L.pos + rl.pos_bytes);
        L.list_len = __float_as_int(L.hdr[2]);
        L.chm = reinterpret_cast<unsigned *>(ub + pl.chm);
        L.info = reinterpret_cast<int *>(ub + pl.info);
        L.nxt_old = reinterpret_cast<int *>(smem + __builtin_amdgcn_readlane(nxo, fl));
        L.nxt_new = reinterpret_cast<int *>(smem + __builtin_amdgcn_readlane(nxn, fl));
        const int64_t nf = n0 + fq;
        L.trie_u = a.trie + nf * (int64_t)a.T * W;
        int ns_, nt__, nk_;
        const bool enough = ctc_frame<false>(fb, reinterpret_cast<const float *>(sb_), L.hdr[0], V, W, t == 0 ? 1 : W, t,
                                             nf, a, DenseCtx{}, L, ns_, nt__, nk_);
        // back into the row
        const bool mine_row = q == fq;
        const int back = lane & 15;
        const float r_nb = shfl_f(fb.nb, back), r_b = shfl_f(fb.b, back);
        const int r_last = shfl_i(fb.last, back), r_len = shfl_i(fb.len, back), r_node = shfl_i(fb.node, back);
        const unsigned r_isp = (unsigned)shfl_i((int)fb.isp, back);
        const int r_origin = shfl_i(fb.origin, back);
        if (mine_row) {
          nb = r_nb; b = r_b; last = r_last; len = r_len; node = r_node; isp = r_isp; origin = r_origin;
          const int tmp = nxo; nxo = nxn; nxn = tmp;
          enough_row = enough;
        }
        // direct children / parent of the row's new beam, from the is-prefix rows and the lengths
        unsigned dch_r = 0u;
        for (int bb = 0; bb < W; ++bb) {
          const int lb = shfl_i(len, fl + bb);
          if (((isp >> bb) & 1u) && bb != k && lb == len + 1) dch_r |= 1u << bb;
        }
        if (!mine_row) dch_r = 0u;
        int dpar_r = -1;
        for (int bb = 0; bb < W; ++bb) {
          const unsigned db = (unsigned)shfl_i((int)dch_r, fl + bb);
          if ((db >> k) & 1u) dpar_r = bb;
        }
        if (mine_row) { dch = dch_r; dpar = dpar_r; }
      }
      // ---- hand the slot back, feedback to the producers, checkpoints -----------------------------
      if (on && k == 0)
        __hip_atomic_store(reinterpret_cast<int *>(smem) + ((ubq + pl.flags) >> 2), t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
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
      if (((t + 1) & ((1 << a.ckpt_shift) - 1)) == 0) {  // checkpoint (CtcArgs::ckpt)
        const int c = ((t + 1) >> a.ckpt_shift) - 1;
        if (on && k < W) {
          a.ckpt[(nq * a.ckpt_count + c) * W + k] = make_int2(node, len | (origin << 24));
          origin = k;
        }
      }
    }
    // ---- outputs (:1188-1200); the final beam goes to LDS for the waves that walk the trie ------
    if (row_exists && k < W) {
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
    int2 *tab = reinterpret_cast<int2 *>(ub);  // [(C + 1) x W]: fits, see ckpt_shift_for
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
bool ctc_packed_applies(int V, int W) {
  if (W > 16 || V + 1 > 8 * PDT_WAVE) return false;
  const char *e = std::getenv("PDT_CTC_PACKED");
  return !(e && e[0] == '0');
}

// ring depth: the deepest of 4 / 3 / 2 slots that still lets four workgroups share a CU's LDS
// (all 1024 workgroups of a 4096-utterance launch resident at once); else three
PackedLayout plan_ctc_packed(int V) {
  for (int ns = 4; ns >= 2; --ns) {
    const PackedLayout p = packed_layout(V, ns);
    if ((size_t)p.utt_bytes * kPackUtts * 4 <= 160 * 1024) return p;
  }
  return packed_layout(V, 3);
}

int ctc_packed_ring_slots(int V) { return plan_ctc_packed(V).rl.nstage; }

template <int NT>
static int launch_packed(const CtcArgs &a, const PackedLayout &pl, hipStream_t stream) {
  const size_t smem = (size_t)pl.utt_bytes * kPackUtts;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_search_packed_kernel<NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  const unsigned grid = (unsigned)((a.N + kPackUtts - 1) / kPackUtts);
  hipLaunchKernelGGL((ctc_search_packed_kernel<NT>), dim3(grid), dim3(64 * (kPackUtts + 1)), smem, stream, a, pl);
  return (int)hipGetLastError();
}

int launch_ctc_search_packed(CtcArgs a, hipStream_t stream) {
  const PackedLayout pl = plan_ctc_packed(a.V);
  // checkpoint spacing from this form's ring (the table of the output walk overlays it)
  const size_t ring = (size_t)pl.rl.slot_bytes * pl.rl.nstage;
  int sh = 5;
  while (((size_t)(a.T >> sh) + 1) * a.W * sizeof(int2) > ring) ++sh;
  a.ckpt_shift = sh;
  a.ckpt_count = (a.T >> sh) + 1;
  if (a.V / PDT_WAVE == 4) return launch_packed<4>(a, pl, stream);
  return launch_packed<-1>(a, pl, stream);
}

}  // namespace pdt
