// One iteration of BeamSearch.forward as ONE kernel (reference _decoding.py:410-486, with the
// default update_log_probs_for_step):
//   eos bookkeeping (which paths have ended, which batch elements are finished, :413-431),
//   log_softmax of the language model's scores (:441) -- never materialised: a row's maximum and
//   log-sum once, candidates ranked through them --, the eos-mass reallocation of ended paths
//   (:448-458), the top-`width` of the K' x V candidates with their histories (beam_search_advance,
//   :41-155), lengths that do not grow for ended sources (:465-468), and finished batch elements
//   keeping the beam they had (:479-486).
// Against the reference's loop this removes the (N, K', V) passes of log_softmax / masked_fill /
// where, the per-iteration clamp of the whole history and the per-iteration host read: the number
// of unfinished batch elements of every iteration goes to a device array the host looks at every
// few iterations (pydrobert_amd/_decoding.py: BeamSearch._forward_fused).
//
// The history `y` is kept CLAMPED to [0, V - 1] (what the reference hands its language model,
// :434); rows a finished element is padded with are written as clamp(pad_value) and their first
// index recorded in `pad_from`, so the host restores pad_value once, at the end.
//
// One workgroup per batch element, as beam_advance_kernel: the waves take the prefixes' selections
// in turn, wave 0 merges, all waves copy the history.  Ties: lowest flat index k * V + v.
#include "ctc_frame.hpp"
#include "switches.hpp"

namespace pdt {

struct BeamStepArgs {
  const float *scores;  int64_t sc_sn, sc_sk, sc_sv;   // LM output (N, Kp, V), any normalisation
  const float *lpp;     int64_t lp_sn, lp_sk;          // log_probs_prev (N, Kp)
  const int64_t *y_prev; int64_t yp_ss, yp_sn, yp_sk;  // (S, N, Kp), clamped
  const int64_t *lens;  int64_t le_sn, le_sk;          // (N, Kp)
  int N, Kp, V, W, S;
  int has_eos, finish_all;
  int64_t eos, pad_clamped;
  int64_t *y_next;                   // (S + 1, N, W)
  int64_t *y_next_lens, *next_src;   // (N, W)
  float *lp_next;                    // (N, W)
  int32_t *active;                   // [1]: set to 1 when some batch element was NOT finished at the start of this iteration
  int32_t *pad_from;                 // (N,): first row of y that is padding (INT32_MAX: none yet)
  int waves_per_wg;
  // table form (pdt_beam_search_step_table): the scores of prefix (n, k) are row rows[n * Kp + k] of
  // `scores` (row stride sc_sk; sc_sn unused), its (maximum, log-sum-exp) row_stats[2 r], [2 r + 1]
  const int64_t *rows;
  const float *row_stats;
};

// maximum and log-sum-exp of a strided row (two passes, eight loads in flight)
__device__ __forceinline__ void row_log_softmax_stats(const float *x, const int64_t sx, const int V, float &mx_out,
                                                      float &lse_out) {
  const int lane = lane_id();
  float mx = -PDT_INF;
  int v = lane;
  for (; v + 7 * PDT_WAVE < V; v += 8 * PDT_WAVE) {
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = x[(int64_t)(v + i * PDT_WAVE) * sx];
#pragma unroll
    for (int i = 0; i < 8; ++i) mx = fmaxf(mx, t[i]);
  }
  for (; v < V; v += PDT_WAVE) mx = fmaxf(mx, x[(int64_t)v * sx]);
  mx = wave_max_f(mx);
  float s = 0.0f;
  v = lane;
  for (; v + 7 * PDT_WAVE < V; v += 8 * PDT_WAVE) {
    float t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = x[(int64_t)(v + i * PDT_WAVE) * sx];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += expf(t[i] - mx);
  }
  for (; v < V; v += PDT_WAVE) s += expf(x[(int64_t)v * sx] - mx);
  s = wave_sum_f(s);
  mx_out = mx;
  lse_out = logf(s);
}

__global__ void __launch_bounds__(512) beam_step_kernel(const BeamStepArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6), NW = a.waves_per_wg;
  const int64_t n = blockIdx.x;
  const int V = a.V, W = a.W, Kp = a.Kp, S = a.S;
  const int K = min(W, (int)min((int64_t)Kp * V, (int64_t)PDT_WAVE));  // :121
  const int M = min(V, K);
  u64 *surv = reinterpret_cast<u64 *>(smem) + (size_t)wave * PDT_SURV_CAP;  // one scratch per wave
  int *tl = reinterpret_cast<int *>(reinterpret_cast<u64 *>(smem) + (size_t)NW * PDT_SURV_CAP);
  float *tlm = reinterpret_cast<float *>(tl + Kp * PDT_WAVE);
  int *srcs = reinterpret_cast<int *>(tlm + Kp * PDT_WAVE);
  int *toks = srcs + W;
  int *plens = toks + W;
  int *ended = plens + W;   // [Kp]
  int *cnt = ended + Kp;    // [Kp] entries of prefix k's list

  // ---- which paths have ended (:413-420), is this element finished (:421-424) ------------------
  if ((int)threadIdx.x < Kp) {
    const int k = (int)threadIdx.x;
    const int64_t len = a.lens[n * a.le_sn + k * a.le_sk];
    bool e = false;
    if (a.has_eos && S > 0 && len > 0)
      e = a.y_prev[(len - 1) * a.yp_ss + n * a.yp_sn + k * a.yp_sk] == a.eos;
    ended[k] = e ? 1 : 0;
  }
  __syncthreads();
  bool done = false;
  if (a.has_eos && S > 0) {
    done = ended[0] != 0;
    if (a.finish_all)
      for (int k = 1; k < Kp; ++k) done = done && ended[k] != 0;
  }
  if (done) {
    // the beam it had, brought to `width` (:479-486; K' == width from the second iteration on),
    // one more row of padding
    if (threadIdx.x == 0 && a.pad_from[n] > S) a.pad_from[n] = S;
    for (int i = (int)threadIdx.x; i < W; i += NW * PDT_WAVE) {
      const bool has = i < Kp;
      a.lp_next[n * W + i] = has ? a.lpp[n * a.lp_sn + i * a.lp_sk] : -PDT_INF;
      a.y_next_lens[n * W + i] = has ? a.lens[n * a.le_sn + i * a.le_sk] : 0;
      a.next_src[n * W + i] = has ? i : 0;
    }
    for (int idx = (int)threadIdx.x; idx < (S + 1) * W; idx += NW * PDT_WAVE) {
      const int s = idx / W, i = idx - s * W;
      int64_t v = 0;
      if (s == S)
        v = a.pad_clamped;
      else if (i < Kp)
        v = a.y_prev[(int64_t)s * a.yp_ss + n * a.yp_sn + i * a.yp_sk];
      a.y_next[((int64_t)s * a.N + n) * W + i] = v;
    }
    return;
  }
  // (a flag, not a count: a thousand atomic increments of one word are ~15 us of memory-side
  // serialisation per launch -- round 5; the host only asks whether it is zero)
  if (threadIdx.x == 0 && *(volatile int32_t *)a.active == 0) *(volatile int32_t *)a.active = 1;

  // ---- per-prefix lists of the best tokens, ranked by the candidate value itself ------------------
  // log_probs_prev[k] + log_softmax(scores[k])[v] (:441, :122); an ended path offers eos alone, at
  // no cost (:448-458)
  for (int k = wave; k < Kp; k += NW) {
    const float bias = a.lpp[n * a.lp_sn + k * a.lp_sk];
    if (ended[k]) {
      if (lane == 0) {
        tl[k * PDT_WAVE] = (int)a.eos;
        tlm[k * PDT_WAVE] = (bias + 0.0f) + 0.0f;
        cnt[k] = 1;
      }
    } else {
      const int64_t r = a.rows ? a.rows[n * Kp + k] : 0;
      const float *row = a.rows ? a.scores + r * a.sc_sk : a.scores + n * a.sc_sn + k * a.sc_sk;
      float mx, lse;
      if (a.row_stats) {
        mx = a.row_stats[2 * r];
        lse = a.row_stats[2 * r + 1];
      } else {
        row_log_softmax_stats(row, a.sc_sv, V, mx, lse);
      }
      // rows of up to 1024 entries whose statistics are known are read ONCE, all loads in flight, and
      // ranked out of registers (the general form walks the row twice)
      const u64 tk = (a.row_stats && V <= 16 * PDT_WAVE)
                         ? wave_top_sorted_regs<16, true, true>(row, a.sc_sv, V, M, surv, bias, mx, lse)
                         : wave_top_sorted_strided<true, false, true, true>(row, a.sc_sv, V, M, surv, nullptr, nullptr, 1,
                                                                            bias, mx, lse);
      if (lane < M) {
        tl[k * PDT_WAVE + lane] = (int)idx_of(tk);
        tlm[k * PDT_WAVE + lane] = fkey_inv(key_of(tk));  // the value that was ranked
      }
      if (lane == 0) cnt[k] = M;
    }
    wave_sync();
  }
  __syncthreads();
  if (wave == 0) {
    const bool live = lane < Kp;
    const int *mytl = tl + (live ? lane : 0) * PDT_WAVE;
    const float *mytlm = tlm + (live ? lane : 0) * PDT_WAVE;
    const int mine = live ? cnt[lane] : 0;
    int ptr = 0;
    int new_src = 0, new_tok = 0;
    float new_lp = -PDT_INF;
    bool valid = false;
    for (int i = 0; i < K; ++i) {
      const bool has = live && ptr < mine;
      const int tok = has ? mytl[ptr] : 0;
      const float mass = has ? mytlm[ptr] : 0.0f;  // :122
      const unsigned key = has ? fkey(mass) : 0u;
      const unsigned mx = wave_max_u32(key);
      if (mx == 0u) break;
      const int win = (int)__builtin_ctzll(__ballot(key == mx));
      const int wtok = __builtin_amdgcn_readlane(tok, win);
      const float wmass = readlane_f(mass, win);
      if (lane == i) {
        new_src = win;
        new_tok = wtok;
        new_lp = wmass;
        valid = true;
      }
      if (lane == win) ++ptr;
    }
    if (lane < W) {
      const int plen = valid ? (int)a.lens[n * a.le_sn + new_src * a.le_sk] : -1;
      const int grew = valid ? 1 - ended[new_src] : 0;  // ended sources stay as long as they were (:465-468)
      a.lp_next[n * W + lane] = valid ? new_lp : -PDT_INF;  // :145-153 for the overflow
      a.next_src[n * W + lane] = valid ? new_src : 0;
      a.y_next_lens[n * W + lane] = valid ? plen + grew : 0;
      srcs[lane] = valid ? new_src : -1;
      toks[lane] = new_tok;
      plens[lane] = plen;
    }
  }
  __syncthreads();
  for (int idx = (int)threadIdx.x; idx < (S + 1) * W; idx += NW * PDT_WAVE) {
    const int s = idx / W, i = idx - s * W;
    const int src = srcs[i];
    const int pl = plens[i];
    int64_t v;
    if (src < 0)
      v = 0;
    else if (s == pl || s >= S)  // :130/:135 the appended token row, :137 the scatter
      v = toks[i];
    else
      v = a.y_prev[(int64_t)s * a.yp_ss + n * a.yp_sn + src * a.yp_sk];
    a.y_next[((int64_t)s * a.N + n) * W + i] = v;
  }
}


// -------------------------------------------------------------------------------------------
// The FLAT form (round 5; beam_advance.hip's beam_advance_flat_kernel has the argument), for the table
// route: the K winners are the K largest of ALL the candidates -- no sorted list per prefix.  The rows of
// the K' prefixes are separate rows of the model's table, so the 64-token chunks are numbered row by row,
// c = k * ceil(V / 64) + chunk, up to kStepFlatRegs per wave; an ended path offers eos alone (:448-458).
//   1. per-lane maxima -> one "column" maximum per lane over the eight waves; the K-th largest of the 64
//      is a lower bound tau of the K-th largest candidate;
//   2. every wave appends its candidates >= tau to one survivor list (LDS cursor);
//   3. wave 0 sorts them by (value, lowest k * V + v first): lane i holds winner i.
// tau = -inf (fewer than K columns with a finite maximum: most paths ended): the survivors are the finite
// candidates, -inf ones of paths that have not ended fill up in flat order.  A survivor list that overflows:
// every candidate again, from memory, through the chunked top-64 merge.  Same results as the list form.
constexpr int kStepFlatWaves = 8, kStepFlatRegs = 32;

__global__ void __launch_bounds__(64 * kStepFlatWaves, 6) beam_step_flat_kernel(const BeamStepArgs a) {
  __shared__ unsigned colmax[kStepFlatWaves * PDT_WAVE];
  __shared__ u64 surv[PDT_SURV_CAP];
  __shared__ unsigned ctl[4];  // [0] the threshold (float bits), [1] the survivor cursor
  __shared__ int srcs[PDT_WAVE], toks[PDT_WAVE], plens[PDT_WAVE];
  int lane = lane_id();
  asm volatile("" : "+v"(lane));
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (scalar: so is everything derived)
  const int64_t n = blockIdx.x;
  const int V = a.V, W = a.W, Kp = a.Kp, S = a.S;
  const int K = min(W, (int)min((int64_t)Kp * V, (int64_t)PDT_WAVE));  // :121
  if (threadIdx.x == 0) ctl[1] = 0u;

  // Two round trips to memory before the arithmetic, not five: what a chunk needs of its prefix k is
  // fetched by lane k of EVERY wave (the same few words: cache hits) -- length, table row, addend first;
  // then, together, the token that tells whether the path has ended (:413-420), the row's statistics and
  // the wave's chunks of the rows themselves (an ended path's are fetched for nothing).
  const int len_v = lane < Kp ? (int)a.lens[n * a.le_sn + lane * a.le_sk] : 0;
  const int64_t r_v = lane < Kp ? a.rows[n * Kp + lane] : 0;
  const float bias_v = lane < Kp ? a.lpp[n * a.lp_sn + lane * a.lp_sk] : 0.0f;
  const bool may_end = lane < Kp && a.has_eos && S > 0 && len_v > 0;
  const int64_t last_tok = may_end ? a.y_prev[(int64_t)(len_v - 1) * a.yp_ss + n * a.yp_sn + lane * a.yp_sk] : -1;
  const float mx_v = lane < Kp ? a.row_stats[2 * r_v] : 0.0f;
  const float lse_v = lane < Kp ? a.row_stats[2 * r_v + 1] : 0.0f;
  const unsigned r_lo = (unsigned)r_v, r_hi = (unsigned)((u64)r_v >> 32);
  const int eos = (int)a.eos;
  const int CH = (V + PDT_WAVE - 1) >> 6, total = Kp * CH;
  const int per = (total + kStepFlatWaves - 1) / kStepFlatWaves;  // chunks per wave, <= kStepFlatRegs (the launcher)
  const int c0 = wave * per;
  const int k0 = c0 / CH, ci0 = c0 - k0 * CH;
  float x[kStepFlatRegs];
  {
    int k = k0, ci = ci0;
#pragma unroll
    for (int j = 0; j < kStepFlatRegs; ++j) {
      x[j] = -PDT_INF;  // (beyond a row, beyond the candidates: never a survivor, see the threshold below)
      if (j < per && c0 + j < total) {
        const u64 r = ((u64)(unsigned)__builtin_amdgcn_readlane((int)r_hi, k) << 32) |
                      (unsigned)__builtin_amdgcn_readlane((int)r_lo, k);
        const float *row = a.scores + (int64_t)r * a.sc_sk;
        const int v = ci * PDT_WAVE + lane;
        if (ci + 1 < CH || v < V) x[j] = row[v];
        if (++ci == CH) ci = 0, ++k;
      }
    }
  }
  const u64 ended_mask = __ballot(may_end && last_tok == a.eos);

  // ---- is this element finished (:421-424) -----------------------------------------------------
  bool done = false;
  if (a.has_eos && S > 0) {
    const u64 all = Kp >= 64 ? ~0ull : ((1ull << Kp) - 1ull);
    done = a.finish_all ? (ended_mask & all) == all : (ended_mask & 1ull) != 0;
  }
  if (done) {
    // the beam it had, brought to `width` (:479-486; K' == width from the second iteration on),
    // one more row of padding
    if (threadIdx.x == 0 && a.pad_from[n] > S) a.pad_from[n] = S;
    for (int i = (int)threadIdx.x; i < W; i += kStepFlatWaves * PDT_WAVE) {
      const bool has = i < Kp;
      a.lp_next[n * W + i] = has ? a.lpp[n * a.lp_sn + i * a.lp_sk] : -PDT_INF;
      a.y_next_lens[n * W + i] = has ? a.lens[n * a.le_sn + i * a.le_sk] : 0;
      a.next_src[n * W + i] = has ? i : 0;
    }
    for (int idx = (int)threadIdx.x; idx < (S + 1) * W; idx += kStepFlatWaves * PDT_WAVE) {
      const int s = idx / W, i = idx - s * W;
      int64_t v = 0;
      if (s == S)
        v = a.pad_clamped;
      else if (i < Kp)
        v = a.y_prev[(int64_t)s * a.yp_ss + n * a.yp_sn + i * a.yp_sk];
      a.y_next[((int64_t)s * a.N + n) * W + i] = v;
    }
    return;
  }
  // (a flag, not a count: the host only asks whether it is zero, and a thousand atomic increments of one
  // word serialise)
  if (threadIdx.x == 0 && *(volatile int32_t *)a.active == 0) *(volatile int32_t *)a.active = 1;

  // ---- 1. the candidates: log_probs_prev[k] + log_softmax(scores[k])[v] (:441, :122) -----------------
  // (every load in flight before the first is used: the values are formed in a second loop)
  float lmax = -PDT_INF;
  {
    int k = k0, ci = ci0;
#pragma unroll
    for (int j = 0; j < kStepFlatRegs; ++j) {
      if (j < per && c0 + j < total) {
        const float bias = readlane_f(bias_v, k);
        const int v = ci * PDT_WAVE + lane;
        if ((ended_mask >> k) & 1ull) {
          x[j] = v == eos ? (bias + 0.0f) + 0.0f : -PDT_INF;  // eos alone, at no cost (:448-458)
        } else {
          const float val = (bias + ((x[j] - readlane_f(mx_v, k)) - readlane_f(lse_v, k))) + 0.0f;
          x[j] = (ci + 1 < CH || v < V) ? val : -PDT_INF;
        }
        lmax = fmaxf(lmax, x[j]);
        if (++ci == CH) ci = 0, ++k;
      }
    }
  }
  colmax[wave * PDT_WAVE + lane] = fkey(lmax);
  __syncthreads();
  if (wave == 0) {
    unsigned cm = colmax[lane];
#pragma unroll
    for (int w = 1; w < kStepFlatWaves; ++w) cm = max(cm, colmax[w * PDT_WAVE + lane]);
    const unsigned sorted_max = wave_sort_desc<unsigned>(cm);
    // x >= the lowest finite float <=> x > -inf: the padding and the -inf candidates stay out
    const float tau = fmaxf(fkey_inv((unsigned)__builtin_amdgcn_readlane((int)sorted_max, K - 1)), -3.4028234664e38f);
    if (lane == 0) ctl[0] = __float_as_uint(tau);
  }
  __syncthreads();
  // ---- 2. survivors ------------------------------------------------------------------------
  const float tau = __uint_as_float(ctl[0]);
  if (__ballot(lmax >= tau)) {
#pragma unroll
    for (int j = 0; j < kStepFlatRegs; ++j) {
      const bool pred = x[j] >= tau;
      if (__ballot(pred)) {
        const int c = c0 + j, k = c / CH, ci = c - k * CH;
        if (pred) {
          const unsigned at = atomicAdd(&ctl[1], 1u);
          if (at < PDT_SURV_CAP) surv[at] = pack_key(fkey(x[j]), ((unsigned)k << 20) | (unsigned)(ci * PDT_WAVE + lane));
        }
      }
    }
  }
  __syncthreads();
  // ---- 3. the winners in order ---------------------------------------------------------------
  if (wave == 0) {
    const int count = (int)ctl[1];
    auto value_at = [&](const int k, const int v) {  // candidate (k, v) of a path that has not ended, from memory
      const int64_t r = a.rows[n * Kp + k];
      return (a.lpp[n * a.lp_sn + k * a.lp_sk] + ((a.scores[r * a.sc_sk + v] - a.row_stats[2 * r]) - a.row_stats[2 * r + 1])) + 0.0f;
    };
    u64 tk = 0ull;
    int have = 0;  // winners that are in tk
    if (count <= PDT_SURV_CAP) {
      tk = wave_sort_desc<u64>(lane < count ? surv[lane] : 0ull);
      have = min(count, K);
    } else {
      // every candidate, -inf ones included, from memory
      int cands = 0;
      for (int k = 0; k < Kp; ++k) {
        if ((ended_mask >> k) & 1ull) {
          const float xv = (a.lpp[n * a.lp_sn + k * a.lp_sk] + 0.0f) + 0.0f;
          tk = wave_merge_top64(tk, lane == 0 ? pack_key(fkey(xv), ((unsigned)k << 20) | (unsigned)eos) : 0ull);
          cands += 1;
        } else {
          for (int v0 = 0; v0 < V; v0 += PDT_WAVE) {
            const int v = v0 + lane;
            const float xv = v < V ? value_at(k, v) : 0.0f;
            tk = wave_merge_top64(tk, v < V ? pack_key(fkey(xv), ((unsigned)k << 20) | (unsigned)v) : 0ull);
          }
          cands += V;
        }
      }
      have = min(cands, K);
    }
    int new_src = (int)(idx_of(tk) >> 20), new_tok = (int)(idx_of(tk) & 0xfffffu);
    float new_lp = fkey_inv(key_of(tk));
    if (have < K && count <= PDT_SURV_CAP) {
      // tau = -inf and fewer than K finite candidates: -inf candidates of the paths that have not ended, in
      // flat order, behind them
      const int filled = have;
      for (int k = 0; k < Kp && have < K; ++k) {
        if ((ended_mask >> k) & 1ull) continue;
        for (int v0 = 0; v0 < V && have < K; v0 += PDT_WAVE) {
          const int v = v0 + lane;
          const bool is = v < V && value_at(k, v) == -PDT_INF;
          const u64 b = __ballot(is);
          const int rank = have + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
          if (is && rank < K) {  // winner number `rank`: handed to lane `rank`
            srcs[rank] = k;
            toks[rank] = v;
          }
          have += __popcll(b);
        }
      }
      have = min(have, K);
      wave_sync();
      if (lane >= filled && lane < have) {
        new_src = srcs[lane];
        new_tok = toks[lane];
        new_lp = -PDT_INF;
      }
      wave_sync();
    }
    const int src_len = shfl_i(len_v, new_src);  // (the lengths are in lane k since the start; every lane takes part)
    if (lane < W) {
      const bool valid = lane < have;
      const int plen = valid ? src_len : -1;
      const int grew = valid ? 1 - (int)((ended_mask >> (new_src & 63)) & 1ull) : 0;  // ended sources stay as long as they were (:465-468)
      a.lp_next[n * W + lane] = valid ? new_lp : -PDT_INF;  // :145-153 for the overflow
      a.next_src[n * W + lane] = valid ? new_src : 0;
      a.y_next_lens[n * W + lane] = valid ? plen + grew : 0;
      srcs[lane] = valid ? new_src : -1;
      toks[lane] = new_tok;
      plens[lane] = plen;
    }
  }
  __syncthreads();
  for (int idx = (int)threadIdx.x; idx < (S + 1) * W; idx += kStepFlatWaves * PDT_WAVE) {
    const int s = idx / W, i = idx - s * W;
    const int src = srcs[i];
    const int pl = plens[i];
    int64_t v;
    if (src < 0)
      v = 0;
    else if (s == pl || s >= S)  // :130/:135 the appended token row, :137 the scatter
      v = toks[i];
    else
      v = a.y_prev[(int64_t)s * a.yp_ss + n * a.yp_sn + src * a.yp_sk];
    a.y_next[((int64_t)s * a.N + n) * W + i] = v;
  }
}


// -------------------------------------------------------------------------------------------
// The whole search over a bigram table in ONE launch (round 5, pdt_beam_search_table): the batch elements
// are independent and the model is a table, so a workgroup runs its element's iterations back to back --
// the beam (log-probability, length, last token = next table row) in LDS, the selection of
// beam_step_flat_kernel per iteration, NO history copies: an iteration leaves one (source, token) word per
// beam entry in a trie (N, T, W), and beam_search_walk_kernel reads the final paths off it at the end
// (row s of a path is the token its ancestor chose in iteration s -- exactly what the copies of the
// step-by-step form add up to, rows beyond a path's length included).  An element stops at the iteration
// that finds it finished (:421-424) and records it; the host cuts y at the largest such iteration.
#ifndef PDT_BS_WAVES  // waves per SIMD the ROWS16 form is compiled for (8: 64 registers, 13 of them spilled; 6: 80)
#define PDT_BS_WAVES 8
#endif
struct BeamSearchArgs {
  const float *table;  int64_t tb_sr;   // (U, V), token stride 1
  const float *row_stats;               // (U, 2)
  int N, V, W, n_iters, sos_row;
  int has_eos, finish_all, eos;
  unsigned *trie;       // (N, n_iters, W): source << 20 | token
  float *lp_out;        // (N, W)
  int64_t *lens_out;    // (N, W)
  int32_t *finish;      // (N,): the iteration that found the element finished (n_iters: none did)
  int32_t *t_stop;      // [1]: max over the elements of finish (atomicMax; the caller zeroes it)
};

// ROWS16: width <= 16 and V <= 1024 -- a wave owns whole rows (prefix k = wave, wave + 8), sixteen chunk slots
// each, so every register, offset and guard of the candidate loops is static and a row's pointer, addend and
// statistics are fetched once per row instead of once per chunk (10 000 -> ~2 000 instructions per
// element and iteration).  Other shapes: chunks numbered across the rows as in beam_step_flat_kernel.
template <bool ROWS16>
__global__ void __launch_bounds__(64 * kStepFlatWaves, ROWS16 ? PDT_BS_WAVES : 6) beam_search_table_kernel(const BeamSearchArgs a) {
  __shared__ unsigned colmax[kStepFlatWaves * PDT_WAVE];
  __shared__ u64 surv[PDT_SURV_CAP];
  __shared__ unsigned ctl[4];  // [0] the threshold (float bits), [1] the survivor cursor
  __shared__ int srcs[PDT_WAVE], toks[PDT_WAVE];
  __shared__ int st_row[PDT_WAVE], st_len[PDT_WAVE], st_tok[PDT_WAVE];  // the beam between iterations
  __shared__ float st_lp[PDT_WAVE];
  int lane = lane_id();
  asm volatile("" : "+v"(lane));
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t n = blockIdx.x;
  const int W = a.W, eos = a.eos;
  const int CH = (a.V + PDT_WAVE - 1) >> 6;
  if (threadIdx.x < PDT_WAVE) {  // one empty path
    st_row[threadIdx.x] = a.sos_row;
    st_len[threadIdx.x] = 0;
    st_tok[threadIdx.x] = -1;
    st_lp[threadIdx.x] = threadIdx.x == 0 ? 0.0f : -PDT_INF;
  }
  __syncthreads();
  int Kp = 1, t = 0;
  const int lane0 = lane;
  for (; t < a.n_iters; ++t) {
    // (laundered per iteration: nothing derived from the lane index or V is loop-invariant to the compiler --
    // hoisted, the sixteen clamped offsets and guards of the candidate loops are live across the whole loop
    // and spill: 149 spilled registers before this line)
    int lane = lane0, V = a.V;
    asm volatile("" : "+v"(lane), "+s"(V));
    if (threadIdx.x == 0) ctl[1] = 0u;
    const int K = min(W, (int)min((int64_t)Kp * V, (int64_t)PDT_WAVE));  // :121
    // lane k of every wave: prefix k
    const int len_v = lane < Kp ? st_len[lane] : 0;
    const int r_v = lane < Kp ? st_row[lane] : 0;
    const float bias_v = lane < Kp ? st_lp[lane] : 0.0f;
    const u64 ended_mask = __ballot(lane < Kp && a.has_eos && t > 0 && len_v > 0 && st_tok[lane] == eos);  // :413-420
    if (a.has_eos && t > 0) {  // :421-424
      const u64 all = Kp >= 64 ? ~0ull : ((1ull << Kp) - 1ull);
      if (a.finish_all ? (ended_mask & all) == all : (ended_mask & 1ull) != 0) break;
    }
    const float mx_v = lane < Kp ? a.row_stats[2 * (int64_t)r_v] : 0.0f;
    const float lse_v = lane < Kp ? a.row_stats[2 * (int64_t)r_v + 1] : 0.0f;
    // ---- 1. the candidates: log_probs_prev[k] + log_softmax(scores[k])[v] (:441, :122) ---------------
    const int total = Kp * CH;
    const int per = (total + kStepFlatWaves - 1) / kStepFlatWaves;  // chunks per wave, <= kStepFlatRegs (the launcher)
    const int c0 = wave * per;
    const int k0 = c0 / CH, ci0 = c0 - k0 * CH;
    float x[kStepFlatRegs];
    float lmax = -PDT_INF;
    if constexpr (ROWS16) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int k = wave + rr * kStepFlatWaves;
#pragma unroll
        for (int ci = 0; ci < 16; ++ci) x[rr * 16 + ci] = -PDT_INF;
        if (k < Kp && !((ended_mask >> k) & 1ull)) {
          const float *row = a.table + (int64_t)__builtin_amdgcn_readlane(r_v, k) * a.tb_sr + lane;
          // (straight-line: slots beyond the row re-read its last token and are masked below)
#pragma unroll
          for (int ci = 0; ci < 16; ++ci) x[rr * 16 + ci] = row[min(ci * PDT_WAVE, V - 1 - lane)];
        }
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int k = wave + rr * kStepFlatWaves;
        if (k < Kp) {
          const float bias = readlane_f(bias_v, k);
          if ((ended_mask >> k) & 1ull) {  // eos alone, at no cost (:448-458)
#pragma unroll
            for (int ci = 0; ci < 16; ++ci) x[rr * 16 + ci] = ci * PDT_WAVE + lane == eos ? (bias + 0.0f) + 0.0f : -PDT_INF;
          } else {
            const float mx = readlane_f(mx_v, k), lse = readlane_f(lse_v, k);
#pragma unroll
            for (int ci = 0; ci < 16; ++ci) {
              const float val = (bias + ((x[rr * 16 + ci] - mx) - lse)) + 0.0f;
              x[rr * 16 + ci] = ci * PDT_WAVE + lane < V ? val : -PDT_INF;
            }
          }
#pragma unroll
          for (int ci = 0; ci < 16; ci += 2) lmax = fmaxf(lmax, fmaxf(x[rr * 16 + ci], x[rr * 16 + ci + 1]));
        }
      }
    } else {
      int k = k0, ci = ci0;
#pragma unroll
      for (int j = 0; j < kStepFlatRegs; ++j) {
        x[j] = -PDT_INF;  // (beyond a row, beyond the candidates: never a survivor, see the threshold below)
        if (j < per && c0 + j < total) {
          const float *row = a.table + (int64_t)__builtin_amdgcn_readlane(r_v, k) * a.tb_sr;
          const int v = ci * PDT_WAVE + lane;
          if (ci + 1 < CH || v < V) x[j] = row[v];
          if (++ci == CH) ci = 0, ++k;
        }
      }
    }
    if constexpr (!ROWS16) {
      int k = k0, ci = ci0;
#pragma unroll
      for (int j = 0; j < kStepFlatRegs; ++j) {
        if (j < per && c0 + j < total) {
          const float bias = readlane_f(bias_v, k);
          const int v = ci * PDT_WAVE + lane;
          if ((ended_mask >> k) & 1ull) {
            x[j] = v == eos ? (bias + 0.0f) + 0.0f : -PDT_INF;  // eos alone, at no cost (:448-458)
          } else {
            const float val = (bias + ((x[j] - readlane_f(mx_v, k)) - readlane_f(lse_v, k))) + 0.0f;
            x[j] = (ci + 1 < CH || v < V) ? val : -PDT_INF;
          }
          lmax = fmaxf(lmax, x[j]);
          if (++ci == CH) ci = 0, ++k;
        }
      }
    }
    colmax[wave * PDT_WAVE + lane] = fkey(lmax);
    __syncthreads();
    if (wave == 0) {
      unsigned cm = colmax[lane];
#pragma unroll
      for (int w = 1; w < kStepFlatWaves; ++w) cm = max(cm, colmax[w * PDT_WAVE + lane]);
      const unsigned sorted_max = wave_sort_desc<unsigned>(cm);
      const float tau = fmaxf(fkey_inv((unsigned)__builtin_amdgcn_readlane((int)sorted_max, K - 1)), -3.4028234664e38f);
      if (lane == 0) ctl[0] = __float_as_uint(tau);
    }
    __syncthreads();
    // ---- 2. survivors ----------------------------------------------------------------------
    const float tau = __uint_as_float(ctl[0]);
    if (__ballot(lmax >= tau)) {
#pragma unroll
      for (int j = 0; j < kStepFlatRegs; ++j) {
        const bool pred = x[j] >= tau;
        if (__ballot(pred)) {
          int k, ci;
          if constexpr (ROWS16) {
            k = wave + (j >> 4) * kStepFlatWaves, ci = j & 15;
          } else {
            const int c = c0 + j;
            k = c / CH, ci = c - k * CH;
          }
          if (pred) {
            const unsigned at = atomicAdd(&ctl[1], 1u);
            if (at < PDT_SURV_CAP) surv[at] = pack_key(fkey(x[j]), ((unsigned)k << 20) | (unsigned)(ci * PDT_WAVE + lane));
          }
        }
      }
    }
    __syncthreads();
    // ---- 3. the winners in order, the new beam ----------------------------------------------------
    if (wave == 0) {
      const int count = (int)ctl[1];
      auto value_at = [&](const int k, const int v) {  // candidate (k, v) of a path that has not ended, from memory
        const int64_t r = st_row[k];
        return (st_lp[k] + ((a.table[r * a.tb_sr + v] - a.row_stats[2 * r]) - a.row_stats[2 * r + 1])) + 0.0f;
      };
      u64 tk = 0ull;
      int have = 0;  // winners that are in tk
      if (count <= PDT_SURV_CAP) {
        tk = wave_sort_desc<u64>(lane < count ? surv[lane] : 0ull);
        have = min(count, K);
      } else {
        int cands = 0;  // every candidate, -inf ones included, from memory
        for (int k = 0; k < Kp; ++k) {
          if ((ended_mask >> k) & 1ull) {
            const float xv = (st_lp[k] + 0.0f) + 0.0f;
            tk = wave_merge_top64(tk, lane == 0 ? pack_key(fkey(xv), ((unsigned)k << 20) | (unsigned)eos) : 0ull);
            cands += 1;
          } else {
            for (int v0 = 0; v0 < V; v0 += PDT_WAVE) {
              const int v = v0 + lane;
              const float xv = v < V ? value_at(k, v) : 0.0f;
              tk = wave_merge_top64(tk, v < V ? pack_key(fkey(xv), ((unsigned)k << 20) | (unsigned)v) : 0ull);
            }
            cands += V;
          }
        }
        have = min(cands, K);
      }
      int new_src = (int)(idx_of(tk) >> 20), new_tok = (int)(idx_of(tk) & 0xfffffu);
      float new_lp = fkey_inv(key_of(tk));
      if (have < K && count <= PDT_SURV_CAP) {
        // fewer than K finite candidates: -inf ones of the paths that have not ended, in flat order, behind them
        const int filled = have;
        for (int k = 0; k < Kp && have < K; ++k) {
          if ((ended_mask >> k) & 1ull) continue;
          for (int v0 = 0; v0 < V && have < K; v0 += PDT_WAVE) {
            const int v = v0 + lane;
            const bool is = v < V && value_at(k, v) == -PDT_INF;
            const u64 b = __ballot(is);
            const int rank = have + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
            if (is && rank < K) {
              srcs[rank] = k;
              toks[rank] = v;
            }
            have += __popcll(b);
          }
        }
        have = min(have, K);
        wave_sync();
        if (lane >= filled && lane < have) {
          new_src = srcs[lane];
          new_tok = toks[lane];
          new_lp = -PDT_INF;
        }
        wave_sync();
      }
      const bool valid = lane < have;
      const int src = valid ? new_src : 0;
      const int src_len = shfl_i(len_v, src);
      const int grew = 1 - (int)((ended_mask >> (src & 63)) & 1ull);  // ended sources stay as long as they were (:465-468)
      wave_sync();  // (every lane has read the old beam)
      if (lane < W) {
        st_lp[lane] = valid ? new_lp : -PDT_INF;  // :145-153 for the overflow
        st_len[lane] = valid ? src_len + grew : 0;
        st_tok[lane] = valid ? new_tok : 0;
        st_row[lane] = valid ? new_tok : 0;  // the next row of the table: the path's last token
        a.trie[(n * a.n_iters + t) * W + lane] = valid ? ((unsigned)src << 20) | (unsigned)new_tok : 0u;
      }
    }
    Kp = W;
    __syncthreads();
  }
  // t: the iteration that found the element finished, or n_iters
  if (threadIdx.x < (unsigned)W) {
    const bool has = (int)threadIdx.x < Kp;
    a.lp_out[n * W + threadIdx.x] = has ? st_lp[threadIdx.x] : -PDT_INF;
    a.lens_out[n * W + threadIdx.x] = has ? st_len[threadIdx.x] : 0;
  }
  if (threadIdx.x == 0) {
    a.finish[n] = t;
    atomicMax(a.t_stop, t);
  }
}

// y[s, n, i] for s < T: the token the ancestor of final path i chose in iteration s; pad_value from the
// element's finishing iteration on (:479-486).  One workgroup per element; the trie slab goes through LDS
// in tiles from the end, the paths' ancestors carried in registers across tiles.
constexpr int kWalkTile = 256;
__global__ void __launch_bounds__(256) beam_search_walk_kernel(const unsigned *trie, const int32_t *finish, const int N,
                                                               const int n_iters, const int W, const int T,
                                                               const int64_t pad_value, int64_t *y) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned *tile = reinterpret_cast<unsigned *>(smem);  // [kWalkTile x W] trie words, then tokens in place
  const int64_t n = blockIdx.x;
  const int tid = (int)threadIdx.x;
  const int fin = min(finish[n], T);  // rows [0, fin) are paths, [fin, T) padding
  for (int idx = fin * W + tid; idx < T * W; idx += 256) {
    const int s = idx / W, i = idx - s * W;
    y[((int64_t)s * N + n) * W + i] = pad_value;
  }
  int anc = tid < W ? tid : 0;  // the ancestor of final path `tid` after the iterations walked so far
  for (int hi = fin; hi > 0; hi -= kWalkTile) {
    const int lo = max(0, hi - kWalkTile), rows = hi - lo;
    for (int idx = tid; idx < rows * W; idx += 256) tile[idx] = trie[(n * n_iters + lo) * W + idx];
    __syncthreads();
    if (tid < W) {
      // (the words of a row are read before they are overwritten with tokens: one lane per column, rows
      // from the end -- a column's word may be another column's ancestor, so tokens go to the second half)
      for (int s = rows - 1; s >= 0; --s) {
        const unsigned w = tile[s * W + anc];
        tile[kWalkTile * W + s * W + tid] = w & 0xfffffu;
        anc = (int)(w >> 20);
      }
    }
    __syncthreads();
    for (int idx = tid; idx < rows * W; idx += 256) {
      const int s = idx / W, i = idx - s * W;
      y[((int64_t)(lo + s) * N + n) * W + i] = (int64_t)tile[kWalkTile * W + idx];
    }
    __syncthreads();
  }
}

}  // namespace pdt

namespace pdt {

// (maximum, log-sum-exp) of every row of a (U, V) table: what beam_step_kernel computes per prefix and
// iteration, once per table (the same routine: the same bits)
__global__ void __launch_bounds__(256) row_stats_kernel(const float *table, const int64_t tb_sr, const int64_t tb_sv,
                                                        const int U, const int V, float *stats) {
  const int r = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (r >= U) return;
  float mx, lse;
  row_log_softmax_stats(table + (int64_t)r * tb_sr, tb_sv, V, mx, lse);
  if (lane_id() == 0) {
    stats[2 * r] = mx;
    stats[2 * r + 1] = lse;
  }
}

static int beam_step_entry(const float *scores, int64_t sc_sn, int64_t sc_sk, int64_t sc_sv, const int64_t *rows,
                           const float *row_stats, int64_t N, int64_t Kp, int64_t V, int64_t width,
                           const float *log_probs_prev, int64_t lp_sn, int64_t lp_sk, const int64_t *y_prev, int64_t S,
                           int64_t yp_ss, int64_t yp_sn, int64_t yp_sk, const int64_t *y_prev_lens, int64_t le_sn,
                           int64_t le_sk, int has_eos, int64_t eos, int finish_all_paths, int64_t pad_value,
                           int64_t *y_next, int64_t *y_next_lens, float *log_probs_next, int64_t *next_src,
                           int32_t *active, int32_t *pad_from, void *stream) {
  if (N < 0 || Kp < 1 || V < 1 || width < 1 || S < 0) return PDT_E_ARG;
  if (N == 0) return PDT_OK;
  if (!scores || !log_probs_prev || (S > 0 && !y_prev) || !y_prev_lens || !y_next || !y_next_lens ||
      !log_probs_next || !next_src || !active || !pad_from)
    return PDT_E_ARG;
  if (has_eos && (eos < 0 || eos >= V)) return PDT_E_ARG;
  if (V >= (1 << 30) || S >= (1 << 26) || N >= (1ll << 31)) return PDT_E_TOO_LONG;
  if (width > PDT_WAVE || Kp > PDT_WAVE) return PDT_E_TOO_LONG;  // (wider beams: the step-by-step form)
  BeamStepArgs a{};
  a.scores = scores; a.sc_sn = sc_sn; a.sc_sk = sc_sk; a.sc_sv = sc_sv;
  a.rows = rows; a.row_stats = row_stats;
  a.lpp = log_probs_prev; a.lp_sn = lp_sn; a.lp_sk = lp_sk;
  a.y_prev = y_prev; a.yp_ss = yp_ss; a.yp_sn = yp_sn; a.yp_sk = yp_sk;
  a.lens = y_prev_lens; a.le_sn = le_sn; a.le_sk = le_sk;
  a.N = (int)N; a.Kp = (int)Kp; a.V = (int)V; a.W = (int)width; a.S = (int)S;
  a.has_eos = has_eos; a.finish_all = finish_all_paths; a.eos = eos;
  a.pad_clamped = pad_value < 0 ? 0 : (pad_value > V - 1 ? V - 1 : pad_value);
  a.y_next = y_next; a.y_next_lens = y_next_lens; a.lp_next = log_probs_next; a.next_src = next_src;
  a.active = active; a.pad_from = pad_from;
  if (switches().step_flat != 0 && rows && row_stats && sc_sv == 1 && V > PDT_WAVE && V < (1 << 20) &&
      Kp * ((V + PDT_WAVE - 1) / PDT_WAVE) <= kStepFlatWaves * kStepFlatRegs) {
    hipLaunchKernelGGL(beam_step_flat_kernel, dim3((unsigned)a.N), dim3(64 * kStepFlatWaves), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  int nw = 1;
  while (nw < 8 && nw * 2 <= a.Kp) nw *= 2;  // waves per element: a power of two <= min(Kp, 8)
  a.waves_per_wg = nw;
  const size_t smem = ((size_t)nw * PDT_SURV_CAP * 8 + (size_t)a.Kp * PDT_WAVE * 8 + (size_t)a.W * 12 +
                       (size_t)a.Kp * 8 + 15) & ~(size_t)15;
  hipLaunchKernelGGL(beam_step_kernel, dim3((unsigned)a.N), dim3(64 * nw), smem, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // namespace pdt

extern "C" int pdt_beam_search_step(const float *scores, int64_t sc_sn, int64_t sc_sk, int64_t sc_sv, int64_t N,
                                    int64_t Kp, int64_t V, int64_t width, const float *log_probs_prev,
                                    int64_t lp_sn, int64_t lp_sk, const int64_t *y_prev, int64_t S,
                                    int64_t yp_ss, int64_t yp_sn, int64_t yp_sk, const int64_t *y_prev_lens,
                                    int64_t le_sn, int64_t le_sk, int has_eos, int64_t eos, int finish_all_paths,
                                    int64_t pad_value, int64_t *y_next, int64_t *y_next_lens,
                                    float *log_probs_next, int64_t *next_src, int32_t *active,
                                    int32_t *pad_from, void *stream) {
  return pdt::beam_step_entry(scores, sc_sn, sc_sk, sc_sv, nullptr, nullptr, N, Kp, V, width, log_probs_prev, lp_sn,
                              lp_sk, y_prev, S, yp_ss, yp_sn, yp_sk, y_prev_lens, le_sn, le_sk, has_eos, eos,
                              finish_all_paths, pad_value, y_next, y_next_lens, log_probs_next, next_src, active,
                              pad_from, stream);
}

extern "C" int pdt_beam_search_step_table(const float *table, int64_t tb_sr, int64_t tb_sv, int64_t U,
                                          const float *row_stats, const int64_t *rows, int64_t N, int64_t Kp,
                                          int64_t V, int64_t width, const float *log_probs_prev, int64_t lp_sn,
                                          int64_t lp_sk, const int64_t *y_prev, int64_t S, int64_t yp_ss,
                                          int64_t yp_sn, int64_t yp_sk, const int64_t *y_prev_lens, int64_t le_sn,
                                          int64_t le_sk, int has_eos, int64_t eos, int finish_all_paths,
                                          int64_t pad_value, int64_t *y_next, int64_t *y_next_lens,
                                          float *log_probs_next, int64_t *next_src, int32_t *active,
                                          int32_t *pad_from, void *stream) {
  if (!rows || U < 1) return PDT_E_ARG;
  return pdt::beam_step_entry(table, 0, tb_sr, tb_sv, rows, row_stats, N, Kp, V, width, log_probs_prev, lp_sn, lp_sk,
                              y_prev, S, yp_ss, yp_sn, yp_sk, y_prev_lens, le_sn, le_sk, has_eos, eos,
                              finish_all_paths, pad_value, y_next, y_next_lens, log_probs_next, next_src, active,
                              pad_from, stream);
}

extern "C" int pdt_row_log_softmax_stats(const float *table, int64_t tb_sr, int64_t tb_sv, int64_t U, int64_t V,
                                         float *stats, void *stream) {
  if (U < 0 || V < 1 || U >= (1ll << 31) || V >= (1 << 30)) return PDT_E_ARG;
  if (U == 0) return PDT_OK;
  if (!table || !stats) return PDT_E_ARG;
  hipLaunchKernelGGL(pdt::row_stats_kernel, dim3((unsigned)((U + 3) / 4)), dim3(256), 0, (hipStream_t)stream, table,
                     tb_sr, tb_sv, (int)U, (int)V, stats);
  return (int)hipGetLastError();
}

extern "C" int pdt_beam_search_table(const float *table, int64_t tb_sr, int64_t tb_sv, int64_t U, const float *row_stats,
                                     int64_t sos_row, int64_t N, int64_t V, int64_t width, int64_t n_iters, int has_eos,
                                     int64_t eos, int finish_all_paths, uint32_t *trie, float *log_probs_out,
                                     int64_t *lens_out, int32_t *finish, int32_t *t_stop, void *stream) {
  using namespace pdt;
  if (N < 0 || V < 1 || width < 1 || n_iters < 1 || U < 1 || sos_row < 0 || sos_row >= U) return PDT_E_ARG;
  if (N == 0) return PDT_OK;
  if (!table || !row_stats || !trie || !log_probs_out || !lens_out || !finish || !t_stop) return PDT_E_ARG;
  if (has_eos && (eos < 0 || eos >= V)) return PDT_E_ARG;
  if (N >= (1ll << 31) || n_iters >= (1 << 24)) return PDT_E_TOO_LONG;
  if (tb_sv != 1 || V <= PDT_WAVE || V >= (1 << 20) || V > U || width > PDT_WAVE ||
      width * ((V + PDT_WAVE - 1) / PDT_WAVE) > kStepFlatWaves * kStepFlatRegs)
    return PDT_E_UNSUPPORTED;  // (the per-iteration entry points serve those)
  BeamSearchArgs a{};
  a.table = table; a.tb_sr = tb_sr; a.row_stats = row_stats;
  a.N = (int)N; a.V = (int)V; a.W = (int)width; a.n_iters = (int)n_iters; a.sos_row = (int)sos_row;
  a.has_eos = has_eos; a.finish_all = finish_all_paths; a.eos = (int)eos;
  a.trie = trie; a.lp_out = log_probs_out; a.lens_out = lens_out; a.finish = finish; a.t_stop = t_stop;
  if (width <= 16 && V <= 16 * PDT_WAVE)
    hipLaunchKernelGGL(beam_search_table_kernel<true>, dim3((unsigned)N), dim3(64 * kStepFlatWaves), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(beam_search_table_kernel<false>, dim3((unsigned)N), dim3(64 * kStepFlatWaves), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int pdt_beam_search_table_paths(const uint32_t *trie, const int32_t *finish, int64_t N, int64_t n_iters,
                                           int64_t width, int64_t T, int64_t pad_value, int64_t *y, void *stream) {
  using namespace pdt;
  if (N < 0 || n_iters < 1 || width < 1 || width > PDT_WAVE || T < 0 || T > n_iters) return PDT_E_ARG;
  if (N == 0 || T == 0) return PDT_OK;
  if (!trie || !finish || !y) return PDT_E_ARG;
  const size_t smem = (size_t)2 * kWalkTile * width * 4;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(beam_search_walk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(beam_search_walk_kernel, dim3((unsigned)N), dim3(256), smem, (hipStream_t)stream, trie, finish, (int)N,
                     (int)n_iters, (int)width, (int)T, pad_value, y);
  return (int)hipGetLastError();
}
