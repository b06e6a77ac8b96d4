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
  int32_t *active;                   // [1]: += batch elements NOT finished at the start of this iteration
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
  if (threadIdx.x == 0) atomicAdd(a.active, 1);

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
