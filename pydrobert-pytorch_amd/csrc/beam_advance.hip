// Step functions of the two beam searches for gfx950, one workgroup per batch element:
//   * pdt_ctc_prefix_search_advance -- ctc_prefix_search_advance (reference
//     _decoding.py:636-934) with arbitrary per-prefix extension probabilities (language-model
//     fusion) and dense (S, N, K') histories, as the reference's signature requires;
//   * pdt_beam_search_advance       -- beam_search_advance (_decoding.py:41-155).
// Both select the top-K of K'*V (+K') candidates WITHOUT materialising them: each old prefix
// gets the sorted list of its best tokens (wave_top_sorted; the waves of the workgroup share the
// prefixes), then K rounds of a wave-wide max-reduce over the list heads pick the winners in order.  Ties go to the lowest flat
// candidate index (the reference's torch.topk leaves them unspecified).
#include "advance_args.hpp"
#include "ctc_frame.hpp"
#include "row_reduce.hpp"
#include "switches.hpp"

#ifndef PDT_FUSED_STEP_WAVES
#define PDT_FUSED_STEP_WAVES 4
#endif
#ifndef PDT_ADV_PHASES  // (diagnostic builds: 1 = stop after the per-prefix lists, 2 = before the history copy)
#define PDT_ADV_PHASES 0
#endif

namespace pdt {

// One WORKGROUP per batch element (a.waves_per_wg waves).  The Kp per-prefix selections over the
// dense extension probabilities are independent and each is a chain of round trips to HBM for a
// lone wave: the waves take prefixes k = w, w + NW, ... in turn (every wave with its own survivor
// scratch), wave 0 runs the frame on the finished lists, all waves copy the histories.
// FUSED (round 5, pdt_ctc_prefix_search_advance_lm): the extension probabilities are never written --
// a wave reads its prefix's row of language-model scores once (registers), reduces it, mixes it with the
// frame's probabilities (the arithmetic of fusion_ext.hip, to the bit) into a row of its own in LDS and
// selects from there; what the frame reads of a row besides its list -- the entries at the prefixes' last
// tokens -- goes to a K' x K' table (DenseCtx::etab).  One kernel and 4 V bytes per prefix instead of two
// kernels and 12 V: fusion_ext 43 us + step 45 us -> see DESIGN.md section 4.4.
template <bool FUSED>
__global__ void __launch_bounds__(512, FUSED ? 4 : 8) ctc_advance_kernel(const CtcAdvArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6), NW = a.waves_per_wg;
  const int64_t n = blockIdx.x;
  const int V = a.V, W = a.W, Kp = a.Kp, S = a.S;
  float *p = reinterpret_cast<float *>(smem);
  FrameLds L;
  L.carve(smem + (size_t)((V + 1 + 3) & ~3) * 4, V, W, Kp, true);
  int *srcs = reinterpret_cast<int *>(L.surv);  // reused after the frame
  u64 *my_surv = reinterpret_cast<u64 *>(smem + a.frame_bytes) + (size_t)wave * PDT_SURV_CAP;

  const int M = ctc_list_len(V, W, Kp);
  // (prefixes that share ONE row of extension probabilities -- the search without a language model hands
  // over nonext.unsqueeze(1).expand(N, K', V), stride 0 -- share one list: built once, copied below)
  const int n_lists = FUSED ? 0 : (a.ext_shared ? 1 : Kp);
  float *etab = nullptr;
  if constexpr (FUSED) {
    const int row_floats = (V + 3) & ~3;
    float *rows_w = reinterpret_cast<float *>(smem + a.frame_bytes + (size_t)NW * PDT_SURV_CAP * 8);
    etab = rows_w;  // [Kp x Kp]; behind it ONE row for the rare fall-back below (wave 0's turn only)
    float *row = etab + ((Kp * Kp + 3) & ~3);
    (void)row_floats;
    const float keep = 1.0f - a.beta, beta = a.beta;
    const bool vm = a.valid_mixture != 0;
    const float scale = vm ? 1.0f - a.blank[n * a.bl_sn] : 0.0f;
    const float *pc = a.nonext + n * a.ne_sn;
    const int lastc = lane < Kp ? (int)min(max(a.last[n * a.la_sn + lane * a.la_sk], (int64_t)0), (int64_t)(V - 1)) : 0;
    const float p_last = lane < Kp ? pc[(int64_t)lastc * a.ne_sv] : 0.0f;
    unsigned overflowed = 0u;  // bit k / NW (wave-uniform): rows of this wave whose survivor buffer overflowed
    // the frame's probabilities are the same for every row of the element: once, in registers
    float pr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int v = lane + i * PDT_WAVE;
      pr[i] = (i * PDT_WAVE < V && v < V) ? pc[(int64_t)v * a.ne_sv] : 0.0f;
    }
    for (int k = wave; k < Kp; k += NW) {
      const float *x = a.lm + (n * Kp + k) * (int64_t)V;
      float r[16];
      const RowStats st = row_stats<false, true, 16>(x, 1, V, r);  // (V <= 1024: the launcher)
      const float log_sum = logf(st.sum);
      auto mix = [&](const float p, const float xv) {
        if (vm) {
          const float lm_p = (expf(xv - st.mx) / st.sum) * scale;
          return keep * p + beta * lm_p;
        }
        return p * expf(beta * ((xv - st.mx) - log_sum));
      };
      // what the frame reads of this row besides its list: the entries at the prefixes' last tokens
      if (lane < Kp) etab[k * Kp + lane] = mix(p_last, x[lastc]);
      unsigned keys[16];  // ordering keys of the mixed values (every one >= +0), 0 beyond the row
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int v = lane + i * PDT_WAVE;
        keys[i] = 0u;
        if (i * PDT_WAVE < V && v < V) keys[i] = fkey_nonneg(mix(pr[i], r[i]));
      }
      u64 tk;
      if (wave_top_sorted_keys<16>(keys, M, my_surv, tk)) {
        if (lane < M) {
          L.tl_tok[k * PDT_WAVE + lane] = (int)idx_of(tk);
          L.tl_p[k * PDT_WAVE + lane] = fkey_nonneg_inv(key_of(tk));
        }
      } else {
        overflowed |= 1u << (k / NW);
      }
      wave_sync();
    }
    // heavy ties in a row (more survivors than the buffer holds): the row through LDS and the chunked merge,
    // one wave at a time -- they share the one LDS row
    for (int w = 0; w < NW; ++w) {
      if (__syncthreads_or(wave == w && overflowed != 0u)) {
        if (wave == w) {
          for (int k = wave; k < Kp; k += NW) {
            if (!((overflowed >> (k / NW)) & 1u)) continue;
            const float *x = a.lm + (n * Kp + k) * (int64_t)V;
            float r[16];
            const RowStats st = row_stats<false, true, 16>(x, 1, V, r);
            const float log_sum = logf(st.sum);
            for (int v = lane; v < V; v += PDT_WAVE) {
              const float p = pc[(int64_t)v * a.ne_sv], xv = x[v];
              row[v] = vm ? keep * p + beta * ((expf(xv - st.mx) / st.sum) * scale) : p * expf(beta * ((xv - st.mx) - log_sum));
            }
            wave_sync();
            const u64 tk = wave_top_sorted<false, true>(row, V, M, my_surv);
            if (lane < M) {
              L.tl_tok[k * PDT_WAVE + lane] = (int)idx_of(tk);
              L.tl_p[k * PDT_WAVE + lane] = fkey_nonneg_inv(key_of(tk));
            }
            wave_sync();
          }
        }
        __syncthreads();
      }
    }
  }
  for (int k = wave; k < n_lists; k += NW) {
    // (rows of 513 .. 1024 elements are read once, into 16 registers per lane: 0.056 -> 0.047 ms at
    // V = 1000; shorter rows measured no better that way, longer ones are streamed twice)
    const float *xk = a.ext + n * a.ext_sn + k * a.ext_sk;
    const u64 tk = V > 8 * PDT_WAVE ? wave_top_sorted_regs<16>(xk, a.ext_sv, V, M, my_surv)
                                    : wave_top_sorted_strided<true>(xk, a.ext_sv, V, M, my_surv);
    if (lane < M) {
      L.tl_tok[k * PDT_WAVE + lane] = (int)idx_of(tk);
      L.tl_p[k * PDT_WAVE + lane] = fkey_inv(key_of(tk));
    }
    wave_sync();
  }
  for (int v = (int)threadIdx.x; v < V; v += NW * PDT_WAVE) p[v] = a.nonext[n * a.ne_sn + v * a.ne_sv];
  if (threadIdx.x == 0) p[V] = a.blank[n * a.bl_sn];
  __syncthreads();
  if (!FUSED && n_lists < Kp) {
    for (int idx = PDT_WAVE + (int)threadIdx.x; idx < Kp * PDT_WAVE; idx += NW * PDT_WAVE) {
      L.tl_tok[idx] = L.tl_tok[idx & (PDT_WAVE - 1)];
      L.tl_p[idx] = L.tl_p[idx & (PDT_WAVE - 1)];
    }
    __syncthreads();
  }
#if PDT_ADV_PHASES == 1
  return;
#endif

  DenseCtx dc;
  dc.ext = FUSED ? nullptr : a.ext + n * a.ext_sn;
  dc.ext_sk = FUSED ? 0 : a.ext_sk;
  dc.ext_sv = FUSED ? 0 : a.ext_sv;
  if constexpr (FUSED) {
    dc.etab = etab;
    dc.etab_stride = Kp;
  }
  dc.y_prev = a.y_prev + n * a.yp_sn;
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
    unsigned pdt_stamp_acc[14] = {0};  // wave-uniform: scalar registers
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
      L.info[lane] = bm.len;        // the frame's scratch is free again: per-entry length and
      L.info[W + lane] = new_kind;  // kind for the copy loop
      // the new token sits right after the source prefix (:862-864)
      if (valid && new_kind != 2) a.y_next[((int64_t)(bm.len - 1) * a.N + n) * W + lane] = new_tok;
    }
  }
  __syncthreads();
#if PDT_ADV_PHASES == 2
  return;
#endif
  // history rows of the source prefix, below the position just written
  for (int idx = (int)threadIdx.x; idx < (S + 1) * W; idx += NW * PDT_WAVE) {
    const int s = idx / W, i = idx - s * W;
    const int src = srcs[i];
    const int len_i = L.info[i], kind_i = L.info[W + i];
    const bool ext_i = kind_i == 0 || kind_i == 1;
    const int plen = len_i - (ext_i ? 1 : 0);
    if (src < 0)
      a.y_next[((int64_t)s * a.N + n) * W + i] = 0;
    else if (!(ext_i && s == plen))
      a.y_next[((int64_t)s * a.N + n) * W + i] = s < S ? dc.y_prev[(int64_t)s * dc.yp_ss + src * dc.yp_sk] : 0;
  }
}

int launch_ctc_advance(CtcAdvArgs a, hipStream_t stream) {
  if (a.W < 1 || a.Kp < 1) return PDT_E_ARG;
  const bool force_wide = switches().step_wide != 0;
  if (a.lm && (force_wide || a.W > kMaxWidth || a.Kp > kMaxWidth || a.V > 16 * PDT_WAVE)) return PDT_E_UNSUPPORTED;
  if (force_wide || a.W > kMaxWidth || a.Kp > kMaxWidth) return launch_ctc_advance_wide(a, stream);  // (advance_wide.hip)
  int nw = 1;
  while (nw < 8 && nw * 2 <= a.Kp) nw *= 2;  // waves per element: a power of two <= min(Kp, 8)
  size_t frame = (size_t)((a.V + 1 + 3) & ~3) * 4 + FrameLds::bytes(a.V, a.W, a.Kp, true);
  frame = (frame + 15) & ~(size_t)15;
  const bool fused = a.lm != nullptr;
  // (the fused form needs ~110 vector registers: four waves per SIMD.  Four-wave workgroups keep every batch
  // element of N = 1024 resident at once -- four rows per wave -- where eight-wave ones run in two rounds)
  if (fused && nw > PDT_FUSED_STEP_WAVES) nw = PDT_FUSED_STEP_WAVES;
  size_t smem = frame + (size_t)nw * PDT_SURV_CAP * 8;  // + one survivor scratch per wave
  if (fused) smem += ((size_t)((a.V + 3) & ~3) + (size_t)((a.Kp * a.Kp + 3) & ~3)) * 4;  // + the K' x K' table, one mixed row (ties)
  if (smem > 160 * 1024) return fused ? PDT_E_UNSUPPORTED : PDT_E_TOO_LONG;
  a.waves_per_wg = nw;
  a.frame_bytes = (int)frame;
  a.ext_shared = (!fused && a.Kp > 1 && a.ext_sk == 0 && switches().step_flat != 0) ? 1 : 0;
  const void *kern = fused ? reinterpret_cast<const void *>(ctc_advance_kernel<true>)
                           : reinterpret_cast<const void *>(ctc_advance_kernel<false>);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  if (fused)
    hipLaunchKernelGGL(ctc_advance_kernel<true>, dim3((unsigned)a.N), dim3(64 * nw), smem, stream, a);
  else
    hipLaunchKernelGGL(ctc_advance_kernel<false>, dim3((unsigned)a.N), dim3(64 * nw), smem, stream, a);
  return (int)hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// One WORKGROUP per batch element (a.waves_per_wg waves): the Kp selections are independent and
// each is a chain of round trips to HBM for a lone wave, so the waves take prefixes k = w, w + NW,
// ... in turn; wave 0 merges the lists (values are kept with the tokens: no global load in the K
// rounds); the history copy is spread over all the waves again.  N = 1024, K = 16, V = 1000:
// 0.164 ms with one wave per element doing everything -> see DESIGN.md section 4.4.
__global__ void __launch_bounds__(512, 8) beam_advance_kernel(const BeamAdvArgs a) {
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

  // per-prefix lists of the best tokens, ranked by the candidate value itself -- the float32 sum
  // log_probs_prev[k] + log_probs_t[k, v] (:122) -- so that candidates whose SUMS are equal come
  // out lowest token first although their log_probs_t differ: with "lowest lane first" in the
  // merge below an exact tie goes to the lowest flat index k * V + v, the oracle's order
  for (int k = wave; k < Kp; k += NW) {
    // (rows in HBM: read once, into registers, where they are short enough -- ctc_advance_kernel's choice)
    const float *row = a.lpt + n * a.lt_sn + k * a.lt_sk;
    const float bias = a.lpp[n * a.lp_sn + k * a.lp_sk];
    const u64 tk = V > 8 * PDT_WAVE ? wave_top_sorted_regs<16, true>(row, a.lt_sv, V, M, surv, bias)
                                    : wave_top_sorted_strided<true, false, true>(row, a.lt_sv, V, M, surv, nullptr, nullptr, 1, bias);
    if (lane < M) {
      tl[k * PDT_WAVE + lane] = (int)idx_of(tk);
      tlm[k * PDT_WAVE + lane] = fkey_inv(key_of(tk));  // the sum that was ranked
    }
    wave_sync();
  }
  __syncthreads();
#if PDT_ADV_PHASES == 1
  return;
#endif
  if (wave == 0) {
    const bool live = lane < Kp;
    const int *mytl = tl + (live ? lane : 0) * PDT_WAVE;
    const float *mytlm = tlm + (live ? lane : 0) * PDT_WAVE;
    int ptr = 0;
    int new_src = 0, new_tok = 0;
    float new_lp = -PDT_INF;
    bool valid = false;
    for (int i = 0; i < K; ++i) {
      const bool has = live && ptr < M;
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
      const int plen = valid ? (a.lens ? (int)a.lens[n * a.le_sn + new_src * a.le_sk] : S) : -1;
      a.lp_next[n * W + lane] = valid ? new_lp : -PDT_INF;            // :145-153 for the overflow
      a.next_src[n * W + lane] = valid ? new_src : 0;
      a.y_next_lens[n * W + lane] = valid ? plen + 1 : 0;
      srcs[lane] = valid ? new_src : -1;
      toks[lane] = new_tok;
      plens[lane] = plen;
    }
  }
  __syncthreads();
#if PDT_ADV_PHASES == 2
  return;
#endif
  for (int idx = (int)threadIdx.x; idx < a.S_out * W; idx += NW * PDT_WAVE) {
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
// The FLAT form (round 5), for dense log_probs_t (the K' * V candidates of a batch element contiguous):
// the K winners are the K largest of ALL the candidates -- no per-prefix lists (sixteen 64-key sorts +
// sixteen survivor sorts per batch element: the lists were 26 of the kernel's 35 us, vector-issue bound).
// The candidates sit in the registers of the element's eight waves, flat index f = k * V + v = 64 * chunk +
// lane, up to kFlatRegs chunks per wave, every load in flight at once:
//   1. per-lane maxima -> one "column" maximum per lane over the eight waves (LDS); the K-th largest of
//      the 64 columns is a lower bound tau of the K-th largest candidate (wave 0: one 64-key sort);
//   2. every wave appends its candidates >= tau to ONE survivor list (LDS cursor), typically K .. K + 4;
//   3. wave 0 sorts them by (sum, lowest flat index first): lane i holds winner i.
// Fewer than K columns with a finite maximum (tau = -inf: finished beams, whose rows are -inf but for one
// token): the survivors are the finite candidates, and -inf candidates fill the rest in flat order, as the
// lists did.  More survivors than the list holds (top candidates crowding a few lanes): every candidate
// again from memory through the chunked top-64 merge.  Same winners, same order, same bits as the list form.
constexpr int kFlatWaves = 8, kFlatRegs = 32;

__global__ void __launch_bounds__(64 * kFlatWaves, 6) beam_advance_flat_kernel(const BeamAdvArgs a) {
  __shared__ unsigned colmax[kFlatWaves * PDT_WAVE];
  __shared__ u64 surv[PDT_SURV_CAP];
  __shared__ unsigned ctl[4];  // [0] the threshold (float bits), [1] the survivor cursor
  __shared__ int srcs[PDT_WAVE], toks[PDT_WAVE], plens[PDT_WAVE];
  int lane = lane_id();
  asm volatile("" : "+v"(lane));
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (scalar: so is everything derived)
  const int64_t n = blockIdx.x;
  const int V = a.V, W = a.W, Kp = a.Kp, S = a.S;
  const int K = min(W, (int)min((int64_t)Kp * V, (int64_t)PDT_WAVE));  // :121
  const int FV = Kp * V, total = (FV + PDT_WAVE - 1) >> 6;
  const int per = (total + kFlatWaves - 1) / kFlatWaves;  // chunks per wave, <= kFlatRegs (the launcher)
  const float *lpt = a.lpt + n * a.lt_sn;
  if (threadIdx.x == 0) ctl[1] = 0u;
  // log_probs_prev of prefix k in lane k (K' <= 64): a candidate's addend is one permute away
  const float bias_v = lane < Kp ? a.lpp[n * a.lp_sn + lane * a.lp_sk] : 0.0f;

  // ---- 1. the candidates, (log_probs_prev[k] + log_probs_t[k, v]) + 0 (:122; + 0.0f: the two zeros tie)
  const int c0 = wave * per;
  const int f0 = c0 * PDT_WAVE + lane;
  float x[kFlatRegs];
#pragma unroll
  for (int j = 0; j < kFlatRegs; ++j) {
    x[j] = -PDT_INF;  // (beyond the candidates: never a survivor, see the threshold below)
    if (j < per) {
      if ((c0 + j + 1) * PDT_WAVE <= FV)
        x[j] = lpt[f0 + j * PDT_WAVE];
      else if ((c0 + j) * PDT_WAVE < FV && f0 + j * PDT_WAVE < FV)
        x[j] = lpt[f0 + j * PDT_WAVE];
    }
  }
  // (every load in flight before the first is used: the sums in a second loop; a lane's prefix k moves on
  // by at most one per chunk, V > 64)
  float lmax = -PDT_INF;
  {
    int k = f0 / V, v = f0 - k * V;
#pragma unroll
    for (int j = 0; j < kFlatRegs; ++j) {
      if ((j & 7) == 0) __builtin_amdgcn_sched_barrier(0);  // (eight permutes ahead at most: registers)
      x[j] = (shfl_f(bias_v, k) + x[j]) + 0.0f;
      lmax = fmaxf(lmax, x[j]);
      v += PDT_WAVE;
      const bool wrap = v >= V;
      v -= wrap ? V : 0;
      k += wrap ? 1 : 0;
    }
  }
  colmax[wave * PDT_WAVE + lane] = fkey(lmax);
  __syncthreads();
#if PDT_ADV_PHASES == 1
  if (lmax != 12345.0f) return;
#endif
  if (wave == 0) {
    unsigned cm = colmax[lane];
#pragma unroll
    for (int w = 1; w < kFlatWaves; ++w) cm = max(cm, colmax[w * PDT_WAVE + lane]);
    const unsigned sorted_max = wave_sort_desc<unsigned>(cm);
    // x >= the lowest finite float <=> x > -inf: the padding and the -inf candidates stay out
    const float tau = fmaxf(fkey_inv((unsigned)__builtin_amdgcn_readlane((int)sorted_max, K - 1)), -3.4028234664e38f);
    if (lane == 0) ctl[0] = __float_as_uint(tau);
  }
  __syncthreads();
  // ---- 2. survivors ------------------------------------------------------------------------
  const float tau = __uint_as_float(ctl[0]);
#if PDT_ADV_PHASES == 3
  if (tau != 12345.0f) return;
#endif
  if (__ballot(lmax >= tau)) {
#pragma unroll
    for (int j = 0; j < kFlatRegs; ++j) {
      const bool pred = x[j] >= tau;
      if (__ballot(pred)) {
        if (pred) {
          const unsigned at = atomicAdd(&ctl[1], 1u);
          if (at < PDT_SURV_CAP) surv[at] = pack_key(fkey(x[j]), (unsigned)(f0 + j * PDT_WAVE));
        }
      }
    }
  }
  __syncthreads();
#if PDT_ADV_PHASES == 4
  return;
#endif
  // ---- 3. the winners in order ---------------------------------------------------------------
  if (wave == 0) {
    const int count = (int)ctl[1];
    u64 tk = 0ull;
    int filled = 0;  // winners that are in tk
    if (count <= PDT_SURV_CAP) {
      tk = wave_sort_desc<u64>(lane < count ? surv[lane] : 0ull);
      filled = min(count, K);
    } else {
      // every candidate, -inf ones included, from memory
      for (int g0 = 0; g0 < FV; g0 += PDT_WAVE) {
        const int f = g0 + lane;
        const float xv = f < FV ? (a.lpp[n * a.lp_sn + (f / V) * a.lp_sk] + lpt[f]) + 0.0f : 0.0f;
        tk = wave_merge_top64(tk, f < FV ? pack_key(fkey(xv), (unsigned)f) : 0ull);
      }
      filled = K;
    }
    const int flat = (int)idx_of(tk);
    int new_src = flat / V, new_tok = flat - new_src * V;
    float new_lp = fkey_inv(key_of(tk));
    if (filled < K) {
      // tau = -inf and fewer than K finite candidates: -inf candidates in flat order behind them
      int have = filled;
      for (int g0 = 0; g0 < FV && have < K; g0 += PDT_WAVE) {
        const int f = g0 + lane;
        const int k = f / V;
        const bool is = f < FV && (a.lpp[n * a.lp_sn + k * a.lp_sk] + lpt[f]) + 0.0f == -PDT_INF;
        const u64 b = __ballot(is);
        const int rank = have + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
        // winner number `rank` is this lane's candidate: hand it to lane `rank`
        if (is && rank < K) {
          srcs[rank] = k;
          toks[rank] = f - k * V;
        }
        have += __popcll(b);
      }
      wave_sync();
      if (lane >= filled && lane < K) {
        new_src = srcs[lane];
        new_tok = toks[lane];
        new_lp = -PDT_INF;
      }
      wave_sync();
    }
    if (lane < W) {
      const bool valid = lane < K;
      const int plen = valid ? (a.lens ? (int)a.lens[n * a.le_sn + new_src * a.le_sk] : S) : -1;
      a.lp_next[n * W + lane] = valid ? new_lp : -PDT_INF;  // :145-153 for the overflow
      a.next_src[n * W + lane] = valid ? new_src : 0;
      a.y_next_lens[n * W + lane] = valid ? plen + 1 : 0;
      srcs[lane] = valid ? new_src : -1;
      toks[lane] = new_tok;
      plens[lane] = plen;
    }
  }
  __syncthreads();
#if PDT_ADV_PHASES == 2
  return;
#endif
  for (int idx = (int)threadIdx.x; idx < a.S_out * W; idx += kFlatWaves * PDT_WAVE) {
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

// What beam_search_advance's caller must know before it can size y_next (reference :133-140): does some path
// reach the history's S rows (bit 0), is some length non-zero (bit 1: invalid at S = 0).  One workgroup, one
// plain store to a word in pinned host memory the host reads after synchronising the stream -- instead of a
// reduction kernel, a device-to-host copy and the synchronisation.
__global__ void __launch_bounds__(1024) lens_reach_kernel(const int64_t *lens, const int64_t le_sn, const int64_t le_sk,
                                                         const int N, const int Kp, const int64_t S, int32_t *flag) {
  __shared__ int any_[2];
  if (threadIdx.x < 2) any_[threadIdx.x] = 0;
  __syncthreads();
  bool reach = false, nonzero = false;
  for (int64_t i = threadIdx.x; i < (int64_t)N * Kp; i += 1024) {
    const int64_t n = i / Kp, k = i - n * Kp;
    const int64_t l = lens[n * le_sn + k * le_sk];
    reach = reach || l >= S;
    nonzero = nonzero || l != 0;
  }
  if (__ballot(reach) && lane_id() == 0) any_[0] = 1;
  if (__ballot(nonzero) && lane_id() == 0) any_[1] = 1;
  __syncthreads();
  // (bit 2: the word has been written -- a system-scope store the host may poll for instead of synchronising)
  if (threadIdx.x == 0) __hip_atomic_store(flag, any_[0] | (any_[1] << 1) | 4, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int launch_beam_advance(BeamAdvArgs a, hipStream_t stream) {
  if (a.W < 1 || a.Kp < 1) return PDT_E_ARG;
  const bool force_wide = switches().step_wide != 0;
  if (force_wide || a.W > PDT_WAVE || a.Kp > PDT_WAVE) return launch_beam_advance_wide(a, stream);  // (advance_wide.hip)
  const int64_t cands = (int64_t)a.Kp * a.V;
  if (switches().step_flat != 0 && a.V > PDT_WAVE && a.lt_sv == 1 && a.lt_sk == a.V &&
      cands <= (int64_t)kFlatWaves * kFlatRegs * PDT_WAVE) {
    hipLaunchKernelGGL(beam_advance_flat_kernel, dim3((unsigned)a.N), dim3(64 * kFlatWaves), 0, stream, a);
    return (int)hipGetLastError();
  }
  int nw = 1;
  while (nw < 8 && nw * 2 <= a.Kp) nw *= 2;  // waves per element: a power of two <= min(Kp, 8)
  const size_t smem = ((size_t)nw * PDT_SURV_CAP * 8 + (size_t)a.Kp * PDT_WAVE * 8 + (size_t)a.W * 12 + 15) & ~(size_t)15;
  a.waves_per_wg = nw;
  hipLaunchKernelGGL(beam_advance_kernel, dim3((unsigned)a.N), dim3(64 * nw), smem, stream, a);
  return (int)hipGetLastError();
}

}  // namespace pdt

extern "C" {

int pdt_ctc_prefix_search_advance(
    const float *ext, int64_t ext_sn, int64_t ext_sk, int64_t ext_sv, const float *nonext,
    int64_t ne_sn, int64_t ne_sv, const float *blank, int64_t bl_sn, int64_t N, int64_t Kp,
    int64_t V, int64_t width, const float *nb_prev, int64_t nb_sn, int64_t nb_sk,
    const float *b_prev, int64_t b_sn, int64_t b_sk, const int64_t *y_prev, int64_t S,
    int64_t yp_ss, int64_t yp_sn, int64_t yp_sk, const int64_t *y_prev_last, int64_t la_sn,
    int64_t la_sk, const int64_t *y_prev_lens, int64_t le_sn, int64_t le_sk,
    const uint8_t *prev_is_prefix, int64_t ip_sn, int64_t ip_sa, int64_t ip_sb, int64_t *y_next,
    int64_t *y_next_last, int64_t *y_next_lens, float *nb_next, float *b_next,
    uint8_t *next_is_prefix, int64_t *next_src, uint8_t *next_is_nonext, void *stream) {
  using namespace pdt;
  if (N < 0 || Kp < 1 || V < 1 || width < 1 || S < 0) return PDT_E_ARG;
  if (N == 0) return PDT_OK;
  if (!ext || !nonext || !blank || !nb_prev || !b_prev || !y_prev_last || !y_prev_lens ||
      !prev_is_prefix || (S > 0 && !y_prev) || !y_next || !y_next_last || !y_next_lens ||
      !nb_next || !b_next || !next_is_prefix || !next_src || !next_is_nonext)
    return PDT_E_ARG;
  if (V >= (1 << 30) || S >= (1 << 26) || N >= (1ll << 31)) return PDT_E_TOO_LONG;
  CtcAdvArgs a{};
  a.ext = ext; a.ext_sn = ext_sn; a.ext_sk = ext_sk; a.ext_sv = ext_sv;
  a.nonext = nonext; a.ne_sn = ne_sn; a.ne_sv = ne_sv;
  a.blank = blank; a.bl_sn = bl_sn;
  a.nb_prev = nb_prev; a.pb_sn = nb_sn; a.pb_sk = nb_sk;
  a.b_prev = b_prev; a.pbb_sn = b_sn; a.pbb_sk = b_sk;
  a.y_prev = y_prev; a.yp_ss = yp_ss; a.yp_sn = yp_sn; a.yp_sk = yp_sk;
  a.last = y_prev_last; a.la_sn = la_sn; a.la_sk = la_sk;
  a.lens = y_prev_lens; a.le_sn = le_sn; a.le_sk = le_sk;
  a.isp = prev_is_prefix; a.ip_sn = ip_sn; a.ip_sa = ip_sa; a.ip_sb = ip_sb;
  a.N = (int)N; a.Kp = (int)Kp; a.V = (int)V; a.W = (int)width; a.S = (int)S;
  a.y_next = y_next; a.y_next_last = y_next_last; a.y_next_lens = y_next_lens;
  a.next_src = next_src; a.nb_next = nb_next; a.b_next = b_next;
  a.next_isp = next_is_prefix; a.next_nonext = next_is_nonext;
  return launch_ctc_advance(a, (hipStream_t)stream);
}

int pdt_ctc_prefix_search_advance_lm(
    const float *lm_log_probs, float beta, int valid_mixture, const float *nonext, int64_t ne_sn,
    int64_t ne_sv, const float *blank, int64_t bl_sn, int64_t N, int64_t Kp, int64_t V, int64_t width,
    const float *nb_prev, int64_t nb_sn, int64_t nb_sk, const float *b_prev, int64_t b_sn, int64_t b_sk,
    const int64_t *y_prev, int64_t S, int64_t yp_ss, int64_t yp_sn, int64_t yp_sk,
    const int64_t *y_prev_last, int64_t la_sn, int64_t la_sk, const int64_t *y_prev_lens, int64_t le_sn,
    int64_t le_sk, const uint8_t *prev_is_prefix, int64_t ip_sn, int64_t ip_sa, int64_t ip_sb,
    int64_t *y_next, int64_t *y_next_last, int64_t *y_next_lens, float *nb_next, float *b_next,
    uint8_t *next_is_prefix, int64_t *next_src, uint8_t *next_is_nonext, void *stream) {
  using namespace pdt;
  if (N < 0 || Kp < 1 || V < 1 || width < 1 || S < 0) return PDT_E_ARG;
  if (N == 0) return PDT_OK;
  if (!lm_log_probs || !nonext || !blank || !nb_prev || !b_prev || !y_prev_last || !y_prev_lens ||
      !prev_is_prefix || (S > 0 && !y_prev) || !y_next || !y_next_last || !y_next_lens ||
      !nb_next || !b_next || !next_is_prefix || !next_src || !next_is_nonext)
    return PDT_E_ARG;
  if (V >= (1 << 30) || S >= (1 << 26) || N >= (1ll << 31)) return PDT_E_TOO_LONG;
  CtcAdvArgs a{};
  a.lm = lm_log_probs; a.beta = beta; a.valid_mixture = valid_mixture;
  a.nonext = nonext; a.ne_sn = ne_sn; a.ne_sv = ne_sv;
  a.blank = blank; a.bl_sn = bl_sn;
  a.nb_prev = nb_prev; a.pb_sn = nb_sn; a.pb_sk = nb_sk;
  a.b_prev = b_prev; a.pbb_sn = b_sn; a.pbb_sk = b_sk;
  a.y_prev = y_prev; a.yp_ss = yp_ss; a.yp_sn = yp_sn; a.yp_sk = yp_sk;
  a.last = y_prev_last; a.la_sn = la_sn; a.la_sk = la_sk;
  a.lens = y_prev_lens; a.le_sn = le_sn; a.le_sk = le_sk;
  a.isp = prev_is_prefix; a.ip_sn = ip_sn; a.ip_sa = ip_sa; a.ip_sb = ip_sb;
  a.N = (int)N; a.Kp = (int)Kp; a.V = (int)V; a.W = (int)width; a.S = (int)S;
  a.y_next = y_next; a.y_next_last = y_next_last; a.y_next_lens = y_next_lens;
  a.next_src = next_src; a.nb_next = nb_next; a.b_next = b_next;
  a.next_isp = next_is_prefix; a.next_nonext = next_is_nonext;
  return launch_ctc_advance(a, (hipStream_t)stream);
}

int pdt_lens_reach(const int64_t *lens, int64_t le_sn, int64_t le_sk, int64_t N, int64_t Kp, int64_t S,
                   int32_t *host_flag, void *stream) {
  using namespace pdt;
  if (N < 0 || Kp < 0 || !host_flag || N >= (1ll << 31) || Kp >= (1ll << 31)) return PDT_E_ARG;
  if (N * Kp == 0) {
    *host_flag = 4;
    return PDT_OK;
  }
  if (!lens) return PDT_E_ARG;
  hipLaunchKernelGGL(lens_reach_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, lens, le_sn, le_sk, (int)N, (int)Kp, S,
                     host_flag);
  return (int)hipGetLastError();
}

int pdt_beam_search_advance(const float *log_probs_t, int64_t lt_sn, int64_t lt_sk, int64_t lt_sv,
                            int64_t N, int64_t Kp, int64_t V, int64_t width,
                            const float *log_probs_prev, int64_t lp_sn, int64_t lp_sk,
                            const int64_t *y_prev, int64_t S, int64_t yp_ss, int64_t yp_sn,
                            int64_t yp_sk, const int64_t *y_prev_lens, int64_t le_sn,
                            int64_t le_sk, int64_t S_out, int64_t *y_next, int64_t *y_next_lens,
                            float *log_probs_next, int64_t *next_src, void *stream) {
  using namespace pdt;
  if (N < 0 || Kp < 1 || V < 1 || width < 1 || S < 0 || (S_out != S && S_out != S + 1))
    return PDT_E_ARG;
  if (N == 0) return PDT_OK;
  if (!log_probs_t || !log_probs_prev || (S > 0 && !y_prev) || !y_next || !y_next_lens ||
      !log_probs_next || !next_src)
    return PDT_E_ARG;
  if (V >= (1 << 30) || S >= (1 << 26) || N >= (1ll << 31)) return PDT_E_TOO_LONG;
  BeamAdvArgs a{};
  a.lpt = log_probs_t; a.lt_sn = lt_sn; a.lt_sk = lt_sk; a.lt_sv = lt_sv;
  a.lpp = log_probs_prev; a.lp_sn = lp_sn; a.lp_sk = lp_sk;
  a.y_prev = y_prev; a.yp_ss = yp_ss; a.yp_sn = yp_sn; a.yp_sk = yp_sk;
  a.lens = y_prev_lens; a.le_sn = le_sn; a.le_sk = le_sk;
  a.N = (int)N; a.Kp = (int)Kp; a.V = (int)V; a.W = (int)width; a.S = (int)S; a.S_out = (int)S_out;
  a.y_next = y_next; a.y_next_lens = y_next_lens; a.lp_next = log_probs_next; a.next_src = next_src;
  return launch_beam_advance(a, (hipStream_t)stream);
}

}  // extern "C"
