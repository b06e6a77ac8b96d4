// Row-synchronous Levenshtein for gfx950: all 64 lanes of a wave work on the SAME DP row.
//
// Used where the whole row must be visible at once:
//   * optimal_completion's row-minimum mask (reference _string.py:319-339, :347-355), and
//   * cost-mode distances with costs whose partial sums are NOT exactly representable in
//     float32, where the reference's O(R^2) "deletion unroll" (_string.py:258-266, :317)
//     rounds differently from the textbook recurrence; EXACT = true reproduces the unroll
//     term by term so results stay bit-identical.
//
// One wave per utterance; lane l holds DP columns l*CPL+1 .. (l+1)*CPL in registers, column 0
// is a wave-uniform scalar.  The deletion closure row[c] = min_k<=c (t[k] + (c-k)*del) is a
// min-plus prefix scan: in-lane running minimum + one 7-step DPP wave scan (EXACT = false).
//
// Tokens are replaced by their CLASS RANK (rank among the distinct reference tokens, found
// by an in-LDS bitonic sort), so equality tests are int32 whatever the int64 token values,
// and optimal-completion sets come out as class bitmasks whose set bits are already in the
// ascending token order the reference produces with sort + masked_scatter (:503-514).
#include "lev_common.hpp"

namespace pdt {

struct RowsyncLds {
  // byte offsets inside a wave's LDS slice.  The sort buffer is dead once the distinct
  // tokens have been compacted into `ctok`, so hyp ids / row buffers / bitmask alias it.
  int off_sort, off_ctok, off_hyp, off_tbuf, off_row0, off_bnd, off_bm;
  int P;  // sort capacity (power of two >= R)
};

__device__ __forceinline__ int lower_bound_i64(const int64_t *tab, int n, int64_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (tab[mid] < v)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// class rank of v among tab[0..U), or -1 when absent
__device__ __forceinline__ int class_of(const int64_t *tab, int U, int64_t v) {
  const int k = lower_bound_i64(tab, U, v);
  return (k < U && tab[k] == v) ? k : -1;
}

template <int CPL, bool EXACT>
__device__ __forceinline__ void rowsync_body(const LevArgs &a, const int64_t n, const int ref_len,
                                             const int hyp_len, const int Heff, const int U,
                                             const int64_t *ctok, const int *hyp_l, float *tbuf,
                                             float *row0_l, float *bnd, unsigned *bm,
                                             int &max_cnt) {
  const int lane = lane_id();
  const float ins = a.ins, del = a.del, sub = a.sub;
  const bool mask_mode = a.bitmask != nullptr;
  const int cbase = lane * CPL + 1;
  int rid[CPL], rnext[CPL];
  float prev[CPL], cdel[CPL], cdel_back[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = cbase + j;
    rid[j] = -2;
    if (c <= ref_len) rid[j] = class_of(ctok, U, a.ref[(int64_t)(c - 1) * a.ref_st + n * a.ref_sn]);
    cdel[j] = (float)c * del;
    // added back after the scan: +inf beyond ref_len, which IS the masked_fill(inf) of :332
    // (the scan's operands stay finite there, so no select per column and row is needed)
    cdel_back[j] = c <= ref_len ? cdel[j] : PDT_INF;
    prev[j] = cdel_back[j];  // row 0 (:258-263)
  }
  // class of ref[c]: the optimal "next token" when the row minimum sits in column c
  const int rid_right = __shfl_down(rid[0], 1);
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int nxt = (j + 1 < CPL) ? rid[(j + 1) % CPL] : rid_right;
    rnext[j] = (cbase + j) < ref_len ? nxt : -1;
  }
  const int rank0 = __builtin_amdgcn_readfirstlane(rid[0]);  // class of ref[0]
  if (EXACT) {
    for (int c = lane; c <= ref_len; c += PDT_WAVE) row0_l[c] = (float)c * del;
  }
  float col0 = 0.0f;
  const int W = a.W;
  const int Hout = a.H + (a.exclude_last ? 0 : 1);

  if (mask_mode) {  // h = 0: only column 0 (:271-278)
    if (lane < W) {
      unsigned w = 0;
      if (ref_len > 0 && (rank0 >> 5) == lane) w = 1u << (rank0 & 31);
      a.bitmask[((int64_t)0 * a.N + n) * W + lane] = w;
    }
    if (ref_len > 0 && max_cnt < 1) max_cnt = 1;
  }
  // which register holds column ref_len (for FINAL / PREFIX extraction)
  const int sel_lane = ref_len > 0 ? (ref_len - 1) / CPL : 0;
  const int sel_j = ref_len > 0 ? (ref_len - 1) % CPL : 0;

  for (int h = 1; h <= Heff; ++h) {
    const int tok = hyp_l[h - 1];
    const float col0_new = col0 + ins;  // :292
    float diag = shr1(prev[CPL - 1], col0);
    float t[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const float sc = diag + ((rid[j] != tok) ? sub : 0.0f);  // :293
      t[j] = fminf(prev[j] + ins, sc);                          // :316
      diag = prev[j];
    }
    if (!EXACT) {
      float loc[CPL];
      float run = PDT_INF;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        run = fminf(run, t[j] - cdel[j]);
        loc[j] = run;
      }
      const float incl = wave_incl_scan_min(run);
      const float carry = fminf(shr1(incl, PDT_INF), col0_new);
#pragma unroll
      for (int j = 0; j < CPL; ++j) prev[j] = fminf(loc[j], carry) + cdel_back[j];
    } else {
      // row[c] = min_{k<=c} ((row0[c] - row0[k]) + t[k])   (_string.py:264-266, :317)
      wave_sync();
      if (lane == 0) tbuf[0] = col0_new;
#pragma unroll
      for (int j = 0; j < CPL; ++j)
        if (cbase + j <= ref_len) tbuf[cbase + j] = t[j];
      wave_sync();
      float best[CPL];
#pragma unroll
      for (int j = 0; j < CPL; ++j) best[j] = PDT_INF;
      for (int k = 0; k <= ref_len; ++k) {
        const float tk = tbuf[k], r0k = row0_l[k];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const float v = (cdel[j] - r0k) + tk;
          best[j] = (k <= cbase + j) ? fminf(best[j], v) : best[j];
        }
      }
#pragma unroll
      for (int j = 0; j < CPL; ++j) prev[j] = best[j];
    }
    if (EXACT) {
#pragma unroll
      for (int j = 0; j < CPL; ++j)
        if (cbase + j > ref_len) prev[j] = PDT_INF;  // :332
    }
    col0 = col0_new;

    if (mask_mode) {
      float m = col0;
#pragma unroll
      for (int j = 0; j < CPL; ++j) m = fminf(m, prev[j]);
      m = wave_min(m);  // :333
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        if (prev[j] == m && rnext[j] >= 0)  // :334 and the r < ref_len cut of :349-354
          atomicOr(&bm[rnext[j] >> 5], 1u << (rnext[j] & 31));
      }
      if (lane == 0 && col0 == m && ref_len > 0) atomicOr(&bm[rank0 >> 5], 1u << (rank0 & 31));
      wave_sync();
      int cnt = 0;
      if (lane < W) {
        const unsigned w = bm[lane];
        bm[lane] = 0u;
        a.bitmask[((int64_t)h * a.N + n) * W + lane] = w;
        cnt = __popc(w);
      }
      cnt = W <= 16 ? row0_sum(cnt) : wave_sum(cnt);  // (only lanes < W hold words)
      max_cnt = cnt > max_cnt ? cnt : max_cnt;
      wave_sync();
    } else if (a.mode == PDT_MODE_PREFIX) {
      float v = col0;
#pragma unroll
      for (int j = 0; j < CPL; ++j) v = (j == sel_j && ref_len > 0) ? prev[j] : v;
      if (lane == sel_lane) bnd[h] = v;
    }
  }

  if (mask_mode) {
    // rows of finished hypotheses carry empty sets (`& not_done`, :334)
    for (int h = Heff + 1; h < Hout; ++h)
      if (lane < W) a.bitmask[((int64_t)h * a.N + n) * W + lane] = 0u;
    return;
  }
  // FINAL / PREFIX epilogue (cost semantics only) --------------------------------------
  float fin = col0;
#pragma unroll
  for (int j = 0; j < CPL; ++j) fin = (j == sel_j && ref_len > 0) ? prev[j] : fin;
  fin = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fin), __builtin_amdgcn_readfirstlane(sel_lane)));
  wave_sync();
  const float r0 = (float)ref_len * del;
  if (a.mode == PDT_MODE_FINAL) {
    if (lane == 0)
      a.out[n * a.out_sn] =
          lev_finish(Heff > 0 ? fin : r0, a.mult, a.norm, ref_len, hyp_len > 0 ? 1.0f : 0.0f);
  } else {
    const int pad_from = hyp_len + (a.exclude_last ? 0 : 1);
    for (int h = lane; h < Hout; h += PDT_WAVE) {
      float v;
      if (h >= pad_from)
        v = a.padding;
      else
        v = lev_finish(h == 0 ? r0 : bnd[h], a.mult, a.norm, ref_len, h > 0 ? 1.0f : 0.0f);
      a.out[(int64_t)h * a.out_sh + n * a.out_sn] = v;
    }
  }
}

// BIG = false: R <= 512 (<= 8 columns per lane, ~60 VGPRs, high occupancy);
// BIG = true : R <= 2048 (up to 32 columns per lane).
template <bool EXACT, bool BIG>
__global__ void __launch_bounds__(256) lev_rowsync_kernel(const LevArgs a, const RowsyncLds L) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int64_t n = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * a.waves_per_wg + wave;
  if (n >= a.N) return;
  unsigned char *base = smem + (size_t)wave * a.lds_per_wave;
  int *hyp_l = reinterpret_cast<int *>(base + L.off_hyp);
  int64_t *srt = reinterpret_cast<int64_t *>(base + L.off_sort);
  int64_t *ctok = reinterpret_cast<int64_t *>(base + L.off_ctok);
  float *tbuf = reinterpret_cast<float *>(base + L.off_tbuf);
  float *row0_l = reinterpret_cast<float *>(base + L.off_row0);
  float *bnd = reinterpret_cast<float *>(base + L.off_bnd);
  unsigned *bm = reinterpret_cast<unsigned *>(base + L.off_bm);

  // ---- lengths (_string.py:195-228) -----------------------------------------------------
  int ref_len = a.R, hyp_len = a.H;
  bool rmiss = false, hmiss = false;
  const int64_t roff = n * a.ref_sn, hoff = n * a.hyp_sn;
  if (a.has_eos) {
    for (int t0 = 0; t0 < a.R; t0 += PDT_WAVE) {
      const int t = t0 + lane;
      const unsigned long long b = __ballot(t < a.R && a.ref[(int64_t)t * a.ref_st + roff] == a.eos);
      if (b) {
        ref_len = t0 + (int)__builtin_ctzll(b);
        break;
      }
    }
    for (int t0 = 0; t0 < a.H; t0 += PDT_WAVE) {
      const int t = t0 + lane;
      const unsigned long long b = __ballot(t < a.H && a.hyp[(int64_t)t * a.hyp_st + hoff] == a.eos);
      if (b) {
        hyp_len = t0 + (int)__builtin_ctzll(b);
        break;
      }
    }
    if (a.include_eos) {
      if (ref_len == a.R) rmiss = true; else ref_len += 1;
      if (hyp_len == a.H) hmiss = true; else hyp_len += 1;
    }
  }
  int Heff = a.exclude_last ? hyp_len - 1 : hyp_len;
  if (Heff < 0) Heff = 0;

  // ---- distinct reference tokens in ascending order: bitonic sort in LDS ---------------
  const int P = L.P;
  for (int i = lane; i < P; i += PDT_WAVE)
    srt[i] = i < ref_len ? a.ref[(int64_t)i * a.ref_st + roff] : INT64_MAX;
  wave_sync();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < (P >> 1); t += PDT_WAVE) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const int64_t x = srt[i], y = srt[l];
        const bool up = (i & k) == 0;
        if ((x > y) == up) {
          srt[i] = y;
          srt[l] = x;
        }
      }
      wave_sync();
    }
  }
  // unique-compact the first ref_len sorted entries -> ctok[0..U) (blocked per lane)
  int U = 0;
  {
    const int B = (P + PDT_WAVE - 1) / PDT_WAVE;
    const int i0 = lane * B;
    int nfirst = 0;
    for (int q = 0; q < B; ++q) {
      const int i = i0 + q;
      if (i < ref_len && (i == 0 || srt[i] != srt[i - 1])) ++nfirst;
    }
    const int incl = wave_incl_scan_add(nfirst);
    U = __builtin_amdgcn_readlane(incl, PDT_WAVE - 1);
    int pos = incl - nfirst;
    for (int q = 0; q < B; ++q) {
      const int i = i0 + q;
      if (i < ref_len && (i == 0 || srt[i] != srt[i - 1])) ctok[pos++] = srt[i];
    }
  }
  wave_sync();  // ctok complete; the sort buffer is dead from here on (hyp_l etc. alias it)
  if (a.class_tokens)
    for (int k = lane; k < U; k += PDT_WAVE) a.class_tokens[n * (int64_t)a.R + k] = ctok[k];

  // hypothesis tokens -> class ranks (-1: token does not occur in ref)
  for (int t = lane; t < hyp_len && t < a.H; t += PDT_WAVE)
    hyp_l[t] = class_of(ctok, U, a.hyp[(int64_t)t * a.hyp_st + hoff]);
  if (lane < a.W && a.bitmask) bm[lane] = 0u;
  wave_sync();

  int max_cnt = 0;
  const int cpl = ref_len > 0 ? (ref_len + PDT_WAVE - 1) / PDT_WAVE : 1;
#define PDT_RS(C) rowsync_body<C, EXACT>(a, n, ref_len, hyp_len, Heff, U, ctok, hyp_l, tbuf, row0_l, bnd, bm, max_cnt)
  if (EXACT) {
    if (!BIG || cpl <= 8) PDT_RS(8);
    else if (BIG) PDT_RS(32);
  } else {
    if (cpl <= 1) PDT_RS(1);
    else if (cpl <= 2) PDT_RS(2);
    else if (cpl <= 3) PDT_RS(3);
    else if (cpl <= 4) PDT_RS(4);
    else if (cpl <= 6) PDT_RS(6);
    else if (!BIG || cpl <= 8) PDT_RS(8);
    else if (BIG) {
      if (cpl <= 12) PDT_RS(12);
      else if (cpl <= 16) PDT_RS(16);
      else if (cpl <= 24) PDT_RS(24);
      else PDT_RS(32);
    }
  }
#undef PDT_RS
  if (lane == 0) {
    int flags = 0;
    if (rmiss) flags |= PDT_WARN_REF_NO_EOS;
    if (hmiss) flags |= PDT_WARN_HYP_NO_EOS;
    if (a.norm && ref_len == 0 && !a.bitmask) flags |= PDT_WARN_EMPTY_REF;
    if (flags && a.status) atomicOr(a.status, flags);
    if (a.max_count && max_cnt > 0) atomicMax(a.max_count, max_cnt);
    if (a.ref_lens_out) a.ref_lens_out[n] = ref_len;
    if (a.hyp_lens_out) a.hyp_lens_out[n] = hyp_len;
  }
}

int launch_lev_rowsync(LevArgs a, bool exact, hipStream_t stream) {
  if (a.R > 64 * 32) return PDT_E_TOO_LONG;
  const bool mask_mode = a.bitmask != nullptr;
  RowsyncLds L;
  int P = 2;
  while (P < a.R) P <<= 1;
  L.P = P;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 15) & ~(size_t)15;
    return (int)o;
  };
  L.off_sort = 0;
  L.off_hyp = take((size_t)a.H * 4 + 4);
  L.off_tbuf = take(exact ? (size_t)(a.R + 1) * 4 : 4);
  L.off_row0 = take(exact ? (size_t)(a.R + 1) * 4 : 4);
  L.off_bnd = take(!mask_mode ? (size_t)(a.H + 1) * 4 : 4);
  L.off_bm = take(mask_mode ? (size_t)a.W * 4 : 4);
  if (off < (size_t)P * 8) off = (size_t)P * 8;
  L.off_ctok = take((size_t)a.R * 8 + 8);
  const size_t per_wave = off;
  const size_t soft_cap = 64 * 1024, hard_cap = 160 * 1024;
  if (per_wave > hard_cap) return PDT_E_TOO_LONG;
  int wpw = (int)(soft_cap / per_wave);
  if (wpw > 4) wpw = 4;
  if (wpw < 1) wpw = 1;
  a.waves_per_wg = wpw;
  a.lds_per_wave = (int)per_wave;
  const size_t smem = per_wave * wpw;
  const unsigned grid = (unsigned)((a.N + wpw - 1) / wpw);
  const bool big = a.R > 64 * 8;
  auto kern = exact ? (big ? lev_rowsync_kernel<true, true> : lev_rowsync_kernel<true, false>)
                    : (big ? lev_rowsync_kernel<false, true> : lev_rowsync_kernel<false, false>);
  if (smem > soft_cap) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpw), smem, stream, a, L);
  return (int)hipGetLastError();
}

// ---- phase 2: class bitmasks -> padded ascending token lists (_string.py:509-514) --------
// One workgroup per utterance: the class-token table is staged once in LDS, the four waves take
// rows h = wave, wave + 4, ... and keep the next row's bitmask word in flight while the current
// row is expanded and written (C * 8 contiguous bytes per row).
__global__ void __launch_bounds__(256)
oc_expand_kernel(const uint32_t *__restrict__ bitmask, const int64_t *__restrict__ class_tokens,
                 int R, int W, int Hout, int64_t N, int C, int64_t padding,
                 int64_t *__restrict__ targets, int64_t tgt_sh, int64_t tgt_sn) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int64_t n = xcd_remap(blockIdx.x, gridDim.x);
  int64_t *ctok = reinterpret_cast<int64_t *>(smem);
  int64_t *stage = ctok + R + (size_t)wave * W * 32;
  for (int k = (int)threadIdx.x; k < R; k += 256) ctok[k] = class_tokens[n * (int64_t)R + k];
  __syncthreads();
  const bool wide = (C & 1) == 0 && ((tgt_sh | tgt_sn) & 1) == 0 &&
                    (reinterpret_cast<uintptr_t>(targets) & 15) == 0;
  unsigned w_next = (wave < Hout && lane < W) ? bitmask[((int64_t)wave * N + n) * W + lane] : 0u;
  for (int h = wave; h < Hout; h += 4) {
    unsigned w = w_next;
    if (h + 4 < Hout && lane < W) w_next = bitmask[((int64_t)(h + 4) * N + n) * W + lane];
    const int cnt = __popc(w);
    const int incl = wave_incl_scan_add(cnt);
    const int total = __builtin_amdgcn_readlane(incl, PDT_WAVE - 1);
    int pos = incl - cnt;
    while (w) {
      const int b = __builtin_ctz(w);
      w &= w - 1u;
      stage[pos++] = ctok[lane * 32 + b];
    }
    wave_sync();
    int64_t *dst = targets + (int64_t)h * tgt_sh + n * tgt_sn;
    if (wide) {  // 16-byte stores: two targets per lane
      for (int i = lane; i < (C >> 1); i += PDT_WAVE) {
        longlong2 v;
        v.x = 2 * i < total ? stage[2 * i] : padding;
        v.y = 2 * i + 1 < total ? stage[2 * i + 1] : padding;
        *reinterpret_cast<longlong2 *>(dst + 2 * i) = v;
      }
    } else {
      for (int i = lane; i < C; i += PDT_WAVE) dst[i] = i < total ? stage[i] : padding;
    }
    wave_sync();
  }
}

int launch_oc_expand(const uint32_t *bitmask, const int64_t *class_tokens, int R, int Hout,
                     int64_t N, int C, int64_t padding, int64_t *targets, int64_t tgt_sh,
                     int64_t tgt_sn, hipStream_t stream) {
  const int W = (int)pdt_oc_mask_words(R);
  if (W > PDT_WAVE) return PDT_E_TOO_LONG;
  const size_t smem = (((size_t)R + (size_t)4 * W * 32) * 8 + 15) & ~(size_t)15;
  if (smem > 160 * 1024) return PDT_E_TOO_LONG;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(oc_expand_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(oc_expand_kernel, dim3((unsigned)N), dim3(256), smem, stream, bitmask,
                     class_tokens, R, W, Hout, N, C, padding, targets, tgt_sh, tgt_sn);
  return (int)hipGetLastError();
}

}  // namespace pdt
