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
#include "lev_classes.hpp"
#include "switches.hpp"

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
  {  // classes of this lane's columns: all loads in flight, then CPL searches side by side
    int64_t rt[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j)
      rt[j] = cbase + j <= ref_len ? a.ref[(int64_t)(cbase + j - 1) * a.ref_st + n * a.ref_sn] : 0;
    classes_of<CPL>(ctok, U, search_depth(U), rt, rid);
  }
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = cbase + j;
    if (c > ref_len) rid[j] = -2;
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
  // where column j's class bit lives and which lanes have one at all: the same for every row, so
  // the mask phase of a row is a compare, a scalar AND and one LDS OR per column
  // (up to 8 columns per lane: beyond, the hoisted values would not fit the register files)
  constexpr bool kHoistMask = CPL <= 8;
  constexpr int HC = kHoistMask ? CPL : 1;
  unsigned bm_bit[HC];
  int bm_word[HC];
  u64 has_next[HC];
  if (kHoistMask) {
#pragma unroll
    for (int j = 0; j < HC; ++j) {
      bm_bit[j] = 1u << (rnext[j] & 31);
      bm_word[j] = max(rnext[j], 0) >> 5;
      has_next[j] = __ballot(rnext[j] >= 0);
    }
  }
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
        // :334 and the r < ref_len cut of :349-354
        if (kHoistMask) {
          if (__builtin_amdgcn_inverse_ballot_w64(__ballot(prev[j] == m) & has_next[j < HC ? j : 0]))
            atomicOr(&bm[bm_word[j < HC ? j : 0]], bm_bit[j < HC ? j : 0]);
        } else if (prev[j] == m && rnext[j] >= 0) {
          atomicOr(&bm[rnext[j] >> 5], 1u << (rnext[j] & 31));
        }
      }
      if (lane == 0 && col0 == m && ref_len > 0) atomicOr(&bm[rank0 >> 5], 1u << (rank0 & 31));
      // (no wait here or below: the LDS serves one wave's instructions in order -- the read sees
      // the ds_or of every lane, the next row's ds_or see the zeros; only the compiler is told)
      __builtin_amdgcn_wave_barrier();
      int cnt = 0;
      if (lane < W) {
        const unsigned w = bm[lane];
        bm[lane] = 0u;
        a.bitmask[((int64_t)h * a.N + n) * W + lane] = w;
        cnt = __popc(w);
      }
      cnt = W <= 16 ? row0_sum(cnt) : wave_sum(cnt);  // (only lanes < W hold words)
      max_cnt = cnt > max_cnt ? cnt : max_cnt;
      __builtin_amdgcn_wave_barrier();
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
    ref_len = first_eos(a.ref, a.R, a.ref_st, roff, a.eos);
    hyp_len = first_eos(a.hyp, a.H, a.hyp_st, hoff, a.eos);
    if (a.include_eos) {
      if (ref_len == a.R) rmiss = true; else ref_len += 1;
      if (hyp_len == a.H) hmiss = true; else hyp_len += 1;
    }
  }
  int Heff = a.exclude_last ? hyp_len - 1 : hyp_len;
  if (Heff < 0) Heff = 0;

  // ---- distinct reference tokens in ascending order -> ctok[0..U) ------------------------
  int U = 0;
  if (!BIG) {  // R <= 512: sorted in registers (lev_classes.hpp)
    int64_t xt[8];
    U = distinct_sorted<8>(a.ref, ref_len, a.ref_st, roff, xt, ctok);
  } else {  // bitonic sort in LDS
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
  wave_sync();  // ctok complete; a sort buffer is dead from here on (hyp_l etc. alias it)
  if (a.class_tokens)
    for (int k = lane; k < U; k += PDT_WAVE) a.class_tokens[n * (int64_t)a.R + k] = ctok[k];

  // hypothesis tokens -> class ranks (-1: token does not occur in ref), eight look-ups side by side
  {
    const int lg = search_depth(U), hl = hyp_len < a.H ? hyp_len : a.H;
    for (int t0 = 0; t0 < hl; t0 += 8 * PDT_WAVE) {
      int64_t ht[8];
      int c[8];
      load_tokens(a.hyp, hl, a.hyp_st, hoff, t0, 0, ht);
      classes_of<8>(ctok, U, lg, ht, c);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (t0 + lane + q * PDT_WAVE < hl) hyp_l[t0 + lane + q * PDT_WAVE] = c[q];
    }
  }
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
  const bool big = a.R > 64 * 8;
  if (big && off < (size_t)P * 8) off = (size_t)P * 8;  // (R <= 512 sorts in registers)
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

// ---- phase 2, tiled form ---------------------------------------------------------------------
// Two things hold the row-at-a-time form at 3.6 TB/s (measured, profiles/microbench `stores`):
// a row is C * 8 bytes (992 at the bench shape), so almost every store instruction starts and
// ends inside a 128-byte line (62-lane stores of such rows: 3.4 TB/s whatever their order, against
// 5.5-6.2 for whole KiB); and vmcnt counts loads and stores together, in order, so a wave that
// loads the next bitmask words after storing a row waits for that store to reach memory.
// Here (a) the rows of NB consecutive utterances at one h -- or, batch-first, of NB consecutive h
// of one utterance -- are ONE contiguous run of NB * C * 8 bytes: a wave expands them into an LDS
// image of the run and streams it out with 16-byte stores, 1 KiB of consecutive bytes per
// instruction; (b) a workgroup loads every bitmask word and class-token table it will need into
// LDS up front, so its main loop issues no global load at all and never waits for a store.
//   over_n = 1: tile = utterances n0 .. n0 + NB at one h; the workgroup owns `chunk` values of h
//               and its four waves take them in turn;
//   over_n = 0: tile = rows h0 .. h0 + NB of one utterance (one table); the workgroup owns
//               `chunk` tiles.
// A pass expands 64 / Wp rows at once (Wp = bitmask words per row rounded up to a power of two:
// lane = (row, word)); positions inside a row come from one wave scan minus the scan value at
// the row's first lane.  The image holds int32 indices into the token tables (-1 = padding), so
// the per-bit loop only writes LDS and the tokens are looked up on the way out by all 64 lanes.
struct OcTileArgs {
  const uint32_t *bitmask;
  const int64_t *class_tokens;
  int64_t *targets;
  int64_t N, padding, outer_stride;  // outer_stride: elements between tiles' outer index
  int R, W, lgWp, Hout, C, NB, over_n, chunk, ntiles, wide;
  int nw;  // waves per workgroup
};

__host__ __device__ inline size_t oc_tile_lds(int R, int W, int C, int NB, int over_n, int chunk, int nw,
                                              size_t *bm_off, size_t *stage_off) {
  const size_t ctok = (((size_t)(over_n ? NB : 1) * R + 1) & ~(size_t)1) * 8;
  const size_t bm = (((size_t)chunk * NB * W + 3) & ~(size_t)3) * 4;  // rows of the chunk x W words
  const size_t stage = (((size_t)NB * C + 3) & ~(size_t)3) * 4;
  if (bm_off) *bm_off = ctok;
  if (stage_off) *stage_off = ctok + bm;
  return ctok + bm + (size_t)nw * stage;
}

template <int NW>
__global__ void __launch_bounds__(NW * PDT_WAVE) oc_expand_tiles_kernel(const OcTileArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int C = a.C, NB = a.NB, W = a.W, R = a.R;
  size_t bm_off, stage_off;
  constexpr int kOcWaves = NW, kOcThreads = NW * PDT_WAVE;
  oc_tile_lds(R, W, C, NB, a.over_n, a.chunk, a.nw, &bm_off, &stage_off);
  int64_t *ctok = reinterpret_cast<int64_t *>(smem);
  unsigned *bm = reinterpret_cast<unsigned *>(smem + bm_off);
  int *stage = reinterpret_cast<int *>(smem + stage_off) + (size_t)wave * (((size_t)NB * C + 3) & ~(size_t)3);
  const unsigned item = xcd_remap(blockIdx.x, gridDim.x);
  // over_n: item = (h chunk, n tile); else item = (utterance, chunk of h tiles)
  const int inner_items = a.over_n ? a.ntiles : (a.ntiles + a.chunk - 1) / a.chunk;
  const int64_t outer = item / inner_items;
  const int inner = (int)(item - outer * inner_items);
  const int64_t n_first = a.over_n ? (int64_t)inner * NB : outer;
  const int tabs_valid = a.over_n ? (int)min((int64_t)NB, a.N - n_first) : 1;
  // the h rows this workgroup covers, and (over_n) the utterances of its tile
  const int h_base = a.over_n ? (int)outer * a.chunk : inner * a.chunk * NB;
  const int h_count = min(a.Hout - h_base, a.over_n ? a.chunk : a.chunk * NB);
  // (eight loads in flight per thread: left to itself the compiler waits for every load before
  // the LDS write that follows it, and a workgroup's preamble becomes 32 round trips to L2)
  {  // the tables of consecutive utterances are one contiguous block of class_tokens
    const int64_t *src = a.class_tokens + n_first * (int64_t)R;
    const int total = tabs_valid * R;
    for (int k0 = (int)threadIdx.x; k0 < total; k0 += kOcThreads * 8) {
      int64_t v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = k0 + q * kOcThreads < total ? src[k0 + q * kOcThreads] : 0;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (k0 + q * kOcThreads < total) ctok[k0 + q * kOcThreads] = v[q];
    }
  }
  {  // bm[(hi * tabs_valid + u) * W + word]: tabs_valid * W consecutive words per h
    const int per_h = tabs_valid * W, total = h_count * per_h;
    for (int k0 = (int)threadIdx.x; k0 < total; k0 += kOcThreads * 8) {
      unsigned v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = k0 + q * kOcThreads;
        const int hi = k / per_h, r = k - hi * per_h;
        v[q] = k < total ? a.bitmask[((int64_t)(h_base + hi) * a.N + n_first) * W + r] : 0u;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (k0 + q * kOcThreads < total) bm[k0 + q * kOcThreads] = v[q];
    }
  }
  __syncthreads();  // from here on: LDS reads and global stores only
  const int Wp = 1 << a.lgWp, rows_per_pass = PDT_WAVE >> a.lgWp;
  const int ur = lane >> a.lgWp, word = lane & (Wp - 1);
  const int seg = lane & ~(Wp - 1);
  // the jobs of this wave: over_n -> h = h_base + wave, + 4, ...; else tiles of NB rows
  const int j_end = a.over_n ? h_base + h_count : min(a.ntiles, (inner + 1) * a.chunk);
  for (int j = (a.over_n ? h_base : inner * a.chunk) + wave; j < j_end; j += kOcWaves) {
    const int h_first = a.over_n ? j : j * NB;
    const int rows = a.over_n ? tabs_valid : min(NB, a.Hout - h_first);
    // image of the run as indices into the class-token tables: -1 (padding) everywhere, then
    // the classes of every row
    {
      const int4 neg = make_int4(-1, -1, -1, -1);
      int4 *s4 = reinterpret_cast<int4 *>(stage);
      for (int i = lane; i < (rows * C + 3) >> 2; i += PDT_WAVE) s4[i] = neg;
    }
    wave_sync();
    for (int u0 = 0; u0 < rows; u0 += rows_per_pass) {
      const int u = u0 + ur;
      const bool live = u < rows && word < W;
      // row (h, n) of the chunk: over_n -> (h_first, u), else (h_first + u, the utterance)
      const int row = a.over_n ? (h_first - h_base) * tabs_valid + u : h_first - h_base + u;
      unsigned w = live ? bm[row * W + word] : 0u;
      const int cnt = __popc(w);
      const int incl = wave_incl_scan_add(cnt);
      const int before = __builtin_amdgcn_ds_bpermute((seg > 0 ? seg - 1 : 0) << 2, incl);
      int pos = incl - cnt - (seg > 0 ? before : 0);
      const int tab = (a.over_n ? u : 0) * R + word * 32;
      int *srow = stage + u * C;
      while (w) {
        const int b = __builtin_ctz(w);
        w &= w - 1u;
        srow[pos++] = tab + b;
      }
    }
    wave_sync();
    int64_t *dst = a.targets + (a.over_n ? (int64_t)h_first * a.outer_stride + n_first * C
                                         : n_first * a.outer_stride + (int64_t)h_first * C);
    const int total = rows * C;
    auto tok_of = [&](int id) { return id < 0 ? a.padding : ctok[id]; };
    if (a.wide) {  // 16-byte stores, four in flight per lane (total is even here or the odd last
                   // element goes out alone)
      const int2 *s2 = reinterpret_cast<const int2 *>(stage);
      longlong2 *d2 = reinterpret_cast<longlong2 *>(dst);
      const int pairs = total >> 1;
      int i = lane;
      for (; i + 3 * PDT_WAVE < pairs; i += 4 * PDT_WAVE) {
        int2 id[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) id[q] = s2[i + q * PDT_WAVE];
        longlong2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q].x = tok_of(id[q].x);
          v[q].y = tok_of(id[q].y);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // (non-temporal: 1.95 GB written once and never read by this kernel -- 0.74 -> 0.71 ms for the operator)
          typedef long long ll2 __attribute__((ext_vector_type(2)));
          ll2 t;
          t.x = v[q].x;
          t.y = v[q].y;
          __builtin_nontemporal_store(t, reinterpret_cast<ll2 *>(&d2[i + q * PDT_WAVE]));
        }
      }
      for (; i < pairs; i += PDT_WAVE) {
        const int2 id = s2[i];
        longlong2 v;
        v.x = tok_of(id.x);
        v.y = tok_of(id.y);
        d2[i] = v;
      }
      if ((total & 1) && lane == 0) dst[total - 1] = tok_of(stage[total - 1]);
    } else {
      for (int i = lane; i < total; i += PDT_WAVE) dst[i] = tok_of(stage[i]);
    }
    // (the next tile's fill may not overtake these reads of the image: the LDS serves a wave's
    // instructions in order, so only the compiler has to be told)
    __builtin_amdgcn_wave_barrier();
  }
}

int launch_oc_expand(const uint32_t *bitmask, const int64_t *class_tokens, int R, int Hout,
                     int64_t N, int C, int64_t padding, int64_t *targets, int64_t tgt_sh,
                     int64_t tgt_sn, hipStream_t stream) {
  const int W = (int)pdt_oc_mask_words(R);
  if (W > PDT_WAVE) return PDT_E_TOO_LONG;
  // tiled form: rows that follow each other in memory (stride C along n or along h)
  const bool over_n = tgt_sn == C, over_h = tgt_sh == C;
  if ((over_n || over_h) && N < (1ll << 31)) {
    OcTileArgs a{};
    a.bitmask = bitmask; a.class_tokens = class_tokens; a.targets = targets;
    a.N = N; a.padding = padding; a.R = R; a.W = W; a.Hout = Hout; a.C = C;
    a.over_n = over_n ? 1 : 0;
    a.outer_stride = over_n ? tgt_sh : tgt_sn;
    while ((1 << a.lgWp) < W) ++a.lgWp;
    // rows per tile: 4 (a run of 4 * C * 8 bytes: 31 whole lines at the bench shape; measured
    // 0.55 ms against 0.61 with 8 rows, whose tables leave room for two workgroups per CU only),
    // fewer while the tables + bitmask words + four images exceed 64 KiB; jobs per workgroup:
    // 64 values of h per table load (over_n) / 16 tiles of one utterance
    const size_t cap = 64 * 1024;
#ifndef PDT_OC_TILE_ROWS
#define PDT_OC_TILE_ROWS 4
#endif
#ifndef PDT_OC_CHUNK
#define PDT_OC_CHUNK 64
#endif
    int NB = PDT_OC_TILE_ROWS;
    int chunk = over_n ? PDT_OC_CHUNK : 16;
    while (oc_tile_lds(R, W, C, NB, a.over_n, chunk, 4, nullptr, nullptr) > cap && (NB > 1 || chunk > 8)) {
      if (chunk > 16 || NB == 1) chunk >>= 1; else NB >>= 1;
    }
    // waves per workgroup (they share the tables): eight when that puts more waves on a CU than
    // four do (bench shape: 3 x 8 against 4 x 4, 1.05-1.07 -> 0.98-1.00 ms for the whole op,
    // batch-first 1.04 -> 0.95; with the wide images of V = 5000 a workgroup of eight would be
    // alone on its CU)
    int nw = 4;
    {
      const size_t l4 = oc_tile_lds(R, W, C, NB, a.over_n, chunk, 4, nullptr, nullptr);
      const size_t l8 = oc_tile_lds(R, W, C, NB, a.over_n, chunk, 8, nullptr, nullptr);
      const size_t lds_cu = 160 * 1024;
      if (min(lds_cu / l8, (size_t)4) * 8 > min(lds_cu / l4, (size_t)8) * 4) nw = 8;
      const int force = switches().oc_waves;
      if (force == 4 || force == 8) nw = force;
    }
    a.nw = nw;
    const size_t smem = oc_tile_lds(R, W, C, NB, a.over_n, chunk, nw, nullptr, nullptr);
    if (smem <= 160 * 1024) {
      a.NB = NB;
      a.chunk = chunk;
      const int64_t inner_len = over_n ? N : Hout;
      a.ntiles = (int)((inner_len + NB - 1) / NB);
      const int64_t grid = over_n ? (int64_t)a.ntiles * ((Hout + a.chunk - 1) / a.chunk)
                                  : N * ((a.ntiles + a.chunk - 1) / a.chunk);
      a.wide = ((a.outer_stride & 1) == 0 && (((int64_t)NB * C) & 1) == 0 &&
                (reinterpret_cast<uintptr_t>(targets) & 15) == 0) ? 1 : 0;
      if (grid > 0 && grid < (1ll << 31)) {
        auto kern = nw == 8 ? oc_expand_tiles_kernel<8> : oc_expand_tiles_kernel<4>;
        if (smem > cap) {
          hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
          if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(nw * PDT_WAVE), smem, stream, a);
        return (int)hipGetLastError();
      }
    }
  }
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
