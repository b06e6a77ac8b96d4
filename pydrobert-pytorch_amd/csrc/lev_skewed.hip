// Anti-diagonal ("skewed") wavefront Levenshtein for gfx950.
//
// Replaces the row loop of _string_matching (reference _string.py:286-346) for the FINAL
// and PREFIX output modes, in both arithmetic flavours:
//   COUNT = false  cost recurrence        (edit_distance, prefix_edit_distances, and
//                                          error_rate with uniform costs, _string.py:316-317)
//   COUNT = true   (cost, #mistakes) pairs with the sub < ins < del tie-break
//                                          (error_rate with non-uniform costs, :294-314)
//
// Mapping: ONE WAVE PER UTTERANCE.  Lane l owns CPL consecutive DP columns, right-aligned
// so that lane 63's last register is column ref_len (the column every output reads).
// At step s lane l updates row h = s - l + 1: the wave sweeps an anti-diagonal band, the
// only cross-lane traffic being one v_mov_b32_dpp wave_shr:1 per tracked quantity per step
// (left neighbour's freshly computed last column).  The previous row lives in registers;
// LDS holds the utterance's token ids and one (H+1)-entry column buffer that serves both as
// the boundary between 64*CPL-column chunks (R > 512) and as the prefix output.
// Every cell is computed with exactly the reference's float32 operations in the reference's
// order, so COUNT mode is bit-exact for any costs; cost mode is bit-exact whenever all
// partial sums are exactly representable (the host routes other costs to lev_rowsync.hip).
#include "lev_common.hpp"

namespace pdt {

constexpr int kMaxCPL = 8;

// UNIT (cost mode with ins = del = sub = 1, what every uniform-cost call becomes after the host's
// rescaling): D[h][c] = D[h-1][c-1] when the tokens match, else 1 + min of the three
// neighbours -- four VALU per cell instead of six, same values (all small integers).
template <int CPL, bool COUNT, bool FROM_LDS, bool UNIT = false>
__device__ __forceinline__ void skew_sweep(const int *ref_l, const int *hyp_l, float *bnd_c,
                                           float *bnd_m, const int c_first, const int Heff,
                                           const float ins, const float del, const float sub) {
  const int lane = lane_id();
  const int cbase = c_first + lane * CPL;  // DP column held in register 0
  int rtok[CPL];
  float pc[CPL], pm[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = cbase + j;
    rtok[j] = c >= 1 ? ref_l[c - 1] : 0;
    pc[j] = c < 0 ? PDT_INF : (float)c * del;  // row 0: _string.py:258-263
    pm[j] = (float)c;
  }
  // D[0][cbase - 1]: the diagonal input of this lane's first column at its first row
  float dprev_c = (cbase - 1) < 0 ? PDT_INF : (float)(cbase - 1) * del;
  float dprev_m = (float)(cbase - 1);
  // lane 0's left input on the first chunk: column c_first-1 is either column 0
  // (row[0] = last_row[0] + ins, _string.py:292) or a virtual column (+inf)
  float chain_c = (c_first == 1) ? ins : PDT_INF;
  const float chain_step = (c_first == 1) ? ins : 0.0f;
  float chain_m = 1.0f;
  float last_c = pc[CPL - 1], last_m = pm[CPL - 1];
  const int nsteps = Heff > 0 ? Heff + PDT_WAVE - 1 : 0;
  for (int s = 0; s < nsteps; ++s) {
    float l0c, l0m = 0.0f;
    if (FROM_LDS) {
      const int hh = s + 1 < Heff ? s + 1 : Heff;
      l0c = bnd_c[hh];
      if (COUNT) l0m = bnd_m[hh];
    } else {
      l0c = chain_c;
      chain_c += chain_step;
      l0m = chain_m;
      chain_m += 1.0f;
    }
    float left_c = shr1(last_c, l0c);
    float left_m = COUNT ? shr1(last_m, l0m) : 0.0f;
    const int h = s - lane + 1;
    if ((unsigned)(h - 1) < (unsigned)Heff) {
      const int tok = hyp_l[h - 1];
      float dc = dprev_c, dm = dprev_m;
      dprev_c = left_c;
      dprev_m = left_m;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const bool neq = rtok[j] != tok;  // _string.py:291
        const float up_c = pc[j], up_m = pm[j];
        float c_, m_ = 0.0f;
        if (UNIT) {
          // columns < 1 are virtual (+inf, or D[h][0] = h in column 0): never a "match"
          const float m3 = fminf(fminf(dc, up_c), left_c) + 1.0f;
          c_ = (neq || cbase + j < 1) ? m3 : dc;
        } else if (COUNT) {
          c_ = up_c + ins;  // :292
          m_ = up_m + 1.0f; // :299
          const float sc = dc + (neq ? sub : 0.0f);  // :293
          const float sm = dm + (neq ? 1.0f : 0.0f); // :300
          const bool pick_sub = c_ >= sc;            // :296
          c_ = pick_sub ? sc : c_;
          m_ = pick_sub ? sm : m_;
          const float dl = left_c + del;  // :308
          const bool keep = dl >= c_;     // :309
          m_ = keep ? m_ : left_m + 1.0f;
          c_ = keep ? c_ : dl;
        } else {
          const float a = up_c + ins;
          const float b = dc + (neq ? sub : 0.0f);
          c_ = fminf(fminf(a, b), left_c + del);  // :316-317 in recurrence form
        }
        dc = up_c;
        dm = up_m;
        left_c = c_;
        left_m = m_;
        pc[j] = c_;
        pm[j] = m_;
      }
      last_c = left_c;
      last_m = left_m;
      if (lane == PDT_WAVE - 1) {
        bnd_c[h] = last_c;
        if (COUNT) bnd_m[h] = last_m;
      }
    }
  }
}

template <int CPL, bool COUNT>
__device__ __forceinline__ void skew_first(const int *ref_l, const int *hyp_l, float *bnd_c,
                                           float *bnd_m, int c_first, int Heff, float ins,
                                           float del, float sub) {
  if (!COUNT && ins == 1.0f && del == 1.0f && sub == 1.0f)
    skew_sweep<CPL, false, false, true>(ref_l, hyp_l, bnd_c, bnd_m, c_first, Heff, ins, del, sub);
  else
    skew_sweep<CPL, COUNT, false>(ref_l, hyp_l, bnd_c, bnd_m, c_first, Heff, ins, del, sub);
}

template <bool COUNT>
__global__ void __launch_bounds__(256) lev_skewed_kernel(const LevArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int64_t n = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * a.waves_per_wg + wave;
  if (n >= a.N) return;  // waves never synchronise with each other
  int *ref_l = reinterpret_cast<int *>(smem + (size_t)wave * a.lds_per_wave);
  int *hyp_l = ref_l + a.R;
  float *bnd_c = reinterpret_cast<float *>(hyp_l + a.H);
  float *bnd_m = bnd_c + (a.H + 1);

  bool rfits, hfits, rmiss, hmiss;
  const int ref_len = stage_tokens(a.ref, a.R, a.ref_st, n * a.ref_sn, a.has_eos, a.eos,
                                   a.include_eos, ref_l, rfits, rmiss);
  const int hyp_len = stage_tokens(a.hyp, a.H, a.hyp_st, n * a.hyp_sn, a.has_eos, a.eos,
                                   a.include_eos, hyp_l, hfits, hmiss);
  if (!(rfits && hfits)) {
    wave_sync();
    remap_tokens_by_first_occurrence(a, n, ref_len, hyp_len, ref_l, hyp_l);
  }
  wave_sync();

  // rows that are actually updated (_string.py:286-288)
  int Heff = a.exclude_last ? hyp_len - 1 : hyp_len;
  if (Heff < 0) Heff = 0;
  const float ins = a.ins, del = a.del, sub = a.sub;

  const int chunk_cols = PDT_WAVE * kMaxCPL;
  const int nch = ref_len > chunk_cols ? (ref_len + chunk_cols - 1) / chunk_cols : 1;
  if (nch == 1) {
    const int cpl = ref_len > 0 ? (ref_len + PDT_WAVE - 1) / PDT_WAVE : 1;
    const int c_first = ref_len - PDT_WAVE * cpl + 1;
    switch (cpl) {
      case 1: skew_first<1, COUNT>(ref_l, hyp_l, bnd_c, bnd_m, c_first, Heff, ins, del, sub); break;
      case 2: skew_first<2, COUNT>(ref_l, hyp_l, bnd_c, bnd_m, c_first, Heff, ins, del, sub); break;
      case 3: skew_first<3, COUNT>(ref_l, hyp_l, bnd_c, bnd_m, c_first, Heff, ins, del, sub); break;
      case 4: skew_first<4, COUNT>(ref_l, hyp_l, bnd_c, bnd_m, c_first, Heff, ins, del, sub); break;
      case 5: skew_first<5, COUNT>(ref_l, hyp_l, bnd_c, bnd_m, c_first, Heff, ins, del, sub); break;
      case 6: skew_first<6, COUNT>(ref_l, hyp_l, bnd_c, bnd_m, c_first, Heff, ins, del, sub); break;
      case 7: skew_first<7, COUNT>(ref_l, hyp_l, bnd_c, bnd_m, c_first, Heff, ins, del, sub); break;
      default: skew_first<8, COUNT>(ref_l, hyp_l, bnd_c, bnd_m, c_first, Heff, ins, del, sub); break;
    }
  } else {
    for (int i = 0; i < nch; ++i) {
      const int c_first = ref_len - (nch - i) * chunk_cols + 1;
      if (i == 0)
        skew_sweep<kMaxCPL, COUNT, false>(ref_l, hyp_l, bnd_c, bnd_m, c_first, Heff, ins, del, sub);
      else
        skew_sweep<kMaxCPL, COUNT, true>(ref_l, hyp_l, bnd_c, bnd_m, c_first, Heff, ins, del, sub);
      wave_sync();
    }
  }
  wave_sync();

  // ---- epilogue: bnd_*[h] = D[h][ref_len] for 1 <= h <= Heff -------------------------
  const float *res = COUNT ? bnd_m : bnd_c;
  const float row0 = COUNT ? (float)ref_len : (float)ref_len * del;  // :285, :258-263
  int flags = 0;
  if (rmiss) flags |= PDT_WARN_REF_NO_EOS;
  if (hmiss) flags |= PDT_WARN_HYP_NO_EOS;
  if (a.norm && ref_len == 0) flags |= PDT_WARN_EMPTY_REF;
  if (a.mode == PDT_MODE_FINAL) {
    if (lane == 0) {
      const float v = Heff > 0 ? res[Heff] : row0;
      a.out[n * a.out_sn] =
          lev_finish(v, a.mult, a.norm, ref_len, hyp_len > 0 ? 1.0f : 0.0f);  // :394-405
    }
  } else {
    const int Hout = a.H + (a.exclude_last ? 0 : 1);
    const int pad_from = hyp_len + (a.exclude_last ? 0 : 1);  // :379-386
    for (int h = lane; h < Hout; h += PDT_WAVE) {
      float v;
      if (h >= pad_from) {
        v = a.padding;
      } else {
        v = h == 0 ? row0 : res[h];
        v = lev_finish(v, a.mult, a.norm, ref_len, h > 0 ? 1.0f : 0.0f);  // :357-378
      }
      a.out[(int64_t)h * a.out_sh + n * a.out_sn] = v;
    }
  }
  if (lane == 0) {
    if (a.ref_lens_out) a.ref_lens_out[n] = ref_len;
    if (a.hyp_lens_out) a.hyp_lens_out[n] = hyp_len;
    if (flags && a.status) atomicOr(a.status, flags);
  }
}

// host side -------------------------------------------------------------------------------
int launch_lev_skewed(LevArgs a, hipStream_t stream) {
  const size_t per_wave =
      (((size_t)a.R + a.H + (size_t)(a.H + 1) * (a.count ? 2 : 1)) * 4 + 15) & ~(size_t)15;
  const size_t soft_cap = 64 * 1024, hard_cap = 160 * 1024;
  if (per_wave > hard_cap) return PDT_E_TOO_LONG;
  int wpw = (int)(soft_cap / per_wave);
  if (wpw > 4) wpw = 4;
  if (wpw < 1) wpw = 1;
  a.waves_per_wg = wpw;
  a.lds_per_wave = (int)per_wave;
  const size_t smem = per_wave * wpw;
  const unsigned grid = (unsigned)((a.N + wpw - 1) / wpw);
  auto kern = a.count ? lev_skewed_kernel<true> : lev_skewed_kernel<false>;
  if (smem > soft_cap) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpw), smem, stream, a);
  return (int)hipGetLastError();
}

}  // namespace pdt
