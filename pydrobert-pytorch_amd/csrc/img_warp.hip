// SpecAugment application, polyharmonic splines and image warps for gfx950.
//
// Replaces, from the reference's _img.py:
//   polyharmonic_spline (:59-150)            -> spline_solve_kernel + spline_apply_kernel
//   warp_1d_grid (:268-303)                  -> warp_1d_grid_kernel (3-knot spline, 5x5 solve)
//   spec_augment_apply_parameters (:1142-1211)-> spec_augment_apply_kernel: ONE pass that reads
//       feats once and writes the result once; the reference materialises an (N,T,F,2) grid
//       (64 % of its time is the torch.stack building it) and calls grid_sample + masked_fill
//   dense_image_warp (:393-439), sparse_image_warp (:520-714) -> image_warp_kernel: per pixel,
//       sampling position from a flow field or straight from the spline (knots in LDS), then the
//       grid_sample gather (bilinear / nearest; zeros / border / reflection) for every channel.
// The small dense systems are solved in float64 (partial pivoting); gathers follow
// torch.nn.functional.grid_sample(align_corners=False) arithmetic in float32.
#include <cfloat>
#include <cmath>
#include <algorithm>

#include <type_traits>

#include "pdt_common.hpp"
#include "switches.hpp"

namespace pdt {

// T + I + 1 up to which the augmented system of one batch element fits the 160 KB of LDS (with
// O <= 4 right-hand sides); larger systems are eliminated in a global-memory workspace
constexpr size_t kSplineLdsCap = 160 * 1024 - 64;

__device__ __forceinline__ double phi_d(double r, int order) {
  // _img.py:59-64; eps = float32 epsilon (train/query points are cast to float, :142-143)
  double rk = 1.0;
  for (int i = 0; i < order; ++i) rk *= r;
  if (order & 1) return rk;
  return rk * log(fmax(r, (double)FLT_EPSILON));
}
__device__ __forceinline__ float phi_f(float r, int order) {
  float rk = 1.0f;
  for (int i = 0; i < order; ++i) rk *= r;
  if (order & 1) return rk;
  return rk * logf(fmaxf(r, FLT_EPSILON));
}

// phi(r) from the SQUARED distance, float32, for the per-pixel spline evaluation of the sparse
// warp (7-100 centres per pixel: this is where its time goes).  Even orders need no square
// root (r^k log r = d2^(k/2) * log(d2) / 2) and the logarithm is the hardware v_log_f32
// (1 ulp) instead of OCML's logf; order 2 -- the default -- costs ~8 VALU per centre
// instead of ~35.
// ORDER = 1, 2, 3: that order, compiled without branches; ORDER = 0: any order (runtime).
// ln(x) for finite x >= eps^2 (1.4e-14: no denormals, no infinities): the arithmetic of __logf
// without its guards -- v_log_f32 (log2) and the compensated product with ln 2 in two pieces --
// five instructions instead of twelve, the same bits for these arguments.
__device__ __forceinline__ float ln_fast(float x) {
  const float r = __builtin_amdgcn_logf(x);
  const float hi = __uint_as_float(0x3f317217u), lo = __uint_as_float(0x3377d1cfu);  // ln 2 = hi + lo
  const float t = r * hi;
  float e = __builtin_fmaf(r, hi, -t);
  e = __builtin_fmaf(r, lo, e);
  return t + e;
}

template <int ORDER>
__device__ __forceinline__ float phi_from_d2(float d2, int order) {
#ifndef PDT_WARP_REFERENCE_PHI
  // d2 + 1e-37 is d2 itself for every distance that is not 0 (and 0 * ln(1e-37) = 0 there); the
  // half and ln 2 folded into one factor
  if (ORDER == 2) return d2 * (__builtin_amdgcn_logf(d2 + 1e-37f) * 0.34657359f);
#endif
  if (ORDER == 2) return d2 * (0.5f * ln_fast(fmaxf(d2, FLT_EPSILON * FLT_EPSILON)));
  if (ORDER == 1) return sqrtf(d2);
  if (ORDER == 3) return d2 * sqrtf(d2);
  float pw = 1.0f;  // d2^(order / 2)
  for (int i = 0; i < (order >> 1); ++i) pw *= d2;
  if (order & 1) return pw * sqrtf(d2);
  return pw * (0.5f * ln_fast(fmaxf(d2, FLT_EPSILON * FLT_EPSILON)));
}

// Solve the bordered system [[A + reg*I, B], [B^T, 0]] [w; v] = [f; 0] (_img.py:79-130) for one
// batch element with the whole workgroup.  a: (S, S + O) augmented matrix in LDS (doubles).
__device__ void solve_in_lds(double *a, int S, int O, int *piv_row) {
  const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
  const int ld = S + O;
  for (int p = 0; p < S; ++p) {
    if (tid == 0) {  // partial pivoting
      int best = p;
      double bv = fabs(a[p * ld + p]);
      for (int r = p + 1; r < S; ++r) {
        const double v = fabs(a[r * ld + p]);
        if (v > bv) {
          bv = v;
          best = r;
        }
      }
      *piv_row = best;
    }
    __syncthreads();
    const int pr = *piv_row;
    if (pr != p)
      for (int c = tid; c < ld; c += nt) {
        const double t = a[p * ld + c];
        a[p * ld + c] = a[pr * ld + c];
        a[pr * ld + c] = t;
      }
    __syncthreads();
    const double inv = 1.0 / a[p * ld + p];
    // eliminate column p from every other row (Gauss-Jordan): rows x columns over the threads
    const int ncol = ld - p - 1;
    for (int i = tid; i < S * ncol; i += nt) {
      const int r = i / ncol, c = p + 1 + (i - r * ncol);
      if (r != p) a[r * ld + c] -= a[r * ld + p] * inv * a[p * ld + c];
    }
    __syncthreads();
    for (int r = tid; r < S; r += nt)
      if (r != p) a[r * ld + p] = 0.0;
    __syncthreads();
  }
  for (int i = tid; i < S * O; i += nt) {
    const int r = i / O, c = S + (i - r * O);
    a[r * ld + c] /= a[r * ld + r];
  }
  __syncthreads();
}

// train points c (N,T,I), values f (N,T,O) -> wv (N, T+I+1, O) doubles.  `tail` (N, I+1, O) or
// null: the last I + 1 rows of the right-hand side (zeros for the interpolation problem itself;
// the adjoint system of the backward pass has them).  `gmat` non-null: the augmented matrix of
// batch element n lives at gmat + n * S * (S + O) in global memory instead of LDS (systems too
// large for LDS; a workgroup's own global writes are visible to it after __syncthreads()).
__global__ void __launch_bounds__(256)
spline_solve_kernel(const float *__restrict__ c, const float *__restrict__ f,
                    const float *__restrict__ tail, int T, int I, int O, int order, float reg,
                    double *__restrict__ wv, double *gmat) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int S = T + I + 1, ld = S + O;
  const int64_t n = blockIdx.x;
  double *a = gmat ? gmat + n * (int64_t)S * ld : reinterpret_cast<double *>(smem);
  int *piv = gmat ? reinterpret_cast<int *>(smem) : reinterpret_cast<int *>(a + (size_t)S * ld);
  const float *cn = c + n * (int64_t)T * I;
  const float *fn = f + n * (int64_t)T * O;
  for (int i = (int)threadIdx.x; i < S * ld; i += (int)blockDim.x) {
    const int r = i / ld, col = i - r * ld;
    double v = 0.0;
    if (r < T && col < T) {
      double d2 = 0.0;
      for (int k = 0; k < I; ++k) {
        const double d = (double)cn[r * I + k] - (double)cn[col * I + k];
        d2 += d * d;
      }
      v = phi_d(sqrt(d2), order);
      if (r == col && reg > 0.0f) v += (double)reg;
    } else if (r < T && col < S) {  // B
      v = (col - T) < I ? (double)cn[r * I + (col - T)] : 1.0;
    } else if (r >= T && col < T) {  // B^T
      v = (r - T) < I ? (double)cn[col * I + (r - T)] : 1.0;
    } else if (r < T && col >= S) {
      v = (double)fn[r * O + (col - S)];
    } else if (r >= T && col >= S && tail) {
      v = (double)tail[(n * (I + 1) + (r - T)) * O + (col - S)];
    }
    a[i] = v;
  }
  __syncthreads();
  solve_in_lds(a, S, O, piv);
  for (int i = (int)threadIdx.x; i < S * O; i += (int)blockDim.x) {
    const int r = i / O, o = i - r * O;
    wv[(n * S + r) * O + o] = a[r * ld + S + o];
  }
}

// out[n,q,o] = sum_t phi(|x_q - c_t|) w[t,o] + x_q . v[:I,o] + v[I,o]     (_img.py:67-76)
__global__ void __launch_bounds__(256)
spline_apply_kernel(const float *__restrict__ c, const double *__restrict__ wv,
                    const float *__restrict__ x, int T, int I, int O, int Q, int order,
                    float *__restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem[];
  double *lw = reinterpret_cast<double *>(smem);            // (T + I + 1, O)
  float *lc = reinterpret_cast<float *>(lw + (size_t)(T + I + 1) * O);  // (T, I)
  const int64_t n = blockIdx.y;
  for (int i = (int)threadIdx.x; i < (T + I + 1) * O; i += (int)blockDim.x)
    lw[i] = wv[n * (int64_t)(T + I + 1) * O + i];
  for (int i = (int)threadIdx.x; i < T * I; i += (int)blockDim.x) lc[i] = c[n * (int64_t)T * I + i];
  __syncthreads();
  const int q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (q >= Q) return;
  const float *xq = x + (n * (int64_t)Q + q) * I;
  for (int o = 0; o < O; ++o) {
    double acc = lw[(T + I) * O + o];
    for (int k = 0; k < I; ++k) acc += (double)xq[k] * lw[(T + k) * O + o];
    for (int t = 0; t < T; ++t) {
      double d2 = 0.0;
      for (int k = 0; k < I; ++k) {
        const double d = (double)xq[k] - (double)lc[t * I + k];
        d2 += d * d;
      }
      acc += phi_d(sqrt(d2), order) * lw[t * O + o];
    }
    out[(n * (int64_t)Q + q) * O + o] = (float)acc;
  }
}

// warp_1d_grid (_img.py:268-303): one workgroup per batch element
__global__ void __launch_bounds__(256)
warp_1d_grid_kernel(const float *__restrict__ src, const float *__restrict__ flow,
                    const float *__restrict__ lengths, int T, int order,
                    float *__restrict__ grid) {
  __shared__ double a[5 * 6];
  __shared__ int piv;
  __shared__ double knots[3];
  const int64_t n = blockIdx.x;
  const double len = (double)lengths[n];
  if (threadIdx.x == 0) {
    const double eps = (double)FLT_EPSILON;
    double s = fmax(fmin((double)src[n], len - 1.0), 0.0);
    double d = fmax(fmin(s + (double)flow[n], len - 1.0), 0.0);
    s = (2.0 * s + 1.0) / T - 1.0;
    d = (2.0 * d + 1.0) / T - 1.0;
    const double lo = 1.0 / T - 1.0 - eps, up = (2.0 * len - 1.0) / T - 1.0 + eps;
    const double cp[3] = {lo, d, up}, fv[3] = {lo, s, up};  // spline FROM dst TO src
    for (int r = 0; r < 5; ++r)
      for (int c = 0; c < 6; ++c) {
        double v = 0.0;
        if (r < 3 && c < 3) v = phi_d(fabs(cp[r] - cp[c]), order);
        else if (r < 3 && c == 3) v = cp[r];
        else if (r < 3 && c == 4) v = 1.0;
        else if (r == 3 && c < 3) v = cp[c];
        else if (r == 4 && c < 3) v = 1.0;
        else if (r < 3 && c == 5) v = fv[r];
        a[r * 6 + c] = v;
      }
    for (int k = 0; k < 3; ++k) knots[k] = cp[k];
  }
  __syncthreads();
  solve_in_lds(a, 5, 1, &piv);
  const double w0 = a[0 * 6 + 5], w1 = a[1 * 6 + 5], w2 = a[2 * 6 + 5];
  const double v0 = a[3 * 6 + 5], v1 = a[4 * 6 + 5];
  for (int j = (int)threadIdx.x; j < T; j += (int)blockDim.x) {
    const double t = (2.0 * j + 1.0) / T - 1.0;
    const double g = w0 * phi_d(fabs(t - knots[0]), order) + w1 * phi_d(fabs(t - knots[1]), order) +
                     w2 * phi_d(fabs(t - knots[2]), order) + v0 * t + v1;
    grid[n * (int64_t)T + j] = (float)g;
  }
}

// ---- grid_sample arithmetic (align_corners = False) -----------------------------------------
enum { PAD_ZEROS = 0, PAD_BORDER = 1, PAD_REFLECTION = 2 };
enum { INTERP_BILINEAR = 0, INTERP_NEAREST = 1 };

// (CT: the coordinate type -- float, or double for float64 images, whose grid the reference forms
// and samples in float64, _img.py:420-436)
template <typename CT>
__device__ __forceinline__ CT unnormalize(CT g, int size) {
  return ((g + CT(1)) * (CT)size - CT(1)) * CT(0.5);
}
template <typename CT>
__device__ __forceinline__ CT clip_coord(CT x, int size) {
  return fmin((CT)(size - 1), fmax(x, CT(0)));
}
__device__ __forceinline__ float clip_coord(float x, int size) {
  return fminf((float)(size - 1), fmaxf(x, 0.0f));
}
template <typename CT>
__device__ __forceinline__ CT reflect_coord(CT x, int twice_low, int twice_high) {
  if (twice_low == twice_high) return CT(0);
  const CT mn = (CT)twice_low * CT(0.5), span = (CT)(twice_high - twice_low) * CT(0.5);
  x = fabs(x - mn);
  const CT extra = fmod(x, span);
  const int flips = (int)floor(x / span);
  return (flips & 1) ? span - extra + mn : extra + mn;
}
__device__ __forceinline__ float reflect_coord(float x, int twice_low, int twice_high) {
  if (twice_low == twice_high) return 0.0f;
  const float mn = (float)twice_low * 0.5f, span = (float)(twice_high - twice_low) * 0.5f;
  x = fabsf(x - mn);
  const float extra = fmodf(x, span);
  const int flips = (int)floorf(x / span);
  return (flips & 1) ? span - extra + mn : extra + mn;
}
template <typename CT>
__device__ __forceinline__ CT source_index(CT g, int size, int padding) {
  CT x = unnormalize(g, size);
  if (padding == PAD_BORDER) x = clip_coord(x, size);
  else if (padding == PAD_REFLECTION) x = clip_coord(reflect_coord(x, -1, 2 * size - 1), size);
  return x;
}

struct SpecAugArgs {
  const float *feats; int64_t f_sn, f_st, f_sf;
  const float *tgrid, *fgrid;          // (N,T) / (N,F) normalised grids or null
  const int64_t *t0, *tl, *f0, *fl;    // (N,MT) / (N,MF) masks or null
  int N, T, F, MT, MF;
  float *out;                          // (N,T,F) contiguous
  // the time warp by its PARAMETERS instead of a grid (spec_augment_rows_kernel only): w_0, w (N,)
  // float and the lengths (N,) int64 (null: all T) -- the three-knot spline of warp_1d_grid is solved
  // in closed form and evaluated where the rows are planned, no (N, T) grid in memory
  const float *tw_src, *tw_flow; const int64_t *tw_len; int tw_order;
  // (with tw_len) set to 1 by an utterance whose length is not in [1, T]: the reference's input check
  // (_img.py:1037-1041) made where the lengths are read; device-visible memory the caller zeroed, or null
  int32_t *bad_lengths;
};

// warp_1d_grid's spline (_img.py:283-302) in closed form.  Knots c0 < c1 < c2 = {lo, dst, up}, values
// {lo, src, up}; the bordered 5 x 5 system [[A, [c 1]], [[c 1]^T, 0]] [w; v] = [f; 0] with A_ij =
// phi(|c_i - c_j|): the two constraints leave w = alpha u, u = (c1 - c2, c2 - c0, c0 - c1); u kills
// the affine part, so alpha = u.f / u^T A u, and v from rows 0 and 2 of f - alpha A u = v0 c + v1.
struct Warp1D {
  double c[3], w[3], v0, v1;
};
__device__ inline Warp1D warp_1d_spline(double src, double flow, double len, int T, int order) {
  const double eps = (double)FLT_EPSILON;
  double s = fmax(fmin(src, len - 1.0), 0.0);
  double d = fmax(fmin(s + flow, len - 1.0), 0.0);
  s = (2.0 * s + 1.0) / T - 1.0;
  d = (2.0 * d + 1.0) / T - 1.0;
  const double lo = 1.0 / T - 1.0 - eps, up = (2.0 * len - 1.0) / T - 1.0 + eps;
  Warp1D r;
  r.c[0] = lo; r.c[1] = d; r.c[2] = up;
  const double f0 = lo, f1 = s, f2 = up;
  const double u0 = d - up, u1 = up - lo, u2 = lo - d;
  const double a = phi_d(fabs(d - lo), order), b = phi_d(fabs(up - lo), order), e = phi_d(fabs(up - d), order);
  const double Au0 = a * u1 + b * u2, Au1 = a * u0 + e * u2, Au2 = b * u0 + e * u1;
  const double den = u0 * Au0 + u1 * Au1 + u2 * Au2;
  const double alpha = den != 0.0 ? (u0 * f0 + u1 * f1 + u2 * f2) / den : 0.0;
  r.w[0] = alpha * u0; r.w[1] = alpha * u1; r.w[2] = alpha * u2;
  const double r0 = f0 - alpha * Au0, r2 = f2 - alpha * Au2;
  r.v0 = (r2 - r0) / (up - lo);
  r.v1 = r0 - r.v0 * lo;
  return r;
}
__device__ __forceinline__ float warp_1d_eval(const Warp1D &sp, int j, int T, int order) {
  const double t = (2.0 * j + 1.0) / T - 1.0;
  return (float)(sp.w[0] * phi_d(fabs(t - sp.c[0]), order) + sp.w[1] * phi_d(fabs(t - sp.c[1]), order) +
                 sp.w[2] * phi_d(fabs(t - sp.c[2]), order) + sp.v0 * t + sp.v1);
}

// One pass: bilinear gather along time and frequency + band masks.  Workgroup = 256 threads
// walking a contiguous range of (t, f) positions of one utterance.
__global__ void __launch_bounds__(256) spec_augment_apply_kernel(const SpecAugArgs a, int tiles) {
  const int64_t n = blockIdx.x / tiles;
  const int tile = (int)(blockIdx.x % tiles);
  const int T = a.T, F = a.F;
  const int rows_per_tile = (T + tiles - 1) / tiles;
  const int t_begin = tile * rows_per_tile, t_end = min(T, t_begin + rows_per_tile);
  const float *fn = a.feats + n * a.f_sn;
  for (int idx = t_begin * F + (int)threadIdx.x; idx < t_end * F; idx += 256) {
    const int t = idx / F, f = idx - t * F;
    bool masked = false;
    for (int m = 0; m < a.MT; ++m) {
      const int64_t s = a.t0[n * a.MT + m];
      masked = masked || (t >= s && t < s + a.tl[n * a.MT + m]);
    }
    for (int m = 0; m < a.MF; ++m) {
      const int64_t s = a.f0[n * a.MF + m];
      masked = masked || (f >= s && f < s + a.fl[n * a.MF + m]);
    }
    float v = 0.0f;
    if (!masked) {
      if (!a.tgrid && !a.fgrid) {
        v = fn[(int64_t)t * a.f_st + (int64_t)f * a.f_sf];
      } else {
        // identity grids when only one axis is warped (:1173-1180)
        const float gy = a.tgrid ? a.tgrid[n * T + t] : (2.0f * (float)t + 1.0f) / (float)T - 1.0f;
        const float gx = a.fgrid ? a.fgrid[n * F + f] : (2.0f * (float)f + 1.0f) / (float)F - 1.0f;
        const float iy = clip_coord(unnormalize(gy, T), T), ix = clip_coord(unnormalize(gx, F), F);
        const float y0f = floorf(iy), x0f = floorf(ix);
        const int y0 = (int)y0f, x0 = (int)x0f, y1 = y0 + 1, x1 = x0 + 1;
        const float wy1 = iy - y0f, wx1 = ix - x0f, wy0 = (y0f + 1.0f) - iy, wx0 = (x0f + 1.0f) - ix;
        const float *r0 = fn + (int64_t)y0 * a.f_st;
        const float *r1 = fn + (int64_t)min(y1, T - 1) * a.f_st;
        const int64_t c0 = (int64_t)x0 * a.f_sf, c1 = (int64_t)min(x1, F - 1) * a.f_sf;
        // taps outside the image carry weight 0 under border padding
        v = r0[c0] * (wx0 * wy0);
        if (x1 < F) v += r0[c1] * (wx1 * wy0);
        if (y1 < T) v += r1[c0] * (wx0 * wy1);
        if (x1 < F && y1 < T) v += r1[c1] * (wx1 * wy1);
      }
    }
    a.out[(n * T + t) * (int64_t)F + f] = v;
  }
}

// Fast path of the above for the common SpecAugment setting (no frequency warp, F % 4 == 0,
// unit stride along F): an output row is a 2-tap blend of two source rows, so a thread moves
// a float4 -- 16-byte coalesced loads and stores.  Per-row (source row, weights, time mask) and
// per-column (frequency mask) decisions are made once per workgroup and kept in LDS.
constexpr int kRowsPerTile = 256;
__global__ void __launch_bounds__(256) spec_augment_rows_kernel(const SpecAugArgs a, int tiles) {
  __shared__ int row_y0[kRowsPerTile];      // source row, or -1 when the row is masked
  __shared__ float row_w1[kRowsPerTile];    // weight of source row y0 + 1
  __shared__ unsigned col_keep[64];         // per float4 column: 4 keep bits
  const int64_t n = blockIdx.x / tiles;
  const int tile = (int)(blockIdx.x % tiles);
  const int T = a.T, F4 = a.F >> 2;
  const int t_begin = tile * kRowsPerTile, t_end = min(T, t_begin + kRowsPerTile);
  const int tid = (int)threadIdx.x;
  const bool warped = a.tgrid != nullptr || a.tw_src != nullptr;
  __shared__ Warp1D spline;
  if (a.tw_src) {  // (uniform: one extra barrier per workgroup)
    if (tid == 0) {
      const int64_t len = a.tw_len ? a.tw_len[n] : (int64_t)T;
      if (a.bad_lengths && tile == 0 && (len > T || len <= 0)) *a.bad_lengths = 1;
      spline = warp_1d_spline((double)a.tw_src[n], (double)a.tw_flow[n], (double)len, T, a.tw_order);
    }
    __syncthreads();
  }
  if (t_begin + tid < t_end) {
    const int t = t_begin + tid;
    bool masked = false;
    for (int m = 0; m < a.MT; ++m) {
      const int64_t s = a.t0[n * a.MT + m];
      masked = masked || (t >= s && t < s + a.tl[n * a.MT + m]);
    }
    int y0 = t;
    float w1 = 0.0f;
    if (warped) {
      const float g = a.tw_src ? warp_1d_eval(spline, t, T, a.tw_order) : a.tgrid[n * T + t];
      const float iy = clip_coord(unnormalize(g, T), T);
      const float y0f = floorf(iy);
      y0 = (int)y0f;
      w1 = iy - y0f;
    }
    row_y0[tid] = masked ? -1 : y0;
    row_w1[tid] = w1;
  }
  for (int c = tid; c < F4; c += 256) {
    unsigned keep = 0u;
    for (int j = 0; j < 4; ++j) {
      const int f = 4 * c + j;
      bool fm = false;
      for (int m = 0; m < a.MF; ++m) {
        const int64_t s = a.f0[n * a.MF + m];
        fm = fm || (f >= s && f < s + a.fl[n * a.MF + m]);
      }
      keep |= fm ? 0u : (1u << j);
    }
    col_keep[c] = keep;
  }
  __syncthreads();
  const float *fn = a.feats + n * a.f_sn;
  float *on = a.out + n * (int64_t)T * a.F;
  const float inv = 1.0f / (float)F4;
  const int total = (t_end - t_begin) * F4;
  for (int idx = tid; idx < total; idx += 256) {
    int r = (int)(((float)idx + 0.5f) * inv);
    int f4 = idx - r * F4;
    if (f4 < 0) { --r; f4 += F4; }
    if (f4 >= F4) { ++r; f4 -= F4; }
    const int y0 = row_y0[r];
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (y0 >= 0) {
      const float w1 = row_w1[r];
      const float4 r0 = *reinterpret_cast<const float4 *>(fn + (int64_t)y0 * a.f_st + 4 * f4);
      if (warped && y0 + 1 < T) {
        // same arithmetic as the 4-tap form with wx0 = 1, wx1 = 0; a row at y0 + 1 == T lies
        // outside the image and carries weight 0 under border padding
        const float w0 = ((float)y0 + 1.0f) - ((float)y0 + w1);
        const float4 r1 = *reinterpret_cast<const float4 *>(fn + (int64_t)(y0 + 1) * a.f_st + 4 * f4);
        v.x = r0.x * w0 + r1.x * w1;
        v.y = r0.y * w0 + r1.y * w1;
        v.z = r0.z * w0 + r1.z * w1;
        v.w = r0.w * w0 + r1.w * w1;
      } else if (warped) {
        const float w0 = ((float)y0 + 1.0f) - ((float)y0 + w1);
        v.x = r0.x * w0; v.y = r0.y * w0; v.z = r0.z * w0; v.w = r0.w * w0;
      } else {
        v = r0;
      }
      const unsigned keep = col_keep[f4];
      v.x = (keep & 1u) ? v.x : 0.0f;
      v.y = (keep & 2u) ? v.y : 0.0f;
      v.z = (keep & 4u) ? v.z : 0.0f;
      v.w = (keep & 8u) ? v.w : 0.0f;
    }
    // (non-temporal: written once, never read here -- apply 0.30 -> 0.29 ms on the box that measured both)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4 *>(on + (int64_t)(t_begin + r) * a.F + 4 * f4));
  }
}

// Adjoint of the two kernels above with respect to the features: every unmasked output element
// scatters its gradient to its (up to) four taps.  grad_feats is zeroed by the caller; the
// accumulation uses the hardware float atomic (order of addition is not fixed, like
// grid_sample's own backward).
__global__ void __launch_bounds__(256)
spec_augment_backward_kernel(const SpecAugArgs a, const float *__restrict__ grad_out,
                             float *__restrict__ grad_feats, int tiles) {
  const int64_t n = blockIdx.x / tiles;
  const int tile = (int)(blockIdx.x % tiles);
  const int T = a.T, F = a.F;
  const int rows_per_tile = (T + tiles - 1) / tiles;
  const int t_begin = tile * rows_per_tile, t_end = min(T, t_begin + rows_per_tile);
  float *gn = grad_feats + n * (int64_t)T * F;
  for (int idx = t_begin * F + (int)threadIdx.x; idx < t_end * F; idx += 256) {
    const int t = idx / F, f = idx - t * F;
    bool masked = false;
    for (int m = 0; m < a.MT; ++m) {
      const int64_t s = a.t0[n * a.MT + m];
      masked = masked || (t >= s && t < s + a.tl[n * a.MT + m]);
    }
    for (int m = 0; m < a.MF; ++m) {
      const int64_t s = a.f0[n * a.MF + m];
      masked = masked || (f >= s && f < s + a.fl[n * a.MF + m]);
    }
    if (masked) continue;
    const float g = grad_out[(n * T + t) * (int64_t)F + f];
    if (!a.tgrid && !a.fgrid) {
      gn[(int64_t)t * F + f] = g;  // one-to-one: no other writer
      continue;
    }
    const float gy = a.tgrid ? a.tgrid[n * T + t] : (2.0f * (float)t + 1.0f) / (float)T - 1.0f;
    const float gx = a.fgrid ? a.fgrid[n * F + f] : (2.0f * (float)f + 1.0f) / (float)F - 1.0f;
    const float iy = clip_coord(unnormalize(gy, T), T), ix = clip_coord(unnormalize(gx, F), F);
    const float y0f = floorf(iy), x0f = floorf(ix);
    const int y0 = (int)y0f, x0 = (int)x0f, y1 = y0 + 1, x1 = x0 + 1;
    const float wy1 = iy - y0f, wx1 = ix - x0f, wy0 = (y0f + 1.0f) - iy, wx0 = (x0f + 1.0f) - ix;
    float *r0 = gn + (int64_t)y0 * F, *r1 = gn + (int64_t)min(y1, T - 1) * F;
    unsafeAtomicAdd(r0 + x0, g * (wx0 * wy0));
    if (x1 < F && wx1 != 0.0f) unsafeAtomicAdd(r0 + x1, g * (wx1 * wy0));
    if (y1 < T && wy1 != 0.0f) unsafeAtomicAdd(r1 + x0, g * (wx0 * wy1));
    if (x1 < F && y1 < T && wx1 != 0.0f && wy1 != 0.0f) unsafeAtomicAdd(r1 + x1, g * (wx1 * wy1));
  }
}

// Adjoint of spec_augment_rows_kernel written as a GATHER (no atomics, no zero fill,
// deterministic): a workgroup owns 256 source rows of one utterance; a source row y collects
// w0(t) * g[t] from the output rows t sampled at y0(t) = y and w1(t) * g[t] from those at
// y0(t) = y - 1.  Warp grids are non-decreasing (warp_1d_grid pins both ends), so those rows
// form one contiguous range found by binary search over the per-row table in LDS; for any
// other grid the range degrades to all rows (still exact, just slower).
__global__ void __launch_bounds__(256)
spec_augment_rows_backward_kernel(const SpecAugArgs a, const float *__restrict__ grad_out,
                                  float *__restrict__ grad_feats, int tiles) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int T = a.T, F4 = a.F >> 2;
  int *sy0 = reinterpret_cast<int *>(smem);             // [T] source row of output row t
  float *sw0 = reinterpret_cast<float *>(sy0 + T);      // [T] weight of row y0 (0 if masked)
  float *sw1 = sw0 + T;                                 // [T] weight of row y0 + 1
  int *lb = reinterpret_cast<int *>(sw1 + T);           // [258] first t with y0(t) >= y_b - 1 + i
  unsigned *col_keep = reinterpret_cast<unsigned *>(lb + 260);  // [64]
  const int64_t n = blockIdx.x / tiles;
  const int tile = (int)(blockIdx.x % tiles);
  const int y_b = tile * kRowsPerTile, y_e = min(T, y_b + kRowsPerTile);
  const int tid = (int)threadIdx.x;
  bool bad = false;
  for (int t = tid; t < T; t += 256) {
    bool masked = false;
    for (int m = 0; m < a.MT; ++m) {
      const int64_t s = a.t0[n * a.MT + m];
      masked = masked || (t >= s && t < s + a.tl[n * a.MT + m]);
    }
    int y0 = t;
    float w0 = 1.0f, w1 = 0.0f;
    if (a.tgrid) {
      const float iy = clip_coord(unnormalize(a.tgrid[n * T + t], T), T);
      const float y0f = floorf(iy);
      y0 = (int)y0f;
      w1 = iy - y0f;
      w0 = ((float)y0 + 1.0f) - ((float)y0 + w1);  // as the forward kernel
      if (y0 + 1 >= T) w1 = 0.0f;                  // tap outside the image
      if (t > 0) {
        const float ip = clip_coord(unnormalize(a.tgrid[n * T + t - 1], T), T);
        bad = bad || ((int)floorf(ip) > y0);
      }
    }
    sy0[t] = y0;
    sw0[t] = masked ? 0.0f : w0;
    sw1[t] = masked ? 0.0f : w1;
  }
  for (int c = tid; c < F4; c += 256) {
    unsigned keep = 0u;
    for (int j = 0; j < 4; ++j) {
      const int f = 4 * c + j;
      bool fm = false;
      for (int m = 0; m < a.MF; ++m) {
        const int64_t s = a.f0[n * a.MF + m];
        fm = fm || (f >= s && f < s + a.fl[n * a.MF + m]);
      }
      keep |= fm ? 0u : (1u << j);
    }
    col_keep[c] = keep;
  }
  const bool monotone = !__syncthreads_or(bad);
  for (int i = tid; i < kRowsPerTile + 2; i += 256) {
    int lo = 0, hi = T;
    if (monotone) {
      const int target = y_b - 1 + i;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sy0[mid] < target) lo = mid + 1; else hi = mid;
      }
    } else {
      lo = i == 0 ? 0 : T;  // every range becomes [0, T)
    }
    lb[i] = lo;
  }
  __syncthreads();
  const float *gn = grad_out + n * (int64_t)T * a.F;
  float *on = grad_feats + n * (int64_t)T * a.F;
  const float inv = 1.0f / (float)F4;
  const int total = (y_e - y_b) * F4;
  for (int idx = tid; idx < total; idx += 256) {
    int r = (int)(((float)idx + 0.5f) * inv);
    int f4 = idx - r * F4;
    if (f4 < 0) { --r; f4 += F4; }
    if (f4 >= F4) { ++r; f4 -= F4; }
    const int y = y_b + r;
    const int t_lo = monotone ? lb[r] : 0, t_hi = monotone ? lb[r + 2] : T;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int t = t_lo; t < t_hi; ++t) {
      const int y0 = sy0[t];
      const float w = y0 == y ? sw0[t] : (y0 + 1 == y ? sw1[t] : 0.0f);
      if (w != 0.0f) {
        const float4 g = *reinterpret_cast<const float4 *>(gn + (int64_t)t * a.F + 4 * f4);
        acc.x += g.x * w; acc.y += g.y * w; acc.z += g.z * w; acc.w += g.w * w;
      }
    }
    const unsigned keep = col_keep[f4];
    acc.x = (keep & 1u) ? acc.x : 0.0f;
    acc.y = (keep & 2u) ? acc.y : 0.0f;
    acc.z = (keep & 4u) ? acc.z : 0.0f;
    acc.w = (keep & 8u) ? acc.w : 0.0f;
    *reinterpret_cast<float4 *>(on + (int64_t)y * a.F + 4 * f4) = acc;
  }
}

// PT: the pixel type (float; double for float64 images -- image_warp_kernel only, see there)
template <typename PT>
struct WarpArgsT {
  const PT *image;     // (N,C,H,W) contiguous
  PT *out;             // (N,C,H,W)
  int N, C, H, W;
  int mode, padding;
  // source of the sampling position, one of:
  const float *flow;   // dense: (N,H,W,2); position = pixel - flow (x = last dim 0 unless flip)
  int flip;            // dense: flow[..., 0] is the H component ("hw" indexing)
  const float *knots;  // sparse: (N,M,2) spline centres (x, y), float
  const float *wv;     // sparse: (N, M+3, 2) float weights (w, then v_x, v_y, v_1)
  int M, order, as_grid;  // as_grid: the spline yields the normalised grid itself (no-flow form)
  float inv_w;         // 1 / W (sparse_warp_bands_kernel)
  float *flow_out;     // sparse, optional (N,H,W,2)
  int flow_out_flip;
  PT *grad_image;      // BACKWARD: (N,C,H,W), zeroed by the caller; `out` then holds grad_out
};
using WarpArgs = WarpArgsT<float>;

// BACKWARD = adjoint with respect to the image: the same sampling positions, each pixel scatters
// its gradient to its taps with the hardware float atomic.
// PT = double: a float64 image.  The reference keeps flows and spline points in float32 whatever
// the image's type (_img.py:420, :537-538) but forms the sampling grid and samples it in the
// image's type (:423-436), so here the flow / spline value stays float and everything from the grid
// on (un-normalisation, padding, weights, blend, the adjoint's atomics) is double.
constexpr int kPixPerWG = 2048;
template <bool BACKWARD, typename PT = float>
__global__ void __launch_bounds__(256) image_warp_kernel(const WarpArgsT<PT> a) {
  using CT = PT;  // coordinate type
  extern __shared__ __align__(16) unsigned char smem[];
  float *lk = reinterpret_cast<float *>(smem);  // knots (M,2) then weights (M+3,2)
  float *lw = lk + 2 * a.M;
  const int64_t n = blockIdx.y;
  const int H = a.H, W = a.W;
  if (a.knots) {
    for (int i = (int)threadIdx.x; i < 2 * a.M; i += 256) lk[i] = a.knots[n * 2 * a.M + i];
    for (int i = (int)threadIdx.x; i < 2 * (a.M + 3); i += 256) lw[i] = a.wv[n * 2 * (a.M + 3) + i];
    __syncthreads();
  }
  const float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;  // wave-uniform: two divisions per wave
  // a workgroup covers kPixPerWG pixels: the LDS staging + barrier above is paid once per
  // 2048 pixels instead of once per 256 (it dominated at one pixel per thread)
  for (int pix = (int)(blockIdx.x * kPixPerWG + threadIdx.x);
       pix < min(H * W, (int)((blockIdx.x + 1) * kPixPerWG)); pix += 256) {
    // (h, w) of the pixel without an integer division where a float holds pix exactly:
    // reciprocal estimate + one-step fix-up
    int h, w;
    if (H * W < (1 << 23)) {
      h = (int)(((float)pix + 0.5f) * inv_w);
      w = pix - h * W;
      if (w < 0) { --h; w += W; }
      if (w >= W) { ++h; w -= W; }
    } else {
      h = pix / W;
      w = pix - h * W;
    }
    CT gx, gy;
    const CT inv_wc = std::is_same<CT, float>::value ? (CT)inv_w : CT(1) / (CT)W;
    const CT inv_hc = std::is_same<CT, float>::value ? (CT)inv_h : CT(1) / (CT)H;
    if (a.knots) {
      const float x = (float)w, y = (float)h;
      float sx = lw[2 * a.M + 0] * x + lw[2 * (a.M + 1) + 0] * y + lw[2 * (a.M + 2) + 0];
      float sy = lw[2 * a.M + 1] * x + lw[2 * (a.M + 1) + 1] * y + lw[2 * (a.M + 2) + 1];
      // the order is wave-uniform: pick the specialised loop once, outside the centre loop
      auto centres = [&](auto tag) {
        constexpr int ORDER = decltype(tag)::value;
#pragma unroll 4
        for (int m = 0; m < a.M; ++m) {
          const float dx = x - lk[2 * m], dy = y - lk[2 * m + 1];
          const float p = phi_from_d2<ORDER>(dx * dx + dy * dy, a.order);
          sx += p * lw[2 * m];
          sy += p * lw[2 * m + 1];
        }
      };
      if (a.order == 2) centres(std::integral_constant<int, 2>{});
      else if (a.order == 1) centres(std::integral_constant<int, 1>{});
      else if (a.order == 3) centres(std::integral_constant<int, 3>{});
      else centres(std::integral_constant<int, 0>{});
      if (a.as_grid) {
        gx = sx;
        gy = sy;
      } else {
        if (!BACKWARD && a.flow_out) {
          float *fo = a.flow_out + ((n * H + h) * (int64_t)W + w) * 2;
          fo[0] = a.flow_out_flip ? sy : sx;
          fo[1] = a.flow_out_flip ? sx : sy;
        }
        gx = (CT(2) * (CT)x - CT(2) * (CT)sx + CT(1)) * inv_wc - CT(1);  // _img.py:432
        gy = (CT(2) * (CT)y - CT(2) * (CT)sy + CT(1)) * inv_hc - CT(1);
      }
    } else {
      const float *fl = a.flow + ((n * H + h) * (int64_t)W + w) * 2;
      const float fx = a.flip ? fl[1] : fl[0], fy = a.flip ? fl[0] : fl[1];
      gx = (CT(2) * (CT)w - CT(2) * (CT)fx + CT(1)) * inv_wc - CT(1);
      gy = (CT(2) * (CT)h - CT(2) * (CT)fy + CT(1)) * inv_hc - CT(1);
    }
    const CT ix = source_index<CT>(gx, W, a.padding), iy = source_index<CT>(gy, H, a.padding);
    const int64_t plane = (int64_t)H * W;
    const PT *img = a.image + n * a.C * plane;
    PT *o = a.out + n * a.C * plane + pix;
    PT *gi = a.grad_image + n * a.C * plane;
    if (a.mode == INTERP_NEAREST) {
      const int xn = (int)nearbyint(ix), yn = (int)nearbyint(iy);
      const bool ok = xn >= 0 && xn < W && yn >= 0 && yn < H;
      if (BACKWARD) {
        if (ok)
          for (int c = 0; c < a.C; ++c) unsafeAtomicAdd(gi + c * plane + (int64_t)yn * W + xn, o[c * plane]);
        continue;
      }
      for (int c = 0; c < a.C; ++c) o[c * plane] = ok ? img[c * plane + (int64_t)yn * W + xn] : PT(0);
      continue;
    }
    const CT x0f = floor(ix), y0f = floor(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const CT wx1 = ix - x0f, wy1 = iy - y0f, wx0 = (x0f + CT(1)) - ix, wy0 = (y0f + CT(1)) - iy;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
    const bool vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    if (BACKWARD) {
      for (int c = 0; c < a.C; ++c) {
        PT *pl = gi + c * plane;
        const PT g = o[c * plane];
        if (vx0 && vy0) unsafeAtomicAdd(pl + (int64_t)y0 * W + x0, g * (wx0 * wy0));
        if (vx1 && vy0) unsafeAtomicAdd(pl + (int64_t)y0 * W + x1, g * (wx1 * wy0));
        if (vx0 && vy1) unsafeAtomicAdd(pl + (int64_t)y1 * W + x0, g * (wx0 * wy1));
        if (vx1 && vy1) unsafeAtomicAdd(pl + (int64_t)y1 * W + x1, g * (wx1 * wy1));
      }
      continue;
    }
    for (int c = 0; c < a.C; ++c) {
      const PT *pl = img + c * plane;
      PT v = PT(0);
      if (vx0 && vy0) v += pl[(int64_t)y0 * W + x0] * (wx0 * wy0);
      if (vx1 && vy0) v += pl[(int64_t)y0 * W + x1] * (wx1 * wy0);
      if (vx0 && vy1) v += pl[(int64_t)y1 * W + x0] * (wx0 * wy1);
      if (vx1 && vy1) v += pl[(int64_t)y1 * W + x1] * (wx1 * wy1);
      o[c * plane] = v;
    }
}
}

// Sparse warp, the common shape (bilinear, forward, no flow output, at most kWarpFastM spline
// centres -- SpecAugment-style calls have 3 + 4 pinned): image_warp_kernel's per-pixel work with
//   * the centres and weights in SCALAR registers (wave-uniform; image_warp_kernel re-reads them
//     from LDS for every pixel: 34 broadcast reads),
//   * kWarpPix pixels per lane, 256 apart (a wave's lanes stay on consecutive pixels: coalesced taps
//     and stores), evaluated side by side -- independent chains that hide v_log_f32 and the taps'
//     latency behind each other,
//   * the spline sum in fused multiply-adds (one rounding per term instead of two; the results
//     differ from image_warp_kernel's in the last bits, far inside the 1e-4 the spline is good to).
constexpr int kWarpFastM = 8;
constexpr int kWarpPix = 4;
template <int ORDER, int PADDING>
__global__ void __launch_bounds__(256) sparse_warp_fast_kernel(const WarpArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  float *lk = reinterpret_cast<float *>(smem);
  float *lw = lk + 2 * a.M;
  const int64_t n = blockIdx.y;
  const int H = a.H, W = a.W, M = a.M;
  for (int i = (int)threadIdx.x; i < 2 * M; i += 256) lk[i] = a.knots[n * 2 * M + i];
  for (int i = (int)threadIdx.x; i < 2 * (M + 3); i += 256) lw[i] = a.wv[n * 2 * (M + 3) + i];
  __syncthreads();
#ifdef PDT_WARP_VGPR
  auto uni = [](float v) { asm volatile("" : "+v"(v)); return v; };
#else
  auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
#endif
  float kx[kWarpFastM], ky[kWarpFastM], wx[kWarpFastM], wy[kWarpFastM];
#pragma unroll
  for (int m = 0; m < kWarpFastM; ++m) {
    const bool in = m < M;
    kx[m] = uni(in ? lk[2 * m] : 0.0f);
    ky[m] = uni(in ? lk[2 * m + 1] : 0.0f);
    wx[m] = uni(in ? lw[2 * m] : 0.0f);
    wy[m] = uni(in ? lw[2 * m + 1] : 0.0f);
  }
  const float ax = uni(lw[2 * M]), ay = uni(lw[2 * M + 1]), bx = uni(lw[2 * (M + 1)]), by = uni(lw[2 * (M + 1) + 1]);
  const float cx = uni(lw[2 * (M + 2)]), cy = uni(lw[2 * (M + 2) + 1]);
  const float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;
  const int HW = H * W;
  const int base = (int)(blockIdx.x * (256 * kWarpPix) + threadIdx.x);
  float x[kWarpPix], y[kWarpPix], sx[kWarpPix], sy[kWarpPix];
#pragma unroll
  for (int j = 0; j < kWarpPix; ++j) {
    const int pix = min(base + j * 256, HW - 1);  // (beyond the image: a duplicate of the last pixel, not stored)
    int h = (int)(((float)pix + 0.5f) * inv_w), w = pix - h * W;  // (H * W < 2^23: checked by the launcher)
    if (w < 0) { --h; w += W; }
    if (w >= W) { ++h; w -= W; }
    x[j] = (float)w;
    y[j] = (float)h;
    sx[j] = __builtin_fmaf(ax, x[j], __builtin_fmaf(bx, y[j], cx));
    sy[j] = __builtin_fmaf(ay, x[j], __builtin_fmaf(by, y[j], cy));
  }
#pragma unroll
  for (int m = 0; m < kWarpFastM; ++m) {
    if (m < M) {  // (wave-uniform)
#pragma unroll
      for (int j = 0; j < kWarpPix; ++j) {
        const float dx = x[j] - kx[m], dy = y[j] - ky[m];
        const float p = phi_from_d2<ORDER>(__builtin_fmaf(dx, dx, dy * dy), a.order);
        sx[j] = __builtin_fmaf(p, wx[m], sx[j]);
        sy[j] = __builtin_fmaf(p, wy[m], sy[j]);
      }
    }
  }
  float ix[kWarpPix], iy[kWarpPix];
#pragma unroll
  for (int j = 0; j < kWarpPix; ++j) {
    float gx, gy;
    if (a.as_grid) {
      gx = sx[j];
      gy = sy[j];
    } else {
      gx = (2.0f * x[j] - 2.0f * sx[j] + 1.0f) * inv_w - 1.0f;  // _img.py:432
      gy = (2.0f * y[j] - 2.0f * sy[j] + 1.0f) * inv_h - 1.0f;
    }
    ix[j] = source_index(gx, W, PADDING);
    iy[j] = source_index(gy, H, PADDING);
  }
  float x0f[kWarpPix], y0f[kWarpPix];
#pragma unroll
  for (int j = 0; j < kWarpPix; ++j) {
    x0f[j] = floorf(ix[j]);
    y0f[j] = floorf(iy[j]);
  }
  for (int c = 0; c < a.C; ++c) {
    const float *pl = a.image + (n * a.C + c) * (int64_t)HW;
    float t00[kWarpPix], t01[kWarpPix], t10[kWarpPix], t11[kWarpPix];
#pragma unroll
    for (int j = 0; j < kWarpPix; ++j) {  // all the taps in flight (taps outside the image: any valid address)
      const int x0 = (int)x0f[j], y0 = (int)y0f[j];
      // (border / reflection padding: the coordinates are inside [0, size - 1] already)
      const int xc0 = PADDING == PAD_ZEROS ? min(max(x0, 0), W - 1) : x0, xc1 = min(max(x0 + 1, 0), W - 1);
      const int yc0 = (PADDING == PAD_ZEROS ? min(max(y0, 0), H - 1) : y0) * W, yc1 = min(max(y0 + 1, 0), H - 1) * W;
      // (unsigned 32-bit byte offsets from the plane's uniform base: no 64-bit address arithmetic per tap)
      const unsigned char *plb = reinterpret_cast<const unsigned char *>(pl);
      t00[j] = *reinterpret_cast<const float *>(plb + ((unsigned)(yc0 + xc0) << 2));
      t01[j] = *reinterpret_cast<const float *>(plb + ((unsigned)(yc0 + xc1) << 2));
      t10[j] = *reinterpret_cast<const float *>(plb + ((unsigned)(yc1 + xc0) << 2));
      t11[j] = *reinterpret_cast<const float *>(plb + ((unsigned)(yc1 + xc1) << 2));
    }
#pragma unroll
    for (int j = 0; j < kWarpPix; ++j) {
      const int x0 = (int)x0f[j], y0 = (int)y0f[j], x1 = x0 + 1, y1 = y0 + 1;
      const float wx1 = ix[j] - x0f[j], wy1 = iy[j] - y0f[j], wx0 = (x0f[j] + 1.0f) - ix[j], wy0 = (y0f[j] + 1.0f) - iy[j];
      const bool vx0 = PADDING != PAD_ZEROS || (x0 >= 0 && x0 < W), vx1 = x1 >= 0 && x1 < W;
      const bool vy0 = PADDING != PAD_ZEROS || (y0 >= 0 && y0 < H), vy1 = y1 >= 0 && y1 < H;
      // (a tap outside the image is left out, as image_warp_kernel does -- a select, not a product
      // with 0: that would turn an inf / NaN pixel into NaN)
      float acc = (vx0 && vy0) ? t00[j] * (wx0 * wy0) : 0.0f;
      acc += (vx1 && vy0) ? t01[j] * (wx1 * wy0) : 0.0f;
      acc += (vx0 && vy1) ? t10[j] * (wx0 * wy1) : 0.0f;
      acc += (vx1 && vy1) ? t11[j] * (wx1 * wy1) : 0.0f;
      const int pix = base + j * 256;
      if (pix < HW)
        *reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(a.out + (n * a.C + c) * (int64_t)HW) + ((unsigned)pix << 2)) = acc;
    }
  }
}

// The same call shape with a lane = one COLUMN of a band of ROWS rows: the pixels (h .. h + ROWS - 1, w)
// share x, so a centre's dx, dx^2 and the x part of the affine term are formed once per lane, the
// pixel -> (h, w) split once, and the chains are written as float2 chains (v_pk_add / v_pk_mul /
// v_pk_fma carry two pixels per instruction; v_log_f32 stays scalar).  A wave's lanes are consecutive
// columns (bands flattened with their columns: lane order = memory order within a row), so each
// row-round of taps and stores is coalesced.
//
// What the lanes do NOT do: everything that is the same for a whole image sits in a TABLE that
// warp_table_kernel writes once per call (per image: MC centres x (kx, ky, wx, wy), then ax, ay, bx,
// by, cx, cy) and the kernel reads with scalar loads -- no LDS staging, no barrier, no readfirstlane.
// The table's weights carry (a) grid_sample's un-normalisation ((g + 1) * size - 1) / 2 -- or, in
// the flow form, pixel - flow -- so the spline's value IS the source pixel coordinate, (b) order 2's
// ln 2 / 2, so phi is d2 * log2(d2) here; both products are formed in double before the cast.
// Centres beyond M have zero weights: fma(phi, 0, s) = s exactly, phi finite everywhere.
//
// Taps (border / reflection padding, coordinates inside [0, size - 1]): buffer loads with the plane's
// base in scalar registers and 32-bit byte offsets; the first tap's offset is one float fma + convert
// (exact below 2^23 pixels), the others add 4 / 4W -- or, where the neighbour lies outside the image,
// an offset beyond the buffer: the load returns 0 and the tap drops out as in image_warp_kernel
// (never a product of an inf / NaN pixel with a zero weight).
typedef float wf2 __attribute__((ext_vector_type(2)));
constexpr int kBandRows = 4;  // (8 measured slower: 0.42 ms against 0.40 at C4)
constexpr int kWarpTableFloats = 40;  // per image: 4 * 8 + 6, rounded up to 16 bytes (workspace stride)
__host__ __device__ constexpr int warp_table_stride(int MC) { return (4 * MC + 6 + 3) & ~3; }

__global__ void warp_table_kernel(const double *__restrict__ wv, const float *__restrict__ knots, float *__restrict__ tab,
                                  int64_t N, int M, int MC, int as_grid, int H, int W, int order) {
  const int stride = warp_table_stride(MC);
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N * stride) return;
  const int64_t n = i / stride;
  const int e = (int)(i - n * stride);
  const double *sol = wv + n * (int64_t)(M + 3) * 2;
  const int d = e & 1;                                     // 0: x (columns), 1: y (rows)
  const double size = d ? (double)H : (double)W;
  // source coordinate = scale * spline + (identity part) + shift
  const double scale = as_grid ? 0.5 * size : -1.0;
  const double phi_scale = order == 2 ? 0.34657359027997264 : 1.0;
  float v = 0.0f;
  if (e < 4 * MC) {
    const int m = e >> 2, k = e & 3;
    if (m < M) v = k < 2 ? knots[(n * M + m) * 2 + k] : (float)(sol[m * 2 + (k - 2)] * scale * phi_scale);
  } else if (e < 4 * MC + 6) {
    const int r = (e - 4 * MC) >> 1;                       // 0: coefficient of x, 1: of y, 2: constant
    double t = sol[(M + r) * 2 + d] * scale;
    if (as_grid) {
      if (r == 2) t += 0.5 * (size - 1.0);
    } else if (r == d) {
      t += 1.0;
    }
    v = (float)t;
  }
  tab[i] = v;
}

template <int ORDER>
__device__ __forceinline__ wf2 phi2_unscaled(const wf2 d2, const int order) {
  if (ORDER == 2) {  // d2 * log2(d2); d2 + 1e-37 is d2 for every distance but 0, and 0 * log2(1e-37) = 0
    const wf2 t = d2 + wf2{1e-37f, 1e-37f};
    return d2 * wf2{__builtin_amdgcn_logf(t.x), __builtin_amdgcn_logf(t.y)};
  }
  return wf2{phi_from_d2<ORDER>(d2.x, order), phi_from_d2<ORDER>(d2.y, order)};
}

template <int ORDER, int PADDING, int MC, int ROWS>
__global__ void __launch_bounds__(256) sparse_warp_bands_kernel(const WarpArgs a) {
  static_assert(ROWS % 2 == 0, "rows are carried in pairs");
  constexpr int RP = ROWS / 2;
  const int64_t n = blockIdx.y;
  const int H = a.H, W = a.W;
  const float *__restrict__ tab = a.wv + n * warp_table_stride(MC);  // (wave-uniform addresses: scalar loads)
  const int HW = H * W, bands = (H + ROWS - 1) / ROWS;
  const int idx = (int)(blockIdx.x * 256 + threadIdx.x);
  if (idx >= bands * W) return;
  int band = (int)(((float)idx + 0.5f) * a.inv_w), w = idx - band * W;  // (bands * W < 2^23: checked by the launcher)
  if (w < 0) { --band; w += W; }
  if (w >= W) { ++band; w -= W; }
  const int h0 = band * ROWS;
  const float x = (float)w, yb = (float)h0;
  wf2 yv[RP], sx[RP], sy[RP];
  {
    const float *af = tab + 4 * MC;
    const float axc = __builtin_fmaf(af[0], x, af[4]), ayc = __builtin_fmaf(af[1], x, af[5]);
    const wf2 bx = {af[2], af[2]}, by = {af[3], af[3]};
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      yv[i] = wf2{yb + (float)(2 * i), yb + (float)(2 * i + 1)};
      sx[i] = __builtin_elementwise_fma(yv[i], bx, wf2{axc, axc});
      sy[i] = __builtin_elementwise_fma(yv[i], by, wf2{ayc, ayc});
    }
  }
#pragma unroll
  for (int m = 0; m < MC; ++m) {
    const float kx = tab[4 * m], ky = tab[4 * m + 1], wx = tab[4 * m + 2], wy = tab[4 * m + 3];
    const float dx = x - kx, dx2 = dx * dx;
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const wf2 dy = yv[i] - wf2{ky, ky};
      const wf2 p = phi2_unscaled<ORDER>(__builtin_elementwise_fma(dy, dy, wf2{dx2, dx2}), a.order);
      sx[i] = __builtin_elementwise_fma(p, wf2{wx, wx}, sx[i]);
      sy[i] = __builtin_elementwise_fma(p, wf2{wy, wy}, sy[i]);
    }
  }
  float ix[ROWS], iy[ROWS], x0f[ROWS], y0f[ROWS];
  const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
#pragma unroll
  for (int j = 0; j < ROWS; ++j) {
    float px = (j & 1) ? sx[j >> 1].y : sx[j >> 1].x, py = (j & 1) ? sy[j >> 1].y : sy[j >> 1].x;
    if (PADDING == PAD_REFLECTION) {
      px = reflect_coord(px, -1, 2 * W - 1);
      py = reflect_coord(py, -1, 2 * H - 1);
    }
    if (PADDING != PAD_ZEROS) {  // clip_coord
      px = fminf(wm1, fmaxf(px, 0.0f));
      py = fminf(hm1, fmaxf(py, 0.0f));
    }
    ix[j] = px;
    iy[j] = py;
    x0f[j] = floorf(px);
    y0f[j] = floorf(py);
  }
  const float wf = (float)W;
  for (int c = 0; c < a.C; ++c) {
    const float *pl = a.image + (n * a.C + c) * (int64_t)HW;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(pl), 0, HW * 4, 0x00020000);
    unsigned t00[ROWS], t01[ROWS], t10[ROWS], t11[ROWS];  // (the pixels' BITS: the loads return integers)
    if (PADDING == PAD_ZEROS) {
#pragma unroll
      for (int j = 0; j < ROWS; ++j) {  // taps outside the image: any valid address, left out below
        const int x0 = (int)x0f[j], y0 = (int)y0f[j];
        const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1);
        const int yc0 = min(max(y0, 0), H - 1) * W, yc1 = min(max(y0 + 1, 0), H - 1) * W;
        t00[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, (yc0 + xc0) << 2, 0, 0);
        t01[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, (yc0 + xc1) << 2, 0, 0);
        t10[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, (yc1 + xc0) << 2, 0, 0);
        t11[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, (yc1 + xc1) << 2, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < ROWS; ++j) {  // all the taps in flight
        // (unsigned: with both neighbours outside the two out-of-buffer steps add up to 2^31, which
        //  wraps by definition and is still beyond num_records -- the load returns 0)
        const unsigned o00 = (unsigned)(int)__builtin_fmaf(y0f[j], wf, x0f[j]) << 2;
        const unsigned right = x0f[j] < wm1 ? 4u : 0x40000000u, down = y0f[j] < hm1 ? 4u * (unsigned)W : 0x40000000u;
        const unsigned o10 = o00 + down;
        t00[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)o00, 0, 0);
        t01[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(o00 + right), 0, 0);
        t10[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)o10, 0, 0);
        t11[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(o10 + right), 0, 0);
      }
    }
    float *po = a.out + (n * a.C + c) * (int64_t)HW;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(po, 0, HW * 4, 0x00020000);
    const int obase = (h0 * W + w) << 2;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const float wx1 = ix[j] - x0f[j], wy1 = iy[j] - y0f[j], wx0 = (x0f[j] + 1.0f) - ix[j], wy0 = (y0f[j] + 1.0f) - iy[j];
      float acc;
      if (PADDING == PAD_ZEROS) {
        const int x0 = (int)x0f[j], y0 = (int)y0f[j], x1 = x0 + 1, y1 = y0 + 1;
        const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
        // (a select, not a product with 0: that would turn an inf / NaN pixel into NaN)
        acc = (vx0 && vy0) ? __uint_as_float(t00[j]) * (wx0 * wy0) : 0.0f;
        acc += (vx1 && vy0) ? __uint_as_float(t01[j]) * (wx1 * wy0) : 0.0f;
        acc += (vx0 && vy1) ? __uint_as_float(t10[j]) * (wx0 * wy1) : 0.0f;
        acc += (vx1 && vy1) ? __uint_as_float(t11[j]) * (wx1 * wy1) : 0.0f;
      } else {
        acc = __uint_as_float(t00[j]) * (wx0 * wy0);
        acc += __uint_as_float(t01[j]) * (wx1 * wy0);
        acc += __uint_as_float(t10[j]) * (wx0 * wy1);
        acc += __uint_as_float(t11[j]) * (wx1 * wy1);
      }
      if (h0 + j < H) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc), ro, obase + j * (4 * W), 0, 0);
    }
  }
}

// spec_augment_draw_parameters (_img.py:1056-1139) from ONE tensor of uniform draws: the reference
// makes six torch.rand calls and ~30 tiny tensor ops around them (clamp, floor, masked_fill, long) --
// launch-bound, 0.2-0.5 ms at N = 2048 against the 0.25 ms of the kernel that applies the parameters.
// Here column c of u (N, R) is the c-th draw of utterance n, in the reference's order (w_0, w, v_0, v,
// then the time masks' t, t_0, the frequency masks' f, f_0), and every expression is the reference's
// float32 expression.  (Bitwise parity of the DRAWS with the reference is no goal -- different
// generators per device, SURVEY A.12 -- the mapping from a uniform to a parameter is.)
struct SpecDrawArgs {
  const float *u;
  int R;
  const int64_t *lengths;  // (N,) or null (all T)
  int N, T, F, MT, MF;
  int time_warp, freq_warp, time_mask, freq_mask;  // which groups are drawn
  float max_time_warp, Vf, max_time_mask, max_time_mask_proportion, num_time_mask, num_time_mask_proportion;
  float maxf, eps, omeps;
  float *w_0, *w, *v_0, *v;
  int64_t *t_0, *t, *f_0, *f;
};

__global__ void __launch_bounds__(256) spec_augment_draw_kernel(const SpecDrawArgs a) {
  const int n = (int)(blockIdx.x * 256 + threadIdx.x);
  if (n >= a.N) return;
  const float *u = a.u + (int64_t)n * a.R;
  const float len = a.lengths ? (float)a.lengths[n] : (float)a.T;
  int c = 0;
  if (a.time_warp) {  // :1082-1090
    const float Wt = fminf(fmaxf(len / 2.0f - a.eps, 0.0f), a.max_time_warp);
    a.w_0[n] = u[c] * (len - 2.0f * Wt) + Wt;
    a.w[n] = u[c + 1] * (2.0f * Wt) - Wt;
    c += 2;
  }
  if (a.freq_warp) {  // :1091-1098
    a.v_0[n] = u[c] * ((float)a.F - 2.0f * a.Vf) + a.Vf;
    a.v[n] = u[c + 1] * (2.0f * a.Vf) - a.Vf;
    c += 2;
  }
  if (a.time_mask) {  // :1099-1126
    const float max_ = floorf(fminf(len * a.max_time_mask_proportion, a.max_time_mask));
    const float nums_ = floorf(fminf(len * a.num_time_mask_proportion, a.num_time_mask));
    for (int m = 0; m < a.MT; ++m) {
      int64_t t = (int64_t)(u[c + m] * (max_ + a.omeps));
      if (nums_ <= (float)m) t = 0;
      a.t[(int64_t)n * a.MT + m] = t;
      a.t_0[(int64_t)n * a.MT + m] = (int64_t)(u[c + a.MT + m] * ((len - (float)t) + a.omeps));
    }
    c += 2 * a.MT;
  }
  if (a.freq_mask) {  // :1127-1137
    for (int m = 0; m < a.MF; ++m) {
      const int64_t f = (int64_t)(u[c + m] * (a.maxf + a.omeps));
      a.f[(int64_t)n * a.MF + m] = f;
      a.f_0[(int64_t)n * a.MF + m] = (int64_t)(u[c + a.MF + m] * (((float)a.F - (float)f) + a.omeps));
    }
  }
}

// copy the double solution into float (w, v) laid out (N, M+3, 2) for image_warp_kernel
__global__ void cast_wv_kernel(const double *__restrict__ wv, float *__restrict__ out, int64_t total) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < total) out[i] = (float)wv[i];
}

}  // namespace pdt

extern "C" {

int64_t pdt_spline_workspace_bytes(int64_t N, int64_t T, int64_t I, int64_t O) {
  if (N < 0 || T < 0 || I < 0 || O < 0) return 0;
  const int64_t S = T + I + 1;
  int64_t bytes = N * S * O * (int64_t)(sizeof(double) + sizeof(float)) + 64;
  bytes += N * pdt::kWarpTableFloats * (int64_t)sizeof(float);  // sparse_warp_bands_kernel's per-image table
  // systems beyond the LDS are eliminated in global memory, after the solutions
  if ((size_t)S * (S + O) * sizeof(double) + 16 > pdt::kSplineLdsCap) bytes += N * S * (S + O) * (int64_t)sizeof(double);
  return bytes;
}

static int spline_solve(const float *c, const float *f, const float *tail, int64_t N, int64_t T,
                        int64_t I, int64_t O, int order, float reg, double *wv, hipStream_t stream) {
  using namespace pdt;
  const int64_t S = T + I + 1;
  if (S > 4096 || O > 64) return PDT_E_TOO_LONG;
  size_t smem = (size_t)S * (S + O) * sizeof(double) + 16;
  double *gmat = nullptr;
  if (smem > kSplineLdsCap) {  // the matrix follows the (N, S, O) doubles + floats of the solutions
    gmat = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(wv) +
                                      (((size_t)N * S * O * (sizeof(double) + sizeof(float)) + 63) & ~(size_t)63));
    smem = 16;
  }
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(spline_solve_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(spline_solve_kernel, dim3((unsigned)N), dim3(256), smem, stream, c, f, tail,
                     (int)T, (int)I, (int)O, order, reg, wv, gmat);
  return (int)hipGetLastError();
}

int pdt_spline_solve(const float *train_points, const float *train_values, const float *tail,
                     int64_t N, int64_t T, int64_t I, int64_t O, int order,
                     float regularization_weight, double *solution, void *workspace, void *stream) {
  using namespace pdt;
  if (N < 0 || T < 1 || I < 1 || O < 1 || order < 1) return PDT_E_ARG;
  if (N == 0) return PDT_OK;
  if (!train_points || !train_values || !solution || !workspace) return PDT_E_ARG;
  if (N > 65535) return PDT_E_TOO_LONG;
  double *wv = reinterpret_cast<double *>(workspace);
  int rc = spline_solve(train_points, train_values, tail, N, T, I, O, order, regularization_weight, wv,
                        (hipStream_t)stream);
  if (rc != PDT_OK) return rc;
  return (int)hipMemcpyAsync(solution, wv, (size_t)N * (T + I + 1) * O * sizeof(double),
                             hipMemcpyDeviceToDevice, (hipStream_t)stream);
}

int pdt_polyharmonic_spline(const float *train_points, const float *train_values,
                            const float *query_points, int64_t N, int64_t T, int64_t I, int64_t O,
                            int64_t Q, int order, float regularization_weight, float *out,
                            void *workspace, void *stream) {
  using namespace pdt;
  if (N < 0 || T < 1 || I < 1 || O < 1 || Q < 0 || order < 1) return PDT_E_ARG;
  if (N == 0 || Q == 0) return PDT_OK;
  if (!train_points || !train_values || !query_points || !out || !workspace) return PDT_E_ARG;
  if (N > 65535) return PDT_E_TOO_LONG;
  double *wv = reinterpret_cast<double *>(workspace);
  int rc = spline_solve(train_points, train_values, nullptr, N, T, I, O, order, regularization_weight, wv,
                        (hipStream_t)stream);
  if (rc != PDT_OK) return rc;
  const size_t smem = (size_t)(T + I + 1) * O * sizeof(double) + (size_t)T * I * sizeof(float);
  hipLaunchKernelGGL(spline_apply_kernel, dim3((unsigned)((Q + 255) / 256), (unsigned)N), dim3(256),
                     smem, (hipStream_t)stream, train_points, wv, query_points, (int)T, (int)I,
                     (int)O, (int)Q, order, out);
  return (int)hipGetLastError();
}

int pdt_warp_1d_grid(const float *src, const float *flow, const float *lengths, int64_t N, int64_t T,
                     int order, float *grid, void *stream) {
  using namespace pdt;
  if (N < 0 || T < 0 || order < 1) return PDT_E_ARG;
  if (N == 0 || T == 0) return PDT_OK;
  if (!src || !flow || !lengths || !grid) return PDT_E_ARG;
  hipLaunchKernelGGL(warp_1d_grid_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, src,
                     flow, lengths, (int)T, order, grid);
  return (int)hipGetLastError();
}

int pdt_spec_augment_draw(const float *u, int64_t N, int64_t R, const int64_t *lengths, int64_t T,
                          int64_t F, float max_time_warp, float max_freq_warp, int64_t max_time_mask,
                          int64_t max_freq_mask, float max_time_mask_proportion, int64_t num_time_mask,
                          float num_time_mask_proportion, int64_t num_freq_mask, int is_double,
                          float *w_0, float *w, float *v_0, float *v, int64_t *t_0, int64_t *t,
                          int64_t *f_0, int64_t *f, void *stream) {
  using namespace pdt;
  if (N < 0 || T < 0 || F < 0 || R < 0 || num_time_mask < 0 || num_freq_mask < 0) return PDT_E_ARG;
  SpecDrawArgs a{};
  a.time_warp = max_time_warp != 0.0f;
  a.freq_warp = max_freq_warp != 0.0f;
  a.time_mask = max_time_mask != 0 && max_time_mask_proportion != 0.0f && num_time_mask != 0 &&
                num_time_mask_proportion != 0.0f;
  a.freq_mask = max_freq_mask != 0 && num_freq_mask != 0;
  a.MT = a.time_mask ? (int)num_time_mask : 0;
  a.MF = a.freq_mask ? (int)num_freq_mask : 0;
  if (R < 2 * a.time_warp + 2 * a.freq_warp + 2 * a.MT + 2 * a.MF) return PDT_E_ARG;
  if (N == 0 || (!a.time_warp && !a.freq_warp && !a.time_mask && !a.freq_mask)) return PDT_OK;  // nothing to draw
  if (!u || (a.time_warp && (!w_0 || !w)) || (a.freq_warp && (!v_0 || !v)) || (a.time_mask && (!t_0 || !t)) ||
      (a.freq_mask && (!f_0 || !f)))
    return PDT_E_ARG;
  // (the reference's eps is that of the features' dtype; its arithmetic on the draws is float32 either way)
  const double eps = is_double ? 2.220446049250313e-16 : 1.1920928955078125e-07;
  a.u = u; a.R = (int)R; a.lengths = lengths;
  a.N = (int)N; a.T = (int)T; a.F = (int)F;
  a.eps = (float)eps;
  a.omeps = (float)(1.0 - eps);
  a.max_time_warp = max_time_warp;
  a.Vf = (float)std::fmin(std::fmax((double)F / 2.0 - eps, 0.0), (double)max_freq_warp);
  a.max_time_mask = (float)max_time_mask;
  a.max_time_mask_proportion = max_time_mask_proportion;
  a.num_time_mask = (float)num_time_mask;
  a.num_time_mask_proportion = num_time_mask_proportion;
  a.maxf = (float)std::min<int64_t>(max_freq_mask, F);
  a.w_0 = w_0; a.w = w; a.v_0 = v_0; a.v = v; a.t_0 = t_0; a.t = t; a.f_0 = f_0; a.f = f;
  hipLaunchKernelGGL(spec_augment_draw_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int pdt_spec_augment_apply(const float *feats, int64_t N, int64_t T, int64_t F, int64_t f_sn,
                           int64_t f_st, int64_t f_sf, const float *time_grid,
                           const float *freq_grid, const int64_t *t_0, const int64_t *t_len,
                           int64_t MT, const int64_t *f_0, const int64_t *f_len, int64_t MF,
                           float *out, void *stream) {
  using namespace pdt;
  if (N < 0 || T < 0 || F < 0 || MT < 0 || MF < 0) return PDT_E_ARG;
  if (N == 0 || T == 0 || F == 0) return PDT_OK;
  if (!feats || !out || (MT > 0 && (!t_0 || !t_len)) || (MF > 0 && (!f_0 || !f_len))) return PDT_E_ARG;
  if (T * F >= (1ll << 31)) return PDT_E_TOO_LONG;
  SpecAugArgs a{};
  a.feats = feats; a.f_sn = f_sn; a.f_st = f_st; a.f_sf = f_sf;
  a.tgrid = time_grid; a.fgrid = freq_grid;
  a.t0 = t_0; a.tl = t_len; a.f0 = f_0; a.fl = f_len;
  a.N = (int)N; a.T = (int)T; a.F = (int)F; a.MT = (int)MT; a.MF = (int)MF;
  a.out = out;
  // ~16K elements per workgroup keeps >= 8 workgroups per CU in flight at N = 2048
  int tiles = (int)((T * F + 16383) / 16384);
  if (tiles < 1) tiles = 1;
  const bool rows_path = !freq_grid && (F % 4 == 0) && f_sf == 1 && (f_st % 4 == 0) &&
                         (f_sn % 4 == 0) && ((uintptr_t)feats % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (rows_path && F <= 256) {
    const int rtiles = (int)((T + kRowsPerTile - 1) / kRowsPerTile);
    hipLaunchKernelGGL(spec_augment_rows_kernel, dim3((unsigned)(N * rtiles)), dim3(256), 0,
                       (hipStream_t)stream, a, rtiles);
  }
  else
    hipLaunchKernelGGL(spec_augment_apply_kernel, dim3((unsigned)(N * tiles)), dim3(256), 0,
                       (hipStream_t)stream, a, tiles);
  return (int)hipGetLastError();
}

int pdt_spec_augment_apply_warp(const float *feats, int64_t N, int64_t T, int64_t F, int64_t f_sn,
                                int64_t f_st, int64_t f_sf, const float *warp_src, const float *warp_flow,
                                const int64_t *lengths, int order, const int64_t *t_0,
                                const int64_t *t_len, int64_t MT, const int64_t *f_0,
                                const int64_t *f_len, int64_t MF, float *out, int32_t *bad_lengths,
                                void *stream) {
  using namespace pdt;
  if (N < 0 || T < 0 || F < 0 || MT < 0 || MF < 0 || order < 1) return PDT_E_ARG;
  if (N == 0 || T == 0 || F == 0) return PDT_OK;
  if (!feats || !out || !warp_src || !warp_flow || (MT > 0 && (!t_0 || !t_len)) || (MF > 0 && (!f_0 || !f_len)))
    return PDT_E_ARG;
  if (T * F >= (1ll << 31)) return PDT_E_TOO_LONG;
  const bool rows_path = (F % 4 == 0) && F <= 256 && f_sf == 1 && (f_st % 4 == 0) && (f_sn % 4 == 0) &&
                         ((uintptr_t)feats % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (!rows_path) return PDT_E_UNSUPPORTED;
  SpecAugArgs a{};
  a.feats = feats; a.f_sn = f_sn; a.f_st = f_st; a.f_sf = f_sf;
  a.t0 = t_0; a.tl = t_len; a.f0 = f_0; a.fl = f_len;
  a.N = (int)N; a.T = (int)T; a.F = (int)F; a.MT = (int)MT; a.MF = (int)MF;
  a.out = out;
  a.tw_src = warp_src; a.tw_flow = warp_flow; a.tw_len = lengths; a.tw_order = order;
  a.bad_lengths = lengths ? bad_lengths : nullptr;
  const int rtiles = (int)((T + kRowsPerTile - 1) / kRowsPerTile);
  hipLaunchKernelGGL(spec_augment_rows_kernel, dim3((unsigned)(N * rtiles)), dim3(256), 0, (hipStream_t)stream, a,
                     rtiles);
  return (int)hipGetLastError();
}

int pdt_spec_augment_apply_backward(const float *grad_out, int64_t N, int64_t T, int64_t F,
                                    const float *time_grid, const float *freq_grid,
                                    const int64_t *t_0, const int64_t *t_len, int64_t MT,
                                    const int64_t *f_0, const int64_t *f_len, int64_t MF,
                                    float *grad_feats, void *stream) {
  using namespace pdt;
  if (N < 0 || T < 0 || F < 0 || MT < 0 || MF < 0) return PDT_E_ARG;
  if (N == 0 || T == 0 || F == 0) return PDT_OK;
  if (!grad_out || !grad_feats || (MT > 0 && (!t_0 || !t_len)) || (MF > 0 && (!f_0 || !f_len)))
    return PDT_E_ARG;
  if (T * F >= (1ll << 31)) return PDT_E_TOO_LONG;
  SpecAugArgs a{};
  a.tgrid = time_grid; a.fgrid = freq_grid;
  a.t0 = t_0; a.tl = t_len; a.f0 = f_0; a.fl = f_len;
  a.N = (int)N; a.T = (int)T; a.F = (int)F; a.MT = (int)MT; a.MF = (int)MF;
  const size_t rows_smem = (size_t)T * 12 + 260 * 4 + 64 * 4;
  if (!freq_grid && F % 4 == 0 && F <= 256 && rows_smem <= 64 * 1024 &&
      ((uintptr_t)grad_out % 16 == 0) && ((uintptr_t)grad_feats % 16 == 0)) {
    const int rtiles = (int)((T + kRowsPerTile - 1) / kRowsPerTile);
    hipLaunchKernelGGL(spec_augment_rows_backward_kernel, dim3((unsigned)(N * rtiles)), dim3(256),
                       rows_smem, (hipStream_t)stream, a, grad_out, grad_feats, rtiles);
    return (int)hipGetLastError();
  }
  hipError_t e = hipMemsetAsync(grad_feats, 0, (size_t)(N * T * F) * sizeof(float), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  int tiles = (int)((T * F + 16383) / 16384);
  if (tiles < 1) tiles = 1;
  hipLaunchKernelGGL(spec_augment_backward_kernel, dim3((unsigned)(N * tiles)), dim3(256), 0,
                     (hipStream_t)stream, a, grad_out, grad_feats, tiles);
  return (int)hipGetLastError();
}

extern "C++" {
template <typename PT>
static int dense_warp_launch(const PT *image, const float *flow, int64_t N, int64_t C, int64_t H,
                             int64_t W, int flow_is_hw, int mode, int padding, PT *out,
                             PT *grad_image, void *stream) {
  using namespace pdt;
  if (N < 0 || C < 0 || H < 0 || W < 0 || mode < 0 || mode > 1 || padding < 0 || padding > 2)
    return PDT_E_ARG;
  if (N == 0 || C == 0 || H == 0 || W == 0) return PDT_OK;
  if ((!image && !grad_image) || !flow || !out) return PDT_E_ARG;
  if (H * W >= (1ll << 31) || N > 65535) return PDT_E_TOO_LONG;
  WarpArgsT<PT> a{};
  a.image = image; a.out = out; a.N = (int)N; a.C = (int)C; a.H = (int)H; a.W = (int)W;
  a.mode = mode; a.padding = padding; a.flow = flow; a.flip = flow_is_hw;
  a.grad_image = grad_image;
  const dim3 grid((unsigned)((H * W + kPixPerWG - 1) / kPixPerWG), (unsigned)N);
  if (grad_image) {
    hipError_t e = hipMemsetAsync(grad_image, 0, (size_t)(N * C * H * W) * sizeof(PT),
                                  (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((image_warp_kernel<true, PT>), grid, dim3(256), 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL((image_warp_kernel<false, PT>), grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  return (int)hipGetLastError();
}
}  // extern "C++"

int pdt_dense_image_warp(const float *image, const float *flow, int64_t N, int64_t C, int64_t H,
                         int64_t W, int flow_is_hw, int mode, int padding, float *out,
                         void *stream) {
  return dense_warp_launch<float>(image, flow, N, C, H, W, flow_is_hw, mode, padding, out, nullptr, stream);
}

int pdt_dense_image_warp_backward(const float *grad_out, const float *flow, int64_t N, int64_t C,
                                  int64_t H, int64_t W, int flow_is_hw, int mode, int padding,
                                  float *grad_image, void *stream) {
  if (!grad_image && N && C && H && W) return PDT_E_ARG;
  return dense_warp_launch<float>(nullptr, flow, N, C, H, W, flow_is_hw, mode, padding,
                                  const_cast<float *>(grad_out), grad_image, stream);
}

extern "C++" {
template <typename PT>
static int sparse_warp_launch(const PT *image, const float *train_points,
                              const float *train_values, int64_t N, int64_t C, int64_t H, int64_t W,
                              int64_t M, int order, float regularization_weight, int values_are_grid,
                              int mode, int padding, PT *out, float *flow_out, int flow_out_is_hw,
                              PT *grad_image, void *workspace, void *stream) {
  using namespace pdt;
  if (N < 0 || C < 0 || H < 0 || W < 0 || M < 1 || order < 1 || mode < 0 || mode > 1 ||
      padding < 0 || padding > 2)
    return PDT_E_ARG;
  if (N == 0 || C == 0 || H == 0 || W == 0) return PDT_OK;
  if ((!image && !grad_image) || !train_points || !train_values || !out || !workspace)
    return PDT_E_ARG;
  if (H * W >= (1ll << 31) || N > 65535) return PDT_E_TOO_LONG;
  double *wv = reinterpret_cast<double *>(workspace);
  int rc = spline_solve(train_points, train_values, nullptr, N, M, 2, 2, order, regularization_weight, wv,
                        (hipStream_t)stream);
  if (rc != PDT_OK) return rc;
  const int64_t total = N * (M + 3) * 2;
  float *wvf = reinterpret_cast<float *>(wv + total);
  // (the bands kernel reads its own table, written from the double solution by warp_table_kernel)
  constexpr bool kFloat = std::is_same<PT, float>::value;  // (the fast forms are float32 kernels)
  const bool fast_shape = kFloat && !grad_image && mode == INTERP_BILINEAR && !flow_out && M <= kWarpFastM && H * W < (1 << 23);
  if (!(fast_shape && switches().warp_bands != 0))
    hipLaunchKernelGGL(cast_wv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, wv, wvf, total);
  WarpArgsT<PT> a{};
  a.image = image; a.out = out; a.N = (int)N; a.C = (int)C; a.H = (int)H; a.W = (int)W;
  a.mode = mode; a.padding = padding;
  a.knots = train_points; a.wv = wvf; a.M = (int)M; a.order = order; a.as_grid = values_are_grid;
  a.flow_out = flow_out; a.flow_out_flip = flow_out_is_hw;
  a.grad_image = grad_image;
  const size_t smem = (size_t)(2 * M + 2 * (M + 3)) * sizeof(float);
  const dim3 grid((unsigned)((H * W + kPixPerWG - 1) / kPixPerWG), (unsigned)N);
  if (grad_image) {
    hipError_t e = hipMemsetAsync(grad_image, 0, (size_t)(N * C * H * W) * sizeof(PT),
                                  (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((image_warp_kernel<true, PT>), grid, dim3(256), smem, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  if constexpr (kFloat) {
   if (fast_shape) {
    // a lane = a column of kBandRows rows, the image's constants from a table (sparse_warp_bands_kernel);
    // PDT_WARP_BANDS=0: four pixels 256 apart per lane (sparse_warp_fast_kernel, for comparisons)
    const bool rows4 = switches().warp_bands != 0;
    const int per_wg = 256 * kWarpPix;
    const int64_t lanes = ((H + kBandRows - 1) / kBandRows) * W;
    const dim3 gf(rows4 ? (unsigned)((lanes + 255) / 256) : (unsigned)((H * W + per_wg - 1) / per_wg), (unsigned)N);
    auto go = [&](auto ord) {
      constexpr int O = decltype(ord)::value;
      if (rows4) {
        auto bands = [&](auto mc) {
          constexpr int MC = decltype(mc)::value;
          float *tab = wvf + total;
          const int64_t entries = N * warp_table_stride(MC);
          hipLaunchKernelGGL(warp_table_kernel, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                             wv, train_points, tab, N, (int)M, MC, values_are_grid, (int)H, (int)W, order);
          a.wv = tab;
          a.inv_w = 1.0f / (float)W;
          if (padding == PAD_BORDER) hipLaunchKernelGGL((sparse_warp_bands_kernel<O, PAD_BORDER, MC, kBandRows>), gf, dim3(256), 0, (hipStream_t)stream, a);
          else if (padding == PAD_ZEROS) hipLaunchKernelGGL((sparse_warp_bands_kernel<O, PAD_ZEROS, MC, kBandRows>), gf, dim3(256), 0, (hipStream_t)stream, a);
          else hipLaunchKernelGGL((sparse_warp_bands_kernel<O, PAD_REFLECTION, MC, kBandRows>), gf, dim3(256), 0, (hipStream_t)stream, a);
        };
        // (seven centres = three control points + four pinned corners, the SpecAugment-style call)
        if (M == 7) bands(std::integral_constant<int, 7>{}); else bands(std::integral_constant<int, kWarpFastM>{});
        return;
      }
      if (padding == PAD_BORDER) hipLaunchKernelGGL((sparse_warp_fast_kernel<O, PAD_BORDER>), gf, dim3(256), smem, (hipStream_t)stream, a);
      else if (padding == PAD_ZEROS) hipLaunchKernelGGL((sparse_warp_fast_kernel<O, PAD_ZEROS>), gf, dim3(256), smem, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((sparse_warp_fast_kernel<O, PAD_REFLECTION>), gf, dim3(256), smem, (hipStream_t)stream, a);
    };
    if (order == 2) go(std::integral_constant<int, 2>{});
    else if (order == 1) go(std::integral_constant<int, 1>{});
    else if (order == 3) go(std::integral_constant<int, 3>{});
    else go(std::integral_constant<int, 0>{});
    return (int)hipGetLastError();
   }
  }
  hipLaunchKernelGGL((image_warp_kernel<false, PT>), grid, dim3(256), smem, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
}  // extern "C++"

int pdt_sparse_image_warp(const float *image, const float *train_points,
                          const float *train_values, int64_t N, int64_t C, int64_t H, int64_t W,
                          int64_t M, int order, float regularization_weight, int values_are_grid,
                          int mode, int padding, float *out, float *flow_out, int flow_out_is_hw,
                          void *workspace, void *stream) {
  return sparse_warp_launch<float>(image, train_points, train_values, N, C, H, W, M, order,
                                   regularization_weight, values_are_grid, mode, padding, out, flow_out,
                                   flow_out_is_hw, nullptr, workspace, stream);
}

int pdt_sparse_image_warp_backward(const float *grad_out, const float *train_points,
                                   const float *train_values, int64_t N, int64_t C, int64_t H,
                                   int64_t W, int64_t M, int order, float regularization_weight,
                                   int values_are_grid, int mode, int padding, float *grad_image,
                                   void *workspace, void *stream) {
  if (!grad_image && N && C && H && W) return PDT_E_ARG;
  return sparse_warp_launch<float>(nullptr, train_points, train_values, N, C, H, W, M, order,
                                   regularization_weight, values_are_grid, mode, padding,
                                   const_cast<float *>(grad_out), nullptr, 0, grad_image, workspace, stream);
}

// float64 images (the reference samples a double image on a double grid, _img.py:423-436; flows and
// spline points are float32 there whatever the image's type, :420, :537-538): image_warp_kernel in
// double from the grid on.  Same arguments as the float32 entries.
int pdt_dense_image_warp_f64(const double *image, const float *flow, int64_t N, int64_t C, int64_t H,
                             int64_t W, int flow_is_hw, int mode, int padding, double *out,
                             void *stream) {
  return dense_warp_launch<double>(image, flow, N, C, H, W, flow_is_hw, mode, padding, out, nullptr, stream);
}

int pdt_dense_image_warp_backward_f64(const double *grad_out, const float *flow, int64_t N, int64_t C,
                                      int64_t H, int64_t W, int flow_is_hw, int mode, int padding,
                                      double *grad_image, void *stream) {
  if (!grad_image && N && C && H && W) return PDT_E_ARG;
  return dense_warp_launch<double>(nullptr, flow, N, C, H, W, flow_is_hw, mode, padding,
                                   const_cast<double *>(grad_out), grad_image, stream);
}

int pdt_sparse_image_warp_f64(const double *image, const float *train_points,
                              const float *train_values, int64_t N, int64_t C, int64_t H, int64_t W,
                              int64_t M, int order, float regularization_weight, int values_are_grid,
                              int mode, int padding, double *out, float *flow_out, int flow_out_is_hw,
                              void *workspace, void *stream) {
  return sparse_warp_launch<double>(image, train_points, train_values, N, C, H, W, M, order,
                                    regularization_weight, values_are_grid, mode, padding, out, flow_out,
                                    flow_out_is_hw, nullptr, workspace, stream);
}

int pdt_sparse_image_warp_backward_f64(const double *grad_out, const float *train_points,
                                       const float *train_values, int64_t N, int64_t C, int64_t H,
                                       int64_t W, int64_t M, int order, float regularization_weight,
                                       int values_are_grid, int mode, int padding, double *grad_image,
                                       void *workspace, void *stream) {
  if (!grad_image && N && C && H && W) return PDT_E_ARG;
  return sparse_warp_launch<double>(nullptr, train_points, train_values, N, C, H, W, M, order,
                                    regularization_weight, values_are_grid, mode, padding,
                                    const_cast<double *>(grad_out), nullptr, 0, grad_image, workspace, stream);
}

}  // extern "C"
