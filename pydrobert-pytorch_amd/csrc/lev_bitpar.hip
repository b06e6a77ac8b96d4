// Bit-parallel Levenshtein for unit costs (ins = del = sub = 1, what every uniform-cost call
// becomes after the rescaling of reference _string.py:168-174) on gfx950.
//
// The cell recurrence of _string_matching (reference _string.py:286-346) moves by -1 / 0 / +1
// between neighbouring cells, so a whole DP column is two bit-vectors of vertical deltas
// (Pv: +1, Mv: -1) and one column update costs ~20 word operations per 32 cells (Myers 1999;
// the block form with carried horizontal deltas is Hyyro 2003) instead of ~5 per cell in
// lev_skewed.hip.  Values are small integers either way: results are identical.
//
// Two kernels, fed from a caller-provided workspace (pdt_lev_workspace_bytes):
//   lev_classify_kernel  one wave per utterance: sequence lengths, the distinct tokens of the
//                        bit-vector sequence X (= hyp) in ascending order (bitonic sort in
//                        registers), and the match masks Eq[class][block] in compressed rows --
//                        per class a 32-bit block-presence word + an offset into a packed array of
//                        mask words (one word per (class, block) pair that has a match: at most
//                        |X| words).  Every position of the consumed sequence Y (= ref) gets its
//                        class's (presence, offset) pair, so the DP loop never sees a token.
//   lev_bitpar_kernel    L = 2^k >= |X| / 32 lanes per utterance, 64 / L utterances per wave.
//                        Lane b owns block b (rows 32 b + 1 .. 32 b + 32) and runs one column
//                        behind lane b - 1, which hands it the horizontal delta of its last row
//                        through one DPP shift -- an anti-diagonal pipeline over blocks.  After
//                        the last reference token the vectors hold D[ref_len][h] - D[ref_len][h-1]
//                        for every h: all prefix distances come out of one final prefix sum.
// (With X = hyp the transposed table is computed; unit costs make it the same table.)
//
// Optimal completion stays on lev_rowsync.hip: a form of this pipeline that dumped every column's
// (Pv, Mv) words and searched them for the arg-min rows (nibble-table block minima, bounds from
// popcounts, one lane per column) measured 0.48 ms against 0.51 -- its cost is the arg-min sets
// (13 rows per column at the bench shape, each a class look-up), not the recurrence.
#include <algorithm>
#include <type_traits>

#include "lev_classes.hpp"

namespace pdt {

struct BitparArgs {
  const int64_t *ref, *hyp;
  int64_t ref_st, ref_sn, hyp_st, hyp_sn;
  int R, H, N;
  int has_eos, include_eos;
  int64_t eos;
  int exclude_last, norm, mode;
  float mult, padding;
  float *out;
  int64_t out_sh, out_sn;
  int64_t *ref_lens_out, *hyp_lens_out;
  int32_t *status;
  int X, Y;      // capacities of the bit-vector sequence and of the consumed sequence
  int lgL, upw;  // lanes per utterance = 1 << lgL; utterances per wave (<= 64 >> lgL)
  int32_t *lens;   // [N][2]     ref_len, hyp_len
  uint2 *yh;       // [N][Y]     (block presence, offset) of the class of Y[j]; (0, 0) = no match
  uint32_t *msk;   // [N][X + 1] packed match-mask words
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// the recurrence kernel's ring of match words: two buffers of 16 words per lane, lane stride 20
// words (conflict-free 16-byte accesses)
constexpr int kBitparChunk = 16, kBitparLaneStride = 20;
constexpr size_t kBitparRingBytes = (size_t)2 * PDT_WAVE * kBitparLaneStride * 4;

// X: length of the bit-vector sequence, Y: of the consumed one.  Blocks are 32 rows and the
// presence word has 32 bits: X <= 1024.
BitparPlan plan_bitpar(int64_t X, int64_t Y, int64_t N) {
  BitparPlan p{};
  if (X > 1024 || Y > (1 << 20) || N <= 0) return p;
  int lgL = 0;
  while ((32 << lgL) < X) ++lgL;
  p.lgL = lgL;
  const size_t Xs = (size_t)(X > 0 ? X : 1), Ys = (size_t)(Y > 0 ? Y : 1);
  // classify: [(presence, offset) per class X * 8] [tokens (X + 1) * 8 (later: the mask words)]
  //           [classes of Y, 2 bytes each]
  p.lds_classify = align_up(Xs * 8 + (Xs + 1) * 8 + Ys * 2, 16);
  // DP: per utterance [yh Y * 8] [mask words (X + 1) * 4]; per workgroup the ring of match words
  // (two chunks of kRingWords per lane), which the distances (X + 1) * 4 per utterance take over
  p.lds_sub = align_up(Ys * 8 + (Xs + 1) * 4, 16);
  int upw = 64 >> lgL;
  auto tail = [&](int u) { return align_up(std::max<size_t>((size_t)u * (Xs + 1) * 4, kBitparRingBytes), 16); };
  while (upw > 1 && p.lds_sub * upw + tail(upw) > 40 * 1024) upw >>= 1;  // (four workgroups per CU)
  if (p.lds_sub * upw + tail(upw) > 150 * 1024 || p.lds_classify * 4 > 160 * 1024) return p;
  p.upw = upw;
  p.lds_tail = tail(upw);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  p.off_lens = take((size_t)N * 8);
  p.off_yh = take((size_t)N * Ys * 8);
  p.off_msk = take((size_t)N * (Xs + 1) * 4);
  p.total = off;
  p.ok = 1;
  return p;
}

// ---- kernel 1: lengths, classes, compressed match masks -------------------------------------
// (Time-major inputs cost this kernel ~15 us at the bench shape -- a wave's tokens sit in 512
// different 32-byte sectors; reading (T, 4) strips with the whole workgroup and exchanging them
// through LDS measured no better.)
template <int NR>
__global__ void __launch_bounds__(256) lev_classify_kernel(const BitparArgs a, const int lds_per_wave) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int64_t n = (int64_t)blockIdx.x * 4 + wave;
  if (n >= a.N) return;  // waves never synchronise with each other
  unsigned char *base = smem + (size_t)wave * lds_per_wave;
  const int X = a.X > 0 ? a.X : 1, Y = a.Y > 0 ? a.Y : 1;
  // [(presence, offset) per class: X * 8] [distinct tokens (X + 1) * 8; later the packed mask
  // words] [classes of Y, 2 bytes each]
  uint2 *po = reinterpret_cast<uint2 *>(base);
  int64_t *ctok = reinterpret_cast<int64_t *>(base + (size_t)X * 8);
  unsigned *msk = reinterpret_cast<unsigned *>(ctok);
  short *yc = reinterpret_cast<short *>(ctok + X + 1);

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
  int Heff = a.exclude_last ? hyp_len - 1 : hyp_len;  // rows that are updated (:286-288)
  if (Heff < 0) Heff = 0;
  // bit-vectors along the hypothesis (only its first Heff tokens matter), reference consumed
  const int64_t *x = a.hyp, *y = a.ref;
  const int64_t x_st = a.hyp_st, y_st = a.ref_st, xoff = hoff, yoff = roff;
  const int x_len = Heff, y_len = ref_len;

  // ---- distinct tokens of X in ascending order (lev_classes.hpp) ---------------------------
  int64_t xt[NR];
  const int U = distinct_sorted<NR>(x, x_len, x_st, xoff, xt, ctok);
  wave_sync();
  for (int k = lane; k < U; k += PDT_WAVE) po[k] = make_uint2(0u, 0u);
  const int lgP = search_depth(U);
  int xc[NR];  // classes of X[lane + 64 q]
  classes_of<NR>(ctok, U, lgP, xt, xc);
  for (int j0 = 0; j0 < y_len; j0 += 8 * PDT_WAVE) {
    int64_t yt[8];
    int c[8];
    load_tokens(y, y_len, y_st, yoff, j0, 0, yt);
    classes_of<8>(ctok, U, lgP, yt, c);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (j0 + lane + q * PDT_WAVE < y_len) yc[j0 + lane + q * PDT_WAVE] = (short)c[q];
  }
  wave_sync();  // the token table is dead from here on: the mask words take it over
  for (int i = lane; i <= x_len; i += PDT_WAVE) msk[i] = 0u;
  auto for_x = [&](auto &&f) {  // f(position, class) over this lane's positions of X
#pragma unroll
    for (int q = 0; q < NR; ++q)
      if (lane + q * PDT_WAVE < x_len) f(lane + q * PDT_WAVE, xc[q]);
  };
  for_x([&](const int i, const int c) {
    atomicOr(&po[c].x, 1u << (i >> 5));  // blocks that hold the class
  });
  wave_sync();
  {  // offsets = exclusive scan of the presence popcounts
    const int B = (U + PDT_WAVE - 1) / PDT_WAVE;
    const int i0 = lane * B;
    int sum = 0;
    for (int q = 0; q < B; ++q)
      if (i0 + q < U) sum += __popc(po[i0 + q].x);
    const int incl = wave_incl_scan_add(sum);
    int pos = incl - sum;
    for (int q = 0; q < B; ++q)
      if (i0 + q < U) {
        po[i0 + q].y = (unsigned)pos;
        pos += __popc(po[i0 + q].x);
      }
  }
  wave_sync();
  for_x([&](const int i, const int c) {
    const uint2 e = po[c];
    atomicOr(&msk[e.y + (unsigned)__popc(e.x & ((1u << (i >> 5)) - 1u))], 1u << (i & 31));
  });
  wave_sync();
  for (int j = lane; j < y_len; j += PDT_WAVE) {
    const int c = yc[j];
    a.yh[n * (int64_t)Y + j] = c >= 0 ? po[c] : make_uint2(0u, 0u);
  }
  for (int i = lane; i <= x_len; i += PDT_WAVE) a.msk[n * (int64_t)(X + 1) + i] = msk[i];
  if (lane == 0) {
    a.lens[2 * n] = ref_len;
    a.lens[2 * n + 1] = hyp_len;
    int flags = 0;
    if (rmiss) flags |= PDT_WARN_REF_NO_EOS;
    if (hmiss) flags |= PDT_WARN_HYP_NO_EOS;
    if (a.norm && ref_len == 0) flags |= PDT_WARN_EMPTY_REF;
    if (flags && a.status) atomicOr(a.status, flags);
    if (a.ref_lens_out) a.ref_lens_out[n] = ref_len;
    if (a.hyp_lens_out) a.hyp_lens_out[n] = hyp_len;
  }
}

// ---- kernel 2: the column recurrence ----------------------------------------------------------
// Two waves per workgroup: wave 1 looks the match words up one chunk of steps ahead and leaves
// them in an LDS ring, wave 0 runs the recurrence -- a lone wave issues an instruction every ~5.5
// cycles whatever it is, and the look-ups were 12 of the loop's 40 instructions per step.
__global__ void __launch_bounds__(128) lev_bitpar_kernel(const BitparArgs a, const int lds_per_sub, const int ring_off) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int L = 1 << a.lgL;
  const int sub = lane >> a.lgL, b = lane & (L - 1);
  const int64_t n_raw = (int64_t)blockIdx.x * a.upw + sub;
  const bool valid = sub < a.upw && n_raw < a.N;
  const int64_t n = valid ? n_raw : (int64_t)blockIdx.x * a.upw;  // (a safe utterance to address)
  const int X = a.X > 0 ? a.X : 1, Y = a.Y > 0 ? a.Y : 1;
  unsigned char *base = smem + (size_t)(valid ? sub : 0) * lds_per_sub;
  uint2 *yh_l = reinterpret_cast<uint2 *>(base);
  unsigned *msk_l = reinterpret_cast<unsigned *>(yh_l + Y);
  unsigned *ring = reinterpret_cast<unsigned *>(smem + ring_off);
  float *bnd = reinterpret_cast<float *>(smem + ring_off) + (size_t)(valid ? sub : 0) * (X + 1);  // (after the loop)

  const int ref_len = a.lens[2 * n], hyp_len = a.lens[2 * n + 1];
  int Heff = a.exclude_last ? hyp_len - 1 : hyp_len;
  if (Heff < 0) Heff = 0;
  const int x_len = valid ? Heff : 0, y_len = valid ? ref_len : 0;

  // ---- stage this utterance's lookups in LDS (eight loads in flight per lane) --------------
  {
    const uint2 *src = a.yh + n * (int64_t)Y;
    for (int j0 = b + wave * 8 * L; j0 < y_len; j0 += 16 * L) {
      uint2 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = j0 + q * L < y_len ? src[j0 + q * L] : make_uint2(0u, 0u);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (j0 + q * L < y_len) yh_l[j0 + q * L] = v[q];
    }
    if (valid && y_len == 0 && b == 0 && wave == 0) yh_l[0] = make_uint2(0u, 0u);
    const unsigned *msrc = a.msk + n * (int64_t)(X + 1);
    for (int i0 = b + wave * 8 * L; i0 <= x_len && valid; i0 += 16 * L) {
      unsigned v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = i0 + q * L <= x_len ? msrc[i0 + q * L] : 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (i0 + q * L <= x_len) msk_l[i0 + q * L] = v[q];
    }
  }
  __syncthreads();
  int ymax = y_len;
#pragma unroll
  for (int t = 1; t < PDT_WAVE; t <<= 1) ymax = max(ymax, __shfl_xor(ymax, t));
  ymax = __builtin_amdgcn_readfirstlane(ymax);
  const int nsteps = ymax > 0 ? ymax + L - 1 : 0;

  // ---- the pipeline: lane b handles Y[s - b] at step s ---------------------------------------
  // A lone wave issues a dependent instruction every ~8.5 cycles and nothing else runs on its SIMD
  // (four utterances per SIMD at the bench shape, all in this wave), so the loop is as long as its
  // instruction count: the match masks of kChunk steps are looked up first (independent LDS reads,
  // batched), then kChunk recurrence steps run out of registers.
  //
  // Steps need no guard while every lane is either working or still waiting for its first column:
  // a waiting lane sees Eq = 0 (its lookups are forced to 0) and the horizontal delta 0 its waiting
  // neighbour emits, which leaves the column-0 state (Pv = ~0, Mv = 0) as it is and emits 0 again.
  // Only the last steps (the first utterance of the wave to finish, onwards) are guarded.
  constexpr int kChunk = kBitparChunk;
  unsigned Pv = 0xffffffffu, Mv = 0u;  // column 0: D[i][0] = i
  unsigned hop = 0u, hon = 0u;         // horizontal delta of this block's last row: +1 / -1 flags
  const unsigned lowmask = (1u << b) - 1u, bbit = 1u << b;
  const int jcap = y_len > 0 ? y_len - 1 : 0;
  int ymin = (sub < a.upw && n_raw < a.N) ? y_len : (1 << 30);
#pragma unroll
  for (int t = 1; t < PDT_WAVE; t <<= 1) ymin = min(ymin, __shfl_xor(ymin, t));
  ymin = __builtin_amdgcn_readfirstlane(ymin);
  const int nfree = (ymin / kChunk) * kChunk;  // steps [0, nfree): no lane has run out of columns
  const bool row16 = a.lgL == 4;               // a DPP row is one utterance: row_shr:1 feeds +1 into b = 0
  auto lookups = [&](int s0, unsigned (&eq)[kChunk]) {
    uint2 hq[kChunk];
#pragma unroll
    for (int q = 0; q < kChunk; ++q) hq[q] = yh_l[min(max(s0 + q - b, 0), jcap)];
#pragma unroll
    for (int q = 0; q < kChunk; ++q) eq[q] = msk_l[hq[q].y + (unsigned)__popc(hq[q].x & lowmask)];
#pragma unroll
    for (int q = 0; q < kChunk; ++q) eq[q] = ((hq[q].x & bbit) && s0 + q >= b) ? eq[q] : 0u;
  };
  auto step = [&](const unsigned eq0, const unsigned hp, const unsigned hn) {
    const unsigned Xv = eq0 | Mv;
    const unsigned Eq = eq0 | hn;
    const unsigned Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    unsigned Ph = Mv | ~(Xh | Pv);
    unsigned Mh = Pv & Xh;
    hop = Ph >> 31;
    hon = Mh >> 31;
    Ph = (Ph << 1) | hp;
    Mh = (Mh << 1) | hn;
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
  };
  // ROW16 form of the step: the neighbour's WHOLE Ph / Mh words travel (xp, xm; lane b = 0 of a
  // row is never written by the row shift and keeps the D[0][j] = j deltas: bit 31 of xp set, of xm
  // clear), and (Ph << 1) | (xp >> 31) is one v_alignbit -- three instructions per step less than
  // shifting the top bits out first and presetting the shift's `old` operand every step.
  unsigned xp = b == 0 ? 0x80000000u : 0u, xm = 0u, Phw = 0u, Mhw = 0u;
  auto step16 = [&](const unsigned eq0) {
    const unsigned hn = xm >> 31;
    const unsigned Xv = eq0 | Mv;
    const unsigned Eq = eq0 | hn;
    const unsigned Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    Phw = Mv | ~(Xh | Pv);
    Mhw = Pv & Xh;
    const unsigned Ph = __builtin_amdgcn_alignbit(Phw, xp, 31);
    const unsigned Mh = __builtin_amdgcn_alignbit(Mhw, xm, 31);
    Pv = Mh | ~(Xv | Ph);
    Mv = Ph & Xv;
  };
  auto ring_at = [&](const int s0) {
    return reinterpret_cast<uint4 *>(ring + ((s0 / kChunk) & 1) * (PDT_WAVE * kBitparLaneStride) + lane * kBitparLaneStride);
  };
  auto sweep = [&](auto row16_tag, auto guarded_tag, const int s_begin, const int s_end) {
    constexpr bool ROW16 = decltype(row16_tag)::value, GUARDED = decltype(guarded_tag)::value;
    for (int s0 = s_begin; s0 < s_end; s0 += kChunk) {
      __syncthreads();  // chunk s0 is in the ring (and wave 1 may fill the other buffer)
      unsigned eq[kChunk];
      {
        const uint4 *src = ring_at(s0);
#pragma unroll
        for (int q = 0; q < kChunk / 4; ++q) {
          const uint4 v = src[q];
          eq[4 * q] = v.x; eq[4 * q + 1] = v.y; eq[4 * q + 2] = v.z; eq[4 * q + 3] = v.w;
        }
      }
#pragma unroll
      for (int q = 0; q < kChunk; ++q) {
        unsigned hp = 0u, hn = 0u;
        if (ROW16) {
          xp = (unsigned)__builtin_amdgcn_update_dpp((int)xp, (int)Phw, PDT_DPP_ROW_SHR(1), 0xf, 0xf, false);
          xm = (unsigned)__builtin_amdgcn_update_dpp((int)xm, (int)Mhw, PDT_DPP_ROW_SHR(1), 0xf, 0xf, false);
        } else {
          hp = (unsigned)shr1((int)hop, 0);
          hn = (unsigned)shr1((int)hon, 0);
          if (b == 0) {  // D[0][j] = j
            hp = 1u;
            hn = 0u;
          }
        }
        const int j = s0 + q - b;
        if (!GUARDED || (unsigned)j < (unsigned)y_len) {
          if (ROW16) step16(eq[q]);
          else step(eq[q], hp, hn);
        }
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  if (wave == 1) {  // the look-ups, one chunk ahead of the recurrence (one barrier per chunk on both sides)
    for (int s0 = 0; s0 < nsteps; s0 += kChunk) {
      unsigned eq[kChunk];
      lookups(s0, eq);
      uint4 *dst = ring_at(s0);
#pragma unroll
      for (int q = 0; q < kChunk / 4; ++q) dst[q] = make_uint4(eq[4 * q], eq[4 * q + 1], eq[4 * q + 2], eq[4 * q + 3]);
      __syncthreads();
    }
    return;
  }
  if (row16) {
    sweep(T{}, F{}, 0, nfree);
    sweep(T{}, T{}, nfree, nsteps);
  } else {
    sweep(F{}, F{}, 0, nfree);
    sweep(F{}, T{}, nfree, nsteps);
  }

  // ---- distances: D[ref_len][h] = ref_len + sum_{k <= h} (Pv_k - Mv_k) (_string.py:357-405) --
  const int nvalid = min(max(x_len - 32 * b, 0), 32);
  const unsigned vm = nvalid == 32 ? 0xffffffffu : (1u << nvalid) - 1u;
  const int bs = __popc(Pv & vm) - __popc(Mv & vm);
  const int incl = wave_incl_scan_add(bs);
  const int seg = lane & ~(L - 1);
  const int before = __builtin_amdgcn_ds_bpermute((seg > 0 ? seg - 1 : 0) << 2, incl);
  int run = ref_len + incl - bs - (seg > 0 ? before : 0);
  if (valid) {
    if (b == 0) bnd[0] = (float)ref_len;
    for (int k = 0; k < nvalid; ++k) {
      run += (int)((Pv >> k) & 1u) - (int)((Mv >> k) & 1u);
      bnd[32 * b + k + 1] = (float)run;
    }
  }
  wave_sync();
  if (!valid) return;
  if (a.mode == PDT_MODE_FINAL) {
    if (b == 0)
      a.out[n * a.out_sn] = lev_finish(bnd[Heff], a.mult, a.norm, ref_len, hyp_len > 0 ? 1.0f : 0.0f);
  } else {
    const int Hout = a.H + (a.exclude_last ? 0 : 1);
    const int pad_from = hyp_len + (a.exclude_last ? 0 : 1);  // :379-386
    for (int h = b; h < Hout; h += L) {
      float v;
      if (h >= pad_from)
        v = a.padding;
      else
        v = lev_finish(bnd[h], a.mult, a.norm, ref_len, h > 0 ? 1.0f : 0.0f);
      a.out[(int64_t)h * a.out_sh + n * a.out_sn] = v;
    }
  }
}

// host side -------------------------------------------------------------------------------
static int set_lds(const void *kern, size_t smem) {
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

// LevArgs -> the three launches.  `ws` must hold plan.total bytes.
int launch_lev_bitpar(const LevArgs &la, const BitparPlan &p, void *ws, hipStream_t stream, bool classified) {
  BitparArgs a{};
  a.ref = la.ref; a.hyp = la.hyp;
  a.ref_st = la.ref_st; a.ref_sn = la.ref_sn; a.hyp_st = la.hyp_st; a.hyp_sn = la.hyp_sn;
  a.R = la.R; a.H = la.H; a.N = la.N;
  a.has_eos = la.has_eos; a.include_eos = la.include_eos; a.eos = la.eos;
  a.exclude_last = la.exclude_last; a.norm = la.norm; a.mode = la.mode;
  a.mult = la.mult; a.padding = la.padding;
  a.out = la.out; a.out_sh = la.out_sh; a.out_sn = la.out_sn;
  a.ref_lens_out = la.ref_lens_out; a.hyp_lens_out = la.hyp_lens_out; a.status = la.status;
  a.X = la.H;
  a.Y = la.R;
  a.lgL = p.lgL; a.upw = p.upw;
  unsigned char *w = reinterpret_cast<unsigned char *>(ws);
  a.lens = reinterpret_cast<int32_t *>(w + p.off_lens);
  a.yh = reinterpret_cast<uint2 *>(w + p.off_yh);
  a.msk = reinterpret_cast<uint32_t *>(w + p.off_msk);

  int rc = 0;
  if (!classified) {  // (pdt_lev_classified: the workspace holds these inputs' tables already)
    auto ck = a.X <= 8 * PDT_WAVE ? lev_classify_kernel<8> : lev_classify_kernel<16>;
    rc = set_lds(reinterpret_cast<const void *>(ck), p.lds_classify * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(ck, dim3((unsigned)((a.N + 3) / 4)), dim3(256), p.lds_classify * 4,
                       stream, a, (int)p.lds_classify);
  }
  const size_t smem = p.lds_sub * p.upw + p.lds_tail;
  const unsigned grid = (unsigned)((a.N + p.upw - 1) / p.upw);
  auto kern = lev_bitpar_kernel;
  rc = set_lds(reinterpret_cast<const void *>(kern), smem);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(128), smem, stream, a, (int)p.lds_sub, (int)(p.lds_sub * p.upw));
  return (int)hipGetLastError();
}

}  // namespace pdt
