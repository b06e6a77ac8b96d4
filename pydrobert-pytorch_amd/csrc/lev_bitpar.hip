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
// Optimal completion (unit costs, references of up to 512 tokens) has its own kernel at the end of
// this file, oc_bitpar_kernel, with the bit-vectors along the REFERENCE: a row's profile is then a
// prefix sum of the vector's +1 / -1 bits and the arg-min columns come from table look-ups.  (A form
// of the hypothesis-major pipeline above that dumped every column's (Pv, Mv) words and searched them
// for the arg-min rows measured 0.48 ms against the row-synchronous kernel's 0.51; the
// reference-major kernel runs the C2 shape in 0.24 ms.)
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
  // optimal-completion form (oc_bitpar_kernel): bit-vectors along the REFERENCE, hypothesis consumed
  int oc;
  int64_t *class_tokens;  // [N][R]  the distinct reference tokens, ascending (pdt_oc_mask's output)
  uint16_t *xcls;         // [N][X]  class of every reference position
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
  // (the token table's region also holds the presence map of small tokens, lev_classes.hpp: at
  // least kDirectWords * 8 bytes)
  p.lds_classify = align_up(Xs * 8 + std::max((Xs + 1) * 8, (size_t)kDirectWords * 8) + Ys * 2, 16);
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
  // (time-major tokens: one 128-byte line of a row holds 16 neighbouring utterances' tokens = four
  // workgroups; with the XCD-aware order those four run on ONE XCD and its L2 fetches the line once
  // -- under the dispatcher's round-robin they sat on four XCDs and HBM delivered it four times)
  const int64_t n = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (n >= a.N) return;  // waves never synchronise with each other
  unsigned char *base = smem + (size_t)wave * lds_per_wave;
  const int X = a.X > 0 ? a.X : 1, Y = a.Y > 0 ? a.Y : 1;
  // [(presence, offset) per class: X * 8] [distinct tokens (X + 1) * 8; later the packed mask
  // words] [classes of Y, 2 bytes each]
  uint2 *po = reinterpret_cast<uint2 *>(base);
  int64_t *ctok = reinterpret_cast<int64_t *>(base + (size_t)X * 8);
  unsigned *msk = reinterpret_cast<unsigned *>(ctok);
  uint2 *pmap = reinterpret_cast<uint2 *>(ctok);  // (instead of the token table: one or the other)
  short *yc = reinterpret_cast<short *>(ctok + max(X + 1, kDirectWords));

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
  // bit-vectors along the hypothesis (only its first Heff tokens matter), reference consumed;
  // the other way round for optimal completion, whose row minima run along the reference
  const bool oc = a.oc != 0;
  const int64_t *x = oc ? a.ref : a.hyp, *y = oc ? a.hyp : a.ref;
  const int64_t x_st = oc ? a.ref_st : a.hyp_st, y_st = oc ? a.hyp_st : a.ref_st;
  const int64_t xoff = oc ? roff : hoff, yoff = oc ? hoff : roff;
  const int x_len = oc ? ref_len : Heff, y_len = oc ? Heff : ref_len;

  // ---- distinct tokens of X in ascending order (lev_classes.hpp) ---------------------------
  // (vocabulary indices below kDirectBits: ranks from a presence map, no sort -- lev_classes.hpp;
  // anything else: the sorted table and binary searches.  The classes are the same numbers.)
  int64_t xt[NR];
  load_sequence<NR>(x, x_len, x_st, xoff, xt);
  const bool direct = tokens_are_small<NR>(x_len, xt);
  int U, lgP = 0;
  int xc[NR];  // classes of X[lane + 64 q]
  if (direct) {
    U = presence_map<NR>(x_len, xt, pmap);
    if (oc) tokens_from_map(pmap, a.class_tokens + n * (int64_t)a.R);
    classes_from_map<NR>(pmap, xt, xc);
#pragma unroll
    for (int q = 0; q < NR; ++q) xc[q] = lane + q * PDT_WAVE < x_len ? xc[q] : -1;
  } else {
    U = distinct_sorted_regs<NR>(x_len, xt, ctok);
    wave_sync();
    if (oc)
      for (int k = lane; k < U; k += PDT_WAVE) a.class_tokens[n * (int64_t)a.R + k] = ctok[k];
    lgP = search_depth(U);
    classes_of<NR>(ctok, U, lgP, xt, xc);
  }
  for (int k = lane; k < U; k += PDT_WAVE) po[k] = make_uint2(0u, 0u);
  if (oc) {
#pragma unroll
    for (int q = 0; q < NR; ++q)
      if (lane + q * PDT_WAVE < x_len) a.xcls[n * (int64_t)X + lane + q * PDT_WAVE] = (uint16_t)xc[q];
  }
  for (int j0 = 0; j0 < y_len; j0 += 8 * PDT_WAVE) {
    int64_t yt[8];
    int c[8];
    load_tokens(y, y_len, y_st, yoff, j0, 0, yt);
    if (direct) classes_from_map<8>(pmap, yt, c);
    else classes_of<8>(ctok, U, lgP, yt, c);
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

// ---- optimal completion: the arg-min sets of every row, bit-parallel --------------------------
// (reference _string.py:271-278, :333-354: after hypothesis token h, the set of reference tokens
// ref[c] over the columns c < ref_len where D[h][c] is the row minimum.)
//
// Bit-vectors run along the REFERENCE here: after h hypothesis tokens Pv / Mv hold
// D[h][c] - D[h][c-1] for every column, so the row's profile relative to D[h][0] = h is a prefix sum
// of +1 / -1 bits and the row minimum needs no DP value at all.  16 lanes (one DPP row) per
// utterance, four utterances per wave, lane b owns columns 32 b + 1 .. 32 b + 32, every lane on the
// SAME row:
//   * the 512-bit addition of Myers' step is 16 word additions whose carries are resolved on the
//     scalar unit -- generate / propagate lane masks (v_add_co's carry mask, one v_cmp), one 64-bit
//     add (Gm << 1) + Pm, xor, and a v_addc_co that takes the result as its carry-in mask;
//   * the shifts take the neighbour's word through a row_shr:1 and one v_alignbit;
//   * a block's (total, min prefix, arg-min bits) come from a 256-entry table over 4 columns at a
//     time (index = plus nibble | minus nibble << 4), block starts from a 4-step row scan, the row
//     minimum from a 4-step row all-reduce;
//   * the arg-min bits are permuted so that neighbouring columns sit in different lanes (oc_spread)
//     and turned into class bits by LDS ORs; the row's W words leave with one exchange each.
// Utterances whose hypothesis has ended write empty sets.
//
// What bounds it (profiles/r03_oc_*): the four utterances of a SIMD are one wave's worth of
// lanes, and a lone wave issues an instruction every ~5.5 cycles.  One wave doing everything took
// 0.375 ms (0.46 before the rows of a pass were processed phase by phase); splitting the rows'
// independent part over consumer waves (below) 0.24 ms, at which point the SIMDs' vector issue is
// ~80 % busy (profiles/tools/micro/valu_cost.hip: with several waves per SIMD most integer
// instructions other than add / sub / and / or / xor / right shifts issue at half rate).
#ifndef PDT_OC_CHUNK
#define PDT_OC_CHUNK 4
#endif
#ifndef PDT_OC_TABLE32
#define PDT_OC_TABLE32 1  // block-minimum table with byte fields (SDWA decode); 0: packed 16-bit entries
#endif
#ifndef PDT_OC_SLOTS
#define PDT_OC_SLOTS(nc) (2 * (nc))  // ring slots (passes in flight) per workgroup
#endif
#ifndef PDT_OC_CONSUMERS
#define PDT_OC_CONSUMERS 3  // consumer waves per workgroup of oc_bitpar_kernel
#endif
constexpr int kOcChunk = PDT_OC_CHUNK;  // rows per pass of oc_bitpar_kernel

struct OcBitArgs {
  int N, X, Y, W, Hout, exclude_last;
  const int32_t *lens;
  const uint2 *yh;
  const uint32_t *msk;
  const uint16_t *xcls;
  uint32_t *bitmask;
  int32_t *max_count;
};

#define PDT_DPP_QUAD_XOR1 0xB1
#define PDT_DPP_QUAD_XOR2 0x4E
#define PDT_DPP_ROW_HALF_MIRROR 0x141
#define PDT_DPP_ROW_MIRROR 0x140

// entry: (sum + G) | (min prefix + G) << 4 | arg-min bits << 8 over G columns
template <int G>
__device__ __forceinline__ void oc_build_table(uint16_t *tab) {
  for (int idx = (int)threadIdx.x; idx < (1 << (2 * G)); idx += (int)blockDim.x) {
    const int p = idx & ((1 << G) - 1), m = idx >> G;
    int run = 0, mn = 99, am = 0;
    for (int j = 0; j < G; ++j) {
      run += ((p >> j) & 1) - ((m >> j) & 1);
      if (run < mn) {
        mn = run;
        am = 1 << j;
      } else if (run == mn) {
        am |= 1 << j;
      }
    }
    tab[idx] = (uint16_t)((run + G) | ((mn + G) << 4) | (am << 8));
  }
}

// the same table with byte fields, 32 bits per entry: byte 0 = sum (signed), byte 1 = min prefix
// (signed), byte 2 = arg-min bits -- SDWA operands then fold the field extraction (and the sign
// extension) into the additions and the shift of the decode: 5 instructions per nibble instead of 8
__device__ __forceinline__ void oc_build_table32(unsigned *tab) {
  for (int idx = (int)threadIdx.x; idx < 256; idx += (int)blockDim.x) {
    const int p = idx & 15, m = idx >> 4;
    int run = 0, mn = 99, am = 0;
    for (int j = 0; j < 4; ++j) {
      run += ((p >> j) & 1) - ((m >> j) & 1);
      if (run < mn) {
        mn = run;
        am = 1 << j;
      } else if (run == mn) {
        am |= 1 << j;
      }
    }
    tab[idx] = (unsigned)(run & 0xff) | ((unsigned)(mn & 0xff) << 8) | ((unsigned)am << 16);
  }
}

// total of the block's deltas, its minimum prefix (over the prefixes of length >= 1) and the
// columns that reach it
// byte `I` of z, times two (one SDWA shift: the table's byte offset)
template <int I>
__device__ __forceinline__ unsigned byte_times2(const unsigned z, const unsigned one) {
  unsigned r;
  if (I == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(one), "v"(z));
  if (I == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(one), "v"(z));
  if (I == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(one), "v"(z));
  if (I == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(one), "v"(z));
  return r;
}

// the eight table entries of a block: index = plus nibble | minus nibble << 4, a byte of one of two
// interleaved words
__device__ __forceinline__ void oc_table_reads(const uint16_t *tab, const unsigned pv, const unsigned mv,
                                               unsigned (&e)[8]) {
  const unsigned ze = (pv & 0x0f0f0f0fu) | ((mv << 4) & 0xf0f0f0f0u);  // nibbles 0, 2, 4, 6
  const unsigned zo = ((pv >> 4) & 0x0f0f0f0fu) | (mv & 0xf0f0f0f0u);  // nibbles 1, 3, 5, 7
  unsigned one = 1u;
  asm volatile("" : "+v"(one));
  const unsigned char *t = reinterpret_cast<const unsigned char *>(tab);
  e[0] = *reinterpret_cast<const uint16_t *>(t + byte_times2<0>(ze, one));
  e[1] = *reinterpret_cast<const uint16_t *>(t + byte_times2<0>(zo, one));
  e[2] = *reinterpret_cast<const uint16_t *>(t + byte_times2<1>(ze, one));
  e[3] = *reinterpret_cast<const uint16_t *>(t + byte_times2<1>(zo, one));
  e[4] = *reinterpret_cast<const uint16_t *>(t + byte_times2<2>(ze, one));
  e[5] = *reinterpret_cast<const uint16_t *>(t + byte_times2<2>(zo, one));
  e[6] = *reinterpret_cast<const uint16_t *>(t + byte_times2<3>(ze, one));
  e[7] = *reinterpret_cast<const uint16_t *>(t + byte_times2<3>(zo, one));
}

// total of the block's deltas, its minimum prefix (over the prefixes of length >= 1) and the
// columns that reach it
__device__ __forceinline__ void oc_block_decode(const unsigned (&e)[8], int &total, int &best, unsigned &am) {
  constexpr int G = 4, NG = 8;
  int run = 0, cand[NG];
  best = 1 << 20;
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    cand[k] = run + (int)((e[k] >> 4) & 15u) - G * (k + 1);
    best = min(best, cand[k]);
    run += (int)(e[k] & 15u);
  }
  am = 0u;
#pragma unroll
  for (int k = 0; k < NG; ++k) am |= cand[k] == best ? (e[k] >> 8) << (G * k) : 0u;
  total = run - G * NG;
}

template <int I>
__device__ __forceinline__ unsigned byte_times4(const unsigned z, const unsigned two) {
  unsigned r;
  if (I == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(z));
  if (I == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(two), "v"(z));
  if (I == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(two), "v"(z));
  if (I == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(two), "v"(z));
  return r;
}
__device__ __forceinline__ void oc_table_reads32(const unsigned *tab, const unsigned pv, const unsigned mv,
                                                 unsigned (&e)[8]) {
  const unsigned ze = (pv & 0x0f0f0f0fu) | ((mv << 4) & 0xf0f0f0f0u);  // nibbles 0, 2, 4, 6
  const unsigned zo = ((pv >> 4) & 0x0f0f0f0fu) | (mv & 0xf0f0f0f0u);  // nibbles 1, 3, 5, 7
  unsigned two = 2u;
  asm volatile("" : "+v"(two));
  const unsigned char *t = reinterpret_cast<const unsigned char *>(tab);
  e[0] = *reinterpret_cast<const unsigned *>(t + byte_times4<0>(ze, two));
  e[1] = *reinterpret_cast<const unsigned *>(t + byte_times4<0>(zo, two));
  e[2] = *reinterpret_cast<const unsigned *>(t + byte_times4<1>(ze, two));
  e[3] = *reinterpret_cast<const unsigned *>(t + byte_times4<1>(zo, two));
  e[4] = *reinterpret_cast<const unsigned *>(t + byte_times4<2>(ze, two));
  e[5] = *reinterpret_cast<const unsigned *>(t + byte_times4<2>(zo, two));
  e[6] = *reinterpret_cast<const unsigned *>(t + byte_times4<3>(ze, two));
  e[7] = *reinterpret_cast<const unsigned *>(t + byte_times4<3>(zo, two));
}
// R rows at once, their chains interleaved instruction by instruction (a dependent SDWA instruction
// right behind its producer costs wait states: 20 s_nop per row when the rows were decoded one after
// the other)
template <int R>
__device__ __forceinline__ void oc_block_decode32(const unsigned (&e)[R][8], int (&total)[R], int (&best)[R],
                                                  unsigned (&am)[R]) {
  int run[R], cand[R][8];
#pragma unroll
  for (int r = 0; r < R; ++r) run[r] = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(cand[r][k]) : "v"(run[r]), "v"(e[r][k]));
#pragma unroll
    for (int r = 0; r < R; ++r)
      asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(run[r]) : "v"(run[r]), "v"(e[r][k]));
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    best[r] = min(min(min(cand[r][0], cand[r][1]), min(cand[r][2], cand[r][3])),
                  min(min(cand[r][4], cand[r][5]), min(cand[r][6], cand[r][7])));
    am[r] = 0u;
    total[r] = run[r];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    unsigned t[R];
    const unsigned sh = 4u * k;  // (a scalar operand: no vector move per nibble)
#pragma unroll
    for (int r = 0; r < R; ++r)
      asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(t[r]) : "s"(sh), "v"(e[r][k]));
#pragma unroll
    for (int r = 0; r < R; ++r) am[r] |= cand[r][k] == best[r] ? t[r] : 0u;
  }
}

// min over the 16 lanes of a DPP row, in every lane (written out: left to the compiler every
// stage is two moves, the DPP move and the v_min)
__device__ __forceinline__ int row_all_min(int v) {
  asm volatile(
      "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}

// The arg-min columns of a row come in runs (a dozen neighbouring columns at the bench shape), i.e.
// in ONE lane's word, and turning a bit into a class bit is a dependent LDS read + LDS OR.  Four
// exchange stages (partner lanes l^8, l^7, l^2, l^1 inside the DPP row; keep half of the bits, take
// the partner's other half rotated by the stage's distance) permute the 512 bits of an utterance so
// that any 16 neighbouring columns end up in 16 different lanes.  Which column a (lane, bit) pair
// holds afterwards is found once per kernel by sending the nine bit-planes of the column index
// through the same network; the class table is staged in that order.
struct OcSpread {
  unsigned keep[4];  // bits this lane keeps at each stage
  unsigned amt[4];   // v_alignbit shift that rotates the partner's word the right way
};
__device__ __forceinline__ OcSpread oc_spread_setup(const int b) {
  OcSpread sp;
  const unsigned clear[4] = {0x00ff00ffu, 0x0f0f0f0fu, 0x33333333u, 0x55555555u};  // bit d of the position clear
  const int dist[4] = {8, 4, 2, 1};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const bool hi = (b & dist[s]) != 0;
    sp.keep[s] = hi ? ~clear[s] : clear[s];
    sp.amt[s] = hi ? (unsigned)dist[s] : (unsigned)(32 - dist[s]);  // alignbit(x, x, k) rotates right by k
  }
  return sp;
}
__device__ __forceinline__ unsigned oc_spread(const OcSpread &sp, unsigned x) {
  unsigned p, t;
  auto bfi = [](const unsigned mask, const unsigned a, const unsigned b) {  // (mask & a) | (~mask & b)
    unsigned r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
  };
  p = (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x128 /* row_ror:8 */, 0xf, 0xf, true);
  t = __builtin_amdgcn_alignbit(p, p, sp.amt[0]);
  x = bfi(sp.keep[0], x, t);
  p = (unsigned)__builtin_amdgcn_mov_dpp((int)x, PDT_DPP_ROW_HALF_MIRROR, 0xf, 0xf, true);
  t = __builtin_amdgcn_alignbit(p, p, sp.amt[1]);
  x = bfi(sp.keep[1], x, t);
  p = (unsigned)__builtin_amdgcn_mov_dpp((int)x, PDT_DPP_QUAD_XOR2, 0xf, 0xf, true);
  t = __builtin_amdgcn_alignbit(p, p, sp.amt[2]);
  x = bfi(sp.keep[2], x, t);
  p = (unsigned)__builtin_amdgcn_mov_dpp((int)x, PDT_DPP_QUAD_XOR1, 0xf, 0xf, true);
  t = __builtin_amdgcn_alignbit(p, p, sp.amt[3]);
  x = bfi(sp.keep[3], x, t);
  return x;
}

// Workgroup = four utterances (one per DPP row of every wave) and 1 + NC waves.  The recurrence is
// the only part of a row that depends on the row before and it is a tenth of the row's
// instructions, so ONE wave (the producer) runs it and leaves each row's masked (Pv, Mv) words in an
// LDS ring, a pass of kOcChunk rows per slot; NC consumer waves take passes in turn and do the rest
// (block minima, row minimum, spreading, class bits, the store).  A lone wave issues an instruction
// every ~5.5 cycles; with the waves of four such workgroups on a CU every SIMD has several to pick
// from.  Roles rotate with the workgroup index so that producers do not all land on one SIMD.
struct OcLds {
  size_t flags, ring, bm, sub, total;  // byte offsets; sub = first utterance's tables
  size_t per_sub;
};
static __host__ __device__ inline OcLds oc_lds(const int X, const int NC) {
  OcLds l;
  const size_t Xs = (size_t)(X > 0 ? X : 1);
  const int S = PDT_OC_SLOTS(NC);
  l.flags = 1024;                                          // after the 256-entry table (room for 32-bit entries)
  l.ring = l.flags + 256;                                  // ready[S], done[S]
  l.bm = l.ring + (size_t)S * kOcChunk * PDT_WAVE * 8;     // a slot: kOcChunk rows of (Pv, Mv) per lane
  l.sub = l.bm + (size_t)NC * kOcChunk * PDT_WAVE * 4;     // per consumer: kOcChunk rows of 16 words per utterance
  l.per_sub = ((Xs + 1) * 4 + 15) / 16 * 16 + 256 + 512 * 2;  // match words; 2 x 16 look-ups; classes in spread order
  l.total = l.sub + 4 * l.per_sub;
  return l;
}

template <int NC>
__global__ void __launch_bounds__(64 * (NC + 1)) oc_bitpar_kernel(const OcBitArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int S = PDT_OC_SLOTS(NC), kChunk = kOcChunk, NG = 8;
#if PDT_OC_TABLE32
  unsigned *tab = reinterpret_cast<unsigned *>(smem);
  oc_build_table32(tab);
#else
  uint16_t *tab = reinterpret_cast<uint16_t *>(smem);
  oc_build_table<4>(tab);
#endif
  const OcLds L = oc_lds(a.X, NC);
  int *ready = reinterpret_cast<int *>(smem + L.flags), *done = ready + S;
  uint2 *ring = reinterpret_cast<uint2 *>(smem + L.ring);
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int role = (wave + (int)blockIdx.x) % (NC + 1);  // 0: producer, 1 .. NC: consumers
  const int q = lane >> 4, b = lane & 15;
  const int64_t n_raw = (int64_t)blockIdx.x * 4 + q;
  const bool valid = n_raw < a.N;
  const int64_t n = valid ? n_raw : (int64_t)a.N - 1;
  const int X = a.X > 0 ? a.X : 1, Y = a.Y > 0 ? a.Y : 1;
  unsigned char *base = smem + L.sub + (size_t)q * L.per_sub;
  unsigned *msk_l = reinterpret_cast<unsigned *>(base);
  uint16_t *xc_l = reinterpret_cast<uint16_t *>(base + L.per_sub - 1024);

  const int ref_len = valid ? a.lens[2 * n] : 0, hyp_len = valid ? a.lens[2 * n + 1] : 0;
  int Heff = a.exclude_last ? hyp_len - 1 : hyp_len;
  if (Heff < 0) Heff = 0;
  if (threadIdx.x < 2 * S) ready[threadIdx.x] = 0;
  const OcSpread sp = oc_spread_setup(b);
  const uint16_t *csrc = a.xcls + n * (int64_t)X;
  const int rank0 = ref_len > 0 ? (int)csrc[0] : 0;  // class of ref[0]
  if (role == 0) {  // the match words of this utterance: 16 lanes, eight loads in flight each
    const unsigned *msrc = a.msk + n * (int64_t)(X + 1);
    for (int i0 = b; i0 <= ref_len; i0 += 8 * 16) {
      unsigned v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = i0 + r * 16 <= ref_len ? msrc[i0 + r * 16] : 0u;
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (i0 + r * 16 <= ref_len) msk_l[i0 + r * 16] = v[r];
    }
  } else {
    // xc_l[32 b + t] = class of ref[c] for the column c that the network leaves in bit t of lane b
    unsigned *bm0 = reinterpret_cast<unsigned *>(smem + L.bm) + (size_t)(role - 1) * kChunk * PDT_WAVE;
    for (int r = 0; r < kChunk; ++r) bm0[r * PDT_WAVE + lane] = 0u;
    unsigned plane[9];
    const unsigned low[5] = {0xaaaaaaaau, 0xccccccccu, 0xf0f0f0f0u, 0xff00ff00u, 0xffff0000u};
#pragma unroll
    for (int k = 0; k < 9; ++k) plane[k] = oc_spread(sp, k < 5 ? low[k] : (((b >> (k - 5)) & 1) ? 0xffffffffu : 0u));
    for (int t0 = 8 * (role - 1); t0 < 32; t0 += 8 * NC) {
      uint16_t v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int c = 1;  // column = (bit index of the unspread layout) + 1
#pragma unroll
        for (int k = 0; k < 9; ++k) c += (int)((plane[k] >> (t0 + r)) & 1u) << k;
        v[r] = c < ref_len ? csrc[c] : (uint16_t)0;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) xc_l[32 * b + t0 + r] = v[r];
    }
  }
  __syncthreads();  // (the only one: from here on the waves meet through the ring's flags)

  const int W = a.W;
  // columns of this block that exist (<= ref_len) and those that have a next token (< ref_len)
  const int nvalid = min(max(ref_len - 32 * b, 0), 32), nnext = min(max(ref_len - 1 - 32 * b, 0), 32);
  const unsigned vmask = nvalid == 32 ? 0xffffffffu : (1u << nvalid) - 1u;
  const unsigned nmask = nnext == 32 ? 0xffffffffu : (1u << nnext) - 1u;
  int hmax = Heff;
#pragma unroll
  for (int t = 16; t < PDT_WAVE; t <<= 1) hmax = max(hmax, __shfl_xor(hmax, t));
  hmax = __builtin_amdgcn_readfirstlane(hmax);
  const int nchunks = (hmax + kChunk - 1) / kChunk;
  const int64_t row_stride = (int64_t)a.N * W;
  uint32_t *out_row = a.bitmask + n * (int64_t)W + b;  // row 0 of this lane's word

  if (role == 0) {
    // ---- producer: Myers' step on the 512-bit column, four utterances side by side ------------
    if (valid && b < W) {  // h = 0: only column 0 (:271-278)
      unsigned w = 0u;
      if (ref_len > 0 && (rank0 >> 5) == b) w = 1u << (rank0 & 31);
      out_row[0] = w;
    }
    // rows nobody in the workgroup reaches (`& not_done`, :334)
    for (int h = hmax + 1; h < a.Hout; ++h)
      if (valid && b < W) out_row[h * row_stride] = 0u;
    const unsigned lowmask = (1u << b) - 1u, bbit = 1u << b;
    const int jcap = Heff > 0 ? Heff - 1 : 0;
    const uint2 *ysrc = a.yh + n * (int64_t)Y;
    unsigned Pv = 0xffffffffu, Mv = 0u;  // row 0: D[0][c] = c
    unsigned xp = b == 0 ? 0x80000000u : 0u, xm = 0u;  // (lane 0 of a row keeps D[h][0] - D[h-1][0] = +1)
    u64 notop = 0x7fff7fff7fff7fffull;  // carries stay inside an utterance's 16 lanes
    asm volatile("" : "+s"(notop));     // (in a register pair: the literal would split every AND in two)
    // (presence, offset) of the hypothesis tokens' classes: lane b fetches row 16 k + b a block of 16
    // rows ahead (a load per pass would cost its whole latency every pass) and leaves it in LDS
    static_assert(16 % kChunk == 0, "a block of look-ups is a whole number of passes");
    uint2 *ybuf = reinterpret_cast<uint2 *>(base + L.per_sub - 1024 - 256);
    uint2 pre = Heff > 0 ? ysrc[min(b, jcap)] : make_uint2(0u, 0u);
    for (int i = 0; i < nchunks; ++i) {
      if ((i * kChunk) % 16 == 0) {
        const int blk = (i * kChunk) >> 4;
        ybuf[(blk & 1) * 16 + b] = pre;
        pre = Heff > 0 ? ysrc[min((blk + 1) * 16 + b, jcap)] : make_uint2(0u, 0u);
      }
      uint2 hq[kChunk];
#pragma unroll
      for (int r = 0; r < kChunk; ++r) hq[r] = ybuf[(i * kChunk + r) & 31];
      unsigned eq[kChunk];
#pragma unroll
      for (int r = 0; r < kChunk; ++r) eq[r] = msk_l[hq[r].y + (unsigned)__popc(hq[r].x & lowmask)];
#pragma unroll
      for (int r = 0; r < kChunk; ++r) eq[r] = (hq[r].x & bbit) ? eq[r] : 0u;
      const int slot = i % S;
      if (i >= S)
        while (__hip_atomic_load(&done[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < i - S + 1)
          __builtin_amdgcn_s_sleep(2);
      uint2 *dst = ring + (size_t)slot * kChunk * PDT_WAVE + lane;
#pragma unroll
      for (int r = 0; r < kChunk; ++r) {
        const unsigned Eq = eq[r];
        const unsigned Xv = Eq | Mv;
        const unsigned A = Eq & Pv;
        unsigned sum;
        u64 Gm;
        asm volatile("v_add_co_u32 %0, %1, %2, %3" : "=v"(sum), "=s"(Gm) : "v"(A), "v"(Pv));
        Gm &= notop;
        const u64 Pm = __ballot(sum == 0xffffffffu) & notop;
        const u64 C = ((Gm << 1) + Pm) ^ Pm;
        asm volatile("v_addc_co_u32 %0, %1, 0, %2, %3" : "=v"(sum), "=s"(Gm) : "v"(sum), "s"(C));
        const unsigned Xh = (sum ^ Pv) | Eq;
        const unsigned Phw = Mv | ~(Xh | Pv);
        const unsigned Mhw = Pv & Xh;
        xp = (unsigned)__builtin_amdgcn_update_dpp((int)xp, (int)Phw, PDT_DPP_ROW_SHR(1), 0xf, 0xf, false);
        xm = (unsigned)__builtin_amdgcn_update_dpp((int)xm, (int)Mhw, PDT_DPP_ROW_SHR(1), 0xf, 0xf, false);
        const unsigned Ph = __builtin_amdgcn_alignbit(Phw, xp, 31);
        const unsigned Mh = __builtin_amdgcn_alignbit(Mhw, xm, 31);
        Pv = Mh | ~(Xv | Ph);
        Mv = Ph & Xv;
        dst[r * PDT_WAVE] = make_uint2(Pv & vmask, Mv & vmask);
      }
      if (lane == 0) __hip_atomic_store(&ready[slot], i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return;
  }

  // ---- consumers: passes role - 1, role - 1 + NC, ... ---------------------------------------------
  unsigned *bm = reinterpret_cast<unsigned *>(smem + L.bm) + (size_t)(role - 1) * kChunk * PDT_WAVE + 16 * q;
  int max_cnt = (role == 1 && ref_len > 0) ? 1 : 0;  // (row 0)
  for (int i = role - 1; i < nchunks; i += NC) {
    const int slot = i % S, h0 = i * kChunk;
    while (__hip_atomic_load(&ready[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < i + 1)
      __builtin_amdgcn_s_sleep(2);
    const uint2 *src = ring + (size_t)slot * kChunk * PDT_WAVE + lane;
    uint2 pm[kChunk];
#pragma unroll
    for (int r = 0; r < kChunk; ++r) pm[r] = src[r * PDT_WAVE];
    unsigned e[kChunk][NG];
#pragma unroll
#if PDT_OC_TABLE32
    for (int r = 0; r < kChunk; ++r) oc_table_reads32(tab, pm[r].x, pm[r].y, e[r]);
#else
    for (int r = 0; r < kChunk; ++r) oc_table_reads(tab, pm[r].x, pm[r].y, e[r]);
#endif
    if (lane == 0) __hip_atomic_store(&done[slot], i + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned rest[kChunk], or0[kChunk], or1[kChunk];
    int c0[kChunk], c1[kChunk];
    bool zero_min[kChunk];
#if PDT_OC_TABLE32
    int total_r[kChunk], best_r[kChunk];
    unsigned am_r[kChunk];
    oc_block_decode32<kChunk>(e, total_r, best_r, am_r);
#endif
#pragma unroll
    for (int r = 0; r < kChunk; ++r) {
      const bool active = h0 + r + 1 <= Heff;
      int total, best;
      unsigned am;
#if PDT_OC_TABLE32
      total = total_r[r];
      best = best_r[r];
      am = am_r[r];
#else
      oc_block_decode(e[r], total, best, am);
#endif
      int incl = total;  // D[h][32 b + 32] - D[h][0]
      incl += dpp_or<PDT_DPP_ROW_SHR(1)>(incl, 0);
      incl += dpp_or<PDT_DPP_ROW_SHR(2)>(incl, 0);
      incl += dpp_or<PDT_DPP_ROW_SHR(4), 0xf, 0xe>(incl, 0);
      incl += dpp_or<PDT_DPP_ROW_SHR(8), 0xf, 0xc>(incl, 0);
      const int cand = incl - total + best;
      const int m = row_all_min(min(cand, 0));  // (0: column 0)
      zero_min[r] = active && m == 0;
      unsigned bits = (active && cand == m) ? am & nmask : 0u;  // :334 and the c < ref_len cut of :349-354
      bits = oc_spread(sp, bits);
      // the first two bits of a lane go the short way (a lane rarely holds more)
      const int j0 = bits ? __builtin_ctz(bits) : 0;
      or0[r] = bits ? 1u : 0u;
      bits &= bits - 1u;
      const int j1 = bits ? __builtin_ctz(bits) : 0;
      or1[r] = bits ? 1u : 0u;
      bits &= bits - 1u;
      rest[r] = bits;
      c0[r] = xc_l[32 * b + j0];
      c1[r] = xc_l[32 * b + j1];
    }
#pragma unroll
    for (int r = 0; r < kChunk; ++r) {  // (an OR of 0 where there is no bit: no branches)
      unsigned *bmr = bm + PDT_WAVE * r;
      atomicOr(&bmr[c0[r] >> 5], or0[r] << (c0[r] & 31));
      atomicOr(&bmr[c1[r] >> 5], or1[r] << (c1[r] & 31));
      if (b == 0) atomicOr(&bmr[rank0 >> 5], (zero_min[r] && ref_len > 0) ? 1u << (rank0 & 31) : 0u);
    }
    unsigned any_rest = 0u;
#pragma unroll
    for (int r = 0; r < kChunk; ++r) any_rest |= rest[r];
    if (__ballot(any_rest != 0u)) {
#pragma unroll
      for (int r = 0; r < kChunk; ++r) {
        unsigned bits = rest[r];
        while (__ballot(bits != 0u)) {
          if (bits) {
            const int cls = xc_l[32 * b + __builtin_ctz(bits)];
            bits &= bits - 1u;
            atomicOr(&bm[PDT_WAVE * r + (cls >> 5)], 1u << (cls & 31));
          }
        }
      }
    }
    // (the LDS serves one wave's instructions in order: the exchanges see every lane's OR)
    __builtin_amdgcn_wave_barrier();
    unsigned w[kChunk];
#pragma unroll
    for (int r = 0; r < kChunk; ++r)
      w[r] = __hip_atomic_exchange(&bm[PDT_WAVE * r + b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __builtin_amdgcn_wave_barrier();
    if (valid && b < W) {
#pragma unroll
      for (int r = 0; r < kChunk; ++r)
        if (h0 + r + 1 <= hmax && h0 + r + 1 < a.Hout) out_row[(h0 + r + 1) * row_stride] = w[r];
    }
#pragma unroll
    for (int r = 0; r < kChunk; ++r) {
      int cnt = __popc(w[r]);
      cnt += dpp_or<PDT_DPP_QUAD_XOR1>(cnt, 0);
      cnt += dpp_or<PDT_DPP_QUAD_XOR2>(cnt, 0);
      cnt += dpp_or<PDT_DPP_ROW_HALF_MIRROR>(cnt, 0);
      cnt += dpp_or<PDT_DPP_ROW_MIRROR>(cnt, 0);
      max_cnt = max(max_cnt, cnt);
    }
  }
  if (valid && b == 0 && a.max_count && max_cnt > 0) atomicMax(a.max_count, max_cnt);
}

constexpr int64_t kOcBitparMaxR = 512;  // 16 lanes of 32 columns

int64_t oc_bitpar_workspace_bytes(int64_t R, int64_t H, int64_t N) {
  if (R > kOcBitparMaxR || H < 0 || N <= 0) return 0;
  const BitparPlan p = plan_bitpar(R, H, N);
  if (!p.ok) return 0;
  return (int64_t)(p.total + align_up((size_t)N * (size_t)(R > 0 ? R : 1) * 2, 256));
}

// Unit costs only.  Returns -1 when the shape is not served (the caller falls back to
// lev_rowsync.hip), else the launch status.
int launch_oc_mask_bitpar(const LevArgs &la, void *ws, int64_t ws_bytes, hipStream_t stream) {
  if (la.R > kOcBitparMaxR || !ws) return -1;
  const BitparPlan p = plan_bitpar(la.R, la.H, la.N);
  const int64_t need = oc_bitpar_workspace_bytes(la.R, la.H, la.N);
  if (!p.ok || need == 0 || need > ws_bytes) return -1;
  BitparArgs a{};
  a.ref = la.ref; a.hyp = la.hyp;
  a.ref_st = la.ref_st; a.ref_sn = la.ref_sn; a.hyp_st = la.hyp_st; a.hyp_sn = la.hyp_sn;
  a.R = la.R; a.H = la.H; a.N = la.N;
  a.has_eos = la.has_eos; a.include_eos = la.include_eos; a.eos = la.eos;
  a.exclude_last = la.exclude_last; a.norm = 0; a.mode = -1;
  a.status = la.status;
  a.X = la.R;
  a.Y = la.H;
  a.lgL = 4; a.upw = 4;
  unsigned char *w = reinterpret_cast<unsigned char *>(ws);
  a.lens = reinterpret_cast<int32_t *>(w + p.off_lens);
  a.yh = reinterpret_cast<uint2 *>(w + p.off_yh);
  a.msk = reinterpret_cast<uint32_t *>(w + p.off_msk);
  a.oc = 1;
  a.class_tokens = la.class_tokens;
  a.xcls = reinterpret_cast<uint16_t *>(w + p.total);
  auto ck = lev_classify_kernel<8>;
  int rc = set_lds(reinterpret_cast<const void *>(ck), p.lds_classify * 4);
  if (rc) return rc;
  hipLaunchKernelGGL(ck, dim3((unsigned)((a.N + 3) / 4)), dim3(256), p.lds_classify * 4, stream, a,
                     (int)p.lds_classify);

  OcBitArgs o{};
  o.N = la.N; o.X = la.R; o.Y = la.H; o.W = la.W;
  o.Hout = la.H + (la.exclude_last ? 0 : 1);
  o.exclude_last = la.exclude_last;
  o.lens = a.lens; o.yh = a.yh; o.msk = a.msk; o.xcls = a.xcls;
  o.bitmask = la.bitmask; o.max_count = la.max_count;
  constexpr int NC = PDT_OC_CONSUMERS;
  const OcLds L = oc_lds(o.X, NC);
  auto kern = oc_bitpar_kernel<NC>;
  rc = set_lds(reinterpret_cast<const void *>(kern), L.total);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)((la.N + 3) / 4)), dim3(64 * (NC + 1)), L.total, stream, o);
  return (int)hipGetLastError();
}

}  // namespace pdt
