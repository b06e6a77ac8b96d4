// Token classes for the Levenshtein kernels: the distinct tokens of one sequence, sorted in
// registers, and class (= rank) look-ups against that table.  Used by lev_bitpar.hip (classify
// kernel) and lev_rowsync.hip (optimal completion emits class bitmasks in ascending token order,
// reference _string.py:503-514).
#pragma once
#include "lev_common.hpp"
#include "wave_select.hpp"

namespace pdt {

// Classes (ranks in the sorted table of U distinct tokens, -1 = absent) of NQ tokens per lane:
// NQ branch-free binary searches of lgP steps advance together, so their LDS reads overlap.
template <int NQ>
__device__ __forceinline__ void classes_of(const int64_t *tab, const int U, const int lgP,
                                           const int64_t (&v)[NQ], int (&cls)[NQ]) {
  int pos[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) pos[q] = 0;
  if (U > 0) {
    for (int st = 1 << (lgP - 1); st >= 1; st >>= 1) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int k = pos[q] + st - 1;
        const int64_t t = tab[min(k, U - 1)];
        if (k < U && t < v[q]) pos[q] += st;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) cls[q] = (pos[q] < U && tab[pos[q]] == v[q]) ? pos[q] : -1;
}

// Sorts NR * 64 keys held as k[r] of lane l = element r * 64 + l, DESCENDING in that order
// (wave_select.hpp sorts the 64 keys of one register; registers are then merged pairwise: the
// lower run against the lane- and register-reversed upper run gives two bitonic halves, cleaned by
// register-to-register exchanges and one 6-stage merge inside every register).
template <int NR>
__device__ __forceinline__ void sort_regs_desc(u64 (&k)[NR]) {
#pragma unroll
  for (int r = 0; r < NR; ++r) k[r] = wave_sort_desc<u64>(k[r]);
#pragma unroll
  for (int m = 1; m < NR; m <<= 1) {  // runs of m registers -> runs of 2 m
#pragma unroll
    for (int base = 0; base < NR; base += 2 * m) {
#pragma unroll
      for (int i = 0; i < m; ++i) {
        const int lo = base + i, hi = base + 2 * m - 1 - i;
        const u64 rev = xor_shfl<63>(k[hi]);
        const u64 big = k[lo] > rev ? k[lo] : rev, small = k[lo] > rev ? rev : k[lo];
        k[lo] = big;
        k[hi] = small;  // (each half is bitonic now; the order inside k[hi] does not matter yet)
      }
#pragma unroll
      for (int d = m >> 1; d >= 1; d >>= 1) {  // half-cleaners between registers
#pragma unroll
        for (int i = 0; i < 2 * m; ++i) {
          if ((i & d) == 0) {
            const u64 x = k[base + i], y = k[base + i + d];
            k[base + i] = x > y ? x : y;
            k[base + i + d] = x > y ? y : x;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) k[r] = bitonic_merge<32>(k[r]);
  }
}

// tokens t0 + lane + 64 q, q < 8, of one sequence (all eight loads in flight); `fill` past T
__device__ __forceinline__ void load_tokens(const int64_t *tok, const int T, const int64_t st,
                                            const int64_t off, const int t0, const int64_t fill,
                                            int64_t (&v)[8]) {
  const int lane = lane_id();
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int t = t0 + lane + q * PDT_WAVE;
    v[q] = t < T ? tok[(int64_t)t * st + off] : fill;
  }
}

// first position of `eos` in tok[0..T) (T when absent): _lens_from_eos, _string.py:137-143
__device__ __forceinline__ int first_eos(const int64_t *tok, const int T, const int64_t st,
                                         const int64_t off, const int64_t eos) {
  const int lane = lane_id();
  for (int t0 = 0; t0 < T; t0 += 8 * PDT_WAVE) {
    int64_t v[8];
    load_tokens(tok, T, st, off, t0, 0, v);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned long long hit = __ballot(t0 + lane + q * PDT_WAVE < T && v[q] == eos);
      if (hit) return t0 + q * PDT_WAVE + (int)__builtin_ctzll(hit);
    }
  }
  return T;
}

// Distinct tokens of tok[0..len) (len <= 64 NR) in ascending order -> tab[0..U); returns U.  The
// tokens stay in registers (xt[q] = token lane + 64 q, INT64_MAX past len) for later look-ups.
// key = ~(token with its sign bit flipped): a descending sort of the keys is an ascending sort of
// the tokens with the padding last; the first element of every run of equal keys is kept.
template <int NR>
__device__ __forceinline__ void load_sequence(const int64_t *tok, const int len, const int64_t st,
                                              const int64_t off, int64_t (&xt)[NR]) {
  int64_t part[8];
#pragma unroll
  for (int c = 0; c < NR / 8; ++c) {
    load_tokens(tok, len, st, off, c * 8 * PDT_WAVE, INT64_MAX, part);
#pragma unroll
    for (int q = 0; q < 8; ++q) xt[c * 8 + q] = part[q];
  }
}

template <int NR>
__device__ __forceinline__ int distinct_sorted_regs(const int len, const int64_t (&xt)[NR], int64_t *tab) {
  const int lane = lane_id();
  constexpr u64 kSign = 0x8000000000000000ull;
  int U = 0;
  u64 key[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) key[q] = lane + q * PDT_WAVE < len ? ~((u64)xt[q] ^ kSign) : 0ull;
  sort_regs_desc<NR>(key);
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int idx = q * PDT_WAVE + lane;
    // key of element idx - 1: the lane below, or lane 63 of the register below
    unsigned plo = 0u, phi = 0u;
    if (q > 0) {
      plo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key[q - 1], 63);
      phi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key[q - 1] >> 32), 63);
    }
    const unsigned lo = (unsigned)shr1((int)(unsigned)key[q], (int)plo);
    const unsigned hi = (unsigned)shr1((int)(unsigned)(key[q] >> 32), (int)phi);
    const u64 prev = ((u64)hi << 32) | lo;
    const bool first = idx < len && (idx == 0 || key[q] != prev);
    const unsigned long long firsts = __ballot(first);
    if (first)
      tab[U + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(firsts >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)firsts, 0u))] =
          (int64_t)(~key[q] ^ kSign);
    U += (int)__popcll(firsts);
  }
  return U;
}

template <int NR>
__device__ __forceinline__ int distinct_sorted(const int64_t *tok, const int len, const int64_t st,
                                               const int64_t off, int64_t (&xt)[NR], int64_t *tab) {
  load_sequence<NR>(tok, len, st, off, xt);
  return distinct_sorted_regs<NR>(len, xt, tab);
}

// ---- small non-negative tokens: classes without a sort ----------------------------------------
// When every token of the sequence lies in [0, kDirectBits) -- vocabulary indices, the usual case --
// its distinct tokens are the set bits of a kDirectBits-bit presence map in LDS, and the class of a
// token (its rank among them, the same number the sorted table gives) is the count of set bits
// below it: the map's words carry their exclusive prefix count, so a look-up is ONE 8-byte LDS read
// and a popcount instead of a log2(U)-step binary search, and the 64 x NR-key sort disappears.
constexpr int kDirectBits = 8192, kDirectWords = kDirectBits / 32;
static_assert(kDirectWords % PDT_WAVE == 0, "word w belongs to lane w % 64");

template <int NR>
__device__ __forceinline__ bool tokens_are_small(const int len, const int64_t (&xt)[NR]) {
  const int lane = lane_id();
  bool big = false;
#pragma unroll
  for (int q = 0; q < NR; ++q) big = big || (lane + q * PDT_WAVE < len && (u64)xt[q] >= (u64)kDirectBits);
  return __ballot(big) == 0ull;
}

// map[w] = (presence word w, number of distinct tokens below 32 w); returns U
template <int NR>
__device__ __forceinline__ int presence_map(const int len, const int64_t (&xt)[NR], uint2 *map) {
  const int lane = lane_id();
  constexpr int kRounds = kDirectWords / PDT_WAVE;
#pragma unroll
  for (int j = 0; j < kRounds; ++j) map[lane + j * PDT_WAVE] = make_uint2(0u, 0u);
  wave_sync();
#pragma unroll
  for (int q = 0; q < NR; ++q)
    if (lane + q * PDT_WAVE < len) atomicOr(&map[(int)xt[q] >> 5].x, 1u << ((int)xt[q] & 31));
  wave_sync();
  int U = 0;
#pragma unroll
  for (int j = 0; j < kRounds; ++j) {  // word lane + 64 j: prefix within the round + the rounds before
    const int cnt = __popc(map[lane + j * PDT_WAVE].x);
    const int incl = wave_incl_scan_add(cnt);
    map[lane + j * PDT_WAVE].y = (unsigned)(U + incl - cnt);
    U += __builtin_amdgcn_readlane(incl, PDT_WAVE - 1);
  }
  wave_sync();
  return U;
}

template <int NQ>
__device__ __forceinline__ void classes_from_map(const uint2 *map, const int64_t (&v)[NQ], int (&cls)[NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const bool in = (u64)v[q] < (u64)kDirectBits;
    const int t = in ? (int)v[q] : 0;
    const uint2 e = map[t >> 5];
    const unsigned bit = 1u << (t & 31);
    cls[q] = (in && (e.x & bit)) ? (int)(e.y + (unsigned)__popc(e.x & (bit - 1u))) : -1;
  }
}

// the distinct tokens themselves, ascending: out[k * stride] = k-th token (callers that list them)
template <typename T>
__device__ __forceinline__ void tokens_from_map(const uint2 *map, T *out) {
  const int lane = lane_id();
#pragma unroll 1
  for (int j = 0; j < kDirectWords / PDT_WAVE; ++j) {
    const uint2 e = map[lane + j * PDT_WAVE];
    unsigned w = e.x;
    int at = (int)e.y;
    while (w) {
      const int b = __builtin_ctz(w);
      w &= w - 1u;
      out[at++] = (T)((lane + j * PDT_WAVE) * 32 + b);
    }
  }
}

// search depth for classes_of: steps 2^(lg-1) .. 1 reach every index below U
__device__ __forceinline__ int search_depth(const int U) {
  int lg = 1;
  while ((1 << lg) < U) ++lg;
  return lg;
}

}  // namespace pdt
