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
__device__ __forceinline__ int distinct_sorted(const int64_t *tok, const int len, const int64_t st,
                                               const int64_t off, int64_t (&xt)[NR], int64_t *tab) {
  const int lane = lane_id();
  {
    int64_t part[8];
#pragma unroll
    for (int c = 0; c < NR / 8; ++c) {
      load_tokens(tok, len, st, off, c * 8 * PDT_WAVE, INT64_MAX, part);
#pragma unroll
      for (int q = 0; q < 8; ++q) xt[c * 8 + q] = part[q];
    }
  }
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

// search depth for classes_of: steps 2^(lg-1) .. 1 reach every index below U
__device__ __forceinline__ int search_depth(const int U) {
  int lg = 1;
  while ((1 << lg) < U) ++lg;
  return lg;
}

}  // namespace pdt
