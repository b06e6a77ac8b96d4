// fill_after_eos (reference _string.py:30-42): everything after the first `eos` along one
// dimension is replaced by a fill value.
//
// The reference takes seven passes over int64 temporaries (eq, long, cumsum, clamp, cumsum, gt,
// masked_fill).  Here the tensor is viewed as (outer, L, inner) around the sequence dimension and
// every sequence is walked once: tokens are read once, the value tensor (the tokens themselves
// unless the caller passes another) once, the output written once.  Values move as opaque
// 1/2/4/8-byte words, so every dtype is served by four instantiations.  HBM-bound.
//   inner > 1: a workgroup owns 64 consecutive columns (coalesced at every row); its waves take
//              the rows l = w, w + NW, ... -- first the position of the first eos per column (an
//              LDS minimum), then, after one barrier, the output rows.  A thread walking its
//              column alone is one round trip to HBM per row: 0.15 ms for (512, 4096) where the
//              tensor is 10 us of traffic;
//   inner = 1: one wave per sequence, 64 consecutive positions per load, eight loads in flight,
//              the first eos found by ballots.
#include "pdt_common.hpp"

namespace pdt {

struct FillArgs {
  const int64_t *tok;  // (outer, L, inner) contiguous
  const void *val;     // same shape, element size = sizeof(W)
  void *out;
  int64_t outer, L, inner, eos;
  unsigned long long fill;  // the fill value's bits in the low bytes
};

template <typename W>
__global__ void __launch_bounds__(1024) fill_after_eos_columns(const FillArgs a) {
  __shared__ int first[64];
  const int c = (int)(threadIdx.x & 63u), w = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
  const int64_t col = (int64_t)blockIdx.x * 64 + c;
  const bool live = col < a.outer * a.inner;
  const int64_t o = live ? col / a.inner : 0, i = live ? col - o * a.inner : 0;
  const int64_t base = o * a.L * a.inner + i;
  const W *val = reinterpret_cast<const W *>(a.val);
  W *out = reinterpret_cast<W *>(a.out);
  const W fill = (W)a.fill;
  const int L = (int)a.L;
  if (w == 0) first[c] = L;
  __syncthreads();
  int mine = L;  // first eos among this thread's rows
  for (int l0 = w; l0 < L && mine == L; l0 += 8 * nw) {
    int64_t t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = (live && l0 + q * nw < L) ? a.tok[base + (int64_t)(l0 + q * nw) * a.inner] : a.eos + 1;
#pragma unroll
    for (int q = 7; q >= 0; --q)
      if (l0 + q * nw < L && t[q] == a.eos) mine = l0 + q * nw;
  }
  if (mine < L) atomicMin(&first[c], mine);
  __syncthreads();
  const int f = first[c];
  for (int l0 = w; l0 < L; l0 += 8 * nw) {
    W v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int l = l0 + q * nw;
      v[q] = (live && l < L && l <= f) ? val[base + (int64_t)l * a.inner] : fill;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int l = l0 + q * nw;
      if (live && l < L) out[base + (int64_t)l * a.inner] = v[q];
    }
  }
}

template <typename W>
__global__ void __launch_bounds__(256) fill_after_eos_rows(const FillArgs a) {
  const int lane = lane_id();
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.outer) return;
  const int64_t base = row * a.L;
  const W *val = reinterpret_cast<const W *>(a.val);
  W *out = reinterpret_cast<W *>(a.out);
  const W fill = (W)a.fill;
  bool seen = false;  // wave-uniform: an eos in an earlier chunk
  for (int64_t l0 = 0; l0 < a.L; l0 += 8 * PDT_WAVE) {
    int64_t t[8];
    W v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t l = l0 + q * PDT_WAVE + lane;
      t[q] = l < a.L ? a.tok[base + l] : a.eos + 1;
      v[q] = l < a.L ? val[base + l] : fill;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t l = l0 + q * PDT_WAVE + lane;
      const unsigned long long hits = __ballot(t[q] == a.eos);
      // positions strictly after the first eos of this chunk
      const bool after = seen || (hits != 0ull && lane > (int)__builtin_ctzll(hits));
      if (l < a.L) out[base + l] = after ? fill : v[q];
      seen = seen || hits != 0ull;
    }
  }
}

template <typename W>
static int launch_fill(const FillArgs &a, hipStream_t stream) {
  if (a.inner == 1) {
    const int64_t grid = (a.outer + 3) / 4;
    if (grid > 0x7fffffffll) return PDT_E_TOO_LONG;
    hipLaunchKernelGGL(fill_after_eos_rows<W>, dim3((unsigned)grid), dim3(256), 0, stream, a);
  } else {
    if (a.L > 0x7fffffffll) return PDT_E_TOO_LONG;
    const int64_t grid = (a.outer * a.inner + 63) / 64;
    if (grid > 0x7fffffffll) return PDT_E_TOO_LONG;
    int nw = 1;  // waves along the sequence dimension
    while (nw < 16 && nw * 8 < a.L) nw *= 2;
    hipLaunchKernelGGL(fill_after_eos_columns<W>, dim3((unsigned)grid), dim3(64 * nw), 0, stream, a);
  }
  return (int)hipGetLastError();
}

}  // namespace pdt

extern "C" int pdt_fill_after_eos(const int64_t *tokens, int64_t outer, int64_t L, int64_t inner,
                                  int64_t eos, const void *value, int64_t elem_bytes,
                                  int64_t fill_bits, void *out, void *stream) {
  using namespace pdt;
  if (outer < 0 || L < 0 || inner < 0) return PDT_E_ARG;
  if (outer == 0 || L == 0 || inner == 0) return PDT_OK;
  if (!tokens || !value || !out) return PDT_E_ARG;
  FillArgs a{tokens, value, out, outer, L, inner, eos, (unsigned long long)fill_bits};
  switch (elem_bytes) {
    case 1: return launch_fill<uint8_t>(a, (hipStream_t)stream);
    case 2: return launch_fill<uint16_t>(a, (hipStream_t)stream);
    case 4: return launch_fill<uint32_t>(a, (hipStream_t)stream);
    case 8: return launch_fill<uint64_t>(a, (hipStream_t)stream);
    default: return PDT_E_ARG;
  }
}
