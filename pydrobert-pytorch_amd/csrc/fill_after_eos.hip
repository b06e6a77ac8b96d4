// fill_after_eos (reference _string.py:30-42): everything after the first `eos` along one
// dimension is replaced by a fill value.
//
// The reference takes seven passes over int64 temporaries (eq, long, cumsum, clamp, cumsum, gt,
// masked_fill).  Here the tensor is viewed as (outer, L, inner) around the sequence dimension and
// every sequence is walked once: tokens are read once, the value tensor (the tokens themselves
// unless the caller passes another) once, the output written once.  Values move as opaque
// 1/2/4/8-byte words, so every dtype is served by four instantiations.  HBM-bound.
//   inner > 1: one thread per (outer, inner) column, consecutive lanes on consecutive `inner`
//              (coalesced at every step of the walk);
//   inner = 1: one wave per sequence, 64 consecutive positions per step, the first eos found by
//              a ballot.
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
__global__ void __launch_bounds__(256) fill_after_eos_columns(const FillArgs a) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= a.outer * a.inner) return;
  const int64_t o = col / a.inner, i = col - o * a.inner;
  const int64_t base = o * a.L * a.inner + i;
  const W *val = reinterpret_cast<const W *>(a.val);
  W *out = reinterpret_cast<W *>(a.out);
  const W fill = (W)a.fill;
  bool seen = false;
  for (int64_t l = 0; l < a.L; ++l) {
    const int64_t at = base + l * a.inner;
    out[at] = seen ? fill : val[at];
    seen = seen || a.tok[at] == a.eos;
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
  bool seen = false;  // wave-uniform: an eos at an earlier chunk
  for (int64_t l0 = 0; l0 < a.L; l0 += PDT_WAVE) {
    const int64_t l = l0 + lane;
    const bool in = l < a.L;
    const bool is_eos = in && a.tok[base + l] == a.eos;
    const unsigned long long hits = __ballot(is_eos);
    // positions strictly after the first eos of this chunk
    const bool after = seen || (hits != 0ull && lane > (int)__builtin_ctzll(hits));
    if (in) out[base + l] = after ? fill : val[base + l];
    seen = seen || hits != 0ull;
  }
}

template <typename W>
static int launch_fill(const FillArgs &a, hipStream_t stream) {
  if (a.inner == 1) {
    const int64_t grid = (a.outer + 3) / 4;
    if (grid > 0x7fffffffll) return PDT_E_TOO_LONG;
    hipLaunchKernelGGL(fill_after_eos_rows<W>, dim3((unsigned)grid), dim3(256), 0, stream, a);
  } else {
    const int64_t grid = (a.outer * a.inner + 255) / 256;
    if (grid > 0x7fffffffll) return PDT_E_TOO_LONG;
    hipLaunchKernelGGL(fill_after_eos_columns<W>, dim3((unsigned)grid), dim3(256), 0, stream, a);
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
