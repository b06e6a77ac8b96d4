// Variable-length padding (reference _pad.py:108-149) -- the data movement behind RandomShift
// (_img.py:883-908).
//
// The reference builds five (N, T', F) boolean masks, two gather buffers and three
// masked_scatter passes.  Here it is one pass: every output element decides which region it is
// in (left pad / sequence / right pad / beyond the new length) and copies its source element or
// the fill value (inputs up to 2^32 elements per call).  Elements are moved as opaque 1/2/4/8-byte words, so every dtype is served by
// four instantiations.  HBM-bound: one read of the valid input, one write of the output.
//
// Backward (float32) is the adjoint written as a gather, so it is deterministic: an input
// element collects the gradient of its own copy plus those of its reflections / replications.
#include "pdt_common.hpp"

namespace pdt {

enum { PADMODE_CONSTANT = 0, PADMODE_REFLECT = 1, PADMODE_REPLICATE = 2 };

struct PadArgs {
  const void *x;        // (N, T, F) contiguous
  const int64_t *lens;  // (N,)
  const int64_t *pad;   // (2, N): left, right
  int N, T, F, Tp, mode;
  void *out;            // (N, Tp, F)
  const void *fill;     // one element
};

// source time index of output position t of sequence (len, left, right), or -1 for fill
__device__ __forceinline__ int64_t pad_source(int64_t t, int64_t len, int64_t left, int64_t right,
                                              int mode) {
  if (t < left) {
    if (mode == PADMODE_REFLECT) return left - t;       // _pad.py:60-66
    if (mode == PADMODE_REPLICATE) return 0;            // :86-88
    return -1;
  }
  if (t < left + len) return t - left;
  if (t < left + len + right) {
    const int64_t j = t - left - len;
    if (mode == PADMODE_REFLECT) return len - j - 2;    // :67-72
    if (mode == PADMODE_REPLICATE) return len - 1;      // :94-98
  }
  return -1;
}

// One thread per output element, flat over (row, f): consecutive lanes move consecutive words
// of a row (coalesced on both sides) whatever F is; a workgroup covers 256 * kPerThread
// consecutive output words.  The per-element region test re-reads lens / pad from L1.
constexpr int kPerThread = 4;
template <typename W>
__global__ void __launch_bounds__(256) pad_variable_kernel(const PadArgs a, unsigned total) {
  const W fill = *reinterpret_cast<const W *>(a.fill);
  const unsigned F = (unsigned)a.F, Tp = (unsigned)a.Tp;
#pragma unroll
  for (int i = 0; i < kPerThread; ++i) {
    const unsigned gid = (blockIdx.x * kPerThread + i) * 256u + threadIdx.x;
    if (gid >= total) return;
    const unsigned row = gid / F, f = gid - row * F;
    const unsigned n = row / Tp, t = row - n * Tp;
    const int64_t len = a.lens[n], left = a.pad[n], right = a.pad[a.N + n];
    int64_t s = pad_source((int64_t)t, len, left, right, a.mode);
    if (s >= a.T) s = -1;  // lens beyond T: nothing to read
    W v = fill;
    if (s >= 0) v = reinterpret_cast<const W *>(a.x)[((int64_t)n * a.T + s) * F + f];
    reinterpret_cast<W *>(a.out)[gid] = v;
  }
}

// grad_x[n, s, :] = sum of grad_out over the output positions that read x[n, s, :]
__global__ void __launch_bounds__(256)
pad_variable_backward_kernel(const PadArgs a, const float *__restrict__ grad_out,
                             float *__restrict__ grad_x, unsigned total) {
  const unsigned F = (unsigned)a.F, T = (unsigned)a.T;
#pragma unroll
  for (int i = 0; i < kPerThread; ++i) {
    const unsigned gid = (blockIdx.x * kPerThread + i) * 256u + threadIdx.x;
    if (gid >= total) return;
    const unsigned row = gid / F, f = gid - row * F;
    const unsigned n = row / T;
    const int64_t s = row - n * T;
    const int64_t len = a.lens[n], left = a.pad[n], right = a.pad[a.N + n];
    const float *go = grad_out + (int64_t)n * a.Tp * F + f;
    float acc = 0.0f;
    if (s < len) {
      acc = go[(left + s) * F];
      if (a.mode == PADMODE_REFLECT) {
        if (s >= 1 && s <= left) acc += go[(left - s) * F];
        const int64_t j = len - s - 2;
        if (j >= 0 && j < right) acc += go[(left + len + j) * F];
      } else if (a.mode == PADMODE_REPLICATE) {
        if (s == 0)
          for (int64_t t = 0; t < left; ++t) acc += go[t * F];
        if (s == len - 1)
          for (int64_t t = left + len; t < left + len + right; ++t) acc += go[t * F];
      }
    }
    grad_x[gid] = acc;
  }
}

}  // namespace pdt

extern "C" {

int pdt_pad_variable(const void *x, int64_t N, int64_t T, int64_t F, int64_t elem_bytes,
                     const int64_t *lens, const int64_t *pad, int mode, const void *fill,
                     int64_t Tp, void *out, void *stream) {
  using namespace pdt;
  if (N < 0 || T < 0 || F < 0 || Tp < 0 || mode < 0 || mode > 2) return PDT_E_ARG;
  if (N == 0 || Tp == 0 || F == 0) return PDT_OK;
  if (!lens || !pad || !fill || !out || (T > 0 && !x)) return PDT_E_ARG;
  if (N * Tp * F >= (1ll << 32) - 1024 * kPerThread) return PDT_E_TOO_LONG;
  PadArgs a{};
  a.x = x; a.lens = lens; a.pad = pad; a.N = (int)N; a.T = (int)T; a.F = (int)F; a.Tp = (int)Tp;
  a.mode = mode; a.out = out; a.fill = fill;
  const unsigned total = (unsigned)(N * Tp * F);
  const dim3 grid((total + 256 * kPerThread - 1) / (256 * kPerThread));
  hipStream_t s = (hipStream_t)stream;
  switch (elem_bytes) {
    case 1: hipLaunchKernelGGL(pad_variable_kernel<uint8_t>, grid, dim3(256), 0, s, a, total); break;
    case 2: hipLaunchKernelGGL(pad_variable_kernel<uint16_t>, grid, dim3(256), 0, s, a, total); break;
    case 4: hipLaunchKernelGGL(pad_variable_kernel<uint32_t>, grid, dim3(256), 0, s, a, total); break;
    case 8: hipLaunchKernelGGL(pad_variable_kernel<uint64_t>, grid, dim3(256), 0, s, a, total); break;
    default: return PDT_E_ARG;
  }
  return (int)hipGetLastError();
}

int pdt_pad_variable_backward(const float *grad_out, int64_t N, int64_t T, int64_t F,
                              const int64_t *lens, const int64_t *pad, int mode, int64_t Tp,
                              float *grad_x, void *stream) {
  using namespace pdt;
  if (N < 0 || T < 0 || F < 0 || Tp < 0 || mode < 0 || mode > 2) return PDT_E_ARG;
  if (N == 0 || T == 0 || F == 0) return PDT_OK;
  if (!grad_out || !lens || !pad || !grad_x) return PDT_E_ARG;
  if (N * T * F >= (1ll << 32) - 1024 * kPerThread) return PDT_E_TOO_LONG;
  PadArgs a{};
  a.lens = lens; a.pad = pad; a.N = (int)N; a.T = (int)T; a.F = (int)F; a.Tp = (int)Tp;
  a.mode = mode;
  const unsigned total = (unsigned)(N * T * F);
  hipLaunchKernelGGL(pad_variable_backward_kernel, dim3((total + 256 * kPerThread - 1) / (256 * kPerThread)),
                     dim3(256), 0, (hipStream_t)stream, a, grad_out, grad_x, total);
  return (int)hipGetLastError();
}

}  // extern "C"
