// Step functions of the two beam searches for beams wider than a wave (beam_search_advance with
// width or K' above 64, ctc_prefix_search_advance with width or K' above 32): the plain forms.
// One workgroup of eight waves per batch element.  The top K of the K' * V (+ K') candidates come
// from a radix select on the order-preserving float keys -- three histogram passes (11 + 11 + 10
// bits) find the K-th largest key, one more pass collects everything above it, and candidates
// EQUAL to it are taken lowest flat index first (each wave owns a contiguous range of the flat
// index, so "first" is well defined) -- then the K winners are ranked by counting.  The candidates
// are never materialised: every pass re-evaluates them from the inputs (L2 resident: K' * V floats
// per element).  Semantics: reference _decoding.py:41-155 and :636-934; ties as in the wave forms
// (beam_advance.hip), i.e. to the lowest flat index.
#include "advance_args.hpp"
#include "wave_select.hpp"

namespace pdt {

constexpr int kWideWaves = 8;
constexpr int kWideThreads = kWideWaves * PDT_WAVE;

struct WideSel {
  u64 *list, *sorted;  // [K] the winners: as collected, then best first
  unsigned *hist;      // [2048]
  unsigned *part;      // [256]
  int *ctl;            // [16]: 0 collected count, 1 digit, 2 count above it, 3 count in it, 8.. per-wave ties
  static size_t bytes(int K) { return (size_t)K * 16 + 2048 * 4 + 256 * 4 + 16 * 4; }
  __device__ unsigned char *carve(unsigned char *p, int K) {
    list = reinterpret_cast<u64 *>(p);
    sorted = list + K;
    hist = reinterpret_cast<unsigned *>(sorted + K);
    part = hist + 2048;
    ctl = reinterpret_cast<int *>(part + 256);
    return reinterpret_cast<unsigned char *>(ctl + 16);
  }
};

__device__ __forceinline__ int lanes_below(u64 bal) {  // set bits of bal below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
}

// scan(f): every wave calls f(key, flat index, active) on its own contiguous range of the flat
// candidate index, ascending, all 64 lanes together (inactive lanes pad the last chunk of a row).
// The candidates hold at least K entries.  Result: s.sorted[0..K) = (key, index) best first.
template <typename SCAN>
__device__ __forceinline__ void block_top_k(const int K, SCAN &&scan, const WideSel &s) {
  const int tid = (int)threadIdx.x, wave = tid >> 6;
  unsigned prefix = 0u;
  int need = K, cnt_eq = 0;
#pragma unroll 1
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0;
    const int hi = pass == 1 ? 21 : 10;  // (pass 0 looks at every key)
    const unsigned mask = pass == 2 ? 1023u : 2047u;
    for (int i = tid; i < 2048; i += kWideThreads) s.hist[i] = 0u;
    __syncthreads();
    scan([&](const unsigned key, const unsigned, const bool active) {
      if (active && (pass == 0 || (key >> hi) == (prefix >> hi))) atomicAdd(&s.hist[(key >> shift) & mask], 1u);
    });
    __syncthreads();
    if (tid < 256) {
      unsigned sum = 0u;
      for (int b = 0; b < 8; ++b) sum += s.hist[tid * 8 + b];
      s.part[tid] = sum;
    }
    __syncthreads();
    if (tid < 256) {  // the one thread whose eight bins hold the need-th largest key says which
      unsigned above = 0u;
      for (int u = 255; u > tid; --u) above += s.part[u];
      if (above < (unsigned)need && (unsigned)need <= above + s.part[tid]) {
        unsigned acc = above;
        for (int b = tid * 8 + 7; b >= tid * 8; --b) {
          const unsigned h = s.hist[b];
          if (acc + h >= (unsigned)need) {
            s.ctl[1] = b;
            s.ctl[2] = (int)acc;
            s.ctl[3] = (int)h;
            break;
          }
          acc += h;
        }
      }
    }
    __syncthreads();
    prefix |= (unsigned)s.ctl[1] << shift;
    need -= s.ctl[2];
    cnt_eq = s.ctl[3];
    __syncthreads();
  }
  // prefix: the K-th largest key; need (>= 1) of the cnt_eq candidates that equal it are wanted
  const unsigned T = prefix;
  const bool all_eq = cnt_eq == need;
  if (tid == 0) s.ctl[0] = 0;
  __syncthreads();
  int eq_seen = 0;
  scan([&](const unsigned key, const unsigned idx, const bool active) {
    const bool take = active && (key > T || (all_eq && key == T));
    const u64 bal = __ballot(take);
    if (bal) {
      int base = 0;
      if (lane_id() == (int)__builtin_ctzll(bal)) base = atomicAdd(&s.ctl[0], (int)__popcll(bal));
      base = __builtin_amdgcn_readlane(base, (int)__builtin_ctzll(bal));
      if (take) s.list[base + lanes_below(bal)] = pack_key(key, idx);
    }
    if (!all_eq) eq_seen += (int)__popcll(__ballot(active && key == T));
  });
  if (lane_id() == 0) s.ctl[8 + wave] = eq_seen;
  __syncthreads();
  if (!all_eq) {  // ties at the threshold: the lowest flat indexes
    int rank = 0;
    for (int u = 0; u < wave; ++u) rank += s.ctl[8 + u];
    const int first = K - need;
    if (rank < need)
      scan([&](const unsigned key, const unsigned idx, const bool active) {
        const bool eq = active && key == T;
        const u64 bal = __ballot(eq);
        const int r = rank + lanes_below(bal);
        if (eq && r < need) s.list[first + r] = pack_key(key, idx);
        rank += (int)__popcll(bal);
      });
    __syncthreads();
  }
  for (int i = tid; i < K; i += kWideThreads) {
    const u64 e = s.list[i];
    int r = 0;
    for (int j = 0; j < K; ++j) r += s.list[j] > e ? 1 : 0;
    s.sorted[r] = e;
  }
  __syncthreads();
}

// rows [k0, k1) of the K' prefixes for this wave: contiguous, so that the flat index ascends
__device__ __forceinline__ void wave_rows(const int Kp, int &k0, int &k1) {
  const int wave = (int)(threadIdx.x >> 6), per = (Kp + kWideWaves - 1) / kWideWaves;
  k0 = min(Kp, wave * per);
  k1 = min(Kp, k0 + per);
}

// -------------------------------------------------------------------------------------------
// beam_search_advance (_decoding.py:41-155).  Candidate (k, v): log_probs_prev[k] + log_probs_t[k, v].
__global__ void __launch_bounds__(kWideThreads) beam_advance_wide_kernel(const BeamAdvArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = (int)threadIdx.x, lane = lane_id();
  const int64_t n = blockIdx.x;
  const int V = a.V, W = a.W, Kp = a.Kp, S = a.S;
  const int K = (int)min((int64_t)W, (int64_t)Kp * V);  // :121
  WideSel sel;
  int *srcs = reinterpret_cast<int *>(sel.carve(smem, K));
  int *toks = srcs + W;
  int *plens = toks + W;
  int k0, k1;
  wave_rows(Kp, k0, k1);
  auto scan = [&](auto &&f) {
    for (int k = k0; k < k1; ++k) {
      const float lp = a.lpp[n * a.lp_sn + k * a.lp_sk];
      const float *row = a.lpt + n * a.lt_sn + k * a.lt_sk;
      for (int v0 = 0; v0 < V; v0 += 4 * PDT_WAVE) {
        float x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int v = v0 + q * PDT_WAVE + lane;
          x[q] = v < V ? row[(int64_t)v * a.lt_sv] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int v = v0 + q * PDT_WAVE + lane;
          if (v0 + q * PDT_WAVE < V) f(fkey((lp + x[q]) + 0.0f), (unsigned)(k * V + v), v < V);  // :122 (-0.0 ties with +0.0)
        }
      }
    }
  };
  block_top_k(K, scan, sel);
  for (int i = tid; i < W; i += kWideThreads) {
    const bool valid = i < K;
    const u64 e = valid ? sel.sorted[i] : 0ull;
    const int c = valid ? (int)idx_of(e) : 0;
    const int src = c / V, tok = c - src * V;
    const int plen = valid ? (a.lens ? (int)a.lens[n * a.le_sn + src * a.le_sk] : S) : -1;
    a.lp_next[n * W + i] = valid ? fkey_inv(key_of(e)) : -PDT_INF;  // :145-153 for the overflow
    a.next_src[n * W + i] = valid ? src : 0;
    a.y_next_lens[n * W + i] = valid ? plen + 1 : 0;
    srcs[i] = valid ? src : -1;
    toks[i] = tok;
    plens[i] = plen;
  }
  __syncthreads();
  for (int64_t idx = tid; idx < (int64_t)a.S_out * W; idx += kWideThreads) {
    const int s = (int)(idx / W), i = (int)(idx - (int64_t)s * W);
    const int src = srcs[i], pl = plens[i];
    int64_t v;
    if (src < 0)
      v = 0;
    else if (s == pl || s >= S)  // :130/:135 the appended token row, :137 the scatter
      v = toks[i];
    else
      v = a.y_prev[(int64_t)s * a.yp_ss + n * a.yp_sn + src * a.yp_sk];
    a.y_next[((int64_t)s * a.N + n) * W + i] = v;
  }
}

static int wide_launch(const void *kern, size_t smem) {
  if (smem > 160 * 1024) return PDT_E_TOO_LONG;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  return PDT_OK;
}

int launch_beam_advance_wide(BeamAdvArgs a, hipStream_t stream) {
  if ((int64_t)a.Kp * a.V >= (1ll << 31)) return PDT_E_TOO_LONG;
  const int K = (int)min((int64_t)a.W, (int64_t)a.Kp * a.V);
  const size_t smem = (WideSel::bytes(K) + (size_t)a.W * 12 + 15) & ~(size_t)15;
  if (int rc = wide_launch(reinterpret_cast<const void *>(beam_advance_wide_kernel), smem)) return rc;
  hipLaunchKernelGGL(beam_advance_wide_kernel, dim3((unsigned)a.N), dim3(kWideThreads), smem, stream, a);
  return (int)hipGetLastError();
}

// -------------------------------------------------------------------------------------------
// ctc_prefix_search_advance (_decoding.py:636-934).  Candidates: the K' * V extensions
//   E[k, v] = ((v == last[k] ? 0 : nb[k]) + b[k]) * ext[k, v]           (:784-789)
// -- -inf where prefix k extended by v IS another prefix k' of the beam, whose non-extension mass
// receives E[k, v] instead (:804-837) -- followed by the K' non-extensions NB[k] + B[k] (:842-845).
// A row's merged tokens are listed per wave right before the row is scanned (the list has at most
// K' entries; one for a beam of distinct prefixes).
struct CtcWide {
  const CtcAdvArgs &a;
  int64_t n;
  const float *nb, *b;  // LDS (K')
  const int *last, *lens;
  __device__ bool exact(int k, int kp) const {  // :823-825
    return lens[k] + 1 == lens[kp] && a.isp[n * a.ip_sn + k * a.ip_sa + kp * a.ip_sb] != 0;
  }
  __device__ int need(int k, int kp) const {  // the token that turns prefix k into prefix k' (:808)
    if (a.S <= 0) return 0;
    const int pos = max(0, min(lens[k], a.S - 1));
    const int64_t t = a.y_prev[(int64_t)pos * a.yp_ss + n * a.yp_sn + kp * a.yp_sk];
    return (int)min(max(t, (int64_t)0), (int64_t)a.V - 1);
  }
  __device__ float ext_mass(int k, int v) const {
    const float w = (v == last[k] ? 0.0f : nb[k]) + b[k];
    return w * a.ext[n * a.ext_sn + k * a.ext_sk + v * a.ext_sv];
  }
};

__global__ void __launch_bounds__(kWideThreads) ctc_advance_wide_kernel(const CtcAdvArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
  const int64_t n = blockIdx.x;
  const int V = a.V, W = a.W, Kp = a.Kp, S = a.S;
  const int64_t total = (int64_t)Kp * (V + 1);
  const int K = (int)min((int64_t)W, total);  // :775
  WideSel sel;
  float *nb = reinterpret_cast<float *>(sel.carve(smem, K));
  float *b = nb + Kp, *NB = b + Kp, *B = NB + Kp;
  int *last = reinterpret_cast<int *>(B + Kp), *lens = last + Kp;
  int *merged = lens + Kp + wave * Kp;  // this wave's list of merged tokens of the row at hand
  int *o_src = lens + Kp + kWideWaves * Kp, *o_len = o_src + W, *o_non = o_len + W, *o_tok = o_non + W;

  for (int k = tid; k < Kp; k += kWideThreads) {
    nb[k] = a.nb_prev[n * a.pb_sn + k * a.pb_sk];
    b[k] = a.b_prev[n * a.pbb_sn + k * a.pbb_sk];
    last[k] = (int)min(max(a.last[n * a.la_sn + k * a.la_sk], (int64_t)0), (int64_t)V - 1);  // :779
    lens[k] = (int)a.lens[n * a.le_sn + k * a.le_sk];
  }
  __syncthreads();
  const CtcWide cw{a, n, nb, b, last, lens};
  const float blank = a.blank[n * a.bl_sn];
  for (int kp = tid; kp < Kp; kp += kWideThreads) {
    float add = 0.0f;
    for (int k = 0; k < Kp; ++k)  // :829-831, summed over k in index order
      if (cw.exact(k, kp)) add += cw.ext_mass(k, cw.need(k, kp));
    NB[kp] = nb[kp] * a.nonext[n * a.ne_sn + last[kp] * a.ne_sv] + add;  // :794
    B[kp] = (nb[kp] + b[kp]) * blank;                                      // :777, :791
  }
  __syncthreads();

  int k0, k1;
  wave_rows(Kp, k0, k1);
  auto scan = [&](auto &&f) {
    for (int k = k0; k < k1; ++k) {
      int nm = 0;
      for (int q0 = 0; q0 < Kp; q0 += PDT_WAVE) {
        const int kp = q0 + lane;
        const bool ex = kp < Kp && cw.exact(k, kp);
        const int tm = ex ? cw.need(k, kp) : 0;
        const u64 bal = __ballot(ex);
        if (ex) merged[nm + lanes_below(bal)] = tm;
        nm += (int)__popcll(bal);
      }
      wave_sync();
      const float wn = nb[k], wb = b[k];
      const int lk = last[k];
      const float *row = a.ext + n * a.ext_sn + k * a.ext_sk;
      for (int v0 = 0; v0 < V; v0 += 4 * PDT_WAVE) {
        float x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int v = v0 + q * PDT_WAVE + lane;
          x[q] = v < V ? row[(int64_t)v * a.ext_sv] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int v = v0 + q * PDT_WAVE + lane;
          if (v0 + q * PDT_WAVE < V) {
            float e = ((v == lk ? 0.0f : wn) + wb) * x[q];
            for (int j = 0; j < nm; ++j)
              if (merged[j] == v) e = -PDT_INF;  // :833-837
            f(fkey(e + 0.0f), (unsigned)(k * V + v), v < V);
          }
        }
      }
      wave_sync();
    }
    if (wave == kWideWaves - 1)  // the non-extensions come last in the flat order
      for (int q0 = 0; q0 < Kp; q0 += PDT_WAVE) {
        const int k = q0 + lane;
        f(fkey(k < Kp ? (NB[k] + B[k]) + 0.0f : 0.0f), (unsigned)(Kp * V + k), k < Kp);
      }
  };
  block_top_k(K, scan, sel);

  for (int j = tid; j < W; j += kWideThreads) {
    const bool valid = j < K;
    const u64 e = valid ? sel.sorted[j] : 0ull;
    const int ind = valid ? (int)idx_of(e) : 0;
    const bool non = ind >= Kp * V;                       // :849
    const int src = non ? ind - Kp * V : ind / V;         // :850-852
    const int tok = ind % V;                              // :853
    const int plen = lens[src];
    o_src[j] = valid ? src : -1;
    o_len[j] = valid ? plen + (non ? 0 : 1) : 0;          // :865
    o_non[j] = non ? 1 : 0;
    o_tok[j] = tok;
    a.y_next_last[n * W + j] = valid ? (non ? last[src] : tok) : 0;  // :878-880
    a.y_next_lens[n * W + j] = o_len[j];
    a.nb_next[n * W + j] = valid ? (non ? NB[src] : fkey_inv(key_of(e))) : -PDT_INF;  // :868-872
    a.b_next[n * W + j] = valid ? (non ? B[src] : B[src] * 0.0f) : -PDT_INF;          // :875
    a.next_src[n * W + j] = valid ? src : 0;
    a.next_nonext[n * W + j] = (uint8_t)(valid && non);
  }
  __syncthreads();
  // y_next[s, j] before the rows past the new length are cleared (what :883-898 reads)
  auto hist = [&](const int s, const int j) -> int64_t {
    const int src = o_src[j];
    if (s == lens[src]) return o_tok[j];
    return s < S ? a.y_prev[(int64_t)s * a.yp_ss + n * a.yp_sn + src * a.yp_sk] : 0;
  };
  for (int64_t idx = tid; idx < (int64_t)W * W; idx += kWideThreads) {  // :883-898
    const int x = (int)(idx / W), y = (int)(idx - (int64_t)x * W);
    bool ok = false;
    if (x < K && y < K) {
      const int la = o_len[x], lb = o_len[y];
      ok = la <= lb && a.isp[n * a.ip_sn + o_src[x] * a.ip_sa + o_src[y] * a.ip_sb] != 0;
      if (ok && !o_non[x]) ok = hist(max(la - 1, 0), y) == (int64_t)o_tok[x];
    }
    a.next_isp[(n * W + x) * W + y] = (uint8_t)ok;
  }
  for (int64_t idx = tid; idx < (int64_t)(S + 1) * W; idx += kWideThreads) {  // :855-864
    const int s = (int)(idx / W), j = (int)(idx - (int64_t)s * W);
    a.y_next[((int64_t)s * a.N + n) * W + j] = (j < K && s < o_len[j]) ? hist(s, j) : 0;
  }
}

int launch_ctc_advance_wide(CtcAdvArgs a, hipStream_t stream) {
  if ((int64_t)a.Kp * (a.V + 1) >= (1ll << 31)) return PDT_E_TOO_LONG;
  const int K = (int)min((int64_t)a.W, (int64_t)a.Kp * (a.V + 1));
  const size_t smem = (WideSel::bytes(K) + (size_t)a.Kp * (24 + 4 * kWideWaves) + (size_t)a.W * 16 + 15) & ~(size_t)15;
  if (int rc = wide_launch(reinterpret_cast<const void *>(ctc_advance_wide_kernel), smem)) return rc;
  hipLaunchKernelGGL(ctc_advance_wide_kernel, dim3((unsigned)a.N), dim3(kWideThreads), smem, stream, a);
  return (int)hipGetLastError();
}

}  // namespace pdt
