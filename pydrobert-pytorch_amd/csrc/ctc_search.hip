// CTC prefix beam search for gfx950 -- one wave per utterance, one LANE per beam entry.
//
// Replaces CTCPrefixSearch.forward without a language model (reference
// _decoding.py:1064-1202) and its step function ctc_prefix_search_advance (:636-934),
// which the reference evaluates as ~45 dense tensor ops per frame (a (N,K,K,V) one-hot and a
// gather over the whole (t,N,K) history among them).  Here the whole T-frame search of one
// utterance runs inside one wave:
//   * beam state (nb, b, last token, length, trie node, is-prefix row as a 64-bit mask) lives
//     in the registers of lane k;
//   * a frame's logits are staged once into LDS, softmax statistics by DPP reductions;
//   * candidates are never materialised: the mass of extending prefix k with token v is
//     w_k(v) * p[v], monotone in p[v] for fixed k, so a K-way merge over per-prefix streams
//     that walk ONE shared list of the top (K + K') tokens (wave_top_sorted) yields the exact
//     global top-K in K rounds of a 32-bit DPP max-reduce;
//   * prefixes are a trie in HBM ((parent, token) per created node) instead of dense
//     (t, N, K) histories, so a frame writes K records instead of gathering t*K tokens; the
//     only history look-ups the algorithm needs (token of prefix b at the position where
//     prefix a ends) are served from a K x K table in LDS maintained incrementally.
//
// Tie policy: equal-mass candidates are taken lowest beam index first (the reference's
// torch.topk leaves ties unspecified); parity is defined on tie-free inputs.
#include "ctc_frame.hpp"

namespace pdt {

__global__ void __launch_bounds__(256) ctc_search_kernel(const CtcArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  const int wave = (int)(threadIdx.x >> 6);
  const int64_t n = (int64_t)xcd_remap(blockIdx.x, gridDim.x) * a.waves_per_wg + wave;
  if (n >= a.N) return;
  const int V = a.V, W = a.W;
  unsigned char *base = smem + (size_t)wave * a.lds_per_wave;
  float *p = reinterpret_cast<float *>(base);
  FrameLds L;
  L.carve(base + (size_t)((V + 1 + 3) & ~3) * 4, V, W, W, false);
  for (int v = lane; v < V; v += PDT_WAVE) L.pos[v] = 0xFF;

  const int Tn = a.lens ? (int)min((int64_t)a.T, max((int64_t)0, a.lens[n])) : a.T;
  Beam bm;  // :1097-1105: one empty prefix with all the mass on "ends in blank"
  bm.nb = lane == 0 ? 0.0f : -PDT_INF;
  bm.b = lane == 0 ? 1.0f : -PDT_INF;
  bm.last = 0;
  bm.len = 0;
  bm.node = -1;
  bm.isp = lane == 0 ? 1ull : 0ull;
  int Kp = 1;
  constexpr int kPrefetch = 8;
  float pre[kPrefetch];
  if (Tn > 0) {
    const float *row0 = a.logits + n * a.lg_sn;
#pragma unroll
    for (int i = 0; i < kPrefetch; ++i) {
      const int v = lane + i * PDT_WAVE;
      pre[i] = v <= V ? row0[(int64_t)v * a.lg_sv] : 0.0f;
    }
  }
#ifdef PDT_STAMPS
  unsigned long long pdt_stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int t = 0; t < Tn; ++t) {
    PDT_STAMP_BEGIN;
    // softmax statistics of frame t (:1093) -- p[v] = exp(x[v] - max), sum over v in [0, V].
    // The first kPrefetch*64 logits of the frame were fetched into registers one frame ago.
    float mx = -PDT_INF;
#pragma unroll
    for (int i = 0; i < kPrefetch; ++i) {
      const int v = lane + i * PDT_WAVE;
      if (v <= V) {
        p[v] = pre[i];
        mx = fmaxf(mx, pre[i]);
      }
    }
    {
      const float *row = a.logits + (int64_t)t * a.lg_st + n * a.lg_sn;
      for (int v = lane + kPrefetch * PDT_WAVE; v <= V; v += PDT_WAVE) {
        const float x = row[(int64_t)v * a.lg_sv];
        p[v] = x;
        mx = fmaxf(mx, x);
      }
    }
    if (t + 1 < Tn) {
      const float *nrow = a.logits + (int64_t)(t + 1) * a.lg_st + n * a.lg_sn;
#pragma unroll
      for (int i = 0; i < kPrefetch; ++i) {
        const int v = lane + i * PDT_WAVE;
        if (v <= V) pre[i] = nrow[(int64_t)v * a.lg_sv];
      }
    }
    mx = wave_max_f(mx);
    float s = 0.0f;
    for (int v = lane; v <= V; v += PDT_WAVE) {
      const float e = expf(p[v] - mx);
      p[v] = e;
      s += e;
    }
    s = wave_sum_f(s);
    wave_sync();
    PDT_STAMP(0);
    int ns, nt, nk;
    ctc_frame<false>(bm, p, s, V, W, Kp, t, n, a, DenseCtx{}, L, ns, nt, nk PDT_STAMP_ARG);
    int *tmp = L.nxt_old;
    L.nxt_old = L.nxt_new;
    L.nxt_new = tmp;
    Kp = W;
  }

#ifdef PDT_STAMPS
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_stamps[i], pdt_stamp_acc[i]);
  unsigned long long stamp_last_ = __builtin_readcyclecounter();
#endif
  // ---- outputs (:1188-1200): probabilities, lengths, and the prefixes read off the trie --
  if (lane < W) {
    a.y_probs[n * W + lane] = bm.nb + bm.b;
    a.y_lens[n * W + lane] = bm.len;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  // rows beyond a prefix's length stay 0: the caller hands in a zero-filled y
  if (lane < W) {
    int node = bm.node;
    for (int pos = bm.len - 1; pos >= 0 && node >= 0; --pos) {
      const int tt = node / W, ii = node - tt * W;
      const int2 *rec = a.trie + (((int64_t)tt * a.N + n) * W + ii);
      const int par = __hip_atomic_load(&rec->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int tok = __hip_atomic_load(&rec->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.y[((int64_t)pos * a.N + n) * W + lane] = tok;
      node = par;
    }
  }
#ifdef PDT_STAMPS
  if (lane == 0) atomicAdd(&g_stamps[6], __builtin_readcyclecounter() - stamp_last_);
#endif
}

size_t ctc_lds_per_wave(int V, int W) {
  return (size_t)((V + 1 + 3) & ~3) * 4 + FrameLds::bytes(V, W, W, false);
}

int launch_ctc_search(CtcArgs a, hipStream_t stream) {
  if (a.W < 1 || a.W > kMaxWidth) return PDT_E_TOO_LONG;
  const size_t per_wave = ctc_lds_per_wave(a.V, a.W);
  const size_t soft_cap = 64 * 1024, hard_cap = 160 * 1024;
  if (per_wave > hard_cap) return PDT_E_TOO_LONG;
  int wpw = (int)(soft_cap / per_wave);
  wpw = wpw > 4 ? 4 : (wpw < 1 ? 1 : wpw);
  a.waves_per_wg = wpw;
  a.lds_per_wave = (int)per_wave;
  const size_t smem = per_wave * wpw;
  if (smem > soft_cap) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_search_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
  }
  const unsigned grid = (unsigned)((a.N + wpw - 1) / wpw);
  hipLaunchKernelGGL(ctc_search_kernel, dim3(grid), dim3(64 * wpw), smem, stream, a);
  return (int)hipGetLastError();
}

}  // namespace pdt

extern "C" {

int64_t pdt_ctc_prefix_search_workspace_bytes(int64_t T, int64_t N, int64_t width) {
  if (T < 0 || N < 0 || width < 0) return 0;
  return T * N * width * (int64_t)sizeof(int2) + 16;
}

int pdt_ctc_prefix_search(const float *logits, int64_t T, int64_t N, int64_t V, int64_t lg_st,
                          int64_t lg_sn, int64_t lg_sv, const int64_t *lens, int64_t width,
                          int64_t S, int64_t *y, int64_t *y_lens, float *y_probs,
                          void *workspace, void *stream) {
  using namespace pdt;
  if (T < 0 || N < 0 || V < 1 || width < 1 || S < 0) return PDT_E_ARG;
  if (N == 0) return PDT_OK;
  if (!y_lens || !y_probs || (T > 0 && (!logits || !workspace)) || (S > 0 && !y)) return PDT_E_ARG;
  if (width > kMaxWidth) return PDT_E_TOO_LONG;
  if (T * width >= (1ll << 31) || V >= (1 << 30) || N >= (1ll << 31)) return PDT_E_TOO_LONG;
  CtcArgs a{};
  a.logits = logits; a.lg_st = lg_st; a.lg_sn = lg_sn; a.lg_sv = lg_sv;
  a.lens = lens;
  a.T = (int)T; a.N = (int)N; a.V = (int)V; a.W = (int)width; a.S = (int)S;
  a.y = y; a.y_lens = y_lens; a.y_probs = y_probs;
  a.trie = reinterpret_cast<int2 *>(workspace);
  return launch_ctc_search(a, (hipStream_t)stream);
}

}  // extern "C"

#ifdef PDT_STAMPS
extern "C" int pdt_debug_read_stamps(unsigned long long *host16, int reset) {
  hipError_t e = hipMemcpyFromSymbol(host16, HIP_SYMBOL(pdt::g_stamps), sizeof(unsigned long long) * 16);
  if (e == hipSuccess && reset) {
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(pdt::g_stamps), z, sizeof(z));
  }
  return (int)e;
}
#endif
