// Optimal completion for references beyond the 2048 tokens lev_rowsync.hip holds in registers
// (reference _string.py:464-517 has no such bound): the same class bitmasks and class-token
// tables, from a plain formulation -- ONE WORKGROUP per utterance, the DP rows, the sort buffer and
// the class ids in a global workspace (pdt_oc_mask_workspace_bytes), workgroup barriers between
// the phases of a row.  A way to get an answer, not a fast one (~10 us per DP row and utterance).
// The same kernel serves the cost-mode DISTANCES for such references when the costs are not exact
// in float32 (the other cost sets take lev_skewed.hip at any length): bitmask == nullptr, the
// value of column ref_len of every row is what comes out.
//
// Per row h:  t[c] = min(prev[c] + ins, prev[c-1] + sub * [ref[c-1] != hyp[h-1]])   (:293, :316)
//             row[c] = min_k<=c (t[k] + (c - k) * del)                               (:264-266, :317)
// as a prefix minimum of t[k] - k * del when the costs are exact in float32; otherwise (inexact = 1)
// the reference's unrolled form term by term, row[c] = min_k<=c ((row0[c] - row0[k]) + t[k]) with
// row0[k] = float(k) * del, O(R^2) per row like lev_rowsync.hip's EXACT path.  Then the row minimum
// and the class bits of the columns that attain it (:333-334, :347-355), OR-ed into the bitmask row
// in HBM.
#include "lev_common.hpp"

namespace pdt {

struct GenericOcArgs {
  LevArgs l;
  unsigned char *ws;
  int64_t ws_per_utt;
  int P;  // sort capacity: a power of two >= R
  int inexact;  // replay the reference's deletion unroll term by term
};

int64_t generic_oc_ws_per_utt(int64_t R, int64_t H, int *P_out) {
  int P = 2;
  while (P < R) P <<= 1;
  if (P_out) *P_out = P;
  // sort buffer, class tokens (distance mode has no caller's table), class of every reference
  // position, class of every hypothesis position, three rows, the per-row distances
  int64_t b = (int64_t)P * 8 + (R + 1) * 8 + (R + 1) * 4 + (H + 1) * 4 + 3 * (R + 2) * 4 + (H + 2) * 4;
  return (b + 255) & ~(int64_t)255;
}

constexpr int kGenThreads = 256;

__device__ __forceinline__ float block_min(float v, float *red) {
  const int tid = (int)threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = kGenThreads / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fminf(red[tid], red[tid + s]);
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}
__device__ __forceinline__ int block_sum(int v, int *red) {
  const int tid = (int)threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = kGenThreads / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const int r = red[0];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(kGenThreads) oc_mask_generic_kernel(const GenericOcArgs g) {
  __shared__ float fred[kGenThreads];
  __shared__ int ired[kGenThreads];
  __shared__ int s_len[2];
  const LevArgs &a = g.l;
  const int tid = (int)threadIdx.x;
  const int64_t n = blockIdx.x;
  const int R = a.R, H = a.H, W = a.W, P = g.P;
  unsigned char *w = g.ws + n * g.ws_per_utt;
  int64_t *srt = reinterpret_cast<int64_t *>(w);
  int64_t *ctok_ws = srt + P;                               // [R + 1]
  int *rid = reinterpret_cast<int *>(ctok_ws + (R + 1));    // [R + 1]
  int *hcls = rid + (R + 1);                                // [H + 1]
  float *rowA = reinterpret_cast<float *>(hcls + (H + 1));  // [R + 2]
  float *rowB = rowA + (R + 2), *rowC = rowB + (R + 2);
  float *bnd = rowC + (R + 2);                              // [H + 2] distance mode: D[h][ref_len]
  const bool mask_mode = a.bitmask != nullptr;
  int64_t *ctok = mask_mode ? a.class_tokens + n * (int64_t)R : ctok_ws;
  const int64_t roff = n * a.ref_sn, hoff = n * a.hyp_sn;
  auto ref_at = [&](int i) { return a.ref[(int64_t)i * a.ref_st + roff]; };
  auto hyp_at = [&](int i) { return a.hyp[(int64_t)i * a.hyp_st + hoff]; };

  // ---- lengths (_string.py:195-228) ----------------------------------------------------------
  if (tid < 2) s_len[tid] = tid == 0 ? R : H;
  __syncthreads();
  if (a.has_eos) {
    for (int i = tid; i < R; i += kGenThreads)
      if (ref_at(i) == a.eos) {
        atomicMin(&s_len[0], i);
        break;
      }
    for (int i = tid; i < H; i += kGenThreads)
      if (hyp_at(i) == a.eos) {
        atomicMin(&s_len[1], i);
        break;
      }
  }
  __syncthreads();
  int ref_len = s_len[0], hyp_len = s_len[1];
  bool rmiss = false, hmiss = false;
  if (a.has_eos && a.include_eos) {
    if (ref_len == R) rmiss = true; else ref_len += 1;
    if (hyp_len == H) hmiss = true; else hyp_len += 1;
  }
  int Heff = a.exclude_last ? hyp_len - 1 : hyp_len;
  if (Heff < 0) Heff = 0;
  const int Hout = H + (a.exclude_last ? 0 : 1);

  // ---- distinct reference tokens in ascending order: bitonic sort in the workspace -----------
  for (int i = tid; i < P; i += kGenThreads) srt[i] = i < ref_len ? ref_at(i) : INT64_MAX;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (P >> 1); t += kGenThreads) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const int64_t x = srt[i], y = srt[l];
        if ((x > y) == ((i & k) == 0)) {
          srt[i] = y;
          srt[l] = x;
        }
      }
      __syncthreads();
    }
  }
  // unique-compact: every thread owns a contiguous piece of the sorted array
  const int per = (P + kGenThreads - 1) / kGenThreads, i0 = tid * per;
  int nfirst = 0;
  for (int q = 0; q < per; ++q) {
    const int i = i0 + q;
    if (i < ref_len && (i == 0 || srt[i] != srt[i - 1])) ++nfirst;
  }
  ired[tid] = nfirst;
  __syncthreads();
  int before = 0;
  for (int t = 0; t < tid; ++t) before += ired[t];
  int U = 0;
  for (int t = 0; t < kGenThreads; ++t) U += ired[t];
  __syncthreads();
  {
    int pos = before;
    for (int q = 0; q < per; ++q) {
      const int i = i0 + q;
      if (i < ref_len && (i == 0 || srt[i] != srt[i - 1])) ctok[pos++] = srt[i];
    }
  }
  __syncthreads();
  auto class_of_tok = [&](int64_t v) {
    int lo = 0, hi = U;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (ctok[mid] < v) lo = mid + 1; else hi = mid;
    }
    return (lo < U && ctok[lo] == v) ? lo : -1;
  };
  for (int i = tid; i < ref_len; i += kGenThreads) rid[i] = class_of_tok(ref_at(i));
  for (int i = tid; i < hyp_len && i < H; i += kGenThreads) hcls[i] = class_of_tok(hyp_at(i));

  // ---- row 0 and its set: only column 0 exists (:271-278) -------------------------------------
  const float ins = a.ins, del = a.del, sub = a.sub;
  float *prev = rowA, *cur = rowB, *nxt = rowC;
  for (int c = tid; c <= ref_len; c += kGenThreads) prev[c] = (float)c * del;
  uint32_t *brow = mask_mode ? a.bitmask + ((int64_t)0 * a.N + n) * W : nullptr;
  if (mask_mode)
    for (int i = tid; i < W; i += kGenThreads) brow[i] = 0u;
  __syncthreads();
  int max_cnt = 0;
  if (mask_mode && ref_len > 0) {
    if (tid == 0) brow[rid[0] >> 5] = 1u << (rid[0] & 31);
    max_cnt = 1;
  }

  const int cper = (ref_len + 1 + kGenThreads - 1) / kGenThreads, c0 = tid * cper;
  for (int h = 1; h <= Heff; ++h) {
    const int tok = hcls[h - 1];
    if (mask_mode) {
      brow = a.bitmask + ((int64_t)h * a.N + n) * W;
      for (int i = tid; i < W; i += kGenThreads) brow[i] = 0u;
    }
    float m = PDT_INF;
    float *row;  // the finished row h
    if (g.inexact) {
      // t into `cur`, then every column's minimum over the columns at or before it, in the
      // reference's own arithmetic: (row0[c] - row0[k]) + t[k]
      for (int q = 0; q < cper; ++q) {
        const int c = c0 + q;
        if (c > ref_len) break;
        float t = prev[c] + ins;
        if (c > 0) t = fminf(t, prev[c - 1] + ((rid[c - 1] != tok) ? sub : 0.0f));
        cur[c] = t;
      }
      __syncthreads();
      for (int c = tid; c <= ref_len; c += kGenThreads) {  // (interleaved: the work per column grows with c)
        const float rc = (float)c * del;
        float best = PDT_INF;
        for (int k = 0; k <= c; ++k) best = fminf(best, (rc - (float)k * del) + cur[k]);
        nxt[c] = best;
        m = fminf(m, best);
      }
      row = nxt;
    } else {
      // t[c] - c * del over this thread's columns, and their running minimum
      float run = PDT_INF;
      for (int q = 0; q < cper; ++q) {
        const int c = c0 + q;
        if (c > ref_len) break;
        float t = prev[c] + ins;
        if (c > 0) t = fminf(t, prev[c - 1] + ((rid[c - 1] != tok) ? sub : 0.0f));
        run = fminf(run, t - (float)c * del);
        cur[c] = run;  // (prefix minimum inside the piece; the pieces before it are folded in below)
      }
      fred[tid] = run;
      __syncthreads();
      float carry = PDT_INF;
      for (int t = 0; t < tid; ++t) carry = fminf(carry, fred[t]);
      __syncthreads();
      for (int q = 0; q < cper; ++q) {
        const int c = c0 + q;
        if (c > ref_len) break;
        const float v = fminf(cur[c], carry) + (float)c * del;
        cur[c] = v;
        m = fminf(m, v);
      }
      row = cur;
    }
    m = block_min(m, fred);  // (its barriers also order the row's writes and the zeroing of the bitmask row)
    if (mask_mode) {
      for (int c = tid; c < ref_len; c += kGenThreads)  // the r < ref_len cut of :349-354
        if (row[c] == m) atomicOr(&brow[rid[c] >> 5], 1u << (rid[c] & 31));
      __syncthreads();
      int cnt = 0;
      for (int i = tid; i < W; i += kGenThreads) cnt += __popc(brow[i]);
      cnt = block_sum(cnt, ired);
      max_cnt = cnt > max_cnt ? cnt : max_cnt;
    } else if (tid == 0) {
      bnd[h] = row[ref_len];
    }
    // rotate: the finished row becomes `prev`; the old `prev` is free
    if (g.inexact) {
      float *tmp = prev;
      prev = nxt;
      nxt = tmp;
    } else {
      float *tmp = prev;
      prev = cur;
      cur = tmp;
    }
  }
  if (!mask_mode) {  // cost-mode distances (the epilogue of lev_rowsync.hip)
    __syncthreads();
    const float r0 = (float)ref_len * del;
    if (a.mode == PDT_MODE_FINAL) {
      if (tid == 0)
        a.out[n * a.out_sn] = lev_finish(Heff > 0 ? bnd[Heff] : r0, a.mult, a.norm, ref_len, hyp_len > 0 ? 1.0f : 0.0f);
    } else {
      const int pad_from = hyp_len + (a.exclude_last ? 0 : 1);
      for (int h = tid; h < Hout; h += kGenThreads) {
        float v;
        if (h >= pad_from)
          v = a.padding;
        else
          v = lev_finish(h == 0 ? r0 : bnd[h], a.mult, a.norm, ref_len, h > 0 ? 1.0f : 0.0f);
        a.out[(int64_t)h * a.out_sh + n * a.out_sn] = v;
      }
    }
  }
  // rows of finished hypotheses carry empty sets (`& not_done`, :334)
  for (int h = Heff + 1; h < Hout && mask_mode; ++h) {
    brow = a.bitmask + ((int64_t)h * a.N + n) * W;
    for (int i = tid; i < W; i += kGenThreads) brow[i] = 0u;
  }
  if (tid == 0) {
    int flags = 0;
    if (rmiss) flags |= PDT_WARN_REF_NO_EOS;
    if (hmiss) flags |= PDT_WARN_HYP_NO_EOS;
    if (!mask_mode && a.norm && ref_len == 0) flags |= PDT_WARN_EMPTY_REF;
    if (flags && a.status) atomicOr(a.status, flags);
    if (a.max_count && max_cnt > 0) atomicMax(a.max_count, max_cnt);
    if (a.ref_lens_out) a.ref_lens_out[n] = ref_len;
    if (a.hyp_lens_out) a.hyp_lens_out[n] = hyp_len;
  }
}

// class bitmasks of any width -> padded ascending token lists: one wave per (h, n) row, 64 words
// at a time
__global__ void __launch_bounds__(256)
oc_expand_generic_kernel(const uint32_t *__restrict__ bitmask, const int64_t *__restrict__ class_tokens,
                         int R, int W, int64_t rows, int64_t N, int C, int64_t padding,
                         int64_t *__restrict__ targets, int64_t tgt_sh, int64_t tgt_sn) {
  const int lane = lane_id();
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t h = row / N, n = row - h * N;
  int64_t *dst = targets + h * tgt_sh + n * tgt_sn;
  const int64_t *ctok = class_tokens + n * (int64_t)R;
  int total = 0;
  for (int w0 = 0; w0 < W; w0 += PDT_WAVE) {
    unsigned w = w0 + lane < W ? bitmask[row * W + w0 + lane] : 0u;
    const int cnt = __popc(w);
    const int incl = wave_incl_scan_add(cnt);
    int pos = total + incl - cnt;
    while (w) {
      const int b = __builtin_ctz(w);
      w &= w - 1u;
      if (pos < C) dst[pos] = ctok[(w0 + lane) * 32 + b];
      ++pos;
    }
    total += __builtin_amdgcn_readlane(incl, PDT_WAVE - 1);
  }
  for (int i = total + lane; i < C; i += PDT_WAVE) dst[i] = padding;
}

int launch_oc_mask_generic(const LevArgs &a, bool inexact, void *ws, int64_t ws_bytes, hipStream_t stream) {
  GenericOcArgs g{};
  g.l = a;
  g.inexact = inexact ? 1 : 0;
  g.ws_per_utt = generic_oc_ws_per_utt(a.R, a.H, &g.P);
  if (!ws || ws_bytes < g.ws_per_utt * a.N) return PDT_E_TOO_LONG;  // (no workspace: the bounded kernels only)
  g.ws = reinterpret_cast<unsigned char *>(ws);
  hipLaunchKernelGGL(oc_mask_generic_kernel, dim3((unsigned)a.N), dim3(kGenThreads), 0, stream, g);
  return (int)hipGetLastError();
}

int launch_oc_expand_generic(const uint32_t *bitmask, const int64_t *class_tokens, int R, int Hout,
                             int64_t N, int C, int64_t padding, int64_t *targets, int64_t tgt_sh,
                             int64_t tgt_sn, hipStream_t stream) {
  const int W = (int)pdt_oc_mask_words(R);
  const int64_t rows = (int64_t)Hout * N;
  if ((rows + 3) / 4 >= (1ll << 31)) return PDT_E_TOO_LONG;
  hipLaunchKernelGGL(oc_expand_generic_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, bitmask,
                     class_tokens, R, W, rows, N, C, padding, targets, tgt_sh, tgt_sn);
  return (int)hipGetLastError();
}

}  // namespace pdt
