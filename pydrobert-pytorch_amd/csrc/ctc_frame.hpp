// One frame of CTC prefix beam search at wave level (shared by the fused search kernel in
// ctc_search.hip and the step-function kernel in beam_advance.hip).  See ctc_search.hip for
// the design notes.
#pragma once
#include "wave_select.hpp"

namespace pdt {

constexpr int kMaxWidth = 32;  // K + K' <= 64 tokens fit one per lane

struct CtcArgs {
  const float *logits;  // (T, N, V + 1), element strides below
  int64_t lg_st, lg_sn, lg_sv;
  const int64_t *lens;  // (N,) or null
  int T, N, V, W, S;    // S = rows of y
  int64_t *y;           // (S, N, W) contiguous
  int64_t *y_lens;      // (N, W)
  float *y_probs;       // (N, W)
  int2 *trie;           // (T, N, W) records (parent node, token)
  int lds_per_wave, waves_per_wg;
};

struct Beam {
  float nb, b;
  int last, len, node;
  u64 isp;  // bit k' set <=> this prefix is a prefix of beam entry k'
};

__device__ __forceinline__ u64 readlane_u64(u64 v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float shfl_f(float v, int l) { return __shfl(v, l); }

// Extra inputs of the step-function form (ctc_prefix_search_advance): per-prefix extension
// probabilities (language-model fusion) and a dense (S, N, K') history instead of the trie.
struct DenseCtx {
  const float *ext;      // ext_probs_t[n] : element (k, v) at ext[k * ext_sk + v * ext_sv]
  int64_t ext_sk, ext_sv;
  const int64_t *y_prev; // y_prev[:, n, :] : element (s, k) at y_prev[s * yp_ss + k * yp_sk]
  int64_t yp_ss, yp_sk;
  int S;
};

// One frame of the search.  `p` holds the (unnormalised) non-extension probabilities of
// v in [0, V] (index V = blank) and `sum` their normaliser (1 when already normalised).
// Kp = number of live lanes (1 at t = 0, then W).  Per-wave LDS scratch:
//   tl: sorted token list(s) -- one shared list of 64, or Kp lists when DENSE;
//   rem[RS*RS] removed tokens per parent (RS = max(W, Kp)), nxt_old/nxt_new[W*W] (trie form).
// On return new_src / new_tok / new_kind describe where lane i's new prefix came from.
template <bool DENSE>
__device__ __forceinline__ void ctc_frame(Beam &bm, const float *p, const float sum, const int V,
                                          const int W, const int Kp, const int RS, const int t,
                                          const int64_t n, const CtcArgs &a, const DenseCtx &dc,
                                          u64 *surv, int *tl, int *rem, int *nxt_old,
                                          int *nxt_new, int &new_src, int &new_tok,
                                          int &new_kind) {
  const int lane = lane_id();
  const bool live = lane < Kp;
  const int K = min(W, Kp * (V + 1));  // _decoding.py:775
  const int M = min(V, K + Kp);

  // ---- sorted token list(s): tokens by descending extension probability ----------------
  if (!DENSE) {
    const u64 tk = wave_top_sorted(p, V, M, surv);
    if (lane < M) tl[lane] = (int)idx_of(tk);
  } else {
    for (int k = 0; k < Kp; ++k) {
      const u64 tk = wave_top_sorted_strided(dc.ext + k * dc.ext_sk, dc.ext_sv, V, M, surv);
      if (lane < M) tl[k * PDT_WAVE + lane] = (int)idx_of(tk);
      wave_sync();
    }
  }
  const int *mytl = DENSE ? tl + (lane < Kp ? lane : 0) * PDT_WAVE : tl;
  // extension probability of (this lane's prefix, token v)
  auto ext_prob = [&](int v) -> float {
    if (DENSE) return dc.ext[(lane < Kp ? lane : 0) * dc.ext_sk + v * dc.ext_sv];
    return __fdiv_rn(p[v], sum);
  };
  const float p_blank = __fdiv_rn(p[V], sum);

  // ---- candidate masses that do not depend on the token (:777-794) ----------------------
  const int lastc = min(max(bm.last, 0), V - 1);
  const float tot = bm.nb + bm.b;
  const float B = tot * p_blank;
  float NB = bm.nb * __fdiv_rn(p[lastc], sum);
  wave_sync();

  // ---- merge: an extension of k that equals an existing prefix k' feeds k' (:804-837) ---
  float add = 0.0f;
  int nrem = 0;  // number of removed tokens of THIS lane's prefix
  for (int kk = 0; kk < Kp; ++kk) {
    const u64 isp_kk = readlane_u64(bm.isp, kk);
    if ((isp_kk & ~(1ull << kk)) == 0ull) continue;
    const int len_kk = __builtin_amdgcn_readlane(bm.len, kk);
    const bool child = live && ((isp_kk >> lane) & 1ull) && (len_kk + 1 == bm.len);
    const u64 cm = __ballot(child);
    if (cm == 0ull) continue;
    const float nb_kk = readlane_f(bm.nb, kk), b_kk = readlane_f(bm.b, kk);
    const int last_kk = min(max(__builtin_amdgcn_readlane(bm.last, kk), 0), V - 1);
    if (child) {
      // to_match = the last token of the child (its length is len_kk + 1)
      const float w = (lastc == last_kk ? 0.0f : nb_kk) + b_kk;
      const float e = DENSE ? dc.ext[kk * dc.ext_sk + lastc * dc.ext_sv] : __fdiv_rn(p[lastc], sum);
      add += w * e;
      const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
      rem[kk * RS + pos] = lastc;
    }
    if (lane == kk) nrem = __popcll(cm);
  }
  NB = NB + add;
  wave_sync();

  // ---- K-way merge of the per-prefix candidate streams ----------------------------------
  // stream 0: tokens tl[ptr..] except `lastc` and removed ones, mass (nb + b) * p[v]
  // stream 1: the token `lastc`, mass b * p[lastc]            (:784-789)
  // stream 2: not extending, mass NB + B                       (:842-845)
  auto removed = [&](int v) {
    for (int i = 0; i < nrem; ++i)
      if (rem[lane * RS + i] == v) return true;
    return false;
  };
  const bool valid_beam = live && tot > -PDT_INF;
  int ptr = 0;
  auto skip = [&]() {
    while (ptr < M) {
      const int v = mytl[ptr];
      if (v != lastc && !removed(v)) break;
      ++ptr;
    }
  };
  skip();
  bool s1_open = valid_beam && !removed(lastc);
  bool s2_open = valid_beam;
  const float m1 = bm.b * ext_prob(lastc);
  const float m2 = NB + B;

  new_src = 0, new_tok = 0, new_kind = -1;  // kind: 0/1 extension, 2 non-extension, -1 invalid
  float new_mass = -PDT_INF;
  for (int i = 0; i < K; ++i) {
    float best = 0.0f;
    int kind = -1;
    if (valid_beam && ptr < M) {
      best = tot * ext_prob(mytl[ptr]);
      kind = 0;
    }
    if (s1_open && (kind < 0 || m1 > best || (m1 == best && lastc < mytl[ptr]))) {
      best = m1;
      kind = 1;
    }
    if (s2_open && (kind < 0 || m2 > best)) {
      best = m2;
      kind = 2;
    }
    const unsigned key = kind >= 0 ? fkey(best) : 0u;
    const unsigned mx = wave_max_u32(key);
    if (mx == 0u) break;  // fewer valid candidates than K: the rest stay invalid (:902-924)
    const int win = (int)__builtin_ctzll(__ballot(key == mx));
    const int wkind = __builtin_amdgcn_readlane(kind, win);
    const int wtok = __builtin_amdgcn_readlane(kind == 0 ? mytl[ptr < M ? ptr : 0] : lastc, win);
    const float wmass = readlane_f(best, win);
    if (lane == i) {
      new_src = win;
      new_tok = wtok;
      new_kind = wkind;
      new_mass = wmass;
    }
    if (lane == win) {
      if (kind == 0) {
        ++ptr;
        skip();
      } else if (kind == 1) {
        s1_open = false;
      } else {
        s2_open = false;
      }
    }
  }

  // ---- new beam state of lane i (:868-880) ---------------------------------------------
  const int srcl = new_kind >= 0 ? new_src : lane;
  const float NB_s = shfl_f(NB, srcl), B_s = shfl_f(B, srcl);
  const int last_s = __shfl(lastc, srcl), len_s = __shfl(bm.len, srcl), node_s = __shfl(bm.node, srcl);
  const u64 isp_s = shfl_u64(bm.isp, srcl);
  const bool is_ext = new_kind == 0 || new_kind == 1;
  const bool is_valid = new_kind >= 0;
  Beam nw;
  nw.nb = !is_valid ? -PDT_INF : (is_ext ? new_mass : NB_s);
  nw.b = !is_valid ? -PDT_INF : (is_ext ? 0.0f : B_s);
  nw.last = !is_valid ? 0 : (is_ext ? new_tok : last_s);
  nw.len = !is_valid ? 0 : len_s + (is_ext ? 1 : 0);
  nw.node = !is_valid ? -1 : (is_ext ? t * W + lane : node_s);
  if (!DENSE && is_valid && is_ext)
    a.trie[((int64_t)t * a.N + n) * W + lane] = make_int2(node_s, new_tok);

  // ---- is-prefix relation and next-token table of the new beam (:883-898) ---------------
  // nxt[a * W + b] = token of prefix b at position len(a), defined when a is a strict prefix
  u64 isp_new = 0ull;
  bool need_walk = false;
  for (int b = 0; b < K; ++b) {
    const int kind_b = __builtin_amdgcn_readlane(new_kind, b);
    if (kind_b < 0) continue;
    const int src_b = __builtin_amdgcn_readlane(new_src, b);
    const int tok_b = __builtin_amdgcn_readlane(new_tok, b);
    const int lenB = __builtin_amdgcn_readlane(len_s, b);  // length of b's source prefix
    const bool ext_b = kind_b != 2;
    const int len_b = lenB + (ext_b ? 1 : 0);
    bool ok = is_valid && ((isp_s >> src_b) & 1ull) && nw.len <= len_b;
    int tok_at = -1;  // token of new prefix b at position len_s (= len of my source prefix)
    if (ok) {
      if (lenB > len_s)
        tok_at = DENSE ? (int)dc.y_prev[(int64_t)len_s * dc.yp_ss + src_b * dc.yp_sk]
                       : nxt_old[new_src * W + src_b];
      else
        tok_at = ext_b ? tok_b : -1;  // lenB == len_s
      if (is_ext) ok = tok_at == new_tok;
    }
    if (ok) {
      isp_new |= 1ull << b;
      if (!DENSE && nw.len < len_b) {  // strict prefix: record the token that follows me inside b
        int nx;
        if (!is_ext) {
          nx = tok_at;
        } else if (lenB == len_s + 1) {
          nx = tok_b;  // b = (my new prefix) + tok_b
        } else {
          nx = -(2 + b);  // deeper than the table reaches: resolved below by a trie walk
          need_walk = true;
        }
        nxt_new[lane * W + b] = nx;
      }
    }
  }
  if (!DENSE && __ballot(need_walk)) {
    // rare: a re-created intermediate prefix.  Token of b at position nw.len = token of the
    // ancestor of b's source node at depth nw.len + 1.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    for (int b = 0; b < K; ++b) {
      const int node_b = __builtin_amdgcn_readlane(node_s, b);
      const int lenB = __builtin_amdgcn_readlane(len_s, b);
      if (need_walk && ((isp_new >> b) & 1ull) && nxt_new[lane * W + b] == -(2 + b)) {
        int node = node_b, depth = lenB, tok = -1;
        while (node >= 0) {
          const int tt = node / W, ii = node - tt * W;
          const int2 *rec = a.trie + (((int64_t)tt * a.N + n) * W + ii);
          const int par = __hip_atomic_load(&rec->x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          tok = __hip_atomic_load(&rec->y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (depth == nw.len + 1) break;
          node = par;
          --depth;
        }
        nxt_new[lane * W + b] = tok;
      }
    }
  }
  nw.isp = isp_new;
  bm = nw;
  wave_sync();
}

}  // namespace pdt
