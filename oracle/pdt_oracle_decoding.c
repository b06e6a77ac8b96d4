/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C restatement of the beam-search step functions of pydrobert-pytorch
 * (reference: src/pydrobert/torch/_decoding.py).  Follows the reference's tensor
 * program literally (dense candidate tables, gathers over the whole history), one
 * utterance at a time.  Parity status: PINNED against the live reference
 * (tests/golden/make_golden.py) on tie-free inputs.
 *
 * Tie policy: torch.topk's order among equal values is unspecified
 * (_decoding.py:123, :846); this restatement takes the LOWEST flat index first.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pdt_oracle.h"

typedef struct {
  float v;
  int64_t i;
} cand_t;

static int cand_cmp(const void *a, const void *b) {
  const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
  /* NaN-free inputs assumed; descending value, ascending index */
  if (x->v > y->v) return -1;
  if (x->v < y->v) return 1;
  return (x->i > y->i) - (x->i < y->i);
}

static int64_t clampi(int64_t x, int64_t lo, int64_t hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* _decoding.py:41-155.  Layouts (contiguous):
 *   log_probs_t (N, Kp, V), log_probs_prev (N, Kp), y_prev (S, N, Kp), y_prev_lens (N, Kp) or NULL
 *   y_next (S_out, N, width) with S_out given by the caller (S or S + 1, see below),
 *   y_next_lens / log_probs_next / next_src (N, width).
 * Returns S_out when y_next == NULL (query), else 0 on success, <0 on error. */
int64_t pdt_oracle_beam_search_advance(const float *log_probs_t, int64_t N, int64_t Kp,
                                       int64_t V, int64_t width, const float *log_probs_prev,
                                       const int64_t *y_prev, int64_t S,
                                       const int64_t *y_prev_lens, int64_t *y_next,
                                       int64_t *y_next_lens, float *log_probs_next,
                                       int64_t *next_src) {
  if (width < 1) return -1;
  const int64_t K = width < Kp * V ? width : Kp * V; /* :121 */
  int grow = 1;
  if (S > 0 && y_prev_lens) { /* :133-135 */
    int64_t mx = 0;
    for (int64_t i = 0; i < N * Kp; ++i)
      if (y_prev_lens[i] > mx) mx = y_prev_lens[i];
    grow = mx >= S;
  }
  if (S == 0 && y_prev_lens) /* :139-140 */
    for (int64_t i = 0; i < N * Kp; ++i)
      if (y_prev_lens[i] != 0) return -2;
  const int64_t S_out = S + (grow ? 1 : 0);
  if (!y_next) return S_out;
  cand_t *c = (cand_t *)malloc(sizeof(cand_t) * (size_t)(Kp * V));
  for (int64_t n = 0; n < N; ++n) {
    for (int64_t k = 0; k < Kp; ++k)
      for (int64_t v = 0; v < V; ++v) {
        c[k * V + v].v = log_probs_prev[n * Kp + k] + log_probs_t[(n * Kp + k) * V + v]; /* :122 */
        c[k * V + v].i = k * V + v;
      }
    qsort(c, (size_t)(Kp * V), sizeof(cand_t), cand_cmp); /* :123 */
    for (int64_t j = 0; j < width; ++j) {
      if (j >= K) { /* :145-153 */
        log_probs_next[n * width + j] = -INFINITY;
        y_next_lens[n * width + j] = 0;
        next_src[n * width + j] = 0;
        for (int64_t s = 0; s < S_out; ++s) y_next[(s * N + n) * width + j] = 0;
        continue;
      }
      const int64_t src = c[j].i / V, tok = c[j].i % V; /* :124-125 */
      const int64_t plen = y_prev_lens ? y_prev_lens[n * Kp + src] : S;
      for (int64_t s = 0; s < S_out; ++s) /* :128 gather, :130/:135 cat of the token row */
        y_next[(s * N + n) * width + j] = s < S ? y_prev[(s * N + n) * Kp + src] : tok;
      if (plen < S_out) y_next[(plen * N + n) * width + j] = tok; /* :130, :137 */
      y_next_lens[n * width + j] = plen + 1;
      log_probs_next[n * width + j] = c[j].v;
      next_src[n * width + j] = src;
    }
  }
  free(c);
  return 0;
}

/* _decoding.py:636-934 for one batch; contiguous layouts:
 *   ext_probs_t (N, Kp, V) addressed with stride ext_sk over Kp (0 = broadcast of (N, V)),
 *   nonext_probs_t (N, V), blank_probs_t (N), nb/b_probs_prev (N, Kp), y_prev (S, N, Kp),
 *   y_prev_last / y_prev_lens (N, Kp), prev_is_prefix (N, Kp, Kp) bytes.
 * Outputs: y_next (S + 1, N, width), y_next_last / y_next_lens (N, width),
 *   nb/b_probs_next (N, width), next_is_prefix (N, width, width) bytes, next_src (N, width),
 *   next_is_nonext (N, width) bytes.  Entries of y_next beyond y_next_lens are set to 0
 *   (the reference leaves them undefined). */
int pdt_oracle_ctc_prefix_search_advance(
    const float *ext_probs_t, int64_t ext_sn, int64_t ext_sk, const float *nonext_probs_t,
    const float *blank_probs_t, int64_t N, int64_t Kp, int64_t V, int64_t width,
    const float *nb_probs_prev, const float *b_probs_prev, const int64_t *y_prev, int64_t S,
    const int64_t *y_prev_last, const int64_t *y_prev_lens, const uint8_t *prev_is_prefix,
    int64_t *y_next, int64_t *y_next_last, int64_t *y_next_lens, float *nb_probs_next,
    float *b_probs_next, uint8_t *next_is_prefix, int64_t *next_src, uint8_t *next_is_nonext) {
  if (width < 1) return -1;
  const int64_t K = width < Kp * (V + 1) ? width : Kp * (V + 1); /* :775 */
  const int64_t W = width;
  float *E = (float *)malloc(sizeof(float) * (size_t)(Kp * V));
  float *NB = (float *)malloc(sizeof(float) * (size_t)Kp);
  float *B = (float *)malloc(sizeof(float) * (size_t)Kp);
  int64_t *last = (int64_t *)malloc(sizeof(int64_t) * (size_t)Kp);
  cand_t *c = (cand_t *)malloc(sizeof(cand_t) * (size_t)(Kp * (V + 1)));
  int64_t *ext_tok = (int64_t *)malloc(sizeof(int64_t) * (size_t)W);
  for (int64_t n = 0; n < N; ++n) {
    const float *nbp = nb_probs_prev + n * Kp, *bp = b_probs_prev + n * Kp;
    const int64_t *lens = y_prev_lens + n * Kp;
    const uint8_t *isp = prev_is_prefix + n * Kp * Kp;
    for (int64_t k = 0; k < Kp; ++k) {
      last[k] = clampi(y_prev_last[n * Kp + k], 0, V - 1); /* :779 */
      const float tot = nbp[k] + bp[k];                     /* :777 */
      for (int64_t v = 0; v < V; ++v) {                     /* :784-789 */
        const float w = (v == last[k] ? 0.0f : nbp[k]) + bp[k];
        E[k * V + v] = w * ext_probs_t[n * ext_sn + k * ext_sk + v];
      }
      B[k] = tot * blank_probs_t[n];                   /* :791 */
      NB[k] = nbp[k] * nonext_probs_t[n * V + last[k]]; /* :794 */
    }
    /* :804-831 merge extensions that equal an existing longer prefix into it */
    for (int64_t kp = 0; kp < Kp; ++kp) {
      float add = 0.0f;
      for (int64_t k = 0; k < Kp; ++k) {
        const int exact = (lens[k] + 1 == lens[kp]) && isp[k * Kp + kp]; /* :823-825 */
        int64_t tm = 0;
        if (S > 0) {
          const int64_t pos = lens[k] < S - 1 ? lens[k] : S - 1; /* :808 clamp(max=tm1-1) */
          tm = clampi(y_prev[((pos < 0 ? 0 : pos) * N + n) * Kp + kp], 0, V - 1);
        }
        add += exact ? E[k * V + tm] : 0.0f; /* :829-831 (sum over k in index order) */
      }
      NB[kp] = NB[kp] + add;
    }
    for (int64_t k = 0; k < Kp; ++k) /* :833-837 */
      for (int64_t kp = 0; kp < Kp; ++kp) {
        const int exact = (lens[k] + 1 == lens[kp]) && isp[k * Kp + kp];
        if (!exact) continue;
        int64_t tm = 0;
        if (S > 0) {
          const int64_t pos = lens[k] < S - 1 ? lens[k] : S - 1;
          tm = clampi(y_prev[((pos < 0 ? 0 : pos) * N + n) * Kp + kp], 0, V - 1);
        }
        E[k * V + tm] = -INFINITY;
      }
    for (int64_t i = 0; i < Kp * V; ++i) { /* :842-845 */
      c[i].v = E[i];
      c[i].i = i;
    }
    for (int64_t k = 0; k < Kp; ++k) {
      c[Kp * V + k].v = NB[k] + B[k];
      c[Kp * V + k].i = Kp * V + k;
    }
    qsort(c, (size_t)(Kp * (V + 1)), sizeof(cand_t), cand_cmp); /* :846 */
    for (int64_t j = 0; j < K; ++j) {
      const int64_t ind = c[j].i;
      const int nonext = ind >= Kp * V;                        /* :849 */
      const int64_t src = nonext ? ind - Kp * V : ind / V;      /* :850-852 */
      const int64_t tok = ind % V;                              /* :853 */
      const int64_t plen = lens[src];
      ext_tok[j] = tok;
      for (int64_t s = 0; s < S + 1; ++s) /* :855-864 */
        y_next[(s * N + n) * W + j] = s < S ? y_prev[(s * N + n) * Kp + src] : 0;
      if (plen <= S && plen >= 0) y_next[(plen * N + n) * W + j] = tok;
      y_next_lens[n * W + j] = plen + (nonext ? 0 : 1);                      /* :865 */
      nb_probs_next[n * W + j] = nonext ? NB[src] : E[src * V + tok];        /* :868-872 */
      b_probs_next[n * W + j] = nonext ? B[src] : B[src] * 0.0f;             /* :875 */
      y_next_last[n * W + j] = nonext ? last[src] : tok;                     /* :878-880 */
      next_src[n * W + j] = src;
      next_is_nonext[n * W + j] = (uint8_t)nonext;
    }
    for (int64_t a = 0; a < K; ++a) /* :883-898 */
      for (int64_t b = 0; b < K; ++b) {
        const int64_t sa = next_src[n * W + a], sb = next_src[n * W + b];
        const int64_t la = y_next_lens[n * W + a], lb = y_next_lens[n * W + b];
        const int64_t pos = la - 1 < 0 ? 0 : la - 1;
        const int64_t tm = y_next[(pos * N + n) * W + b];
        const int ok = isp[sa * Kp + sb] && la <= lb &&
                       (next_is_nonext[n * W + a] || tm == ext_tok[a]);
        next_is_prefix[(n * W + a) * W + b] = (uint8_t)ok;
      }
    /* scrub the undefined tail of y_next so outputs are comparable */
    for (int64_t j = 0; j < K; ++j)
      for (int64_t s = y_next_lens[n * W + j]; s < S + 1; ++s) y_next[(s * N + n) * W + j] = 0;
    for (int64_t j = K; j < W; ++j) { /* :902-924 */
      for (int64_t s = 0; s < S + 1; ++s) y_next[(s * N + n) * W + j] = 0;
      y_next_last[n * W + j] = 0;
      y_next_lens[n * W + j] = 0;
      nb_probs_next[n * W + j] = -INFINITY;
      b_probs_next[n * W + j] = -INFINITY;
      next_is_nonext[n * W + j] = 0;
      next_src[n * W + j] = 0;
      for (int64_t b = 0; b < W; ++b) {
        next_is_prefix[(n * W + j) * W + b] = 0;
        next_is_prefix[(n * W + b) * W + j] = 0;
      }
    }
  }
  free(E);
  free(NB);
  free(B);
  free(last);
  free(c);
  free(ext_tok);
  return 0;
}

/* softmax over the last axis in float32 (logits.softmax(2), _decoding.py:1093).  The
 * normaliser is accumulated in double and rounded once: ATen sums float32 lanes pairwise
 * (error ~ sqrt(n / lanes) ulp), a sequential float32 sum would drift ~ sqrt(n) ulp away from
 * it for vocabularies in the thousands -- more than the 1e-5 parity tolerance after a few
 * dozen frames.  For n <= a few hundred the two are indistinguishable (goldens: n = 13). */
static void softmax_f32(const float *x, int64_t n, float *y) {
  float mx = -INFINITY;
  for (int64_t i = 0; i < n; ++i)
    if (x[i] > mx) mx = x[i];
  double acc = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    y[i] = expf(x[i] - mx);
    acc += (double)y[i];
  }
  const float s = (float)acc;
  for (int64_t i = 0; i < n; ++i) y[i] = y[i] / s;
}

/* CTCPrefixSearch.forward without a language model (_decoding.py:1064-1202).
 *   logits (T, N, V + 1) contiguous; lens (N) or NULL.
 *   y (S, N, width) with S = max(lens) (T if lens == NULL), y_lens (N, width),
 *   y_probs (N, width).  Entries of y beyond y_lens are 0. */
int pdt_oracle_ctc_prefix_search(const float *logits, int64_t T, int64_t N, int64_t Vp1,
                                 const int64_t *lens, int64_t width, int64_t *y,
                                 int64_t *y_lens, float *y_probs) {
  const int64_t V = Vp1 - 1, W = width;
  if (V < 1 || W < 1) return -1;
  int64_t S = T;
  if (lens) {
    S = 0;
    for (int64_t n = 0; n < N; ++n)
      if (lens[n] > S) S = lens[n];
  }
  float *probs = (float *)malloc(sizeof(float) * (size_t)Vp1);
  /* single-utterance state, ping-pong */
  int64_t *yp = (int64_t *)calloc((size_t)((S + 1) * W), sizeof(int64_t));
  int64_t *yn = (int64_t *)calloc((size_t)((S + 1) * W), sizeof(int64_t));
  int64_t *last = (int64_t *)malloc(sizeof(int64_t) * (size_t)W), *last2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)W);
  int64_t *ln = (int64_t *)malloc(sizeof(int64_t) * (size_t)W), *ln2 = (int64_t *)malloc(sizeof(int64_t) * (size_t)W);
  float *nb = (float *)malloc(sizeof(float) * (size_t)W), *nb2 = (float *)malloc(sizeof(float) * (size_t)W);
  float *bb = (float *)malloc(sizeof(float) * (size_t)W), *bb2 = (float *)malloc(sizeof(float) * (size_t)W);
  uint8_t *isp = (uint8_t *)malloc((size_t)(W * W)), *isp2 = (uint8_t *)malloc((size_t)(W * W));
  int64_t *src = (int64_t *)malloc(sizeof(int64_t) * (size_t)W);
  uint8_t *nonext = (uint8_t *)malloc((size_t)W);
  for (int64_t n = 0; n < N; ++n) {
    const int64_t Tn = lens ? lens[n] : T;
    int64_t Kp = 1; /* :1097-1105 */
    nb[0] = 0.0f;
    bb[0] = 1.0f;
    last[0] = 0;
    ln[0] = 0;
    isp[0] = 1;
    for (int64_t t = 0; t < Tn; ++t) {
      softmax_f32(logits + (t * N + n) * Vp1, Vp1, probs);
      /* compact (t, 1, Kp) history view: yp holds rows of width Kp */
      int rc = pdt_oracle_ctc_prefix_search_advance(
          probs, 0, 0, probs, probs + V, 1, Kp, V, W, nb, bb, yp, t, last, ln, isp, yn, last2,
          ln2, nb2, bb2, isp2, src, nonext);
      if (rc) return rc;
      memcpy(yp, yn, sizeof(int64_t) * (size_t)((t + 1) * W));
      memcpy(last, last2, sizeof(int64_t) * (size_t)W);
      memcpy(ln, ln2, sizeof(int64_t) * (size_t)W);
      memcpy(nb, nb2, sizeof(float) * (size_t)W);
      memcpy(bb, bb2, sizeof(float) * (size_t)W);
      memcpy(isp, isp2, (size_t)(W * W));
      Kp = W;
    }
    for (int64_t k = 0; k < W; ++k) {
      if (Kp == 1 && k > 0) { /* :1190-1200 */
        y_lens[n * W + k] = ln[0];
        y_probs[n * W + k] = -INFINITY;
      } else {
        y_lens[n * W + k] = ln[k];
        y_probs[n * W + k] = nb[k] + bb[k]; /* :1188 */
      }
      for (int64_t s = 0; s < S; ++s) {
        const int64_t kk = (Kp == 1) ? 0 : k;
        y[(s * N + n) * W + k] = (s < Tn && s < ln[kk]) ? yp[s * Kp + kk] : 0;
      }
    }
  }
  free(probs); free(yp); free(yn); free(last); free(last2); free(ln); free(ln2);
  free(nb); free(nb2); free(bb); free(bb2); free(isp); free(isp2); free(src); free(nonext);
  return 0;
}
