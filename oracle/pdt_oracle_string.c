/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C restatement of the batched Levenshtein family of pydrobert-pytorch
 * (reference: src/pydrobert/torch/_string.py).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  Parity status: PINNED --
 * checked bit-for-bit against the live reference in the build container
 * (tests/golden/make_golden.py) and against the reference's own known-answer tests
 * restated in tests/test_oracle_string.py.
 *
 * Two variants are provided:
 *   faithful=1  follows the reference's arithmetic literally, including the
 *               O(R^2)-per-row "deletion unroll" of _string.py:258-266,316-317,
 *               so float32 roundings agree for *any* costs.
 *   faithful=0  textbook O(H*R) recurrence; identical results
 *               whenever every partial sum is exactly representable in float32.
 *
 * All arithmetic is IEEE float32, no contraction (compile with -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pdt_oracle.h"

/* _string.py:137-143 -- index of first eos along the sequence axis, else T */
void pdt_oracle_lens_from_eos(const int64_t *tok, int64_t T, int64_t N, int64_t st,
                              int64_t sn, int64_t eos, int64_t *lens) {
  for (int64_t n = 0; n < N; ++n) {
    int64_t len = T;
    for (int64_t t = 0; t < T; ++t) {
      if (tok[t * st + n * sn] == eos) {
        len = t;
        break;
      }
    }
    lens[n] = len;
  }
}

/* _string.py:195-228 -- sequence lengths incl. the include_eos fix-ups.
 * returns 1 if some sequence lacked an eos although include_eos was set. */
static int seq_lens(const int64_t *tok, int64_t T, int64_t N, int64_t st, int64_t sn,
                    int has_eos, int64_t eos, int include_eos, int64_t *lens) {
  int missing = 0;
  if (!has_eos) {
    for (int64_t n = 0; n < N; ++n) lens[n] = T; /* :222-228 */
    return 0;
  }
  pdt_oracle_lens_from_eos(tok, T, N, st, sn, eos, lens);
  if (include_eos) {
    for (int64_t n = 0; n < N; ++n) {
      if (lens[n] == T)
        missing = 1; /* :199-208: +1 then -1 again */
      else
        lens[n] += 1;
    }
  }
  return missing;
}

int pdt_oracle_string_matching(const int64_t *ref, int64_t R, int64_t ref_st,
                               int64_t ref_sn, const int64_t *hyp, int64_t H,
                               int64_t hyp_st, int64_t hyp_sn, int64_t N, int has_eos,
                               int64_t eos, int include_eos, float ins_cost,
                               float del_cost, float sub_cost, int norm, int mode,
                               int exclude_last, float padding, int return_mistakes,
                               int faithful, float *out, uint8_t *mask_out,
                               int64_t *ref_lens, int64_t *hyp_lens, int *warn_flags) {
  if (R < 0 || H < 0 || N < 0) return -1;
  if (mode != PDT_MODE_FINAL && mode != PDT_MODE_PREFIX && mode != PDT_MODE_MASK)
    return -1;
  if (exclude_last && mode == PDT_MODE_FINAL) return -1; /* :165 */
  /* The loop bound of :286 and the number of OUTPUT rows are different things in mask mode:
   * the initial row mask is appended before the loop (:271-278), so a mask has max(1, Hloop)
   * rows -- one row even for H == 0 with exclude_last.  The reference indexes row 0 of a
   * (R, N) mask (:275) resp. of a (Hloop, N) prefix buffer (:285): IndexError when that
   * dimension is empty.  PDT_ORACLE_E_INDEX tells the caller to raise it; nothing is written. */
  const int64_t Hloop = H + (exclude_last ? 0 : 1); /* :281, :286 */
  const int64_t Hout = (mode == PDT_MODE_MASK && Hloop < 1) ? 1 : Hloop;
  if (mode == PDT_MODE_MASK && R == 0) return PDT_ORACLE_E_INDEX;
  if (mode == PDT_MODE_PREFIX && Hloop == 0) return PDT_ORACLE_E_INDEX;
  int flags = 0;
  float mult = 1.0f;
  /* :168-174 uniform-cost shortcut */
  if (ins_cost == del_cost && del_cost == sub_cost && sub_cost > 0.0f) {
    if (!return_mistakes) mult = ins_cost;
    ins_cost = del_cost = sub_cost = 1.0f;
    return_mistakes = 0;
  }
  int64_t *rl = ref_lens, *hl = hyp_lens;
  int own_rl = 0, own_hl = 0;
  if (!rl) {
    rl = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N ? N : 1));
    own_rl = 1;
  }
  if (!hl) {
    hl = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N ? N : 1));
    own_hl = 1;
  }
  if (seq_lens(ref, R, N, ref_st, ref_sn, has_eos, eos, include_eos, rl))
    flags |= PDT_WARN_REF_NO_EOS;
  if (seq_lens(hyp, H, N, hyp_st, hyp_sn, has_eos, eos, include_eos, hl))
    flags |= PDT_WARN_HYP_NO_EOS;

  const int64_t R1 = R + 1;
  float *row0 = (float *)malloc(sizeof(float) * (size_t)R1);
  float *row = (float *)malloc(sizeof(float) * (size_t)R1);
  float *last = (float *)malloc(sizeof(float) * (size_t)R1);
  float *tmp = (float *)malloc(sizeof(float) * (size_t)R1);
  float *mis = (float *)malloc(sizeof(float) * (size_t)R1);
  float *lmis = (float *)malloc(sizeof(float) * (size_t)R1);
  float *tmis = (float *)malloc(sizeof(float) * (size_t)R1);
  for (int64_t r = 0; r < R1; ++r) row0[r] = (float)r * del_cost; /* :258-263 */

  for (int64_t n = 0; n < N; ++n) {
    const int64_t ref_len = rl[n], hyp_len = hl[n];
    for (int64_t r = 0; r < R1; ++r) {
      row[r] = row0[r];
      mis[r] = (float)r; /* :260 */
    }
    if (mode == PDT_MODE_MASK) { /* :271-278 */
      for (int64_t r = 0; r < R; ++r) mask_out[(0 * R + r) * N + n] = 0; /* Hout >= 1, R >= 1 */
      mask_out[(0 * R + 0) * N + n] = ref_len > 0;
    } else if (mode == PDT_MODE_PREFIX) { /* :285 */
      out[0 * N + n] = (float)ref_len * (return_mistakes ? 1.0f : del_cost);
    }
    for (int64_t h = 1; h < Hloop; ++h) { /* :286 */
      const int not_done = (h - (exclude_last ? 0 : 1)) < hyp_len; /* :288 */
      const float ins_mask = hyp_len >= h ? 1.0f : 0.0f;            /* :290 */
      const int64_t hy = hyp[(h - 1) * hyp_st + n * hyp_sn];
      memcpy(last, row, sizeof(float) * (size_t)R1);
      memcpy(lmis, mis, sizeof(float) * (size_t)R1);
      const float ins_add = ins_cost * ins_mask;
      for (int64_t r = 0; r < R1; ++r) tmp[r] = last[r] + ins_add; /* :292 */
      if (return_mistakes) {
        /* :296-313 (cost, mistakes) pairs with sub < ins < del tie-break */
        for (int64_t r = 0; r < R1; ++r) tmis[r] = lmis[r] + ins_mask; /* :299 */
        for (int64_t r = 1; r < R1; ++r) {
          const float neq =
              (ref[(r - 1) * ref_st + n * ref_sn] != hy) ? 1.0f : 0.0f; /* :291 */
          const float sub_c = last[r - 1] + sub_cost * neq;               /* :293 */
          const float sub_m = lmis[r - 1] + neq;                          /* :300 */
          if (tmp[r] >= sub_c) {                                          /* :296 */
            tmp[r] = sub_c;
            tmis[r] = sub_m;
          }
        }
        for (int64_t r = 1; r < R1; ++r) { /* :307-313 */
          const float del_c = tmp[r - 1] + del_cost;
          if (!(del_c >= tmp[r])) {
            tmp[r] = del_c;
            tmis[r] = tmis[r - 1] + 1.0f;
          }
        }
        if (not_done) { /* :314, :318 */
          memcpy(row, tmp, sizeof(float) * (size_t)R1);
          memcpy(mis, tmis, sizeof(float) * (size_t)R1);
        }
      } else {
        for (int64_t r = 1; r < R1; ++r) { /* :316 */
          const float neq = (ref[(r - 1) * ref_st + n * ref_sn] != hy) ? 1.0f : 0.0f;
          const float sub_c = last[r - 1] + sub_cost * neq;
          if (sub_c < tmp[r]) tmp[r] = sub_c;
        }
        if (faithful) {
          /* :264-266,:317  row[r] = min_k (row0[r]-row0[k]) + tmp[k],  k <= r */
          for (int64_t r = 0; r < R1; ++r) {
            float best = INFINITY;
            for (int64_t k = 0; k <= r; ++k) {
              const float d = row0[r] - row0[k];
              const float v = d + tmp[k];
              if (v < best) best = v;
            }
            tmis[r] = best; /* scratch */
          }
          if (not_done) memcpy(row, tmis, sizeof(float) * (size_t)R1);
        } else {
          for (int64_t r = 1; r < R1; ++r) {
            const float del_c = tmp[r - 1] + del_cost;
            if (del_c < tmp[r]) tmp[r] = del_c;
          }
          if (not_done) memcpy(row, tmp, sizeof(float) * (size_t)R1);
        }
      }
      if (mode == PDT_MODE_MASK) { /* :332-339 */
        float mn = INFINITY;
        for (int64_t r = 0; r < R1; ++r) {
          if (r > ref_len) row[r] = INFINITY;
          if (row[r] < mn) mn = row[r];
        }
        for (int64_t r = 0; r < R; ++r)
          mask_out[(h * R + r) * N + n] = (uint8_t)((row[r] == mn) && not_done);
      } else if (mode == PDT_MODE_PREFIX) { /* :340-346 */
        out[h * N + n] = return_mistakes ? mis[ref_len] : row[ref_len];
      }
    }
    if (mode == PDT_MODE_MASK) { /* :349-354 */
      for (int64_t h = 0; h < Hout; ++h)
        for (int64_t r = ref_len; r < R; ++r) mask_out[(h * R + r) * N + n] = 0;
    } else if (mode == PDT_MODE_PREFIX) { /* :357-386 */
      for (int64_t h = 0; h < Hout; ++h) {
        float v = out[h * N + n] * mult;
        if (norm) {
          v = v / (float)ref_len;
          if (ref_len == 0) {
            flags |= PDT_WARN_EMPTY_REF;
            v = h > 0 ? 1.0f : 0.0f;
          }
        }
        if (h >= hyp_len + (exclude_last ? 0 : 1)) v = padding;
        out[h * N + n] = v;
      }
    } else { /* :390-405 */
      float er = (return_mistakes ? mis[ref_len] : row[ref_len]) * mult;
      if (norm) {
        er = er / (float)ref_len;
        if (ref_len == 0) {
          flags |= PDT_WARN_EMPTY_REF;
          er = hyp_len > 0 ? 1.0f : 0.0f;
        }
      }
      out[n] = er;
    }
  }
  free(row0);
  free(row);
  free(last);
  free(tmp);
  free(mis);
  free(lmis);
  free(tmis);
  if (own_rl) free(rl);
  if (own_hl) free(hl);
  if (warn_flags) *warn_flags = flags;
  return 0;
}

static int cmp_i64(const void *a, const void *b) {
  const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return (x > y) - (x < y);
}

/* _string.py:492-517 -- per (h, n) the ascending unique set of ref tokens under the
 * mask.  Two calls: targets == NULL returns C (the max set size, the reference's
 * counts.max().item() at :511); then with a (Hout, N, C) buffer it fills it. */
int64_t pdt_oracle_optimal_completion_from_mask(const uint8_t *mask, const int64_t *ref,
                                                int64_t R, int64_t ref_st,
                                                int64_t ref_sn, int64_t Hout, int64_t N,
                                                int64_t padding, int64_t C,
                                                int64_t *targets) {
  int64_t *buf = (int64_t *)malloc(sizeof(int64_t) * (size_t)(R ? R : 1));
  int64_t maxc = 0;
  for (int64_t h = 0; h < Hout; ++h) {
    for (int64_t n = 0; n < N; ++n) {
      int64_t cnt = 0;
      for (int64_t r = 0; r < R; ++r)
        if (mask[(h * R + r) * N + n]) buf[cnt++] = ref[r * ref_st + n * ref_sn];
      qsort(buf, (size_t)cnt, sizeof(int64_t), cmp_i64);
      int64_t u = 0;
      for (int64_t i = 0; i < cnt; ++i)
        if (i == 0 || buf[i] != buf[i - 1]) buf[u++] = buf[i];
      if (u > maxc) maxc = u;
      if (targets) {
        int64_t *dst = targets + (h * N + n) * C;
        for (int64_t i = 0; i < C; ++i) dst[i] = i < u ? buf[i] : padding;
      }
    }
  }
  free(buf);
  return maxc;
}
