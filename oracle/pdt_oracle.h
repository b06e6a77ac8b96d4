/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 * CPU restatement of the pydrobert-pytorch sequence-level hot path, used as the
 * parity checker for the HIP kernels.  Never linked into the product library.
 */
#ifndef PDT_ORACLE_H
#define PDT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* output modes of _string_matching (reference _string.py:146-163) */
#define PDT_MODE_FINAL 0  /* one value per utterance             */
#define PDT_MODE_PREFIX 1 /* return_prf_dsts: one value per hyp prefix */
#define PDT_MODE_MASK 2   /* return_mask: row-minimum mask (optimal completion) */

/* data-irregularity flags (the reference emits warnings.warn for each) */
#define PDT_WARN_REF_NO_EOS 1 /* _string.py:201-207 */
#define PDT_WARN_HYP_NO_EOS 2 /* _string.py:211-217 */
#define PDT_WARN_EMPTY_REF 4  /* _string.py:361-367, :398-404 */

/* return codes: 0 ok, -1 invalid argument, -2 the reference raises IndexError on this shape */
#define PDT_ORACLE_E_INDEX (-2)

void pdt_oracle_lens_from_eos(const int64_t *tok, int64_t T, int64_t N, int64_t st,
                              int64_t sn, int64_t eos, int64_t *lens);

/* Strides are in elements: token (t, n) lives at tok[t * st + n * sn].
 * out: (N,) for FINAL, (Hout, N) row-major for PREFIX; mask_out: (Hout, R, N) bytes for
 * MASK, Hout = H + (exclude_last ? 0 : 1).  ref_lens / hyp_lens / warn_flags optional. */
int pdt_oracle_string_matching(const int64_t *ref, int64_t R, int64_t ref_st,
                               int64_t ref_sn, const int64_t *hyp, int64_t H,
                               int64_t hyp_st, int64_t hyp_sn, int64_t N, int has_eos,
                               int64_t eos, int include_eos, float ins_cost,
                               float del_cost, float sub_cost, int norm, int mode,
                               int exclude_last, float padding, int return_mistakes,
                               int faithful, float *out, uint8_t *mask_out,
                               int64_t *ref_lens, int64_t *hyp_lens, int *warn_flags);

int64_t pdt_oracle_optimal_completion_from_mask(const uint8_t *mask, const int64_t *ref,
                                                int64_t R, int64_t ref_st,
                                                int64_t ref_sn, int64_t Hout, int64_t N,
                                                int64_t padding, int64_t C,
                                                int64_t *targets);

/* ---- decoding (pdt_oracle_decoding.c; reference _decoding.py) ---- */
int64_t pdt_oracle_beam_search_advance(const float *log_probs_t, int64_t N, int64_t Kp,
                                       int64_t V, int64_t width, const float *log_probs_prev,
                                       const int64_t *y_prev, int64_t S,
                                       const int64_t *y_prev_lens, int64_t *y_next,
                                       int64_t *y_next_lens, float *log_probs_next,
                                       int64_t *next_src);

int pdt_oracle_ctc_prefix_search_advance(
    const float *ext_probs_t, int64_t ext_sn, int64_t ext_sk, const float *nonext_probs_t,
    const float *blank_probs_t, int64_t N, int64_t Kp, int64_t V, int64_t width,
    const float *nb_probs_prev, const float *b_probs_prev, const int64_t *y_prev, int64_t S,
    const int64_t *y_prev_last, const int64_t *y_prev_lens, const uint8_t *prev_is_prefix,
    int64_t *y_next, int64_t *y_next_last, int64_t *y_next_lens, float *nb_probs_next,
    float *b_probs_next, uint8_t *next_is_prefix, int64_t *next_src, uint8_t *next_is_nonext);

int pdt_oracle_ctc_prefix_search(const float *logits, int64_t T, int64_t N, int64_t Vp1,
                                 const int64_t *lens, int64_t width, int64_t *y,
                                 int64_t *y_lens, float *y_probs);

#ifdef __cplusplus
}
#endif
#endif
