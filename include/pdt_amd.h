/*
 * pdt_amd.h -- C ABI of libpdt_amd.so: MI355X (gfx950) kernels for the sequence-level
 * hot path of pydrobert-pytorch.
 *
 * The reference (sdrobert/pydrobert-pytorch) is pure Python on stock ATen ops and has
 * no FFI of its own; its boundary for this path is the Python functional API
 * (src/pydrobert/torch/functional.py:24-58).  Each entry point below is what a binding
 * for one of those functions calls; the citation names the reference code it replaces.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers that the
 *     library borrows for the duration of the call (never frees, never retains);
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised;
 *   - strides are in ELEMENTS: token (t, n) of a sequence tensor is tok[t*st + n*sn], so
 *     batch_first inputs/outputs are expressed by swapping strides, not by copies;
 *   - return value: 0 ok, <0 invalid argument (PDT_E_*), >0 a hipError_t;
 *   - `status` (optional, device int32) is OR-ed with PDT_WARN_* data-irregularity bits
 *     (the reference raises warnings.warn for them when warn=True).
 */
#ifndef PDT_AMD_H
#define PDT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDT_OK 0
#define PDT_E_ARG -1      /* malformed argument (null pointer, negative size, bad mode) */
#define PDT_E_TOO_LONG -2 /* a sequence dimension exceeds what the kernel supports */
#define PDT_E_UNSUPPORTED -3 /* this entry point has no kernel for the layout given; the general entry point has */

#define PDT_MODE_FINAL 0
#define PDT_MODE_PREFIX 1

#define PDT_WARN_REF_NO_EOS 1 /* _string.py:201-207 */
#define PDT_WARN_HYP_NO_EOS 2 /* _string.py:211-217 */
#define PDT_WARN_EMPTY_REF 4  /* _string.py:361-367, :398-404 */

/* Library / build identification. */
int pdt_amd_abi_version(void);

/* Run-time switches (kernel selection for comparisons; INTEGRATION.md "Switches").  Each is
 * initialised ONCE from the environment variable of the same name the first time the library
 * needs one -- nothing on a launch path calls getenv -- and can be changed / read afterwards.
 * Unknown names return PDT_E_ARG.  No counterpart in the reference (it has one route per operator). */
int pdt_amd_set_switch(const char *name, int value);
int pdt_amd_get_switch(const char *name, int *value);

/* ---------------------------------------------------------------------------------------
 * Batched Levenshtein: error_rate, edit_distance, prefix_error_rates,
 * prefix_edit_distances.  Replaces _string_matching (_string.py:146-406) for
 * return_mask=False, including _lens_from_eos (_string.py:137-143), the include_eos
 * fix-ups (:195-218), the uniform-cost shortcut (:168-174), normalisation and padding
 * (:356-405).
 *
 *   ref (R, N), hyp (H, N) int64 tokens addressed through element strides.
 *   mode FINAL : out[n * out_sn]                          (N,)      float32
 *   mode PREFIX: out[h * out_sh + n * out_sn], h < Hout   (Hout, N) float32,
 *                Hout = H + (exclude_last ? 0 : 1)
 *   return_mistakes: 1 = error-count semantics (error_rate / prefix_error_rates),
 *                    0 = cost semantics (edit_distance / prefix_edit_distances).
 *   ref_lens_out / hyp_lens_out (optional, (N,) int64) receive the sequence lengths.
 *   workspace (optional): pdt_lev_workspace_bytes(R, H, N) bytes of device memory, 256-byte
 *   aligned, contents irrelevant.  With it, unit (i.e. uniform) costs run on the bit-parallel
 *   kernels of lev_bitpar.hip; without it (NULL / too small), or when that function returns 0
 *   (a hypothesis longer than 1024 tokens), on the cell-by-cell kernels.  Same results.
 *   Cost semantics with costs that are NOT exact in float32 and R > 2048 need the workspace (the
 *   plain workgroup kernel of lev_generic.hip keeps its rows there): PDT_E_TOO_LONG without it.
 * ------------------------------------------------------------------------------------- */
int64_t pdt_lev_workspace_bytes(int64_t R, int64_t H, int64_t N);

int pdt_lev(const int64_t *ref, int64_t R, int64_t ref_st, int64_t ref_sn,
            const int64_t *hyp, int64_t H, int64_t hyp_st, int64_t hyp_sn, int64_t N,
            int has_eos, int64_t eos, int include_eos, float ins_cost, float del_cost,
            float sub_cost, int norm, int mode, int exclude_last, float padding,
            int return_mistakes, float *out, int64_t out_sh, int64_t out_sn,
            int64_t *ref_lens_out, int64_t *hyp_lens_out, int32_t *status, void *workspace,
            int64_t workspace_bytes, void *stream);

/* The same call for inputs a previous pdt_lev has already classified into `workspace`: same ref /
 * hyp contents, shapes and strides, same eos / include_eos, same workspace bytes untouched since,
 * issued on the same stream (or ordered after it).  On the bit-parallel path (unit costs) the
 * lengths, token classes and match tables are then read from the workspace instead of being rebuilt
 * -- half the work of a call; error_rate followed by prefix_error_rates on one (ref, hyp) pair is
 * the case (the Python host keeps the last workspace for exactly that, _string.py).  The warning
 * bits are those the classifying call reported; `status` is not written.  Everywhere else it is
 * pdt_lev. */
int pdt_lev_classified(const int64_t *ref, int64_t R, int64_t ref_st, int64_t ref_sn,
                       const int64_t *hyp, int64_t H, int64_t hyp_st, int64_t hyp_sn, int64_t N,
                       int has_eos, int64_t eos, int include_eos, float ins_cost, float del_cost,
                       float sub_cost, int norm, int mode, int exclude_last, float padding,
                       int return_mistakes, float *out, int64_t out_sh, int64_t out_sn,
                       int64_t *ref_lens_out, int64_t *hyp_lens_out, int32_t *status, void *workspace,
                       int64_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * fill_after_eos (_string.py:30-42): out = value, except that every position strictly after
 * the first `eos` along the sequence dimension holds the fill value.
 *   tokens int64, value / out of elem_bytes (1, 2, 4 or 8) per element, all contiguous and
 *   viewed as (outer, L, inner) around the sequence dimension; fill_bits = the fill value's bit
 *   pattern in the low elem_bytes bytes.
 * ------------------------------------------------------------------------------------- */
int pdt_fill_after_eos(const int64_t *tokens, int64_t outer, int64_t L, int64_t inner,
                       int64_t eos, const void *value, int64_t elem_bytes, int64_t fill_bits,
                       void *out, void *stream);

/* ---------------------------------------------------------------------------------------
 * Optimal completion, two phases (optimal_completion, _string.py:464-517; the DP is
 * _string_matching with return_mask=True, :271-278, :319-355).
 *
 * Phase 1, pdt_oc_mask: per utterance, ranks the distinct reference tokens
 * (class k = k-th smallest distinct token of ref[:ref_len]), runs the DP and for each
 * hypothesis prefix h records WHICH classes are optimal next tokens as a bitmask:
 *     bitmask[(h * N + n) * W + w], W = pdt_oc_mask_words(R) 32-bit words, bit k of the
 *     row set iff class k is in the completion set of prefix h.
 *     class_tokens[n * R + k] = token value of class k            (N, R) int64
 *     max_count (device int32, must be zeroed by the caller) = max set size = the
 *     reference's `C = counts.max().item()` (:511).
 *     workspace: pdt_oc_mask_workspace_bytes(R, H, N) bytes.  R <= 512: the tables of the bit-parallel
 *     kernel (uniform costs; a caller that passes no workspace, or other costs, gets the same masks
 *     from the row-synchronous kernel, whose DP row lives in registers); 512 < R <= 2048: 0; longer
 *     references run a plain one-workgroup-per-utterance form whose rows, sort buffer and class ids
 *     live there (costs that are not exact in float32 replay the reference's unrolled deletion term
 *     by term there as well, O(R^2) per row).
 * Phase 2, pdt_oc_expand (after the caller has read max_count and allocated targets):
 *     targets[h * tgt_sh + n * tgt_sn + i], i < C, ascending tokens then `padding`.
 * ------------------------------------------------------------------------------------- */
int64_t pdt_oc_mask_words(int64_t R);
int64_t pdt_oc_mask_workspace_bytes(int64_t R, int64_t H, int64_t N);

int pdt_oc_mask(const int64_t *ref, int64_t R, int64_t ref_st, int64_t ref_sn,
                const int64_t *hyp, int64_t H, int64_t hyp_st, int64_t hyp_sn, int64_t N,
                int has_eos, int64_t eos, int include_eos, float ins_cost, float del_cost,
                float sub_cost, int exclude_last, uint32_t *bitmask,
                int64_t *class_tokens, int32_t *max_count, int32_t *status, void *workspace,
                int64_t workspace_bytes, void *stream);

int pdt_oc_expand(const uint32_t *bitmask, const int64_t *class_tokens, int64_t R,
                  int64_t Hout, int64_t N, int64_t C, int64_t padding, int64_t *targets,
                  int64_t tgt_sh, int64_t tgt_sn, void *stream);

/* ---------------------------------------------------------------------------------------
 * Hard optimal-completion distillation loss, fused (hard_optimal_completion_distillation_loss,
 * _string.py:1188-1251).  Consumes pdt_oc_mask's outputs directly (exclude_last = 1, so the
 * bitmask has H rows) instead of the expanded (H, N, C) targets:
 *   loss[h*N + n]  = mean over the completion set S of  -weight[t] * log_softmax(logits[h,n])[t]
 *   count[h*N + n] = |S| (targets equal to ignore_index are skipped), loss uses max(count, 1).
 * logits (H, N, V) float32 through element strides; weight (V,) or NULL.
 * Backward: grad_logits (H, N, V) contiguous from grad_loss (H, N).
 * status (optional) bit 0 is set when a target lies outside [0, V).
 * ------------------------------------------------------------------------------------- */
int pdt_ocd_loss_forward(const float *logits, int64_t H, int64_t N, int64_t V, int64_t lg_sh,
                         int64_t lg_sn, int64_t lg_sv, const uint32_t *bitmask,
                         const int64_t *class_tokens, int64_t R, const float *weight,
                         int64_t ignore_index, float *loss, int32_t *count, int32_t *status,
                         void *stream);

int pdt_ocd_loss_backward(const float *logits, int64_t H, int64_t N, int64_t V, int64_t lg_sh,
                          int64_t lg_sn, int64_t lg_sv, const uint32_t *bitmask,
                          const int64_t *class_tokens, int64_t R, const float *weight,
                          int64_t ignore_index, const float *grad_loss, float *grad_logits,
                          void *stream);

/* ---------------------------------------------------------------------------------------
 * CTC prefix beam search with an N-GRAM language model in the loop, every frame from one launch
 * (reference _decoding.py:1064-1202 with shallow fusion / valid mixture, :1113-1135, around a
 * LookupLanguageModel, _lm.py:403-515, whose contexts -- U^(order - 1) of them -- fit a table).
 *
 *   pdt_lm_factor_table: the model's factor of the mix for every context token, from its scores
 *     lm_log_probs (rows, V) contiguous (row c = the model's scores after context token c):
 *     out[c][v] = exp(beta * log_softmax(lm_log_probs[c])[v])   (valid_mixture = 0; ext = p * out)
 *              or softmax(lm_log_probs[c])[v]                    (valid_mixture = 1;
 *                 ext = (1 - beta) * p + beta * (out * (1 - p_blank))) -- pdt_fusion_ext's expressions.
 *     Built once per model and mix by the host; rows out_stride floats apart.
 *   pdt_ctc_lm_table_search: logits / lens / width / S / y / y_lens / y_probs as
 *     pdt_ctc_prefix_search (the softmax of :1093 is fused); factors (contexts, V) the table above,
 *     factor_max (contexts,) the largest value of every row (it bounds a prefix's extension masses:
 *     lists are built only for prefixes whose extensions can be among a frame's winners);
 *     sos_row the row of the empty prefix's context.  A prefix's row is its last (order - 1) tokens read as
 *     digits in base ctx_base, start-of-sequence padding in front: an extension by token v moves row r to
 *     (r mod ctx_mod) * ctx_base + v, with ctx_mod = ctx_base^(order - 2) and contexts = ctx_base * ctx_mod
 *     (a bigram model: ctx_mod = 1, the row is the last token; a trigram model over U context symbols:
 *     ctx_base = ctx_mod = U, U^2 rows -- 4 GB at U = 1001, V = 1000: sized for this card's HBM, built once
 *     per model by the model's own scoring kernel over every context).
 *     width <= 32, V <= 5119.  workspace: pdt_ctc_lm_table_search_workspace_bytes (trie + checkpoints).
 * ------------------------------------------------------------------------------------- */
int pdt_lm_factor_table(const float *lm_log_probs, int64_t rows, int64_t V, float beta, int valid_mixture,
                        float *out, int64_t out_stride, void *stream);
int64_t pdt_ctc_lm_table_search_workspace_bytes(int64_t T, int64_t N, int64_t V, int64_t width);
int pdt_ctc_lm_table_search(const float *logits, int64_t T, int64_t N, int64_t V, int64_t lg_st, int64_t lg_sn,
                            int64_t lg_sv, const int64_t *lens, int64_t width, int64_t S, const float *factors,
                            const float *factor_max, int64_t contexts, int64_t f_stride, int64_t sos_row,
                            int64_t ctx_base, int64_t ctx_mod, float beta, int valid_mixture,
                            int64_t *y, int64_t *y_lens, float *y_probs, void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * CTC prefix beam search without a language model: CTCPrefixSearch(width)(logits, lens)
 * (reference _decoding.py:1064-1202; the per-frame step is ctc_prefix_search_advance,
 * :636-934; the softmax of :1093 is fused).
 *
 *   logits (T, N, V + 1) float32 through element strides, blank = index V.
 *   lens   (N,) int64 or NULL (all T).  S = number of rows of y = max(lens) (T if NULL).
 *   y      (S, N, width) int64 contiguous; every element is written (rows beyond a prefix's
 *          length are 0; the reference leaves them undefined);
 *   y_lens (N, width) int64, y_probs (N, width) float32 (probabilities, not logs).
 *   workspace: pdt_ctc_prefix_search_workspace_bytes(T, N, V, width) bytes of scratch
 *          (the prefix trie: one (parent, token) record per frame and beam entry, the
 *          checkpoints of the output walk and, for rows beyond the LDS, the rows of the ring).
 *   width <= 32 (wider beams: one pdt_ctc_prefix_search_advance per frame).  S must be at least min(T, max lens): frames beyond S are not decoded.
 *   Rows of up to 511 tokens and of 512 .. 16 447 tokens (contiguous logits) are held in the
 *   registers of the producer wave that reads them (ctc_search.hip / ctc_rowreg.hip); other rows
 *   of up to about 10 000 tokens live in LDS (one row of probabilities per slot of a three-slot
 *   ring); longer ones stay in the workspace (L2-resident), any V below 2^30.
 * pdt_ctc_prefix_search_plan (host only, no device work): the launch configuration the library
 *   picks for rows of V tokens and this width -- plan5 = {producer waves per utterance, ring
 *   slots, utterances per workgroup, where a row is held: 1 the producer's registers (short
 *   rows), 3 the same for long rows (the ring then carries lists, not rows), 0 LDS, 2 the
 *   workspace; for 3: 64-token register chunks of the instantiation, else 0}.  Lets callers and
 *   tests see where the configurations change.
 * ------------------------------------------------------------------------------------- */
int64_t pdt_ctc_prefix_search_workspace_bytes(int64_t T, int64_t N, int64_t V, int64_t width);

int pdt_ctc_prefix_search_plan(int64_t V, int64_t width, int32_t *plan5);

int pdt_ctc_prefix_search(const float *logits, int64_t T, int64_t N, int64_t V, int64_t lg_st,
                          int64_t lg_sn, int64_t lg_sv, const int64_t *lens, int64_t width,
                          int64_t S, int64_t *y, int64_t *y_lens, float *y_probs,
                          void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * ctc_prefix_search_advance (reference _decoding.py:636-934): one CTC prefix-search step with
 * per-prefix extension probabilities (shallow fusion) and dense histories.
 * Inputs through element strides (a stride of 0 expresses a broadcast):
 *   ext (N, Kp, V), nonext (N, V), blank (N,) float32 probabilities;
 *   nb_prev, b_prev (N, Kp) float32; y_prev (S, N, Kp) int64; y_prev_last, y_prev_lens
 *   (N, Kp) int64; prev_is_prefix (N, Kp, Kp) bool bytes.
 * Outputs, contiguous: y_next (S + 1, N, width) int64 (entries beyond y_next_lens
 *   unspecified, as in the reference); y_next_last, y_next_lens, next_src (N, width) int64;
 *   nb_next, b_next (N, width) float32; next_is_prefix (N, width, width) and next_is_nonext
 *   (N, width) bool bytes.  Kp or width above 32: the plain workgroup form (csrc/advance_wide.hip),
 *   same results; PDT_E_TOO_LONG once its LDS (about 56 Kp + 32 width bytes + 10 KB) exceeds 160 KB.
 * ------------------------------------------------------------------------------------- */
int pdt_ctc_prefix_search_advance(
    const float *ext, int64_t ext_sn, int64_t ext_sk, int64_t ext_sv, const float *nonext,
    int64_t ne_sn, int64_t ne_sv, const float *blank, int64_t bl_sn, int64_t N, int64_t Kp,
    int64_t V, int64_t width, const float *nb_prev, int64_t nb_sn, int64_t nb_sk,
    const float *b_prev, int64_t b_sn, int64_t b_sk, const int64_t *y_prev, int64_t S,
    int64_t yp_ss, int64_t yp_sn, int64_t yp_sk, const int64_t *y_prev_last, int64_t la_sn,
    int64_t la_sk, const int64_t *y_prev_lens, int64_t le_sn, int64_t le_sk,
    const uint8_t *prev_is_prefix, int64_t ip_sn, int64_t ip_sa, int64_t ip_sb, int64_t *y_next,
    int64_t *y_next_last, int64_t *y_next_lens, float *nb_next, float *b_next,
    uint8_t *next_is_prefix, int64_t *next_src, uint8_t *next_is_nonext, void *stream);

/* ---------------------------------------------------------------------------------------
 * The same step with the extension probabilities formed inside it (reference _decoding.py:1110-1135
 * in front of :636-934): lm_log_probs (N * Kp, V) float32 contiguous -- the language model's scores of
 * the frame, any normalisation -- are mixed with nonext / blank as pdt_fusion_ext mixes them
 * (shallow fusion p * exp(beta * log_softmax(lm)), or the valid mixture), to the bit, and never
 * written: what a call of pdt_fusion_ext followed by pdt_ctc_prefix_search_advance returns, in one
 * kernel.  PDT_E_UNSUPPORTED for V above 1024, Kp or width above 32, or LDS beyond 160 KB: the two
 * calls serve those.
 * ------------------------------------------------------------------------------------- */
int pdt_ctc_prefix_search_advance_lm(
    const float *lm_log_probs, float beta, int valid_mixture, const float *nonext, int64_t ne_sn,
    int64_t ne_sv, const float *blank, int64_t bl_sn, int64_t N, int64_t Kp, int64_t V, int64_t width,
    const float *nb_prev, int64_t nb_sn, int64_t nb_sk, const float *b_prev, int64_t b_sn, int64_t b_sk,
    const int64_t *y_prev, int64_t S, int64_t yp_ss, int64_t yp_sn, int64_t yp_sk,
    const int64_t *y_prev_last, int64_t la_sn, int64_t la_sk, const int64_t *y_prev_lens, int64_t le_sn,
    int64_t le_sk, const uint8_t *prev_is_prefix, int64_t ip_sn, int64_t ip_sa, int64_t ip_sb,
    int64_t *y_next, int64_t *y_next_last, int64_t *y_next_lens, float *nb_next, float *b_next,
    uint8_t *next_is_prefix, int64_t *next_src, uint8_t *next_is_nonext, void *stream);

/* ---------------------------------------------------------------------------------------
 * What a caller of pdt_beam_search_advance must know to choose S_out (reference _decoding.py:133-140):
 * *host_flag (a word in pinned host memory, valid once the stream has been synchronised) = bit 0: some
 * y_prev_lens[n, k] >= S; bit 1: some length is non-zero; bit 2: written (a system-scope release store: a
 * host that zeroed the word before the call may poll for it instead of synchronising).  One small kernel
 * and no device-to-host copy.
 * ------------------------------------------------------------------------------------- */
int pdt_lens_reach(const int64_t *lens, int64_t le_sn, int64_t le_sk, int64_t N, int64_t Kp, int64_t S,
                   int32_t *host_flag, void *stream);

/* ---------------------------------------------------------------------------------------
 * beam_search_advance (reference _decoding.py:41-155): one beam-search step.
 *   log_probs_t (N, Kp, V), log_probs_prev (N, Kp) float32; y_prev (S, N, Kp) int64;
 *   y_prev_lens (N, Kp) int64 or NULL (all S).  S_out = rows of y_next: S + 1 when
 *   y_prev_lens is NULL or some length equals S (the caller decides, :133-135), else S.
 * Outputs, contiguous: y_next (S_out, N, width), y_next_lens / next_src (N, width) int64,
 *   log_probs_next (N, width).  Slots beyond min(width, Kp * V) get -inf / length 0 / source 0.
 *   Kp or width above 64: the plain workgroup form (csrc/advance_wide.hip), same results.
 * ------------------------------------------------------------------------------------- */
int pdt_beam_search_advance(const float *log_probs_t, int64_t lt_sn, int64_t lt_sk, int64_t lt_sv,
                            int64_t N, int64_t Kp, int64_t V, int64_t width,
                            const float *log_probs_prev, int64_t lp_sn, int64_t lp_sk,
                            const int64_t *y_prev, int64_t S, int64_t yp_ss, int64_t yp_sn,
                            int64_t yp_sk, const int64_t *y_prev_lens, int64_t le_sn,
                            int64_t le_sk, int64_t S_out, int64_t *y_next, int64_t *y_next_lens,
                            float *log_probs_next, int64_t *next_src, void *stream);
/* ---------------------------------------------------------------------------------------
 * One iteration of BeamSearch.forward (_decoding.py:410-486 with the default
 * update_log_probs_for_step) as one kernel: eos bookkeeping, log_softmax of the language model's
 * scores (not materialised), eos-mass reallocation of ended paths, beam_search_advance, lengths
 * that do not grow for ended sources, finished batch elements keeping their beam.
 *   scores (N, Kp, V) float32 through element strides: the LM's output, any normalisation;
 *   log_probs_prev (N, Kp); y_prev (S, N, Kp) int64 with every token in [0, V - 1] (the clamped
 *   history the reference hands its LM, :434); y_prev_lens (N, Kp).
 *   has_eos / eos in [0, V); finish_all_paths; pad_value (written clamped, see pad_from).
 *   y_next (S + 1, N, width) int64 contiguous, y_next_lens / next_src (N, width) int64,
 *   log_probs_next (N, width) float32.
 *   active [1] int32: set to 1 if some batch element was NOT finished at the start of this
 *   iteration (the caller zeroes it; reading 0 means the reference would have left its loop before
 *   this iteration -- every later row of y is padding).  A flag, not a count: plain stores.
 *   pad_from (N,) int32, INT32_MAX initially: the first row of y that is padding for a finished
 *   element; the caller writes pad_value into rows >= pad_from[n] at the end.
 *   width, Kp <= 64.
 * ------------------------------------------------------------------------------------- */
/* ---------------------------------------------------------------------------------------
 * BeamSearch.forward over a bigram table, EVERY iteration in one launch (reference _decoding.py:383-504
 * with the default step hook; the table and its row statistics as for pdt_beam_search_step_table):
 * a workgroup runs its batch element's iterations back to back and stops at the one that finds the
 * element finished.  Starts from one empty path whose table row is sos_row.
 *   trie (N, n_iters, width) uint32: an iteration's (source << 20 | token) per beam entry;
 *   log_probs_out / lens_out (N, width): the final beam; finish (N,) int32: the iteration that found the
 *   element finished (n_iters: none); t_stop [1] int32, zeroed by the caller: max over the elements --
 *   the rows y has (the reference leaves its loop there).
 * pdt_beam_search_table_paths then writes y (T, N, width) int64 from the trie (T = t_stop): row s of a
 * path = the token its ancestor chose in iteration s, pad_value from finish[n] on.
 * PDT_E_UNSUPPORTED for rows of at most 64 tokens, width above 64 or width * ceil(V / 64) above 256: the
 * per-iteration entry points serve those.
 * ------------------------------------------------------------------------------------- */
int pdt_beam_search_table(const float *table, int64_t tb_sr, int64_t tb_sv, int64_t U, const float *row_stats,
                          int64_t sos_row, int64_t N, int64_t V, int64_t width, int64_t n_iters, int has_eos,
                          int64_t eos, int finish_all_paths, uint32_t *trie, float *log_probs_out,
                          int64_t *lens_out, int32_t *finish, int32_t *t_stop, void *stream);
int pdt_beam_search_table_paths(const uint32_t *trie, const int32_t *finish, int64_t N, int64_t n_iters,
                                int64_t width, int64_t T, int64_t pad_value, int64_t *y, void *stream);

/* ---------------------------------------------------------------------------------------
 * One frame of CTCPrefixSearch with a LookupLanguageModel in the loop as one kernel
 * (_decoding.py:1110-1163 around :636-934; scores: _lm.py:403-515): the back-off n-gram scores of
 * every prefix's context (its last max_ngram - 1 tokens, read from y_prev), shallow fusion
 * (valid_mixture = 0: ext = p_ctc * exp(beta * log_softmax(lm))) or the valid mixture, the
 * per-prefix sorted lists and the prefix step.  State arguments and outputs are those of
 * pdt_ctc_prefix_search_advance (without ext); the model's buffers are those of
 * pdt_lookup_lm_log_probs (the forward index is required); max_ngram >= 2.
 * history_bytes: 8 -- y_prev / y_next hold int64 tokens, as the step functions exchange them; 2 --
 * int16 tokens (V <= 32767): a caller that runs frame after frame keeps the (t, N, K) history in
 * this narrow form between its frames (copying it is what a long search pays per frame).
 * yn_ss / yn_sn / yn_sk: element strides of y_next, rows 0 .. S written ((S + 1, N, width) contiguous is
 * N * width, width, 1; a frame loop that keeps token-contiguous (N, width, Smax) int16 histories --
 * strides 1, width * Smax, Smax, Smax a multiple of 8 -- has them copied 16 bytes at a time).
 * frame_lens (N,) int64 or NULL, frame: a batch element with frame_lens[n] <= frame has no such frame
 * and keeps its beam -- y / lens / nb / b as they were (brought to `width`), one more row of zeros
 * (_decoding.py:1165-1181); its last tokens and is-prefix outputs are unspecified.
 * width, Kp <= 32; max_ngram <= 16.  Same bits as pdt_lookup_lm_log_probs -> pdt_fusion_ext ->
 * pdt_ctc_prefix_search_advance.
 * ------------------------------------------------------------------------------------- */
int pdt_ctc_lookup_lm_advance(
    const float *nonext, int64_t ne_sn, int64_t ne_sv, const float *blank, int64_t bl_sn, int64_t N, int64_t Kp,
    int64_t V, int64_t width, const float *nb_prev, int64_t nb_sn, int64_t nb_sk, const float *b_prev,
    int64_t b_sn, int64_t b_sk, const void *y_prev, int64_t S, int64_t yp_ss, int64_t yp_sn, int64_t yp_sk,
    const int64_t *y_prev_last, int64_t la_sn, int64_t la_sk, const int64_t *y_prev_lens, int64_t le_sn,
    int64_t le_sk, const uint8_t *prev_is_prefix, int64_t ip_sn, int64_t ip_sa, int64_t ip_sb,
    const float *logps, const float *logbs, const int32_t *child_start, const int32_t *ids,
    const int32_t *succ_start, const int32_t *succ_tok, const int32_t *succ_node, int64_t max_ngram, int64_t U,
    int64_t sos, float beta, int valid_mixture, void *y_next, int64_t *y_next_last, int64_t *y_next_lens,
    float *nb_next, float *b_next, uint8_t *next_is_prefix, int64_t *next_src, uint8_t *next_is_nonext,
    int history_bytes, const int64_t *frame_lens, int64_t frame, int64_t yn_ss, int64_t yn_sn, int64_t yn_sk,
    void *stream);

/* ---------------------------------------------------------------------------------------
 * The whole CTCPrefixSearch with a LookupLanguageModel in the loop (_decoding.py:1083-1202 with
 * :1110-1163 per frame): the frame kernel of pdt_ctc_lookup_lm_advance launched n_frames times
 * from this call, the beam's state kept in `workspace` between the frames.
 *   probs (T, N, V + 1) float32 through element strides: softmax of the logits, blank last; frames
 *   0 .. n_frames - 1 are read (n_frames >= 1).  frame_lens (N,) int64 or NULL: a batch element keeps
 *   its beam from frame frame_lens[n] on (:1165-1181).
 *   Histories are not copied from frame to frame: every batch element has 2 * width history slots;
 *   a prefix that survives a frame keeps its slot, an extended prefix gets a free slot, a copy of its
 *   source's tokens and the new token (int16 tokens when V <= 32767, int64 otherwise).
 *   Outputs, contiguous: y (n_frames, N, width) int64, zero beyond an entry's length and for absent
 *   entries; y_lens (N, width) int64; nb, b (N, width) float32 -- the two masses of every entry (the
 *   module returns nb + b).  Same bits as n_frames calls of pdt_ctc_lookup_lm_advance.
 *   The model's factor of the mix depends on the context only; for a bigram model whose table of
 *   U rows of V floats stays below 1 GiB the workspace keeps every row once it has been computed.
 *   workspace: pdt_ctc_lookup_lm_search_workspace_bytes(n_frames, N, V, width, max_ngram, U) bytes.
 * width <= 32; max_ngram <= 16.
 * ------------------------------------------------------------------------------------- */
int64_t pdt_ctc_lookup_lm_search_workspace_bytes(int64_t n_frames, int64_t N, int64_t V, int64_t width,
                                                 int64_t max_ngram, int64_t U);
int pdt_ctc_lookup_lm_search(
    const float *probs, int64_t p_st, int64_t p_sn, int64_t p_sv, const int64_t *frame_lens, int64_t n_frames,
    int64_t N, int64_t V, int64_t width, const float *logps, const float *logbs, const int32_t *child_start,
    const int32_t *ids, const int32_t *succ_start, const int32_t *succ_tok, const int32_t *succ_node,
    int64_t max_ngram, int64_t U, int64_t sos, float beta, int valid_mixture, int64_t *y, int64_t *y_lens,
    float *nb, float *b, void *workspace, int64_t workspace_bytes, void *stream);

int pdt_beam_search_step(const float *scores, int64_t sc_sn, int64_t sc_sk, int64_t sc_sv, int64_t N,
                         int64_t Kp, int64_t V, int64_t width, const float *log_probs_prev,
                         int64_t lp_sn, int64_t lp_sk, const int64_t *y_prev, int64_t S,
                         int64_t yp_ss, int64_t yp_sn, int64_t yp_sk, const int64_t *y_prev_lens,
                         int64_t le_sn, int64_t le_sk, int has_eos, int64_t eos, int finish_all_paths,
                         int64_t pad_value, int64_t *y_next, int64_t *y_next_lens,
                         float *log_probs_next, int64_t *next_src, int32_t *active,
                         int32_t *pad_from, void *stream);


/* Table form of pdt_beam_search_step: the scores of prefix (n, k) are row rows[n * Kp + k] (int64,
 * in [0, U)) of a (U, V) table -- e.g. the rows of a bigram LookupLanguageModel by context token,
 * computed once -- and, if row_stats (U, 2) is given (pdt_row_log_softmax_stats of the same table), a
 * row's maximum and log-sum-exp are read from it instead of being recomputed every iteration.
 * Everything else as pdt_beam_search_step; the same bits. */
int pdt_beam_search_step_table(const float *table, int64_t tb_sr, int64_t tb_sv, int64_t U,
                               const float *row_stats, const int64_t *rows, int64_t N, int64_t Kp,
                               int64_t V, int64_t width, const float *log_probs_prev, int64_t lp_sn,
                               int64_t lp_sk, const int64_t *y_prev, int64_t S, int64_t yp_ss,
                               int64_t yp_sn, int64_t yp_sk, const int64_t *y_prev_lens, int64_t le_sn,
                               int64_t le_sk, int has_eos, int64_t eos, int finish_all_paths,
                               int64_t pad_value, int64_t *y_next, int64_t *y_next_lens,
                               float *log_probs_next, int64_t *next_src, int32_t *active,
                               int32_t *pad_from, void *stream);
/* stats[2 r], stats[2 r + 1] = max_v table[r][v], log sum_v exp(table[r][v] - max) */
int pdt_row_log_softmax_stats(const float *table, int64_t tb_sr, int64_t tb_sv, int64_t U, int64_t V,
                              float *stats, void *stream);

/* ---------------------------------------------------------------------------------------
 * ctc_greedy_search (reference _decoding.py:507-558).  logits (T, N, V) through element
 * strides; blank_idx already normalised to [0, V).  max_out (N,): sum of the per-frame maximum
 * log-probabilities (is_probs: product of the maxima, no normalisation) over frames below
 * in_lens; paths (T, N) int64 through strides: the first out_lens[n] entries of column n are
 * the collapsed tokens, the rest the raw per-frame arg-max (as the reference leaves them).
 *
 * sequence_log_probs on tensors (_decoding.py:1516-1551).  hyp viewed as (A, S, B) with S the
 * step axis, logits (A, S, B, V) contiguous; out (A, B) = sum over valid steps of
 * log_softmax(logits)[hyp]; tokens outside [0, V) and steps after the first eos are skipped.
 * Backward: grad_logits (same layout) from grad_out (A, B).
 * ------------------------------------------------------------------------------------- */
int pdt_ctc_greedy_search(const float *logits, int64_t T, int64_t N, int64_t V, int64_t lg_st,
                          int64_t lg_sn, int64_t lg_sv, const int64_t *in_lens, int64_t blank_idx,
                          int is_probs, float *max_out, int64_t *paths, int64_t pa_st,
                          int64_t pa_sn, int64_t *out_lens, void *stream);

int pdt_sequence_log_probs_forward(const float *logits, const int64_t *hyp, int64_t A, int64_t S,
                                   int64_t B, int64_t V, int has_eos, int64_t eos, float *out,
                                   void *stream);

int pdt_sequence_log_probs_backward(const float *logits, const int64_t *hyp, int64_t A, int64_t S,
                                    int64_t B, int64_t V, int has_eos, int64_t eos,
                                    const float *grad_out, float *grad_logits, void *stream);

/* ---------------------------------------------------------------------------------------
 * Feature augmentation / image warps (reference _img.py).  All tensors float32 and
 * contiguous unless strides are given.
 *
 * pdt_polyharmonic_spline: polyharmonic_spline (_img.py:59-150).  train_points (N,T,I),
 *   train_values (N,T,O), query_points (N,Q,I) -> out (N,Q,O).  The bordered system is solved
 *   exactly (float64, partial pivoting; in LDS up to ~140 unknowns, in the workspace beyond),
 *   which covers both of the reference's `full_matrix` evaluation orders.
 *   workspace: pdt_spline_workspace_bytes(N,T,I,O) bytes.
 * pdt_spline_solve: the solve alone, with an optional right-hand-side tail: solution (N, T+I+1, O)
 *   float64 of [[phi(|c_i - c_j|) + reg I, [c 1]], [[c 1]^T, 0]] X = [train_values; tail]
 *   (tail (N, I+1, O) or NULL = zeros).  The spline's weights (_img.py:79-130), and -- the matrix
 *   being symmetric -- the adjoint system of its backward pass.
 * pdt_warp_1d_grid: warp_1d_grid (_img.py:268-303): src, flow, lengths (N,) -> grid (N,T).
 * pdt_spec_augment_apply: spec_augment_apply_parameters (_img.py:1142-1211).  feats (N,T,F)
 *   through element strides; time_grid (N,T) / freq_grid (N,F) normalised sampling grids or
 *   NULL (no warp along that axis); t_0/t_len (N,MT), f_0/f_len (N,MF) int64 mask bands
 *   (MT / MF = 0: none); out (N,T,F) contiguous.
 * pdt_dense_image_warp: dense_image_warp (_img.py:393-439).  image (N,C,H,W), flow (N,H,W,2);
 *   flow_is_hw: last dim of flow is (h, w) ("hw" indexing) instead of (w, h).
 *   mode: 0 bilinear, 1 nearest; padding: 0 zeros, 1 border, 2 reflection.
 * pdt_sparse_image_warp: sparse_image_warp (_img.py:520-714) after the caller has appended the
 *   pinned boundary points and put points in (x=w, y=h) order: train_points (N,M,2) = dest
 *   points; train_values (N,M,2) = dest - source (flow form) or the normalised source grid
 *   (values_are_grid = 1, the include_flow=False form).  flow_out (N,H,W,2) optional
 *   (flow form only), stored (h, w)-ordered when flow_out_is_hw.
 *   workspace: pdt_spline_workspace_bytes(N, M, 2, 2) bytes.
 * pdt_*_backward: adjoints of the three resampling operators with respect to the features /
 *   image (what autograd derives through grid_sample in the reference, _img.py:436, :1203):
 *   grad_out has the forward output's shape, contiguous; grad_feats / grad_image (contiguous)
 *   is overwritten.  The sampling arguments are the forward call's.  Accumulation uses the
 *   hardware float atomic, so sums are not bit-reproducible from run to run.
 * ------------------------------------------------------------------------------------- */
int64_t pdt_spline_workspace_bytes(int64_t N, int64_t T, int64_t I, int64_t O);

int pdt_spline_solve(const float *train_points, const float *train_values, const float *tail,
                     int64_t N, int64_t T, int64_t I, int64_t O, int order,
                     float regularization_weight, double *solution, void *workspace, void *stream);

int pdt_polyharmonic_spline(const float *train_points, const float *train_values,
                            const float *query_points, int64_t N, int64_t T, int64_t I, int64_t O,
                            int64_t Q, int order, float regularization_weight, float *out,
                            void *workspace, void *stream);

int pdt_warp_1d_grid(const float *src, const float *flow, const float *lengths, int64_t N, int64_t T,
                     int order, float *grid, void *stream);

/* spec_augment_draw_parameters (_img.py:1056-1139) from one (N, R) tensor of uniform draws u in
 * [0, 1): column c is utterance n's c-th draw in the reference's order (w_0, w | v_0, v | t x MT,
 * t_0 x MT | f x MF, f_0 x MF; groups the configuration disables take no columns and their outputs
 * may be NULL); every parameter is the reference's float32 expression of its uniform.  lengths (N,)
 * int64 or NULL (all T).  is_double: the features are float64 (the reference's eps follows them). */
int pdt_spec_augment_draw(const float *u, int64_t N, int64_t R, const int64_t *lengths, int64_t T,
                          int64_t F, float max_time_warp, float max_freq_warp, int64_t max_time_mask,
                          int64_t max_freq_mask, float max_time_mask_proportion, int64_t num_time_mask,
                          float num_time_mask_proportion, int64_t num_freq_mask, int is_double,
                          float *w_0, float *w, float *v_0, float *v, int64_t *t_0, int64_t *t,
                          int64_t *f_0, int64_t *f, void *stream);

int pdt_spec_augment_apply(const float *feats, int64_t N, int64_t T, int64_t F, int64_t f_sn,
                           int64_t f_st, int64_t f_sf, const float *time_grid,
                           const float *freq_grid, const int64_t *t_0, const int64_t *t_len,
                           int64_t MT, const int64_t *f_0, const int64_t *f_len, int64_t MF,
                           float *out, void *stream);

/* spec_augment_apply_parameters with the TIME WARP given by its parameters (w_0 = warp_src, w =
 * warp_flow, float (N,); lengths int64 (N,) or NULL = all T; interpolation order) instead of a grid:
 * warp_1d_grid's three-knot spline (_img.py:283-302) is solved in closed form (float64) and evaluated
 * inside the one pass over the features -- one launch, no (N, T) grid in memory.  No frequency warp.
 * PDT_E_UNSUPPORTED when the one-pass kernel does not take the layout (F not a multiple of 4 or above
 * 256, strided or unaligned rows): form the grid with pdt_warp_1d_grid and call pdt_spec_augment_apply.
 * bad_lengths (optional): an int32 in device-visible memory (e.g. pinned host memory) the caller has
 * set to 0; an utterance whose length is outside [1, T] stores 1 there -- the reference's
 * "values of lengths must be between (1, T)" check (_img.py:1037-1041) decided where the lengths are
 * read, for the caller to look at once the stream has drained (the result is then to be dropped). */
int pdt_spec_augment_apply_warp(const float *feats, int64_t N, int64_t T, int64_t F, int64_t f_sn,
                                int64_t f_st, int64_t f_sf, const float *warp_src, const float *warp_flow,
                                const int64_t *lengths, int order, const int64_t *t_0,
                                const int64_t *t_len, int64_t MT, const int64_t *f_0,
                                const int64_t *f_len, int64_t MF, float *out, int32_t *bad_lengths,
                                void *stream);

int pdt_dense_image_warp(const float *image, const float *flow, int64_t N, int64_t C, int64_t H,
                         int64_t W, int flow_is_hw, int mode, int padding, float *out,
                         void *stream);

int pdt_sparse_image_warp(const float *image, const float *train_points,
                          const float *train_values, int64_t N, int64_t C, int64_t H, int64_t W,
                          int64_t M, int order, float regularization_weight, int values_are_grid,
                          int mode, int padding, float *out, float *flow_out, int flow_out_is_hw,
                          void *workspace, void *stream);

/* float64 images: the reference samples a double image on a double grid (_img.py:423-436) while
 * flows and spline points stay float32 (:420, :537-538).  Same arguments as the float32 entries. */
int pdt_dense_image_warp_f64(const double *image, const float *flow, int64_t N, int64_t C, int64_t H,
                             int64_t W, int flow_is_hw, int mode, int padding, double *out,
                             void *stream);
int pdt_dense_image_warp_backward_f64(const double *grad_out, const float *flow, int64_t N, int64_t C,
                                      int64_t H, int64_t W, int flow_is_hw, int mode, int padding,
                                      double *grad_image, void *stream);
int pdt_sparse_image_warp_f64(const double *image, const float *train_points,
                              const float *train_values, int64_t N, int64_t C, int64_t H, int64_t W,
                              int64_t M, int order, float regularization_weight, int values_are_grid,
                              int mode, int padding, double *out, float *flow_out, int flow_out_is_hw,
                              void *workspace, void *stream);
int pdt_sparse_image_warp_backward_f64(const double *grad_out, const float *train_points,
                                       const float *train_values, int64_t N, int64_t C, int64_t H,
                                       int64_t W, int64_t M, int order, float regularization_weight,
                                       int values_are_grid, int mode, int padding, double *grad_image,
                                       void *workspace, void *stream);

int pdt_spec_augment_apply_backward(const float *grad_out, int64_t N, int64_t T, int64_t F,
                                    const float *time_grid, const float *freq_grid,
                                    const int64_t *t_0, const int64_t *t_len, int64_t MT,
                                    const int64_t *f_0, const int64_t *f_len, int64_t MF,
                                    float *grad_feats, void *stream);

int pdt_dense_image_warp_backward(const float *grad_out, const float *flow, int64_t N, int64_t C,
                                  int64_t H, int64_t W, int flow_is_hw, int mode, int padding,
                                  float *grad_image, void *stream);

int pdt_sparse_image_warp_backward(const float *grad_out, const float *train_points,
                                   const float *train_values, int64_t N, int64_t C, int64_t H,
                                   int64_t W, int64_t M, int order, float regularization_weight,
                                   int values_are_grid, int mode, int padding, float *grad_image,
                                   void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------
 * Back-off n-gram language model scoring (reference _lm.py:403-515, LookupLanguageModel
 * :518-1110), the on-device LM of CTCPrefixSearch / BeamSearch with shallow fusion.
 *
 * The model is the reference's reverse trie (:609-677) in widened form: logps [P] and
 * logbs [O] float32 as stored; child_start [O] int32 = node index + offsets[node] (absolute
 * index of the node's first child; the children of i are [child_start[i], child_start[i+1]));
 * ids [I] int32 = labels of the nodes >= U (index node - U).  V vocabulary size, N >= 2 the
 * n-gram order, U = V + shift + 1 the unigram count + dummy (shift = 1 when sos is not a
 * vocabulary id; it is then stored as unigram V).
 *
 * hist (S, B) int64 through element strides.  idx != NULL: rows == B, row b is queried at
 * position idx[b * idx_stride] in [0, S] (idx_stride 0 = one shared position) --
 * calc_idx_log_probs (:769-790).  idx == NULL: rows == (S + 1) * B, row t * B + b is batch
 * element b at position t -- calc_full_log_probs (:793-848).  Positions before the start of
 * the history read sos (:452-461).  out (rows, V) float32: log P(v | the N - 1 tokens before
 * the position).  status bit 0: a position outside [0, S] (clamped).
 *
 * succ_start [U + 1], succ_tok [E], succ_node [E] int32 (all three or NULL): a forward index of
 * the trie's second level, derived from the buffers above -- for a context token c, entries
 * succ_start[c] .. succ_start[c + 1] list the last tokens v (ascending) and nodes of the bigrams
 * "c v".  With it a row searches the children of the listed successors only (same results).
 * ------------------------------------------------------------------------------------- */
int pdt_lookup_lm_log_probs(const int64_t *hist, int64_t S, int64_t B, int64_t h_ss, int64_t h_sb,
                            const int64_t *idx, int64_t idx_stride, int64_t rows,
                            const float *logps, const float *logbs, const int32_t *child_start,
                            const int32_t *ids, const int32_t *succ_start, const int32_t *succ_tok,
                            const int32_t *succ_node, int64_t V, int64_t N, int64_t U, int64_t sos,
                            float *out, int32_t *status, void *stream);

/* ---------------------------------------------------------------------------------------
 * Extension probabilities of one frame of CTCPrefixSearch with a language model (reference
 * _decoding.py:1110-1135), fused into one pass over the LM scores:
 *   lm_log_probs (N, Kp, V) float32 contiguous (unnormalised scores, as the LM returns them),
 *   nonext (N, V) / blank (N,) the frame's CTC probabilities through element strides;
 *   valid_mixture == 0: out = nonext * exp(beta * log_softmax(lm_log_probs))        (shallow fusion)
 *   valid_mixture != 0: out = (1 - beta) * nonext + beta * softmax(lm_log_probs) * (1 - blank)
 *   out (N, Kp, V) float32 contiguous.  No gradient (the differentiable path composes torch ops).
 * ------------------------------------------------------------------------------------- */
int pdt_fusion_ext(const float *lm_log_probs, int64_t N, int64_t Kp, int64_t V, const float *nonext,
                   int64_t ne_sn, int64_t ne_sv, const float *blank, int64_t bl_sn, float beta,
                   int valid_mixture, float *out, void *stream);

/* ---------------------------------------------------------------------------------------
 * Variable-length padding (reference _pad.py:108-149), the data movement of RandomShift
 * (_img.py:883-908).  x (N, T, F) contiguous, elements of elem_bytes in {1, 2, 4, 8} moved as
 * opaque words; lens (N,), pad (2, N) int64; out (N, Tp, F) with Tp >= max(lens + pad sums):
 *   out[n, :pad[0,n]]                          left padding
 *   out[n, pad[0,n] : pad[0,n] + lens[n]]      x[n, :lens[n]]
 *   ... + pad[1,n]                             right padding;  the rest: *fill
 * mode 0 constant (*fill), 1 reflect (x[n, pad0 - t], x[n, lens - 2 - j]; the caller checks
 * pad < lens), 2 replicate (x[n, 0], x[n, lens - 1]; the caller checks lens >= 1).
 * pdt_pad_variable_backward: float32 adjoint with respect to x, written as a gather
 * (deterministic); grad_out (N, Tp, F), grad_x (N, T, F).
 * ------------------------------------------------------------------------------------- */
int pdt_pad_variable(const void *x, int64_t N, int64_t T, int64_t F, int64_t elem_bytes,
                     const int64_t *lens, const int64_t *pad, int mode, const void *fill,
                     int64_t Tp, void *out, void *stream);

int pdt_pad_variable_backward(const float *grad_out, int64_t N, int64_t T, int64_t F,
                              const int64_t *lens, const int64_t *pad, int mode, int64_t Tp,
                              float *grad_x, void *stream);

#ifdef __cplusplus
}
#endif
#endif
