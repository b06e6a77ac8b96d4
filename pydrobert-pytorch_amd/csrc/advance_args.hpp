// Argument blocks of the step kernels (beam_advance.hip: the wave forms; advance_wide.hip: beams
// wider than a wave).
#pragma once
#include "pdt_common.hpp"

namespace pdt {

struct CtcAdvArgs {
  const float *ext;     int64_t ext_sn, ext_sk, ext_sv;   // (N, Kp, V)
  const float *nonext;  int64_t ne_sn, ne_sv;             // (N, V)
  const float *blank;   int64_t bl_sn;                    // (N,)
  const float *nb_prev; const float *b_prev; int64_t pb_sn, pb_sk, pbb_sn, pbb_sk;  // (N, Kp)
  const int64_t *y_prev; int64_t yp_ss, yp_sn, yp_sk;     // (S, N, Kp)
  const int64_t *last;  int64_t la_sn, la_sk;             // (N, Kp)
  const int64_t *lens;  int64_t le_sn, le_sk;             // (N, Kp)
  const uint8_t *isp;   int64_t ip_sn, ip_sa, ip_sb;      // (N, Kp, Kp) bool
  int N, Kp, V, W, S;
  // outputs, contiguous
  int64_t *y_next;      // (S + 1, N, W)
  int64_t *y_next_last, *y_next_lens, *next_src;  // (N, W)
  float *nb_next, *b_next;                        // (N, W)
  uint8_t *next_isp;                              // (N, W, W)
  uint8_t *next_nonext;                           // (N, W)
  int frame_bytes, waves_per_wg;  // LDS of the frame routine (the per-wave survivor scratch follows it)
  int ext_shared;                 // ext_sk == 0: every prefix reads the same row of extension probabilities
  // the fused form (pdt_ctc_prefix_search_advance_lm): no ext -- the language model's scores (N * Kp, V),
  // contiguous, mixed with the frame's probabilities on the fly (fusion_ext.hip's arithmetic)
  const float *lm;
  float beta;
  int valid_mixture;
};

struct BeamAdvArgs {
  const float *lpt;     int64_t lt_sn, lt_sk, lt_sv;   // log_probs_t (N, Kp, V)
  const float *lpp;     int64_t lp_sn, lp_sk;          // log_probs_prev (N, Kp)
  const int64_t *y_prev; int64_t yp_ss, yp_sn, yp_sk;  // (S, N, Kp)
  const int64_t *lens;  int64_t le_sn, le_sk;          // (N, Kp) or null
  int N, Kp, V, W, S, S_out;
  int64_t *y_next;      // (S_out, N, W)
  int64_t *y_next_lens, *next_src;  // (N, W)
  float *lp_next;                   // (N, W)
  int waves_per_wg;
};

int launch_ctc_advance_wide(CtcAdvArgs a, hipStream_t stream);
int launch_beam_advance_wide(BeamAdvArgs a, hipStream_t stream);

}  // namespace pdt
