// Every run-time switch of libpdt_amd.so in one place.  The PDT_* environment variables are read
// ONCE, the first time an entry point asks (no getenv on a launch path); pdt_amd_set_switch /
// pdt_amd_get_switch (include/pdt_amd.h) change / read one afterwards -- that is how the tests run two
// routes in one process.  The table of names is in INTEGRATION.md ("Switches").
#pragma once

namespace pdt {

struct Switches {
  int lev_bitpar;      // PDT_LEV_BITPAR     1  bit-parallel distance kernels for unit costs (0: cell-by-cell kernels)
  int oc_bitpar;       // PDT_OC_BITPAR      1  bit-parallel optimal-completion mask (0: row-synchronous kernel)
  int oc_waves;        // PDT_OC_WAVES       0  waves per workgroup of the expansion kernel (0: by shape; 4 or 8)
  int ctc_exact_div;   // PDT_CTC_EXACT_DIV  0  probabilities as the IEEE quotient e / sum (1) instead of e * (1 / sum)
  int ctc_rowreg;      // PDT_CTC_ROWREG     1  rows of 320+ tokens of the CTC search held in the producers' registers (0: LDS ring of rows; 2: from 128 tokens)
  int step_wide;       // PDT_STEP_WIDE      0  step functions always on the radix-select kernels (1)
  int lm_cache;        // PDT_LM_CACHE       0  ctc_lm_step.hip's search: bigram factor rows kept per context across workgroups behind a
                       //                       relaxed flag (outside the HIP memory model; comparisons only -- bigram models take
                       //                       ctc_lm_table.hip, whose table is built before the launch)
  int lm_persistent;   // PDT_LM_PERSISTENT  1  n-gram search: every frame in one launch (0: a launch per frame)
  int warp_bands;      // PDT_WARP_BANDS     1  sparse_image_warp: a lane = a column of four rows (0: four pixels 256 apart)
  int lm_step_waves;   // PDT_LM_STEP_WAVES  0  waves per utterance of the n-gram frame kernel (0: by shape; 1, 2, 4 or 8)
  int ctc_pair;        // PDT_CTC_PAIR       1  CTC search at V = 256, W = 16: the producer takes two frames per pass, one per half wave (0: one; same bits)
  int step_flat;       // PDT_STEP_FLAT      1  step functions: one selection over all K' * V candidates (beam) / one list for prefixes that share
                       //                       their extension row (CTC) (0: a sorted list per prefix; same results)
  int ctc_lean_extra;  // PDT_CTC_LEAN_EXTRA 1  CTC frame: one-prefix and tie frames decided beside the lean tier (0: by the full tiers, same results)
};

Switches &switches();

}  // namespace pdt
