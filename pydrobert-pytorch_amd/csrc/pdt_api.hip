// extern "C" entry points of libpdt_amd.so (see include/pdt_amd.h): argument validation,
// the host-side part of _string_matching's preamble (uniform-cost shortcut,
// reference _string.py:168-174) and kernel selection.  No allocation, no synchronisation.
#include <cmath>
#include <algorithm>
#include <cstdlib>

#include "lev_common.hpp"
#include "switches.hpp"
#include <cstring>

namespace pdt {
int launch_lev_skewed(LevArgs a, hipStream_t stream);
int launch_lev_rowsync(LevArgs a, bool exact, hipStream_t stream);
int launch_lev_bitpar(const LevArgs &la, const BitparPlan &p, void *ws, hipStream_t stream, bool classified);
int launch_oc_mask_generic(const LevArgs &a, bool inexact, void *ws, int64_t ws_bytes, hipStream_t stream);
int launch_oc_mask_bitpar(const LevArgs &a, void *ws, int64_t ws_bytes, hipStream_t stream);
int64_t oc_bitpar_workspace_bytes(int64_t R, int64_t H, int64_t N);
int launch_oc_expand_generic(const uint32_t *bitmask, const int64_t *class_tokens, int R, int Hout,
                             int64_t N, int C, int64_t padding, int64_t *targets, int64_t tgt_sh,
                             int64_t tgt_sn, hipStream_t stream);
int64_t generic_oc_ws_per_utt(int64_t R, int64_t H, int *P_out);
int launch_oc_expand(const uint32_t *bitmask, const int64_t *class_tokens, int R, int Hout,
                     int64_t N, int C, int64_t padding, int64_t *targets, int64_t tgt_sh,
                     int64_t tgt_sn, hipStream_t stream);

// True when every partial sum the DP can form is an integer multiple of 2^-q below 2^24 in
// those units, i.e. exactly representable in float32.  Then the textbook recurrence and the
// reference's unrolled deletion matrix (_string.py:258-266) agree bit for bit.
static bool costs_exact_in_f32(float ins, float del, float sub, int64_t R, int64_t H) {
  const float c[3] = {ins, del, sub};
  float mx = 0.0f;
  for (float v : c) {
    if (!std::isfinite(v)) return false;
    mx = std::fmax(mx, std::fabs(v));
  }
  for (int q = 0; q <= 20; ++q) {
    bool ok = true;
    for (float v : c) {
      const float s = std::ldexp(v, q);
      ok = ok && (s == std::floor(s));
    }
    if (ok) {
      const double bound = (double)(R + H + 2) * (double)std::ldexp(mx, q);
      return bound < 16777216.0;
    }
  }
  return false;
}

// The bit-parallel kernels (lev_bitpar.hip) serve unit costs when the caller passed the workspace
// their plan asks for; PDT_LEV_BITPAR=0 keeps the cell-by-cell kernels (for comparisons).
static bool bitpar_enabled() { return switches().lev_bitpar != 0; }

// PDT_OC_BITPAR=0 keeps optimal completion on the row-synchronous kernel (the tests run both in one
// process through pdt_amd_set_switch).
static bool oc_bitpar_enabled() { return switches().oc_bitpar != 0; }

namespace {
struct SwitchName {
  const char *name;
  int Switches::*field;
  int deft;
};
const SwitchName kSwitchNames[] = {
    {"PDT_LEV_BITPAR", &Switches::lev_bitpar, 1},       {"PDT_OC_BITPAR", &Switches::oc_bitpar, 1},
    {"PDT_OC_WAVES", &Switches::oc_waves, 0},           {"PDT_CTC_EXACT_DIV", &Switches::ctc_exact_div, 0},
    {"PDT_CTC_ROWREG", &Switches::ctc_rowreg, 1},       {"PDT_STEP_WIDE", &Switches::step_wide, 0},
    {"PDT_LM_CACHE", &Switches::lm_cache, 0},           {"PDT_LM_PERSISTENT", &Switches::lm_persistent, 1},
    {"PDT_LM_STEP_WAVES", &Switches::lm_step_waves, 0}, {"PDT_WARP_BANDS", &Switches::warp_bands, 1},
    {"PDT_CTC_LEAN_EXTRA", &Switches::ctc_lean_extra, 1}, {"PDT_CTC_PAIR", &Switches::ctc_pair, 1},
    {"PDT_STEP_FLAT", &Switches::step_flat, 1},
};
}  // namespace

Switches &switches() {
  static Switches s = [] {
    Switches v{};
    for (const SwitchName &n : kSwitchNames) {
      const char *e = std::getenv(n.name);
      v.*(n.field) = (e && e[0]) ? std::atoi(e) : n.deft;
    }
    return v;
  }();
  return s;
}

static int fill_common(LevArgs &a, const int64_t *ref, int64_t R, int64_t ref_st, int64_t ref_sn,
                       const int64_t *hyp, int64_t H, int64_t hyp_st, int64_t hyp_sn, int64_t N,
                       int has_eos, int64_t eos, int include_eos, float ins, float del, float sub) {
  if (R < 0 || H < 0 || N < 0) return PDT_E_ARG;
  if (N > 0 && ((R > 0 && !ref) || (H > 0 && !hyp))) return PDT_E_ARG;
  if (R > (1 << 28) || H > (1 << 28) || N > (1ll << 31) - 1) return PDT_E_TOO_LONG;
  a = LevArgs{};
  a.ref = ref; a.hyp = hyp;
  a.ref_st = ref_st; a.ref_sn = ref_sn; a.hyp_st = hyp_st; a.hyp_sn = hyp_sn;
  a.R = (int)R; a.H = (int)H; a.N = (int)N;
  a.has_eos = has_eos; a.eos = eos; a.include_eos = include_eos;
  a.ins = ins; a.del = del; a.sub = sub; a.mult = 1.0f;
  return PDT_OK;
}
}  // namespace pdt

extern "C" {

// 2: pdt_ctc_prefix_search_workspace_bytes takes V; 3: pdt_lev takes a workspace;
// 4: pdt_lookup_lm_log_probs takes the forward index of the trie's second level
// 5: pdt_lev_classified
// 6: pdt_ctc_lookup_lm_search; pdt_oc_mask takes a workspace for references of up to 512 tokens;
//    pdt_beam_search_step_table, pdt_row_log_softmax_stats
// 7: pdt_amd_set_switch / pdt_amd_get_switch; the four-utterances-per-wave CTC form left the library
// 9: pdt_spec_augment_apply_warp; pdt_ctc_lm_table_search takes ctx_base / ctx_mod; PDT_E_UNSUPPORTED
int pdt_amd_abi_version(void) { return 10; }

int pdt_amd_set_switch(const char *name, int value) {
  if (!name) return PDT_E_ARG;
  for (const pdt::SwitchName &n : pdt::kSwitchNames)
    if (std::strcmp(n.name, name) == 0) return pdt::switches().*(n.field) = value, PDT_OK;
  return PDT_E_ARG;
}

int pdt_amd_get_switch(const char *name, int *value) {
  if (!name || !value) return PDT_E_ARG;
  for (const pdt::SwitchName &n : pdt::kSwitchNames)
    if (std::strcmp(n.name, name) == 0) return *value = pdt::switches().*(n.field), PDT_OK;
  return PDT_E_ARG;
}

int64_t pdt_lev_workspace_bytes(int64_t R, int64_t H, int64_t N) {
  if (R < 0 || H < 0 || N <= 0) return 0;
  int64_t need = 0;
  if (pdt::bitpar_enabled()) {
    const pdt::BitparPlan p = pdt::plan_bitpar(H, R, N);
    if (p.ok) need = (int64_t)p.total;
  }
  // references beyond the row-synchronous kernel's 2048 columns: costs that are inexact in float32
  // take the plain workgroup kernel (lev_generic.hip), whose rows live in the workspace
  if (R > 64 * 32) need = std::max(need, pdt::generic_oc_ws_per_utt(R, H, nullptr) * N);
  return need;
}

static int lev_entry(const int64_t *ref, int64_t R, int64_t ref_st, int64_t ref_sn, const int64_t *hyp,
                     int64_t H, int64_t hyp_st, int64_t hyp_sn, int64_t N, int has_eos, int64_t eos,
                     int include_eos, float ins_cost, float del_cost, float sub_cost, int norm, int mode,
                     int exclude_last, float padding, int return_mistakes, float *out, int64_t out_sh,
                     int64_t out_sn, int64_t *ref_lens_out, int64_t *hyp_lens_out, int32_t *status,
                     void *workspace, int64_t workspace_bytes, void *stream, bool classified) {
  using namespace pdt;
  if (mode != PDT_MODE_FINAL && mode != PDT_MODE_PREFIX) return PDT_E_ARG;
  if (exclude_last && mode != PDT_MODE_PREFIX) return PDT_E_ARG;  // _string.py:165
  if (exclude_last && H == 0) return PDT_E_ARG;
  if (N > 0 && !out) return PDT_E_ARG;
  // _string.py:168-174: uniform costs run with unit costs; the result is rescaled only for
  // the cost flavours, and the error-count flavour degenerates to the plain recurrence
  float mult = 1.0f;
  if (ins_cost == del_cost && del_cost == sub_cost && sub_cost > 0.0f) {
    if (!return_mistakes) mult = ins_cost;
    ins_cost = del_cost = sub_cost = 1.0f;
    return_mistakes = 0;
  }
  LevArgs a;
  int rc = fill_common(a, ref, R, ref_st, ref_sn, hyp, H, hyp_st, hyp_sn, N, has_eos, eos,
                       include_eos, ins_cost, del_cost, sub_cost);
  if (rc != PDT_OK) return rc;
  if (N == 0) return PDT_OK;
  a.mult = mult;
  a.norm = norm; a.mode = mode; a.exclude_last = exclude_last; a.count = return_mistakes;
  a.padding = padding;
  a.out = out; a.out_sh = out_sh; a.out_sn = out_sn;
  a.ref_lens_out = ref_lens_out; a.hyp_lens_out = hyp_lens_out; a.status = status;
  if (!return_mistakes && !costs_exact_in_f32(ins_cost, del_cost, sub_cost, R, H)) {
    if (R > 64 * 32) return launch_oc_mask_generic(a, /*inexact=*/true, workspace, workspace_bytes, (hipStream_t)stream);
    return launch_lev_rowsync(a, /*exact=*/true, (hipStream_t)stream);
  }
  if (!return_mistakes && ins_cost == 1.0f && del_cost == 1.0f && sub_cost == 1.0f && workspace &&
      bitpar_enabled()) {
    const BitparPlan p = plan_bitpar(H, R, N);
    if (p.ok && (int64_t)p.total <= workspace_bytes)
      return launch_lev_bitpar(a, p, workspace, (hipStream_t)stream, classified);
  }
  return launch_lev_skewed(a, (hipStream_t)stream);
}

int pdt_lev(const int64_t *ref, int64_t R, int64_t ref_st, int64_t ref_sn, const int64_t *hyp,
            int64_t H, int64_t hyp_st, int64_t hyp_sn, int64_t N, int has_eos, int64_t eos,
            int include_eos, float ins_cost, float del_cost, float sub_cost, int norm, int mode,
            int exclude_last, float padding, int return_mistakes, float *out, int64_t out_sh,
            int64_t out_sn, int64_t *ref_lens_out, int64_t *hyp_lens_out, int32_t *status,
            void *workspace, int64_t workspace_bytes, void *stream) {
  return lev_entry(ref, R, ref_st, ref_sn, hyp, H, hyp_st, hyp_sn, N, has_eos, eos, include_eos, ins_cost,
                   del_cost, sub_cost, norm, mode, exclude_last, padding, return_mistakes, out, out_sh, out_sn,
                   ref_lens_out, hyp_lens_out, status, workspace, workspace_bytes, stream, false);
}

int pdt_lev_classified(const int64_t *ref, int64_t R, int64_t ref_st, int64_t ref_sn, const int64_t *hyp,
                       int64_t H, int64_t hyp_st, int64_t hyp_sn, int64_t N, int has_eos, int64_t eos,
                       int include_eos, float ins_cost, float del_cost, float sub_cost, int norm, int mode,
                       int exclude_last, float padding, int return_mistakes, float *out, int64_t out_sh,
                       int64_t out_sn, int64_t *ref_lens_out, int64_t *hyp_lens_out, int32_t *status,
                       void *workspace, int64_t workspace_bytes, void *stream) {
  return lev_entry(ref, R, ref_st, ref_sn, hyp, H, hyp_st, hyp_sn, N, has_eos, eos, include_eos, ins_cost,
                   del_cost, sub_cost, norm, mode, exclude_last, padding, return_mistakes, out, out_sh, out_sn,
                   ref_lens_out, hyp_lens_out, status, workspace, workspace_bytes, stream, true);
}

int64_t pdt_oc_mask_words(int64_t R) { return R <= 0 ? 1 : (R + 31) / 32; }

int64_t pdt_oc_mask_workspace_bytes(int64_t R, int64_t H, int64_t N) {
  if (H < 0 || N <= 0) return 0;
  if (R <= 64 * 32)  // the bit-parallel kernel's tables (unit costs, up to 512 columns); the
    return pdt::oc_bitpar_enabled() ? pdt::oc_bitpar_workspace_bytes(R, H, N) : 0;  // register-resident kernel needs none
  return pdt::generic_oc_ws_per_utt(R, H, nullptr) * N;
}

int pdt_oc_mask(const int64_t *ref, int64_t R, int64_t ref_st, int64_t ref_sn, const int64_t *hyp,
                int64_t H, int64_t hyp_st, int64_t hyp_sn, int64_t N, int has_eos, int64_t eos,
                int include_eos, float ins_cost, float del_cost, float sub_cost, int exclude_last,
                uint32_t *bitmask, int64_t *class_tokens, int32_t *max_count, int32_t *status,
                void *workspace, int64_t workspace_bytes, void *stream) {
  using namespace pdt;
  if (exclude_last && H == 0) return PDT_E_ARG;
  if (N > 0 && (!bitmask || !class_tokens || !max_count)) return PDT_E_ARG;
  // optimal_completion never asks for mistakes (_string.py:479-491), so uniform costs only
  // rescale the row and leave the arg-min set unchanged: run with unit costs
  if (ins_cost == del_cost && del_cost == sub_cost && sub_cost > 0.0f)
    ins_cost = del_cost = sub_cost = 1.0f;
  LevArgs a;
  int rc = fill_common(a, ref, R, ref_st, ref_sn, hyp, H, hyp_st, hyp_sn, N, has_eos, eos,
                       include_eos, ins_cost, del_cost, sub_cost);
  if (rc != PDT_OK) return rc;
  if (N == 0) return PDT_OK;
  a.exclude_last = exclude_last;
  a.mode = -1;
  a.bitmask = bitmask; a.class_tokens = class_tokens; a.max_count = max_count;
  a.status = status;
  a.W = (int)pdt_oc_mask_words(R);
  const bool exact = !costs_exact_in_f32(ins_cost, del_cost, sub_cost, R, H);
  if (a.W > 64) {  // beyond the 2048 columns the row-synchronous kernel holds: the plain formulation
    return launch_oc_mask_generic(a, exact, workspace, workspace_bytes, (hipStream_t)stream);
  }
  if (ins_cost == 1.0f && del_cost == 1.0f && sub_cost == 1.0f && oc_bitpar_enabled()) {
    rc = launch_oc_mask_bitpar(a, workspace, workspace_bytes, (hipStream_t)stream);
    if (rc >= 0) return rc;  // (-1: shape not served)
  }
  return launch_lev_rowsync(a, exact, (hipStream_t)stream);
}

int pdt_oc_expand(const uint32_t *bitmask, const int64_t *class_tokens, int64_t R, int64_t Hout,
                  int64_t N, int64_t C, int64_t padding, int64_t *targets, int64_t tgt_sh,
                  int64_t tgt_sn, void *stream) {
  if (R < 0 || Hout < 0 || N < 0 || C < 0) return PDT_E_ARG;
  if (N == 0 || Hout == 0 || C == 0) return PDT_OK;
  if (!bitmask || !class_tokens || !targets) return PDT_E_ARG;
  if (R > 64 * 32)
    return pdt::launch_oc_expand_generic(bitmask, class_tokens, (int)R, (int)Hout, N, (int)C, padding, targets,
                                         tgt_sh, tgt_sn, (hipStream_t)stream);
  return pdt::launch_oc_expand(bitmask, class_tokens, (int)R, (int)Hout, N, (int)C, padding,
                               targets, tgt_sh, tgt_sn, (hipStream_t)stream);
}

}  // extern "C"
