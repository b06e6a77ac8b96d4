"""Shared by the CPU (oracle) and GPU (kernels) suites: replay ``tests/golden/string_edge.npz`` --
what the LIVE reference returned, or the class of what it raised, on degenerate shapes of the string
operators (empty hypothesis / reference, one utterance; ``make_golden.py::string_edge_goldens``)."""
import builtins
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def replay(impl, to_arg, to_np):
    """``impl``: namespace with the five operators; ``to_arg`` makes its input from an int64 array,
    ``to_np`` turns its output into numpy.  Returns the number of checks made."""
    g = np.load(os.path.join(G, "string_edge.npz"))
    checked = 0

    def expect(tag, fn):
        nonlocal checked
        checked += 1
        if "err_" + tag in g.files:
            exc = getattr(builtins, str(g["err_" + tag]))
            try:
                fn()
            except exc:
                return
            raise AssertionError("{}: the reference raises {}".format(tag, exc.__name__))
        got, want = to_np(fn()), g[tag]
        assert got.shape == want.shape, (tag, got.shape, want.shape)
        assert got.dtype == want.dtype, (tag, got.dtype, want.dtype)
        assert np.array_equal(got, want), tag

    for i in range(len(g["shapes"])):
        ref, hyp = g["ref_{}".format(i)], g["hyp_{}".format(i)]
        for bf in (0, 1):
            a, b = (to_arg(ref.T), to_arg(hyp.T)) if bf else (to_arg(ref), to_arg(hyp))
            for ex in (0, 1):
                kw = dict(exclude_last=bool(ex), batch_first=bool(bf))
                t = "_{}_x{}_b{}".format(i, ex, bf)
                expect("oc" + t, lambda: impl.optimal_completion(a, b, **kw))
                expect("per" + t, lambda: impl.prefix_error_rates(a, b, **kw))
                expect("ped" + t, lambda: impl.prefix_edit_distances(a, b, ins_cost=2.0, del_cost=0.5, sub_cost=1.0, **kw))
            t = "_{}_b{}".format(i, bf)
            expect("er" + t, lambda: impl.error_rate(a, b, batch_first=bool(bf)))
            expect("ed" + t, lambda: impl.edit_distance(a, b, batch_first=bool(bf)))
    for j in range(3):
        ref, hyp = to_arg(g["eref_{}".format(j)]), to_arg(g["ehyp_{}".format(j)])
        for inc in (0, 1):
            for ex in (0, 1):
                tag = "{}_i{}_x{}".format(j, inc, ex)
                kw = dict(eos=0, include_eos=bool(inc), exclude_last=bool(ex))
                expect("eoc_" + tag, lambda: impl.optimal_completion(ref, hyp, **kw))
                expect("eper_" + tag, lambda: impl.prefix_error_rates(ref, hyp, **kw))
    return checked
