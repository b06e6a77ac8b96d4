#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the LIVE reference.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONPATH=/root/reference/src python tests/golden/make_golden.py

Every fixture is DATA: the inputs fed to the reference and the outputs it returned, stored
as .npz.  The tfa_*.npy files next to this script are the TensorFlow-addons arrays the
reference's own tests ship (tests/polyharmonic_spline, tests/dense_image_warp,
tests/sparse_image_warp), copied verbatim.  sclite.npz is the NIST sclite known answer of
tests/sclite (50 utterances, costs 3/3/4) turned into token-id arrays with the reference's
own .trn reader.
"""
import os
import sys
import warnings

import numpy as np
import torch

REF = os.environ.get("PDT_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src"))
HERE = os.path.dirname(os.path.abspath(__file__))

import pydrobert.torch.functional as F  # noqa: E402
from pydrobert.torch import modules as M  # noqa: E402

warnings.simplefilter("ignore")


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


def string_goldens():
    rng = np.random.default_rng(0x5EED0001)
    # G-S1: config-1 shape, several cost triples
    ref = rng.integers(0, 32, (128, 64))
    hyp = rng.integers(0, 32, (128, 64))
    costs = [(1, 1, 1), (2, 2, 2), (3, 3, 4), (2, 0.5, 1), (0.1, 0.7, 1.3)]
    d = dict(ref=ref, hyp=hyp, costs=np.array(costs, np.float64))
    tr, th = torch.from_numpy(ref), torch.from_numpy(hyp)
    for i, c in enumerate(costs):
        for norm in (False, True):
            kw = dict(norm=norm, ins_cost=c[0], del_cost=c[1], sub_cost=c[2], warn=False)
            d["er_c{}_n{}".format(i, int(norm))] = F.error_rate(tr, th, **kw)
            d["ed_c{}_n{}".format(i, int(norm))] = F.edit_distance(tr, th, **kw)
    save("string_s1", **d)
    # G-S2/S3/S4: ragged, tie-rich, with empty ref / hyp rows
    N, R, H, V = 48, 40, 37, 6
    ref = rng.integers(0, V, (R, N))
    hyp = rng.integers(0, V, (H, N))
    rl = rng.integers(0, R + 1, N)
    hl = rng.integers(0, H + 1, N)
    rl[:3], hl[:3] = [0, 5, 0], [4, 0, 0]
    for n in range(N):
        if rl[n] < R:
            ref[rl[n], n] = V
        if hl[n] < H:
            hyp[hl[n], n] = V
    d = dict(ref=ref, hyp=hyp, eos=np.array(V))
    tr, th = torch.from_numpy(ref), torch.from_numpy(hyp)
    for inc in (False, True):
        for bf in (False, True):
            a, b = (tr.t().contiguous(), th.t().contiguous()) if bf else (tr, th)
            tag = "i{}_b{}".format(int(inc), int(bf))
            kw = dict(eos=V, include_eos=inc, batch_first=bf, warn=False)
            d["er_" + tag] = F.error_rate(a, b, ins_cost=3.0, del_cost=3.0, sub_cost=4.0, **kw)
            d["ed_" + tag] = F.edit_distance(a, b, ins_cost=2.0, del_cost=0.5, sub_cost=1.0, **kw)
            for ex in (False, True):
                for norm in (False, True):
                    t2 = tag + "_x{}_n{}".format(int(ex), int(norm))
                    d["per_" + t2] = F.prefix_error_rates(
                        a, b, norm=norm, exclude_last=ex, padding=-100, ins_cost=1.0, del_cost=2.0,
                        sub_cost=3.0, **kw)  # fmt: skip
                    d["ped_" + t2] = F.prefix_edit_distances(
                        a, b, norm=norm, exclude_last=ex, padding=-7, **kw)  # fmt: skip
                d["oc_" + tag + "_x{}".format(int(ex))] = F.optimal_completion(
                    a, b, exclude_last=ex, padding=-100, **kw)  # fmt: skip
    save("string_s2", **d)


def string_edge_goldens():
    """Degenerate shapes of the string operators (round 5): an EMPTY hypothesis / reference, a batch of
    one.  The reference's optimal_completion appends the initial row mask before its loop
    (_string.py:271-278, :286), so ``H == 0, exclude_last=True`` returns ONE row; shapes on which it
    raises are recorded as the exception's class name (``err_*``)."""
    d = {}
    shapes = [(3, 0, 2), (5, 0, 1), (0, 3, 2), (0, 0, 2), (1, 1, 1), (3, 3, 1), (4, 2, 1), (2, 5, 3), (1, 0, 4)]
    d["shapes"] = np.array(shapes)
    ops = dict(
        oc=lambda r, h, ex, bf: F.optimal_completion(r, h, exclude_last=ex, batch_first=bf, warn=False),
        per=lambda r, h, ex, bf: F.prefix_error_rates(r, h, exclude_last=ex, batch_first=bf, warn=False),
        ped=lambda r, h, ex, bf: F.prefix_edit_distances(r, h, exclude_last=ex, batch_first=bf, ins_cost=2.0,
                                                         del_cost=0.5, sub_cost=1.0, warn=False),
    )
    finals = dict(
        er=lambda r, h, bf: F.error_rate(r, h, batch_first=bf, warn=False),
        ed=lambda r, h, bf: F.edit_distance(r, h, batch_first=bf, warn=False),
    )
    for i, (R, H, N) in enumerate(shapes):
        ref = (torch.arange(R * N).view(R, N) * 7 + 1) % 4 + 1
        hyp = (torch.arange(H * N).view(H, N) * 5 + 2) % 3 + 1
        d["ref_{}".format(i)], d["hyp_{}".format(i)] = ref, hyp
        for bf in (False, True):
            a, b = (ref.t().contiguous(), hyp.t().contiguous()) if bf else (ref, hyp)
            for name, fn in ops.items():
                for ex in (False, True):
                    tag = "{}_{}_x{}_b{}".format(name, i, int(ex), int(bf))
                    try:
                        d[tag] = fn(a, b, ex, bf)
                    except Exception as e:  # noqa: BLE001 -- the class is the datum
                        d["err_" + tag] = np.array(type(e).__name__)
            for name, fn in finals.items():
                tag = "{}_{}_b{}".format(name, i, int(bf))
                try:
                    d[tag] = fn(a, b, bf)
                except Exception as e:  # noqa: BLE001
                    d["err_" + tag] = np.array(type(e).__name__)
    # one utterance with an eos: lengths 0 and full among the cases
    rng = np.random.default_rng(0x5EED0051)
    for j, (R, H) in enumerate([(6, 4), (1, 7), (7, 1)]):
        ref = torch.from_numpy(rng.integers(0, 3, (R, 1)))
        hyp = torch.from_numpy(rng.integers(0, 3, (H, 1)))
        d["eref_{}".format(j)], d["ehyp_{}".format(j)] = ref, hyp
        for inc in (False, True):
            for ex in (False, True):
                tag = "{}_i{}_x{}".format(j, int(inc), int(ex))
                kw = dict(eos=0, include_eos=inc, exclude_last=ex, warn=False)
                d["eoc_" + tag] = F.optimal_completion(ref, hyp, **kw)
                d["eper_" + tag] = F.prefix_error_rates(ref, hyp, **kw)
    save("string_edge", **d)


def sclite_golden():
    from pydrobert.torch._parsing import read_trn

    d = os.path.join(REF, "tests", "sclite")
    tok2id = {}
    with open(os.path.join(d, "token2id.txt")) as f:
        for line in f:
            t, i = line.split()
            tok2id[t] = int(i)
    refs = dict(read_trn(os.path.join(d, "ref.trn")))
    hyps = dict(read_trn(os.path.join(d, "hyp.trn")))
    utts = sorted(refs)
    per_utt = {}
    with open(os.path.join(d, "per_utt.txt")) as f:
        for line in f:
            u, v = line.split()
            per_utt[u] = float(v)
    with open(os.path.join(d, "total.txt")) as f:
        total = float(f.read().strip())

    def to_ids(seq):
        return [tok2id[t] for t in seq]

    R = max(len(refs[u]) for u in utts)
    H = max(len(hyps[u]) for u in utts)
    ref = np.full((R + 1, len(utts)), -1, np.int64)
    hyp = np.full((H + 1, len(utts)), -1, np.int64)
    for n, u in enumerate(utts):
        r, h = to_ids(refs[u]), to_ids(hyps[u])
        ref[: len(r), n] = r
        hyp[: len(h), n] = h
    errs = F.error_rate(torch.from_numpy(ref), torch.from_numpy(hyp), eos=-1, norm=False,
                        ins_cost=3.0, del_cost=3.0, sub_cost=4.0, warn=False)  # fmt: skip
    save("sclite", ref=ref, hyp=hyp, per_utt=np.array([per_utt[u] for u in utts]),
         total=np.array(total), ref_errs=errs)  # fmt: skip


class BigramLM(M.MixableSequentialLanguageModel):
    """Stateless bigram table LM used for the search fixtures (table rows: previous token,
    last row = start of sequence)."""

    def __init__(self, table):
        super().__init__(table.shape[1])
        self.register_buffer("table", table)

    def calc_idx_log_probs(self, hist, prev, idx):
        V = self.vocab_size
        N = hist.shape[1]
        if idx.dim() == 0:
            idx = idx.expand(N)
        prev_tok = torch.full((N,), V, dtype=torch.long, device=hist.device)
        if hist.shape[0]:
            last = hist.gather(0, (idx - 1).clamp(min=0).unsqueeze(0)).squeeze(0).clamp(0, V - 1)
            prev_tok = torch.where(idx > 0, last, prev_tok)
        return self.table[prev_tok], prev

    def extract_by_src(self, prev, src):
        return prev

    def mix_by_mask(self, prev_true, prev_false, mask):
        return prev_true


def decoding_goldens():
    rng = np.random.default_rng(0x5EED0003)
    torch.manual_seed(3)
    # G-D2: CTC prefix search, peaky logits, ragged lens
    T, N, V, K = 30, 8, 12, 4
    lg = rng.normal(size=(T, N, V + 1)).astype(np.float32)
    peak = rng.integers(0, V + 1, (T, N))
    np.put_along_axis(lg, peak[..., None], np.take_along_axis(lg, peak[..., None], 2) + 6.0, 2)
    lens = rng.integers(10, T + 1, N)
    y, yl, yp = M.CTCPrefixSearch(K)(torch.from_numpy(lg), torch.from_numpy(lens))
    mask = torch.arange(y.shape[0]).view(-1, 1, 1) < yl.unsqueeze(0)
    d = dict(logits=lg, lens=lens, width=np.array(K), y=torch.where(mask, y, torch.zeros_like(y)),
             y_lens=yl, y_probs=yp)  # fmt: skip
    # with shallow fusion / valid mixture
    table = torch.from_numpy(rng.normal(size=(V + 1, V)).astype(np.float32)).log_softmax(-1)
    lm = BigramLM(table)
    d["lm_table"] = table
    for name, vm in (("fusion", False), ("valid", True)):
        y, yl, yp = M.CTCPrefixSearch(K, 0.3, lm, valid_mixture=vm)(torch.from_numpy(lg), torch.from_numpy(lens))
        mask = torch.arange(y.shape[0]).view(-1, 1, 1) < yl.unsqueeze(0)
        d["y_" + name] = torch.where(mask, y, torch.zeros_like(y))
        d["y_lens_" + name] = yl
        d["y_probs_" + name] = yp
    save("ctc_search", **d)

    # G-D1: one ctc_prefix_search_advance call from a state reached by the reference itself
    N, V, W = 8, 10, 6
    nb, b = torch.zeros(N, 1), torch.ones(N, 1)
    yprev = torch.empty((0, N, 1), dtype=torch.long)
    last = lens_ = torch.zeros((N, 1), dtype=torch.long)
    isp = torch.ones((N, 1, 1), dtype=torch.bool)
    state = None
    for t in range(6):
        p = torch.from_numpy(rng.dirichlet(np.ones(V + 1) * 0.5, N).astype(np.float32))
        Kp = nb.shape[1]
        lmp = torch.from_numpy(rng.dirichlet(np.ones(V), (N, Kp)).astype(np.float32))
        ext = lmp.sqrt() * p[:, None, :V]
        inp = ((ext, p[:, :V].contiguous(), p[:, V].contiguous()), W, (nb, b), yprev, last, lens_, isp)
        out = F.ctc_prefix_search_advance(*inp)
        state = (inp, out)
        yprev, last, lens_, (nb, b), isp = out[0], out[1], out[2], out[3], out[4]
    (probs_t, W_, probs_prev, yprev_, last_, lens__, isp_), out = state
    ynext = out[0]
    mask = torch.arange(ynext.shape[0]).view(-1, 1, 1) < out[2].unsqueeze(0)
    save("ctc_advance", ext=probs_t[0], nonext=probs_t[1], blank=probs_t[2], width=np.array(W_),
         nb_prev=probs_prev[0], b_prev=probs_prev[1], y_prev=yprev_, y_prev_last=last_,
         y_prev_lens=lens__, prev_is_prefix=isp_,
         y_next=torch.where(mask, ynext, torch.zeros_like(ynext)), y_next_last=out[1],
         y_next_lens=out[2], nb_next=out[3][0], b_next=out[3][1], next_is_prefix=out[4],
         next_src=out[5], next_is_nonext=out[6])  # fmt: skip

    # G-D3: beam_search_advance with / without lens, and width > K' * V
    d = {}
    for tag, (N, Kp, V, W, S, with_lens) in {
        "a": (4, 3, 7, 5, 4, False), "b": (4, 3, 7, 5, 4, True), "c": (3, 2, 3, 9, 0, False),
    }.items():  # fmt: skip
        lpt = torch.from_numpy(rng.normal(size=(N, Kp, V)).astype(np.float32)).log_softmax(-1)
        lpp = torch.from_numpy(rng.normal(size=(N, Kp)).astype(np.float32))
        yp_ = torch.from_numpy(rng.integers(0, V, (S, N, Kp)))
        ypl = torch.from_numpy(rng.integers(1, S + 1, (N, Kp))) if with_lens else None
        if ypl is not None:
            ypl[0, 0] = S
        out = F.beam_search_advance(lpt, W, lpp, yp_, ypl)
        K = min(W, Kp * V)
        d.update({tag + "_lpt": lpt, tag + "_lpp": lpp, tag + "_yprev": yp_, tag + "_width": np.array(W),
                  tag + "_ynext": out[0][..., :K], tag + "_lens": out[1], tag + "_lp": out[2], tag + "_src": out[3]})
        if ypl is not None:
            d[tag + "_yprevlens"] = ypl
    save("beam_advance", **d)

    # BeamSearch module with the bigram LM
    V = 9
    table = torch.from_numpy((rng.normal(size=(V + 1, V)) * 2).astype(np.float32)).log_softmax(-1)
    lm = BigramLM(table)
    d = dict(lm_table=table)
    for tag, (kw, call) in {
        "eos": (dict(width=4, eos=0), dict(batch_size=5, max_iters=12)),
        "all": (dict(width=3, eos=2, finish_all_paths=True), dict(batch_size=2, max_iters=10)),
        "noeos": (dict(width=5), dict(batch_size=3, max_iters=6)),
    }.items():
        y, yl, lp = M.BeamSearch(lm, **kw)(dict(), **call)
        d["y_" + tag], d["lens_" + tag], d["lp_" + tag] = y, yl, lp
    save("beam_search", **d)


def image_goldens():
    rng = np.random.default_rng(0x5EED0004)
    torch.manual_seed(4)
    d = {}
    c = torch.from_numpy(rng.uniform(-2, 2, (3, 6, 2)).astype(np.float32))
    f = torch.from_numpy(rng.normal(size=(3, 6, 2)).astype(np.float32))
    q = torch.from_numpy(rng.uniform(-2, 2, (3, 20, 2)).astype(np.float32))
    d.update(sp_c=c, sp_f=f, sp_q=q)
    for o in (1, 2, 3):
        d["sp_o{}".format(o)] = F.polyharmonic_spline(c, f, q, o)
    src = torch.tensor([3.0, 2.5, 1.0, 4.0, 2.0])
    flow = torch.tensor([1.0, -1.0, 2.0, -2.5, 0.5])
    lens = torch.tensor([7.0, 6.0, 5.0, 7.0, 4.0])
    d.update(w1_src=src, w1_flow=flow, w1_lens=lens)
    for o in (1, 2, 3):
        d["w1_o{}".format(o)] = F.warp_1d_grid(src, flow, lens, 7, o)
    # SpecAugment apply with fixed (non-degenerate) parameters
    N, T, Fq = 4, 50, 10
    feats = torch.from_numpy(rng.normal(size=(N, T, Fq)).astype(np.float32))
    lens = torch.tensor([50, 37, 20, 44])
    params = (
        torch.tensor([20.0, 15.0, 9.0, 30.0]), torch.tensor([3.0, -2.0, 1.5, -4.0]),
        torch.tensor([4.0, 5.0, 3.0, 6.0]), torch.tensor([1.0, -1.0, 0.5, -0.5]),
        torch.tensor([[5, 30], [0, 20], [3, 10], [40, 1]]), torch.tensor([[3, 2], [4, 0], [1, 5], [2, 2]]),
        torch.tensor([[1, 7], [0, 4], [2, 2], [8, 5]]), torch.tensor([[2, 1], [1, 0], [3, 1], [1, 2]]),
    )  # fmt: skip
    d.update(sa_feats=feats, sa_lens=lens)
    for i, p in enumerate(params):
        d["sa_p{}".format(i)] = p
    for o in (1, 2, 3):
        d["sa_o{}".format(o)] = F.spec_augment_apply_parameters(feats, params, o, lens)
    # sparse image warp
    N, C, H, W, Mp = 2, 1, 12, 9, 4
    img = torch.from_numpy(rng.uniform(size=(N, C, H, W)).astype(np.float32))
    sp = torch.from_numpy((rng.uniform(size=(N, Mp, 2)) * [H - 1, W - 1]).astype(np.float32))
    dp = sp + torch.from_numpy(rng.normal(size=(N, Mp, 2)).astype(np.float32))
    d.update(siw_img=img, siw_src=sp, siw_dst=dp)
    for pinned in (0, 1, 2):
        w, fl = F.sparse_image_warp(img, sp, dp, pinned_boundary_points=pinned, include_flow=True)
        d["siw_w{}".format(pinned)], d["siw_f{}".format(pinned)] = w, fl
        d["siw_n{}".format(pinned)] = F.sparse_image_warp(img, sp, dp, pinned_boundary_points=pinned, include_flow=False)
    save("image", **d)


if __name__ == "__main__" and os.environ.get("PDT_GOLDEN_ONLY", "") == "":
    string_goldens()
    sclite_golden()
    decoding_goldens()
    image_goldens()


def loss_goldens():
    rng = np.random.default_rng(0x5EED0005)
    N, R, H, V = 6, 9, 8, 7
    ref = rng.integers(0, V, (R, N))
    hyp = rng.integers(0, V, (H, N))
    ref[5, 0] = hyp[4, 1] = 0
    logits = torch.from_numpy(rng.normal(size=(H, N, V)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy(rng.uniform(0.5, 2.0, V).astype(np.float32))
    d = dict(ref=ref, hyp=hyp, logits=logits.detach(), weight=w)
    for tag, kw in {"a": dict(eos=0), "b": dict(eos=None, weight=w), "c": dict(eos=0, include_eos=False, weight=w)}.items():
        for red in ("mean", "sum", "none"):
            loss = F.hard_optimal_completion_distillation_loss(
                logits, torch.from_numpy(ref), torch.from_numpy(hyp), reduction=red, warn=False, **kw)
            (g,) = torch.autograd.grad(loss.sum(), logits)
            d["hocd_{}_{}".format(tag, red)] = loss.detach()
            d["hocd_{}_{}_grad".format(tag, red)] = g
    S = 3
    hyp3 = rng.integers(0, V, (H, N, S))
    lp = torch.from_numpy(rng.normal(size=(N, S)).astype(np.float32))
    d.update(mer_hyp=hyp3, mer_lp=lp)
    for red in ("mean", "none"):
        for sub_avg in (True, False):
            d["mer_{}_{}".format(red, int(sub_avg))] = F.minimum_error_rate_loss(
                lp, torch.from_numpy(ref), torch.from_numpy(hyp3), eos=0, sub_avg=sub_avg,
                reduction=red, warn=False)
    save("losses", **d)


if __name__ == "__main__" and os.environ.get("PDT_GOLDEN_ONLY", "") in ("", "losses"):
    loss_goldens()


def lm_goldens():
    """LookupLanguageModel tries and scores from the live reference.  The tables are large
    enough for int16 offsets: with uint8 offsets the reference's builder fails under NumPy 2
    (defect 1)."""
    rng = np.random.default_rng(0x5EED0006)
    d = {}

    def make(V, sos, N, counts, p_uni=0.9):
        toks = list(range(V)) + ([sos] if not (0 <= sos < V) else [])
        dicts = []
        for n in range(N):
            dd = {}
            if n == 0:
                for v in toks:
                    if rng.random() < p_uni:
                        dd[v] = (float(rng.normal()), float(rng.normal())) if N > 1 else float(rng.normal())
            else:
                keys = set()
                while len(keys) < counts[n]:
                    keys.add(tuple(int(toks[i]) for i in rng.integers(0, len(toks), n + 1)))
                for k in keys:
                    dd[k] = float(rng.normal()) if n == N - 1 else (float(rng.normal()), float(rng.normal()))
            dicts.append(dd)
        return dicts

    cases = {
        "A": (20, -1, 2, [0, 300]),
        "B": (20, 3, 3, [0, 200, 400]),
        "C": (30, 30, 4, [0, 150, 300, 300]),
        "U": (12, -5, 1, [0]),
    }
    lms = {}
    for tag, (V, sos, N, counts) in cases.items():
        dicts = make(V, sos, N, counts)
        for n, dd in enumerate(dicts):
            keys = np.array([[k] if n == 0 else list(k) for k in dd.keys()], dtype=np.int64).reshape(len(dd), n + 1)
            vals = np.array([[v] if n == N - 1 else list(v) for v in dd.values()], dtype=np.float64)
            d["{}_keys{}".format(tag, n)], d["{}_vals{}".format(tag, n)] = keys, vals
        lm = M.LookupLanguageModel(V, sos, [x.copy() for x in dicts])
        lms[tag] = lm
        d[tag + "_cfg"] = np.array([V, sos, N, lm.max_ngram_nodes, lm.max_direct_descendants])
        for name in ("logps", "logbs", "ids", "offsets"):
            d[tag + "_" + name] = getattr(lm, name)
        S, B = 9, 6
        toks = list(range(V)) + ([sos] if not (0 <= sos < V) else [])
        hist = torch.from_numpy(np.asarray(toks)[rng.integers(0, len(toks), (S, B))])
        idx = torch.from_numpy(rng.integers(0, S + 1, B))
        d[tag + "_hist"], d[tag + "_idx"] = hist, idx
        d[tag + "_full"] = lm(hist)
        d[tag + "_at_idx"] = lm(hist, None, idx)[0]
        d[tag + "_at_4"] = lm(hist, None, 4)[0]
        d[tag + "_at_0"] = lm(hist[:0], None, 0)[0]
    # uniform default model
    d["D_full"] = M.LookupLanguageModel(7, 2)(torch.zeros((3, 2), dtype=torch.long))
    # CTC prefix search with the trigram model in shallow fusion
    V, K, T, N = 20, 5, 18, 4
    lg = rng.normal(size=(T, N, V + 1)).astype(np.float32)
    peak = rng.integers(0, V + 1, (T, N))
    np.put_along_axis(lg, peak[..., None], np.take_along_axis(lg, peak[..., None], 2) + 5.0, 2)
    lens = rng.integers(8, T + 1, N)
    y, yl, yp = M.CTCPrefixSearch(K, 0.4, lms["B"])(torch.from_numpy(lg), torch.from_numpy(lens))
    mask = torch.arange(y.shape[0]).view(-1, 1, 1) < yl.unsqueeze(0)
    d.update(ctc_logits=lg, ctc_lens=lens, ctc_width=np.array(K), ctc_beta=np.array(0.4),
             ctc_y=torch.where(mask, y, torch.zeros_like(y)), ctc_y_lens=yl, ctc_y_probs=yp)  # fmt: skip
    # BeamSearch driven by the 4-gram model
    y, yl, lp = M.BeamSearch(lms["C"], 4, eos=0)(dict(), batch_size=3, max_iters=10)
    d.update(beam_y=y, beam_lens=yl, beam_lp=lp)
    save("lm", **d)


if __name__ == "__main__" and os.environ.get("PDT_GOLDEN_ONLY", "") in ("", "lm"):
    lm_goldens()


def pad_goldens():
    rng = np.random.default_rng(0x5EED0007)
    N, T, F = 6, 9, 3
    x = rng.normal(size=(N, T, F)).astype(np.float32)
    lens = np.array([9, 4, 1, 7, 5, 9])
    pad = np.array([[0, 3, 0, 6, 2, 1], [4, 2, 0, 1, 4, 0]])
    d = dict(x=x, lens=lens, pad=pad, xi=rng.integers(-5, 5, (N, T)))
    for mode in ("constant", "reflect", "replicate"):
        d["out_" + mode] = F_.pad_variable(torch.from_numpy(x), torch.from_numpy(lens), torch.from_numpy(pad), mode, -1.5)
        d["outi_" + mode] = F_.pad_variable(torch.from_numpy(d["xi"]), torch.from_numpy(lens), torch.from_numpy(pad), mode, 7)
    save("pad", **d)


if __name__ == "__main__" and os.environ.get("PDT_GOLDEN_ONLY", "") in ("", "pad"):
    F_ = F
    pad_goldens()


def search_sweep_goldens():
    """Many small searches with a language model in the loop (CTCPrefixSearch with shallow
    fusion / valid mixture, BeamSearch with every option) from the live reference: pins the
    host-side frame loops, which the oracle does not restate."""
    rng = np.random.default_rng(0x5EED0008)
    d = {}
    n_ctc = n_beam = 0
    while n_ctc < 24:
        V, K = int(rng.integers(2, 15)), int(rng.integers(1, 9))
        if K > V + 1:
            continue
        T, N = int(rng.integers(1, 25)), int(rng.integers(1, 5))
        lg = rng.normal(size=(T, N, V + 1)).astype(np.float32)
        peak = rng.integers(0, V + 1, (T, N))
        np.put_along_axis(lg, peak[..., None], np.take_along_axis(lg, peak[..., None], 2) + 4.0, 2)
        lens = None if rng.random() < 0.3 else rng.integers(0, T + 1, N)
        table = torch.from_numpy((rng.normal(size=(V + 1, V)) * 1.5).astype(np.float32)).log_softmax(-1)
        beta = float(rng.uniform(0.05, 0.9))
        vm = bool(rng.integers(0, 2))
        y, yl, yp = M.CTCPrefixSearch(K, beta, BigramLM(table), valid_mixture=vm)(
            torch.from_numpy(lg), None if lens is None else torch.from_numpy(lens)
        )
        srt = yp.sort(1, descending=True).values
        if not torch.isfinite(yp).all() or (K > 1 and ((srt[:, :-1] - srt[:, 1:]) < 1e-6 * srt[:, :-1]).any()):
            continue  # padded beams / near ties: outcome unspecified
        mask = torch.arange(y.shape[0]).view(-1, 1, 1) < yl.unsqueeze(0)
        tag = "ctc%d_" % n_ctc
        d[tag + "logits"], d[tag + "table"] = lg, table
        d[tag + "lens"] = np.array([-1]) if lens is None else lens
        d[tag + "cfg"] = np.array([K, beta, float(vm)])
        d[tag + "y"], d[tag + "y_lens"], d[tag + "y_probs"] = torch.where(mask, y, torch.zeros_like(y)), yl, yp
        n_ctc += 1
    while n_beam < 24:
        V, K = int(rng.integers(2, 12)), int(rng.integers(1, 8))
        table = torch.from_numpy((rng.normal(size=(V + 1, V)) * 2).astype(np.float32)).log_softmax(-1)
        eos = None if rng.random() < 0.3 else int(rng.integers(-V, V))
        fin = bool(rng.integers(0, 2))
        N = None if rng.random() < 0.2 else int(rng.integers(1, 5))
        iters = int(rng.integers(0, 12))
        y, yl, lp = M.BeamSearch(BigramLM(table), K, eos, fin, -7)(dict(), N, iters)
        fl = lp.reshape(-1, K)
        srt = fl.sort(1, descending=True).values
        if K > 1 and torch.isfinite(srt).all() and (srt[:, :-1] - srt[:, 1:]).min() < 1e-6:
            continue
        tag = "beam%d_" % n_beam
        d[tag + "table"] = table
        d[tag + "cfg"] = np.array([K, -1000 if eos is None else eos, int(fin), -1 if N is None else N, iters])
        d[tag + "y"], d[tag + "y_lens"], d[tag + "lp"] = y, yl, lp
        n_beam += 1
    save("search_sweep", **d)


if __name__ == "__main__" and os.environ.get("PDT_GOLDEN_ONLY", "") in ("", "sweep"):
    search_sweep_goldens()


class CounterLM(M.MixableSequentialLanguageModel):
    """A model WITH state under the reference's interface (twins: oracle.CounterLM, tests/_toy_lm.py)."""

    def __init__(self, table):
        super().__init__(table.shape[1])
        self.register_buffer("table", table)

    def update_input(self, prev, hist):
        if "count" not in prev:
            prev = {"count": torch.zeros((hist.size(1),))}
        return prev

    def calc_idx_log_probs(self, hist, prev, idx):
        V = self.vocab_size
        N = hist.shape[1]
        if idx.dim() == 0:
            idx = idx.expand(N)
        prev_tok = torch.full((N,), V, dtype=torch.long)
        if hist.shape[0]:
            last = hist.gather(0, (idx - 1).clamp(min=0).unsqueeze(0)).squeeze(0).clamp(0, V - 1)
            prev_tok = torch.where(idx > 0, last, prev_tok)
        x = self.table[prev_tok] * (1.0 + 0.1 * prev["count"]).unsqueeze(1)
        return x.log_softmax(-1), {"count": prev["count"] + 1.0}

    def extract_by_src(self, prev, src):
        return {"count": prev["count"].index_select(0, src)}

    def mix_by_mask(self, prev_true, prev_false, mask):
        return {"count": torch.where(mask, prev_true["count"], prev_false["count"])}


def _tie_free(p, K, rel):
    """Beam entries of one batch element far enough apart that their order is not a rounding matter."""
    fl = p.reshape(-1, K)
    if not torch.isfinite(fl).all():
        return False
    srt = fl.sort(1, descending=True).values
    gap = srt[:, :-1] - srt[:, 1:]
    return K == 1 or bool((gap >= rel * srt[:, :-1].abs().clamp(min=1e-30)).all())


def lm_search_goldens():
    """Searches with the SHIPPED n-gram model (LookupLanguageModel) in the loop, from the live reference:
    CTCPrefixSearch with orders 2-4, shallow fusion and valid mixture, ragged lens, sos inside / outside
    the vocabulary; BeamSearch with eos / finish_all_paths / neither; and both searches around a model
    WITH state (CounterLM), which pins extract_by_src / mix_by_mask.  Tables are large enough for int16
    trie offsets (the reference's builder fails on uint8 ones under NumPy 2)."""
    rng = np.random.default_rng(0x5EED0009)
    d = {}

    def make(V, sos, N, counts):
        toks = list(range(V)) + ([sos] if not (0 <= sos < V) else [])
        dicts = []
        for n in range(N):
            dd = {}
            if n == 0:
                for v in toks:
                    dd[v] = (float(rng.normal() - 2.0), float(rng.normal() * 0.3)) if N > 1 else float(rng.normal())
            else:
                keys = set()
                while len(keys) < counts[n]:
                    keys.add(tuple(int(toks[i]) for i in rng.integers(0, len(toks), n + 1)))
                for k in keys:
                    dd[k] = float(rng.normal() - 1.0) if n == N - 1 else (float(rng.normal() - 1.0), float(rng.normal() * 0.3))
            dicts.append(dd)
        return dicts

    def store(tag, V, sos, dicts):
        N = len(dicts)
        d[tag + "_cfg"] = np.array([V, sos, N])
        for n, dd in enumerate(dicts):
            d["{}_keys{}".format(tag, n)] = np.array(
                [[k] if n == 0 else list(k) for k in dd.keys()], dtype=np.int64).reshape(len(dd), n + 1)
            d["{}_vals{}".format(tag, n)] = np.array([[v] if n == N - 1 else list(v) for v in dd.values()], dtype=np.float64)

    models = {"m2in": (24, 5, 2, [0, 400]), "m2out": (24, -1, 2, [0, 400]), "m3in": (20, 0, 3, [0, 250, 500]),
              "m3out": (20, 20, 3, [0, 250, 500]), "m4out": (16, -3, 4, [0, 200, 400, 400])}
    lms = {}
    for tag, (V, sos, N, counts) in models.items():
        dicts = make(V, sos, N, counts)
        store(tag, V, sos, dicts)
        lms[tag] = (V, M.LookupLanguageModel(V, sos, [x.copy() for x in dicts]))
    n_ctc = n_beam = 0
    tags = list(models)
    while n_ctc < 30:
        tag = tags[n_ctc % len(tags)]
        V, lm = lms[tag]
        K = int(rng.integers(1, 9))
        T, N = int(rng.integers(1, 26)), int(rng.integers(1, 5))
        lg = rng.normal(size=(T, N, V + 1)).astype(np.float32)
        peak = rng.integers(0, V + 1, (T, N))
        np.put_along_axis(lg, peak[..., None], np.take_along_axis(lg, peak[..., None], 2) + 4.0, 2)
        lens = None if rng.random() < 0.3 else rng.integers(0, T + 1, N)
        beta = float(rng.uniform(0.05, 0.9))
        vm = bool((n_ctc // len(tags)) % 2)
        y, yl, yp = M.CTCPrefixSearch(K, beta, lm, valid_mixture=vm)(
            torch.from_numpy(lg), None if lens is None else torch.from_numpy(lens))
        if not _tie_free(yp, K, 1e-4):
            continue
        mask = torch.arange(y.shape[0]).view(-1, 1, 1) < yl.unsqueeze(0)
        t_ = "ctc%d_" % n_ctc
        d[t_ + "logits"] = lg
        d[t_ + "lens"] = np.array([-1]) if lens is None else lens
        d[t_ + "cfg"] = np.array([K, beta, float(vm), tags.index(tag)])
        d[t_ + "y"], d[t_ + "y_lens"], d[t_ + "y_probs"] = torch.where(mask, y, torch.zeros_like(y)), yl, yp
        n_ctc += 1
    while n_beam < 20:
        tag = tags[n_beam % len(tags)]
        V, lm = lms[tag]
        K = int(rng.integers(1, 8))
        eos = None if n_beam % 4 == 3 else int(rng.integers(-V, V))
        fin = bool(n_beam % 2)
        N = None if rng.random() < 0.2 else int(rng.integers(1, 5))
        iters = int(rng.integers(0, 14))
        y, yl, lp = M.BeamSearch(lm, K, eos, fin, -7)(dict(), N, iters)
        fl = lp.reshape(-1, K)
        fin_ = torch.isfinite(fl)
        srt = fl.sort(1, descending=True).values
        if K > 1 and fin_.all() and (srt[:, :-1] - srt[:, 1:]).min() < 1e-4:
            continue
        t_ = "beam%d_" % n_beam
        d[t_ + "cfg"] = np.array([K, -1000 if eos is None else eos, int(fin), -1 if N is None else N, iters, tags.index(tag)])
        d[t_ + "y"], d[t_ + "y_lens"], d[t_ + "lp"] = y, yl, lp
        n_beam += 1
    d["model_tags"] = np.array(tags)
    # a model with state in both searches
    n_c = 0
    while n_c < 12:
        V, K = int(rng.integers(3, 12)), int(rng.integers(1, 7))
        if K > V + 1:
            continue
        T, N = int(rng.integers(2, 20)), int(rng.integers(1, 4))
        table = torch.from_numpy((rng.normal(size=(V + 1, V)) * 1.5).astype(np.float32))
        lm = CounterLM(table)
        lg = rng.normal(size=(T, N, V + 1)).astype(np.float32)
        peak = rng.integers(0, V + 1, (T, N))
        np.put_along_axis(lg, peak[..., None], np.take_along_axis(lg, peak[..., None], 2) + 3.0, 2)
        lens = None if rng.random() < 0.3 else rng.integers(0, T + 1, N)
        beta, vm = float(rng.uniform(0.2, 0.9)), bool(n_c % 2)
        y, yl, yp = M.CTCPrefixSearch(K, beta, lm, valid_mixture=vm)(
            torch.from_numpy(lg), None if lens is None else torch.from_numpy(lens))
        if not _tie_free(yp, K, 1e-4):
            continue
        eos = None if n_c % 3 == 0 else int(rng.integers(0, V))
        iters = int(rng.integers(1, 10))
        by, byl, blp = M.BeamSearch(lm, K, eos, bool(n_c % 2), -9)(dict(), N, iters)
        if not _tie_free(blp, K, 0.0) and torch.isfinite(blp).all():
            continue
        mask = torch.arange(y.shape[0]).view(-1, 1, 1) < yl.unsqueeze(0)
        t_ = "cnt%d_" % n_c
        d[t_ + "table"], d[t_ + "logits"] = table, lg
        d[t_ + "lens"] = np.array([-1]) if lens is None else lens
        d[t_ + "cfg"] = np.array([K, beta, float(vm), -1000 if eos is None else eos, iters])
        d[t_ + "y"], d[t_ + "y_lens"], d[t_ + "y_probs"] = torch.where(mask, y, torch.zeros_like(y)), yl, yp
        d[t_ + "by"], d[t_ + "by_lens"], d[t_ + "blp"] = by, byl, blp
        n_c += 1
    save("lm_search", **d)


if __name__ == "__main__" and os.environ.get("PDT_GOLDEN_ONLY", "") in ("", "lm_search"):
    lm_search_goldens()


def fill_goldens():
    """fill_after_eos (_string.py:30-42): every dim of a 3-D tensor, default and explicit fill,
    a separate value tensor (float and bool), rows with no / several / leading eos."""
    rng = np.random.default_rng(0x5EED0009)
    tok = rng.integers(0, 4, (7, 5, 6))
    tok[:, 0, 0] = 1  # no eos in this column
    tok[0, 1, :] = 3  # eos first along dim 0
    val = rng.normal(size=tok.shape).astype(np.float32)
    valb = rng.integers(0, 2, tok.shape).astype(bool)
    d = dict(tokens=tok, value=val, value_bool=valb, eos=np.array(3))
    t = torch.from_numpy(tok)
    for dim in (0, 1, 2, -1):
        d["default_d%d" % dim] = F.fill_after_eos(t, 3, dim)
        d["fill_d%d" % dim] = F.fill_after_eos(t, 3, dim, -7.0)
        d["value_d%d" % dim] = F.fill_after_eos(t, 3, dim, 0.5, torch.from_numpy(val))
        d["bool_d%d" % dim] = F.fill_after_eos(t, 3, dim, 1.0, torch.from_numpy(valb))
    d["module"] = M.FillAfterEndOfSequence(3, 1, 9.0)(t)
    save("fill", **d)


if __name__ == "__main__" and os.environ.get("PDT_GOLDEN_ONLY", "") in ("", "fill"):
    fill_goldens()


def c4_goldens():
    """BASELINE config 4 at its real frame count (T=1000, F=80) for two utterances: the live
    reference's SpecAugment application (time warp up to 80 frames, 2 + 2 masks) and its
    sparse_image_warp of the same tensor; float16 inputs keep the fixture small and are exact
    in float32."""
    rng = np.random.default_rng(0x5EED000A)
    N, T, Fq = 2, 1000, 80
    feats = torch.from_numpy(rng.normal(size=(N, T, Fq)).astype(np.float16).astype(np.float32))
    lens = torch.tensor([1000, 731])
    params = (
        torch.tensor([412.0, 250.0]), torch.tensor([63.0, -71.0]), torch.zeros(0), torch.zeros(0),
        torch.tensor([[100, 640], [20, 500]]), torch.tensor([[37, 22], [29, 11]]),
        torch.tensor([[3, 50], [10, 61]]), torch.tensor([[20, 9], [27, 5]]),
    )  # fmt: skip
    d = dict(feats=feats.numpy().astype(np.float16), lens=lens)
    for i, p in enumerate(params):
        d["p{}".format(i)] = p
    out = F.spec_augment_apply_parameters(feats, params, 1, lens)
    d["sa_o1"] = out
    src = torch.tensor([[[300.0, 20.0], [700.0, 60.0], [500.0, 40.0]], [[100.0, 10.0], [650.0, 70.0], [400.0, 33.0]]])
    dst = src + torch.tensor([[[2.0, -1.0], [-3.0, 0.5], [1.0, 1.0]], [[-2.0, 0.5], [1.5, -1.0], [0.0, 2.0]]])
    d.update(siw_src=src, siw_dst=dst)
    d["siw"] = F.sparse_image_warp(feats.unsqueeze(1), src, dst, pinned_boundary_points=1, include_flow=False)
    save("c4_sample", **d)


if __name__ == "__main__" and os.environ.get("PDT_GOLDEN_ONLY", "") in ("", "c4"):
    c4_goldens()


def spec_draw_goldens():
    """spec_augment_draw_parameters of the live reference with its uniform draws PINNED: torch.rand is
    replaced, for the duration of the call, by a function that hands out the columns of a known (N, R)
    tensor in the reference's call order -- so the mapping from draws to parameters (not the generator)
    is what the fixture holds.  Several configurations, groups enabled and disabled, the extreme draws 0
    and 1 - 2^-24, ragged lengths and none."""
    rng = np.random.default_rng(0x5EED000B)
    d = {}
    cfgs = [
        dict(max_time_warp=80.0, max_freq_warp=0.0, max_time_mask=100, max_freq_mask=27, max_time_mask_proportion=0.04,
             num_time_mask=2, num_time_mask_proportion=1.0, num_freq_mask=2),  # C4
        dict(max_time_warp=10.0, max_freq_warp=2.0, max_time_mask=20, max_freq_mask=5, max_time_mask_proportion=0.1,
             num_time_mask=3, num_time_mask_proportion=0.02, num_freq_mask=2),
        dict(max_time_warp=0.0, max_freq_warp=3.5, max_time_mask=0, max_freq_mask=8, max_time_mask_proportion=0.2,
             num_time_mask=4, num_time_mask_proportion=0.5, num_freq_mask=1),
        dict(max_time_warp=5.0, max_freq_warp=0.0, max_time_mask=7, max_freq_mask=0, max_time_mask_proportion=1.0,
             num_time_mask=5, num_time_mask_proportion=0.01, num_freq_mask=3),
    ]
    real_rand = torch.rand
    for i, cfg in enumerate(cfgs):
        N, T, Fq = 9, 60 + 17 * i, 12 + 5 * i
        tw, fw = cfg["max_time_warp"] != 0.0, cfg["max_freq_warp"] != 0.0
        tm = all(cfg[k] != 0 for k in ("max_time_mask", "max_time_mask_proportion", "num_time_mask", "num_time_mask_proportion"))
        fm = cfg["max_freq_mask"] != 0 and cfg["num_freq_mask"] != 0
        MT, MF = (cfg["num_time_mask"] if tm else 0), (cfg["num_freq_mask"] if fm else 0)
        R = 2 * tw + 2 * fw + 2 * MT + 2 * MF
        u = rng.random((N, R)).astype(np.float32)
        u[0] = 0.0
        u[1] = np.float32(1.0 - 2.0 ** -24)
        lens = None if i == 2 else torch.from_numpy(rng.integers(T // 3, T + 1, N))
        cols = [0]
        tu = torch.from_numpy(u)

        def fake_rand(shape, *a, **k):
            n = 1 if len(shape) == 1 else shape[1]
            out = tu[:, cols[0] : cols[0] + n]
            cols[0] += n
            return out.reshape(shape).clone()

        torch.rand = fake_rand
        try:
            out = F.spec_augment_draw_parameters(torch.zeros((N, T, Fq)), lengths=lens, **cfg)
        finally:
            torch.rand = real_rand
        assert cols[0] == R, (cols[0], R)
        t_ = "d%d_" % i
        d[t_ + "u"], d[t_ + "shape"] = u, np.array([N, T, Fq])
        d[t_ + "lens"] = np.array([-1]) if lens is None else lens
        d[t_ + "cfg"] = np.array([cfg[k] for k in ("max_time_warp", "max_freq_warp", "max_time_mask", "max_freq_mask",
                                                    "max_time_mask_proportion", "num_time_mask",
                                                    "num_time_mask_proportion", "num_freq_mask")], dtype=np.float64)
        for name, o in zip(("w_0", "w", "v_0", "v", "t_0", "t", "f_0", "f"), out):
            d[t_ + name] = o
    save("spec_draw", **d)


if __name__ == "__main__" and os.environ.get("PDT_GOLDEN_ONLY", "") in ("", "spec_draw"):
    spec_draw_goldens()


def image_f64_goldens():
    """The reference's float64 image path: dense_image_warp / sparse_image_warp(include_flow=True) on a
    float64 image (the grid is formed and sampled in float64, _img.py:423-436; the flow and the
    spline stay float32, :420, :537-538).  Its other image operators RAISE on float64 values (a dtype
    mismatch in linalg.solve / grid_sample), which the last entries record."""
    rng = np.random.default_rng(0x5EED000C)
    d = {}
    N, C, H, W = 2, 2, 11, 7
    img = torch.from_numpy(rng.normal(size=(N, C, H, W)))
    flow = torch.from_numpy((rng.normal(size=(N, H, W, 2)) * 2.5).astype(np.float32))
    d.update(img=img, flow=flow)
    for mode in ("bilinear", "nearest"):
        for pad in ("border", "zeros", "reflection"):
            for ind in ("hw", "wh"):
                d["dense_{}_{}_{}".format(mode, pad, ind)] = F.dense_image_warp(img, flow, ind, mode, pad)
    Mp = 4
    sp = torch.from_numpy((rng.uniform(size=(N, Mp, 2)) * [H - 1, W - 1]).astype(np.float32))
    dp = sp + torch.from_numpy(rng.normal(size=(N, Mp, 2)).astype(np.float32))
    d.update(src=sp, dst=dp)
    for order in (1, 2, 3):
        w, fl = F.sparse_image_warp(img, sp, dp, field_interpolation_order=order, include_flow=True)
        assert w.dtype == torch.double
        d["sparse_w{}".format(order)], d["sparse_f{}".format(order)] = w, fl
    raises = []
    for name, fn in (
        ("sparse_image_warp(include_flow=True, pinned_boundary_points=1)",
         lambda: F.sparse_image_warp(img, sp, dp, pinned_boundary_points=1, include_flow=True)),
        ("sparse_image_warp(include_flow=False)", lambda: F.sparse_image_warp(img, sp, dp, include_flow=False)),
        ("polyharmonic_spline(float64 values)", lambda: F.polyharmonic_spline(sp, dp.double(), sp, 2)),
        ("spec_augment_apply_parameters(float64 feats, warp)", lambda: F.spec_augment_apply_parameters(
            img[:, 0], F.spec_augment_draw_parameters(img[:, 0], 2.0, 1.0, 2, 2, 0.5, 1, 0.5, 1), 1)),
    ):
        try:
            fn()
            raises.append(name + ": ok")
        except RuntimeError:
            raises.append(name + ": RuntimeError")
    d["reference_behaviour"] = np.array(raises)
    save("image_f64", **d)


if __name__ == "__main__" and os.environ.get("PDT_GOLDEN_ONLY", "") in ("", "image_f64"):
    image_f64_goldens()


if __name__ == "__main__" and os.environ.get("PDT_GOLDEN_ONLY", "") in ("", "string_edge"):
    string_edge_goldens()
