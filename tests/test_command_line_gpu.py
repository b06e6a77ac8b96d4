"""The directory scorer end to end on the GPU.

Acceptance test = the reference's own (tests/test_command_line.py:495-531): the NIST sclite
known answer for tests/golden/sclite, through trn -> token data dirs -> error rates.  Plus the
randomized check of :396-494 (ignore / replace / id2token / missing utterances) against
the oracle's Levenshtein."""
import os

import numpy as np
import pytest
import torch

import oracle
from pydrobert_amd import command_line

pytestmark = pytest.mark.gpu
SCLITE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sclite")


def test_error_rates_match_sclite(tmp_path):
    token2id = os.path.join(SCLITE, "token2id.txt")
    ref_dir, hyp_dir = str(tmp_path / "ref"), str(tmp_path / "hyp")
    assert not command_line.trn_to_torch_token_data_dir([os.path.join(SCLITE, "ref.trn"), token2id, ref_dir])
    assert not command_line.trn_to_torch_token_data_dir([os.path.join(SCLITE, "hyp.trn"), token2id, hyp_dir])
    total, per_utt = str(tmp_path / "total.txt"), str(tmp_path / "per_utt.txt")
    assert not command_line.compute_torch_token_data_dir_error_rates(
        [ref_dir, hyp_dir, total, "--nist-costs", "--quiet"]
    )
    assert not command_line.main(
        ["compute-torch-token-data-dir-error-rates", ref_dir, hyp_dir, per_utt, "--nist-costs", "--per-utt",
         "--quiet", "--batch-size", "7"]
    )  # fmt: skip

    def table(path):
        with open(path) as f:
            return {u: "{:.03f}".format(float(v)) for u, v in (line.split() for line in f)}

    exp, act = table(os.path.join(SCLITE, "per_utt.txt")), table(per_utt)
    assert len(exp) == 50 and exp == act
    with open(os.path.join(SCLITE, "total.txt")) as f:
        exp_total = "{:.03f}".format(float(f.read()))
    with open(total) as f:
        assert exp_total == "{:.03f}".format(float(f.read()))


@pytest.mark.parametrize("per_utt", [True, False])
@pytest.mark.parametrize("norm", [True, False])
@pytest.mark.parametrize("tokens", [None, "id2token", "token2id"])
@pytest.mark.parametrize("layout", ["timed", "flat"])
def test_error_rates_random_dirs(tmp_path, per_utt, norm, tokens, layout):
    rng = np.random.default_rng(17 + 2 * per_utt + 4 * norm + 8 * (layout == "flat"))
    V, n_utts = 12, 40
    names = [chr(ord("a") + i) for i in range(V)]
    ignore_ids, replace_ids = {3, 7}, {5: 1, 9: 2}
    ref_dir, hyp_dir = tmp_path / "ref", tmp_path / "hyp"
    ref_dir.mkdir()
    hyp_dir.mkdir()
    exp, tot_err, tot_len, kept = {}, 0.0, 0, 0
    for u in range(n_utts):
        ref = rng.integers(0, V, rng.integers(1, 20))
        hyp = rng.integers(0, V, rng.integers(0, 20))
        for seq, d in ((ref, ref_dir), (hyp, hyp_dir)):
            t = torch.from_numpy(seq)
            if layout == "timed":
                t = torch.stack([t, torch.full_like(t, -1), torch.full_like(t, -1)], 1)
            torch.save(t, str(d / "{}.pt".format(u)))
        missing = rng.random() < 0.1
        if missing:
            os.remove(str((ref_dir if rng.random() < 0.5 else hyp_dir) / "{}.pt".format(u)))
            continue

        def clean(seq):
            out = [replace_ids.get(int(t), int(t)) for t in seq]
            return np.array([t for t in out if t not in ignore_ids], dtype=np.int64)

        r, h = clean(ref), clean(hyp)
        if len(r) == 0:  # a rate over an empty reference is undefined; keep the case out
            os.remove(str(ref_dir / "{}.pt".format(u)))
            os.remove(str(hyp_dir / "{}.pt".format(u)))
            continue
        T = max(len(r), len(h)) + 1
        rp, hp = np.full((T, 1), -2), np.full((T, 1), -2)
        rp[: len(r), 0], rp[len(r), 0] = r, -1
        hp[: len(h), 0], hp[len(h), 0] = h, -1
        er = float(oracle.error_rate(rp, hp, eos=-1, norm=False)[0])
        exp[str(u)] = er / (len(r) if norm else 1)
        tot_err += er
        tot_len += len(r)
        kept += 1
    fmt = (lambda i: names[i]) if tokens else str
    (tmp_path / "ignore").write_text(" ".join(fmt(i) for i in sorted(ignore_ids)) + "\n")
    (tmp_path / "replace").write_text("".join("{} {}\n".format(fmt(a), fmt(b)) for a, b in replace_ids.items()))
    out = str(tmp_path / "out.txt")
    args = [str(ref_dir), str(hyp_dir), out, "--ignore", str(tmp_path / "ignore"), "--replace",
            str(tmp_path / "replace"), "--warn-missing", "--batch-size", "9"]  # fmt: skip
    if not norm:
        args.append("--distances")
    if per_utt:
        args.append("--per-utt")
    if tokens:
        rows = ["{} {}".format(i, n) if tokens == "id2token" else "{} {}".format(n, i) for i, n in enumerate(names)]
        (tmp_path / "map").write_text("\n".join(rows) + "\n")
        args += ["--id2token", str(tmp_path / "map")]
        if tokens == "token2id":
            args.append("--swap")
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert not command_line.compute_torch_token_data_dir_error_rates(args)
    with open(out) as f:
        if per_utt:
            act = {u: float(v) for u, v in (line.split() for line in f)}
            assert set(act) == set(exp)
            for u in exp:
                assert abs(exp[u] - act[u]) < 1e-5, u
        else:
            assert abs(float(f.read()) - tot_err / (tot_len if norm else kept)) < 1e-4


def test_missing_utterance_is_an_error_by_default(tmp_path):
    for d in ("ref", "hyp"):
        (tmp_path / d).mkdir()
        torch.save(torch.tensor([1, 2, 3]), str(tmp_path / d / "a.pt"))
    torch.save(torch.tensor([1]), str(tmp_path / "ref" / "b.pt"))
    with pytest.raises(ValueError, match='contains utterance "b"'):
        command_line.compute_torch_token_data_dir_error_rates([str(tmp_path)])
    out = str(tmp_path / "o")
    with pytest.warns(UserWarning, match="Skipping"):
        assert not command_line.compute_torch_token_data_dir_error_rates([str(tmp_path), "--warn-missing"] )
