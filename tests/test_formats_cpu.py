"""Transcript formats around the error-rate path: "trn" files and
token data dirs.  Known answers are the facts of the reference's tests/test_parsing.py:141-191,
:229-267, :300-430."""
import os
from io import StringIO

import pytest
import torch

from pydrobert_amd import _parsing as P
from pydrobert_amd import command_line

SCLITE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sclite")


def test_read_trn_known_answers():
    trn = StringIO("here is a simple example (a)\nnothing should go wrong (b)\n")
    assert P.read_trn(trn) == [
        ("a", ["here", "is", "a", "simple", "example"]),
        ("b", ["nothing", "should", "go", "wrong"]),
    ]
    trn = StringIO(
        "here is an { example /with} some alternates (a)\n"
        "} and /here/ is {something really / {really}} (stupid) { ignore this (b)\n"
        "(c)\n"
        "a11 (d)\n"
    )
    assert P.read_trn(trn, warn=False) == [
        ("a", ["here", "is", "an", ([["example"], ["with"]], -1, -1), "some", "alternates"]),
        ("b", ["}", "and", "/here/", "is", ([["something", "really"], [[["really"]]]], -1, -1), "(stupid)"]),
        ("c", []),
        ("d", ["a11"]),
    ]
    trn.seek(0)
    with pytest.warns(UserWarning, match="alternate"):
        P.read_trn(trn)
    with pytest.raises(IOError):
        P.read_trn(StringIO("no utterance id here\n"))
    with pytest.raises(IOError):
        P.read_trn(StringIO("an { } empty alternate (x)\n"))


def test_write_trn_known_answers():
    out = StringIO()
    P.write_trn([("a", ["again", "a", "simple", "example"]), ("b", ["should", "get", "right", "no", "prob"])], out)
    assert out.getvalue() == "again a simple example (a)\nshould get right no prob (b)\n"
    out = StringIO()
    P.write_trn(
        [
            (" c ", [("unnecessary", -1, -1), ([["complexity", [["can"]]], ["also", "be"]], 10, 4), "handled"]),
            ("d", []),
            ("e", ["a11"]),
        ],
        out,
    )
    assert out.getvalue() == "unnecessary { complexity { can } / also be } handled ( c )\n(d)\na11 (e)\n"
    # round trip through the reader
    out.seek(0)
    back = P.read_trn(out, warn=False)
    assert back[0] == (" c ", ["unnecessary", ([["complexity", [["can"]]], ["also", "be"]], -1, -1), "handled"])


@pytest.mark.parametrize(
    "transcript,token2id,unk,skip,exp",
    [
        ([], None, None, False, torch.empty((0, 3), dtype=torch.long)),
        ([1, 2, 3, 4], None, None, True, torch.tensor([1, 2, 3, 4])),
        (
            [1, ("a", 4, 10), "a", 3], {"a": 2}, None, False,
            torch.tensor([[1, -1, -1], [2, 4, 10], [2, -1, -1], [3, -1, -1]]),
        ),
        (
            ["foo", 1, "bar"], {"foo": 0, "baz": 3}, "baz", False,
            torch.tensor([[0, -1, -1], [3, -1, -1], [3, -1, -1]]),
        ),
    ],
)  # fmt: skip
def test_transcript_to_token(transcript, token2id, unk, skip, exp):
    act = P.transcript_to_token(transcript, token2id, unk=unk, skip_frame_times=skip)
    assert act.dtype == torch.long and torch.equal(exp, act)
    with pytest.raises(Exception):
        P.transcript_to_token(["foo"] + transcript, token2id)


def test_frame_shift_conversions():
    trans = [(12, 0.5, 0.81), 420, (1, 2.1, 2.2), (3, 2.8, 2.815), (12, 2.9, 3.0025)]
    tok = P.transcript_to_token(trans, frame_shift_ms=10)
    assert tok.tolist() == [[12, 50, 81], [420, -1, -1], [1, 210, 220], [3, 280, 282], [12, 290, 300]]
    tok = P.transcript_to_token(trans, frame_shift_ms=1 / 8)
    assert tok.tolist() == [
        [12, 4000, 6480], [420, -1, -1], [1, 16800, 17600], [3, 22400, 22520], [12, 23200, 24020],
    ]  # fmt: skip
    tok = torch.tensor([[1, -1, 10], [2, 1000, 2000], [3, 12345, 678910]])
    assert P.token_to_transcript(tok, frame_shift_ms=10) == [1, (2, 10.0, 20.0), (3, 123.45, 6789.10)]
    assert P.token_to_transcript(tok, frame_shift_ms=1 / 8) == [
        1, (2, 1000 / 8000, 2000 / 8000), (3, 12345 / 8000, 678910 / 8000),
    ]  # fmt: skip


@pytest.mark.parametrize(
    "tok,id2token,exp",
    [
        (torch.empty((0, 3), dtype=torch.long), None, []),
        (torch.tensor([[1, -1, -1], [2, -1, -1], [3, -1, -1], [4, -1, -1]]), None, [1, 2, 3, 4]),
        (torch.tensor([[1, 3, 4], [3, 4, 5], [2, -1, -1]]), {1: "a", 2: "b"}, [("a", 3, 4), (3, 4, 5), "b"]),
        (torch.arange(10), None, list(range(10))),
        (torch.arange(5).unsqueeze(-1), None, list(range(5))),
    ],
)
def test_token_to_transcript(tok, id2token, exp):
    assert P.token_to_transcript(tok, id2token) == exp


def test_parse_token2id():
    f = StringIO("a 0\nb 1\n\nc -3\n")
    f.name = "map"
    assert P.parse_token2id(f, False, False) == {"a": 0, "b": 1, "c": -3}
    f.seek(0)
    assert P.parse_token2id(f, False, True) == {0: "a", 1: "b", -3: "c"}
    g = StringIO("0 a\n1 b\n")
    g.name = "map"
    assert P.parse_token2id(g, True, False) == {0: "a", 1: "b"}
    bad = StringIO("a b c\n")
    bad.name = "bad"
    with pytest.raises(ValueError, match="Cannot parse line 1"):
        P.parse_token2id(bad, False, False)
    dup = StringIO("a 0\na 1\n")
    dup.name = "dup"
    with pytest.warns(UserWarning, match="already exists"):
        P.parse_token2id(dup, False, False)


def test_trn_to_token_dir_and_back(tmp_path):
    """The sclite fixture through both converters: trn -> one .pt per utterance -> trn."""
    token2id = os.path.join(SCLITE, "token2id.txt")
    ref_dir = str(tmp_path / "ref")
    assert command_line.trn_to_torch_token_data_dir([os.path.join(SCLITE, "ref.trn"), token2id, ref_dir]) == 0
    exp = P.read_trn(os.path.join(SCLITE, "ref.trn"))
    assert sorted(os.listdir(ref_dir)) == sorted(u + ".pt" for u, _ in exp)
    tok = torch.load(os.path.join(ref_dir, exp[0][0] + ".pt"))
    assert tok.dtype == torch.long and tok.shape == (len(exp[0][1]), 3) and (tok[:, 1:] == -1).all()
    out = str(tmp_path / "back.trn")
    assert command_line.torch_token_data_dir_to_trn([ref_dir, token2id, out, "--swap"]) == 0
    assert sorted(P.read_trn(out)) == sorted(exp)
    # the (R,) and (R, 1) layouts
    for flag, shape in (("--skip-frame-times", (len(exp[0][1]),)), ("--feat-sizing", (len(exp[0][1]), 1))):
        d = str(tmp_path / flag.strip("-"))
        assert command_line.trn_to_torch_token_data_dir([os.path.join(SCLITE, "ref.trn"), token2id, d, flag]) == 0
        assert torch.load(os.path.join(d, exp[0][0] + ".pt")).shape == shape


def test_trn_alternates_need_a_handler(tmp_path):
    trn = tmp_path / "alt.trn"
    trn.write_text("a { b / c } d (u)\n")
    t2i = tmp_path / "t2i"
    t2i.write_text("a 0\nb 1\nc 2\nd 3\n")
    with pytest.warns(UserWarning), pytest.raises(ValueError, match="alternate"):
        command_line.trn_to_torch_token_data_dir([str(trn), str(t2i), str(tmp_path / "x")])
    with pytest.warns(UserWarning):
        assert command_line.trn_to_torch_token_data_dir(
            [str(trn), str(t2i), str(tmp_path / "y"), "--alt-handler", "first", "--skip-frame-times"]
        ) == 0
    assert torch.load(str(tmp_path / "y" / "u.pt")).tolist() == [0, 1, 3]


def test_error_rate_command_argument_errors(tmp_path, capsys):
    (tmp_path / "ref").mkdir()
    assert command_line.compute_torch_token_data_dir_error_rates([str(tmp_path)]) == 1
    assert "is not a directory" in capsys.readouterr().err
    assert command_line.compute_torch_token_data_dir_error_rates([str(tmp_path / "nope")]) == 2
    assert command_line.main(["no-such-command"]) == 2
