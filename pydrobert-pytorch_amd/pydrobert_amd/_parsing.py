"""Transcript formats on either side of the error-rate path.

Mirrors the parts of the reference's ``_parsing.py`` that the scoring caller needs: NIST
"trn" files (:283-440) and the token-sequence <-> transcript conversions (:700-900) that define
what a SpectDataSet ``ref/`` or ``hyp/`` ``.pt`` file holds.  Host-side text work: none of it
touches the GPU.
"""
import warnings
from typing import Dict, Iterable, Iterator, List, Optional, TextIO, Tuple, Union

import numpy as np
import torch

__all__ = [
    "parse_token2id",
    "read_trn",
    "read_trn_iter",
    "token_to_transcript",
    "transcript_to_token",
    "write_trn",
]


def _parse_trn_line(line: str, warn: bool):
    """One "trn" line -> (utt_id, transcript) or None for a blank line.

    sclite's conventions (reference _parsing.py:308-320): the LAST parenthesised group is the
    utterance id and anything after it is ignored; ``{ a / b }`` is a set of alternates, kept
    as ``([[a...], [b...]], -1, -1)`` (they nest); ``/`` and ``}`` outside an alternate are
    ordinary characters; an alternate left open at the end of the line is dropped.
    """
    line = line.strip()
    if not line:
        return None
    close = line.rfind(")")
    open_ = line.rfind("(")
    if open_ < 0 or close < 0 or open_ > close:
        raise IOError("Line does not end in utterance id")
    utt_id = line[open_ + 1 : close]
    body = line[:open_].strip()
    transcript: List = []
    # stack of open alternates: each is a list of branches, each branch a list of items
    stack: List[List[List]] = []
    word = ""
    saw_alternate = False

    def flush():
        nonlocal word
        if word:
            (stack[-1][-1] if stack else transcript).append(word)
            word = ""

    for c in body:
        if c == "{":
            saw_alternate = True
            flush()
            stack.append([[]])
        elif c == "/" and stack:
            flush()
            stack[-1].append([])
        elif c == "}" and stack:
            flush()
            if not stack[-1][-1]:
                raise IOError('Empty alternate found ("{ }")')
            done = stack.pop()
            if stack:
                stack[-1][-1].append(done)
            else:
                transcript.append((done, -1, -1))
        elif c == " ":
            flush()
        else:
            word += c
    if not stack:
        flush()
    if saw_alternate and warn:
        warnings.warn(
            'Found an alternate in transcription for utt="{}". Transcript will contain an array '
            "of alternates at that point, and will not be compatible with transcript_to_token "
            "until resolved. To suppress this warning, set warn=False".format(utt_id)
        )
    return utt_id, transcript


def read_trn_iter(
    trn: Union[TextIO, str], warn: bool = True, processes: int = 0, chunk_size: int = 1000
) -> Iterator[Tuple[str, List]]:
    """Yield ``(utt_id, transcript)`` from a NIST "trn" file (reference _parsing.py:283-337).
    ``processes`` / ``chunk_size`` are accepted for signature parity; parsing is a linear
    scan and stays on the calling thread."""
    if isinstance(trn, str):
        with open(trn) as f:
            yield from read_trn_iter(f, warn, processes, chunk_size)
        return
    for line in trn:
        parsed = _parse_trn_line(line, warn)
        if parsed is not None:
            yield parsed


def read_trn(
    trn: Union[TextIO, str], warn: bool = True, processes: int = 0, chunk_size: int = 1000
) -> List[Tuple[str, List]]:
    """Read a NIST "trn" file into a list of ``(utt_id, transcript)`` (reference :340-390)."""
    return list(read_trn_iter(trn, warn, processes, chunk_size))


def _is_timed(x) -> bool:
    try:
        return len(x) == 3 and not isinstance(x, str) and np.isreal(x[1]) and np.isreal(x[2])
    except TypeError:
        return False


def write_trn(transcripts: Iterable[Tuple[str, List]], trn: Union[str, TextIO]) -> None:
    """Write ``(utt_id, transcript)`` pairs as a "trn" file (reference :393-440): start / end
    times are dropped, alternates are written back as ``{ a / b }``."""
    if isinstance(trn, str):
        with open(trn, "w") as f:
            return write_trn(transcripts, f)

    def render(x) -> str:
        if isinstance(x, str):
            return x + " "
        return "{ " + "/ ".join("".join(render(item) for item in branch) for branch in x) + "} "

    for utt_id, transcript in transcripts:
        parts = [render(x[0] if _is_timed(x) else x) for x in transcript]
        trn.write("".join(parts) + "(" + utt_id + ")\n")


def transcript_to_token(
    transcript: List,
    token2id: Optional[dict] = None,
    frame_shift_ms: Optional[float] = None,
    unk: Optional[Union[str, int]] = None,
    skip_frame_times: bool = False,
) -> torch.Tensor:
    """Transcript -> the long tensor a token data dir stores (reference :700-855): ``(R, 3)``
    rows of ``(id, start_frame, end_frame)`` (-1 when unknown), or ``(R,)`` ids when
    ``skip_frame_times``."""
    if token2id is not None and unk in token2id:
        unk = token2id[unk]
    rows = np.empty((len(transcript), 3), dtype=np.int64)
    for i, token in enumerate(transcript):
        start = end = -1
        if _is_timed(token):
            token, start, end = token
            if frame_shift_ms:
                if start == end:
                    start = end = (1000 * start) // frame_shift_ms
                else:
                    start = (1000 * start) // frame_shift_ms
                    end = max((1000 * end + 0.5 * frame_shift_ms) // frame_shift_ms, start + 1)
            else:
                start, end = int(start), int(end)
        if token2id is None:
            id_ = token
        else:
            id_ = token2id.get(token, token if unk is None else unk)
        rows[i] = (id_, start, end)  # raises for a token that is neither mapped nor an int
    tok = torch.from_numpy(rows)
    return tok[:, 0].clone() if skip_frame_times else tok


def token_to_transcript(
    ref: torch.Tensor, id2token: Optional[Dict[int, str]] = None, frame_shift_ms: Optional[float] = None
) -> List:
    """Inverse of :func:`transcript_to_token` (reference :858-902): ``ref`` is ``(R, 3)``,
    ``(R, 1)`` or ``(R,)``."""
    arr = ref.detach().cpu().numpy()
    if arr.ndim == 1:
        arr = arr[:, None]
    timed = arr.shape[1] == 3
    out = []
    for row in arr:
        id_ = int(row[0])
        token = id2token.get(id_, id_) if id2token is not None else id_
        start, end = (int(row[1]), int(row[2])) if timed else (-1, -1)
        if start == -1 or end == -1:
            out.append(token)
        elif frame_shift_ms:
            out.append((token, start * frame_shift_ms / 1000, end * frame_shift_ms / 1000))
        else:
            out.append((token, start, end))
    return out


def parse_token2id(file: TextIO, swap: bool, return_swap: bool) -> dict:
    """A two-column ``token id`` (or ``id token`` with ``swap``) file -> dictionary (reference
    command_line.py:265-289); ``return_swap`` returns the inverse map."""
    forward, backward = dict(), dict()
    for line_no, line in enumerate(file):
        line = line.strip()
        if not line:
            continue
        cols = line.split()
        if len(cols) != 2 or not cols[1 - int(swap)].lstrip("-").isdigit():
            raise ValueError("Cannot parse line {} of {}".format(line_no + 1, file.name))
        key, value = (int(cols[0]), cols[1]) if swap else (cols[0], int(cols[1]))
        for table, k in ((forward, key), (backward, value)):
            if k in table:
                warnings.warn(
                    '{} line {}: "{}" already exists. Mapping will be ambiguous'.format(file.name, line_no + 1, k)
                )
        forward[key] = value
        backward[value] = key
    return backward if return_swap else forward
