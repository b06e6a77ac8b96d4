"""Command-line callers of the error-rate path.

The reference's ``compute-torch-token-data-dir-error-rates`` (command_line.py:848-1149) and
the two converters that produce / consume its inputs (``trn-to-torch-token-data-dir``
:305-392, ``torch-token-data-dir-to-trn`` :469-521), with the same arguments, outputs and
exit codes.  The directory scan and text handling stay on the host; every error count comes
from the Levenshtein kernel on the GPU (``pydrobert_amd.functional.error_rate``) -- there is
no CPU scoring path, so the scorer needs a ROCm device.

    python -m pydrobert_amd.command_line compute-torch-token-data-dir-error-rates DIR [HYP] [OUT]
"""
import argparse
import os
import sys
import warnings
from typing import Optional, Sequence

import numpy as np
import torch

from . import _parsing, config
from ._string import error_rate

__all__ = [
    "compute_torch_token_data_dir_error_rates",
    "torch_token_data_dir_to_trn",
    "trn_to_torch_token_data_dir",
]

_ID2TOKEN_HELP = (
    'A file containing mappings from unique IDs to tokens (e.g. words or phones). Each line has '
    'the format "<id> <token>". The flag "--swap" can be used to swap the expected ordering '
    '(i.e. to "<token> <id>")'
)
_TOKEN2ID_HELP = (
    'A file containing mappings from tokens (e.g. words or phones) to unique IDs. Each line has '
    'the format "<token> <id>". The flag "--swap" can be used to swap the expected ordering '
    '(i.e. to "<id> <token>")'
)


def _as_dir(path: str) -> str:
    if not os.path.isdir(path):
        raise argparse.ArgumentTypeError("'{}' is not a directory".format(path))
    return path


def _as_nat(val: str) -> int:
    i = int(val)
    if i < 1:
        raise argparse.ArgumentTypeError("{} is not a natural number".format(val))
    return i


def _add_file_args(parser):
    parser.add_argument("--file-prefix", default="",
                        help="The file prefix indicating a torch data file")  # fmt: skip
    parser.add_argument("--file-suffix", default=".pt",
                        help="The file suffix indicating a torch data file")  # fmt: skip
    parser.add_argument("--swap", action="store_true", default=False,
                        help="If set, swaps the order of the key and value in the mapping file")  # fmt: skip


def _list_utts(dir_: str, prefix: str, suffix: str):
    return sorted(
        x[len(prefix) : len(x) - len(suffix)]
        for x in os.listdir(dir_)
        if x.startswith(prefix) and x.endswith(suffix)
    )


def _load_transcripts(dir_, id2token, prefix, suffix, strip_timing):
    """Sorted ``(utt_id, transcript)`` of a token data dir (reference :394-466)."""
    for utt in _list_utts(dir_, prefix, suffix):
        tok = torch.load(os.path.join(dir_, prefix + utt + suffix))
        transcript = _parsing.token_to_transcript(tok, id2token)
        for i, token in enumerate(transcript):
            if isinstance(token, tuple):
                token = token[0]
                if strip_timing:
                    transcript[i] = token
            if isinstance(token, int) and id2token is not None:
                raise ValueError("Utterance '{}': ID '{}' could not be found in id2token".format(utt, token))
        yield utt, transcript


def trn_to_torch_token_data_dir(args: Optional[Sequence[str]] = None):
    """Convert a NIST "trn" file to a SpectDataSet token data dir (one ``.pt`` per utterance)."""
    parser = argparse.ArgumentParser(description=trn_to_torch_token_data_dir.__doc__)
    parser.add_argument("trn", type=argparse.FileType("r"), help="The input trn file")
    parser.add_argument("token2id", type=argparse.FileType("r"), help=_TOKEN2ID_HELP)
    parser.add_argument("dir", help="The directory to store token sequences to (created if missing)")
    parser.add_argument("--alt-handler", default="error", choices=("error", "first"),
                        help='How to handle transcription alternates: "error" out, or take the "first"')  # fmt: skip
    _add_file_args(parser)
    parser.add_argument("--unk-symbol", default=None,
                        help="If set, will map out-of-vocabulary tokens to this symbol")  # fmt: skip
    size = parser.add_mutually_exclusive_group()
    size.add_argument("--skip-frame-times", action="store_true", default=False,
                      help="Store token sequences of shape (R,) instead of (R, 3)")  # fmt: skip
    size.add_argument("--feat-sizing", action="store_true", default=False,
                      help="Store token sequences of shape (R, 1) instead of (R, 3)")  # fmt: skip
    try:
        options = parser.parse_args(args)
    except SystemExit as ex:
        return ex.code
    token2id = _parsing.parse_token2id(options.token2id, options.swap, options.swap)
    if options.unk_symbol is not None and options.unk_symbol not in token2id:
        print('Unk symbol "{}" is not in token2id'.format(options.unk_symbol), file=sys.stderr)
        return 1
    os.makedirs(options.dir, exist_ok=True)
    for utt_id, transcript in _parsing.read_trn_iter(options.trn):
        pending, flat = list(transcript), []
        while pending:
            x = pending.pop(0)
            if not isinstance(x, str) and len(x) == 3 and x[1] == -1:
                x = x[0]
            if isinstance(x, str):
                flat.append(x)
            elif options.alt_handler == "error":
                raise ValueError("Cannot handle alternate in '{}'".format(utt_id))
            else:  # the first alternate is canon
                pending = list(x[0]) + pending
        tok = _parsing.transcript_to_token(
            flat, token2id, None, options.unk_symbol, options.skip_frame_times or options.feat_sizing
        )
        if options.feat_sizing:
            tok = tok.unsqueeze(-1)
        torch.save(tok, os.path.join(options.dir, options.file_prefix + utt_id + options.file_suffix))
    return 0


def torch_token_data_dir_to_trn(args: Optional[Sequence[str]] = None):
    """Convert a SpectDataSet token data dir to a NIST "trn" file."""
    parser = argparse.ArgumentParser(description=torch_token_data_dir_to_trn.__doc__)
    parser.add_argument("dir", type=_as_dir, help="The directory to read token sequences from")
    parser.add_argument("id2token", type=argparse.FileType("r"), help=_ID2TOKEN_HELP)
    parser.add_argument("trn", type=argparse.FileType("w"), help='The "trn" file to write to')
    _add_file_args(parser)
    try:
        options = parser.parse_args(args)
    except SystemExit as ex:
        return ex.code
    id2token = _parsing.parse_token2id(options.id2token, not options.swap, options.swap)
    _parsing.write_trn(
        _load_transcripts(options.dir, id2token, options.file_prefix, options.file_suffix, True),
        options.trn,
    )
    return 0


def _pack(seqs, eos: int, padding: int) -> np.ndarray:
    """Right-pad ``seq + [eos]`` columns into one (T, N) int64 array."""
    T = max(len(s) for s in seqs) + 1
    out = np.full((T, len(seqs)), padding, dtype=np.int64)
    for n, s in enumerate(seqs):
        out[: len(s), n] = s
        out[len(s), n] = eos
    return out


def compute_torch_token_data_dir_error_rates(args: Optional[Sequence[str]] = None):
    """Compute error rates between reference and hypothesis token data dirs

    The error rate of the partition is the total number of insertions, deletions and
    substitutions over all transcriptions divided by the total reference length (or, with
    --distances, the mean distance per utterance).  Rates are printed as ratios, not
    percentages.  Counts come from the GPU Levenshtein kernel."""
    parser = argparse.ArgumentParser(description=compute_torch_token_data_dir_error_rates.__doc__)
    parser.add_argument("dir", type=_as_dir,
                        help="Parent of 'ref/' and 'hyp/', or the reference directory if 'hyp' is given")  # fmt: skip
    parser.add_argument("hyp", nargs="?", type=_as_dir, default=None, help="The hypothesis directory")
    parser.add_argument("out", nargs="?", type=argparse.FileType("w"), default=sys.stdout,
                        help="Where to print the error rate to. Defaults to stdout")  # fmt: skip
    parser.add_argument("--id2token", type=argparse.FileType("r"), default=None, help=_ID2TOKEN_HELP)
    parser.add_argument("--replace", type=argparse.FileType("r"), default=None,
                        help="File of pairs per line: the element to replace and its replacement "
                        "(tokens with --id2token, else integer IDs). Processed before --ignore")  # fmt: skip
    parser.add_argument("--ignore", type=argparse.FileType("r"), default=None,
                        help="File with a whitespace-delimited list of elements to drop from both "
                        "transcripts (tokens with --id2token, else integer IDs)")  # fmt: skip
    _add_file_args(parser)
    parser.add_argument("--warn-missing", action="store_true", default=False,
                        help="Warn about and exclude utterances missing a reference or a "
                        "hypothesis. The default is to error")  # fmt: skip
    parser.add_argument("--distances", action="store_true", default=False,
                        help="Return the average distance per utterance instead of the total "
                        "errors over the number of reference tokens")  # fmt: skip
    parser.add_argument("--per-utt", action="store_true", default=False,
                        help="Return lines of ``<utt_id> <error_rate>`` instead of the average")  # fmt: skip
    parser.add_argument("--batch-size", type=_as_nat, default=100,
                        help="The number of error rates to compute at once")  # fmt: skip
    parser.add_argument("--quiet", action="store_true", default=False,
                        help="Suppress warnings which arise from edit distance computations")  # fmt: skip
    parser.add_argument("--device", default="cuda", help="The ROCm device the kernel runs on")
    group = parser.add_mutually_exclusive_group()
    group.add_argument("--costs", nargs=3, type=float, metavar=("INS", "DEL", "SUB"),
                       default=(config.DEFT_INS_COST, config.DEFT_DEL_COST, config.DEFT_SUB_COST),
                       help="The costs of an insertion, deletion, and substitution, respectively")  # fmt: skip
    group.add_argument("--nist-costs", action="store_true", default=False,
                       help="Use NIST (sclite, score) default costs (3/3/4)")  # fmt: skip
    try:
        options = parser.parse_args(args)
    except SystemExit as ex:
        return ex.code
    if options.nist_costs:
        options.costs = (3.0, 3.0, 4.0)
    if options.hyp:
        ref_dir, hyp_dir = options.dir, options.hyp
    else:
        ref_dir, hyp_dir = os.path.join(options.dir, "ref"), os.path.join(options.dir, "hyp")
    for d in (ref_dir, hyp_dir):
        if not os.path.isdir(d):
            print('"{}" is not a directory'.format(d), file=sys.stderr)
            return 1
    id2token = None
    if options.id2token:
        id2token = _parsing.parse_token2id(options.id2token, not options.swap, options.swap)

    def element(x, file):
        if id2token is not None:
            return x
        try:
            return int(x)
        except ValueError:
            raise ValueError(
                'If --id2token is not set, all elements in "{}" must be integers'.format(file.name)
            )

    replace = dict()
    if options.replace:
        for line in options.replace:
            old, new = line.strip().split()
            replace[element(old, options.replace)] = element(new, options.replace)
    ignore = set()
    if options.ignore:
        ignore = {element(x, options.ignore) for x in options.ignore.read().strip().split()}

    refs = dict(_load_transcripts(ref_dir, id2token, options.file_prefix, options.file_suffix, True))
    hyps = dict(_load_transcripts(hyp_dir, id2token, options.file_prefix, options.file_suffix, True))
    for utt in sorted(set(refs) ^ set(hyps)):
        have, lack = (ref_dir, hyp_dir) if utt in refs else (hyp_dir, ref_dir)
        msg = 'Directory "{}" contains utterance "{}" which directory "{}" does not contain'.format(
            have, utt, lack
        )
        if not options.warn_missing:
            raise ValueError(msg)
        warnings.warn(msg + ". Skipping")
    utts = sorted(set(refs) & set(hyps))

    # tokens (after --replace / --ignore) -> dense non-negative ids shared by both sides
    ids = dict()

    def encode(transcript):
        out = []
        for t in transcript:
            t = replace.get(t, t)
            if t not in ignore:
                out.append(ids.setdefault(t, len(ids)))
        return out

    eos, padding = -1, -2
    device = torch.device(options.device)
    per_utt, tot_errs, tot_ref = [], 0.0, 0
    for lo in range(0, len(utts), options.batch_size):
        batch = utts[lo : lo + options.batch_size]
        ref_seqs = [encode(refs[u]) for u in batch]
        hyp_seqs = [encode(hyps[u]) for u in batch]
        ers = error_rate(
            torch.from_numpy(_pack(ref_seqs, eos, padding)).to(device),
            torch.from_numpy(_pack(hyp_seqs, eos, padding)).to(device),
            eos=eos, include_eos=False, norm=False, ins_cost=options.costs[0],
            del_cost=options.costs[1], sub_cost=options.costs[2], warn=not options.quiet,
        ).tolist()  # fmt: skip
        for u, seq, er in zip(batch, ref_seqs, ers):
            per_utt.append((u, er / (1 if options.distances else len(seq))))
            tot_errs += er
            tot_ref += len(seq)
    if options.per_utt:
        for u, er in per_utt:
            options.out.write("{} {}\n".format(u, er))
    else:
        options.out.write("{}\n".format(tot_errs / (len(per_utt) if options.distances else tot_ref)))
    if options.out is not sys.stdout:
        options.out.flush()
    return 0


_COMMANDS = {
    "compute-torch-token-data-dir-error-rates": compute_torch_token_data_dir_error_rates,
    "torch-token-data-dir-to-trn": torch_token_data_dir_to_trn,
    "trn-to-torch-token-data-dir": trn_to_torch_token_data_dir,
}


def main(argv: Optional[Sequence[str]] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in _COMMANDS:
        print("usage: python -m pydrobert_amd.command_line {{{}}} ...".format(",".join(_COMMANDS)), file=sys.stderr)
        return 2
    return _COMMANDS[argv[0]](argv[1:]) or 0


if __name__ == "__main__":
    sys.exit(main())
