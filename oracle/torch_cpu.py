"""TEST / BENCH INFRASTRUCTURE ONLY -- never imported by the product package.

The CPU baseline of bench.py: what a user of the reference experiences on the host cores, i.e.
the reference's *tensor-level* algorithm (dense torch ops over the whole batch, one Python step
per hypothesis token / per frame) restated with stock torch CPU operators.  The C restatement in
pdt_oracle_*.c computes the same values with scalar loops and is timed separately ("port").

Restated from (reference src/pydrobert/torch/):
  * `levenshtein_rows`    -- _string.py:146-406 for the uniform-cost case (:168-174 rescales any
                            uniform costs to 1 and drops the mistake counters), including the
                            O(R^2) deletion closure through an (R+1, R+1, 1) offset matrix
                            (:258-266, :316-317), per-prefix read-outs (:341-346) and row-minimum
                            masks (:319-338);
  * `optimal_completion`  -- _string.py:464-517 (duplicate closure over an (H, N, R, R)
                            comparison, sort, masked scatter into (H, N, C));
  * `ctc_advance`         -- _decoding.py:775-934, the dense candidate tensors of one frame;
  * `ctc_prefix_search`   -- _decoding.py:1064-1202 without a language model and with full
                            lengths (the bench's configuration).
Checked against the C oracle in tests/test_oracle_golden.py::test_torch_cpu_restatement.
"""
import torch

INF = float("inf")


def levenshtein_rows(ref, hyp, want="final", exclude_last=False):
    """ref (R, N), hyp (H, N) int64, every sequence full length, unit costs.

    want = "final": (N,) distances; "prefix": (H+1 or H, N) distances of every hypothesis
    prefix; "mask": (H+1 or H, R, N) bool, True where a DP row attains its minimum."""
    R, N = ref.shape
    H = hyp.shape[0]
    steps = H + (0 if exclude_last else 1)
    cols = torch.arange(R + 1, dtype=torch.float)
    # offsets[i, j] = (i - j) deletions to reach column i from column j <= i, inf above the diagonal
    offsets = cols.unsqueeze(1) - cols
    offsets = (offsets + torch.full_like(offsets, INF).triu(1)).unsqueeze(-1)
    row = cols.unsqueeze(1).expand(R + 1, N)
    ref_lens = torch.full((N,), R, dtype=torch.long)
    hyp_lens = torch.full((N,), H, dtype=torch.long)
    masks, prefix = [], None
    if want == "mask":
        first = torch.zeros((R, N), dtype=torch.bool)
        first[0] = ref_lens > 0
        masks.append(first)
    elif want == "prefix":
        prefix = torch.empty((steps, N))
        prefix[0] = ref_lens.float()
    for h in range(1, steps):
        live = (h - (0 if exclude_last else 1)) < hyp_lens
        above = row
        can_insert = (hyp_lens >= h).float()
        differs = (ref != hyp[h - 1]).float()
        row = above + can_insert  # insertion everywhere
        diag = above[:-1] + differs  # substitution / match
        row[1:] = torch.min(row[1:], diag)
        row = (offsets + row).min(1)[0]  # deletions: min over every column to the left
        row = torch.where(live, row, above)
        if want == "mask":
            row = row.masked_fill(cols.unsqueeze(1) > ref_lens, INF)
            low = row.min(0, keepdim=True)[0]
            masks.append((row[:-1] == low) & live)
        elif want == "prefix":
            prefix[h] = row.gather(0, ref_lens.unsqueeze(0)).squeeze(0)
    if want == "mask":
        inside = torch.arange(R).unsqueeze(1).expand(R, N) < ref_lens
        return torch.stack(masks, 0) & inside.unsqueeze(0)
    if want == "prefix":
        return prefix
    return row.gather(0, ref_lens.unsqueeze(0)).squeeze(0)


def error_rate(ref, hyp, norm=True):
    er = levenshtein_rows(ref, hyp)
    return er / ref.shape[0] if norm else er


def prefix_error_rates(ref, hyp, norm=True, exclude_last=False):
    er = levenshtein_rows(ref, hyp, "prefix", exclude_last)
    return er / ref.shape[0] if norm else er


def optimal_completion(ref, hyp, padding=-100, exclude_last=False):
    mask = levenshtein_rows(ref, hyp, "mask", exclude_last)  # (H', R, N)
    toks = ref.t()  # (N, R)
    Hp, R, N = mask.shape
    # a token that is optimal at one position is optimal wherever it occurs
    same = toks.unsqueeze(1) == toks.unsqueeze(2)  # (N, R, R)
    mask = (mask.transpose(1, 2).unsqueeze(2) & same).any(3)  # (H', N, R)
    toks, order = toks.sort(1)
    mask = mask.gather(2, order.expand_as(mask))
    keep_last = (toks[:, :-1] != toks[:, 1:]).expand(Hp, -1, -1)  # one survivor per run of equals
    mask = torch.cat([mask[..., :-1] & keep_last, mask[..., -1:]], 2)
    flat = toks.expand_as(mask).masked_select(mask)
    counts = mask.sum(2)
    C = int(counts.max().item())
    out = torch.full((Hp, N, C), padding, dtype=torch.long)
    out.masked_scatter_(counts.unsqueeze(-1) > torch.arange(C), flat)
    return out


def ctc_advance(ext, nonext, blank, width, nb, b, y, last, lens, is_prefix):
    """One frame over dense (N, K', V) candidate tensors; returns the new state."""
    N, Kp, V = ext.shape
    S = y.shape[0]
    K = min(width, Kp * (V + 1))
    total = nb + b
    last = last.clamp(0, V - 1)
    # extension by v: (nb unless v repeats the last token) + b, times the extension probability
    grow = (nb.unsqueeze(2).expand(N, Kp, V).scatter(2, last.unsqueeze(2), 0.0) + b.unsqueeze(2)) * ext
    stay_b = total * blank.unsqueeze(1)
    stay_nb = nb * nonext.gather(1, last)
    # token prefix k would need next in order to stay a prefix of k'
    if S:
        at = lens.clamp(max=S - 1).unsqueeze(2).expand(N, Kp, Kp).transpose(0, 1)
        need = y.gather(0, at).transpose(0, 1).clamp(0, V - 1)
    else:
        need = torch.zeros((N, Kp, Kp), dtype=y.dtype)
    becomes = ((lens + 1).unsqueeze(2) == lens.unsqueeze(1)) & is_prefix  # k + need == k' exactly
    stay_nb = stay_nb + grow.gather(2, need).masked_fill(~becomes, 0.0).sum(1)
    taken = (torch.nn.functional.one_hot(need, V).to(torch.bool) & becomes.unsqueeze(3)).any(2)
    grow = grow.masked_fill(taken, -INF)
    cand = torch.cat([grow.view(N, Kp * V), stay_nb + stay_b], 1)
    pick = cand.topk(K, 1)[1]
    kept = pick >= Kp * V
    src = torch.where(kept, pick - Kp * V, torch.div(pick, V, rounding_mode="trunc"))
    tok = pick % V
    base_len = lens.gather(1, src)
    y_new = torch.cat([y.gather(2, src.unsqueeze(0).expand(S, N, K)), torch.empty((1, N, K), dtype=y.dtype)], 0)
    y_new = y_new.scatter(0, base_len.unsqueeze(0), tok.unsqueeze(0))
    lens_new = base_len + (~kept)
    nb_new = torch.where(kept, stay_nb.gather(1, src), grow.view(N, Kp * V).gather(1, pick.clamp(max=Kp * V - 1)))
    b_new = stay_b.gather(1, src) * kept
    last_new = last.gather(1, src) * kept + tok * (~kept)
    rel = is_prefix.gather(1, src.unsqueeze(2).expand(N, K, Kp)).gather(2, src.unsqueeze(1).expand(N, K, K))
    shorter = lens_new.unsqueeze(2) <= lens_new.unsqueeze(1)
    at = (lens_new - 1).clamp(min=0).unsqueeze(2).expand(N, K, K).transpose(0, 1)
    agrees = y_new.gather(0, at).transpose(0, 1) == tok.unsqueeze(2)
    is_prefix_new = rel & shorter & (kept.unsqueeze(2) | (~kept.unsqueeze(2) & agrees))
    if K < width:
        pad = width - K
        y_new = torch.cat([y_new, y_new.new_empty(S + 1, N, pad)], 2)
        zeros = torch.zeros((N, pad), dtype=last_new.dtype)
        last_new, lens_new = torch.cat([last_new, zeros], 1), torch.cat([lens_new, zeros], 1)
        minus = torch.full((N, pad), -INF)
        nb_new, b_new = torch.cat([nb_new, minus], 1), torch.cat([b_new, minus], 1)
        no = torch.zeros((N, pad), dtype=torch.bool)
        is_prefix_new = torch.cat([is_prefix_new, no.unsqueeze(1).expand(N, K, pad)], 2)
        is_prefix_new = torch.cat([is_prefix_new, no.unsqueeze(2).expand(N, pad, width)], 1)
    return y_new, last_new, lens_new, nb_new, b_new, is_prefix_new


def ctc_prefix_search(logits, width):
    """logits (T, N, V + 1) float32, blank last, every utterance T frames, no language model."""
    T, N, V1 = logits.shape
    V = V1 - 1
    probs = logits.softmax(2)
    nb, b = torch.zeros((N, 1)), torch.ones((N, 1))
    y = torch.empty((0, N, 1), dtype=torch.long)
    lens = last = torch.zeros((N, 1), dtype=torch.long)
    is_prefix = torch.ones((N, 1, 1), dtype=torch.bool)
    for t in range(T):
        nonext, blank = probs[t, :, :V], probs[t, :, V]
        ext = nonext.unsqueeze(1).expand(N, nb.shape[1], V)
        y, last, lens, nb, b, is_prefix = ctc_advance(ext, nonext, blank, width, nb, b, y, last, lens, is_prefix)
    total = nb + b
    if T == 0 and width != 1:
        y, lens = y.repeat(1, 1, width), lens.repeat(1, width)
        total = torch.cat([total, total.new_full((N, width - 1), -INF)], 1)
    return y, lens, total
