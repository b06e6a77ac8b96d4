"""Static instruction mix of one kernel in a hipcc -S listing, priced with the issue costs of
profiles/r03_valu_issue_cost.txt (ns per wave64 instruction per SIMD with four waves resident):
    python profiles/tools/valu_mix.py file.s <kernel-name-substring> [first_line last_line]
Fast class (1.1 ns): v_add/sub(rev)_u32, v_and/or/xor/not_b32, v_mov_b32, v_lshrrev/ashrrev, v_add/sub/mul_f32
with VGPR / inline / literal operands.  Everything else vector: 1.8 ns; transcendentals 3.4; an SGPR
operand makes a fast instruction slow."""
import re, sys, collections

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_subrev_f32"}
TRANS = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_swap_b32"}


def price(line):
    parts = line.split()
    op = parts[0]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if not op.startswith("v_"):
        return None
    if base in TRANS:
        return "trans", 3.4
    if op.endswith("_dpp") or op.endswith("_sdwa") or " row_" in line or "quad_perm" in line:
        return "slow", 1.8
    if base in FAST:
        ops = " ".join(parts[1:])
        srcs = ops.split(",")[1:]
        if any(re.match(r"\s*(s\d+|s\[|vcc|exec)", x) for x in srcs):
            return "fast+sgpr", 1.8
        return "fast", 1.1
    return "slow", 1.8


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^\S*" + re.escape(pat) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, end - start)
    cls = collections.Counter(); ns = collections.Counter(); ops = collections.Counter(); other = collections.Counter()
    for l in lines[start + lo:start + hi]:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            continue
        p = price(l)
        if p is None:
            other[l.split()[0].split("_")[0]] += 1
            continue
        cls[p[0]] += 1; ns[p[0]] += p[1]; ops[re.sub(r"_(e32|e64)$", "", l.split()[0])] += 1
    tot = sum(cls.values())
    print("vector instructions %d, priced %.0f ns per pass (%.2f ns each); scalar/lds/other: %s" % (tot, sum(ns.values()), sum(ns.values()) / max(tot, 1), dict(other)))
    for k in cls:
        print("  %-10s %5d  %7.0f ns" % (k, cls[k], ns[k]))
    print("  top:", ", ".join("%s %d" % kv for kv in ops.most_common(14)))


main()
