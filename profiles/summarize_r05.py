"""Turns the raw output of profiles/collect_r05.sh (gpurun_out/r05) into the committed summaries:

  r05_bench_kernel_stats.csv        rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 10 --no-extra ...`
  r05_bench_line.json               the JSON line that run printed
  r05_bench_full_kernel_stats.csv   the same for the whole line (every other config's kernels by name)
  r05_bench_full_line.json
  r05_ctc_traffic.json              the headline kernel: HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE), SQ
                                    instruction counts, wave-time split, effective clock -- the record bench.py reads
  r05_c5_rowreg.json, r05_c3_rowreg.json, r05_lm_table.json, r05_sparse_warp.json
                                    the same for the C5 shard decode, the C3 search, C3 + bigram model, C4 sparse warp
"""
import glob, json, os, shutil, subprocess, sys

src = os.path.join(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out", "r05")
here = os.path.dirname(os.path.abspath(__file__))


def one(pattern):
    m = glob.glob(os.path.join(src, pattern), recursive=True)
    assert m, pattern
    return m[0]


for name in ("bench", "bench_full"):
    shutil.copy(one(name + "_stats/**/*kernel_stats.csv"), os.path.join(here, "r05_%s_kernel_stats.csv" % name))
    line = [l for l in open(os.path.join(src, name + "_line.json")) if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(here, "r05_%s_line.json" % name), "w"), indent=1)


def summary(sub, needle, out, extra=None):
    rec = json.loads(subprocess.check_output(
        [sys.executable, os.path.join(here, "tools", "pmc_summary.py"), os.path.join(src, sub), needle]).decode())
    rec.update(extra or {})
    json.dump(rec, open(os.path.join(here, out), "w"), indent=1)
    return rec


alg = lambda T, N, V, K: (4 * T * (V + 1) + 8 * T * K + 12 * K) * N  # noqa: E731
head = summary("pmc_headline", "ctc_search_kernel<1, 4", "r05_headline_raw.json")
os.remove(os.path.join(here, "r05_headline_raw.json"))
issue = {"fast_cycles": 2.5, "other_cycles": 4.3, "transcendental_cycles": 8.2}
rec = {
    "kernel": head["kernel"], "config": {"N": 4096, "T": 512, "V": 256, "beam": 16},
    "avg_ms_profiled": head["avg_ms"],
    "algorithmic_bytes_per_launch": alg(512, 4096, 256, 16),
    "hbm_bytes_per_launch": head["hbm_bytes_per_launch"],
    "raw": head["raw_KB"],
    "correction": "2 x FETCH_SIZE + 1 x WRITE_SIZE (MI355X_MICROARCH.md, HBM; profiles/r02_traffic_calibration.json)",
    "source": "profiles/collect_r05.sh: separate --pmc passes over profiles/prof_ctc.py, averaged over its launches",
    "effective_clock_GHz": head.get("effective_clock_GHz"),
    "sq": {
        "SQ_INSTS_VALU_per_launch": head["SQ_INSTS_VALU"], "SQ_INSTS_SALU_per_launch": head["SQ_INSTS_SALU"],
        "SQ_INSTS_LDS_per_launch": head["SQ_INSTS_LDS"], "wave_time_split": head["wave_time_split"],
        "valu_pipe_busy_frac_measured": head.get("valu_pipe_busy_frac"), "valu_cycles_per_inst_measured": head.get("valu_cycles_per_inst"),
        "valu_issue_cycles": issue,
        "valu_issue_note": "profiles/r04_valu_issue_cost.txt (round 4, not re-measured): ns per wave64 instruction per SIMD with four waves resident, "
                           "turned into shader cycles by the s_memtime / s_memrealtime clock of the same kernel "
                           "(profiles/tools/micro/valu_cost.hip): add / sub / and / or / xor / mov / right shifts / f32 add, "
                           "mul 2.5; everything else (fma, min / max, compares, cndmask, DPP, lshl, any SGPR operand) 4.3; "
                           "exp / log / rcp 8.2",
    },
}
json.dump(rec, open(os.path.join(here, "r05_ctc_traffic.json"), "w"), indent=1)
summary("pmc_c5", "ctc_rowreg", "r05_c5_rowreg.json",
        {"config": {"N": 4096, "T": 512, "V": 5000, "beam": 16}, "algorithmic_bytes_per_launch": alg(512, 4096, 5000, 16)})
summary("pmc_c3", "ctc_rowreg", "r05_c3_rowreg.json",
        {"config": {"N": 1024, "T": 1000, "V": 1000, "beam": 16}, "algorithmic_bytes_per_launch": alg(1000, 1024, 1000, 16)})
summary("pmc_lm", "ctc_lm_table", "r05_lm_table.json",
        {"config": {"N": 1024, "T": 1000, "V": 1000, "beam": 16, "lm": "bigram LookupLanguageModel, shallow fusion 0.2, "
                    "speechlike logits"}, "algorithmic_bytes_per_launch": alg(1000, 1024, 1000, 16)})
summary("pmc_warp", "sparse_warp_bands", "r05_sparse_warp.json",
        {"config": {"N": 2048, "C": 1, "H": 1000, "W": 80, "centres": 7, "order": 2},
         "algorithmic_bytes_per_launch": 2 * 4 * 2048 * 1000 * 80})
S = 100
summary("pmc_beam_advance", "beam_advance_flat_kernel", "r05_advance_beam.json",
        {"config": {"N": 1024, "K": 16, "V": 1000, "S": S}, "algorithmic_bytes_per_launch": (4 * 16 * 1000 + 8 * S * 16 + 8 * (S + 1) * 16 + 4 * 8 * 16) * 1024})
summary("pmc_ctc_advance", "ctc_advance_kernel", "r05_advance_ctc.json",
        {"config": {"N": 1024, "K": 16, "V": 1000, "S": S},
         "algorithmic_bytes_per_launch": (4 * 1001 + 8 * S * 16 + 8 * (S + 1) * 16 + 16 * 16 + 5 * 8 * 16) * 1024,
         "note": "avg over the launches of profiles/prof_steps.py ctc: the 100 steps that build the history (K' = 1, then 16) and 5 at S = 100"})
summary("pmc_spec", "spec_augment_rows_kernel", "r05_spec_augment.json",
        {"config": {"N": 2048, "T": 1000, "F": 80}, "algorithmic_bytes_per_launch": 2 * 4 * 2048 * 1000 * 80})
for f in sorted(glob.glob(os.path.join(here, "r05_*.json"))):
    print(f)
