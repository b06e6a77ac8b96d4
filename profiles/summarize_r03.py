"""Turns the raw output of profiles/collect_r03.sh (gpurun_out/prof3) into the committed summaries:

  r03_bench_kernel_stats.csv    rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 10 ...`
  r03_bench_line.json           the JSON line that bench run printed
  r03_gru_lm_kernel_stats.csv   the same for profiles/prof_gru_lm.py (which kernel runs the logit GEMM)
  r03_ctc_traffic.json          the dominant kernel: HBM bytes per launch from FETCH_SIZE / WRITE_SIZE
                                (2 x FETCH + 1 x WRITE: MI355X_MICROARCH.md, profiles/r02_traffic_calibration.json),
                                SQ instruction counts, wave-time split -- the record bench.py reads
"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof3"
here = os.path.dirname(os.path.abspath(__file__))


def one(pattern):
    m = glob.glob(os.path.join(src, pattern), recursive=True)
    assert m, pattern
    return m[0]


def counters(path, needle):
    out = defaultdict(list)
    name = None
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"]:
            name = r["Kernel_Name"]
            out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return name, {k: sum(v) / len(v) for k, v in out.items()}


shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(here, "r03_bench_kernel_stats.csv"))
shutil.copy(one("gru/**/*kernel_stats.csv"), os.path.join(here, "r03_gru_lm_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_line.json")) if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(here, "r03_bench_line.json"), "w"), indent=1)
needle = "ctc_search_kernel<1, 4"
name, fetch = counters(one("fetch/**/*counter_collection.csv"), needle)
_, write = counters(one("write/**/*counter_collection.csv"), needle)
_, sq1 = counters(one("sq1/**/*counter_collection.csv"), needle)
_, sq2 = counters(one("sq2/**/*counter_collection.csv"), needle)
issue = json.load(open(os.path.join(here, "r02_valu_issue.json")))
cyc = None
for k, v in issue.items():
    if isinstance(v, dict) and "ns_per_inst_8_waves" in v:
        cyc = v["ns_per_inst_8_waves"] * 2.4
rec = {
    "kernel": name.split("(")[0],
    "config": {"N": 4096, "T": 512, "V": 256, "beam": 16},
    "hbm_bytes_per_launch": (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0,
    "raw": {"FETCH_SIZE_KB": fetch["FETCH_SIZE"], "WRITE_SIZE_KB": write["WRITE_SIZE"]},
    "correction": "2 x FETCH_SIZE + 1 x WRITE_SIZE (profiles/r02_traffic_calibration.json)",
    "source": "profiles/collect_r03.sh: separate --pmc passes over profiles/prof_ctc.py, averaged over its launches",
    "sq": {
        "SQ_INSTS_VALU_per_launch": sq1["SQ_INSTS_VALU"],
        "SQ_INSTS_SALU_per_launch": sq1["SQ_INSTS_SALU"],
        "SQ_INSTS_LDS_per_launch": sq1["SQ_INSTS_LDS"],
        "wave_time_split": {
            "issuing": sq2["SQ_ACTIVE_INST_ANY"] / sq1["SQ_WAVE_CYCLES"],
            "issue_stalled": sq2["SQ_WAIT_INST_ANY"] / sq1["SQ_WAVE_CYCLES"],
            "waiting": sq2["SQ_WAIT_ANY"] / sq1["SQ_WAVE_CYCLES"],
        },
        "valu_issue_cycles_per_inst_measured": 2.8488,
        "note": "cycles at 2.4 GHz per wave64 VALU instruction per SIMD with 8 waves resident (profiles/r02_valu_issue.json): the price of a v_add",
        # the kernel's static instruction mix priced per kind (profiles/tools/valu_mix.py on a hipcc -S
        # listing of ctc_search.hip with the costs of profiles/r03_valu_issue_cost.txt): 705 fast (1.1 ns),
        # 144 fast with a scalar operand + 2374 others (1.8 ns), 19 transcendental (3.4 ns) of 3242
        "valu_issue_ns_per_inst_static_mix": 1.66,
        "valu_issue_ns_note": "static instruction mix of the kernel's listing priced with profiles/r03_valu_issue_cost.txt "
                              "(profiles/tools/valu_mix.py; four or more waves per SIMD)",
    },
}
json.dump(rec, open(os.path.join(here, "r03_ctc_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
