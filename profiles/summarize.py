"""Turns the raw output of profiles/collect.sh (gpurun_out/prof) into the committed summaries:
r01_bench_kernel_stats.csv, r01_bench_line.json, r01_ctc_sq_counters.csv, r01_ctc_traffic.json.

    python profiles/summarize.py [gpurun_out/prof]
"""
import csv, glob, json, os, shutil, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
here = os.path.dirname(os.path.abspath(__file__))
KERNEL = "ctc_search_kernel<1, 4>"


def one(pattern):
    m = glob.glob(os.path.join(src, pattern), recursive=True)
    assert m, pattern
    return m[0]


def counters(path):
    """{kernel name: {counter: [values per dispatch]}}"""
    out = {}
    for r in csv.DictReader(open(path)):
        out.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return out


def mean(x):
    return sum(x) / len(x)


# kernel-trace statistics of the bench command, and its JSON line
shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(here, "r01_bench_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_line.json")) if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(here, "r01_bench_line.json"), "w"), indent=1)

# SQ counters of the dominant kernel (rows of the two passes, that kernel only)
rows, header = [], None
for p in ("sq1/**/*counter_collection.csv", "sq2/**/*counter_collection.csv"):
    rd = csv.reader(open(one(p)))
    header = next(rd)
    rows += [r for r in rd if KERNEL in r[header.index("Kernel_Name")]]
with open(os.path.join(here, "r01_ctc_sq_counters.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(header)
    w.writerows(rows)

# HBM traffic: FETCH_SIZE / WRITE_SIZE passes, rescaled by a kernel of known size (the randn
# that writes the logits: N*T*(V+1) floats)
fetch, write = counters(one("fetch/**/*counter_collection.csv")), counters(one("write/**/*counter_collection.csv"))
k = [n for n in fetch if KERNEL in n][0]
cal = [n for n in write if "normal_kernel" in n or "distribution" in n]
cal_name = max(cal, key=lambda n: mean(write[n]["WRITE_SIZE"]))
known = 512 * 4096 * 257 * 4
cal_kb = max(write[cal_name]["WRITE_SIZE"])
scale = known / 1024.0 / cal_kb
f_kb, w_kb = mean(fetch[k]["FETCH_SIZE"]), mean(write[k]["WRITE_SIZE"])
sq = counters(one("sq1/**/*counter_collection.csv"))[k]
rec = {
    "kernel": "pdt::ctc_search_kernel<1, 4>",
    "config": {"N": 4096, "T": 512, "V": 256, "beam": 16},
    "raw": {"FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb},
    "calibration": {
        "kernel": "at::native distribution (randn) writing a (512,4096,257) f32 tensor",
        "known_bytes": known, "WRITE_SIZE_KB": cal_kb, "scale_vs_KB": scale,
    },
    "hbm_bytes_per_launch": (f_kb + w_kb) * 1024.0 * scale,
    "note": "FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc passes (profiles/collect.sh, "
            "profiles/prof_ctc.py); on this gfx950 image both counters read a fixed fraction of a known "
            "streaming byte count, so they are rescaled by the calibration kernel's known size "
            "(MI355X_MICROARCH.md, HBM section).",
    "sq": {
        "SQ_INSTS_VALU_per_launch": mean(sq["SQ_INSTS_VALU"]),
        "SQ_INSTS_SALU_per_launch": mean(sq["SQ_INSTS_SALU"]),
        "SQ_INSTS_LDS_per_launch": mean(sq["SQ_INSTS_LDS"]),
        "note": "wave-level instruction counts of one launch (profiles/r01_ctc_sq_counters.csv); a wave64 VALU "
                "instruction occupies its SIMD for 4 cycles, 1024 SIMDs at 2.4 GHz",
    },
}
json.dump(rec, open(os.path.join(here, "r01_ctc_traffic.json"), "w"), indent=1)
print(json.dumps({"hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "scale": scale,
                  "valu_per_frame_utt": rec["sq"]["SQ_INSTS_VALU_per_launch"] / (4096 * 512)}))
