"""Turns the raw output of profiles/collect.sh (gpurun_out/prof) into the committed summaries
profiles/r02_*:

    python profiles/summarize.py [gpurun_out/prof]

  r02_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 10 ...`
  r02_ops_kernel_stats.csv     the same for profiles/prof_ops.py (every hot kernel at its BASELINE shape)
  r02_bench_line.json          the JSON line that bench run printed
  r02_valu_issue.json          microbench: VALU issue rate / dependent-chain latencies
  r02_traffic_calibration.json FETCH_SIZE / WRITE_SIZE read on streaming kernels of known size
  r02_store_patterns.json      microbench: 16-byte store streams by layout
  r02_kernels.json             per kernel: duration, HBM bytes (corrected counters), SQ counters,
                               instructions per unit of work
  r02_ctc_traffic.json         the dominant kernel's record in the form bench.py reads
"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
here = os.path.dirname(os.path.abspath(__file__))


def one(pattern):
    m = glob.glob(os.path.join(src, pattern), recursive=True)
    assert m, pattern
    return m[0]


def counters(path):
    """{kernel name: {counter: [value per dispatch]}}"""
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))  # launch order
    out = defaultdict(lambda: defaultdict(list))
    for r in rows:
        out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def durations(path):
    out = defaultdict(list)
    for r in sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"])):  # launch order
        out[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out


def mean(x):
    return sum(x) / len(x)


shutil.copy(one("stats/**/*kernel_stats.csv"), os.path.join(here, "r02_bench_kernel_stats.csv"))
shutil.copy(one("opstats/**/*kernel_stats.csv"), os.path.join(here, "r02_ops_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_line.json")) if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(here, "r02_bench_line.json"), "w"), indent=1)
issue = json.load(open(os.path.join(src, "valu_issue.json")))
json.dump(issue, open(os.path.join(here, "r02_valu_issue.json"), "w"), indent=1)
stores = json.load(open(os.path.join(src, "store_patterns.json")))
stores["tiled_traversal"] = json.load(open(os.path.join(src, "store_tiles.json")))
json.dump(stores, open(os.path.join(here, "r02_store_patterns.json"), "w"), indent=1)

# ---- calibration of the two traffic counters on kernels of known size (1 GiB each)
GiB_KB = 1024.0 * 1024.0
cal = {"bytes_each": 1 << 30, "FETCH_SIZE_KB": {}, "WRITE_SIZE_KB": {}}
for k, v in counters(one("fetch_cal/**/*counter_collection.csv")).items():
    if "k_read" in k:
        cal["FETCH_SIZE_KB"]["read4" if "<float>" in k else "read16"] = mean(v["FETCH_SIZE"])
for k, v in counters(one("write_cal/**/*counter_collection.csv")).items():
    if "k_write" in k:
        name = "write4" if "<float>" in k else ("write8" if "2u" in k else "write16")
        cal["WRITE_SIZE_KB"][name] = mean(v["WRITE_SIZE"])
fetch_scale = GiB_KB / mean(list(cal["FETCH_SIZE_KB"].values()))
write_scale = GiB_KB / mean(list(cal["WRITE_SIZE_KB"].values()))
cal["fetch_bytes_per_counter_KB"] = 1024.0 * fetch_scale
cal["write_bytes_per_counter_KB"] = 1024.0 * write_scale
cal["note"] = ("FETCH_SIZE reads half of a known streaming read (4 and 16 B per lane alike), WRITE_SIZE reads "
               "4 / 8 / 16 B per lane stores exactly: as MI355X_MICROARCH.md states; every HBM figure below "
               "is 2 x FETCH_SIZE + WRITE_SIZE")
json.dump(cal, open(os.path.join(here, "r02_traffic_calibration.json"), "w"), indent=1)

# ---- per-kernel records
ns_per_inst = issue["independent_v_add_f32"]["waves_per_simd_8"]["ns_per_wave_inst_per_simd_by_event"]
fetch, write = counters(one("fetch/**/*counter_collection.csv")), counters(one("write/**/*counter_collection.csv"))
sq1, sq2 = counters(one("sq1/**/*counter_collection.csv")), counters(one("sq2/**/*counter_collection.csv"))
dur = durations(one("opstats/**/*kernel_trace.csv"))
# units of work per launch (prof_ops.py shapes) and algorithmic bytes per unit
C_oc = None
for l in open(os.path.join(src, "ops.log")):
    if l.startswith("optimal_completion C ="):
        C_oc = int(l.split("=")[1])
shapes = {
    "lev_classify_kernel<8>": dict(what="error_rate / prefix_error_rates, N=4096 T=512: token classes + match masks", units=4096,
                                   alg=8192 + 512 * 8 + 513 * 4, per_unit="utterance (tokens in, look-up tables out)"),
    "lev_bitpar_kernel": dict(what="error_rate / prefix_error_rates, N=4096 T=512: bit-parallel recurrence", units=4096,
                              alg=512 * 8 + 513 * 4 + (4 + 2052) / 2, per_unit="utterance (tables in; mean of the two calls out)",
                              cells=512 * 512),
    "lev_skewed_kernel<false>": dict(what="edit_distance with costs 1/2/3 (cell-by-cell), N=4096 T=512", units=4096, alg=8196,
                                     per_unit="utterance", cells=512 * 512),
    "lev_rowsync_kernel<false, false>": dict(what="optimal_completion masks, N=4096 T=512", units=4096, alg=8192 + 513 * 64,
                                             per_unit="utterance (tokens in, class bitmasks out)", cells=512 * 512),
    "oc_expand_tiles_kernel": dict(what="optimal_completion expansion", units=4096, alg=8 * 513 * (C_oc or 0),
                                   per_unit="utterance ((H+1) x C int64 out)"),
    "ctc_search_kernel<1, 4, true": dict(what="fused CTC search N=4096 T=512 V=256 K=16 (+12 and +6 logits)", units=4096,
                                          alg=4 * 512 * 257 + 8 * 512 * 16 + 12 * 16, per_unit="utterance", frames=512),
    "ctc_search_kernel<3, -1, false": dict(what="fused CTC search C3: N=1024 T=1000 V=1000 K=16", units=1024,
                                            alg=4 * 1000 * 1001 + 8 * 1000 * 16 + 12 * 16, per_unit="utterance", frames=1000),
    "spec_augment_rows_kernel": dict(what="spec_augment_apply C4: 2048 x 1000 x 80", units=2048, alg=640000, per_unit="utterance"),
    "image_warp_kernel<false>": dict(what="sparse_image_warp C4: (2048,1,1000,80)", units=2048, alg=640000, per_unit="image"),
}
records = {}
for key, meta in shapes.items():
    names = [k for k in dur if key in k]
    if not names:
        continue
    k = names[0]
    d = dur[k]
    rec = dict(meta)
    rec["kernel"] = k.split("(")[0]
    split = key.startswith("ctc_search_kernel<1")  # first half of the dispatches: +12 logits, second: +6
    def part(x):
        x = list(x)
        return x[: len(x) // 2] if split else x
    rec["launches"] = len(part(d))
    med = lambda x: sorted(x)[len(x) // 2]  # noqa: E731  (the first launch of a process runs cold)
    rec["avg_us"] = med(part(d))
    rec["all_us"] = [round(x, 1) for x in d]
    if split:
        rec["avg_us_flat_logits"] = med(list(d)[len(d) // 2:])
    f_kb, w_kb = mean(part(fetch[k]["FETCH_SIZE"])), mean(part(write[k]["WRITE_SIZE"]))
    rec["FETCH_SIZE_KB_raw"], rec["WRITE_SIZE_KB_raw"] = f_kb, w_kb
    rec["hbm_bytes_per_launch"] = f_kb * 1024 * fetch_scale + w_kb * 1024 * write_scale
    rec["algorithmic_bytes_per_launch"] = meta["alg"] * meta["units"]
    rec["hbm_over_algorithmic"] = rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    rec["achieved_GBs_algorithmic"] = rec["algorithmic_bytes_per_launch"] / rec["avg_us"] / 1e3
    rec["roofline_frac_of_8TBs"] = rec["achieved_GBs_algorithmic"] / 8000.0
    c = {n: mean(part(v)) for n, v in list(sq1[k].items()) + list(sq2[k].items())}
    rec["sq"] = c
    per = meta["units"] * meta.get("frames", 1)
    rec["per_unit"] = {"unit": ("frame and utterance" if "frames" in meta else meta["per_unit"]),
                       "VALU": c["SQ_INSTS_VALU"] / per, "SALU": c["SQ_INSTS_SALU"] / per,
                       "LDS": c["SQ_INSTS_LDS"] / per}
    if "cells" in meta:
        rec["per_unit"]["VALU_per_DP_cell"] = c["SQ_INSTS_VALU"] * 64 / (meta["units"] * meta["cells"])
        rec["cell_updates_per_s"] = meta["units"] * meta["cells"] / (rec["avg_us"] * 1e-6)
    # share of the chip's VALU issue capacity (measured: ns per wave instruction per SIMD at 8 waves)
    rec["valu_pipe_busy_frac"] = c["SQ_INSTS_VALU"] * ns_per_inst * 1e-3 / 1024 / rec["avg_us"]
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        rec["wave_time_split"] = {"issuing": c["SQ_ACTIVE_INST_ANY"] / wc, "issue_stalled": c["SQ_WAIT_INST_ANY"] / wc,
                                  "waiting_s_waitcnt_or_sleep": c["SQ_WAIT_ANY"] / wc}
    records[key] = rec
json.dump({"valu_issue_ns_per_wave_inst_per_simd": ns_per_inst, "kernels": records},
          open(os.path.join(here, "r02_kernels.json"), "w"), indent=1)

ctc = records["ctc_search_kernel<1, 4, true"]
json.dump({
    "kernel": ctc["kernel"],
    "config": {"N": 4096, "T": 512, "V": 256, "beam": 16},
    "hbm_bytes_per_launch": ctc["hbm_bytes_per_launch"],
    "raw": {"FETCH_SIZE_KB": ctc["FETCH_SIZE_KB_raw"], "WRITE_SIZE_KB": ctc["WRITE_SIZE_KB_raw"]},
    "correction": "2 x FETCH_SIZE + 1 x WRITE_SIZE (profiles/r02_traffic_calibration.json)",
    "sq": {"SQ_INSTS_VALU_per_launch": ctc["sq"]["SQ_INSTS_VALU"], "SQ_INSTS_SALU_per_launch": ctc["sq"]["SQ_INSTS_SALU"],
           "SQ_INSTS_LDS_per_launch": ctc["sq"]["SQ_INSTS_LDS"],
           "valu_issue_cycles_per_inst_measured": ns_per_inst * 2.4,
           "note": "cycles at the 2.4 GHz bench.py prices against = measured ns per wave64 VALU instruction per SIMD "
                   "with 8 waves resident (profiles/r02_valu_issue.json) x 2.4"},
}, open(os.path.join(here, "r02_ctc_traffic.json"), "w"), indent=1)
for k, r in records.items():
    print("%-34s %8.1f us  hbm %6.2f GB (x%.2f of algorithmic)  %5.1f %% of 8 TB/s  VALU busy %4.1f %%  VALU/unit %.1f" % (
        k, r["avg_us"], r["hbm_bytes_per_launch"] / 1e9, r["hbm_over_algorithmic"], 100 * r["roofline_frac_of_8TBs"],
        100 * r["valu_pipe_busy_frac"], r["per_unit"]["VALU"]))
