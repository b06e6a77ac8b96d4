"""Summary of profiles/tools/pmc_passes.sh output: python profiles/tools/pmc_summary.py gpurun_out/<name> <kernel-needle> [out.json]
Per launch of the kernels whose name contains the needle: average duration (kernel-trace stats), HBM
bytes (2 x FETCH_SIZE + WRITE_SIZE, KB units: MI355X_MICROARCH.md, HBM), SQ instruction counts and the
wave-time split."""
import csv, glob, json, os, sys
from collections import defaultdict

src, needle = sys.argv[1], sys.argv[2]


def one(pattern):
    m = glob.glob(os.path.join(src, pattern), recursive=True)
    return m[0] if m else None


def counters(path):
    out = defaultdict(list)
    if path is None:
        return {}
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"]:
            out[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


rec = {"needle": needle, "source": "profiles/tools/pmc_passes.sh (separate --pmc passes)"}
st = one("stats/**/*kernel_stats.csv")
if st:
    for r in csv.DictReader(open(st)):
        if needle in r["Name"]:
            rec["kernel"] = r["Name"].split("(")[0]
            rec["calls"] = int(r["Calls"])
            rec["avg_ms"] = float(r["AverageNs"]) / 1e6
c = {}
for p in ("fetch", "write", "sq1", "sq2", "grbm"):
    c.update(counters(one(p + "/**/*counter_collection.csv")))
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    rec["hbm_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    rec["raw_KB"] = {"FETCH_SIZE": c["FETCH_SIZE"], "WRITE_SIZE": c["WRITE_SIZE"]}
for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
          "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"):
    if k in c:
        rec[k] = c[k]
if "SQ_WAVE_CYCLES" in c and "SQ_ACTIVE_INST_ANY" in c:
    w = c["SQ_WAVE_CYCLES"]
    rec["wave_time_split"] = {"issuing": c["SQ_ACTIVE_INST_ANY"] / w, "issue_stalled": c["SQ_WAIT_INST_ANY"] / w,
                              "parked": c["SQ_WAIT_ANY"] / w, "valu": c["SQ_ACTIVE_INST_VALU"] / w,
                              "scalar": c["SQ_ACTIVE_INST_SCA"] / w, "lds": c["SQ_ACTIVE_INST_LDS"] / w}
if "GRBM_GUI_ACTIVE" in c and "avg_ms" in rec:
    rec["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (rec["avg_ms"] * 1e-3) / 1e9
if "GRBM_GUI_ACTIVE" in c and "SQ_ACTIVE_INST_VALU" in c:
    # SQ_ACTIVE_INST_VALU counts, chip-wide, quad-cycles in which a SIMD's vector pipe is executing:
    # x 4 / 1024 SIMDs = busy cycles per SIMD, against the launch's cycles (GRBM_GUI_ACTIVE is summed over 8 XCDs)
    rec["valu_pipe_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (c["GRBM_GUI_ACTIVE"] / 8.0)
    rec["valu_cycles_per_inst"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / c["SQ_INSTS_VALU"] if c.get("SQ_INSTS_VALU") else None
print(json.dumps(rec, indent=1))
if len(sys.argv) > 3:
    json.dump(rec, open(sys.argv[3], "w"), indent=1)
