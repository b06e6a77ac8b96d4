#!/bin/bash
# SQ counters of the kernels whose name contains <pattern>, for any script:
#   bash profiles/tools/pmc_kernel.sh <pattern> <script.py> [tag]
set -e
pat=$1; script=$2; tag=${3:-k}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
  --kernel-trace -d $out/sq1 -o p --output-format csv -- python3 $script > $out/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM \
  --kernel-trace -d $out/sq2 -o p --output-format csv -- python3 $script > $out/sq2.log 2>&1
python3 - "$out" "$pat" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
for p in ("sq1", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat not in r["Kernel_Name"]:
                continue
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(r["Kernel_Name"][:70], r["Counter_Name"])] += 1
    for kn, d in acc.items():
        print(kn)
        for c, v in sorted(d.items()):
            print("   %-24s %16.0f per launch" % (c, v / cnt[(kn, c)]))
PY
