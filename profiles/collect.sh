#!/bin/bash
# Re-collects what profiles/r02_* is made of, on a GPU box (run from the repo root through gpurun):
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh'
# then, back in the build container:  python profiles/summarize.py gpurun_out/prof
# Kernel-trace statistics of the default bench command; the VALU issue microbenchmark; then the
# PMC passes (one counter group per pass, --kernel-trace only, as MI355X_MICROARCH.md prescribes)
# over profiles/prof_ops.py and over the streaming kernels of known size that calibrate
# FETCH_SIZE / WRITE_SIZE on this image.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof
rm -rf $out && mkdir -p $out
MB=profiles/microbench/microbench
$MB issue > $out/valu_issue.json
$MB traffic > $out/traffic_rates.json
$MB stores > $out/store_patterns.json
$MB tiles > $out/store_tiles.json
rocprofv3 --kernel-trace --stats -d $out/stats -o bench --output-format csv -- \
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $out/bench_line.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/opstats -o ops --output-format csv -- python3 profiles/prof_ops.py > $out/ops.log 2>&1
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $out/$name -o p --output-format csv -- python3 profiles/prof_ops.py > $out/$name.log 2>&1
  if [ "$name" = fetch ] || [ "$name" = write ]; then
    rocprofv3 --pmc "$@" --kernel-trace -d $out/${name}_cal -o p --output-format csv -- $MB traffic > $out/${name}_cal.log 2>&1
  fi
  echo "pass $name done"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
find $out -name "*.csv" | head -40
