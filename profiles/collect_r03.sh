#!/bin/bash
# Round-3 profiles, on a GPU box (from the repo root through gpurun):
#   gpurun --timeout 1200 -- 'bash profiles/collect_r03.sh'
# then, back in the build container:  python profiles/summarize_r03.py gpurun_out/prof3
# Kernel-trace statistics of the bench command and of the GRU-LM frames; then PMC passes over the
# CTC search (one counter group per pass, --kernel-trace only, as MI355X_MICROARCH.md prescribes).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof3
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -o bench --output-format csv -- \
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $out/bench_line.json 2> $out/bench.err
echo "bench stats done"
rocprofv3 --kernel-trace --stats -d $out/gru -o gru --output-format csv -- python3 profiles/prof_gru_lm.py > $out/gru.log 2>&1
echo "gru stats done"
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $out/$name -o p --output-format csv -- python3 profiles/prof_ctc.py > $out/$name.log 2>&1
  echo "pass $name done"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
find $out -name "*.csv" | head -30
