#!/bin/bash
# Re-collects every file under profiles/ on a GPU box (run from the repo root through gpurun):
#   gpurun --timeout 900 -- 'bash profiles/collect.sh'
# Kernel-trace statistics of the default bench command, then the PMC passes for the dominant
# kernel (separate passes per counter group, as MI355X_MICROARCH.md prescribes).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -o bench --output-format csv -- \
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o f --output-format csv -- python3 profiles/prof_ctc.py > $out/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o w --output-format csv -- python3 profiles/prof_ctc.py > $out/w.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES --kernel-trace \
  -d $out/sq1 -o s --output-format csv -- python3 profiles/prof_ctc.py > $out/s1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace \
  -d $out/sq2 -o s --output-format csv -- python3 profiles/prof_ctc.py > $out/s2.log 2>&1
ls -R $out | head -40
