#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gru_k
timeout -k 10 300 python3 profiles/tools/gru_loop.py 300 2>&1 | grep -v amdgpu.ids
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/gru_k -o x --output-format csv -- python3 profiles/tools/gru_loop.py 300 > gpurun_out/gru_k.log 2>&1 || exit 1
f=$(find gpurun_out/gru_k -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -32 "$f" | cut -d, -f1-5 | cut -c1-140
rm -f gpurun_out/gru_k/x_kernel_trace.csv
