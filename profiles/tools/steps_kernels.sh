#!/bin/bash
# Kernel durations of the two step functions at the C3 shape (rocprofv3 --kernel-trace --stats over
# profiles/tools/steps_only.py), then the calls with their host glue (profiles/tools/time_steps.py).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/steps_k
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/steps_k -o x --output-format csv -- python3 profiles/tools/steps_only.py > gpurun_out/steps_k.log 2>&1 || exit 1
f=$(find gpurun_out/steps_k -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -h "advance" "$f" | cut -d, -f1-5
timeout -k 10 240 python3 profiles/tools/time_steps.py 2>&1 | grep -v amdgpu.ids
