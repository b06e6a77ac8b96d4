#!/bin/bash
# kernel-trace stats of optimal completion (mask + expansion kernels), bit-parallel mask kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_oc; rm -rf $out; mkdir -p $out
PDT_OC_BITPAR=${1:-1} rocprofv3 --kernel-trace --stats -d $out -o oc --output-format csv -- python3 profiles/tools/time_oc.py > $out/run.log 2>&1
tail -1 $out/run.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | head -8 | cut -c1-160
