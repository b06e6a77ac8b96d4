#!/bin/bash
# kernel-trace stats of any script: bash profiles/tools/prof_script.sh <script.py> <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$2; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python3 $1 > $out/run.log 2>&1
tail -1 $out/run.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | head -${3:-10} | cut -c1-170
