#!/bin/bash
# one SQ pass over prof_ctc.py under a variant library
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_skipcons; rm -rf $out; mkdir -p $out
export PDT_AMD_LIB=pydrobert-pytorch_amd/csrc/build/variants/skipcons/lib.so
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $out/sq1 -o p --output-format csv -- python3 profiles/prof_ctc.py > $out/sq1.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 profiles/prof_ctc.py > $out/stats.log 2>&1
echo done
