#!/bin/bash
# The fused step kernel inside the GRU-LM loop (profiles/tools/gru_loop.py) under rocprofv3, per library variant
# (VARIANTS="default adv_ph1 ..."; profiles/tools/build_var.sh): the kernel's own average duration.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then unset PDT_AMD_LIB; else export PDT_AMD_LIB=$PWD/pydrobert-pytorch_amd/csrc/build/variants/$v/lib.so; fi
  rm -rf gpurun_out/gru_k
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/gru_k -o x --output-format csv -- python3 profiles/tools/gru_loop.py 200 > gpurun_out/gru_k.log 2>&1 || exit 1
  echo "== $v"; grep -h "T=200" gpurun_out/gru_k.log | tail -1
  f=$(find gpurun_out/gru_k -name "*kernel_stats.csv" | head -1)
  grep -h "ctc_advance\|fusion_ext" "$f" | cut -d, -f1-5
done
rm -rf gpurun_out/gru_k
