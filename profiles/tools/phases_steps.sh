#!/bin/bash
# Kernel time of the two step functions with the kernels cut short after each phase (variant builds
# adv_ph1 / adv_ph2 of beam_advance.hip: -DPDT_ADV_PHASES=1 / 2; profiles/tools/build_var.sh).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$PWD/pydrobert-pytorch_amd/csrc/build/variants
for v in ${PHASES:-full adv_ph1 adv_ph2}; do
  if [ $v = full ]; then unset PDT_AMD_LIB; else export PDT_AMD_LIB=$V/$v/lib.so; fi
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/ph_$v -o x --output-format csv -- python3 profiles/tools/steps_only.py > gpurun_out/ph_$v.log 2>&1 || exit 1
  echo "== $v"
  f=$(find gpurun_out/ph_$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -h "advance" "$f" | cut -d, -f1-5
done
