#!/bin/bash
# Counter passes over one python script, one counter group per pass (--kernel-trace only, as
# MI355X_MICROARCH.md prescribes):  bash profiles/tools/pmc_passes.sh <out-name> <script.py> [args...]
# Output under gpurun_out/<out-name>/{stats,fetch,write,sq1,sq2}; summarise with profiles/tools/pmc_summary.py.
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$name
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 "$@" > $out/stats.log 2>&1
echo "stats done"
pass() {
  local p=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $out/$p -o p --output-format csv -- python3 "${ARGS[@]}" > $out/$p.log 2>&1
  echo "pass $p done"
}
ARGS=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
pass grbm GRBM_GUI_ACTIVE
