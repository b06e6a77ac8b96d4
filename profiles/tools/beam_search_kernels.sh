cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/bs_k
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/bs_k -o x --output-format csv -- python3 profiles/tools/time_beam_search.py > gpurun_out/bs_k.log 2>&1 || exit 1
f=$(find gpurun_out/bs_k -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f" | cut -d, -f1-5 | cut -c1-150
