#!/bin/bash
# Round-5 profiles, on a GPU box (from the repo root through gpurun):
#   gpurun --timeout 1200 -- 'bash profiles/collect_r05.sh'
# then, back in the build container:  python profiles/summarize_r05.py gpurun_out
# Kernel-trace statistics of the bench command (timed step only, and the whole line with every other
# config); then counter passes (one counter group per pass, --kernel-trace only, as
# MI355X_MICROARCH.md prescribes) over the dominant kernel of every BASELINE config that has one of
# its own: the headline CTC search (C2 shape), the C5 shard decode, the C3 search, C3 with the bigram
# model in the loop, C4 sparse_image_warp and SpecAugment's one-launch application, and the two step functions (C3).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
rocprofv3 --kernel-trace --stats -d gpurun_out/r05/bench_stats -o bench --output-format csv -- \
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/r05/bench_line.json 2> gpurun_out/r05/bench.err
echo "bench stats done"
rocprofv3 --kernel-trace --stats -d gpurun_out/r05/bench_full_stats -o bench --output-format csv -- \
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r05/bench_full_line.json 2> gpurun_out/r05/bench_full.err
echo "bench full stats done"
bash profiles/tools/pmc_passes.sh r05/pmc_headline profiles/prof_ctc.py
bash profiles/tools/pmc_passes.sh r05/pmc_c5 profiles/prof_ctc_shape.py 512 4096 5000
bash profiles/tools/pmc_passes.sh r05/pmc_c3 profiles/prof_ctc_shape.py 1000 1024 1000
bash profiles/tools/pmc_passes.sh r05/pmc_lm profiles/prof_lm_table.py
bash profiles/tools/pmc_passes.sh r05/pmc_warp profiles/tools/time_warp.py
bash profiles/tools/pmc_passes.sh r05/pmc_beam_advance profiles/prof_steps.py beam
bash profiles/tools/pmc_passes.sh r05/pmc_ctc_advance profiles/prof_steps.py ctc
bash profiles/tools/pmc_passes.sh r05/pmc_spec profiles/tools/time_spec.py
echo "all passes done"
