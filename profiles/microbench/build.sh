#!/bin/bash
# hipcc --offload-arch=gfx950 -> profiles/microbench/microbench (travels to the GPU box; git-ignored)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 microbench.hip -o microbench
