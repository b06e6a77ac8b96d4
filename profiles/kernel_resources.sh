#!/bin/bash
# Register / spill / scratch figures of every kernel of one .hip file, read from the gfx950 code
# object's notes (what DESIGN.md quotes):  profiles/kernel_resources.sh ctc_search.hip [-D...]
set -e
src="$1"; shift
here="$(cd "$(dirname "$0")/.." && pwd)"
out="$(mktemp -d)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off "$@" --cuda-device-only -c \
  "$here/pydrobert-pytorch_amd/csrc/$src" -o "$out/k.bundle"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --unbundle \
  --input="$out/k.bundle" --output="$out/k.co"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$out/k.co" | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    print("{:90s} vgpr {:>3} (spill {:>2})  sgpr {:>3} (spill {:>2})  scratch {:>4} B  lds {:>6} B".format(
        name[:90], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
        g("private_segment_fixed_size"), g("group_segment_fixed_size")))
' | (command -v c++filt >/dev/null && c++filt || cat)
rm -rf "$out"
