#!/bin/bash
# the same script under several library variants (csrc/build/variants/<name>/lib.so), same box:
#   bash profiles/tools/ab_lib.sh <script.py> <variant> [<variant> ...]     ("base" = the shipped library)
s=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then echo "base: $(python $s 2>&1 | tail -1)";
  else echo "$v: $(PDT_AMD_LIB=pydrobert-pytorch_amd/csrc/build/variants/$v/lib.so python $s 2>&1 | tail -1)"; fi
done
