#!/bin/bash
# optimal completion: row-synchronous mask kernel against the bit-parallel one (and build variants
# given as arguments: names under csrc/build/variants), same box
set -e
echo "PDT_OC_BITPAR=0"; PDT_OC_BITPAR=0 python profiles/tools/time_oc.py
echo "PDT_OC_BITPAR=1"; PDT_OC_BITPAR=1 python profiles/tools/time_oc.py
for v in "$@"; do
  echo "variant $v"; PDT_AMD_LIB=pydrobert-pytorch_amd/csrc/build/variants/$v/lib.so python profiles/tools/time_oc.py
done
