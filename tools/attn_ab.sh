#!/bin/bash
# forward-attention A/B on one box: tools/attn_ab.sh <tag> <lib> [<lib> ...]   (lib = file under composer_amd/lib without .so)
tag=$1; shift
mkdir -p gpurun_out/$tag
for r in 1 2; do
for l in "$@"; do
  echo "== $l" | tee -a gpurun_out/$tag/attn_ab.txt
  KB_B=128 COMPOSER_HIP_LIB=composer_amd/lib/$l.so timeout 120 python tools/kbench.py attnfwd 2>&1 | grep "attn fwd" | tee -a gpurun_out/$tag/attn_ab.txt
done
done
