#!/bin/bash
# inference forward of C2 (or $1) with the memory-policy experiment builds, twice each, ALT = tile-walk alternation mode
cfg=${1:-c2}
for rep in 1 2; do
for lib in libcomposer_hip x_r x_ra x_raq x_raqg; do
  for a in 0 1; do
    echo -n "$lib ALT=$a: "; COMPOSER_HIP_LIB=composer_amd/lib/$lib.so COMPOSER_GEMM_ALT=$a python tools/fwd_only.py $cfg 20 2>/dev/null | tail -1
  done
done
done
echo -n "unfused: "; COMPOSER_LN_FUSED=0 python tools/fwd_only.py $cfg 20 2>/dev/null | tail -1
