#!/bin/bash
cfg=${1:-c2}
for rep in 1 2; do
for lib in libcomposer_hip x_g; do
  for n in 1 2 4; do
    echo -n "$lib CHUNKS=$n: "; COMPOSER_HIP_LIB=composer_amd/lib/$lib.so COMPOSER_MLP_CHUNKS=$n python tools/fwd_only.py $cfg 20 2>/dev/null | tail -1
  done
done
done
