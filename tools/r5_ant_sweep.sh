#!/bin/bash
for rep in 1 2; do
for cfg in c4 c2; do
for lib in libcomposer_hip ant0 ant1 ant2; do
  echo -n "$cfg $lib: "; COMPOSER_HIP_LIB=composer_amd/lib/$lib.so python tools/fwd_only.py $cfg 20 2>/dev/null | tail -1
done
done
done
