#!/bin/bash
# attention forward / backward with the pair's light query block first (lib lightfirst.so) against the product order: times, then
# HBM-side traffic and L2 hit counts of both (separate PMC passes)
for rep in 1 2; do
for l in libcomposer_hip lightfirst; do echo "== $l"; KB_B=128 COMPOSER_HIP_LIB=composer_amd/lib/$l.so python tools/kbench.py attn 2>&1 | grep "attn"; done
done
for l in libcomposer_hip lightfirst; do echo "== PMC $l"; bash tools/pmc_tcc.sh r5_attn_$l attn attn_ $l 2>&1 | grep -v "^W2026" ; done
