#!/bin/bash
# per-kernel statistics of the C2 train step, LayerNorm-fused training path (COMPOSER_LN_FUSED=2) against the default:  bash tools/r6_prof_train.sh <tag> [cfg]
export TMPDIR=/tmp
o=gpurun_out/${1:-r6_pt}; cfg=${2:-c2}
mkdir -p $o
COMPOSER_LN_FUSED=${FUSED_MODE:-2} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/fused_$cfg -o k -- python3 tools/train_only.py $cfg 12 > $o/fused_$cfg.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/unfused_$cfg -o k -- python3 tools/train_only.py $cfg 12 > $o/unfused_$cfg.log 2>&1
tail -n 1 $o/fused_$cfg.log $o/unfused_$cfg.log
python3 tools/kstats.py $o/fused_$cfg 24 | tee $o/fused_${cfg}_k.txt
python3 tools/kstats.py $o/unfused_$cfg 24 | tee $o/unfused_${cfg}_k.txt
find gpurun_out -name "*.db" -delete
find gpurun_out -name "k_kernel_trace.csv" -delete
