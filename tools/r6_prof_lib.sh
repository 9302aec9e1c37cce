#!/bin/bash
# per-kernel statistics of the train step, this library against another build of it:  bash tools/r6_prof_lib.sh <tag> <other.so> [cfg]
export TMPDIR=/tmp
o=gpurun_out/${1:-r6_pl}; other=${2:-composer_amd/lib/r5_baseline.so}; cfg=${3:-c2}
mkdir -p $o
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/new_$cfg -o k -- python3 tools/train_only.py $cfg 12 > $o/new_$cfg.log 2>&1
COMPOSER_HIP_LIB=$other timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/old_$cfg -o k -- python3 tools/train_only.py $cfg 12 > $o/old_$cfg.log 2>&1
python3 tools/kstats.py $o/new_$cfg 14 | tee $o/new_${cfg}_k.txt
python3 tools/kstats.py $o/old_$cfg 14 | tee $o/old_${cfg}_k.txt
find gpurun_out -name "*.db" -delete
find gpurun_out -name "k_kernel_trace.csv" -delete
