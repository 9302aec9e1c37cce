#!/bin/bash
# per-kernel statistics of the C2 / C4 inference forward, fused against unfused block path:  bash tools/r5_prof_fwd.sh <tag> [c2|c4]
export TMPDIR=/tmp
o=gpurun_out/${1:-r5_p1}; cfg=${2:-c2}
mkdir -p $o
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/fused_$cfg -o k -- python3 tools/fwd_only.py $cfg 10 > $o/fused_$cfg.log 2>&1
COMPOSER_LN_FUSED=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/unfused_$cfg -o k -- python3 tools/fwd_only.py $cfg 10 > $o/unfused_$cfg.log 2>&1
tail -n 2 $o/fused_$cfg.log $o/unfused_$cfg.log
python3 tools/kstats.py $o/fused_$cfg 16 | tee $o/fused_${cfg}_k.txt
python3 tools/kstats.py $o/unfused_$cfg 16 | tee $o/unfused_${cfg}_k.txt
find gpurun_out -name "*.db" -delete
find gpurun_out -name "k_kernel_trace.csv" -delete
