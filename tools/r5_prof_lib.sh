#!/bin/bash
# per-kernel statistics of the inference forward with another build of the library:  bash tools/r5_prof_lib.sh <tag> <lib.so> [c2|c4]
export TMPDIR=/tmp
o=gpurun_out/${1:-r5_pl}; lib=$2; cfg=${3:-c2}
mkdir -p $o
COMPOSER_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/lib_$cfg -o k -- python3 tools/fwd_only.py $cfg 10 > $o/lib_$cfg.log 2>&1
tail -n 1 $o/lib_$cfg.log
python3 tools/kstats.py $o/lib_$cfg 12 | tee $o/lib_${cfg}_k.txt
find gpurun_out -name "*.db" -delete
find gpurun_out -name "k_kernel_trace.csv" -delete
