#!/bin/bash
# kernel statistics of 30 train steps of the reference's default configuration (E=256, 16 heads of 16, 8 blocks, window 1024, batch 1)
export TMPDIR=/tmp
o=gpurun_out/${1:-r5_def}; mkdir -p $o
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/def -o k -- python3 tools/default_config_steps.py > $o/def.log 2>&1
python3 tools/kstats.py $o/def 40 | tee $o/def_k.txt
find gpurun_out -name "*.db" -delete; find gpurun_out -name "k_kernel_trace.csv" -delete
