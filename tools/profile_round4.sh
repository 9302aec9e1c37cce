#!/bin/bash
# Round-4 evidence pass on the GPU box (everything under gpurun_out/<tag>/; tools/collect_profiles4.sh copies what is judged
# into profiles/).  Every profiler command runs under `timeout`, python3 directly after `--`; PMC passes carry --kernel-trace only.
#   bash tools/profile_round4.sh <tag> [quick]
tag=${1:-r4}; quick=$2
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-decode --no-extras"
# kernel-trace statistics of the bench command (C2 at B=128, C2 at B=32, C4)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2 -o k -- $BENCH --steps 10 --warmup 3 > $out/stats_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2b32 -o k -- $BENCH --batch 32 --steps 10 --warmup 3 > $out/stats_c2b32.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c4 -o k -- $BENCH --config c4 --steps 6 --warmup 2 > $out/stats_c4.log 2>&1
# HBM-side traffic of every kernel class INSIDE the step: FETCH_SIZE and WRITE_SIZE in separate passes over bench.py itself
for cfg in "c2 128 131072" "c2 32 32768" "c4 32 65536"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$1_$2_$c -o k -- $BENCH --config $1 --batch $2 --steps 2 --warmup 1 > $out/pmc_$1_$2_$c.log 2>&1
  done
  python3 tools/make_traffic_json.py $1_tokens$3 $out/pmc_$1_$2_FETCH_SIZE $out/pmc_$1_$2_WRITE_SIZE $out/hbm_traffic.json
done
if [ -z "$quick" ]; then
  # MFMA-pipe busy cycles per kernel at C4 (north_star: "rocprof MFMA util"): SQ counters, their own run
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_c4 -o k -- $BENCH --config c4 --steps 3 --warmup 1 > $out/pmc_c4.log 2>&1
  python3 tools/pmc_summary.py $out/pmc_c4 > $out/pmc_c4_sq_summary.txt
  KB_B=128 timeout 200 python tools/kbench.py gemm > $out/kbench_gemm.txt 2>&1
  KB_B=128 timeout 200 python tools/kbench.py attn ln > $out/kbench_attn_ln.txt 2>&1
  timeout 300 python tools/decode_bench.py > $out/decode_bench.txt 2>&1
  DECODE_EAGER_ONLY=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_decode -o k -- python3 tools/decode_bench.py > $out/trace_decode.log 2>&1
  python3 tools/decode_trace.py $out/trace_decode > $out/decode_trace.txt
fi
find $out -name "*.db" -delete
rm -f $out/stats_*/k_kernel_trace.csv $out/pmc_*/k_kernel_trace.csv $out/pmc_*/k_counter_collection.csv $out/trace_decode/k_kernel_trace.csv
ls -R $out | head -80
