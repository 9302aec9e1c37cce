#!/bin/bash
# One GPU-box pass that regenerates the evidence under gpurun_out/<tag>/ (copy what is judged into profiles/):
#   gpu test log, default bench line, rocprofv3 kernel stats of the bench command, HBM traffic of the GEMM kernels from
#   PMC (FETCH_SIZE and WRITE_SIZE in SEPARATE passes: together they exceed the TCC counter slots and rocprofv3 aborts
#   and then hangs -- every profiler command runs under `timeout`), kernel micro-benchmarks.
# usage (on the box, from the repo root):  bash tools/profile_round.sh r1_05
tag=${1:-round}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; grep -E "passed|failed" $out/pytest_gpu.log | tail -1
timeout 600 python bench.py > $out/bench_c2.json 2> $out/bench_c2.err; cut -c1-200 $out/bench_c2.json
timeout 600 python bench.py --batch 32 --no-cpu-baseline --no-decode > $out/bench_c2_b32.json 2>> $out/bench_c2.err; cut -c1-200 $out/bench_c2_b32.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 bench.py --no-cpu-baseline --no-decode --steps 10 --warmup 3 > $out/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  KB_B=128 timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o k -- python3 tools/kbench.py gemmfwd > $out/pmc_$c.log 2>&1
  python3 tools/pmc_summary.py $out/pmc_$c gemm > $out/pmc_${c}_summary.txt
done
KB_B=128 timeout 200 python tools/kbench.py gemm > $out/kbench_gemm.txt 2>&1
KB_B=128 timeout 200 python tools/kbench.py attn ln > $out/kbench_attn_ln.txt 2>&1
find $out -name "*.db" -delete
rm -f $out/stats/k_kernel_trace.csv $out/pmc_*/k_kernel_trace.csv $out/pmc_*/k_counter_collection.csv
ls -R $out | head -30
