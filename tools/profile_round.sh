#!/bin/bash
# One GPU-box pass that regenerates the evidence under gpurun_out/<tag>/ (copy what is judged into profiles/):
#   gpu test log, default bench line, rocprofv3 kernel stats of the bench command, PMC pass for HBM traffic
#   (FETCH_SIZE / WRITE_SIZE in their own run: gpurun refuses --pmc together with hip/sys traces), kernel micro-benchmarks.
# usage (on the box, from the repo root):  bash tools/profile_round.sh r1_04
tag=${1:-round}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -1 $out/pytest_gpu.log
python bench.py > $out/bench_c2.json 2> $out/bench_c2.err; cut -c1-300 $out/bench_c2.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o k -- python3 bench.py --no-cpu-baseline --no-decode --steps 10 --warmup 3 > $out/stats.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $out/pmc_hbm -o k -- python3 bench.py --no-cpu-baseline --no-decode --steps 4 --warmup 2 > $out/pmc_hbm.log 2>&1
KB_B=128 python tools/kbench.py gemm > $out/kbench_gemm.txt 2>&1
KB_B=128 python tools/kbench.py attn ln > $out/kbench_attn_ln.txt 2>&1
find $out -name "*.db" -delete
ls -R $out | head -40
