#!/bin/bash
# Round-2 evidence pass on the GPU box (everything under gpurun_out/<tag>/; copy what is judged into profiles/).
# Every profiler command runs under `timeout`; PMC passes never share a run with tracing other than --kernel-trace.
tag=${1:-r2}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; grep -E "passed|failed" $out/pytest_gpu.log | tail -1
timeout 600 python bench.py > $out/bench_c2.json 2> $out/bench_c2.err; cut -c1-200 $out/bench_c2.json
timeout 600 python bench.py --config c4 --no-decode > $out/bench_c4.json 2> $out/bench_c4.err; cut -c1-200 $out/bench_c4.json
timeout 600 python bench.py --batch 32 --no-cpu-baseline --no-decode --no-extras > $out/bench_c2_b32.json 2>> $out/bench_c2.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2 -o k -- python3 bench.py --no-cpu-baseline --no-decode --no-extras --steps 10 --warmup 3 > $out/stats_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c4 -o k -- python3 bench.py --config c4 --no-cpu-baseline --no-decode --no-extras --steps 6 --warmup 2 > $out/stats_c4.log 2>&1
# MFMA-pipe busy cycles per kernel at C4 (north_star: "rocprof MFMA util"): SQ counters, their own run
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_c4 -o k -- python3 bench.py --config c4 --no-cpu-baseline --no-decode --no-extras --steps 3 --warmup 1 > $out/pmc_c4.log 2>&1
python3 tools/pmc_summary.py $out/pmc_c4 > $out/pmc_c4_sq_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  KB_B=128 timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o k -- python3 tools/kbench.py gemmfwd > $out/pmc_$c.log 2>&1
  python3 tools/pmc_summary.py $out/pmc_$c gemm > $out/pmc_${c}_summary.txt
done
KB_B=128 timeout 200 python tools/kbench.py gemm > $out/kbench_gemm.txt 2>&1
KB_B=128 timeout 200 python tools/kbench.py attn ln > $out/kbench_attn_ln.txt 2>&1
timeout 300 python tools/fwd_bench.py > $out/forward_only.txt 2>&1
timeout 300 python tools/decode_bench.py > $out/decode_bench.txt 2>&1
COMPOSER_DECODE_V1=1 timeout 300 python tools/decode_bench.py > $out/decode_bench_v1.txt 2>&1
timeout 300 python tools/decode_diag.py > $out/decode_diag.txt 2>&1
timeout 300 ./tools/ubench/bin/graph_chain > $out/graph_chain.txt 2>&1
timeout 900 python tools/train_cli_bench.py > $out/train_cli.txt 2>&1; cat $out/train_cli.txt
# persistent-GEMM CU cap under a (1-rank) communicator: its 1-GPU cost
for cus in 0 248 240 224; do
  COMPOSER_DP_GEMM_CUS=$cus timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline --no-decode --no-extras --steps 20 > $out/bench_dp1_cus$cus.json 2> $out/bench_dp1_cus$cus.err
  python3 -c "import json,sys; d=json.loads(open('$out/bench_dp1_cus$cus.json').read().strip().split('\n')[-1]); print('DP 1-rank, GEMM CUs $cus:', round(d['ms_per_step'],2), 'ms/step', round(d['value']/1e6,3), 'M tok/s')" | tee -a $out/dp_cu_cap.txt
done
find $out -name "*.db" -delete
rm -f $out/stats_*/k_kernel_trace.csv $out/pmc_*/k_kernel_trace.csv $out/pmc_*/k_counter_collection.csv
ls -R $out | head -60
