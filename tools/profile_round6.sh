#!/bin/bash
# Round-6 evidence pass on the GPU box: ONE pass at the last library commit (one pass; same-box A/Bs are tools/ab_step.py).  Everything under gpurun_out/<tag>/;
# tools/collect_profiles5.sh copies what is judged into profiles/.  Every profiler command runs under `timeout`, python3 directly
# after `--`; PMC passes carry --kernel-trace only.
#   bash tools/profile_round6.sh <tag>
tag=${1:-r6}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-decode --no-extras"
# 0. the GPU suite, its summary line written by pytest itself (the r4_10..13 tails held library banners only)
timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider --junitxml=$out/pytest_gpu.xml > $out/pytest_gpu.log 2>&1
python3 - "$out" <<'PY'
import sys, xml.etree.ElementTree as ET
o = sys.argv[1]
r = ET.parse(o + "/pytest_gpu.xml").getroot()
s = r if r.tag == "testsuite" else r.find("testsuite")
line = "pytest -m gpu: tests=%s failures=%s errors=%s skipped=%s time=%ss" % tuple(s.get(k) for k in ("tests", "failures", "errors", "skipped", "time"))
open(o + "/pytest_gpu_tail.txt", "w").write(line + "\n")
print(line)
PY
# 1. kernel-trace statistics of the bench command (C2 at B=128, C2 at B=32, C4)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2 -o k -- $BENCH --steps 10 --warmup 3 > $out/stats_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2b32 -o k -- $BENCH --batch 32 --steps 10 --warmup 3 > $out/stats_c2b32.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c4 -o k -- $BENCH --config c4 --steps 6 --warmup 2 > $out/stats_c4.log 2>&1
# 2. HBM-side traffic of every kernel class INSIDE the step: FETCH_SIZE and WRITE_SIZE in separate passes over bench.py itself
for cfg in "c2 128 131072" "c2 32 32768" "c4 32 65536"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$1_$2_$c -o k -- $BENCH --config $1 --batch $2 --steps 2 --warmup 1 > $out/pmc_$1_$2_$c.log 2>&1
  done
  python3 tools/make_traffic_json.py $1_tokens$3 $out/pmc_$1_$2_FETCH_SIZE $out/pmc_$1_$2_WRITE_SIZE $out/hbm_traffic.json
done
# ... into profiles/ BEFORE the bench line below is taken, so that line carries the traffic of THIS library build (bench.py prints a
# run's traffic only while its build key is the loaded library's)
cp $out/hbm_traffic.json profiles/hbm_traffic.json
# 3. the default bench line (driver form)
timeout 900 python3 bench.py > $out/bench_c2.json 2> $out/bench_c2.err
# 4. MFMA-pipe busy cycles per kernel at C4 (north_star: "rocprof MFMA util"): SQ counters, their own run
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_c4 -o k -- $BENCH --config c4 --steps 3 --warmup 1 > $out/pmc_c4.log 2>&1
python3 tools/pmc_summary.py $out/pmc_c4 > $out/pmc_c4_sq_summary.txt
# 5. the inference forward, fused against unfused block path, per kernel (C2, C4)
for cfg in c2 c4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fwd_fused_$cfg -o k -- python3 tools/fwd_only.py $cfg 10 > $out/fwd_fused_$cfg.log 2>&1
  COMPOSER_LN_FUSED=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/fwd_unfused_$cfg -o k -- python3 tools/fwd_only.py $cfg 10 > $out/fwd_unfused_$cfg.log 2>&1
  { echo "## $cfg inference forward, LayerNorm-fused block path (12 passes)"; python3 tools/kstats.py $out/fwd_fused_$cfg 12
    echo; echo "## $cfg inference forward, COMPOSER_LN_FUSED=0 (12 passes)"; python3 tools/kstats.py $out/fwd_unfused_$cfg 12; } > $out/fwd_kernels_$cfg.txt
done
{ python3 tools/fwd_bench.py; echo "COMPOSER_LN_FUSED=0:"; COMPOSER_LN_FUSED=0 python3 tools/fwd_bench.py; } 2>/dev/null > $out/forward_only.txt
# 6. kernel micro-benchmarks, decode, the CLI train loop, the reference's default configuration
KB_B=128 timeout 200 python3 tools/kbench.py gemm > $out/kbench_gemm.txt 2>&1
KB_B=128 timeout 200 python3 tools/kbench.py attn ln > $out/kbench_attn_ln.txt 2>&1
timeout 300 python3 tools/decode_bench.py > $out/decode_bench.txt 2>&1
timeout 900 python3 tools/train_cli_bench.py > $out/train_cli.txt 2>&1
$BENCH --steps 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py (same box): %.2f ms/step' % d['ms_per_step'])" >> $out/train_cli.txt
timeout 300 python3 tools/default_config_probe.py 2>/dev/null > $out/default_config.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/def -o k -- python3 tools/default_config_steps.py > $out/def.log 2>&1
{ echo; echo "## rocprofv3 --kernel-trace --stats over 30 train steps (tools/default_config_steps.py)"; python3 tools/kstats.py $out/def 22; } >> $out/default_config.txt
# 7. round 6: the train step per kernel, weight gradients on raw LayerNorm rows (COMPOSER_LN_FUSED=3) against the default; bench.py under the
#    launcher with one rank (ranks / runtime objects, RCCL diagnostics on stderr) and the gradient-exchange pattern alone
COMPOSER_LN_FUSED=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_raw -o k -- python3 tools/train_only.py c2 12 > $out/train_raw.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_default -o k -- python3 tools/train_only.py c2 12 > $out/train_default.log 2>&1
{ echo "## C2 train step, COMPOSER_LN_FUSED=3 (15 steps)"; grep "train:" $out/train_raw.log; python3 tools/kstats.py $out/train_raw 24
  echo; echo "## C2 train step, default (15 steps)"; grep "train:" $out/train_default.log; python3 tools/kstats.py $out/train_default 24; } > $out/train_kernels_c2.txt
AB_ENV_1=COMPOSER_LN_FUSED=3 timeout 600 python3 tools/ab_step.py composer_amd/lib/libcomposer_hip.so composer_amd/lib/libcomposer_hip.so --rounds 3 --cfg c2,c2b32,c4 > $out/ab_ln_raw.txt 2>&1
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-decode > $out/bench_dp1.json 2> $out/bench_dp1.err
timeout 300 python3 bench.py --allreduce-only --steps 20 > $out/allreduce_only.json 2> $out/allreduce_only.err
find $out -name "*.db" -delete
find $out -name "k_kernel_trace.csv" -delete
find $out -name "k_counter_collection.csv" -delete
ls $out | head -80
