#!/bin/bash
# Round 3: static striding vs per-XCD item counters in the persistent GEMMs (same box, alternating): plain, beside a CU hog on
# the communication stream (32 / 8 workgroups for 3 ms of every step: what an RCCL kernel does to the chip), and under a
# 1-rank communicator with and without the CU cap.
out=gpurun_out/${1:-r3b}
mkdir -p $out
b() { python bench.py --no-cpu-baseline --no-decode --no-extras --steps 20 --warmup 5 "$@" 2>>$out/err.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['ms_per_step'],3))"; }
dp() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline --no-decode --no-extras --steps 20 --warmup 5 2>>$out/err.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['ms_per_step'],3))"; }
{
for i in 1 2; do
  echo "plain          dynamic $(COMPOSER_GEMM_ITEMS=dynamic b)   static $(COMPOSER_GEMM_ITEMS=static b)"
done
for h in 32,3000 8,3000 32,10000 64,20000; do
  echo "hog $h   dynamic $(COMPOSER_GEMM_ITEMS=dynamic b --hog $h)   static $(COMPOSER_GEMM_ITEMS=static b --hog $h)"
done
echo "B=32           dynamic $(COMPOSER_GEMM_ITEMS=dynamic b --batch 32)   static $(COMPOSER_GEMM_ITEMS=static b --batch 32)"
echo "c4             dynamic $(COMPOSER_GEMM_ITEMS=dynamic b --config c4)   static $(COMPOSER_GEMM_ITEMS=static b --config c4)"
for cus in 0 248; do
  echo "dp1 cus=$cus  dynamic $(COMPOSER_GEMM_ITEMS=dynamic COMPOSER_DP_GEMM_CUS=$cus dp)   static $(COMPOSER_GEMM_ITEMS=static COMPOSER_DP_GEMM_CUS=$cus dp)"
done
} | tee $out/ab_sched.txt
