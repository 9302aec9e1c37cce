#!/bin/bash
# Steps 2 and 3 of tools/profile_round6.sh alone: the PMC traffic of every kernel class inside the step (stamped with the loaded library's
# build key) and the default bench line that prints it -- for a library commit that changes the source text but not one device
# instruction (tools/isa_hash.py says so), after the full evidence pass.   bash tools/refresh_traffic.sh <tag>
tag=${1:-r6_tr}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-decode --no-extras"
for cfg in "c2 128 131072" "c2 32 32768" "c4 32 65536"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$1_$2_$c -o k -- $BENCH --config $1 --batch $2 --steps 2 --warmup 1 > $out/pmc_$1_$2_$c.log 2>&1
  done
  python3 tools/make_traffic_json.py $1_tokens$3 $out/pmc_$1_$2_FETCH_SIZE $out/pmc_$1_$2_WRITE_SIZE $out/hbm_traffic.json
done
cp $out/hbm_traffic.json profiles/hbm_traffic.json
timeout 900 python3 bench.py > $out/bench_c2.json 2> $out/bench_c2.err
find $out -name "*.db" -delete
find $out -name "k_kernel_trace.csv" -delete
find $out -name "k_counter_collection.csv" -delete
