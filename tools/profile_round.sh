#!/bin/bash
# Evidence passes of one round on the GPU box (each under rocprofv3, the program directly after `--`):
#   1. --kernel-trace --stats of the default bench run            -> gpurun_out/prof_rN/ (copy to profiles/ afterwards): rN_bench_kernel_stats.csv + rN_bench.json
#   2. --pmc FETCH_SIZE, 3. --pmc WRITE_SIZE (separate passes)    -> gpurun_out/prof_rN/ (copy to profiles/ afterwards): rN_traffic_cfg2.json (tools/traffic.py, stamped with the source hash)
#   4. --pmc SQ counters (MFMA busy, VALU activity)               -> gpurun_out/prof_rN/ (copy to profiles/ afterwards): rN_pmc_busy_cfg2.json (tools/pmc_busy.py)
# usage (from the repo root, on the GPU box): bash tools/profile_round.sh r2
set -o pipefail
R=${1:-r2}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err && cp $OUT/bench.json $OUT/${R}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/${R}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write.log 2>&1
python tools/traffic.py $(find $OUT/fetch -name '*counter_collection.csv' | head -1) $(find $OUT/write -name '*counter_collection.csv' | head -1) $OUT/${R}_traffic_cfg2.json > $OUT/traffic.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq.log 2>&1
python tools/pmc_busy.py $(find $OUT/sq -name '*counter_collection.csv' | head -1) $OUT/${R}_pmc_busy_cfg2.json > $OUT/busy.txt
cat $OUT/traffic.txt $OUT/busy.txt
