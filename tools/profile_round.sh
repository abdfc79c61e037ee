#!/bin/bash
# Evidence passes of one round on the GPU box (each under rocprofv3, the program directly after `--`):
#   1. --kernel-trace --stats of the bench run                    -> <R>_bench[_TAG]_kernel_stats.csv + <R>_bench[_TAG].json
#   2. --pmc FETCH_SIZE, 3. --pmc WRITE_SIZE (separate passes)    -> <R>_traffic_<TAG>.json (tools/traffic.py, stamped with the source hash)
#   4. --pmc SQ counters (MFMA busy, VALU activity)               -> <R>_pmc_busy_<TAG>.json (tools/pmc_busy.py)
# usage (from the repo root, on the GPU box): bash tools/profile_round.sh r3 [TAG [bench args...]]
#   (every pass runs under `timeout` and leaves a line in progress.txt: a pass that stops writing is killed, not waited out)
#   TAG defaults to cfg2 (the default workload, configs[1]); e.g.  bash tools/profile_round.sh r3 cfg4 --config 4 --batch 128
# Output under gpurun_out/prof_<R>_<TAG>/ — copy what is to be judged to profiles/.
set -o pipefail
R=${1:-r3}
TAG=${2:-cfg2}
shift; shift
ARGS="$@"
OUT=gpurun_out/prof_${R}_${TAG}
mkdir -p $OUT
SUF=$([ "$TAG" = cfg2 ] && echo "" || echo "_$TAG")
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err && cp $OUT/bench.json $OUT/${R}_bench${SUF}.json
echo "[profile_round] kernel-trace pass" >> $OUT/progress.txt; timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --no-cpu-baseline $ARGS > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/${R}_bench${SUF}_kernel_stats.csv
echo "[profile_round] FETCH_SIZE pass" >> $OUT/progress.txt; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --no-cpu-baseline $ARGS --steps 1 --warmup 0 > $OUT/fetch.log 2>&1
echo "[profile_round] WRITE_SIZE pass" >> $OUT/progress.txt; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --no-cpu-baseline $ARGS --steps 1 --warmup 0 > $OUT/write.log 2>&1
python tools/traffic.py $(find $OUT/fetch -name '*counter_collection.csv' | head -1) $(find $OUT/write -name '*counter_collection.csv' | head -1) $OUT/${R}_traffic_${TAG}.json > $OUT/traffic.txt
echo "[profile_round] SQ pass" >> $OUT/progress.txt; timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- python bench.py --no-cpu-baseline $ARGS --steps 1 --warmup 0 > $OUT/sq.log 2>&1
python tools/pmc_busy.py $(find $OUT/sq -name '*counter_collection.csv' | head -1) $OUT/${R}_pmc_busy_${TAG}.json > $OUT/busy.txt
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq
cat $OUT/traffic.txt $OUT/busy.txt
