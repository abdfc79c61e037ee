#!/bin/bash
# usage (on the GPU box, repo root): bash tools/kstats.sh <tag> [bench args]   -> gpurun_out/<tag>_kernel_stats.csv + gpurun_out/<tag>.json
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/_st_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_st_$TAG -- python bench.py --no-cpu-baseline "$@" > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err || exit 1
cp "$(find gpurun_out/_st_$TAG -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/_st_$TAG
python - "$TAG" <<'P'
import csv, json, sys
tag = sys.argv[1]
d = json.loads(open(f"gpurun_out/{tag}.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"])
rows = list(csv.DictReader(open(f"gpurun_out/{tag}_kernel_stats.csv")))
for r in rows[:32]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), f'{float(r["AverageNs"]) / 1e3:9.1f} us', f'{float(r["TotalDurationNs"]) / 1e6:8.2f} ms')
P
