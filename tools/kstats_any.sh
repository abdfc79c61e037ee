#!/bin/bash
# Per-kernel table of ANY python tool: bash tools/kstats_any.sh <tag> <script.py> [args]   -> gpurun_out/<tag>_kernel_stats.csv + gpurun_out/<tag>.json (the tool's stdout)
# (rocprofv3 --kernel-trace --stats with the program directly after `--`; no counters in this pass)
set -o pipefail
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/_st_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_st_$TAG -- python "$@" > gpurun_out/$TAG.json 2> gpurun_out/$TAG.err || { tail -5 gpurun_out/$TAG.err; exit 1; }
cp "$(find gpurun_out/_st_$TAG -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/_st_$TAG
tail -1 gpurun_out/$TAG.json
python - "$TAG" <<'P'
import csv, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/{tag}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':90s} calls   avg us  total ms   share")
for r in rows[:28]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), f'{float(r["AverageNs"]) / 1e3:8.1f}', f'{float(r["TotalDurationNs"]) / 1e6:9.2f}', f'{100 * float(r["TotalDurationNs"]) / tot:6.1f}%')
P
