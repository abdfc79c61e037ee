#!/usr/bin/env python3
"""Per-kernel MFMA-busy / VALU-busy from one rocprofv3 --pmc pass (SQ counters), summed over a kernel's dispatches:
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)   MFMA-pipe cycles per SIMD (4 SIMDs per CU) over the cycles the CU had waves
  valu_busy = SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES               SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md): x4 cycles, /4 SIMDs
Calibration: the K=256 -> N=512 A-stationary GEMM issues 25.8 GFLOP / 16384 = 1.57 M MFMAs of 16 cycles per launch = 25 M SIMD-cycles
against ~82 M SIMD-cycles of a 40 us launch at 2 GHz (0.31); the formula gives 0.27.  ROCm 7.2 ships no gfx950 derived-counter section,
so the table keeps every raw sum and the ratios can be re-derived.  usage: pmc_busy.py <counter_collection.csv> <out.json>"""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ishara_amd.build import source_hash
from tools.traffic import short

agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        n[k] += 1
out = {}
for k, c in agg.items():
    busy = c.get("SQ_BUSY_CU_CYCLES") or c.get("SQ_BUSY_CYCLES") or 0.0
    rec = dict(counters=dict(c), dispatches=n.get(k, 0))
    if busy:
        rec["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * busy)
        if "SQ_ACTIVE_INST_VALU" in c: rec["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] / busy
        if "SQ_WAIT_INST_ANY" in c: rec["issue_stall_frac_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    out[k] = rec
json.dump(dict(source_hash=source_hash(), note="rocprofv3 --pmc SQ pass over `python bench.py --steps 1 --warmup 0 --no-cpu-baseline`; raw counter sums per kernel; "
               "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES); valu_busy_frac = SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES", kernels=out), open(sys.argv[2], "w"), indent=1)
top = sorted(out.items(), key=lambda kv: -kv[1]["counters"].get("SQ_WAVE_CYCLES", 0))[:14]
for k, v in top:
    print(f"{k:46s} mfma_busy={v.get('mfma_busy_frac', float('nan')):6.3f} valu_busy={v.get('valu_busy_frac', float('nan')):6.3f}")
