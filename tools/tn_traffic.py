#!/usr/bin/env python3
"""HBM read bytes of single weight-gradient launches, 128 x 128 tile kernel vs the 256 x 256 tile kernel (run under rocprofv3 --pmc FETCH_SIZE, then
`tools/tn_traffic.py --parse <counter_collection.csv> gpurun_out/tn_traffic_plan.txt`): prints, per launch in dispatch order, the fetched bytes
(x2, gfx950 wide reads) next to the algorithmic M x (K + N) x 2.  Without the profiler it prints the time per launch incl. the slab sums.
(profiles/r3_tn_traffic_pacing.txt was made with an earlier form of this script on a build that paced the sibling workgroups: DESIGN.md 7c.)"""
import sys
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import csv
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if r["Counter_Name"] == "FETCH_SIZE" and ("gemm_tn_tr" in r["Kernel_Name"] or "gemm_tn_big" in r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    plan = [l.split() for l in open(sys.argv[3])]
    for r, p in zip(rows, plan):
        M, K, N = int(p[1]), int(p[2]), int(p[3]); alg = M * (K + N) * 2
        got = 2.0 * float(r["Counter_Value"]) * 1024
        print(f"{p[0]:8s} M{M} K{K} N{N} grid={r['Grid_Size']:>8s} fetched {got/1e6:8.1f} MB  algorithmic {alg/1e6:8.1f} MB  x{got/alg:.2f}")
    sys.exit(0)
import ctypes as C, torch
sys.path.insert(0, ".")
from ishara_amd import _lib
lib = _lib.load()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [(262144, 512, 512), (262144, 512, 1024), (98304, 256, 256), (98304, 256, 512), (98304, 256, 768)]
V = {"tile128": 0, "tile256": 1}          # ishara_debug_set_nt_big: the 128 x 128 tile kernel / the 256 x 256 tile kernel where it applies
plan = open("gpurun_out/tn_traffic_plan.txt", "w")
for (M, K, N) in SHAPES:
    x = torch.randn(M, K, device="cuda").bfloat16(); dy = torch.randn(M, N, device="cuda").bfloat16()
    W = torch.randn(K, N, device="cuda") / K ** 0.5
    dW = torch.zeros(K, N, device="cuda"); db = torch.zeros(N, device="cuda")
    sc = torch.empty(int(lib.ishara_op_scratch_bytes(M, K, N)) + 256, dtype=torch.uint8, device="cuda")
    scp = C.c_void_p(sc.data_ptr() + (-sc.data_ptr()) % 256)
    for name, bits in V.items():
        lib.ishara_debug_set_nt_big(bits)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(3):
            if i == 1: e0.record()
            lib.ishara_op_dense_bwd(1, _lib.ptr(x), _lib.ptr(W), _lib.ptr(dy), None, _lib.ptr(dW), _lib.ptr(db), M, K, N, scp, st())
            plan.write(f"{name} {M} {K} {N}\n")
        e1.record(); torch.cuda.synchronize()
        print(f"{name:8s} M{M} K{K} N{N}: {e0.elapsed_time(e1)*1e3/2:.0f} us (wgrad + slab sums)", flush=True)
    lib.ishara_debug_set_nt_big(1)
    del x, dy, sc
plan.close()
