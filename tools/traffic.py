#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md §HBM prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of a wide coalesced read stream (x2); WRITE_SIZE is exact for 16-B-per-lane stores.
usage: traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
The output is stamped with ishara_amd.build.source_hash() of the tree it is run in: run it on the same sources the profiled
.so was built from (bench.py refuses the table when the stamp differs from the running build)."""
import collections, csv, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ishara_amd.build import source_hash

def short(name):
    m = re.match(r"_Z\d+(\w+?_kernel)(I.*?E)?v?", name)
    base = name
    if name.startswith("_Z"):
        mm = re.match(r"_Z(\d+)", name); n = int(mm.group(1)); base = name[len(mm.group(0)):len(mm.group(0)) + n]
        rest = name[len(mm.group(0)) + n:]
        targs = re.match(r"I((?:DF16b|f|Li\d+E)+)E", rest)
        if targs:
            parts = re.findall(r"DF16b|f|Li\d+E", targs.group(1))
            base += "<" + ",".join({"DF16b": "bf16", "f": "f32"}.get(p, p[2:-1]) for p in parts) + ">"
    else:
        base = re.sub(r"\(.*", "", name).replace("void ", "")
    return base

def load(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = agg[short(r["Kernel_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg

def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, [0, 0]); w, nw = write.get(k, [0, 0])
        n = max(nf, nw, 1)
        out[k] = dict(launches=n, read_bytes_per_launch=2.0 * f * 1024 / n, write_bytes_per_launch=w * 1024 / n,
                      hbm_bytes_per_launch=(2.0 * f + w) * 1024 / n)
    json.dump(dict(source_hash=source_hash(), note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `python bench.py --steps 1 --warmup 0 --no-cpu-baseline`; "
                        "KiB units, FETCH_SIZE x2 (gfx950 wide-read correction)", kernels=out), open(sys.argv[3], "w"), indent=1)
    top = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]
    for k, v in top:
        print(f"{k:44s} n={v['launches']:4d} read={v['read_bytes_per_launch']/1e6:8.1f} MB write={v['write_bytes_per_launch']/1e6:8.1f} MB")


if __name__ == "__main__":
    main()
