#!/usr/bin/env python3
"""Print per-kernel resource usage (VGPR/AGPR/LDS/scratch/spills) from a gfx950 .s file
produced by `hipcc -save-temps`."""
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size", txt, re.S):
    blk = m.group(0)
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    print(f"{g('name')[:70]:70s} vgpr={g('vgpr_count'):>3s} agpr={g('agpr_count'):>3s} sgpr={g('sgpr_count'):>3s} "
          f"lds={g('group_segment_fixed_size'):>6s} scratch={g('private_segment_fixed_size'):>4s} "
          f"vspill={g('vgpr_spill_count')} sspill={g('sgpr_spill_count')}")
