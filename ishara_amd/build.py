"""Build libishara_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")


def build(verbose: bool = True, jobs: int = 8) -> str:
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stdout.write(res.stdout[-4000:])
    if res.returncode != 0:
        raise RuntimeError("building libishara_hip.so failed")
    out = os.path.join(HERE, "libishara_hip.so")
    if not os.path.exists(out):
        raise RuntimeError(f"{out} was not produced")
    return out


if __name__ == "__main__":
    print(build())
