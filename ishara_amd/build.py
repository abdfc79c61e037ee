"""Build libishara_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")


def source_hash() -> str:
    """Hash of every source the library is built from (csrc/*.hip, *.h, the public header): stamps profiles/*_traffic.json so
    bench.py only quotes PMC traffic measured on the build it is running."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(os.path.dirname(HERE), "include", "ishara_hip.h"),
                                                                                                 os.path.join(CSRC, "Makefile")]      # build flags are part of the build
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


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
