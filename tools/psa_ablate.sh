#!/bin/bash
# timing ablation of the per-sample-affine wgrad kernel (gemm.hip, TnPsa.dbg): bash tools/psa_ablate.sh   (GPU box, repo root)
for d in 0 1 5 21; do
  export ISHARA_PSA_DBG=$d
  bash tools/kstats.sh psa_abl_$d --steps 6 --warmup 2 > gpurun_out/psa_abl_$d.txt 2>&1 || { echo "dbg $d failed"; tail -3 gpurun_out/psa_abl_$d.txt; exit 1; }
  echo "dbg=$d $(grep 'false, true' gpurun_out/psa_abl_$d.txt) | $(grep 'false, false' gpurun_out/psa_abl_$d.txt | cut -c92-)"
done
