#!/bin/bash
# A/B of the downsweep configurations (SA_AMD_SORT_VARIANT) on one workload; prints kernel ms per build
W=${1:-c2_uniform_64m}
for v in ${VARIANTS:-0 1 2 3}; do      # the product library has four tile-scatter shapes; ablations and stamps: diagnostic library only (tools/phase_stamps.py)
  SA_AMD_SORT_VARIANT=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-end-to-end --no-cpu-baseline --workload $W 2>/dev/null > gpurun_out/sweep_$v.log
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/sweep_$v.log").read().strip().splitlines()[-1])
    k=d["kernels"]
    print("variant $v", "$W", "ms/step", d["ms_per_step"], "verified", d["verified"], "down", k["k_radix_downsweep"]["ms_per_step"], "up", k["k_radix_upsweep"]["ms_per_step"], "roofline", d["roofline"]["achieved"])
except Exception as e:
    print("variant $v failed", e)
PY
done
