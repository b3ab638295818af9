for v in 1 3 4; do export SA_AMD_SORT32_VARIANT=$v; echo "sort32 variant $v"; timeout -k 10 200 python -m pytest tests -m gpu -x -q -k "32bit or two_stage or radix" 2>&1 | tail -1; timeout -k 10 300 python bench.py --steps 4 --warmup 1 --verify --no-cpu-baseline --workload c5_uniform_512m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print(d['config']['workload'][:14], d['ms_per_step'], 'ms verified', d['verified'], 'down32', k['k_radix_downsweep32']['ms_per_step'])"; done; unset SA_AMD_SORT32_VARIANT
for v in 22 35 36 37; do export SA_AMD_SORT_VARIANT=$v; echo "sort variant $v"; timeout -k 10 200 python -m pytest tests -m gpu -x -q -k "radix" 2>&1 | tail -1; timeout -k 10 300 python bench.py --steps 4 --warmup 1 --verify --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print(d['config']['workload'][:14], d['ms_per_step'], 'ms verified', d['verified'], 'down', k['k_radix_downsweep']['ms_per_step'])"; done
