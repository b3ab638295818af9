# bash tools/ab_env_bench.sh <outdir> <workload> <tag>=<ENV=VAL[,ENV=VAL]> ...   -- one bench line per environment
O=$1; W=$2; shift; shift
mkdir -p $O
for spec in "$@"; do
  tag=${spec%%=*}; envs=${spec#*=}
  ( for kv in $(echo "$envs" | tr ',' ' '); do [ "$kv" != "-" ] && export "$kv"; done
    timeout -k 10 300 python bench.py --steps 5 --warmup 1 --workload $W --no-cpu-baseline --no-end-to-end > $O/ab_${W}_$tag.json 2> $O/ab_${W}_$tag.err )
  python - "$O/ab_${W}_$tag.json" "$W" "$tag" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    print(sys.argv[2], sys.argv[3], r["ms_per_step"], "ms verified", r["verified"], {k: v["ms_per_step"] for k, v in r["kernels"].items() if v["ms_per_step"] > 0.3})
except Exception as ex:
    print(sys.argv[2], sys.argv[3], "FAILED", ex)
PY
done
