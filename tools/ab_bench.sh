# A/B of the sort engines on the GPU box:  bash tools/ab_bench.sh <outdir> <workload> ...
# one bench line per workload with the single-pass tile scatter (default) and one with the three-kernel pass
O=$1; shift
mkdir -p $O
for w in "$@"; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --workload $w --no-cpu-baseline --no-end-to-end > $O/b_os_$w.json 2> $O/b_os_$w.err
  SA_AMD_NO_ONESWEEP=1 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --workload $w --no-cpu-baseline --no-end-to-end > $O/b_old_$w.json 2> $O/b_old_$w.err
done
python - "$O" "$@" <<'PY'
import json, sys
O = sys.argv[1]
for w in sys.argv[2:]:
    for e in ("os", "old"):
        try:
            r = json.load(open(f"{O}/b_{e}_{w}.json"))
            print(w, e, r["ms_per_step"], "ms verified", r["verified"], {k: v["ms_per_step"] for k, v in r["kernels"].items() if v["ms_per_step"] > 0.3})
        except Exception as ex:
            print(w, e, "FAILED", ex)
PY
