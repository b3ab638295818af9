# same-box A/B of two BUILDS of the library on the GPU box:  bash tools/ab_lib.sh <variant.so> <outdir> <workload> ...
# (the box's copy of the repo is scratch: the variant is copied over the product library for the second run of each workload)
V=$1; O=$2; shift; shift
mkdir -p $O
cp suffix_array_amd/libsuffix_array_amd.so /tmp/base_lib.so
for w in "$@"; do
  cp /tmp/base_lib.so suffix_array_amd/libsuffix_array_amd.so
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --workload $w --no-cpu-baseline --no-end-to-end --no-batch-api > $O/b_base_$w.json 2> $O/b_base_$w.err
  cp $V suffix_array_amd/libsuffix_array_amd.so
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --workload $w --no-cpu-baseline --no-end-to-end --no-batch-api > $O/b_var_$w.json 2> $O/b_var_$w.err
done
cp /tmp/base_lib.so suffix_array_amd/libsuffix_array_amd.so
python - "$O" "$@" <<'PY'
import json, sys
O = sys.argv[1]
for w in sys.argv[2:]:
    for e in ("base", "var"):
        try:
            r = json.load(open(f"{O}/b_{e}_{w}.json"))
            print(w, e, r["ms_per_step"], "ms verified", r["verified"], {k: v["ms_per_step"] for k, v in r["kernels"].items() if v["ms_per_step"] > 0.3})
        except Exception as ex:
            print(w, e, "FAILED", ex)
PY
