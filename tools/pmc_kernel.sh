# SQ counters of one kernel: bash tools/pmc_kernel.sh <kernel substring> [bench args]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift
cd /tmp
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"; do
  rm -rf $R/gpurun_out/pmck
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmck -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end "$@" > $R/gpurun_out/pmck.log 2>&1
  python3 - "$K" <<'PY'
import csv, glob, os, sys, collections
k = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmck/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for c, (n, v) in sorted(agg.items()):
    print(f"{c:28s} launches {n:3d}  total {v:16.0f}  per launch {v/max(n,1):14.0f}")
PY
done
