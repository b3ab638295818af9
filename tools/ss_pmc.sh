# SQ counters of the sample-sort kernels on a C3 build (GPU box, from the repo root): bash tools/ss_pmc.sh <out dir>
set -e
R=$GRAFT_REPO_ROOT
O=$R/${1:-gpurun_out/ss_pmc}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/p_$tag -o d -- python3 $R/tools/mid_build.py english_corpus 268435456 2 > $O/log_$tag.txt 2>&1 || true
  python3 - <<PY >> $O/summary.txt
import csv, glob, collections
for f in glob.glob("$O/p_$tag/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "k_ss_" in k or "k_onesweep<1024, 8" in k or "k_bucket_sort" in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        print(k, {c: (v, cnt[(k, c)]) for c, v in d.items()})
PY
  rm -rf $O/p_$tag
done
cat $O/summary.txt
