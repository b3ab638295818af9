# per-kernel rocprofv3 stats of the default bench (or: bash tools/prof_stats.sh --workload c4_dna_1g)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/prof_stats
# 5 timed steps + 1 warm-up + the extra fully-profiled build of bench.py = 7 builds (the divisor below); the end-to-end leg is
# switched off so that no other build runs under the profiler
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end "$@" > $R/gpurun_out/prof_stats.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].split('(')[0].replace('void sa::', '').replace('sa::', '')
    print(f"{n[:64]:64s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['TotalDurationNs'])/1e6/7:7.3f} ms/build")
PY
