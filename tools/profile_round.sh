# Round evidence, run on the GPU box from the repo root:  bash tools/profile_round.sh <tag> [part]   (tag: r04, ...)
# part 1: per WORKLOAD (every BASELINE config + north_star's literal 256 MiB random text): rocprofv3 --kernel-trace --stats of the bench
#         command on that workload -> <tag>_<workload>_kernel_stats.csv, so that every roofline fraction of the bench line can be
#         reproduced from profiles/; PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) on the headline workload -> traffic.json;
#         the default `python bench.py` line (headline + configs + batch_api + cpu_baseline) and the two-rank rehearsal
# part 2: timelines, adversarial families, host path (early download off / on, copy engine probes), small / mid-size latency, next-row timings, A/B knobs
# everything lands in gpurun_out/<tag>/, to be copied into profiles/ by tools/copy_evidence.sh
set -e
TAG=${1:-r04}
PART=${2:-all}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
WORKLOADS="c3_english_256m c2_uniform_64m c2_uniform_256m c4_dna_1g c5_uniform_512m"
if [ "$PART" = "all" ] || [ "$PART" = "1" ]; then
cd /tmp
for w in $WORKLOADS; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_$w -o k -- python3 $R/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs --no-batch-api > $O/prof_stats_$w.log 2>&1
  cp $O/prof_stats_$w/k_kernel_stats.csv $O/${w}_kernel_stats.csv
  rm -rf $O/prof_stats_$w
  echo "kernel stats $w done"
done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc/pmc_FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs --no-batch-api > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc/pmc_WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs --no-batch-api > $O/pmc_write.log 2>&1
cd $R
# builds per PMC pass: first touch + profiled build + 1 warm-up + 2 timed = 5 (bench.py measure_resident()); the verification kernels run once
python tools/pmc_traffic.py $O/pmc $O/traffic.json c3_english_256m 268435456 5
cp $O/traffic.json profiles/traffic.json     # bench.py quotes the figures from there: measured first, on this code
find $O/pmc -name "*.csv" -delete
echo "pmc done"
timeout -k 10 550 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "default bench done"
for w in c3_iid_256m c4_dna_repeats_1g; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --workload $w --no-configs --no-batch-api --cpu-sample 67108864 > $O/bench_$w.json 2> $O/bench_$w.err
done
SA_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_shared_gpu.json 2> $O/bench_2ranks_shared_gpu.err
python - <<PY
import json
r = json.load(open("$O/bench_default.json"))
print("c3", r["value"], "MB/s", r["ms_per_step"], "ms verified", r["verified"], "e2e fresh", r["end_to_end"]["fresh_buffer"]["ms"], "reused", r["end_to_end"]["reused_buffer"]["ms"], "roofline", r["roofline"]["kernel"], r["roofline"]["frac"])
for k, c in r["configs"].items():
    if not k.startswith("_"):
        print(k, c["ms_per_build"], "ms", c["MB_per_s"], "MB/s verified", c["verified"], c["roofline"]["kernel"], c["roofline"]["frac"], "e2e fresh", c["end_to_end"]["fresh_buffer"]["ms"])
PY
fi
if [ "$PART" = "all" ] || [ "$PART" = "2" ]; then
for w in c3_english_256m c3_iid_256m adv:one:268435456 adv:twice:268435456 adv:p1000:268435456 adv:fib:268435456; do
  f=$(echo $w | tr ':' '_')
  SA_AMD_VERBOSE=3 timeout -k 10 200 python tools/round_trace.py $w > $O/round_trace_$f.txt 2>&1
done
echo "round traces done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof_seq -o d --output-format csv -- python3 $R/tools/ab_env.py c3_english_256m - > $O/prof_seq.log 2>&1
cd $R
python tools/trace_sequence.py $O/prof_seq/d_kernel_trace.csv -2 > $O/c3_dispatch_sequence.txt
rm -rf $O/prof_seq
timeout -k 10 900 python tools/adversarial_timing.py 268435456 > $O/adversarial_256m.txt 2>&1
echo "adversarial done"
( for d in 0 4; do echo "== SA_AMD_EARLY_DIV=$d (0: the whole array is downloaded behind the build)"; SA_AMD_EARLY_DIV=$d SA_AMD_VERBOSE=2 timeout -k 10 100 python tools/mid_build.py english_corpus 268435456 6 2>&1 | grep -E "n=268435456|best" | cut -c1-260; done ) > $O/early_download.txt 2>&1
( if [ -x tools/bin/pcie_probe ]; then timeout -k 10 300 tools/bin/pcie_probe 1024; fi; timeout -k 10 300 python tools/host_api_timing.py ) > $O/host_path.txt 2>&1
timeout -k 10 300 python tools/small_latency.py > $O/small_latency.txt 2>&1
timeout -k 10 300 python tools/midsize_timing.py > $O/midsize_timing.txt 2>&1
( for sz in 131072 1048576 4194304; do SA_AMD_VERBOSE=3 timeout -k 10 60 python tools/mid_build.py english_corpus $sz 8 2>&1 | grep -E "best|read-backs" | cut -c1-120 | uniq; done ) >> $O/midsize_timing.txt 2>&1
timeout -k 10 300 python tools/midsize_knobs.py - SA_AMD_NO_FLAT_RULE=1 SA_AMD_NO_POSTED_READBACK=1 SA_AMD_NO_UPFRONT_COUNTS=1 SA_AMD_NO_UPFRONT_COUNTS=1,SA_AMD_COUNT_NEXT_MIN_N=0,SA_AMD_COUNT_NEXT_BELOW_N=0 > $O/midsize_knobs.txt 2>&1
# the copy engine's two rates, the read-back's two forms (plain HIP programs), and the library's download in both states
mkdir -p tools/bin
for pgm in readback_probe realloc_dma_probe; do hipcc --offload-arch=gfx950 -O2 -o tools/bin/$pgm tools/$pgm.hip > /dev/null 2>&1; done
timeout -k 10 100 tools/bin/readback_probe > $O/readback_probe.txt 2>&1
( for v in 1 2 5; do echo "== variant $v"; timeout -k 10 60 tools/bin/realloc_dma_probe $v; done ) > $O/realloc_dma_probe.txt 2>&1
timeout -k 10 400 python tools/d2h_ab.py 2>&1 | grep -E "MiB random, |C3|copy engine out" > $O/d2h_ab.txt
timeout -k 10 300 python tools/extras_bench.py > $O/extras.txt 2>&1
timeout -k 10 300 python tools/search_bench.py > $O/search_throughput.txt 2>&1
timeout -k 10 300 python tools/ab_env.py c3_english_256m - SA_AMD_NO_DEFER=1 SA_AMD_ONESWEEP_FLAGS=1 SA_AMD_NO_REPEAT_PROBE=1 SA_AMD_FORCE_DENSE=1 SA_AMD_CHASE=1 SA_AMD_CHASE=3 SA_AMD_CHASE=15 SA_AMD_SCATTER_LEVELS=1 SA_AMD_BINNED_MIN=4194304 SA_AMD_GROUP_CAP=512 SA_AMD_NO_GRAM_KEYS=1 > $O/ab_knobs_c3.txt 2>&1
tail -3 $O/adversarial_256m.txt; tail -4 $O/ab_knobs_c3.txt; cat $O/early_download.txt | tail -4
fi
