# Round evidence, run on the GPU box from the repo root:  bash tools/profile_round.sh <tag> [part]   (tag: r03, ...)
# part 1: rocprofv3 kernel stats of the headline command, PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) -> traffic.json,
#         bench lines of every BASELINE config + the 256 MiB random text + the round-1 iid corpus + the C4 repeat variant
# part 2: timelines, adversarial families, host path, A/B runs, small-text latency, next-row timings, stamps, scatter probe
# everything lands in gpurun_out/<tag>/, to be copied into profiles/ by tools/copy_evidence.sh
set -e
TAG=${1:-r03}
PART=${2:-all}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ "$PART" = "all" ] || [ "$PART" = "1" ]; then
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o c3 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/prof_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc/pmc_FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc/pmc_WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/pmc_write.log 2>&1
cd $R
# builds per PMC pass: first touch + profiled build + 1 warm-up + 2 timed = 5 (bench.py run()); the verification kernels run once
python tools/pmc_traffic.py $O/pmc $O/traffic.json c3_english_256m 268435456 5
cp $O/traffic.json profiles/traffic.json     # bench.py quotes the figures from there: measured first, on this code
# the same two passes on the 256 MiB random text (the 32-bit first stage: two global passes + k_bucket_sort), evidence only
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc2/pmc_FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --workload c2_uniform_256m --no-cpu-baseline --no-end-to-end > $O/pmc2_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc2/pmc_WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --workload c2_uniform_256m --no-cpu-baseline --no-end-to-end > $O/pmc2_write.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc2 $O/traffic_c2_uniform_256m.json c2_uniform_256m 268435456 5
find $O/pmc2 -name "*.csv" -delete
for w in c3_english_256m c3_iid_256m c2_uniform_256m c2_uniform_64m c4_dna_1g c4_dna_repeats_1g c5_uniform_512m; do
  timeout -k 10 500 python bench.py --steps 5 --warmup 1 --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
done
SA_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_shared_gpu.json 2> $O/bench_2ranks_shared_gpu.err
find $O/pmc -name "*.csv" -delete
find $O/prof_stats -name "*kernel_trace.csv" -delete
for w in c3_english_256m c3_iid_256m c2_uniform_256m c2_uniform_64m c4_dna_1g c4_dna_repeats_1g c5_uniform_512m; do python - <<PY
import json
r = json.load(open("$O/bench_$w.json"))
e = r["end_to_end"]
print("$w", r["value"], "MB/s", r["ms_per_step"], "ms  verified", r["verified"], " e2e reused", e["reused_buffer"]["ms"], "fresh", e["fresh_buffer"]["ms"], " roofline", r["roofline"]["kernel"], r["roofline"]["frac"], [(k["name"], k["ms_per_step"], k.get("frac")) for k in r["roofline"]["kernels"]])
PY
done
fi
if [ "$PART" = "all" ] || [ "$PART" = "2" ]; then
for w in c3_english_256m c3_iid_256m adv:one:268435456 adv:twice:268435456 adv:p1000:268435456 adv:fib:268435456; do
  f=$(echo $w | tr ':' '_')
  SA_AMD_VERBOSE=3 timeout -k 10 200 python tools/round_trace.py $w > $O/round_trace_$f.txt 2>&1
done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof_seq -o d --output-format csv -- python3 $R/tools/ab_env.py c3_english_256m - > $O/prof_seq.log 2>&1
cd $R
python tools/trace_sequence.py $O/prof_seq/d_kernel_trace.csv -2 > $O/c3_dispatch_sequence.txt
rm -rf $O/prof_seq
timeout -k 10 900 python tools/adversarial_timing.py 268435456 > $O/adversarial_256m.txt 2>&1
( if [ -x tools/bin/pcie_probe ]; then timeout -k 10 300 tools/bin/pcie_probe 1024; fi; timeout -k 10 300 python tools/host_api_timing.py ) > $O/host_path.txt 2>&1
timeout -k 10 300 python tools/small_latency.py > $O/small_latency.txt 2>&1
timeout -k 10 300 python tools/midsize_timing.py > $O/midsize_timing.txt 2>&1
timeout -k 10 300 python tools/midsize_d2h_probe.py > $O/midsize_d2h_probe.txt 2>&1
timeout -k 10 300 python tools/extras_bench.py > $O/extras.txt 2>&1
timeout -k 10 300 python tools/search_bench.py > $O/search_throughput.txt 2>&1
( timeout -k 10 120 python tools/onesweep_stamps.py 32 26 0; timeout -k 10 120 python tools/onesweep_stamps.py 64 26 0 ) > $O/onesweep_stamps.txt 2>&1
timeout -k 10 300 python tools/group_sort_stamps.py > $O/group_sort_stamps.txt 2>&1
( if [ -x tools/bin/scatter_probe ]; then timeout -k 10 60 tools/bin/scatter_probe 28 8 8 0; timeout -k 10 60 tools/bin/scatter_probe 28 4 12 0; fi ) > $O/scatter_probe.txt 2>&1 || true
timeout -k 10 300 python tools/ab_env.py c3_english_256m - SA_AMD_NO_ONESWEEP=1 SA_AMD_ONESWEEP_FLAGS=1 SA_AMD_NO_REPEAT_PROBE=1 SA_AMD_NO_REPEAT_PROBE=1,SA_AMD_MAX_TEXT_ROUNDS=1 SA_AMD_FORCE_DENSE=1 SA_AMD_CHASE=1 SA_AMD_CHASE=3 SA_AMD_CHASE=15 SA_AMD_SCATTER_LEVELS=1 SA_AMD_BINNED_MIN=4194304 SA_AMD_BINNED_MIN=268435457 SA_AMD_GROUP_CAP=512 SA_AMD_NO_GRAM_KEYS=1 > $O/ab_knobs_c3.txt 2>&1
timeout -k 10 300 python tools/ab_env.py c2_uniform_256m - SA_AMD_NO_ONESWEEP=1 SA_AMD_ONESWEEP_FLAGS=1 SA_AMD_ONESWEEP32_SHAPE=1 SA_AMD_ONESWEEP32_SHAPE=2 SA_AMD_ONESWEEP32_SHAPE=3 >> $O/ab_knobs_c3.txt 2>&1
tail -3 $O/adversarial_256m.txt; tail -4 $O/ab_knobs_c3.txt
fi
