set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --verify > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
for w in c2_uniform_64m c4_dna_1g c5_uniform_512m; do timeout -k 10 300 python bench.py --steps 5 --warmup 1 --verify --workload $w > gpurun_out/bench_$w.json 2>gpurun_out/bench_$w.err; done
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/pmc_FETCH_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/pmc_WRITE_SIZE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
cd $R
find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -2
find gpurun_out/pmc -name "*counter_collection.csv" | xargs ls -la | head
find gpurun_out/pmc -name "*.csv" -size +30M -delete
tail -1 gpurun_out/bench_c3.json | cut -c1-300
