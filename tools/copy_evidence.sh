# copy the judged files of one evidence run (bash tools/profile_round.sh <tag>, merged into gpurun_out/<tag>/) into profiles/
TAG=${1:-r04}
O=gpurun_out/$TAG
for w in c3_english_256m c2_uniform_64m c2_uniform_256m c4_dna_1g c5_uniform_512m; do [ -s $O/${w}_kernel_stats.csv ] && cp $O/${w}_kernel_stats.csv profiles/${TAG}_${w}_kernel_stats.csv; done
[ -s $O/bench_default.json ] && grep "^{" $O/bench_default.json > profiles/${TAG}_bench_default.json
for w in c3_iid_256m c4_dna_repeats_1g; do [ -s $O/bench_$w.json ] && grep "^{" $O/bench_$w.json > profiles/${TAG}_bench_$w.json; done
[ -s $O/bench_2ranks_shared_gpu.json ] && grep "^{" $O/bench_2ranks_shared_gpu.json > profiles/${TAG}_bench_2ranks_shared_gpu.json
[ -s $O/traffic.json ] && cp $O/traffic.json profiles/traffic.json
[ -s $O/c3_dispatch_sequence.txt ] && cp $O/c3_dispatch_sequence.txt profiles/${TAG}_c3_dispatch_sequence.txt
for f in $O/round_trace_*.txt; do [ -s $f ] && grep -v "amdgpu.ids" $f > profiles/${TAG}_$(basename $f); done
for f in adversarial_256m host_path small_latency midsize_timing midsize_knobs readback_probe extras search_throughput ab_knobs_c3 early_download; do
  [ -s $O/$f.txt ] && grep -v "amdgpu.ids" $O/$f.txt > profiles/${TAG}_$f.txt
done
true
