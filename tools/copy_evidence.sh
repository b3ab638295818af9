# copy the judged files of one evidence run (bash tools/profile_round.sh <tag>, merged into gpurun_out/<tag>/) into profiles/
TAG=${1:-r02}
O=gpurun_out/$TAG
for w in c3_english_256m c3_iid_256m c2_uniform_256m c2_uniform_64m c4_dna_1g c5_uniform_512m; do cp $O/bench_$w.json profiles/${TAG}_bench_$w.json; done
grep "^{" $O/bench_2ranks_shared_gpu.json > profiles/${TAG}_bench_2ranks_shared_gpu.json
cp $O/prof_stats/c3_kernel_stats.csv profiles/${TAG}_c3_english_256m_kernel_stats.csv
cp $O/traffic.json profiles/traffic.json
cp $O/c3_dispatch_sequence.txt profiles/${TAG}_c3_dispatch_sequence.txt
for f in $O/round_trace_*.txt; do grep -v "amdgpu.ids" $f > profiles/${TAG}_$(basename $f); done
cp $O/adversarial_256m.txt profiles/${TAG}_adversarial_256m.txt
for f in host_path ab_text_rounds_vs_doubling ab_refinement_knobs; do grep -v "amdgpu.ids" $O/$f.txt > profiles/${TAG}_$f.txt; done
