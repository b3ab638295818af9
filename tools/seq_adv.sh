# dispatch sequence of one adversarial build: bash tools/seq_adv.sh adv:fib:268435456
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/prof_adv
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_adv -o d -- python3 $R/tools/round_trace.py "$1" > $R/gpurun_out/prof_adv.log 2>&1
cd $R
python3 tools/trace_sequence.py $(ls gpurun_out/prof_adv/*kernel_trace.csv gpurun_out/prof_adv/*/*kernel_trace.csv 2>/dev/null | head -1) -1
