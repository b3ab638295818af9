# kernel + memory-copy traces of a C3 host-pointer build with the early download off and on (run on the GPU box from the repo root):
#   bash tools/early_trace.sh <out dir>
# writes seq_<div>.txt (dispatch sequence of the last build, tools/trace_sequence.py) and memcpy_<div>.txt (the long copies)
set -e
R=$GRAFT_REPO_ROOT
O=$R/${1:-gpurun_out/early_trace}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for d in 0 4; do
  SA_AMD_EARLY_DIV=$d timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace -d $O/p$d -o d --output-format csv -- python3 $R/tools/mid_build.py english_corpus 268435456 3 > $O/mid_$d.log 2>&1
  python3 $R/tools/trace_sequence.py $O/p$d/d_kernel_trace.csv -1 > $O/seq_$d.txt
  python3 $R/tools/memcpy_trace.py $O/p$d/d_memory_copy_trace.csv > $O/memcpy_$d.txt
  rm -rf $O/p$d
done
tail -1 $O/seq_0.txt $O/seq_4.txt
