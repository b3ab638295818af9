"""Where do the mid-size host-pointer calls spend their download?  Texts of 16 / 64 MiB of random bytes through
sa_amd_saca_u8 (`saca()`, reference src/saca.rs:9-15) with the download route varied per call: plain hipMemcpy into the
caller's pageable array (SA_AMD_COPY_THREADS=0), the staged route with 4 / 12 helpers, each with the main thread where
the scheduler put it and pinned to the GPU's NUMA node.  Prints the phases of the best call.
python tools/midsize_d2h_probe.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import suffix_array_amd as sa
from suffix_array_amd import corpus

libc = ctypes.CDLL(None)
def cpu_now():
    return libc.sched_getcpu()

def node_of_cpu(c):
    for d in sorted(os.listdir("/sys/devices/system/node")):
        if d.startswith("node") and os.path.exists(f"/sys/devices/system/node/{d}/cpu{c}"):
            return int(d[4:])
    return -1

def cpus_of_node(k):
    out = set()
    for part in open(f"/sys/devices/system/node/node{k}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out

all_cpus = os.sched_getaffinity(0)
print(f"affinity: {len(all_cpus)} cpus, main thread on cpu {cpu_now()} = node {node_of_cpu(cpu_now())}", flush=True)
os.environ["SA_AMD_VERBOSE"] = "2"
t = corpus.uniform(1 << 20, 3); out = np.zeros(t.size + 1, dtype=np.uint32); sa.saca(t, out)      # prints the GPU's node
os.environ.pop("SA_AMD_VERBOSE")
nodes = [int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node")]
for mb in (16, 64, 256):
    n = mb << 20
    t = corpus.uniform(n, 3)
    for where in ["free"] + [f"node{k}" for k in sorted(nodes)]:
        if where == "free":
            os.sched_setaffinity(0, all_cpus)
        else:
            want = cpus_of_node(int(where[4:])) & all_cpus
            if not want:
                continue
            os.sched_setaffinity(0, want)
        out = np.zeros(n + 1, dtype=np.uint32)       # first touched by this thread where it runs now
        out[:] = 1
        for threads, smin in (("0", None), ("4", "0"), ("12", "0")):
            os.environ["SA_AMD_COPY_THREADS"] = threads
            if smin is None: os.environ.pop("SA_AMD_STAGED_MIN_BYTES", None)
            else: os.environ["SA_AMD_STAGED_MIN_BYTES"] = smin
            sa.saca(t, out)
            best, bt = 1e9, None
            for _ in range(5):
                t0 = time.perf_counter(); sa.saca(t, out); dt = time.perf_counter() - t0
                if dt < best: best, bt = dt, sa.last_host_timing()
            print(f"{mb:4d} MiB main {where:6s} (cpu {cpu_now():3d}) copy_threads={threads:2s}: {best*1e3:7.2f} ms | h2d {bt['h2d']:.2f} build {bt['build']:.2f} "
                  f"d2h {bt['d2h']:.2f} = {4*(n+1)/bt['d2h']/1e6:5.1f} GB/s", flush=True)
os.sched_setaffinity(0, all_cpus)
