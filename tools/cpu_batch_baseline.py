"""SURVEY.md section 8d, batch config: min(8, nproc) independent single-thread CPU builds at once (the oracle's SA-IS,
stand-in for divsufsort), aggregate input MB/s.  python tools/cpu_batch_baseline.py [bytes per text]"""
import multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def one(args):
    rank, n = args
    import numpy as np
    from suffix_array_amd import corpus
    from conftest import Oracle
    t = corpus.uniform(n, 50 + rank)          # C5-style text (seeds 50..57), bounded length
    orc = Oracle()
    t0 = time.perf_counter(); orc.sais(t); return time.perf_counter() - t0

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else (32 << 20)
    procs = min(8, os.cpu_count() or 1)
    t0 = time.perf_counter()
    with mp.Pool(procs) as pool:
        each = pool.map(one, [(r, n) for r in range(procs)])
    wall = time.perf_counter() - t0
    print(f"{procs} concurrent single-thread builds of {n} bytes: slowest {max(each):.1f} s, wall {wall:.1f} s (incl. text generation), "
          f"aggregate {procs * n / 1e6 / max(each):.1f} MB/s, per core {n / 1e6 / (sum(each) / procs):.1f} MB/s; nproc = {os.cpu_count()}")
