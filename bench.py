#!/usr/bin/env python3
"""bench.py -- suffix-array construction throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one complete SuffixArray::new-equivalent build (device-resident text in, device
resident SA out) of one synthetic text.  Default workload: the configuration the metric is
quoted on, C3 = 256 MiB English-like text.  With N ranks every rank
builds its own independent text (seed + rank) on its own GPU: weak scaling, no data-path
collective; torch.distributed is used for the barrier and the
max-over-ranks time only.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed inside the
timed region; the other kernels' table `kernels` comes from one extra build outside it, so that
their event records do not sit in the measured time) and `cpu_baseline` (the oracle's
single-thread SA-IS on a bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DOMINANT = ("k_radix_downsweep", "k_radix_downsweep32")     # 64-bit-key / 32-bit-key tile scatter: the one with more time
# algorithmic bytes per unit (element) of each kernel class, DESIGN.md section 3
ALGO_BYTES = {"k_byte_hist": 1, "k_build_keys": 9, "k_radix_upsweep": 8, "k_radix_downsweep": 24,
              "k_rr_count": 8, "k_rr_apply": 24, "k_gather_key2": 20, "k_scatter_pairs": 16,
              "k_radix_upsweep32": 4, "k_radix_downsweep32": 16}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3_english_256m",
                    help="c2_uniform_64m | c3_english_256m | c4_dna_1g | c5_uniform_512m")
    ap.add_argument("--n", type=int, default=None, help="override the text length (bytes)")
    ap.add_argument("--verify", action="store_true", help="check the last SA with the oracle's linear verifier")
    ap.add_argument("--verify-gpu", action="store_true", help="check the last SA with the HIP integrity check (any n)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=64 << 20, help="bytes of the workload timed on the CPU")
    return ap.parse_args()


def load_oracle():
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.check_call(["make", "-s", "all"], cwd=os.path.join(ROOT, "oracle"))
    orc = ctypes.CDLL(path)
    orc.oracle_sais.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    orc.oracle_verify_sa.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64]
    return orc


def cpu_baseline(text, sample):
    """oracle SA-IS (single thread) on the first `sample` bytes of the same text -- a reported
    baseline only; kind 'port' because the reference's divsufsort cannot be built offline."""
    import numpy as np
    orc = load_oracle()
    t = np.ascontiguousarray(text[:sample])
    out = np.empty(t.size + 1, dtype=np.uint32)
    t0 = time.perf_counter()
    rc = orc.oracle_sais(t.ctypes.data, out.ctypes.data, t.size)
    dt = time.perf_counter() - t0
    assert rc == 0
    return {"value": round(t.size / 1e6 / dt, 3), "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": f"first {t.size} bytes of the workload; oracle_sais = own single-thread SA-IS "
                      f"(stand-in: divsufsort unavailable offline); {dt:.1f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import numpy as np
    import torch
    import suffix_array_amd as sa
    from suffix_array_amd import corpus

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # rehearsal on a one-GPU box: SA_BENCH_SHARE_GPU=1 puts every rank on device 0 and uses gloo for the
    # control plane (RCCL refuses two ranks on one device); the driver's real runs use one GPU per rank + nccl
    share = os.environ.get("SA_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    ctl_dev = dev
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group(backend="gloo")
            ctl_dev = torch.device("cpu")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    # ---- synthetic input, resident in HBM before the timed region ----
    text_h = corpus.workload(args.workload, rank=rank, n_override=args.n)
    n = int(text_h.size)
    text = torch.from_numpy(text_h).to(dev)
    out = torch.empty(n + 1, dtype=torch.int32, device=dev)
    wbytes = sa.workspace_bytes(n)
    work = torch.empty(wbytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    stats = sa.Stats()

    def step():
        sa.saca_device_ptr(text.data_ptr(), out.data_ptr(), n, work.data_ptr(), wbytes, stream, stats)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    L = sa.lib()
    L.sa_amd_profile_begin_classes.argtypes = [ctypes.c_uint64]
    L.sa_amd_profile_begin_classes.restype = None
    names = []
    while True:
        nm = L.sa_amd_profile_kernel_name(len(names)).decode()
        if not nm:
            break
        names.append(nm)
    dom_mask = sum(1 << i for i, nm in enumerate(names) if nm in DOMINANT)
    # timed region: HIP events (on the launch stream) around the launches of the dominant kernel only -- an event pair around
    # each of the ~150 launches of a build costs ~0.7 ms of host time per step; the other kernels are timed below
    L.sa_amd_profile_begin_classes(dom_mask)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    cap = 32
    ms = (ctypes.c_double * cap)()
    launches = (ctypes.c_int64 * cap)()
    units = (ctypes.c_int64 * cap)()
    ncls = L.sa_amd_profile_end(ms, launches, units, cap)
    # per-kernel table: one extra, untimed build with events around every launch
    ms_all = (ctypes.c_double * cap)()
    launches_all = (ctypes.c_int64 * cap)()
    units_all = (ctypes.c_int64 * cap)()
    L.sa_amd_profile_begin()
    step()
    torch.cuda.synchronize()
    L.sa_amd_profile_end(ms_all, launches_all, units_all, cap)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    verified = None
    if args.verify:
        orc = load_oracle()
        got = out.cpu().numpy().view(np.uint32)
        verified = bool(orc.oracle_verify_sa(text_h.ctypes.data, n, got.ctypes.data, n + 1) == 1)

    if args.verify_gpu:
        ci_bytes = 4 * (n + 1) + 256
        ci_work = torch.empty(ci_bytes, dtype=torch.uint8, device=dev)
        L.sa_amd_check_integrity_device.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                                    ctypes.c_int64, ctypes.c_void_p]
        L.sa_amd_check_integrity_device.restype = ctypes.c_int32
        rc = L.sa_amd_check_integrity_device(text.data_ptr(), n, out.data_ptr(), ci_work.data_ptr(), ci_bytes, stream)
        verified = bool(rc == 1) if verified is None else (verified and rc == 1)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    kernels = {}                      # from the extra build (one step)
    for i in range(ncls):
        name = L.sa_amd_profile_kernel_name(i).decode()
        if launches_all[i]:
            kernels[name] = {"ms_per_step": round(ms_all[i], 3), "launches_per_step": float(launches_all[i]),
                             "units_per_step": int(units_all[i])}
    d_ms, d_launch, d_units, dom = max((ms[i], launches[i], units[i], L.sa_amd_profile_kernel_name(i).decode())
                                       for i in range(ncls) if L.sa_amd_profile_kernel_name(i).decode() in DOMINANT)
    avg_ms = d_ms / max(d_launch, 1)
    achieved = (ALGO_BYTES[dom] * d_units) / (d_ms * 1e-3) / 1e9 if d_ms > 0 else 0.0
    # HBM traffic of the dominant kernel from the committed PMC passes of this same command (if any)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("workload") == args.workload and tj.get("n_bytes") == n and dom in tj.get("kernels", {}):
            traffic = tj["kernels"][dom]["hbm_bytes_per_launch"]
    per_step = dt / args.steps
    device_ms = sum(ms_all[i] for i in range(ncls))
    job_gbs = (5 * n + 4) / per_step / 1e9
    result = {
        "metric": "input MB/s indexed (SA build)",
        "value": round(world * n / 1e6 / per_step, 3),
        "unit": "MB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(per_step * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n} bytes per GPU, seed base + rank, one independent text per GPU",
                   "n_bytes": n, "sigma": stats.sigma, "bits_per_symbol": stats.bits_per_symbol,
                   "symbols_per_key": stats.symbols_per_key, "doubling_rounds": stats.rounds,
                   "radix_passes": stats.sort_passes, "unresolved_after_initial_sort": stats.unresolved_after_initial},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "algorithmic_bytes_per_launch": round(ALGO_BYTES[dom] * d_units / max(d_launch, 1)),
                     "algorithmic_bytes_per_element": ALGO_BYTES[dom], "avg_launch_ms": round(avg_ms, 4),
                     "launches_per_step": d_launch / args.steps,
                     "whole_job": {"algorithmic_bytes": 5 * n + 4, "achieved": round(job_gbs, 3),
                                   "frac": round(job_gbs / HBM_PEAK_GBS, 6)}},
        "kernels": kernels,
        "device_ms_per_step": round(device_ms, 3),
        "verified": verified,
    }
    if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
        result["cpu_baseline"] = cpu_baseline(text_h, min(args.cpu_sample, n))
    else:
        result["cpu_baseline"] = None
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
