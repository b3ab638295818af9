#!/usr/bin/env python3
"""bench.py -- suffix-array construction throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one complete SuffixArray::new-equivalent build (device-resident text in, device-resident SA out) of one
synthetic text.  Default workload: the configuration the metric is quoted on, C3 = 256 MiB English-like corpus
(BASELINE.md section 2).  With N ranks every rank builds its own independent text (seed + rank) on its own GPU: weak
scaling, no data-path collective; torch.distributed carries the barrier and the max-over-ranks
time only.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(torch.distributed.run, before this process touches a GPU); under an external torchrun it runs as one rank.

Prints ONE JSON line on rank 0:
  value / ms_per_step   device-resident builds, whole job (all ranks), timed as the contract says
  roofline              the kernel class with the MOST device time in the timed region (HIP events on the launch stream around
                        every class that had >= 5 % of the device time of a profiled build just before it), roofline.kernels:
                        the same figures for every class with >= 10 %, roofline.whole_build: PMC bytes / device time
  kernels               per-kernel table from one extra build outside the timed region (every class)
  verified              sa_amd_check_integrity_device (reference src/sa.rs:72-84, linear time) on the last array of
                        EVERY rank, outside the timed region -- always on
  end_to_end            median of 5 sa_amd_saca_u8 calls on host buffers (what the reference's `SuffixArray::new`
                        hands over: src/sa.rs:23-27): H2D + build + D2H, all ranks concurrently
  configs               (N = 1) the other BASELINE.json configs behind the headline workload, same measurement each: C2 64 MiB uniform,
                        north_star's literal 256 MiB random-byte text, C4 1 GiB DNA, C5's 512 MiB per-GPU text -- device-resident
                        ms per build (1 warm-up + --config-steps timed), MB/s, verified, the class with the most device time
                        against the HBM roofline (HIP events inside the timed region), (5n + 4) / t, host-pointer calls
  ranks                 every rank's HIP device, PCI bus id, own ms per step and own verification (N > 1: the bus ids must be
                        distinct unless SA_BENCH_SHARE_GPU=1 asks for a rehearsal on one GPU)
  batch_c5              (N > 1) BASELINE config 5: one 512 MiB uniform text per rank, device-resident and end to end
  batch_api             (N = 1) BASELINE config 5 through its C-ABI form: ONE call of sa_amd_saca_batch with 8 texts of 512 MiB
                        over all visible devices, host pointers in and out (+ small_texts: 4 096 texts of 4 KiB through one
                        call of the same entry point, the reference's own size domain)
  cpu_baseline          CPU suffix sorter on a bounded sample, one pinned core (N = 1 only): libdivsufsort if a probe finds
                        one (kind "reference"), else the oracle's own SA-IS (kind "port"); the probe log is in the object
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes per unit (element) of each kernel class, DESIGN.md section 3.  k_group_sort: 12 B of list in (suffix,
# group head, slot) + 4 B of rank (or the text symbols) gathered + 13 B out (key, suffix, flag) per tied suffix;
# k_finish_sorted: the sorted 32-bit key + the suffix of every slot (the few tied slots also gather text and rewrite SA)
ALGO_BYTES = {"k_byte_hist": 1, "k_build_keys": 9, "k_radix_upsweep": 8, "k_radix_downsweep": 24,
              "k_rr_count": 8, "k_rr_apply": 24, "k_gather_key2": 20, "k_scatter_pairs": 16,
              "k_radix_upsweep32": 4, "k_radix_downsweep32": 16, "k_onesweep": 24, "k_onesweep32": 16,
              "k_group_sort": 29, "k_finish_sorted": 8, "k_bucket_sort": 16}
EVENT_SHARE = 0.05             # classes with at least this share of a build's device time get events in the timed region
REPORT_SHARE = 0.10            # ... and at least this share are listed in roofline.kernels


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3_english_256m",
                    help="c2_uniform_64m | c2_uniform_256m | c3_english_256m | c3_iid_256m | c4_dna_1g | c4_dna_repeats_1g | c5_uniform_512m")
    ap.add_argument("--text-bytes", "--n", dest="n", type=int, default=None, help="override the text length (bytes)")
    ap.add_argument("--verify-cpu", action="store_true", help="additionally check the last SA with the oracle's linear verifier")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="N > 1: skip the config-5 batch leg")
    ap.add_argument("--no-batch-api", action="store_true", help="N = 1: skip the sa_amd_saca_batch leg (8 x 512 MiB, host pointers)")
    ap.add_argument("--batch-texts", type=int, default=8)
    ap.add_argument("--small-batch-texts", type=int, default=4096, help="texts of 4 KiB in the small-text part of the sa_amd_saca_batch leg (0: skip)")
    ap.add_argument("--cpu-sample", type=int, default=256 << 20,
                    help="bytes of the workload timed on the CPU (default: the whole 256 MiB headline text, ~30 s of the stand-in)")
    ap.add_argument("--batch-api-only", action="store_true", help="(internal) run only the sa_amd_saca_batch leg and print its JSON object")
    ap.add_argument("--batch-api-inline", action="store_true", help="run the sa_amd_saca_batch leg in this process even when several GPUs are visible")
    ap.add_argument("--batch-api-timeout", type=int, default=300)
    ap.add_argument("--configs", default="c2_uniform_64m,c2_uniform_256m,c4_dna_1g,c5_uniform_512m",
                    help="N = 1: the other BASELINE configs measured behind the headline workload (comma list; '' or --no-configs: none)")
    ap.add_argument("--no-configs", action="store_true")
    ap.add_argument("--config-steps", type=int, default=5, help="timed builds per extra config (after 1 warm-up)")
    ap.add_argument("--e2e-calls", type=int, default=5)
    return ap.parse_args(argv)


# ---- rank launcher ---------------------------------------------------------------------------------

def spawn_ranks(args, argv):
    """--gpus N without an external launcher: start N ranks with torch.distributed.run as a CHILD process (this
    process has not touched the GPU and never does) and pass its output and exit code through."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # (torch.distributed.run's own parser would read a bare `--n` as an abbreviation of its --nnodes / --nproc-per-node)
    argv = ["--text-bytes" if a == "--n" else a for a in argv]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ---- CPU side --------------------------------------------------------------------------------------

def load_oracle():
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "all"], cwd=os.path.join(ROOT, "oracle"))
    orc = ctypes.CDLL(path)
    orc.oracle_sais.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    orc.oracle_verify_sa_mt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]
    return orc


def _pin_one_core():
    """pin this process to one of its CPUs (the engines timed here are single-threaded); returns (cpu, restore())"""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        cpu = allowed[len(allowed) // 2]
        os.sched_setaffinity(0, {cpu})
        return cpu, (lambda: os.sched_setaffinity(0, set(allowed)))
    except (AttributeError, OSError):
        return None, (lambda: None)


def _probe_divsufsort(probe):
    """SURVEY.md section 8d / BASELINE.md section 2: the crate's CPU engine is libdivsufsort (cdivsufsort = "2.0", reference
    Cargo.toml:17, called at src/saca.rs:14).  Probe order: cargo + a resolvable cdivsufsort crate -> a system libdivsufsort
    -> the Python package pydivsufsort -> none.  Returns a callable (text: np.uint8 array) -> seconds, or None."""
    import ctypes.util
    import shutil
    import numpy as np
    cargo = shutil.which("cargo")
    if cargo is None:
        probe.append({"step": "cargo + cdivsufsort crate", "result": "cargo not on PATH"})
    else:
        # an offline build of a ten-line timing binary against the crate's own dependency
        work = os.path.join(ROOT, "oracle", "_ref", "rust_probe")
        try:
            os.makedirs(os.path.join(work, "src"), exist_ok=True)
            open(os.path.join(work, "Cargo.toml"), "w").write(
                '[package]\nname = "dss_probe"\nversion = "0.0.0"\nedition = "2018"\n[dependencies]\ncdivsufsort = "2.0"\n')
            open(os.path.join(work, "src", "main.rs"), "w").write(
                "use std::io::Read;\nfn main() {\n    let mut t = Vec::new();\n    std::io::stdin().read_to_end(&mut t).unwrap();\n"
                "    let mut sa = vec![0i32; t.len()];\n    let t0 = std::time::Instant::now();\n"
                "    cdivsufsort::sort_in_place(&t, &mut sa);\n    println!(\"{}\", t0.elapsed().as_secs_f64());\n}\n")
            r = subprocess.run([cargo, "build", "--offline", "--release"], cwd=work, capture_output=True, text=True, timeout=300)
            if r.returncode == 0:
                exe = os.path.join(work, "target", "release", "dss_probe")
                probe.append({"step": "cargo + cdivsufsort crate", "result": "built"})

                def run_cargo(t):
                    out = subprocess.run([exe], input=t.tobytes(), capture_output=True, timeout=3600)
                    return float(out.stdout.decode().strip())
                return run_cargo, "cdivsufsort crate (cargo, offline registry)"
            probe.append({"step": "cargo + cdivsufsort crate", "result": "cargo build --offline failed: " + r.stderr.strip().splitlines()[-1][:160]})
        except Exception as ex:                                   # no registry, no compiler, time-out ...
            probe.append({"step": "cargo + cdivsufsort crate", "result": f"{type(ex).__name__}: {ex}"[:200]})
    name = ctypes.util.find_library("divsufsort")
    if name is None:
        probe.append({"step": "system libdivsufsort (ctypes.util.find_library)", "result": "not found"})
    else:
        try:
            lib = ctypes.CDLL(name)
            lib.divsufsort.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
            lib.divsufsort.restype = ctypes.c_int32
            probe.append({"step": "system libdivsufsort (ctypes.util.find_library)", "result": name})

            def run_lib(t):
                out = np.empty(t.size, dtype=np.int32)
                t0 = time.perf_counter()
                rc = lib.divsufsort(t.ctypes.data, out.ctypes.data, t.size)
                dt = time.perf_counter() - t0
                assert rc == 0
                return dt
            return run_lib, f"system {name}"
        except (OSError, AttributeError) as ex:
            probe.append({"step": "system libdivsufsort (ctypes.util.find_library)", "result": f"{name}: {ex}"[:200]})
    try:
        import pydivsufsort

        def run_py(t):
            t0 = time.perf_counter()
            pydivsufsort.divsufsort(t)
            return time.perf_counter() - t0
        probe.append({"step": "python package pydivsufsort", "result": "imported"})
        return run_py, "pydivsufsort"
    except ImportError:
        probe.append({"step": "python package pydivsufsort", "result": "not installed"})
    return None, None


def cpu_baseline(text, sample):
    """The CPU path timed beside the GPU one, on the first `sample` bytes of the same text, one pinned core.  kind
    "reference" only if a real libdivsufsort was found by the probe; otherwise the oracle's own single-thread SA-IS (kind
    "port": a stand-in, probably several times slower than divsufsort -- a reported baseline, not a target)."""
    import numpy as np
    probe = []
    t = np.ascontiguousarray(text[:sample])
    cpu, restore = _pin_one_core()
    try:
        run, what = _probe_divsufsort(probe)
        if run is not None:
            dt = run(t)
            kind, engine = "reference", what
        else:
            orc = load_oracle()
            out = np.empty(t.size + 1, dtype=np.uint32)
            t0 = time.perf_counter()
            rc = orc.oracle_sais(t.ctypes.data, out.ctypes.data, t.size)
            dt = time.perf_counter() - t0
            assert rc == 0
            probe.append({"step": "stand-in: oracle_sais (own single-thread SA-IS)", "result": "ran"})
            kind, engine = "port", "oracle_sais = own single-thread SA-IS (stand-in: divsufsort unavailable offline)"
    finally:
        restore()
    what = "the whole workload" if t.size == text.size else f"first {t.size} of the workload's {text.size} bytes"
    return {"value": round(t.size / 1e6 / dt, 3), "unit": "MB/s", "cores": 1, "kind": kind, "pinned_cpu": cpu, "probe": probe,
            "sample": f"{what} ({t.size} bytes); {engine}; {dt:.1f} s"}


def host_info():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"nproc": os.cpu_count(), "cpu_model": model}


def corpus_facts(workload, n, text):
    """mean / max LCP of the benchmark text (profiles/r02_corpus_stats.json: oracle SA-IS + Kasai on the full text,
    tools/corpus_stats.py) and its order-0 entropy, computed here"""
    import numpy as np
    c = np.bincount(text, minlength=256)
    p = c[c > 0] / max(text.size, 1)
    facts = {"H0_bits_per_byte": round(float(-(p * np.log2(p)).sum()), 4), "mean_lcp": None, "max_lcp": None}
    path = os.path.join(ROOT, "profiles", "r02_corpus_stats.json")
    if os.path.exists(path):
        for row in json.load(open(path)):
            if row.get("workload") == workload and row.get("n_bytes") == n:
                facts["mean_lcp"], facts["max_lcp"] = row["mean_lcp"], row["max_lcp"]
                facts["frac_lcp_ge_32"] = row.get("frac_lcp_ge", {}).get("32")
    return facts


# ---- the GPU backend: everything that touches the device goes through the C ABI ----------------------

class HipBackend:
    """device memory and streams from torch (plumbing), construction through libsuffix_array_amd.so"""

    def __init__(self, device_index):
        import torch
        import suffix_array_amd as sa
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
        self.torch, self.sa = torch, sa
        torch.cuda.set_device(device_index)
        self.dev = torch.device("cuda", device_index)
        self.L = sa.lib()
        self.stats = sa.Stats()
        self.name = "hip"

    def load(self, text_h):
        torch, sa = self.torch, self.sa
        self.n = int(text_h.size)
        self.text = torch.from_numpy(text_h).to(self.dev)
        self.out = torch.empty(self.n + 1, dtype=torch.int32, device=self.dev)
        self.wbytes = sa.workspace_bytes(self.n)
        self.work = torch.empty(self.wbytes, dtype=torch.uint8, device=self.dev)
        self.stream = torch.cuda.current_stream().cuda_stream

    def unload(self):
        self.text = self.out = self.work = None
        self.torch.cuda.empty_cache()

    def step(self):
        self.sa.saca_device_ptr(self.text.data_ptr(), self.out.data_ptr(), self.n, self.work.data_ptr(), self.wbytes,
                                self.stream, self.stats)

    def sync(self):
        self.torch.cuda.synchronize()

    def kernel_names(self):
        names = []
        while True:
            nm = self.L.sa_amd_profile_kernel_name(len(names)).decode()
            if not nm:
                return names
            names.append(nm)

    def profile_begin(self, mask):
        self.L.sa_amd_profile_begin_classes(mask)

    def profile_end(self):
        cap = 32
        ms, launches, units = (ctypes.c_double * cap)(), (ctypes.c_int64 * cap)(), (ctypes.c_int64 * cap)()
        ncls = self.L.sa_amd_profile_end(ms, launches, units, cap)
        return [(ms[i], launches[i], units[i]) for i in range(ncls)]

    def verify(self):
        """reference src/sa.rs:72-84 in linear time on the device (k_ci_scatter / k_ci_check), on the array the last
        step left in HBM; outside the timed region"""
        ci_bytes = int(self.L.sa_amd_check_integrity_work_bytes(self.n))       # (the streaming form of the check)
        ci_work = self.torch.empty(ci_bytes, dtype=self.torch.uint8, device=self.dev)
        rc = self.L.sa_amd_check_integrity_device(self.text.data_ptr(), self.n, self.out.data_ptr(), ci_work.data_ptr(),
                                                  ci_bytes, self.stream)
        return rc == 1

    def download(self):
        import numpy as np
        return self.out.cpu().numpy().view(np.uint32)

    def build_host(self, text_h, out_h):
        self.sa.saca(text_h, out_h)

    def device_count(self):
        return int(self.L.sa_amd_device_count())

    def build_batch(self, texts, outs):
        """ONE call of sa_amd_saca_batch (device = NULL: text i -> device i mod the number of visible devices)"""
        k = len(texts)
        T = (ctypes.c_void_p * k)(*[t.ctypes.data for t in texts])
        S = (ctypes.c_void_p * k)(*[o.ctypes.data for o in outs])
        N = (ctypes.c_int32 * k)(*[int(t.size) for t in texts])
        st = (ctypes.c_int32 * k)()
        rc = self.L.sa_amd_saca_batch(T, S, N, None, k, st)
        return rc, list(st)

    def prepare_batch(self, texts, outs):
        """the pointer arrays of a batch call, built once (thousands of small texts: ctypes spends ~2 us per text on them)"""
        k = len(texts)
        return (k, (ctypes.c_void_p * k)(*[t.ctypes.data for t in texts]), (ctypes.c_void_p * k)(*[o.ctypes.data for o in outs]),
                (ctypes.c_int32 * k)(*[int(t.size) for t in texts]), (ctypes.c_int32 * k)())

    def run_batch(self, prep):
        k, T, S, N, st = prep
        rc = self.L.sa_amd_saca_batch(T, S, N, None, k, st)
        return rc, list(st)

    def check_host(self, text_h, sa_h):
        """sa_amd_check_integrity on host arrays (reference src/sa.rs:72-84, linear time on the device)"""
        return self.L.sa_amd_check_integrity(text_h.ctypes.data, int(text_h.size), sa_h.ctypes.data, int(sa_h.size)) == 1

    def host_timing(self):
        return self.sa.last_host_timing()

    def stats_dict(self):
        return self.stats.as_dict()

    def identity(self):
        """which physical GPU this rank runs on (the N > 1 line lists it per rank)"""
        idx = self.dev.index
        try:
            bus = self.sa.device_pci_bus_id(idx)
        except Exception as ex:                                   # (reported, not fatal: the line then says the ids are unknown)
            bus = None
            print(f"bench.py: rank device {idx}: no PCI bus id ({ex})", file=sys.stderr)
        return {"hip_device": idx, "pci_bus_id": bus, "gpu": self.torch.cuda.get_device_name(idx), "pid": os.getpid()}

    def ctl_device(self, share):
        return self.torch.device("cpu") if share else self.dev


def timed_steps(backend, barrier, steps, warmup, mask=0):
    for _ in range(warmup):
        backend.step()
    backend.profile_begin(mask)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        backend.step()
    barrier()
    dt = time.perf_counter() - t0
    return dt, backend.profile_end()


def end_to_end(backend, text_h, calls, barrier):
    """host pointers in, host pointers out, as `saca()` is called by SuffixArray::new / ::set: one warm-up, then `calls`
    timed calls into a REUSED output buffer (::set, src/sa.rs:30-33) and `calls` into a FRESH zeroed one each
    (::new, src/sa.rs:23-27: vec![0; n + 1] is first touched by the download)"""
    import numpy as np
    n = int(text_h.size)
    out = np.zeros(n + 1, dtype=np.uint32)
    backend.build_host(text_h, out)
    res = {}
    for mode in ("reused_buffer", "fresh_buffer"):
        times, phases = [], None
        for _ in range(calls):
            if mode == "fresh_buffer":
                out = np.zeros(n + 1, dtype=np.uint32)
            barrier()
            t0 = time.perf_counter()
            backend.build_host(text_h, out)
            times.append(time.perf_counter() - t0)
            phases = backend.host_timing()
        med = statistics.median(times)
        res[mode] = {"ms": round(med * 1e3, 3), "MB_per_s": round(n / 1e6 / med, 1),
                     "phases_ms_last_call": {k: round(v, 2) for k, v in (phases or {}).items()}}
    res["ok"] = bool(out[0] == n)
    return res


def batch_api_leg(backend, args):
    """BASELINE config 5 in its C-ABI form (reference analogue: a loop over SuffixArray::new,
    src/sa.rs:23-27): ONE process, ONE call of sa_amd_saca_batch with `--batch-texts` independent 512 MiB texts (seeds 50 +
    i), host pointers in and out, text i on device i mod the number of visible devices.  On a one-GPU box all texts go
    through device 0 (two host threads overlap one text's copies with the next one's build): a rehearsal of the call the
    8-GPU run makes, not a scaling figure."""
    import numpy as np
    from suffix_array_amd import corpus
    k = max(int(args.batch_texts), 1)
    texts = [corpus.workload("c5_uniform_512m", rank=i, n_override=args.n) for i in range(k)]
    outs = [np.zeros(t.size + 1, dtype=np.uint32) for t in texts]
    times = []
    rc, st = 0, []
    for _ in range(2):                                        # first call: pool warm-up (device blocks, pinned staging)
        t0 = time.perf_counter()
        rc, st = backend.build_batch(texts, outs)
        times.append(time.perf_counter() - t0)
    ok = rc == 0 and all(x == 0 for x in st) and all(int(o[0]) == t.size for t, o in zip(texts, outs))
    ok = ok and all(backend.check_host(t, o) for t, o in zip(texts, outs))
    total = sum(int(t.size) for t in texts)
    res = {"entry_point": "sa_amd_saca_batch", "texts": k, "bytes_each": int(texts[0].size), "devices": backend.device_count(),
           "ms_first_call": round(times[0] * 1e3, 1), "ms": round(times[1] * 1e3, 1),
           "MB_per_s": round(total / 1e6 / times[1], 1), "verified": bool(ok)}
    ks = int(getattr(args, "small_batch_texts", 0) or 0)
    if ks > 0:
        # the same entry point on the reference's own size domain (src/tests.rs:13-17: texts below 4 KiB): MANY small texts in one
        # call -- the library builds them together, one workgroup per text, one launch per chunk (kernels/small.hpp)
        del texts, outs
        small = [corpus.english(4096, 9000 + i) for i in range(ks)]
        souts = [np.zeros(4097, dtype=np.uint32) for _ in small]
        stimes = []
        prep = backend.prepare_batch(small, souts) if hasattr(backend, "prepare_batch") else None
        for _ in range(3):
            t0 = time.perf_counter()
            rc, st = backend.run_batch(prep) if prep is not None else backend.build_batch(small, souts)
            stimes.append(time.perf_counter() - t0)
        oks = rc == 0 and all(x == 0 for x in st) and all(int(o[0]) == 4096 for o in souts)
        oks = oks and all(backend.check_host(small[i], souts[i]) for i in range(0, ks, max(ks // 16, 1)))
        best = min(stimes[1:])
        res["small_texts"] = {"texts": ks, "bytes_each": 4096, "ms": round(best * 1e3, 3), "us_per_text": round(best * 1e6 / ks, 3),
                              "MB_per_s": round(ks * 4096 / 1e6 / best, 1), "verified": bool(oks),
                              "what": "one sa_amd_saca_batch call over all of them (host pointers in and out; the pointer arrays are built before the clock starts)"}
    return res


def load_traffic(workload, n):
    """HBM traffic per kernel class from the committed PMC passes of this same command (builder-run: tools/profile_round.sh ->
    tools/pmc_traffic.py -> profiles/traffic.json); None when the file is for another workload"""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None
    tj = json.load(open(tpath))
    if tj.get("workload") != workload or tj.get("n_bytes") != n:
        return None
    return tj


def roofline_lines(names, prof, steps, tj=None):
    """one line per kernel class timed inside a timed region (HIP events on the launch stream), most device time first:
    ms per step, algorithmic bytes per element and launch (ALGO_BYTES x units), achieved GB/s, fraction of the HBM peak and --
    where profiles/traffic.json has the class -- the PMC-measured bytes per launch over the algorithmic ones"""
    timed_ms = sum(ms for ms, _, _ in prof)
    lines = []
    for i in sorted((i for i in range(len(prof)) if prof[i][1]), key=lambda i: -prof[i][0]):
        ms, launches, units = prof[i]
        nm = names[i]
        ab = ALGO_BYTES.get(nm)
        line = {"name": nm, "ms_per_step": round(ms / max(steps, 1), 3), "launches_per_step": launches / max(steps, 1),
                "share_of_timed_classes": round(ms / timed_ms, 4) if timed_ms > 0 else None,
                "algorithmic_bytes_per_element": ab, "avg_launch_ms": round(ms / max(launches, 1), 4)}
        if ab and ms > 0:
            line["algorithmic_bytes_per_launch"] = round(ab * units / max(launches, 1))
            line["achieved"] = round(ab * units / (ms * 1e-3) / 1e9, 2)
            line["frac"] = round(line["achieved"] / HBM_PEAK_GBS, 5)
        if tj and nm in tj.get("kernels", {}):
            line["traffic"] = tj["kernels"][nm]["hbm_bytes_per_launch"]
            if ab and units:
                line["traffic_over_algorithmic"] = round(line["traffic"] / (ab * units / max(launches, 1)), 3)
            # the rate the kernel moves FETCHED bytes at: a kernel of random 4-byte look-ups sits far below the roofline on
            # algorithmic bytes and AT it on these -- every look-up is a 128-byte request on gfx950 (tools/gather_probe.hip,
            # profiles/r03_gather_probe.txt)
            if ms > 0 and launches:
                line["traffic_rate"] = round(line["traffic"] / (ms / launches * 1e-3) / 1e9, 1)
                line["traffic_frac"] = round(line["traffic_rate"] / HBM_PEAK_GBS, 4)
        lines.append(line)
    return lines


def measure_resident(backend, barrier, steps, warmup, bcast_mask=None):
    """the text is loaded: first touch, one build with events around every launch (which classes matter), then `warmup` untimed
    and `steps` timed builds with events around the classes that had >= EVENT_SHARE of the device time.  Returns
    (seconds for the timed steps on this rank, per-class rows of the timed region, rows of the profiled build, its device ms)."""
    backend.step()                                            # (first touch of the workspace, pool, pinned read-back buffer)
    backend.profile_begin(~0 & 0xFFFFFFFFFFFFFFFF)
    backend.step()
    backend.sync()
    prof_all = backend.profile_end()
    device_ms = sum(ms for ms, _, _ in prof_all)
    ev_mask = sum(1 << i for i, (ms, _, _) in enumerate(prof_all) if device_ms > 0 and ms >= EVENT_SHARE * device_ms)
    if bcast_mask is not None:                                # (every rank times the same classes: rank 0 decides)
        ev_mask = bcast_mask(ev_mask)
    dt, prof = timed_steps(backend, barrier, steps, warmup, ev_mask)
    return dt, prof, prof_all, device_ms


def config_leg(backend, args, workload, barrier, names):
    """one of the other BASELINE.json configs behind the headline workload (N = 1): the same measurement -- device-resident
    builds timed as the contract says, verified on the device, the class with the most device time against the HBM roofline
    from HIP events inside the timed region, (5n + 4) / t, and the host-pointer call -- at `--config-steps` timed builds"""
    from suffix_array_amd import corpus
    t_gen = time.perf_counter()
    text_h = corpus.workload(workload, rank=0, n_override=args.n)
    t_gen = time.perf_counter() - t_gen
    n = int(text_h.size)
    backend.load(text_h)
    steps = max(int(args.config_steps), 1)
    dt, prof, prof_all, device_ms = measure_resident(backend, barrier, steps, 1)
    stats = backend.stats_dict()
    ok = bool(backend.verify())
    per = dt / steps
    lines = roofline_lines(names, prof, steps)
    dom = lines[0] if lines else {"name": None}
    res = {"workload": workload, "n_bytes": n, "steps": steps, "warmup": 1, "ms_per_build": round(per * 1e3, 3),
           "MB_per_s": round(n / 1e6 / per, 1), "device_ms_per_build": round(device_ms, 3), "verified": ok,
           "sigma": stats["sigma"], "refinement_rounds": stats["rounds"], "radix_passes": stats["sort_passes"],
           "roofline": {"kernel": dom.get("name"), "achieved": dom.get("achieved"), "frac": dom.get("frac"), "unit": "GB/s",
                        "avg_launch_ms": dom.get("avg_launch_ms"), "launches_per_step": dom.get("launches_per_step"),
                        "algorithmic_bytes_per_element": dom.get("algorithmic_bytes_per_element"),
                        "algorithmic_bytes_per_launch": dom.get("algorithmic_bytes_per_launch"),
                        "kernels": [{k: ln.get(k) for k in ("name", "ms_per_step", "launches_per_step", "avg_launch_ms", "achieved", "frac")}
                                    for ln in lines if device_ms > 0 and ln["ms_per_step"] >= REPORT_SHARE * device_ms]},
           "whole_job": {"algorithmic_bytes": 5 * n + 4, "achieved": round((5 * n + 4) / per / 1e9, 3),
                         "frac": round((5 * n + 4) / per / 1e9 / HBM_PEAK_GBS, 6)},
           "corpus_generation_s": round(t_gen, 2)}
    if not args.no_end_to_end:
        e = end_to_end(backend, text_h, max(min(args.e2e_calls, 3), 1), barrier)
        res["end_to_end"] = {m: {"ms": e[m]["ms"], "MB_per_s": e[m]["MB_per_s"], "phases_ms_last_call": e[m]["phases_ms_last_call"]}
                             for m in ("reused_buffer", "fresh_buffer")}
    backend.unload()
    return res


def batch_api_subprocess(args):
    """batch_api_leg in a child process (`bench.py --batch-api-only`, its own GPU context) with a time limit"""
    cmd = [sys.executable, os.path.abspath(__file__), "--batch-api-only", "--batch-texts", str(args.batch_texts),
           "--small-batch-texts", str(args.small_batch_texts)]
    if args.n is not None:
        cmd += ["--text-bytes", str(args.n)]
    try:
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=args.batch_api_timeout)
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
        if proc.returncode == 0 and lines:
            out = json.loads(lines[-1])
            out["ran_in"] = "child process (more than one GPU visible)"
            return out
        return {"entry_point": "sa_amd_saca_batch", "verified": False, "ran_in": "child process",
                "error": f"exit code {proc.returncode}: {proc.stderr.strip()[-300:]}"}
    except subprocess.TimeoutExpired:
        return {"entry_point": "sa_amd_saca_batch", "verified": False, "ran_in": "child process", "error": f"no answer within {args.batch_api_timeout} s"}
    except Exception as ex:
        return {"entry_point": "sa_amd_saca_batch", "verified": False, "ran_in": "child process", "error": f"{type(ex).__name__}: {ex}"[:300]}


def run(args, backend, rank, world, dist=None, share=False):
    """one rank of the benchmark; returns the result dict on rank 0 (None elsewhere).  `backend` is the HipBackend; the
    world-size-2 CPU test of tests/test_dist.py injects its own object with the same methods."""
    import numpy as np
    from suffix_array_amd import corpus

    def barrier():
        backend.sync()
        if dist is not None:
            dist.barrier()
        backend.sync()

    def reduce(value, op):
        if dist is None:
            return value
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=backend.ctl_device(share))
        dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
        return float(t.item())

    def bcast_mask(mask):
        if dist is None:
            return mask
        import torch
        tm = torch.tensor([mask], dtype=torch.int64, device=backend.ctl_device(share))
        dist.broadcast(tm, src=0)
        return int(tm.item())

    # ---- synthetic input, resident in HBM before the timed region ----
    text_h = corpus.workload(args.workload, rank=rank, n_override=args.n)
    n = int(text_h.size)
    backend.load(text_h)
    names = backend.kernel_names()
    # per-kernel table first: one untimed build with events around every launch.  It also says which classes matter: the
    # timed region then carries events around every class with at least EVENT_SHARE of the device time (an event pair
    # around each of the ~200 launches of a build would cost ~0.7 ms of host time per step)
    dt_rank, prof, prof_all, device_ms = measure_resident(backend, barrier, args.steps, args.warmup, bcast_mask)
    dt = reduce(dt_rank, "MAX")
    stats = backend.stats_dict()
    # correctness gate of every run, outside the timed region: every rank's last array
    ok = bool(backend.verify())
    if args.verify_cpu:
        orc = load_oracle()
        got = backend.download()
        ok = ok and orc.oracle_verify_sa_mt(text_h.ctypes.data, n, got.ctypes.data, n + 1, 16) == 1
    verified = reduce(1.0 if ok else 0.0, "MIN") == 1.0

    # who ran where: every rank's device identity, its own time and its own check (the driver's N > 1 runs: N distinct GPUs)
    ident = dict(backend.identity()) if hasattr(backend, "identity") else {}
    mine = {"rank": rank, **ident, "ms_per_step": round(dt_rank / max(args.steps, 1) * 1e3, 3), "verified": ok}
    ranks = [mine]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    bus_ids = [r.get("pci_bus_id") for r in ranks]
    distinct = len(set(bus_ids)) == len(bus_ids) and None not in bus_ids
    if world > 1 and not distinct and os.environ.get("SA_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"bench.py: {world} ranks but their GPUs are not distinct ({bus_ids}); SA_BENCH_SHARE_GPU=1 allows a rehearsal")

    e2e = None
    if not args.no_end_to_end and world == 1:
        try:                                                  # (N = 1: a failure here is recorded; at N > 1 the ranks must stay in step, so it is not caught)
            e = end_to_end(backend, text_h, args.e2e_calls, barrier)
        except Exception as ex:
            e, e2e = None, {"error": f"{type(ex).__name__}: {ex}"[:300]}
    elif not args.no_end_to_end:
        e = end_to_end(backend, text_h, args.e2e_calls, barrier)
    else:
        e = None
    if e is not None:
        # whole job: all ranks run their calls concurrently (shared PCIe switches / host DRAM are part of the figure)
        e2e = {}
        for mode in ("reused_buffer", "fresh_buffer"):
            ms = reduce(e[mode]["ms"], "MAX")
            e2e[mode] = {"ms": round(ms, 3), "MB_per_s": round(world * n / 1e3 / ms, 1),
                         "phases_ms_rank0_last_call": e[mode]["phases_ms_last_call"]}
        e2e["calls"] = args.e2e_calls
        e2e["what"] = "median wall time of sa_amd_saca_u8 (host text in, host SA out: n bytes up, 4(n+1) bytes down), max over ranks"

    batch = None
    if world > 1 and not args.no_batch:
        # BASELINE config 5: one independent 512 MiB uniform text per rank (seeds 50 + rank), no collective
        backend.unload()
        t5 = corpus.workload("c5_uniform_512m", rank=rank, n_override=args.n)
        backend.load(t5)
        k5 = max(args.steps // 2, 2)
        dt5, _ = timed_steps(backend, barrier, k5, 1)
        dt5 = reduce(dt5, "MAX")
        ok5 = reduce(1.0 if backend.verify() else 0.0, "MIN") == 1.0
        e5 = end_to_end(backend, t5, 3, barrier)
        ms5 = reduce(e5["reused_buffer"]["ms"], "MAX")
        batch = {"workload": f"c5_uniform_512m: {t5.size} bytes per GPU, seeds 50 + rank", "texts": world,
                 "device_resident": {"ms_per_text": round(dt5 / k5 * 1e3, 3), "MB_per_s": round(world * t5.size / 1e6 / (dt5 / k5), 1)},
                 "end_to_end": {"ms_per_text": round(ms5, 3), "MB_per_s": round(world * t5.size / 1e3 / ms5, 1)},
                 "verified": ok5}

    # the other BASELINE configs, same measurement (N = 1: they are single-GPU configs; C5's per-GPU text is one of them)
    configs = None
    cfg_names = [c for c in (args.configs or "").split(",") if c and c != args.workload]
    if world == 1 and not args.no_configs and cfg_names:
        backend.unload()
        t_cfg = time.perf_counter()
        configs = {}
        for c in cfg_names:
            # (a leg behind the headline must not take the line down with it: a failure is recorded, the headline stands)
            try:
                configs[c] = config_leg(backend, args, c, barrier, names)
            except Exception as ex:
                configs[c] = {"workload": c, "verified": False, "error": f"{type(ex).__name__}: {ex}"[:300]}
                try:
                    backend.unload()
                except Exception:
                    pass
        configs["_seconds"] = round(time.perf_counter() - t_cfg, 1)

    batch_api = None
    if world == 1 and not args.no_batch_api and not args.no_end_to_end:
        backend.unload()
        try:
            ndev = backend.device_count()
        except Exception:
            ndev = 1
        if ndev > 1 and not args.batch_api_inline:
            # several GPUs are visible to this one-rank run (the driver's N = 1 run on a whole node): sa_amd_saca_batch then drives
            # them ALL from one process -- a path no builder session (one GPU) has ever executed.  It runs in a child process with a
            # time limit, so that whatever it does the headline line above is still printed; the child's JSON is passed through.
            batch_api = batch_api_subprocess(args)
        else:
            try:
                batch_api = batch_api_leg(backend, args)
            except Exception as ex:
                batch_api = {"entry_point": "sa_amd_saca_batch", "verified": False, "error": f"{type(ex).__name__}: {ex}"[:300]}

    if rank != 0:
        return None

    kernels = {}
    for i, (ms, launches, units) in enumerate(prof_all):
        if launches:
            kernels[names[i]] = {"ms_per_step": round(ms, 3), "launches_per_step": float(launches), "units_per_step": int(units)}
    tj = load_traffic(args.workload, n)
    lines = roofline_lines(names, prof, args.steps, tj)
    dom = lines[0] if lines else {"name": None, "achieved": 0.0, "frac": 0.0}
    reported = [ln for ln in lines if device_ms > 0 and ln["ms_per_step"] >= REPORT_SHARE * device_ms]
    whole_build = None
    if tj and tj.get("builds_per_pass"):
        # (the verification kernels k_ci_* run once per bench run, outside the builds)
        pmc_bytes = sum(k.get("hbm_bytes_per_build", 0) for nm, k in tj["kernels"].items() if not nm.startswith("k_ci_"))
        whole_build = {"pmc_hbm_bytes_per_build": pmc_bytes, "device_ms_per_build": round(device_ms, 3),
                       "achieved": round(pmc_bytes / (device_ms * 1e-3) / 1e9, 1) if device_ms > 0 else None,
                       "frac": round(pmc_bytes / (device_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if device_ms > 0 else None,
                       "over_compulsory": round(pmc_bytes / (5 * n + 4), 1),
                       "what": "sum of the PMC-measured HBM bytes of every kernel of one build (profiles/traffic.json) over the "
                               "sum of the kernels' device time: the rate the whole build moves bytes at"}
    per_step = dt / args.steps
    job_gbs = (5 * n + 4) / per_step / 1e9
    facts = corpus_facts(args.workload, n, text_h)
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
                   "n_bytes": n, "sigma": stats["sigma"], "bits_per_symbol": stats["bits_per_symbol"],
                   "symbols_per_key": stats["symbols_per_key"], "refinement_rounds": stats["rounds"],
                   "text_rounds": stats["text_rounds"], "radix_passes": stats["sort_passes"],
                   "unresolved_after_initial_sort": stats["unresolved_after_initial"], **facts},
        "roofline": {"bound": "hbm", "kernel": dom["name"], "achieved": dom.get("achieved", 0.0), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": dom.get("frac", 0.0), "traffic": dom.get("traffic"),
                     "traffic_source": "profiles/traffic.json (builder-run PMC passes of this command, tools/profile_round.sh); "
                                       "not measured in this run" if tj else None,
                     "algorithmic_bytes_per_launch": dom.get("algorithmic_bytes_per_launch"),
                     "algorithmic_bytes_per_element": dom.get("algorithmic_bytes_per_element"),
                     "avg_launch_ms": dom.get("avg_launch_ms"), "launches_per_step": dom.get("launches_per_step"),
                     "traffic_over_algorithmic": dom.get("traffic_over_algorithmic"), "traffic_frac": dom.get("traffic_frac"),
                     "chosen": "the class with the most device time among the classes timed inside the timed region "
                               f"(every class with >= {EVENT_SHARE:.0%} of a profiled build's device time)",
                     "kernels": reported,
                     "whole_build": whole_build,
                     "whole_job": {"algorithmic_bytes": 5 * n + 4, "achieved": round(job_gbs, 3),
                                   "frac": round(job_gbs / HBM_PEAK_GBS, 6)}},
        "kernels": kernels,
        "device_ms_per_step": round(device_ms, 3),
        "verified": verified,
        "ranks": ranks,
        "distinct_gpus": distinct if world > 1 else None,
        "end_to_end": e2e,
        "configs": configs,
        "batch_c5": batch,
        "batch_api": batch_api,
        "host": host_info(),
    }
    if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
        try:
            result["cpu_baseline"] = cpu_baseline(text_h, min(args.cpu_sample, n))
        except Exception as ex:
            result["cpu_baseline"] = {"value": None, "unit": "MB/s", "cores": 1, "kind": "port", "error": f"{type(ex).__name__}: {ex}"[:300]}
    else:
        result["cpu_baseline"] = None
    return result


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)                       # (nothing above this line initialises the GPU)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("SA_BENCH_FORCE_LOCAL_RANK", os.environ.get("LOCAL_RANK", "0")))      # (forced: tests only)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal on a one-GPU box: SA_BENCH_SHARE_GPU=1 puts every rank on device 0; the driver's real runs use one GPU per rank.
    # Control plane: the data path has NO collective (one independent text per GPU); what the ranks exchange is a
    # barrier, a mask and a few scalars, so they go over gloo on CPU tensors by default -- bringing RCCL up across 8 GPUs is
    # failure surface the path never needs.  SA_BENCH_CONTROL=nccl uses RCCL (backend "nccl" on ROCm) for them instead.
    share = os.environ.get("SA_BENCH_SHARE_GPU") == "1"
    use_nccl = os.environ.get("SA_BENCH_CONTROL", "gloo") == "nccl" and not share
    backend = HipBackend(0 if share else local_rank)
    if args.batch_api_only:                                  # (the child of batch_api_subprocess)
        print(json.dumps(batch_api_leg(backend, args)), flush=True)
        return 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_nccl:
            dist.init_process_group(backend="nccl", device_id=backend.dev)
        else:
            dist.init_process_group(backend="gloo")
    try:
        result = run(args, backend, rank, world, dist, share or not use_nccl)        # (True: control tensors live on the CPU)
        if result is not None:
            print(json.dumps(result), flush=True)
    finally:
        if dist is not None:
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
