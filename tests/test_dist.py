"""CPU suite: the N > 1 path of bench.py itself -- rank launcher, per-rank independent texts (weak scaling, no data-path
collective: SURVEY.md section 8e), barrier + max-over-ranks timing, whole-job aggregation, the config-5 batch leg --
under torch.distributed with the gloo backend, world size 2.

bench.run() takes the object that touches the device as a parameter.  The real one (bench.HipBackend) needs a GPU; here
a stand-in DEFINED IN THIS TEST FILE builds the arrays with the CPU oracle, so that the host logic of the benchmark can
run without a GPU.  Nothing of the kind exists in bench.py or in the package: the product path has no CPU builder.  The
same code with the real backend and two ranks is exercised on the GPU box by
tests/test_gpu_parity.py::test_bench_two_ranks_share_one_gpu."""
import json
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleStandIn:
    """test double for bench.HipBackend: same methods, arrays from the oracle"""
    name = "oracle stand-in (tests/test_dist.py)"

    def __init__(self):
        from conftest import Oracle
        self.orc = Oracle()
        self.builds = 0

    def load(self, text_h):
        self.text, self.n, self.out = text_h, int(text_h.size), None

    def unload(self):
        self.text = self.out = None

    def step(self):
        self.out = self.orc.sais(self.text)
        self.builds += 1

    def sync(self):
        pass

    def kernel_names(self):
        return ["k_onesweep", "k_onesweep32", "k_group_sort", "misc"]

    def profile_begin(self, mask):
        self.mask = mask

    def profile_end(self):
        # (ms, launches, units) per class; classes outside the mask get no events
        rows = [(2.0, 4, 4 * self.n), (0.0, 0, 0), (3.0, 2, self.n), (0.1, 5, 10)]
        return [r if (self.mask >> i) & 1 else (0.0, 0, 0) for i, r in enumerate(rows)]

    def device_count(self):
        return 1

    def build_batch(self, texts, outs):
        for t, o in zip(texts, outs):
            o[:] = self.orc.sais(t)
        return 0, [0] * len(texts)

    def check_host(self, text_h, sa_h):
        return self.orc.verify(text_h, sa_h) == 1

    def verify(self):
        return self.orc.verify(self.text, self.out) == 1

    def download(self):
        return self.out

    def build_host(self, text_h, out_h):
        out_h[:] = self.orc.sais(text_h)

    def host_timing(self):
        return {"h2d": 0.0, "build": 0.0, "d2h": 0.0}

    def stats_dict(self):
        return {"sigma": 256, "bits_per_symbol": 8, "symbols_per_key": 8, "rounds": 0, "text_rounds": 0, "sort_passes": 4,
                "unresolved_after_initial": 0}

    def ctl_device(self, share):
        import torch
        return torch.device("cpu")

    def identity(self):
        # (two stand-in "GPUs": one per rank, as on the driver's multi-GPU node)
        r = int(os.environ.get("RANK", "0"))
        return {"hip_device": r, "pci_bus_id": f"0000:{0x10 + r:02x}:00.0", "gpu": "stand-in", "pid": os.getpid()}


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = bench.parse(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--workload", "c2_uniform_64m", "--n", "40000",
                        "--e2e-calls", "2"])
    backend = OracleStandIn()
    res = bench.run(args, backend, rank, world, dist, share=True)
    q.put((rank, res, backend.builds))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_runs_bench_rank_logic():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, out0, b0), (r1, out1, b1) = res
    assert out1 is None and out0 is not None                 # one JSON line, from rank 0
    assert out0["n_gpus"] == 2 and out0["steps"] == 3 and out0["scaling"] == "weak" and out0["verified"] is True
    n = 40000
    # whole-job value: both ranks' bytes over the max-over-ranks time
    assert out0["value"] == pytest.approx(2 * n / 1e6 / (out0["ms_per_step"] / 1e3), rel=1e-3)
    assert out0["config"]["n_bytes"] == n and out0["cpu_baseline"] is None
    assert out0["end_to_end"]["reused_buffer"]["MB_per_s"] > 0 and out0["end_to_end"]["fresh_buffer"]["ms"] > 0
    assert out0["batch_c5"]["texts"] == 2 and out0["batch_c5"]["verified"] is True
    # the dominant class is the one with the most measured time, not a fixed name; classes under 5 % get no events
    rf = out0["roofline"]
    assert rf["kernel"] == "k_group_sort" and rf["bound"] == "hbm" and rf["algorithmic_bytes_per_element"] == 29
    assert [k["name"] for k in rf["kernels"]] == ["k_group_sort", "k_onesweep"] and rf["kernels"][1]["achieved"] > 0
    assert out0["batch_api"] is None and out0["configs"] is None       # (N = 1 only)
    # who ran where: one entry per rank, distinct devices, each with its own time and its own verification
    assert [r["rank"] for r in out0["ranks"]] == [0, 1] and out0["distinct_gpus"] is True
    assert len({r["pci_bus_id"] for r in out0["ranks"]}) == 2 and all(r["verified"] and r["ms_per_step"] > 0 for r in out0["ranks"])
    assert out0["ms_per_step"] == pytest.approx(max(r["ms_per_step"] for r in out0["ranks"]), rel=0.05)
    json.dumps(out0)                                          # serialisable as the one line the driver reads
    assert b0 == b1 and b0 >= 2 + 1 + 3                       # first touch + profiled build, warm-up, timed steps -- on every rank


def test_single_rank_path_with_stand_in():
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse(["--steps", "2", "--warmup", "0", "--workload", "c4_dna_1g", "--n", "30000", "--cpu-sample", "20000",
                        "--e2e-calls", "1", "--batch-texts", "3", "--small-batch-texts", "40", "--configs", "c2_uniform_64m,c5_uniform_512m",
                        "--config-steps", "2"])
    out = bench.run(args, OracleStandIn(), 0, 1)
    # the other BASELINE configs ride behind the headline workload, each verified and with its own dominant class
    cf = out["configs"]
    assert set(cf) == {"c2_uniform_64m", "c5_uniform_512m", "_seconds"}
    for name in ("c2_uniform_64m", "c5_uniform_512m"):
        c = cf[name]
        assert c["verified"] is True and c["n_bytes"] == 30000 and c["steps"] == 2 and c["MB_per_s"] > 0
        assert c["roofline"]["kernel"] == "k_group_sort" and c["roofline"]["frac"] > 0 and c["whole_job"]["frac"] > 0
        assert c["end_to_end"]["fresh_buffer"]["ms"] > 0
    assert out["ranks"][0]["rank"] == 0 and out["distinct_gpus"] is None
    assert out["n_gpus"] == 1 and out["verified"] is True and out["batch_c5"] is None
    assert out["batch_api"]["texts"] == 3 and out["batch_api"]["verified"] is True and out["batch_api"]["MB_per_s"] > 0
    assert out["batch_api"]["small_texts"]["texts"] == 40 and out["batch_api"]["small_texts"]["verified"] is True
    cb = out["cpu_baseline"]
    assert cb["cores"] == 1 and cb["kind"] in ("port", "reference") and len(cb["probe"]) >= 1
    # the probe order of SURVEY.md 8d: cargo + crate, system library, python package, stand-in
    assert cb["probe"][0]["step"].startswith("cargo")
    assert (cb["kind"] == "port") == (cb["probe"][-1]["step"].startswith("stand-in"))
    assert out["host"]["nproc"] >= 1 and "H0_bits_per_byte" in out["config"]


def test_batch_api_child_process_failure_is_a_record_not_a_crash():
    """with several GPUs visible the N = 1 line runs its sa_amd_saca_batch leg in a child process (bench.batch_api_subprocess): here,
    without a GPU, the child cannot start a backend -- the parent gets an object that says so instead of an exception, and a leg
    that raises inside run() is recorded the same way (the headline line must survive everything behind it)"""
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse(["--batch-texts", "2", "--text-bytes", "4096", "--small-batch-texts", "0", "--batch-api-timeout", "120"])
    out = bench.batch_api_subprocess(args)
    assert out["verified"] is False and out["ran_in"] == "child process" and "error" in out
    json.dumps(out)

    class Failing(OracleStandIn):
        def build_batch(self, texts, outs):
            raise RuntimeError("device lost")

        def build_host(self, text_h, out_h):
            raise MemoryError("no staging")
    args = bench.parse(["--steps", "2", "--warmup", "0", "--workload", "c2_uniform_64m", "--n", "20000", "--cpu-sample", "10000", "--e2e-calls", "1",
                        "--batch-texts", "2", "--small-batch-texts", "0", "--configs", "c5_uniform_512m", "--config-steps", "1"])
    line = bench.run(args, Failing(), 0, 1)
    assert line["verified"] is True and line["value"] > 0                       # the headline stands
    assert "MemoryError" in line["end_to_end"]["error"] and "RuntimeError" in line["batch_api"]["error"]
    assert "error" in line["configs"]["c5_uniform_512m"] and line["configs"]["c5_uniform_512m"]["verified"] is False
    json.dumps(line)


def test_gpus_flag_starts_the_ranks_itself(monkeypatch):
    """`python bench.py --gpus 2` without WORLD_SIZE launches torch.distributed.run as a child process and never builds a
    backend (= never touches the GPU) in the parent"""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(bench, "HipBackend", lambda *a, **k: (_ for _ in ()).throw(AssertionError("parent touched the GPU")))
    assert bench.main(["--gpus", "2", "--steps", "4"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert cmd[-4:] == ["--gpus", "2", "--steps", "4"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_per_rank_texts_are_independent():
    from suffix_array_amd import corpus
    a, b = corpus.workload("c5_uniform_512m", rank=0, n_override=5000), corpus.workload("c5_uniform_512m", rank=1, n_override=5000)
    assert a.size == b.size == 5000 and not np.array_equal(a, b)
    assert np.array_equal(a, corpus.workload("c5_uniform_512m", rank=0, n_override=5000))       # seeded: reproducible
