"""bench.py prints ONE JSON line with the fields the driver's contract names (task statement ④) — a short run on the GPU box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    env = dict(os.environ, DDP_BENCH_REHEARSALS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--preheat", "10", "--cpu-sample", "2",
                        "--no-other-configs", "--fill-batch", "0"], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d.get("sh_timeouts") == 0, (d.get("sh_timeouts"), d.get("sh_timeout_info"))      # no tile of the shared kernel gave up in the timed run
    assert d["steps"] == 6 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["dtype"] == "f64" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1024 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-3 * d["value"]                 # value = units / elapsed
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert abs(rf["achieved"] - rf["bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1.0       # algorithmic bytes / HIP-event time
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    # roofline.traffic is measured by this run (two rocprofv3 --pmc child runs), not replayed: within 5 % of the algorithmic bytes or above
    # (a box whose image is still paging in can run the profiler's child past its time limit: bench.py then says so and replays the
    # committed figure of the same kernel and batch — the line must still carry a traffic figure)
    assert rf["traffic"] is not None, rf
    if "measured in this run" not in rf["traffic_source"]:
        # the replayed figure says nothing about THIS tree's kernels: no bound is asserted on it
        assert "timed out" in rf["traffic_source"], rf["traffic_source"]
        pytest.skip("rocprofv3 child ran past its time limit on this box: traffic replayed from profiles/pmc_traffic.json, bounds not checked (%s)" % rf["traffic_source"][:120])
    assert 0.95 * rf["bytes_per_launch"] < rf["traffic"] < 1.5 * rf["bytes_per_launch"]


def _torchrun_bench(collective, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--preheat", "0", "--no-cpu-baseline", "--no-other-configs",
           "--fill-batch", "0", "--no-traffic", "--collective", collective]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=dict(os.environ, DDP_BENCH_REHEARSALS="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_one_rank_collective_torch_and_capi_agree():
    """The N > 1 code path as far as ONE GPU can run it (RCCL refuses two ranks on one device): under torch.distributed.run with one rank
    the per-step statistics vector goes through an RCCL communicator — torch's, or the C ABI's own (ddp_comm_*).  Both must deliver the
    same vector; the C ABI must use the librccl instance torch already loaded (never a second copy) and see a version >= its declared ABI."""
    import socket
    ports = []
    for _ in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            ports.append(s.getsockname()[1])
    a, b = _torchrun_bench("torch", ports[0]), _torchrun_bench("capi", ports[1])
    sa, sb = a["collective"]["stats"], b["collective"]["stats"]
    assert a["collective"]["issued_by"] == "torch" and b["collective"]["issued_by"] == "capi"
    assert len(sa) == 4 and sa[3] == 0.0 and sa[0] > 0.0
    assert all(abs(x - y) <= 1e-12 * max(1.0, abs(x)) for x, y in zip(sa, sb)), (sa, sb)
    assert b["collective"]["rccl_version"] >= 21800 and b["collective"]["rccl_instance"].startswith("the one already resident")


@pytest.mark.gpu
def test_bare_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the way the driver calls `--gpus 1`): bench.py starts its two ranks itself under
    torch.distributed.run and rank 0 prints the one line.  Test mode: gloo (two ranks may share the box's single GPU, RCCL refuses
    that); `n_ranks_seen` comes from the communicator."""
    env = dict(os.environ, DDP_BENCH_BACKEND="gloo", DDP_BENCH_REHEARSALS="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--preheat", "0", "--batch", "256",
                        "--no-cpu-baseline", "--no-other-configs", "--fill-batch", "0", "--no-traffic"], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["scaling"] == "weak" and d["steps"] == 4
    assert abs(d["value"] - 2 * 256 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-3 * d["value"]             # whole-job units / max-over-ranks time
    assert len(d["collective"]["stats"]) == 4 and d["collective"]["stats"][0] > 0.0


def test_spawn_ranks_builds_the_driver_command(monkeypatch):
    """CPU: the command line bench.py hands to subprocess when it is asked for N GPUs without a launcher"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class R:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    assert bench.spawn_ranks(4) is None
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert cmd[-7].endswith("bench.py") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
