"""N>1 path on CPU: two gloo processes, each solving its shard of a batch (with the CPU oracle standing in for the
GPU solver — the sharding / statistics / collective code is what is under test), one all-reduce."""
import os
import socket

import numpy as np
import pytest

from ddp_amd import sharding


def test_shard_range_partitions_the_batch():
    for B in (1, 7, 8, 1024, 1000):
        for world in (1, 2, 3, 8):
            got = [sharding.shard_range(B, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            sizes = [hi - lo for lo, hi in got]
            assert max(sizes) - min(sizes) <= 1


def _oracle_solver(problem, x0, u0, **kw):
    """CPU stand-in with the return convention of ddp_amd.iLQG (batched)"""
    from oracle import oracle_ctypes as oc
    n, B = x0.shape
    m, N = u0.shape[:2]
    p = oc.make_problem("lq", n, m, N, A=problem["A"], B=problem["B"], Q=problem["Q"], R=problem["R"])
    stats = np.zeros((8, B)); costs = np.zeros((N, B))
    for b in range(B):
        x, u, (K, k, Quu), Vx, Vxx, cost, info = oc.ilqg(p, x0[:, b], u0[:, :, b])
        stats[:, b] = [info["status"], info["iter"], info["accepted_iter"], info["n_backpass"], info["n_forward"], info["lam"],
                       info["g_norm"], cost.sum()]
        costs[:, b] = cost
    return None, None, None, None, None, costs, dict(stats=stats)


def _make_batch():
    from oracle import np_restatement as npr
    rng = np.random.default_rng(5)
    P = npr.make_lq_problem(rng, T=60)
    B = 5
    x0 = np.ones((10, B)) + 0.1 * rng.standard_normal((10, B))
    u0 = 0.1 * rng.standard_normal((2, 60, B))
    return P, x0, u0


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, x0, u0 = _make_batch()
    res, g, (lo, hi) = sharding.solve_sharded(P, x0, u0, solver=_oracle_solver)
    q.put((rank, lo, hi, g))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_allreduce_matches_unsharded():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    assert (out[0][1], out[0][2], out[1][1], out[1][2]) == (0, 3, 3, 5)
    # both ranks hold the same global statistics, equal to the unsharded solve
    P, x0, u0 = _make_batch()
    _, _, _, _, _, cost, tr = _oracle_solver(P, x0, u0)
    ref = sharding.stats_from_solve(tr["stats"], cost)
    for _, _, _, g in out:
        got = np.array([g[k] for k in sharding.STAT_NAMES])
        assert np.allclose(got, ref, rtol=1e-12, atol=0)
    assert out[0][3]["n_traj"] == 5 and out[0][3]["n_converged"] == 5



def test_file_exchange_ships_the_id(tmp_path):
    """the rendezvous of CApiComm without torch.distributed: rank 0 writes the 128-byte id, the others wait for it"""
    import threading
    path = str(tmp_path / "rccl_id")
    blob = bytes(range(128))
    got = {}

    def other():
        got[1] = sharding.file_exchange(path, 1, timeout=30.0)(None)
    t = threading.Thread(target=other)
    t.start()
    assert sharding.file_exchange(path, 0)(blob) == blob
    t.join(timeout=30)
    assert got[1] == blob
    with pytest.raises(TimeoutError):
        sharding.file_exchange(str(tmp_path / "missing"), 1, timeout=0.05)(None)


def test_solve_sharded_without_torch_distributed_uses_the_capi_comm(monkeypatch):
    """torch.distributed not initialised (a Julia / C launcher): rank and world size from the environment, the statistics vector through
    the communicator object (CApiComm on a GPU box; a stand-in with its interface here) — one call with the SUM / MAX split"""
    P, x0, u0 = _make_batch()
    calls = []

    class FakeComm:
        def allreduce_host(self, v, nsum):
            calls.append((np.array(v), nsum))
            return np.concatenate([2.0 * v[:nsum], v[nsum:]])        # "two ranks with the same shard"
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "2")
    res, g, (lo, hi) = sharding.solve_sharded(P, x0, u0, solver=_oracle_solver, comm=FakeComm())
    assert (lo, hi) == (3, 5) and len(calls) == 1 and calls[0][1] == sharding.N_SUM
    assert g["n_traj"] == 4.0 and g["max_iters"] == calls[0][0][10]
    monkeypatch.delenv("DDP_COMM_ID_FILE", raising=False)
    with pytest.raises(RuntimeError, match="DDP_COMM_ID_FILE"):
        sharding.solve_sharded(P, x0, u0, solver=_oracle_solver)


def test_allreduce_stats_single_rank_needs_no_torch(monkeypatch):
    """the torch-absent mode the docstring of solve_sharded promises: one rank, no process group -> the vector comes back as it is and
    torch is not even imported"""
    import sys
    monkeypatch.setitem(sys.modules, "torch", None)               # `import torch` now raises ImportError
    monkeypatch.setitem(sys.modules, "torch.distributed", None)
    v = np.arange(len(sharding.STAT_NAMES), dtype=float)
    assert np.array_equal(sharding.allreduce_stats(v), v)


@pytest.mark.timeout(600)
def test_bench_gpus_8_launch_contract_dry_run():
    """`python bench.py --gpus 8` without a launcher starts its eight ranks itself (torch.distributed.run, 127.0.0.1); --dry-run keeps
    everything but the device work: rank 0's line must report the world size IT SAW in the communicator (n_ranks_seen, not the command line),
    every rank its own seed and its own trajectories, all of them the same shared operands (A, B: LTI), and the statistics vector the sum
    over the ranks.  No hardware with more than one RCCL rank has run this yet (the driver's SCALE leg does); this is what a CPU host can pin."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, DDP_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=550)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["value"] is None        # nobody can mistake it for a measurement
    assert out["n_ranks_seen"] == 8 and out["n_gpus"] == 8 and out["scaling"] == "weak"
    ranks = sorted(out["ranks"], key=lambda d: d["rank"])
    assert [d["rank"] for d in ranks] == list(range(8))
    assert len({d["seed"] for d in ranks}) == 8 and len({d["inputs_digest"] for d in ranks}) == 8
    assert len({d["shared_operands_digest"] for d in ranks}) == 1
    assert abs(out["collective"]["stats"][0] - sum(d["local_stats0"] for d in ranks)) < 1e-9 * abs(out["collective"]["stats"][0])
    assert out["collective"]["stats"][2] == 8 * 64
