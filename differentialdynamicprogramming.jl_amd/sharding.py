"""Multi-GPU use of the path: batch sharding + ONE small all-reduce.

The reference solves one trajectory; a batch of independent trajectories / MPC rollouts has no coupling
(each has its own λ schedule, line search and divergence flag), so the path shards by contiguous batch ranges,
one process per GPU, with no data-path exchange.  The only collective is an all-reduce (RCCL on GPUs,
gloo in the CPU tests) of a short statistics vector per outer iteration / per solve, used for batch-level
reporting and global termination.  It is latency-bound (tens of bytes); xGMI bandwidth is irrelevant.
"""
from __future__ import annotations

import numpy as np

STAT_NAMES = ("sum_cost", "n_traj", "n_converged", "n_lambda_exit", "n_maxiter", "n_init_diverged", "sum_iters",
              "sum_backpass", "sum_forward", "sum_gnorm", "max_iters")
N_SUM = 10          # the first N_SUM entries reduce with SUM, the rest with MAX


def shard_range(B: int, rank: int, world: int):
    """contiguous shard [lo, hi) of a batch of B trajectories for `rank` of `world` (sizes differ by at most one)"""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def stats_from_solve(stats, cost):
    """local statistics vector from the per-trajectory summary `stats[8,B]` of ddp_ilqg_f64 (include/ddp_amd.h)
    and the final cost vectors `cost[CL,B]`"""
    st = np.asarray(stats, dtype=np.float64).reshape(8, -1)
    status = st[0]
    v = np.zeros(len(STAT_NAMES))
    v[0] = float(np.sum(cost))
    v[1] = st.shape[1]
    v[2] = float(np.sum((status == 1) | (status == 2)))
    v[3] = float(np.sum(status == 3))
    v[4] = float(np.sum(status == 4))
    v[5] = float(np.sum(status == -1))
    v[6] = float(np.sum(st[1]))
    v[7] = float(np.sum(st[3]))
    v[8] = float(np.sum(st[4]))
    v[9] = float(np.sum(st[6]))
    v[10] = float(np.max(st[1])) if st.shape[1] else 0.0
    return v


def allreduce_stats(v, device=None):
    """all-reduce a statistics vector over the default torch.distributed group (no-op when not initialised)"""
    dist = _torch_dist()
    if dist is None or dist.get_world_size() == 1:           # also the torch-absent, single-rank mode: nothing to import, nothing to reduce
        return np.asarray(v, dtype=np.float64)
    import torch
    # ONE collective: gather the tiny vectors, reduce locally (SUM for the first N_SUM entries, MAX for the rest) — the same
    # scheme as ddp_allreduce_stats_f64_dev of the C ABI (csrc/comm.hip)
    t = torch.as_tensor(np.asarray(v, dtype=np.float64), device=device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    g = torch.stack(parts)
    return torch.cat([g[:, :N_SUM].sum(dim=0), g[:, N_SUM:].max(dim=0).values]).cpu().numpy()


def file_exchange(path, rank, timeout=120.0):
    """an `exchange` for CApiComm on hosts without torch.distributed (or MPI): rank 0 writes the 128-byte RCCL id to `path`
    (atomically, via rename), the other ranks wait for the file.  `path` must be visible to every rank and fresh per job."""
    import os
    import time

    def exchange(b):
        if rank == 0:
            tmp = "%s.%d.tmp" % (path, os.getpid())
            with open(tmp, "wb") as f:
                f.write(b)
            os.replace(tmp, path)
            return b
        t0 = time.time()
        while True:
            try:
                with open(path, "rb") as f:
                    raw = f.read()
                if len(raw) == 128:
                    return raw
            except FileNotFoundError:
                pass
            if time.time() - t0 > timeout:
                raise TimeoutError("no RCCL id at %s after %.0f s" % (path, timeout))
            time.sleep(0.01)
    return exchange


class CApiComm:
    """The RCCL communicator owned by the C ABI (ddp_comm_create / ddp_allreduce_stats_f64_dev, include/ddp_amd.h) — what a Julia
    or C host without torch uses for the one collective of the path.  `exchange(id_bytes_or_None) -> id_bytes` ships the
    128-byte RCCL id from rank 0 to every rank; by default torch.distributed's object broadcast does it (rendezvous only)."""

    def __init__(self, handle, rank, world, exchange=None):
        import ctypes as C
        from . import _lib
        self._lib, self.handle, self.rank, self.world = _lib, handle, rank, world
        idbuf = (C.c_char * 128)()
        if rank == 0:
            _lib.check(_lib.lib().ddp_comm_unique_id(idbuf))
        if exchange is None:
            def exchange(b):
                import torch.distributed as dist
                if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
                    return b
                box = [b]
                dist.broadcast_object_list(box, src=0)
                return box[0]
        raw = exchange(bytes(idbuf.raw) if rank == 0 else None)
        idbuf.raw = raw
        self._c = C.c_void_p()
        _lib.check(_lib.lib().ddp_comm_create(handle.raw, int(world), int(rank), idbuf, C.byref(self._c)))

    def allreduce(self, dptr, nsum, nmax=0):
        """in place on the device vector at `dptr` (ctypes pointer / int): SUM of the first nsum entries, MAX of the next nmax"""
        import ctypes as C
        self._lib.check(self._lib.lib().ddp_allreduce_stats_f64_dev(self.handle.raw, self._c, C.c_void_p(getattr(dptr, "value", dptr)),
                                                                    int(nsum), int(nmax)))

    def allreduce_host(self, v, nsum):
        """a host vector through the device collective: SUM of the first nsum entries, MAX of the rest; returns a new array"""
        v = np.ascontiguousarray(v, dtype=np.float64)
        d = self.handle.to_device(v)
        try:
            self.allreduce(d, nsum, v.size - nsum)
            self.handle.sync()
            return self.handle.to_host(d, v.shape)
        finally:
            self.handle.free(d)

    @staticmethod
    def rccl_info():
        """(NCCL version code of the librccl the library uses, whether that instance was already resident in the process)"""
        import ctypes as C
        from . import _lib
        ver, pre = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().ddp_comm_rccl_info(C.byref(ver), C.byref(pre)))
        return ver.value, bool(pre.value)

    def close(self):
        if self._c:
            self._lib.lib().ddp_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _torch_dist():
    """torch.distributed when it is importable AND initialised, else None"""
    try:
        import torch.distributed as dist
    except Exception:
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def solve_sharded(problem, x0, u0, *, solver=None, device=None, comm=None, handle=None, **kw):
    """Every rank solves its contiguous shard of the batch (x0[n,B], u0[m,N,B]) and the ranks all-reduce the
    statistics vector.  `solver(problem, x0_shard, u0_shard, **kw)` must return the tuple of ``iLQG``; it defaults
    to the GPU solver of this package.  Returns ``(local_result, global_stats_dict, (lo, hi))``.

    The collective: torch.distributed when a process group is initialised (backend nccl = RCCL, gloo in the CPU tests); otherwise —
    torch absent or not initialised, as under a Julia / C launcher — rank and world size come from RANK / WORLD_SIZE and the vector
    goes through the C ABI's own RCCL communicator (`comm`: a CApiComm, or one is created with the id exchanged through the file
    named by DDP_COMM_ID_FILE)."""
    import os
    dist = _torch_dist()
    if dist is not None:
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    B = u0.shape[2]
    lo, hi = shard_range(B, rank, world)
    own_solver = solver is None
    if own_solver:
        from . import iLQG as solver        # GPU path (raises without a GPU: no CPU fallback)
        # ONE handle for the solve and for the collective: the caller's, or the default handle of this rank's device.  With per-rank
        # HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES isolation every rank sees its GPU as device 0; without it LOCAL_RANK names it.
        if handle is None:
            from . import default_handle
            from . import _lib
            local = int(os.environ.get("LOCAL_RANK", rank))
            ndev = _lib.lib().ddp_device_count()
            handle = default_handle(local if local < ndev else 0)
        kw = dict(kw, handle=handle)
    res = solver(problem, np.ascontiguousarray(x0[:, lo:hi]), np.ascontiguousarray(u0[:, :, lo:hi]), **kw)
    x, u, pol, Vx, Vxx, cost, trace = res
    v = stats_from_solve(trace["stats"], cost)
    if comm is None and dist is None and world > 1:
        path = os.environ.get("DDP_COMM_ID_FILE")
        if not path:
            raise RuntimeError("solve_sharded without torch.distributed: pass comm=CApiComm(...) or name a shared file in DDP_COMM_ID_FILE")
        if handle is None:
            from . import default_handle
            from . import _lib
            local = int(os.environ.get("LOCAL_RANK", rank))
            handle = default_handle(local if local < _lib.lib().ddp_device_count() else 0)
        comm = CApiComm(handle, rank, world, exchange=file_exchange(path, rank))
        if rank == 0:
            # every rank has the id once the communicator exists (ddp_comm_create is collective): a path that is reused by a later
            # job must not hand that job a stale id
            try:
                os.unlink(path)
            except OSError:
                pass
    g = comm.allreduce_host(v, N_SUM) if comm is not None else allreduce_stats(v, device=device)
    return res, dict(zip(STAT_NAMES, g)), (lo, hi)
