"""Capacity of the shared-LTI backward pass's work-item list (csrc/back_pass_sh.hip: ddp_sh_max_tiles sizes items[] and the grid on the host,
sh_group_kernel chooses the tile size on the device from the number of λ groups it finds).  Round 4 sized the list for ONE group holding
the whole batch; 16 groups of 500 trajectories made 480 tiles for a 272-entry list.  Here the device's choice is restated in Python and
swept over batch sizes around the k * 32 * (ncu - G) edges, group counts and group-size distributions: the count it produces must never
exceed what the host function (called through the built library, no GPU needed) allocates."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT

LIB = os.path.join(ROOT, "differentialdynamicprogramming.jl_amd", "libddp_amd.so")
TMAX, GMAX = 32, 16


def device_tiles(counts, ncu, wmax):
    """sh_group_kernel's tile choice (back_pass_sh.hip, `if (tid == 0)` block): returns (T, W)"""
    G, start = len(counts), sum(counts)
    slots = max(ncu - G, 8)
    R = (start + slots * TMAX - 1) // (slots * TMAX)
    T = max(4, (start + R * slots - 1) // (R * slots))
    cap = min(R * slots, wmax)
    while T < TMAX:
        if sum((c + T - 1) // T for c in counts) <= cap:
            break
        T += 1
    return T, sum((c + T - 1) // T for c in counts)


@pytest.fixture(scope="module")
def max_tiles():
    if not os.path.exists(LIB):
        pytest.skip("libddp_amd.so not built")
    try:
        L = C.CDLL(LIB)
    except OSError as e:                      # no HIP runtime on this host
        pytest.skip(str(e))
    f = L.ddp_sh_max_tiles
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int]
    return f


def test_advisor_example(max_tiles):
    """ncu = 256, B = 8 000, 16 groups of 500: the device wants R = 2, T = 17, 480 tiles"""
    w = max_tiles(8000, 256)
    T, W = device_tiles([500] * 16, 256, w)
    assert (T, W) == (17, 480) and W <= w


@pytest.mark.parametrize("ncu", [256, 304, 64, 20])
def test_tiles_never_exceed_the_capacity(max_tiles, ncu):
    rng = np.random.default_rng(ncu)
    worst = 0.0
    for G in (1, 2, 3, 8, 15, 16):
        slots = max(ncu - G, 8)
        edges = {k * TMAX * slots + d for k in (1, 2, 3, 4, 5) for d in (-33, -1, 0, 1, 31, 32, 33, 100)}
        for B in sorted(edges | {G * 2, 1024, 2048, 8000, 8160, 32768, 100000}):
            if B < 2 * G:
                continue
            w = max_tiles(B, ncu)
            assert w >= (B + TMAX - 1) // TMAX + GMAX
            shapes = [np.full(G, B // G)]                                              # equal groups (the remainder ungrouped)
            shapes.append(np.r_[np.full(G - 1, 2), B - 2 * (G - 1)])                   # one big group, the others minimal
            shapes.append(np.maximum(2, rng.multinomial(B - 2 * G, rng.dirichlet(np.ones(G))) + 2))
            shapes.append(np.maximum(2, (np.full(G, B // G) * rng.uniform(0.3, 1.0, G)).astype(int)))      # part of the batch ungrouped
            for c in shapes:
                c = [int(v) for v in c if v >= 2]
                if sum(c) > B:
                    continue
                T, W = device_tiles(c, ncu, w)
                assert W <= w, (ncu, G, B, c[:4], T, W, w)
                assert 4 <= T <= TMAX
                worst = max(worst, W / w)
    assert worst > 0.5          # the bound is not wildly loose either
