"""A bounded, seeded slice of the randomised GPU-vs-oracle sweep (tests/fuzz_gpu_parity.py) inside `-m gpu`, so that the driver's
GPU run exercises random shapes, operand layouts, forced kernels (DDP_BACKPASS, DDP_MX2, DDP_FORWARD_PIPE, DDP_DPPW, DDP_FORWARD_FAST), limits, divergences and
odd horizons every round — not only when somebody runs the sweep by hand.  ~2 500 cases (under a minute); every case is reproducible on its own
(`python tests/fuzz_gpu_parity.py --cond 31 <case>` shows its conditioning)."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 31


@pytest.fixture(scope="module")
def fz():
    spec = importlib.util.spec_from_file_location("fuzz_gpu_parity", os.path.join(HERE, "fuzz_gpu_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import ddp_amd
    import ddp_amd.kl  # noqa: F401
    ddp_amd.default_handle()
    return mod, ddp_amd


@pytest.fixture(autouse=True)
def _clean_env():
    keys = ("DDP_BACKPASS", "DDP_MX2", "DDP_FORWARD_PIPE", "DDP_FORWARD", "DDP_FORWARD_LANE", "DDP_FORWARD_PEND", "DDP_DPPW", "DDP_FORWARD_FAST")
    old = {k: os.environ.get(k) for k in keys}
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("chunk", range(16))
def test_fuzz_pass_cases(fz, chunk):
    """back_pass + forward_pass (1..11 step sizes) on random LQ-family cases, every output of every trajectory at 1e-8"""
    mod, ddp = fz
    from oracle import oracle_ctypes as oc
    worst = 0.0
    for c in range(100 * chunk, 100 * (chunk + 1)):
        worst = max(worst, mod.one_case(ddp, oc, np.random.default_rng([SEED, c]), c))
    assert worst < mod.RTOL


@pytest.mark.parametrize("chunk", range(4))
def test_fuzz_row_kernel_cases(fz, chunk):
    """the same sweep over EVERY (n, m) the padded row kernels hold (n <= 14, m <= 4: 45 pairs), with the row kernel forced, dispatched by
    default, or the run-time-sized kernel: back_pass + 1..11 rollouts per case against the oracle (round 5: the shape range)"""
    mod, ddp = fz
    from oracle import oracle_ctypes as oc
    worst = 0.0
    for c in range(100 * chunk, 100 * (chunk + 1)):
        worst = max(worst, mod.one_case(ddp, oc, np.random.default_rng([SEED, 500000 + c]), c, shapes=mod.ROW_SHAPES, impls=mod.ROW_IMPLS))
    assert worst < mod.RTOL


def test_fuzz_ilqg_solves(fz):
    mod, ddp = fz
    from oracle import oracle_ctypes as oc
    rng = np.random.default_rng(SEED)
    for c in range(200):
        mod.ilqg_case(ddp, oc, rng, c)


def test_fuzz_pendcart_cases(fz):
    mod, ddp = fz
    from oracle import oracle_ctypes as oc
    for c in range(400):
        mod.pendcart_case(ddp, oc, np.random.default_rng([SEED, 100000 + c]), c)


def test_fuzz_kl_path_cases(fz):
    mod, ddp = fz
    from oracle import oracle_ctypes as oc
    for c in range(300):
        mod.gps_case(ddp, oc, np.random.default_rng([SEED, 200000 + c]), c)


def test_fuzz_escapes_stay_rare(fz):
    """VERDICT r03 / ADVICE: the tolerance escapes of the sweep (ill-conditioned draws judged at the distance of the two CPU restatements,
    draws the restatements cannot judge, iLQG solves on a boxQP knife edge) are COUNTED, printed and bounded by committed ceilings — a
    kernel defect that hides behind them makes the counts grow.  Runs after the slices above (same module, same counters)."""
    mod, _ = fz
    n = mod.escape_counts()
    print("fuzz escapes over 1 600 pass cases + 200 solves + 300 KL cases: %s" % n)
    assert n["ill_conditioned"] <= mod.MAX_ILL_PER_1000 * 1.6, n
    assert n["unjudged"] <= max(1.0, mod.MAX_UNJUDGED_PER_1000 * 1.6), n
    assert n["knife_edge"] <= mod.MAX_KNIFE_PER_1000 * 0.2, n
    assert n["exploded_kl_draws"] <= 0.1 * 300, n
