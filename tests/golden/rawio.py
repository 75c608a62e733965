"""Dependency-free exchange format between the Python fixtures and the Julia reference script
(julia/make_reference_fixtures.jl):  <dir>/<case>.bin  holds the arrays of a case back to back as little-endian
f64 / i64 in COLUMN-MAJOR order (Julia's memory order), <dir>/manifest.txt one line per array:

    case key dtype ndim d1 ... dn offset_bytes

    python tests/golden/rawio.py          # exports the INPUTS of every tests/golden/*.npz to tests/golden/raw/

The outputs stay in the .npz files (they are what the builder's restatement produced); the Julia script writes the
reference's own outputs to tests/golden/julia/ in the same format and tests/test_julia_fixtures.py compares."""
import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
RAW = os.path.join(HERE, "raw")
JULIA = os.path.join(HERE, "julia")

# keys of each fixture family that are RESULTS (everything else is an input the Julia script needs)
OUTPUTS = {
    "bp_": {"diverge", "K", "k", "Quu", "Vx", "Vxx", "dV"},
    "boxqp": {"x", "result", "free", "Hfree"},
    "fwd_": {"xnew", "unew", "cnew"},
    "df_": {"fx", "fu", "cx", "cu"},
    "ilqg_": {"x", "u", "K", "k", "Quu", "Vx", "Vxx", "cost", "status", "iter", "lam", "n_backpass", "n_forward", "tr_cost", "tr_lambda", "tr_dlambda",
              "tr_alpha", "tr_improvement", "tr_reduce_ratio", "tr_grad_norm"},
    "kl_gps_": {"cxkl", "cukl", "cxxkl", "cxukl", "cuukl", "diverge", "K", "k", "Quui", "Quu", "Vx", "Vxx", "dV", "sigmanew", "kldiv",
                # calc_η (src/klutils.jl:110-133) on the fixture's own divergence: written by the Julia script only (the step sizes are
                # chosen around ITS mean divergence), compared in tests/test_julia_fixtures.py when present
                "eta_kl_steps", "eta_out", "eta_satisfied", "eta_divergence"},
    # the whole KL-constrained solve (src/iLQGkl.jl:25-178).  forward_covariance needs df / covariance of the un-vendored
    # LinearTimeVaryingModelsBase: the Julia script supplies a fixture model with the given fx, R1 (as for kl_gps_*)
    "kl_ilqgkl_": {"xnew", "unew", "K", "S", "Si", "Vx", "Vxx", "cost", "status", "iter", "eta", "divergence", "n_backpass"},
}


def family(case):
    for pre in OUTPUTS:
        if case.startswith(pre):
            return pre
    raise KeyError(case)


def write_case(dirname, case, arrays, manifest_lines):
    off = 0
    with open(os.path.join(dirname, case + ".bin"), "wb") as f:
        for key, a in arrays.items():
            a = np.asarray(a)
            if a.dtype.kind in "iub":
                a = a.astype("<i8"); dt = "i64"
            else:
                a = a.astype("<f8"); dt = "f64"
            data = a.tobytes(order="F")
            manifest_lines.append(" ".join([case, key, dt, str(a.ndim)] + [str(d) for d in a.shape] + [str(off)]))
            f.write(data)
            off += len(data)


def read_dir(dirname):
    """{case: {key: array}} of a directory written by write_case / by the Julia script"""
    out = {}
    man = os.path.join(dirname, "manifest.txt")
    if not os.path.exists(man):
        return out
    blobs = {}
    for line in open(man):
        t = line.split()
        if not t or t[0].startswith("#"):
            continue
        case, key, dt, nd = t[0], t[1], t[2], int(t[3])
        shape = tuple(int(v) for v in t[4:4 + nd])
        off = int(t[4 + nd])
        if case not in blobs:
            blobs[case] = open(os.path.join(dirname, case + ".bin"), "rb").read()
        cnt = int(np.prod(shape)) if shape else 1
        a = np.frombuffer(blobs[case], dtype="<i8" if dt == "i64" else "<f8", count=cnt, offset=off)
        out.setdefault(case, {})[key] = a.reshape(shape, order="F").copy() if shape else a[0]
    return out


def export_inputs():
    os.makedirs(RAW, exist_ok=True)
    lines = ["# case key dtype ndim dims... offset_bytes   (column-major little-endian; written by tests/golden/rawio.py)"]
    for f in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        case = os.path.basename(f)[:-4]
        outs = OUTPUTS[family(case)]
        if outs is None:
            continue
        with np.load(f) as z:
            arrays = {k: z[k] for k in z.files if k not in outs}
        write_case(RAW, case, arrays, lines)
    with open(os.path.join(RAW, "manifest.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return len(lines) - 1


if __name__ == "__main__":
    n = export_inputs()
    print("exported %d input arrays to %s" % (n, RAW))
