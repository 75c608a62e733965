"""The Julia binding (differentialdynamicprogramming.jl_amd/julia/DDPAmd.jl) cannot be executed in the build image (no Julia);
what CAN be checked statically is checked here against include/ddp_amd.h: every `@ccall` names an exported function, passes
the right number of arguments with the right C types and return type, and every mirrored struct has the header's field order
and types.  (tests/test_capi_cpu.py checks the same header against the ctypes host.)"""
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "ddp_amd.h")
JL = os.path.join(ROOT, "differentialdynamicprogramming.jl_amd", "julia", "DDPAmd.jl")

# C type -> the Julia type(s) a @ccall may declare for it
CTYPES = {
    "ddp_handle": {"Ptr{Cvoid}"}, "ddp_handle *": {"Ptr{Ptr{Cvoid}}"},
    "void *": {"Ptr{Cvoid}"}, "const void *": {"Ptr{Cvoid}"}, "void **": {"Ptr{Ptr{Cvoid}}"},
    "const double *": {"Ptr{Float64}", "Ptr{Cdouble}"}, "double *": {"Ptr{Float64}", "Ptr{Cdouble}"},
    "const int32_t *": {"Ptr{Int32}"}, "int32_t *": {"Ptr{Int32}"}, "uint8_t *": {"Ptr{UInt8}"},
    "int *": {"Ptr{Cint}", "Ptr{Int32}"}, "float *": {"Ptr{Cfloat}", "Ptr{Float32}"},
    "int": {"Cint", "Int32"}, "size_t": {"Csize_t", "UInt"}, "double": {"Cdouble", "Float64"},
    "const ddp_bp_desc *": {"Ptr{BPDesc}"}, "const ddp_problem *": {"Ptr{CProblem}"}, "const ddp_ilqg_opts *": {"Ptr{ILQGOpts}"},
    "ddp_ilqg_opts *": {"Ptr{ILQGOpts}"}, "const ddp_qp_opts *": {"Ptr{QPOpts}"}, "const ddp_kl_cost_terms *": {"Ptr{KLCostTerms}"},
    "const ddp_kl_dual *": {"Ptr{KLDual}"}, "const ddp_ilqgkl_opts *": {"Ptr{ILQGKLOpts}"}, "ddp_ilqgkl_opts *": {"Ptr{ILQGKLOpts}"},
    "const char *": {"Cstring"}, "void": {"Cvoid"},
}
STRUCTS = {"ddp_bp_desc": "BPDesc", "ddp_qp_opts": "QPOpts", "ddp_problem": "CProblem", "ddp_ilqg_opts": "ILQGOpts",
           "ddp_kl_cost_terms": "KLCostTerms", "ddp_kl_dual": "KLDual", "ddp_ilqgkl_opts": "ILQGKLOpts"}
FIELD = {"int": {"Cint"}, "uint32_t": {"Cuint", "UInt32"}, "double": {"Cdouble", "Float64"}, "const double *": {"Ptr{Float64}"}, "double *": {"Ptr{Float64}"},
         "int32_t *": {"Ptr{Int32}"}}


def _strip_comments(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def _norm(t):
    t = re.sub(r"\s+", " ", t.strip())
    return re.sub(r"\s*\*\s*", " *", t).replace("* *", "**").strip()


def c_prototypes():
    src = _strip_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[A-Za-z_][\w]*(?:\s*\*+)?)\s+\**\s*(ddp_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        full = src[m.start():m.end()]
        ret = _norm(re.match(r"\s*(.*?)\s*" + name, full.replace("\n", " ")).group(1))
        params = []
        if args.strip() not in ("", "void"):
            for a in args.split(","):
                a = _norm(re.sub(r"/\*.*?\*/", "", a))
                mm = re.match(r"(.*?)(\w+)$", a)           # drop the parameter name
                typ = _norm(mm.group(1)) if mm and mm.group(1).strip() else a
                params.append(typ)
        protos[name] = (ret, params)
    return protos


def c_structs():
    src = _strip_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"typedef struct \{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"((?:const\s+)?\w+)\s*(.*)$", decl, flags=re.S)
            base, names = mm.group(1), mm.group(2)
            for nm in names.split(","):
                nm = nm.strip()
                ptr = nm.startswith("*")
                nm = nm.lstrip("* ")
                arr = re.match(r"(\w+)\[(\d+)\]", nm)
                if arr:
                    fields.append((arr.group(1), base, int(arr.group(2))))
                else:
                    fields.append((nm, _norm(base + (" *" if ptr else "")), 0))
        out[m.group(2)] = fields
    return out


def split_top(s, sep=","):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def jl_ccalls():
    src = open(JL).read()
    src = "\n".join(line.split("#")[0] if not line.lstrip().startswith("#") else "" for line in src.split("\n"))
    calls = []
    for m in re.finditer(r"@ccall\s+libddp\.(\w+)\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        args = src[m.end():i - 1]
        ret = re.match(r"::([\w{}]+)", src[i:]).group(1)
        types = []
        for a in split_top(args):
            # the declared type follows the LAST top-level `::`
            depth, pos = 0, -1
            for j, ch in enumerate(a):
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                elif ch == ":" and depth == 0 and a[j:j + 2] == "::":
                    pos = j
            assert pos >= 0, "argument without a type in @ccall %s: %r" % (m.group(1), a)
            types.append(a[pos + 2:].strip())
        calls.append((m.group(1), types, ret, src[:m.start()].count("\n") + 1))
    return calls


def jl_structs():
    src = open(JL).read()
    out = {}
    for m in re.finditer(r"\nstruct (\w+)\n(.*?)\nend", src, flags=re.S):
        fields = []
        for line in m.group(2).split("\n"):
            line = line.split("#")[0].strip()
            if "::" in line:
                nm, ty = line.split("::")
                fields.append((nm.strip(), ty.strip()))
        out[m.group(1)] = fields
    return out


def test_header_parser_sees_every_export():
    from ddp_amd import _lib
    protos = c_prototypes()
    assert set(protos) == set(_lib.EXPORTS), set(protos) ^ set(_lib.EXPORTS)


def test_every_ccall_matches_the_header():
    protos = c_prototypes()
    calls = jl_ccalls()
    assert len(calls) >= 25
    for name, types, ret, line in calls:
        assert name in protos, "DDPAmd.jl:%d calls %s which include/ddp_amd.h does not declare" % (line, name)
        cret, cparams = protos[name]
        assert len(types) == len(cparams), "DDPAmd.jl:%d %s: %d arguments, the header has %d" % (line, name, len(types), len(cparams))
        assert ret in CTYPES[cret], "DDPAmd.jl:%d %s returns %s, header: %s" % (line, name, ret, cret)
        for i, (jt, ct) in enumerate(zip(types, cparams)):
            assert ct in CTYPES, "unmapped C type %r (%s argument %d)" % (ct, name, i)
            assert jt in CTYPES[ct], "DDPAmd.jl:%d %s argument %d: Julia %s, header %s" % (line, name, i + 1, jt, ct)


def test_binding_covers_the_hot_path_entry_points():
    called = {c[0] for c in jl_ccalls()}
    need = {"ddp_create", "ddp_destroy", "ddp_last_error", "ddp_back_pass_f64", "ddp_back_pass_f64_dev", "ddp_boxqp_f64", "ddp_forward_pass_f64",
            "ddp_forward_pass_f64_dev", "ddp_df_f64", "ddp_df_f64_dev", "ddp_ilqg_ex_f64", "ddp_ilqg_ex_f64_dev", "ddp_ilqg_set_timing",
            "ddp_malloc", "ddp_free", "ddp_memcpy_h2d", "ddp_memcpy_d2h", "ddp_mpc_shift_f64_dev", "ddp_kl_terms_f64", "ddp_back_pass_gps_f64",
            "ddp_forward_covariance_f64", "ddp_kl_div_f64", "ddp_ilqgkl_f64"}
    assert need <= called, need - called


def test_struct_layouts_match_the_header():
    cs, js = c_structs(), jl_structs()
    for cname, jname in STRUCTS.items():
        cf, jf = cs[cname], js[jname]
        assert len(cf) == len(jf), (cname, [f[0] for f in cf], [f[0] for f in jf])
        for (cn, ct, carr), (jn, jt) in zip(cf, jf):
            if carr:
                assert jt == "NTuple{%d,Cdouble}" % carr or jt == "NTuple{%d,Float64}" % carr, (cname, cn, jt)
            else:
                assert jt in FIELD[ct], (cname, cn, ct, jt)
            assert cn.rstrip("_") == jn.rstrip("_") or {cn, jn} <= {"lambda", "lambda_"}, (cname, cn, jn)


def test_drop_in_signature_keeps_the_reference_keywords():
    """iLQG(f,costfun,df,x0,u0; ...) of src/iLQG.jl:143-163: the device-driver method lists every keyword of the reference"""
    src = open(JL).read()
    m = re.search(r"function iLQG\(problem::RegisteredProblem, x0, u0;(.*?)\)\n", src, flags=re.S)
    kws = {k.split("=")[0].strip() for k in split_top(m.group(1))}
    ref = {"lims", "α", "tol_fun", "tol_grad", "max_iter", "λ", "dλ", "λfactor", "λmax", "λmin", "regType", "reduce_ratio_min", "diff_fun", "plot",
           "verbosity", "plot_fun", "cost", "traj_prev", "print_head"}
    assert ref <= kws, ref - kws
    assert re.search(r"function iLQG\(f, costfun, df_, x0, u0; kwargs\.\.\.\)", src)
    assert "BPDesc(n, m, N, 1," not in src.split("function back_pass(")[1].split("\nfunction ")[0]      # the batch axis is real
