"""Structural lint of the shipped Julia sources (no Julia in the build image, so nothing can `include` them here).

`tests/test_julia_binding.py` checks every `@ccall` against include/ddp_amd.h with regular expressions; a file whose functions sit INSIDE
string literals (round 4: a lost triple quote put `iLQG_queue` / `iLQG_mpc` into `install!`'s docstring — a ParseError for
`include("DDPAmd.jl")`) passes that check.  This one tokenises the files with pygments' JuliaLexer and verifies the structure a parser
would need:

 (i)   no definition text (`function name(`, `struct Name`, `@ccall lib.f(`) inside a string token;
 (ii)  every top-level triple-quoted block is a docstring: the next token starts a definition;
 (iii) block openers and `end` keywords nest and balance (with `end` inside `[...]` read as an index and `for` / `if` inside brackets
       read as generator clauses), brackets balance, every string is closed;
 (iv)  the public names the reference's users call are defined at top level of `module DDPAmd` (iLQG keeps the signature of
       /root/reference/src/iLQG.jl:143-163).
"""
import glob
import os
import re

import pytest
from pygments.lexers import JuliaLexer
from pygments.token import Comment, Error, Keyword, Name, Punctuation, String, Text, Whitespace

from conftest import ROOT

JL_FILES = sorted(glob.glob(os.path.join(ROOT, "differentialdynamicprogramming.jl_amd", "julia", "*.jl"))
                  + glob.glob(os.path.join(ROOT, "julia", "*.jl")) + glob.glob(os.path.join(ROOT, "bench", "*.jl")))

OPENERS = {"function", "macro", "begin", "let", "do", "quote", "try", "struct", "module", "baremodule", "while", "for", "if"}
SOFT = {"for", "if", "while"}            # generator / filter clauses when met deeper in brackets than the enclosing block
DEF_IN_STRING = re.compile(r"^\s*(?:function\s+[\w.!]+\s*\(|(?:mutable\s+)?struct\s+\w+\s*$|module\s+\w+\s*$)|@ccall\s+\w+\.\w+\(", re.M)


def tokens(src):
    """(type, text, line) of every token"""
    line = 1
    for t, v in JuliaLexer().get_tokens(src):
        yield t, v, line
        line += v.count("\n")


def lint(src):
    """list of 'line N: problem' — empty for a well-formed file"""
    problems = []
    toks = [(t, v, ln) for t, v, ln in tokens(src)]

    # strings: pygments emits the delimiter as its own String token; an unterminated literal runs to the end of the file
    in_str = None            # (delimiter, line)
    string_body = []
    code = []                # the tokens outside strings, with triple-quoted blocks collapsed to one ('DOC', text, line) entry
    depth_interp = 0         # inside $( ... ) of a string the lexer returns to code tokens; we do not lint those
    for t, v, ln in toks:
        if t in String and t is not String.Symbol and t is not String.Char and t not in String.Interpol:
            if in_str is None and v in ('"""', '"', "`", "```"):
                in_str = (v, ln); string_body = []
                continue
            if in_str is not None and v == in_str[0] and depth_interp == 0:
                body = "".join(string_body)
                m = DEF_IN_STRING.search(body)
                if m:
                    problems.append(f"line {in_str[1] + body[:m.start()].count(chr(10))}: definition text inside a string literal: "
                                    f"{m.group(0).strip()!r}")
                code.append(("DOC" if in_str[0] == '"""' else "STR", body, in_str[1]))
                in_str = None
                continue
            if in_str is not None:
                string_body.append(v)
                continue
            code.append((t, v, ln))        # a string-like token outside a literal (prefix such as r"..." is emitted whole)
            continue
        if in_str is not None:
            # code inside $( ... ) interpolation
            if t in String.Interpol:
                continue
            if v == "(":
                depth_interp += 1
            elif v == ")":
                depth_interp -= 1
            string_body.append(v)
            continue
        if t in Error:
            problems.append(f"line {ln}: lexer error at {v!r}")
        code.append((t, v, ln))
    if in_str is not None:
        problems.append(f"line {in_str[1]}: string opened with {in_str[0]} is never closed")

    # structure
    stack = []               # ('block', keyword, bracket depth, line) and ('br', char, line)
    closing = {")": "(", "]": "[", "}": "{"}

    def bdepth():
        return sum(1 for e in stack if e[0] == "br")

    sig = [(t, v, ln) for t, v, ln in code if not (t in Comment or t in Whitespace or (t in Text and not v.strip()))]
    prev_kw = None
    for i, (t, v, ln) in enumerate(sig):
        if t == "DOC":
            if not any(e[0] == "block" and e[1] not in ("module", "baremodule") for e in stack) and bdepth() == 0:
                # top level (possibly inside `module`): must document something
                nxt = sig[i + 1] if i + 1 < len(sig) else (None, "", ln)
                ok = (nxt[0] in Keyword and nxt[1] in ("function", "struct", "mutable", "macro", "module", "const", "abstract", "primitive")) \
                    or nxt[0] in Name or nxt[0] is Name.Decorator or nxt[0] in Keyword.Type
                prv = sig[i - 1] if i > 0 else (None, "", 0)
                is_value = prv[1] in ("=", "(", ",", "return") or prv[0] is Name.Decorator     # a triple-quoted VALUE, not a docstring
                if not ok and not is_value:
                    problems.append(f"line {ln}: top-level triple-quoted block is not followed by a definition (next: {nxt[1]!r})")
                if nxt[0] == "DOC":
                    problems.append(f"line {ln}: two triple-quoted blocks in a row")
            continue
        if t == "STR":
            continue
        if t in Punctuation or v in "()[]{}":
            if v in "([{" and len(v) == 1:
                stack.append(("br", v, ln))
            elif v in ")]}" and len(v) == 1:
                if not stack or stack[-1][0] != "br" or stack[-1][1] != closing[v]:
                    problems.append(f"line {ln}: {v!r} closes {stack[-1][1:] if stack else 'nothing'}")
                    if stack and stack[-1][0] == "br":
                        stack.pop()
                else:
                    stack.pop()
            continue
        if t in Keyword and t not in Keyword.Type:
            if v == "end":
                if stack and stack[-1][0] == "br":
                    if stack[-1][1] != "[":
                        problems.append(f"line {ln}: `end` inside {stack[-1][1]!r} opened on line {stack[-1][2]}")
                    continue                                   # a[end]
                if not stack:
                    problems.append(f"line {ln}: `end` without an opener")
                else:
                    stack.pop()
            elif v in OPENERS:
                if v == "struct" and prev_kw == ("mutable", i - 1):
                    pass
                enclosing = 0
                for e in reversed(stack):
                    if e[0] == "block":
                        enclosing = e[2]; break
                d = bdepth()
                after_semicolon = i > 0 and sig[i - 1][1] == ";"                # (a = 1; for c in cs; ...; end; a): a statement
                if v in SOFT and d > enclosing and not after_semicolon:
                    pass                                       # [x for x in xs if p(x)]
                else:
                    stack.append(("block", v, d, ln))
            elif v in ("abstract", "primitive"):
                # `abstract type X end`
                stack.append(("block", v, bdepth(), ln))
            prev_kw = (v, i)
    for e in stack:
        problems.append(f"line {e[-1]}: {e[1]!r} is never closed")
    return problems


def top_level_functions(src):
    """names defined by `function name(` directly inside the outermost module (or at file level)"""
    names, depth = set(), 0
    sig = [(t, v) for t, v, _ in tokens(src) if not (t in Comment or t in Whitespace or (t in Text and not v.strip()) or t in String)]
    br = 0
    for i, (t, v) in enumerate(sig):
        if v in ("(", "[", "{"):
            br += 1
        elif v in (")", "]", "}"):
            br -= 1
        elif t in Keyword and t not in Keyword.Type:
            if v == "end" and br == 0:
                depth -= 1
            elif v in OPENERS and not (v in SOFT and br > 0):
                if v == "function" and depth <= 1 and i + 1 < len(sig):
                    j = i + 1
                    name = sig[j][1]
                    while j + 2 < len(sig) and sig[j + 1][1] == ".":       # Base.show
                        name = sig[j + 2][1]; j += 2
                    names.add(name)
                depth += 1
    return names


def test_the_linter_sees_the_round4_breakage():
    """the shape of commit a91704b: a function pasted between a docstring's text and its closing quotes"""
    broken = '''module M
"""
    install!(ref)

text
"""
    queue(p) -> x

more text with `backticks`...
"""
function queue(p)
    @ccall lib.f(p::Ptr{Cvoid})::Cint
end

"""
function install!(ref)
    ref
end
end
'''
    probs = lint(broken)
    assert probs, "a function inside a string literal went unnoticed"
    assert any("inside a string literal" in p for p in probs), probs
    good = '''module M
"""
    queue(p) -> x
"""
function queue(p)
    y = [i for i in p if i > p[end]]
    @ccall lib.f(p::Ptr{Cvoid})::Cint
end
mutable struct A; x::Int; end
const S = """
value, not a docstring
"""
f(x) = x[end] + 1
end
'''
    assert lint(good) == []
    assert lint("function f(x)\n  if x\n  end\n") != []                      # missing end
    assert lint('x = "abc\n') != []                                          # unterminated string
    assert lint('"""\ndoc\n"""\n\n1 + 1\n') != []                            # docstring documenting nothing
    assert lint("f(x) = (x[1]\n") != []                                      # unbalanced bracket


@pytest.mark.parametrize("path", JL_FILES, ids=[os.path.relpath(p, ROOT) for p in JL_FILES])
def test_julia_file_is_structurally_sound(path):
    probs = lint(open(path, encoding="utf-8").read())
    assert probs == [], "\n".join(probs)


def test_all_four_julia_files_are_linted():
    rel = {os.path.relpath(p, ROOT) for p in JL_FILES}
    assert {"differentialdynamicprogramming.jl_amd/julia/DDPAmd.jl", "differentialdynamicprogramming.jl_amd/julia/test_result_lifetime.jl",
            "julia/make_reference_fixtures.jl", "bench/reference_julia.jl"} <= rel


def test_public_entry_points_are_real_definitions():
    """`iLQG(f,costfun,df,x0,u0; lims...)` (iLQG.jl:143-163) and the batched / queue / MPC / install! entries must be FUNCTIONS of the module,
    not text: the round-4 file 'defined' iLQG_queue and iLQG_mpc inside a string."""
    src = open(os.path.join(ROOT, "differentialdynamicprogramming.jl_amd", "julia", "DDPAmd.jl"), encoding="utf-8").read()
    names = top_level_functions(src)
    for want in ("iLQG", "iLQG_queue", "iLQG_mpc", "install!", "back_pass", "forward_pass", "boxQP"):
        assert want in names, f"{want} is not defined at top level of DDPAmd.jl (found: {sorted(names)})"
    # the keyword names of the reference's signature that the registered-problem iLQG must keep
    m = re.search(r"^function iLQG\((.*?)\)\n", src, flags=re.S | re.M)
    assert m, "function iLQG(...) not found"
    for kw in ("lims", "α", "tol_fun", "tol_grad", "max_iter", "λ", "dλ", "λfactor", "λmax", "λmin", "regType", "reduce_ratio_min", "diff_fun"):
        assert re.search(rf"\b{kw}\s*=", m.group(1)), f"iLQG lost its keyword {kw}"


# ---- install! is a SAFE drop-in (missing 3) ------------------------------------------------------------------------------
# the annotations of the reference's three linear-system methods, /root/reference/src/backward_pass.jl:162,179,217 (cxx, fx)
REF_BACK_PASS_ANNOTATIONS = {("AbstractArray{T,2}", "AbstractArray{T,3}"), ("AbstractArray{T,3}", "AbstractArray{T,3}"),
                             ("AbstractArray{T,2}", "AbstractMatrix{T}")}


def _ddpamd_src():
    return open(os.path.join(ROOT, "differentialdynamicprogramming.jl_amd", "julia", "DDPAmd.jl"), encoding="utf-8").read()


def _function_body(src, name):
    m = re.search(rf"^function {re.escape(name)}\(.*?^end\n", src, flags=re.S | re.M)
    assert m, f"function {name} not found"
    return m.group(0)


def test_install_adds_narrower_methods_and_keeps_the_reference_ones():
    """`install!` must not overwrite backward_pass.jl:162,179,217: its three methods carry annotations that are strictly narrower than the
    reference's (concrete Float64 array types), so the reference's own `AbstractArray{T}` methods stay the fallback."""
    body = _function_body(_ddpamd_src(), "install!")
    sigs = re.findall(r"^\s*back_pass\(cx, cu, cxx::(.+?), cxu, cuu, fx::(.+?), fu, λ, regType, lims, x, u\)\s*=", body, flags=re.M)
    assert len(sigs) == 3, sigs
    for cxx_t, fx_t in sigs:
        assert (cxx_t.strip(), fx_t.strip()) not in REF_BACK_PASS_ANNOTATIONS, "install! would OVERWRITE a reference method: %s / %s" % (cxx_t, fx_t)
        assert "AbstractArray{T" not in cxx_t + fx_t and "AbstractMatrix{T" not in fx_t
        assert "Float64" in fx_t and ("Float64" in cxx_t or cxx_t.strip().startswith("$Dense")), (cxx_t, fx_t)
    assert " where " not in "".join(l for l in body.splitlines() if l.lstrip().startswith("back_pass(")), "no type parameter: Float64 only"
    src = _ddpamd_src()
    assert re.search(r"^const DenseCost2 = Union\{Matrix\{Float64\},\s*Diagonal\{Float64", src, flags=re.M)
    assert re.search(r"^const DenseCost3 = Array\{Float64,3\}", src, flags=re.M)


def test_install_has_a_fallback_to_the_reference_methods():
    """n > 64, m > 8, a non-Float64 operand or a shape the library refuses (DDPError) must reach the reference's method through `invoke`
    with the REFERENCE's declared signature, and `uninstall!` must delete what `install!` added."""
    src = _ddpamd_src()
    body = _function_body(src, "install!")
    assert "invoke(ref.back_pass, _ref_sig(kind)" in body
    assert "_gpu_takes(" in body and "fallback()" in body
    assert re.search(r"err isa DDPError \|\| rethrow\(\)", body)
    takes = _function_body(src, "_gpu_takes")
    assert "MAX_N" in takes and "MAX_M" in takes and "AbstractArray{Float64}" in takes
    assert re.search(r"^const MAX_N = 64\b", src, flags=re.M) and re.search(r"^const MAX_M = 8\b", src, flags=re.M)
    # the three invoke signatures are the reference's (Float64 instances of its AbstractArray annotations)
    for kind, cxx_t, fx_t in ((":ltv_ti", "AbstractArray{Float64,2}", "AbstractArray{Float64,3}"),
                              (":ltv_tv", "AbstractArray{Float64,3}", "AbstractArray{Float64,3}"),
                              (":lti", "AbstractArray{Float64,2}", "AbstractMatrix{Float64}")):
        m = re.search(rf"^_ref_sig\(::Val{{{kind}}}\)\s*=\s*Tuple{{Any,Any,([^,]+,\d+}}|[^,]+}}),Any,Any,([^,]+,\d+}}|[^,]+}}),", src, flags=re.M)
        assert m and m.group(1) == cxx_t and m.group(2) == fx_t, (kind, m and m.groups())
    un = _function_body(src, "uninstall!")
    assert "Base.delete_method" in un and "empty!(_installed)" in un
    names = top_level_functions(src)
    assert "uninstall!" in names and "_gpu_takes" in names
    assert re.search(r"^struct DDPError <: Exception", src, flags=re.M)
    assert "throw(DDPError(" in src


def test_back_pass_does_not_guess_the_batch_axis():
    """a batched call with B == N and per-trajectory time-invariant dynamics was read as time-varying (`B != N` guess): now a keyword"""
    body = _function_body(_ddpamd_src(), "back_pass")
    assert "B != N" not in body
    assert "batched_dynamics::Union{Nothing,Bool}=nothing" in body and "batched_cost::Union{Nothing,Bool}=nothing" in body
    assert "fx_batched = batched_dynamics === nothing ? ndims(fx) == 4 : batched_dynamics" in body
