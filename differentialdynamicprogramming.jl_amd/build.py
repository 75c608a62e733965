"""Builds libddp_amd.so (HIP, gfx950) in-tree:  python build.py [--force]

One hipcc invocation per translation unit (run in parallel), then a link step.  hipcc
cross-compiles for gfx950 without a GPU, so this also runs in the CPU-only build container.
"""
import concurrent.futures as cf
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libddp_amd.so")
SOURCES = ["capi.hip", "back_pass.hip", "back_pass_dpp.hip", "back_pass_dppw.hip", "back_pass_row.hip", "back_pass_row_hi.hip", "back_pass_mid.hip", "back_pass_big.hip", "back_pass_gps_lane.hip", "back_pass_mfma.hip", "back_pass_mfma_lims.hip", "back_pass_mf2.hip", "back_pass_mf2_lims.hip", "back_pass_mx.hip", "back_pass_mxg.hip", "back_pass_mx2.hip", "back_pass_sh.hip", "back_pass_q4.hip",
           "forward_pass.hip", "forward_pass_dpp.hip", "forward_pass_row.hip", "forward_pass_pipe.hip", "forward_pass_big.hip", "df.hip", "ilqg.hip", "kl.hip", "comm.hip", "boxqp_big.hip"]
HEADERS = ["ddp_internal.h", "boxqp_dev.h", "boxqp_rows.h", "arena.h", os.path.join("..", "..", "include", "ddp_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Rpass-analysis=kernel-resource-usage", "-Wall", "-Wno-unused-function",
         "-Wno-unused-but-set-variable", "-Wno-unused-variable"]


# MFMA accumulators stay in the (unified) VGPR file: no v_accvgpr_read/write shuffles around every product
EXTRA_FLAGS = {"back_pass_mx.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "back_pass_mxg.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "back_pass_mx2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "back_pass_sh.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "back_pass_mid.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "back_pass_q4.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "back_pass_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "back_pass_mf2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "back_pass_mf2_lims.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


EXTRA_DEPS = {"back_pass_mf2.hip": ["back_pass_mf2_kernel.h"], "back_pass_mf2_lims.hip": ["back_pass_mf2_kernel.h"], "back_pass_row_hi.hip": ["back_pass_row.hip"], "back_pass_mfma.hip": ["back_pass_mfma_kernel.h"], "back_pass_mfma_lims.hip": ["back_pass_mfma_kernel.h"],
              "back_pass_mx.hip": ["back_pass_mx_common.h"], "back_pass_mxg.hip": ["back_pass_mx_common.h"], "back_pass_mx2.hip": ["back_pass_mx_common.h"], "back_pass_sh.hip": ["back_pass_mx_common.h"],
              "forward_pass_dpp.hip": ["pend_math.h"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


_digests = {}


def _code_digest(path):
    """digest of a header's CODE (comments and white space removed): an edit to a comment of include/ddp_amd.h does not recompile
    eighteen translation units (four minutes)"""
    if path not in _digests:
        txt = open(path).read()
        txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
        txt = re.sub(r"//[^\n]*", " ", txt)
        _digests[path] = hashlib.sha1(" ".join(txt.split()).encode()).hexdigest()
    return _digests[path]


def _save_usage(obj, stderr):
    """registers, scratch and LDS of every kernel of a translation unit, from the compiler's kernel-resource-usage remarks ->
    build/<unit>.usage.json (tests/test_capi_cpu.py: the kernels that count their own vector-memory operations — s_waitcnt vmcnt(N)
    with N from the number of loads they issue — must not spill to scratch: a spill is a vector-memory operation the count does not know)"""
    import json
    out, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass-analysis", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    json.dump(out, open(obj + ".usage.json", "w"), indent=0)


def _compile(src):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    hdrs = [os.path.join(CSRC, h) for h in HEADERS + EXTRA_DEPS.get(src, [])]
    stamp = obj + ".hdr"
    want = "\n".join("%s %s" % (os.path.basename(h), _code_digest(h)) for h in hdrs) + "\n" + " ".join(FLAGS + EXTRA_FLAGS.get(src, []))
    have = open(stamp).read() if os.path.exists(stamp) else None
    if _stale(obj, [os.path.join(CSRC, src)]) or have != want:
        cmd = ["hipcc"] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        _save_usage(obj, r.stderr)
        open(stamp, "w").write(want)
        return src, True
    return src, False


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs]
    if any(changed for _, changed in results) or _stale(LIB, objs):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    if verbose:
        print("libddp_amd.so:", ", ".join("%s%s" % (s, "*" if c else "") for s, c in results))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
