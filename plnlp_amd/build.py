"""Build libplnlp_hip.so in-tree with hipcc for gfx950 (no torch headers: the
library is a plain C-ABI shared object, see include/plnlp_hip.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libplnlp_hip.so")
SOURCES = ["csr_aggregate.hip", "csr_aggregate_max.hip", "gemm_f32.hip", "edge_ops.hip", "train_ops.hip", "incidence.hip",
           "host_perm.hip", "launch_log.hip"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "plnlp_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    import time
    started = time.time()        # the library is stamped with the time its sources were READ: an edit made while the
    objs = []                    # compilers run must count as newer than the result
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    # gemm_f32.hip is compiled three times: K-tile depth 32 (with the host entry points), depth 16 (kernels
    # only) and depth 16 with the operands split into bf16 terms (kernels only)
    units = [(src, src.replace(".hip", ".o"), []) for src in SOURCES]
    units.append(("gemm_f32.hip", "gemm_f32_bk16.o", ["-DPLNLP_GEMM_BK=16"]))
    # (-fno-slp-vectorize: packed f32 adds formed from the operand split cost issue slots beside the MFMAs and
    # sent the split's residuals through scratch memory)
    units.append(("gemm_f32.hip", "gemm_f32_x3.o", ["-DPLNLP_GEMM_BK=16", "-DPLNLP_GEMM_X3=1", "-fno-slp-vectorize"]))
    units.append(("gemm_x3s.hip", "gemm_x3s.o", ["-fno-slp-vectorize"]))
    units.append(("gemm_wgw.hip", "gemm_wgw.o", ["-fno-slp-vectorize"]))
    units.append(("gemm_x3b.hip", "gemm_x3b.o", ["-fno-slp-vectorize"]))
    units.append(("aggregate_dense.hip", "aggregate_dense.o", ["-fno-slp-vectorize"]))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "plnlp_hip.h"))
    newest_header = max(os.path.getmtime(h) for h in headers)
    for src, oname, extra in units:
        obj = os.path.join(HERE, "build", oname)
        objs.append(obj)
        # a unit is recompiled when its source or any header is newer than its object (the objects carry the time their
        # sources were read, like the library)
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) >= max(os.path.getmtime(os.path.join(CSRC, src)), newest_header)):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
               os.path.join(CSRC, src), "-o", obj] + extra
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or verbose:
            print(f"--- {src} ---\n{out}", file=sys.stderr)
        failed |= p.returncode != 0
        if p.returncode == 0:
            os.utime(obj, (started, started))
    if failed:
        raise RuntimeError("hipcc failed")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    os.utime(LIB, (started, started))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
