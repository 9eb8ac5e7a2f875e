"""Build recipe for libpcvae_hip.so (gfx950 only; hipcc cross-compiles without a GPU).

    python -m pivotcvae_amd.build [--force] [--verbose]

The library is built IN-TREE (pivotcvae_amd/lib/) so it travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libpcvae_hip.so")
SOURCES = ["error.cpp", "elementwise.hip", "gemm_f32.hip", "catalog_f32.hip", "catalog_bf16.hip", "catalog_sparse.hip", "candidate_ce.hip", "catalog_sample.hip",
           "catalog_api.cpp"]
HEADERS = ["common.h", "catalog_plan.h", "catalog_x3.h", os.path.join("..", "..", "include", "pcvae.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile every translation unit to an object, then link the shared library."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, procs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {s} ---\n{out}\n")
        elif verbose and out.strip():
            print(f"--- {s} ---\n{out}")
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
