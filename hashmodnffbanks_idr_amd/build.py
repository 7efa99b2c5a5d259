"""Builds libhashmod.so (hand-written HIP for gfx950 behind the C ABI of include/hashmod.h).

    python -m hashmodnffbanks_idr_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is kept in-tree (git-ignored) so that it travels
to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libhashmod.so")
SOURCES = ["hm_api.hip", "hm_encode.hip", "hm_sdf.hip", "hm_gemm.hip", "hm_trace.hip", "hm_elem.hip", "hm_pack.hip", "hm_optim.hip", "hm_loss.hip", "hm_nffb.hip", "hm_sdf_bf16.hip", "hm_encode_dx.hip", "hm_exchange.hip", "hm_sdf_split.hip", "hm_sort.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-DHM_BUILDING=1", "-I", os.path.join(ROOT, "include")]


def _deps():
    out = [os.path.join(ROOT, "include", "hashmod.h"), os.path.abspath(__file__)]
    for f in os.listdir(CSRC):
        out.append(os.path.join(CSRC, f))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), src))
        objs.append(obj)
    for p, src in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
