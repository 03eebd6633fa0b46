"""Build libvgpu.so (HIP/gfx950 kernels + host prover + C ABI) in-tree, and the oracle library.

hipcc cross-compiles for gfx950 without a GPU.  Objects go to build/, the shared library next to this
file (git-ignored, but shipped to the GPU box by gpurun).
"""
import concurrent.futures
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "valida_amd", "csrc")
BUILD = os.path.join(ROOT, "build")
LIB = os.path.join(ROOT, "valida_amd", "libvgpu.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")

SOURCES = [
    "kernels/ntt.hip",
    "kernels/layout.hip",
    "kernels/merkle.hip",
    "kernels/perm.hip",
    "kernels/quotient.hip",
    "kernels/open.hip",
    "host/prover.cpp",
    "capi.cpp",
]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _newer_than(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _all_headers():
    out = []
    for base, _, files in os.walk(CSRC):
        for f in files:
            if f.endswith((".hpp", ".h")):
                out.append(os.path.join(base, f))
    out.append(os.path.join(ROOT, "include", "vgpu.h"))
    return out


def _compile(src):
    obj = os.path.join(BUILD, src.replace("/", "_") + ".o")
    path = os.path.join(CSRC, src)
    if _newer_than(obj, [path] + _all_headers()):
        return obj
    cmd = ["hipcc"] + FLAGS + ["-x", "hip", "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
    return obj


def build_vgpu(force=False):
    os.makedirs(BUILD, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if not force and _newer_than(LIB, srcs + _all_headers()):
        return LIB
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(_compile, SOURCES))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return LIB


def build_oracle(force=False):
    deps = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".hpp", ".cpp"))]
    deps += [os.path.join(CSRC, "chips", "basic_machine.hpp"), os.path.join(CSRC, "air", "builder.hpp")]
    if not force and _newer_than(ORACLE_LIB, deps):
        return ORACLE_LIB
    r = subprocess.run(["make", "-C", ORACLE_DIR, "-B"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stderr)
    return ORACLE_LIB


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_vgpu(force))
    print(build_oracle(force))
