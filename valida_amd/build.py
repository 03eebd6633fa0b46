"""Build libvgpu.so (HIP/gfx950 kernels + host prover + C ABI) in-tree.

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

SOURCES = [
    "kernels/ntt.hip",
    "kernels/layout.hip",
    "kernels/merkle.hip",
    "kernels/poseidon_mmcs.hip",
    "kernels/perm.hip",
    "kernels/quotient.hip",
    "kernels/open.hip",
    "kernels/tracegen.hip",
    "host/prover.cpp",
    "host/sharded_prover.cpp",
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
            if f.endswith((".hpp", ".h", ".inc")):
                out.append(os.path.join(base, f))
    out.append(os.path.join(ROOT, "include", "vgpu.h"))
    return out


LAST = {"compiled": [], "reused": [], "linked": False}  # what the last build_vgpu() call did (printed by __graft_entry__.build)
_FORCE = False


def _compile(src):
    obj = os.path.join(BUILD, src.replace("/", "_") + ".o")
    path = os.path.join(CSRC, src)
    if not _FORCE and _newer_than(obj, [path] + _all_headers()):
        LAST["reused"].append(src)
        return obj
    LAST["compiled"].append(src)
    cmd = ["hipcc"] + FLAGS + ["-x", "hip", "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
    return obj


def build_vgpu(force=False):
    """Returns the library's path.  force=True recompiles every source (hipcc --offload-arch=gfx950) and relinks; otherwise objects newer
    than their source and every header are reused.  LAST records which of the two happened to each source."""
    global _FORCE
    os.makedirs(BUILD, exist_ok=True)
    LAST["compiled"], LAST["reused"], LAST["linked"] = [], [], False
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if not force and _newer_than(LIB, srcs + _all_headers()):
        LAST["reused"] = list(SOURCES)
        return LIB
    _FORCE = bool(force)
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(_compile, SOURCES))
    # linked beside the library and renamed onto it: a process that has the old libvgpu.so mapped keeps its (unlinked) file instead of seeing the
    # linker rewrite the pages under it; librccl.so is dlopen'ed on first use (host/comm.hpp)
    tmp = os.path.join(BUILD, "libvgpu.%d.so" % os.getpid())
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise RuntimeError("link failed:\n" + r.stderr)
    os.replace(tmp, LIB)
    LAST["linked"] = True
    return LIB


if __name__ == "__main__":
    print(build_vgpu("--force" in sys.argv))
