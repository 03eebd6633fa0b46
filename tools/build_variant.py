#!/usr/bin/env python3
"""A/B builds of ONE kernel file: compile it with extra -D flags and link it with the current objects of everything else into
build/variants/<name>/libvgpu.so (picked up through VGPU_LIB_PATH; tools/gpu_session.sh alternates candidates "label=LIB:build/variants/<name>/libvgpu.so" inside one GPU session).

    python tools/build_variant.py <name> kernels/ntt.hip|ALL -DVGPU_MID12_WAVES=4 [-D...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from valida_amd import build as b  # noqa: E402


def main():
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    b.build_vgpu()  # the baseline objects
    out_dir = os.path.join(b.BUILD, "variants", name)
    os.makedirs(out_dir, exist_ok=True)
    srcs = list(b.SOURCES) if src == "ALL" else [src]  # ALL: a switch in a shared header (field.hpp)
    procs = []
    for one in srcs:
        obj = os.path.join(out_dir, one.replace("/", "_") + ".o")
        procs.append(subprocess.Popen(["hipcc"] + b.FLAGS + flags + ["-x", "hip", "-c", os.path.join(b.CSRC, one), "-o", obj]))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed")
    objs = [os.path.join(out_dir if s in srcs else b.BUILD, s.replace("/", "_") + ".o") for s in b.SOURCES]
    lib = os.path.join(out_dir, "libvgpu.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"], check=True)
    print(lib)


if __name__ == "__main__":
    main()
