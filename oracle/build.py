"""ORACLE — TEST INFRASTRUCTURE ONLY.  Builds oracle/liboracle.so (g++ -O3 -fopenmp) via the Makefile."""
import os
import subprocess

ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(ORACLE_DIR)
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")


def build_oracle(force=False):
    deps = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".hpp", ".cpp"))]
    if not force and os.path.exists(ORACLE_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(ORACLE_LIB) for d in deps):
        return ORACLE_LIB
    r = subprocess.run(["make", "-C", ORACLE_DIR, "-B"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stderr)
    return ORACLE_LIB


if __name__ == "__main__":
    print(build_oracle(True))
