"""Build liblinefront.so (HIP, gfx950) in-tree with hipcc.  No CPU fallback is ever built.

    python -m lineslam_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblinefront.so")

# -ffp-contract=off: the reference arithmetic is IEEE double without FMA contraction; hipcc
# contracts to v_fma_f64 by default.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "linefront.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def _build_micro(hipcc, force):
    """tools/micro/clock.hip -> tools/micro/clock_cal: the clock / fp64-rate calibration bench.py reports next to its line
    (a stand-alone binary, not part of the library; bench.py goes on without it)."""
    micro = os.path.join(os.path.dirname(HERE), "tools", "micro")
    src, exe = os.path.join(micro, "clock.hip"), os.path.join(micro, "clock_cal")
    if os.path.exists(src) and (force or not os.path.exists(exe) or os.path.getmtime(src) > os.path.getmtime(exe)):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-w", src, "-o", exe], check=False)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    _build_micro(hipcc, force)
    if not force and not _stale():
        return LIB
    cmd = [hipcc] + FLAGS + os.environ.get("LF_EXTRA_CFLAGS", "").split() + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
