"""Builds haf_grasping_amd/libhafgrasp.so (gfx950 only) with hipcc.  In-tree, so the .so travels to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhafgrasp.so")
SOURCES = ["kernels.hip", "screen.hip", "engine.cpp", "parsers.cpp"]
# per-file extra flags (screen.hip: see its header)
EXTRA = {"screen.hip": ["-fno-slp-vectorize"]}
HEADERS = ["kernels.h", "parsers.h", "decq.h", os.path.join("..", "..", "include", "hafgrasp.h"),
           os.path.join("..", "cli", "haf_grasp_cli.cpp")]
# -ffp-contract=off: the bit-exact stages spell out every rounding; nothing may be fused behind their back
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-result", "-Wno-inline-asm"]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # ROS-free command line front end (C++ host code over the C-ABI)
    cli = os.path.join(HERE, "haf_grasp_cli")
    cmd = [hipcc, "-O2", "-std=c++17", os.path.join(HERE, "cli", "haf_grasp_cli.cpp"), "-o", cli, "-L" + HERE, "-lhafgrasp",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
