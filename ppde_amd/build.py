"""Builds the gfx950 HIP library in-tree: ppde_amd/libppde_hip.so (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "ppde_api.hip")
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".hip", ".h"))) + \
       [os.path.join(os.path.dirname(HERE), "include", "ppde_hip.h")]
OUT = os.path.join(HERE, "libppde_hip.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, extra=(), out=None):
    out = out or OUT
    if out == OUT and not force and not is_stale():
        return OUT
    cmd = [hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-fno-fast-math",
           "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wl,-rpath,/opt/rocm/lib", *extra, SRC, "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else [])
    print("built", OUT)
