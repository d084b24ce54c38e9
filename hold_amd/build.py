"""In-tree build of libholdhip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "libholdhip_dev.so" if os.environ.get("HOLD_DEV") == "1" else "libholdhip.so")
# -pragma-unroll-threshold: the register-resident kernels (rmlp.hip / rchain.hip) write one k step as 48 MFMA gaps with
# constant-index slices of work behind each; LLVM refuses to fully unroll `#pragma unroll` nests beyond 16 384 instructions
# (pre-simplification) and the accumulator arrays would then be indexed dynamically -- i.e. live in scratch memory
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-pragma-unroll-threshold=1048576"]
# developer build (HOLD_DEV=1 in the environment of the BUILD): adds the diagnostics kernels (csrc/dev/diag.hip) and
# compiles the timing-ablation / variant-selection environment switches into the launchers (-DHOLD_DEV).  The product
# library has neither: its entry points are stateless and read no environment variables.
DEV = os.environ.get("HOLD_DEV") == "1"


def sources():
    src = sorted(glob.glob(os.path.join(SRC_DIR, "*.hip")))
    if DEV:
        src += sorted(glob.glob(os.path.join(SRC_DIR, "dev", "*.hip")))
    return src


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(SRC_DIR, "*.h")) + glob.glob(os.path.join(_HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(s) > t for s in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    objs = []
    procs = []
    bdir = os.path.join(_HERE, "build", "dev" if DEV else "")
    os.makedirs(bdir, exist_ok=True)
    for s in sources():
        o = os.path.join(bdir, os.path.basename(s) + ".o")
        objs.append(o)
        procs.append((s, subprocess.Popen(["hipcc", *FLAGS, *(["-DHOLD_DEV"] if DEV else []), "-c", s, "-o", o],
                                          stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode())
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
