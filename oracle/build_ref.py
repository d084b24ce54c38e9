"""Build recipe for oracle/_ref (TEST INFRASTRUCTURE): compiles the REFERENCE's own sources WHERE THEY LIE under
/root/reference into oracle/_ref/, to validate the restatements in oracle/ against the real thing.  Nothing is copied into
the repository (oracle/_ref/ is git-ignored; it travels to the GPU box like the other built artefacts, where the tests use
the prebuilt module and never read /root/reference).

Currently one module:
  mise   the reference's MISE octree refinement (code/src/libmise/mise.pyx, Cython / C++; the extractor behind
         generate_mesh, code/src/utils/meshing.py:9-72) -> oracle/_ref/mise.so.  Recipe: `cython --cplus -3` writes the
         generated C++ to oracle/_ref/mise.cpp, g++ compiles it against the CPython + numpy headers.  The reference's own build
         system (setup.py of the whole repo) is not run.

    python -m oracle.build_ref          # or __graft_entry__.build(), which calls build() when /root/reference is present
"""
from __future__ import annotations

import importlib.util
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("HOLD_REFERENCE_ROOT", "/root/reference")
MISE_PYX = os.path.join(REF, "code", "src", "libmise", "mise.pyx")
MISE_SO = os.path.join(OUT, "mise.so")


def build(force: bool = False):
    """-> path of oracle/_ref/mise.so, or None when the reference tree (or Cython) is not available"""
    if not os.path.exists(MISE_PYX):
        return MISE_SO if os.path.exists(MISE_SO) else None
    if not force and os.path.exists(MISE_SO) and os.path.getmtime(MISE_SO) >= os.path.getmtime(MISE_PYX):
        return MISE_SO
    try:
        import Cython  # noqa: F401
        import numpy as np
    except Exception:
        return None
    os.makedirs(OUT, exist_ok=True)
    cpp = os.path.join(OUT, "mise.cpp")
    subprocess.run([sys.executable, "-m", "cython", "--cplus", "-3", MISE_PYX, "-o", cpp], check=True)
    inc = [sysconfig.get_paths()["include"], np.get_include()]
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-std=c++14", "-w", *[f"-I{i}" for i in inc], cpp, "-o", MISE_SO], check=True)
    os.remove(cpp)  # generated from the reference's source: only the binary stays
    return MISE_SO


def load_mise():
    """the compiled reference module (oracle/_ref/mise.so), or None"""
    if not os.path.exists(MISE_SO):
        return None
    spec = importlib.util.spec_from_file_location("mise", MISE_SO)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


if __name__ == "__main__":
    print(build(force=True))
