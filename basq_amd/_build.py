"""In-tree build of ``libbasq_hip.so`` (hipcc cross-compiles gfx950 without a GPU).

``python -m basq_amd._build`` or ``__graft_entry__.build()``.  The shared library is
written next to the sources (``basq_amd/csrc/libbasq_hip.so``): it is git-ignored but
travels with the tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libbasq_hip.so")
SOURCES = ["basq_hip.hip"]
DEPS = ["exp_coeffs.inc", os.path.join("..", "..", "include", "basq_hip.h")]
ARCH = "gfx950"
EXTRA_FLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + DEPS)


def build(force: bool = False, verbose: bool = True, defines: dict | None = None, out: str | None = None) -> str:
    """Build the library.  ``defines``/``out`` build a tuning variant (tools/ only; e.g. BASQ_ST, BASQ_TJ)."""
    if defines and out is None:
        raise ValueError("tuning variants (defines=...) must be written to their own file (out=...): the in-tree product library "
                         "is always built from the sources' defaults")
    target = out or LIB
    if not force and out is None and not needs_build():
        return LIB
    # -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (no v_accvgpr_read per kernel value; the block sums
    # consume every MFMA result on the VALU right away and have registers to spare)
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-Wno-unused-const-variable"] + EXTRA_FLAGS + ["-o", target] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    for k, v in (defines or {}).items():
        cmd.append(f"-D{k}={v}")
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return target


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
