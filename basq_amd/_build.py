"""In-tree build of ``libbasq_hip.so`` (hipcc cross-compiles gfx950 without a GPU).

``python -m basq_amd._build`` or ``__graft_entry__.build()``.  Three translation units -- the pairwise-kernel family, the dense
linear algebra, the per-round reductions -- are compiled side by side (objects under ``basq_amd/csrc/build/``, only the stale
ones again) and linked into ``basq_amd/csrc/libbasq_hip.so``: git-ignored, but it travels with the tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libbasq_hip.so")
OBJ_DIR = os.path.join(CSRC, "build")
SOURCES = ["basq_pairwise.hip", "basq_linalg.hip", "basq_reduction.hip"]
COMMON_DEPS = ["basq_common.hpp", os.path.join("..", "..", "include", "basq_hip.h")]
DEPS = {"basq_pairwise.hip": ["exp_coeffs.inc"]}
ARCH = "gfx950"
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (no v_accvgpr_read per kernel value; the block sums
# consume every MFMA result on the VALU right away and have registers to spare)
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-const-variable",
         "-mllvm", "-amdgpu-mfma-vgpr-form",
         # every kernel's registers / scratch / occupancy, kept beside the object (build/<unit>.resources.txt) and checked by
         # tests/test_kernel_resources.py: two registers too many halve a kernel's occupancy without a word from the compiler
         "-Rpass-analysis=kernel-resource-usage"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _mtime(rel):
    return os.path.getmtime(os.path.join(CSRC, rel))


def _obj(src, obj_dir=OBJ_DIR):
    return os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")


def resources_path(src, obj_dir=OBJ_DIR):
    return os.path.join(obj_dir, os.path.splitext(src)[0] + ".resources.txt")


def _stale(src, obj_dir=OBJ_DIR):
    obj = _obj(src, obj_dir)
    if not os.path.exists(obj) or not os.path.exists(resources_path(src, obj_dir)):
        return True
    t = os.path.getmtime(obj)
    return any(_mtime(f) > t for f in [src] + COMMON_DEPS + DEPS.get(src, []))


def _all_inputs():
    return SOURCES + COMMON_DEPS + [d for ds in DEPS.values() for d in ds]


def source_hash() -> str:
    """sha256 over the kernel sources, the shared header, the generated table AND the compiler flags: names the binary a counter
    pass or a resource report was taken on (``tools/pmc_summary.py`` stores it, ``bench.py`` compares it with the tree's)."""
    import hashlib

    h = hashlib.sha256()
    for rel in sorted(_all_inputs()):
        h.update(os.path.basename(rel).encode() + b"\0")
        with open(os.path.join(CSRC, rel), "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


STAMP = os.path.join(CSRC, "libbasq_hip.stamp")              # source_hash() of the tree the in-tree library was built from


def _stamp_matches() -> bool:
    try:
        with open(STAMP) as f:
            return f.read().strip() == source_hash()
    except OSError:
        return False


def needs_build() -> bool:
    """The LIBRARY against its sources and flags only (the objects and resource reports under ``build/`` do not travel to the GPU
    box: their absence there must not trigger a rebuild over a library a running process has mapped)."""
    if not os.path.exists(LIB):
        return True
    if os.path.exists(STAMP):
        return not _stamp_matches()                              # sources or FLAGS changed since the library was linked
    t = os.path.getmtime(LIB)
    return any(_mtime(f) > t for f in _all_inputs())


def kernel_resources(obj_dir=OBJ_DIR):
    """-> ``{kernel (mangled): {vgprs, agprs, scratch, occupancy, lds, sgprs, unit}}`` from the build's resource reports."""
    import re

    out = {}
    for src in SOURCES:
        path = resources_path(src, obj_dir)
        if not os.path.exists(path):
            continue
        name = None
        for ln in open(path):
            ln = ln.strip()
            m = re.match(r"Function Name: (\S+)", ln)
            if m:
                name = m.group(1)
                out[name] = {"unit": src}
                continue
            if name is None:
                continue
            for key, pat in (("sgprs", r"TotalSGPRs: (\d+)"), ("vgprs", r"^VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"),
                             ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"),
                             ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)")):
                m = re.search(pat, ln)
                if m:
                    out[name][key] = int(m.group(1))
    return out


def build(force: bool = False, verbose: bool = True, defines: dict | None = None, out: str | None = None) -> str:
    """Build the library.  ``defines``/``out`` build a tuning variant (tools/ only; e.g. BASQ_ST, BASQ_TJ)."""
    if defines and out is None:
        raise ValueError("tuning variants (defines=...) must be written to their own file (out=...): the in-tree product library "
                         "is always built from the sources' defaults")
    target = out or LIB
    if not force and out is None and not needs_build():
        return LIB
    # a variant keeps its objects apart from the product's
    obj_dir = OBJ_DIR if out is None else os.path.join(OBJ_DIR, "variant_" + os.path.splitext(os.path.basename(out))[0])
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = hipcc_path()
    dflags = [f"-D{k}={v}" for k, v in (defines or {}).items()]

    def compile_one(src):
        cmd = [hipcc] + FLAGS + dflags + ["-c", "-o", _obj(src, obj_dir), os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
        remarks = [ln for ln in r.stderr.splitlines() if "[-Rpass-analysis=kernel-resource-usage]" in ln]
        other = [ln for ln in r.stderr.splitlines() if "[-Rpass-analysis=kernel-resource-usage]" not in ln
                 and not ln.lstrip().startswith(("|", "^")) and not ln[:6].strip().isdigit()]
        if r.returncode != 0 or any("warning:" in ln or "error:" in ln for ln in other):
            sys.stderr.write("\n".join(other) + "\n")
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        with open(resources_path(src, obj_dir), "w") as f:
            f.write("\n".join(ln.split("remark: ", 1)[1].replace(" [-Rpass-analysis=kernel-resource-usage]", "") for ln in remarks) + "\n")

    todo = [s for s in SOURCES if force or out is not None or _stale(s, obj_dir)]
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        list(ex.map(compile_one, todo))
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", target] + [_obj(s, obj_dir) for s in SOURCES]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True, cwd=CSRC)
    if out is None:
        with open(STAMP, "w") as f:
            f.write(source_hash() + "\n")
    return target


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
