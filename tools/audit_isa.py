"""Audit the generated ISA of the scalar-broadcast block-sum kernel (run on the build host, no GPU).

The candidate rows are loaded with inline-asm ``s_load_dwordx8`` into SGPR tuples that hipcc does not know
to be in flight (cdna_hip_programming.md §5.7).  That is only safe if no instruction touches (reads, copies, spills or
overwrites) a destination SGPR between its load and the following ``s_waitcnt lgkmcnt(0)``, and the kernel
uses no scratch.  This script compiles the library source to assembly and checks exactly that, instruction by
instruction, for every instantiation of ``blocksum_valu_kernel``.

    python tools/audit_isa.py            # exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "basq_amd", "csrc", "basq_hip.hip")


def compile_to_asm(workdir):
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-save-temps", SRC, "-o", os.devnull],
                   check=True, cwd=workdir, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(workdir):
        if f.endswith("gfx950.s"):
            return open(os.path.join(workdir, f)).read()
    raise RuntimeError("no device assembly produced")


SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def audit(asm: str):
    """Hazard check: between an inline-asm s_load and the next s_waitcnt lgkmcnt(0), no instruction may name
    (read, copy, spill or overwrite) a destination SGPR of a load that is still in flight."""
    problems, seen = [], 0
    for m in re.finditer(r"^(_Z20blocksum_valu_kernel\w+):[^\n]*\n(.*?)\n\s*s_endpgm", asm, re.S | re.M):
        name, body = m.group(1), m.group(2)
        seen += 1
        inflight = set()
        n_loads = 0
        for ln, line in enumerate(body.split("\n")):
            code = line.split(";")[0].strip()
            if not code or code.endswith(":") or code.startswith("."):
                continue                       # labels / directives / comments; branches end a basic block but the
            op = code.split()[0]               # kernel issues no load across one (every path waits first)
            if op.startswith("s_load_dword"):
                dst = code.split(None, 1)[1].split(",")[0]
                used = sregs(code.split(",", 1)[1])
                hit = used & inflight
                if hit:
                    problems.append(f"{name}: line {ln}: load address uses in-flight SGPRs {sorted(hit)[:4]}: {code}")
                inflight |= sregs(dst)
                n_loads += 1
                continue
            if op == "s_waitcnt":
                if "lgkmcnt(0)" in code or "lgkmcnt" not in code and "vmcnt" not in code:
                    inflight.clear()
                continue
            hit = sregs(code) & inflight
            if hit:
                problems.append(f"{name}: line {ln}: touches in-flight SGPRs {sorted(hit)[:4]}: {code}")
        if n_loads == 0:
            problems.append(f"{name}: no scalar loads found")
    for m in re.finditer(r"\.name:\s+(_Z20blocksum_valu_kernel\w+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s*(\d+)", asm):
        if int(m.group(2)) != 0:
            problems.append(f"{m.group(1)}: uses {m.group(2)} B of scratch")
    return seen, problems


def main():
    with tempfile.TemporaryDirectory() as d:
        asm = compile_to_asm(d)
    seen, problems = audit(asm)
    print(f"audited {seen} blocksum_valu_kernel instantiations")
    for p in problems:
        print("PROBLEM:", p)
    return 1 if problems or seen == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
