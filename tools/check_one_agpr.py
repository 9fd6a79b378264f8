#!/usr/bin/env python3
"""rollout_one_kernel keeps a chain's factor in AGPRs a[0:243] as C++ doubles that are only ever touched through asm operands with
PHYSICAL register constraints, and some of its writes are hidden from hipcc (EXEC-masked v_accvgpr_write under uniform branches,
declared afterwards by an instruction-free statement: csrc/rollout_one.hip).  That is only sound while hipcc itself
never touches those registers between the statements: no copy (v_accvgpr_mov / read / write of its own), no live-range split, no
AGPR used as a spill slot for a VGPR.  This script compiles rollout_one.hip exactly as csrc/build.py does and scans the ISA of
every rollout_one_kernel instantiation: ANY instruction outside an inline-asm statement (between ;;#ASMEND and the next
;;#ASMSTART) that names an AGPR, and any scratch spill inside the kernel, is reported.  A toolchain or flag change that makes the
compiler move a panel fails here (a CPU test runs it), not in a wrong trajectory.

    python tools/check_one_agpr.py [file.s]        exit code 0 = hipcc never touches an AGPR of the kernel itself
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "sampling_gpmpc_amd", "csrc")


def compile_to_isa():
    sys.path.insert(0, CSRC)
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpmpc_build", os.path.join(CSRC, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    out = os.path.join(tempfile.mkdtemp(prefix="gpmpc_one_"), "rollout_one.s")
    cmd = [b.HIPCC, "-x", "hip", "-S", "--cuda-device-only", os.path.join(CSRC, "rollout_one.hip"), "-o", out] + \
        [f for f in b.FLAGS if f != "-fPIC"] + b.EXTRA_FLAGS.get("rollout_one.hip", [])
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def check(path):
    kernel, in_asm, n_asm, n_out = None, False, 0, 0
    problems = []
    for ln, raw in enumerate(open(path), 1):
        t = raw.strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            kernel = m.group(1) if "rollout_one_kernel" in m.group(1) else None
            in_asm = False
            continue
        if kernel is None:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if t.startswith("s_endpgm"):
            kernel = None
            continue
        code = t.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        if in_asm:
            n_asm += 1
            continue
        n_out += 1
        if re.search(r"\ba\[\d+:\d+\]|\ba\d+\b", code) or code.startswith("v_accvgpr"):
            problems.append((kernel, ln, code, "an AGPR named by a compiler-generated instruction"))
        elif code.startswith("scratch_") or code.startswith("buffer_store_dword") and "offen" in code and "s[0:3]" in code:
            problems.append((kernel, ln, code, "scratch traffic (a spill) inside the kernel"))
    return n_asm, n_out, problems


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else compile_to_isa()
    n_asm, n_out, probs = check(path)
    print(f"{path}: {n_asm} instructions inside asm statements, {n_out} compiler-generated; {len(probs)} of those touch an AGPR or spill")
    for k, ln, code, why in probs[:20]:
        print(f"  line {ln}: `{code}`: {why}")
    sys.exit(1 if probs or n_asm == 0 else 0)
