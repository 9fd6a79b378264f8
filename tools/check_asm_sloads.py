#!/usr/bin/env python3
"""Scan the gfx950 ISA of rollout_indep.hip for the one hazard of its hand-issued scalar loads.

rollout_indep_grid_kernel requests table units with inline-asm `s_load_dwordx16` and waits with an inline-asm
`s_waitcnt lgkmcnt(0)`.  The compiler does not know that the destination SGPRs are stale between the two; a spill
(`v_writelane`), copy or use it inserted there would read garbage.  This script compiles the file to ISA (or takes a
.s file) and checks, per kernel, that no instruction between an asm `s_load_dwordx16 s[a:b]` and the next asm
`s_waitcnt lgkmcnt(0)` reads or overwrites any of s[a..b].  Exit code 0 = clean.

    python tools/check_asm_sloads.py [file.s]
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "sampling_gpmpc_amd", "csrc")


def compile_to_isa():
    out = os.path.join(tempfile.mkdtemp(prefix="gpmpc_isa_"), "rollout_indep.s")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-x", "hip", "-S", "--cuda-device-only",
           os.path.join(CSRC, "rollout_indep.hip"), "-o", out, "--offload-arch=gfx950", "-O3", "-std=c++17",
           "-fno-gpu-rdc", "-ffp-contract=on", "-I", os.path.join(REPO, "include"), "-I", CSRC]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def sregs(tok):
    """SGPR numbers named by an operand token: s5, s[8:23]."""
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    problems, n_loads, kernel = [], 0, None
    pending = {}            # sgpr -> line number of the request
    in_asm = False
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";;#")[0] if ";;#" not in raw[:8] else raw
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if re.match(r"^_Z\w+:", t):
            kernel, pending = t[:-1], {}
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        code = t.split(";")[0].strip()
        toks = re.findall(r"s\[\d+:\d+\]|s\d+", code)
        if in_asm and code.startswith("s_load_dwordx16"):
            n_loads += 1
            dst = sregs(toks[0])
            for r in dst:
                pending[r] = ln
            used = set().union(*[sregs(x) for x in toks[1:]]) if len(toks) > 1 else set()
            bad = (used & set(pending)) - dst
            if bad:
                problems.append((kernel, ln, code, sorted(bad)))
            continue
        if in_asm and code.startswith("s_waitcnt"):
            pending = {}
            continue
        if code.startswith("s_endpgm"):
            pending = {}
            continue
        touched = set().union(*[sregs(x) for x in toks]) if toks else set()
        bad = touched & set(pending)
        if bad:
            problems.append((kernel, ln, code, sorted(bad)))
    return n_loads, problems


if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 else compile_to_isa()
    n, probs = check(src)
    print(f"{src}: {n} asm scalar-load requests checked, {len(probs)} hazard(s)")
    for k, ln, code, regs in probs[:20]:
        print(f"  {k} line {ln}: `{code}` touches in-flight SGPRs {regs}")
    sys.exit(1 if probs or n == 0 else 0)
