#!/usr/bin/env python3
"""Scan the gfx950 ISA of rollout_indep.hip for the DPP read-after-VALU-write hazard of its inline-asm table FMAs.

rollout_indep_grid_kernel reads its register-resident tables as the DPP source of inline-asm `v_fmac_f64_dpp`.  A VGPR
written by a VALU instruction may be read through DPP only two wait states later; the compiler inserts the wait states
for DPP instructions it generates itself, but it cannot see into inline asm.  The table registers are never written in
the step loop, so the hazard can only arise if the register allocator inserts a copy of a table register right in
front of an asm statement.  This script compiles the file to ISA (or takes a .s file) and checks that none of the two
instructions in front of every asm `v_fmac_f64_dpp` writes its DPP source register pair (an `s_nop N` counts N + 1
wait states).  Exit code 0 = clean.

    python tools/check_dpp_hazard.py [file.s]
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


def vregs(tok):
    m = re.fullmatch(r"-?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"-?v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(path):
    n_dpp, problems, kernel = 0, [], None
    window = []                                   # (wait states it provides, VGPRs it writes) of the preceding instructions
    for ln, raw in enumerate(open(path), 1):
        t = raw.strip()
        if re.match(r"^_Z\w+:", t):
            kernel, window = t.split(":")[0], []
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        code = t.split(";")[0].strip()
        if not code:
            continue
        op = code.split()[0]
        toks = re.findall(r"-?v\[\d+:\d+\]|-?v\d+", code)
        if op == "v_fmac_f64_dpp" and "row_newbcast" in code:
            n_dpp += 1
            src = vregs(toks[1]) if len(toks) > 1 else set()
            ws = 0
            for states, written in reversed(window):
                if ws >= 2:
                    break
                if written & src:
                    problems.append((kernel, ln, code, sorted(written & src)))
                    break
                ws += states
        if op == "s_nop":
            m = re.search(r"s_nop\s+(\d+)", code)
            window.append((int(m.group(1)) + 1 if m else 1, set()))
        elif op.startswith("v_") and toks:
            window.append((1, vregs(toks[0])))        # VALU: destination is the first operand
        else:
            window.append((1, set()))
        window = window[-4:]
    return n_dpp, problems


if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 else compile_to_isa()
    n, probs = check(src)
    print(f"{src}: {n} DPP table reads checked, {len(probs)} hazard(s)")
    for k, ln, code, regs in probs[:20]:
        print(f"  {k} line {ln}: `{code}` reads v{regs} written less than two wait states earlier")
    sys.exit(1 if probs or n == 0 else 0)
