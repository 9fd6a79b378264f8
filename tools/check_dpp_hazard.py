#!/usr/bin/env python3
"""Scan the gfx950 ISA of the kernels that issue v_fmac_f64_dpp / v_mfma_f64 from inline asm for the hazards hipcc does
not pad there (it neither sees into an asm statement nor pads its boundary beyond one wait state).

Rules (measured on MI355X: tools/ubench/mfma64_hazard2.hip, mfma64_chain.hip, mfma64_war.hip; an `s_nop N` is N + 1 wait
states, every other instruction one - EXCEPT an independent FP64 MFMA standing between an MFMA and the reader of its result:
tools/ubench/mfma64_fill.hip (round 4) shows that ONE such MFMA leaves nothing to pad for a VALU, a DPP or an MFMA SrcA/B reader
(the reader then waits 16+ cycles behind the producer); IN rollout_one.hip ONLY it counts as 6 wait states for the rules M2-M4
(the next tile row's independent MFMAs stand where the s_nop of the previous row would be; `gpmpc_selftest` and the parity tests
run that kernel against the oracle on the part at hand) - everywhere else, and for D1 / M1, an MFMA is ONE wait state like any
instruction:

  D1  VALU write of a VGPR -> the same VGPR read through DPP (v_fmac_f64_dpp / v_mov_b32_dpp source)   2 wait states
  M1  VALU write of a VGPR -> MFMA reading it as SrcA / SrcB / SrcC                           2
  M2  MFMA D -> MFMA reading it as SrcC                                                        4
  M3  MFMA D -> MFMA reading it as SrcA / SrcB                                                 6
  M4  MFMA D -> ANY other instruction reading it (VALU, v_accvgpr_write, a store's data)       6

rollout_indep.hip keeps its tables in registers that the step loop never writes, rollout_tiles.hip's broadcast sources
are written a phase earlier: D1 can only arise if the register allocator puts a copy or a reload right in front of an
asm statement - which is exactly what this script looks for in the ISA hipcc produced.  M1-M4 guard the MFMA chains of
rollout_tiles.hip (statements open with s_nop 1 and close with s_nop 5; a regression in the generator shows up here).

    python tools/check_dpp_hazard.py [file.s ...]          exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "sampling_gpmpc_amd", "csrc")
SOURCES = ["rollout_indep.hip", "rollout_tiles.hip", "rollout_one.hip", "joint_chol.hip"]
EXTRA = {"rollout_one.hip": ["-mllvm", "-disable-machine-licm"]}        # as csrc/build.py compiles it


def compile_to_isa(src):
    out = os.path.join(tempfile.mkdtemp(prefix="gpmpc_isa_"), src.replace(".hip", ".s"))
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-x", "hip", "-S", "--cuda-device-only",
           os.path.join(CSRC, src), "-o", out, "--offload-arch=gfx950", "-O3", "-std=c++17",
           "-fno-gpu-rdc", "-ffp-contract=on", "-I", os.path.join(REPO, "include"), "-I", CSRC] + EXTRA.get(src, [])
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def regs(tok):
    """register file and numbers of an operand token: ('v', {..}) / ('a', {..}) / (None, set())"""
    m = re.fullmatch(r"-?([va])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.fullmatch(r"-?([va])(\d+)", tok)
    return (m.group(1), {int(m.group(2))}) if m else (None, set())


TOK = r"-?[va]\[\d+:\d+\]|-?[va]\d+"
MFMA_WS_RELAXED = 6         # rollout_one.hip only (its solve statements are scheduled on the mfma64_fill.hip measurement)
RELAXED_FILES = ("rollout_one",)
# joint_chol.hip issues its MFMAs through the compiler builtin (hipcc's own hazard recogniser pads them) and its DPP broadcasts from
# inline asm: rule D1 only
DPP_ONLY_FILES = ("joint_chol", "jc")


def check(path, mfma_ws=None):
    if mfma_ws is None:     # STRICT by default: an MFMA is one wait state like any instruction; relaxed only for the named files
        mfma_ws = MFMA_WS_RELAXED if os.path.basename(path).startswith(RELAXED_FILES) else 1
    counts = {"dpp": 0, "mfma": 0}
    dpp_only = os.path.basename(path).startswith(DPP_ONLY_FILES)
    problems, kernel = [], None
    window = []          # preceding instructions, newest last: (wait states, wait states for M2-M4, kind, set of ('v'|'a', n) written)
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
        toks = re.findall(TOK, code)
        ops = [regs(x) for x in toks]
        tagged = [{(f, n) for n in ns} for f, ns in ops]

        def scan(read, rule, need, kinds):
            ws = 0
            for states, mstates, kind, written in reversed(window):
                if ws >= need:
                    return
                if kind in kinds and (written & read):
                    problems.append((kernel, ln, code, rule, sorted(written & read)))
                    return
                ws += mstates if rule in ("M2", "M3", "M4") else states

        is_mfma = op.startswith("v_mfma")
        if (op == "v_fmac_f64_dpp" and "row_newbcast" in code) or op == "v_mov_b32_dpp":
            counts["dpp"] += 1
            if len(tagged) > 1:
                scan(tagged[1], "D1", 2, ("valu",))
        if is_mfma and dpp_only:
            counts["mfma"] += 1
        elif is_mfma:
            counts["mfma"] += 1
            srcs = tagged[1:]
            for i, rd in enumerate(srcs):
                scan(rd, "M1", 2, ("valu",))
                scan(rd, "M2" if i == 2 else "M3", 4 if i == 2 else 6, ("mfma",))
        else:
            reads = set().union(*tagged[(1 if (op.startswith("v_") or op.startswith("ds_read") or op.startswith("buffer_load")
                                               or op.startswith("global_load") or op.startswith("scratch_load")) else 0):]) if tagged else set()
            if reads and not dpp_only:
                scan(reads, "M4", 6, ("mfma",))
        if op == "s_nop":
            m = re.search(r"s_nop\s+(\d+)", code)
            n = int(m.group(1)) + 1 if m else 1
            window.append((n, n, "nop", set()))
        elif is_mfma:
            window.append((1, mfma_ws, "mfma", tagged[0] if tagged else set()))
        elif op.startswith("v_") and tagged:
            window.append((1, 1, "valu", tagged[0]))      # VALU: the destination is the first operand
        else:
            window.append((1, 1, "other", set()))
        window = window[-12:]
    return counts, problems


if __name__ == "__main__":
    paths = sys.argv[1:] or [compile_to_isa(s) for s in SOURCES]
    bad = 0
    for src in paths:
        c, probs = check(src)
        print(f"{src}: {c['dpp']} DPP broadcast reads and {c['mfma']} MFMAs checked, {len(probs)} hazard(s)")
        for k, ln, code, rule, rg in probs[:20]:
            print(f"  [{rule}] {k} line {ln}: `{code}` <- {rg}")
        bad += len(probs) + (0 if (c["dpp"] or c["mfma"]) else 1)
    sys.exit(1 if bad else 0)
