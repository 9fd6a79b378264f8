#!/usr/bin/env python3
"""The computed jumps of joint_mfma_gen.inc land on `table + (index << shift)`: every case of a table must fit its stride, or the
assembler's alignment pushes the following cases one stride down and the jumps land in the wrong accumulator tile - silently.
This compiles joint_mfma.hip for gfx950 (no GPU needed), reads the symbol table of the device object and checks

    jm_tend<shift>n<tiles>_N - jm_case0_N == tiles << shift       (jm_acc_diag / jm_acc_set* / jm_acc_get*)
    jm_tab1_N - jm_tab0_N == jm_tab2_N - jm_tab1_N == jm_exit_N - jm_tab2_N == NT << 7      (jm_acc_fma_run)

    python tools/check_joint_mfma_tables.py        # exit code 0 = every table has its stride
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "sampling_gpmpc_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
NT = 26


def symbols():
    with tempfile.TemporaryDirectory() as d:
        obj, dev = os.path.join(d, "jm.o"), os.path.join(d, "jm_gfx950.o")
        subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "-c", os.path.join(CSRC, "joint_mfma.hip"), "-o", obj, "--offload-arch=gfx950",
                        "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-ffp-contract=on", "-I", os.path.join(REPO, "include"), "-I", CSRC,
                        "--cuda-device-only", "-w"], check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + obj,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + dev], check=True)
        out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-t", dev], check=True, capture_output=True, text=True).stdout
    sym = collections.defaultdict(dict)
    for line in out.splitlines():
        m = re.match(r"^([0-9a-f]+) .* (jm_[a-z0-9]+)_(\d+)$", line.strip())
        if m:
            sym[int(m.group(3))][m.group(2)] = int(m.group(1), 16)
    return sym


def main():
    bad, n = [], 0
    for uid, s in sorted(symbols().items()):
        if "jm_tab0" in s:
            n += 1
            want = NT << 7
            got = [s["jm_tab1"] - s["jm_tab0"], s["jm_tab2"] - s["jm_tab1"], s["jm_exit"] - s["jm_tab2"]]
            if got != [want] * 3 or s["jm_tab0"] % 128:
                bad.append(f"statement {uid}: run tables {got}, expected {want} each (table at {s['jm_tab0']:#x})")
        for key in s:
            m = re.match(r"jm_tend(\d)n(\d+)$", key)
            if m:
                n += 1
                shift, ntile = int(m.group(1)), int(m.group(2))
                if s[key] - s["jm_case0"] != ntile << shift or s["jm_case0"] % (1 << shift):
                    bad.append(f"statement {uid}: {s[key] - s['jm_case0']} bytes of cases, expected {ntile << shift} (stride {1 << shift})")
    print(f"{n} jump tables checked, {len(bad)} with a wrong stride")
    for b in bad:
        print("  " + b)
    return 1 if bad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
