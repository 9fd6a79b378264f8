#!/usr/bin/env python3
"""Generates sampling_gpmpc_amd/csrc/rollout_one_gen.inc: the register-pinned leaf operations of rollout_one.hip.

rollout_one_kernel keeps a chain's whole factor in AGPRs as A operands of v_mfma_f64_4x4x4_4b_f64: a PANEL is one FP64
register (an AGPR pair) holding a 16 x 4 block of L - rows 16 R .. 16 R + 15 (the four MFMA blocks = four tile rows of
SUPER ROW R), columns 4 p .. 4 p + 3 - in the A-operand lane map (lane 16 k + 4 b + m holds L[16 R + 4 b + m][4 p + k]).
Rows are appended three per step, i.e. a few LANES of a panel change per step: a VALU instruction cannot address half of a
64-bit inline-asm operand, and a panel that hipcc is free to move gets copied around (measured: v_accvgpr_mov / read pairs
on every use).  So every panel is a C++ double that is ONLY ever touched through asm operands with a PHYSICAL register
constraint "{a[2n:2n+1]}": the register allocator then has one choice, the asm text names the halves (a<2n>, a<2n+1>), and
a masked lane update is s_mov exec + two v_accvgpr_write_b32.

Panel order (index n, registers a[2n:2n+1]) for super rows R = 0 .. NRES-1:
    PR[R][kt], kt < NKT   real-data block (whitened: the grid root), NKT = ceil(N_r / 4) column tiles
    PH[R][p],  p < 4 R    appended rows against earlier super rows
    PC[R][q],  q < 3      inside the diagonal super block: column tile q, blocks b > q (zero elsewhere)
    GD[R]                 the four diagonal tiles, inverted: block b holds (L_bb^-1)^T in the natural map
MFMA hazards as in tools/gen_mfma_chains.py: every statement opens with s_nop 1 and closes with s_nop 5; accumulating
MFMAs alternate between two accumulators.  The subtracting form uses the FP64 MFMA's neg modifier (neg:[1,0,0]: -A B + C).
"""
import os
import sys

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sampling_gpmpc_amd", "csrc", "rollout_one_gen.inc")
NRES = 5          # super rows resident in AGPRs (the sixth lives in LDS: three steps of a 30-step horizon touch it)
MF = "v_mfma_f64_4x4x4_4b_f64"


class Map:
    def __init__(self, nkt):
        self.nkt = nkt
        self.idx = {}
        n = 0
        for R in range(NRES):
            for kt in range(nkt):
                self.idx[("pr", R, kt)] = n; n += 1
            for p in range(4 * R):
                self.idx[("ph", R, p)] = n; n += 1
            for q in range(3):
                self.idx[("pc", R, q)] = n; n += 1
            self.idx[("gd", R, 0)] = n; n += 1
        self.count = n
        assert 2 * n <= 256, "panels exceed the AGPR file"

    def reg(self, kind, R, i):
        n = self.idx[(kind, R, i)]
        return 2 * n

    def member(self, kind, R, i):
        if kind == "ph":
            return f"P.ph[{2 * R * (R - 1) + i}]"
        if kind == "gd":
            return f"P.gd[{R}]"
        return f"P.{kind}[{R}][{i}]"


def chain_stmt(items, neg):
    """items: list of (areg or None, a_expr, b_expr): one asm statement accumulating alternately into c0 / c1.
    areg None: the A operand is an ordinary VGPR value."""
    lines = ['"s_nop 1\\n\\t"']
    ops_in = []
    k = len(items)
    for i, (areg, aexpr, bexpr) in enumerate(items):
        c = i & 1
        a_txt = f"a[{areg}:{areg + 1}]" if areg is not None else f"%{2 + 2 * i}"
        b_txt = f"%{2 + 2 * i + 1}"
        lines.append(f'"{MF} %{c}, {a_txt}, {b_txt}, %{c}{" neg:[1,0,0]" if neg else ""}\\n\\t"')
        if i + 1 < k:
            lines.append('"s_nop 1\\n\\t"')
        ops_in.append(f'"{{a[{areg}:{areg + 1}]}}"({aexpr})' if areg is not None else f'"v"({aexpr})')
        ops_in.append(f'"v"({bexpr})')
    lines.append('"s_nop 5"')
    body = "\n        ".join(lines)
    return f"    asm volatile({body}\n        : \"+v\"(c0), \"+v\"(c1)\n        : {', '.join(ops_in)});\n"


def set_stmt(items):
    """items: list of (areg, member_expr, value_expr): masked lane update of up to 6 panels in one statement."""
    k = len(items)
    # operands: 0 = saved exec, 1.. = k tied panels, then mask, then lo / hi pairs
    lines = ['"s_mov_b64 %0, exec\\n\\t"', f'"s_mov_b64 exec, %{1 + k}\\n\\t"']
    outs = ['"=&s"(sv_)']
    ins = ['"s"(mask)']
    for i, (areg, member, val) in enumerate(items):
        outs.append(f'"+{{a[{areg}:{areg + 1}]}}"({member})')
        lo = 2 + k + 2 * i
        lines.append(f'"v_accvgpr_write_b32 a{areg}, %{lo}\\n\\t"')
        lines.append(f'"v_accvgpr_write_b32 a{areg + 1}, %{lo + 1}\\n\\t"')
        ins.append(f'"v"(__double2loint({val}))')
        ins.append(f'"v"(__double2hiint({val}))')
    lines.append('"s_mov_b64 exec, %0"')
    body = "\n        ".join(lines)
    return f"    asm volatile({body}\n        : {', '.join(outs)}\n        : {', '.join(ins)});\n"


def emit(nkt, f):
    m = Map(nkt)
    sfx = f"k{nkt}"
    f.write(f"// ---- NKT = {nkt}: {m.count} resident panels in a[0:{2 * m.count - 1}] -------------------------------------------------------\n")
    f.write(f"struct OnePanels_{sfx} {{\n    double pr[{NRES}][{nkt}];\n    double ph[{2 * NRES * (NRES - 1)}];\n    double pc[{NRES}][3];\n    double gd[{NRES}];\n}};\n")
    P = f"OnePanels_{sfx}"
    # off-diagonal part of super row R:  c -= PR[R] VrRep, c -= PH[R] Vrep
    for R in range(NRES):
        items = [(m.reg("pr", R, kt), m.member("pr", R, kt), f"VrRep[{kt}]") for kt in range(nkt)]
        items += [(m.reg("ph", R, p), m.member("ph", R, p), f"Vrep[{p}]") for p in range(4 * R)]
        f.write(f"__device__ __forceinline__ void one_off_{sfx}_{R}({P}& P, double& c0, double& c1, const double* VrRep, const double* Vrep) {{\n")
        for s in range(0, len(items), 12):
            f.write(chain_stmt(items[s:s + 12], True))
        f.write("}\n")
        # W = GD[R] acc
        g = m.reg("gd", R, 0)
        f.write(f"__device__ __forceinline__ double one_gdm_{sfx}_{R}({P}& P, double acc) {{\n    double d;\n"
                f"    asm volatile(\"s_nop 1\\n\\t{MF} %0, a[{g}:{g + 1}], %2, 0\\n\\ts_nop 5\" : \"=&v\"(d) : \"{{a[{g}:{g + 1}]}}\"({m.member('gd', R, 0)}), \"v\"(acc));\n"
                f"    return d;\n}}\n")
        for q in range(3):
            r = m.reg("pc", R, q)
            f.write(f"__device__ __forceinline__ void one_pcm_{sfx}_{R}_{q}({P}& P, double& acc, double w) {{\n"
                    f"    asm volatile(\"s_nop 1\\n\\t{MF} %0, a[{r}:{r + 1}], %2, %0 neg:[1,0,0]\\n\\ts_nop 5\" : \"+v\"(acc) : \"{{a[{r}:{r + 1}]}}\"({m.member('pc', R, q)}), \"v\"(w));\n}}\n")
        # masked lane updates
        f.write(f"__device__ __forceinline__ void one_set_pr_{sfx}_{R}({P}& P, unsigned long long mask, const double* V) {{\n    unsigned long long sv_;\n")
        items = [(m.reg("pr", R, kt), m.member("pr", R, kt), f"V[{kt}]") for kt in range(nkt)]
        for s in range(0, len(items), 6):
            f.write(set_stmt(items[s:s + 6]))
        f.write("}\n")
        f.write(f"__device__ __forceinline__ void one_set_ph_{sfx}_{R}({P}& P, unsigned long long mask, const double* V) {{\n    unsigned long long sv_;\n    (void)sv_; (void)mask; (void)V; (void)P;\n")
        items = [(m.reg("ph", R, p), m.member("ph", R, p), f"V[{p}]") for p in range(4 * R)]
        for s in range(0, len(items), 6):
            f.write(set_stmt(items[s:s + 6]))
        f.write("}\n")
        for q in range(3):
            f.write(f"__device__ __forceinline__ void one_set_pc_{sfx}_{R}_{q}({P}& P, unsigned long long mask, double v) {{\n    unsigned long long sv_;\n")
            f.write(set_stmt([(m.reg("pc", R, q), m.member("pc", R, q), "v")]))
            f.write("}\n")
        f.write(f"__device__ __forceinline__ void one_set_gd_{sfx}_{R}({P}& P, unsigned long long mask, double v) {{\n    unsigned long long sv_;\n")
        f.write(set_stmt([(m.reg("gd", R, 0), m.member("gd", R, 0), "v")]))
        f.write("}\n")
    # compile-time dispatchers
    def disp(name, ret, params, args, per_q=False):
        f.write(f"template <int R{', int Q' if per_q else ''}>\n__device__ __forceinline__ {ret} {name}_{sfx}({params}) {{\n")
        first = True
        for R in range(NRES):
            if per_q:
                for q in range(3):
                    f.write(f"    {'if' if first else 'else if'} constexpr (R == {R} && Q == {q}) {'return ' if ret != 'void' else ''}{name}_{sfx}_{R}_{q}({args});\n")
                    first = False
            else:
                f.write(f"    {'if' if first else 'else if'} constexpr (R == {R}) {'return ' if ret != 'void' else ''}{name}_{sfx}_{R}({args});\n")
                first = False
        f.write("}\n")
    disp("one_off", "void", f"{P}& P, double& c0, double& c1, const double* VrRep, const double* Vrep", "P, c0, c1, VrRep, Vrep")
    disp("one_gdm", "double", f"{P}& P, double acc", "P, acc")
    disp("one_pcm", "void", f"{P}& P, double& acc, double w", "P, acc, w", per_q=True)
    disp("one_set_pr", "void", f"{P}& P, unsigned long long mask, const double* V", "P, mask, V")
    disp("one_set_ph", "void", f"{P}& P, unsigned long long mask, const double* V", "P, mask, V")
    disp("one_set_pc", "void", f"{P}& P, unsigned long long mask, double v", "P, mask, v", per_q=True)
    disp("one_set_gd", "void", f"{P}& P, unsigned long long mask, double v", "P, mask, v")


def generic(f):
    """chains with every operand in ordinary registers (the LDS-resident super row, the Gram products)"""
    for neg in (False, True):
        nm = "one_nchain" if neg else "one_pchain"
        for k in range(1, 13):
            args = ", ".join([f"double a{i}" for i in range(k)] + [f"double b{i}" for i in range(k)])
            items = [(None, f"a{i}", f"b{i}") for i in range(k)]
            f.write(f"__device__ __forceinline__ void {nm}{k}(double& c0, double& c1, {args}) {{\n")
            f.write(chain_stmt(items, neg))
            f.write("}\n")
        f.write(f"template <int K>\n__device__ __forceinline__ void {nm}(double& c0, double& c1, const double* A, const double* B) {{\n")
        for k in range(1, 13):
            call = ", ".join([f"A[{i}]" for i in range(k)] + [f"B[{i}]" for i in range(k)])
            f.write(f"    {'if' if k == 1 else 'else if'} constexpr (K == {k}) {nm}{k}(c0, c1, {call});\n")
        f.write("}\n")


if __name__ == "__main__":
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_rollout_one.py - do not edit.  Register-pinned panel operations of rollout_one.hip.\n")
        generic(f)
        for nkt in (9,):
            emit(nkt, f)
    print(OUT)
