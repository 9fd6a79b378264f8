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
MF = "v_mfma_f64_4x4x4_4b_f64"


NKT = 9           # real-data column tiles (N_r = 36)
NTR = 22          # tile rows of appended labels (3 (H - 1) <= 88)


def u_of(r):
    return NKT + r


class Map:
    """panel (r, g): tile row r of the appended labels against column GROUP g (unified column tiles 4 g .. 4 g + 3, the
    real-data tiles first); the group that holds the row's own diagonal tile stops in front of it and is absent when the
    diagonal tile is the group's first block."""
    def __init__(self):
        self.idx = {}
        self.groups = {}
        n = 0
        for r in range(NTR):
            u = u_of(r)
            gs = list(range(u >> 2)) + ([u >> 2] if (u & 3) else [])
            self.groups[r] = gs
            for g in gs:
                self.idx[("p", r, g)] = n; n += 1
        self.gd = list(range(NKT >> 2, (u_of(NTR - 1) >> 2) + 1))
        for g in self.gd:
            self.idx[("gd", g, 0)] = n; n += 1
        self.count = n
        assert 2 * n <= 256, "panels exceed the AGPR file"

    def reg(self, kind, a, b):
        return 2 * self.idx[(kind, a, b)]


def chain_stmt(items, neg, fresh=False):
    """items: list of (areg or None, a_expr, b_expr): one asm statement accumulating alternately into c0 / c1.
    areg None: the A operand is an ordinary VGPR value.  fresh: the accumulators start at zero (SrcC = 0 in the first MFMA of
    each: no v_mov of zeros in front of every chain); needs at least two MFMAs."""
    lines = ['"s_nop 1\\n\\t"']
    ops_in = []
    k = len(items)
    assert not fresh or k >= 2
    for i, (areg, aexpr, bexpr) in enumerate(items):
        c = i & 1
        a_txt = f"a[{areg}:{areg + 1}]" if areg is not None else f"%{2 + 2 * i}"
        b_txt = f"%{2 + 2 * i + 1}"
        src_c = "0" if (fresh and i < 2) else f"%{c}"
        lines.append(f'"{MF} %{c}, {a_txt}, {b_txt}, {src_c}{" neg:[1,0,0]" if neg else ""}\\n\\t"')
        if i + 1 < k:
            lines.append('"s_nop 1\\n\\t"')
        ops_in.append(f'"{{a[{areg}:{areg + 1}]}}"({aexpr})' if areg is not None else f'"v"({aexpr})')
        ops_in.append(f'"v"({bexpr})')
    lines.append('"s_nop 5"')
    body = "\n        ".join(lines)
    cons = '"=&v"(c0), "=&v"(c1)' if fresh else '"+v"(c0), "+v"(c1)'
    return f"    asm volatile({body}\n        : {cons}\n        : {', '.join(ops_in)});\n"


def set_stmt(items, mask="mask"):
    """items: list of (areg, member_expr, value_expr): masked lane update of up to 6 panels in one statement."""
    k = len(items)
    # operands: 0 = saved exec, 1.. = k tied panels, then mask, then lo / hi pairs
    # an empty mask skips the writes INSIDE the statement: a C++ `if` around a tied physical-register operand makes hipcc keep
    # the panel in a virtual register across the branch (copies in and out of the pinned register, spills of the rest)
    lines = ['"s_mov_b64 %0, exec\\n\\t"', f'"s_mov_b64 exec, %{1 + k}\\n\\t"', '"s_cbranch_execz .Lone_skip_%=\\n\\t"']
    outs = ['"=&s"(sv_)']
    ins = [f'"s"({mask})']
    for i, (areg, member, val) in enumerate(items):
        outs.append(f'"+{{a[{areg}:{areg + 1}]}}"({member})')
        lo = 2 + k + 2 * i
        lines.append(f'"v_accvgpr_write_b32 a{areg}, %{lo}\\n\\t"')
        lines.append(f'"v_accvgpr_write_b32 a{areg + 1}, %{lo + 1}\\n\\t"')
        ins.append(f'"v"(__double2loint({val}))')
        ins.append(f'"v"(__double2hiint({val}))')
    lines.append('".Lone_skip_%=:\\n\\t"')
    lines.append('"s_mov_b64 exec, %0"')
    body = "\n        ".join(lines)
    return f"    asm volatile({body}\n        : {', '.join(outs)}\n        : {', '.join(ins)});\n"


def hidden_set_stmt(items, mask):
    """the same masked update WITHOUT naming the panels as operands: for use under a C++ branch (a tied physical-register
    operand defined inside a branch makes hipcc carry the panel in a virtual register across it); one_touch() afterwards tells
    the compiler that the panels may have changed."""
    k = len(items)
    lines = ['"s_mov_b64 %0, exec\\n\\t"', '"s_mov_b64 exec, %1\\n\\t"']
    ins = [f'"s"({mask})']
    for i, (areg, val) in enumerate(items):
        lo = 2 + 2 * i
        lines.append(f'"v_accvgpr_write_b32 a{areg}, %{lo}\\n\\t"')
        lines.append(f'"v_accvgpr_write_b32 a{areg + 1}, %{lo + 1}\\n\\t"')
        ins.append(f'"v"(__double2loint({val}))')
        ins.append(f'"v"(__double2hiint({val}))')
    lines.append('"s_mov_b64 exec, %0"')
    body = "\n        ".join(lines)
    return f"    asm volatile({body}\n        : \"=&s\"(sv_)\n        : {', '.join(ins)});\n"


def emit(f):
    m = Map()
    f.write(f"// ---- NKT = {NKT}, {NTR} tile rows: {m.count} panels in a[0:{2 * m.count - 1}] ----------------------------------------------------\n")
    f.write(f"struct OnePanels {{\n    double p[{m.count - len(m.gd)}];\n    double gd[{max(m.gd) + 1}];\n}};\n")
    f.write(f"constexpr int kOneNKT = {NKT}, kOneNTR = {NTR};\n")
    member = lambda r, g: f"P.p[{m.idx[('p', r, g)]}]"
    for r in range(NTR):
        gs = m.groups[r]
        # off-diagonal part of tile row r: c -= panel(r, g) Vu[g]  (block b of the product: column tile 4 g + b)
        f.write(f"__device__ __forceinline__ void one_row_{r}(OnePanels& P, double& c0, double& c1, const double* Vu) {{\n")
        f.write(chain_stmt([(m.reg("p", r, g), member(r, g), f"Vu[{g}]") for g in gs], True, fresh=True))
        f.write("}\n")
        # masked lane update: the full groups under mBase, the row's own (partial) group under mLast
        u = u_of(r)
        full = [g for g in gs if g < (u >> 2)]
        f.write(f"__device__ __forceinline__ void one_set_row_{r}(OnePanels& P, unsigned long long mBase, unsigned long long mLast, const double* V) {{\n    unsigned long long sv_;\n    (void)mLast;\n")
        for s in range(0, len(full), 6):
            f.write(set_stmt([(m.reg("p", r, g), member(r, g), f"V[{g}]") for g in full[s:s + 6]], "mBase"))
        if u & 3:
            g = u >> 2
            f.write(set_stmt([(m.reg("p", r, g), member(r, g), f"V[{g}]")], "mLast"))
        f.write("}\n")
    for r in range(NTR):
        u = u_of(r)
        full = [g for g in m.groups[r] if g < (u >> 2)]
        f.write(f"__device__ __forceinline__ void one_hset_row_{r}(unsigned long long mBase, unsigned long long mLast, const double* V) {{\n    unsigned long long sv_;\n    (void)mLast;\n")
        for s_ in range(0, len(full), 12):
            f.write(hidden_set_stmt([(m.reg("p", r, g), f"V[{g}]") for g in full[s_:s_ + 12]], "mBase"))
        if u & 3:
            g = u >> 2
            f.write(hidden_set_stmt([(m.reg("p", r, g), f"V[{g}]")], "mLast"))
        f.write("}\n")
    # "the panels of tile rows R .. R + N - 1 may have changed": empty statements with tied operands
    f.write("template <int R>\n__device__ __forceinline__ void one_touch_row(OnePanels& P) {\n")
    for r in range(NTR):
        ops = ", ".join([f'"+{{a[{m.reg("p", r, g)}:{m.reg("p", r, g) + 1}]}}"({member(r, g)})' for g in m.groups[r]])
        f.write(f"    {'if' if r == 0 else 'else if'} constexpr (R == {r}) asm volatile(\"\" : {ops});\n")
    f.write("}\n")
    for g in m.gd:
        a = m.reg("gd", g, 0)
        f.write(f"__device__ __forceinline__ double one_gdm_{g}(OnePanels& P, double acc) {{\n    double d;\n"
                f"    asm volatile(\"s_nop 1\\n\\t{MF} %0, a[{a}:{a + 1}], %2, 0\\n\\ts_nop 5\" : \"=&v\"(d) : \"{{a[{a}:{a + 1}]}}\"(P.gd[{g}]), \"v\"(acc));\n"
                f"    return d;\n}}\n")
        f.write(f"__device__ __forceinline__ void one_hset_gd_{g}(double v) {{\n"
                f"    asm volatile(\"v_accvgpr_write_b32 a{a}, %0\\n\\tv_accvgpr_write_b32 a{a + 1}, %1\" : : \"v\"(__double2loint(v)), \"v\"(__double2hiint(v)));\n}}\n")
        f.write(f"__device__ __forceinline__ void one_touch_gd_{g}(OnePanels& P) {{ asm volatile(\"\" : \"+{{a[{a}:{a + 1}]}}\"(P.gd[{g}])); }}\n")
        f.write(f"__device__ __forceinline__ void one_set_gd_{g}(OnePanels& P, unsigned long long mask, double v) {{\n    unsigned long long sv_;\n")
        f.write(set_stmt([(a, f"P.gd[{g}]", "v")]))
        f.write("}\n")
    # everything to zero, the diagonal tiles to the identity
    f.write("__device__ __forceinline__ void one_init(OnePanels& P, double inat) {\n    unsigned long long sv_;\n    unsigned long long mask = ~0ull;\n    double z = 0.0;\n")
    allp = [(m.reg("p", r, g), member(r, g), "z") for r in range(NTR) for g in m.groups[r]]
    for s in range(0, len(allp), 6):
        f.write(set_stmt(allp[s:s + 6]))
    for g in m.gd:
        f.write(set_stmt([(m.reg("gd", g, 0), f"P.gd[{g}]", "inat")]))
    f.write("}\n")

    def disp(name, ret, params, args, rng):
        f.write(f"template <int R>\n__device__ __forceinline__ {ret} {name}({params}) {{\n")
        for i, r in enumerate(rng):
            f.write(f"    {'if' if i == 0 else 'else if'} constexpr (R == {r}) {'return ' if ret != 'void' else ''}{name}_{r}({args});\n")
        f.write("}\n")
    disp("one_row", "void", "OnePanels& P, double& c0, double& c1, const double* Vu", "P, c0, c1, Vu", range(NTR))
    disp("one_set_row", "void", "OnePanels& P, unsigned long long mBase, unsigned long long mLast, const double* V", "P, mBase, mLast, V", range(NTR))
    disp("one_hset_row", "void", "unsigned long long mBase, unsigned long long mLast, const double* V", "mBase, mLast, V", range(NTR))
    disp("one_gdm", "double", "OnePanels& P, double acc", "P, acc", m.gd)
    disp("one_set_gd", "void", "OnePanels& P, unsigned long long mask, double v", "P, mask, v", m.gd)
    disp("one_hset_gd", "void", "double v", "v", m.gd)
    disp("one_touch_gd", "void", "OnePanels& P", "P", m.gd)


def generic(f):
    """chains with every operand in ordinary registers (the LDS-resident super row, the Gram products)"""
    for neg, fresh in ((False, False), (True, False), (False, True)):
        nm = "one_fchain" if fresh else ("one_nchain" if neg else "one_pchain")
        for k in range(2 if fresh else 1, 13):
            args = ", ".join([f"double a{i}" for i in range(k)] + [f"double b{i}" for i in range(k)])
            items = [(None, f"a{i}", f"b{i}") for i in range(k)]
            f.write(f"__device__ __forceinline__ void {nm}{k}(double& c0, double& c1, {args}) {{\n")
            f.write(chain_stmt(items, neg, fresh))
            f.write("}\n")
        f.write(f"template <int K>\n__device__ __forceinline__ void {nm}(double& c0, double& c1, const double* A, const double* B) {{\n")
        for k in range(2 if fresh else 1, 13):
            call = ", ".join([f"A[{i}]" for i in range(k)] + [f"B[{i}]" for i in range(k)])
            f.write(f"    {'if' if k == (2 if fresh else 1) else 'else if'} constexpr (K == {k}) {nm}{k}(c0, c1, {call});\n")
        f.write("}\n")


if __name__ == "__main__":
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_rollout_one.py - do not edit.  Register-pinned panel operations of rollout_one.hip.\n")
        generic(f)
        emit(f)
    print(OUT)
