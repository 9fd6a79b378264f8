#!/usr/bin/env python3
"""Generates sampling_gpmpc_amd/csrc/rollout_one_gen.inc: the register-pinned leaf operations of rollout_one.hip.

rollout_one_kernel keeps a chain's whole factor in AGPRs as A operands of v_mfma_f64_4x4x4_4b_f64: a PANEL (r, g) is one FP64
register (an AGPR pair) holding tile row r of the appended labels against the four column tiles of GROUP g (unified column
tiles 4 g .. 4 g + 3, the NKT real-data tiles first; one tile per MFMA block) in the A-operand lane map: lane (kq, bm, jq)
holds L[4 r + jq][4 (4 g + bm) + kq].  Rows are appended three per step, i.e. a few LANES of a panel change per step: a VALU
instruction cannot address half of a 64-bit inline-asm operand, and a panel that hipcc is free to move gets copied around
(measured: v_accvgpr_mov / read pairs on every use).  So every panel is a C++ double that is ONLY ever touched through asm
operands with a PHYSICAL register constraint "{a[2n:2n+1]}": the register allocator then has one choice, the asm text names the
halves (a<2n>, a<2n+1>), and a masked lane update is s_mov exec + two v_accvgpr_write_b32.

Panel order (class Map; index n, registers a[2n:2n+1]): for r = 0 .. NTR-1 the groups of row r (every group in front of the
row's own tile; the own group is absent when the row's tile is the group's first block), then GD[g], g = NKT >> 2 .. : the four
diagonal tiles of group g, inverted - block b holds (L_bb^-1)^T in the natural map.  122 panels for NKT = 9, NTR = 22.

Statements: the Gram / generic chains open with s_nop 1 and close with s_nop 5 (hipcc pads nothing around inline asm); the
forward substitution of a step is ONE statement per epoch (solve_stmt) whose wait states hold the next tile row's independent
MFMAs; tools/check_dpp_hazard.py checks the ISA of the result.  The subtracting form uses the FP64 MFMA's neg modifier
(neg:[1,0,0]: -A B + C).  --variant ... --out ...: schedule experiments of tools/ubench/one_solve_chain.hip.
"""
import os
import sys

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sampling_gpmpc_amd", "csrc", "rollout_one_gen.inc")
MF = "v_mfma_f64_4x4x4_4b_f64"


VARIANT = set()   # schedule experiments of tools/ubench/one_solve_chain.hip (--variant a,b,...); the shipped file has none
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
    head, tail, acc_first = True, "s_nop 5", 0
    lines = ['"s_nop 1\\n\\t"']
    ops_in = []
    k = len(items)
    assert not fresh or k >= 2
    for i, (areg, aexpr, bexpr) in enumerate(items):
        c = (i + acc_first) & 1
        a_txt = f"a[{areg}:{areg + 1}]" if areg is not None else f"%{2 + 2 * i}"
        b_txt = f"%{2 + 2 * i + 1}"
        src_c = "0" if (fresh and i < 2) else f"%{c}"
        lines.append(f'"{MF} %{c}, {a_txt}, {b_txt}, {src_c}{" neg:[1,0,0]" if neg else ""}\\n\\t"')
        if i + 1 < k:
            lines.append('"s_nop 1\\n\\t"')
        ops_in.append(f'"{{a[{areg}:{areg + 1}]}}"({aexpr})' if areg is not None else f'"v"({aexpr})')
        ops_in.append(f'"v"({bexpr})')
    if tail:
        lines.append(f'"{tail}"')
    else:
        lines[-1] = lines[-1][:-5] + '"'          # drop the trailing \n\t of the last instruction
    body = "\n        ".join(lines)
    cons = '"=&v"(c0), "=&v"(c1)' if fresh else '"+v"(c0), "+v"(c1)'
    return f"    asm volatile({body}\n        : {cons}\n        : {', '.join(ops_in)});\n"



# ---- the forward substitution of one step as ONE statement ------------------------------------------------------------------
# Registers of the statement (the kernel itself stays below v214; build.py / the ISA check assert it):
#   v[216 + 2 g : 217 + 2 g]   Vu[g], the solution of column group g (tied operands with a physical constraint: the DPP merge
#                              writes HALVES of it under a bank mask, and inline asm cannot name half of a %operand)
#   v[232:233] T   v[234:235] T2 (the rotated copy)   v[236:237] W   v[238:245] X0 .. X3: the accumulators of two tile rows
#   v[246:247] RNm: the row's right-hand side masked to the row's own block (the start value of its second accumulator)
#   v[248:249], v[250:251] G0 / G1: the accumulators of the Gram product S' = sum_g Vu[g]^T Vu[g] (outputs of the statement)
SOLVE_VU, SOLVE_T, SOLVE_T2, SOLVE_W, SOLVE_X, SOLVE_RNM, SOLVE_G = 216, 232, 234, 236, 238, 246, 248
SOLVE_CLOBBER = list(range(232, 248))


def vp(n):
    return f"v[{n}:{n + 1}]"


def solve_stmt(f, m, K, member):
    """Row r: acc = rhs_r - sum_g panel(r, g) Vu[g]; W = G_r (block sum of acc); block b of Vu[g_r] := W.  Left to the compiler
    (one statement per chain, per MFMA, per DPP pair) every MFMA result is fenced by s_nop 5 (the compiler neither sees the
    MFMA nor may it schedule into the statement), twice per row: 48 of a row's ~190 cycles.  Here the row is software
    pipelined by hand: only the row's LAST panel (the group that holds tile r - 1) depends on the previous row, the others are
    issued INSIDE the previous row - behind its dependent MFMA, behind its diagonal MFMA and in the wait states in front of
    the three DPP reads (tools/ubench/mfma64_fill.hip: ONE independent FP64 MFMA between an MFMA and the reader of its result
    leaves nothing to pad).  The rows that may be absent this step (4 r >= n_h) are left by scalar branches inside the
    statement; the last row of a step has nothing to hide behind and takes the padded form."""
    R0 = 4 * K - NKT
    rows = list(range(0, min(R0 + 4, NTR)))
    L = []
    used = []

    def areg(kind, x, y):
        n = m.reg(kind, x, y)
        used.append((n, member(x, y) if kind == "p" else f"P.gd[{x}]"))
        return f"a[{n}:{n + 1}]"
    VU = lambda g: SOLVE_VU + 2 * g
    T, T2, W = SOLVE_T, SOLVE_T2, SOLVE_W
    accs = lambda r: (SOLVE_X + 4 * (r & 1), SOLVE_X + 4 * (r & 1) + 2)
    NOUT = K + 1 + 2                                    # Vu[0 .. K], S0, S1
    rn_op = lambda g: f"%{NOUT + (g - m.gd[0])}"
    nh_op = f"%{NOUT + (K - m.gd[0] + 1)}"
    mk_op = lambda b: f"%{NOUT + (K - m.gd[0] + 1) + 1 + b}"      # 1.0 in the lanes of block b, 0.0 elsewhere
    GRAM = "nogram" not in VARIANT

    def gram_items(gs):
        """S' += Vu[g]^T Vu[g] (a natural register used as the A operand acts as its transpose): the step's Gram product rides in
        the wait states of the LAST tile row - which has no next row to hide behind - and behind its merge"""
        return [("gram", f"{MF} {vp(SOLVE_G + 2 * (g & 1))}, {vp(VU(g))}, {vp(VU(g))}, {'0' if g < 2 else vp(SOLVE_G + 2 * (g & 1))}")
                for g in gs]
    RNM = SOLVE_RNM
    # --variant rnfold: the right-hand side starts the row's second accumulator (masked to the row's block) instead of being
    # added on the chain.  Measured (tools/ubench/one_solve_chain.hip, 22 rows): -3 cycles per row alone, +6 together with the
    # post-merge MFMA below - not shipped.
    FOLD = "rnfold" in VARIANT

    def ra_list(r):
        # (accumulator, A panel, B register, start value: None = accumulate)
        return [(accs(r)[i & 1], areg("p", r, g), VU(g), (vp(RNM) if (i == 1 and FOLD) else "0") if i < 2 else None)
                for i, g in enumerate(m.groups[r][:-1])]

    def emit_ra(item):
        if item[0] == "gram":
            L.append(item[1])
            return
        acc, a, b, start = item
        L.append(f"{MF} {vp(acc)}, {a}, {vp(b)}, {start if start is not None else vp(acc)} neg:[1,0,0]")

    def emit_rnm(r):
        """RNm of row r: rhs_r in the lanes of the row's own block only - the cross-block sum then adds it exactly once, and the
        `+ rhs` add leaves the row's dependent chain (one FP64 multiply by a 0 / 1 mask, off the chain)"""
        u = u_of(r)
        L.append(f"v_mul_f64 {vp(RNM)}, {rn_op(u >> 2)}, {mk_op(u & 3)}")

    def rot(ctrl):
        L.append(f"v_mov_b32_dpp v{T2}, v{T} {ctrl} row_mask:0xf bank_mask:0xf bound_ctrl:1")
        L.append(f"v_mov_b32_dpp v{T2 + 1}, v{T + 1} {ctrl} row_mask:0xf bank_mask:0xf bound_ctrl:1")
        L.append(f"v_add_f64 {vp(T)}, {vp(T)}, {vp(T2)}")

    def row(r, nxt, head=True):
        """nxt: the independent MFMAs of row r + 1 (None: the step's last row - the Gram products of the groups in front of
        the row's own stand in their place, the own group's follows the merge); head: the statement in front ended on the merge
        (VALU write of Vu[dep] -> MFMA read: two wait states).  Returns whether the NEXT row needs that head."""
        u = u_of(r)
        g, b = u >> 2, u & 3
        last = nxt is None
        if last and GRAM:
            nxt = gram_items(range(0, g))
        c0, c1 = accs(r)
        dep = m.groups[r][-1]
        n_ind = len(m.groups[r]) - 1
        dacc = accs(r)[n_ind & 1]                       # the accumulator the independent MFMAs did not write last
        q = list(nxt) if nxt is not None else []
        # one MFMA of the next row is kept for BEHIND the merge, where the next row's dependent MFMA would wait for Vu[dep]
        post = q.pop() if (len(q) >= 3 and "nopost" not in VARIANT) else None
        keep = 1                                        # (one stays for behind the diagonal MFMA)

        def slot(pad):
            """an MFMA of the next row where a wait would stand (it counts as one state for a VALU -> DPP / MFMA read)"""
            if len(q) > keep and "noslots" not in VARIANT:
                emit_ra(q.pop(0))
                L.append("s_nop 0")
            else:
                L.append(f"s_nop {pad}")
        if not last and FOLD:
            emit_rnm(r + 1)                             # (read by the next row's second MFMA, in a slot below)
            if head:
                L.append("s_nop 0")
        elif head:
            L.append("s_nop 1")                         # the merge in front wrote Vu[dep]: VALU write -> MFMA read
        L.append(f"{MF} {vp(dacc)}, {areg('p', r, dep)}, {vp(VU(dep))}, {vp(dacc)} neg:[1,0,0]")
        if q:
            emit_ra(q.pop(0))
        else:
            L.append("s_nop 5")
        L.append(f"v_add_f64 {vp(T)}, {vp(c0)}, {vp(c1)}")
        if "nosum" in VARIANT:                          # (timing floor only: wrong results)
            L.append(f"v_add_f64 {vp(T)}, {vp(T)}, {rn_op(g)}")
            slot(1)
        else:
            slot(1)
            rot("row_ror:8")
            slot(1)
            rot("row_ror:4")
            if not FOLD:
                L.append(f"v_add_f64 {vp(T)}, {vp(T)}, {rn_op(g)}")
            slot(1)
        L.append(f"{MF} {vp(W)}, {areg('gd', g, 0)}, {vp(T)}, 0")
        if q:
            while q:
                emit_ra(q.pop(0))
                if q and "nomfmanop" not in VARIANT:
                    L.append("s_nop 1")
        else:
            L.append("s_nop 5")
        L.append(f"v_mov_b32_dpp v{VU(g)}, v{W} quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x{1 << b:x}")
        L.append(f"v_mov_b32_dpp v{VU(g) + 1}, v{W + 1} quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x{1 << b:x}")
        if post is not None:
            emit_ra(post)
            L.append("s_nop 0")
            if not last:
                return False
        if last and GRAM:                               # the own group's Gram product: Vu[g] was merged two instructions ago
            if post is None:
                L.append("s_nop 1")
            emit_ra(gram_items([g])[0])
        return True

    if R0 <= 0:
        L.append(f"s_cmp_eq_u32 {nh_op}, 0")
        L.append("s_cbranch_scc1 .Lone_none_%=")
    if FOLD:
        emit_rnm(0)
    L.append("s_nop 1")
    first = ra_list(0)
    for i, it in enumerate(first):
        emit_ra(it)
        if i + 1 < len(first):
            L.append("s_nop 1")
    lasts = []
    head = False                                        # (the prologue ends on MFMAs)
    for r in rows:
        if r + 1 in rows:
            if r + 1 >= R0:                             # row r + 1 may be absent this step
                L.append(f"s_cmp_gt_u32 {nh_op}, {4 * (r + 1)}")
                L.append(f"s_cbranch_scc0 .Lone_last{r}_%=")
                lasts.append((r, head))
            head = row(r, ra_list(r + 1), head)
        else:
            row(r, None, head)
    L.append("s_branch .Lone_end_%=")
    for r, hd in lasts:
        L.append(f".Lone_last{r}_%=:")
        row(r, None, True)                              # (reached by a branch: padded whatever stood in front - 8 cycles per step)
        L.append("s_branch .Lone_end_%=")
    if R0 <= 0:                                         # no appended row yet: the Gram product of the real-data tiles alone
        L.append(".Lone_none_%=:")
        if GRAM:
            L.append("s_nop 1")
            for i, it in enumerate(gram_items(range(0, K + 1))):
                if i:
                    L.append("s_nop 1")
                emit_ra(it)
    else:
        L.pop()                                         # the last of them falls through
    L.append(".Lone_end_%=:")
    if GRAM:
        L.append("s_nop 5")                             # S0 / S1 are read by the compiler's code next
    seen, ins = set(), []
    for n, mem in used:
        if n not in seen:
            seen.add(n)
            ins.append(f'"{{a[{n}:{n + 1}]}}"({mem})')
    outs = ", ".join(f'"+{{v[{VU(g)}:{VU(g) + 1}]}}"(Vu[{g}])' for g in range(K + 1))
    outs += f', "=&{{v[{SOLVE_G}:{SOLVE_G + 1}]}}"(S0), "=&{{v[{SOLVE_G + 2}:{SOLVE_G + 3}]}}"(S1)'
    rn = ", ".join(f'"v"(RN[{g}])' for g in range(m.gd[0], K + 1))
    mk = ", ".join(f'"v"(MK[{b}])' for b in range(4))
    clob = ", ".join(f'"v{n}"' for n in SOLVE_CLOBBER) + ', "scc"'
    body = "\n        ".join(f'"{x}\\n\\t"' for x in L[:-1]) + f'\n        "{L[-1]}"'
    f.write(f"__device__ __forceinline__ void one_solve_{K}(OnePanels& P, double* Vu, const double* RN, const double* MK, int nh, double& S0, double& S1) {{\n")
    f.write(f"    asm volatile({body}\n        : {outs}\n        : {rn}, \"s\"(nh), {mk},\n          {', '.join(ins)}\n        : {clob});\n}}\n")


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
    for K in m.gd:
        solve_stmt(f, m, K, member)
    disp("one_solve", "void", "OnePanels& P, double* Vu, const double* RN, const double* MK, int nh, double& S0, double& S1", "P, Vu, RN, MK, nh, S0, S1", m.gd)
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
    if "--variant" in sys.argv:
        VARIANT = set(sys.argv[sys.argv.index("--variant") + 1].split(","))
    if "--out" in sys.argv:
        OUT = sys.argv[sys.argv.index("--out") + 1]
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_rollout_one.py - do not edit.  Register-pinned panel operations of rollout_one.hip.\n")
        generic(f)
        emit(f)
    print(OUT)
