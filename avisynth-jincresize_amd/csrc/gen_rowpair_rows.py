#!/usr/bin/env python3
"""Writes kernel_rowpair_rows.inc: for N = 6 .. 17 taps per kernel row, one kernel row of ewa_periodic_rowpair_kernel for a lane's four
chain pairs as two inline-assembly statements.

Why generated text and not templates: the chord of a (q, kernel row) -- TR taps left out on either side, a wave-uniform run-time
value -- is taken INSIDE the statements (scalar compares and branches to the first tap / past the last), so that the compiler
sees straight-line code around them.  As a C++ switch between unrolled bodies the accumulators went through phi nodes: 24
register moves per kernel row, 10 % of the kernel's VALU time (profiles/round5/pmc_rows/: 94.5 M of 550 M VALU instructions per
launch were not taps).  A statement's operand numbers are literal text, hence the generator.

Tap lx of chain k (the lane's period k = 0 .. 3) multiplies sample u = lx + k of the row segment -- half u % 2 of register pair
w[u / 2], broadcast to both halves of the product by op_sel -- by the coefficient pair c[lx] = (set(p = 0, q), set(p = 1, q))[ly][lx]
(an SGPR pair) and adds the product pair to the chain pair: v_pk_mul_f32 + v_pk_add_f32, un-fused, taps in lx order per chain
(ref /root/reference/src/JincResize.cpp:570-579).  The four multiplies of a tap are issued together, then the four adds.

Run: python3 gen_rowpair_rows.py > kernel_rowpair_rows.inc   (the Makefile does not run it; the .inc is committed)
"""
LO = "op_sel_hi:[0,1]"
HI = "op_sel:[1,0] op_sel_hi:[1,1]"


def tap(lx, pbase, cop):
    """Instructions of tap lx; operand numbers: %0-3 chains, %4-7 temporaries, %8+ pairs from pair index pbase, cop = the coefficient operand."""
    out = []
    for k in range(4):
        u = lx + k
        out.append(f"v_pk_mul_f32 %{4 + k}, %{8 + u // 2 - pbase}, %{cop} {HI if u % 2 else LO}")
    for k in range(4):
        out.append(f"v_pk_add_f32 %{k}, %{k}, %{4 + k}")
    return out


def statement(lines, pairs, coefs, n_pairs_name="w", first_pair=0, first_coef=0):
    ins = [f'"v"({n_pairs_name}[{first_pair + i}])' for i in range(pairs)] + [f'"s"(c[{first_coef + i}])' for i in range(coefs)] + ['"s"(tr)']
    body = "\n".join(f'        "{ln}\\n\\t"' for ln in lines)
    return (f"    asm volatile(\n{body}\n"
            f'        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)\n'
            f"        : {', '.join(ins)}\n"
            f'        : "scc");\n')


def max_trim(n):
    """Taps a kernel row may leave out per side: at most 5, and at least two taps stay."""
    return min(5, (n - 2) // 2)


def entry(tr_op, mt, tag):
    """Jumps to label a<k> for tr == k (k = 0 .. mt; tr >= mt lands on a<mt>)."""
    out = []
    for k in range(mt):
        out += [f"s_cmp_lt_u32 %{tr_op}, {k + 1}", f"s_cbranch_scc1 .Ljrp_{tag}{k}_%="]
    if mt:
        out.append(f"s_branch .Ljrp_{tag}{mt}_%=")
    return out


def row(n):
    mt = max_trim(n)
    nw = 2 * ((n + 3 + 3) // 4)
    head = (f"template <>\nstruct RowPairRow<{n}> {{\n"
            f"    static __device__ __forceinline__ void run(f32x2 (&a)[4], const f32x2 (&w)[{nw}], const f32x2 (&c)[{n}], uint32_t tr) {{\n"
            f"    f32x2 t0, t1, t2, t3;\n")
    if n <= 9:   # one statement: entry by tr, taps, exit by tr
        pairs = (n - 1 + 3) // 2 + 1
        tr_op = 8 + pairs + n
        lines = entry(tr_op, mt, "a")
        for lx in range(n):
            k = n - lx                  # tap lx is executed iff lx >= tr (entry) and tr < n - lx (exit)
            if k <= mt:
                lines += [f"s_cmp_lt_u32 %{tr_op}, {k}", f"s_cbranch_scc0 .Ljrp_end_%="]
            if lx <= mt:
                lines.append(f".Ljrp_a{lx}_%=:")
            lines += tap(lx, 0, 8 + pairs + lx)
        lines.append(".Ljrp_end_%=:")
        assert tr_op + 1 <= 30
        return head + statement(lines, pairs, n) + "    }\n};\n"
    split = min(8, n - mt)          # statement A: taps 0 .. split - 1 (the left flank 0 .. mt - 1 and what follows), B: the rest
    assert split > mt
    # ---- A: entry by tr ----
    a_pairs = (split - 1 + 3) // 2 + 1
    tr_op_a = 8 + a_pairs + split
    la = entry(tr_op_a, mt, "a")
    for lx in range(split):
        if lx <= mt:
            la.append(f".Ljrp_a{lx}_%=:")
        la += tap(lx, 0, 8 + a_pairs + lx)
    # ---- B: exit by tr ----
    pbase = split // 2
    b_last_pair = (n - 1 + 3) // 2
    b_pairs = b_last_pair - pbase + 1
    b_coefs = n - split
    tr_op_b = 8 + b_pairs + b_coefs
    lb = []
    for lx in range(split, n):
        k = n - lx                 # tap lx is executed iff tr < n - lx
        if k <= mt:
            lb += [f"s_cmp_lt_u32 %{tr_op_b}, {k}", f"s_cbranch_scc0 .Ljrp_end_%="]
        lb += tap(lx, pbase, 8 + b_pairs + (lx - split))
    lb.append(".Ljrp_end_%=:")
    assert tr_op_a + 1 <= 30 and tr_op_b + 1 <= 30
    return head + statement(la, a_pairs, split) + statement(lb, b_pairs, b_coefs, first_pair=pbase, first_coef=split) + "    }\n};\n"


print("// kernel_rowpair_rows.inc -- GENERATED by gen_rowpair_rows.py (see there); do not edit.\n"
      "// RowPairRow<N>::run(a, w, c, tr): taps lx = tr .. N - 1 - tr of one kernel row onto the lane's four chain pairs.\n"
      "template <int N>\nstruct RowPairRow;\n")
for n in range(6, 18):
    print(row(n))
