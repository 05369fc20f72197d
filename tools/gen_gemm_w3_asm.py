#!/usr/bin/env python3
"""Generates labrador_snark_amd/csrc/gemm_w3_asm.inc: the K loop of k_gemm_i8_w3 (mfma_kernels.hip) as gfx950
assembly, one inline-asm block per (column tiles NT, non-temporal A loads) variant.

Why assembly: the loop keeps three stages of A operands (registers) and B operands (LDS-DMA) in flight behind the stage
being multiplied.  hipcc's wait-count pass cannot follow loads that are consumed several loop trips after they were
issued — it drains the vector-memory queue (s_waitcnt vmcnt(0)) inside every stage, which serialises the HBM stream with
the matrix instructions (round 1's k_gemm_i8: load time + multiply time instead of their maximum) — and it assigns ONE
register class to the results of all matrix instructions of a kernel, while 288 accumulators need the 256 accumulation
registers plus 32 vector registers.  Here every wait is counted by hand.

Register plan of the block (everything else is the compiler's):
  v[120:215]  A ring: slot r (4), k-tile ks (2), plane i (3) -> 4 registers at 120 + ((r*2 + ks)*3 + i)*4
  v[216:227]  B fragments in flight from LDS (3 x 4)
  s[40:51]    loop state: s40 next stage to issue, s41 stage being multiplied, s[42:43] / s44 address increments,
              s[46:51] the A addresses of the three planes (advanced here)
Operands: %0 .. %(3*NT-1): accumulator tile (plane i, column tile n) = operand i*NT + n, the first 16 in accumulation
registers, the rest in vector registers; then
  BG          (+v, 64-bit)  address of this wave's first DMA piece of the stage to be issued next, + lane*16
  A0, A1, A2  (s, 64-bit)   address of this wave's A operands of plane i, stage 0
  VOFF        (v)           lane*16
  VLDS        (v)           LDS byte address of ring slot 0 + lane*16
  M0B         (s)           LDS byte address of ring slot 0 + this wave's first piece
  NST         (s)           number of stages (even, >= 2)

    python tools/gen_gemm_w3_asm.py > labrador_snark_amd/csrc/gemm_w3_asm.inc
"""
import sys

KS, RING, PF = 2, 4, 3
RING0, FR0 = 120, 216


def gen(NT, nt_loads, mode="full", persistent=False, merged=False):
    """mode: "full"; the diagnostic variants "noload" (no global loads / LDS-DMA: the multiply side alone) and "nomfma"
    (no matrix instructions or LDS reads: the load side alone) keep everything else — waits, barriers, address updates.
    persistent: the loop of a block that works off several jobs (k_gemm_i8_w3p) — the B stages 0..2 of the job were issued by
    gen_bissue()'s block BEFORE the previous job's results were recombined (LDS-DMA: no register is in flight while the
    compiler's code runs), so the prologue here issues the A operands only; the first k-tile's matrix instructions take 0 as
    their C operand (the accumulators are never zeroed).
    merged (persistent only): products of equal weight share an accumulator tile — (matrix plane i, vector plane i') adds into tile
    (i + i', half h), operand (i + i') * NT/3 + h: 5 * NT/3 tiles instead of 3 * NT.  For sums short enough that three tile entries
    add up in 32 bits (the caller checks); the recombination then has five terms instead of nine."""
    halves = NT // 3
    nacc = 5 * halves if merged else 3 * NT

    def acc_of(i, n):
        if not merged:
            return i * NT + n
        iv, h = divmod(n, halves)
        return (i + iv) * halves + h
    fresh = set()                  # merged + first: tiles that have received their first product
    pieces = KS * NT
    ppw = (pieces + 3) // 4
    slot_bytes = pieces * 1024
    ops = ppw + 3 * KS
    left = ops + 3 * KS            # may stay in flight at the top of a stage: stage st+2, and the A operands of stage st+1
    # operand numbers: outputs first (accumulators, then BG), then the inputs
    o_bg = nacc
    o_a = [nacc + 1 + i for i in range(3)]
    o_voff, o_vlds, o_m0b, o_nst = nacc + 4, nacc + 5, nacc + 6, nacc + 7
    sa = ["s[%d:%d]" % (46 + 2 * i, 47 + 2 * i) for i in range(3)]     # the A addresses, advanced here
    nt = " nt" if nt_loads else ""
    L = []
    emit = L.append

    def ring(r, ks, i):
        b = RING0 + ((r * KS + ks) * 3 + i) * 4
        return "v[%d:%d]" % (b, b + 3)

    def fr(k):
        b = FR0 + 4 * k
        return "v[%d:%d]" % (b, b + 3)

    # s40: index of the next stage to issue, s41: stage being multiplied, s[42:43]: 64-bit B increment, s44: A increment
    def vmem_ops(slot):
        """the instructions that issue one stage, in groups that are spread between the matrix instructions"""
        ld = mode != "noload"
        grp = [["s_add_u32 m0, %%%d, %d" % (o_m0b, slot * slot_bytes), "s_nop 0"] + (["global_load_lds_dwordx4 %%%d, off" % o_bg] if ld else [])]
        for q in range(1, ppw):
            grp.append(["global_load_lds_dwordx4 %%%d, off offset:%d" % (o_bg, q * 1024)] if ld else [])
        for ks in range(KS):
            for i in range(3):
                grp.append(["global_load_dwordx4 %s, %%%d, %s offset:%d%s" % (ring(slot, ks, i), o_voff, sa[i], ks * 1024, nt)] if ld else [])
        # advance to the next stage unless this was the last one (then the last stage is fetched again, into a slot nobody reads)
        adv = ["s_add_u32 s40, s40, 1", "s_cmp_lt_u32 s40, %%%d" % o_nst,
               "s_cselect_b32 s42, %d, 0" % (pieces * 1024), "s_cselect_b32 s44, %d, 0" % (KS * 1024),
               "v_lshl_add_u64 %%%d, %%%d, 0, s[42:43]" % (o_bg, o_bg)]
        for i in range(3):
            adv += ["s_add_u32 s%d, s%d, s44" % (46 + 2 * i, 46 + 2 * i), "s_addc_u32 s%d, s%d, 0" % (47 + 2 * i, 47 + 2 * i)]
        grp.append(adv)
        return grp

    def issue_now(slot):
        for grp in vmem_ops(slot):
            for ins in grp:
                emit(ins)

    def issue_a_only(slot):
        """persistent prologue: the A operands of one stage (its B pieces are already on their way)"""
        for ks in range(KS):
            for i in range(3):
                emit("global_load_dwordx4 %s, %%%d, %s offset:%d%s" % (ring(slot, ks, i), o_voff, sa[i], ks * 1024, nt))
        emit("s_add_u32 s40, s40, 1")
        emit("s_cmp_lt_u32 s40, %%%d" % o_nst)
        emit("s_cselect_b32 s44, %d, 0" % (KS * 1024))
        for i in range(3):
            emit("s_add_u32 s%d, s%d, s44" % (46 + 2 * i, 46 + 2 * i))
            emit("s_addc_u32 s%d, s%d, 0" % (47 + 2 * i, 47 + 2 * i))

    def stage(k, first=False):
        """multiply ring slot k; issue the loads of three stages ahead into slot (k+3)%4, spread between the matrix instructions"""
        emit("s_waitcnt vmcnt(%d)" % left)
        emit("s_barrier")
        pending = vmem_ops((k + 3) % RING)
        nm = KS * NT
        for f in range(nm):
            ks, n = divmod(f, NT)
            emit("s_waitcnt lgkmcnt(%d)" % (PF - 1))
            for i in range(3):
                if mode != "nomfma":
                    t_ = acc_of(i, n)
                    zero_c = first and ks == 0 and (not merged or t_ not in fresh)
                    if first and ks == 0:
                        fresh.add(t_)
                    c_op = "0" if zero_c else "%%%d" % t_
                    emit("v_mfma_i32_32x32x32_i8 %%%d, %s, %s, %s" % (t_, ring(k, ks, i), fr(f % PF), c_op))
            nxt = f + PF
            if mode == "nomfma":
                pass
            elif nxt < nm:
                emit("ds_read_b128 %s, %%%d offset:%d" % (fr(f % PF), o_vlds, k * slot_bytes + nxt * 1024))
            else:
                emit("ds_read_b128 %s, %%%d offset:%d" % (fr(f % PF), o_vlds, ((k + 1) % RING) * slot_bytes + (nxt - nm) * 1024))
            if pending:
                for ins in pending.pop(0):
                    emit(ins)
        while pending:
            for ins in pending.pop(0):
                emit(ins)

    if persistent:
        # The queue holds, oldest first: the compiler's result stores of the previous job, the B stages 0..2 (gen_bissue) and now
        # the A operands of stages 0..2.  Everything but the last two A stages has landed at vmcnt(2 * 3 * KS).
        emit("s_waitcnt lgkmcnt(0)")
        for i in range(3):
            emit("s_mov_b64 %s, %%%d" % (sa[i], o_a[i]))
        emit("s_mov_b32 s40, 0")
        emit("s_mov_b32 s41, 0")
        emit("s_mov_b32 s43, 0")
        for s_ in range(3):
            issue_a_only(s_)
        emit("s_waitcnt vmcnt(%d)" % (2 * 3 * KS))
    else:
        emit("s_waitcnt vmcnt(0) lgkmcnt(0)")     # nothing of the compiler's may be counted with this block's operations
        for i in range(3):
            emit("s_mov_b64 %s, %%%d" % (sa[i], o_a[i]))
        emit("s_mov_b32 s40, 0")
        emit("s_mov_b32 s41, 0")
        emit("s_mov_b32 s43, 0")
        for s_ in range(3):
            issue_now(s_)
        emit("s_waitcnt vmcnt(%d)" % (2 * ops))
    emit("s_barrier")
    for f in range(PF):
        if mode != "nomfma":
            emit("ds_read_b128 %s, %%%d offset:%d" % (fr(f), o_vlds, f * 1024))
    if persistent:   # the first two stages once more in front of the loop, the very first k-tile writing fresh sums
        stage(0, first=True)
        stage(1)
        emit("s_add_u32 s41, s41, 2")
        emit("s_cmp_ge_u32 s41, %%%d" % o_nst)
        emit("s_cbranch_scc1 2f")
        stage(2)
        stage(3)
        emit("s_add_u32 s41, s41, 2")
        emit("s_cmp_ge_u32 s41, %%%d" % o_nst)
        emit("s_cbranch_scc1 2f")
    emit("1:")
    stage(0)
    stage(1)
    emit("s_add_u32 s41, s41, 2")
    emit("s_cmp_ge_u32 s41, %%%d" % o_nst)
    emit("s_cbranch_scc1 2f")
    stage(2)
    stage(3)
    emit("s_add_u32 s41, s41, 2")
    emit("s_cmp_lt_u32 s41, %%%d" % o_nst)
    emit("s_cbranch_scc1 1b")
    emit("2:")
    emit("s_waitcnt vmcnt(0) lgkmcnt(0)")   # no DMA may still be writing this block's LDS when it ends
    emit("s_nop 15")                        # the last matrix instructions are still writing their tiles (16 passes)
    emit("s_nop 15")
    emit("s_nop 7")
    clob = ['"v%d"' % r for r in range(RING0, FR0 + 4 * PF)] + ['"s%d"' % r for r in range(40, 52)] + ['"scc"', '"memory"']
    return L, clob


def gen_bissue(NT):
    """B stages 0..2 of a job by LDS-DMA into ring slots 0..2 (k_gemm_i8_w3p issues them for job b+1 as soon as the K loop of job b
    has ended, before its results are recombined).  Operands: %0 BG (+v, 64-bit, advanced by three stages like the loop's own
    issue would), %1 M0B (s), %2 NST (s).  The barrier: no wave may still be reading the previous job's fragments from the ring."""
    pieces = KS * NT
    ppw = (pieces + 3) // 4
    slot_bytes = pieces * 1024
    L = ["s_waitcnt lgkmcnt(0)", "s_barrier", "s_mov_b32 s45, 0", "s_mov_b32 s43, 0"]
    for slot in range(3):
        L += ["s_add_u32 m0, %%1, %d" % (slot * slot_bytes), "s_nop 0", "global_load_lds_dwordx4 %0, off"]
        for q in range(1, ppw):
            L.append("global_load_lds_dwordx4 %%0, off offset:%d" % (q * 1024))
        L += ["s_add_u32 s45, s45, 1", "s_cmp_lt_u32 s45, %2", "s_cselect_b32 s42, %d, 0" % (pieces * 1024),
              "v_lshl_add_u64 %0, %0, 0, s[42:43]"]
    return L, ['"s42"', '"s43"', '"s45"', '"scc"', '"memory"']


def main():
    out = ["// GENERATED by tools/gen_gemm_w3_asm.py — do not edit.  The K loop of k_gemm_i8_w3 (see mfma_kernels.hip).", ""]
    for NT in (3, 6):
        for ntl in (0, 1):
            L, clob = gen(NT, ntl)
            name = "NT%d_%s" % (NT, "NTLOAD" if ntl else "PLAIN")
            out.append("#define W3_LOOP_ASM_%s \\" % name)
            for k, ins in enumerate(L):
                out.append('    "%s\\n\\t"%s' % (ins, " \\" if k + 1 < len(L) else ""))
            out.append("")
            out.append("#define W3_LOOP_CLOBBERS_%s %s" % (name, ", ".join(clob)))
            out.append("")
    # the persistent-block variant (k_gemm_i8_w3p: many jobs share the matrix): plain A loads, 64 sets
    L, clob = gen(6, 0, persistent=True)
    out.append("#define W3P_LOOP_ASM_NT6 \\")
    for k, ins in enumerate(L):
        out.append('    "%s\\n\\t"%s' % (ins, " \\" if k + 1 < len(L) else ""))
    out.append("")
    # ... with the products of equal weight sharing an accumulator tile (10 tiles)
    L, clob = gen(6, 0, persistent=True, merged=True)
    out.append("#define W3PM_LOOP_ASM_NT6 \\")
    for k, ins in enumerate(L):
        out.append('    "%s\\n\\t"%s' % (ins, " \\" if k + 1 < len(L) else ""))
    out.append("")
    L, clob = gen_bissue(6)
    out.append("#define W3P_BISSUE_ASM_NT6 \\")
    for k, ins in enumerate(L):
        out.append('    "%s\\n\\t"%s' % (ins, " \\" if k + 1 < len(L) else ""))
    out.append("")
    out.append("#define W3P_BISSUE_CLOBBERS %s" % ", ".join(clob))
    out.append("")
    # diagnostic variants of the outer commitment's loop (NT = 6, non-temporal A loads): lab_ctx_gemm_diag modes 2 and 3
    for mode in ("noload", "nomfma"):
        L, clob = gen(6, 1, mode)
        out.append("#define W3_LOOP_ASM_NT6_%s \\" % mode.upper())
        for k, ins in enumerate(L):
            out.append('    "%s\\n\\t"%s' % (ins, " \\" if k + 1 < len(L) else ""))
        out.append("")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
