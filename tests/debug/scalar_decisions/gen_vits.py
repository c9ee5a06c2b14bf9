#!/usr/bin/env python3
"""Generates abracadabra_amd/csrc/dabx_vits.inc: inline-asm text of k_viterbi_s
(decisions as 64-bit lane masks in SGPRs -> scalar stores; traceback entirely on the scalar unit).

Forward, 32 steps per asm statement and starting phase (0, 2, 4).  Per step:
    v_dot4 S, v_dot4 K          path metric +- branch metric (send / keep)
    [v_readlane x(j+1)]         soft values of the next step (MODE readlane) -- or they are
                                already in SGPRs x0..x31 from an s_load (MODE sload)
    v_cmp_gt_i32 pair(j-1)      decision mask of the PREVIOUS step: pm(j-1) > K(j-1)  <=>  received > kept
    s_store_dwordx4 / s_nop     two steps of masks to the wave's scratch slot
    v_max_i32 (DPP-fused exchange, or LDS crossbar exchange for xor 16 / xor 32)
gfx950 needs 3 wait states between a DOT write and another VALU's read of it (2 before a DPP read);
the compare/store/nop instructions fill that gap.

Traceback, 96 steps per asm statement: 8-step chunks (one s_load_dwordx16) double buffered in SGPRs;
per step  s_bitcmp1_b64 mask, L ; s_cselect_b32 T, basis[ph], 0 ; s_xor_b32 L, L, T  (L = survivor's LANE);
per 6 steps the lane number is appended to a 48-bit accumulator; lane -> basis coordinates (= the
six decoded bits) is done on the packed accumulators.
"""
import os

DPP = {0: "quad_perm:[1,0,3,2]", 1: "quad_perm:[2,3,0,1]", 2: "row_half_mirror", 3: "row_ror:8"}
BASIS = [1, 2, 7, 8, 16, 32]          # lane-number toggle of the step with phase ph
QBASE = 36                            # forward: decision quads s[36:51]
NQUAD = 4
TBASE = 52                            # traceback: mask buffers s[52:83], temporaries s[84:97]
XREGS = ("s98", "s99")                # merged text: soft values of the current / next step
STORE_PHASES = (0, 1, 2)              # not in the steps right before the LDS exchanges: their lgkmcnt(0) also waits for stores


def pair(j):
    k = j // 2
    b = QBASE + 4 * (k % NQUAD) + 2 * (j % 2)
    return f"s[{b}:{b + 1}]"


def quad(k):
    b = QBASE + 4 * (k % NQUAD)
    return f"s[{b}:{b + 3}]"


def fwd_steps(ph0, nsteps, xv_of=lambda j: ("xv", j), idx0=None, xregs=("%[xa]", "%[xb]")):
    """Forward pass as a list of steps; every step is a list of (kind, text) with kind in
    'dot' (the two dot4), 'mid' (instructions between the dot4 pair and the max), 'max' (exchange + max).
    Decisions of step j go to [dpf + 8 j]."""
    steps = []
    xv0, l0 = xv_of(0)
    pre = [f"v_readlane_b32 {xregs[0]}, %[{xv0}], {idx0 if idx0 is not None else l0}", "s_nop 1"]
    pending = []                       # (k, step of its last compare): quads complete but not yet stored
    held = {}                          # quad register set -> k it holds (not yet stored)
    for j in range(nsteps + 1):
        ph = (ph0 + j) % 6
        last = j == nsteps             # pseudo step: only the decision of step nsteps-1
        k_cur, k_prev = ("Ka", "Kb") if j % 2 == 0 else ("Kb", "Ka")
        xs, xn = (xregs[0], xregs[1]) if j % 2 == 0 else (xregs[1], xregs[0])
        t = []
        if not last:
            t += [("dot", f"v_dot4_i32_i8 %[S], %[n{ph}], {xs}, %[pm]"),
                  ("dot", f"v_dot4_i32_i8 %[{k_cur}], %[s{ph}], {xs}, %[pm]")]
        if j + 1 < nsteps:
            xvn, ln = xv_of(j + 1)
            t.append(("mid", f"v_readlane_b32 {xn}, %[{xvn}], {ln}"))
        if j > 0:
            kq = (j - 1) // 2
            rs = kq % NQUAD
            assert held.get(rs, kq) == kq, "decision quad overwritten before it was stored"
            held[rs] = kq
            t.append(("mid", f"v_cmp_gt_i32_e64 {pair(j - 1)}, %[pm], %[{k_prev}]"))
            if (j - 1) % 2 == 1 or last:
                pending.append((kq, j))            # complete after this compare
        if last:
            t.append(("mid", "s_nop 3"))
            for kq, _ in pending:
                if 2 * kq + 1 < nsteps:
                    t.append(("mid", f"s_store_dwordx4 {quad(kq)}, %[dpf], {16 * kq}"))
                else:                               # odd count: only the first half of the quad is valid
                    b = QBASE + 4 * (kq % NQUAD)
                    t.append(("mid", f"s_store_dwordx2 s[{b}:{b + 1}], %[dpf], {16 * kq}"))
            pending = []
        else:
            can = [p for p in pending if p[1] < j]  # the last compare is at least one full step old
            if ph in STORE_PHASES:
                for p in can[:2]:
                    kq, _ = p
                    pending.remove(p)
                    del held[kq % NQUAD]
                    t.append(("mid", f"s_store_dwordx4 {quad(kq)}, %[dpf], {16 * kq}"))
            if ph < 4:
                t.append(("max", f"v_max_i32_dpp %[pm], %[S], %[{k_cur}] {DPP[ph]} row_mask:0xf bank_mask:0xf"))
            else:
                ex = "ds_swizzle_b32 %[D], %[S] offset:swizzle(SWAP,16)" if ph == 4 else "ds_bpermute_b32 %[D], %[ad], %[S]"
                t += [("max", ex), ("max", "s_waitcnt lgkmcnt(0)"), ("max", f"v_max_i32 %[pm], %[{k_cur}], %[D]")]
        steps.append(t)
    return pre, steps


def weave(pre, steps, extra):
    """Flatten the forward steps, dropping the instructions of `extra` (the traceback stream, in order)
    between them; s_nop pads keep 3 wait states between the second dot4 and the max where needed."""
    lines = list(pre)
    n = len(steps)
    pos = 0
    for j, t in enumerate(steps):
        want = (len(extra) - pos + (n - j) - 1) // (n - j)            # spread what is left evenly
        take = extra[pos:pos + want]
        pos += want
        dots = [x for k, x in t if k == "dot"]
        mids = [x for k, x in t if k == "mid"]
        maxs = [x for k, x in t if k == "max"]
        # interleave: one traceback instruction after each forward 'mid' instruction, the rest after the max
        body, ti = [], 0
        for m in mids:
            body.append(m)
            if ti < len(take):
                body.append(take[ti]); ti += 1
        if dots and maxs and len(body) < 3:
            body.append(f"s_nop {2 - len(body)}")
        lines += dots + body + maxs + take[ti:]
    assert pos == len(extra)
    return lines


def fwd_text(ph0, nsteps, mode="readlane", idx0=None):
    pre, steps = fwd_steps(ph0, nsteps, idx0=idx0)
    return weave(pre, steps, [])


def tb_text(noload=False, nostore_dummy=False):
    """96 steps [0, 96) relative to %[dpt]; in/out %[L]; out %[o0..o2]: step 32 k + j at bit 31 - j of o_k."""
    buf = [TBASE, TBASE + 16]
    r = lambda k: f"s{TBASE + 32 + k}"
    rr = lambda k: f"s[{TBASE + 32 + k}:{TBASE + 33 + k}]"
    T, ACC, MSK, TMP, TMP2, HI, LO = r(0), rr(2), rr(4), rr(6), rr(8), rr(10), rr(12)
    acc_lo = r(2)
    lines = [f"s_load_dwordx16 s[{buf[1]}:{buf[1] + 15}], %[dpt], {11 * 64}",
             f"s_mov_b32 {r(4)}, 0x41041041", f"s_mov_b32 {r(5)}, 0x410"]

    def convert(dst):                  # lane numbers -> coordinates on 8 packed groups
        return [f"s_lshr_b64 {TMP}, {ACC}, 2", f"s_and_b64 {TMP}, {TMP}, {MSK}", f"s_lshl_b64 {TMP2}, {TMP}, 1",
                f"s_or_b64 {TMP}, {TMP}, {TMP2}", f"s_xor_b64 {dst}, {ACC}, {TMP}"]

    for c in range(11, -1, -1):
        b = buf[c % 2]
        lines.append("s_waitcnt lgkmcnt(0)")
        if c > 0:
            nb = buf[(c - 1) % 2]
            lines.append(f"s_load_dwordx16 s[{nb}:{nb + 15}], %[dpt], {(c - 1) * 64}")
        for i in range(7, -1, -1):
            t = 8 * c + i
            ph = t % 6
            if ph == 5:
                if t in (95, 47):
                    lines.append(f"s_mov_b64 {ACC}, 0")
                lines += [f"s_lshl_b64 {ACC}, {ACC}, 6", f"s_or_b32 {acc_lo}, {acc_lo}, %[L]"]
            lines += [f"s_bitcmp1_b64 s[{b + 2 * i}:{b + 2 * i + 1}], %[L]", f"s_cselect_b32 {T}, {BASIS[ph]}, 0",
                      f"s_xor_b32 %[L], %[L], {T}"]
            if t == 48:                # steps 48..95 done
                lines += convert(HI)
    lines += convert(LO)               # steps 0..47
    # o0 = steps 0..31, o1 = steps 32..63, o2 = steps 64..95 (bit i of the stream = step i), then bit-reverse
    if noload:                         # debug variant (tests/debug/vits_cost.hip): same SALU work, no SMEM
        lines = [l for l in lines if not l.startswith("s_load") and not l.startswith("s_waitcnt")]
    lines += [f"s_brev_b32 %[o0], {r(12)}",
              f"s_lshl_b32 {T}, {r(10)}, 16", f"s_or_b32 {T}, {T}, {r(13)}", f"s_brev_b32 %[o1], {T}",
              f"s_lshr_b64 {ACC}, {HI}, 16", f"s_brev_b32 %[o2], {acc_lo}"]
    return lines


def emit(out, name, ls):
    out.append(f"#define {name} \\")
    for i, l in enumerate(ls):
        end = "" if i == len(ls) - 1 else " \\"
        sep = "" if i == len(ls) - 1 else "\\n\\t"
        out.append(f'    "{l}{sep}"{end}')
    out.append("")


def merged_text_debug(noload, nostore):
    pre, steps = fwd_steps(0, 96, xv_of=lambda j: (f"xv{j // 32}", j % 32), xregs=XREGS)
    if nostore:
        steps = [[(k, x) for k, x in t if not x.startswith("s_store")] for t in steps]
    return weave(pre, steps, tb_text(noload))


def merged_text():
    """96 forward steps of one codeword (soft values: lanes 0..31 of xv0, xv1, xv2) woven with the
    96-step traceback block of the previous codeword."""
    pre, steps = fwd_steps(0, 96, xv_of=lambda j: (f"xv{j // 32}", j % 32), xregs=XREGS)
    return weave(pre, steps, tb_text())


def main():
    out = ["// GENERATED by tools/gen_vits.py — do not edit.",
           "// forward: pm (+v)  S,Ka,Kb,D (=&v)  xa,xb (=&s)  xv,ad,s0..s5,n0..n5 (v)  dpf (s, 64-bit)",
           "// traceback: L (+s)  o0,o1,o2 (=&s)  dpt (s, 64-bit)",
           f"// merged: both, with xv0,xv1,xv2 instead of xv and {XREGS[0]},{XREGS[1]} instead of xa,xb",
           f"// clobbers: s[{QBASE}:{QBASE + 4 * NQUAD - 1}] (forward), s[{TBASE}:{TBASE + 45}] (traceback), scc, memory"]
    for ph0 in (0, 2, 4):
        emit(out, f"DABX_FWD32_TEXT_{ph0}", fwd_text(ph0, 32))
    for ph0 in range(6):
        emit(out, f"DABX_FWD1_TEXT_{ph0}", fwd_text(ph0, 1, idx0="%[idx]"))
    emit(out, "DABX_TB96_TEXT", tb_text())
    emit(out, "DABX_FWDTB96_TEXT", merged_text())
    if os.environ.get("GEN_DEBUG"):    # variants for tests/debug/vits_cost.hip
        dbg = ["// GENERATED by GEN_DEBUG=1 tools/gen_vits.py — microbenchmark variants"]
        emit(dbg, "DABX_TB96_NOLOAD_TEXT", tb_text(True))
        emit(dbg, "DABX_FWDTB96_NOLOAD_TEXT", merged_text_debug(True, False))
        emit(dbg, "DABX_FWDTB96_NOSTORE_TEXT", merged_text_debug(False, True))
        emit(dbg, "DABX_FWDTB96_NOMEM_TEXT", merged_text_debug(True, True))
        pre, steps = fwd_steps(0, 96, xv_of=lambda j: (f"xv{j // 32}", j % 32), xregs=XREGS)
        emit(dbg, "DABX_FWD96_NOSTORE_TEXT", weave(pre, [[(k, x) for k, x in t if not x.startswith("s_store")] for t in steps], []))
        dpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "debug", "vits_debug.inc")
        open(dpath, "w").write("\n".join(dbg))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "abracadabra_amd", "csrc", "dabx_vits.inc")
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
