#!/usr/bin/env python3
"""Writes motion324_amd/csrc/gemm_hp_*.inc: the hand-placed instruction streams of schedule v15 (gemm_hp.hip), the K = 768 GEMM
with the PREVIOUS tile's epilogue issued between the MFMAs of the current tile's main loop.  Run it after editing; the .inc files
are committed, the build does not need this script (tests/test_static.py checks that they are what this script writes).

Why (profiles/r05_gemm_labs.md, profiles/r05_gemm_hp.md): the K = 768 GEMMs are bound by what a CU can issue -- 12.3 k cycles of MFMA
per 256 x 128 tile, 1200-2000 VALU instructions of epilogue per wave and tile, and neither a partner workgroup (v14) nor
compiler-scheduled fillers (tools/lab_src/de_lab.hip) hide the epilogue; a wave's OWN plain VALU instructions do issue in the shadow of
its MFMAs when they sit between them in program order (tools/issue_lab, round 4; the attention stream of gen_attn_pwg.py rests on the
same fact).

Structure: ONE persistent 4-wave workgroup per CU (one wave per SIMD, 512 registers), 256 x 128 output tiles, wave (wm, wn) owns
128 rows x 64 columns = 4 x 2 accumulator blocks of 32 x 32 (swapped operands: a lane holds ONE output row and 4-column runs).
  * K-stages of 64: three 48-KiB LDS buffers [X 256 rows x 128 B | W 128 rows x 128 B], filled by LDS-DMA two stages ahead;
    ONE barrier per stage; fragments are read two k-steps (16 MFMAs) ahead of their MFMAs into three register sets, so the first
    16 MFMAs behind a barrier are the previous stage's last two k-steps.
  * two accumulator sets in AGPRs (a0-127 / a128-255) that swap roles every tile (the tile body is emitted twice): while set X
    accumulates tile t, the epilogue of tile t - 1 reads set Y: v_accvgpr_read, LayerNorm fold / bias, the 9-term erf polynomial of
    gemm_tile.h, bf16 pack, v_permlane32_swap (16 contiguous bytes of one row per lane).  The arithmetic is PLAIN fp32 VALU: packed
    fp32 (v_pk_fma_f32 ...) does not issue in the MFMAs' shadow (measured: 832 packed instructions per tile cost 4.2 cycles each,
    1856 plain ones in their place 2.6); only the exposed tail behind a workgroup's last tile uses the packed forms.
  * stores: a row block (32 rows x 64 columns of the wave) leaves through the wave's 16-row x 128-byte LDS scratch in two halves
    (exec-masked ds_write_b128 of the packed rows, read back as whole rows) and out as 8 rows x 128 contiguous bytes per instruction,
    clipped by the store resource's record count (ragged last row tile: no predicates).  The cache flag of the stores is the
    includer's macro HP_ST_FLAG ("" or " nt": outputs above M324_NT_MB leave nontemporal -- 402 MB of plain stores push the
    operands out of the L2s).  A bounce's wait never directly follows its read-back: the parts are woven into the next row block's
    first instructions.
  * tiles are seamless: stages 10, 11 of a tile fetch stages 0, 1 of the workgroup's next tile (resource words from the tile table
    the C++ side leaves in LDS); a workgroup's first tile runs a body WITHOUT epilogue fillers, its last tile is followed by the
    exposed epilogue.
  * the generator resolves every s_waitcnt from the instruction order (loads return in order among themselves, LDS operations too:
    a wait for operation T may leave as many in flight as were issued behind T; stores count in vmcnt in hardware but not here --
    the bound stays valid), checks that the counts do not depend on the path by which a body is entered (prologue, first tile, the
    other set's body), and checks the hazards the assembler does not handle inside an asm statement.
Register map: see `alloc` below (v16-255, a0-255, s40-87 belong to the asm statement).  `--lab`: timing-only ablation streams for
tools/hp_lab (no epilogue, no stores, no LDS-DMA, no fragment reads, no barriers, packed arithmetic, nt / sc1 stores, stamps).
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_attn_pwg import I, v, a, regs, salu, nop  # noqa: E402  (instruction records, register names)

HERE = os.path.dirname(os.path.abspath(__file__))

OPT = {"fold": False, "gelu": True, "noepi": False, "nostore": False, "trace": False, "nsets": 3, "nodma": False, "nobar": False, "nofrag": False, "scalar": True, "bounce": True, "stflag": "@ST", "dma_span": 28.0, "frag_span": 6.0, "epi_end": 356.0, "first": True, "weave_first": 10, "weave_gap": 36}

NS = 12                      # K-stages per tile (K = 768)
STAGE = 49152                # bytes per ring buffer: X 32 KiB + W 16 KiB
TABLE = 3 * STAGE            # the tile table behind the ring
ENTRY = 48                   # bytes per table entry

# ------------------------------------------------------------------------------------------------ registers
_next_v = 16


def alloc(n, align=None):
    global _next_v
    al = align or (4 if n >= 4 else (2 if n == 2 else 1))
    _next_v = (_next_v + al - 1) // al * al
    r = _next_v
    _next_v += n
    assert _next_v <= 256, "out of VGPRs"
    return r


XF = [[alloc(4) for _ in range(4)] for _ in range(3)]        # token-row fragments (MFMA B operand) [set][row block]
WF = [[alloc(4) for _ in range(2)] for _ in range(3)]        # weight-row fragments (MFMA A operand) [set][column block]
BIAS = alloc(32)             # bias[j * 16 + r]: column wn*64 + j*32 + 8 (r >> 2) + 4 hi + (r & 3)
CS = alloc(32)               # colsum, same indexing
RS = alloc(8)                # (rstd, -rstd mean) of the lane's row of row block i: RS + 2 i
EX, EU, ET, EQ = alloc(8), alloc(8), alloc(8), alloc(8)
P = [alloc(8), alloc(8)]     # packed bf16 of a block (two sets, alternating blocks)
FA = [alloc(4), alloc(4)]    # X fragment address per k-step: [0] buffers 0 / 1 (immediate 0 / STAGE), [1] buffer 2
FW = [alloc(4), alloc(4)]    # W fragment address
DX, DW = alloc(8), alloc(4)  # LDS-DMA lane offsets of the wave's 8 X pieces / 4 W pieces
TT = alloc(8)                # table words on their way to SGPRs; the same registers receive the bounced rows (RB) -- never at the same time
RB = TT
VTAB, CLAMPV, Q7V = alloc(1), alloc(1), alloc(1)
WA = alloc(4)                # bounce: the lane's scratch write address for chunk pair k = 2 j + half
CV = WA                      # (lab: direct stores, OPT bounce off) store offset of the lane's row in row block i (+ 16 hi)
RDA, SVO = alloc(1), alloc(1)  # bounce: scratch read address (row L >> 3, chunk L & 7), store offset of that row / chunk
V_END = _next_v

RA, RW, RC, RR = 40, 44, 48, 52          # buffer resources: X / W of the tile being fetched, C / rowstat of the epilogue's tile
GQ = 56                                  # s56-65: (Q8,Q7) (Q6,Q5) (Q4,Q3) (Q2,Q1) (Q0,-clamp)
SK, ST, NOFF, SROW = 66, 67, 68, 69       # K offset of the stage to fetch, tiles left, byte offset of the tile's columns in bias / colsum, store row offset
STMP = 70                                # s70-81 scratch / trace
EXLO, EXHI = 84, 86                      # exec masks: rows 0-15 / 16-31 of a 32-row block (lanes 0-15 + 32-47 / 16-31 + 48-63)
GELU_Q = [3.989031257e-01, -6.634691738e-02, 9.818113584e-03, -1.109398132e-03, 9.359017959e-05, -5.646163474e-06, 2.268262506e-07,
          -5.371752709e-09, 5.626603458e-11]
GELU_CLAMP = 4.2426405


def ACC(s, i, j, r=0):
    return s * 128 + (i * 2 + j) * 16 + r


def f32_bits(x):
    import struct
    import numpy as np
    return struct.unpack("<I", struct.pack("<f", float(np.float32(x))))[0]


def sreg(i, n=1):
    return f"s{i}" if n == 1 else f"s[{i}:{i + n - 1}]"


# ------------------------------------------------------------------------------------------------ instruction helpers
def valu(text, rd=(), wr=(), kind="valu"):
    return I(text, kind, rd, wr)


def vr(base, n):
    return regs("v", base, n)


class Wait(I):
    """placeholder: resolved to s_waitcnt by resolve_waits.  need: tags of the memory operations whose results must be there"""

    def __init__(self, vm=None, lgkm=None, vm0=False, lgkm0=False):
        super().__init__("s_waitcnt ?", "wait")
        self.vm, self.lgkm, self.vm0, self.lgkm0 = vm, lgkm, vm0, lgkm0


def tagged(ins, tag, q):
    ins.tag, ins.queue = tag, q       # q: "vm" (loads), "st" (stores: counted by the hardware, not by us), "lgkm"
    return ins


def mfma(s, i, j, f, zero):
    d = ACC(s, i, j)
    c = "0" if zero else a(d, 16)
    return I(f"v_mfma_f32_32x32x16_bf16 {a(d, 16)}, {v(WF[f][j], 4)}, {v(XF[f][i], 4)}, {c}", "mfma",
             rd=vr(WF[f][j], 4) + vr(XF[f][i], 4) + ([] if zero else regs("a", d, 16)), wr=regs("a", d, 16))


def frag_reads(stage, ks, f):
    """fragments of k-step ks of the stage in ring buffer stage % 3 into set f: 2 weight blocks, 4 token blocks"""
    b = stage % 3
    sel, imm = (1, 0) if b == 2 else (0, b * STAGE)
    out = []
    for j in range(2):
        out.append(tagged(I(f"ds_read_b128 {v(WF[f][j], 4)}, {v(FW[sel] + ks)} offset:{imm + j * 4096}", "ds", rd=[f"v{FW[sel] + ks}"], wr=vr(WF[f][j], 4)),
                          ("frag", stage, ks), "lgkm"))
    for i in range(4):
        out.append(tagged(I(f"ds_read_b128 {v(XF[f][i], 4)}, {v(FA[sel] + ks)} offset:{imm + i * 4096}", "ds", rd=[f"v{FA[sel] + ks}"], wr=vr(XF[f][i], 4)),
                          ("frag", stage, ks), "lgkm"))
    return out


def dma_stage(fs, bdst, tag, first=False):
    """LDS-DMA of fetch stage fs (0..11 of the tile whose resources sit in RA / RW) into ring buffer bdst: 8 X + 4 W pieces"""
    out = []
    if not first:
        out.append(salu(f"s_mov_b32 {sreg(SK)}, {fs * 128}"))
    for i in range(8):
        out.append(salu(f"s_add_u32 m0, %[wldsx], {bdst * STAGE + i * 1024}"))
        out.append(tagged(I(f"buffer_load_dwordx4 {v(DX + i)}, {sreg(RA, 4)}, {sreg(SK)} offen lds", "vmem", rd=[f"v{DX + i}"]), tag, "vm"))
    for i in range(4):
        out.append(salu(f"s_add_u32 m0, %[wldsw], {bdst * STAGE + i * 1024}"))
        out.append(tagged(I(f"buffer_load_dwordx4 {v(DW + i)}, {sreg(RW, 4)}, {sreg(SK)} offen lds", "vmem", rd=[f"v{DW + i}"]), tag, "vm"))
    return out


def fix_m0(seq):
    out = []
    for ins in seq:
        if ins.kind == "vmem" and " lds" in ins.text and out and out[-1].kind == "salu" and " m0," in out[-1].text:
            out.append(nop(1))
        out.append(ins)
    return out


def e_loads(tag):
    """bias / colsum / row statistics of the tile whose words sit in NOFF / RR, for the epilogue that runs one tile later"""
    out = []
    for j in range(2):
        for q in range(4):
            out.append(tagged(I(f"buffer_load_dwordx4 {v(BIAS + j * 16 + 4 * q, 4)}, %[bo], %[rb], {sreg(NOFF)} offen offset:{j * 128 + q * 32}", "vmem",
                                wr=vr(BIAS + j * 16 + 4 * q, 4)), tag, "vm"))
    if OPT["fold"]:
        for j in range(2):
            for q in range(4):
                out.append(tagged(I(f"buffer_load_dwordx4 {v(CS + j * 16 + 4 * q, 4)}, %[bo], %[rcs], {sreg(NOFF)} offen offset:{j * 128 + q * 32}", "vmem",
                                    wr=vr(CS + j * 16 + 4 * q, 4)), tag, "vm"))
        for i in range(4):
            out.append(tagged(I(f"buffer_load_dwordx2 {v(RS + 2 * i, 2)}, %[ro], {sreg(RR, 4)}, 0 offen offset:{i * 256}", "vmem", wr=vr(RS + 2 * i, 2)), tag, "vm"))
    return out


def table_read(dst, off, tag):
    return tagged(I(f"ds_read_b128 {v(dst, 4)}, {v(VTAB)} offset:{off}", "ds", rd=[f"v{VTAB}"], wr=vr(dst, 4)), tag, "lgkm")


def rfl(s, vreg):
    return I(f"v_readfirstlane_b32 {sreg(s)}, {v(vreg)}", "valu", rd=[f"v{vreg}"], wr=[f"s{s}"])


# ------------------------------------------------------------------------------------------------ the epilogue of one block
def arith(op, d, *srcs):
    """d (register pair base) = op(srcs) on two values.  A source is ("v", base): the pair base, base + 1 | ("vb", r): VGPR r for both |
    ("s", r): SGPR r for both | ("c", text): an inline constant.  OPT scalar: two plain VALU instructions (v_pk_* does not issue in
    the shadow of the wave's MFMAs: measured, 4.2 cycles each on top of the main loop); else one packed instruction."""
    name = {"fma": "fma", "mul": "mul", "add": "add"}[op]
    if OPT["scalar"]:
        out = []
        for e in range(2):
            ops, rd = [], []
            for kind, x in srcs:
                if kind == "v":
                    ops.append(v(x + e)); rd.append(f"v{x + e}")
                elif kind == "vb":
                    ops.append(v(x)); rd.append(f"v{x}")
                elif kind == "s":
                    ops.append(sreg(x))
                else:
                    ops.append(x)
            if op == "fma":
                out.append(valu(f"v_fma_f32 {v(d + e)}, {', '.join(ops)}", rd=rd, wr=[f"v{d + e}"]))
            else:
                # VOP2: src0 any, src1 a VGPR
                if not ops[1].startswith("v"):
                    ops = [ops[1], ops[0]]
                out.append(valu(f"v_{name}_f32_e32 {v(d + e)}, {', '.join(ops)}", rd=rd, wr=[f"v{d + e}"]))
        return out
    ops, rd, sel, selhi = [], [], [], []
    for kind, x in srcs:
        if kind == "v":
            ops.append(v(x, 2)); rd += vr(x, 2); sel.append(0); selhi.append(1)
        elif kind == "vb":
            ops.append(v(x & ~1, 2)); rd.append(f"v{x}"); sel.append(x & 1); selhi.append(x & 1)
        elif kind == "s":
            ops.append(sreg(x & ~1, 2)); sel.append(x & 1); selhi.append(x & 1)
        else:
            ops.append(x); sel.append(0); selhi.append(0)
    mods = f" op_sel:[{','.join(map(str, sel))}] op_sel_hi:[{','.join(map(str, selhi))}]" if op == "fma" else ""
    if op != "fma":
        assert all(k == "v" for k, _ in srcs)
    return [valu(f"v_pk_{name}_f32 {v(d, 2)}, {', '.join(ops)}{mods}", rd=rd, wr=vr(d, 2))]


def epi_block(s, i, j, pset, store=True):
    """epilogue of accumulator block (i, j) of set s -> (items, stores).  16 values per lane: row l31 of row block i, columns
    j*32 + 8 q + 4 hi + e (register 4 q + e)."""
    out = []
    for h in range(2):
        for k in range(8):
            out.append(valu(f"v_accvgpr_read_b32 {v(EX + k)}, {a(ACC(s, i, j, 8 * h + k))}", rd=[f"a{ACC(s, i, j, 8 * h + k)}"], wr=[f"v{EX + k}"]))
        b0 = BIAS + j * 16 + 8 * h
        c0 = CS + j * 16 + 8 * h
        if OPT["fold"]:
            for p in range(4):         # t = rs.y * colsum + bias
                out += arith("fma", ET + 2 * p, ("vb", RS + 2 * i + 1), ("v", c0 + 2 * p), ("v", b0 + 2 * p))
            for p in range(4):         # x = rs.x * acc + t
                out += arith("fma", EX + 2 * p, ("vb", RS + 2 * i), ("v", EX + 2 * p), ("v", ET + 2 * p))
        else:
            for p in range(4):
                out += arith("add", EX + 2 * p, ("v", EX + 2 * p), ("v", b0 + 2 * p))
        if OPT["gelu"]:
            for k in range(8):
                out.append(valu(f"v_med3_f32 {v(EU + k)}, {v(EX + k)}, {sreg(GQ + 9)}, {v(CLAMPV)}", rd=[f"v{EX + k}", f"v{CLAMPV}"], wr=[f"v{EU + k}"]))
            for p in range(4):
                out += arith("mul", ET + 2 * p, ("v", EU + 2 * p), ("v", EU + 2 * p))
            for p in range(4):         # q = Q8 t + Q7 (one scalar operand per instruction: Q7 comes from a VGPR)
                out += arith("fma", EQ + 2 * p, ("v", ET + 2 * p), ("s", GQ), ("vb", Q7V) if OPT["scalar"] else ("s", GQ + 1))
            for step in range(6, -1, -1):      # q = q t + Q_step
                for p in range(4):
                    out += arith("fma", EQ + 2 * p, ("v", EQ + 2 * p), ("v", ET + 2 * p), ("s", GQ + 8 - step))
            for p in range(4):         # q = u q + 1/2
                out += arith("fma", EQ + 2 * p, ("v", EU + 2 * p), ("v", EQ + 2 * p), ("c", "0.5"))
            for p in range(4):
                out += arith("mul", EX + 2 * p, ("v", EX + 2 * p), ("v", EQ + 2 * p))
        for p in range(4):
            d = P[pset] + 4 * h + p
            out.append(valu(f"v_cvt_pk_bf16_f32 {v(d)}, {v(EX + 2 * p)}, {v(EX + 2 * p + 1)}", rd=[f"v{EX + 2 * p}", f"v{EX + 2 * p + 1}"], wr=[f"v{d}"]))
    out.append(nop(2))
    for half in range(2):
        for k in range(2):
            x, y = P[pset] + 4 * half + k, P[pset] + 4 * half + 2 + k
            out.append(valu(f"v_permlane32_swap_b32_e32 {v(x)}, {v(y)}", rd=[f"v{x}", f"v{y}"], wr=[f"v{x}", f"v{y}"], kind="perm"))
    out.append(nop(2))
    stores = []
    if store and not OPT["nostore"] and not OPT["bounce"]:
        for half in range(2):
            stores.append(tagged(I(f"buffer_store_dwordx4 {v(P[pset] + 4 * half, 4)}, {v(CV + i)}, {sreg(RC, 4)}, 0 offen offset:{j * 64 + half * 32}", "vmem",
                                   rd=vr(P[pset] + 4 * half, 4) + [f"v{CV + i}"]), ("store",), "st"))
    return out, stores


def bounce(i):
    """row block i of the tile (both column blocks packed: P[0] = j 0, P[1] = j 1): through the wave's 16-row x 128-byte scratch in two
    halves (rows l31 < 16, then the rest), back as whole rows -- lane L holds chunk L & 7 of row (L >> 3) + 8 p -- and out: a store
    instruction writes 8 rows x 128 contiguous bytes instead of 32 rows x 32.  The writes of a half run under an exec mask: ONE
    item, so that no other stream's instruction lands inside.  Returns four parts [write + read-back A, wait + stores A + write +
    read-back B, wait + stores B]: the caller puts other work between them (a wait right behind its read-back stalls the wave for
    the LDS round trip -- nothing else of this wave issues meanwhile, MFMAs included)."""
    parts = []
    for half16 in range(2):
        lines = [f"s_mov_b64 exec, {sreg(EXHI if half16 else EXLO, 2)}"]
        rd = []
        for k in range(4):            # chunk pair k = 2 j + half: registers P[j] + 4 half .. + 3
            src = P[k >> 1] + 4 * (k & 1)
            lines.append(f"ds_write_b128 {v(WA + k)}, {v(src, 4)}")
            rd += vr(src, 4) + [f"v{WA + k}"]
        lines.append("s_mov_b64 exec, -1")
        w = I("\\n".join(lines), "ds", rd=rd)
        w.states = 6
        w.lds_ops = 4
        issue = [tagged(w, ("bw", i, half16), "lgkm")]
        for p in range(2):
            issue.append(tagged(I(f"ds_read_b128 {v(RB + 4 * p, 4)}, {v(RDA)} offset:{p * 1024}", "ds", rd=[f"v{RDA}"], wr=vr(RB + 4 * p, 4)), ("br", i, half16), "lgkm"))
        drain = [Wait(lgkm=("br", i, half16))]
        for p in range(2):
            if not OPT["nostore"]:
                drain.append(tagged(I(f"buffer_store_dwordx4 {v(RB + 4 * p, 4)}, {v(SVO)}, {sreg(RC, 4)}, {sreg(SROW)} offen" + OPT['stflag'], "vmem",
                                      rd=vr(RB + 4 * p, 4) + [f"v{SVO}"]), ("store",), "st"))
            drain.append(salu(f"s_add_u32 {sreg(SROW)}, {sreg(SROW)}, %[cs8]"))
        parts += [issue, drain]
    return [parts[0], parts[1] + parts[2], parts[3]]


def weave(main, parts, first=8, gap=40):
    """main with parts[0] in front, parts[k] behind the first + (k - 1) * gap-th instruction of main"""
    out = list(parts[0])
    cuts = [first + k * gap for k in range(len(parts) - 1)]
    pos = 0
    for k, c in enumerate(cuts):
        c = min(c, len(main))
        out += main[pos:c] + parts[k + 1]
        pos = c
    return out + main[pos:]


# ------------------------------------------------------------------------------------------------ placement
COST = {"valu": 1.0, "perm": 1.0, "trans": 2.0, "ds": 1.0, "vmem": 1.0, "salu": 0.7, "nop": 0.3, "wait": 0.3, "barrier": 0.3}


def spread(items, lo, hi):
    tot = sum(COST[i.kind] for i in items) or 1.0
    acc, out = 0.0, []
    for it in items:
        out.append((lo + (hi - lo) * acc / tot, it))
        acc += COST[it.kind]
    return out


def at(items, g, step=1e-4):
    return [(g + k * step, it) for k, it in enumerate(items)]


def interleave(mfmas, timed):
    timed = sorted(enumerate(timed), key=lambda e: (e[1][0], e[0]))
    out, k = [], 0
    while k < len(timed) and timed[k][1][0] < 0:          # in front of the first MFMA
        out.append(timed[k][1][1])
        k += 1
    for g, m in enumerate(mfmas):
        out.append(m)
        while k < len(timed) and timed[k][1][0] < g + 1:
            out.append(timed[k][1][1])
            k += 1
    out += [e[1][1] for e in timed[k:]]
    return out


def body(sx, label, epilogue=True):
    """one tile into accumulator set sx with the epilogue of set 1 - sx; VTAB points at the table entry of the PREVIOUS tile.
    epilogue False: a workgroup's first tile (nothing to finish: the same stream without the filler instructions)"""
    sy = 1 - sx
    ns = OPT["nsets"]
    mf, tim = [], []
    for W in range(4 * NS):                     # window W: 8 MFMAs of k-step W - 2 (W < 2: the previous tile's k-steps 46, 47 into set sy)
        K = W - 2
        s_, kk, zero = (sy, K + 4 * NS, False) if K < 0 else (sx, K, K == 0)
        mf += [mfma(s_, i, j, kk % ns, zero) for i in range(4) for j in range(2)]
    for s in range(NS):
        g0 = 32 * s
        top = [Wait(vm=None if OPT["nodma"] else ("piece", s), lgkm0=True)] + ([] if OPT["nobar"] else [I("s_barrier", "barrier")])
        if OPT["trace"]:
            # s78:79 = stamp in front of the wait, s74:75 = behind the barrier; s77 += time spent in (wait + barrier) of the PREVIOUS
            # top (its stamps are retired by this top's lgkmcnt(0)); s80 = the very first stamp, s76 scratch
            top = [I(f"s_memtime {sreg(STMP + 8, 2)}", "salu"), top[0],
                   salu(f"s_sub_u32 {sreg(STMP + 6)}, {sreg(STMP + 4)}, {sreg(STMP + 2)}"), salu(f"s_add_u32 {sreg(STMP + 7)}, {sreg(STMP + 7)}, {sreg(STMP + 6)}"),
                   salu(f"s_mov_b32 {sreg(STMP + 2)}, {sreg(STMP + 8)}"), salu(f"s_cmp_eq_u32 {sreg(STMP + 10)}, 0"),
                   salu(f"s_cselect_b32 {sreg(STMP + 10)}, {sreg(STMP + 8)}, {sreg(STMP + 10)}")] + top[1:] + [I(f"s_memtime {sreg(STMP + 4, 2)}", "salu")]
        tim += at(top, g0 - 0.5)
        # fetch stage s + 2 (stages 10, 11: stages 0, 1 of the next tile, whose resource words were read in stage 9)
        fs = (s + 2) % NS
        if s == 10:
            tim += at([rfl(RA + 0, TT + 0), rfl(RA + 1, TT + 1), rfl(RA + 2, TT + 2), rfl(RW + 0, TT + 3), rfl(RW + 1, TT + 4)], g0 - 0.4)
        if not OPT["nodma"]:
            tim += spread(dma_stage(fs, (s + 2) % 3, ("piece", (s + 2) % NS)), g0 + 1.5, g0 + 1.5 + (10.0 if s == 10 else OPT["dma_span"]))      # stage 10: the epilogue loads follow its pieces
        for q in range(4):
            Kp = 4 * s + q
            if q > 0:                            # window 0's MFMAs follow the top wait (lgkmcnt(0))
                tim += at([Wait(lgkm=("frag", ((Kp - 2) // 4) % NS, (Kp - 2) % 4))], g0 + 8 * q - 0.01)
            if not OPT["nofrag"]:
                tim += spread(frag_reads(s, q, Kp % ns), g0 + 8 * q + 0.0, g0 + 8 * q + OPT["frag_span"])
    # table: C resource of the previous tile (entry + 16: W hi, C lo, C hi, C rec)
    tim += at([table_read(TT + 4, 16, ("tabC",))], 1.0)
    tim += at([Wait(lgkm=("tabC",)), rfl(RC + 0, TT + 5), rfl(RC + 1, TT + 6), rfl(RC + 2, TT + 7)], 12.5)
    # next tile's X / W resources: entry + 96 (A lo, A hi, A rec, W lo), entry + 112 (W hi, ...): read in stage 9, to SGPRs at stage 10's top
    tim += at([table_read(TT + 0, 2 * ENTRY, ("tabF",)), table_read(TT + 4, 2 * ENTRY + 16, ("tabF",))], 32 * 9 + 20.0)
    # this tile's epilogue words: entry + 48 + 32 (n_off, rs lo, rs hi, rs rec): read in stage 9 behind the fetch words' trip to SGPRs
    tim += at([table_read(TT + 0, ENTRY + 32, ("tabE",))], 32 * 10 + 1.0)
    tim += at([Wait(lgkm=("tabE",)), rfl(NOFF, TT + 0), rfl(RR + 0, TT + 1), rfl(RR + 1, TT + 2), rfl(RR + 2, TT + 3)], 32 * 10 + 12.5)
    tim += spread(e_loads(("eload",)), 32 * 10 + 16.0, 32 * 10 + 28.0)
    tim += at([valu(f"v_add_u32_e32 {v(VTAB)}, {ENTRY}, {v(VTAB)}", rd=[f"v{VTAB}"], wr=[f"v{VTAB}"])], 32 * 11 + 20.0)
    # the previous tile's epilogue
    if not OPT["noepi"] and epilogue:
        items, blocks = [Wait(vm=("eload",)), salu(f"s_mov_b32 {sreg(SROW)}, 0")], []
        k = 0
        pending = None                            # the previous row block's bounce, woven into this row block's first instructions
        for i in range(4):
            row = []
            for j in range(2):
                it, st = epi_block(sy, i, j, k & 1)
                if not OPT["bounce"]:
                    blocks.append((len(items) + len(row), len(items) + len(row) + len(it), st))
                row += it
                k += 1
            items += weave(row, pending, OPT["weave_first"], OPT["weave_gap"]) if pending else row
            pending = bounce(i) if OPT["bounce"] else None
        if pending:                               # the last row block's: MFMAs and the main loop's own fillers sit between the parts
            items += pending[0] + [nop(1)] * 0 + pending[1] + pending[2]
        placed = spread(items, 20.0, OPT["epi_end"])
        tim += placed
        for lo, hi, st in blocks:                 # direct stores of a block: behind the top barrier of the next stage
            g_end = placed[hi - 1][0]
            g_st = 32 * (int(g_end) // 32 + 1) + 0.6
            tim += at(st, g_st)
    seq = [I(f"{label}%=:", "label")] + fix_m0(interleave(mf, tim))
    return seq


def tail(sx, label):
    """behind a workgroup's last tile: its last two k-steps, then the exposed epilogue of set sx"""
    ns = OPT["nsets"]
    seq = [I(f"{label}%=:", "label"), Wait(vm0=True, lgkm0=True)]
    seq += [table_read(TT + 4, 16, ("tabC",)), Wait(lgkm0=True), rfl(RC + 0, TT + 5), rfl(RC + 1, TT + 6), rfl(RC + 2, TT + 7)]
    for K in (4 * NS - 2, 4 * NS - 1):
        seq += [mfma(sx, i, j, K % ns, False) for i in range(4) for j in range(2)]
    seq += [nop(16), nop(16)]
    if not OPT["noepi"]:
        seq.append(salu(f"s_mov_b32 {sreg(SROW)}, 0"))
        k = 0
        was = OPT["scalar"]
        OPT["scalar"] = False            # no MFMAs to hide under: packed fp32 arithmetic halves the instruction count here
        for i in range(4):
            for j in range(2):
                it, st = epi_block(sx, i, j, k & 1)
                seq += it + st
                k += 1
            if OPT["bounce"]:
                seq += [x for part in bounce(i) for x in part]
        OPT["scalar"] = was
    seq += [Wait(vm0=True), I("s_branch L_end%=", "branch")]
    return seq


def prologue():
    L = [nop(5)]
    # GELU constants
    qs = [GELU_Q[8], GELU_Q[7], GELU_Q[6], GELU_Q[5], GELU_Q[4], GELU_Q[3], GELU_Q[2], GELU_Q[1], GELU_Q[0], -GELU_CLAMP]
    for k, c in enumerate(qs):
        L.append(salu(f"s_mov_b32 {sreg(GQ + k)}, 0x{f32_bits(c):08x}"))
    L.append(valu(f"v_mov_b32_e32 {v(CLAMPV)}, 0x{f32_bits(GELU_CLAMP):08x}", wr=[f"v{CLAMPV}"]))
    L.append(valu(f"v_mov_b32_e32 {v(Q7V)}, 0x{f32_bits(GELU_Q[7]):08x}", wr=[f"v{Q7V}"]))
    L += [salu(f"s_mov_b32 {sreg(EXLO)}, 0x0000ffff"), salu(f"s_mov_b32 {sreg(EXLO + 1)}, 0x0000ffff"),
          salu(f"s_mov_b32 {sreg(EXHI)}, 0xffff0000"), salu(f"s_mov_b32 {sreg(EXHI + 1)}, 0xffff0000")]
    for k in range(4):
        L.append(valu(f"v_xor_b32_e32 {v(WA + k)}, {k << 5}, %[wa0]", wr=[f"v{WA + k}"]))
    L += [valu(f"v_mov_b32_e32 {v(RDA)}, %[rda]", wr=[f"v{RDA}"]), valu(f"v_mov_b32_e32 {v(SVO)}, %[svo]", wr=[f"v{SVO}"])]
    for r0 in (RA, RW, RC, RR):
        L.append(salu(f"s_mov_b32 {sreg(r0 + 3)}, 0x00020000"))
    L.append(salu(f"s_mov_b32 {sreg(RW + 2)}, 0x7fffffff"))
    L += [salu(f"s_mov_b32 {sreg(ST)}, %[ntl]"), salu(f"s_mov_b32 {sreg(SK)}, 0")]
    L.append(valu(f"v_mov_b32_e32 {v(VTAB)}, %[tab]", wr=[f"v{VTAB}"]))
    # LDS-DMA lane offsets: piece i = piece (i & 1) + (i >> 1) * 16 rows
    L += [valu(f"v_mov_b32_e32 {v(DX + 0)}, %[xo0]", wr=[f"v{DX}"]), valu(f"v_mov_b32_e32 {v(DX + 1)}, %[xo1]", wr=[f"v{DX + 1}"]),
          valu(f"v_mov_b32_e32 {v(DW + 0)}, %[wo0]", wr=[f"v{DW}"]), valu(f"v_mov_b32_e32 {v(DW + 1)}, %[wo1]", wr=[f"v{DW + 1}"])]
    for i in range(2, 8):
        L.append(valu(f"v_add_u32_e32 {v(DX + i)}, %[xs16], {v(DX + i - 2)}", rd=[f"v{DX + i - 2}"], wr=[f"v{DX + i}"]))
    for i in range(2, 4):
        L.append(valu(f"v_add_u32_e32 {v(DW + i)}, %[ws16], {v(DW + i - 2)}", rd=[f"v{DW + i - 2}"], wr=[f"v{DW + i}"]))
    # fragment addresses
    for ks in range(4):
        L.append(valu(f"v_xor_b32_e32 {v(FA[0] + ks)}, {ks << 5}, %[fb]", wr=[f"v{FA[0] + ks}"]))
    for ks in range(4):
        L.append(valu(f"v_add_u32_e32 {v(FW[0] + ks)}, %[wno], {v(FA[0] + ks)}", rd=[f"v{FA[0] + ks}"], wr=[f"v{FW[0] + ks}"]))
    for ks in range(4):
        L.append(valu(f"v_add_u32_e32 {v(FA[0] + ks)}, %[wmo], {v(FA[0] + ks)}", rd=[f"v{FA[0] + ks}"], wr=[f"v{FA[0] + ks}"]))
    for ks in range(4):
        L.append(valu(f"v_add_u32_e32 {v(FA[1] + ks)}, 0x{2 * STAGE:x}, {v(FA[0] + ks)}", rd=[f"v{FA[0] + ks}"], wr=[f"v{FA[1] + ks}"]))
        L.append(valu(f"v_add_u32_e32 {v(FW[1] + ks)}, 0x{2 * STAGE:x}, {v(FW[0] + ks)}", rd=[f"v{FW[0] + ks}"], wr=[f"v{FW[1] + ks}"]))
    assert OPT["bounce"], "direct stores were the first form of the stream (profiles/r05_gemm_hp.md); the kernel passes the bounce's addresses"
    if OPT["trace"]:
        L += [salu(f"s_mov_b32 {sreg(STMP + k)}, 0") for k in range(2, 12)]
    # the table is complete (the C++ side wrote it): first tile's resources = entry 1
    L += [I("s_waitcnt vmcnt(0) lgkmcnt(0)", "wait"), I("s_barrier", "barrier")]
    L += [I(f"ds_read_b128 {v(TT, 4)}, {v(VTAB)} offset:{ENTRY}", "ds"), I(f"ds_read_b128 {v(TT + 4, 4)}, {v(VTAB)} offset:{ENTRY + 16}", "ds"),
          I("s_waitcnt lgkmcnt(0)", "wait")]
    L += [rfl(RA + 0, TT + 0), rfl(RA + 1, TT + 1), rfl(RA + 2, TT + 2), rfl(RW + 0, TT + 3), rfl(RW + 1, TT + 4)]
    # null previous tile: C words of entry 0 (record count 0), row statistics of entry 0 (record count 0), n_off 0
    L += [I(f"ds_read_b128 {v(TT, 4)}, {v(VTAB)} offset:32", "ds"), I("s_waitcnt lgkmcnt(0)", "wait")]
    L += [rfl(NOFF, TT + 0), rfl(RR + 0, TT + 1), rfl(RR + 1, TT + 2), rfl(RR + 2, TT + 3), nop(6)]
    # the ring as a tile body leaves it: [stage 0 pieces] [epilogue loads] [stage 1 pieces]
    L += dma_stage(0, 0, ("piece", 0), first=True)
    L += e_loads(("eload",))
    L += dma_stage(1, 1, ("piece", 1))
    return fix_m0(L)


# ------------------------------------------------------------------------------------------------ waits, hazards
def resolve_waits(seq, what):
    """fills in the Wait placeholders of a linear sequence.  Loads return in order among themselves (LDS operations too): a wait
    for tag T may leave as many operations in flight as were issued behind the LAST operation tagged T.  Stores count in vmcnt in
    hardware but not here -- the bound stays valid (an incomplete load keeps every younger load incomplete), it only waits longer."""
    vmq, lgq = [], []
    for ins in seq:
        q = getattr(ins, "queue", None)
        if q == "vm":
            vmq.append(ins.tag)
        elif q == "lgkm":
            lgq += [ins.tag] * getattr(ins, "lds_ops", 1)
        if isinstance(ins, Wait):
            parts = []
            if ins.vm0:
                parts.append("vmcnt(0)")
                vmq = []
            elif ins.vm is not None:
                idx = [k for k, t in enumerate(vmq) if t == ins.vm]
                if not idx:               # cold start of the steady-state resolution only: nothing to wait for
                    if what != "steady":
                        raise SystemExit(f"{what}: wait for {ins.vm} with nothing in flight")
                    n = min(len(vmq), 63)
                else:
                    n = len(vmq) - 1 - idx[-1]
                assert n <= 63, (what, ins.vm, n)
                parts.append(f"vmcnt({n})")
                vmq = vmq[len(vmq) - n:] if n else []
                ins.n_vm = n
            if ins.lgkm0:
                parts.append("lgkmcnt(0)")
                lgq = []
            elif ins.lgkm is not None:
                idx = [k for k, t in enumerate(lgq) if t == ins.lgkm]
                n = (len(lgq) - 1 - idx[-1]) if idx else min(len(lgq), 15)
                assert n <= 15, (what, ins.lgkm, n)
                parts.append(f"lgkmcnt({n})")
                lgq = lgq[len(lgq) - n:] if n else []
                ins.n_lgkm = n
            ins.text = "s_waitcnt " + " ".join(parts)


def check(seq, what):
    """hazard distances the assembler does not insert inside an asm statement (wait states: every instruction 1, s_nop N = N + 1)"""
    last_mfma_wr, last_valu_wr, last_sgpr_wr, last_perm_wr = {}, {}, {}, {}
    pos = 0
    for ins in seq:
        if ins.kind in ("valu", "perm"):
            for r in ins.rd | ins.wr:
                if r in last_mfma_wr and pos - last_mfma_wr[r] < 13:
                    raise SystemExit(f"{what}: MFMA result {r} touched by VALU after {pos - last_mfma_wr[r]} states: {ins.text}")
        if ins.kind == "perm":
            for r in ins.rd:
                if r in last_valu_wr and pos - last_valu_wr[r] < 2:
                    raise SystemExit(f"{what}: permlane reads {r} {pos - last_valu_wr[r]} states after its VALU write: {ins.text}")
        if ins.kind == "mfma":
            for r in ins.rd:
                if r in last_valu_wr and pos - last_valu_wr[r] < 3:
                    raise SystemExit(f"{what}: VALU result {r} read by MFMA after {pos - last_valu_wr[r]} states: {ins.text}")
        if ins.kind == "vmem":
            for r in ins.rd:
                if r in last_perm_wr and pos - last_perm_wr[r] < 2:
                    raise SystemExit(f"{what}: store reads {r} {pos - last_perm_wr[r]} states after permlane: {ins.text}")
            for s_ in range(40, 70):
                if f"s{s_}" in last_sgpr_wr and pos - last_sgpr_wr[f"s{s_}"] < 5:
                    lo = [x for x in (RA, RW, RC, RR) if x <= s_ < x + 4]
                    used = (lo and f"s[{lo[0]}:{lo[0] + 3}]" in ins.text) or f" s{s_} " in ins.text + " "
                    if used:
                        raise SystemExit(f"{what}: VMEM uses s{s_} {pos - last_sgpr_wr[f's{s_}']} states after v_readfirstlane: {ins.text}")
        if ins.kind == "mfma":
            for r in ins.wr:
                last_mfma_wr[r] = pos
        if ins.kind in ("valu", "perm"):
            for r in ins.wr:
                if r.startswith("s"):
                    last_sgpr_wr[r] = pos
                else:
                    last_valu_wr[r] = pos
                    last_mfma_wr.pop(r, None)
                    if ins.kind == "perm":
                        last_perm_wr[r] = pos
                    else:
                        last_perm_wr.pop(r, None)
        pos += ins.states


def check_stores(seq, what):
    """a block's packed registers are not rewritten before they were consumed (direct store, or the bounce's masked scratch writes of BOTH
    halves), consumers follow the swaps; bounced rows are stored behind a wait before the next read-back overwrites them"""
    pend = {}          # packed register -> number of consumers still to come
    rb_pend = set()
    need = 2 if OPT["bounce"] else 1
    for ins in seq:
        is_store = ins.kind == "vmem" and ins.text.startswith("buffer_store")
        is_bw = ins.kind == "ds" and "ds_write_b128" in ins.text
        if is_store or is_bw:
            for r in ins.rd:
                if not r.startswith("v"):
                    continue
                n = int(r[1:])
                if any(P[k] <= n < P[k] + 8 for k in range(2)):
                    if r not in pend:
                        raise SystemExit(f"{what}: {r} consumed without a fresh value: {ins.text[:60]}")
                    pend[r] -= 1
                    if pend[r] == 0:
                        del pend[r]
                if is_store and RB <= n < RB + 8 and OPT["bounce"]:
                    if r not in rb_pend:
                        raise SystemExit(f"{what}: store of {r} without a bounced row")
                    rb_pend.discard(r)
        elif ins.text.startswith("v_cvt_pk_bf16"):
            for r in ins.wr:
                if r in pend:
                    raise SystemExit(f"{what}: {r} rewritten before it was consumed: {ins.text}")
        elif ins.kind == "perm":
            for r in ins.wr:
                pend[r] = need
        elif ins.kind == "ds" and ins.text.startswith("ds_read_b128") and OPT["bounce"]:
            for r in ins.wr:
                n = int(r[1:])
                if RB <= n < RB + 8:
                    if r in rb_pend and not OPT["nostore"]:
                        raise SystemExit(f"{what}: {r} overwritten before its store: {ins.text}")
                    if "offset" in ins.text and f"{v(RDA)} " in ins.text + " ":
                        rb_pend.add(r)
    if pend and not OPT["nostore"]:
        raise SystemExit(f"{what}: values never consumed: {sorted(pend)}")


def check_eloads(seq, what):
    """the loads of the NEXT epilogue's bias / colsum / row statistics come behind the last instruction of this epilogue that reads them"""
    first_load = {}
    last_read = {}
    watched = set(vr(BIAS, 32) + vr(CS, 32) + vr(RS, 8))
    for pos, ins in enumerate(seq[:len(seq) // 2]):
        if ins.kind == "vmem" and getattr(ins, "tag", None) == ("eload",):
            for r in ins.wr:
                first_load.setdefault(r, pos)
        elif ins.kind in ("valu", "perm"):
            for r in ins.rd & watched:
                last_read[r] = pos
    for r, p in first_load.items():
        if r in last_read and last_read[r] > p:
            raise SystemExit(f"{what}: {r} reloaded at {p} before its last use at {last_read[r]}")


def program():
    P_ = prologue()
    # steady-state waits: resolve over A B A B (four separate instances), emit the second pair
    warm = [body(0, "L_warm_A"), body(1, "L_warm_B")]
    bodies = [body(0, "L_body_A"), body(1, "L_body_B")]
    resolve_waits(warm[0] + warm[1] + bodies[0] + bodies[1], "steady")
    again = [body(0, "L_x"), body(1, "L_y")]
    resolve_waits(warm[1] + again[0] + again[1], "steady")
    for k in range(2):
        if [w.text for w in bodies[k] if isinstance(w, Wait)] != [w.text for w in again[k] if isinstance(w, Wait)]:
            raise SystemExit("waits of a tile body depend on the entry path")
    # from the prologue the same counts must come out (the prologue leaves the queue a tile body leaves)
    trial = body(0, "L_t")
    resolve_waits(prologue() + trial, "prologue")
    for w0, w1 in zip([w for w in bodies[0] if isinstance(w, Wait)], [w for w in trial if isinstance(w, Wait)]):
        if w0.text != w1.text:
            raise SystemExit(f"prologue leaves another queue than a tile body: {w0.text} vs {w1.text}")
    check(bodies[0] + bodies[1] + bodies[0], "loop")
    if not OPT["noepi"] and not OPT["nostore"]:
        check_stores(bodies[0] + bodies[1] + bodies[0], "loop")
    check_eloads(bodies[0] + bodies[1], "loop")
    tails = [tail(0, "L_tail_A"), tail(1, "L_tail_B")]
    for t in tails:
        resolve_waits(t, "tail")
        check(t, "tail")
    out = list(P_)
    first = None
    if OPT["first"] and not OPT["noepi"]:
        # a workgroup's first tile: set A without an epilogue (the null tile's would cost its issue slots: ~2.5 us of a 4-tile launch)
        first = body(0, "L_first", epilogue=False)
        resolve_waits(prologue() + first, "prologue")
        nxt = body(1, "L_z")
        resolve_waits(prologue() + body(0, "L_first2", epilogue=False) + nxt, "prologue")
        if [w.text for w in bodies[1] if isinstance(w, Wait)] != [w.text for w in nxt if isinstance(w, Wait)]:
            raise SystemExit("waits of body B depend on whether the first tile ran before it")
        check(first + bodies[1], "first")
        out += first
        out += [salu(f"s_sub_u32 {sreg(ST)}, {sreg(ST)}, 1"), salu(f"s_cmp_eq_u32 {sreg(ST)}, 0"), I("s_cbranch_scc1 L_tail_A%=", "branch"),
                I("s_branch L_body_B%=", "branch")]
    else:
        out.append(I("s_branch L_body_A%=", "branch"))
    for k in range(2):
        out += bodies[k]
        out += [salu(f"s_sub_u32 {sreg(ST)}, {sreg(ST)}, 1"), salu(f"s_cmp_eq_u32 {sreg(ST)}, 0"),
                I(f"s_cbranch_scc1 L_tail_{'AB'[k]}%=", "branch")]
    out.append(I("s_branch L_body_A%=", "branch"))
    out += tails[0] + tails[1]
    out.append(I("L_end%=:", "label"))
    if OPT["trace"]:          # dbg0 = cycles in top waits + barriers, dbg1 = first stamp, dbg2 = last stamp (in front of the last top wait)
        out += [valu(f"v_mov_b32_e32 %[dbg0], {sreg(STMP + 7)}"), valu(f"v_mov_b32_e32 %[dbg1], {sreg(STMP + 10)}"),
                valu(f"v_mov_b32_e32 %[dbg2], {sreg(STMP + 8)}"), valu(f"v_mov_b32_e32 %[dbg3], {sreg(STMP + 4)}")]
    return out


LAB_DIR = os.path.join(HERE, "..", "..", "tools", "lab_src")      # --lab: the ablation streams are lab material, not product


def write(name, Pg, where=HERE):
    out = os.path.normpath(os.path.join(where, name))
    n_ins = sum(1 for i in Pg if i.kind != "label")
    with open(out, "w") as f:
        f.write("// GENERATED by gen_gemm_hp.py -- do not edit; an instruction stream of gemm_hp.hip's asm statement.\n")
        f.write(f"// {n_ins} instructions.  Register map and schedule: see the generator's docstring.\n")
        for ins in Pg:
            if ins.text.endswith("@ST"):          # the store's cache flag is the includer's macro HP_ST_FLAG ("" or " nt")
                f.write('"' + ins.text[:-3] + '" HP_ST_FLAG "\\n"\n')
            else:
                f.write('"' + ins.text + '\\n"\n')
    mf = sum(1 for i in Pg if i.kind == "mfma")
    print(f"{out}: {n_ins} instructions, {mf} MFMAs")


VARIANTS = {"gemm_hp_gelu.inc": dict(fold=False, gelu=True), "gemm_hp_fold_gelu.inc": dict(fold=True, gelu=True),
            "gemm_hp_plain.inc": dict(fold=False, gelu=False), "gemm_hp_fold.inc": dict(fold=True, gelu=False)}
# Every stream is included twice by gemm_hp.hip: HP_ST_FLAG "" and " nt".  Outputs larger than the last-level cache keeps (ep.stream)
# leave through nontemporal stores -- plain ones push the operands out of the L2s (M = 65536, N = 3072: 374 -> 278 us; at M = 10368 nt
# costs 4 us: profiles/r05_gemm_hp.md)


def main():
    for name, o in VARIANTS.items():
        OPT.update(o)
        OPT["stflag"] = "@ST"
        write(name, program())
    with open(os.path.join(HERE, "gemm_hp_clobbers.inc"), "w") as f:
        f.write("// GENERATED by gen_gemm_hp.py: registers the asm statement of gemm_hp_kernel owns.\n")
        names = [f"v{i}" for i in range(16, 256)] + [f"a{i}" for i in range(0, 256)] + [f"s{i}" for i in range(40, 88)]
        f.write(", ".join(f'"{n}"' for n in names) + "\n")
    if "--lab" in sys.argv:
        OPT.update(fold=False, gelu=True, stflag="")
        labs = [("noepi",), ("nostore",), ("trace",), ("noepi", "nodma"), ("noepi", "nofrag"), ("noepi", "nobar"), ("noepi", "nodma", "nofrag", "nobar"),
                ("packed",), ("packed", "nostore"), ("nt",), ("sc1",)]
        for k, keys in enumerate(labs):
            for key in keys:
                if key == "packed":
                    OPT["scalar"] = False
                elif key in ("nt", "sc1"):
                    OPT["stflag"] = " " + key
                else:
                    OPT[key] = True
            write(f"gemm_hp_lab{k + 1}.inc", program(), LAB_DIR)
            for key in keys:
                if key == "packed":
                    OPT["scalar"] = True
                elif key in ("nt", "sc1"):
                    OPT["stflag"] = ""
                else:
                    OPT[key] = False


if __name__ == "__main__":
    main()
