#!/usr/bin/env python
"""Generates the head_dim-128 main loop of the bf16 / fp16 QK-Skip forward on the OTHER matrix shape of gfx950:
v_mfma_f32_16x16x32_{bf16,f16} instead of the 32x32x16 form gen_fwd_x64.py uses.

Why: on random operands the 16x16x32 form costs ~8 % less energy per FLOP (a 4-register accumulator is read-modified-written per 16 Ki
FLOP, against 16 registers per 32 Ki; profiles/r04_power_ceiling.md), and this kernel sits at the socket's power cap, where energy per
FLOP is throughput. What it costs: twice the MFMA instructions (64 + 64 per step) for a lone wave that already issues ~5.7 fillers per
32-cycle MFMA gap. Round 4 only priced it with stand-ins (gen_fwd_x64.py `mfma16:*`, wrong results); this is the real body: a new
register map and cross-lane scheme, the same shell, tile (256 x 64), LDS-DMA staging, tile-address table, vote bit and list protocol.

Same structure as gen_fwd_x64.py (one wave per SIMD, 64 query rows per wave, one barrier per step, two S buffers software-pipelined
over two phases); what differs is where a query row lives:

  S^T = K Q^T per (q-block qb of 16 queries, key block kb of 16 keys):  A = K fragment [16 keys x 32 d] (ds_read_b128 from the same
  XOR-swizzled K image), B = Q fragment [32 d x 16 queries] (AGPRs), result 4 registers: lane (j = lane & 15, g = lane >> 4) holds
  keys 16 kb + 4 g + r (r = 0..3) of query 16 qb + j. A query row is spread over the FOUR lanes j, j + 16, j + 32, j + 48, and a
  lane carries FOUR queries (one per q-block), 16 scores of each.

  O^T += V^T P^T per (qb, d-block db of 16, key step kk of 32):  B = P straight from the S^T accumulators - the 8 k-slots of lane
  group g are the keys 32 kk + 4 g + r (from S(qb, 2 kk)) and 32 kk + 16 + 4 g + r (from S(qb, 2 kk + 1)), packed in place into the
  first 4 registers of every 8 - and A = V^T fragment by two ds_read_b64_tr_b16 whose 16 lanes name exactly those rows. The V image
  needs its own swizzle for that: lanes 0-31 of a transpose read touch 8 rows x 32 bytes, so the XOR acts on 32-byte granules with
  row & 7 (the 32x32 form: 64-byte segments with row & 3). The shell's first DMA and the per-lane DMA offsets here apply it.

  Row statistics: the in-lane max of a lane's 16 scores per q-block (4 values), then a TRANSPOSING reduction over the four lanes of
  a row - v_permlane16_swap on (qb 0, qb 1) and on (qb 2, qb 3), max; v_permlane32_swap on the two results, max: three swaps, three
  max - which leaves lane (j, g) with the COMPLETE row max of query 16 g + j. The running state (true max, vote, lazy-rescale test)
  lives in that transposed form: ONE register per lane for 64 queries (the 32x32 form: two per lane). -m_ref c for the exponent and
  the row-sum accumulators stay per (lane, q-block); the broadcast back happens only in the rare rescale block and in the epilogue.

tools/debug/probe_m16_layout.hip checks all of this index math (fragment layouts, the V swizzle, the transposing reduction, the store
swap) on the GPU against a CPU product before any assembly is involved.

Register file: AGPR a[0:127] O^T (4 q-blocks x 8 d-blocks x 4), a[128:191] Q (4 q-blocks x 4 k-steps x 4), a[192:255] K fragments
of the next tile (4 key blocks x 4 k-steps x 4); VGPR v[0:63] / v[64:127] S^T ping / pong (base 16 qb + 4 kb), v[128:159] V^T ring
(8 x 4), then addresses, state, temporaries.
"""
import os
import sys

OPT = set(x for x in os.environ.get("LA_X64_OPT", "").split(",") if x)


def opt_val(key, default):
    for o in OPT:
        if o.startswith(key + ":"):
            return o[len(key) + 1:]
    return default


SCHEDULE_ONLY = {"x", "dmagaps", "dmapol", "align", "pad4", "safe", "kgaps", "vgap0"}      # see gen_fwd_x64.py: same results bit for bit


def option_tag():
    wrong = sorted(o for o in OPT if o.split(":")[0] not in SCHEDULE_ONLY)
    return (f"// la_body_options: {','.join(sorted(OPT | {'m16'}))}; wrong_results={1 if wrong else 0}"
            + (f" (PRICING ONLY, results are wrong: {','.join(wrong)})" if wrong else ""))


DTYPE = os.environ.get("LA_X64_DTYPE", "bf16")
MFMA_OP = {"bf16": "v_mfma_f32_16x16x32_bf16", "f16": "v_mfma_f32_16x16x32_f16"}[DTYPE]
CVT_OP = {"bf16": "v_cvt_pk_bf16_f32", "f16": "v_cvt_pk_f16_f32"}[DTYPE]
D = int(os.environ.get("LA_X64_D", "128"))
assert D == 128, "the 16x16x32 body exists at head_dim 128"
NQB, NKB, KS, DB, KK = 4, 4, D // 32, D // 16, 2       # q-blocks / key blocks of 16, k-steps of 32 d, d-blocks of 16, key steps of 32
NG = NQB * NKB * KS                                     # 64 MFMAs per phase
assert NG == NQB * DB * KK
ROW, ROW_SHIFT = 256, 8
KV_TILE = 64 * ROW
V_REGION = 2 * KV_TILE
NQUADS = NQB * NKB                                      # softmax units: the 4 scores of one (q-block, key block) accumulator
XQ = int(opt_val("x", "5"))                             # quads of the NEXT tile done in phase 2 (of 16), the rest in phase 1
SAFE_GAPS = int(opt_val("safe", "4"))                   # MFMA gaps at the head of phase 2 that hold nothing that reads S_nxt
DMA_GAPS = [int(x) for x in opt_val("dmagaps", "2.4.8.12.16.20.22.26.30.34").replace(".", ",").split(",")]
DMA_POLICY = {"": "", "nt": " nt", "sc0": " sc0", "sc1": " sc1"}[opt_val("dmapol", "")]
DMA_BIAS = 3072

# ---------------------------------------------------------------- register map
Q_A0, K_A0 = 128, 192


def O_(qb, db):
    return 32 * qb + 4 * db


def QA(qb, ks):
    return Q_A0 + 16 * qb + 4 * ks


def KFRAG(kb, ks):
    j = 4 * kb + ks
    return f"a[{K_A0 + 4 * j}:{K_A0 + 4 * j + 3}]"


def S_(sset, qb, kb):
    return 64 * sset + 16 * qb + 4 * kb


VF = [128 + 4 * i for i in range(8)]
KADDR = list(range(160, 164))             # per k-step
VADDR = list(range(164, 172))             # per d-block
LK = list(range(172, 176))
LV = list(range(176, 180))
NMS = list(range(180, 184))               # -m_ref c of the lane's query in q-block qb
L = list(range(184, 188))                 # row-sum accumulator of the lane's 16 keys per tile, per q-block
M4 = list(range(188, 192))                # in-lane tile max per q-block
MTRUE_T, MREF_T, MTHR_T = 192, 193, 194   # transposed state: lane (j, g) = query 16 g + j
ALPHA = list(range(195, 199))             # rescale factor per q-block (rare block -> O rescale at the step's tail)
T = list(range(200, 216))                 # temporaries (T[0] even: 64-bit tuples such as T[4:5] must be even-aligned)
NEGINF, G4, QROW_T, TABV, LANE, J16, RAGK, RAGV = 199, 216, 217, 218, 219, 220, 221, 222

S_KBASE, S_VBASE, S_QBASE = 36, 38, 40    # 64-bit
S_TB, S_VB, S_EXEC, S_T64, S_T64B = 42, 44, 46, 48, 50
(S_KRS, S_VRS, S_LASTROW, S_NTILES, S_C, S_THR, S_TAILVALID, S_FIRSTLAST, S_TAB, S_DOFLAGS, S_WAVE, S_I, S_DOMASK,
 S_FREE0, S_FREE1, S_FREE2, S_LDS, S_T0, S_T1, S_T2, S_T3, S_NM1, S_QRS, S_QROW0, S_SEQLENQ, S_EXPORT, S_PARAM, S_DOWORD, S_NEGC,
 S_FREE3, S_DMAW, S_FREE4, S_TAU, S_RESC, S_FREE5) = range(52, 87)
S_FREE6, S_TB2, S_VB2, S_BIT = 87, 88, 90, 92
TBS, VBS = [S_TB, S_TB2], [S_VB, S_VB2]

out = []          # IR: str | ("LDS", str, tag) | ("WAIT", tag) | ("DRAIN",)


def emit(x):
    out.append(x if isinstance(x, tuple) else "    " + x)


def label(s_):
    out.append(s_ + ":")


def v(i):
    return f"v{i}"


def vr(a, n):
    return f"v[{a}:{a + n - 1}]"


def ar(a, n):
    return f"a[{a}:{a + n - 1}]"


def s(i):
    return f"s{i}"


def sr(a, n=2):
    return f"s[{a}:{a + n - 1}]"


uid = [0]


def new_label(prefix):
    uid[0] += 1
    return f".LM{prefix}_{uid[0]}_%="


def finalize(items):
    """Counted lgkmcnt waits: LDS operations of one wave return in order."""
    lines, q = [], []
    for it in items:
        if isinstance(it, str):
            lines.append(it)
        elif it[0] == "LDS":
            lines.append("    " + it[1])
            q.append(it[2])
        elif it[0] == "WAIT":
            if it[1] in q:
                idx = max(i for i, t in enumerate(q) if t == it[1])
                lines.append(f"    s_waitcnt lgkmcnt({min(len(q) - 1 - idx, 15)})")
                q = q[idx + 1:]
        elif it[0] == "DRAIN":
            lines.append("    s_waitcnt vmcnt(0) lgkmcnt(0)")
            q = []
    return lines


# ---------------------------------------------------------------- building blocks
def k_read(kbuf_imm, kb, ks):
    """K fragment (kb, ks): key 16 kb + j, 16-byte chunk 4 ks + g, XOR-swizzled by the row (row & 15 = j): KADDR[ks] + 4 KiB per key block."""
    return ("LDS", f"ds_read_b128 {KFRAG(kb, ks)}, {v(KADDR[ks])} offset:{kbuf_imm + kb * 16 * ROW}", ("k", kb, ks))


def v_read(slot, vbuf_imm, kk, db):
    """V^T fragment (kk, db): d = 16 db + j, k-slots = keys 32 kk + 4 g + e and 32 kk + 16 + 4 g + e: two transpose reads 16 rows apart."""
    off = vbuf_imm + kk * 32 * ROW
    return [("LDS", f"ds_read_b64_tr_b16 {vr(VF[slot], 2)}, {v(VADDR[db])} offset:{V_REGION + off}", ("v", kk, db, 0)),
            ("LDS", f"ds_read_b64_tr_b16 {vr(VF[slot] + 2, 2)}, {v(VADDR[db])} offset:{V_REGION + off + 16 * ROW}", ("v", kk, db, 1))]


def mfma_qk(sset, kb, ks, qb):
    d = S_(sset, qb, kb)
    c = "0" if ks == 0 else vr(d, 4)
    return f"    {MFMA_OP} {vr(d, 4)}, {KFRAG(kb, ks)}, {ar(QA(qb, ks), 4)}, {c}"


def mfma_pv(sset, slot, kk, db, qb):
    return f"    {MFMA_OP} {ar(O_(qb, db), 4)}, {vr(VF[slot], 4)}, {vr(S_(sset, qb, 2 * kk), 4)}, {ar(O_(qb, db), 4)}"


# order of the phase-1 MFMAs: k-step outer (an accumulator is touched every 16 MFMAs), key block, q-block inner (a K fragment feeds
# 4 consecutive MFMAs: its operand bus does not toggle between them)
QK_ORDER = [(ks, kb, qb) for ks in range(KS) for kb in range(NKB) for qb in range(NQB)]
# phase 2: key step outer (the second half of P is needed 32 MFMAs later), d-block, q-block inner
PV_ORDER = [(kk, db, qb) for kk in range(KK) for db in range(DB) for qb in range(NQB)]
QUAD_ORDER = [(qb, kb) for kb in range(NKB) for qb in range(NQB)]       # softmax units: key block major (P of key step 0 first)


def quad_parts(sset, n):
    """Softmax of quad n = (qb, kb): P = exp2(S c - m_ref c) in place (4 fma + 4 exp), row sum (4 add), pack into the B operand of
    the PV MFMA (2 cvt): the 8 registers [S(qb, 2 kk) | S(qb, 2 kk + 1)] end as 4 packed registers at their start."""
    qb, kb = QUAD_ORDER[n]
    r = [S_(sset, qb, kb) + i for i in range(4)]
    dst = S_(sset, qb, kb & ~1) + 2 * (kb & 1)
    F = [f"    v_fma_f32 {v(x)}, {v(x)}, {s(S_C)}, {v(NMS[qb])}" for x in r]
    E = [f"    v_exp_f32 {v(x)}, {v(x)}" for x in r]
    A = [f"    v_add_f32 {v(L[qb])}, {v(L[qb])}, {v(x)}" for x in r]
    C = [f"    {CVT_OP} {v(dst)}, {v(r[0])}, {v(r[1])}", f"    {CVT_OP} {v(dst + 1)}, {v(r[2])}, {v(r[3])}"]
    return F, E, A, C


def softmax_stream(sset, quads):
    """Software-pipelined softmax of several quads: a transcendental occupies its unit for ~4 quad-cycles but two issue slots, so no
    two exps are adjacent: exp i of quad q is followed by fma i of quad q + 1 and add i of quad q - 1; the packs of quad q - 1 close
    the group (after its adds, which read the un-packed values; and after quad q - 1's predecessor of the same q-block was packed:
    the quads come key block major, so (qb, kb - 1) is 4 quads back)."""
    if "nosoftmax" in OPT or not quads:
        return []
    parts = [quad_parts(sset, n) for n in quads]
    o = list(parts[0][0])
    n = len(parts)
    for g in range(n):
        Fn = parts[g + 1][0] if g + 1 < n else [None] * 4
        Ap, Cp = (parts[g - 1][2], parts[g - 1][3]) if g > 0 else ([None] * 4, [])
        for i in range(4):
            o.append(parts[g][1][i])
            if Fn[i]:
                o.append(Fn[i])
            if Ap[i]:
                o.append(Ap[i])
        o += Cp
    o += parts[-1][2] + parts[-1][3]
    return o


def row_max_ops(sset):
    """In-lane max of the 16 scores of each q-block into M4[qb] (one max3 chain each), interleaved over the four q-blocks."""
    per = []
    for qb in range(NQB):
        regs = [S_(sset, qb, kb) + r for kb in range(NKB) for r in range(4)]
        ops = [f"    v_max3_f32 {v(M4[qb])}, {v(regs[0])}, {v(regs[1])}, {v(regs[2])}"]
        rest = regs[3:]
        while len(rest) >= 2:
            ops.append(f"    v_max3_f32 {v(M4[qb])}, {v(M4[qb])}, {v(rest[0])}, {v(rest[1])}")
            rest = rest[2:]
        if rest:
            ops.append(f"    v_max_f32 {v(M4[qb])}, {v(M4[qb])}, {v(rest[0])}")
        per.append(ops)
    return [x for grp in zip(*per) for x in grp]


def transposing_reduce(x, op, dst):
    """x[0..3]: per-lane partials of the lane's query in q-block 0..3 (DESTROYED). dst: lane (j, g) = op over the four lanes of query
    16 g + j. v_permlane16_swap a, b swaps a's odd 16-lane rows with b's even rows; v_permlane32_swap the same on 32-lane halves. A
    VALU write needs 2 wait states before a permlane swap reads it, and the swap's results 1 before a VALU reads them: the
    independent instructions between are arranged for that (`filler` ops are supplied by the caller where there are none)."""
    return [f"    v_permlane16_swap_b32 {v(x[0])}, {v(x[1])}",
            f"    v_permlane16_swap_b32 {v(x[2])}, {v(x[3])}",
            f"    {op} {v(x[0])}, {v(x[0])}, {v(x[1])}",
            f"    {op} {v(x[2])}, {v(x[2])}, {v(x[3])}",
            "    s_nop 1",
            f"    v_permlane32_swap_b32 {v(x[0])}, {v(x[2])}",
            "    s_nop 0",
            f"    {op} {v(dst)}, {v(x[0])}, {v(x[2])}"]


def broadcast_ops(src_t, dst4):
    """The inverse: src_t (lane (j, g) = value of query 16 g + j) -> dst4[qb] = value of the lane's query in q-block qb."""
    a, b, a2, b2 = dst4
    return [f"    v_mov_b32 {v(a)}, {v(src_t)}", f"    v_mov_b32 {v(a2)}, {v(src_t)}", "    s_nop 1",
            f"    v_permlane32_swap_b32 {v(a)}, {v(a2)}",                  # a = [R0 R1 R0 R1], a2 = [R2 R3 R2 R3]
            "    s_nop 0",
            f"    v_mov_b32 {v(b)}, {v(a)}", f"    v_mov_b32 {v(b2)}, {v(a2)}", "    s_nop 1",
            f"    v_permlane16_swap_b32 {v(a)}, {v(b)}",                   # a = R0 everywhere, b = R1
            f"    v_permlane16_swap_b32 {v(a2)}, {v(b2)}",                 # a2 = R2, b2 = R3
            "    s_nop 0"]


def stats_ops(rare_label, back_label, flush_label, flush_back, inval_label, inval_back):
    """Transposing max, skip vote (one bit per position), true running max, lazy-rescale test - all on ONE register per lane."""
    o = []
    a = o.append
    a(f"    v_add_u32 {v(TABV)}, 16, {v(TABV)}")
    o += transposing_reduce(M4, "v_max_f32", T[0])
    # the step past the end of the walk (i == n - 1: tile i + 1 does not exist) must not touch the state
    a(f"    s_cmp_eq_u32 {s(S_I)}, {s(S_NM1)}")
    a(f"    s_cbranch_scc1 {inval_label}")
    o.append(inval_back + ":")
    # vote: (m_loc - m_prev) c > thr (softmax.h:194), m_prev = the running max BEFORE this tile
    a(f"    v_sub_f32 {v(T[1])}, {v(T[0])}, {v(MTRUE_T)}")
    a(f"    v_max_f32 {v(MTRUE_T)}, {v(MTRUE_T)}, {v(T[0])}")
    a(f"    v_mul_f32 {v(T[1])}, {s(S_C)}, {v(T[1])}")
    a(f"    v_cmp_gt_f32 vcc, {v(T[1])}, {s(S_THR)}")
    a("    s_cmp_lg_u64 vcc, 0")                                          # SCC = some row of the wave voted "do"
    a(f"    s_cselect_b32 {s(S_T0)}, {s(S_BIT)}, 0")
    a(f"    s_or_b32 {s(S_DOMASK)}, {s(S_DOMASK)}, {s(S_T0)}")
    a(f"    v_cmp_gt_f32 vcc, {v(MTRUE_T)}, {v(MTHR_T)}")                 # lazy rescale: m_true > m_ref + tau / c on some row
    a(f"    s_cbranch_vccnz {rare_label}")
    o.append(back_label + ":")
    a(f"    s_lshl_b32 {s(S_BIT)}, {s(S_BIT)}, 1")
    a(f"    s_cbranch_scc0 {flush_label}")
    o.append(flush_back + ":")
    return o


def set_reference_ops():
    """m_ref := m_true for every row (transposed), and its broadcast products: NMS[qb] = -m_ref c, MTHR = m_ref + tau / c."""
    o = [f"    v_mov_b32 {v(MREF_T)}, {v(MTRUE_T)}", f"    v_add_f32 {v(MTHR_T)}, {s(S_TAU)}, {v(MTRUE_T)}",
         f"    v_mul_f32 {v(T[2])}, {s(S_NEGC)}, {v(MTRUE_T)}"]
    o += broadcast_ops(T[2], [NMS[0], NMS[1], NMS[2], NMS[3]])
    return o


def rare_rescale_block(rare_label, back_label):
    """Out of line: m_ref follows m_true; alpha = exp2((m_ref_old - m_true) c) per row, broadcast to the q-blocks; l *= alpha; O flagged."""
    label(rare_label)
    emit(f"v_sub_f32 {v(T[2])}, {v(MREF_T)}, {v(MTRUE_T)}")
    emit(f"v_mul_f32 {v(T[2])}, {s(S_C)}, {v(T[2])}")
    emit(f"v_exp_f32 {v(T[3])}, {v(T[2])}")
    emit("s_nop 1")                                            # a transcendental's result is not forwarded to the next VALU read
    for op in broadcast_ops(T[3], ALPHA):
        out.append(op)
    for op in set_reference_ops():
        out.append(op)
    for qb in range(NQB):
        emit(f"v_mul_f32 {v(L[qb])}, {v(L[qb])}, {v(ALPHA[qb])}")
    emit(f"s_mov_b32 {s(S_RESC)}, 1")
    emit(f"s_branch {back_label}")


def inval_block(lbl, back):
    """Out of line (last step of a walk): the tile max becomes -inf (no vote, no new max) and -m_ref c becomes -inf, so the part of
    P(i + 1) computed in this phase - exp2(S c - inf) = 0 - adds nothing to the row sums."""
    label(lbl)
    emit(f"v_mov_b32 {v(T[0])}, {v(NEGINF)}")
    for qb in range(NQB):
        emit(f"v_mov_b32 {v(NMS[qb])}, {v(NEGINF)}")
    emit(f"s_branch {back}")


def flush_domask():
    emit(f"v_mov_b32 {v(T[4])}, {s(S_DOWORD)}")
    emit(f"v_mov_b32 {v(T[5])}, {s(S_DOMASK)}")
    emit(f"s_mov_b64 {sr(S_EXEC)}, exec")
    emit("s_mov_b64 exec, 1")
    emit(f"ds_or_b32 {v(T[4])}, {v(T[5])}")
    emit(f"s_mov_b64 exec, {sr(S_EXEC)}")
    emit(f"s_mov_b32 {s(S_DOMASK)}, 0")


def flush_block(flush_label, back_label):
    label(flush_label)
    flush_domask()
    emit(f"s_add_u32 {s(S_DOWORD)}, {s(S_DOWORD)}, 4")
    emit(f"s_mov_b32 {s(S_BIT)}, 1")
    emit("s_waitcnt lgkmcnt(0)")
    emit(f"s_branch {back_label}")


def rescale_o_block(lbl, back):
    """Out of line (rare): O^T *= alpha of the lane's query per q-block (AGPR -> VGPR -> AGPR), after the PV MFMAs have drained."""
    label(lbl)
    emit("s_nop 15")
    emit("s_nop 15")
    for qb in range(NQB):
        for base in range(0, 4 * DB, 8):
            for k in range(8):
                emit(f"v_accvgpr_read_b32 {v(T[k])}, a{O_(qb, 0) + base + k}")
            for k in range(8):
                emit(f"v_mul_f32 {v(T[k])}, {v(T[k])}, {v(ALPHA[qb])}")
            for k in range(8):
                emit(f"v_accvgpr_write_b32 a{O_(qb, 0) + base + k}, {v(T[k])}")
    emit(f"s_mov_b32 {s(S_RESC)}, 0")
    emit("s_nop 7")
    emit(f"s_branch {back}")


def dma_ops(kbuf_imm, vbuf_imm, do_k=True, do_v=True, st=0):
    """[m0K, K0..K3, m0V, V0..V3]: one M0 per tensor, the piece index on the instruction offset (see gen_fwd_x64.py dma_ops)."""
    if "nodma" in OPT:
        return []
    o = []
    if do_k:
        o.append(f"    s_add_u32 m0, {s(S_DMAW)}, {kbuf_imm}")
        o += [f"    global_load_lds_dwordx4 {v(LK[j])}, {sr(TBS[st])} offset:{1024 * j}{DMA_POLICY}" for j in range(4)]
    if do_v:
        o.append(f"    s_add_u32 m0, {s(S_DMAW)}, {V_REGION + vbuf_imm}")
        o += [f"    global_load_lds_dwordx4 {v(LV[j])}, {sr(VBS[st])} offset:{1024 * j}{DMA_POLICY}" for j in range(4)]
    return o


def weight(it):
    if isinstance(it, str):
        if it.endswith(":"):
            return 0
        if "v_exp_f32" in it:
            return 2
    return 1


def n_fill(items):
    return sum(weight(it) for it in items)


def distribute(queue, post, start, cap=0, end=NG):
    """Append the ops of `queue` (order kept) to post[start..end-1], topping every gap up to `cap` fillers (0: balance evenly)."""
    q = list(queue)
    if cap <= 0:
        total = sum(n_fill(post[t]) for t in range(start, end)) + n_fill(q)
        cap = -(-total // (end - start))
    for t in range(start, end):
        while q and n_fill(post[t]) < cap:
            post[t].append(q.pop(0))
            while q and isinstance(q[0], str) and q[0].endswith(":"):      # a label sticks to the op before it
                post[t].append(q.pop(0))
    post[end - 1] += q


def emit_gaps(pre, mf, post):
    for t in range(NG):
        for it in pre[t] + [mf[t]] + post[t]:
            out.append(it)


deferred = []


def step(variant):
    """One pipeline step; variant = parity of i (S_cur = S set `variant`; K(i+2) / V(i) in LDS buffer `variant`)."""
    cur, nxt = variant, variant ^ 1
    kbuf_read, kbuf_stage = cur * KV_TILE, nxt * KV_TILE
    vbuf_cur, vbuf_stage = cur * KV_TILE, nxt * KV_TILE

    # ---- phase 1: S_nxt = K(i+1) Q^T || rest of softmax(i), DMA issue (V(i+1), K(i+3)), first 8 V^T fragments
    pre = [[] for _ in range(NG)]
    post = [[] for _ in range(NG)]
    mf = [mfma_qk(nxt, kb, ks, qb) if "nomfma1" not in OPT else "    s_nop 0" for ks, kb, qb in QK_ORDER]
    for g_, op in zip(DMA_GAPS, dma_ops(kbuf_stage, vbuf_stage, st=variant)):
        post[g_].append(op)
    v0 = int(opt_val("vgap0", "36"))
    for f in range(8):                                         # key step 0's fragments, spread over the back of the phase
        if "novread" not in OPT:
            post[v0 + f * ((NG - v0) // 8)] += v_read(f, vbuf_cur, 0, f)
    distribute(softmax_stream(cur, list(range(XQ, NQUADS))), post, 0)
    emit_gaps(pre, mf, post)

    # ---- phase 2: O^T += V(i)^T P(i)^T || K(i+2) fragments -> AGPRs, key step 1's V^T fragments, next step's DMA bases,
    #               statistics of tile i+1, first quads of softmax(i+1)
    pre = [[] for _ in range(NG)]
    post = [[] for _ in range(NG)]
    mf = []
    kq = [(kb, ks) for ks in range(KS) for kb in range(NKB)]                     # the order phase 1 of the next step consumes them
    kgaps = int(opt_val("kgaps", "2"))                                           # one K fragment read every `kgaps` gaps
    for t, (kk, db, qb) in enumerate(PV_ORDER):
        f = kk * DB + db
        if qb == 0 and "novread" not in OPT:
            pre[t].append(("WAIT", ("v", kk, db, 1)))
        mf.append(mfma_pv(cur, f % 8, kk, db, qb) if "nomfma2" not in OPT else "    s_nop 0")
        if qb == NQB - 1 and kk == 0 and "novread" not in OPT:                   # the slot's four MFMAs have issued: refill it for key step 1
            post[t] += v_read(f % 8, vbuf_cur, 1, db)
        if t % kgaps == 0 and t // kgaps < len(kq) and "nokread" not in OPT:
            post[t].append(k_read(kbuf_read, *kq[t // kgaps]))
    rare, back = new_label("rare"), new_label("rare_back")
    fl, flback = new_label("flush"), new_label("flush_back")
    inv, invback = new_label("inval"), new_label("inval_back")
    st2 = variant ^ 1
    head = [("LDS", f"ds_read_b64 {vr(T[4], 2)}, {v(TABV)} offset:8", "tabv"),
            ("LDS", f"ds_read_b64 {vr(T[6], 2)}, {v(TABV)} offset:32", "tabk")]
    rm = row_max_ops(nxt) if "norowmax" not in OPT else []
    nb = [f"    v_readfirstlane_b32 {s(VBS[st2])}, {v(T[4])}", f"    v_readfirstlane_b32 {s(VBS[st2] + 1)}, {v(T[5])}",
          f"    v_readfirstlane_b32 {s(TBS[st2])}, {v(T[6])}", f"    v_readfirstlane_b32 {s(TBS[st2] + 1)}, {v(T[7])}"]
    vq = rm[:8] + [("WAIT", "tabk")]
    rm = rm[8:]
    while rm or nb:
        vq += rm[:3]
        rm = rm[3:]
        if nb:
            vq.append(nb.pop(0))
    if "nostats" not in OPT:
        vq += stats_ops(rare, back, fl, flback, inv, invback)
        deferred.append(lambda: inval_block(inv, invback))
        deferred.append(lambda: rare_rescale_block(rare, back))
        deferred.append(lambda: flush_block(fl, flback))
    vq += softmax_stream(nxt, list(range(XQ)))
    # the first SAFE_GAPS gaps hold nothing that reads S_nxt (MFMA result -> VALU read hazard: the last QK MFMA has 8 passes)
    distribute(head, post, 0, 4, end=SAFE_GAPS)
    distribute(vq, post, SAFE_GAPS)
    emit_gaps(pre, mf, post)

    # ---- tail: the rare O rescale, drain, barrier
    resc, resc_back = new_label("resc"), new_label("resc_back")
    emit(f"s_cmp_lg_u32 {s(S_RESC)}, 0")
    emit(f"s_cbranch_scc1 {resc}")
    label(resc_back)
    deferred.append(lambda: rescale_o_block(resc, resc_back))
    emit(("DRAIN",))
    if "nobarrier" not in OPT:
        emit("s_barrier")
    emit(f"s_add_u32 {s(S_I)}, {s(S_I)}, 1")


def prologue():
    emit("; ---- lane id, parameter block -> SGPRs")
    emit(f"v_mbcnt_lo_u32_b32 {v(LANE)}, -1, 0")
    emit(f"v_mbcnt_hi_u32_b32 {v(LANE)}, -1, {v(LANE)}")
    emit(f"s_mov_b32 {s(S_WAVE)}, %0")
    emit(f"s_mov_b32 {s(S_PARAM)}, %1")
    emit(f"v_mov_b32 {v(T[0])}, {s(S_PARAM)}")
    for q in range(6):
        emit(f"ds_read_b128 {vr(4 * q, 4)}, {v(T[0])} offset:{16 * q}")
    emit("s_waitcnt lgkmcnt(0)")
    plist = [S_KBASE, S_KBASE + 1, S_VBASE, S_VBASE + 1, S_KRS, S_VRS, S_LASTROW, S_NTILES, S_C, S_THR, S_TAILVALID,
             S_FIRSTLAST, S_TAB, S_DOFLAGS, S_QBASE, S_QBASE + 1, S_QRS, S_QROW0, S_SEQLENQ, S_EXPORT, S_LDS, S_NEGC, S_TAU]
    for idx, sg in enumerate(plist):
        emit(f"v_readfirstlane_b32 {s(sg)}, {v(idx)}")
    emit("s_nop 4")
    emit(f"s_sub_u32 {s(S_NM1)}, {s(S_NTILES)}, 1")
    emit(f"s_lshl_b32 {s(S_DMAW)}, {s(S_WAVE)}, {ROW_SHIFT + 4}")          # a wave stages 16 rows = 4 KiB of a tile
    emit(f"s_add_u32 {s(S_DMAW)}, {s(S_DMAW)}, {s(S_LDS)}")
    emit(f"s_mov_b32 {s(S_I)}, 0")
    emit(f"s_mov_b32 {s(S_RESC)}, 0")
    emit(f"v_mov_b32 {v(NEGINF)}, 0xff800000")

    emit("; ---- per-lane constants: j = lane & 15 (query / key / d inside a 16-block), g = lane >> 4 (k-slot group)")
    emit(f"v_and_b32 {v(J16)}, 15, {v(LANE)}")
    emit(f"v_lshrrev_b32 {v(T[0])}, 4, {v(LANE)}")                           # g
    emit(f"v_lshlrev_b32 {v(G4)}, 2, {v(T[0])}")                             # 4 g
    # K fragment addresses: row j, chunk (4 ks + g) ^ j
    emit(f"v_lshlrev_b32 {v(T[1])}, {ROW_SHIFT}, {v(J16)}")
    emit(f"v_add_u32 {v(T[1])}, {s(S_LDS)}, {v(T[1])}")
    for ks in range(KS):
        emit(f"v_add_u32 {v(T[2])}, {4 * ks}, {v(T[0])}")
        emit(f"v_xor_b32 {v(T[2])}, {v(T[2])}, {v(J16)}")
        emit(f"v_lshl_add_u32 {v(KADDR[ks])}, {v(T[2])}, 4, {v(T[1])}")
    # V^T fragment addresses: row 4 g + (j >> 2), 8 bytes at chunk 2 db + ((j & 3) >> 1), half (j & 1); chunk ^= (row & 7) << 1
    emit(f"v_lshrrev_b32 {v(T[2])}, 2, {v(J16)}")                            # j >> 2
    emit(f"v_add_u32 {v(T[3])}, {v(G4)}, {v(T[2])}")                         # row = 4 g + (j >> 2)
    emit(f"v_and_b32 {v(T[4])}, 7, {v(T[3])}")                               # row & 7
    emit(f"v_lshlrev_b32 {v(T[3])}, {ROW_SHIFT}, {v(T[3])}")
    emit(f"v_add_u32 {v(T[3])}, {s(S_LDS)}, {v(T[3])}")
    emit(f"v_and_b32 {v(T[5])}, 1, {v(J16)}")
    emit(f"v_lshl_add_u32 {v(T[3])}, {v(T[5])}, 3, {v(T[3])}")               # + 8 (j & 1)
    emit(f"v_bfe_u32 {v(T[5])}, {v(J16)}, 1, 1")                             # (j & 3) >> 1
    for db in range(DB):
        emit(f"v_xor_b32 {v(T[6])}, {db}, {v(T[4])}")                        # db ^ (row & 7)
        emit(f"v_lshl_or_b32 {v(T[6])}, {v(T[6])}, 1, {v(T[5])}")            # swizzled chunk
        emit(f"v_lshl_add_u32 {v(VADDR[db])}, {v(T[6])}, 4, {v(T[3])}")
    # DMA image: a 1-KiB piece = 4 rows of 256 bytes; lane -> (rip = lane >> 4 = g, cpos = lane & 15 = j); row = 16 w + 4 p + rip
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, 4")
    emit(f"v_add_u32 {v(T[8])}, {s(S_T0)}, {v(T[0])}")                       # 16 w + rip
    emit(f"v_xor_b32 {v(RAGK)}, {v(J16)}, {v(T[0])}")
    emit(f"v_lshlrev_b32 {v(RAGK)}, 4, {v(RAGK)}")                           # K: (cpos ^ rip) << 4, piece p: ^ (4 p) << 4
    emit(f"v_lshlrev_b32 {v(RAGV)}, 1, {v(T[0])}")
    emit(f"v_xor_b32 {v(RAGV)}, {v(J16)}, {v(RAGV)}")
    emit(f"v_lshlrev_b32 {v(RAGV)}, 4, {v(RAGV)}")                           # V: (cpos ^ (rip << 1)) << 4, piece p: ^ (8 (p & 1)) << 4
    emit(f"s_mov_b32 {s(S_T1)}, {s(S_LASTROW)}")
    for p in range(4):
        bias = DMA_BIAS - 1024 * p
        emit(f"v_add_u32 {v(T[4])}, {4 * p}, {v(T[8])}")
        emit(f"v_min_i32 {v(T[4])}, {v(T[4])}, {s(S_T1)}")
        emit(f"v_mul_lo_u32 {v(LK[p])}, {v(T[4])}, {s(S_KRS)}")
        emit(f"v_xor_b32 {v(T[5])}, {(4 * p) << 4}, {v(RAGK)}")
        emit(f"v_add_u32 {v(LK[p])}, {v(LK[p])}, {v(T[5])}")
        emit(f"v_mul_lo_u32 {v(LV[p])}, {v(T[4])}, {s(S_VRS)}")
        emit(f"v_xor_b32 {v(T[5])}, {(8 * (p & 1)) << 4}, {v(RAGV)}")
        emit(f"v_add_u32 {v(LV[p])}, {v(LV[p])}, {v(T[5])}")
        if bias:
            emit(f"v_add_u32 {v(LK[p])}, {bias}, {v(LK[p])}")
            emit(f"v_add_u32 {v(LV[p])}, {bias}, {v(LV[p])}")

    emit("; ---- Q fragments -> AGPRs: row q_row0 + 64 wave + 16 qb + j, d = 32 ks + 8 g; rows past seqlen_q are ZERO rows")
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, 6")
    emit(f"s_add_u32 {s(S_T0)}, {s(S_T0)}, {s(S_QROW0)}")
    emit(f"s_sub_u32 {s(S_T1)}, {s(S_SEQLENQ)}, 1")
    emit(f"v_add_u32 {v(QROW_T)}, {s(S_T0)}, {v(LANE)}")                     # the lane's row in the transposed form
    emit(f"v_lshlrev_b32 {v(T[6])}, 4, {v(T[0])}")                           # g * 16 bytes
    for qb in range(NQB):                                                    # all 16 loads in flight together
        emit(f"v_add_u32 {v(T[8])}, {s(S_T0)}, {v(J16)}")
        if qb:
            emit(f"v_add_u32 {v(T[8])}, {16 * qb}, {v(T[8])}")
        emit(f"v_min_i32 {v(T[3])}, {v(T[8])}, {s(S_T1)}")
        emit(f"v_mad_u64_u32 {vr(T[4], 2)}, {sr(S_T64)}, {v(T[3])}, {s(S_QRS)}, 0")
        emit(f"v_add_co_u32 {v(T[4])}, vcc, {v(T[4])}, {v(T[6])}")
        emit(f"v_addc_co_u32 {v(T[5])}, vcc, 0, {v(T[5])}, vcc")
        emit(f"v_add_co_u32 {v(T[4])}, vcc, {s(S_QBASE)}, {v(T[4])}")
        emit(f"v_mov_b32 {v(T[7])}, {s(S_QBASE + 1)}")
        emit(f"v_addc_co_u32 {v(T[5])}, vcc, {v(T[5])}, {v(T[7])}, vcc")
        for ks in range(KS):
            emit(f"global_load_dwordx4 {vr(16 * qb + 4 * ks, 4)}, {vr(T[4], 2)}, off offset:{64 * ks}")
    emit("s_waitcnt vmcnt(0)")
    for qb in range(NQB):
        emit(f"v_add_u32 {v(T[8])}, {s(S_T0)}, {v(J16)}")
        if qb:
            emit(f"v_add_u32 {v(T[8])}, {16 * qb}, {v(T[8])}")
        emit(f"v_cmp_gt_i32 vcc, {s(S_SEQLENQ)}, {v(T[8])}")
        for r in range(16):
            emit(f"v_cndmask_b32 {v(16 * qb + r)}, 0, {v(16 * qb + r)}, vcc")
    for r in range(16 * NQB):
        emit(f"v_accvgpr_write_b32 a{Q_A0 + r}, {v(r)}")
    emit("; ---- state")
    for r in range(4 * DB * NQB):
        emit(f"v_accvgpr_write_b32 a{r}, 0")
    for qb in range(NQB):
        emit(f"v_mov_b32 {v(L[qb])}, 0")
        emit(f"v_mov_b32 {v(ALPHA[qb])}, 1.0")

    emit("; ---- tile addresses of positions 1..3 from the table; K(0) fragments -> AGPRs, S(0) = K(0) Q^T, then K(1) fragments")
    emit(f"v_mov_b32 {v(T[6])}, {s(S_TAB)}")
    emit(f"ds_read_b64 {vr(T[8], 2)}, {v(T[6])} offset:32")          # tab[2].k : K(2), staged below
    emit(f"ds_read_b64 {vr(T[10], 2)}, {v(T[6])} offset:48")         # tab[3].k : K(3), staged by step 0
    emit(f"ds_read_b64 {vr(T[12], 2)}, {v(T[6])} offset:24")         # tab[1].v : V(1), staged by step 0
    emit(f"v_add_u32 {v(TABV)}, 32, {v(T[6])}")                      # step 0 reads tab[2].v and tab[4].k
    for ks in range(KS):
        for kb in range(NKB):
            emit(k_read(0, kb, ks))
    emit(("DRAIN",))
    emit(f"v_readfirstlane_b32 {s(TBS[0])}, {v(T[8])}")
    emit(f"v_readfirstlane_b32 {s(TBS[0] + 1)}, {v(T[9])}")
    for ks, kb, qb in QK_ORDER:
        out.append(mfma_qk(0, kb, ks, qb))
    for ks in range(KS):
        for kb in range(NKB):
            emit(k_read(KV_TILE, kb, ks))
    emit(("DRAIN",))
    emit("s_barrier")                                          # every wave has read K(0) and K(1): both K buffers are free
    for it in dma_ops(0, 0, do_k=True, do_v=False):            # K(2) -> K buffer 0. V(1) / K(3) are staged by step 0.
        out.append(it)
        if "m0" in it:
            emit("s_nop 0")
    for dst, src in ((TBS[0], T[10]), (TBS[0] + 1, T[11]), (VBS[0], T[12]), (VBS[0] + 1, T[13])):
        emit(f"v_readfirstlane_b32 {s(dst)}, {v(src)}")
    emit("s_nop 7")
    # seqlen-k mask: only if the first walked tile is tile k_tiles - 1 and tail_valid < 64 (mask.h:44-78; mainloop...:1626)
    nomask = new_label("nomask")
    emit(f"s_cmp_eq_u32 {s(S_FIRSTLAST)}, 1")
    emit(f"s_cbranch_scc0 {nomask}")
    emit(f"s_cmp_lt_i32 {s(S_TAILVALID)}, 64")
    emit(f"s_cbranch_scc0 {nomask}")
    for kb in range(NKB):
        for r in range(4):
            emit(f"v_add_u32 {v(T[0])}, {16 * kb + r}, {v(G4)}")                  # key 16 kb + 4 g + r
            emit(f"v_cmp_gt_i32 vcc, {s(S_TAILVALID)}, {v(T[0])}")               # key < tail_valid -> keep
            for qb in range(NQB):
                emit(f"v_cndmask_b32 {v(S_(0, qb, kb) + r)}, {v(NEGINF)}, {v(S_(0, qb, kb) + r)}, vcc")
    label(nomask)
    for op in row_max_ops(0):
        out.append(op)
    emit("s_nop 1")
    # first-tile statistics: m_true = m_ref = row max; position 0 is never flagged (softmax.h:153)
    for op in transposing_reduce(M4, "v_max_f32", MTRUE_T):
        out.append(op)
    for op in set_reference_ops():
        out.append(op)
    emit(f"s_mov_b32 {s(S_DOMASK)}, 1")                        # position 0 is never flagged; position 1 votes into bit 1
    emit(f"s_mov_b32 {s(S_BIT)}, 2")
    emit(f"s_mov_b32 {s(S_DOWORD)}, {s(S_DOFLAGS)}")
    for op in softmax_stream(0, list(range(XQ))):
        out.append(op)
    emit(("DRAIN",))
    emit("s_barrier")


def epilogue():
    """finalize (softmax.h:275-296) + store (epilogue_fwd.hpp:214-403) straight from the accumulators.
    Parameter words (LDS block, written by the C++ shell; the same as gen_epilogue.py reads): [24] [25] O row 0 of this (batch, head),
    [26] O row stride in bytes, [27] c ln 2, [28] [29] &lse[row 0] or 0, [30] O scale, [31] added to the LSE."""
    S_OBASE, S_LSEB, S_ORS, S_CLN2, S_OSCALE, S_LSEADD = S_TB, S_VB, S_T0, S_T1, S_T2, S_T3
    emit("; ---- flush the last (partial) vote word")
    nofl = new_label("nolastflush")
    emit(f"s_cmp_eq_u32 {s(S_DOMASK)}, 0")
    emit(f"s_cbranch_scc1 {nofl}")
    flush_domask()
    label(nofl)
    emit("s_nop 15")                                           # the last PV MFMAs have written the accumulators
    emit("s_nop 15")
    emit("; ---- finalize + store O and LSE straight from the accumulators")
    emit(f"v_mov_b32 {v(T[0])}, {s(S_PARAM)}")
    emit(f"ds_read_b128 {vr(T[4], 4)}, {v(T[0])} offset:96")
    emit(f"ds_read_b128 {vr(T[8], 4)}, {v(T[0])} offset:112")
    emit("s_waitcnt lgkmcnt(0)")
    for dst, src in ((S_OBASE, T[4]), (S_OBASE + 1, T[5]), (S_ORS, T[6]), (S_CLN2, T[7]), (S_LSEB, T[8]), (S_LSEB + 1, T[9]),
                     (S_OSCALE, T[10]), (S_LSEADD, T[11])):
        emit(f"v_readfirstlane_b32 {s(dst)}, {v(src)}")
    emit("s_nop 4")
    # l of every row, transposed: lane (j, g) = row 16 g + j of the wave = row `lane`
    for op in transposing_reduce(L, "v_add_f32", T[0]):
        out.append(op)
    emit(f"v_rcp_f32 {v(T[2])}, {v(T[0])}")
    emit(f"v_log_f32 {v(T[3])}, {v(T[0])}")
    emit("s_nop 0")
    emit(f"v_fma_f32 {v(T[1])}, -{v(T[0])}, {v(T[2])}, 1.0")            # one Newton step: 1/l to < 1 ulp
    emit(f"v_fma_f32 {v(T[2])}, {v(T[1])}, {v(T[2])}, {v(T[2])}")
    emit(f"v_mul_f32 {v(T[2])}, {s(S_OSCALE)}, {v(T[2])}")
    emit(f"v_mul_f32 {v(T[3])}, 0x3f317218, {v(T[3])}")                  # ln 2
    emit(f"v_fma_f32 {v(T[3])}, {v(MREF_T)}, {s(S_CLN2)}, {v(T[3])}")
    emit(f"v_add_f32 {v(T[3])}, {s(S_LSEADD)}, {v(T[3])}")
    emit(f"v_cmp_lg_f32 vcc, 0, {v(T[0])}")                               # false for l == 0 and for NaN
    emit(f"v_cndmask_b32 {v(T[2])}, 0, {v(T[2])}, vcc")
    emit(f"v_cndmask_b32 {v(T[3])}, {v(NEGINF)}, {v(T[3])}, vcc")
    # LSE: every lane stores its own row (coalesced), if the caller wants it and the row exists
    nolse = new_label("nolse")
    emit(f"s_cmp_eq_u64 {sr(S_LSEB)}, 0")
    emit(f"s_cbranch_scc1 {nolse}")
    emit(f"v_cmp_gt_i32 vcc, {s(S_SEQLENQ)}, {v(QROW_T)}")
    emit(f"s_and_saveexec_b64 {sr(S_EXEC)}, vcc")
    emit(f"v_lshlrev_b32 {v(T[6])}, 2, {v(QROW_T)}")
    emit(f"v_add_co_u32 {v(T[6])}, vcc, {s(S_LSEB)}, {v(T[6])}")
    emit(f"v_mov_b32 {v(T[7])}, {s(S_LSEB + 1)}")
    emit(f"v_addc_co_u32 {v(T[7])}, vcc, 0, {v(T[7])}, vcc")
    emit(f"global_store_dword {vr(T[6], 2)}, {v(T[3])}, off")
    emit(f"s_mov_b64 exec, {sr(S_EXEC)}")
    label(nolse)
    # 1 / l back to the q-blocks: INV[qb] = scale of the lane's query in q-block qb (the ALPHA registers are dead here)
    INV = ALPHA
    for op in broadcast_ops(T[2], INV):
        out.append(op)
    # per-lane byte offset inside a pair of d-blocks after the swap: even groups keep block a (16 contiguous bytes at d = 4 g of it),
    # odd groups block b (at d = 16 + 4 (g - 1)):  g = 0, 1, 2, 3 -> 0, 32, 16, 48 bytes
    LOFF = M4[0]
    emit(f"v_lshrrev_b32 {v(T[0])}, 2, {v(G4)}")                          # g
    emit(f"v_and_b32 {v(T[1])}, 1, {v(T[0])}")
    emit(f"v_lshlrev_b32 {v(LOFF)}, 5, {v(T[1])}")                        # 32 (g & 1)
    emit(f"v_lshrrev_b32 {v(T[1])}, 1, {v(T[0])}")
    emit(f"v_lshl_or_b32 {v(LOFF)}, {v(T[1])}, 4, {v(LOFF)}")             # + 16 (g >> 1)
    emit(f"s_lshl_b32 {s(S_EXPORT)}, {s(S_WAVE)}, 6")
    emit(f"s_add_u32 {s(S_EXPORT)}, {s(S_EXPORT)}, {s(S_QROW0)}")         # first row of the wave
    for qb in range(NQB):
        emit(f"v_add_u32 {v(T[12])}, {s(S_EXPORT)}, {v(J16)}")
        if qb:
            emit(f"v_add_u32 {v(T[12])}, {16 * qb}, {v(T[12])}")
        emit(f"v_cmp_gt_i32 vcc, {s(S_SEQLENQ)}, {v(T[12])}")              # rows past seqlen_q are not stored
        emit(f"s_and_saveexec_b64 {sr(S_EXEC)}, vcc")
        emit(f"v_mad_u64_u32 {vr(T[4], 2)}, {sr(S_T64)}, {v(T[12])}, {s(S_ORS)}, 0")
        emit(f"v_add_co_u32 {v(T[4])}, vcc, {v(T[4])}, {v(LOFF)}")
        emit(f"v_addc_co_u32 {v(T[5])}, vcc, 0, {v(T[5])}, vcc")
        emit(f"v_add_co_u32 {v(T[4])}, vcc, {s(S_OBASE)}, {v(T[4])}")
        emit(f"v_mov_b32 {v(T[6])}, {s(S_OBASE + 1)}")
        emit(f"v_addc_co_u32 {v(T[5])}, vcc, {v(T[5])}, {v(T[6])}, vcc")
        # pairs of d-blocks (2 p, 2 p + 1): 8 accumulators -> scale -> 4 packed registers X0 X1 (block 2 p) Y0 Y1 (block 2 p + 1) ->
        # v_permlane16_swap (X0, Y0), (X1, Y1) -> one 16-byte store of {X0, X1, Y0, Y1}. Software-pipelined over two register sets
        # (the S buffers are dead): read(p + 1) sits between pack(p) and swap(p) - the 2 wait states a swap needs behind a VALU write.
        # The swaps execute with the row mask of EXEC: a lane whose partner row is masked off still exchanges garbage with it, which
        # only ever lands in a lane that does not store (both lanes of a pair belong to the same query row: j is the same).
        def stage_read(p):
            r = 8 * (p % 2)
            return [f"v_accvgpr_read_b32 {v(r + k)}, a{O_(qb, 2 * p) + k}" for k in range(8)]

        def stage_pack(p):
            r = 8 * (p % 2)
            return [f"v_mul_f32 {v(r + k)}, {v(r + k)}, {v(INV[qb])}" for k in range(8)] + \
                   [f"{CVT_OP} {v(r + k)}, {v(r + 2 * k)}, {v(r + 2 * k + 1)}" for k in range(4)]

        def stage_store(p):
            r = 8 * (p % 2)
            # after the packs: r, r+1 = block 2p (X0, X1), r+2, r+3 = block 2p+1 (Y0, Y1). Swap (X0, Y0) and (X1, Y1); the 16 bytes of
            # a lane are then {X0, X1, Y0, Y1} = registers r, r+1, r+2, r+3 in this order for BOTH parities:
            #   even g: X = own block-a half, Y = the odd partner's block-a half (d + 4)      -> a[4g .. 4g+7]
            #   odd g:  X = the even partner's block-b half (d - 4), Y = own block-b half     -> b[4(g-1) .. 4(g-1)+7]
            return [f"v_permlane16_swap_b32 {v(r)}, {v(r + 2)}", f"v_permlane16_swap_b32 {v(r + 1)}, {v(r + 3)}", "s_nop 0",
                    f"global_store_dwordx4 {vr(T[4], 2)}, {vr(r, 4)}, off offset:{64 * p}"]

        seq = stage_read(0) + stage_pack(0)
        for p in range(4):
            seq += stage_read(p + 1) if p + 1 < 4 else ["s_nop 1"]
            seq += stage_store(p)
            if p + 1 < 4:
                seq += stage_pack(p + 1)
        for op in seq:
            emit(op)
        emit(f"s_mov_b64 exec, {sr(S_EXEC)}")
    emit("s_waitcnt lgkmcnt(0)")


def main():
    prologue()
    loop, done = new_label("loop"), new_label("done")
    if opt_val("align", "") or opt_val("pad4", ""):           # code placement: see gen_fwd_x64.py main(); this body is pinned at phase 8 (not swept)
        if opt_val("align", ""):
            out.append(f".p2align {opt_val('align', '')}")
        for _ in range(int(opt_val("pad4", "0"))):
            emit("s_nop 0")
    else:
        out.append(".p2align 5")
        emit("s_nop 0")
        emit("s_nop 0")
    label(loop)
    for variant in (0, 1):
        emit(f"s_cmp_lt_u32 {s(S_I)}, {s(S_NTILES)}")
        emit(f"s_cbranch_scc0 {done}")
        step(variant)
    emit(f"s_branch {loop}")
    for blk in deferred:
        blk()
    label(done)
    epilogue()
    lines = finalize(out)
    text = "\n".join(lines)
    path = sys.argv[1] if len(sys.argv) > 1 else "la_fwd_x64_m16_body.inc"
    with open(path, "w") as f:
        f.write("// GENERATED by gen_fwd_x64_m16.py — do not edit. Inline-asm body of la_fwd_x64_kernel<.., 128> on v_mfma_f32_16x16x32.\n")
        f.write(option_tag() + "\n")
        f.write('R"ASM(\n' + text + '\n)ASM"\n')
    print(f"wrote {path}: {len(lines)} lines, {text.count('v_mfma')} MFMAs")


if __name__ == "__main__":
    main()
