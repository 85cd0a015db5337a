#!/usr/bin/env python
"""Generates la_fwd_x64[_d<D>][_f16]_body.inc: hand-scheduled gfx950 main loop of the bf16 / fp16 QK-Skip forward with
ONE wave per SIMD (one 4-wave workgroup per CU). The text below describes the head_dim-128 form: 64 query rows per wave, q-tile
256 x k-tile 64. LA_X64_D selects the other forms (96: the same with 12 of 16 fragments; 256 / 192: one 32-row q-block per wave,
q-tile 128 - see the comments at D / DL / NQB below), LA_X64_DTYPE the 16-bit element type.

Why this shape (measured, HISTORY.md section 4.6): with two 32-row waves per SIMD the MFMA pipe idles 37 % of the
cycles and the chip clocks at 1.69 GHz, because every wave re-reads the whole K/V tile from LDS for only 32 rows and
the two unsynchronised waves fight for issue slots. One wave with the whole 512-register file halves the LDS bytes
per FLOP, and its single in-order stream is scheduled here gap by gap.

Register file (per lane):  AGPR  a[0:127]   O^T  (2 q-blocks x 4 d-blocks x 16)
                                 a[128:191] Q    (2 q-blocks x 8 k-steps x 4, B operand of S^T = K Q^T)
                                 a[192:255] K    fragments of the NEXT tile (16 x 4, A operand)
                           VGPR  v[0:63] / v[64:127]  S^T ping / pong (2 key blocks x 2 q-blocks x 16); P (bf16) is
                                 compacted IN PLACE into the first 4 registers of every 8, which is exactly the B operand
                                 of the PV MFMA, so P needs no registers of its own
                                 v[128:159] V^T fragment ring (8 x 4), then addresses / running state / temporaries.

Step i (S_cur = scores of tile i, partly exponentiated; K fragments of tile i+1 in AGPRs; V(i), K(i+2) in LDS):
  phase 1  32 MFMA  S_nxt = K(i+1) Q^T   ||  rest of P(i) = exp2(S_cur*c - m_ref*c), row sums, bf16 compaction;
                                              LDS-DMA issue of V(i+1), K(i+3); first 8 V^T fragment reads
  phase 2  32 MFMA  O^T += V(i)^T P(i)^T ||  K(i+2) fragment reads -> AGPRs; remaining V^T fragment reads; the global addresses
                                              of the tiles the NEXT step stages, read from the tile-address table the C++ shell
                                              built in LDS (no tile lookup or address arithmetic on the scalar unit); row max of
                                              S_nxt, skip vote (rotating bit), running max, lazy-rescale decision; first part of P(i+1)
  tail     rare O rescale, vmcnt/lgkmcnt drain, ONE barrier.
After the walk: 1/l, LSE and the bf16 O store straight from the accumulators (gen_epilogue.py).
Lazy rescale: O and l are kept relative to a reference max m_ref that only follows the true running max m_true when
it has grown by more than `tau` (log2 units); P stays <= 2^tau. The skip vote uses m_true, so lists are bit-exact.
"""
import os
import sys

from gen_epilogue import store_epilogue

OPT = set(x for x in os.environ.get("LA_X64_OPT", "").split(",") if x)


def opt_val(key, default):
    for o in OPT:
        if o.startswith(key + ":"):
            return o[len(key) + 1:]
    return default


# Options that only MOVE instructions (placement, split points, cache policy, code alignment): the body computes the same results bit
# for bit. Every other option drops work or changes the arithmetic - pricing experiments (tools/asm_variants.py). The first line of a
# generated body says which kind went in; liteattention_amd/build.py refuses the latter for the product library and records both in
# la_build_info() for A/B builds (--out=).
SCHEDULE_ONLY = {"snake", "x", "cap1", "cap2", "dmagaps", "dmapol", "align", "pad4", "pad4b", "e64", "wp2", "wp2b", "wc2", "kearly", "klate", "klate2", "expblock", "norot", "w2", "pk"}


def option_tag():
    wrong = sorted(o for o in OPT if o.split(":")[0] not in SCHEDULE_ONLY)
    return (f"// la_body_options: {','.join(sorted(OPT)) or '-'}; wrong_results={1 if wrong else 0}"
            + (f" (PRICING ONLY, results are wrong: {','.join(wrong)})" if wrong else ""))


# 16-bit element type of Q / K / V / P / O: "bf16" (default) or "f16" (LA_X64_DTYPE; the build generates one body per type). Only the
# MFMA opcode and the fp32 -> 16-bit pack differ: the fragment layout, the transpose reads and the schedule are type-agnostic.
DTYPE = os.environ.get("LA_X64_DTYPE", "bf16")
MFMA_OP = {"bf16": "v_mfma_f32_32x32x16_bf16", "f16": "v_mfma_f32_32x32x16_f16"}[DTYPE]
CVT_OP = {"bf16": "v_cvt_pk_bf16_f32", "f16": "v_cvt_pk_f16_f32"}[DTYPE]          # both round to nearest even
# Head dim of the body: 128 (two 32-row q-blocks per wave, q-tile 256) or 256 (LA_X64_D=256: ONE q-block per wave, q-tile 128 - O^T
# alone is 128 registers per q-block there). Both have 32 + 32 MFMAs per step; the 256 form does half the softmax per step and
# reads twice the K/V fragment bytes per FLOP. See the register-map comments below for what moves.
# Head dims 96 and 192 (LA_X64_D=96 / 192) are the 128 / 256 forms with three quarters of the MFMAs: 12 / 24 K fragments and V^T
# fragments per tile instead of 16 / 32, the LDS image keeps the 256 / 512-byte row pitch (both XOR swizzles are defined on it; the
# quarter of each LDS row behind the data is filled with duplicates by the DMA lanes that have no column of their own).
# Head dim 64 (LA_X64_D=64, round 3) is the 128 form with half the MFMAs (8 K fragments, 8 V^T fragments per tile: 16 + 16 MFMAs per
# step) under the SAME softmax: bound by the vector unit like the fp8 kernel. It has its own LDS image: rows of 128 bytes (tiles of
# 8 KiB, 2 DMA pieces per wave and tensor), K chunks swizzled by (row >> 1) & 7 and V 64-byte segments by (row >> 1) & 1 - the
# conflict-free forms for a 128-byte pitch (16 consecutive rows of one chunk column / 4 rows x 64 bytes of a transpose read cover
# all 64 banks once).
# W2 (LA_X64_OPT=w2, head_dim 64 only; round 4): TWO waves per SIMD - a workgroup of EIGHT waves of 32 query rows (q-tile 256, as before).
# At head_dim 64 a step has half the MFMAs of head_dim 128 under the same softmax, and a lone wave issues one instruction per ~4-5
# cycles whatever it is: the one-wave body is issue-bound (MFMA busy 46 % at 2.1 GHz, waves issuing 78 % of the time). A second wave
# per SIMD doubles the issue bandwidth, and one wave's softmax runs under the other's MFMAs. Two waves share the SIMD's 512 registers,
# so a wave owns ONE 32-row q-block (as at head dims 192 / 256): O^T 32 + Q 16 + K fragments 32 AGPRs, S 32 + a 4-deep V^T ring + state
# in 90 VGPRs, which leaves the C++ shell the registers it keeps across the body. Every K / V^T fragment then feeds ONE MFMA instead of
# two: LDS reads double to 1 KiB per MFMA - 128 KiB per step and CU, half of what the LDS delivers in the 1024 cycles the step's MFMAs
# take (256 B/clk/CU), and the second wave hides their latency. The intra-wave software pipeline of the one-wave body (two S buffers) is
# gone: a step is QK -> row max / vote -> softmax -> PV in program order (step_w2), with the LDS reads and the DMA issue inside the two
# MFMA runs; K and V are staged ONE tile ahead.
W2 = "w2" in OPT
D = int(os.environ.get("LA_X64_D", "128"))
assert D in (64, 96, 128, 192, 256)
assert not W2 or D == 64
# LA_X64_FORM=half (round 6; LA_FLAG_HALF_VOTE, la_fwd_x64_half*_body.inc): the skip lists are kept per 128-ROW HALF of the 256-row
# workgroup (waves 0-1 / waves 2-3; kBlockM = 128 as the reference's bf16 head_dim-128 tile, tile_size.h:35-39). The workgroup walks
# the UNION of its two lists (the C++ shell merges them into one descending tile sequence and one activity bit per position and
# half); a wave whose half does not list the tile at a position sits that tile out: no MFMA, no softmax, no fragment reads - it only
# stages its DMA pieces, follows the tile-address table and meets the barrier. The software pipeline of a step mixes two tiles (QK,
# statistics and the first part of softmax of tile i+1; the rest of softmax and PV of tile i), so a step comes in four forms by
# (a(i), a(i+1)), dispatched at its top from a 64-bit activity window in SGPRs (ACT: bit k = a(i + k), shifted every step, refilled
# from LDS when a vote word is flushed). Votes go to the half's own vote words. Results per half are exactly those of an independent
# 128-row q-tile walking its own list (the oracle at block_m = 128).
HALF = os.environ.get("LA_X64_FORM", "") == "half"
assert not HALF or (D in (64, 96, 128) and not W2)
DL = 64 if D == 64 else (128 if D <= 128 else 256)   # layout head dim: LDS row pitch, q-blocks per wave, DMA pieces
NQB = 1 if W2 else (2 if DL <= 128 else 1)   # 32-row q-blocks per wave
NW = 8 if W2 else 4                       # waves per workgroup
ROW_SHIFT = {64: 7, 128: 8, 256: 9}[DL]   # log2 of the LDS row pitch in bytes
KS = D // 16                              # k-steps of S^T = K Q^T
DB = D // 32                              # 32-wide d-blocks of O^T
ROW = 2 * DL                              # bytes per K / V row in LDS
NKF, NVF = 2 * KS, 4 * DB                 # K fragments (A operands of QK) / V^T fragments (A operands of PV) per tile
NG = NKF * NQB                            # MFMAs (= gaps for the other pipes) per phase: 32, or 24 for head dims 96 / 192
assert NG == NVF * NQB
PW = (64 // NW) * ROW // 1024             # 1-KiB DMA pieces per wave per tile (a wave stages 64 / NW of the 64 rows)
# Where the K fragments of tile i+2 are read from LDS. Two q-blocks per wave: during phase 2 of step i (the LDS pipe has room there).
# One q-block per wave (head dims 192 / 256): a phase moves the same 32 MFMAs over half the rows, so LDS bytes per MFMA double and
# phase 2 (24 V^T + 32 K fragments = 56 KiB per wave, 85 % of the CU's LDS read rate) stalls the MFMAs; there every K fragment
# register is refilled in PHASE 1, right behind the QK MFMA of tile i+1 that consumed it (K(i+2) has been in LDS since the barrier
# of step i-1): 38 / 26 KiB per phase instead of 8 / 56. Measured (dense S=16384 H=40, same box): head_dim 256 1201 -> 1218 TFLOP/s
# (+1.4 %), head_dim 192 1111 -> 1107 (24 + 24 KiB there: nothing to balance) - small, because these forms sit at the power limit
# too (1.8-1.9 GHz). Default: 256 only; `kearly` / `klate2` force it on / off for A/B.
K_EARLY = NQB == 1 and (D == 256 or "kearly" in OPT) and "klate2" not in OPT
XPAIRS = int(opt_val("x", {64: "4", 256: "6"}.get(D, "5")))   # pair-groups (of 16; one group = the same pair of both q-blocks) done in phase 2 (head_dim 64: 4 since round 4,
                                                      # +1.8-2.2 % over round 3's 8 on two boxes; 1 / 2 / 3 / 6 / 10 / 12 lose to it; head_dim 256: 6, +1 % over 5; 96 / 192: 5 stays)
# `snake` (A/B): the q-block order of a fragment's two MFMAs flips with the fragment's parity, so that every MFMA shares an operand register
# with its predecessor inside a group - (K0,Q0) (K0,Q1) (K1,Q1) (K1,Q0) instead of (K0,Q0) (K0,Q1) (K1,Q0) (K1,Q1). The matrix pipe
# ALONE gains 0.9 % from it on N(0,1) data (tools/debug/mfma_order_bench.py: operand toggling is what an MFMA's energy depends on).
SNAKE = "snake" in OPT


def qb_of(t):
    """q-block of the MFMA at position t of a phase (NQB MFMAs per fragment, fragment t // NQB)"""
    pos, f = t % NQB, t // NQB
    return pos ^ (f & 1) if SNAKE and NQB == 2 else pos


CAP1 = int(opt_val("cap1", "0"))          # fillers per MFMA gap the distributor may place (0 = balance evenly)
CAP2 = int(opt_val("cap2", "0"))
DMA_GAPS = [int(x) for x in opt_val("dmagaps", {128: "1,2,4,6,8,10,11,13,15,17", 96: "0,1,3,4,6,7,8,9,11,12", 64: "1,2,4,8,9,11"}[D] if DL <= 128 else
                                    "1,2,3,4,5,7,8,9,10,11,13,14,15,16,17,19,20,21,22,23").replace(".", ",").split(",")]   # m0K,K0..3,m0V,V0..3 (phase 1)
assert W2 or max(DMA_GAPS) < NG          # (step_w2 places its DMA pieces itself)

Q_A0, K_A0 = (32, 48) if W2 else (128, 192)     # first AGPR of the Q fragments / of the K fragments
LOOP_PHASE = {64: 24, 96: 8, 128: 8, 192: 8, 256: 0}[D]      # bytes past a 32-byte boundary at which the loop head is placed (see main())


# ---------------------------------------------------------------- AGPR map
def O_(qb, db):
    return 16 * DB * qb + 16 * db


def QA(qb, ks):
    return Q_A0 + 4 * KS * qb + 4 * ks


def KFRAG(j):
    """Operand of K fragment j = kb * KS + ks. head_dim 128: a[192:255]. head_dim 256: 32 fragments; key block 0 in a[192:255],
    key block 1 in v[64:127] (the S buffers are half the size there; an MFMA A operand may be either file)."""
    return f"a[{K_A0 + 4 * j}:{K_A0 + 4 * j + 3}]" if j < 16 else f"v[{64 + 4 * (j - 16)}:{64 + 4 * (j - 16) + 3}]"


# ---------------------------------------------------------------- VGPR map
def S_(sset, kb, qb):
    return 32 * NQB * sset + 16 * NQB * kb + 16 * qb


VF = [128 + 4 * i for i in range(8)]
KADDR = list(range(160, 168))
VADDR = list(range(168, 172))
# per-lane DMA source offsets, one per piece: head_dim 256 has 8 pieces per tensor; the second four sit in the registers the
# second q-block's running state has at head_dim 128 (MTRUE[1], MREF[1], NMS[1], L0[1], L1[1], MLOC[1], MLOC2[1], ALPHA[1])
LK = list(range(172, 176)) + ([181, 183, 185, 188] if DL == 256 else [])
LV = list(range(176, 180)) + ([189, 191, 193, 195] if DL == 256 else [])
# (L0[qb], L1[qb]) and (NMS[0], NMS[1]) are even-aligned 64-bit pairs: operands of the packed-fp32 VALU ops
MTRUE, MREF, NMS, L0, L1, MLOC, MLOC2, ALPHA = ([180, 181], [182, 183], [184, 185], [186, 188], [187, 189], [190, 191],
                                                [192, 193], [194, 195])
T = list(range(196, 212))                 # temporaries; (T[4],T[5]) even-aligned 64-bit pair
NEGINF, HH4, LANE, RIPROW = 212, 213, 214, 215
QROW = [216, 217]
RAGK, RAGV = 218, 219                     # ragged-path swizzled chunk offsets (constants)
MTHR = [220, 221]                         # m_ref + tau/c: the lazy-rescale trigger level
TABV = 222                                # LDS address of tab[i + 2], the tile-address table entry step i reads from
if W2:                                    # 90 VGPRs: S v[0:31] (one buffer), V^T ring v[32:47] (4 slots), addresses / state / temporaries
    VF = [32 + 4 * i for i in range(4)]
    KADDR = list(range(48, 52)) + [0] * 4
    VADDR = [52, 53, 0, 0]
    LK, LV = [54], [55]
    MTRUE, MREF, NMS, MLOC, L0, L1, MTHR, QROW = [56], [57], [58], [59], [60], [61], [62], [63]
    NEGINF, HH4, TABV = 64, 65, 66
    T = list(range(68, 84))               # (T[4], T[5]) an even-aligned pair
    ALPHA, MLOC2 = [84], [85]
    LANE, RIPROW, RAGK, RAGV = 86, 87, 88, 89

# ---------------------------------------------------------------- SGPR map (s32-s34 are ABI-reserved: unused)
S_KBASE, S_VBASE, S_QBASE = 36, 38, 40    # 64-bit
S_TB, S_VB, S_EXEC, S_T64, S_T64B = 42, 44, 46, 48, 50   # 64-bit temps
(S_KRS, S_VRS, S_LASTROW, S_NTILES, S_C, S_THR, S_TAILVALID, S_FIRSTLAST, S_TAB, S_DOFLAGS, S_WAVE, S_I, S_DOMASK,
 S_FREE0, S_FREE1, S_FREE2, S_LDS, S_T0, S_T1, S_T2, S_T3, S_NM1, S_QRS, S_QROW0, S_SEQLENQ, S_EXPORT, S_PARAM, S_DOWORD, S_NEGC,
 S_FREE3, S_DMAW, S_FREE4, S_TAU, S_RESC, S_FREE5) = range(52, 87)
S_FREE6, S_TB2, S_VB2, S_BIT = 87, 88, 90, 92
S_STATE = S_FREE3         # HALF: the form of the NEXT step, computed in front of the drain: (a(i), a(i+1)) as bits 0, 1, or 4 = the walk is over
S_ACT, S_ACTPTR, S_HSTRIDE = S_T64B, S_FREE1, S_FREE2    # HALF: s[50:51] = activity window (bit k = a(i + k)), LDS address of the next activity word, bytes between the halves' flag blocks
S_ONES = S_FREE0                                 # packed (1.0, 1.0) of the element type: src0 of the row-sum dot (dotsum)    # second set of DMA bases (the loop is unrolled by two); the rotating vote bit
S_CC = 94                                        # s[94:95] = (c, c): scalar operand of v_pk_fma_f32
TBS, VBS = [S_TB, S_TB2], [S_VB, S_VB2]

KV_TILE = 64 * ROW
V_REGION = 2 * KV_TILE
DMA_POLICY = {"": "", "nt": " nt", "sc0": " sc0", "sc1": " sc1"}[opt_val("dmapol", "")]    # cache-policy experiments on the K/V stream
DMA_BIAS = 3072           # S_KBASE / S_VBASE hold (tensor base - DMA_BIAS); LK / LV[j] hold (+DMA_BIAS - 1024 j): see dma_ops

out = []          # IR: str | ("LDS", str, tag) | ("WAIT", tag) | ("DRAIN",)


def emit(x):
    out.append(x if isinstance(x, tuple) else "    " + x)


def label(s):
    out.append(s + ":")


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
    return f".LX{prefix}_{uid[0]}_%="


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
            lines.append("    s_waitcnt lgkmcnt(0)" if "nowaitvm" in OPT else "    s_waitcnt vmcnt(0) lgkmcnt(0)")
            q = []
    return lines


# ---------------------------------------------------------------- building blocks (return item lists)
def k_read(kbuf_imm, j):
    """K fragment j: rows 32 kb + (lane & 31), 16-byte chunk 2 ks + hh (XOR-swizzled by the row: KADDR[ks & 7]); the chunks of
    k-steps 8..15 (head_dim 256) are the same addresses + 256 bytes (the swizzle stays inside a 256-byte half row)."""
    kb, ks = j // KS, j % KS
    if "lds23" in OPT and j % 3 == 2:      # PRICING ONLY (wrong results): every third fragment read dropped - what a body with 1.5 x the rows per fragment would read
        return "    ; (lds23: K fragment read dropped)"
    return ("LDS", f"ds_read_b128 {KFRAG(j)}, {v(KADDR[ks & 7])} offset:{kbuf_imm + kb * 32 * ROW + (ks >> 3) * 256}", ("k", j))


def v_read(slot, vbuf_imm, m):
    """V^T fragment m = 4 db + kk: keys 16 kk .. 16 kk + 15, d-block db (VADDR[db & 3]; d-blocks 4..7 of head_dim 256: + 256 bytes)."""
    db, kk = m >> 2, m & 3
    if "lds23" in OPT and m % 3 == 2:
        return ["    ; (lds23: V^T fragment read dropped)"]
    if "vfake" in OPT:      # pricing only: ONE b128 read per fragment, as a pre-transposed V^T image would need
        return [("LDS", f"ds_read_b128 {vr(VF[slot], 4)}, {v(KADDR[2 * kk])} offset:{V_REGION + vbuf_imm + db * 4096}", ("v", m, 1))]
    off = vbuf_imm + kk * 16 * ROW + (db >> 2) * 256
    return [("LDS", f"ds_read_b64_tr_b16 {vr(VF[slot], 2)}, {v(VADDR[db & 3])} offset:{off}", ("v", m, 0)),
            ("LDS", f"ds_read_b64_tr_b16 {vr(VF[slot] + 2, 2)}, {v(VADDR[db & 3])} offset:{off + 8 * ROW}", ("v", m, 1))]


# PRICING ONLY (wrong results; profiles/r04_power_ceiling.md): every 32x32x16 MFMA replaced by TWO 16x16x32 MFMAs on the same operand
# registers (the same FLOPs, 4-register accumulators taken round-robin from the 16 of the original) - what the matrix pipe costs in the
# other shape, inside the real loop with its real operand data. `mfma16:qk` / `mfma16:pv` / `mfma16:both`.
MFMA16 = opt_val("mfma16", "")
MFMA16_OP = {"bf16": "v_mfma_f32_16x16x32_bf16", "f16": "v_mfma_f32_16x16x32_f16"}[DTYPE]


def mfma_qk(sset, j, qb):
    kb, ks = j // KS, j % KS
    d = S_(sset, kb, qb)
    if MFMA16 in ("qk", "both"):
        quads = [(2 * ks) % 4, (2 * ks + 1) % 4]
        return "\n".join(f"    {MFMA16_OP} {vr(d + 4 * qd, 4)}, {KFRAG(j)}, {ar(QA(qb, ks), 4)}, {'0' if ks < 2 else vr(d + 4 * qd, 4)}" for qd in quads)
    c = "0" if ks == 0 else vr(d, 16)
    return f"    {MFMA_OP} {vr(d, 16)}, {KFRAG(j)}, {ar(QA(qb, ks), 4)}, {c}"


def mfma_pv(sset, slot, m, qb):
    db, kk = m >> 2, m & 3
    pf = S_(sset, kk >> 1, qb) + 8 * (kk & 1)
    if MFMA16 in ("pv", "both"):
        o = O_(qb, db)
        return "\n".join(f"    {MFMA16_OP} {ar(o + 4 * qd, 4)}, {vr(VF[slot], 4)}, {vr(pf, 4)}, {ar(o + 4 * qd, 4)}" for qd in ((2 * kk) % 4, (2 * kk + 1) % 4))
    return f"    {MFMA_OP} {ar(O_(qb, db), 16)}, {vr(VF[slot], 4)}, {vr(pf, 4)}, {ar(O_(qb, db), 16)}"


# Row sums of the ROUNDED P by v_dot2c_f32_<type> (l += p0 * 1 + p1 * 1 on the packed 16-bit pair, fp32 accumulate): ONE instruction per
# pair of scores behind the pack instead of two v_add_f32 in front of it - 32 of the step's VALU instructions fewer. l is then the sum
# of exactly the weights the PV MFMA uses (O = sum P~ V / sum P~ is self-normalised); the LSE carries the rounding of P~ (RNE: unbiased,
# |LSE - exact| <= 2^-9 for bf16, 2^-12 for fp16, on a row of one dominant key; 2^-9 / sqrt(n) on n comparable keys). The reference
# sums the un-rounded fp32 P (softmax.h:275-296): this form changes the LSE's rounding, so it is a pricing option of the generator (`dotsum`, recorded as wrong_results=1), never a product body and not a flag of the C-ABI.
DOTSUM = "dotsum" in OPT
# `stamps` (tools/debug/body_stage_profile.py; results are WRONG by construction): wave 0 reads the shader clock at six points of an item -
# body start, Q loads issued, Q in registers, loop entry, loop exit, body end - and lanes 0..4 overwrite the first five LSE values of the
# item's q-tile with the five differences (as floats): what the body's fixed part is made of, per item, without a byte of C++ changed.
STAMPS = "stamps" in OPT
STAMP_REGS = [65, 66, 67, 81, 83, 86]           # S_FREE0..5; the clock lands in s[50:51] (S_T64B, unused by this body)


def stamp(k):
    if STAMPS:
        emit(f"s_memtime {sr(S_T64B)}")
        emit("s_waitcnt lgkmcnt(0)")
        emit(f"s_mov_b32 {s(STAMP_REGS[k])}, {s(S_T64B)}")


def stamps_out():
    if not STAMPS:
        return
    skip = new_label("nostamps")
    emit(f"s_cmp_eq_u32 {s(S_WAVE)}, 0")
    emit(f"s_cbranch_scc0 {skip}")
    emit(f"s_cmp_eq_u64 {sr(S_VB)}, 0")                     # the epilogue's LSE base (gen_epilogue.py S_LSEB): no LSE asked for
    emit(f"s_cbranch_scc1 {skip}")
    for i in range(5):
        emit(f"s_sub_u32 {s(STAMP_REGS[i])}, {s(STAMP_REGS[i + 1])}, {s(STAMP_REGS[i])}")
    emit("s_waitcnt vmcnt(0)")                               # the item's own LSE stores have left
    emit(f"v_mbcnt_lo_u32_b32 {v(T[7])}, -1, 0")
    emit(f"v_mbcnt_hi_u32_b32 {v(T[7])}, -1, {v(T[7])}")
    emit(f"v_mov_b32 {v(T[0])}, {s(STAMP_REGS[0])}")
    for i in range(1, 5):
        emit(f"v_mov_b32 {v(T[1])}, {s(STAMP_REGS[i])}")
        emit(f"v_cmp_eq_u32 vcc, {i}, {v(T[7])}")
        emit(f"v_cndmask_b32 {v(T[0])}, {v(T[0])}, {v(T[1])}, vcc")
    emit(f"v_cvt_f32_u32 {v(T[0])}, {v(T[0])}")
    emit(f"v_add_u32 {v(T[2])}, {s(S_QROW0)}, {v(T[7])}")
    emit(f"v_lshlrev_b32 {v(T[2])}, 2, {v(T[2])}")
    emit(f"v_add_co_u32 {v(T[4])}, vcc, {s(S_VB)}, {v(T[2])}")
    emit(f"v_mov_b32 {v(T[5])}, {s(S_VB + 1)}")
    emit(f"v_addc_co_u32 {v(T[5])}, vcc, 0, {v(T[5])}, vcc")
    emit("s_mov_b64 exec, 0x1f")
    emit(f"global_store_dword {vr(T[4], 2)}, {v(T[0])}, off")
    emit("s_mov_b64 exec, -1")
    label(skip)
DOT_OP = {"bf16": "v_dot2c_f32_bf16", "f16": "v_dot2c_f32_f16"}[DTYPE]
ONES_BITS = {"bf16": "0x3f803f80", "f16": "0x3c003c00"}[DTYPE]
PK = "pk" in OPT           # packed-fp32 VALU (v_pk_fma_f32 / v_pk_add_f32). MEASURED ANTI-LEVER beside MFMAs: -64 issue slots
                           # per step but +300 quad-cycles of issue stall (1163 vs 1316 TFLOP/s); kept for A/B only


def softmax_parts(sset, p):
    """Pair p (elements 2p, 2p+1 of the 32 per lane) of BOTH q-blocks: per q-block (fma(s), 2 exp, add(s), cvt).
    Packed form: ONE v_pk_fma_f32 (c from s[S_CC:S_CC+1], -m_ref*c broadcast from one half of v[NMS0:NMS1] by op_sel)
    and ONE v_pk_add_f32 into the (L0, L1) pair do the work of two fmas / two adds."""
    F, E, A, C = [], [], [], []
    for qb in range(NQB):
        e0 = 2 * p
        kb, r = e0 >> 4, e0 & 15
        r0 = S_(sset, kb, qb) + r
        r1 = r0 + 1
        dst = S_(sset, kb, qb) + 8 * (r >> 3) + ((r & 7) >> 1)
        ta, tb = T[8 + 2 * qb], T[9 + 2 * qb]
        if PK:
            sel = "op_sel_hi:[1,0,0]" if qb == 0 else "op_sel:[0,0,1] op_sel_hi:[1,0,1]"
            F.append([f"    v_pk_fma_f32 {vr(ta, 2)}, {vr(r0, 2)}, {sr(S_CC)}, {vr(NMS[0], 2)} {sel}"])
            A.append([f"    v_pk_add_f32 {vr(L0[qb], 2)}, {vr(L0[qb], 2)}, {vr(r0, 2)}"])
        else:
            F.append([f"    v_fma_f32 {v(ta)}, {v(r0)}, {s(S_C)}, {v(NMS[qb])}", f"    v_fma_f32 {v(tb)}, {v(r1)}, {s(S_C)}, {v(NMS[qb])}"])
            if "mfmasum" in OPT:     # pricing only (wrong results): row sums from the matrix pipe, see step()
                A.append([])
            elif DOTSUM:       # behind the pack (emitted after C by the stream): alternate the two accumulators to keep the chains short
                A.append([f"    {DOT_OP} {v(L0[qb] if (p & 1) == 0 else L1[qb])}, {s(S_ONES)}, {v(dst)}"])
            else:
                A.append([f"    v_add_f32 {v(L0[qb])}, {v(L0[qb])}, {v(r0)}", f"    v_add_f32 {v(L1[qb])}, {v(L1[qb])}, {v(r1)}"])
        E.append([f"    v_exp_f32 {v(r0)}, {v(ta)}", f"    v_exp_f32 {v(r1)}, {v(tb)}"])
        C.append([f"    {CVT_OP} {v(dst)}, {v(r0)}, {v(r1)}"])
    return F, E, A, C


def softmax_group(sset, p):
    if "nosoftmax" in OPT:
        return []
    F, E, A, C = softmax_parts(sset, p)
    return [op for part in ((F, E, C, A) if DOTSUM else (F, E, A, C)) for per_qb in part for op in per_qb]


def softmax_stream(sset, groups):
    """Software-pipelined softmax of several pair-groups. A transcendental occupies its unit for ~4 quad-cycles but only
    two issue slots, so exps are never adjacent: every v_exp of group g is followed by the fma of group g+1 (same temp,
    just consumed), and the add / cvt of group g-1."""
    if "nosoftmax" in OPT or not groups:
        return []
    if "expblock" in OPT:
        return [op for p in groups for op in softmax_group(sset, p)]
    parts = [softmax_parts(sset, p) for p in groups]
    o = [x for per_qb in parts[0][0] for x in per_qb]       # F(first), every q-block
    n = len(parts)
    for g in range(n):
        Fn = parts[g + 1][0] if g + 1 < n else None
        Ap, Cp = (parts[g - 1][2], parts[g - 1][3]) if g > 0 else (None, None)
        E = parts[g][1]
        for qb in range(NQB):
            # between / after the two exps of a q-block: the add(s) and the cvt of group g-1 (the cvt after BOTH of its
            # adds: it may overwrite r0 in place), and the fma(s) of group g+1 (their temporaries were just consumed)
            if DOTSUM:                                      # the dot reads the packed pair: behind the cvt of its group
                fill0 = []
                fill1 = (list(Fn[qb]) if Fn else []) + (list(Cp[qb]) if Cp else []) + (list(Ap[qb]) if Ap else [])
            else:
                fill0 = list(Ap[qb][:1]) if Ap else []
                fill1 = (list(Ap[qb][1:]) if Ap else []) + (list(Fn[qb]) if Fn else []) + (list(Cp[qb]) if Cp else [])
            if Fn and not PK:                               # scalar form: fma of temp a right after exp a
                fill0 = [Fn[qb][0]] + fill0
                fill1 = [x for x in fill1 if x is not Fn[qb][0]]
            o += [E[qb][0]] + fill0 + [E[qb][1]] + fill1
    for part in ((3, 2) if DOTSUM else (2, 3)):
        for qb in range(NQB):
            o += parts[-1][part][qb]
    return o


def row_max_ops(sset):
    """In-lane max of the 32 scores of each q-block into MLOC[qb] (two max3 chains each), interleaved over q-blocks."""
    if "norowmax" in OPT and W2:
        return []
    per = []
    for qb in range(NQB):
        regs = [S_(sset, 0, qb) + r for r in range(16)] + [S_(sset, 1, qb) + r for r in range(16)]
        ops = [f"    v_max_f32 {v(MLOC[qb])}, {v(regs[0])}, {v(regs[1])}", f"    v_max_f32 {v(MLOC2[qb])}, {v(regs[2])}, {v(regs[3])}"]
        rest = regs[4:]
        chains = [MLOC[qb], MLOC2[qb]]
        for n_, i in enumerate(range(0, len(rest), 2)):
            ch = chains[n_ & 1]
            ops.append(f"    v_max3_f32 {v(ch)}, {v(ch)}, {v(rest[i])}, {v(rest[i + 1])}")
        ops.append(f"    v_max_f32 {v(MLOC[qb])}, {v(MLOC[qb])}, {v(MLOC2[qb])}")
        per.append(ops)
    return [x for pair in zip(*per) for x in pair]


def stats_bookkeeping(flush_label, flush_back):
    """HALF, a(i + 1) = 0: what a step's statistics block does besides the statistics - the table pointer and the rotating vote bit."""
    return [f"    v_add_u32 {v(TABV)}, 16, {v(TABV)}", f"    s_lshl_b32 {s(S_BIT)}, {s(S_BIT)}, 1", f"    s_cbranch_scc0 {flush_label}", flush_back + ":"]


def stats_ops(rare_label, back_label, flush_label, flush_back, inval_label, inval_back):
    """Half-wave max exchange, skip vote (one bit per position, OR over both q-blocks), true running max, lazy-rescale test.
    The vote bit of the position rides in S_BIT (shifted left every step; when it falls off the word is flushed), so no
    position arithmetic is needed; the independent TABV increment sits in the two-wait-state shadow a VALU write needs before
    v_permlane32_swap reads it."""
    o = []
    a = o.append
    QBS = range(NQB)
    for qb in QBS:
        a(f"    v_mov_b32 {v(T[qb])}, {v(MLOC[qb])}")
    a(f"    v_add_u32 {v(TABV)}, 16, {v(TABV)}")
    if NQB == 1:
        a("    s_nop 0")                         # 2 wait states between the VALU write of T[0] and the swap that reads it
    for qb in QBS:
        a(f"    v_permlane32_swap_b32 {v(MLOC[qb])}, {v(T[qb])}")
    a("    s_nop 0")
    for qb in QBS:
        a(f"    v_max_f32 {v(MLOC[qb])}, {v(MLOC[qb])}, {v(T[qb])}")
    # the step past the end of the walk (i == n - 1: tile i + 1 does not exist, S_nxt came from a clamped duplicate) must not
    # touch the state: its row max becomes -inf (no vote, no new max) and -m_ref*c becomes -inf (the part of P(i+1) computed in
    # this phase, added to the row sums, is 0)
    if inval_label is not None:             # (HALF: positions past the end of the walk have a = 0, the step form without statistics)
        a(f"    s_cmp_eq_u32 {s(S_I)}, {s(S_NM1)}")
        a(f"    s_cbranch_scc1 {inval_label}")
        o.append(inval_back + ":")
    # vote: (m_loc - m_prev) * c > thr   (softmax.h:194), m_prev = the running max BEFORE this tile
    for qb in QBS:
        a(f"    v_sub_f32 {v(T[2 + qb])}, {v(MLOC[qb])}, {v(MTRUE[qb])}")
    for qb in QBS:
        a(f"    v_max_f32 {v(MTRUE[qb])}, {v(MTRUE[qb])}, {v(MLOC[qb])}")
    for qb in QBS:
        a(f"    v_mul_f32 {v(T[2 + qb])}, {s(S_C)}, {v(T[2 + qb])}")
    if NQB == 2:
        a(f"    v_cmp_gt_f32 {sr(S_T64)}, {v(T[2])}, {s(S_THR)}")
        a(f"    v_cmp_gt_f32 vcc, {v(T[3])}, {s(S_THR)}")
        a(f"    s_or_b64 vcc, vcc, {sr(S_T64)}")                          # SCC = some row of the wave voted "do"
    else:
        a(f"    v_cmp_gt_f32 vcc, {v(T[2])}, {s(S_THR)}")
        a("    s_cmp_lg_u64 vcc, 0")                                      # SCC = some row of the wave voted "do"
    a(f"    s_cselect_b32 {s(S_T0)}, {s(S_BIT)}, 0")
    a(f"    s_or_b32 {s(S_DOMASK)}, {s(S_DOMASK)}, {s(S_T0)}")
    # lazy rescale: m_true > m_ref + tau/c on any lane of either q-block -> rare block
    if NQB == 2:
        a(f"    v_cmp_gt_f32 {sr(S_T64)}, {v(MTRUE[0])}, {v(MTHR[0])}")
        a(f"    v_cmp_gt_f32 vcc, {v(MTRUE[1])}, {v(MTHR[1])}")
        a(f"    s_or_b64 vcc, vcc, {sr(S_T64)}")
    else:
        a(f"    v_cmp_gt_f32 vcc, {v(MTRUE[0])}, {v(MTHR[0])}")
    a(f"    s_cbranch_vccnz {rare_label}")
    o.append(back_label + ":")
    # next position's bit; when it falls off the 32-bit word (SCC = 0: result is zero) the word is complete: flush it
    a(f"    s_lshl_b32 {s(S_BIT)}, {s(S_BIT)}, 1")
    a(f"    s_cbranch_scc0 {flush_label}")
    o.append(flush_back + ":")
    return o


def rare_rescale_block(rare_label, back_label):
    """Out of line: m_ref follows m_true; alpha = exp2((m_ref_old - m_true)*c); l *= alpha; O rescale flagged."""
    label(rare_label)
    for qb in range(NQB):
        emit(f"v_sub_f32 {v(T[2 + qb])}, {v(MREF[qb])}, {v(MTRUE[qb])}")
    for qb in range(NQB):
        emit(f"v_mul_f32 {v(T[2 + qb])}, {s(S_C)}, {v(T[2 + qb])}")
    for qb in range(NQB):
        emit(f"v_exp_f32 {v(ALPHA[qb])}, {v(T[2 + qb])}")
    for qb in range(NQB):
        emit(f"v_mov_b32 {v(MREF[qb])}, {v(MTRUE[qb])}")
    for qb in range(NQB):
        emit(f"v_mul_f32 {v(NMS[qb])}, {s(S_NEGC)}, {v(MREF[qb])}")
        emit(f"v_add_f32 {v(MTHR[qb])}, {s(S_TAU)}, {v(MREF[qb])}")
    for qb in range(NQB):
        emit(f"v_mul_f32 {v(L0[qb])}, {v(L0[qb])}, {v(ALPHA[qb])}")
        emit(f"v_mul_f32 {v(L1[qb])}, {v(L1[qb])}, {v(ALPHA[qb])}")
    emit(f"s_mov_b32 {s(S_RESC)}, 1")
    emit(f"s_branch {back_label}")


def inval_block(lbl, back):
    """Out of line (last step of a walk): -m_ref*c := -inf, so exp2(S*c - inf) = 0 for the tile that does not exist."""
    label(lbl)
    for qb in range(NQB):
        emit(f"v_mov_b32 {v(MLOC[qb])}, {v(NEGINF)}")
        emit(f"v_mov_b32 {v(NMS[qb])}, {v(NEGINF)}")
    emit(f"s_branch {back}")


def flush_block(flush_label, back_label):
    """Out of line: doflags word |= domask by one lane; next word, domask = 0, bit = 1. Drains lgkmcnt (keeps counted waits valid)."""
    label(flush_label)
    flush_domask()
    emit(f"s_add_u32 {s(S_DOWORD)}, {s(S_DOWORD)}, 4")
    emit(f"s_mov_b32 {s(S_BIT)}, 1")
    if HALF:
        # the flush runs inside step i = 32 m + 30, before its end-of-step shift: the window holds a(i), a(i + 1) in bits 0..1 and the
        # NEXT activity word (positions 32 (m + 1) ...) goes to bits 2..33
        emit(f"v_mov_b32 {v(T[4])}, {s(S_ACTPTR)}")
        emit(f"ds_read_b32 {v(T[4])}, {v(T[4])}")
        emit(f"s_add_u32 {s(S_ACTPTR)}, {s(S_ACTPTR)}, 4")
        emit("s_waitcnt lgkmcnt(0)")
        emit(f"v_readfirstlane_b32 {s(S_T0)}, {v(T[4])}")
        emit(f"s_and_b32 {s(S_ACT)}, {s(S_ACT)}, 3")
        emit(f"s_lshl_b32 {s(S_T1)}, {s(S_T0)}, 2")
        emit(f"s_lshr_b32 {s(S_ACT + 1)}, {s(S_T0)}, 30")
        emit(f"s_or_b32 {s(S_ACT)}, {s(S_ACT)}, {s(S_T1)}")
    else:
        emit("s_waitcnt lgkmcnt(0)")
    emit(f"s_branch {back_label}")


def flush_domask():
    emit(f"v_mov_b32 {v(T[4])}, {s(S_DOWORD)}")
    emit(f"v_mov_b32 {v(T[5])}, {s(S_DOMASK)}")
    emit(f"s_mov_b64 {sr(S_EXEC)}, exec")
    emit("s_mov_b64 exec, 1")
    emit(f"ds_or_b32 {v(T[4])}, {v(T[5])}")
    emit(f"s_mov_b64 exec, {sr(S_EXEC)}")
    emit(f"s_mov_b32 {s(S_DOMASK)}, 0")


def rescale_o_block(lbl, back):
    """Out of line (rare): O^T *= alpha for both q-blocks (AGPR -> VGPR -> AGPR), after the PV MFMAs have drained."""
    label(lbl)
    emit("s_nop 15")
    emit("s_nop 15")
    for qb in range(NQB):
        for base in range(0, 16 * DB, 8):
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
    """[m0K, K0..K3, m0V, V0..V3]: one M0 per tensor, the piece index rides on the instruction offset (applied to both the
    global and the LDS address). The global side is compensated in the per-lane offsets: LK/LV[j] carry +(3072 - 1024*j) and
    the tensor bases in S_KBASE / S_VBASE carry -3072 (DMA_BIAS), so every per-lane offset is >= 0 — the VGPR of the SADDR form
    is an UNSIGNED 32-bit offset, and with plain -1024*j a row clamped to a short sequence (seqlen_k < 13 rows at one head)
    went negative = +4 GiB (found by the varlen tests: a 1-key sequence faulted)."""
    if "nodma" in OPT:
        return []
    o = []
    # head_dim 256: 8 pieces per tensor in two groups of 4 (the instruction offset is a 13-bit signed field: 0..3072 only), M0 moved
    # by 4 KiB for the second group; the lane offsets of piece j carry +(3072 - 1024 (j & 3))
    per = min(4, PW)                       # pieces per M0 group (head_dim 64: 2 pieces per tensor in all)
    for grp in range(PW // per):
        if do_k:
            o.append(f"    s_add_u32 m0, {s(S_DMAW)}, {kbuf_imm + 4096 * grp}")
            o += [f"    global_load_lds_dwordx4 {v(LK[4 * grp + j])}, {sr(TBS[st])} offset:{1024 * j}{DMA_POLICY}" for j in range(per)]
    for grp in range(PW // per):
        if do_v:
            o.append(f"    s_add_u32 m0, {s(S_DMAW)}, {V_REGION + vbuf_imm + 4096 * grp}")
            o += [f"    global_load_lds_dwordx4 {v(LV[4 * grp + j])}, {sr(VBS[st])} offset:{1024 * j}{DMA_POLICY}" for j in range(per)]
    if "dma2x" in OPT:                     # pricing only (HISTORY.md section 8, two 128-row workgroups per CU): every piece staged twice
        o = [x + "\n" + x if "global_load_lds" in x else x for x in o]
    return o


def weight(it):
    """Issue cost in quad-cycles as measured (PMC: SQ_ACTIVE_INST_VALU): a transcendental takes two slots, labels none."""
    if isinstance(it, str):
        if it.endswith(":"):
            return 0
        if "v_exp_f32" in it:
            return 2
    return 1


def n_fill(items):
    return sum(weight(it) for it in items)


def distribute(queue, post, start, cap):
    """Append the ops of `queue` (order kept) to post[start..NG-1], topping every gap up to `cap` fillers."""
    q = list(queue)
    if cap <= 0:
        total = sum(n_fill(post[t]) for t in range(start, NG)) + n_fill(q)
        cap = -(-total // (NG - start))
    for t in range(start, NG):
        while q and n_fill(post[t]) < cap:
            post[t].append(q.pop(0))
            while q and isinstance(q[0], str) and q[0].endswith(":"):      # a label sticks to the op before it
                post[t].append(q.pop(0))
    post[NG - 1] += q


def emit_gaps(pre, mf, post):
    """One phase: per gap the pre items, the MFMA, the fillers. Pricing variant mfma16 (two 16x16x32 MFMAs per gap): with `spread` the
    second MFMA of a gap sits behind the first half of the gap's fillers instead of right behind the first (a 16x16x32 MFMA holds the
    matrix pipe ~17 cycles: back to back, the second one stalls at issue)."""
    for t in range(NG):
        if MFMA16 and "spread" in OPT and "\n" in mf[t]:
            a, b = mf[t].split("\n")
            h = (len(post[t]) + 1) // 2
            for it in pre[t] + [a] + post[t][:h] + [b] + post[t][h:]:
                out.append(it)
        else:
            for it in pre[t] + [mf[t]] + post[t]:
                out.append(it)


deferred = []     # out-of-line blocks emitted after the loop: callables


def widen_last(n, since):
    """Code-placement experiments (profiles/r05_code_placement.md): shift everything BEHIND this point by 4 n bytes at no issue cost -
    the last n register-operand VOP1 / VOP2 instructions emitted since index `since` of `out` take their 8-byte VOP3 encoding."""
    import re
    pat = re.compile(r"^(\s*)(v_exp_f32|v_add_f32|v_max_f32|v_mul_f32|v_sub_f32)(\s+[vs]\d+(?:,\s*-?[vs]\d+)+\s*)$")
    i = len(out) - 1
    while n > 0 and i >= since:
        if isinstance(out[i], str) and pat.match(out[i]):
            out[i] = pat.sub(lambda m: m.group(1) + m.group(2) + "_e64" + m.group(3), out[i])
            n -= 1
        i -= 1
    assert n == 0, "not enough 4-byte instructions to widen"


HALFSKIP = int(opt_val("halfskip", "0"))     # PRICING ONLY: every wave sits out one step in HALFSKIP (no MFMA, no softmax, no fragment reads)


def step(variant, light=False, a_cur=True, a_nxt=True):
    if light:                             # the step of a wave whose 128-row half does not list this tile: it still stages its DMA pieces,
        saved = set(OPT)                  # follows the tile-address table and meets the barrier
        OPT.update({"nomfma1", "nomfma2", "nosoftmax", "norowmax", "novread", "nokread", "nowaitv"})
        try:
            return _step(variant)
        finally:
            OPT.clear()
            OPT.update(saved)
    return _step(variant, a_cur, a_nxt)


def _step(variant, a_cur=True, a_nxt=True):
    """One pipeline step; variant = parity of i: S_cur = S set `variant`, K(i+2)/V(i) in LDS buffer `variant`, DMA bases
    in SGPR set `variant` (computed during the previous step). Nothing but the drain, the barrier and the loop test sits
    between the last MFMA of a step and the first of the next.
    HALF: a_cur / a_nxt = this wave's half lists tile i / tile i + 1. Tile i's work of the step (rest of softmax(i), V^T reads, PV(i))
    is there iff a_cur; tile i + 1's (QK(i+1), row max / vote / running max, first part of softmax(i+1)) iff a_nxt. The DMA issue, the
    table reads, the vote-bit / table-pointer bookkeeping, the drain and the barrier are in every form. K(i+2) fragments: read beside
    the PV MFMAs when a_cur (whatever a(i+2): the gaps cannot branch), else in a block of their own iff a(i+2)."""
    cur, nxt = variant, variant ^ 1
    kbuf_read = cur * KV_TILE                # K(i+2)
    kbuf_stage = nxt * KV_TILE               # K(i+3) goes where K(i+1) was
    vbuf_cur = cur * KV_TILE                 # V(i)
    vbuf_stage = nxt * KV_TILE               # V(i+1)
    ord1 = [(f & 1) * KS + (f >> 1) for f in range(NKF)]     # K fragment order: alternate key blocks
    ord2 = [(f % DB) * 4 + (f // DB) for f in range(NVF)]    # V^T fragment order: kk outer, d-block inner
    full = a_cur and a_nxt

    # ---- phase 1: QK^T(i+1) || rest of softmax(i), DMA issue (V(i+1), K(i+3)), first V^T fragments
    pre = [[] for _ in range(NG)]
    post = [[] for _ in range(NG)]
    mf = []
    for t in range(NG):
        mf.append(mfma_qk(nxt, ord1[t // NQB], qb_of(t)) if ("nomfma1" not in OPT and a_nxt) else "    s_nop 0")
    for g, op in zip(DMA_GAPS, dma_ops(kbuf_stage, vbuf_stage, st=variant)):
        post[g].append(op)
    if K_EARLY and "nokread" not in OPT:
        for t in range(NG):
            if t % NQB == NQB - 1:                           # the last MFMA that reads fragment ord1[t // NQB] has issued
                post[t].append(k_read(kbuf_read, ord1[t // NQB]))
    if "novread" not in OPT and a_cur:
        for f in range(8):                                   # the first 8 V^T fragments, spread over the second half of the phase
            post[NG // 2 + f * (NG // 2) // 8] += v_read(f, vbuf_cur, ord2[f])
    vq = softmax_stream(cur, list(range(XPAIRS, 16))) if a_cur else []
    distribute(vq, post, 0, CAP1)
    mark = len(out)
    emit_gaps(pre, mf, post)
    if full:
        widen_last(int(opt_val("wp2", "0")) if variant == 0 else int(opt_val("wp2b", opt_val("wp2", "0"))), mark)      # shift phase 2 (copy 0 / copy 1 of the step)
    if a_nxt and not a_cur:
        # no PV MFMAs stand between the QK MFMAs and the first VALU read of their results (the row max): the MFMA results are
        # readable 18 wait states after the last one issued
        emit("s_nop 15")
        emit("s_nop 7")

    # ---- phase 2: PV(i) || K(i+2) fragments -> AGPRs, rest of the V^T fragments, next step's DMA bases,
    #               stats(i+1), first part of softmax(i+1)
    pre = [[] for _ in range(NG)]
    post = [[] for _ in range(NG)]
    mf = []
    for t in range(NG):
        f, qb = t // NQB, t % NQB
        if qb == 0 and "novread" not in OPT and "nowaitv" not in OPT and a_cur:
            pre[t].append(("WAIT", ("v", ord2[f], 1)))
        mf.append(mfma_pv(cur, f % 8, ord2[f], qb_of(t)) if ("nomfma2" not in OPT and a_cur) else "    s_nop 0")
        if qb == NQB - 1 and f + 8 < NVF and "novread" not in OPT and a_cur:
            post[t] += v_read(f % 8, vbuf_cur, ord2[f + 8])
        if "nokread" not in OPT and not K_EARLY and a_cur and (t < NKF if "klate" not in OPT else (t & 1) == 0):
            post[t].append(k_read(kbuf_read, ord1[t if "klate" not in OPT else f]))
    if "mfmasum" in OPT:
        # PRICING ONLY: the 64 v_add_f32 of the row sums are gone and one more MFMA per (q-block, 16-key group)
        # stands for "ones x P^T" - 8 per step. No 32 accumulator registers are free in this register map, so the stand-in
        # accumulates into O^T (finite garbage) with a Q fragment as its A operand (real data: prices the energy conservatively).
        for kk in range(4):
            for qb in range(NQB):
                pf = S_(cur, kk >> 1, qb) + 8 * (kk & 1)
                post[4 * kk + 2 * qb + 17].append(f"    {MFMA_OP} {ar(O_(qb, 0), 16)}, {ar(QA(qb, 0), 4)}, {vr(pf, 4)}, {ar(O_(qb, 0), 16)}")
    rare, back = new_label("rare"), new_label("rare_back")
    fl, flback = new_label("flush"), new_label("flush_back")
    # next step (i+1) stages K(i+4) and V(i+2): their global addresses come from the tile-address table the C++ shell built in
    # LDS (tab[pos] = {K address, V address} of the tile at walk position pos, rows clamped, DMA_BIAS applied, padded by 4
    # copies of the last entry): TABV = &tab[i+2]. No tile lookup, no clamp, no 64-bit multiply on the scalar unit.
    st2 = variant ^ 1
    vq = [("LDS", f"ds_read_b64 {vr(T[4], 2)}, {v(TABV)} offset:8", "tabv"),
          ("LDS", f"ds_read_b64 {vr(T[6], 2)}, {v(TABV)} offset:32", "tabk")]
    n_head = len(vq)
    if "norowmax" not in OPT and a_nxt:
        rm = row_max_ops(nxt)
    else:
        rm = []
    vq += rm[:8]
    rm = rm[8:]
    vq += [("WAIT", "tabk")]
    nb = [f"    v_readfirstlane_b32 {s(VBS[st2])}, {v(T[4])}", f"    v_readfirstlane_b32 {s(VBS[st2] + 1)}, {v(T[5])}",
          f"    v_readfirstlane_b32 {s(TBS[st2])}, {v(T[6])}", f"    v_readfirstlane_b32 {s(TBS[st2] + 1)}, {v(T[7])}"]
    mixed = []
    while rm or nb:
        if rm:
            mixed += rm[:2]
            rm = rm[2:]
        if nb:
            mixed.append(nb.pop(0))
    vq += mixed
    if not a_nxt:
        vq += stats_bookkeeping(fl, flback)
        deferred.append(lambda: flush_block(fl, flback))
    elif "notail" not in OPT:
        # (HALF: positions past the end of the walk have a = 0, so a form with statistics never runs at i == n - 1 and the test below never
        # fires there; it stays so that the hot loop is instruction for instruction - gap for gap - the tuned schedule of the form
        # without activity bits)
        inv, invback = new_label("inval"), new_label("inval_back")
        vq += stats_ops(rare, back, fl, flback, inv, invback)
        deferred.append(lambda: inval_block(inv, invback))
        deferred.append(lambda: rare_rescale_block(rare, back))
        deferred.append(lambda: flush_block(fl, flback))
    if a_nxt:
        vq += softmax_stream(nxt, list(range(XPAIRS)))
    # the first two gaps may only hold ops that do not read S_nxt (MFMA -> VALU read hazard): the SALU / seq part
    distribute(vq[:n_head], post, 0, CAP2 if CAP2 > 0 else 6)
    distribute(vq[n_head:], post, 2, CAP2)
    if HALF and a_cur:
        # the loop count and the form of the NEXT step: ADDED to the PV gaps behind the statistics (which read S_I and may refill the
        # window) - on top of the tuned filling, which stays as it is. In front of the drain they were a dependent scalar chain on the
        # critical path: +30 cycles per step, measured.
        t_stats = max(t for t in range(NG) if any(isinstance(it, str) and it.startswith(flback) for it in post[t]))
        ops = state_ops()
        groups = [ops[0:1], ops[1:2], ops[2:3], ops[3:]]           # (the compare and the select that reads its SCC: adjacent)
        for g_, grp in enumerate(groups):                           # one group per gap behind the statistics; what does not fit: the last gap
            post[min(t_stats + 1 + g_, NG - 1)] += grp
    mark = len(out)
    emit_gaps(pre, mf, post)
    if variant == 0 and full:
        # shift the second copy of the step against the first (code-placement experiments). HALF: a step is four scalar instructions (16
        # bytes: five in the PV gaps, one fewer behind the barrier) longer than the tuned form's; four widened encodings more put the second
        # copy of the step back at the fetch phase it was tuned at (profiles/r05_code_placement.md: period 32 bytes)
        widen_last(int(opt_val("wc2", "4" if HALF else "0")), mark)
    if HALF and not a_cur and "nokread" not in OPT:
        # K(i+2) fragments iff this half lists tile i + 2 (bit 2 of the window). Behind every counted LDS wait of the step and in front
        # of the drain: a conditional LDS operation between a counted wait and its target would break the count.
        nok = new_label("nokfrag")
        emit(f"s_bitcmp1_b32 {s(S_ACT)}, 2")
        emit(f"s_cbranch_scc0 {nok}")
        for j in range(NKF):
            emit(k_read(kbuf_read, ord1[j]))
        label(nok)

    # ---- tail: the rare O rescale, drain, barrier
    resc, resc_back = new_label("resc"), new_label("resc_back")
    emit(f"s_cmp_lg_u32 {s(S_RESC)}, 0")
    emit(f"s_cbranch_scc1 {resc}")
    label(resc_back)
    deferred.append(lambda: rescale_o_block(resc, resc_back))
    if HALF and not a_cur:
        # (the forms without PV: behind the conditional K-fragment block, which reads bit 2 of the window)
        for op in state_ops():
            out.append(op)
    emit(("DRAIN",))
    if "nobarrier" not in OPT:
        emit("s_barrier")
    if not HALF:
        emit(f"s_add_u32 {s(S_I)}, {s(S_I)}, 1")


def state_ops():
    """HALF: i += 1, the window moves on, S_STATE = the form of the next step - (a(i), a(i+1)) as bits 0, 1, or 4 when the walk is over.
    Behind the barrier a step then starts with one compare and one branch, like the form without activity bits."""
    return [f"    s_add_u32 {s(S_I)}, {s(S_I)}, 1", f"    s_lshr_b64 {sr(S_ACT)}, {sr(S_ACT)}, 1", f"    s_and_b32 {s(S_STATE)}, {s(S_ACT)}, 3",
            f"    s_cmp_lt_u32 {s(S_I)}, {s(S_NTILES)}", f"    s_cselect_b32 {s(S_STATE)}, {s(S_STATE)}, 4"]


def next_state(shift=True):
    if shift:
        emit(f"s_lshr_b64 {sr(S_ACT)}, {sr(S_ACT)}, 1")
    emit(f"s_and_b32 {s(S_STATE)}, {s(S_ACT)}, 3")
    emit(f"s_cmp_lt_u32 {s(S_I)}, {s(S_NTILES)}")
    emit(f"s_cselect_b32 {s(S_STATE)}, {s(S_STATE)}, 4")


def mask_first_tile_ops():
    """seqlen-k mask of the first walked tile (mask.h:44-78; mainloop...:1626): columns >= tail_valid -> -inf."""
    for kb in range(2):
        for r in range(16):
            key = 32 * kb + (r & 3) + 8 * (r >> 2)
            emit(f"v_add_u32 {v(T[0])}, {key}, {v(HH4)}")
            emit(f"v_cmp_gt_i32 vcc, {s(S_TAILVALID)}, {v(T[0])}")            # key < tail_valid -> keep
            for qb in range(NQB):
                emit(f"v_cndmask_b32 {v(S_(0, kb, qb) + r)}, {v(NEGINF)}, {v(S_(0, kb, qb) + r)}, vcc")


def w2_top_fillers(p):
    """What every step does first, as (slot, item) pairs to be interleaved with the step's first run of instructions: this step's DMA
    addresses from the tile-address table -> SGPRs, K(i+2) -> K buffer p, V(i+1) -> V buffer 1-p (both buffers were last read in step
    i-1, behind its barrier), the first V^T fragments of tile i, TABV -> tab[i+2]."""
    kbuf_stage, vbuf_cur, vbuf_stage = p * KV_TILE, p * KV_TILE, (p ^ 1) * KV_TILE
    ord2 = [(f % DB) * 4 + (f // DB) for f in range(NVF)]
    f = [(0, ("LDS", f"ds_read_b64 {vr(T[4], 2)}, {v(TABV)} offset:8", "tabv")),
         (0, ("LDS", f"ds_read_b64 {vr(T[6], 2)}, {v(TABV)} offset:16", "tabk")),
         (2, ("WAIT", "tabk")),
         (2, f"    v_readfirstlane_b32 {s(VBS[0])}, {v(T[4])}"), (2, f"    v_readfirstlane_b32 {s(VBS[0] + 1)}, {v(T[5])}"),
         (2, f"    v_readfirstlane_b32 {s(TBS[0])}, {v(T[6])}"), (2, f"    v_readfirstlane_b32 {s(TBS[0] + 1)}, {v(T[7])}"),
         (2, f"    v_add_u32 {v(TABV)}, 16, {v(TABV)}")]
    for g_, op in zip(range(3, 64), dma_ops(kbuf_stage, vbuf_stage, st=0)):     # >= 5 wait states behind the v_readfirstlane of their bases
        f.append((g_, op))
    if "novread" not in OPT:
        for k in range(len(VF)):
            for it in v_read(k, vbuf_cur, ord2[k]):
                f.append((4 + k, it))
    return f


def w2_interleave(main, fillers, per_slot=1):
    """main: instruction list; fillers: (slot, item) pairs; slot k = behind the (k * per_slot)-th main instruction (slot 0: in front)."""
    by = {}
    for slot, it in fillers:
        by.setdefault(slot, []).append(it)
    for it in by.pop(0, []):
        out.append(it)
    n_slots = max(by) if by else 0
    if n_slots * per_slot > len(main):              # ablation variants (no softmax): pad with no-ops
        main = list(main) + ["    s_nop 0"] * (n_slots * per_slot - len(main))
    for i, op in enumerate(main):
        out.append(op)
        if (i + 1) % per_slot == 0:
            for it in by.pop((i + 1) // per_slot, []):
                out.append(it)
    assert not by


def w2_qk(fillers=()):
    ord1 = [(f & 1) * KS + (f >> 1) for f in range(NKF)]
    w2_interleave([mfma_qk(0, ord1[t // NQB], t % NQB) for t in range(NG)], fillers)
    emit("s_nop 15")                                 # the MFMA results are readable 18 wait states after the last MFMA issued
    emit("s_nop 7")


def w2_stats(first_tile, can_be_first):
    """Row max, vote (position S_I, or S_I + 1 for the waves that run one QK ahead: the bit bookkeeping only counts calls), running
    max, lazy-rescale decision. first_tile: this IS the first walked tile (no test); can_be_first: test S_I == 0 at run time."""
    first, first_back = new_label("w2first"), new_label("w2first_back")
    rare, back = new_label("w2rare"), new_label("w2rare_back")
    fl, flback = new_label("w2flush"), new_label("w2flush_back")

    def mask_block(msk, msk_back):
        label(msk)
        emit(f"s_cmp_eq_u32 {s(S_FIRSTLAST)}, 1")             # the first walked tile is tile k_tiles - 1 (C++ shell)
        emit(f"s_cbranch_scc0 {msk_back}")
        emit(f"s_cmp_lt_i32 {s(S_TAILVALID)}, 64")
        emit(f"s_cbranch_scc0 {msk_back}")
        mask_first_tile_ops()
        emit(f"s_branch {msk_back}")
    if first_tile or can_be_first:
        msk, msk_back = new_label("w2mask"), new_label("w2mask_back")
        if can_be_first:
            emit(f"s_cmp_eq_u32 {s(S_I)}, 0")
            emit(f"s_cbranch_scc1 {msk}")
        else:
            emit(f"s_branch {msk}")
        label(msk_back)
        deferred.append(lambda msk=msk, msk_back=msk_back: mask_block(msk, msk_back))
    for op in row_max_ops(0):
        out.append(op)
    for qb in range(NQB):
        emit(f"v_mov_b32 {v(T[qb])}, {v(MLOC[qb])}")
    emit("s_nop 1")
    for qb in range(NQB):
        emit(f"v_permlane32_swap_b32 {v(MLOC[qb])}, {v(T[qb])}")
    emit("s_nop 0")
    for qb in range(NQB):
        emit(f"v_max_f32 {v(MLOC[qb])}, {v(MLOC[qb])}, {v(T[qb])}")

    def first_ops():
        # first walked tile: m_true = m_ref = its row max; position 0 is never flagged (softmax.h:153)
        for qb in range(NQB):
            emit(f"v_mov_b32 {v(MTRUE[qb])}, {v(MLOC[qb])}")
            emit(f"v_mov_b32 {v(MREF[qb])}, {v(MLOC[qb])}")
            emit(f"v_mul_f32 {v(NMS[qb])}, {s(S_NEGC)}, {v(MLOC[qb])}")
            emit(f"v_add_f32 {v(MTHR[qb])}, {s(S_TAU)}, {v(MLOC[qb])}")
        emit(f"s_or_b32 {s(S_DOMASK)}, {s(S_DOMASK)}, {s(S_BIT)}")
    if first_tile:
        first_ops()
    else:
        if can_be_first:
            emit(f"s_cmp_eq_u32 {s(S_I)}, 0")
            emit(f"s_cbranch_scc1 {first}")
            deferred.append(lambda: (label(first), first_ops(), emit(f"s_branch {first_back}")))
        # vote: (m_loc - m_prev) * c > thr   (softmax.h:194), m_prev = the running max BEFORE this tile
        for qb in range(NQB):
            emit(f"v_sub_f32 {v(T[2 + qb])}, {v(MLOC[qb])}, {v(MTRUE[qb])}")
        for qb in range(NQB):
            emit(f"v_max_f32 {v(MTRUE[qb])}, {v(MTRUE[qb])}, {v(MLOC[qb])}")
        for qb in range(NQB):
            emit(f"v_mul_f32 {v(T[2 + qb])}, {s(S_C)}, {v(T[2 + qb])}")
        assert NQB == 1
        emit(f"v_cmp_gt_f32 vcc, {v(T[2])}, {s(S_THR)}")
        emit("s_cmp_lg_u64 vcc, 0")                                  # SCC = some row of the wave voted "do"
        emit(f"s_cselect_b32 {s(S_T0)}, {s(S_BIT)}, 0")
        emit(f"s_or_b32 {s(S_DOMASK)}, {s(S_DOMASK)}, {s(S_T0)}")
        emit(f"v_cmp_gt_f32 vcc, {v(MTRUE[0])}, {v(MTHR[0])}")       # lazy rescale: m_true > m_ref + tau / c on some lane
        emit(f"s_cbranch_vccnz {rare}")
        label(back)
        if can_be_first:
            label(first_back)

        def rare_block(rare=rare, back=back):
            # m_ref follows m_true; alpha = exp2((m_ref_old - m_true) c); l and O^T (the tiles before this one) *= alpha, here and now:
            # this tile's PV comes later in program order, and the PV MFMAs of the previous tile have long drained
            label(rare)
            emit("s_nop 15")
            emit("s_nop 15")
            for qb in range(NQB):
                emit(f"v_sub_f32 {v(T[2 + qb])}, {v(MREF[qb])}, {v(MTRUE[qb])}")
            for qb in range(NQB):
                emit(f"v_mul_f32 {v(T[2 + qb])}, {s(S_C)}, {v(T[2 + qb])}")
            for qb in range(NQB):
                emit(f"v_exp_f32 {v(ALPHA[qb])}, {v(T[2 + qb])}")
            for qb in range(NQB):
                emit(f"v_mov_b32 {v(MREF[qb])}, {v(MTRUE[qb])}")
            for qb in range(NQB):
                emit(f"v_mul_f32 {v(NMS[qb])}, {s(S_NEGC)}, {v(MREF[qb])}")
                emit(f"v_add_f32 {v(MTHR[qb])}, {s(S_TAU)}, {v(MREF[qb])}")
            for qb in range(NQB):
                emit(f"v_mul_f32 {v(L0[qb])}, {v(L0[qb])}, {v(ALPHA[qb])}")
                emit(f"v_mul_f32 {v(L1[qb])}, {v(L1[qb])}, {v(ALPHA[qb])}")
            for qb in range(NQB):
                for base in range(0, 16 * DB, 8):
                    for k in range(8):
                        emit(f"v_accvgpr_read_b32 {v(T[k])}, a{O_(qb, 0) + base + k}")
                    for k in range(8):
                        emit(f"v_mul_f32 {v(T[k])}, {v(T[k])}, {v(ALPHA[qb])}")
                    for k in range(8):
                        emit(f"v_accvgpr_write_b32 a{O_(qb, 0) + base + k}, {v(T[k])}")
            emit("s_nop 7")
            emit(f"s_branch {back}")
        deferred.append(rare_block)
    emit(f"s_lshl_b32 {s(S_BIT)}, {s(S_BIT)}, 1")
    emit(f"s_cbranch_scc0 {fl}")
    label(flback)
    deferred.append(lambda: flush_block(fl, flback))


def w2_pv(p):
    """O^T += V(i)^T P^T; beside the MFMAs: the K(i+1) fragments -> AGPRs (the QK of tile i has consumed K(i)'s), the rest of V^T."""
    kbuf_next, vbuf_cur = (p ^ 1) * KV_TILE, p * KV_TILE
    ord1 = [(f & 1) * KS + (f >> 1) for f in range(NKF)]
    ord2 = [(f % DB) * 4 + (f // DB) for f in range(NVF)]
    NS = len(VF)
    emit("s_nop 4")                                  # the last pack of P -> its first MFMA read
    for t in range(NG):
        f, qb = t // NQB, t % NQB
        if qb == 0 and "novread" not in OPT:
            out.append(("WAIT", ("v", ord2[f], 1)))
        out.append(mfma_pv(0, f % NS, ord2[f], qb) if "nomfma2" not in OPT else "    s_nop 0")
        if qb == NQB - 1 and f + NS < NVF and "novread" not in OPT:
            for it in v_read(f % NS, vbuf_cur, ord2[f + NS]):
                out.append(it)
        if t < NKF and "nokread" not in OPT:
            out.append(k_read(kbuf_next, ord1[t]))


def w2_tail():
    emit(("DRAIN",))
    if "nobarrier" not in OPT:
        emit("s_barrier")
    emit(f"s_add_u32 {s(S_I)}, {s(S_I)}, 1")


def step_w2(p, group):
    """W2: one step. p = parity of i = S_I: K(i) came from K buffer p, V(i) is in V buffer p, K(i+1) in K buffer 1-p.
    Waves 0-3 (group "a") run tile i in program order: QK(i) -> statistics(i) -> softmax(i) -> PV(i). Waves 4-7 (group "b") share their
    SIMDs with waves 0-3 and the ONE barrier of the workgroup, so in the same rotation both waves of a SIMD would want the matrix pipe,
    then the vector unit, then the matrix pipe at the same moments; they run the rotation softmax(i) -> PV(i) -> QK(i+1) ->
    statistics(i+1) instead (QK(0) and its statistics in front of their loop), so that one wave's vector work lies under the other's
    MFMAs. The LDS buffer protocol is the same for both groups."""
    if group == "a" or "norot" in OPT:          # norot (A/B): both groups in the same rotation
        w2_qk(w2_top_fillers(p))
        w2_stats(first_tile=False, can_be_first=(p == 0))       # i == 0 is even: only this copy of the step can be the first one
        for op in softmax_stream(0, list(range(16))):
            out.append(op)
        w2_pv(p)
    else:
        w2_interleave(softmax_stream(0, list(range(16))), w2_top_fillers(p), per_slot=6)
        w2_pv(p)
        skip = new_label("w2lastqk")
        emit(f"s_cmp_eq_u32 {s(S_I)}, {s(S_NM1)}")             # the last tile has no successor
        emit(f"s_cbranch_scc1 {skip}")
        emit("s_waitcnt lgkmcnt(0)")                         # the K(i+1) fragments
        w2_qk()
        w2_stats(first_tile=False, can_be_first=False)
        label(skip)
    w2_tail()


def prologue():
    stamp(0)
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
    if HALF:
        plist.append(S_HSTRIDE)            # [23]: bytes between the two halves' flag blocks; [19] (S_EXPORT) = LDS address of half 0's activity words
    for idx, sg in enumerate(plist):
        emit(f"v_readfirstlane_b32 {s(sg)}, {v(idx)}")
    emit("s_nop 4")
    if HALF:
        emit("; ---- half = wave >> 1: its vote words, its activity words; the window starts as positions [0, 64)")
        emit(f"s_lshr_b32 {s(S_T0)}, {s(S_WAVE)}, 1")
        emit(f"s_mul_i32 {s(S_T0)}, {s(S_T0)}, {s(S_HSTRIDE)}")
        emit(f"s_add_u32 {s(S_DOFLAGS)}, {s(S_DOFLAGS)}, {s(S_T0)}")
        emit(f"s_add_u32 {s(S_ACTPTR)}, {s(S_EXPORT)}, {s(S_T0)}")
        emit(f"v_mov_b32 {v(T[0])}, {s(S_ACTPTR)}")
        emit(f"ds_read_b64 {vr(T[4], 2)}, {v(T[0])}")
        emit("s_waitcnt lgkmcnt(0)")
        emit(f"v_readfirstlane_b32 {s(S_ACT)}, {v(T[4])}")
        emit(f"v_readfirstlane_b32 {s(S_ACT + 1)}, {v(T[5])}")
        emit(f"s_add_u32 {s(S_ACTPTR)}, {s(S_ACTPTR)}, 4")           # the first flush (step 30) brings word 1 to bits 2..33
    emit(f"s_mov_b32 {s(S_CC)}, {s(S_C)}")
    emit(f"s_mov_b32 {s(S_CC + 1)}, {s(S_C)}")
    emit(f"s_sub_u32 {s(S_NM1)}, {s(S_NTILES)}, 1")
    emit(f"s_lshl_b32 {s(S_DMAW)}, {s(S_WAVE)}, {ROW_SHIFT + (3 if W2 else 4)}")      # a wave stages 16 rows = 2 / 4 / 8 KiB of a tile (w2: 8 rows)
    emit(f"s_add_u32 {s(S_DMAW)}, {s(S_DMAW)}, {s(S_LDS)}")
    emit(f"s_mov_b32 {s(S_I)}, 0")
    if DOTSUM:
        emit(f"s_mov_b32 {s(S_ONES)}, {ONES_BITS}")
    emit(f"s_mov_b32 {s(S_RESC)}, 0")
    emit(f"v_mov_b32 {v(NEGINF)}, 0xff800000")

    emit("; ---- per-lane constants")
    emit(f"v_lshrrev_b32 {v(T[0])}, 5, {v(LANE)}")            # hh
    emit(f"v_lshlrev_b32 {v(HH4)}, 2, {v(T[0])}")
    emit(f"v_and_b32 {v(T[1])}, 31, {v(LANE)}")               # l31
    emit(f"v_and_b32 {v(T[2])}, 15, {v(LANE)}")               # a16 / cpos
    emit(f"v_lshlrev_b32 {v(T[3])}, {ROW_SHIFT}, {v(T[1])}")            # l31 * ROW
    emit(f"v_add_u32 {v(T[3])}, {s(S_LDS)}, {v(T[3])}")
    if DL == 64:                                               # 128-byte rows: the K chunk swizzle is (row >> 1) & 7
        emit(f"v_lshrrev_b32 {v(T[8])}, 1, {v(T[1])}")
        emit(f"v_and_b32 {v(T[8])}, 7, {v(T[8])}")
    for ks in range(4 if DL == 64 else 8):
        emit(f"v_add_u32 {v(T[4])}, {2 * ks}, {v(T[0])}")
        emit(f"v_xor_b32 {v(T[4])}, {v(T[4])}, {v(T[8] if DL == 64 else T[2])}")
        emit(f"v_lshl_add_u32 {v(KADDR[ks])}, {v(T[4])}, 4, {v(T[3])}")
    emit(f"v_lshrrev_b32 {v(T[4])}, 2, {v(T[2])}")            # kq = a16 >> 2
    emit(f"v_add_u32 {v(T[5])}, {v(HH4)}, {v(T[4])}")         # key0
    emit(f"v_lshlrev_b32 {v(T[5])}, {ROW_SHIFT}, {v(T[5])}")
    emit(f"v_add_u32 {v(T[5])}, {s(S_LDS)}, {v(T[5])}")
    emit(f"v_add_u32 {v(T[5])}, {V_REGION}, {v(T[5])}")
    emit(f"v_lshrrev_b32 {v(T[6])}, 4, {v(LANE)}")            # g = lane >> 4 = rip
    emit(f"v_and_b32 {v(T[7])}, 1, {v(T[6])}")
    emit(f"v_lshlrev_b32 {v(T[7])}, 5, {v(T[7])}")
    emit(f"v_and_b32 {v(T[8])}, 3, {v(T[2])}")                # a3
    emit(f"v_lshl_or_b32 {v(T[7])}, {v(T[8])}, 3, {v(T[7])}")
    emit(f"v_add_u32 {v(T[5])}, {v(T[5])}, {v(T[7])}")
    if DL == 64:                                               # 128-byte rows: two 64-byte segments, swizzled by (row >> 1) & 1 = kq >> 1
        emit(f"v_lshrrev_b32 {v(T[4])}, 1, {v(T[4])}")
    for db in range(2 if DL == 64 else 4):
        emit(f"v_xor_b32 {v(T[7])}, {db}, {v(T[4])}")
        emit(f"v_lshl_add_u32 {v(VADDR[db])}, {v(T[7])}, 6, {v(T[5])}")
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, {3 if W2 else 4}")
    if DL == 64:
        # DMA image: a 1-KiB piece = 8 rows of 128 bytes; lane -> (rip = lane >> 3, cpos = lane & 7). Row = 16 w + 8 j + rip: the K
        # swizzle XORs the chunk with (row >> 1) & 7 = (4 j + (rip >> 1)) & 7, the V swizzle with ((row >> 1) & 1) << 2 = ((rip >> 1) & 1) << 2
        emit(f"v_lshrrev_b32 {v(T[6])}, 3, {v(LANE)}")            # rip
        emit(f"v_and_b32 {v(T[7])}, 7, {v(LANE)}")                # cpos
        emit(f"v_add_u32 {v(RIPROW)}, {s(S_T0)}, {v(T[6])}")      # 16*wave + rip
        emit(f"v_lshrrev_b32 {v(T[8])}, 1, {v(T[6])}")            # rip >> 1
        emit(f"v_xor_b32 {v(RAGK)}, {v(T[7])}, {v(T[8])}")
        emit(f"v_lshlrev_b32 {v(RAGK)}, 4, {v(RAGK)}")            # (cpos ^ (rip >> 1)) << 4
        if W2:                                                     # rows 8 w + rip: (row >> 1) & 7 has the term 4 (w & 1) as well
            emit(f"s_and_b32 {s(S_T2)}, {s(S_WAVE)}, 1")
            emit(f"s_lshl_b32 {s(S_T2)}, {s(S_T2)}, 6")
            emit(f"v_xor_b32 {v(RAGK)}, {s(S_T2)}, {v(RAGK)}")
        emit(f"v_and_b32 {v(T[8])}, 1, {v(T[8])}")
        emit(f"v_lshlrev_b32 {v(T[8])}, 2, {v(T[8])}")
        emit(f"v_xor_b32 {v(RAGV)}, {v(T[7])}, {v(T[8])}")
        emit(f"v_lshlrev_b32 {v(RAGV)}, 4, {v(RAGV)}")            # (cpos ^ (((rip >> 1) & 1) << 2)) << 4
    elif DL == 128:
        # DMA image: a 1-KiB piece = 4 rows of 256 bytes; lane -> (row in piece rip = lane >> 4, chunk cpos = lane & 15)
        emit(f"v_add_u32 {v(RIPROW)}, {s(S_T0)}, {v(T[6])}")      # 16*wave + rip
        emit(f"v_xor_b32 {v(RAGK)}, {v(T[2])}, {v(T[6])}")
        emit(f"v_lshlrev_b32 {v(RAGK)}, 4, {v(RAGK)}")            # (cpos ^ rip) << 4
        emit(f"v_lshlrev_b32 {v(RAGV)}, 2, {v(T[6])}")
        emit(f"v_xor_b32 {v(RAGV)}, {v(T[2])}, {v(RAGV)}")
        emit(f"v_lshlrev_b32 {v(RAGV)}, 4, {v(RAGV)}")            # (cpos ^ (rip<<2)) << 4
    else:
        # head_dim 256: a piece = 2 rows of 512 bytes; lane -> (rip = lane >> 5, cpos = lane & 31). The K swizzle XORs the chunk with
        # row & 15 = 2 j + rip, the V swizzle with (row & 3) << 2 = (2 (j & 1) + rip) << 2: both stay inside bits 0..3 of cpos
        emit(f"v_add_u32 {v(RIPROW)}, {s(S_T0)}, {v(T[0])}")      # 16*wave + rip   (T[0] = lane >> 5)
        emit(f"v_xor_b32 {v(RAGK)}, {v(T[1])}, {v(T[0])}")
        emit(f"v_lshlrev_b32 {v(RAGK)}, 4, {v(RAGK)}")            # (cpos ^ rip) << 4   (T[1] = lane & 31)
        emit(f"v_lshlrev_b32 {v(RAGV)}, 2, {v(T[0])}")
        emit(f"v_xor_b32 {v(RAGV)}, {v(T[1])}, {v(RAGV)}")
        emit(f"v_lshlrev_b32 {v(RAGV)}, 4, {v(RAGV)}")            # (cpos ^ (rip<<2)) << 4
    emit(f"s_mov_b32 {s(S_T1)}, {s(S_LASTROW)}")              # seqlen_k < 64: rows of the only tile stay inside the tensor
    RSTEP = (64 // NW) // PW                                   # rows per piece
    for j in range(PW):
        # LK[j] = (16w + RSTEP j + rip)*k_rs + (RAGK ^ ((RSTEP j) << 4)) + 3072 - 1024 (j & 3); LV[j] likewise with the V swizzle
        bias = DMA_BIAS - 1024 * (j & 3)
        emit(f"v_add_u32 {v(T[4])}, {RSTEP * j}, {v(RIPROW)}")
        emit(f"v_min_i32 {v(T[4])}, {v(T[4])}, {s(S_T1)}")
        emit(f"v_mul_lo_u32 {v(LK[j])}, {v(T[4])}, {s(S_KRS)}")
        def in_row(reg):
            """head dims 96 / 192: a source chunk past the row's 2 D bytes (the LDS image is 2 DL bytes wide) -> 64 bytes lower: a
            duplicate of a valid chunk, written to an LDS position no fragment read ever touches"""
            if D != DL:
                emit(f"v_subrev_u32 {v(T[7])}, {2 * DL - 2 * D}, {v(reg)}")
                emit(f"v_cmp_le_u32 vcc, {2 * D}, {v(reg)}")
                emit(f"v_cndmask_b32 {v(reg)}, {v(reg)}, {v(T[7])}, vcc")
        emit(f"v_xor_b32 {v(T[5])}, {((RSTEP * j) >> (1 if DL == 64 else 0)) << 4}, {v(RAGK)}")
        in_row(T[5])
        emit(f"v_add_u32 {v(LK[j])}, {v(LK[j])}, {v(T[5])}")
        emit(f"v_mul_lo_u32 {v(LV[j])}, {v(T[4])}, {s(S_VRS)}")
        if D == 128 or D == 64:
            emit(f"v_add_u32 {v(LV[j])}, {v(LV[j])}, {v(RAGV)}")
        else:
            emit(f"v_xor_b32 {v(T[5])}, {((RSTEP * j) & 3) << 6}, {v(RAGV)}")
            in_row(T[5])
            emit(f"v_add_u32 {v(LV[j])}, {v(LV[j])}, {v(T[5])}")
        if bias:
            emit(f"v_add_u32 {v(LK[j])}, {bias}, {v(LK[j])}")
            emit(f"v_add_u32 {v(LV[j])}, {bias}, {v(LV[j])}")

    emit("; ---- Q fragments -> AGPRs: row q_row0 + 64*wave + 32*qb + l31, d = 16*ks + 8*hh; rows past seqlen_q are ZERO rows")
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, {5 + NQB - 1}")     # 32 NQB rows per wave
    emit(f"s_add_u32 {s(S_T0)}, {s(S_T0)}, {s(S_QROW0)}")
    emit(f"s_sub_u32 {s(S_T1)}, {s(S_SEQLENQ)}, 1")
    emit(f"v_lshlrev_b32 {v(T[6])}, 4, {v(T[0])}")            # hh * 16 bytes
    for qb in range(NQB):
        emit(f"v_add_u32 {v(QROW[qb])}, {s(S_T0)}, {v(T[1])}")
        if qb:
            emit(f"v_add_u32 {v(QROW[qb])}, 32, {v(QROW[qb])}")
        emit(f"v_min_i32 {v(T[3])}, {v(QROW[qb])}, {s(S_T1)}")
        emit(f"v_mad_u64_u32 {vr(T[4], 2)}, {sr(S_T64)}, {v(T[3])}, {s(S_QRS)}, 0")
        emit(f"v_add_co_u32 {v(T[4])}, vcc, {v(T[4])}, {v(T[6])}")
        emit(f"v_addc_co_u32 {v(T[5])}, vcc, 0, {v(T[5])}, vcc")
        emit(f"v_add_co_u32 {v(T[4])}, vcc, {s(S_QBASE)}, {v(T[4])}")
        emit(f"v_mov_b32 {v(T[7])}, {s(S_QBASE + 1)}")
        emit(f"v_addc_co_u32 {v(T[5])}, vcc, {v(T[5])}, {v(T[7])}, vcc")
        for ks in range(KS):
            emit(f"global_load_dwordx4 {vr(4 * KS * qb + 4 * ks, 4)}, {vr(T[4], 2)}, off offset:{32 * ks}")
    stamp(1)
    emit("s_waitcnt vmcnt(0)")
    stamp(2)
    for qb in range(NQB):
        emit(f"v_cmp_gt_i32 vcc, {s(S_SEQLENQ)}, {v(QROW[qb])}")
        for r in range(4 * KS):
            emit(f"v_cndmask_b32 {v(4 * KS * qb + r)}, 0, {v(4 * KS * qb + r)}, vcc")
    for r in range(4 * KS * NQB):
        emit(f"v_accvgpr_write_b32 a{Q_A0 + r}, {v(r)}")
    emit("; ---- state")
    for r in range(16 * DB * NQB):
        emit(f"v_accvgpr_write_b32 a{r}, 0")
    for qb in range(NQB):
        emit(f"v_mov_b32 {v(MTRUE[qb])}, 0xff800000")
        emit(f"v_mov_b32 {v(L0[qb])}, 0")
        emit(f"v_mov_b32 {v(L1[qb])}, 0")
        emit(f"v_mov_b32 {v(ALPHA[qb])}, 1.0")
        if HALF:
            # a half that does not list the first position starts from the empty state: its first tile's statistics see m_true =
            # -inf (every row votes "do": +inf > thr) and m_true > m_ref + tau / c = -inf, so the rescale block sets m_ref, -m_ref c
            # and the trigger level there (alpha = exp2(-inf) = 0 on l = 0 and O = 0)
            emit(f"v_mov_b32 {v(MREF[qb])}, 0xff800000")
            emit(f"v_mov_b32 {v(MTHR[qb])}, 0xff800000")
            emit(f"v_mov_b32 {v(NMS[qb])}, 0")

    if W2:
        # step i reads tab[i + 1].v (V(i+1)) and tab[i + 2].k (K(i+2)): TABV = &tab[1]. K(0) fragments -> AGPRs; once every wave has
        # them, K buffer 0 is free for K(2), which step 0 stages. The first tile's mask / statistics are step 0's (out-of-line blocks).
        emit("; ---- w2: K(0) fragments -> AGPRs")
        emit(f"v_mov_b32 {v(TABV)}, {s(S_TAB)}")
        emit(f"v_add_u32 {v(TABV)}, 16, {v(TABV)}")
        for j in range(NKF):
            emit(k_read(0, j))
        emit(f"s_mov_b32 {s(S_DOMASK)}, 0")
        emit(f"s_mov_b32 {s(S_BIT)}, 1")
        emit(f"s_mov_b32 {s(S_DOWORD)}, {s(S_DOFLAGS)}")
        emit(("DRAIN",))
        emit("s_barrier")
        return
    emit("; ---- tile addresses of positions 1..3 from the table; K(0) fragments -> AGPRs, S(0) = K(0) Q^T, then K(1) fragments")
    emit(f"v_mov_b32 {v(T[6])}, {s(S_TAB)}")
    emit(f"ds_read_b64 {vr(T[8], 2)}, {v(T[6])} offset:32")          # tab[2].k : K(2), staged below
    emit(f"ds_read_b64 {vr(T[10], 2)}, {v(T[6])} offset:48")         # tab[3].k : K(3), staged by step 0
    emit(f"ds_read_b64 {vr(T[12], 2)}, {v(T[6])} offset:24")         # tab[1].v : V(1), staged by step 0
    emit(f"v_add_u32 {v(TABV)}, 32, {v(T[6])}")                      # step 0 reads tab[2].v and tab[4].k
    for j in range(NKF):
        emit(k_read(0, j))
    emit(("DRAIN",))
    emit(f"v_readfirstlane_b32 {s(TBS[0])}, {v(T[8])}")
    emit(f"v_readfirstlane_b32 {s(TBS[0] + 1)}, {v(T[9])}")
    ord1 = [(f & 1) * KS + (f >> 1) for f in range(NKF)]
    if HALF:
        noqk0 = new_label("noqk0")
        emit(f"s_bitcmp1_b32 {s(S_ACT)}, 0")                   # this half lists the first position
        emit(f"s_cbranch_scc0 {noqk0}")
    for t in range(NG):
        out.append(mfma_qk(0, ord1[t // NQB], qb_of(t)))
    if HALF:
        label(noqk0)
    for j in range(NKF):
        emit(k_read(KV_TILE, j))
    emit(("DRAIN",))
    emit("s_barrier")                                          # every wave has read K(0) and K(1): both K buffers are free
    # K(2) -> K buffer 0. V(1)/K(3) are staged by step 0.
    for it in dma_ops(0, 0, do_k=True, do_v=False):
        out.append(it)
        if "m0" in it:
            emit("s_nop 0")
    for dst, src in ((TBS[0], T[10]), (TBS[0] + 1, T[11]), (VBS[0], T[12]), (VBS[0] + 1, T[13])):   # step 0 stages K(3), V(1)
        emit(f"v_readfirstlane_b32 {s(dst)}, {v(src)}")
    emit("s_nop 7")
    if HALF:
        nofirst = new_label("nofirst")
        emit(f"s_mov_b32 {s(S_DOMASK)}, 0")
        emit(f"s_bitcmp1_b32 {s(S_ACT)}, 0")
        emit(f"s_cbranch_scc0 {nofirst}")
    # seqlen-k mask: only if n0 == k_tiles-1 and tail_valid < 64  (mask.h:44-78; first walked tile only, mainloop...:1626)
    nomask = new_label("nomask")
    emit(f"s_cmp_eq_u32 {s(S_FIRSTLAST)}, 1")                # the first walked tile is tile k_tiles - 1 (C++ shell)
    emit(f"s_cbranch_scc0 {nomask}")
    emit(f"s_cmp_lt_i32 {s(S_TAILVALID)}, 64")
    emit(f"s_cbranch_scc0 {nomask}")
    for kb in range(2):
        for r in range(16):
            key = 32 * kb + (r & 3) + 8 * (r >> 2)
            emit(f"v_add_u32 {v(T[0])}, {key}, {v(HH4)}")
            emit(f"v_cmp_gt_i32 vcc, {s(S_TAILVALID)}, {v(T[0])}")            # key < tail_valid -> keep
            for qb in range(NQB):
                emit(f"v_cndmask_b32 {v(S_(0, kb, qb) + r)}, {v(NEGINF)}, {v(S_(0, kb, qb) + r)}, vcc")
    label(nomask)
    for op in row_max_ops(0):
        out.append(op)
    # first-tile stats: m_true = m_ref = row max; position 0 is never flagged (softmax.h:153)
    for qb in range(NQB):
        emit(f"v_mov_b32 {v(T[qb])}, {v(MLOC[qb])}")
    emit("s_nop 1")
    for qb in range(NQB):
        emit(f"v_permlane32_swap_b32 {v(MLOC[qb])}, {v(T[qb])}")
    emit("s_nop 1")
    for qb in range(NQB):
        emit(f"v_max_f32 {v(MTRUE[qb])}, {v(MLOC[qb])}, {v(T[qb])}")
    for qb in range(NQB):
        emit(f"v_mov_b32 {v(MREF[qb])}, {v(MTRUE[qb])}")
        emit(f"v_mul_f32 {v(NMS[qb])}, {s(S_NEGC)}, {v(MTRUE[qb])}")
        emit(f"v_add_f32 {v(MTHR[qb])}, {s(S_TAU)}, {v(MTRUE[qb])}")
    emit(f"s_mov_b32 {s(S_DOMASK)}, 1")                        # position 0 is never flagged; position 1 votes into bit 1
    emit(f"s_mov_b32 {s(S_BIT)}, 2")
    emit(f"s_mov_b32 {s(S_DOWORD)}, {s(S_DOFLAGS)}")
    for op in softmax_stream(0, list(range(XPAIRS))):
        out.append(op)
    if HALF:
        label(nofirst)                                         # (step 0 starts with the window as loaded: bit k = a(k))
        emit(f"s_mov_b32 {s(S_BIT)}, 2")
        emit(f"s_mov_b32 {s(S_DOWORD)}, {s(S_DOFLAGS)}")
        next_state(shift=False)

    emit(("DRAIN",))
    emit("s_barrier")


def epilogue():
    emit("; ---- flush the last (partial) vote word")
    nofl = new_label("nolastflush")
    emit(f"s_cmp_eq_u32 {s(S_DOMASK)}, 0")
    emit(f"s_cbranch_scc1 {nofl}")
    flush_domask()
    label(nofl)
    emit("s_nop 15")                                           # the last PV MFMAs have written the accumulators
    emit("s_nop 15")
    store_epilogue(globals(), O_)                              # gen_epilogue.py: 1/l, bf16 O and LSE from the registers
    emit("s_waitcnt lgkmcnt(0)")


def main():
    prologue()
    if W2:
        # two loops: waves 0-3 (group a) and waves 4-7 (group b, one QK ahead: see step_w2); every wave meets the same barriers
        done = new_label("done")
        loop_b_entry = new_label("w2groupb")
        emit(f"s_cmp_ge_u32 {s(S_WAVE)}, 4")
        emit(f"s_cbranch_scc1 {loop_b_entry}")
        for group in ("a", "b"):
            loop = new_label("loop" + group)
            if group == "b":
                label(loop_b_entry)
                if "norot" not in OPT:
                    w2_qk()                                    # tile 0 (its K fragments are in the AGPRs)
                    w2_stats(first_tile=True, can_be_first=False)
            label(loop)
            for variant in (0, 1):
                emit(f"s_cmp_lt_u32 {s(S_I)}, {s(S_NTILES)}")
                emit(f"s_cbranch_scc0 {done}")
                step_w2(variant, group)
            emit(f"s_branch {loop}")
        for blk in deferred:
            blk()
        label(done)
        epilogue()
        write_out()
        return
    loop, done = new_label("loop"), new_label("done")
    # Code placement (round 5). Where the loop head falls inside a 32-byte fetch window moves the body's throughput by up to 2-3 % with a
    # period of 32 bytes (tools/debug/phase_sweep_bench.py, profiles/r05_code_placement.md: head_dim 128: 1311-1316 TFLOP/s at phase 0,
    # 1329-1334 at phase 8; head_dim 64: 1011-1015 at 0, 1041 at 24) - and until round 5 that phase was whatever the C++ shell in front of
    # the asm statement happened to leave: an edit to the list writer moved the headline kernel from phase 8 to phase 16 and cost 0.8 %.
    # The head is now pinned: .p2align 5, then PHASE / 4 s_nop (executed once per item), per body the best measured phase.
    # `align:N` / `pad4:N` override it for experiments.
    if opt_val("align", "") or opt_val("pad4", ""):
        if opt_val("align", ""):
            out.append(f".p2align {opt_val('align', '')}")
        for _ in range(int(opt_val("pad4", "0"))):
            emit("s_nop 0")
    else:
        out.append(".p2align 5")
        for _ in range(LOOP_PHASE // 4):
            emit("s_nop 0")
    stamp(3)
    label(loop)
    for variant in (0, 1):
        if variant == 1:
            for _ in range(int(opt_val("pad4b", "0"))):        # code-placement experiments: the second copy of the step against the first
                emit("s_nop 0")
        if not HALF:
            emit(f"s_cmp_lt_u32 {s(S_I)}, {s(S_NTILES)}")
            emit(f"s_cbranch_scc0 {done}")
        if HALF:
            # four forms of the step by (a(i), a(i+1)); S_STATE was computed in front of the previous drain (4 = the walk is over). The
            # full form stays inline (the hot path: one compare and one branch in front of it, as in the form without activity bits),
            # the others are out of line
            notfull, after = new_label("notfull"), new_label("after_step")
            emit(f"s_cmp_eq_u32 {s(S_STATE)}, 3")
            emit(f"s_cbranch_scc0 {notfull}")
            step(variant)
            label(after)

            def partial_forms(notfull=notfull, after=after, variant=variant):
                l10, l01 = new_label("step10"), new_label("step01")
                label(notfull)
                emit(f"s_cmp_eq_u32 {s(S_STATE)}, 4")
                emit(f"s_cbranch_scc1 {done}")
                emit(f"s_cmp_eq_u32 {s(S_STATE)}, 1")
                emit(f"s_cbranch_scc1 {l10}")
                emit(f"s_cmp_eq_u32 {s(S_STATE)}, 2")
                emit(f"s_cbranch_scc1 {l01}")
                step(variant, a_cur=False, a_nxt=False)
                emit(f"s_branch {after}")
                label(l10)
                step(variant, a_cur=True, a_nxt=False)
                emit(f"s_branch {after}")
                label(l01)
                step(variant, a_cur=False, a_nxt=True)
                emit(f"s_branch {after}")
            deferred.append(partial_forms)
        elif HALFSKIP:
            # waves 0-1 sit out the steps with i % HALFSKIP == 0, waves 2-3 those with i % HALFSKIP == HALFSKIP / 2: what a workgroup
            # walking the union of two per-128-row lists would do on the tiles only one half lists (HISTORY.md section 8.6 a)
            assert not W2
            lbl, after = new_label("light"), new_label("after_light")
            emit(f"s_lshr_b32 {s(S_T0)}, {s(S_WAVE)}, 1")
            emit(f"s_mul_i32 {s(S_T0)}, {s(S_T0)}, {HALFSKIP // 2}")
            emit(f"s_and_b32 {s(S_T1)}, {s(S_I)}, {HALFSKIP - 1}")
            emit(f"s_cmp_eq_u32 {s(S_T0)}, {s(S_T1)}")
            emit(f"s_cbranch_scc1 {lbl}")
            step(variant)
            label(after)
            deferred.append(lambda lbl=lbl, after=after, variant=variant: (label(lbl), step(variant, light=True), emit(f"s_branch {after}")))
        else:                            # (the W2 body has its own loops and returned above)
            step(variant)
    emit(f"s_branch {loop}")
    for blk in deferred:
        blk()
    label(done)
    stamp(4)
    epilogue()
    stamp(5)
    stamps_out()
    write_out()


E64_OPS = ("v_exp_f32", "v_add_f32", "v_mul_f32", "v_sub_f32", "v_max_f32", "v_mov_b32", "v_add_u32")


def write_out():
    lines = finalize(out)
    if "e64" in OPT:       # code-placement experiment: every 4-byte VOP1 / VOP2 of the body in its 8-byte VOP3 encoding (same operation, same issue cost)
        import re
        pat = re.compile(r"^(\s*)(" + "|".join(E64_OPS) + r")(\s+[vs]\d+(?:,\s*-?[vs]\d+)+\s*)$")      # register operands only: VOP3 takes no literals on gfx9
        lines = [pat.sub(lambda m: m.group(1) + m.group(2) + "_e64" + m.group(3), ln) for ln in lines]
    text = "\n".join(lines)
    path = sys.argv[1] if len(sys.argv) > 1 else "la_fwd_x64_body.inc"
    with open(path, "w") as f:
        f.write("// GENERATED by gen_fwd_x64.py — do not edit. Inline-asm body of la_fwd_x64_kernel.\n" if D == 128 and not HALF else
                f"// GENERATED by gen_fwd_x64.py (LA_X64_D={D} LA_X64_FORM=half) — do not edit. Inline-asm body of la_fwd_x64_kernel<.., {D}, true>.\n" if HALF else
                f"// GENERATED by gen_fwd_x64.py (LA_X64_D={D}) — do not edit. Inline-asm body of la_fwd_bf16_x64_kernel<.., {D}>.\n")
        f.write(option_tag() + "\n")
        f.write('R"ASM(\n' + text + '\n)ASM"\n')
    print(f"wrote {path}: {len(lines)} lines, {text.count('v_mfma')} MFMAs")


if __name__ == "__main__":
    main()
