#!/usr/bin/env python
"""Generates la_fwd_x64_fp8[_d<D>]_body.inc: hand-scheduled gfx950 main loop of the fp8 (e4m3) QK-Skip forward, head_dim 128 (described here) and -
LA_X64F8_D, see the parameter block below - 64 / 96 / 192 / 256, with
ONE wave per SIMD and 64 query rows per wave (q-tile 256 x k-tile 64) - the structure of gen_fwd_x64.py (bf16) on the
block-scaled MFMA v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 MFMA rate; the non-scaled v_mfma_f32_32x32x16_fp8_fp8 runs at the
bf16 rate). The E8M0 scales are 2^0 everywhere except on P in the default body, which is block-scaled per (row, tile): "mx" below. One instruction contracts 64 indices, 32 bytes of A and of B per lane; the
contraction index is permuted freely (A and B only have to agree), which lets P go from the S^T accumulators straight into
the B operand. Operand layout: K rows / Q fragments as 2 x 16-byte chunks per 64-wide contraction step, V^T tiles
pre-transposed by la_prep_v_fp8 so that the PV operand is two plain ds_read_b128).

Why: the 128-row fp8 kernel (two 32-row waves per SIMD, hipcc-scheduled) is VALU-issue-bound at 38 % MFMA utilisation
(profiles/r01f_fp8): per 64 query rows it pays the per-tile statistics / SALU / waits twice and stalls 30 % of its cycles on
dependencies. Here one wave does 64 rows with ONE set of per-tile overhead, and every filler is placed by hand.

Register file (per lane):  AGPR  a[0:127]   O^T  (2 q-blocks x 4 d-blocks x 16)
                                 a[128:159] Q    (2 q-blocks x 2 contraction steps x 8: B operand of S^T = K Q^T)
                                 a[160:191] K    fragments of the NEXT tile (2 key blocks x 2 steps x 8: A operand)
                           VGPR  v[0:63] / v[64:127]  S^T ping / pong, q-block major: (q-block, key block, 16). P (e4m3) is
                                 compacted IN PLACE into the first 8 registers of a q-block's 32 = the B operand of PV
                                 v[128:159] the four V^T fragments of the tile (A operand of PV, 8 registers each)
Three bodies are generated (LA_X64F8_OPT): the default ("lin" + "mx": the e4m3 byte of P computed directly and block-scaled, row sums
from the matrix pipe), "exp" (v_exp_f32 + hardware rounding, row sums from the matrix pipe) and "lvalu" (that with fp32 row sums on
the vector unit: the reference's arithmetic); the comments at LMFMA / LIN / MX below say what each changes. The step, in "exp" form:
Step i:  phase 1   8 MFMA  S_nxt = K(i+1) Q^T  ||  rest of P(i) = exp2(S c - m_ref c + OFF), row sums, e4m3 compaction;
                                                    LDS-DMA of V^T(i+1), K(i+3); the 8 V^T fragment reads
         phase 2   8 MFMA  O^T += V^T(i) P(i)^T ||  K(i+2) fragment reads -> AGPRs; next step's tile lookup / DMA bases;
                                                    row max of S_nxt, running max, skip vote, lazy-rescale test; start of P(i+1)
P offset ("exp" / "lvalu"): the reference scales P by 2^8 before the e4m3 cast (softmax.h:85-87). With the lazy rescale P can reach 2^tau, so
OFF = 8 - tau with tau = 2: P 2^6 <= 256 < 448 (e4m3 max). e4m3 rounding is scale-invariant away from the subnormal end, so
results equal the offset-8 ones except for P < 2^-12 (absolute 2.4e-4 of a weight <= 1).
"""
import os
import sys

from gen_epilogue import store_epilogue

OPT = set(x for x in os.environ.get("LA_X64F8_OPT", "").split(",") if x)


def opt_val(key, default):
    for o in OPT:
        if o.startswith(key + ":"):
            return o[len(key) + 1:]
    return default


# Options that only move instructions or select one of the three product bodies (exp / lvalu): same results bit for bit as the body of
# that name. Everything else is a pricing experiment; the first line of a generated body says which kind went in (see gen_fwd_x64.py).
SCHEDULE_ONLY = {"x", "dmagaps", "align", "pad4", "smstart", "klate", "pk", "exp", "lvalu", "vspread"}


def option_tag():
    wrong = sorted(o for o in OPT if o.split(":")[0] not in SCHEDULE_ONLY)
    return (f"// la_body_options: {','.join(sorted(OPT)) or '-'}; wrong_results={1 if wrong else 0}"
            + (f" (PRICING ONLY, results are wrong: {','.join(wrong)})" if wrong else ""))



# Head dim (round 6): 128, or 64 = the same step with ONE 64-wide contraction per score block and two 32-wide d-blocks of O^T: 4 QK + 4 PV
# MFMAs per step instead of 8 + 8 under the same softmax. K tile = 64 keys x 64 bytes, held in LDS as 32 pseudo-rows of 128 bytes (key R in
# chunks 0-3, key R + 32 in chunks 4-7) so that the fragment addresses and the swizzle are those of head_dim 128 with `sx` read as the key
# block; prepared V^T tile = its first 64 rows. One LDS-DMA piece of 1 KiB per wave, tensor and step instead of two.
# Head dims 192 / 256 (round 6, "WIDE"): ONE q-block of 32 rows per wave (O^T of 32 rows x 256 is 128 accumulators), q-tile 128 x k-tile 64 - the tiles of
# the bf16 kernels of these head dims, so lists keep their geometry. 3 / 4 contraction steps per score block, 6 / 8 d-blocks; K rows sit in LDS at a
# 256-byte stride (16 chunks, XOR-swizzled by row & 15; at 192 the last four chunks of a row are DMA filler, never read), rings of 16 KiB per stage.
# Per wave and step: 16 K + 16 V^T fragment reads of 1 KiB against 16 MFMAs and half the softmax of the 64-row bodies (measured: the power cap, MFMA busy 56 % at 1.94 GHz, profiles/r06_fp8_dims_pmc.md).
# Head dim 96: the 128 step with three d-blocks of O^T (8 QK + 6 PV MFMAs) - its contraction is one and a half 64-wide steps, so QK^T keeps both: the K tile
# sits in LDS exactly as at 128 with the two chunk positions per row whose source would be chunk 6 / 7 filled by DMA filler (a copy of chunk 5: finite
# data) and the matching quarter of the Q fragments ZERO; the prepared V^T tile is padded to the 128 tile's 8 KiB by the prepare pass.
D = int(os.environ.get("LA_X64F8_D", "128"))
assert D in (64, 96, 128, 192, 256), D
WIDE = D > 128
NQB = 1 if WIDE else 2                    # q-blocks of 32 rows per wave
QBS = tuple(range(NQB))
NSX, ND = (D + 63) // 64, D // 32         # 64-wide contraction steps of S^T = K Q^T; 32-wide d-blocks of O^T
DB = ND                                   # gen_epilogue.py: d-blocks to store
PIECES_K = 4 if WIDE else (D + 63) // 64  # 1 KiB LDS-DMA pieces per wave and K tile (WIDE: the 16 KiB image of 256-byte rows)
PIECES_V = (D + 63) // 64                 # ... and prepared V^T tile (64 D bytes; 96: padded to 8 KiB)
ROWSUM = 99                               # "d-block" index of the row-sum MFMA in PV_ORDER
XPAIRS = int(opt_val("x", {128: {"lin": "8", "exp": "4", "lvalu": "4"}, 64: {"lin": "12", "exp": "2", "lvalu": "6"}, 96: {"lin": "8", "exp": "4", "lvalu": "4"},
                                192: {"lin": "8", "exp": "8", "lvalu": "8"}, 256: {"lin": "8", "exp": "8", "lvalu": "8"}}[D]
                        ["lvalu" if "lvalu" in OPT else "exp" if "exp" in OPT else "lin"]))          # pair-groups (of 16) done in phase 2. EVEN: two pair-groups share one packed-e4m3 destination register
                                          # (lo / hi half by op_sel); an odd split leaves a half-written register across the phase boundary
                                          # (measured x = 2 / 4: 2078-2101 TFLOP/s at 42 %, x = 3 / 5 / 6: 2048-2065; round 6, the lvalu body as the default: x = 2 / 4 / 6 / 8 ->
                                          # 2062 / 2089 / 2037 / 2079). head_dim 64 (phase 1 has 4 MFMAs, not 8;
                                          # tools/debug/fp8_dim_ab.py, dense S = 16 384: lvalu x = 2 / 4 / 6 / 8: 2.40 / 2.41 / 2.30 / 2.40 ms; exp 2 / 4 / 6 / 8: 2.11 / 2.18 /
                                          # 2.13 / 2.19; lin 4 / 8 / 10 / 12 / 14: 1.66 / 1.60 / 1.65 / 1.585 / 1.68 against 1.67 of that session's x = 8)
PK = "pk" in OPT                          # A/B: packed fp32 FMA / add in the softmax (v_pk_fma_f32, v_pk_add_f32). MEASURED ANTI-LEVER here too:
                                          # 64 fewer instructions per step, bit-identical results, 1891 vs 2068 TFLOP/s at 42 % (round 2, tools/ab.py --fp8)
NG = 2 * NSX * NQB                        # QK MFMAs (gaps) of phase 1
# Row sums. "lvalu" (round 2): l = sum of the UN-rounded fp32 P, 64 v_add_f32 per step (the reference's form, softmax.h:275-296).
# Default (round 3): l~ = sum of the e4m3-ROUNDED P, taken from the matrix pipe: one more 32 x 32 x 64 MFMA per q-block and step
# with an all-ones A operand (every row of the result is the column sum of P^T), accumulated in a[192:223] and rescaled with O.
# Why: this kernel is bound by VALU issue (383 issue slots per step against 256 quad-cycles of MFMA; a lone wave issues a slot
# every ~4.8 cycles: tools/valu_microbench.py), the packed / dot2 forms that would halve the adds are VOP3P and serialise with
# the MFMA in flight (same tool), and the matrix pipe idles 55 % of the step. O = (sum P~ V) / (sum P~) is also the better
# numerics under the lazy rescale: the weights sum to 1 exactly, so the e4m3 rounding of a row's dominant P (which is not 2^k
# once m_ref lags m_true) cancels instead of scaling the whole row by up to 2^-4. The LSE inherits the rounding of P~ (HISTORY.md 3.4).
LMFMA = "lvalu" not in OPT
# P itself. "exp" (rounds 1-2; LA_FLAG_FP8_MFMA_ROWSUM; with `lvalu` the default form): P = v_exp_f32(S c - m_ref c + OFF), rounded to e4m3 by v_cvt_pk_fp8_f32 - per score one
# FMA, one transcendental (2 issue slots) and half a convert (which also costs 2 slots: tools/valu_microbench.py) = 4 slots.
# Default (round 3) "lin": the e4m3 BYTE is computed directly, b = sat_u8(rne(8 y + 56 - 8 delta)), y = S c - m_ref c + OFF - one FMA
# and one v_cvt_pk_u8_f32 (RNE, saturating, NaN / -inf -> 0: probed on the hardware), 2 slots per score, no transcendental.
# An e4m3 byte b = 8 E + M decodes to 2^(E - 7) (1 + M / 8): reading b / 8 - 7 as a base-2 logarithm is the classic
# linear-mantissa exponential (1 + f for 2^f, at most +6.1 %, cancelled on average by delta = 0.0575 = log2 of its mean ratio), and
# the byte grid is then a LOG-uniform quantisation of P (step 2^(1/8)) instead of e4m3's round-to-nearest (relative step 1/8 ..
# 1/16). Per element the error is within [-7.9 %, +6.5 %] against +-6.25 % for the hardware rounding; measured on the reference's
# fp8 goldens the output error is 1.0-2.2 x that of the exact form and stays under the reference's own rule (HISTORY.md 3.4).
# What it costs in range: bytes 1..7 (e4m3 subnormals) are reached for y in (-7, -6] only, so P below 2^-7 is dropped where the
# hardware rounding keeps P down to 2^-10. Needs the row sums of the ENCODED P (LMFMA): no fp32 P exists in this form.
LIN = LMFMA and "exp" not in OPT
LIN_DELTA = 0.0575
NG2 = NQB * ND + (NQB if LMFMA else 0)    # MFMAs (gaps) of phase 2: PV + the row-sum MFMA of each q-block
# Lazy-rescale slack TAU (log2 units) and the offset of P. P <= 2^(P_OFFSET + TAU) must stay finite in e4m3 (max 448 = 2^8.8):
# exp / lvalu: the reference's 2^8 ceiling (Max_offset = 8, softmax.h:85-87), TAU = 2. lin: ceiling 2^8.75 (byte 126; byte 127 is NaN)
# and TAU = 1 - the byte grid ends at byte 1 = 2^-9 <-> y = -6.94, so a key is kept while its weight is above 2^-(P_OFFSET + 6.94)
# of the row's reference maximum: 2^-14.7 here (hardware rounding at offset 6: 2^-16); every octave of TAU is an octave of tail.
# The shell takes both numbers from the generated la_fwd_x64_fp8*_consts.h (param[22] = TAU / c, param[31] = ln 2^-P_OFFSET).
# mx (default with lin): P is BLOCK-SCALED, the MX way, on the scale operand the matrix instruction has anyway. The E8M0 byte a lane
# supplies with the B operand multiplies one 32-element block of its column: registers 0-3 of lanes n and n + 32 take lane n's byte,
# registers 4-7 take lane n + 32's (probed: tools/debug/probe_mfma_scale2.hip) - here the two 32-key halves of the tile, for one
# query row. Both get the same exponent, that of the row's largest P in THIS TILE: t = floor((m_loc - m_ref) c) from the row maximum
# the skip vote needs anyway, P is encoded relative to 2^t (the tile's largest P lands in bytes 112..120 = [2^7, 2^8)) and the MFMA
# multiplies by 2^t. What it buys: (1) the byte grid's 15 octaves hang below the TILE's maximum, not the row's - a diffuse tail far
# below the row maximum keeps full relative precision (the plain byte grid drops keys below 2^-14.7 of the reference maximum, the
# reference's e4m3 rounding below 2^-17; at S = 75 600 such tails carry real mass: HISTORY.md 3.4); (2) P cannot overflow whatever
# the row maximum does, so the lazy rescale only guards the fp32 accumulators: TAU = 32, i.e. never on real data. Cost: 8 vector
# instructions per step (exponent, scale byte and encoding offset for both q-blocks).
MX = LIN and "nomx" not in OPT
TAU = float(opt_val("tau", "32" if MX else ("1" if LIN else "2")))
P_CEIL = 8.75 if LIN else 8.0
P_OFFSET = 7.0 if MX else P_CEIL - TAU
DMA_GAPS = [int(x) for x in opt_val("dmagaps", {64: "0,1,2,3", 96: "0,1,1,2,3,3", 128: "0,1,1,2,3,3", 192: "0,1,1,2,2,3,4,4,5", 256: "0,1,1,2,2,3,4,4,5,5"}[D]).replace(".", ",").split(",")]   # m0K,K0,K1,m0V,V0,V1 (phase 1 gaps; an M0 write is never adjacent to its first use)


# ---------------------------------------------------------------- AGPR map
def O_(qb, db):
    return 64 * qb + 16 * db


def QA(qb, sx):
    return 128 + 16 * qb + 8 * sx


def KA(j):          # j = 2*kb + sx (head_dim 64: j = kb)
    return 160 + 8 * j


def LSUM(qb):       # row sums of P~ (LMFMA): 16 accumulator registers per q-block, all rows equal
    return 224 if WIDE else 192 + 16 * qb


# ---------------------------------------------------------------- VGPR map
def S_(sset, qb, kb):
    return 64 * sset + 32 * qb + 16 * kb


VF = [96 + 8 * i for i in range(8)] if WIDE else [128 + 8 * i for i in range(4)]      # WIDE: v[96:159] (no second q-block: v[32:63] and v[96:127] are free)
KADDR = list(range(32, 40)) if WIDE else list(range(160, 164))             # [2*sx + t]
VADDR = [164, 165]                        # [t]
LK = [166, 167, 40, 41] if WIDE else [166, 167]
LV = 168
VSC = 169                                 # 0x7f7f7f7f: four E8M0 exponents of 127 (= 2^0)
ONES = 170                                # v[170:177] = 0x38383838: the all-ones e4m3 A operand of the row-sum MFMA
MTRUE, MREF, NMS, L0, L1, MLOC, MLOC2, ALPHA = ([180, 181], [182, 183], [184, 185], [186, 188], [187, 189], [190, 191],
                                                [192, 193], [194, 195])
T = list(range(196, 212))
NEGINF, HH4, LANE = 212, 213, 214
QROW = [216, 217]
MTHR = [220, 221]
TABV = 222                                # LDS address of tab[i + 2], the tile-address table entry step i reads from
# mx: per row -m_ref c + 127; per row and tile the offset of the byte encoding; the tile's E8M0 scale byte per S set
NMR, NMSB, SCB = [178, 179], [186, 187], [[188, 189], [215, 218]]      # (L0 / L1 are unused with the matrix-pipe row sums)
MXT = [T[10], T[11]]

# ---------------------------------------------------------------- SGPR map (s32-s34 are ABI-reserved: unused)
S_KBASE, S_VBASE, S_QBASE = 36, 38, 40
S_TB, S_VB, S_EXEC, S_T64, S_T64B = 42, 44, 46, 48, 50
(S_KRS, S_VRS, S_LASTROW, S_NTILES, S_C, S_THR, S_TAILVALID, S_FIRSTLAST, S_TAB, S_DOFLAGS, S_WAVE, S_I, S_DOMASK,
 S_FREE0, S_FREE1, S_FREE2, S_LDS, S_T0, S_T1, S_T2, S_T3, S_NM1, S_QRS, S_QROW0, S_SEQLENQ, S_EXPORT, S_PARAM, S_DOWORD, S_NEGC,
 S_FREE3, S_DMAW, S_FREE4, S_TAU, S_RESC, S_FREE5) = range(52, 87)
S_C8, S_NEGC8, S_M8 = S_FREE0, S_FREE1, S_FREE2   # lin: 8 c and -8 c; mx: -8.0
S_FREE6, S_TB2, S_VB2, S_BIT = 87, 88, 90, 92     # second set of DMA bases (the loop is unrolled by two); the rotating vote bit
S_DMAWV = S_FREE4 if PIECES_V != PIECES_K else S_DMAW   # LDS-DMA destination of this wave's V^T pieces (head_dim 192: 3 pieces against 4 of K)
TBS, VBS = [S_TB, S_TB2], [S_VB, S_VB2]

KV_TILE = 16384 if WIDE else 8192         # one stage of the K / V^T rings
V_REGION = 2 * KV_TILE

out = []


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
    return f".LF{prefix}_{uid[0]}_%="


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
            lines.append("    s_waitcnt lgkmcnt(0)" if "nowaitvm" in OPT else "    s_waitcnt vmcnt(0) lgkmcnt(0)")   # nowaitvm: pricing only
            q = []
    return lines


# ---------------------------------------------------------------- building blocks
def k_read(kbuf_imm, j, t):
    if D == 64:     # pseudo-rows of 128 bytes: the key block selects the chunk group, as sx does at 128
        return ("LDS", f"ds_read_b128 {ar(KA(j) + 4 * t, 4)}, {v(KADDR[2 * j + t])} offset:{kbuf_imm}", ("k", j, t))
    kb, sx = j // NSX, j % NSX
    if WIDE:        # rows of 256 bytes: 32 keys of a key block = 8 KiB
        return ("LDS", f"ds_read_b128 {ar(KA(j) + 4 * t, 4)}, {v(KADDR[2 * sx + t])} offset:{kbuf_imm + kb * 8192}", ("k", j, t))
    return ("LDS", f"ds_read_b128 {ar(KA(j) + 4 * t, 4)}, {v(KADDR[2 * sx + t])} offset:{kbuf_imm + kb * 4096}", ("k", j, t))


def v_read(vbuf_imm, db, t):
    return ("LDS", f"ds_read_b128 {vr(VF[db] + 4 * t, 4)}, {v(VADDR[t])} offset:{V_REGION + vbuf_imm + db * 2048}", ("v", db, t))


MFMA = "v_mfma_scale_f32_32x32x64_f8f6f4"
SCALES = f"{v(VSC)}, {v(VSC)} op_sel_hi:[0,0,0]"


def mfma_qk(sset, kb, sx, qb):
    d = S_(sset, qb, kb)
    c = "0" if sx == 0 else vr(d, 16)
    return f"    {MFMA} {vr(d, 16)}, {ar(KA(NSX * kb + sx), 8)}, {ar(QA(qb, sx), 8)}, {c}, {SCALES}"


def mfma_pv(sset, db, qb):
    sc = f"{v(VSC)}, {v(SCB[sset][qb])} op_sel_hi:[0,0,0]" if MX else SCALES      # mx: the lane's block scale on the B operand (P)
    if db == ROWSUM:      # row sums: ones (32 x 64) times P^T
        return f"    {MFMA} {ar(LSUM(qb), 16)}, {vr(ONES, 8)}, {vr(S_(sset, qb, 0), 8)}, {ar(LSUM(qb), 16)}, {sc}"
    return f"    {MFMA} {ar(O_(qb, db), 16)}, {vr(VF[db], 8)}, {vr(S_(sset, qb, 0), 8)}, {ar(O_(qb, db), 16)}, {sc}"


def mx_ops(sset, src):
    """Exponent and encoding offset of the tile whose scores sit in S set `sset` (src = the rows' maxima over this tile, the same in
    both half-waves): u = floor((m_loc - m_ref) c) + 127 -> E8M0 byte (the byte convert saturates: below 2^-127 the scale stops
    following, harmless; a tile of -inf gives NaN offsets = byte 0 for every key), offset = NMS - 8 (u - 127) (NMS carries the + 1016)."""
    o = []
    for qb in QBS:
        o.append(f"    v_fma_f32 {v(MXT[qb])}, {v(src[qb])}, {s(S_C)}, {v(NMR[qb])}")
    for qb in QBS:
        o.append(f"    v_floor_f32 {v(MXT[qb])}, {v(MXT[qb])}")
    if "mxt0" in OPT:        # debug: every block exponent 0
        for qb in QBS:
            o.append(f"    v_mov_b32 {v(MXT[qb])}, 0x{float_bits(127.0):08x}")
    for qb in QBS:
        o.append(f"    v_cvt_pk_u8_f32 {v(SCB[sset][qb])}, {v(MXT[qb])}, 0, 0")
    for qb in QBS:
        o.append(f"    v_fma_f32 {v(NMSB[qb])}, {v(MXT[qb])}, {s(S_M8)}, {v(NMS[qb])}")
    return o


def softmax_parts(sset, p):
    """Pair p (elements 2p, 2p+1 of the 32 per lane and q-block; key block kb = p >> 3) of BOTH q-blocks."""
    F, E, A, C = [], [], [], []
    for qb in QBS:
        e0 = 2 * p
        kb, r = e0 >> 4, e0 & 15
        r0 = S_(sset, qb, kb) + r
        r1 = r0 + 1
        dst = S_(sset, qb, 0) + 4 * kb + (r >> 2)         # 32 e4m3 bytes of a q-block = its first 8 registers
        if LIN:
            # in place: 8 y + 56 - 8 delta replaces the score, then one byte convert each into byte r & 3 of dst. dst is one of the
            # q-block's first 8 registers, which hold the scores of pairs 0-3: pairs are processed in order, so the pair whose
            # scores sit in dst (pair dst_index >> 1 <= p) has consumed them (its own first convert reads and writes dst at once)
            nms = NMSB[qb] if MX else NMS[qb]
            F.append([f"    v_fma_f32 {v(r0)}, {v(r0)}, {s(S_C8)}, {v(nms)}", f"    v_fma_f32 {v(r1)}, {v(r1)}, {s(S_C8)}, {v(nms)}"])
            E.append([])
            A.append([])
            C.append([f"    v_cvt_pk_u8_f32 {v(dst)}, {v(r0)}, {r & 3}, {v(dst)}", f"    v_cvt_pk_u8_f32 {v(dst)}, {v(r1)}, {(r & 3) + 1}, {v(dst)}"])
            continue
        hi = " op_sel:[0,0,1]" if (r >> 1) & 1 else ""
        ta, tb = T[8 + 2 * qb], T[9 + 2 * qb]
        if PK:
            # packed fp32: one instruction = the two FMAs / the two adds of the pair, lane for lane the same IEEE operations
            # (c from the low dword of s[S_C:S_C+1] for both halves; -m.c from the NMS register of this q-block for both)
            assert r0 % 2 == 0 and ta % 2 == 0 and tb == ta + 1 and L1[qb] == L0[qb] + 1 and L0[qb] % 2 == 0 and S_C % 2 == 0
            nb, nsel = NMS[qb] & ~1, NMS[qb] & 1
            F.append([f"    v_pk_fma_f32 {vr(ta, 2)}, {vr(r0, 2)}, {sr(S_C)}, {vr(nb, 2)} op_sel:[0,0,{nsel}] op_sel_hi:[1,0,{nsel}]"])
            A.append([f"    v_pk_add_f32 {vr(L0[qb], 2)}, {vr(L0[qb], 2)}, {vr(r0, 2)}"])
        else:
            F.append([f"    v_fma_f32 {v(ta)}, {v(r0)}, {s(S_C)}, {v(NMS[qb])}", f"    v_fma_f32 {v(tb)}, {v(r1)}, {s(S_C)}, {v(NMS[qb])}"])
            A.append([] if LMFMA else [f"    v_add_f32 {v(L0[qb])}, {v(L0[qb])}, {v(r0)}", f"    v_add_f32 {v(L1[qb])}, {v(L1[qb])}, {v(r1)}"])
        E.append([f"    v_exp_f32 {v(r0)}, {v(ta)}", f"    v_exp_f32 {v(r1)}, {v(tb)}"])
        C.append([f"    v_cvt_pk_fp8_f32 {v(dst)}, {v(r0)}, {v(r1)}{hi}"])
    return F, E, A, C


def softmax_stream(sset, groups):
    """Software-pipelined: every v_exp of group g is followed by the fma of group g+1 (same temp, just consumed) and the add /
    convert of group g-1; exps are never adjacent."""
    if not groups:
        return []
    parts = [softmax_parts(sset, p) for p in groups]
    def both(per_qb):
        return [x for qb in QBS for x in per_qb[qb]]
    if LIN:          # FMAs of pair g + 1 between the FMAs and the converts of pair g: every convert is >= 4 instructions behind its FMA
        o = both(parts[0][0])
        for g in range(len(parts)):
            if g + 1 < len(parts):
                o += both(parts[g + 1][0])
            o += both(parts[g][3])
        return o
    o = both(parts[0][0])
    n = len(parts)
    for g in range(n):
        Fn = parts[g + 1][0] if g + 1 < n else None
        Ap, Cp = (parts[g - 1][2], parts[g - 1][3]) if g > 0 else (None, None)
        E = parts[g][1]
        for qb in QBS:
            if PK:       # the packed FMA of group g+1 overwrites both temps: it goes behind the second exp
                fill0 = list(Ap[qb]) if Ap else []
                fill1 = (list(Fn[qb]) if Fn else []) + (list(Cp[qb]) if Cp else [])
            else:
                fill0 = ([Fn[qb][0]] if Fn else []) + (list(Ap[qb][:1]) if Ap else [])
                fill1 = ([Fn[qb][1]] if Fn else []) + (list(Ap[qb][1:]) if Ap else []) + (list(Cp[qb]) if Cp else [])
            o += [E[qb][0]] + fill0 + [E[qb][1]] + fill1
    for qb in QBS:
        o += parts[-1][2][qb]
    for qb in QBS:
        o += parts[-1][3][qb]
    return o


def row_max_ops(sset):
    per = []
    for qb in QBS:
        regs = [S_(sset, qb, 0) + r for r in range(32)]
        ops = [f"    v_max_f32 {v(MLOC[qb])}, {v(regs[0])}, {v(regs[1])}", f"    v_max_f32 {v(MLOC2[qb])}, {v(regs[2])}, {v(regs[3])}"]
        rest = regs[4:]
        chains = [MLOC[qb], MLOC2[qb]]
        for n_, i in enumerate(range(0, len(rest), 2)):
            ch = chains[n_ & 1]
            ops.append(f"    v_max3_f32 {v(ch)}, {v(ch)}, {v(rest[i])}, {v(rest[i + 1])}")
        ops.append(f"    v_max_f32 {v(MLOC[qb])}, {v(MLOC[qb])}, {v(MLOC2[qb])}")
        per.append(ops)
    return [x for pair in zip(*per) for x in pair]


def stats_ops(rare_label, back_label, flush_label, flush_back, inval_label, inval_back, sset=0):
    """Half-wave max exchange, skip vote, true running max, lazy-rescale test: as gen_fwd_x64.py stats_ops (rotating vote bit in
    S_BIT, the step past the end of the walk recognised by i == n - 1, no position arithmetic)."""
    o = []
    a = o.append

    def any_lane(cmp_ops):
        """vcc = OR of the compares of the q-blocks, SCC = some lane set (one q-block: the OR of the mask with itself sets SCC)."""
        dst = [sr(S_T64)] + ["vcc"] * (NQB - 1)
        for qb in QBS:
            a(cmp_ops[qb].format(dst=dst[qb]))
        a(f"    s_or_b64 vcc, {'vcc' if NQB == 2 else sr(S_T64)}, {sr(S_T64)}")
    for qb in QBS:
        a(f"    v_mov_b32 {v(T[qb])}, {v(MLOC[qb])}")
    a(f"    v_add_u32 {v(TABV)}, 16, {v(TABV)}")
    for qb in QBS:
        a(f"    v_permlane32_swap_b32 {v(MLOC[qb])}, {v(T[qb])}")
    a("    s_nop 0")
    for qb in QBS:
        a(f"    v_max_f32 {v(MLOC[qb])}, {v(MLOC[qb])}, {v(T[qb])}")
    a(f"    s_cmp_eq_u32 {s(S_I)}, {s(S_NM1)}")
    a(f"    s_cbranch_scc1 {inval_label}")
    o.append(inval_back + ":")
    for qb in QBS:
        a(f"    v_sub_f32 {v(T[2 + qb])}, {v(MLOC[qb])}, {v(MTRUE[qb])}")              # vote: (m_loc - m_prev) * c > thr (softmax.h:194)
    for qb in QBS:
        a(f"    v_max_f32 {v(MTRUE[qb])}, {v(MTRUE[qb])}, {v(MLOC[qb])}")
    for qb in QBS:
        a(f"    v_mul_f32 {v(T[2 + qb])}, {s(S_C)}, {v(T[2 + qb])}")
    any_lane([f"    v_cmp_gt_f32 {{dst}}, {v(T[2 + qb])}, {s(S_THR)}" for qb in QBS])         # SCC = some row of the wave voted "do"
    a(f"    s_cselect_b32 {s(S_T0)}, {s(S_BIT)}, 0")
    a(f"    s_or_b32 {s(S_DOMASK)}, {s(S_DOMASK)}, {s(S_T0)}")
    any_lane([f"    v_cmp_gt_f32 {{dst}}, {v(MTRUE[qb])}, {v(MTHR[qb])}" for qb in QBS])       # lazy rescale trigger
    a(f"    s_cbranch_vccnz {rare_label}")
    o.append(back_label + ":")
    if MX:
        o += mx_ops(sset, MLOC)
    a(f"    s_lshl_b32 {s(S_BIT)}, {s(S_BIT)}, 1")                           # falls off the word (SCC = 0): flush it
    a(f"    s_cbranch_scc0 {flush_label}")
    o.append(flush_back + ":")
    return o


def set_nms(qb):
    """-m_ref*c + P_OFFSET (the literal needs the VOP2 encoding: gfx9 VOP3 takes no literals)."""
    if LIN:          # -8 m_ref c + (8 OFF + 56 - 8 delta): the byte of P = 2^OFF (a row's dominant key while m_ref = m_true) is exact
        emit(f"v_mul_f32 {v(NMS[qb])}, {s(S_NEGC8)}, {v(MREF[qb])}")
        emit(f"v_add_f32 {v(NMS[qb])}, 0x{float_bits(8.0 * P_OFFSET + 56.0 - 8.0 * LIN_DELTA + (1016.0 if MX else 0.0)):08x}, {v(NMS[qb])}")
        if MX:       # -m_ref c + 127: the block exponent comes out biased like an E8M0 byte; NMS carries the matching + 8 * 127
            emit(f"v_mul_f32 {v(NMR[qb])}, {s(S_NEGC)}, {v(MREF[qb])}")
            emit(f"v_add_f32 {v(NMR[qb])}, 0x{float_bits(127.0):08x}, {v(NMR[qb])}")
        return
    emit(f"v_mul_f32 {v(NMS[qb])}, {s(S_NEGC)}, {v(MREF[qb])}")
    emit(f"v_add_f32 {v(NMS[qb])}, 0x{float_bits(P_OFFSET):08x}, {v(NMS[qb])}")


def float_bits(x):
    import struct
    return struct.unpack("<I", struct.pack("<f", x))[0]


def rare_rescale_block(rare_label, back_label):
    label(rare_label)
    for qb in QBS:
        emit(f"v_sub_f32 {v(T[2 + qb])}, {v(MREF[qb])}, {v(MTRUE[qb])}")
    for qb in QBS:
        emit(f"v_mul_f32 {v(T[2 + qb])}, {s(S_C)}, {v(T[2 + qb])}")
    for qb in QBS:
        emit(f"v_exp_f32 {v(ALPHA[qb])}, {v(T[2 + qb])}")
    for qb in QBS:
        emit(f"v_mov_b32 {v(MREF[qb])}, {v(MTRUE[qb])}")
    for qb in QBS:
        set_nms(qb)
        emit(f"v_add_f32 {v(MTHR[qb])}, {s(S_TAU)}, {v(MREF[qb])}")
    for qb in QBS:
        if not LMFMA:                        # (LMFMA: the row sums live in accumulators and are rescaled with O)
            emit(f"v_mul_f32 {v(L0[qb])}, {v(L0[qb])}, {v(ALPHA[qb])}")
            emit(f"v_mul_f32 {v(L1[qb])}, {v(L1[qb])}, {v(ALPHA[qb])}")
    emit(f"s_mov_b32 {s(S_RESC)}, 1")
    emit(f"s_branch {back_label}")


def inval_block(lbl, back):
    label(lbl)
    for qb in QBS:
        emit(f"v_mov_b32 {v(MLOC[qb])}, {v(NEGINF)}")
        emit(f"v_mov_b32 {v(NMS[qb])}, {v(NEGINF)}")
    emit(f"s_branch {back}")


def flush_block(flush_label, back_label):
    label(flush_label)
    flush_domask()
    emit(f"s_add_u32 {s(S_DOWORD)}, {s(S_DOWORD)}, 4")
    emit(f"s_mov_b32 {s(S_BIT)}, 1")
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
    label(lbl)
    emit("s_nop 15")
    emit("s_nop 15")
    emit("s_nop 15")
    emit("s_nop 15")
    for qb in QBS:
        for base in range(0, 16 * ND, 8):
            for k in range(8):
                emit(f"v_accvgpr_read_b32 {v(T[k])}, a{64 * qb + base + k}")
            for k in range(8):
                emit(f"v_mul_f32 {v(T[k])}, {v(T[k])}, {v(ALPHA[qb])}")
            for k in range(8):
                emit(f"v_accvgpr_write_b32 a{64 * qb + base + k}, {v(T[k])}")
    if LMFMA:                                # the row sums: only register 0 of each block is ever read
        for qb in QBS:
            emit(f"v_accvgpr_read_b32 {v(T[qb])}, a{LSUM(qb)}")
        for qb in QBS:
            emit(f"v_mul_f32 {v(T[qb])}, {v(T[qb])}, {v(ALPHA[qb])}")
        for qb in QBS:
            emit(f"v_accvgpr_write_b32 a{LSUM(qb)}, {v(T[qb])}")
    emit(f"s_mov_b32 {s(S_RESC)}, 0")
    emit("s_nop 7")
    emit(f"s_branch {back}")


def dma_ops(kbuf_imm, vbuf_imm, do_k=True, do_v=True, st=0):
    """[m0K, K0, K1, m0V, V0, V1]: one M0 per tensor, the piece index rides on the instruction offset (applied to the global
    and the LDS address alike). K: the per-lane offsets LK[j] carry +(1024 - 1024 j) and S_KBASE carries -1024, so they stay
    >= 0 when rows are clamped to a short sequence (the SADDR form's VGPR is an UNSIGNED 32-bit offset; see gen_fwd_x64.py
    dma_ops). V^T: LV is the same for both pieces (the prepared tile is copied linearly)."""
    o = []
    if do_k:
        o.append(f"    s_add_u32 m0, {s(S_DMAW)}, {kbuf_imm}")
        o += [f"    global_load_lds_dwordx4 {v(LK[j])}, {sr(TBS[st])} offset:{1024 * j}" for j in range(PIECES_K)]
    if do_v:
        o.append(f"    s_add_u32 m0, {s(S_DMAWV)}, {V_REGION + vbuf_imm}")
        o += [f"    global_load_lds_dwordx4 {v(LV)}, {sr(VBS[st])} offset:{1024 * j}" for j in range(PIECES_V)]
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


def distribute(queue, post, start, cap=0):
    q = list(queue)
    ng = len(post)
    if cap <= 0:
        total = sum(n_fill(post[t]) for t in range(start, ng)) + n_fill(q)
        cap = -(-total // (ng - start))
    for t in range(start, ng):
        while q and n_fill(post[t]) < cap:
            post[t].append(q.pop(0))
            while q and isinstance(q[0], str) and q[0].endswith(":"):
                post[t].append(q.pop(0))
    post[ng - 1] += q


deferred = []
QK_ORDER = [(sx, kb, qb) for sx in range(NSX) for kb in (0, 1) for qb in QBS]      # dependent pairs are 4 MFMAs apart (one q-block: 2)
PV_ORDER = [(db, qb) for db in range(ND) for qb in QBS] + ([(ROWSUM, qb) for qb in QBS] if LMFMA else [])
K_FRAGS = [(NSX * kb + sx, t) for sx in range(NSX) for kb in (0, 1) for t in (0, 1)]  # sx = 0 fragments first


def step(variant):
    cur, nxt = variant, variant ^ 1
    kbuf_read = cur * KV_TILE                # K(i+2)
    kbuf_stage = nxt * KV_TILE               # K(i+3) goes where K(i+1) was
    vbuf_cur = cur * KV_TILE                 # V^T(i)
    vbuf_stage = nxt * KV_TILE               # V^T(i+1)

    # ---- phase 1
    post = [[] for _ in range(NG)]
    mf = [mfma_qk(nxt, kb, sx, qb) for (sx, kb, qb) in QK_ORDER]
    for g, op in zip(DMA_GAPS, dma_ops(kbuf_stage, vbuf_stage, st=variant)):
        post[g].append(op)
    for f, (db, t) in enumerate([(db, t) for db in range(ND) for t in (0, 1)]):
        v0 = 0 if "vspread" in OPT else NG // 2          # vspread (A/B): the V^T fragment reads over the whole of phase 1 instead of its second half
        post[v0 + f * (NG - v0) // (2 * ND)].append(v_read(vbuf_cur, db, t))
    distribute(softmax_stream(cur, list(range(XPAIRS, 16))), post, int(opt_val("smstart", "0")))
    for t in range(NG):
        out.append(mf[t])
        out.extend(post[t])

    # ---- phase 2
    emit("s_nop 1")                          # the last e4m3 converts (VALU writes) -> first PV MFMA (reads them as B)
    pre = [[] for _ in range(NG2)]
    post = [[] for _ in range(NG2)]
    mf = []
    for t, (db, qb) in enumerate(PV_ORDER):
        if qb == 0 and db != ROWSUM:
            pre[t].append(("WAIT", ("v", db, 1)))
        mf.append(mfma_pv(cur, db, qb))
        if "klate" in OPT:
            assert len(K_FRAGS) == NG2, "klate: one K fragment read per phase-2 gap (the head_dim-128 exp / lvalu bodies only)"
            post[t].append(k_read(kbuf_read, *K_FRAGS[t]))
        elif t < len(K_FRAGS) // 2:              # K(i+2) fragments in the first half: nothing young is left for the drain
            post[t] += [k_read(kbuf_read, *K_FRAGS[2 * t]), k_read(kbuf_read, *K_FRAGS[2 * t + 1])]
    rare, back = new_label("rare"), new_label("rare_back")
    fl, flback = new_label("flush"), new_label("flush_back")
    # next step stages K(i+4) and V^T(i+2): global addresses from the tile-address table the C++ shell built in LDS
    # (gen_fwd_x64.py step(): TABV = &tab[i+2], entries {K address, V^T tile address}, padded by 4 copies of the last one)
    st2 = variant ^ 1
    vq = [("LDS", f"ds_read_b64 {vr(T[4], 2)}, {v(TABV)} offset:8", "tabv"),
          ("LDS", f"ds_read_b64 {vr(T[6], 2)}, {v(TABV)} offset:32", "tabk")]
    n_head = len(vq)
    rm = row_max_ops(nxt)
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
    inv, invback = new_label("inval"), new_label("inval_back")
    vq += stats_ops(rare, back, fl, flback, inv, invback, sset=nxt)
    deferred.append(lambda: inval_block(inv, invback))
    deferred.append(lambda: rare_rescale_block(rare, back))
    deferred.append(lambda: flush_block(fl, flback))
    vq += softmax_stream(nxt, list(range(XPAIRS)))
    # gap 0 holds only ops that do not read S_nxt (its last MFMA was issued just before this phase)
    post[0] += vq[:n_head]
    distribute(vq[n_head:], post, 1)
    for t in range(NG2):
        out.extend(pre[t])
        out.append(mf[t])
        out.extend(post[t])

    # ---- tail: rare O rescale, drain, barrier
    slow, slow_back = new_label("slow"), new_label("slow_back")
    emit(f"s_cmp_lg_u32 {s(S_RESC)}, 0")
    emit(f"s_cbranch_scc1 {slow}")
    label(slow_back)
    deferred.append(lambda: rescale_o_block(slow, slow_back))
    emit(("DRAIN",))
    if "nobarrier" not in OPT and not ("halfbarrier" in OPT and variant == 0):                 # pricing only
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
    if LIN:
        emit(f"v_mul_f32 {v(T[0])}, 8.0, {v(8)}")             # v8 / v21 still hold c / -c of the parameter block
        emit(f"v_mul_f32 {v(T[1])}, 8.0, {v(21)}")
        emit(f"v_readfirstlane_b32 {s(S_C8)}, {v(T[0])}")
        emit(f"v_readfirstlane_b32 {s(S_NEGC8)}, {v(T[1])}")
        emit(f"s_mov_b32 {s(S_M8)}, 0x{float_bits(-8.0):08x}")
    emit(f"s_sub_u32 {s(S_NM1)}, {s(S_NTILES)}, 1")
    emit(f"s_lshl_b32 {s(S_DMAW)}, {s(S_WAVE)}, {10 + PIECES_K.bit_length() - 1}")          # 2 KiB of every 8 KiB tile per wave (head_dim 64: 1 KiB of 4; 192 / 256: 4 of 16)
    emit(f"s_add_u32 {s(S_DMAW)}, {s(S_DMAW)}, {s(S_LDS)}")
    if S_DMAWV != S_DMAW:                                     # head_dim 192: the prepared V^T tile is 12 KiB, 3 pieces per wave
        emit(f"s_mul_i32 {s(S_DMAWV)}, {s(S_WAVE)}, {1024 * PIECES_V}")
        emit(f"s_add_u32 {s(S_DMAWV)}, {s(S_DMAWV)}, {s(S_LDS)}")
    emit(f"s_mov_b32 {s(S_I)}, 0")
    emit(f"s_mov_b32 {s(S_RESC)}, 0")
    emit(f"v_mov_b32 {v(NEGINF)}, 0xff800000")
    emit(f"v_mov_b32 {v(VSC)}, 0x7f7f7f7f")
    if LMFMA:
        for r in range(8):
            emit(f"v_mov_b32 {v(ONES + r)}, 0x38383838")      # e4m3 1.0 in every byte

    emit("; ---- per-lane constants")
    emit(f"v_lshrrev_b32 {v(T[0])}, 5, {v(LANE)}")            # hh
    emit(f"v_lshlrev_b32 {v(HH4)}, 2, {v(T[0])}")
    emit(f"v_and_b32 {v(T[1])}, 31, {v(LANE)}")               # l31
    # K fragment addresses: lds + l31*128 + (((4 sx + 2 t + hh) ^ ((l31 >> 1) & 7)) << 4); WIDE: lds + l31*256 + (((4 sx + 2 t + hh) ^ (l31 & 15)) << 4)
    if WIDE:
        emit(f"v_and_b32 {v(T[2])}, 15, {v(T[1])}")
        emit(f"v_lshlrev_b32 {v(T[3])}, 8, {v(T[1])}")
    else:
        emit(f"v_lshrrev_b32 {v(T[2])}, 1, {v(T[1])}")
        emit(f"v_and_b32 {v(T[2])}, 7, {v(T[2])}")                # k swizzle
        emit(f"v_lshlrev_b32 {v(T[3])}, 7, {v(T[1])}")
    emit(f"v_add_u32 {v(T[3])}, {s(S_LDS)}, {v(T[3])}")
    for sx in (range(NSX) if WIDE else (0, 1)):
        for t in (0, 1):
            emit(f"v_add_u32 {v(T[4])}, {4 * sx + 2 * t}, {v(T[0])}")
            emit(f"v_xor_b32 {v(T[4])}, {v(T[4])}, {v(T[2])}")
            emit(f"v_lshl_add_u32 {v(KADDR[2 * sx + t])}, {v(T[4])}, 4, {v(T[3])}")
    # V^T fragment addresses: lds + l31*64 + (((2 t + hh) ^ ((l31 >> 2) & 3)) << 4)   (V_REGION rides on the offsets)
    emit(f"v_lshrrev_b32 {v(T[2])}, 2, {v(T[1])}")
    emit(f"v_and_b32 {v(T[2])}, 3, {v(T[2])}")
    emit(f"v_lshlrev_b32 {v(T[3])}, 6, {v(T[1])}")
    emit(f"v_add_u32 {v(T[3])}, {s(S_LDS)}, {v(T[3])}")
    for t in (0, 1):
        emit(f"v_add_u32 {v(T[4])}, {2 * t}, {v(T[0])}")
        emit(f"v_xor_b32 {v(T[4])}, {v(T[4])}, {v(T[2])}")
        emit(f"v_lshl_add_u32 {v(VADDR[t])}, {v(T[4])}, 4, {v(T[3])}")
    # DMA lane offsets. K piece j of this wave: rows 16 w + 8 j + rip (rip = lane >> 3), source chunk cpos ^ ((row >> 1) & 7)
    emit(f"v_lshrrev_b32 {v(T[6])}, {4 if WIDE else 3}, {v(LANE)}")            # rip
    emit(f"v_and_b32 {v(T[7])}, {15 if WIDE else 7}, {v(LANE)}")                # cpos
    if WIDE:
        # four pieces per wave: piece j = rows 16 w + 4 j + rip (rip = lane >> 4) of 16 chunks; LDS chunk cpos holds source chunk cpos ^ (row & 15)
        # = cpos ^ (4 j + rip). head_dim 192: source chunks 12..15 do not exist - clamped to 11 (filler: positions no fragment read touches)
        emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, 4")
        for j in range(4):
            emit(f"v_add_u32 {v(T[4])}, {4 * j}, {v(T[6])}")              # row & 15
            emit(f"v_xor_b32 {v(T[5])}, {v(T[4])}, {v(T[7])}")            # source chunk
            if D == 192:
                emit(f"v_min_u32 {v(T[5])}, 11, {v(T[5])}")
            emit(f"v_add_u32 {v(T[4])}, {s(S_T0)}, {v(T[4])}")            # row
            emit(f"v_min_i32 {v(T[4])}, {v(T[4])}, {s(S_LASTROW)}")       # seqlen_k < 64: rows of the only tile stay inside K
            emit(f"v_mul_lo_u32 {v(LK[j])}, {v(T[4])}, {s(S_KRS)}")
            emit(f"v_lshl_add_u32 {v(LK[j])}, {v(T[5])}, 4, {v(LK[j])}")
            if j < 3:
                emit(f"v_add_u32 {v(LK[j])}, {1024 * (3 - j)}, {v(LK[j])}")   # +3072 - 1024 j; S_KBASE carries -3072
    if D == 64:
        # one piece per wave: pseudo-row R = 8 w + rip, LDS chunk cpos holds source chunk cs = cpos ^ ((R >> 1) & 7) = chunk cs & 3 of key R + 32 (cs >> 2)
        emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, 3")
        emit(f"v_add_u32 {v(T[4])}, {s(S_T0)}, {v(T[6])}")
        emit(f"v_lshrrev_b32 {v(T[5])}, 1, {v(T[4])}")
        emit(f"v_and_b32 {v(T[5])}, 7, {v(T[5])}")
        emit(f"v_xor_b32 {v(T[5])}, {v(T[5])}, {v(T[7])}")
        emit(f"v_lshrrev_b32 {v(T[8])}, 2, {v(T[5])}")
        emit(f"v_lshl_add_u32 {v(T[4])}, {v(T[8])}, 5, {v(T[4])}")
        emit(f"v_and_b32 {v(T[5])}, 3, {v(T[5])}")
        emit(f"v_min_i32 {v(T[4])}, {v(T[4])}, {s(S_LASTROW)}")   # seqlen_k < 64: rows of the only tile stay inside K
        emit(f"v_mul_lo_u32 {v(LK[0])}, {v(T[4])}, {s(S_KRS)}")
        emit(f"v_lshl_add_u32 {v(LK[0])}, {v(T[5])}, 4, {v(LK[0])}")
        emit(f"v_add_u32 {v(LK[0])}, 1024, {v(LK[0])}")            # S_KBASE carries -1024 (as at head_dim 128)
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, 4")
    for j in ((0, 1) if D in (96, 128) else ()):
        emit(f"v_add_u32 {v(T[4])}, {s(S_T0)}, {v(T[6])}")
        if j:
            emit(f"v_add_u32 {v(T[4])}, 8, {v(T[4])}")
        emit(f"v_lshrrev_b32 {v(T[5])}, 1, {v(T[4])}")
        emit(f"v_and_b32 {v(T[5])}, 7, {v(T[5])}")
        emit(f"v_xor_b32 {v(T[5])}, {v(T[5])}, {v(T[7])}")
        if D == 96:
            emit(f"v_min_u32 {v(T[5])}, 5, {v(T[5])}")        # a row has six 16-byte chunks: positions whose source would be chunk 6 / 7 take a copy of chunk 5
        emit(f"v_min_i32 {v(T[4])}, {v(T[4])}, {s(S_LASTROW)}")   # seqlen_k < 64: rows of the only tile stay inside K
        emit(f"v_mul_lo_u32 {v(LK[j])}, {v(T[4])}, {s(S_KRS)}")
        emit(f"v_lshl_add_u32 {v(LK[j])}, {v(T[5])}, 4, {v(LK[j])}")
        if not j:
            emit(f"v_add_u32 {v(LK[j])}, 1024, {v(LK[j])}")            # +1024 - 1024 j; S_KBASE carries -1024
    if WIDE:
        emit(f"s_mul_i32 {s(S_T0)}, {s(S_WAVE)}, {1024 * PIECES_V}")
    else:
        emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, {10 + NSX - 1}")
    emit(f"v_lshl_add_u32 {v(LV)}, {v(LANE)}, 4, {s(S_T0)}")  # 2048 w + 16 lane (head_dim 64: 1024 w; 192 / 256: 3072 / 4096 w): linear copy of the prepared tile

    emit("; ---- Q fragments -> AGPRs: row q_row0 + 64 w + 32 qb + l31, d = 64 sx + 32 t + 16 hh + [0,16); rows past seqlen_q are ZERO")
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, {5 + NQB - 1}")      # 64 rows per wave (one q-block: 32)
    emit(f"s_add_u32 {s(S_T0)}, {s(S_T0)}, {s(S_QROW0)}")
    emit(f"s_sub_u32 {s(S_T1)}, {s(S_SEQLENQ)}, 1")
    emit(f"v_lshlrev_b32 {v(T[6])}, 4, {v(T[0])}")            # hh * 16 bytes
    for qb in QBS:
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
        for sx in range(NSX):
            for t in (0, 1):
                if 64 * sx + 32 * t >= D:            # head_dim 96: d 96..127 does not exist - zero fragments (they meet the K tile's filler chunks)
                    for r in range(4):
                        emit(f"v_mov_b32 {v(16 * qb + 8 * sx + 4 * t + r)}, 0")
                    continue
                emit(f"global_load_dwordx4 {vr(16 * qb + 8 * sx + 4 * t, 4)}, {vr(T[4], 2)}, off offset:{64 * sx + 32 * t}")
    emit("s_waitcnt vmcnt(0)")
    for qb in QBS:
        emit(f"v_cmp_gt_i32 vcc, {s(S_SEQLENQ)}, {v(QROW[qb])}")
        for r in range(8 * NSX):
            emit(f"v_cndmask_b32 {v(16 * qb + r)}, 0, {v(16 * qb + r)}, vcc")
    for qb in QBS:
        for r in range(8 * NSX):
            emit(f"v_accvgpr_write_b32 a{128 + 16 * qb + r}, {v(16 * qb + r)}")
    emit("; ---- state")
    for r in list(range(128)) + (list(range(LSUM(0), LSUM(NQB - 1) + 16)) if LMFMA else []):
        emit(f"v_accvgpr_write_b32 a{r}, 0")
    for qb in QBS:
        emit(f"v_mov_b32 {v(MTRUE[qb])}, 0xff800000")
        if not LMFMA:
            emit(f"v_mov_b32 {v(L0[qb])}, 0")
            emit(f"v_mov_b32 {v(L1[qb])}, 0")
        emit(f"v_mov_b32 {v(ALPHA[qb])}, 1.0")

    emit("; ---- tile addresses of positions 1..3 from the table; K(0) fragments -> AGPRs, S(0) = K(0) Q^T, then K(1) fragments")
    emit(f"v_mov_b32 {v(T[6])}, {s(S_TAB)}")
    emit(f"ds_read_b64 {vr(T[8], 2)}, {v(T[6])} offset:32")          # tab[2].k : K(2), staged below
    emit(f"ds_read_b64 {vr(T[10], 2)}, {v(T[6])} offset:48")         # tab[3].k : K(3), staged by step 0
    emit(f"ds_read_b64 {vr(T[12], 2)}, {v(T[6])} offset:24")         # tab[1].v : V^T(1), staged by step 0
    emit(f"v_add_u32 {v(TABV)}, 32, {v(T[6])}")                      # step 0 reads tab[2].v and tab[4].k
    for (j, t) in K_FRAGS:
        emit(k_read(0, j, t))
    emit(("DRAIN",))
    emit(f"v_readfirstlane_b32 {s(TBS[0])}, {v(T[8])}")
    emit(f"v_readfirstlane_b32 {s(TBS[0] + 1)}, {v(T[9])}")
    emit("s_nop 7")                                           # v_accvgpr_write (Q) / ds_read (K) -> MFMA operand reads
    for (sx, kb, qb) in QK_ORDER:
        out.append(mfma_qk(0, kb, sx, qb))
    for (j, t) in K_FRAGS:
        emit(k_read(KV_TILE, j, t))
    emit(("DRAIN",))
    emit("s_barrier")                                          # every wave has read K(0) and K(1): both K buffers are free
    for it in dma_ops(0, 0, do_k=True, do_v=False):            # K(2) -> K buffer 0. V^T(1) / K(3) are staged by step 0.
        out.append(it)
        if "m0" in it:
            emit("s_nop 0")
    for dst, src in ((TBS[0], T[10]), (TBS[0] + 1, T[11]), (VBS[0], T[12]), (VBS[0] + 1, T[13])):   # step 0 stages K(3), V^T(1)
        emit(f"v_readfirstlane_b32 {s(dst)}, {v(src)}")
    emit("s_nop 15")                                           # S(0): the last MFMA's results before the VALU reads them
    emit("s_nop 15")
    nomask = new_label("nomask")
    emit(f"s_cmp_eq_u32 {s(S_FIRSTLAST)}, 1")                  # seqlen-k mask: first walked tile only (mask.h:44-78), if it is tile k_tiles-1
    emit(f"s_cbranch_scc0 {nomask}")
    emit(f"s_cmp_lt_i32 {s(S_TAILVALID)}, 64")
    emit(f"s_cbranch_scc0 {nomask}")
    for kb in range(2):
        for r in range(16):
            key = 32 * kb + (r & 3) + 8 * (r >> 2)
            emit(f"v_add_u32 {v(T[0])}, {key}, {v(HH4)}")
            emit(f"v_cmp_gt_i32 vcc, {s(S_TAILVALID)}, {v(T[0])}")
            for qb in QBS:
                emit(f"v_cndmask_b32 {v(S_(0, qb, kb) + r)}, {v(NEGINF)}, {v(S_(0, qb, kb) + r)}, vcc")
    label(nomask)
    for op in row_max_ops(0):
        out.append(op)
    for qb in QBS:
        emit(f"v_mov_b32 {v(T[qb])}, {v(MLOC[qb])}")
    emit("s_nop 1")
    for qb in QBS:
        emit(f"v_permlane32_swap_b32 {v(MLOC[qb])}, {v(T[qb])}")
    emit("s_nop 1")
    for qb in QBS:
        emit(f"v_max_f32 {v(MTRUE[qb])}, {v(MLOC[qb])}, {v(T[qb])}")
    for qb in QBS:
        emit(f"v_mov_b32 {v(MREF[qb])}, {v(MTRUE[qb])}")
        set_nms(qb)
        emit(f"v_add_f32 {v(MTHR[qb])}, {s(S_TAU)}, {v(MTRUE[qb])}")
    emit(f"s_mov_b32 {s(S_DOMASK)}, 1")                        # position 0 is never flagged; position 1 votes into bit 1
    emit(f"s_mov_b32 {s(S_BIT)}, 2")
    emit(f"s_mov_b32 {s(S_DOWORD)}, {s(S_DOFLAGS)}")
    if MX:
        out.extend(mx_ops(0, MTRUE))
    for op in softmax_stream(0, list(range(XPAIRS))):
        out.append(op)
    emit(("DRAIN",))
    emit("s_barrier")


def epilogue():
    emit("; ---- flush the last (partial) vote word")
    nofl = new_label("nolastflush")
    emit(f"s_cmp_eq_u32 {s(S_DOMASK)}, 0")
    emit(f"s_cbranch_scc1 {nofl}")
    flush_domask()
    label(nofl)
    emit("s_nop 15")                                           # the last PV MFMAs (16 passes each) have written the accumulators
    emit("s_nop 15")
    emit("s_nop 15")
    emit("s_nop 15")
    if LMFMA:
        globals()["LSUM_AGPR"] = [LSUM(qb) for qb in QBS]      # l~ of the lane's row: register 0 of the row-sum accumulators
    store_epilogue(globals(), O_)                              # gen_epilogue.py: v_descale / l, bf16 O and LSE from the registers
    emit("s_waitcnt lgkmcnt(0)")


def main():
    prologue()
    loop, done = new_label("loop"), new_label("done")
    # Code placement (round 5; see gen_fwd_x64.py main()): the loop head is pinned at the best measured phase inside a 32-byte window -
    # default form 0, exact-exp 24 (30.65 ms against 31.25 at phases 8 / 16: 2 %), exact-rowsum 8 (flat). `align:N` / `pad4:N` override.
    if opt_val("align", "") or opt_val("pad4", ""):
        if opt_val("align", ""):
            out.append(f".p2align {opt_val('align', '')}")
        for _ in range(int(opt_val("pad4", "0"))):
            emit("s_nop 0")
    else:
        out.append(".p2align 5")
        for _ in range((8 if not LMFMA else (0 if LIN else 24)) // 4):
            emit("s_nop 0")
    label(loop)
    emit(f"s_cmp_lt_u32 {s(S_I)}, {s(S_NTILES)}")
    emit(f"s_cbranch_scc0 {done}")
    step(0)
    emit(f"s_cmp_lt_u32 {s(S_I)}, {s(S_NTILES)}")
    emit(f"s_cbranch_scc0 {done}")
    step(1)
    emit(f"s_branch {loop}")
    for blk in deferred:
        blk()
    label(done)
    epilogue()
    lines = finalize(out)
    text = "\n".join(lines)
    path = sys.argv[1] if len(sys.argv) > 1 else "la_fwd_x64_fp8_body.inc"
    mode = 2 if not LMFMA else (0 if LIN else 1)               # PMODE of the shell (la_fwd_kernel_x64_fp8.hip)
    # the three bodies of the build are told apart by their FILE NAME in the shell's includes: a body generated under options that belong to
    # another name (e.g. a global LA_X64F8_OPT in the environment of a default build) must fail here, not at link time
    by_name = 1 if path.endswith("_exp_body.inc") else (2 if path.endswith("_lvalu_body.inc") else (0 if path.endswith(("la_fwd_x64_fp8_body.inc", "la_fwd_x64_fp8_d64_body.inc", "la_fwd_x64_fp8_d96_body.inc", "la_fwd_x64_fp8_d192_body.inc", "la_fwd_x64_fp8_d256_body.inc")) else mode))
    tag = "" if D == 128 else f"_d{D}_"
    if tag not in os.path.basename(path) or (D == 128 and any(f"_d{d}_" in os.path.basename(path) for d in (64, 96, 192, 256))):
        raise SystemExit(f"{path}: generated for head_dim {D} (LA_X64F8_D) but named like another head dim's body")
    if by_name != mode:
        raise SystemExit(f"{path}: generated with the options of P mode {mode} (LA_X64F8_OPT={os.environ.get('LA_X64F8_OPT', '')!r}) "
                         f"but named like the body of P mode {by_name}")
    if D == 128:     # (TAU and the offset of P depend on the form of P, not on the head dim: one header per form)
        with open(path.replace("_body.inc", "_consts.h"), "w") as f:
            f.write("// GENERATED by gen_fwd_x64_fp8.py together with the body of the same name — do not edit.\n")
            f.write(f"#define LA_X64F8_TAU_{mode} {TAU!r}f\n#define LA_X64F8_OFFSET_{mode} {P_OFFSET!r}f\n")
    with open(path, "w") as f:
        f.write("// GENERATED by gen_fwd_x64_fp8.py — do not edit. Inline-asm body of la_fwd_x64_fp8_kernel.\n")
        f.write(option_tag() + "\n")
        f.write('R"ASM(\n' + text + '\n)ASM"\n')
    print(f"wrote {path}: {len(lines)} lines, {text.count('v_mfma')} MFMAs")


if __name__ == "__main__":
    main()
