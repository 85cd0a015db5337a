#!/usr/bin/env python
"""Generates la_fwd_asm_body.inc: the hand-scheduled gfx950 main loop of the bf16 / head_dim-128 QK-Skip forward.

Same algorithm, LDS image and per-lane data layout as la_fwd_kernel_v2.hip (see its header); this file only fixes
the instruction ORDER and the REGISTER ALLOCATION, which hipcc (capped at 256 VGPRs by the two-waves-per-SIMD design)
cannot get right: it wants ~330 VGPRs for the loop and therefore sinks every LDS fragment read next to its MFMA — a
read -> wait -> MFMA latency chain worth 25 % of the kernel.

Schedule of one step (tile i in S_cur, K(i+1) / V(i) resident in LDS):
  head     seq[i+1], seq[i+2] from LDS; first 4 K fragments of K(i+1); 8 LDS-DMA pieces (K(i+2), V(i+1)) with an SGPR
           tile base + per-lane 32-bit offset (no per-piece VALU address arithmetic)
  phase 1  16 x { wait K fragment j (counted lgkmcnt, fragments 4 MFMAs ahead) ; MFMA S_nxt += K_j Q ; issue K
           fragment j+4 ; 7 VALU: one pair of P = exp2(S*c - m*c), row-sum adds, cvt to bf16 }
  phase 2  16 x { wait V^T fragment m (4 MFMAs ahead) ; MFMA O += V_m P ; issue V^T fragment m+4 ; 1-2 VALU of the
           row max of S_nxt }
  tail     half-wave max exchange, running max, skip vote into an SGPR bit mask, alpha, rare O rescale, barrier.

Everything the block needs comes from an LDS parameter block written by the C++ prologue; results (O^T, m, l) go
back through LDS (the K/V buffers are free after the last barrier). The block owns v0-v247, s30-s79, vcc, m0, scc.
"""
import os
import sys

# Experiment switches (comma list in LA_ASM_OPT): ablations produce WRONG results and exist only to price a component.
OPT = set(x for x in os.environ.get("LA_ASM_OPT", "").split(",") if x)

# ---------------------------------------------------------------- register map (VGPR)
O = [0, 16, 32, 48]                  # O^T accumulators, 4 d-blocks x 16
SA = [64, 80]                        # S^T ping (key block 0 / 1)
SB = [96, 112]                       # S^T pong
PF = [128, 132, 136, 140]            # P^T as bf16 (k-step kk: 4 regs)
KF = [144, 148, 152, 156]            # K fragment ring
VF = [160, 164, 168, 172]            # V^T fragment ring (lo 2 regs, hi 2 regs)
KADDR = list(range(176, 184))        # per-lane LDS address of K row, chunk of k-step ks (buffer/key-block via imm)
VADDR = list(range(184, 188))        # per-lane LDS address of V^T fragment base for d-block db
MRUN, LRUN, ALPHA, PSUM, NMS = 190, 191, 192, 193, 194   # running max, partial row sum, alpha, tile sum, -m*c
T = [196, 197, 198, 199, 200, 201, 202, 203, 195]   # temporaries; (T[2],T[3]) and (T[4],T[5]) are even-aligned pairs
LK = [204, 205, 206, 207]            # per-lane DMA source offsets of K pieces 0..3 (bytes, relative to the tile base)
LV = [208, 209, 210, 211]            # same for V
HH4, LANE, RIPROW, MLOC, MLOC2 = 212, 213, 214, 215, 188
QROW = 189
Q = [216 + 4 * ks for ks in range(8)]  # Q fragments (B operand), 8 k-steps x 4 regs

# ---------------------------------------------------------------- register map (SGPR)
S_KBASE, S_VBASE = 30, 74            # 64-bit (s32-s34 are ABI-reserved: not used)
S_KRS, S_VRS, S_LASTROW, S_NTILES, S_C, S_THR = 76, 35, 36, 37, 38, 39
S_TAILVALID, S_KTM1, S_SEQ, S_DOFLAGS, S_WAVE, S_I, S_DOMASK = 40, 41, 42, 43, 44, 45, 46
S_NCUR, S_N1, S_N2 = 47, 48, 49
S_TB = 50                            # 64-bit tile base temp (s50:51)
S_EXEC = 52                          # 64-bit exec save
S_LDS, S_T0, S_T1, S_NM1 = 54, 55, 56, 57
S_QBASE, S_QRS, S_QROW0, S_SEQLENQ, S_EXPORT = 58, 60, 61, 62, 63
S_T64 = 64                           # 64-bit temp (s64:65)
S_PARAM, S_HASNEXT, S_NEGC, S_T2, S_T3 = 66, 67, 68, 69, 70
S_VB = 72                            # 64-bit V tile base temp (s72:73)
S_SAFEROW, S_DMAW, S_RAG = 71, 77, 78   # max(seqlen_k-64, 0); LDS base of this wave's DMA pieces; ragged-tile flags of the step

def opt_val(key, default):
    for o in OPT:
        if o.startswith(key + ":"):
            return o[len(key) + 1:]
    return default


DMA_PLACE = opt_val("dma", "head")      # where the 8 LDS-DMA pieces of a step are issued
ILV = "ilv" in OPT                      # alternate accumulators between consecutive MFMAs

KV_TILE = 16384
V_REGION = 32768

out = []


def emit(s):
    out.append("    " + s)


def label(s):
    out.append(s + ":")


def v(i):
    return f"v{i}"


def vr(a, n):
    return f"v[{a}:{a + n - 1}]"


def s(i):
    return f"s{i}"


def sr(a, n=2):
    return f"s[{a}:{a + n - 1}]"


class Lgkm:
    """Counted-wait bookkeeping for LDS operations issued by this wave (they return in order)."""

    def __init__(self):
        self.issued = []
        self.fake = set()

    def issue(self, tag):
        self.issued.append(tag)

    def wait_for(self, tag):
        if tag in self.fake:
            return
        idx = max(i for i, t in enumerate(self.issued) if t == tag)
        after = len(self.issued) - 1 - idx
        emit(f"s_waitcnt lgkmcnt({after})")
        # everything up to idx has completed
        self.issued = self.issued[idx + 1:]

    def drain(self):
        if self.issued:
            emit("s_waitcnt lgkmcnt(0)")
        self.issued = []


def mfma(dst, a, b, c_init_zero=False):
    c = "0" if c_init_zero else vr(dst, 16)
    emit(f"v_mfma_f32_32x32x16_bf16 {vr(dst, 16)}, {vr(a, 4)}, {vr(b, 4)}, {c}")


def k_read(lg, slot, kbuf_imm, j):
    """K(i+1) fragment j = 8*kb + ks from K buffer at byte immediate kbuf_imm."""
    kb, ks = j >> 3, j & 7
    if "nokread" in OPT:
        lg.issue(("k", j)); lg.issued.pop(); lg.fake.add(("k", j))
        return
    emit(f"ds_read_b128 {vr(KF[slot], 4)}, {v(KADDR[ks])} offset:{kbuf_imm + kb * 8192}")
    lg.issue(("k", j))


def v_read(lg, slot, vbuf_imm, m):
    """V^T fragment m = 4*db + kk: two transpose reads (keys +0 and +8)."""
    db, kk = m >> 2, m & 3
    if "novread" in OPT:
        lg.fake.add(("v", m, 1))
        return
    emit(f"ds_read_b64_tr_b16 {vr(VF[slot], 2)}, {v(VADDR[db])} offset:{vbuf_imm + kk * 4096}")
    lg.issue(("v", m, 0))
    emit(f"ds_read_b64_tr_b16 {vr(VF[slot] + 2, 2)}, {v(VADDR[db])} offset:{vbuf_imm + kk * 4096 + 2048}")
    lg.issue(("v", m, 1))


def softmax_pair(scur, pidx):
    """7 VALU: P for accumulator elements 2*pidx, 2*pidx+1 of the 32 (kb = e>>4, r = e&15)."""
    if "nosoftmax" in OPT:
        return
    e0, e1 = 2 * pidx, 2 * pidx + 1
    r0 = scur[e0 >> 4] + (e0 & 15)
    r1 = scur[e1 >> 4] + (e1 & 15)
    emit(f"v_fma_f32 {v(T[0])}, {v(r0)}, {s(S_C)}, {v(NMS)}")
    emit(f"v_fma_f32 {v(T[1])}, {v(r1)}, {s(S_C)}, {v(NMS)}")
    emit(f"v_exp_f32 {v(r0)}, {v(T[0])}")
    emit(f"v_exp_f32 {v(r1)}, {v(T[1])}")
    emit(f"v_add_f32 {v(PSUM)}, {v(PSUM)}, {v(r0)}")
    emit(f"v_add_f32 {v(T[2])}, {v(T[2])}, {v(r1)}")          # second partial sum chain
    # pf register for elements (e0, e1): k-step kk = 2*kb + (r>>3), word (r&7)>>1
    kb, r = e0 >> 4, e0 & 15
    kk, w = 2 * kb + (r >> 3), (r & 7) >> 1
    emit(f"v_cvt_pk_bf16_f32 {v(PF[kk] + w)}, {v(r0)}, {v(r1)}")


def dma_fast(n_sgpr, do_k, kbuf_imm, do_v, vbuf_imm, sk_tmp, sv_tmp):
    """8 (or 4) LDS-DMA pieces of tile n (SGPR): tile base in SGPRs + per-lane offsets LK/LV."""
    # base = tensor_base + n*64 * row_stride   (SALU, 64-bit); the wave's 16-row slice is in the per-lane offsets
    emit(f"s_lshl_b32 {s(S_T0)}, {s(n_sgpr)}, 6")
    if do_k:
        emit(f"s_mul_hi_u32 {s(sk_tmp + 1)}, {s(S_T0)}, {s(S_KRS)}")
        emit(f"s_mul_i32 {s(sk_tmp)}, {s(S_T0)}, {s(S_KRS)}")
        emit(f"s_add_u32 {s(sk_tmp)}, {s(sk_tmp)}, {s(S_KBASE)}")
        emit(f"s_addc_u32 {s(sk_tmp + 1)}, {s(sk_tmp + 1)}, {s(S_KBASE + 1)}")
    if do_v:
        emit(f"s_mul_hi_u32 {s(sv_tmp + 1)}, {s(S_T0)}, {s(S_VRS)}")
        emit(f"s_mul_i32 {s(sv_tmp)}, {s(S_T0)}, {s(S_VRS)}")
        emit(f"s_add_u32 {s(sv_tmp)}, {s(sv_tmp)}, {s(S_VBASE)}")
        emit(f"s_addc_u32 {s(sv_tmp + 1)}, {s(sv_tmp + 1)}, {s(S_VBASE + 1)}")
    # LDS destination of this wave's piece j: region + buf + (4*wave + j)*1024
    emit(f"s_lshl_b32 {s(S_T1)}, {s(S_WAVE)}, 12")
    emit(f"s_add_u32 {s(S_T1)}, {s(S_T1)}, {s(S_LDS)}")
    for j in range(4):
        if do_k:
            emit(f"s_add_u32 m0, {s(S_T1)}, {kbuf_imm + j * 1024}")
            emit("s_nop 0")
            emit(f"global_load_lds_dwordx4 {v(LK[j])}, {sr(sk_tmp)}")
        if do_v:
            emit(f"s_add_u32 m0, {s(S_T1)}, {V_REGION + vbuf_imm + j * 1024}")
            emit("s_nop 0")
            emit(f"global_load_lds_dwordx4 {v(LV[j])}, {sr(sv_tmp)}")


def dma_ragged(n_sgpr, do_k, kbuf_imm, do_v, vbuf_imm):
    """Slow path for a tile whose rows run past seqlen_k: per-lane clamped row (rare: at most one tile per walk)."""
    emit(f"s_lshl_b32 {s(S_T0)}, {s(n_sgpr)}, 6")
    emit(f"s_lshl_b32 {s(S_T1)}, {s(S_WAVE)}, 12")
    emit(f"s_add_u32 {s(S_T1)}, {s(S_T1)}, {s(S_LDS)}")
    for j in range(4):
        # row = min(n*64 + 16*wave + 4j + rip, last_row)
        emit(f"v_add_u32 {v(T[3])}, {s(S_T0)}, {v(RIPROW)}")
        if j:
            emit(f"v_add_u32 {v(T[3])}, {4 * j}, {v(T[3])}")
        emit(f"v_min_i32 {v(T[3])}, {v(T[3])}, {s(S_LASTROW)}")
        for (flag, rs, base, lane_off, region, buf) in ((do_k, S_KRS, S_KBASE, "k", 0, kbuf_imm),
                                                        (do_v, S_VRS, S_VBASE, "v", V_REGION, vbuf_imm)):
            if not flag:
                continue
            emit(f"v_mad_u64_u32 {vr(T[4], 2)}, {sr(S_T64)}, {v(T[3])}, {s(rs)}, 0")
            # swizzled chunk offset inside the row: K: (cpos ^ ((4j+rip)&15))<<4 ; V: (cpos ^ (rip<<2))<<4
            if lane_off == "k":
                emit(f"v_xor_b32 {v(T[6])}, {j << 6}, {v(T[7])}")      # T7 = (cpos ^ rip) << 4
            else:
                emit(f"v_mov_b32 {v(T[6])}, {v(T[8])}")                # T8 = (cpos ^ (rip<<2)) << 4
            emit(f"v_add_co_u32 {v(T[4])}, vcc, {v(T[4])}, {v(T[6])}")
            emit(f"v_addc_co_u32 {v(T[5])}, vcc, 0, {v(T[5])}, vcc")
            emit(f"v_add_co_u32 {v(T[4])}, vcc, {s(base)}, {v(T[4])}")
            emit(f"v_mov_b32 {v(T[6])}, {s(base + 1)}")
            emit(f"v_addc_co_u32 {v(T[5])}, vcc, {v(T[5])}, {v(T[6])}, vcc")
            emit(f"s_add_u32 m0, {s(S_T1)}, {region + buf + j * 1024}")
            emit("s_nop 0")
            emit(f"global_load_lds_dwordx4 {vr(T[4], 2)}, off")


def dma_bases():
    """Tile bases for the spread placement: K(i+2) -> S_TB, V(i+1) -> S_VB, relative to the tile's first row (clamped so
    that a ragged tile never reads past the tensor; such a tile is re-staged by dma_fixup at the end of the step)."""
    emit(f"s_mov_b32 {s(S_RAG)}, 0")
    for (n_sgpr, rs, base, dst, bit) in ((S_N2, S_KRS, S_KBASE, S_TB, 1), (S_N1, S_VRS, S_VBASE, S_VB, 2)):
        emit(f"s_lshl_b32 {s(S_T0)}, {s(n_sgpr)}, 6")
        emit(f"s_cmp_gt_u32 {s(S_T0)}, {s(S_SAFEROW)}")
        emit(f"s_cselect_b32 {s(S_T1)}, {bit}, 0")
        emit(f"s_or_b32 {s(S_RAG)}, {s(S_RAG)}, {s(S_T1)}")
        emit(f"s_min_u32 {s(S_T0)}, {s(S_T0)}, {s(S_SAFEROW)}")
        emit(f"s_mul_hi_u32 {s(dst + 1)}, {s(S_T0)}, {s(rs)}")
        emit(f"s_mul_i32 {s(dst)}, {s(S_T0)}, {s(rs)}")
        emit(f"s_add_u32 {s(dst)}, {s(dst)}, {s(base)}")
        emit(f"s_addc_u32 {s(dst + 1)}, {s(dst + 1)}, {s(base + 1)}")


def piece_pre(kind, j, kbuf_imm, vbuf_imm):
    imm = (kbuf_imm if kind == "k" else V_REGION + vbuf_imm) + j * 1024
    emit(f"s_add_u32 m0, {s(S_DMAW)}, {imm}")


def piece_load(kind, j):
    if kind == "k":
        emit(f"global_load_lds_dwordx4 {v(LK[j])}, {sr(S_TB)}")
    else:
        emit(f"global_load_lds_dwordx4 {v(LV[j])}, {sr(S_VB)}")


def dma_fixup(kbuf_imm, vbuf_imm):
    """Rare: a staged tile was ragged (rows past seqlen_k): re-stage it with per-lane clamped rows after the fast pieces."""
    done = new_label("fix_done")
    nov = new_label("fix_nov")
    emit(f"s_cmp_eq_u32 {s(S_RAG)}, 0")
    emit(f"s_cbranch_scc1 {done}")
    emit("s_waitcnt vmcnt(0)")
    emit(f"s_bitcmp1_b32 {s(S_RAG)}, 0")
    emit(f"s_cbranch_scc0 {nov}")
    dma_ragged(S_N2, True, kbuf_imm, False, 0)
    label(nov)
    emit(f"s_bitcmp1_b32 {s(S_RAG)}, 1")
    emit(f"s_cbranch_scc0 {done}")
    dma_ragged(S_N1, False, 0, True, vbuf_imm)
    label(done)


PLACEMENTS = {   # piece order V0..V3 then K0..K3 unless stated: (phase, slot) per piece
    "p1": [(1, t) for t in (1, 3, 5, 7, 9, 11, 13, 15)],
    "p1a": [(1, t) for t in range(8)],
    "p1b": [(1, t) for t in range(8, 16)],
    "p2": [(2, t) for t in range(8)],
    "p2s": [(2, t) for t in (0, 2, 4, 6, 8, 10, 12, 14)],
    "mix": [(1, 3), (1, 7), (1, 11), (1, 15), (2, 1), (2, 3), (2, 5), (2, 7)],
}

uid = [0]


def new_label(prefix):
    uid[0] += 1
    return f".L{prefix}_{uid[0]}_%="


def dma_tile(n_sgpr, do_k, kbuf_imm, do_v, vbuf_imm, sk_tmp=S_TB, sv_tmp=S_VB):
    """Fast path unless the tile holds rows past seqlen_k (only tile k_tiles-1 can)."""
    slow, done = new_label("dma_slow"), new_label("dma_done")
    emit(f"s_cmp_eq_u32 {s(n_sgpr)}, {s(S_KTM1)}")
    emit(f"s_cselect_b32 {s(S_T2)}, 1, 0")
    emit(f"s_cmp_lt_i32 {s(S_TAILVALID)}, 64")
    emit(f"s_cselect_b32 {s(S_T3)}, 1, 0")
    emit(f"s_and_b32 {s(S_T2)}, {s(S_T2)}, {s(S_T3)}")
    emit(f"s_cmp_lg_u32 {s(S_T2)}, 0")
    emit(f"s_cbranch_scc1 {slow}")
    dma_fast(n_sgpr, do_k, kbuf_imm, do_v, vbuf_imm, sk_tmp, sv_tmp)
    emit(f"s_branch {done}")
    label(slow)
    dma_ragged(n_sgpr, do_k, kbuf_imm, do_v, vbuf_imm)
    label(done)


def row_max_ops(snxt):
    """Row max of the 32 accumulator values of S_nxt into MLOC: two chains of max3, 17 VALU."""
    ops = []
    regs = [snxt[0] + r for r in range(16)] + [snxt[1] + r for r in range(16)]
    ops.append(f"v_max_f32 {v(MLOC)}, {v(regs[0])}, {v(regs[1])}")
    ops.append(f"v_max_f32 {v(MLOC2)}, {v(regs[2])}, {v(regs[3])}")
    rest = regs[4:]
    chains = [MLOC, MLOC2]
    for n_, i in enumerate(range(0, len(rest), 2)):
        ch = chains[n_ & 1]
        ops.append(f"v_max3_f32 {v(ch)}, {v(ch)}, {v(rest[i])}, {v(rest[i + 1])}")
    ops.append(f"v_max_f32 {v(MLOC)}, {v(MLOC)}, {v(MLOC2)}")
    return ops


def stats_tail(pos_expr_sgpr, valid_sgpr):
    """MLOC holds the in-lane max. Half-wave exchange, running max, vote bit for list position (SGPR), alpha, -m*c."""
    emit(f"v_mov_b32 {v(T[0])}, {v(MLOC)}")
    emit("s_nop 1")
    emit(f"v_permlane32_swap_b32 {v(MLOC)}, {v(T[0])}")
    emit("s_nop 1")
    emit(f"v_max_f32 {v(MLOC)}, {v(MLOC)}, {v(T[0])}")
    if valid_sgpr is not None:      # a clamped duplicate past the end of the walk must not touch the state
        emit(f"s_cmp_lg_u32 {s(valid_sgpr)}, 0")
        emit(f"s_cselect_b64 vcc, -1, 0")
        emit(f"v_mov_b32 {v(T[1])}, 0xff800000")
        emit(f"v_cndmask_b32 {v(MLOC)}, {v(T[1])}, {v(MLOC)}, vcc")
    emit(f"v_mov_b32 {v(T[1])}, {v(MRUN)}")                         # m_prev
    emit(f"v_max_f32 {v(MRUN)}, {v(MRUN)}, {v(MLOC)}")
    # vote: (m_loc - m_prev) * c > thr   (softmax.h:194)
    emit(f"v_sub_f32 {v(T[2])}, {v(MLOC)}, {v(T[1])}")
    emit(f"v_mul_f32 {v(T[2])}, {s(S_C)}, {v(T[2])}")
    emit(f"v_cmp_gt_f32 vcc, {v(T[2])}, {s(S_THR)}")
    emit("s_cmp_lg_u64 vcc, 0")
    emit(f"s_cselect_b32 {s(S_T0)}, 1, 0")
    if valid_sgpr is not None:
        emit(f"s_and_b32 {s(S_T0)}, {s(S_T0)}, {s(valid_sgpr)}")
    emit(f"s_and_b32 {s(S_T1)}, {s(pos_expr_sgpr)}, 31")
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_T0)}, {s(S_T1)}")
    emit(f"s_or_b32 {s(S_DOMASK)}, {s(S_DOMASK)}, {s(S_T0)}")
    # alpha = exp2((m_prev - m_new) * c) ; NMS = -(m_new * c)
    emit(f"v_sub_f32 {v(T[2])}, {v(T[1])}, {v(MRUN)}")
    emit(f"v_mul_f32 {v(T[2])}, {s(S_C)}, {v(T[2])}")
    emit(f"v_exp_f32 {v(ALPHA)}, {v(T[2])}")
    emit(f"v_mul_f32 {v(NMS)}, {s(S_NEGC)}, {v(MRUN)}")
    # flush the vote word when position & 31 == 31 (and the position is real)
    skip = new_label("noflush")
    emit(f"s_cmp_eq_u32 {s(S_T1)}, 31")
    emit(f"s_cbranch_scc0 {skip}")
    if valid_sgpr is not None:
        emit(f"s_cmp_lg_u32 {s(valid_sgpr)}, 0")
        emit(f"s_cbranch_scc0 {skip}")
    flush_domask(pos_expr_sgpr)
    label(skip)


def flush_domask(pos_sgpr):
    """doflags[pos >> 5] |= domask by one lane; domask = 0."""
    emit(f"s_lshr_b32 {s(S_T0)}, {s(pos_sgpr)}, 5")
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_T0)}, 2")
    emit(f"s_add_u32 {s(S_T0)}, {s(S_T0)}, {s(S_DOFLAGS)}")
    emit(f"v_mov_b32 {v(T[3])}, {s(S_T0)}")
    emit(f"v_mov_b32 {v(T[4])}, {s(S_DOMASK)}")
    emit(f"s_mov_b64 {sr(S_EXEC)}, exec")
    emit("s_mov_b64 exec, 1")
    emit(f"ds_or_b32 {v(T[3])}, {v(T[4])}")
    emit(f"s_mov_b64 exec, {sr(S_EXEC)}")
    emit(f"s_mov_b32 {s(S_DOMASK)}, 0")


def rescale_o():
    skip = new_label("norescale")
    emit(f"v_cmp_neq_f32 vcc, 1.0, {v(ALPHA)}")
    emit(f"s_cbranch_vccz {skip}")
    emit("s_nop 7")
    for db in range(4):
        for r in range(0, 16, 2):
            emit(f"v_pk_mul_f32 {vr(O[db] + r, 2)}, {vr(ALPHA, 2)}, {vr(O[db] + r, 2)} op_sel_hi:[0,1]")
    label(skip)


def step(variant):
    """One pipeline step; variant 0: S_cur = SA, S_nxt = SB, cur buffer 0; variant 1 mirrored."""
    scur, snxt = (SA, SB) if variant == 0 else (SB, SA)
    cur = variant
    kbuf_next = (cur ^ 1) * KV_TILE          # K(i+1)
    kbuf_stage = cur * KV_TILE               # K(i+2) goes where K(i) was
    vbuf_cur = cur * KV_TILE                 # V(i)
    vbuf_stage = (cur ^ 1) * KV_TILE         # V(i+1)
    lg = Lgkm()

    # ---- head: indices of the next two tiles, first K fragments, DMA issue
    emit(f"s_add_u32 {s(S_T0)}, {s(S_I)}, 1")
    emit(f"s_cmp_lt_u32 {s(S_T0)}, {s(S_NTILES)}")
    emit(f"s_cselect_b32 {s(S_HASNEXT)}, 1, 0")
    emit(f"s_min_u32 {s(S_T0)}, {s(S_T0)}, {s(S_NM1)}")
    emit(f"s_add_u32 {s(S_T1)}, {s(S_I)}, 2")
    emit(f"s_min_u32 {s(S_T1)}, {s(S_T1)}, {s(S_NM1)}")
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_T0)}, 2")
    emit(f"s_lshl_b32 {s(S_T1)}, {s(S_T1)}, 2")
    emit(f"s_add_u32 {s(S_T0)}, {s(S_T0)}, {s(S_SEQ)}")
    emit(f"s_add_u32 {s(S_T1)}, {s(S_T1)}, {s(S_SEQ)}")
    emit(f"v_mov_b32 {v(T[3])}, {s(S_T0)}")
    emit(f"v_mov_b32 {v(T[4])}, {s(S_T1)}")
    emit(f"ds_read_b32 {v(T[5])}, {v(T[3])}")
    lg.issue("n1")
    emit(f"ds_read_b32 {v(T[6])}, {v(T[4])}")
    lg.issue("n2")
    ord1 = [(t & 1) * 8 + (t >> 1) for t in range(16)] if ILV else list(range(16))     # t -> K fragment j = 8*kb + ks
    ord2 = [(t & 3) * 4 + (t >> 2) for t in range(16)] if ILV else list(range(16))     # t -> V^T fragment m = 4*db + kk
    for t in range(4):
        k_read(lg, t, kbuf_next, ord1[t])
    lg.wait_for("n2")
    emit(f"v_readfirstlane_b32 {s(S_N1)}, {v(T[5])}")
    emit(f"v_readfirstlane_b32 {s(S_N2)}, {v(T[6])}")
    emit("s_nop 3")
    if "dmal2" in OPT:      # ablation: every step re-stages tile 0 (always L2-hot)
        emit(f"s_mov_b32 {s(S_N1)}, 0")
        emit(f"s_mov_b32 {s(S_N2)}, 0")
    plan = {}
    if "nodma" not in OPT:
        if DMA_PLACE == "head":
            dma_tile(S_N2, True, kbuf_stage, False, 0)
            dma_tile(S_N1, False, 0, True, vbuf_stage)
        else:
            dma_bases()
            order = [("v", j) for j in range(4)] + [("k", j) for j in range(4)]
            for pc, where in zip(order, PLACEMENTS[DMA_PLACE]):
                if "dmahalf" in OPT and pc[0] == "v":
                    continue
                plan[where] = pc

    # ---- phase 1: QK^T(i+1) || softmax(i)
    emit(f"v_mov_b32 {v(PSUM)}, 0")
    emit(f"v_mov_b32 {v(T[2])}, 0")
    for t in range(16):
        j = ord1[t]
        lg.wait_for(("k", j))
        kb, ks = j >> 3, j & 7
        pc = plan.get((1, t))
        if pc:
            piece_pre(pc[0], pc[1], kbuf_stage, vbuf_stage)
        if "nomfma1" not in OPT:
            mfma(snxt[kb], KF[t % 4], Q[ks], c_init_zero=(ks == 0))
        if t + 4 < 16:
            k_read(lg, t % 4, kbuf_next, ord1[t + 4])
        elif t >= 12:
            # K reads are all issued: start the V^T fragment ring
            v_read(lg, t - 12, vbuf_cur, ord2[t - 12])
        if pc:
            piece_load(pc[0], pc[1])
        softmax_pair(scur, t)
    emit(f"v_add_f32 {v(PSUM)}, {v(PSUM)}, {v(T[2])}")
    emit(f"v_fma_f32 {v(LRUN)}, {v(LRUN)}, {v(ALPHA)}, {v(PSUM)}")

    # ---- phase 2: PV(i) || row max of S_nxt
    rmax = row_max_ops(snxt)
    # S_nxt key block 1 is written by the last MFMAs of phase 1: its max ops go late (>= 12 wait states after them)
    per_slot = [[] for _ in range(16)]
    for idx, op in enumerate(rmax):
        per_slot[min(15, 2 + idx)].append(op) if idx < 9 else per_slot[min(15, idx)].append(op)
    for t in range(16):
        m = ord2[t]
        lg.wait_for(("v", m, 1))
        db, kk = m >> 2, m & 3
        pc = plan.get((2, t))
        if pc:
            piece_pre(pc[0], pc[1], kbuf_stage, vbuf_stage)
        if "nomfma2" not in OPT:
            mfma(O[db], VF[t % 4], PF[kk])
        if t + 4 < 16:
            v_read(lg, t % 4, vbuf_cur, ord2[t + 4])
        if pc:
            piece_load(pc[0], pc[1])
        if "norowmax" not in OPT:
            for op in per_slot[t]:
                emit(op)
    lg.drain()
    if plan:
        dma_fixup(kbuf_stage, vbuf_stage)

    # ---- tail
    emit(f"s_add_u32 {s(S_T2)}, {s(S_I)}, 1")
    if "notail" not in OPT:
        stats_tail(S_T2, S_HASNEXT)
        rescale_o()
    emit("s_waitcnt lgkmcnt(0)" if "nowaitvm" in OPT else "s_waitcnt vmcnt(0) lgkmcnt(0)")
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
        emit(f"ds_read_b128 {vr(O[0] + 4 * q, 4)}, {v(T[0])} offset:{16 * q}")     # O regs as scratch before init
    emit("s_waitcnt lgkmcnt(0)")
    plist = [S_KBASE, S_KBASE + 1, S_VBASE, S_VBASE + 1, S_KRS, S_VRS, S_LASTROW, S_NTILES, S_C, S_THR, S_TAILVALID,
             S_KTM1, S_SEQ, S_DOFLAGS, S_QBASE, S_QBASE + 1, S_QRS, S_QROW0, S_SEQLENQ, S_EXPORT, S_LDS, S_NEGC]
    for idx, sg in enumerate(plist):
        emit(f"v_readfirstlane_b32 {s(sg)}, {v(O[0] + idx)}")
    emit("s_nop 4")
    emit(f"s_sub_u32 {s(S_NM1)}, {s(S_NTILES)}, 1")
    emit(f"s_sub_u32 {s(S_SAFEROW)}, {s(S_LASTROW)}, 63")
    emit(f"s_max_i32 {s(S_SAFEROW)}, {s(S_SAFEROW)}, 0")
    emit(f"s_lshl_b32 {s(S_DMAW)}, {s(S_WAVE)}, 12")
    emit(f"s_add_u32 {s(S_DMAW)}, {s(S_DMAW)}, {s(S_LDS)}")
    emit(f"s_mov_b32 {s(S_I)}, 0")
    emit(f"s_mov_b32 {s(S_DOMASK)}, 1")        # the first walked tile is never flagged (softmax.h:153)

    emit("; ---- per-lane constants")
    emit(f"v_lshrrev_b32 {v(T[0])}, 5, {v(LANE)}")            # hh
    emit(f"v_lshlrev_b32 {v(HH4)}, 2, {v(T[0])}")
    emit(f"v_and_b32 {v(T[1])}, 31, {v(LANE)}")               # l31
    emit(f"v_and_b32 {v(T[2])}, 15, {v(LANE)}")               # l31 & 15 == lane & 15 (a16 / cpos)
    emit(f"v_lshlrev_b32 {v(T[3])}, 8, {v(T[1])}")            # l31 * 256
    emit(f"v_add_u32 {v(T[3])}, {s(S_LDS)}, {v(T[3])}")
    for ks in range(8):
        # ((2ks + hh) ^ (l31 & 15)) << 4
        emit(f"v_add_u32 {v(T[4])}, {2 * ks}, {v(T[0])}")
        emit(f"v_xor_b32 {v(T[4])}, {v(T[4])}, {v(T[2])}")
        emit(f"v_lshl_add_u32 {v(KADDR[ks])}, {v(T[4])}, 4, {v(T[3])}")
    # V^T: key0 = 4hh + (a16>>2); byte = key0*256 + (((db ^ kq) << 6) | (g1 << 5) | (a3 << 3))
    emit(f"v_lshrrev_b32 {v(T[4])}, 2, {v(T[2])}")            # kq = a16 >> 2
    emit(f"v_add_u32 {v(T[5])}, {v(HH4)}, {v(T[4])}")         # key0
    emit(f"v_lshlrev_b32 {v(T[5])}, 8, {v(T[5])}")
    emit(f"v_add_u32 {v(T[5])}, {s(S_LDS)}, {v(T[5])}")
    emit(f"v_add_u32 {v(T[5])}, {V_REGION}, {v(T[5])}")
    emit(f"v_lshrrev_b32 {v(T[6])}, 4, {v(LANE)}")            # g = lane >> 4
    emit(f"v_and_b32 {v(T[7])}, 1, {v(T[6])}")                # g & 1
    emit(f"v_lshlrev_b32 {v(T[7])}, 5, {v(T[7])}")
    emit(f"v_and_b32 {v(T[8])}, 3, {v(T[2])}")                # a3
    emit(f"v_lshl_or_b32 {v(T[7])}, {v(T[8])}, 3, {v(T[7])}")  # (a3<<3) | (g1<<5)
    emit(f"v_add_u32 {v(T[5])}, {v(T[5])}, {v(T[7])}")
    for db in range(4):
        emit(f"v_xor_b32 {v(T[7])}, {db}, {v(T[4])}")
        emit(f"v_lshl_add_u32 {v(VADDR[db])}, {v(T[7])}, 6, {v(T[5])}")
    # DMA: rip = lane>>4 (T6), cpos = lane&15 (T2); RIPROW = 16*wave + rip
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, 4")
    emit(f"v_add_u32 {v(RIPROW)}, {s(S_T0)}, {v(T[6])}")
    emit(f"v_xor_b32 {v(T[7])}, {v(T[2])}, {v(T[6])}")
    emit(f"v_lshlrev_b32 {v(T[7])}, 4, {v(T[7])}")            # T7 = (cpos ^ rip) << 4      (kept for the ragged path)
    emit(f"v_lshlrev_b32 {v(T[8])}, 2, {v(T[6])}")
    emit(f"v_xor_b32 {v(T[8])}, {v(T[2])}, {v(T[8])}")
    emit(f"v_lshlrev_b32 {v(T[8])}, 4, {v(T[8])}")            # T8 = (cpos ^ (rip<<2)) << 4 (kept for the ragged path)
    for j in range(4):
        # LK[j] = (16w + 4j + rip) * k_rs + (T7 ^ (j<<6)) ; LV[j] = (16w + 4j + rip) * v_rs + T8   (relative to the tile)
        emit(f"v_add_u32 {v(T[4])}, {4 * j}, {v(RIPROW)}")
        emit(f"v_mul_lo_u32 {v(LK[j])}, {v(T[4])}, {s(S_KRS)}")
        emit(f"v_xor_b32 {v(T[5])}, {j << 6}, {v(T[7])}")
        emit(f"v_add_u32 {v(LK[j])}, {v(LK[j])}, {v(T[5])}")
        emit(f"v_mul_lo_u32 {v(LV[j])}, {v(T[4])}, {s(S_VRS)}")
        emit(f"v_add_u32 {v(LV[j])}, {v(LV[j])}, {v(T[8])}")

    emit("; ---- Q fragments: row q_row0 + 32*wave + l31, d = 16*ks + 8*hh")
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, 5")
    emit(f"s_add_u32 {s(S_T0)}, {s(S_T0)}, {s(S_QROW0)}")
    emit(f"v_add_u32 {v(QROW)}, {s(S_T0)}, {v(T[1])}")
    emit(f"s_sub_u32 {s(S_T1)}, {s(S_SEQLENQ)}, 1")
    emit(f"v_min_i32 {v(T[3])}, {v(QROW)}, {s(S_T1)}")
    emit(f"v_mad_u64_u32 {vr(T[4], 2)}, {sr(S_T64)}, {v(T[3])}, {s(S_QRS)}, 0")
    emit(f"v_lshlrev_b32 {v(T[6])}, 4, {v(T[0])}")            # hh * 16 bytes
    emit(f"v_add_co_u32 {v(T[4])}, vcc, {v(T[4])}, {v(T[6])}")
    emit(f"v_addc_co_u32 {v(T[5])}, vcc, 0, {v(T[5])}, vcc")
    emit(f"v_add_co_u32 {v(T[4])}, vcc, {s(S_QBASE)}, {v(T[4])}")
    emit(f"v_mov_b32 {v(T[6])}, {s(S_QBASE + 1)}")
    emit(f"v_addc_co_u32 {v(T[5])}, vcc, {v(T[5])}, {v(T[6])}, vcc")
    for ks in range(8):
        emit(f"global_load_dwordx4 {vr(Q[ks], 4)}, {vr(T[4], 2)}, off offset:{32 * ks}")
    emit("s_waitcnt vmcnt(0)")
    emit(f"v_cmp_gt_i32 vcc, {s(S_SEQLENQ)}, {v(QROW)}")      # row valid
    for r in range(32):
        emit(f"v_cndmask_b32 {v(Q[0] + r)}, 0, {v(Q[0] + r)}, vcc")   # rows past seqlen_q are ZERO rows

    emit("; ---- state")
    for r in range(64):
        emit(f"v_mov_b32 {v(O[0] + r)}, 0")
    emit(f"v_mov_b32 {v(MRUN)}, 0xff800000")
    emit(f"v_mov_b32 {v(LRUN)}, 0")
    emit(f"v_mov_b32 {v(ALPHA)}, 0")
    emit(f"v_mov_b32 {v(ALPHA + 1)}, 0")

    emit("; ---- QK^T of the first walked tile (K buffer 0), seqlen-k mask, stats")
    lg = Lgkm()
    emit(f"v_mov_b32 {v(T[3])}, {s(S_SEQ)}")
    emit(f"ds_read_b32 {v(T[5])}, {v(T[3])}")
    lg.issue("n0")
    for j in range(4):
        k_read(lg, j, 0, j)
    for j in range(16):
        lg.wait_for(("k", j))
        kb, ks = j >> 3, j & 7
        mfma(SA[kb], KF[j % 4], Q[ks], c_init_zero=(ks == 0))
        if j + 4 < 16:
            k_read(lg, j % 4, 0, j + 4)
    lg.drain()
    emit(f"v_readfirstlane_b32 {s(S_NCUR)}, {v(T[5])}")
    emit("s_nop 15")
    # mask: only if n0 == k_tiles-1 and tail_valid < 64  (mask.h:44-78; first walked tile only, mainloop...:1626)
    nomask = new_label("nomask")
    emit(f"s_cmp_eq_u32 {s(S_NCUR)}, {s(S_KTM1)}")
    emit(f"s_cbranch_scc0 {nomask}")
    emit(f"s_cmp_lt_i32 {s(S_TAILVALID)}, 64")
    emit(f"s_cbranch_scc0 {nomask}")
    emit(f"v_mov_b32 {v(T[1])}, 0xff800000")
    for kb in range(2):
        for r in range(16):
            key = 32 * kb + (r & 3) + 8 * (r >> 2)
            emit(f"v_add_u32 {v(T[0])}, {key}, {v(HH4)}")
            emit(f"v_cmp_gt_i32 vcc, {s(S_TAILVALID)}, {v(T[0])}")            # key < tail_valid -> keep
            emit(f"v_cndmask_b32 {v(SA[kb] + r)}, {v(T[1])}, {v(SA[kb] + r)}, vcc")
    label(nomask)
    for op in row_max_ops(SA):
        emit(op)
    emit(f"s_mov_b32 {s(S_T2)}, 0")
    emit(f"s_mov_b32 {s(S_DOMASK)}, 0")
    stats_tail(S_T2, None)
    emit(f"s_or_b32 {s(S_DOMASK)}, {s(S_DOMASK)}, 1")        # position 0: forced "do"
    emit(f"v_mov_b32 {v(ALPHA)}, 0")                          # O = 0, l = 0
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")                                          # every wave has read K(0) before step 0 refills buffer 0


def epilogue():
    emit("; ---- flush the last vote word, export O^T / m / l through LDS")
    nofl = new_label("nolastflush")
    emit(f"s_and_b32 {s(S_T0)}, {s(S_NTILES)}, 31")
    emit(f"s_cmp_eq_u32 {s(S_T0)}, 0")
    emit(f"s_cbranch_scc1 {nofl}")
    emit(f"s_sub_u32 {s(S_T2)}, {s(S_NTILES)}, 1")
    flush_domask(S_T2)
    label(nofl)
    emit("s_nop 15")
    # O: register r of d-block db at lds_base + (4*db + r/4)*4096 + tid*16 ; tid = wave*64 + lane
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, 10")
    emit(f"v_lshl_add_u32 {v(T[0])}, {v(LANE)}, 4, {s(S_T0)}")
    emit(f"v_add_u32 {v(T[0])}, {s(S_LDS)}, {v(T[0])}")
    for db in range(4):
        for q4 in range(4):
            emit(f"ds_write_b128 {v(T[0])}, {vr(O[db] + 4 * q4, 4)} offset:{(4 * db + q4) * 4096}")
    # m, l: export + tid*8
    emit(f"s_lshl_b32 {s(S_T0)}, {s(S_WAVE)}, 9")
    emit(f"v_lshl_add_u32 {v(T[1])}, {v(LANE)}, 3, {s(S_T0)}")
    emit(f"v_add_u32 {v(T[1])}, {s(S_EXPORT)}, {v(T[1])}")
    emit(f"v_mov_b32 {v(T[2])}, {v(MRUN)}")
    emit(f"v_mov_b32 {v(T[3])}, {v(LRUN)}")
    emit(f"ds_write_b64 {v(T[1])}, {vr(T[2], 2)}")
    emit("s_waitcnt lgkmcnt(0)")


def main():
    prologue()
    loop, done, odd = new_label("loop"), new_label("done"), new_label("odd")
    label(loop)
    emit(f"s_cmp_lt_u32 {s(S_I)}, {s(S_NTILES)}")
    emit(f"s_cbranch_scc0 {done}")
    step(0)
    emit(f"s_cmp_lt_u32 {s(S_I)}, {s(S_NTILES)}")
    emit(f"s_cbranch_scc0 {done}")
    step(1)
    emit(f"s_branch {loop}")
    label(done)
    epilogue()
    text = "\n".join(out)
    path = sys.argv[1] if len(sys.argv) > 1 else "la_fwd_asm_body.inc"
    with open(path, "w") as f:
        f.write("// GENERATED by gen_fwd_asm.py — do not edit. Inline-asm body of la_fwd_bf16_d128_asm_kernel.\n")
        f.write('R"ASM(\n' + text + '\n)ASM"\n')
    n_mfma = text.count("v_mfma")
    print(f"wrote {path}: {len(out)} lines, {n_mfma} MFMAs")


if __name__ == "__main__":
    main()
