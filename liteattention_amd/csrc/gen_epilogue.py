"""Shared by gen_fwd_x64.py and gen_fwd_x64_fp8.py: the in-register epilogue of the 64-rows-per-wave kernels.

finalize (softmax.h:275-296) + store (epilogue_fwd.hpp:214-403) straight from the accumulators: every lane owns ONE query row
(column lane & 31 of the 32x32 accumulator; lanes l and l ^ 32 split the head dim), so 1/l is lane-local after one half-wave
exchange and the accumulator registers (qb, db, 4t..4t+3) are 4 consecutive d = 32 db + 8 t + 4 hh + [0, 4) of that row: 8 bytes
of bf16; pairs of such groups are exchanged between the half-waves (v_permlane32_swap) into 16 contiguous bytes per lane and
stored with global_store_dwordx4 (16 store instructions per wave instead of 32).
Round 1 exported O^T as fp32 through a 128 KiB LDS overlay of the K/V rings and finished in C++: 32 ds_write_b128 + 32
ds_read_b128 per wave and a workgroup barrier per item for no transposition at all, and the overlay pinned 128 KiB of LDS.

Parameter words read here (LDS parameter block, 16-byte aligned groups; written by the C++ shell):
    [24] [25]  O row 0 of this (batch, head): byte address        [26]  O row stride in bytes      [27]  c * ln 2
    [28] [29]  &lse[row 0 of this (batch, head)] or 0             [30]  O scale (bf16: 1; fp8: v_descale)
    [31]       added to the LSE: ln of the factor P carried (bf16: 0; fp8: ln 2^-(8 - tau))
"""


def store_epilogue(g, o_reg):
    """g: the generator's globals (emit, v, s, sr, vr, label, new_label and its register map); o_reg(qb, db) -> first AGPR."""
    emit, v, s, sr, vr, label, new_label = g["emit"], g["v"], g["s"], g["sr"], g["vr"], g["label"], g["new_label"]
    T, L0, L1, MREF, NEGINF, HH4, QROW = g["T"], g["L0"], g["L1"], g["MREF"], g["NEGINF"], g["HH4"], g["QROW"]
    S_PARAM, S_SEQLENQ, S_EXEC, S_T64 = g["S_PARAM"], g["S_SEQLENQ"], g["S_EXEC"], g["S_T64"]
    # the loop's DMA-base registers are dead here: O base, LSE base and the scalars live in them
    S_OBASE, S_LSEB, S_ORS, S_CLN2, S_OSCALE, S_LSEADD = g["S_TB"], g["S_VB"], g["S_T0"], g["S_T1"], g["S_T2"], g["S_T3"]
    cvt = g.get("CVT_OP", "v_cvt_pk_bf16_f32")                 # the 16-bit output type follows the inputs (fp8 inputs: bf16)
    n_qb, n_db = g.get("NQB", 2), g.get("DB", 4)               # q-blocks per wave, 32-wide d-blocks (head_dim 256: 1 and 8)
    HH16 = g["MLOC"][0]                                        # dead after the loop: hh * 16 bytes
    emit("; ---- finalize + store O (bf16) and LSE straight from the accumulators")
    emit(f"v_mov_b32 {v(T[0])}, {s(S_PARAM)}")
    emit(f"ds_read_b128 {vr(T[4], 4)}, {v(T[0])} offset:96")
    emit(f"ds_read_b128 {vr(T[8], 4)}, {v(T[0])} offset:112")
    emit("s_waitcnt lgkmcnt(0)")
    for dst, src in ((S_OBASE, T[4]), (S_OBASE + 1, T[5]), (S_ORS, T[6]), (S_CLN2, T[7]), (S_LSEB, T[8]), (S_LSEB + 1, T[9]),
                     (S_OSCALE, T[10]), (S_LSEADD, T[11])):
        emit(f"v_readfirstlane_b32 {s(dst)}, {v(src)}")
    emit("s_nop 4")
    emit(f"v_lshlrev_b32 {v(HH16)}, 2, {v(HH4)}")
    for qb in range(n_qb):
        # l = sum over the two half-waves; inv = oscale / l (0 for l == 0 or NaN); lse = m_ref c ln2 + ln l + lse_add
        if g.get("LSUM_AGPR"):
            # row sums accumulated by the matrix pipe (fp8: ones x P~^T): every lane already holds the complete sum of its row
            emit(f"v_accvgpr_read_b32 {v(T[0])}, a{g['LSUM_AGPR'][qb]}")
            emit("s_nop 1")
        else:
            emit(f"v_add_f32 {v(T[0])}, {v(L0[qb])}, {v(L1[qb])}")
            emit(f"v_mov_b32 {v(T[1])}, {v(T[0])}")
            emit("s_nop 1")
            emit(f"v_permlane32_swap_b32 {v(T[0])}, {v(T[1])}")
            emit("s_nop 1")
            emit(f"v_add_f32 {v(T[0])}, {v(T[0])}, {v(T[1])}")
        emit(f"v_rcp_f32 {v(T[2])}, {v(T[0])}")
        emit(f"v_log_f32 {v(T[3])}, {v(T[0])}")
        emit("s_nop 0")
        emit(f"v_fma_f32 {v(T[1])}, -{v(T[0])}, {v(T[2])}, 1.0")        # one Newton step: 1/l to < 1 ulp
        emit(f"v_fma_f32 {v(T[2])}, {v(T[1])}, {v(T[2])}, {v(T[2])}")
        emit(f"v_mul_f32 {v(T[2])}, {s(S_OSCALE)}, {v(T[2])}")
        emit(f"v_mul_f32 {v(T[3])}, 0x3f317218, {v(T[3])}")              # ln 2
        emit(f"v_fma_f32 {v(T[3])}, {v(MREF[qb])}, {s(S_CLN2)}, {v(T[3])}")
        emit(f"v_add_f32 {v(T[3])}, {s(S_LSEADD)}, {v(T[3])}")
        emit(f"v_cmp_lg_f32 vcc, 0, {v(T[0])}")                           # false for l == 0 and for NaN
        emit(f"v_cndmask_b32 {v(T[2])}, 0, {v(T[2])}, vcc")
        emit(f"v_cndmask_b32 {v(T[3])}, {v(NEGINF)}, {v(T[3])}, vcc")
        # rows past seqlen_q are not stored
        emit(f"v_cmp_gt_i32 vcc, {s(S_SEQLENQ)}, {v(QROW[qb])}")
        emit(f"s_and_saveexec_b64 {sr(S_EXEC)}, vcc")
        emit(f"v_mad_u64_u32 {vr(T[4], 2)}, {sr(S_T64)}, {v(QROW[qb])}, {s(S_ORS)}, 0")
        emit(f"v_add_co_u32 {v(T[4])}, vcc, {v(T[4])}, {v(HH16)}")
        emit(f"v_addc_co_u32 {v(T[5])}, vcc, 0, {v(T[5])}, vcc")
        emit(f"v_add_co_u32 {v(T[4])}, vcc, {s(S_OBASE)}, {v(T[4])}")
        emit(f"v_mov_b32 {v(T[6])}, {s(S_OBASE + 1)}")
        emit(f"v_addc_co_u32 {v(T[5])}, vcc, {v(T[5])}, {v(T[6])}, vcc")
        # 16 groups of 4 floats per q-block -> 8 PAIRS (t even, t + 1) of one d-block: after the bf16 pack, two half-wave swaps
        # give the lower lanes cols 8t..8t+7 and the upper lanes cols 8t+8..8t+15 of their row = ONE 16-byte store per pair (upper
        # lanes +16 bytes) instead of two 8-byte ones: the store tail is issue-bound per instruction (cdna_hip_programming.md T21).
        # Software-pipelined over two register sets (the S buffers v0..v15 are dead here): read(p+1) sits between pack(p) and
        # swap(p), which also covers the 2 wait states a VALU write needs before v_permlane32_swap reads it.
        pairs = [(db, t) for db in range(n_db) for t in (0, 2)]

        def stage_read(i):
            db, t = pairs[i]
            r = 8 * (i % 2)
            return [f"v_accvgpr_read_b32 {v(r + k)}, a{o_reg(qb, db) + 4 * t + k}" for k in range(8)]

        def stage_pack(i):
            r = 8 * (i % 2)
            return [f"v_mul_f32 {v(r + k)}, {v(r + k)}, {v(T[2])}" for k in range(8)] + \
                   [f"{cvt} {v(r + k)}, {v(r + 2 * k)}, {v(r + 2 * k + 1)}" for k in range(4)]

        def stage_store(i):
            db, t = pairs[i]
            r = 8 * (i % 2)
            return [f"v_permlane32_swap_b32 {v(r)}, {v(r + 2)}", f"v_permlane32_swap_b32 {v(r + 1)}, {v(r + 3)}",
                    f"global_store_dwordx4 {vr(T[4], 2)}, {vr(r, 4)}, off offset:{64 * db + 16 * t}"]

        seq = stage_read(0) + stage_pack(0)
        for i in range(len(pairs)):
            if i + 1 < len(pairs):
                seq += stage_read(i + 1)
            else:
                seq += ["s_nop 1"]
            seq += stage_store(i)
            if i + 1 < len(pairs):
                seq += stage_pack(i + 1)
        for op in seq:
            emit(op)
        # LSE: one lane per row (hh == 0), only if the caller wants it
        nolse = new_label("nolse")
        emit(f"s_cmp_eq_u64 {sr(S_LSEB)}, 0")
        emit(f"s_cbranch_scc1 {nolse}")
        emit(f"v_cmp_eq_u32 vcc, 0, {v(HH4)}")
        emit("s_and_b64 exec, exec, vcc")
        emit(f"v_lshlrev_b32 {v(T[6])}, 2, {v(QROW[qb])}")
        emit(f"v_add_co_u32 {v(T[6])}, vcc, {s(S_LSEB)}, {v(T[6])}")
        emit(f"v_mov_b32 {v(T[7])}, {s(S_LSEB + 1)}")
        emit(f"v_addc_co_u32 {v(T[7])}, vcc, 0, {v(T[7])}, vcc")
        emit(f"global_store_dword {vr(T[6], 2)}, {v(T[3])}, off")
        label(nolse)
        emit(f"s_mov_b64 exec, {sr(S_EXEC)}")
