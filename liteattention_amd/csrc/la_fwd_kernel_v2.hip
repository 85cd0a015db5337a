// la_fwd_kernel_v2.hip — software-pipelined QK-Skip attention forward (gfx950, bf16, head_dim 128 and 64).
//
// Replaces the reference's Hopper kernel on the LiteAttention path:
//   FlashAttnFwdSm90::operator()            hopper/_internal/cpp/flash_fwd_kernel_sm90.h:213-571
//   CollectiveMainloopFwdSm90::load / mma   hopper/_internal/cpp/mainloop_fwd_sm90_tma_gmma_ws.hpp:808-1237, 1359-2101
//   SkipListReader / SkipListWriter         mainloop_fwd_sm90_tma_gmma_ws.hpp:47-192
//   Softmax::max_get_scale_detect_qk_skip   hopper/_internal/cpp/softmax.h:139-222
//   Softmax::online_softmax / finalize      softmax.h:263-296
//   Mask::apply<Seqlenk_mask>               hopper/_internal/cpp/mask.h:44-78
//   CollectiveEpilogueFwd::store            hopper/_internal/cpp/epilogue_fwd.hpp:214-403
//   SingleTileScheduler                     hopper/_internal/cpp/tile_scheduler.hpp:37-130
// All forward kernels of this library are one new design for CDNA4, not a translation (no TMA / WGMMA / warp specialisation):
//
//   * one workgroup = one (batch, head, q-tile of 128 rows); 4 waves x 32 query rows.
//   * S^T = K Q^T ("swapped" product) with v_mfma_f32_32x32x16_bf16: every lane then owns ONE
//     query row (column lane&31 of the 32x32 accumulator), so row max / row sum / the skip test
//     are in-lane plus a single half-wave exchange, and the O rescale factor is a per-lane scalar.
//   * P^T goes straight from the S^T accumulator registers (cvt to bf16) into the B operand of
//     O^T += V^T P^T: the contraction index is permuted consistently on the V side (the LDS
//     transpose-read addresses), so no cross-lane shuffle of P is needed.
//   * K and V tiles (64 keys) are double-buffered in LDS, 16-byte XOR swizzle for K
//     (ds_read_b128 conflict-free), 64-byte XOR swizzle for V (ds_read_b64_tr_b16 conflict-free).
//   * the read list is expanded once per workgroup into an LDS tile sequence, so the K/V prefetch
//     follows the data-dependent walk with no per-tile global list reads.
//   * per-tile skip votes are OR-ed into an LDS bit vector (one LDS atomic per wave per tile); the
//     write list is serialised once in the epilogue by one lane. No extra pass, no per-tile barrier.
//
// v2's schedule (the first, register-staged version measured MFMA busy 37 %, waves 32 % parked on s_waitcnt/barrier,
// 27 % issue-stalled: profiles/r01a):
//
//   * K/V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction). The DMA
//     image is lane-linear, so the XOR swizzles move to the per-lane SOURCE address. No staging VGPRs
//     (-32), no ds_write pass, no vmcnt wait in front of it.
//   * the freed registers hold a second S^T accumulator: QK^T of tile i+1 is issued BEFORE the softmax of
//     tile i, so a wave's own exp2/max/cvt VALU work runs in the shadow of its own MFMAs (MFMA and VALU
//     are separate pipes) instead of only in the shadow of the co-resident workgroup's. K therefore runs
//     one tile ahead of V in LDS (same 64 KiB: K ring and V ring of two tiles each, one barrier per tile).
//   * the O rescale is skipped — exactly, not approximately — when no row of the wave raised its running
//     max in this tile (alpha == 1.0f for all 64 lanes): multiplying by 1.0f is the identity in IEEE
//     arithmetic, so results are bit-identical to the always-rescale schedule.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "la_fwd_common.h"

namespace la {

namespace {

constexpr int BN = 64;                           // keys per tile

// LDS swizzles (in 16-byte chunks of a row of 2*D bytes), chosen per head_dim so that the fragment reads are
// bank-conflict free (HISTORY.md section 3):
//   K, read with ds_read_b128 by 16-lane groups of distinct rows:  D=128: chunk ^ (row & 15)   (16 chunks/row)
//                                                                  D=64 : chunk ^ ((row>>1) & 7) (8 chunks/row,
//                                                                         two rows per 256-byte bank row)
//   V, read with ds_read_b64_tr_b16 by 32-lane groups covering 4 keys x 64 bytes: the 64-byte segment index is
//      XOR-ed with  D=128: key & 3  (4 segments/row),  D=64: (key>>1) & 1  (2 segments/row).
//   head_dim 256 (512-byte rows = two bank rows each): the same XORs as D = 128 on the low bits of the chunk / segment index.
template <int D> __device__ __forceinline__ constexpr int k_swz(int row) { return D >= 128 ? (row & 15) : ((row >> 1) & 7); }
template <int D> __device__ __forceinline__ constexpr int v_swz(int row) { return D >= 128 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One 1-KiB LDS-DMA piece: every lane moves 16 bytes from its own global address to
// (wave-uniform LDS base) + lane*16.
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)lds_dst, 16, 0, 0);
}

}  // namespace

template <int D, bool SKIPABLE, bool F16>
__global__ void __launch_bounds__(256, D > 128 ? 1 : 2)      // head_dim 256: O 128 + Q 64 + S 64 registers -> one wave per SIMD
la_fwd_v2_kernel(const FwdParams p) {                   // F16 selects fp16 instead of bf16 elements
    typedef Elem16<F16> E;
    typedef typename E::x8 ex8;
    typedef typename E::x4 ex4;
    constexpr int BM = 128;
    constexpr int ROW_BYTES = D * 2;                 // 256 / 128
    constexpr int TILE_BYTES = BN * ROW_BYTES;       // 16 / 8 KiB
    constexpr int KS = D / 16;                       // k-steps of QK^T
    constexpr int DB = D / 32;                       // 32-wide d blocks of O^T
    constexpr int CPR = ROW_BYTES / 16;              // 16-byte chunks per row
    constexpr int RPP = 1024 / ROW_BYTES;            // rows per 1-KiB DMA piece
    constexpr int PPW = TILE_BYTES / 1024 / 4;       // DMA pieces per wave per tile (each wave stages 16 rows)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const k_lds = smem;                        // [2][TILE_BYTES]
    unsigned char* const v_lds = smem + 2 * TILE_BYTES;       // [2][TILE_BYTES]
    int* const meta = reinterpret_cast<int*>(smem + 4 * TILE_BYTES);
    int* const seq = meta + 4;
    unsigned* const doflags = reinterpret_cast<unsigned*>(seq + p.seq_cap);
    unsigned* const endflags = doflags + (p.k_tiles + 31) / 32;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int hh = lane >> 5;
    const int l31 = lane & 31;

    // Work distribution: static (one workgroup per item, XCD-aware map) or, with a ticket counter in the caller's workspace,
    // persistent workgroups that take the next (batch, head, q-tile) item until none is left — see la_fwd_kernel_x64.hip.
    const bool dynamic = p.work_counter != nullptr;
    const int total_work = p.batch * p.num_heads * p.q_tile_count;
    if (dynamic && tid == 0) meta[2] = 0;            // "own queue is empty": workgroup state of next_work_item()
  for (;;) {
    int vid;
    if (dynamic) {
        if (tid == 0) meta[1] = next_work_item(p, D > 128 ? 32 : 64, &meta[2]);     // = workgroups co-resident on an XCD
        __syncthreads();
        vid = meta[1];
        if (static_cast<unsigned>(vid) >= static_cast<unsigned>(total_work)) return;     // -1: no work left
    } else {
        vid = xcd_work_id();
    }
    int bh, m_block;
    if (dynamic) {
        bh = vid / p.q_tile_count;
        m_block = p.q_tile_begin + vid % p.q_tile_count;
    } else {
        work_item(p, vid, bh, m_block);
    }
    const int h = bh % p.num_heads;
    const int b = bh / p.num_heads;
    // Per-item view of the problem. Fixed length: the launch-wide sizes. Varlen (la_fwd_args.cu_seqlens_*; dense launches only,
    // so the SKIPABLE instantiation keeps the plain form): sequence b's own rows and lengths.
    int seqlen_q = p.seqlen_q, seqlen_k = p.seqlen_k, k_tiles = p.k_tiles;
    int64_t q_off = b * p.q_batch_stride, k_off = b * p.k_batch_stride, v_off = b * p.v_batch_stride, o_off = b * p.o_batch_stride;
    float* lse_row0 = p.lse + static_cast<int64_t>(bh) * p.seqlen_q;
    if constexpr (!SKIPABLE) {
        if (p.cu_seqlens_q != nullptr) {
            const SeqView sv = seq_view(p, b, h, BN);
            if (m_block * BM >= sv.seqlen_q) return;           // q-tiles past this sequence's end
            if (sv.k_tiles == 0) {                             // a sequence without keys: o = 0, lse = +inf
                store_empty_rows(p, sv, h, m_block * BM, BM, D, tid, 256);
                return;
            }
            seqlen_q = sv.seqlen_q; seqlen_k = sv.seqlen_k; k_tiles = sv.k_tiles;
            q_off = sv.q_off; k_off = sv.k_off; v_off = sv.v_off; o_off = sv.o_off; lse_row0 = sv.lse_row0;
        }
    }
    const int hk = h / p.h_ratio;                      // K/V head (GQA/MQA: h_ratio query heads share one)
    const int64_t list_off = (static_cast<int64_t>(bh) * p.q_tiles + m_block) * (p.k_tiles + 1);

    if (SKIPABLE) {
        // doflags + endflags. Wave-uniform trip count with a predicated body: a `for (i = tid; ...)` loop ends with EXEC = 0, and the
        // register allocator has been seen to place spill stores right there (la_fwd_kernel_x64.hip, "COMPILER HAZARD")
        for (int base = 0; base < 2 * ((k_tiles + 31) / 32); base += 256)
            if (base + tid < 2 * ((k_tiles + 31) / 32)) doflags[base + tid] = 0u;
        __syncthreads();
        if (wave == 0) {
            const int n = expand_read_list(p.read_list + list_off, seq, endflags, k_tiles, lane);
            if (lane == 0) meta[0] = n;
        }
    }

    // ---- Q fragments (B operand of S^T = K Q^T): query row l31, d = 16*ks + 8*hh + [0,8)
    const int q_row = m_block * BM + wave * 32 + l31;
    ex8 qf[KS];
    {
        const uint16_t* qp = p.q + q_off + static_cast<int64_t>(q_row) * p.q_row_stride +
                             h * p.q_head_stride + hh * 8;
        const bool ok = q_row < seqlen_q;   // rows past seqlen_q are ZERO rows (TMA OOB fill in the reference)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 t = {0u, 0u, 0u, 0u};
            if (ok) t = *reinterpret_cast<const u32x4*>(qp + ks * 16);
            qf[ks] = __builtin_bit_cast(ex8, t);
        }
    }

    // ---- LDS-DMA addressing. A tile is TILE_BYTES/1024 pieces of 1 KiB (RPP rows each); wave w stages rows
    // 16w .. 16w+15 = pieces PPW*w .. PPW*w+PPW-1. Lane: row rip = lane / CPR inside the piece, LDS chunk
    // position cpos = lane % CPR. The DMA image is lane-linear, so the swizzles go on the SOURCE address:
    // LDS position c' of row r holds data chunk c' ^ swz(r).
    const unsigned char* const kg = reinterpret_cast<const unsigned char*>(p.k + k_off + hk * p.k_head_stride);
    const unsigned char* const vg = reinterpret_cast<const unsigned char*>(p.v + v_off + hk * p.v_head_stride);
    const int k_rs = static_cast<int>(p.k_row_stride * 2), v_rs = static_cast<int>(p.v_row_stride * 2);   // bytes (< 2^31, checked on the host)
    const int rip = lane / CPR;
    const int cpos = lane % CPR;
    // K: swz(RPP*j + rip) = swz(rip) ^ 4j for both head dims (RPP*j only touches bits the swizzle maps to 4j),
    // so piece j XORs (4j << 4) into the byte offset; V: swz(RPP*j + rip) = swz(rip).
    const int k_lane = rip * k_rs + ((cpos ^ k_swz<D>(rip)) << 4);
    const int v_lane = rip * v_rs + ((cpos ^ v_swz<D>(rip)) << 4);
    const int last_row = seqlen_k - 1;
    auto dma_tile = [&](int n, int kbuf, bool do_k, int vbuf, bool do_v) {
        const int row_w = n * BN + 16 * wave;                               // wave-uniform first row of this wave's slice
        if (__builtin_expect(row_w + 15 <= last_row, 1)) {
            const unsigned char* kb_ = kg + static_cast<int64_t>(row_w) * k_rs;   // scalar bases
            const unsigned char* vb_ = vg + static_cast<int64_t>(row_w) * v_rs;
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                // swizzle of row RPP*j + rip from the lane's swz(rip): K: ^ (RPP*j mapped through k_swz) = 4j (D = 128: rows
                // 4j+rip; D = 64: rows 8j+rip, >>1) or 2j (D = 256: rows 2j+rip); V: unchanged, except D = 256 where rows
                // 2j+rip flip bit 1 of (row & 3) on odd j
                constexpr int KX = D > 128 ? 2 : 4;
                const int vx = D > 128 ? ((j & 1) << 7) : 0;
                if (do_k) dma16(kb_ + j * RPP * k_rs + (k_lane ^ ((KX * j) << 4)), k_lds + kbuf * TILE_BYTES + (PPW * wave + j) * 1024);
                if (do_v) dma16(vb_ + j * RPP * v_rs + (v_lane ^ vx), v_lds + vbuf * TILE_BYTES + (PPW * wave + j) * 1024);
            }
        } else {
            asm volatile("; ragged K/V tail" ::: "memory");                  // keep this a real (rare) branch
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                const int r = RPP * j + rip;
                const int grow = min(row_w + r, last_row);                    // rows past seqlen_k: clamp (masked, P = 0)
                if (do_k) dma16(kg + static_cast<int64_t>(grow) * k_rs + ((cpos ^ k_swz<D>(r)) << 4),
                                k_lds + kbuf * TILE_BYTES + (PPW * wave + j) * 1024);
                if (do_v) dma16(vg + static_cast<int64_t>(grow) * v_rs + ((cpos ^ v_swz<D>(r)) << 4),
                                v_lds + vbuf * TILE_BYTES + (PPW * wave + j) * 1024);
            }
        }
    };

    __syncthreads();   // seq / meta / doflags visible
    const int n_tiles = SKIPABLE ? meta[0] : k_tiles;
    auto tile_at = [&](int i) -> int {
        const int ii = min(i, n_tiles - 1);
        return SKIPABLE ? __builtin_amdgcn_readfirstlane(seq[ii]) : (k_tiles - 1 - ii);
    };

    // prologue: K(0) -> kbuf0, V(0) -> vbuf0, K(1) -> kbuf1
    dma_tile(tile_at(0), 0, true, 0, true);
    dma_tile(tile_at(1), 1, true, 0, false);
    __syncthreads();

    // ---- per-lane LDS read offsets
    // K A-operand: row = 32*kb + l31, chunk = 2*ks + hh   ->  row*ROW_BYTES + ((chunk ^ k_swz(row)) << 4)
    const int k_rd_row = l31 * ROW_BYTES;
    const int k_rd_sw = k_swz<D>(l31);
    // V^T A-operand via ds_read_b64_tr_b16. 16-lane group g = lane>>4, a = lane&15:
    //   key = 16*kk + 4*hh + (a>>2) (+8 for the second read), d = 32*db + 16*(g&1) + 4*(a&3)
    //   16-byte chunk = 4*db + 2*(g&1) + ((a&3)>>1), swizzled with v_swz(key) (constant per lane), +8 bytes if a&1
    const int a16 = lane & 15;
    const int v_key0 = 4 * hh + (a16 >> 2);
    int v_rd[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
        v_rd[db] = v_key0 * ROW_BYTES +
                   (((4 * db + 2 * ((lane >> 4) & 1) + ((a16 & 3) >> 1)) ^ v_swz<D>(v_key0)) << 4) + 8 * (a16 & 1);

    const float c = p.scale_log2;
    const float thr = p.thr;
    const int tail_valid = seqlen_k - (k_tiles - 1) * BN;   // valid keys in tile k_tiles-1 (1..64)
    unsigned domask = 1u;                                     // wave-uniform "do" bits of 32 consecutive positions; position 0 is never flagged
    float l_run = 0.f;
    f32x16 o_acc[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[db][r] = 0.f;

    // S^T[key][q] for one 64-key tile from K buffer `kbuf`
    auto qk_tile = [&](int kbuf, f32x16 (&s)[2]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
            const unsigned char* kt = k_lds + kbuf * TILE_BYTES + kb * 32 * ROW_BYTES + k_rd_row;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const ex8 kf = *reinterpret_cast<const ex8*>(kt + (((2 * ks + hh) ^ k_rd_sw) << 4));
                s[kb] = E::mfma(kf, qf[ks], s[kb]);
            }
        }
    };

    // Row max of a score tile + running-max update + skip vote for list position `pos`.
    // Returns alpha = exp2((m_prev - m_new) * c). `valid` = the tile really is part of the walk.
    float m_run = -INFINITY;
    auto stats = [&](f32x16 (&s)[2], int pos, bool valid) -> float {
        float m_loc = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) m_loc = fmaxf(m_loc, s[kb][r]);
        m_loc = half_swap_max(m_loc);
        if (!valid) m_loc = -INFINITY;                       // clamped duplicate past the end of the walk: no effect
        const float m_prev = m_run;
        m_run = fmaxf(m_prev, m_loc);
        if (SKIPABLE) {
            // do_qk |= ((m_loc - m_prev) * c) > thr   (softmax.h:194); wave vote -> one scalar bit per position
            const bool do_any = __any(((m_loc - m_prev) * c) > thr) && valid;
            domask |= (do_any ? 1u : 0u) << (pos & 31);
            if ((pos & 31) == 31 && valid) {
                if (lane == 0) atomicOr(&doflags[pos >> 5], domask);
                domask = 0u;
            }
        }
        return fast_exp2((m_prev - m_run) * c);              // first tile: exp2(-inf) = 0
    };
    // Seqlen-k mask (mask.h:44-78). As in the reference it is applied to the FIRST walked tile only
    // (mainloop...:1626): every well-formed list starts with tile k_tiles-1 — init_skip_list, the writer
    // (position 0 is never dropped) and must_skip_row all guarantee it — and only that tile can be ragged.
    auto mask_tail = [&](f32x16 (&s)[2], int n) {
        if (__builtin_expect(n == k_tiles - 1 && tail_valid < BN, 0)) {
            asm volatile("; seqlen-k mask" ::: "memory");    // a real, rare branch (not 32 selects per tile)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= tail_valid) s[kb][r] = -INFINITY;
                }
        }
    };

    // One pipeline step. On entry: s_cur = raw (masked) scores of tile i, m_run already includes tile i,
    // `alpha` is tile i's rescale factor and O has already been multiplied by it; V(i) is in vbuf[i&1],
    // K(i+1) in kbuf[(i+1)&1].
    float alpha = 0.f;
    auto step = [&](int i, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2]) {
        const int cur = i & 1;
        const bool has_next = (i + 1) < n_tiles;
        const int n_next = tile_at(i + 1);
        // stage K(i+2) -> kbuf[cur] (K(i) was consumed by the previous step), V(i+1) -> vbuf[cur^1]
        dma_tile(tile_at(i + 2), cur, true, 0, false);
        dma_tile(n_next, 0, false, cur ^ 1, true);

        // ---- phase 1: QK^T of tile i+1 (MFMA)  ||  P = exp2(S*c - m*c), row sum, bf16 P of tile i (VALU)
        qk_tile(cur ^ 1, s_nxt);
        const float m_scaled = m_run * c;
        float psum = 0.f;
        ex8 pf[4];   // B operand of O^T += V^T P^T: k-step kk = accumulator regs 8*(kk&1).. of block kk>>1
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fast_exp2(__builtin_fmaf(s_cur[kb][r], c, -m_scaled));
                s_cur[kb][r] = pv;
                psum += pv;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x8 t;
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = s_cur[kb][8 * half + e];
                pf[2 * kb + half] = __builtin_convertvector(t, ex8);
            }
        }
        l_run = l_run * alpha + psum;

        // ---- phase 2: O^T += V^T P^T of tile i (MFMA)  ||  row max / running max / vote of tile i+1 (VALU)
        const unsigned char* vt = v_lds + cur * TILE_BYTES;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    LDS_PTR(s16x4, vt + v_rd[db] + kk * 16 * ROW_BYTES));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    LDS_PTR(s16x4, vt + v_rd[db] + kk * 16 * ROW_BYTES + 8 * ROW_BYTES));
                const s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o_acc[db] = E::mfma(__builtin_bit_cast(ex8, vf), pf[kk], o_acc[db]);
            }
        }
        alpha = stats(s_nxt, i + 1, has_next);

        // ---- O^T *= alpha(i+1), after PV(i). Identity (bit-exact) when alpha == 1 on every lane.
        if (!__all(alpha == 1.0f)) {
            float a_in = alpha;
            asm volatile("; rescale O" : "+v"(a_in));           // value defined inside the branch: the multiplies cannot be speculated
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[db][r] *= a_in;
        }
        __syncthreads();   // all waves done with K(i+1)/V(i) of this step; this step's DMA has landed
    };

    f32x16 s_a[2], s_b[2];
    {
        const int n0 = tile_at(0);
        qk_tile(0, s_a);
        mask_tail(s_a, n0);
        domask = 0u;
        (void)stats(s_a, 0, true);
        domask |= 1u;                                        // the first walked tile is never flagged (softmax.h:153)
        alpha = 0.f;                                         // O = 0, l = 0
    }
    __syncthreads();   // every wave has read K(0) before step 0 re-fills kbuf0 with K(2)
    int i = 0;
    for (; i + 1 < n_tiles; i += 2) {
        step(i, s_a, s_b);
        step(i + 1, s_b, s_a);
    }
    if (i < n_tiles) step(i, s_a, s_b);
    if (SKIPABLE) {
        if ((n_tiles & 31) != 0 && lane == 0) atomicOr(&doflags[(n_tiles - 1) >> 5], domask);
        __syncthreads();
    }

    // ---- finalize (softmax.h:275-296) and store (epilogue_fwd.hpp:214-403)
    const float l_tot = half_swap_sum(l_run);
    const bool bad = (l_tot == 0.f) || (l_tot != l_tot);
    const float inv = bad ? 0.f : 1.f / l_tot;
    if (q_row < seqlen_q) {
        uint16_t* op = p.o + o_off + static_cast<int64_t>(q_row) * p.o_row_stride + h * p.o_head_stride;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 x = {o_acc[db][4 * t] * inv, o_acc[db][4 * t + 1] * inv, o_acc[db][4 * t + 2] * inv,
                           o_acc[db][4 * t + 3] * inv};
                *reinterpret_cast<ex4*>(op + 32 * db + 8 * t + 4 * hh) = __builtin_convertvector(x, ex4);
            }
        }
        if (p.lse != nullptr && hh == 0) {
            lse_row0[q_row] =
                bad ? -INFINITY : m_run * (c * 0.69314718055994530942f) + __logf(l_tot);
        }
    }

    if (SKIPABLE) {
        if (wave == 0 && p.write_list != nullptr) {
            const int* md = p.must_do_list ? (p.must_do_is_1d ? p.must_do_list : p.must_do_list + list_off) : nullptr;
#ifndef LA_ABL_NOWRITER
            write_skip_list_wave(seq, endflags, doflags, n_tiles, p.write_list + list_off, md, k_tiles, lane);
#endif
        }
    }
    if (!dynamic) return;
    __syncthreads();   // every wave is done with this item's LDS before the next one is set up
  }
}

size_t fwd_lds_bytes_v2(int head_dim, int k_tiles, int* seq_cap_out) {
    const int seq_cap = (k_tiles + 3) & ~3;
    if (seq_cap_out) *seq_cap_out = seq_cap;
    return 4 * static_cast<size_t>(BN) * head_dim * 2 + 16 + static_cast<size_t>(seq_cap) * 4 +
           2 * static_cast<size_t>((k_tiles + 31) / 32) * 4 + 16;
}

template <int D, bool SKIPABLE, bool F16>
static hipError_t launch_v2(const FwdParams& p, hipStream_t stream) {
    const int total = p.batch * p.num_heads * p.q_tile_count;
    FwdParams pp = p;
    const size_t lds = fwd_lds_bytes_v2(D, p.k_tiles, &pp.seq_cap);
    (void)hipGetLastError();   // drop any stale sticky error of this thread: only OUR launch is reported
    auto kfn = la_fwd_v2_kernel<D, SKIPABLE, F16>;
    const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (err != hipSuccess) return err;
    int grid = total;
    const hipError_t qerr = prepare_work_queue(pp, SKIPABLE, total, D > 128 ? 1 : 2, stream, &grid);     // workgroups per CU
    if (qerr != hipSuccess) return qerr;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, stream, pp);
    return hipGetLastError();
}

template <bool F16>
static hipError_t launch_v2_dtype(const FwdParams& p, int head_dim, bool skipable, hipStream_t stream) {
    if (head_dim == 128) return skipable ? launch_v2<128, true, F16>(p, stream) : launch_v2<128, false, F16>(p, stream);
    if (head_dim == 64) return skipable ? launch_v2<64, true, F16>(p, stream) : launch_v2<64, false, F16>(p, stream);
    if (head_dim == 256) return skipable ? launch_v2<256, true, F16>(p, stream) : launch_v2<256, false, F16>(p, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_fwd_bf16_v2(const FwdParams& p, int head_dim, bool skipable, bool f16, hipStream_t stream) {
    return f16 ? launch_v2_dtype<true>(p, head_dim, skipable, stream) : launch_v2_dtype<false>(p, head_dim, skipable, stream);
}

}  // namespace la
