// la_fwd_kernel.hip — QK-Skip attention forward for gfx950 (MI355X, CDNA4), bf16, head_dim 128.
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
// It is a new design for CDNA4, not a translation (no TMA / WGMMA / warp specialisation):
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
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "la_fwd_common.h"

namespace la {

// ------------------------------------------------------------------------------------------------
// Forward kernel. NW waves of 64 lanes; each wave owns 32 query rows. BN = 64 keys per tile.
// ------------------------------------------------------------------------------------------------
template <int NW, bool SKIPABLE>
__global__ void __launch_bounds__(NW * 64, (NW == 4 ? 2 : 1))
la_fwd_bf16_d128_kernel(const FwdParams p) {
    constexpr int D = 128;
    constexpr int BM = NW * 32;
    constexpr int BN = 64;
    constexpr int NT = NW * 64;
    constexpr int ROW_BYTES = D * 2;                 // 256
    constexpr int TILE_BYTES = BN * ROW_BYTES;       // 16 KiB
    constexpr int CH = (BN * 16) / NT;               // 16-byte chunks per thread per tile (4 @ NW=4)
    constexpr int ROWS_PER_PASS = NT / 16;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const k_lds = smem;                        // [2][TILE_BYTES]
    unsigned char* const v_lds = smem + 2 * TILE_BYTES;       // [2][TILE_BYTES]
    int* const meta = reinterpret_cast<int*>(smem + 4 * TILE_BYTES);  // [4]: n_tiles
    int* const seq = meta + 4;                                // [k_tiles] tile index per position
    unsigned* const doflags = reinterpret_cast<unsigned*>(seq + p.seq_cap);  // [(k_tiles+31)/32]
    unsigned* const endflags = doflags + (p.k_tiles + 31) / 32;               // [(k_tiles+31)/32]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int hh = lane >> 5;      // half-wave: which 4-key group of every 8 / which 8-d group of every 16
    const int l31 = lane & 31;

    // ---- XCD-aware (bijective) block -> (b, h, q-tile) map: blocks b%8 share an XCD/L2, so give
    // each XCD a contiguous run of q-tiles of the same head (they stream the same K/V).
    int vid;
    {
        const int bid = blockIdx.x, nwg = gridDim.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int q8 = nwg >> 3, r8 = nwg & 7;
        vid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int m_block = vid % p.q_tiles;
    const int bh = vid / p.q_tiles;
    const int h = bh % p.num_heads;
    const int b = bh / p.num_heads;

    const int k_tiles = p.k_tiles;
    const int64_t list_off = (static_cast<int64_t>(bh) * p.q_tiles + m_block) * (k_tiles + 1);

    // ---- expand the read list into the LDS tile sequence (wave 0), clear the vote bits
    if (SKIPABLE) {
        for (int i = tid; i < 2 * ((k_tiles + 31) / 32); i += NT) doflags[i] = 0u;   // doflags + endflags
        __syncthreads();
        if (wave == 0) {
            const int n = expand_read_list(p.read_list + list_off, seq, endflags, k_tiles, lane);
            if (lane == 0) meta[0] = n;
        }
    }

    // ---- Q fragments: B operand of S^T = K Q^T. lane: query row l31, d = 16*ks + 8*hh + [0,8)
    const int q_row = m_block * BM + wave * 32 + l31;
    bf16x8 qf[8];
    {
        const uint16_t* qp = p.q + b * p.q_batch_stride + static_cast<int64_t>(q_row) * p.q_row_stride +
                             h * p.q_head_stride + hh * 8;
        const bool ok = q_row < p.seqlen_q;   // rows past seqlen_q are ZERO rows (TMA OOB fill in the reference)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            u32x4 t = {0u, 0u, 0u, 0u};
            if (ok) t = *reinterpret_cast<const u32x4*>(qp + ks * 16);
            qf[ks] = __builtin_bit_cast(bf16x8, t);
        }
    }

    // ---- K/V tile staging (global -> registers -> swizzled LDS)
    const uint16_t* kbase = p.k + b * p.k_batch_stride + h * p.k_head_stride + (tid & 15) * 8;
    const uint16_t* vbase = p.v + b * p.v_batch_stride + h * p.v_head_stride + (tid & 15) * 8;
    const int ld_row0 = tid >> 4;                               // + ROWS_PER_PASS * j
    // K: 16-B chunk c of row r lives at r*256 + ((c ^ (r & 15)) << 4)
    // V: 64-B segment swizzle: r*256 + ((c ^ ((r & 3) << 2)) << 4)
    u32x4 kreg[CH], vreg[CH];
    auto issue_loads = [&](int n) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int key = n * BN + ld_row0 + ROWS_PER_PASS * j;
            u32x4 kz = {0u, 0u, 0u, 0u}, vz = {0u, 0u, 0u, 0u};
            if (key < p.seqlen_k) {
                kz = *reinterpret_cast<const u32x4*>(kbase + static_cast<int64_t>(key) * p.k_row_stride);
                vz = *reinterpret_cast<const u32x4*>(vbase + static_cast<int64_t>(key) * p.v_row_stride);
            }
            kreg[j] = kz;
            vreg[j] = vz;
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int r = ld_row0 + ROWS_PER_PASS * j;
            const int c = tid & 15;
            *reinterpret_cast<u32x4*>(k_lds + buf * TILE_BYTES + r * ROW_BYTES + ((c ^ (r & 15)) << 4)) = kreg[j];
            *reinterpret_cast<u32x4*>(v_lds + buf * TILE_BYTES + r * ROW_BYTES + ((c ^ ((r & 3) << 2)) << 4)) = vreg[j];
        }
    };

    __syncthreads();   // seq / meta / doflags visible
    const int n_tiles = SKIPABLE ? meta[0] : k_tiles;
    auto tile_at = [&](int i) -> int { return SKIPABLE ? seq[i] : (k_tiles - 1 - i); };

    issue_loads(tile_at(0));
    store_lds(0);
    __syncthreads();

    // ---- per-lane LDS read offsets
    // K A-operand: row = 32*kb + l31, chunk = 2*ks + hh  ->  row*256 + ((chunk ^ (row&15)) << 4)
    const int k_rd_row = l31 * ROW_BYTES;              // + kb*32*256
    const int k_rd_sw = l31 & 15;
    // V^T A-operand via ds_read_b64_tr_b16. 16-lane group g = lane>>4, a = lane&15:
    //   key = 16*kk + 4*hh + (a>>2) (+8 for the second read), d = 32*db + 16*(g&1) + 4*(a&3)
    //   byte = key*256 + (((db ^ (a>>2)) << 6) | ((g&1) << 5) | ((a&3) << 3))
    const int a16 = lane & 15;
    const int v_key0 = 4 * hh + (a16 >> 2);
    int v_rd[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
        v_rd[db] = v_key0 * ROW_BYTES + (((db ^ (a16 >> 2)) << 6) | (((lane >> 4) & 1) << 5) | ((a16 & 3) << 3));

    const float c = p.scale_log2;
    const float thr = p.thr;
    float m_run = -INFINITY;   // running row max (raw scores)
    float l_run = 0.f;         // lane-partial row sum
    f32x16 o_acc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[db][r] = 0.f;

    for (int i = 0; i < n_tiles; ++i) {
        const int cur = i & 1;
        const int n = tile_at(i);
        const bool has_next = (i + 1) < n_tiles;
        if (has_next) issue_loads(tile_at(i + 1));

        // ---- S^T[key][q] = sum_d K[key][d] Q[q][d]
        f32x16 s_acc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.f;
            const unsigned char* kt = k_lds + cur * TILE_BYTES + kb * 32 * ROW_BYTES + k_rd_row;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt + (((2 * ks + hh) ^ k_rd_sw) << 4));
                s_acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_acc[kb], 0, 0, 0);
            }
        }
        // accumulator layout: key = 32*kb + (r&3) + 8*(r>>2) + 4*hh, query = l31

        // ---- seqlen-k mask (mask.h:44-78): only tile k_tiles-1 can hold keys >= seqlen_k
        if (n == k_tiles - 1) {
            const int valid = p.seqlen_k - n * BN;   // 1..64
            if (valid < BN) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (key >= valid) s_acc[kb][r] = -INFINITY;
                    }
            }
        }

        // ---- row max, skip vote (softmax.h:139-222), online softmax (softmax.h:81-121, 263-273)
        float m_loc = s_acc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) m_loc = fmaxf(m_loc, s_acc[kb][r]);
        m_loc = half_swap_max(m_loc);
        const float m_prev = m_run;
        m_run = fmaxf(m_prev, m_loc);
        const float alpha = fast_exp2((m_prev - m_run) * c);   // first tile: exp2(-inf) = 0
        if (SKIPABLE) {
            // do_qk |= ((m_loc - m_prev) * c) > thr ; the first processed tile is never flagged
            const bool do_qk = (((m_loc - m_prev) * c) > thr) || (i == 0);
            if (__any(do_qk) && lane == 0) atomicOr(&doflags[i >> 5], 1u << (i & 31));
        }
        const float m_scaled = m_run * c;
        float psum = 0.f;
        bf16x8 pf[4];   // B operand of O^T += V^T P^T: k-step kk uses accumulator regs 8*(kk&1) .. +7 of block kk>>1
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fast_exp2(__builtin_fmaf(s_acc[kb][r], c, -m_scaled));
                s_acc[kb][r] = pv;
                psum += pv;
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x8 t;
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = s_acc[kb][8 * half + e];
                pf[2 * kb + half] = __builtin_convertvector(t, bf16x8);
            }
        }
        l_run = l_run * alpha + psum;

        // ---- O^T = O^T * alpha + V^T P^T
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[db][r] *= alpha;
        }
        const unsigned char* vt = v_lds + cur * TILE_BYTES;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    LDS_PTR(s16x4, vt + v_rd[db] + kk * 16 * ROW_BYTES));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    LDS_PTR(s16x4, vt + v_rd[db] + kk * 16 * ROW_BYTES + 8 * ROW_BYTES));
                const s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf[kk],
                                                                    o_acc[db], 0, 0, 0);
            }
        }

        // ---- stage the prefetched tile; one barrier per tile
        if (has_next) store_lds(cur ^ 1);
        __syncthreads();
    }

    // ---- finalize (softmax.h:275-296) and store (epilogue_fwd.hpp:214-403)
    const float l_tot = half_swap_sum(l_run);
    const bool bad = (l_tot == 0.f) || (l_tot != l_tot);
    const float inv = bad ? 0.f : 1.f / l_tot;
    if (q_row < p.seqlen_q) {
        uint16_t* op = p.o + b * p.o_batch_stride + static_cast<int64_t>(q_row) * p.o_row_stride + h * p.o_head_stride;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // regs 4t..4t+3 -> d = 32*db + 8*t + 4*hh + [0,4)
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                f32x4 x = {o_acc[db][4 * t] * inv, o_acc[db][4 * t + 1] * inv, o_acc[db][4 * t + 2] * inv,
                           o_acc[db][4 * t + 3] * inv};
                *reinterpret_cast<bf16x4*>(op + 32 * db + 8 * t + 4 * hh) = __builtin_convertvector(x, bf16x4);
            }
        }
        if (p.lse != nullptr && hh == 0) {
            // LSE = row_max * softmax_scale + ln(l)   (softmax.h:293; c*ln2 = softmax_scale)
            p.lse[static_cast<int64_t>(bh) * p.seqlen_q + q_row] =
                bad ? -INFINITY : m_run * (c * 0.69314718055994530942f) + __logf(l_tot);
        }
    }

    // ---- skip-list epilogue: AND of the waves' votes is already in doflags (OR of "do" bits)
    if (SKIPABLE) {
        if (tid == 0 && p.write_list != nullptr) {
            const int* md = p.must_do_list ? (p.must_do_is_1d ? p.must_do_list : p.must_do_list + list_off) : nullptr;
            write_skip_list(seq, endflags, doflags, n_tiles, p.write_list + list_off, md, k_tiles);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Host-side launchers (called from la_api.hip)
// ------------------------------------------------------------------------------------------------
size_t fwd_lds_bytes(int k_tiles, int* seq_cap_out) {
    const int seq_cap = (k_tiles + 3) & ~3;
    if (seq_cap_out) *seq_cap_out = seq_cap;
    return 4 * 16384 + 16 + static_cast<size_t>(seq_cap) * 4 + 2 * static_cast<size_t>((k_tiles + 31) / 32) * 4 + 16;
}

hipError_t launch_fwd_bf16_d128(const FwdParams& p, bool skipable, hipStream_t stream) {
    const int total = p.batch * p.num_heads * p.q_tiles;
    FwdParams pp = p;
    const size_t lds = fwd_lds_bytes(p.k_tiles, &pp.seq_cap);
    hipError_t err;
    (void)hipGetLastError();   // drop any stale sticky error of this thread: only OUR launch is reported
    if (skipable) {
        auto kfn = la_fwd_bf16_d128_kernel<4, true>;
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds));
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(kfn, dim3(total), dim3(256), lds, stream, pp);
    } else {
        auto kfn = la_fwd_bf16_d128_kernel<4, false>;
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds));
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(kfn, dim3(total), dim3(256), lds, stream, pp);
    }
    return hipGetLastError();
}

}  // namespace la
