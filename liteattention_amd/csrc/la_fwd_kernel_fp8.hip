// la_fwd_kernel_fp8.hip — QK-Skip attention forward with fp8 (OCP e4m3fn) Q/K/V, bf16 output, head_dim 128.
//
// BASELINE.json configs[4]. Replaces the reference's fp8 branch of the same Hopper kernel:
//   tile (128,224)                          hopper/_internal/cpp/tile_size.h:54-55      -> (128, 64) here
//   softmax_scale_log2 *= q_descale*k_descale   hopper/_internal/cpp/flash_fwd_kernel_sm90.h:505-512
//   P scaled by 2^8 before the e4m3 cast (Max_offset = 8)   flash_fwd_kernel_sm90.h:514, softmax.h:85-87
//   row sum rescaled by 2^-8 only inside the LSE            softmax.h:288-292
//   O *= v_descale / l                                      mainloop_fwd_sm90_tma_gmma_ws.hpp:1852-1853
//   V transposed in shared memory for the PV operand        mainloop_fwd_sm90_tma_gmma_ws.hpp:942-984
//
// Same schedule as la_fwd_kernel_v2.hip (LDS-DMA staging, QK^T(i+1) under softmax(i), exact rescale skip, skip
// votes and write list fused), with the fp8 specifics designed for CDNA4:
//   * both GEMMs use the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 with every E8M0 scale = 127 (2^0): numerically
//     a plain e4m3 x e4m3 -> fp32 MFMA, but the ONLY fp8 form that runs at twice the bf16 rate on gfx950 (the non-scaled
//     v_mfma_f32_32x32x16_fp8_fp8 runs at the bf16 rate; MI355X_MICROARCH.md MFMA table). One instruction contracts 64
//     indices: 32 bytes of A and of B per lane (two ds_read_b128). Per tile and wave: 4 MFMAs for S^T = K Q^T (2 key
//     blocks x 2 halves of d) and 4 for O^T += V^T P^T (4 d-blocks x all 64 keys) instead of 16 + 16.
//     (-DLA_FP8_PLAIN_MFMA builds the non-scaled form on the same operand bytes for A/B runs.)
//   * the contraction index is permuted freely — A and B only have to agree: lane-half hh, 16-byte chunk j of QK^T
//     holds d = 32j + 16hh + [0,16) for K rows and Q fragments alike; MFMA s uses chunks 2s, 2s+1.
//   * no 8-bit transpose read: a prepare kernel (la_prep_v_fp8) rewrites V once per call into V^T tiles
//     [B, H, Kt][128 d][64 keys] whose 64-byte rows already hold the keys in the order the PV operand wants
//     (k-step pair j, lane-half hh, k-step parity, accumulator slot e <-> key 16kk + 4hh + (e&3) + 8(e>>2)) and are
//     already XOR-swizzled for conflict-free ds_read_b128; the forward kernel stages them with a linear LDS-DMA.
//     One extra pass over V (0.4 GB at S=75600, H=40) against ~60 TFLOP of attention.
//   * K tile 8 KiB + V^T tile 8 KiB per stage: 32 KiB of LDS double-buffered.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "la_fwd_common.h"

namespace la {

namespace {

constexpr int F8_D = 128;
constexpr int F8_BN = 64;
constexpr int F8_KROW = F8_D;                 // bytes per K row in LDS
constexpr int F8_TILE = F8_BN * F8_KROW;      // 8 KiB (K tile and V^T tile)
constexpr int F8_VROW = F8_BN;                // bytes per V^T row (64 keys)

typedef const __attribute__((address_space(1))) void* f8_gptr_t;
typedef __attribute__((address_space(3))) void* f8_lptr_t;
typedef long i64x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef long i64x4 __attribute__((ext_vector_type(4)));

constexpr int F8_SCALE_ONE = 0x7f7f7f7f;      // four E8M0 exponents of 127 = 2^0, whichever byte op_sel picks

// acc += A(32 x 64 e4m3) * B(64 x 32 e4m3): lane-half hh carries 32 of the 64 contraction indices as 32 bytes, given here
// as two 16-byte halves (lo = first 16 bytes). Unit block scales: exactly the fp32-accumulated e4m3 products.
__device__ __forceinline__ f32x16 f8_mfma_k64(const i64x2 a_lo, const i64x2 a_hi, const i64x2 b_lo, const i64x2 b_hi, f32x16 acc) {
#ifdef LA_FP8_PLAIN_MFMA
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a_lo[0], b_lo[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a_lo[1], b_lo[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a_hi[0], b_hi[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a_hi[1], b_hi[1], acc, 0, 0, 0);
    return acc;
#else
    const i64x4 a4 = {a_lo[0], a_lo[1], a_hi[0], a_hi[1]};
    const i64x4 b4 = {b_lo[0], b_lo[1], b_hi[0], b_hi[1]};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(__builtin_bit_cast(i32x8, a4), __builtin_bit_cast(i32x8, b4), acc,
                                                           0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, F8_SCALE_ONE, 0, F8_SCALE_ONE);
#endif
}

__device__ __forceinline__ void f8_dma16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((f8_gptr_t)gsrc, (f8_lptr_t)lds_dst, 16, 0, 0);
}
// K rows are 128 bytes = 8 chunks of 16: two rows per 256-byte bank row -> chunk ^ ((row>>1)&7)
__device__ __forceinline__ constexpr int f8_k_swz(int row) { return (row >> 1) & 7; }
// V^T rows are 64 bytes = 4 chunks: four rows per bank row -> chunk ^ ((row>>2)&3)
__device__ __forceinline__ constexpr int f8_v_swz(int row) { return (row >> 2) & 3; }

}  // namespace

// ------------------------------------------------------------------------------------------------
// Prepare kernel: V (B,Sk,H,128) e4m3 -> V^T tiles [B,H,Kt][128][64] (key order and swizzle as above).
// One workgroup per (b, h, k-tile); rows past seqlen_k become zeros (P is 0 there anyway).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) la_prep_v_fp8_kernel(const uint8_t* __restrict__ v, int64_t v_batch_stride,
                                                            int64_t v_row_stride, int64_t v_head_stride,
                                                            uint8_t* __restrict__ vt, int seqlen_k, int num_heads,
                                                            int k_tiles) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[F8_BN][F8_D + 16];   // [key][d], padded rows
    const int n = blockIdx.x % k_tiles;
    const int bh = blockIdx.x / k_tiles;
    const int h = bh % num_heads, b = bh / num_heads;
    const uint8_t* src = v + b * v_batch_stride + h * v_head_stride;
    const int tid = threadIdx.x;
    // coalesced load: 64 rows x 128 bytes = 512 chunks of 16 bytes, 2 per thread
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int cid = tid + 256 * it;
        const int row = cid >> 3, ch = cid & 7;
        u32x4 t = {0u, 0u, 0u, 0u};
        const int key = n * F8_BN + row;
        if (key < seqlen_k) t = *reinterpret_cast<const u32x4*>(src + static_cast<int64_t>(key) * v_row_stride + ch * 16);
        *reinterpret_cast<u32x4*>(&tile[row][ch * 16]) = t;
    }
    __syncthreads();
    // 128 rows (d) x 4 chunks of 16 bytes out, 2 per thread
    uint8_t* dst = vt + (static_cast<int64_t>(bh) * k_tiles + n) * F8_TILE;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int cid = tid + 256 * it;
        const int d = cid >> 2, cpos = cid & 3;
        const int ch = cpos ^ f8_v_swz(d);            // logical chunk = 2*j + hh
        const int j = ch >> 1, hh = ch & 1;
        uint32_t w[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            uint32_t acc = 0;
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
                const int byte = 4 * q4 + bi;             // 0..15 inside the chunk
                const int kk = 2 * j + (byte >> 3), e = byte & 7;
                const int key = 16 * kk + 4 * hh + (e & 3) + 8 * (e >> 2);
                acc |= static_cast<uint32_t>(tile[key][d]) << (8 * bi);
            }
            w[q4] = acc;
        }
        u32x4 o = {w[0], w[1], w[2], w[3]};
        *reinterpret_cast<u32x4*>(dst + d * F8_VROW + cpos * 16) = o;
    }
}

hipError_t launch_prep_v_fp8(const void* v, int64_t v_batch_stride, int64_t v_row_stride, int64_t v_head_stride,
                             void* vt, int batch, int seqlen_k, int num_heads, int k_tiles, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(la_prep_v_fp8_kernel, dim3(batch * num_heads * k_tiles), dim3(256), 0, stream,
                       static_cast<const uint8_t*>(v), v_batch_stride, v_row_stride, v_head_stride,
                       static_cast<uint8_t*>(vt), seqlen_k, num_heads, k_tiles);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Forward kernel
// ------------------------------------------------------------------------------------------------
template <bool SKIPABLE>
__global__ void __launch_bounds__(256, 2)
la_fwd_fp8_d128_kernel(const FwdParams p) {
    constexpr int BM = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const k_lds = smem;                      // [2][F8_TILE]
    unsigned char* const v_lds = smem + 2 * F8_TILE;        // [2][F8_TILE]  (V^T tiles)
    int* const meta = reinterpret_cast<int*>(smem + 4 * F8_TILE);
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
        if (tid == 0) meta[1] = next_work_item(p, 64, &meta[2]);     // 64 = workgroups co-resident on an XCD
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
    const int k_tiles = p.k_tiles;
    const int hk = h / p.h_ratio;                      // K/V head (GQA/MQA: h_ratio query heads share one)
    const int64_t list_off = (static_cast<int64_t>(bh) * p.q_tiles + m_block) * (k_tiles + 1);

    if (SKIPABLE) {
        for (int i = tid; i < 2 * ((k_tiles + 31) / 32); i += 256) doflags[i] = 0u;
        __syncthreads();
        if (wave == 0) {
            const int n = expand_read_list(p.read_list + list_off, seq, endflags, k_tiles, lane);
            if (lane == 0) meta[0] = n;
        }
    }

    // ---- descales (per batch, per K/V head — all three, flash_api.cpp:689-691, flash_fwd_kernel_sm90.h:509-510;
    // NULL = 1). c folds softmax_scale*log2(e) with q and k descales.
    const float qd = p.q_descale ? p.q_descale[b * p.q_descale_batch_stride + hk * p.q_descale_head_stride] : 1.f;
    const float kd = p.k_descale ? p.k_descale[b * p.k_descale_batch_stride + hk * p.k_descale_head_stride] : 1.f;
    const float vd = p.v_descale ? p.v_descale[b * p.v_descale_batch_stride + hk * p.v_descale_head_stride] : 1.f;
    const float c = p.scale_log2 * qd * kd;

    // ---- Q fragments: lane holds query row l31, d = 32*j + 16*hh + [0,16) for j = 0..3 (two k-steps each)
    const int q_row = m_block * BM + wave * 32 + l31;
    i64x2 qf[4];
    {
        const uint8_t* qp = reinterpret_cast<const uint8_t*>(p.q) + b * p.q_batch_stride +
                            static_cast<int64_t>(q_row) * p.q_row_stride + h * p.q_head_stride + hh * 16;
        const bool ok = q_row < p.seqlen_q;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 t = {0u, 0u, 0u, 0u};
            if (ok) t = *reinterpret_cast<const u32x4*>(qp + j * 32);
            qf[j] = __builtin_bit_cast(i64x2, t);
        }
    }

    // ---- LDS-DMA. K tile: 8 pieces of 1 KiB = 8 rows of 128 bytes; wave w stages rows 16w..16w+15 (pieces 2w, 2w+1);
    // lane: row rip = lane>>3, chunk position cpos = lane&7, source chunk cpos ^ f8_k_swz(row).
    // V^T tile: 8 KiB contiguous and pre-swizzled in the workspace: a linear copy.
    const uint8_t* const kg = reinterpret_cast<const uint8_t*>(p.k) + b * p.k_batch_stride + hk * p.k_head_stride;
    const uint8_t* const vtg = reinterpret_cast<const uint8_t*>(p.v) +
                               (static_cast<int64_t>(b) * (p.num_heads / p.h_ratio) + hk) * k_tiles * F8_TILE;
    const int k_rs = static_cast<int>(p.k_row_stride);   // bytes (1-byte elements)
    const int rip = lane >> 3;
    const int cpos = lane & 7;
    const int last_row = p.seqlen_k - 1;
    // per-lane byte offsets inside a full tile (row 16w + 8j + rip, swizzled source chunk): the per-step address is a
    // wave-uniform tile base + this 32-bit offset (saddr + voffset form: no per-step VALU address arithmetic)
    unsigned k_lane_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 16 * wave + 8 * j + rip;
        k_lane_off[j] = static_cast<unsigned>(r) * static_cast<unsigned>(k_rs) + ((cpos ^ f8_k_swz(r)) << 4);
    }
    auto dma_k = [&](int n, int kbuf) {
        const int row0 = n * F8_BN;
        if (__builtin_expect(row0 + F8_BN - 1 <= last_row, 1)) {         // wave-uniform: every tile but a ragged last one
            const uint8_t* base = kg + static_cast<int64_t>(row0) * k_rs;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                f8_dma16(base + k_lane_off[j], k_lds + kbuf * F8_TILE + (2 * wave + j) * 1024);
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 16 * wave + 8 * j + rip;
            const int grow = min(row0 + r, last_row);                    // rows past seqlen_k: clamp (masked / P = 0)
            f8_dma16(kg + static_cast<int64_t>(grow) * k_rs + ((cpos ^ f8_k_swz(r)) << 4),
                     k_lds + kbuf * F8_TILE + (2 * wave + j) * 1024);
        }
    };
    auto dma_v = [&](int n, int vbuf) {
        const uint8_t* src = vtg + static_cast<int64_t>(n) * F8_TILE + lane * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            f8_dma16(src + (2 * wave + j) * 1024, v_lds + vbuf * F8_TILE + (2 * wave + j) * 1024);
    };

    __syncthreads();   // seq / meta / flags visible
    const int n_tiles = SKIPABLE ? meta[0] : k_tiles;
    auto tile_at = [&](int i) -> int {
        const int ii = min(i, n_tiles - 1);
        return SKIPABLE ? __builtin_amdgcn_readfirstlane(seq[ii]) : (k_tiles - 1 - ii);
    };

    dma_k(tile_at(0), 0);
    dma_v(tile_at(0), 0);
    dma_k(tile_at(1), 1);
    __syncthreads();

    // ---- per-lane LDS read offsets
    // K A-operand: row = 32*kb + l31, 16-byte chunk 2*j + hh (two k-steps)  -> row*128 + ((chunk ^ swz(row)) << 4)
    const int k_rd = l31 * F8_KROW;
    const int k_rd_sw = f8_k_swz(l31);
    // V^T A-operand: row d = 32*db + l31, chunk 2*j + hh (k-steps 2j, 2j+1)  -> d*64 + ((chunk ^ swz(d)) << 4)
    const int v_rd = l31 * F8_VROW;
    const int v_rd_sw = f8_v_swz(l31);

    const float thr = p.thr;
    const int tail_valid = p.seqlen_k - (k_tiles - 1) * F8_BN;
    unsigned domask = 1u;
    float l_run = 0.f;
    f32x16 o_acc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[db][r] = 0.f;

    auto qk_tile = [&](int kbuf, f32x16 (&s)[2]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
            const unsigned char* kt = k_lds + kbuf * F8_TILE + kb * 32 * F8_KROW + k_rd;
#pragma unroll
            for (int sx = 0; sx < 2; ++sx) {                 // d = 64*sx + {16hh + [0,16), 32 + 16hh + [0,16)}
                const i64x2 k_lo = *reinterpret_cast<const i64x2*>(kt + (((4 * sx + hh) ^ k_rd_sw) << 4));
                const i64x2 k_hi = *reinterpret_cast<const i64x2*>(kt + (((4 * sx + 2 + hh) ^ k_rd_sw) << 4));
                s[kb] = f8_mfma_k64(k_lo, k_hi, qf[2 * sx], qf[2 * sx + 1], s[kb]);
            }
        }
    };

    float m_run = -INFINITY;
    auto stats = [&](f32x16 (&s)[2], int pos, bool valid) -> float {
        float m_loc = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) m_loc = fmaxf(m_loc, s[kb][r]);
        m_loc = half_swap_max(m_loc);
        if (!valid) m_loc = -INFINITY;
        const float m_prev = m_run;
        m_run = fmaxf(m_prev, m_loc);
        if (SKIPABLE) {
            const bool do_any = __any(((m_loc - m_prev) * c) > thr) && valid;     // softmax.h:194, c incl. descales
            domask |= (do_any ? 1u : 0u) << (pos & 31);
            if ((pos & 31) == 31 && valid) {
                if (lane == 0) atomicOr(&doflags[pos >> 5], domask);
                domask = 0u;
            }
        }
        return fast_exp2((m_prev - m_run) * c);
    };
    auto mask_tail = [&](f32x16 (&s)[2], int n) {
        if (__builtin_expect(n == k_tiles - 1 && tail_valid < F8_BN, 0)) {
            asm volatile("; seqlen-k mask" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= tail_valid) s[kb][r] = -INFINITY;
                }
        }
    };

    float alpha = 0.f;
    auto step = [&](int i, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2]) {
        const int cur = i & 1;
        const bool has_next = (i + 1) < n_tiles;
        const int n_next = tile_at(i + 1);
        dma_k(tile_at(i + 2), cur);          // K(i) was consumed by the previous step
        dma_v(n_next, cur ^ 1);              // V(i-1) likewise

        // ---- phase 1: QK^T of tile i+1  ||  P = exp2(S*c - m*c + 8) (Max_offset = 8), row sum, e4m3 P of tile i
        qk_tile(cur ^ 1, s_nxt);
        const float m_off = 8.f - m_run * c;
        float psum = 0.f;
        long pf[4];   // B operand of O^T += V^T P^T: k-step kk = accumulator regs 8*(kk&1).. of key block kk>>1
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = fast_exp2(__builtin_fmaf(s_cur[kb][r], c, m_off));
                s_cur[kb][r] = pv;
                psum += pv;
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                // `old` of the first convert is a value that dies here (the convert works in place on its register):
                // both halves are overwritten, so no zero has to be materialised per word
                int w0 = __float_as_int(s_cur[kb][8 * hf + 0]), w1 = __float_as_int(s_cur[kb][8 * hf + 4]);
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(s_cur[kb][8 * hf + 0], s_cur[kb][8 * hf + 1], w0, false);
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(s_cur[kb][8 * hf + 2], s_cur[kb][8 * hf + 3], w0, true);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(s_cur[kb][8 * hf + 4], s_cur[kb][8 * hf + 5], w1, false);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(s_cur[kb][8 * hf + 6], s_cur[kb][8 * hf + 7], w1, true);
                pf[2 * kb + hf] = static_cast<long>((static_cast<unsigned long>(static_cast<unsigned>(w1)) << 32) |
                                                    static_cast<unsigned long>(static_cast<unsigned>(w0)));
            }
        }
        l_run = l_run * alpha + psum;

        // ---- phase 2: O^T += V^T P^T of tile i  ||  stats of tile i+1
        const unsigned char* vt = v_lds + cur * F8_TILE + v_rd;
        const i64x2 p_lo = {pf[0], pf[1]}, p_hi = {pf[2], pf[3]};      // all 64 keys of the tile: one MFMA per d-block
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const i64x2 v_lo = *reinterpret_cast<const i64x2*>(vt + db * 32 * F8_VROW + ((hh ^ v_rd_sw) << 4));
            const i64x2 v_hi = *reinterpret_cast<const i64x2*>(vt + db * 32 * F8_VROW + (((2 + hh) ^ v_rd_sw) << 4));
            o_acc[db] = f8_mfma_k64(v_lo, v_hi, p_lo, p_hi, o_acc[db]);
        }
        alpha = stats(s_nxt, i + 1, has_next);
        if (!__all(alpha == 1.0f)) {
            float a_in = alpha;
            asm volatile("; rescale O" : "+v"(a_in));
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[db][r] *= a_in;
        }
        __syncthreads();
    };

    f32x16 s_a[2], s_b[2];
    {
        const int n0 = tile_at(0);
        qk_tile(0, s_a);
        mask_tail(s_a, n0);
        domask = 0u;
        (void)stats(s_a, 0, true);
        domask |= 1u;
        alpha = 0.f;
    }
    __syncthreads();
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

    // ---- finalize: l carries the 2^8 factor of P (softmax.h:288-292); O = O * v_descale / l (mainloop...:1852-1853)
    const float l_tot = half_swap_sum(l_run);
    const bool bad = (l_tot == 0.f) || (l_tot != l_tot);
    const float inv = bad ? 0.f : vd / l_tot;
    if (q_row < p.seqlen_q) {
        uint16_t* op = p.o + b * p.o_batch_stride + static_cast<int64_t>(q_row) * p.o_row_stride + h * p.o_head_stride;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                f32x4 x = {o_acc[db][4 * t] * inv, o_acc[db][4 * t + 1] * inv, o_acc[db][4 * t + 2] * inv,
                           o_acc[db][4 * t + 3] * inv};
                *reinterpret_cast<bf16x4*>(op + 32 * db + 8 * t + 4 * hh) = __builtin_convertvector(x, bf16x4);
            }
        }
        if (p.lse != nullptr && hh == 0)
            p.lse[static_cast<int64_t>(bh) * p.seqlen_q + q_row] =
                bad ? -INFINITY : m_run * (c * 0.69314718055994530942f) + __logf(l_tot * (1.f / 256.f));
    }

    if (SKIPABLE) {
        if (wave == 0 && p.write_list != nullptr) {
            const int* md = p.must_do_list ? (p.must_do_is_1d ? p.must_do_list : p.must_do_list + list_off) : nullptr;
            write_skip_list_wave(seq, endflags, doflags, n_tiles, p.write_list + list_off, md, k_tiles, lane);
        }
    }
    if (!dynamic) return;
    __syncthreads();   // every wave is done with this item's LDS before the next one is set up
  }
}

size_t fwd_lds_bytes_fp8(int k_tiles, int* seq_cap_out) {
    const int seq_cap = (k_tiles + 3) & ~3;
    if (seq_cap_out) *seq_cap_out = seq_cap;
    return 4 * F8_TILE + 16 + static_cast<size_t>(seq_cap) * 4 + 2 * static_cast<size_t>((k_tiles + 31) / 32) * 4 + 16;
}

size_t fp8_workspace_bytes(int batch, int num_heads, int k_tiles) {
    return static_cast<size_t>(batch) * num_heads * k_tiles * F8_TILE;
}

// p.v must already point at the V^T workspace written by launch_prep_v_fp8.
hipError_t launch_fwd_fp8_d128(const FwdParams& p, bool skipable, hipStream_t stream) {
    const int total = p.batch * p.num_heads * p.q_tile_count;
    FwdParams pp = p;
    const size_t lds = fwd_lds_bytes_fp8(p.k_tiles, &pp.seq_cap);
    hipError_t err;
    (void)hipGetLastError();
    if (skipable) {
        auto kfn = la_fwd_fp8_d128_kernel<true>;
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds));
        if (err != hipSuccess) return err;
        int grid = total;
        err = prepare_work_queue(pp, true, total, 2, stream, &grid);                       // two workgroups per CU
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), lds, stream, pp);
    } else {
        auto kfn = la_fwd_fp8_d128_kernel<false>;
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds));
        if (err != hipSuccess) return err;
        pp.work_counter = nullptr;
        hipLaunchKernelGGL(kfn, dim3(total), dim3(256), lds, stream, pp);
    }
    return hipGetLastError();
}

}  // namespace la
