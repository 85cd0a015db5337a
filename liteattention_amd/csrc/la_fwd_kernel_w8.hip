// la_fwd_kernel_w8.hip — QK-Skip attention forward, 8 waves / 256 query rows per workgroup, TWO skip-list
// tiles per workgroup (gfx950, bf16, head_dim 128).
//
// Same algorithm, list tiles (kBlockM, kBlockN) = (128, 64), LDS image, MFMA mapping and per-q-tile arithmetic as
// la_fwd_kernel_v2.hip — results are bit-identical to it. What changes is who shares K/V:
//
//   * measured on real (fragmented) skip lists, 128-row workgroups stop sharing K/V tiles in L2 (co-resident
//     workgroups drift apart in the walk): L2 hit rate 97 % -> 29 %, fabric traffic 5-7 TB/s, +22 % time over
//     ideal at 48 % sparsity (profiles/r01c). A workgroup here is TWO adjacent 128-row q-tiles A (waves 0-3)
//     and B (waves 4-7), each with its own read list, votes and write list, walking the UNION of the two lists
//     (descending). One K/V tile in LDS serves 256 query rows: K/V traffic and LDS-DMA issue per FLOP halve.
//     A half that does not own a union position skips that position's QK^T / softmax / PV (wave-uniform
//     branches) and only takes part in the barrier; adjacent q-tiles have near-identical lists, so the union
//     is barely longer than either list.
//   * one workgroup per CU (still 2 waves per SIMD) frees LDS for 3-deep K and V rings: tiles are staged TWO
//     steps ahead (counted vmcnt across a raw s_barrier), which covers an L2 miss.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "la_fwd_common.h"

namespace la {

namespace {

constexpr int W8_D = 128;
constexpr int W8_BN = 64;
constexpr int W8_ROW = W8_D * 2;                  // 256 bytes
constexpr int W8_TILE = W8_BN * W8_ROW;           // 16 KiB
constexpr int W8_STAGES = 3;
constexpr unsigned OWN_A = 1u << 30, OWN_B = 1u << 31, TILE_MASK = (1u << 30) - 1u;

typedef const __attribute__((address_space(1))) void* w8_gptr_t;
typedef __attribute__((address_space(3))) void* w8_lptr_t;
__device__ __forceinline__ void w8_dma16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((w8_gptr_t)gsrc, (w8_lptr_t)lds_dst, 16, 0, 0);
}

}  // namespace

size_t fwd_w8_lds_bytes(int k_tiles, int* seq_cap_out) {
    const int seq_cap = (k_tiles + 3) & ~3;
    if (seq_cap_out) *seq_cap_out = seq_cap;
    const size_t words = static_cast<size_t>((k_tiles + 31) / 32);
    return 2 * W8_STAGES * W8_TILE + 32 + 3 * static_cast<size_t>(seq_cap) * 4 + 6 * words * 4 + 16;
}

template <bool SKIPABLE>
__global__ void __launch_bounds__(512, 2)
la_fwd_bf16_d128_w8_kernel(const FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const k_lds = smem;                                   // [3][TILE]
    unsigned char* const v_lds = smem + W8_STAGES * W8_TILE;             // [3][TILE]
    int* const meta = reinterpret_cast<int*>(smem + 2 * W8_STAGES * W8_TILE);   // [8]: nA, nB, nU
    unsigned* const seq_u = reinterpret_cast<unsigned*>(meta + 8);       // union walk: tile | OWN_A | OWN_B
    int* const seq_h0 = reinterpret_cast<int*>(seq_u + p.seq_cap);       // per-half own sequences (for the writers)
    int* const seq_h1 = seq_h0 + p.seq_cap;
    const int words = (p.k_tiles + 31) / 32;
    unsigned* const flag0 = reinterpret_cast<unsigned*>(seq_h1 + p.seq_cap);    // do[2], end[2], bit[2] x words

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // 0..7
    const int half = wave >> 2;                                          // 0: q-tile A, 1: q-tile B
    const int wih = wave & 3;                                            // wave inside the half
    const int lane = tid & 63;
    const int hh = lane >> 5;
    const int l31 = lane & 31;

    unsigned* const do_f = flag0 + half * words;
    unsigned* const end_f = flag0 + (2 + half) * words;
    unsigned* const bit_f = flag0 + (4 + half) * words;
    int* const seq_own = half ? seq_h1 : seq_h0;

    const int k_tiles = p.k_tiles;
    const int qt2 = (p.q_tiles + 1) >> 1;
    const int vid = xcd_work_id<32>();
    const int m2 = vid % qt2;
    const int bh = vid / qt2;
    const int h = bh % p.num_heads;
    const int b = bh / p.num_heads;
    const int m_block = 2 * m2 + half;
    const bool present = m_block < p.q_tiles;                            // odd q-tile count: the last pair has no B
    const int64_t list_off = (static_cast<int64_t>(bh) * p.q_tiles + m_block) * (k_tiles + 1);

    if (SKIPABLE) {
        for (int i = tid; i < 6 * words; i += 512) flag0[i] = 0u;
        __syncthreads();
        if (wih == 0) {                                                  // waves 0 and 4: one list each
            int n = 0;
            if (present) {
                n = expand_read_list(p.read_list + list_off, seq_own, end_f, k_tiles, lane);
                for (int j = lane; j < n; j += 64) {
                    const int t = seq_own[j];
                    atomicOr(&bit_f[t >> 5], 1u << (t & 31));
                }
            }
            if (lane == 0) meta[half] = n;
        }
        __syncthreads();
        if (wave == 0) {                                                 // union of the two tile sets, descending
            const unsigned* bit_a = flag0 + 4 * words;
            const unsigned* bit_b = flag0 + 5 * words;
            int pos = 0;
            for (int base = 0; base < words; base += 64) {
                const int wi = words - 1 - (base + lane);
                unsigned ua = 0u, ub = 0u;
                if (wi >= 0) { ua = bit_a[wi]; ub = bit_b[wi]; }
                unsigned u = ua | ub;
                const int cnt = __popc(u);
                int incl = cnt;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int t = __shfl_up(incl, off);
                    if (lane >= off) incl += t;
                }
                int at = pos + incl - cnt;
                while (u) {
                    const int bpos = 31 - __clz(u);
                    seq_u[at++] = static_cast<unsigned>(wi * 32 + bpos) | (((ua >> bpos) & 1u) ? OWN_A : 0u) |
                                  (((ub >> bpos) & 1u) ? OWN_B : 0u);
                    u &= ~(1u << bpos);
                }
                pos += __shfl(incl, 63);
            }
            if (lane == 0) meta[2] = pos;
        }
    }

    // ---- Q fragments (B operand of S^T = K Q^T): query row l31, d = 16*ks + 8*hh + [0,8)
    const int q_row = m_block * 128 + wih * 32 + l31;
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

    // ---- LDS-DMA: a tile is 16 pieces of 1 KiB (4 rows); wave w moves pieces 2w, 2w+1 of K and of V.
    const unsigned char* const kg = reinterpret_cast<const unsigned char*>(p.k + b * p.k_batch_stride + h * p.k_head_stride);
    const unsigned char* const vg = reinterpret_cast<const unsigned char*>(p.v + b * p.v_batch_stride + h * p.v_head_stride);
    const int k_rs = static_cast<int>(p.k_row_stride * 2), v_rs = static_cast<int>(p.v_row_stride * 2);
    const int rip = lane >> 4;
    const int cpos = lane & 15;
    const int last_row = p.seqlen_k - 1;
    auto dma_k = [&](int n, int stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 8 * wave + 4 * j + rip;                         // row inside the tile
            const int grow = min(n * W8_BN + r, last_row);                // rows past seqlen_k: clamp (masked / P = 0)
            w8_dma16(kg + static_cast<int64_t>(grow) * k_rs + ((cpos ^ (r & 15)) << 4),
                     k_lds + stage * W8_TILE + (2 * wave + j) * 1024);
        }
    };
    auto dma_v = [&](int n, int stage) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 8 * wave + 4 * j + rip;
            const int grow = min(n * W8_BN + r, last_row);
            w8_dma16(vg + static_cast<int64_t>(grow) * v_rs + ((cpos ^ ((r & 3) << 2)) << 4),
                     v_lds + stage * W8_TILE + (2 * wave + j) * 1024);
        }
    };

    __syncthreads();   // lists, union, flags visible
    const int n_u = SKIPABLE ? meta[2] : k_tiles;
    const unsigned own_bit = half ? OWN_B : OWN_A;
    // packed union entry of position j (clamped): tile index + ownership bits
    auto entry_at = [&](int j) -> unsigned {
        const int jj = min(max(j, 0), n_u - 1);
        if (SKIPABLE) return __builtin_amdgcn_readfirstlane(seq_u[jj]);
        return static_cast<unsigned>(k_tiles - 1 - jj) | OWN_A | OWN_B;
    };
    auto owns = [&](int j, unsigned e) -> bool { return present && j >= 0 && j < n_u && (e & own_bit) != 0u; };

    // prologue: K(0), K(1), V(0) resident before the first step; ring slot of position j is j % 3
    dma_k(entry_at(0) & TILE_MASK, 0);
    dma_k(entry_at(1) & TILE_MASK, 1);
    dma_v(entry_at(0) & TILE_MASK, 0);
    __syncthreads();

    const int k_rd_row = l31 * W8_ROW;
    const int k_rd_sw = l31 & 15;
    const int a16 = lane & 15;
    const int v_key0 = 4 * hh + (a16 >> 2);
    int v_rd[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
        v_rd[db] = v_key0 * W8_ROW + (((db ^ (a16 >> 2)) << 6) | (((lane >> 4) & 1) << 5) | ((a16 & 3) << 3));

    const float c = p.scale_log2;
    const float thr = p.thr;
    const int tail_valid = p.seqlen_k - (k_tiles - 1) * W8_BN;
    unsigned domask = 0u;
    int own_pos = 0;                   // position inside THIS half's own list of the next tile it will get stats for
    float m_run = -INFINITY;
    float l_run = 0.f;
    float alpha = 0.f;
    f32x16 o_acc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[db][r] = 0.f;

    auto qk_tile = [&](int stage, f32x16 (&s)[2]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
            const unsigned char* kt = k_lds + stage * W8_TILE + kb * 32 * W8_ROW + k_rd_row;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt + (((2 * ks + hh) ^ k_rd_sw) << 4));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
            }
        }
    };

    // stats of an OWNED tile (scores in s): seqlen mask on the half's first tile (mask.h:44-78, mainloop...:1626),
    // row max, running max, skip vote (softmax.h:139-222). Returns the rescale factor.
    auto stats = [&](f32x16 (&s)[2], int n) -> float {
        if (__builtin_expect(own_pos == 0 && n == k_tiles - 1 && tail_valid < W8_BN, 0)) {
            asm volatile("; seqlen-k mask" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= tail_valid) s[kb][r] = -INFINITY;
                }
        }
        float m_loc = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) m_loc = fmaxf(m_loc, s[kb][r]);
        m_loc = half_swap_max(m_loc);
        const float m_prev = m_run;
        m_run = fmaxf(m_prev, m_loc);
        if (SKIPABLE) {
            // the first walked tile of a list is never flagged (softmax.h:153); else do_qk = (m_loc-m_prev)*c > thr
            const bool do_any = __any(((m_loc - m_prev) * c) > thr) || own_pos == 0;
            domask |= (do_any ? 1u : 0u) << (own_pos & 31);
            if ((own_pos & 31) == 31) {
                if (lane == 0) atomicOr(&do_f[own_pos >> 5], domask);
                domask = 0u;
            }
        }
        ++own_pos;
        return fast_exp2((m_prev - m_run) * c);
    };

    // One pipeline step on union position i (i = -1 is the fill step: only QK^T + stats of position 0).
    auto step = [&](int i, f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2]) {
        const unsigned e_cur = entry_at(i), e_nxt = entry_at(i + 1);
        const bool own_cur = owns(i, e_cur), own_nxt = owns(i + 1, e_nxt);
        // stage K(i+3) -> slot (i+3)%3 (K(i) is dead), V(i+2) -> slot (i+2)%3 (V(i-1) is dead)
        dma_k(entry_at(i + 3) & TILE_MASK, (i + 3) % 3);
        dma_v(entry_at(i + 2) & TILE_MASK, (i + 2) % 3);

        // ---- phase 1: QK^T of position i+1 (MFMA)  ||  P = exp2(S*c - m*c), row sum, bf16 P of position i (VALU)
        if (own_nxt) qk_tile((i + 1) % 3, s_nxt);
        bf16x8 pf[4];
        if (own_cur) {
            const float m_scaled = m_run * c;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = fast_exp2(__builtin_fmaf(s_cur[kb][r], c, -m_scaled));
                    s_cur[kb][r] = pv;
                    psum += pv;
                }
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    f32x8 t;
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = s_cur[kb][8 * hf + e];
                    pf[2 * kb + hf] = __builtin_convertvector(t, bf16x8);
                }
            }
            l_run = l_run * alpha + psum;

            // ---- phase 2a: O^T += V^T P^T of position i
            const unsigned char* vt = v_lds + ((i + 3) % 3) * W8_TILE;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        LDS_PTR(s16x4, vt + v_rd[db] + kk * 16 * W8_ROW));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        LDS_PTR(s16x4, vt + v_rd[db] + kk * 16 * W8_ROW + 8 * W8_ROW));
                    const s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    o_acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf[kk],
                                                                        o_acc[db], 0, 0, 0);
                }
            }
        }
        // ---- phase 2b: stats of position i+1, then O^T *= alpha (identity when alpha == 1 on every lane)
        if (own_nxt) {
            alpha = stats(s_nxt, static_cast<int>(e_nxt & TILE_MASK));
            if (!__all(alpha == 1.0f)) {
                float a_in = alpha;
                asm volatile("; rescale O" : "+v"(a_in));
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o_acc[db][r] *= a_in;
            }
        }
        // all waves done with this step's LDS reads; last step's DMA (K(i+2), V(i+1)) has landed, this step's 4
        // pieces per wave may still be in flight.
        asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    f32x16 s_a[2], s_b[2];
    step(-1, s_b, s_a);                        // fill: QK^T(0) -> s_a, stats(0)
    int i = 0;
    for (; i + 1 < n_u; i += 2) {
        step(i, s_a, s_b);
        step(i + 1, s_b, s_a);
    }
    if (i < n_u) step(i, s_a, s_b);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SKIPABLE) {
        if ((own_pos & 31) != 0 && lane == 0 && own_pos > 0) atomicOr(&do_f[(own_pos - 1) >> 5], domask);
        __syncthreads();
    }

    // ---- finalize (softmax.h:275-296) and store (epilogue_fwd.hpp:214-403)
    const float l_tot = half_swap_sum(l_run);
    const bool bad = (l_tot == 0.f) || (l_tot != l_tot);
    const float inv = bad ? 0.f : 1.f / l_tot;
    if (present && q_row < p.seqlen_q) {
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
                bad ? -INFINITY : m_run * (c * 0.69314718055994530942f) + __logf(l_tot);
    }

    if (SKIPABLE) {
        if (wih == 0 && present && p.write_list != nullptr) {                // waves 0 and 4: one writer wave per list
            const int* md = p.must_do_list ? (p.must_do_is_1d ? p.must_do_list : p.must_do_list + list_off) : nullptr;
            write_skip_list_wave(seq_own, end_f, do_f, meta[half], p.write_list + list_off, md, k_tiles, lane);
        }
    }
}

hipError_t launch_fwd_bf16_d128_w8(const FwdParams& p, bool skipable, hipStream_t stream) {
    const int qt2 = (p.q_tiles + 1) / 2;
    const int total = p.batch * p.num_heads * qt2;
    FwdParams pp = p;
    const size_t lds = fwd_w8_lds_bytes(p.k_tiles, &pp.seq_cap);
    hipError_t err;
    (void)hipGetLastError();
    if (skipable) {
        auto kfn = la_fwd_bf16_d128_w8_kernel<true>;
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds));
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(kfn, dim3(total), dim3(512), lds, stream, pp);
    } else {
        auto kfn = la_fwd_bf16_d128_w8_kernel<false>;
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(lds));
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(kfn, dim3(total), dim3(512), lds, stream, pp);
    }
    return hipGetLastError();
}

}  // namespace la
