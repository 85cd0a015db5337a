// la_aux_kernels.hip — small HBM-bound helpers around the forward kernel.
//
//  * skip_list_stats: number of listed tiles of a skip list, computed on the device. Replaces
//    LiteAttention.calc_percentage (/root/reference/hopper/lite_attention.py:61-85), whose
//    arithmetic is wrong; same input, corrected statistic.
//  * blockmask_to_lists: static 0/1 block mask -> read-list rows, one wave per row. The reference's block-sparse adapter converts
//    its mask on the host for a CUDA entry point that exists nowhere (/root/reference/flash_attn/flash_blocksparse_attn_interface.py:7-39,
//    185-200); here the mask becomes the skip lists the forward kernel walks (row format: mainloop_fwd_sm90_tma_gmma_ws.hpp:47-115).
//  * empty_k_fill: o = 0, lse = +inf for a call with seqlen_k == 0 (flash_api.cpp:1241-1245), on strided o.
//  * combine: LSE-weighted merge of partial outputs of K/V splits — the device counterpart of
//    attention_combine_ref (/root/reference/hopper/tests/test_flash_attn.py:1178-1187) and of the
//    reference's (compiled-out) FlashAttnFwdCombine kernel
//    (/root/reference/hopper/_internal/cpp/flash_fwd_combine_kernel.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "la_fwd_common.h"
#include "la_kernel_params.h"

namespace la {

__global__ void __launch_bounds__(256) skip_list_stats_kernel(const int32_t* __restrict__ list, int rows, int k_tiles,
                                                               unsigned long long* out) {
    // one wave per row; lanes stride over the ranges of the row
    const int lane = threadIdx.x & 63;
    const int wave_in_grid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    unsigned long long acc = 0;
    for (int r = wave_in_grid; r < rows; r += n_waves) {
        const int32_t* row = list + static_cast<int64_t>(r) * (k_tiles + 1);
        int len = row[0];
        if (len < 2) len = 2;   // the reader always walks the first range (mainloop...:93-101)
        if (len > k_tiles) len = k_tiles;
        for (int i = 1 + 2 * lane; i + 1 <= len; i += 128) {
            const int d = row[i] - row[i + 1] + 1;
            acc += d > 0 ? d : 0;
        }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0 && acc) atomicAdd(out, acc);
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = static_cast<unsigned long long>(rows);
}

hipError_t launch_skip_list_stats(const int32_t* list, int rows, int k_tiles, int64_t* out, hipStream_t stream) {
    hipError_t err = hipMemsetAsync(out, 0, 2 * sizeof(int64_t), stream);
    if (err != hipSuccess) return err;
    (void)hipGetLastError();
    int blocks = (rows + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(skip_list_stats_kernel, dim3(blocks), dim3(256), 0, stream, list, rows, k_tiles,
                       reinterpret_cast<unsigned long long*>(out));
    return hipGetLastError();
}

// One wave per (batch, head, q-tile) row. Position j of the descending walk is tile kt-1-j; a kept run [start .. end] (start >= end)
// becomes the pair (start, end) of the row, runs in descending order: row = [2 * runs, start0, end0, start1, end1, ..., 0 ...].
// HBM-bound byte work: kt mask bytes in, kt+1 ints out. A lane's position starts a run when it is kept and position j-1 is not, ends
// one when position j+1 is not: both neighbours come out of the BALLOT of the chunk (and of the chunks beside it), so every mask byte
// is loaded once; the loads of 8 chunks (512 positions) are issued together before the first ballot - a row of 1 182 tiles is 3
// memory latencies, not 19. STAGED (rows of up to ~4 000 tiles): the pairs are scattered into an LDS copy of the row and the row leaves in
// coalesced 256-byte wave stores, zero padding included; otherwise they go straight to global memory (partial, scattered stores).
// Round 4 at the Wan2.1 geometry (11 840 rows x 1 183 ints, 56-70 MB): first form 35-44 us, one load per byte 29-32 us, staged: tools/blockmask_bench.py.
template <bool STAGED>
__global__ void __launch_bounds__(256) blockmask_to_lists_kernel(const uint8_t* __restrict__ mask, int64_t mask_batch_stride,
                                                                  int64_t mask_head_stride, int batch, int num_heads, int q_tiles,
                                                                  int k_tiles, const int32_t* __restrict__ q_tiles_valid,
                                                                  const int32_t* __restrict__ k_tiles_valid,
                                                                  int32_t* __restrict__ lists, int32_t* __restrict__ empty_rows) {
    constexpr int G = 8;                         // chunks of 64 positions whose loads are in flight together
    extern __shared__ int32_t bm_rows[];         // STAGED: 4 waves x (k_tiles + 2) ints
    const int lane = threadIdx.x & 63;
    int32_t* const stage = bm_rows + (threadIdx.x >> 6) * (k_tiles + 2);
    const int64_t rows = static_cast<int64_t>(batch) * num_heads * q_tiles;
    const int64_t wave0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
    for (int64_t r = wave0; r < rows; r += n_waves) {
        const int m = static_cast<int>(r % q_tiles);
        const int64_t bh = r / q_tiles;
        const int h = static_cast<int>(bh % num_heads), b = static_cast<int>(bh / num_heads);
        const uint8_t* mrow = mask + b * mask_batch_stride + h * mask_head_stride + static_cast<int64_t>(m) * k_tiles;
        int32_t* out = lists + r * (k_tiles + 1);
        int kv = k_tiles_valid ? k_tiles_valid[b] : k_tiles;          // tiles >= kv do not exist for this sequence
        kv = kv < 0 ? 0 : (kv > k_tiles ? k_tiles : kv);
        const bool all = q_tiles_valid != nullptr && m >= q_tiles_valid[b];   // a q-tile past the sequence's end: never read; full corner
        int runs = 0;                                                  // kept runs that START at positions before this chunk
        unsigned long long before = 0ull;                              // bit 63: is the position in front of this chunk kept?
        for (int j0 = 0; j0 < k_tiles; j0 += 64 * G) {
            unsigned long long kb[G + 1];
#pragma unroll
            for (int c = 0; c <= G; ++c) {                             // G chunks + the first position behind them (one more ballot)
                const int j = j0 + 64 * c + lane, t = k_tiles - 1 - j;
                const bool keep = j < k_tiles && t < kv && (all || mrow[t] != 0);
                kb[c] = __ballot(keep);
            }
#pragma unroll
            for (int c = 0; c < G; ++c) {
                if (j0 + 64 * c >= k_tiles) break;                     // wave-uniform
                const int t = k_tiles - 1 - (j0 + 64 * c + lane);
                const bool keep = (kb[c] >> lane) & 1ull;
                const bool prev = lane == 0 ? (before >> 63) & 1ull : (kb[c] >> (lane - 1)) & 1ull;
                const bool next = lane == 63 ? kb[c + 1] & 1ull : (kb[c] >> (lane + 1)) & 1ull;
                const bool is_start = keep && !prev, is_end = keep && !next;
                const unsigned long long sb = __ballot(is_start);
                const int ridx = runs + __popcll(sb & ((2ull << lane) - 1ull)) - 1;       // the run this position belongs to
                int32_t* const dst = STAGED ? stage : out;
                if (is_start && 1 + 2 * ridx <= k_tiles) dst[1 + 2 * ridx] = t;
                if (is_end && 2 + 2 * ridx <= k_tiles) dst[2 + 2 * ridx] = t;              // a last end behind the row is counted, not stored
                runs += __popcll(sb);
                before = kb[c];
            }
        }
        if (STAGED) {
            // the pairs were scattered by some lanes and are copied out by others of the SAME wave: the hardware executes one wave's LDS
            // accesses in order, but nothing in the source language says so - a wave-scope fence + wave barrier keeps the compiler from
            // moving the loads above the stores (and, at the end, the next row's stores above these loads); no workgroup barrier is needed
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int i = lane; i <= k_tiles; i += 64) out[i] = i == 0 ? 2 * runs : (i <= 2 * runs ? stage[i] : 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            for (int i = 2 * runs + 1 + lane; i <= k_tiles; i += 64) out[i] = 0;
            if (lane == 0) out[0] = 2 * runs;
        }
        if (lane == 0 && runs == 0 && empty_rows != nullptr) atomicAdd(empty_rows, 1);
    }
}

hipError_t launch_blockmask_to_lists(const uint8_t* mask, int64_t mask_batch_stride, int64_t mask_head_stride, int batch,
                                     int num_heads, int q_tiles, int k_tiles, const int32_t* q_tiles_valid,
                                     const int32_t* k_tiles_valid, int32_t* lists, int32_t* empty_rows, hipStream_t stream) {
    if (empty_rows != nullptr) {
        const hipError_t err = hipMemsetAsync(empty_rows, 0, sizeof(int32_t), stream);
        if (err != hipSuccess) return err;
    }
    const int64_t rows = static_cast<int64_t>(batch) * num_heads * q_tiles;
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 16384) blocks = 16384;      // >> 256 CUs; rows beyond are taken by the grid-stride loop
    (void)hipGetLastError();
    const size_t lds = 4 * (static_cast<size_t>(k_tiles) + 2) * sizeof(int32_t);
    if (lds <= 64 * 1024)
        hipLaunchKernelGGL(blockmask_to_lists_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(256), lds, stream, mask, mask_batch_stride,
                           mask_head_stride, batch, num_heads, q_tiles, k_tiles, q_tiles_valid, k_tiles_valid, lists, empty_rows);
    else
        hipLaunchKernelGGL(blockmask_to_lists_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, mask, mask_batch_stride,
                           mask_head_stride, batch, num_heads, q_tiles, k_tiles, q_tiles_valid, k_tiles_valid, lists, empty_rows);
    return hipGetLastError();
}

// seqlen_k == 0: every output row is the empty sum. One thread = 8 consecutive d of one (b, s, h) row of the strided o.
__global__ void __launch_bounds__(256) empty_k_fill_kernel(uint16_t* __restrict__ o, float* __restrict__ lse,
                                                            int64_t o_batch_stride, int64_t o_row_stride, int64_t o_head_stride,
                                                            int batch, int seqlen_q, int num_heads, int dv) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int chunks = dv / 8;
    const int64_t total = static_cast<int64_t>(batch) * seqlen_q * num_heads * chunks;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int ch = static_cast<int>(idx % chunks);
        const int64_t row = idx / chunks;   // (b*S + s)*H + h
        const int h = static_cast<int>(row % num_heads);
        const int64_t bs = row / num_heads;
        const int s_ = static_cast<int>(bs % seqlen_q);
        const int b = static_cast<int>(bs / seqlen_q);
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(o + b * o_batch_stride + s_ * o_row_stride + h * o_head_stride + ch * 8) = z;
        if (lse != nullptr && ch == 0) lse[(static_cast<int64_t>(b) * num_heads + h) * seqlen_q + s_] = INFINITY;
    }
}

hipError_t launch_empty_k_fill(uint16_t* o, float* lse, int64_t o_batch_stride, int64_t o_row_stride, int64_t o_head_stride,
                               int batch, int seqlen_q, int num_heads, int head_dim_v, hipStream_t stream) {
    const int64_t total = static_cast<int64_t>(batch) * seqlen_q * num_heads * (head_dim_v / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(empty_k_fill_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, o, lse, o_batch_stride,
                       o_row_stride, o_head_stride, batch, seqlen_q, num_heads, head_dim_v);
    return hipGetLastError();
}

// la_combine_list: the partials of the splits are SEPARATE tensors (what the calls of a split attention return, e.g. the four calls of
// the reference's text / video recipe, README.md:225-246): stacking them first would move every partial once more than the merge itself.
constexpr int kCombineListMax = 8;
struct CombineList {
    const void* o[kCombineListMax];
    const float* lse[kCombineListMax];
};

// One thread = 8 consecutive d of one (b, s, h) row. Memory-bound; 16-byte accesses.
template <bool PARTIAL_16BIT, bool F16, bool OUT_F32 = false, bool LIST = false>     // 16-bit partials have the element type of o (bf16, or fp16 when F16); OUT_F32: o is fp32
__global__ void __launch_bounds__(256) combine_kernel(const void* __restrict__ o_partial,
                                                       const float* __restrict__ lse_partial,
                                                       uint16_t* __restrict__ o, float* __restrict__ lse_out,
                                                       int num_splits, int batch, int seqlen_q, int num_heads,
                                                       int dv, const CombineList list = CombineList{}, int64_t o_batch_stride = 0,
                                                       int64_t o_row_stride = 0, int64_t o_head_stride = 0) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    typedef typename Elem16<F16>::x8 ex8;
    const int chunks = dv / 8;
    const int64_t total = static_cast<int64_t>(batch) * seqlen_q * num_heads * chunks;
    const int64_t split_elems = static_cast<int64_t>(batch) * seqlen_q * num_heads * dv;
    const int64_t split_rows = static_cast<int64_t>(batch) * num_heads * seqlen_q;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int ch = static_cast<int>(idx % chunks);
        const int64_t row = idx / chunks;   // (b*S + s)*H + h
        const int hh = static_cast<int>(row % num_heads);
        const int64_t bs = row / num_heads;
        const int s = static_cast<int>(bs % seqlen_q);
        const int b = static_cast<int>(bs / seqlen_q);
        const int64_t lse_idx = (static_cast<int64_t>(b) * num_heads + hh) * seqlen_q + s;
        float m = -INFINITY;
        for (int sp = 0; sp < num_splits; ++sp) m = fmaxf(m, LIST ? list.lse[sp][lse_idx] : lse_partial[sp * split_rows + lse_idx]);
        float denom = 0.f;
        f32x8 acc = {0, 0, 0, 0, 0, 0, 0, 0};
        const float m_safe = (m == -INFINITY) ? 0.f : m;
        for (int sp = 0; sp < num_splits; ++sp) {
            const float l = LIST ? list.lse[sp][lse_idx] : lse_partial[sp * split_rows + lse_idx];
            const float w = (l == -INFINITY) ? 0.f : __expf(l - m_safe);
            denom += w;
            const int64_t off = (LIST ? 0 : sp * split_elems) + row * dv + ch * 8;
            const void* const src = LIST ? list.o[sp] : o_partial;
            f32x8 x;
            if (PARTIAL_16BIT) {
                const ex8 t = *reinterpret_cast<const ex8*>(static_cast<const uint16_t*>(src) + off);
                x = __builtin_convertvector(t, f32x8);
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(static_cast<const float*>(src) + off);
                const f32x4 c = *reinterpret_cast<const f32x4*>(static_cast<const float*>(src) + off + 4);
                x = __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            acc += x * w;
        }
        const float inv = denom > 0.f ? 1.f / denom : 0.f;
        acc *= inv;
        // the result: contiguous (b, s, h, d), or the caller's strided tensor (element strides; la_fwd's internal split of long dense walks)
        const int64_t o_off = o_row_stride != 0 ? b * o_batch_stride + s * o_row_stride + hh * o_head_stride + ch * 8 : row * dv + ch * 8;
        if constexpr (OUT_F32) {
            float* const of = reinterpret_cast<float*>(o) + o_off;
            *reinterpret_cast<f32x4*>(of) = __builtin_shufflevector(acc, acc, 0, 1, 2, 3);
            *reinterpret_cast<f32x4*>(of + 4) = __builtin_shufflevector(acc, acc, 4, 5, 6, 7);
        } else {
            *reinterpret_cast<ex8*>(o + o_off) = __builtin_convertvector(acc, ex8);
        }
        if (lse_out != nullptr && ch == 0) lse_out[lse_idx] = denom > 0.f ? m_safe + __logf(denom) : -INFINITY;
    }
}

template <bool P16, bool F16>
static void launch_combine_t(unsigned blocks, hipStream_t stream, const void* o_partial, const float* lse_partial, uint16_t* o,
                             float* lse, int num_splits, int batch, int seqlen_q, int num_heads, int head_dim_v, int64_t o_bs = 0,
                             int64_t o_rs = 0, int64_t o_hs = 0) {
    hipLaunchKernelGGL((combine_kernel<P16, F16>), dim3(blocks), dim3(256), 0, stream, o_partial, lse_partial, o, lse, num_splits,
                       batch, seqlen_q, num_heads, head_dim_v, CombineList{}, o_bs, o_rs, o_hs);
}

hipError_t launch_combine(const void* o_partial, bool partial_is_16bit, bool f16, const float* lse_partial, uint16_t* o,
                          float* lse, int num_splits, int batch, int seqlen_q, int num_heads, int head_dim_v,
                          hipStream_t stream, bool out_f32, int64_t o_bs, int64_t o_rs, int64_t o_hs) {
    const int64_t total = static_cast<int64_t>(batch) * seqlen_q * num_heads * (head_dim_v / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    (void)hipGetLastError();
    const unsigned g = static_cast<unsigned>(blocks);
    if (out_f32)     // fp32 partials -> fp32 result (the reference's default for fp32 partials, flash_attn_interface.py:684-685)
        hipLaunchKernelGGL((combine_kernel<false, false, true>), dim3(g), dim3(256), 0, stream, o_partial, lse_partial, o, lse, num_splits,
                           batch, seqlen_q, num_heads, head_dim_v);
    else if (partial_is_16bit && f16)
        launch_combine_t<true, true>(g, stream, o_partial, lse_partial, o, lse, num_splits, batch, seqlen_q, num_heads, head_dim_v, o_bs, o_rs, o_hs);
    else if (partial_is_16bit)
        launch_combine_t<true, false>(g, stream, o_partial, lse_partial, o, lse, num_splits, batch, seqlen_q, num_heads, head_dim_v, o_bs, o_rs, o_hs);
    else if (f16)
        launch_combine_t<false, true>(g, stream, o_partial, lse_partial, o, lse, num_splits, batch, seqlen_q, num_heads, head_dim_v);
    else
        launch_combine_t<false, false>(g, stream, o_partial, lse_partial, o, lse, num_splits, batch, seqlen_q, num_heads, head_dim_v);
    return hipGetLastError();
}

// The same merge over separately allocated partials (host arrays of num_splits <= 8 device pointers).
hipError_t launch_combine_list(const void* const* o_partials, bool partial_is_16bit, bool f16, const float* const* lse_partials, uint16_t* o,
                               float* lse, int num_splits, int batch, int seqlen_q, int num_heads, int head_dim_v, hipStream_t stream,
                               bool out_f32) {
    if (num_splits < 1 || num_splits > kCombineListMax) return hipErrorInvalidValue;
    CombineList list{};
    for (int i = 0; i < num_splits; ++i) { list.o[i] = o_partials[i]; list.lse[i] = lse_partials[i]; }
    const int64_t total = static_cast<int64_t>(batch) * seqlen_q * num_heads * (head_dim_v / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    (void)hipGetLastError();
    const dim3 g(static_cast<unsigned>(blocks)), b(256);
#define LA_COMBINE_LIST(P16, F16_, F32_) \
    hipLaunchKernelGGL((combine_kernel<P16, F16_, F32_, true>), g, b, 0, stream, nullptr, nullptr, o, lse, num_splits, batch, seqlen_q, num_heads, head_dim_v, list)
    if (out_f32) LA_COMBINE_LIST(false, false, true);
    else if (partial_is_16bit && f16) LA_COMBINE_LIST(true, true, false);
    else if (partial_is_16bit) LA_COMBINE_LIST(true, false, false);
    else if (f16) LA_COMBINE_LIST(false, true, false);
    else LA_COMBINE_LIST(false, false, false);
#undef LA_COMBINE_LIST
    return hipGetLastError();
}

}  // namespace la
