// la_prep_fp8.hip — fp8 (OCP e4m3fn) V -> prepared V^T tiles for the fp8 forward kernel (la_fwd_kernel_x64_fp8.hip).
//
// BASELINE.json configs[4]. The reference transposes V in shared memory inside its kernel for the PV operand
// (hopper/_internal/cpp/mainloop_fwd_sm90_tma_gmma_ws.hpp:942-984); gfx950 has no 8-bit transpose read, so a prepare kernel
// rewrites V ONCE per call into V^T tiles [B, Hk, Kt][D d][64 keys] whose 64-byte rows already hold the keys in the order
// the PV operand of the block-scaled MFMA wants (k-step pair j, lane-half hh, k-step parity, accumulator slot
// e <-> key 16kk + 4hh + (e&3) + 8(e>>2)) and are already XOR-swizzled for conflict-free ds_read_b128; the forward kernel
// stages them with a linear LDS-DMA. One extra pass over V (0.4 GB at S=75600, H=40) against ~60 TFLOP of attention.
// The workspace is caller-owned (C-ABI la_fwd_workspace_bytes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "la_fwd_common.h"

namespace la {

namespace {

constexpr int F8_BN = 64;
constexpr int F8_VROW = F8_BN;                // bytes per V^T row (64 keys)

// V^T rows are 64 bytes = 4 chunks: four rows per bank row -> chunk ^ ((row>>2)&3)
__device__ __forceinline__ constexpr int f8_v_swz(int row) { return (row >> 2) & 3; }

}  // namespace

// ------------------------------------------------------------------------------------------------
// Prepare kernel: V (B,Sk,H,D) e4m3 -> V^T tiles [B,H,Kt][D][64] (key order and swizzle as above), D = 128 or 64 (a head_dim-64 tile is
// the first 64 rows of what a head_dim-128 tile would be: 4 KiB; 192 / 256: 12 / 16 KiB).
// One workgroup per (b, h, k-tile); rows past seqlen_k become zeros (P is 0 there anyway).
// Packed variable-length batches (cu_seqlens_k != nullptr): v is (total_k, H, 128), sequence b owns rows [cu[b], cu[b + 1]) and its
// tiles are written to the same [B, H, Kt] grid (Kt = tiles of the longest sequence; the tiles past a sequence's end are zeros).
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256) la_prep_v_fp8_kernel(const uint8_t* __restrict__ v, int64_t v_batch_stride,
                                                            int64_t v_row_stride, int64_t v_head_stride,
                                                            uint8_t* __restrict__ vt, int seqlen_k, int num_heads,
                                                            int k_tiles, const int* __restrict__ cu_seqlens_k) {
    constexpr int DP = D == 96 ? 128 : D;          // rows of the tile written (96: zero rows up to the 128 tile, so that the forward copies it in 1 KiB pieces)
    __shared__ __attribute__((aligned(16))) uint8_t tile[F8_BN][D + 16];   // [key][d], padded rows
    const int n = blockIdx.x % k_tiles;
    const int bh = blockIdx.x / k_tiles;
    const int h = bh % num_heads, b = bh / num_heads;
    const uint8_t* src = v + b * v_batch_stride + h * v_head_stride;
    if (cu_seqlens_k != nullptr) {
        const int k0 = cu_seqlens_k[b];
        seqlen_k = min(max(cu_seqlens_k[b + 1] - k0, 0), seqlen_k);
        src = v + static_cast<int64_t>(k0) * v_row_stride + h * v_head_stride;
    }
    const int tid = threadIdx.x;
    // coalesced load: 64 rows x D bytes = 4 D chunks of 16 bytes, up to DP / 64 per thread
    constexpr int kPerThread = DP / 64, kRowChunks = D / 16;
#pragma unroll
    for (int it = 0; it < kPerThread; ++it) {
        const int cid = tid + 256 * it;
        if (cid >= F8_BN * kRowChunks) continue;
        const int row = cid / kRowChunks, ch = cid % kRowChunks;
        u32x4 t = {0u, 0u, 0u, 0u};
        const int key = n * F8_BN + row;
        if (key < seqlen_k) t = *reinterpret_cast<const u32x4*>(src + static_cast<int64_t>(key) * v_row_stride + ch * 16);
        *reinterpret_cast<u32x4*>(&tile[row][ch * 16]) = t;
    }
    __syncthreads();
    // DP rows (d) x 4 chunks of 16 bytes out, DP / 64 per thread
    uint8_t* dst = vt + (static_cast<int64_t>(bh) * k_tiles + n) * (F8_BN * DP);
#pragma unroll
    for (int it = 0; it < kPerThread; ++it) {
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
                if (d < D) acc |= static_cast<uint32_t>(tile[key][d]) << (8 * bi);
            }
            w[q4] = acc;
        }
        u32x4 o = {w[0], w[1], w[2], w[3]};
        *reinterpret_cast<u32x4*>(dst + d * F8_VROW + cpos * 16) = o;
    }
}

hipError_t launch_prep_v_fp8(const void* v, int64_t v_batch_stride, int64_t v_row_stride, int64_t v_head_stride,
                             void* vt, int batch, int seqlen_k, int num_heads, int k_tiles, int head_dim, hipStream_t stream,
                             const int* cu_seqlens_k) {
    (void)hipGetLastError();
    if (head_dim == 256)
        hipLaunchKernelGGL(la_prep_v_fp8_kernel<256>, dim3(batch * num_heads * k_tiles), dim3(256), 0, stream,
                           static_cast<const uint8_t*>(v), v_batch_stride, v_row_stride, v_head_stride,
                           static_cast<uint8_t*>(vt), seqlen_k, num_heads, k_tiles, cu_seqlens_k);
    else if (head_dim == 192)
        hipLaunchKernelGGL(la_prep_v_fp8_kernel<192>, dim3(batch * num_heads * k_tiles), dim3(256), 0, stream,
                           static_cast<const uint8_t*>(v), v_batch_stride, v_row_stride, v_head_stride,
                           static_cast<uint8_t*>(vt), seqlen_k, num_heads, k_tiles, cu_seqlens_k);
    else if (head_dim == 96)
        hipLaunchKernelGGL(la_prep_v_fp8_kernel<96>, dim3(batch * num_heads * k_tiles), dim3(256), 0, stream,
                           static_cast<const uint8_t*>(v), v_batch_stride, v_row_stride, v_head_stride,
                           static_cast<uint8_t*>(vt), seqlen_k, num_heads, k_tiles, cu_seqlens_k);
    else if (head_dim == 64)
        hipLaunchKernelGGL(la_prep_v_fp8_kernel<64>, dim3(batch * num_heads * k_tiles), dim3(256), 0, stream,
                           static_cast<const uint8_t*>(v), v_batch_stride, v_row_stride, v_head_stride,
                           static_cast<uint8_t*>(vt), seqlen_k, num_heads, k_tiles, cu_seqlens_k);
    else
        hipLaunchKernelGGL(la_prep_v_fp8_kernel<128>, dim3(batch * num_heads * k_tiles), dim3(256), 0, stream,
                           static_cast<const uint8_t*>(v), v_batch_stride, v_row_stride, v_head_stride,
                           static_cast<uint8_t*>(vt), seqlen_k, num_heads, k_tiles, cu_seqlens_k);
    return hipGetLastError();
}

size_t fp8_workspace_bytes(int batch, int num_heads, int k_tiles, int head_dim) {
    return static_cast<size_t>(batch) * num_heads * k_tiles * F8_BN * (head_dim == 96 ? 128 : head_dim);     // (96: tiles padded to the 128 tile)
}

}  // namespace la
