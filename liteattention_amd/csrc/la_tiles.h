// la_tiles.h — the ONE table of (kBlockM, kBlockN) used by the HIP kernels, the C-ABI
// (la_get_tile_sizes) and, through it, the Python host code (LiteAttention.get_MN).
//
// Replaces tile_size_fwd_sm90 (/root/reference/hopper/_internal/cpp/tile_size.h:10-62) and its
// hand-copied Python twin (/root/reference/hopper/lite_attention.py:87-111). The reference keeps two
// copies that must agree; here Python asks the library.
//
// CDNA4 derivation (HISTORY.md §3, §4.6). kBlockN = 64 keys: K 16 KiB + V 16 KiB per stage, double-buffered = 64 KiB
// of LDS. bf16 head_dim 128 (the headline path): ONE wave per SIMD owning the whole 512-entry register file and 64
// query rows (two 32x32 MFMA column blocks), a workgroup is 4 waves -> kBlockM = 256; 256 x 64 = 16 Ki scores per
// skip decision (reference Hopper tile: 128 x 176 = 22 Ki). fp8 head_dim 128 uses the same structure (kBlockM = 256). bf16
// head_dim 256 / 192 keep 32 rows per wave, kBlockM = 128; 96 and 64 are the 128 kernel with fewer fragments (kBlockM = 256). LA_FLAG_KERNEL_128ROW (la_fwd_args.flags) selects the 128-row A/B
// kernel for bf16 head_dim 128, and la_get_tile_sizes_ex then reports 128 (la_api.hip).
#pragma once

namespace la {

struct TileShape {
    int block_m;
    int block_n;
};

// element_size: 2 = bf16/fp16, 1 = fp8. Returns {0,0} when no kernel is instantiated.
constexpr TileShape tile_shape(int head_dim, int element_size) {
    if (element_size == 2) {
        if (head_dim == 128 || head_dim == 96) return {256, 64};    // 96: the 128 kernel with 12 of the 16 fragments per tile
        if (head_dim == 64) return {256, 64};   // round 3: the hand-scheduled form too (8 of the 16 fragments, LDS rows of 128 bytes);
                                                // LA_FLAG_KERNEL_128ROW: the hipcc-scheduled 128-row template (round 2's kernel)
        if (head_dim == 256 || head_dim == 192) return {128, 64};   // K/V tile 32 KiB each, one workgroup per CU, one wave per SIMD
                                                                    // with 32 rows (O^T of 32 rows x 256 is 128 registers); 192: 24 of 32 fragments
    }
    if (element_size == 1) {
        if (head_dim == 128 || head_dim == 64 || head_dim == 96) return {256, 64};  // fp8 e4m3: x64 structure on the block-scaled MFMA; K 8 KiB + V^T 8 KiB per stage
                                                                  // (round 6: head_dim 64 natively, half the MFMAs under the same softmax: 4 KiB + 4 KiB; 96: three d-blocks of O^T)
        if (head_dim == 192 || head_dim == 256) return {128, 64};   // round 6: one 32-row q-block per wave (O^T of 32 rows x 256 is 128 accumulators),
                                                                    // 3 / 4 contraction steps, 6 / 8 d-blocks; rings of 16 KiB per stage
    }
    return {0, 0};
}

}  // namespace la
