// la_kernel_params.h — device-side parameter block (the slimmed-down Flash_fwd_params,
// /root/reference/hopper/_internal/cpp/flash.h:48-185) filled by la_api.hip from la_fwd_args.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace la {

struct FwdParams {
    const uint16_t* q;
    const uint16_t* k;
    const uint16_t* v;
    uint16_t* o;
    float* lse;
    int64_t q_batch_stride, q_row_stride, q_head_stride;   // elements
    int64_t k_batch_stride, k_row_stride, k_head_stride;
    int64_t v_batch_stride, v_row_stride, v_head_stride;
    int64_t o_batch_stride, o_row_stride, o_head_stride;
    int batch, seqlen_q, seqlen_k, num_heads;
    int h_ratio;            // num_heads / num_heads_k: query heads per K/V head (GQA/MQA; kv head = h / h_ratio)
    int q_tiles, k_tiles;
    int q_tile_begin;       // this launch covers q-tiles [q_tile_begin, q_tile_begin + q_tile_count) of every (batch, head):
    int q_tile_count;       // a window of the SAME problem (tensors, LSE and lists are indexed by the global q-tile)
    int list_q_tiles;       // rows of the skip lists per (batch, head). == q_tiles, except under LA_FLAG_HALF_VOTE (x64 kernel, head_dim 128): the
                            // lists are kept per 128-ROW HALF of a 256-row workgroup item: list_q_tiles = ceil(seqlen_q / 128), and item m
                            // reads / writes rows 2 m and 2 m + 1 (q_tiles, q_tile_begin, q_tile_count stay in items of 256 rows)
    int half_vote;          // 1 = that mode
    int seq_cap;            // int32 slots reserved in LDS for the expanded tile sequence
    int walk_buffers;       // x64 kernels: 2 = a second walk buffer (tile sequence + vote / range-end flags) fits in LDS beside the first: the next
                            // item's read list is expanded while this item's write list is serialised; 1 = serial (very long key sequences, head_dim > 128)
    float scale_log2;       // softmax_scale * log2(e)   (flash_api.cpp:125-126)
    float rescale_tau;      // x64 kernel: O/l follow the running max only when it grew by more than this (log2 units)
    float thr;
    unsigned* work_counter; // x64 + lists: zeroed ticket counter in caller workspace -> persistent workgroups take work items
                            // dynamically (skip lists make items unequal); nullptr -> one workgroup per item, static XCD map
    const int* read_list;
    int* write_list;
    const int* must_do_list;
    int must_do_is_1d;
    // fp8 only: per-(batch, head) descales, NULL = 1.0 (flash_api.cpp:1003-1022); strides in elements
    const float* q_descale;
    const float* k_descale;
    const float* v_descale;
    int64_t q_descale_batch_stride, q_descale_head_stride;
    int64_t k_descale_batch_stride, k_descale_head_stride;
    int64_t v_descale_batch_stride, v_descale_head_stride;
    // variable-length batches (dense only): device prefix sums int32[batch + 1]; seqlen_q / seqlen_k / q_tiles / k_tiles above are
    // then the MAXIMA over the batch, lse is (H, total_q). nullptr = fixed length.
    const int* cu_seqlens_q;
    const int* cu_seqlens_k;
    int64_t total_q;
};

// Varlen launches (la_fwd_args.cu_seqlens_*): sequence b's own rows and lengths.
struct SeqView {
    int seqlen_q, seqlen_k, k_tiles;
    int64_t q_off, k_off, v_off, o_off;   // element offsets of the sequence's first row in q / k / v / o
    float* lse_row0;                      // &lse[row 0 of this (batch, head)] (may be derived from nullptr: check p.lse)
};
__device__ __forceinline__ SeqView seq_view(const FwdParams& p, int b, int h, int block_n) {
    SeqView s;
    const int q0 = p.cu_seqlens_q[b], k0 = p.cu_seqlens_k[b];
    s.seqlen_q = min(max(p.cu_seqlens_q[b + 1] - q0, 0), p.seqlen_q);
    s.seqlen_k = min(max(p.cu_seqlens_k[b + 1] - k0, 0), p.seqlen_k);
    s.k_tiles = (s.seqlen_k + block_n - 1) / block_n;
    s.q_off = q0 * p.q_row_stride; s.k_off = k0 * p.k_row_stride; s.v_off = k0 * p.v_row_stride; s.o_off = q0 * p.o_row_stride;
    s.lse_row0 = p.lse + static_cast<int64_t>(h) * p.total_q + q0;
    return s;
}

// Dynamic work distribution: zero the ticket counter on `stream` and return the persistent grid (workgroups per CU x CUs,
// capped by the number of items). work_counter == nullptr -> static: one workgroup per item.
int compute_units();
inline hipError_t prepare_work_queue(FwdParams& p, bool skipable, int total, int wg_per_cu, hipStream_t stream, int* grid) {
    *grid = total;
    if (!skipable) p.work_counter = nullptr;        // dense: every item costs the same, the static map is balanced
    if (p.work_counter == nullptr) return hipSuccess;
    constexpr size_t kWorkQueueBytes = 16 * 64;     // == kSchedWorkspaceBytes of la_api.hip (what la_fwd_workspace_bytes asks the caller for)
    static_assert(kWorkQueueBytes == 1024, "la_api.hip promises callers a 1024-byte scheduler workspace: change both together");
    const hipError_t err = hipMemsetAsync(p.work_counter, 0, kWorkQueueBytes, stream);       // 8 ticket queues, one 64-byte line each (+ 8 lines the LA_SCHED_GANG A/B build counts finished items in)
    if (err != hipSuccess) return err;
    const int slots = wg_per_cu * compute_units();
    *grid = total < slots ? total : slots;
    return hipSuccess;
}

size_t fwd_lds_bytes_v2(int head_dim, int k_tiles, int* seq_cap_out);
hipError_t launch_fwd_bf16_v2(const FwdParams& p, int head_dim, bool skipable, bool f16, hipStream_t stream);   // 128-row hipcc-scheduled template: head_dim 64; 128 / 256 as the A/B kernels (LA_FLAG_KERNEL_128ROW)
size_t fwd_lds_bytes_x64(int k_tiles, int* seq_cap_out, int head_dim, int* walk_buffers_out = nullptr, bool half_vote = false);
int x64_workgroups_per_cu(int head_dim);      // resident workgroups per CU of the hand-scheduled kernels: 1
hipError_t launch_fwd_x64(const FwdParams& p, int head_dim, bool skipable, bool f16, hipStream_t stream);  // p.half_vote: the half-vote form (head_dim 128, lists only)  // 1 wave/SIMD; head_dim 128: 64 rows/wave, q-tile 256; 256: 32 rows/wave, q-tile 128
size_t fwd_lds_bytes_x64_fp8(int k_tiles, int* seq_cap_out, int* walk_buffers_out = nullptr, int head_dim = 128);
hipError_t launch_fwd_x64_fp8(const FwdParams& p, bool skipable, int p_mode, int head_dim, hipStream_t stream);   // 1 wave/SIMD, 64 rows/wave; p.v = V^T workspace
size_t fp8_workspace_bytes(int batch, int num_heads_k, int k_tiles, int head_dim);   // prepared V^T tiles: 64 keys x head_dim bytes each
hipError_t launch_prep_v_fp8(const void* v, int64_t v_batch_stride, int64_t v_row_stride, int64_t v_head_stride,
                             void* vt, int batch, int seqlen_k, int num_heads_k, int k_tiles, int head_dim, hipStream_t stream,
                             const int* cu_seqlens_k = nullptr);
hipError_t launch_empty_k_fill(uint16_t* o, float* lse, int64_t o_batch_stride, int64_t o_row_stride, int64_t o_head_stride,
                               int batch, int seqlen_q, int num_heads, int head_dim_v, hipStream_t stream);
hipError_t launch_skip_list_stats(const int32_t* list, int rows, int k_tiles, int64_t* out, hipStream_t stream);
hipError_t launch_blockmask_to_lists(const uint8_t* mask, int64_t mask_batch_stride, int64_t mask_head_stride, int batch,
                                     int num_heads, int q_tiles, int k_tiles, const int32_t* q_tiles_valid,
                                     const int32_t* k_tiles_valid, int32_t* lists, int32_t* empty_rows, hipStream_t stream);
hipError_t launch_combine(const void* o_partial, bool partial_is_16bit, bool f16, const float* lse_partial, uint16_t* o,
                          float* lse, int num_splits, int batch, int seqlen_q, int num_heads, int head_dim_v,
                          hipStream_t stream, bool out_f32 = false, int64_t o_batch_stride = 0, int64_t o_row_stride = 0,
                          int64_t o_head_stride = 0);      // strides (elements) of a strided 16-bit result; 0 = contiguous

hipError_t launch_combine_list(const void* const* o_partials, bool partial_is_16bit, bool f16, const float* const* lse_partials, uint16_t* o,
                               float* lse, int num_splits, int batch, int seqlen_q, int num_heads, int head_dim_v, hipStream_t stream,
                               bool out_f32 = false);

}  // namespace la
