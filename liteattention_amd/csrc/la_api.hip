// la_api.hip — the C-ABI (include/lite_attention_amd.h): validation, parameter fill, dispatch.
//
// Replaces mha_fwd + set_params_fprop + run_mha_fwd of the reference
// (/root/reference/hopper/_internal/cpp/flash_api.cpp:45-163, 362-380, 667-1249). The checks mirror
// the reference's TORCH_CHECKs (cited per check) but return codes instead of throwing; the Python
// layer turns codes into the reference's exception types.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lite_attention_amd.h"
#include "la_kernel_params.h"
#include "la_tiles.h"

namespace {
thread_local int g_last_hip_error = 0;     // per-thread error detail of the last failed launch; the only state in the library

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Kernel selection is a pure function of the arguments (no environment variables, no process-wide statics):
//   bf16 / fp16 head_dim 128: the 256-row hand-scheduled kernel (x64) unless LA_FLAG_KERNEL_128ROW asks for the 128-row one (v2);
//   head_dim 256: the hand-scheduled kernel in its 32-rows-per-wave form (q-tile 128) unless LA_FLAG_KERNEL_128ROW asks for the
//   hipcc-scheduled v2 instantiation (same tiles); head_dim 96 / 192: the 128 / 256 hand-scheduled forms with three quarters of
//   the fragments (no v2 instantiation: LA_ERR_HEAD_DIM under LA_FLAG_KERNEL_128ROW, hosts pad to 128 / 256 then);
//   head_dim 64: the hand-scheduled kernel with 8 + 8 fragments per tile (q-tile 256) unless LA_FLAG_KERNEL_128ROW asks for v2 (q-tile 128);
//   fp8 head_dim 128: x64-fp8. The skip lists are indexed by the selected kernel's tile, so
//   la_get_tile_sizes_ex and la_fwd must agree on it: both call uses_128row().
constexpr uint32_t kKnownFlags = LA_FLAG_V_PREPARED | LA_FLAG_STATIC_SCHED | LA_FLAG_KERNEL_128ROW | LA_FLAG_EXACT_RESCALE |
                                LA_FLAG_FP8_MFMA_ROWSUM | LA_FLAG_FP8_ENCODED_P | LA_FLAG_HALF_VOTE;
bool uses_128row(int head_dim, int element_size, uint32_t flags) {      // the flag changes the q-tile (256 -> 128 rows) at head dims 64 and 128
    return element_size == 2 && (head_dim == 128 || head_dim == 64) && (flags & LA_FLAG_KERNEL_128ROW) != 0;
}
// LA_FLAG_HALF_VOTE: the hand-scheduled kernel with skip lists per 128-row half of its 256-row workgroup (bf16 / fp16 head dims 64 / 96 / 128,
// the kernels with a 256-row q-tile; no effect elsewhere, and LA_FLAG_KERNEL_128ROW - another kernel with the same list geometry - wins
// when both are set)
bool uses_half_vote(int head_dim, int element_size, uint32_t flags) {
    return element_size == 2 && (head_dim == 128 || head_dim == 96 || head_dim == 64) && (flags & LA_FLAG_HALF_VOTE) != 0 &&
           (flags & LA_FLAG_KERNEL_128ROW) == 0;
}
// Long DENSE key ranges (hand-scheduled kernels; e4m3 too: la_fwd's fp8 branch). A workgroup keeps the tile-address table of its walk in LDS, which bounds the key
// tiles of ONE launch (~4 800 at head_dim <= 128, ~1 600 at 192 / 256). A skip list names its tiles over the whole key range and keeps that
// bound (LA_ERR_SEQLEN); a dense launch does not need it: with the workspace la_fwd_workspace_bytes() asks for, la_fwd cuts the keys into
// the fewest equal runs of tiles that fit, runs them as launches on partial O / LSE buffers in the workspace and merges them by LSE
// (la_combine's kernel) - the reference has no such bound (its producer reads the list from global memory, mainloop...:47-115).
size_t walk_lds_bytes(int k_tiles, int head_dim, int element_size) {
    return element_size == 1 ? la::fwd_lds_bytes_x64_fp8(k_tiles, nullptr, nullptr, head_dim) : la::fwd_lds_bytes_x64(k_tiles, nullptr, head_dim);
}
int dense_tiles_per_launch(int head_dim, int element_size) {
    int lo = 1, hi = 1 << 20;                         // largest k_tiles whose walk fits (the LDS bytes grow with k_tiles)
    while (lo < hi) {
        const int mid = lo + (hi - lo + 1) / 2;
        if (walk_lds_bytes(mid, head_dim, element_size) <= 160 * 1024) lo = mid; else hi = mid - 1;
    }
    return lo;
}
struct DenseSplit { int n, chunk_tiles; uint64_t o_bytes, lse_bytes; };
// (the partial O of a run is 16-bit for every input type: bf16 / fp16 as the inputs, bf16 for e4m3)
DenseSplit dense_split(int k_tiles, int head_dim, int64_t batch, int64_t seqlen_q, int64_t num_heads, int64_t head_dim_v, int element_size = 2) {
    const int per = dense_tiles_per_launch(head_dim, element_size);
    DenseSplit d{};
    d.n = (k_tiles + per - 1) / per;
    d.chunk_tiles = (k_tiles + d.n - 1) / d.n;
    d.o_bytes = ((static_cast<uint64_t>(batch) * seqlen_q * num_heads * head_dim_v * 2) + 15) & ~15ull;
    d.lse_bytes = static_cast<uint64_t>(batch) * num_heads * seqlen_q * 4;      // (exact: the merge reads run s at s * (elements of one partial); the O partials - always a
                                                                                // multiple of 16 bytes, head_dim_v % 8 == 0 - come first, so every O partial is 16-byte aligned)
    return d;
}
constexpr uint64_t kSchedWorkspaceBytes = 1024;   // 16 ticket / steal counters of 64 bytes, all of them zeroed by prepare_work_queue (no slack)
constexpr float kRescaleTauBf16 = 8.0f;           // lazy-rescale slack of the x64 kernel, log2 units (HISTORY.md section 3.1)
}  // namespace

extern "C" {

int la_abi_version(void) { return LA_ABI_VERSION; }

#ifndef LA_BUILD_INFO          // liteattention_amd/build.py passes the record; a hand-run hipcc gets an honest "unknown"
#define LA_BUILD_INFO "src=unknown;variant=1;wrong_results=0;opts=built outside liteattention_amd/build.py"
#endif
#define LA_STR2(x) #x
#define LA_STR(x) LA_STR2(x)
const char* la_build_info(void) { return "abi=" LA_STR(LA_ABI_VERSION) ";" LA_BUILD_INFO; }

int la_last_hip_error(void) { return g_last_hip_error; }

const char* la_status_string(int status) {
    switch (status) {
        case LA_OK: return "ok";
        case LA_ERR_NULL_ARG: return "required pointer is NULL";
        case LA_ERR_STRUCT_SIZE: return "la_fwd_args.struct_size mismatch (ABI version skew)";
        case LA_ERR_DTYPE: return "FlashAttention only supports fp16, bf16, and fp8_e4m3 type; this build instantiates all three";
        case LA_ERR_HEAD_DIM: return "head_size not instantiated in this build (bf16 / fp16: 64, 96, 128, 192, 256 - under LA_FLAG_KERNEL_128ROW only 64, 128, 256; fp8: 64, 96, 128, 192, 256)";
        case LA_ERR_SHAPE: return "invalid shape (batch, seqlen_q, heads and head_dim must be positive; number of heads in key/value must divide number of heads in query)";
        case LA_ERR_STRIDE: return "Input tensor must have contiguous last dimension and 16-byte aligned rows";
        case LA_ERR_TILE_MISMATCH: return "block_m/block_n do not match la_get_tile_sizes(): skip lists would be mis-indexed";
        case LA_ERR_LISTS: return "attn_read_list and attn_write_list must be given together";
        case LA_ERR_UNSUPPORTED: return "feature outside the QK-Skip hot path (head_dim_v != head_dim, unknown flags, skip lists with cu_seqlens on the 128-row kernels or, for bf16 / fp16, above head_dim 128)";
        case LA_ERR_LAUNCH: return "HIP kernel launch failed (see la_last_hip_error)";
        case LA_ERR_SEQLEN: return "seqlen_k too long: the expanded skip list does not fit in LDS (dense launches are cut into runs and merged when the workspace of la_fwd_workspace_bytes() is given)";
        case LA_ERR_WORKSPACE: return "fp8 needs a 16-byte aligned workspace of la_fwd_workspace_bytes() bytes";
        case LA_ERR_Q_WINDOW: return "q_tile_begin/q_tile_count outside the q-tiles of this problem (LA_FLAG_HALF_VOTE: windows start on an even q-tile and hold an even number unless they reach the last)";
        default: return "unknown la_status";
    }
}

int la_get_tile_sizes_ex(int head_dim, int element_size, uint32_t flags, int* block_m, int* block_n) {
    la::TileShape t = la::tile_shape(head_dim, element_size);
    if (t.block_m == 0) return (element_size == 2 || element_size == 1) ? LA_ERR_HEAD_DIM : LA_ERR_DTYPE;
    if ((flags & ~kKnownFlags) != 0) return LA_ERR_UNSUPPORTED;
    if ((flags & LA_FLAG_KERNEL_128ROW) && element_size == 1) return LA_ERR_UNSUPPORTED;   // the 128-row fp8 kernel is not in this build
    if ((flags & LA_FLAG_KERNEL_128ROW) && (head_dim == 96 || head_dim == 192)) return LA_ERR_HEAD_DIM;   // the hipcc-scheduled template has 64 / 128 / 256
    if (uses_128row(head_dim, element_size, flags)) t.block_m = 128;                       // A/B kernel: 32 rows per wave
    if (uses_half_vote(head_dim, element_size, flags)) t.block_m = 128;                    // lists per 128-row half of the 256-row workgroup
    if (block_m) *block_m = t.block_m;
    if (block_n) *block_n = t.block_n;
    return LA_OK;
}

int la_get_tile_sizes(int head_dim, int element_size, int* block_m, int* block_n) {
    return la_get_tile_sizes_ex(head_dim, element_size, 0u, block_m, block_n);
}

int64_t la_fwd_workspace_bytes(const la_fwd_args* a) {
    if (a == nullptr) return LA_ERR_NULL_ARG;
    if (a->struct_size != sizeof(la_fwd_args)) return LA_ERR_STRUCT_SIZE;
    if (a->dtype == LA_DTYPE_BF16 || a->dtype == LA_DTYPE_FP16) {    // ticket counters of the dynamic work distribution: launches with lists, and
        // (round 5) dense launches of the hand-scheduled kernels; the 128-row template keeps the static map for dense
        int64_t need = ((a->read_list != nullptr || !(a->flags & LA_FLAG_KERNEL_128ROW)) && !(a->flags & LA_FLAG_STATIC_SCHED))
                           ? static_cast<int64_t>(kSchedWorkspaceBytes) : 0;
        // (round 6) a dense key range longer than one launch's walk: + the partial O / LSE of its runs (see dense_split)
        if (a->read_list == nullptr && !(a->flags & LA_FLAG_KERNEL_128ROW) && a->cu_seqlens_q == nullptr && a->seqlen_k > 0 && a->batch > 0 &&
            a->seqlen_q > 0 && a->num_heads > 0 && la::tile_shape(a->head_dim, 2).block_m != 0) {
            const int k_tiles = (a->seqlen_k + 63) / 64;
            if (la::fwd_lds_bytes_x64(k_tiles, nullptr, a->head_dim) > 160 * 1024) {
                const DenseSplit d = dense_split(k_tiles, a->head_dim, a->batch, a->seqlen_q, a->num_heads, a->head_dim_v);
                need = static_cast<int64_t>(kSchedWorkspaceBytes + d.n * (d.o_bytes + d.lse_bytes));
            }
        }
        return need;
    }
    if (a->dtype != LA_DTYPE_FP8_E4M3) return LA_ERR_DTYPE;
    int bm = 0, bn = 0;
    const int trc = la_get_tile_sizes_ex(a->head_dim, 1, a->flags, &bm, &bn);
    if (trc != LA_OK) return trc;
    if (a->batch <= 0 || a->num_heads <= 0 || a->num_heads_k <= 0 || a->seqlen_k < 0) return LA_ERR_SHAPE;
    const int k_tiles = (a->seqlen_k + bn - 1) / bn;
    if (a->read_list == nullptr && a->cu_seqlens_q == nullptr && a->seqlen_q > 0 && walk_lds_bytes(k_tiles, a->head_dim, 1) > 160 * 1024) {
        // a dense key range longer than one launch's walk (see dense_split): the V^T tiles of ONE run (the runs follow each other on the
        // stream), the ticket counters, the partial O / LSE of the runs
        const DenseSplit d = dense_split(k_tiles, a->head_dim, a->batch, a->seqlen_q, a->num_heads, a->head_dim_v, 1);
        return static_cast<int64_t>(la::fp8_workspace_bytes(a->batch, a->num_heads_k, d.chunk_tiles, a->head_dim) + kSchedWorkspaceBytes +
                                    d.n * (d.o_bytes + d.lse_bytes));
    }
    // V^T tiles, then the ticket counter of the dynamic work distribution
    return static_cast<int64_t>(la::fp8_workspace_bytes(a->batch, a->num_heads_k, k_tiles, a->head_dim) + kSchedWorkspaceBytes);
}

int la_fwd(const la_fwd_args* a, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (a == nullptr) return LA_ERR_NULL_ARG;
    if (a->struct_size != sizeof(la_fwd_args)) return LA_ERR_STRUCT_SIZE;
    if (a->dtype != LA_DTYPE_BF16 && a->dtype != LA_DTYPE_FP16 && a->dtype != LA_DTYPE_FP8_E4M3) return LA_ERR_DTYPE;   // flash_api.cpp:715
    const bool fp8 = a->dtype == LA_DTYPE_FP8_E4M3;
    const bool f16 = a->dtype == LA_DTYPE_FP16;      // same kernels as bf16 with the fp16 MFMA and conversions
    const int esize = fp8 ? 1 : 2;
    if (!a->q || !a->o || ((!a->k || !a->v) && a->seqlen_k != 0)) return LA_ERR_NULL_ARG;   // empty K/V tensors may be NULL
    if (a->batch <= 0 || a->seqlen_q <= 0 || a->seqlen_k < 0 || a->num_heads <= 0 || a->num_heads_k <= 0 ||
        a->head_dim <= 0 || a->head_dim_v <= 0)
        return LA_ERR_SHAPE;                                                             // flash_api.cpp:776-778
    if (a->num_heads % a->num_heads_k != 0) return LA_ERR_SHAPE;                          // flash_api.cpp:777
    if (a->head_dim % (fp8 ? 16 : 8) != 0) return LA_ERR_HEAD_DIM;                         // flash_api.cpp:854-856
    if (a->head_dim_v != a->head_dim) return LA_ERR_UNSUPPORTED;
    if (a->reserved0 != 0 || (a->flags & ~kKnownFlags) != 0) return LA_ERR_UNSUPPORTED;
    int bm = 0, bn = 0;
    const int trc = la_get_tile_sizes_ex(a->head_dim, esize, a->flags, &bm, &bn);
    if (trc != LA_OK) return trc;
    if (a->block_m != bm || a->block_n != bn) return LA_ERR_TILE_MISMATCH;
    if ((a->read_list == nullptr) != (a->write_list == nullptr)) return LA_ERR_LISTS;
    // 16-byte vector access on every row: row/head/batch strides multiples of 8 elements, base aligned
    const int64_t strides[] = {a->q_batch_stride, a->q_row_stride, a->q_head_stride, a->k_batch_stride,
                               a->k_row_stride,   a->k_head_stride, a->v_batch_stride, a->v_row_stride,
                               a->v_head_stride,  a->o_batch_stride, a->o_row_stride,  a->o_head_stride};
    for (int i = 0; i < 12; ++i) {                                                       // flash_api.cpp:726-728 (+alignment)
        const int64_t s = strides[i];
        const int gran = (fp8 && i < 9) ? 16 : 8;                                        // 16-byte rows: 16 fp8 / 8 bf16 elements
        if (s % gran != 0 || s < 0) return LA_ERR_STRIDE;
    }
    if (a->k_row_stride > 0x3fffffff || a->v_row_stride > 0x3fffffff || a->o_row_stride > 0x3fffffff || a->q_row_stride > 0x3fffffff)
        return LA_ERR_STRIDE;                                                             // byte strides kept in 32 bits
    if (!aligned16(a->q) || !aligned16(a->k) || !aligned16(a->v) || !aligned16(a->o)) return LA_ERR_STRIDE;

    const bool varlen = a->cu_seqlens_q != nullptr || a->cu_seqlens_k != nullptr;
    if (varlen) {                                                                        // flash_api.cpp:736-760
        if (a->cu_seqlens_q == nullptr || a->cu_seqlens_k == nullptr) return LA_ERR_NULL_ARG;
        if (a->read_list != nullptr && ((a->flags & LA_FLAG_KERNEL_128ROW) || (a->head_dim > 128 && !fp8)))
            return LA_ERR_UNSUPPORTED;                                                    // lists + cu_seqlens: the hand-scheduled kernels, head_dim <= 128 (e4m3: every head dim)
        if (a->total_q < 0 || a->q_tile_count != 0) return LA_ERR_SHAPE;
    }
    if (a->seqlen_k == 0) {
        // flash_api.cpp:1241-1245: no keys -> out = 0, lse = +inf (the write list, if any, is left as it is: nothing was
        // walked). Varlen reaches here only when EVERY sequence is empty (seqlen_k is the maximum); rows then are total_q.
        const hipError_t e0 = varlen
            ? la::launch_empty_k_fill(static_cast<uint16_t*>(a->o), nullptr, 0, a->o_row_stride, a->o_head_stride, 1,
                                      static_cast<int>(a->total_q), a->num_heads, a->head_dim_v, stream)
            : la::launch_empty_k_fill(static_cast<uint16_t*>(a->o), a->lse, a->o_batch_stride, a->o_row_stride, a->o_head_stride,
                                      a->batch, a->seqlen_q, a->num_heads, a->head_dim_v, stream);
        hipError_t e1 = hipSuccess;
        if (e0 == hipSuccess && varlen && a->lse != nullptr && a->total_q > 0)            // +inf = 0x7f800000 is not a memset byte pattern
            e1 = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(a->lse), 0x7f800000, static_cast<size_t>(a->total_q) * a->num_heads, stream);
        if (e0 != hipSuccess || e1 != hipSuccess) { g_last_hip_error = static_cast<int>(e0 != hipSuccess ? e0 : e1); return LA_ERR_LAUNCH; }
        return LA_OK;
    }

    la::FwdParams p{};
    size_t f8_tiles = 0;                               // fp8: bytes of prepared V^T tiles at the front of the workspace
    bool f8_split = false;                             // fp8: a dense key range longer than one launch's walk, cut into runs (dense_split)
    DenseSplit f8d{};
    if (fp8) {
        const int kt = (a->seqlen_k + bn - 1) / bn;
        f8_split = a->read_list == nullptr && !varlen && walk_lds_bytes(kt, a->head_dim, 1) > 160 * 1024;
        if (f8_split) f8d = dense_split(kt, a->head_dim, a->batch, a->seqlen_q, a->num_heads, a->head_dim_v, 1);
        f8_tiles = la::fp8_workspace_bytes(a->batch, a->num_heads_k, f8_split ? f8d.chunk_tiles : kt, a->head_dim);   // split: the tiles of ONE run
        const uint64_t need = f8_tiles + kSchedWorkspaceBytes + (f8_split ? f8d.n * (f8d.o_bytes + f8d.lse_bytes) : 0);
        if (a->workspace == nullptr || a->workspace_bytes < need || !aligned16(a->workspace))
            return LA_ERR_WORKSPACE;
        if (!(a->flags & LA_FLAG_STATIC_SCHED))        // lists or dense: persistent workgroups + ticket queues
            p.work_counter = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(a->workspace) + f8_tiles);
    }
    p.q = static_cast<const uint16_t*>(a->q);
    p.k = static_cast<const uint16_t*>(a->k);
    p.v = static_cast<const uint16_t*>(a->v);
    p.o = static_cast<uint16_t*>(a->o);
    p.lse = a->lse;
    p.q_batch_stride = a->q_batch_stride; p.q_row_stride = a->q_row_stride; p.q_head_stride = a->q_head_stride;
    p.k_batch_stride = a->k_batch_stride; p.k_row_stride = a->k_row_stride; p.k_head_stride = a->k_head_stride;
    p.v_batch_stride = a->v_batch_stride; p.v_row_stride = a->v_row_stride; p.v_head_stride = a->v_head_stride;
    p.o_batch_stride = a->o_batch_stride; p.o_row_stride = a->o_row_stride; p.o_head_stride = a->o_head_stride;
    p.batch = a->batch; p.seqlen_q = a->seqlen_q; p.seqlen_k = a->seqlen_k; p.num_heads = a->num_heads;
    p.h_ratio = a->num_heads / a->num_heads_k;
    p.q_tiles = (a->seqlen_q + bm - 1) / bm;
    p.list_q_tiles = p.q_tiles;
    p.k_tiles = (a->seqlen_k + bn - 1) / bn;
    const bool half_vote = uses_half_vote(a->head_dim, esize, a->flags);
    if (a->q_tile_count == 0) {
        if (a->q_tile_begin != 0) return LA_ERR_Q_WINDOW;
        p.q_tile_begin = 0; p.q_tile_count = p.q_tiles;
    } else {
        if (a->q_tile_begin < 0 || a->q_tile_count < 0 || a->q_tile_begin > p.q_tiles - a->q_tile_count) return LA_ERR_Q_WINDOW;
        // half-vote: q-tiles (list rows) are 128-row halves, the kernel's items are pairs of them: a window starts on an even q-tile
        // and holds an even number of them unless it reaches the last one
        if (half_vote && ((a->q_tile_begin & 1) || ((a->q_tile_count & 1) && a->q_tile_begin + a->q_tile_count != p.q_tiles)))
            return LA_ERR_Q_WINDOW;
        p.q_tile_begin = a->q_tile_begin; p.q_tile_count = a->q_tile_count;
    }
    if (half_vote) {                                   // the kernel's geometry: items of 2 * bm rows; the lists keep bm-row rows
        p.half_vote = 1;
        p.q_tiles = (p.list_q_tiles + 1) / 2;
        p.q_tile_count = (p.q_tile_begin + p.q_tile_count + 1) / 2 - p.q_tile_begin / 2;
        p.q_tile_begin = p.q_tile_begin / 2;
    }
    p.scale_log2 = static_cast<float>(static_cast<double>(a->softmax_scale) * 1.4426950408889634);  // flash_api.cpp:125-126
    p.thr = a->thr;                                                                      // flash_api.cpp:930
    p.rescale_tau = (a->flags & LA_FLAG_EXACT_RESCALE) ? 0.0f : kRescaleTauBf16;
    p.cu_seqlens_q = a->cu_seqlens_q; p.cu_seqlens_k = a->cu_seqlens_k; p.total_q = a->total_q;
    p.read_list = a->read_list;
    p.write_list = a->write_list;
    p.must_do_list = a->must_do_list;
    p.must_do_is_1d = a->must_do_is_1d;
    p.q_descale = a->q_descale; p.k_descale = a->k_descale; p.v_descale = a->v_descale;
    p.q_descale_batch_stride = a->q_descale_batch_stride; p.q_descale_head_stride = a->q_descale_head_stride;
    p.k_descale_batch_stride = a->k_descale_batch_stride; p.k_descale_head_stride = a->k_descale_head_stride;
    p.v_descale_batch_stride = a->v_descale_batch_stride; p.v_descale_head_stride = a->v_descale_head_stride;

    if (!fp8 && (a->flags & LA_FLAG_KERNEL_128ROW) &&
        la::fwd_lds_bytes_v2(a->head_dim <= 128 ? a->head_dim : 256, p.k_tiles, nullptr) > 160 * 1024) return LA_ERR_SEQLEN;
    if (static_cast<int64_t>(p.batch) * p.num_heads * p.q_tiles > 0x7fffffffLL) return LA_ERR_SHAPE;

    if (fp8) {
        const int p_mode = (a->flags & LA_FLAG_FP8_ENCODED_P) ? 0 : (a->flags & LA_FLAG_FP8_MFMA_ROWSUM) ? 1 : 2;   // default: the reference's arithmetic
        if (walk_lds_bytes(p.k_tiles, a->head_dim, 1) > 160 * 1024) {
            // lists keep the bound; a dense launch is cut into runs of tiles that fit, as for bf16 / fp16 below: per run the V^T prepare pass of
            // ITS keys into the one tile region (the runs follow each other on the stream), the forward on bf16 partial O / LSE, then the merge
            if (!f8_split || p.q_tile_count != p.q_tiles || (a->flags & LA_FLAG_V_PREPARED)) return LA_ERR_SEQLEN;
            unsigned char* const ws = static_cast<unsigned char*>(a->workspace);
            uint16_t* const o_part = reinterpret_cast<uint16_t*>(ws + f8_tiles + kSchedWorkspaceBytes);
            float* const lse_part = reinterpret_cast<float*>(ws + f8_tiles + kSchedWorkspaceBytes + f8d.n * f8d.o_bytes);
            hipError_t e = hipSuccess;
            for (int s = 0; s < f8d.n && e == hipSuccess; ++s) {
                la::FwdParams ps = p;
                const int64_t row0 = static_cast<int64_t>(s) * f8d.chunk_tiles * bn;
                const int64_t left = a->seqlen_k - row0, full = static_cast<int64_t>(f8d.chunk_tiles) * bn;
                ps.seqlen_k = static_cast<int>(left < full ? left : full);
                ps.k_tiles = (ps.seqlen_k + bn - 1) / bn;
                ps.k = reinterpret_cast<const uint16_t*>(static_cast<const unsigned char*>(a->k) + row0 * a->k_row_stride);     // 1-byte elements: strides are bytes
                ps.v = static_cast<const uint16_t*>(a->workspace);
                ps.o = o_part + s * (f8d.o_bytes / 2);
                ps.o_head_stride = a->head_dim_v; ps.o_row_stride = static_cast<int64_t>(p.num_heads) * a->head_dim_v;
                ps.o_batch_stride = static_cast<int64_t>(p.seqlen_q) * ps.o_row_stride;
                ps.lse = lse_part + s * (f8d.lse_bytes / 4);
                e = la::launch_prep_v_fp8(static_cast<const unsigned char*>(a->v) + row0 * a->v_row_stride, a->v_batch_stride, a->v_row_stride, a->v_head_stride,
                                          a->workspace, a->batch, ps.seqlen_k, a->num_heads_k, ps.k_tiles, a->head_dim, stream, nullptr);
                if (e == hipSuccess) e = la::launch_fwd_x64_fp8(ps, false, p_mode, a->head_dim, stream);
            }
            if (e == hipSuccess)
                e = la::launch_combine(o_part, true, false, lse_part, p.o, a->lse, f8d.n, p.batch, p.seqlen_q, p.num_heads, a->head_dim_v, stream, false,
                                       p.o_batch_stride, p.o_row_stride, p.o_head_stride);
            if (e != hipSuccess) { g_last_hip_error = static_cast<int>(e); return LA_ERR_LAUNCH; }
            return LA_OK;
        }
        // 1) V -> pre-transposed, pre-swizzled V^T tiles in the caller's workspace; 2) forward on (Q, K, V^T)
        hipError_t e8 = hipSuccess;
        if (!(a->flags & LA_FLAG_V_PREPARED))
            e8 = la::launch_prep_v_fp8(a->v, a->v_batch_stride, a->v_row_stride, a->v_head_stride, a->workspace,
                                       a->batch, a->seqlen_k, a->num_heads_k, p.k_tiles, a->head_dim, stream, a->cu_seqlens_k);
        if (e8 == hipSuccess) {
            p.v = static_cast<const uint16_t*>(a->workspace);
            e8 = la::launch_fwd_x64_fp8(p, a->read_list != nullptr, p_mode, a->head_dim, stream);
        }
        if (e8 != hipSuccess) { g_last_hip_error = static_cast<int>(e8); return LA_ERR_LAUNCH; }
        return LA_OK;
    }
    // every instantiated head_dim: the hand-scheduled x64 kernel unless LA_FLAG_KERNEL_128ROW (the hipcc-scheduled template: 64 / 128 / 256)
    const bool skipable = a->read_list != nullptr;                                      // is_skipable, flash_api.cpp:931
    const bool x64 = !(a->flags & LA_FLAG_KERNEL_128ROW);
    hipError_t err;
    // optional workspace (la_fwd_workspace_bytes): with it, the launch uses persistent workgroups and the ticket queues (lists:
    // every kernel; dense: the hand-scheduled ones); without it, the static one-workgroup-per-item map (same results either way)
    if ((skipable || x64) && a->workspace != nullptr && a->workspace_bytes >= kSchedWorkspaceBytes && aligned16(a->workspace) &&
        !(a->flags & LA_FLAG_STATIC_SCHED))
        p.work_counter = static_cast<unsigned*>(a->workspace);
    if (x64 && la::fwd_lds_bytes_x64(p.k_tiles, nullptr, a->head_dim, nullptr, half_vote && skipable) > 160 * 1024) {
        // lists keep the bound; a dense launch is cut into runs of tiles that fit (dense_split) when the caller gave the workspace for it
        if (skipable || varlen || p.q_tile_count != p.q_tiles) return LA_ERR_SEQLEN;
        const DenseSplit d = dense_split(p.k_tiles, a->head_dim, p.batch, p.seqlen_q, p.num_heads, a->head_dim_v);
        if (a->workspace == nullptr || !aligned16(a->workspace) || a->workspace_bytes < kSchedWorkspaceBytes + d.n * (d.o_bytes + d.lse_bytes))
            return LA_ERR_SEQLEN;
        unsigned char* const ws = static_cast<unsigned char*>(a->workspace);
        uint16_t* const o_part = reinterpret_cast<uint16_t*>(ws + kSchedWorkspaceBytes);
        float* const lse_part = reinterpret_cast<float*>(ws + kSchedWorkspaceBytes + d.n * d.o_bytes);
        for (int s = 0; s < d.n; ++s) {
            la::FwdParams ps = p;
            const int64_t row0 = static_cast<int64_t>(s) * d.chunk_tiles * bn;
            ps.k = p.k + row0 * p.k_row_stride;
            ps.v = p.v + row0 * p.v_row_stride;
            ps.seqlen_k = static_cast<int>(a->seqlen_k - row0 < static_cast<int64_t>(d.chunk_tiles) * bn ? a->seqlen_k - row0 : static_cast<int64_t>(d.chunk_tiles) * bn);
            ps.k_tiles = (ps.seqlen_k + bn - 1) / bn;
            ps.o = o_part + s * (d.o_bytes / 2);
            ps.o_head_stride = a->head_dim_v; ps.o_row_stride = static_cast<int64_t>(p.num_heads) * a->head_dim_v;
            ps.o_batch_stride = static_cast<int64_t>(p.seqlen_q) * ps.o_row_stride;
            ps.lse = lse_part + s * (d.lse_bytes / 4);
            err = la::launch_fwd_x64(ps, a->head_dim, false, f16, stream);
            if (err != hipSuccess) { g_last_hip_error = static_cast<int>(err); return LA_ERR_LAUNCH; }
        }
        err = la::launch_combine(o_part, true, f16, lse_part, p.o, a->lse, d.n, p.batch, p.seqlen_q, p.num_heads, a->head_dim_v, stream, false,
                                 p.o_batch_stride, p.o_row_stride, p.o_head_stride);
    } else if (x64) {
        err = la::launch_fwd_x64(p, a->head_dim, skipable, f16, stream);
    } else {
        err = la::launch_fwd_bf16_v2(p, a->head_dim, skipable, f16, stream);
    }
    if (err != hipSuccess) {
        g_last_hip_error = static_cast<int>(err);
        return LA_ERR_LAUNCH;
    }
    return LA_OK;
}

int la_skip_list_stats(const int32_t* list, int32_t n_batch, int32_t num_heads, int32_t q_tiles, int32_t k_tiles,
                       int64_t* out_counts, void* stream_) {
    if (!list || !out_counts) return LA_ERR_NULL_ARG;
    if (n_batch <= 0 || num_heads <= 0 || q_tiles <= 0 || k_tiles <= 0) return LA_ERR_SHAPE;
    const int64_t rows = static_cast<int64_t>(n_batch) * num_heads * q_tiles;
    if (rows > 0x7fffffffLL) return LA_ERR_SHAPE;
    const hipError_t err = la::launch_skip_list_stats(list, static_cast<int>(rows), k_tiles, out_counts,
                                                      static_cast<hipStream_t>(stream_));
    if (err != hipSuccess) { g_last_hip_error = static_cast<int>(err); return LA_ERR_LAUNCH; }
    return LA_OK;
}

int la_blockmask_to_lists(const uint8_t* blockmask, int64_t mask_batch_stride, int64_t mask_head_stride, int32_t batch,
                          int32_t num_heads, int32_t q_tiles, int32_t k_tiles, const int32_t* q_tiles_valid,
                          const int32_t* k_tiles_valid, int32_t* lists, int32_t* empty_rows, void* stream_) {
    if (!blockmask || !lists) return LA_ERR_NULL_ARG;
    if (batch <= 0 || num_heads <= 0 || q_tiles <= 0 || k_tiles <= 0) return LA_ERR_SHAPE;
    if (mask_batch_stride < 0 || mask_head_stride < 0) return LA_ERR_STRIDE;
    if (static_cast<int64_t>(batch) * num_heads * q_tiles > 0x7fffffffLL) return LA_ERR_SHAPE;
    const hipError_t err = la::launch_blockmask_to_lists(blockmask, mask_batch_stride, mask_head_stride, batch, num_heads, q_tiles,
                                                         k_tiles, q_tiles_valid, k_tiles_valid, lists, empty_rows,
                                                         static_cast<hipStream_t>(stream_));
    if (err != hipSuccess) { g_last_hip_error = static_cast<int>(err); return LA_ERR_LAUNCH; }
    return LA_OK;
}

int la_device_slots(int head_dim, int element_size, uint32_t flags, int* compute_units, int* workgroups_per_cu) {
    int bm = 0, bn = 0;
    const int rc = la_get_tile_sizes_ex(head_dim, element_size, flags, &bm, &bn);
    if (rc != LA_OK) return rc;
    // the hand-scheduled kernels fill a CU with ONE workgroup (512 registers x 4 waves, 64-130 KiB of LDS; the two-waves-per-SIMD A/B body
    // of head_dim 64 is one 8-wave workgroup too); the hipcc-scheduled 128-row template runs two per CU at head_dim <= 128 (la_fwd_kernel_v2.hip)
    const bool v2 = element_size == 2 && (flags & LA_FLAG_KERNEL_128ROW) != 0;
    if (compute_units) *compute_units = la::compute_units();
    if (workgroups_per_cu) *workgroups_per_cu = v2 ? (head_dim <= 128 ? 2 : 1) : (element_size == 2 ? la::x64_workgroups_per_cu(head_dim) : 1);
    return LA_OK;
}

int la_combine(const void* o_partial, int32_t partial_is_16bit, const float* lse_partial, void* o, int32_t o_dtype, float* lse,
               int32_t num_splits, int32_t batch, int32_t seqlen_q, int32_t num_heads, int32_t head_dim_v,
               void* stream_) {
    if (!o_partial || !lse_partial || !o) return LA_ERR_NULL_ARG;
    if (o_dtype != LA_DTYPE_BF16 && o_dtype != LA_DTYPE_FP16 && o_dtype != LA_DTYPE_FP32) return LA_ERR_DTYPE;
    if (o_dtype == LA_DTYPE_FP32 && partial_is_16bit) return LA_ERR_DTYPE;      // an fp32 result is the merge of fp32 partials
    if (num_splits <= 0 || batch <= 0 || seqlen_q <= 0 || num_heads <= 0 || head_dim_v <= 0) return LA_ERR_SHAPE;
    if (head_dim_v % 8 != 0) return LA_ERR_HEAD_DIM;
    if (!aligned16(o_partial) || !aligned16(o)) return LA_ERR_STRIDE;
    const hipError_t err = la::launch_combine(o_partial, partial_is_16bit != 0, o_dtype == LA_DTYPE_FP16, lse_partial, static_cast<uint16_t*>(o), lse,
                                              num_splits, batch, seqlen_q, num_heads, head_dim_v,
                                              static_cast<hipStream_t>(stream_), o_dtype == LA_DTYPE_FP32);
    if (err != hipSuccess) { g_last_hip_error = static_cast<int>(err); return LA_ERR_LAUNCH; }
    return LA_OK;
}

int la_combine_list(const void* const* o_partials, int32_t partial_is_16bit, const float* const* lse_partials, void* o, int32_t o_dtype,
                    float* lse, int32_t num_splits, int32_t batch, int32_t seqlen_q, int32_t num_heads, int32_t head_dim_v, void* stream_) {
    if (!o_partials || !lse_partials || !o) return LA_ERR_NULL_ARG;
    if (o_dtype != LA_DTYPE_BF16 && o_dtype != LA_DTYPE_FP16 && o_dtype != LA_DTYPE_FP32) return LA_ERR_DTYPE;
    if (o_dtype == LA_DTYPE_FP32 && partial_is_16bit) return LA_ERR_DTYPE;
    if (num_splits <= 0 || num_splits > LA_COMBINE_LIST_MAX || batch <= 0 || seqlen_q <= 0 || num_heads <= 0 || head_dim_v <= 0) return LA_ERR_SHAPE;
    if (head_dim_v % 8 != 0) return LA_ERR_HEAD_DIM;
    if (!aligned16(o)) return LA_ERR_STRIDE;
    for (int i = 0; i < num_splits; ++i) {
        if (!o_partials[i] || !lse_partials[i]) return LA_ERR_NULL_ARG;
        if (!aligned16(o_partials[i])) return LA_ERR_STRIDE;
    }
    const hipError_t err = la::launch_combine_list(o_partials, partial_is_16bit != 0, o_dtype == LA_DTYPE_FP16, lse_partials, static_cast<uint16_t*>(o),
                                                   lse, num_splits, batch, seqlen_q, num_heads, head_dim_v, static_cast<hipStream_t>(stream_),
                                                   o_dtype == LA_DTYPE_FP32);
    if (err != hipSuccess) { g_last_hip_error = static_cast<int>(err); return LA_ERR_LAUNCH; }
    return LA_OK;
}

}  // extern "C"
