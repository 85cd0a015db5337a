/*
 * lite_attention_amd.h — C-ABI of the MI355X (gfx950) QK-Skip attention forward.
 *
 * This is the drop-in boundary of the hot path. It replaces, one for one, what the
 * reference reaches through `torch.ops.lite_attention.fwd`:
 *
 *   la_fwd                <- mha_fwd                hopper/_internal/cpp/flash_api.cpp:667-1249
 *                            (+ set_params_fprop    flash_api.cpp:45-163,
 *                               run_mha_fwd         flash_api.cpp:362-380,
 *                               run_flash_fwd       hopper/_internal/cpp/flash_fwd_launch_template.h:52-363)
 *   la_fwd_args           <- Flash_fwd_params + QKSkipMaskArgs
 *                                                   hopper/_internal/cpp/flash.h:12-18,48-185
 *   la_get_tile_sizes     <- tile_size_fwd_sm90     hopper/_internal/cpp/tile_size.h:10-62
 *                            (and its Python twin LiteAttention.get_MN, hopper/lite_attention.py:87-111)
 *   la_skip_list_stats    <- LiteAttention.calc_percentage   hopper/lite_attention.py:61-85
 *                            (device-side, corrected statistic)
 *   la_blockmask_to_lists <- convert_blockmask + the (missing) fwd_block entry point of the block-sparse adapter
 *                            flash_attn/flash_blocksparse_attn_interface.py:7-39,185-200: a static 0/1 block mask becomes
 *                            the read lists la_fwd walks
 *   la_combine            <- mha_combine / flash_fwd_combine (LSE-weighted merge of partial outputs,
 *                            hopper/_internal/cpp/flash_api.cpp fwd_combine; oracle
 *                            hopper/tests/test_flash_attn.py:1178-1187)
 *
 * Rules of the boundary:
 *   - plain C: pointers, sizes, scalars. No torch types, no C++ types, no exceptions.
 *   - every pointer is a DEVICE pointer unless the name says `host`.
 *   - the library owns nothing, allocates nothing on the device and keeps no global mutable
 *     state (no environment variables are read either: every choice is an argument or a flag);
 *     outputs and skip lists are caller-owned. `write_list` is mutated in place.
 *   - launches are asynchronous on the given hipStream_t (passed as void*); no host sync.
 *   - return value: LA_OK (0) or a negative la_status; on error nothing has been launched.
 */
#ifndef LITE_ATTENTION_AMD_H
#define LITE_ATTENTION_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LA_ABI_VERSION 8   /* 8 = 7 with the sense of the two fp8 flags turned round (the reference's arithmetic is the default form of P) + LA_FLAG_HALF_VOTE + la_combine_list;
                            * 7 = 6 + la_build_info; 6 = 5 + la_blockmask_to_lists, la_device_slots (5 = 4 + skip lists and fp8 with cu_seqlens, LA_FLAG_EXACT_ROWSUM /
                            * LA_FLAG_EXACT_EXP (renamed and inverted in 8), LA_DTYPE_FP32 for la_combine); la_fwd_args unchanged since 4 */

typedef enum la_status {
    LA_OK = 0,
    LA_ERR_NULL_ARG = -1,        /* a required pointer is NULL                                   */
    LA_ERR_STRUCT_SIZE = -2,     /* args->struct_size != sizeof(la_fwd_args): ABI mismatch        */
    LA_ERR_DTYPE = -3,           /* dtype not built (flash_api.cpp:715 "only supports fp16, bf16, fp8") */
    LA_ERR_HEAD_DIM = -4,        /* head_dim not instantiated / not a multiple of 8 (flash_api.cpp:854) */
    LA_ERR_SHAPE = -5,           /* non-positive sizes, num_heads % num_heads_k != 0 (flash_api.cpp:777). seqlen_k == 0 is NOT
                                  * an error: la_fwd stores o = 0, lse = +inf like the reference (flash_api.cpp:1241-1245) */
    LA_ERR_STRIDE = -6,          /* last dim must be contiguous (flash_api.cpp:726-728) / 16-byte rows  */
    LA_ERR_TILE_MISMATCH = -7,   /* block_m/block_n echo != la_get_tile_sizes (list indexing would be wrong) */
    LA_ERR_LISTS = -8,           /* read list without write list, or vice versa                   */
    LA_ERR_UNSUPPORTED = -9,     /* feature outside the hot path (causal, dv != d, ...)           */
    LA_ERR_LAUNCH = -10,         /* hipLaunchKernel failed; see la_last_hip_error()               */
    LA_ERR_SEQLEN = -11,         /* sequence too long for the per-workgroup list staging in LDS   */
    LA_ERR_WORKSPACE = -12,      /* fp8: workspace missing or smaller than la_fwd_workspace_bytes() */
    LA_ERR_Q_WINDOW = -13        /* q_tile_begin/q_tile_count outside [0, ceil(seqlen_q/block_m)]  */
} la_status;

/* la_fwd_args.flags */
#define LA_FLAG_KERNEL_128ROW 4u /* A/B: run the hipcc-scheduled 128-row template (32 rows per wave) where the default is the
                                  * hand-scheduled kernel: bf16 / fp16 head_dim 128 (the skip lists then use 128-row q-tiles instead of
                                  * 256-row ones: take the tile sizes from la_get_tile_sizes_ex() with the same flags) and head_dim 256
                                  * (same tiles). head_dim 96 / 192 have no such instantiation: LA_ERR_HEAD_DIM. fp8: LA_ERR_UNSUPPORTED. */
#define LA_FLAG_HALF_VOTE 64u    /* bf16 / fp16 head dims 64 / 96 / 128 - the kernels with a 256-row q-tile (no effect elsewhere; LA_FLAG_KERNEL_128ROW wins when both are set): the hand-scheduled
                                  * kernel keeps its skip lists per 128-ROW HALF of its 256-row workgroup - the reference's own q-granularity for
                                  * this head dim (kBlockM = 128, tile_size.h:35-39). la_get_tile_sizes_ex() reports (128, 64) with it. The
                                  * workgroup walks the UNION of its two halves' lists (never longer than the 256-row list of the same votes), and
                                  * the waves of a half sit out the tiles only the other half lists. Every half's output, LSE and write list are
                                  * exactly those of an independent 128-row q-tile walking its own list. At a given THRESHOLD a 128-row vote drops
                                  * more tiles than a 256-row one (13.7 % / 25 % fewer listed tiles at thr -4.22 / -2.46 on the 50-step workload).
                                  * q-tile windows: q_tile_begin even, q_tile_count even unless the window reaches the last q-tile. Lists are read
                                  * as SETS of tiles (identical to the literal walk for every well-formed list; a list with overlapping or
                                  * ascending ranges walks each named tile once). */
#define LA_FLAG_EXACT_RESCALE 8u /* A/B: the hand-scheduled kernels rescale O on EVERY growth of a row maximum (tau = 0) instead of lazily
                                  * (bf16: only after it grew by more than 2^8; results agree to rounding, lists are identical) */
/* fp8 (e4m3) forms of P. DEFAULT = the reference's arithmetic in full: P = exp2(S c - m c + off) by the transcendental unit, rounded to e4m3 by the
 * hardware convert (softmax.h:85-87 + mainloop...:1645-1647), row sums l = sum of the UN-rounded fp32 P on the vector unit (softmax.h:275-296: LSE
 * exact to fp32). The two flags below trade that for throughput and are for callers who ask for it (round 6: until ABI 7 the encoded form was the
 * default and the reference's arithmetic behind LA_FLAG_EXACT_ROWSUM / LA_FLAG_EXACT_EXP - a drop-in's default must be the reference's results). */
#define LA_FLAG_FP8_MFMA_ROWSUM 16u /* fp8: row sums l~ = sum of the e4m3-ROUNDED P taken from the matrix pipe instead of the fp32 sum of the un-rounded P
                                  * (4-5 % faster; O = (sum P~ V) / (sum P~) is self-consistent, the LSE then carries the rounding of
                                  * P~: |LSE - exact| <= ln(1 + 2^-4), about 1e-2 on rows of a few keys, 1e-4 on long rows).
                                  * Ignored for bf16 / fp16. */
#define LA_FLAG_FP8_ENCODED_P 32u /* fp8: the BLOCK-SCALED LOG-LINEAR ENCODING of P instead of exp2 + hardware rounding (implies the matrix-pipe row sums):
                                  *  - the e4m3 byte of P is computed directly, b = sat_u8(rne(8 y + 56 - 8 delta)), y = log2 P: one FMA + one byte
                                  *    convert per score and no transcendental (the fp8 kernel is bound by vector-unit issue, not by the matrix
                                  *    pipe). Reading byte / 8 - 7 as a base-2 logarithm is the linear-mantissa exponential (1 + f for 2^f; delta =
                                  *    0.0575 centres it): per element P~ / P lies in [0.920, 1.065] (rms 3.1 %) against [0.941, 1.0625] (rms 2.65 %)
                                  *    for round-to-nearest e4m3;
                                  *  - per query row and key tile P is encoded relative to 2^t, t = floor(log2 of the row's largest P in the tile), and
                                  *    the matrix instruction multiplies by 2^t through its E8M0 block-scale operand (MX scaling of P): the 15 octaves
                                  *    of the byte grid hang below the TILE's maximum, so a diffuse tail far below the row maximum is kept with full
                                  *    relative precision (hardware rounding at the reference's offset drops P below 2^-17 of the row maximum), and
                                  *    P cannot overflow, so O is practically never rescaled.
                                  * NOT the reference's arithmetic: on the reference-generated fp8 outputs the error of O is 1.6-2.1 x that of the default
                                  * form (rms), inside the reference's own fp8 rule; |LSE - exact| <= 0.084 (rows of one or two comparable keys), about
                                  * 3e-4 of bias on long rows. +20 ... +25 % throughput at the headline shape over the default (bench.py reports all
                                  * three forms in one line). Ignored for bf16 / fp16. */
#define LA_FLAG_STATIC_SCHED 2u  /* keep one workgroup per item (static XCD map) even when a workspace is given. For
                                  * launches that must share the GPU with another kernel while they run — e.g. an RCCL
                                  * collective on another stream: persistent workgroups would hold every CU until the
                                  * launch ends, per-item workgroups release a CU every item. */
#define LA_FLAG_V_PREPARED 1u    /* fp8: `workspace` already holds the prepared V^T tiles of THIS v (an earlier
                                  * la_fwd call on the same v/workspace): skip the prepare pass. Lets a caller
                                  * split one attention into several q-tile windows (below) and pay for it once. */

typedef enum la_dtype {
    LA_DTYPE_BF16 = 0,
    LA_DTYPE_FP16 = 1,           /* q/k/v/o fp16; every path of LA_DTYPE_BF16 (lists, varlen, flags) */
    LA_DTYPE_FP8_E4M3 = 2,       /* OCP e4m3fn (gfx950), bf16 output                              */
    LA_DTYPE_FP32 = 3            /* la_combine only: fp32 result of fp32 partials                */
} la_dtype;

/*
 * Forward arguments (dense and QK-Skip launches, fixed-length and packed batches, bf16 / fp16 / fp8: every path of la_fwd). Strides are in ELEMENTS (as Flash_fwd_params, flash_api.cpp:84-103).
 * Tensors: q (B,Sq,H,D)  k (B,Sk,Hk,D)  v (B,Sk,Hk,Dv)  o (B,Sq,H,Dv)  lse (B,H,Sq) fp32 contiguous.
 * GQA / MQA: H % Hk == 0, query head h reads K/V head h / (H/Hk) (flash_api.cpp:777; the reference's
 * non-packed GQA path); fp8 descales are per (batch, K/V head) for q, k AND v (flash_api.cpp:689-691).
 *
 * Skip lists (reader/writer semantics of
 * hopper/_internal/cpp/mainloop_fwd_sm90_tma_gmma_ws.hpp:47-192):
 *   int32 [B_alloc, H, Qt, Kt+1] contiguous, Qt = ceil(Sq/block_m), Kt = ceil(Sk/block_n);
 *   row = [L, start_0, end_0, start_1, end_1, ...]; ranges descending, both ends inclusive.
 *   read_list == NULL  -> dense mode (is_skipable = false, flash_api.cpp:931-936).
 *   must_do_list: same row format, token->tile converted by the caller
 *                 (hopper/lite_attention.py:228-235); ranges are (start inclusive, end EXCLUSIVE).
 *                 must_do_is_1d != 0 -> ONE row of Kt+1 ints shared by every (b,h,q-tile)
 *                 (the reference materialises the full 4-D repeat per call, lite_attention.py:239-241).
 */
typedef struct la_fwd_args {
    uint32_t struct_size;        /* = sizeof(la_fwd_args)                                         */
    int32_t  dtype;              /* la_dtype of q,k,v                                             */

    const void* q;
    const void* k;
    const void* v;
    void*       o;               /* bf16 (also for fp8 inputs, flash_api.cpp:859)                 */
    float*      lse;             /* may be NULL: LSE not stored                                   */

    int64_t q_batch_stride, q_row_stride, q_head_stride;
    int64_t k_batch_stride, k_row_stride, k_head_stride;
    int64_t v_batch_stride, v_row_stride, v_head_stride;
    int64_t o_batch_stride, o_row_stride, o_head_stride;

    int32_t batch, seqlen_q, seqlen_k;
    int32_t num_heads, num_heads_k;
    int32_t head_dim, head_dim_v;

    float   softmax_scale;       /* scores = q.k * softmax_scale (flash_api.cpp:125)              */

    /* fp8 only: per-(batch, kv-head) descales, fp32; NULL = 1.0 (flash_api.cpp:1003-1022) */
    const float* q_descale; const float* k_descale; const float* v_descale;
    int64_t q_descale_batch_stride, q_descale_head_stride;
    int64_t k_descale_batch_stride, k_descale_head_stride;
    int64_t v_descale_batch_stride, v_descale_head_stride;

    const int32_t* read_list;    /* QKSkipMaskArgs::attn_read_list   flash.h:13                   */
    int32_t*       write_list;   /* QKSkipMaskArgs::attn_write_list  flash.h:15                   */
    const int32_t* must_do_list; /* QKSkipMaskArgs::attn_must_do_list flash.h:14; may be NULL     */
    int32_t        must_do_is_1d;
    float          thr;          /* QKSkipMaskArgs::thr flash.h:17 (log2 domain)                  */

    int32_t block_m, block_n;    /* echo of la_get_tile_sizes(); checked                          */

    /* Caller-owned scratch (the library allocates nothing). Size from la_fwd_workspace_bytes(); 16-byte aligned;
     * contents are scratch, valid during the call (stream-ordered).
     *   fp8: REQUIRED — the pre-transposed V tiles (64 keys x head_dim bytes per (batch, K/V head, key tile), 96 as 128; head_dim 64, 96, 128, 192, 256).
     *   bf16 / fp16: OPTIONAL, 1 KiB — eight ticket counters (one queue per XCD). With it the launch uses one
     *   persistent workgroup per CU and distributes the (batch, head, q-tile) items dynamically, which removes the
     *   cross-XCD imbalance real skip lists cause (items differ 2-3x in length; the hardware's workgroup->XCD
     *   assignment is static) and, for dense launches too since round 5 (hand-scheduled kernels; the 128-row template
     *   keeps the static map there), the per-item workgroup launch (+3.5 % at S = 75 600, +0.7 % at 16 384).
     *   Without it: one workgroup per item, static map. Results are identical. */
    void*    workspace;
    uint64_t workspace_bytes;

    /* q-tile window (ABI 3). The launch computes q-tiles [q_tile_begin, q_tile_begin + q_tile_count) of every
     * (batch, head) of the SAME problem: q/o/lse/lists are still the full tensors and are indexed by the global
     * q-tile, rows outside the window are not touched. q_tile_count == 0 means "all" (begin must then be 0).
     * Use: several launches on one stream whose outputs leave early (e.g. the head-sharded driver all-gathers
     * window i over xGMI while window i+1 computes). The reference has no equivalent (one launch per call,
     * flash_fwd_launch_template.h:359). */
    int32_t  q_tile_begin, q_tile_count;
    uint32_t flags;              /* LA_FLAG_* */
    uint32_t reserved0;          /* must be 0 */

    /* Variable-length (packed) batches (ABI 4) — the reference's cu_seqlens_q / cu_seqlens_k of mha_fwd
     * (flash_api.cpp:672-674, 736-760; Python: hopper/_internal/flash_attn_interface.py:638-682). Both NULL = fixed length.
     * Both given: q is (total_q, H, D), k/v (total_k, Hk, D), o (total_q, H, D) — the *_batch_stride fields are ignored —
     * cu_seqlens_* are DEVICE int32[batch + 1] prefix sums (sequence b = rows [cu[b], cu[b+1])), seqlen_q / seqlen_k are the
     * MAXIMUM sequence lengths (they size the grid; longer sequences are truncated to them), and lse is (H, total_q):
     * lse[h * total_q + row]. Sequences with no keys get o = 0, lse = +inf. bf16, fp16 and (round 3) fp8: the V^T prepare pass
     * reads cu_seqlens_k itself and lays every sequence's tiles on the [B, Hk, tiles of the longest sequence] grid of `workspace`.
     * One launch for the whole batch, no host sync.
     * Skip lists with cu_seqlens (round 3; the reference's varlen entry point has none): read_list / write_list / a 4-D
     * must_do_list are [>= batch, H, ceil(seqlen_q / block_m), ceil(seqlen_k / block_n) + 1] - the geometry of the MAXIMA - and
     * row (b, h, m) describes q-tile m of sequence b over that sequence's own k-tiles (tile indices relative to the sequence).
     * Rows of q-tiles past a sequence's end, and of sequences without keys, are neither read nor written. Served by the
     * hand-scheduled kernels at head_dim <= 128: with LA_FLAG_KERNEL_128ROW or a larger head_dim it is LA_ERR_UNSUPPORTED. */
    const int32_t* cu_seqlens_q;
    const int32_t* cu_seqlens_k;
    int64_t        total_q;      /* rows of q / o (= cu_seqlens_q[batch]); the head stride of lse */
} la_fwd_args;

/* Tile sizes (kBlockM, kBlockN) of the kernel that la_fwd will run for (head_dim, element size).
 * Skip-list geometry depends on them, so host code must take them from here. */
int la_get_tile_sizes(int head_dim, int element_size, int* block_m, int* block_n);
/* The same for a launch that sets `flags` (LA_FLAG_KERNEL_128ROW and LA_FLAG_HALF_VOTE change the answer). */
int la_get_tile_sizes_ex(int head_dim, int element_size, uint32_t flags, int* block_m, int* block_n);

/* Bytes of `workspace` la_fwd wants for these arguments (fp8: required; bf16 / fp16: optional, see la_fwd_args;
 * 0 under LA_FLAG_STATIC_SCHED and for dense launches of the 128-row template). Negative la_status on bad arguments. */
int64_t la_fwd_workspace_bytes(const la_fwd_args* args);

/* The forward pass. `stream` is a hipStream_t. */
int la_fwd(const la_fwd_args* args, void* stream);

/* Counts listed (= to be computed) tiles of a skip list on the device:
 *   out_counts[0] = sum over rows of sum over ranges (start - end + 1), rows = n_batch*H*Qt
 *   out_counts[1] = number of rows
 * `list` is [>=n_batch, H, Qt, Kt+1]; out_counts is a device int64[2], zeroed by the call. */
int la_skip_list_stats(const int32_t* list, int32_t n_batch, int32_t num_heads, int32_t q_tiles,
                       int32_t k_tiles, int64_t* out_counts, void* stream);

/* LSE-weighted merge of `num_splits` partial attention results (sequence-parallel K/V splits):
 *   o_partial   [num_splits, B, Sq, H, Dv] contiguous: fp32, or (partial_is_16bit != 0) the element type of o
 *   lse_partial fp32 [num_splits, B, H, Sq] contiguous
 *   o [B,Sq,H,Dv] contiguous of o_dtype (LA_DTYPE_BF16, LA_DTYPE_FP16, or LA_DTYPE_FP32 for fp32 partials: the reference's default there,
 *   hopper/_internal/flash_attn_interface.py:684-685), lse fp32 [B,H,Sq] (may be NULL). */
int la_combine(const void* o_partial, int32_t partial_is_16bit, const float* lse_partial,
               void* o, int32_t o_dtype, float* lse, int32_t num_splits, int32_t batch, int32_t seqlen_q,
               int32_t num_heads, int32_t head_dim_v, void* stream);

/* The same merge when the partials of the splits are SEPARATE tensors - what the calls of a split attention return, e.g. the t2t / t2v / v2t / v2v
 * calls of the reference's text + video recipe (README.md:225-246), or a ring step: stacking them into one [num_splits, ...] tensor first would
 * move every partial once more than the merge itself does. o_partials / lse_partials: HOST arrays of num_splits <= LA_COMBINE_LIST_MAX device
 * pointers, each partial [B, Sq, H, Dv] contiguous (fp32, or the element type of o when partial_is_16bit), each LSE fp32 [B, H, Sq] contiguous. */
#define LA_COMBINE_LIST_MAX 8
int la_combine_list(const void* const* o_partials, int32_t partial_is_16bit, const float* const* lse_partials,
                    void* o, int32_t o_dtype, float* lse, int32_t num_splits, int32_t batch, int32_t seqlen_q,
                    int32_t num_heads, int32_t head_dim_v, void* stream);

/* Static block mask -> read-list rows, on the device (one wave per row; no host round trip).
 *   blockmask      uint8 (0 = tile masked out, non-zero = computed), rows [q_tiles, k_tiles] contiguous; element strides between
 *                  batches and heads are arguments (0 = the same mask for every batch / head).
 *   q_tiles_valid, k_tiles_valid   device int32[batch] or NULL: the sequence of batch b has only that many q- / k-tiles (packed batches
 *                  under cu_seqlens use the mask's top-left corner): k-tiles >= k_tiles_valid[b] are dropped, rows of q-tiles >=
 *                  q_tiles_valid[b] (never read by la_fwd) get the whole corner.
 *   lists          int32 [batch, num_heads, q_tiles, k_tiles + 1] contiguous, every element written:
 *                  row = [2 * runs, start_0, end_0, ...] (descending, both ends inclusive), zero padded.
 *   empty_rows     device int32[1] or NULL: number of rows that keep no tile (row[0] = 0). Such a row has no skip-list
 *                  representation - the reader always walks its first range (mainloop_fwd_sm90_tma_gmma_ws.hpp:93-101) - the
 *                  caller decides whether to read the counter back. */
int la_blockmask_to_lists(const uint8_t* blockmask, int64_t mask_batch_stride, int64_t mask_head_stride, int32_t batch,
                          int32_t num_heads, int32_t q_tiles, int32_t k_tiles, const int32_t* q_tiles_valid,
                          const int32_t* k_tiles_valid, int32_t* lists, int32_t* empty_rows, void* stream);

/* How many workgroups of the kernel la_fwd runs for (head_dim, element size, flags) are resident at once on the current device:
 * compute units x workgroups per compute unit. A host that splits one attention into q-tile windows (la_fwd_args.q_tile_begin)
 * sizes them in whole rounds of this number. No counterpart in the reference (one launch per call). */
int la_device_slots(int head_dim, int element_size, uint32_t flags, int* compute_units, int* workgroups_per_cu);

/* What this binary was built from: "abi=7;src=<sha256[:16] of the kernel / API sources, generators and this header>;variant=0|1;
 * wrong_results=0|1;opts=<generator options and -D defines, empty for the product build>". The product build (variant=0) is generated
 * with NO generator option and NO define, whatever the environment of the build held; A/B and pricing builds (python -m
 * liteattention_amd.build --out=...) say variant=1, and wrong_results=1 when an option that changes the arithmetic went in. The Python
 * binding refuses a library in the default location whose record says variant / wrong_results or whose src differs from the tree
 * beside it. No counterpart in the reference (its build flags are only visible in setup.py's environment, hopper/setup.py:47-68). */
const char* la_build_info(void);

const char* la_status_string(int status);
int         la_abi_version(void);
int         la_last_hip_error(void);   /* hipError_t of the last failed launch on this thread */

#ifdef __cplusplus
}
#endif
#endif /* LITE_ATTENTION_AMD_H */
