// la_fwd_common.h — device helpers shared by the forward kernels (types, half-wave exchanges,
// the skip-list reader/writer restatement, the XCD-aware block map, the read-list expansion).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "la_kernel_params.h"

namespace la {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The 16-bit element type of Q / K / V / P / O: bf16 or fp16 (flash_api.cpp:715 accepts both). Only the MFMA opcode and the
// fp32 -> 16-bit conversions (round to nearest even in both) differ; fragment layouts and LDS images are type-agnostic.
template <bool F16> struct Elem16;
template <> struct Elem16<false> {
    typedef bf16x8 x8;
    typedef bf16x4 x4;
    static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Elem16<true> {
    typedef f16x8 x8;
    typedef f16x4 x4;
    static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max over the two half-waves: lanes l and l^32 hold the same query row.
__device__ __forceinline__ float half_swap_max(float x) {
    u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
    u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ------------------------------------------------------------------------------------------------
// Skip-list writer: exact restatement of SkipListWriter (mainloop...:121-192) driven by the
// per-position "do" bits collected during the walk. Runs on ONE lane in the epilogue.
// ------------------------------------------------------------------------------------------------
struct ListReader {
    const int* row;
    int len, idx, start, end;
    __device__ void init(const int* r) {
        row = r; len = r[0]; idx = 1; start = r[1]; end = r[2];
    }
    // the row is width + 1 ints; a pair past its end reads as (0, 0) = "matches nothing", which is what the zero padding of a
    // shorter list gives (the reference reads past the row there, mainloop...:156-159 with len >= k_tiles - 1)
    __device__ void load(int width) {
        const bool in_row = idx + 1 <= width;
        start = in_row ? row[idx] : 0;
        end = in_row ? row[idx + 1] : 0;
    }
    __device__ void advance() { idx += 2; }
    __device__ bool has_more() const { return idx <= len; }
};

// Runs on ONE lane in the epilogue, entirely from LDS: the walked tile sequence `seq`, the range-end markers
// `endflags` (bit p set = position p is the last tile of its read-list range) and the vote bits `doflags` (bit p
// set = some row of the q-tile voted "do" for position p). No global loads of the read list: real lists hold
// hundreds of ranges per row and two dependent global loads per range cost more than the tiles they describe.
__device__ __noinline__ void write_skip_list(const int* seq, const unsigned* endflags, const unsigned* doflags,
                                             int n_tiles, int* __restrict__ write_row,
                                             const int* __restrict__ must_do_row, int k_tiles) {
    ListReader md;
    const bool has_md = must_do_row != nullptr;
    if (has_md) md.init(must_do_row);
    int w = 1;
    bool is_skipping = true;
    for (int pos = 0; pos < n_tiles; ++pos) {
        const int n = seq[pos];
        // fwd_step's raw flag; the first walked tile is recorded with skip = false and no must-do (:1804-1805)
        const bool raw_skip = pos != 0 && !((doflags[pos >> 5] >> (pos & 31)) & 1u);
        bool skip = raw_skip;
        if (has_md && skip) {                                            // record_transition :154-162
            if (md.end > n && md.has_more()) { md.advance(); md.load(k_tiles); }
            const bool must_do = n <= md.start && n > md.end;
            skip = skip && !must_do;
        }
        if (skip != is_skipping) {                                       // :163-168
            if (w <= k_tiles) write_row[w] = n;
            ++w;
            is_skipping = skip;
        }
        if ((endflags[pos >> 5] >> (pos & 31)) & 1u) {                   // record_range_end :173-181 (raw flag)
            is_skipping = true;
            if (!raw_skip) { if (w <= k_tiles) write_row[w] = n; ++w; }
        }
    }
    write_row[0] = min(w - 1, k_tiles);                                  // finalize :185-191
}

// The same over the positions `live` selects (half-vote kernels: bit p set = this list's walk holds position p of the union sequence;
// the other positions belong to the other half's list only and do not exist for this writer; the first LIVE position is the list's
// first walked tile).
__device__ __forceinline__ void write_skip_list_live(const int* seq, const unsigned* endflags, const unsigned* doflags, const unsigned* live,
                                                  int n_tiles, int* __restrict__ write_row,
                                                  const int* __restrict__ must_do_row, int k_tiles) {
    ListReader md;
    const bool has_md = must_do_row != nullptr;
    if (has_md) md.init(must_do_row);
    int w = 1;
    bool is_skipping = true, first = true;
    for (int pos = 0; pos < n_tiles; ++pos) {
        if (!((live[pos >> 5] >> (pos & 31)) & 1u)) continue;
        const int n = seq[pos];
        const bool raw_skip = !first && !((doflags[pos >> 5] >> (pos & 31)) & 1u);
        first = false;
        bool skip = raw_skip;
        if (has_md && skip) {
            if (md.end > n && md.has_more()) { md.advance(); md.load(k_tiles); }
            const bool must_do = n <= md.start && n > md.end;
            skip = skip && !must_do;
        }
        if (skip != is_skipping) {
            if (w <= k_tiles) write_row[w] = n;
            ++w;
            is_skipping = skip;
        }
        if ((endflags[pos >> 5] >> (pos & 31)) & 1u) {
            is_skipping = true;
            if (!raw_skip) { if (w <= k_tiles) write_row[w] = n; ++w; }
        }
    }
    write_row[0] = min(w - 1, k_tiles);
}

// Inclusive prefix sum over the 64 lanes of a wave on the DPP datapath (round 5): four row_shr steps scan each 16-lane row, then
// row_bcast:15 adds lane 15 of rows 0 / 2 to rows 1 / 3 and row_bcast:31 adds lane 31 to rows 2 and 3 - six dependent VALU
// instructions. The __shfl_up form it replaces is six ds_bpermute round trips through the LDS crossbar (~100 cycles each, and the
// list writer / expander are the one place where a lone wave sits on a chain of them while three waves wait).
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);      // row_shr:1, invalid lanes read 0
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3 (other rows keep `old` = 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return v;
}

// Wave-parallel form of write_skip_list (same inputs, run by all 64 lanes of ONE wave). Lane l takes position
// base + l: with skip(p) the must-do-adjusted flag and raw(p) the raw one, the serial writer emits
//     n(p)   if skip(p) != (p is the first position of its range ? true : skip(p-1))       (record_transition)
//     n(p)   if p is the last position of its range and !raw(p)                             (record_range_end)
// in that order; entry slots come from a wave prefix sum. The must-do reader of the reference is stateful (it
// advances at most one range per flagged tile, mainloop...:156-159), but for lists of at most ONE range - the
// default [0,0] and the single-range case - membership reduces to `n <= start && n > end` whatever the reader
// state; longer must-do lists take the serial path, which reproduces the state machine literally.
__device__ __forceinline__ void write_skip_list_wave(const int* seq, const unsigned* endflags, const unsigned* doflags,
                                                     int n_tiles, int* __restrict__ write_row,
                                                     const int* __restrict__ must_do_row, int k_tiles, int lane) {
    int md_start = 0, md_end = 0;
    if (must_do_row != nullptr) {
        if (must_do_row[0] > 2) {                                         // multi-range must-do: literal state machine
            if (lane == 0) write_skip_list(seq, endflags, doflags, n_tiles, write_row, must_do_row, k_tiles);
            return;
        }
        md_start = must_do_row[1];
        md_end = must_do_row[2];
    }
    int w = 1;                  // next free entry slot (wave-uniform)
    int carry_skip = 1;         // "is_skipping" entering the chunk: true at the very start (:125)
    for (int base = 0; base < n_tiles; base += 64) {
        const int pos = base + lane;
        const bool live = pos < n_tiles;
        int n = 0;
        bool raw = false, is_end = false;
        if (live) {
            n = seq[pos];
            raw = pos != 0 && !((doflags[pos >> 5] >> (pos & 31)) & 1u);
            is_end = (endflags[pos >> 5] >> (pos & 31)) & 1u;
        }
        const bool skip = raw && !(n <= md_start && n > md_end);
        // state left behind by this position: forced to "skipping" after a range end. Every quantity here is ONE BIT per lane, so the
        // neighbour's state and the entry slots come from wave ballots and popcounts (round 5; the shuffle-scan form took ~1 100
        // cycles per 64 positions: 12 k cycles per item at the headline's 686 positions, with the other three waves waiting)
        const unsigned long long after_b = __ballot(live && (is_end || skip));
        const unsigned long long below = (1ull << lane) - 1ull;                        // lanes before this one
        const int before = lane == 0 ? carry_skip : static_cast<int>((after_b >> ((lane - 1) & 63)) & 1ull);
        const bool e1 = live && (static_cast<int>(skip) != before);
        const bool e2 = live && is_end && !raw;
        const unsigned long long e1_b = __ballot(e1), e2_b = __ballot(e2);
        int slot = w + __popcll(e1_b & below) + __popcll(e2_b & below);
        if (e1) { if (slot <= k_tiles) write_row[slot] = n; ++slot; }
        if (e2) { if (slot <= k_tiles) write_row[slot] = n; }
        w += __popcll(e1_b) + __popcll(e2_b);
        carry_skip = static_cast<int>((after_b >> 63) & 1ull);
    }
    if (lane == 0) write_row[0] = min(w - 1, k_tiles);
}

// The same for ONE HALF of a half-vote workgroup (LA_FLAG_HALF_VOTE): the walk was the union of two lists, `livebits` (bit p = this
// half's list holds position p) selects this list's positions; the others do not exist for this writer: the state a position sees is
// the one its nearest LIVE predecessor left, and the first live position is the list's first walked tile (never flagged, :1804-1805).
__device__ __forceinline__ void write_skip_list_wave_live(const int* seq, const unsigned* endflags, const unsigned* doflags,
                                                          const unsigned* livebits, int n_tiles, int* __restrict__ write_row,
                                                          const int* __restrict__ must_do_row, int k_tiles, int lane) {
    int md_start = 0, md_end = 0;
    if (must_do_row != nullptr) {
        if (must_do_row[0] > 2) {
            if (lane == 0) write_skip_list_live(seq, endflags, doflags, livebits, n_tiles, write_row, must_do_row, k_tiles);
            return;
        }
        md_start = must_do_row[1];
        md_end = must_do_row[2];
    }
    int w = 1, carry_skip = 1;
    bool seen = false;          // a live position came before this chunk (wave-uniform)
    for (int base = 0; base < n_tiles; base += 64) {
        const int pos = base + lane;
        const bool live = pos < n_tiles && ((livebits[pos >> 5] >> (pos & 31)) & 1u);
        const unsigned long long live_b = __ballot(live);
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned long long lb = live_b & below;                                  // live lanes before this one
        int n = 0;
        bool raw = false, is_end = false;
        if (live) {
            n = seq[pos];
            const bool first = !seen && lb == 0ull;
            raw = !first && !((doflags[pos >> 5] >> (pos & 31)) & 1u);
            is_end = (endflags[pos >> 5] >> (pos & 31)) & 1u;
        }
        const bool skip = raw && !(n <= md_start && n > md_end);
        const unsigned long long after_b = __ballot(live && (is_end || skip));
        const int before = lb != 0ull ? static_cast<int>((after_b >> (63 - __clzll(static_cast<long long>(lb)))) & 1ull) : carry_skip;
        const bool e1 = live && (static_cast<int>(skip) != before);
        const bool e2 = live && is_end && !raw;
        const unsigned long long e1_b = __ballot(e1), e2_b = __ballot(e2);
        int slot = w + __popcll(e1_b & below) + __popcll(e2_b & below);
        if (e1) { if (slot <= k_tiles) write_row[slot] = n; ++slot; }
        if (e2) { if (slot <= k_tiles) write_row[slot] = n; }
        w += __popcll(e1_b) + __popcll(e2_b);
        if (live_b != 0ull) {
            carry_skip = static_cast<int>((after_b >> (63 - __clzll(static_cast<long long>(live_b)))) & 1ull);
            seen = true;
        }
    }
    if (lane == 0) write_row[0] = min(w - 1, k_tiles);
}

// o = 0, lse = +inf for `nrows` query rows of one (sequence, head) that has no keys (flash_api.cpp:1241-1245), by the whole
// workgroup. Varlen launches only: a fixed-length call with seqlen_k == 0 never reaches a forward kernel (la_api.hip).
__device__ __forceinline__ void store_empty_rows(const FwdParams& p, const SeqView& sv, int h, int row0, int nrows, int head_dim,
                                                 int tid, int nthreads) {
    const int chunks = head_dim / 8;
    const int rows = min(nrows, sv.seqlen_q - row0);
    for (int base = 0; base < rows * chunks; base += nthreads) {        // wave-uniform trip count (see la_fwd_kernel_x64.hip)
        const int i = base + tid;
        if (i >= rows * chunks) continue;
        const int r = row0 + i / chunks, ch = i % chunks;
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(p.o + sv.o_off + static_cast<int64_t>(r) * p.o_row_stride + h * p.o_head_stride + ch * 8) = z;
        if (p.lse != nullptr && ch == 0) sv.lse_row0[r] = INFINITY;
    }
}

// XCD-aware (bijective) block -> virtual work id. Blocks b%8 share an XCD/L2 (observed dispatch rule; used for
// speed only). Work is dealt to the XCDs in CHUNKS of 64 consecutive q-tiles of one head: 64 = the workgroups
// co-resident on one XCD (32 CUs x 2), so the co-resident set streams the same K/V tiles in the same order and
// K/V is read from HBM about once per chunk. Chunks go round-robin over the XCDs (chunk j -> XCD j%8), so every
// XCD sees every head: with real skip lists heads differ in sparsity, and one contiguous slab of heads per XCD
// (the first version of this map) left the kernel waiting for the XCD that drew the densest heads.
template <int C = 64>
__device__ __forceinline__ int xcd_work_id() {
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int full = (nwg / (8 * C)) * (8 * C);
    if (bid >= full) return bid;                       // ragged tail: identity (still a bijection)
    const int xcd = bid & 7, idx = bid >> 3;
    return ((idx / C) * 8 + xcd) * C + (idx % C);
}

// Work item of a static virtual id: (batch*head index, q-tile). q-tiles of a head are taken in the order
// [last, 0, 1, 2, ...]: under the descending key walk a tile can only be flagged against the running max SO FAR, so query
// rows whose dominant keys come late in the walk (low q-tiles of a self-attention) keep the longest lists, and the
// zero-padded last q-tile never skips anything (zero rows vote "do"): long items first is the list-scheduling rule (LPT).
__device__ __forceinline__ void work_item(const FwdParams& p, int vid, int& bh, int& m_block) {
    const int cnt = p.q_tile_count;
    bh = vid / cnt;
    m_block = p.q_tile_begin + (vid % cnt + cnt - 1) % cnt;
}

// ------------------------------------------------------------------------------------------------
// Dynamic work distribution (launches with skip lists; persistent workgroups). ONE thread calls next_work_item().
//
// Items differ 2-3x in length (real lists), and the hardware never moves a workgroup between XCDs, so a static map
// leaves the kernel waiting for the XCD that drew the most tiles. Tickets fix that - but ONE global ticket stream
// scatters the q-tiles of a head over all XCDs and the K/V tiles lose their L2 sharing (measured: L2 hit rate 83 % ->
// 50 %, 46 -> 138 GB of L2 fills per launch). So there are EIGHT ticket queues, one per XCD (HW_REG_XCC_ID), each a
// sequence of CHUNKS of C consecutive q-tiles of one head (C = the workgroups co-resident on an XCD): the CUs of an XCD
// walk the same K/V in the same order, as under the static map. Chunks are dealt to the queues round-robin from a chunk
// order that interleaves groups of G = 8 heads - with 8 queues queue x then walks head (8 j + x) from end to end, so a head's K/V is
// filled into ONE L2 (G = 4, rounds 1-2: two XCDs alternate on a head's chunks and both fill it; measured on the real step-49 lists,
// G = 8 is +1.9 % at 44 % and +1.0-3.1 % at 78 % sparsity, +0.5 % on banded lists, 170 vs 180 GB of L2 fills: profiles/r03_sched_sweep.md;
// the last round has 8x as many chunks to pack) and takes each head's q-tiles in the order [last, 0, 1, ...]
// (long items first, see work_item). A workgroup whose own queue is empty STEALS from the queue with the most tickets
// left, so the XCDs finish together (list-scheduling simulation on the per-row counts of a real 78 % list: makespan /
// ideal 1.09 for a head-major global stream, 1.01-1.03 here; tools/frag_bench.py).
// Counters: 8 x 64 bytes in the caller's workspace, zeroed on the launch stream. Every ticket of a queue goes to exactly
// one caller (atomicAdd), and a workgroup leaves only after it has seen all eight queues empty: every item is taken.
// ------------------------------------------------------------------------------------------------
#ifndef LA_SCHED_G
#define LA_SCHED_G 8          // heads whose chunks are interleaved in the ticket order: 8 = the queues, so every XCD queue walks ONE head at a time (its K/V is filled into one L2 only); measured against 1 / 2 / 4 / 16 on real lists: profiles/r03_sched_sweep.md
#endif
#ifndef LA_SCHED_C
#define LA_SCHED_C 32         // q-tiles per chunk of the one-workgroup-per-CU kernels (= workgroups co-resident on an XCD)
#endif
constexpr int kSchedQueues = 8;
constexpr int kSchedCounterStride = 16;      // uints: one 64-byte line per counter

struct SchedGeom {
    int cnt, nbh, C, nch, nv;                // q-tiles per head in this launch, batch*heads, chunk size, chunks/head, chunks
    __device__ SchedGeom(const FwdParams& p, int chunk) : cnt(p.q_tile_count), nbh(p.batch * p.num_heads), C(chunk) {
        nch = (cnt + C - 1) / C;
        nv = nbh * nch;
    }
    __device__ int queue_tickets(int x) const { return x < nv ? ((nv - x + kSchedQueues - 1) / kSchedQueues) * C : 0; }
    // ticket t of queue x -> item (bh * cnt + q-tile offset), or -2 for a padding slot of a head's last chunk
    __device__ int item(int x, unsigned t) const {
        constexpr int G = LA_SCHED_G;
        const int J = x + kSchedQueues * static_cast<int>(t / C);      // virtual chunk
        const int grp = J / (G * nch);
        const int g = min(G, nbh - grp * G);                           // heads in this group (only the last can be short)
        const int r = J - grp * G * nch;
        const int ci = r / g, bh = grp * G + r % g;
        const int qi = ci * C + static_cast<int>(t % C);
        if (qi >= cnt) return -2;
        return bh * cnt + (qi + cnt - 1) % cnt;
    }
};

#ifdef LA_SCHED_GANG
// A/B variant (never the default build): GANG scheduling of the chunks. The C = 32 items of a chunk are taken by the
// 32 workgroups of an XCD within a few microseconds of each other; an item of chunk c of queue x may START only when every item of
// the chunks before c in that queue has FINISHED (eight more counters behind the ticket counters, one per queue: items finished). The
// co-resident workgroups of an XCD then always work on one chunk - q-tiles [32 j, 32 j + 32) of one head, whose real lists overlap
// most - instead of drifting over several chunks; the price is the wait for the slowest of 32 lists per chunk.
// item -> (home queue, chunk index in that queue): the inverse of SchedGeom::item.
__device__ __forceinline__ void gang_home(const FwdParams& p, int chunk, int vid, int* queue, int* chunk_in_queue) {
    const SchedGeom geo(p, chunk);
    constexpr int G = LA_SCHED_G;
    const int bh = vid / geo.cnt, qi = (vid % geo.cnt + 1) % geo.cnt;        // item() stores (qi + cnt - 1) % cnt
    const int grp = bh / G, g = min(G, geo.nbh - grp * G);
    const int J = grp * G * geo.nch + (qi / geo.C) * g + (bh - grp * G);
    *queue = J % kSchedQueues;
    *chunk_in_queue = J / kSchedQueues;
}
// Bounded wait (a scheduling hint must never be able to hang the device): ~50 ms at most, then the item starts anyway.
__device__ __forceinline__ void gang_wait(const FwdParams& p, int chunk, int vid) {
    int x, c;
    gang_home(p, chunk, vid, &x, &c);
    const unsigned need = static_cast<unsigned>(c) * static_cast<unsigned>(chunk);
    unsigned* const done = &p.work_counter[(kSchedQueues + x) * kSchedCounterStride];
    for (int spin = 0; spin < (1 << 16); ++spin) {
        if (atomicAdd(done, 0u) >= need) break;
        __builtin_amdgcn_s_sleep(32);
    }
}
__device__ __forceinline__ void gang_done(const FwdParams& p, int chunk, int vid) {
    int x, c;
    gang_home(p, chunk, vid, &x, &c);
    atomicAdd(&p.work_counter[(kSchedQueues + x) * kSchedCounterStride], 1u);
}
#endif

// Returns the next item for this workgroup (bh * cnt + q-tile offset) or -1 when all queues are empty.
// *own_empty (workgroup state, kept in LDS by the caller) remembers that the XCD's own queue has run dry.
__device__ __forceinline__ int next_work_item(const FwdParams& p, int chunk, int* own_empty) {
    const SchedGeom geo(p, chunk);
    unsigned* const ctr = p.work_counter;
    const int xcd = static_cast<int>(__builtin_amdgcn_s_getreg((3 << 11) | 20)) & (kSchedQueues - 1);   // HW_REG_XCC_ID[3:0]
    auto take = [&](int x) -> int {          // -1: queue x is empty
        const int n = geo.queue_tickets(x);
        for (;;) {
            const unsigned t = atomicAdd(&ctr[x * kSchedCounterStride], 1u);
            if (t >= static_cast<unsigned>(n)) return -1;
            const int it = geo.item(x, t);
            if (it >= 0) return it;          // -2: padding slot, take the next ticket
#ifdef LA_SCHED_GANG
            atomicAdd(&ctr[(kSchedQueues + x) * kSchedCounterStride], 1u);     // a padding slot counts as a finished item of its chunk
#endif
        }
    };
    if (!*own_empty) {
        const int it = take(xcd);
        if (it >= 0) return it;
        *own_empty = 1;
    }
    for (int round = 0; round < 4 * kSchedQueues; ++round) {           // bounded: every failed take() means one more empty queue
        int best = -1, best_left = 0;
        for (int x = 0; x < kSchedQueues; ++x) {
            const int n = geo.queue_tickets(x);
            const unsigned done = atomicAdd(&ctr[x * kSchedCounterStride], 0u);      // coherent read (the L2s are per XCD)
            const int left = done < static_cast<unsigned>(n) ? n - static_cast<int>(done) : 0;
            if (left > best_left) { best_left = left; best = x; }
        }
        if (best < 0) return -1;
        const int it = take(best);
        if (it >= 0) return it;
    }
    // not reached in practice (queues only empty out); finish any remaining queue in order rather than leave items behind
    for (int x = 0; x < kSchedQueues; ++x) {
        const int it = take(x);
        if (it >= 0) return it;
    }
    return -1;
}

// Expand one read-list row into the LDS tile sequence (one wave, 64 ranges per pass). Returns the number of
// tiles (wave-uniform). Real lists hold hundreds of short ranges per row, so the ranges are handled in parallel:
// lane r takes range 64*pass + r, an exclusive wave scan of the range sizes gives its first position, then every
// lane writes its own tiles. The first range is walked even when len == 0 (mainloop...:93-101); indices are
// clamped to [0, k_tiles) and the total to k_tiles (memory safety on malformed lists). `endflags` (zeroed by the
// caller) gets one bit per range end.
__device__ __forceinline__ int expand_read_list(const int* __restrict__ row, int* seq, unsigned* endflags, int k_tiles,
                                                int lane) {
    const int len = max(row[0], 2);
    const int n_ranges = min(len >> 1, (k_tiles + 1) >> 1);
    int pos = 0;
    for (int base = 0; base < n_ranges; base += 64) {
        const int r = base + lane;
        int start = 0, cnt = 0;
        if (r < n_ranges) {
            // the row is k_tiles + 1 ints: a pair whose end would lie behind it (odd k_tiles with (k_tiles + 1) / 2 ranges; k_tiles = 1)
            // reads as (0, 0) - the oracle's reader_load rule - except the first pair, whose start is always in the row
            const bool in_row = 2 + 2 * r <= k_tiles;
            start = (in_row || r == 0) ? min(max(row[1 + 2 * r], 0), k_tiles - 1) : 0;
            const int end = in_row ? min(max(row[2 + 2 * r], 0), k_tiles - 1) : 0;
            cnt = max(start - end + 1, 0);
            if (r == 0) cnt = max(cnt, 1);        // the first tile of the first range is always walked (mainloop...:1614-1660)
        }
        const int incl = wave_inclusive_scan(cnt);            // inclusive scan over the wave (DPP)
        const int first = pos + incl - cnt;
        const int room = max(k_tiles - first, 0);
        cnt = min(cnt, room);
        // short ranges (the common case of real lists: hundreds of ranges of a few tiles): the lane writes its own tiles;
        // the rest of a LONG range (imposed bands, early denoising steps: 1-2 ranges of hundreds of tiles, which one lane
        // would write one LDS store at a time: 16 k cycles per item, tools/phase_profile_real.py) is filled by the whole wave
        // (all loops here have wave-uniform trip counts with predicated bodies: la_fwd_kernel_x64.hip, "COMPILER HAZARD")
        constexpr int kOwn = 4;
#pragma unroll
        for (int j = 0; j < kOwn; ++j)
            if (j < cnt) seq[first + j] = start - j;
        unsigned long long longs = __ballot(cnt > kOwn);
        while (longs) {
            const int src = __builtin_ctzll(longs);
            longs &= longs - 1;
            const int f = __shfl(first, src), s0 = __shfl(start, src), c = __shfl(cnt, src);
            for (int j0 = kOwn; j0 < c; j0 += 64)
                if (j0 + lane < c) seq[f + j0 + lane] = s0 - (j0 + lane);
        }
        if (cnt > 0) atomicOr(&endflags[(first + cnt - 1) >> 5], 1u << ((first + cnt - 1) & 31));
        pos = min(pos + __shfl(incl, 63), k_tiles);
    }
    return pos;
}

// Half-vote kernels: one read-list row -> a bitmap over the key tiles (bit t = the list names tile t) and the bitmap of its range
// ends. Same clamps as expand_read_list (indices into [0, k_tiles), at most (k_tiles + 1) / 2 ranges, the first tile of the first
// range always walked); the list is read as a SET - a well-formed list (descending, disjoint ranges: everything a writer, the
// initial list or a block mask produces) walks exactly its tiles in descending order either way, a malformed one (overlapping or
// ascending ranges) walks each named tile once. One wave; `in_bits` / `end_bits` zeroed by the caller.
__device__ __forceinline__ void expand_read_list_bits(const int* __restrict__ row, unsigned* in_bits, unsigned* end_bits, int k_tiles,
                                                      int lane) {
    const int len = max(row[0], 2);
    const int n_ranges = min(len >> 1, (k_tiles + 1) >> 1);
    for (int base = 0; base < n_ranges; base += 64) {
        const int r = base + lane;
        int hi = 0, cnt = 0;
        if (r < n_ranges) {
            const bool in_row = 2 + 2 * r <= k_tiles;
            hi = (in_row || r == 0) ? min(max(row[1 + 2 * r], 0), k_tiles - 1) : 0;
            const int end = in_row ? min(max(row[2 + 2 * r], 0), k_tiles - 1) : 0;
            cnt = max(hi - end + 1, 0);
            if (r == 0) cnt = max(cnt, 1);
        }
        const int lo = hi - cnt + 1;
        const int w_lo = lo >> 5;
        const int nw = cnt > 0 ? (hi >> 5) - w_lo + 1 : 0;
        for (int j = 0; __any(j < nw); ++j) {                  // wave-uniform trip count, predicated body (la_fwd_kernel_x64.hip, "COMPILER HAZARD")
            if (j < nw) {
                const int wd = w_lo + j;
                const int b_lo = max(lo - 32 * wd, 0), b_hi = min(hi - 32 * wd, 31);
                atomicOr(&in_bits[wd], (0xffffffffu >> (31 - b_hi)) & (0xffffffffu << b_lo));
            }
        }
        if (cnt > 0) atomicOr(&end_bits[lo >> 5], 1u << (lo & 31));
    }
}

}  // namespace la
