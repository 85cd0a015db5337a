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
    __device__ void load() { start = row[idx]; end = row[idx + 1]; }
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
            if (md.end > n && md.has_more()) { md.advance(); md.load(); }
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
        // state left behind by this position: forced to "skipping" after a range end
        const int after = (is_end || skip) ? 1 : 0;
        int before = __shfl_up(after, 1);
        if (lane == 0) before = carry_skip;
        const int e1 = live && (static_cast<int>(skip) != before);
        const int e2 = live && is_end && !raw;
        const int cnt = e1 + e2;
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        int slot = w + incl - cnt;
        if (e1) { if (slot <= k_tiles) write_row[slot] = n; ++slot; }
        if (e2) { if (slot <= k_tiles) write_row[slot] = n; }
        w += __shfl(incl, 63);
        carry_skip = __shfl(after, 63);
    }
    if (lane == 0) write_row[0] = min(w - 1, k_tiles);
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

// Work item of a virtual id / ticket: (batch*head index, q-tile).
//  * q-tiles of a head are taken in the order [last, 0, 1, 2, ...]: under the descending key walk a tile can only be
//    flagged against the running max SO FAR, so query rows whose dominant keys come late in the walk (low q-tiles of a
//    self-attention) keep the longest lists, and the zero-padded last q-tile never skips anything (zero rows vote "do"):
//    long items first is the list-scheduling rule (LPT).
//  * dynamic (ticket) mode deals GROUPS OF 4 HEADS with their q-tiles interleaved: the final group then has 4x as many
//    items to pack into the last round (makespan / ideal 1.088 -> 1.008 in a list-scheduling simulation on the per-row
//    counts of a real 78 % list, tools/frag_bench.py), while the K/V live set stays at 4 heads (155 MB at S = 75 600, inside the
//    256 MB Infinity Cache).
__device__ __forceinline__ void work_item(const FwdParams& p, int vid, bool dynamic, int& bh, int& m_block) {
    const int cnt = p.q_tile_count;
    int qi;
    if (dynamic) {
        constexpr int G = 4;
        const int grp = vid / (G * cnt);
        const int bh0 = grp * G;
        const int g = min(G, p.batch * p.num_heads - bh0);        // heads in this group (only the last group can be short)
        const int i = vid - grp * G * cnt;
        qi = i / g;
        bh = bh0 + i % g;
    } else {
        qi = vid % cnt;
        bh = vid / cnt;
    }
    m_block = p.q_tile_begin + (qi + cnt - 1) % cnt;
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
            start = min(max(row[1 + 2 * r], 0), k_tiles - 1);
            const int end = min(max(row[2 + 2 * r], 0), k_tiles - 1);
            cnt = max(start - end + 1, 0);
            if (r == 0) cnt = max(cnt, 1);        // the first tile of the first range is always walked (mainloop...:1614-1660)
        }
        int incl = cnt;                                       // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        const int first = pos + incl - cnt;
        const int room = max(k_tiles - first, 0);
        cnt = min(cnt, room);
        for (int j = 0; j < cnt; ++j) seq[first + j] = start - j;
        if (cnt > 0) atomicOr(&endflags[(first + cnt - 1) >> 5], 1u << ((first + cnt - 1) & 31));
        pos = min(pos + __shfl(incl, 63), k_tiles);
    }
    return pos;
}

}  // namespace la
