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

__device__ __noinline__ void write_skip_list(const int* __restrict__ read_row, int* __restrict__ write_row,
                                             const int* __restrict__ must_do_row, const unsigned* doflags,
                                             int k_tiles) {
    ListReader rd, md;
    rd.init(read_row);
    const bool has_md = must_do_row != nullptr;
    if (has_md) md.init(must_do_row);
    int w = 1;
    bool is_skipping = true;
    auto transition = [&](bool skip, int n, bool use_md) {
        if (use_md && skip) {
            if (md.end > n && md.has_more()) { md.advance(); md.load(); }
            const bool must_do = n <= md.start && n > md.end;
            skip = skip && !must_do;
        }
        if (skip != is_skipping) {
            if (w <= k_tiles) write_row[w] = n;
            ++w;
            is_skipping = skip;
        }
    };
    int pos = 0;
    int n = min(max(rd.start, 0), k_tiles - 1);
    bool skip = false;
    transition(false, n, false);
    --n; ++pos;
    for (;;) {
        const int end = min(max(rd.end, 0), k_tiles - 1);
        for (; n >= end && pos < k_tiles; --n, ++pos) {
            skip = !((doflags[pos >> 5] >> (pos & 31)) & 1u);
            transition(skip, n, has_md);
        }
        // record_range_end (:173-181)
        is_skipping = true;
        if (!skip) { if (w <= k_tiles) write_row[w] = end; ++w; }
        rd.advance();
        if (!rd.has_more()) break;
        rd.load();
        n = min(max(rd.start, 0), k_tiles - 1);
    }
    write_row[0] = min(w - 1, k_tiles);
}


// XCD-aware (bijective) block -> virtual work id: blocks b%8 share an XCD/L2, so each XCD gets a contiguous
// run of q-tiles of the same head (they stream the same K/V tiles in the same order).
__device__ __forceinline__ int xcd_work_id() {
    const int bid = blockIdx.x, nwg = gridDim.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
}

// Expand one read-list row into the LDS tile sequence (one wave). Returns the number of tiles (lane-uniform).
// The first range is walked even when len == 0 (mainloop...:93-101). Indices are clamped to [0, k_tiles).
__device__ __forceinline__ int expand_read_list(const int* __restrict__ row, int* seq, int k_tiles, int lane) {
    const int len = row[0];
    int idx = 1, pos = 0;
    do {
        const int start = min(max(row[idx], 0), k_tiles - 1);
        const int end = min(max(row[idx + 1], 0), k_tiles - 1);
        const int cnt = min(start - end + 1, k_tiles - pos);
        for (int j = lane; j < cnt; j += 64) seq[pos + j] = start - j;
        pos += max(cnt, 0);
        idx += 2;
    } while (idx <= len);
    return pos;
}

}  // namespace la
