// Per-batch index structures for the backward of the edge gathers, built on the device with no
// host synchronisation:
//   plnlp_incidence_build  -- node-sorted incidence list of an edge batch (which batch edges touch
//                             each node, and the OTHER endpoint of each), via one keys-only radix sort
//                             of (node << shift | item) -- unique keys, so the order is fully
//                             determined and every later reduction is bit-reproducible;
//   plnlp_row_split_build  -- the long-row tables of plnlp_row_split for any CSR rowptr.
// The sort itself is rocPRIM's device radix sort (a ROCm library primitive, used as plumbing).
#include "common.hip.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace plnlp {

// rocPRIM switches to a merge sort below 1 Mi keys (17 launches, ~100 us for the 262 144 keys of a
// collab batch, and blind to the bit range); the digit-pass (onesweep) path sorts the 18 node bits in
// 3 passes.  Limit 0 = always onesweep above one block's worth of keys.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                              rocprim::default_config, 0>;

__device__ __forceinline__ int64_t wrap_node(int64_t i, int64_t n) { return i < 0 ? i + n : i; }

__global__ __launch_bounds__(256) void incidence_keys_kernel(const int64_t* __restrict__ src,
                                                             const int64_t* __restrict__ dst, int64_t n_edges,
                                                             int64_t n_nodes, int shift,
                                                             uint64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * n_edges) return;
    const int64_t node = wrap_node(i < n_edges ? src[i] : dst[i - n_edges], n_nodes);
    keys[i] = ((uint64_t)node << shift) | (uint64_t)i;
}

// sorted keys -> items (edge id, other endpoint) and the node segment pointers
__global__ __launch_bounds__(256) void incidence_items_kernel(const uint64_t* __restrict__ keys,
                                                              const int64_t* __restrict__ src,
                                                              const int64_t* __restrict__ dst, int64_t n_edges,
                                                              int64_t n_nodes, int shift,
                                                              int32_t* __restrict__ item_edge,
                                                              int32_t* __restrict__ item_other,
                                                              int64_t* __restrict__ seg_ptr) {
    const int64_t n_items = 2 * n_edges;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_items) return;
    const uint64_t key = keys[p];
    const int64_t node = (int64_t)(key >> shift);
    const int64_t i = (int64_t)(key & ((1ull << shift) - 1));
    const int64_t e = i < n_edges ? i : i - n_edges;
    item_edge[p] = (int32_t)e;
    item_other[p] = (int32_t)wrap_node(i < n_edges ? dst[e] : src[e], n_nodes);
    // seg_ptr[n] = first sorted position whose node >= n
    const int64_t prev = p == 0 ? -1 : (int64_t)(keys[p - 1] >> shift);
    for (int64_t n = prev + 1; n <= node; ++n) seg_ptr[n] = p;
    if (p == n_items - 1)
        for (int64_t n = node + 1; n <= n_nodes; ++n) seg_ptr[n] = n_items;
}

__global__ __launch_bounds__(256) void fill_i64_kernel(int64_t* __restrict__ p, int64_t n, int64_t v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}

// rows longer than `threshold` get a slot (atomic compaction: slot ORDER is arbitrary, which is
// harmless -- each long row's chunks are still reduced in chunk order) and a run of chunk ids
__global__ __launch_bounds__(256) void row_split_build_kernel(const int64_t* __restrict__ rowptr, int64_t n_rows,
                                                              int64_t threshold, int64_t n_long_cap,
                                                              int64_t n_chunks_cap,
                                                              int64_t* __restrict__ long_rows,
                                                              int64_t* __restrict__ chunk_beg,
                                                              int32_t* __restrict__ chunk_cnt,
                                                              int32_t* __restrict__ chunk_long,
                                                              unsigned long long* __restrict__ counters) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t deg = rowptr[r + 1] - rowptr[r];
    if (deg <= threshold) return;
    const int64_t nch = (deg + threshold - 1) / threshold;
    // the counters start at ~0 (the tables and the counters are cleared by ONE 0xFF fill): +1 wraps to 0
    const int64_t slot = (int64_t)(atomicAdd(&counters[0], 1ull) + 1ull);
    const int64_t cb = (int64_t)(atomicAdd(&counters[1], (unsigned long long)nch) + 1ull);
    if (slot >= n_long_cap || cb + nch > n_chunks_cap) { atomicAdd(&counters[2], 1ull); return; }  // overflow flag
    long_rows[slot] = r;
    chunk_beg[slot] = cb;
    chunk_cnt[slot] = (int32_t)nch;
    for (int64_t j = 0; j < nch; ++j) chunk_long[cb + j] = (int32_t)slot;
}

// uniform random walks (torch_cluster.random_walk semantics, main.py:242): one thread per walker,
// next = col[rowptr[cur] + floor(u * deg)], stay put on a node without neighbours.  u comes from the
// counter hash of (seed, walker * L + step), so a walk is reproducible and order-independent.
__global__ __launch_bounds__(256) void random_walk_kernel(const int64_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col,
                                                          const int64_t* __restrict__ start, int64_t n_walkers,
                                                          int walk_length, uint32_t seed_lo, uint32_t seed_hi,
                                                          int64_t* __restrict__ walks) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= n_walkers) return;
    int64_t cur = start[w];
    int64_t* o = walks + w * (walk_length + 1);
    o[0] = cur;
    for (int l = 0; l < walk_length; ++l) {
        const int64_t beg = rowptr[cur];
        const uint64_t deg = (uint64_t)(rowptr[cur + 1] - beg);
        if (deg > 0) {
            const uint64_t idx = (uint64_t)w * (uint64_t)walk_length + (uint64_t)l;
            const uint32_t h = counter_hash(idx, seed_lo, seed_hi);
            cur = col[beg + (int64_t)(((uint64_t)h * deg) >> 32)];
        }
        o[l + 1] = cur;
    }
}

// ---- R-MAT edge stream (BASELINE.json config 5: the skewed HBM stress graph) ----------------------
// Edge e of the stream (a global edge id, so any rank can replay any part of the stream without holding the
// rest): `scale` quadrant draws, one per bit, each from the counter hash of (seed, e * 64 + level) compared
// against 32-bit integer thresholds (a, a+b, a+b+c as fractions of 2^32 -- no floating point, so the numpy
// restatement in oracle/reference_path.py::rmat_edges_ref is bit-exact); the raw 2^scale ids then go through a
// seeded bijection of [0, 2^scale) (Graph500-style relabelling: raw R-MAT ids put every hub at a power-of-two
// id) and are folded mod n_nodes.
__host__ __device__ inline uint64_t rmat_relabel(uint64_t x, int scale, uint64_t seed) {
    const uint64_t M = scale >= 64 ? ~0ull : ((1ull << scale) - 1ull);
    const int sh = (scale + 1) / 2;
    const uint64_t c1 = (seed * 0x9E3779B97F4A7C15ull) >> 7, c2 = (seed ^ 0xD6E8FEB86659FD93ull) * 0xBF58476D1CE4E5B9ull;
    x = (x * 0x9E3779B97F4A7C15ull + c1) & M;      // odd multiplier: a bijection of the low `scale` bits
    x ^= x >> sh;                                  // xor-shift: a bijection
    x = (x * 0xBF58476D1CE4E5B9ull) & M;
    x ^= x >> sh;
    x = (x * 0x94D049BB133111EBull + c2) & M;
    x ^= x >> sh;
    return x;
}

__global__ __launch_bounds__(256) void rmat_edges_kernel(int scale, int64_t n_nodes, int64_t edge_lo, int64_t n_edges,
                                                         uint32_t seed_lo, uint32_t seed_hi, uint32_t t_a,
                                                         uint32_t t_ab, uint32_t t_abc, int relabel,
                                                         int32_t* __restrict__ rows, int32_t* __restrict__ cols) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_edges) return;
    const uint64_t e = (uint64_t)(edge_lo + i);
    uint64_t r = 0, c = 0;
    for (int l = 0; l < scale; ++l) {
        const uint32_t h = counter_hash(e * 64ull + (uint64_t)l, seed_lo, seed_hi);
        const uint32_t right = ((h >= t_a) & (h < t_ab)) | (h >= t_abc);      // quadrants b, d -> column bit
        const uint32_t down = h >= t_ab;                                       // quadrants c, d -> row bit
        r = (r << 1) | down;
        c = (c << 1) | right;
    }
    if (relabel) {
        const uint64_t seed = ((uint64_t)seed_hi << 32) | seed_lo;
        r = rmat_relabel(r, scale, seed);
        c = rmat_relabel(c, scale, seed);
    }
    rows[i] = (int32_t)(r % (uint64_t)n_nodes);
    cols[i] = (int32_t)(c % (uint64_t)n_nodes);
}

// ---- non-empty rows of a CSR, in order (three launches: block counts, scan of the counts, write) --
constexpr int CB = 1024;

__device__ __forceinline__ int lanes_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

__global__ __launch_bounds__(CB) void compact_count_kernel(const int64_t* __restrict__ rowptr, int64_t n_rows,
                                                           int32_t* __restrict__ block_cnt) {
    __shared__ int wcnt[CB / 64];
    const int64_t r = (int64_t)blockIdx.x * CB + threadIdx.x;
    const bool f = r < n_rows && rowptr[r + 1] > rowptr[r];
    const uint64_t m = __ballot(f);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
#pragma unroll
        for (int i = 0; i < CB / 64; ++i) sum += wcnt[i];
        block_cnt[blockIdx.x] = sum;
    }
}

// one block: exclusive scan of block_cnt in place, total to *count
__global__ __launch_bounds__(CB) void compact_scan_kernel(int32_t* __restrict__ block_cnt, int64_t nb,
                                                          int64_t* __restrict__ count) {
    __shared__ int wsum[CB / 64];
    __shared__ int64_t carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += CB) {
        const int64_t i = base + tid;
        const int v = i < nb ? block_cnt[i] : 0;
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int64_t excl = carry + woff + x - v;
        if (i < nb) block_cnt[i] = (int32_t)excl;
        __syncthreads();
        if (tid == CB - 1) carry = excl + v;
        __syncthreads();
    }
    if (tid == 0) *count = carry;
}

__global__ __launch_bounds__(CB) void compact_write_kernel(const int64_t* __restrict__ rowptr, int64_t n_rows,
                                                           const int32_t* __restrict__ block_off,
                                                           int32_t* __restrict__ rows, int32_t* __restrict__ node_map,
                                                           int64_t* __restrict__ rowptr_c,
                                                           const int64_t* __restrict__ count) {
    __shared__ int wcnt[CB / 64];
    const int64_t r = (int64_t)blockIdx.x * CB + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    int64_t beg = 0;
    bool f = false;
    if (r < n_rows) { beg = rowptr[r]; f = rowptr[r + 1] > beg; }
    const uint64_t m = __ballot(f);
    if ((threadIdx.x & 63) == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wcnt[w];
    const int idx = block_off[blockIdx.x] + woff + lanes_below(m);
    if (r < n_rows) {
        node_map[r] = f ? idx : -1;
        if (f) {
            rows[idx] = (int32_t)r;
            rowptr_c[idx] = beg;
        } else {
            // the u-th empty row pads the tail: past `count` the compact CSR continues with empty rows
            // (row id 0, offset = end), so it can be walked over its CAPACITY without knowing the count
            const int64_t c = *count, u = r - idx;
            rows[c + u] = 0;
            rowptr_c[c + 1 + u] = rowptr[n_rows];
        }
        if (r == n_rows - 1) rowptr_c[idx + (f ? 1 : 0)] = rowptr[n_rows];
    }
}

// ---- the endpoint lists of a batch in one launch each (they were 2 cat + 3 gather + 3 cast launches of stock ops) ----
// src / dst = column 0 / 1 of [pos (na rows) ; neg (nb rows)], both row-major [*, 2] int64
__global__ __launch_bounds__(256) void edge_endpoints_kernel(const int64_t* __restrict__ a, int64_t na,
                                                             const int64_t* __restrict__ b, int64_t nb,
                                                             int64_t* __restrict__ src, int64_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= na + nb) return;
    const int64_t* row = i < na ? a + 2 * i : b + 2 * (i - na);
    src[i] = row[0];
    dst[i] = row[1];
}
// endpoints and the incidence lists' other endpoints as rows of the matrix that holds only the touched nodes
__global__ __launch_bounds__(256) void compact_endpoints_kernel(const int32_t* __restrict__ node_map,
                                                                const int64_t* __restrict__ src,
                                                                const int64_t* __restrict__ dst, int64_t n_edges,
                                                                const int32_t* __restrict__ item_other, int64_t n_items,
                                                                int64_t* __restrict__ src_c, int64_t* __restrict__ dst_c,
                                                                int32_t* __restrict__ other_c) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_edges) {
        src_c[i] = (int64_t)node_map[src[i]];
        dst_c[i] = (int64_t)node_map[dst[i]];
    }
    if (i < n_items) other_c[i] = node_map[item_other[i]];
}

static inline int bits_for(int64_t n) {  // smallest b with (1 << b) >= n
    int b = 0;
    while (((int64_t)1 << b) < n) ++b;
    return b;
}

}  // namespace plnlp

extern "C" int64_t plnlp_incidence_temp_bytes(int64_t n_edges) {
    size_t bytes = 0;
    uint64_t* k = nullptr;
    if (n_edges <= 0) return 0;
    rocprim::radix_sort_keys<plnlp::SortConfig>(nullptr, bytes, k, k, (size_t)(2 * n_edges), 0, 64, (hipStream_t)0);
    return (int64_t)bytes + 256;
}

extern "C" int plnlp_incidence_build(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes,
                                     uint64_t* keys_a, uint64_t* keys_b, void* temp, int64_t temp_bytes,
                                     int32_t* item_edge, int32_t* item_other, int64_t* seg_ptr, void* stream) {
    using namespace plnlp;
    if (n_edges < 0 || n_nodes <= 0 || n_edges >= (1ll << 30)) return PLNLP_E_SHAPE;
    if (!seg_ptr) return PLNLP_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    if (n_edges == 0) {
        hipLaunchKernelGGL(fill_i64_kernel, dim3(64), dim3(256), 0, s, seg_ptr, n_nodes + 1, (int64_t)0);
        return launch_status();
    }
    if (!src || !dst || !keys_a || !keys_b || !temp || !item_edge || !item_other) return PLNLP_E_NULL;
    const int64_t n_items = 2 * n_edges;
    const int shift = bits_for(n_items);
    const int node_bits = bits_for(n_nodes);
    if (shift + node_bits > 64) return PLNLP_E_SHAPE;
    // the keys are generated in increasing item order and the radix sort is stable, so sorting on the
    // NODE bits alone already yields the (node, item) order: half the digit passes
    const unsigned bit0 = (unsigned)shift, bit1 = (unsigned)(shift + (node_bits > 0 ? node_bits : 1));
    size_t need = 0;
    rocprim::radix_sort_keys<SortConfig>(nullptr, need, keys_a, keys_b, (size_t)n_items, bit0, bit1, s);
    if ((int64_t)need > temp_bytes) return PLNLP_E_WORKSPACE;
    const unsigned blocks = (unsigned)((n_items + 255) / 256);
    hipLaunchKernelGGL(incidence_keys_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n_edges, n_nodes, shift, keys_a);
    if (int rc = launch_status()) return rc;
    size_t tb = (size_t)temp_bytes;
    hipError_t err = rocprim::radix_sort_keys<SortConfig>(temp, tb, keys_a, keys_b, (size_t)n_items, bit0, bit1, s);
    if (err != hipSuccess) return (int)err;
    hipLaunchKernelGGL(incidence_items_kernel, dim3(blocks), dim3(256), 0, s, keys_b, src, dst, n_edges, n_nodes,
                       shift, item_edge, item_other, seg_ptr);
    return launch_status();
}

extern "C" int plnlp_edge_endpoints(const int64_t* pos, int64_t n_pos, const int64_t* neg, int64_t n_neg,
                                    int64_t* src, int64_t* dst, void* stream) {
    using namespace plnlp;
    if (n_pos < 0 || n_neg < 0 || n_pos + n_neg >= (1ll << 31)) return PLNLP_E_SHAPE;
    if (n_pos + n_neg == 0) return 0;
    if ((n_pos > 0 && !pos) || (n_neg > 0 && !neg) || !src || !dst) return PLNLP_E_NULL;
    const int64_t n = n_pos + n_neg;
    hipLaunchKernelGGL(edge_endpoints_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos,
                       n_pos, neg, n_neg, src, dst);
    return launch_status();
}

extern "C" int plnlp_compact_endpoints(const int32_t* node_map, const int64_t* src, const int64_t* dst, int64_t n_edges,
                                       const int32_t* item_other, int64_t n_items, int64_t* src_c, int64_t* dst_c,
                                       int32_t* other_c, void* stream) {
    using namespace plnlp;
    if (n_edges < 0 || n_items < 0 || n_edges >= (1ll << 31) || n_items >= (1ll << 31)) return PLNLP_E_SHAPE;
    const int64_t n = n_edges > n_items ? n_edges : n_items;
    if (n == 0) return 0;
    if (!node_map || (n_edges > 0 && (!src || !dst || !src_c || !dst_c)) || (n_items > 0 && (!item_other || !other_c)))
        return PLNLP_E_NULL;
    hipLaunchKernelGGL(compact_endpoints_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       node_map, src, dst, n_edges, item_other, n_items, src_c, dst_c, other_c);
    return launch_status();
}

extern "C" int64_t plnlp_compact_rows_workspace(int64_t n_rows) {
    return n_rows <= 0 ? 1 : (n_rows + plnlp::CB - 1) / plnlp::CB;
}

extern "C" int plnlp_compact_rows(const int64_t* rowptr, int64_t n_rows, int32_t* rows, int32_t* node_map,
                                  int64_t* rowptr_c, int64_t* count, int32_t* block_ws, void* stream) {
    using namespace plnlp;
    if (!rowptr || !rowptr_c || !count) return PLNLP_E_NULL;
    if (n_rows < 0 || n_rows > 0x7FFFFFFF) return PLNLP_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (n_rows == 0) {
        hipError_t e = hipMemsetAsync(count, 0, sizeof(int64_t), s);
        if (e == hipSuccess) e = hipMemcpyAsync(rowptr_c, rowptr, sizeof(int64_t), hipMemcpyDeviceToDevice, s);
        return e == hipSuccess ? 0 : (int)e;
    }
    if (!rows || !node_map || !block_ws) return PLNLP_E_NULL;
    const int64_t nb = (n_rows + CB - 1) / CB;
    hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)nb), dim3(CB), 0, s, rowptr, n_rows, block_ws);
    if (int rc = launch_status()) return rc;
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(CB), 0, s, block_ws, nb, count);
    if (int rc = launch_status()) return rc;
    hipLaunchKernelGGL(compact_write_kernel, dim3((unsigned)nb), dim3(CB), 0, s, rowptr, n_rows, block_ws, rows,
                       node_map, rowptr_c, count);
    return launch_status();
}

extern "C" int plnlp_row_split_build(const int64_t* rowptr, int64_t n_rows, int64_t threshold, int64_t n_long_cap,
                                     int64_t n_chunks_cap, int64_t* long_rows, int64_t* chunk_beg,
                                     int32_t* chunk_cnt, int32_t* chunk_long, int64_t* counters, void* stream) {
    using namespace plnlp;
    if (!rowptr || !long_rows || !chunk_beg || !chunk_cnt || !chunk_long || !counters) return PLNLP_E_NULL;
    if (n_rows < 0 || threshold < 64 || n_long_cap <= 0 || n_chunks_cap <= 0) return PLNLP_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    // -1 everywhere: idle slots (long_rows, chunk_long), counters at ~0.  When the caller laid the five
    // arrays out back to back starting at `counters` (plnlp_amd/graph.py does) this is ONE fill.
    hipError_t e;
    const char* base = reinterpret_cast<const char*>(counters);
    const bool packed = reinterpret_cast<const char*>(long_rows) == base + 32 &&
                        reinterpret_cast<const char*>(chunk_beg) == base + 32 + 8 * n_long_cap &&
                        reinterpret_cast<const char*>(chunk_cnt) == base + 32 + 16 * n_long_cap &&
                        reinterpret_cast<const char*>(chunk_long) == base + 32 + 20 * n_long_cap;
    if (packed) {
        e = hipMemsetAsync(counters, 0xFF, 32 + 20 * n_long_cap + 4 * n_chunks_cap, s);
    } else {
        e = hipMemsetAsync(long_rows, 0xFF, sizeof(int64_t) * n_long_cap, s);
        if (e == hipSuccess) e = hipMemsetAsync(chunk_long, 0xFF, sizeof(int32_t) * n_chunks_cap, s);
        if (e == hipSuccess) e = hipMemsetAsync(counters, 0xFF, sizeof(int64_t) * 4, s);
    }
    if (e != hipSuccess) return (int)e;
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(row_split_build_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s, rowptr,
                       n_rows, threshold, n_long_cap, n_chunks_cap, long_rows, chunk_beg, chunk_cnt, chunk_long,
                       reinterpret_cast<unsigned long long*>(counters));
    return launch_status();
}

extern "C" int plnlp_random_walk(const int64_t* rowptr, const int32_t* col, const int64_t* start,
                                 int64_t n_walkers, int walk_length, uint64_t seed, int64_t* walks, void* stream) {
    using namespace plnlp;
    if (n_walkers < 0 || walk_length < 0 || walk_length > (1 << 20)) return PLNLP_E_SHAPE;
    if (n_walkers == 0) return 0;
    if (!rowptr || !start || !walks || (walk_length > 0 && !col)) return PLNLP_E_NULL;
    hipLaunchKernelGGL(random_walk_kernel, dim3((unsigned)((n_walkers + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, rowptr, col, start, n_walkers, walk_length, (uint32_t)seed,
                       (uint32_t)(seed >> 32), walks);
    return launch_status();
}

extern "C" int plnlp_rmat_edges(int scale, int64_t n_nodes, int64_t edge_lo, int64_t n_edges, uint64_t seed,
                                uint32_t t_a, uint32_t t_ab, uint32_t t_abc, int relabel, int32_t* rows,
                                int32_t* cols, void* stream) {
    using namespace plnlp;
    if (scale < 1 || scale > 40 || n_nodes < 1 || n_nodes > 0x7FFFFFFF || edge_lo < 0 || n_edges < 0)
        return PLNLP_E_SHAPE;
    if (!(t_a <= t_ab && t_ab <= t_abc)) return PLNLP_E_SHAPE;
    if (n_edges == 0) return 0;
    if (!rows || !cols) return PLNLP_E_NULL;
    if (n_edges > ((int64_t)1 << 38)) return PLNLP_E_SHAPE;
    constexpr int64_t MAX_BLOCKS = (int64_t)1 << 22;            // < 2^32 threads per launch
    const int64_t blocks = (n_edges + 255) / 256;
    for (int64_t b0 = 0; b0 < blocks; b0 += MAX_BLOCKS) {
        const int64_t nb = (blocks - b0) < MAX_BLOCKS ? (blocks - b0) : MAX_BLOCKS;
        const int64_t off = b0 * 256;
        const int64_t cnt = (n_edges - off) < nb * 256 ? (n_edges - off) : nb * 256;
        hipLaunchKernelGGL(rmat_edges_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, scale, n_nodes,
                           edge_lo + off, cnt, (uint32_t)seed, (uint32_t)(seed >> 32), t_a, t_ab, t_abc, relabel,
                           rows + off, cols + off);
        if (int rc = launch_status()) return rc;
    }
    return 0;
}
