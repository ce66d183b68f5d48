// plnlp_edge_lists_build -- everything the backward of the edge gathers needs that depends on the batch's edges alone, in
// SEVEN hand-written launches and no library sort:
//
//   the node-sorted incidence lists of the batch   (item_edge, item_other, seg_ptr)        == plnlp_incidence_build
//   the touched-node compaction                     (rows, node_map, rowptr_c, count)       == plnlp_compact_rows
//   the endpoints / other endpoints as compact rows (src_c, dst_c, other_c)                 == plnlp_compact_endpoints
//
// bit for bit what those three entry points produce together (they ran a keys kernel, rocPRIM's onesweep radix sort -- four
// launches and five memset fills --, an items kernel, three compaction kernels and the compact-endpoints kernel: 15 launches,
// 0.2 ms of GPU time on the side stream of every training step).
//
// A stable sort by node of items that are generated in increasing item order is: per node, its items in increasing order.
// So no sort is needed, only each item's RANK among the items of its node:
//   1. count     cnt[node] += 1 for every item (integer atomics: the counts do not depend on the order of arrival); the
//                arrival ticket is kept as a provisional slot;
//   2. scan      seg_ptr = exclusive prefix sum of cnt over the nodes -- one three-kernel scan that ALSO yields the touched
//                rows, node_map, the compact row pointers and the count (a second prefix sum over the cnt > 0 flags rides
//                along), and appends the nodes with 3 .. WAVE_MAX items / with more to two lists;
//   3. scatter   item i goes to seg_ptr[node] + ticket: inside a node's segment the order is the atomics' -- arbitrary;
//   4. order     each segment is put into increasing item order and the final lists (edge id, other endpoint, its compact
//                row) are written: one thread for a segment of 1 or 2 items, one wave for 3 .. 64 (rank by v_readlane), one
//                workgroup beyond (rank sort in LDS up to 1024 items, a BITMAP over the item ids in LDS for hub nodes) --
//                deterministic whatever the arrival order was.
// Item ids fit a bitmap of 2 e bits in LDS up to e = 2^19 edges per batch (128 KiB); larger batches keep the sort-based
// entry points (the caller decides: plnlp_edge_lists_supported).
#include "common.hip.h"

namespace plnlp {
namespace el {

constexpr int CB = 1024;          // nodes per scan block
constexpr int WAVE_MAX = 64;      // segments of 3 .. WAVE_MAX items are ordered by one wave
constexpr int RANK_MAX = 1024;    // ... up to this many by one workgroup's rank sort, longer ones through the bitmap
constexpr int LONG_BLOCKS = 512;  // workgroups of the workgroup-per-segment pass (each walks its list with this stride)
constexpr int MED_BLOCKS = 1024;  // workgroups (of four waves) of the wave-per-segment pass

__device__ __forceinline__ int64_t wrap_node(int64_t i, int64_t n) { return i < 0 ? i + n : i; }
__device__ __forceinline__ int lanes_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// slot of this lane's item in a list that grows by one atomic per wave (want: wave-wide predicate; all active lanes call)
__device__ __forceinline__ int wave_append(int32_t* counter, bool want, int lane) {
    const uint64_t m = __ballot(want);
    if (m == 0) return 0;
    const int leader = __ffsll((unsigned long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader, 64);
    return base + lanes_below(m);
}

__global__ __launch_bounds__(256) void zero_kernel(int32_t* __restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0;
}

__global__ __launch_bounds__(256) void count_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                    int64_t n_edges, int64_t n_nodes, int32_t* __restrict__ cnt,
                                                    int32_t* __restrict__ slot) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * n_edges) return;
    const int64_t node = wrap_node(i < n_edges ? src[i] : dst[i - n_edges], n_nodes);
    slot[i] = atomicAdd(&cnt[node], 1);
}

// per block of CB nodes: items and touched nodes in the block
__global__ __launch_bounds__(CB) void block_sums_kernel(const int32_t* __restrict__ cnt, int64_t n_nodes,
                                                        int32_t* __restrict__ part) {
    __shared__ int w_items[CB / 64], w_rows[CB / 64];
    const int64_t r = (int64_t)blockIdx.x * CB + threadIdx.x;
    const int c = r < n_nodes ? cnt[r] : 0;
    int x = c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    const uint64_t m = __ballot(c > 0);
    if ((threadIdx.x & 63) == 0) { w_items[threadIdx.x >> 6] = x; w_rows[threadIdx.x >> 6] = __popcll(m); }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < CB / 64; ++i) { a += w_items[i]; b += w_rows[i]; }
        part[2 * blockIdx.x] = a;
        part[2 * blockIdx.x + 1] = b;
    }
}

// one block: exclusive scan of both columns of `part` in place; the number of touched nodes to *count
__global__ __launch_bounds__(CB) void scan_parts_kernel(int32_t* __restrict__ part, int64_t nb, int64_t* __restrict__ count) {
    __shared__ int wsum[2][CB / 64];
    __shared__ int carry[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2) carry[tid] = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += CB) {
        const int64_t i = base + tid;
        int v[2] = {i < nb ? part[2 * i] : 0, i < nb ? part[2 * i + 1] : 0};
        int x[2] = {v[0], v[1]};
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y0 = __shfl_up(x[0], o, 64), y1 = __shfl_up(x[1], o, 64);
            if (lane >= o) { x[0] += y0; x[1] += y1; }
        }
        if (lane == 63) { wsum[0][wave] = x[0]; wsum[1][wave] = x[1]; }
        __syncthreads();
        int woff[2] = {0, 0};
        for (int w = 0; w < wave; ++w) { woff[0] += wsum[0][w]; woff[1] += wsum[1][w]; }
        const int e0 = carry[0] + woff[0] + x[0] - v[0], e1 = carry[1] + woff[1] + x[1] - v[1];
        if (i < nb) { part[2 * i] = e0; part[2 * i + 1] = e1; }
        __syncthreads();
        if (tid == CB - 1) { carry[0] = e0 + v[0]; carry[1] = e1 + v[1]; }
        __syncthreads();
    }
    if (tid == 0 && count) *count = carry[1];
}

// seg_ptr, the compaction (the exact layout plnlp_compact_rows writes: touched rows first, then empty padding rows whose
// offsets repeat the end), and the list of long segments
__global__ __launch_bounds__(CB) void segments_kernel(const int32_t* __restrict__ cnt, int64_t n_nodes, int64_t n_items,
                                                      const int32_t* __restrict__ part, const int64_t* __restrict__ count,
                                                      int64_t* __restrict__ seg_ptr, int32_t* __restrict__ rows,
                                                      int32_t* __restrict__ node_map, int64_t* __restrict__ rowptr_c,
                                                      int32_t* __restrict__ long_list, int32_t* __restrict__ med_list,
                                                      int32_t* __restrict__ n_lists) {
    __shared__ int w_items[CB / 64], w_rows[CB / 64];
    const int64_t r = (int64_t)blockIdx.x * CB + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = r < n_nodes ? cnt[r] : 0;
    int x = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    const bool f = c > 0;
    const uint64_t m = __ballot(f);
    if (lane == 63) w_items[wave] = x;
    if (lane == 0) w_rows[wave] = __popcll(m);
    __syncthreads();
    int ioff = 0, roff = 0;
    for (int w = 0; w < wave; ++w) { ioff += w_items[w]; roff += w_rows[w]; }
    if (r >= n_nodes) return;
    const int64_t beg = (int64_t)part[2 * blockIdx.x] + ioff + x - c;
    const int idx = part[2 * blockIdx.x + 1] + roff + lanes_below(m);
    seg_ptr[r] = beg;
    // (one atomic per wave and list, written out: left to the compiler, the two appends were merged into a single per-lane
    // atomic on a lane-dependent address once the statements around them moved -- ~25 000 serialised atomics, the kernel 3x
    // longer and 0.14 ms on the collab step it overlaps, profiles/r04_same_box_ab.txt)
    const int at_long = wave_append(&n_lists[0], c > WAVE_MAX, lane);
    const int at_med = wave_append(&n_lists[1], c > 2 && c <= WAVE_MAX, lane);
    if (c > WAVE_MAX) long_list[at_long] = (int32_t)r;
    else if (c > 2) med_list[at_med] = (int32_t)r;
    if (r == n_nodes - 1) seg_ptr[n_nodes] = n_items;
    if (!rows) return;                      // the lists alone (no touched-node compaction asked for)
    node_map[r] = f ? idx : -1;
    if (f) {
        rows[idx] = (int32_t)r;
        rowptr_c[idx] = beg;
    } else {
        const int64_t cn = *count, u = r - idx;
        rows[cn + u] = 0;
        rowptr_c[cn + 1 + u] = n_items;
    }
    if (r == n_nodes - 1) rowptr_c[idx + (f ? 1 : 0)] = n_items;
}

// item i -> its node's segment at the arrival slot; the edges' endpoints as compact rows
__global__ __launch_bounds__(256) void scatter_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                      int64_t n_edges, int64_t n_nodes, const int64_t* __restrict__ seg_ptr,
                                                      const int32_t* __restrict__ slot, const int32_t* __restrict__ node_map,
                                                      int32_t* __restrict__ tmp, int64_t* __restrict__ src_c,
                                                      int64_t* __restrict__ dst_c) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * n_edges) return;
    const bool first = i < n_edges;
    const int64_t node = wrap_node(first ? src[i] : dst[i - n_edges], n_nodes);
    tmp[seg_ptr[node] + slot[i]] = (int32_t)i;
    if (src_c) {
        const int32_t cidx = node_map[node];
        if (first) src_c[i] = cidx; else dst_c[i - n_edges] = cidx;
    }
}

__device__ __forceinline__ void emit(int64_t p, int32_t item, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                     int64_t n_edges, int64_t n_nodes, const int32_t* __restrict__ node_map,
                                     int32_t* __restrict__ item_edge, int32_t* __restrict__ item_other,
                                     int32_t* __restrict__ other_c) {
    const bool first = item < n_edges;
    const int64_t e = first ? item : item - n_edges;
    const int64_t other = wrap_node(first ? dst[e] : src[e], n_nodes);
    item_edge[p] = (int32_t)e;
    item_other[p] = (int32_t)other;
    if (other_c) other_c[p] = node_map[other];
}

// Three kinds of workgroup in ONE launch (block-uniform choice):
//   [0, LONG_BLOCKS)                 segments of more than WAVE_MAX items, one workgroup each (list stride LONG_BLOCKS):
//                                    up to RANK_MAX items by a rank sort in LDS (an item's place = how many of the segment's
//                                    items are smaller: c / 256 items per thread, c broadcast reads each); beyond that through
//                                    a BITMAP over the item ids (set a bit per item, prefix-popcount the words, emit in bit
//                                    order);
//   [LONG_BLOCKS, +MED_BLOCKS)       segments of 3 .. WAVE_MAX items, one WAVE each: lane j holds item j, its place is the
//                                    number of smaller items among the c lanes (c v_readlane steps, no memory);
//   the rest                         one thread per node for the segments of 1 or 2 items (most nodes of a batch).
// Every path writes item ranks, so the result does not depend on the arrival order the atomics produced.
__global__ __launch_bounds__(256) void order_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                    int64_t n_edges, int64_t n_nodes, const int32_t* __restrict__ cnt,
                                                    const int64_t* __restrict__ seg_ptr, const int32_t* __restrict__ node_map,
                                                    const int32_t* __restrict__ long_list, const int32_t* __restrict__ med_list,
                                                    const int32_t* __restrict__ n_lists, const int32_t* __restrict__ tmp,
                                                    int32_t* __restrict__ item_edge, int32_t* __restrict__ item_other,
                                                    int32_t* __restrict__ other_c, int words) {
    extern __shared__ unsigned int bm[];          // [words]: one bit per item id (the bitmap path only)
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (blockIdx.x >= LONG_BLOCKS + MED_BLOCKS) {
        const int64_t r = (int64_t)(blockIdx.x - LONG_BLOCKS - MED_BLOCKS) * 256 + t;
        if (r >= n_nodes) return;
        const int c = cnt[r];
        if (c == 0 || c > 2) return;
        const int64_t beg = seg_ptr[r];
        const int32_t a = tmp[beg], b = c == 2 ? tmp[beg + 1] : a;
        emit(beg, a < b ? a : b, src, dst, n_edges, n_nodes, node_map, item_edge, item_other, other_c);
        if (c == 2) emit(beg + 1, a < b ? b : a, src, dst, n_edges, n_nodes, node_map, item_edge, item_other, other_c);
        return;
    }
    if (blockIdx.x >= LONG_BLOCKS) {
        const int nm = n_lists[1];
        for (int li = (blockIdx.x - LONG_BLOCKS) * 4 + wave; li < nm; li += MED_BLOCKS * 4) {
            const int64_t r = med_list[li];
            const int c = __builtin_amdgcn_readfirstlane(cnt[r]);                  // 3 .. WAVE_MAX
            const int64_t beg = seg_ptr[r];
            const int32_t v = lane < c ? tmp[beg + lane] : 0x7fffffff;
            int rank = 0;
            for (int j = 0; j < c; ++j) rank += (__builtin_amdgcn_readlane(v, j) < v) ? 1 : 0;
            if (lane < c) emit(beg + rank, v, src, dst, n_edges, n_nodes, node_map, item_edge, item_other, other_c);
        }
        return;
    }
    __shared__ unsigned int wtot[4];
    __shared__ int32_t vals[RANK_MAX];
    const int nl = n_lists[0];
    const int per = (words + 255) / 256;          // bitmap words per thread
    for (int li = blockIdx.x; li < nl; li += LONG_BLOCKS) {
        const int64_t r = long_list[li];
        const int c = cnt[r];
        const int64_t beg = seg_ptr[r];
        if (c <= RANK_MAX) {                      // block-uniform
            for (int a = t; a < c; a += 256) vals[a] = tmp[beg + a];
            __syncthreads();
            for (int a = t; a < c; a += 256) {
                const int32_t v = vals[a];
                int rank = 0;
                for (int j = 0; j < c; ++j) rank += (vals[j] < v) ? 1 : 0;
                emit(beg + rank, v, src, dst, n_edges, n_nodes, node_map, item_edge, item_other, other_c);
            }
            __syncthreads();
            continue;
        }
        for (int w = t; w < words; w += 256) bm[w] = 0u;
        __syncthreads();
        for (int a = t; a < c; a += 256) {
            const unsigned int it = (unsigned int)tmp[beg + a];
            atomicOr(&bm[it >> 5], 1u << (it & 31));
        }
        __syncthreads();
        const int w0 = t * per, w1 = (w0 + per < words) ? w0 + per : words;
        unsigned int mine = 0;
        for (int w = w0; w < w1; ++w) mine += __popc(bm[w]);
        unsigned int x = mine;                    // exclusive scan of the 256 per-thread counts
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        unsigned int base = x - mine;
        for (int w = 0; w < wave; ++w) base += wtot[w];
        int64_t p = beg + base;
        for (int w = w0; w < w1; ++w) {
            unsigned int bits = bm[w];
            while (bits) {
                const int b = __ffs((int)bits) - 1;
                bits &= bits - 1;
                emit(p++, (int32_t)(w * 32 + b), src, dst, n_edges, n_nodes, node_map, item_edge, item_other, other_c);
            }
        }
        __syncthreads();
    }
}

}  // namespace el
}  // namespace plnlp

// int32 words of workspace for a batch of n_edges edges on n_nodes nodes
extern "C" int64_t plnlp_edge_lists_workspace(int64_t n_edges, int64_t n_nodes) {
    if (n_edges < 0 || n_nodes <= 0) return 0;
    const int64_t nb = (n_nodes + plnlp::el::CB - 1) / plnlp::el::CB;
    // cnt [n] | n_lists [2] (+2 pad) | slot [2e] | tmp [2e] | part [2 nb] | long_list [2e / WAVE_MAX + 1] | med_list [2e / 3 + 1]
    return n_nodes + 4 + 4 * n_edges + 2 * nb + (2 * n_edges) / plnlp::el::WAVE_MAX + (2 * n_edges) / 3 + 8;
}

// 1 when the bitmap of a long segment (2 n_edges bits) fits the LDS: otherwise use the sort-based entry points
extern "C" int plnlp_edge_lists_supported(int64_t n_edges, int64_t n_nodes) {
    return n_edges > 0 && n_edges <= ((int64_t)1 << 19) && n_nodes > 0 && n_nodes < ((int64_t)1 << 31) ? 1 : 0;
}

extern "C" int plnlp_edge_lists_build(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes,
                                      int32_t* item_edge, int32_t* item_other, int64_t* seg_ptr, int32_t* rows,
                                      int32_t* node_map, int64_t* rowptr_c, int64_t* count, int64_t* src_c, int64_t* dst_c,
                                      int32_t* other_c, int32_t* workspace, int64_t workspace_ints, void* stream) {
    using namespace plnlp;
    using namespace plnlp::el;
    if (!plnlp_edge_lists_supported(n_edges, n_nodes)) return PLNLP_E_UNSUPPORTED;
    if (!src || !dst || !item_edge || !item_other || !seg_ptr || !workspace) return PLNLP_E_NULL;
    // rows == NULL: the node-sorted lists alone (a batch whose backward is not row-sparse); then no compaction output at all
    if (rows ? (!node_map || !rowptr_c || !count) : (node_map || rowptr_c || count || src_c || dst_c || other_c))
        return PLNLP_E_NULL;
    if ((src_c == nullptr) != (dst_c == nullptr)) return PLNLP_E_NULL;
    if (workspace_ints < plnlp_edge_lists_workspace(n_edges, n_nodes)) return PLNLP_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_items = 2 * n_edges;
    const int64_t nb = (n_nodes + CB - 1) / CB;
    int32_t* cnt = workspace;
    int32_t* n_lists = cnt + n_nodes;
    int32_t* slot = n_lists + 4;
    int32_t* tmp = slot + n_items;
    int32_t* part = tmp + n_items;
    int32_t* long_list = part + 2 * nb;
    int32_t* med_list = long_list + n_items / WAVE_MAX + 1;
    const unsigned item_blocks = (unsigned)((n_items + 255) / 256);
    int64_t zb = (n_nodes + 4 + 255) / 256;
    if (zb > 2048) zb = 2048;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)zb), dim3(256), 0, s, cnt, n_nodes + 4);
    hipLaunchKernelGGL(count_kernel, dim3(item_blocks), dim3(256), 0, s, src, dst, n_edges, n_nodes, cnt, slot);
    hipLaunchKernelGGL(block_sums_kernel, dim3((unsigned)nb), dim3(CB), 0, s, cnt, n_nodes, part);
    hipLaunchKernelGGL(scan_parts_kernel, dim3(1), dim3(CB), 0, s, part, nb, count);
    hipLaunchKernelGGL(segments_kernel, dim3((unsigned)nb), dim3(CB), 0, s, cnt, n_nodes, n_items, part, count, seg_ptr, rows,
                       node_map, rowptr_c, long_list, med_list, n_lists);
    hipLaunchKernelGGL(scatter_kernel, dim3(item_blocks), dim3(256), 0, s, src, dst, n_edges, n_nodes, seg_ptr, slot, node_map,
                       tmp, src_c, dst_c);
    const int words = (int)((n_items + 31) / 32);
    const unsigned order_blocks = (unsigned)(LONG_BLOCKS + MED_BLOCKS + (n_nodes + 255) / 256);
    static unsigned long long big_lds = 0;       // per device (bit = device ordinal; idempotent: a race sets a bit twice)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return PLNLP_E_UNSUPPORTED;
    if (dev < 0 || dev >= 64 || !((big_lds >> dev) & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                128 * 1024) != hipSuccess)
            return PLNLP_E_UNSUPPORTED;
        if (dev >= 0 && dev < 64) big_lds |= 1ull << dev;
    }
    hipLaunchKernelGGL(order_kernel, dim3(order_blocks), dim3(256), (size_t)words * 4, s, src, dst, n_edges, n_nodes, cnt,
                       seg_ptr, node_map, long_list, med_list, n_lists, tmp, item_edge, item_other, other_c, words);
    return launch_status();
}
