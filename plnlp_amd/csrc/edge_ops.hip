// K3 -- edge endpoint gather + score, forward and backward (gfx950).
//
// Forward: the reference materialises h[src], h[dst] ([E,F] each) and then
// multiplies (plnlp/model.py:155-156 -> layer.py:81,175).  Here the two gathers
// and the product / dot are one pass: LPR lanes own one edge, 16 B per lane per
// endpoint row, both endpoint rows in flight together.
//
// Backward: index_put_(accumulate=True) in the reference.  Two forms:
//   * plnlp_edge_scatter_bwd_f32  -- hardware fp32 atomics (order not fixed);
//   * plnlp_edge_segment_bwd_f32  -- atomic-free gather-reduce over a node-sorted
//     incidence list (deterministic); also overwrites untouched rows with 0 so no
//     memset of gh is needed.
#include "common.hip.h"

namespace plnlp {

__device__ __forceinline__ int64_t wrap_index(int64_t i, int64_t n_rows) { return i < 0 ? i + n_rows : i; }

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------- forward: dot ---------------------------------------------------
template <int LPR, bool VEC>
__global__ __launch_bounds__(256) void edge_dot_fwd_kernel(const float* __restrict__ h, int64_t ldh, int64_t n_rows,
                                                           const int64_t* __restrict__ src,
                                                           const int64_t* __restrict__ dst, int64_t n_edges,
                                                           int feat, float* __restrict__ out) {
    constexpr int GPB = 256 / LPR;  // edges per block iteration
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    for (int64_t e = (int64_t)blockIdx.x * GPB + grp; e < n_edges; e += (int64_t)gridDim.x * GPB) {
        const float* a = h + wrap_index(src[e], n_rows) * ldh;
        const float* b = h + wrap_index(dst[e], n_rows) * ldh;
        float acc = 0.f;
        if constexpr (VEC) {
            const int nslots = feat >> 2;
            for (int s = sub; s < nslots; s += LPR) {
                const float4 u = reinterpret_cast<const float4*>(a)[s];
                const float4 v = reinterpret_cast<const float4*>(b)[s];
                acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc);
                acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
            }
        } else {
            for (int f = sub; f < feat; f += LPR) acc = fmaf(a[f], b[f], acc);
        }
        acc = group_sum<LPR>(acc);
        if (sub == 0) out[e] = acc;
    }
}

// ---------------- forward: hadamard ----------------------------------------------
template <int LPR, bool VEC>
__global__ __launch_bounds__(256) void edge_hadamard_fwd_kernel(const float* __restrict__ h, int64_t ldh,
                                                                int64_t n_rows, const int64_t* __restrict__ src,
                                                                const int64_t* __restrict__ dst, int64_t n_edges,
                                                                int feat, float* __restrict__ out, int64_t ldo) {
    constexpr int GPB = 256 / LPR;
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    for (int64_t e = (int64_t)blockIdx.x * GPB + grp; e < n_edges; e += (int64_t)gridDim.x * GPB) {
        const float* a = h + wrap_index(src[e], n_rows) * ldh;
        const float* b = h + wrap_index(dst[e], n_rows) * ldh;
        float* o = out + e * ldo;
        if constexpr (VEC) {
            const int nslots = feat >> 2;
            for (int s = sub; s < nslots; s += LPR) {
                const float4 u = reinterpret_cast<const float4*>(a)[s];
                const float4 v = reinterpret_cast<const float4*>(b)[s];
                reinterpret_cast<float4*>(o)[s] = make_float4(u.x * v.x, u.y * v.y, u.z * v.z, u.w * v.w);
            }
        } else {
            for (int f = sub; f < feat; f += LPR) o[f] = a[f] * b[f];
        }
    }
}

// A table beyond one XCD's L2 (ddi: 4 267 x 2 KB = 8.7 MB against 4 MB) whose rows every edge gathers: the columns in EIGHT slabs,
// workgroup b takes slab b % 8 (consecutive workgroups run on consecutive XCDs), so an XCD's L2 holds one slab of the table and the
// gathers stay in it -- what goes over the fabric is the result.  SL float4 slots per slab (feat == 32 SL); same values.
template <int SL>
__global__ __launch_bounds__(256) void edge_hadamard_fwd_slab_kernel(const float* __restrict__ h, int64_t ldh, int64_t n_rows,
                                                                     const int64_t* __restrict__ src,
                                                                     const int64_t* __restrict__ dst, int64_t n_edges,
                                                                     float* __restrict__ out, int64_t ldo) {
    constexpr int GPB = 256 / SL;
    const int slab = blockIdx.x & 7;
    const int64_t wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    const int sub = threadIdx.x % SL, grp = threadIdx.x / SL;
    const int col = (slab * SL + sub) * 4;
    for (int64_t e = wg * GPB + grp; e < n_edges; e += nwg * GPB) {
        const float4 u = *reinterpret_cast<const float4*>(h + wrap_index(src[e], n_rows) * ldh + col);
        const float4 v = *reinterpret_cast<const float4*>(h + wrap_index(dst[e], n_rows) * ldh + col);
        *reinterpret_cast<float4*>(out + e * ldo + col) = make_float4(u.x * v.x, u.y * v.y, u.z * v.z, u.w * v.w);
    }
}

// ---------------- backward: atomic scatter -----------------------------------------
template <int LPR, bool GVEC>
__global__ __launch_bounds__(256) void edge_scatter_bwd_kernel(const float* __restrict__ h, int64_t ldh,
                                                               const int64_t* __restrict__ src,
                                                               const int64_t* __restrict__ dst, int64_t n_edges,
                                                               int feat, const float* __restrict__ g, int64_t ldg,
                                                               float* __restrict__ gh, int64_t ldgh) {
    constexpr int GPB = 256 / LPR;
    const int sub = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    for (int64_t e = (int64_t)blockIdx.x * GPB + grp; e < n_edges; e += (int64_t)gridDim.x * GPB) {
        const int64_t si = src[e], di = dst[e];
        const float* a = h + si * ldh;
        const float* b = h + di * ldh;
        float* ga = gh + si * ldgh;
        float* gb = gh + di * ldgh;
        const float gs = GVEC ? 0.f : g[e];
        for (int f = sub; f < feat; f += LPR) {
            const float gv = GVEC ? g[e * ldg + f] : gs;
            unsafeAtomicAdd(ga + f, gv * b[f]);
            unsafeAtomicAdd(gb + f, gv * a[f]);
        }
    }
}

// ---------------- backward: deterministic segmented gather-reduce -----------------
// one wave per node slot; items are (edge id, other endpoint).
// NK: float4 slots per lane and pass (a pass covers 256 NK columns).  Rows of up to 256 floats take NK = 1: the launch is a chain
// of dependent round trips per segment (node -> segment -> items -> rows), bound by waves in flight, and the second slot's
// registers (x, gg: 32 VGPRs that hold zeros at F = 200) cost the vector-gradient form three of its eight waves per SIMD.
template <bool GVEC, bool VEC, int NK = 2>
__global__ __launch_bounds__(256) void edge_segment_bwd_kernel(
    const float* __restrict__ h, int64_t ldh, const int64_t* __restrict__ seg_ptr,
    const int64_t* __restrict__ seg_node, int64_t n_seg, const int32_t* __restrict__ item_edge,
    const int32_t* __restrict__ item_other, int feat, const float* __restrict__ g, int64_t ldg,
    float* __restrict__ gh, int64_t ldgh, Epi epi) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t s = (int64_t)blockIdx.x * 4 + wave;
    if (s >= n_seg) return;
    const int64_t node = seg_node ? seg_node[s] : s;
    const int64_t beg = seg_ptr[s], end = seg_ptr[s + 1];
    float* orow = gh + node * ldgh;
    if constexpr (VEC) {
        const int nslots = feat >> 2;
        for (int s0 = 0; s0 < nslots; s0 += 64 * NK) {
            float4 acc[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t i0 = beg; i0 < end; i0 += 64) {
                const int n = (int)((end - i0) < 64 ? (end - i0) : 64);
                int ev = 0, ov = 0;
                float gv = 0.f;
                if (lane < n) {
                    ev = item_edge[i0 + lane];
                    ov = item_other[i0 + lane];
                    if constexpr (!GVEC) gv = g[ev];
                }
                for (int j = 0; j < n; j += 4) {
                    float4 x[4][NK], gg[4][NK];
                    float w[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (j + u < n) {
                            const int o = __builtin_amdgcn_readlane(ov, j + u);
                            const float4* p = reinterpret_cast<const float4*>(h + (int64_t)o * ldh);
                            if constexpr (GVEC) {
                                const int ed = __builtin_amdgcn_readlane(ev, j + u);
                                const float4* q = reinterpret_cast<const float4*>(g + (int64_t)ed * ldg);
#pragma unroll
                                for (int k = 0; k < NK; ++k) {
                                    const int sl = s0 + lane + 64 * k;
                                    gg[u][k] = sl < nslots ? q[sl] : make_float4(0.f, 0.f, 0.f, 0.f);
                                }
                            } else {
                                w[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gv), j + u));
                            }
#pragma unroll
                            for (int k = 0; k < NK; ++k) {
                                const int sl = s0 + lane + 64 * k;
                                x[u][k] = sl < nslots ? p[sl] : make_float4(0.f, 0.f, 0.f, 0.f);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (j + u < n) {
#pragma unroll
                            for (int k = 0; k < NK; ++k) {
                                if constexpr (GVEC) {
                                    acc[k].x = fmaf(gg[u][k].x, x[u][k].x, acc[k].x);
                                    acc[k].y = fmaf(gg[u][k].y, x[u][k].y, acc[k].y);
                                    acc[k].z = fmaf(gg[u][k].z, x[u][k].z, acc[k].z);
                                    acc[k].w = fmaf(gg[u][k].w, x[u][k].w, acc[k].w);
                                } else {
                                    acc[k].x = fmaf(w[u], x[u][k].x, acc[k].x);
                                    acc[k].y = fmaf(w[u], x[u][k].y, acc[k].y);
                                    acc[k].z = fmaf(w[u], x[u][k].z, acc[k].z);
                                    acc[k].w = fmaf(w[u], x[u][k].w, acc[k].w);
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int sl = s0 + lane + 64 * k;
                if (sl < nslots) {
                    const float4 y = epi_apply4(epi, acc[k], node, (int64_t)sl * 4, feat, orow);
                    reinterpret_cast<float4*>(orow)[sl] = y;
                }
            }
        }
    } else {
        for (int f = lane; f < feat; f += 64) {
            float acc = 0.f;
            for (int64_t i = beg; i < end; ++i) {
                const int ed = item_edge[i], o = item_other[i];
                const float gv = GVEC ? g[(int64_t)ed * ldg + f] : g[ed];
                acc = fmaf(gv, h[(int64_t)o * ldh + f], acc);
            }
            const float prev = (epi.flags & PLNLP_EPI_ACCUM) ? orow[f] : 0.f;
            orow[f] = epi_apply(epi, acc, node, f, feat, prev);
        }
    }
}

// The vector path again, with the work of a LONG segment shared by the waves of a workgroup.  One wave per segment (above) makes
// the launch as long as its longest segment: the hub's items are a serial chain of batches of four rows (citation2's step: 263 K
// touched nodes with two items each -- 0.18 ms when no segment is long -- and one with several hundred: 0.61 ms; ddi: 4 267 nodes,
// 123 items on average, 250+ at the hubs).
// A workgroup of W waves takes S consecutive segments (S = W: many segments; S = 1: fewer segments than the chip holds waves):
//   phase A (S = W): wave w sums segment w alone if it has at most LONG items -- item order, the bits of the kernel above;
//   phase B: every longer segment of the group, one after the other, by ALL W waves: contiguous W-ths of its items in order,
//     the partial sums added in wave order through LDS by wave 0 -- a fixed association that depends on
//     the segment's length only.  S = 1: every segment goes this way.
template <bool GVEC, int NK, int U>
__device__ __forceinline__ void segment_sum(float4 (&acc)[NK], int64_t lo, int64_t hi, int s0, int nslots, int lane,
                                            const float* __restrict__ h, int64_t ldh, const int32_t* __restrict__ item_edge,
                                            const int32_t* __restrict__ item_other, const float* __restrict__ g, int64_t ldg) {
    for (int64_t i0 = lo; i0 < hi; i0 += 64) {
        const int n = (int)((hi - i0) < 64 ? (hi - i0) : 64);
        int ev = 0, ov = 0;
        float gv = 0.f;
        if (lane < n) {
            ev = item_edge[i0 + lane];
            ov = item_other[i0 + lane];
            if constexpr (!GVEC) gv = g[ev];
        }
        for (int j = 0; j < n; j += U) {
            float4 x[U][NK], gg[U][NK];
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j + u < n) {
                    const int o = __builtin_amdgcn_readlane(ov, j + u);
                    const float4* p = reinterpret_cast<const float4*>(h + (int64_t)o * ldh);
                    if constexpr (GVEC) {
                        const int ed = __builtin_amdgcn_readlane(ev, j + u);
                        const float4* q = reinterpret_cast<const float4*>(g + (int64_t)ed * ldg);
#pragma unroll
                        for (int c = 0; c < NK; ++c) {
                            const int sl = s0 + lane + 64 * c;
                            gg[u][c] = sl < nslots ? q[sl] : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    } else {
                        w[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gv), j + u));
                    }
#pragma unroll
                    for (int c = 0; c < NK; ++c) {
                        const int sl = s0 + lane + 64 * c;
                        x[u][c] = sl < nslots ? p[sl] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j + u < n) {
#pragma unroll
                    for (int c = 0; c < NK; ++c) {
                        if constexpr (GVEC) {
                            acc[c].x = fmaf(gg[u][c].x, x[u][c].x, acc[c].x); acc[c].y = fmaf(gg[u][c].y, x[u][c].y, acc[c].y);
                            acc[c].z = fmaf(gg[u][c].z, x[u][c].z, acc[c].z); acc[c].w = fmaf(gg[u][c].w, x[u][c].w, acc[c].w);
                        } else {
                            acc[c].x = fmaf(w[u], x[u][c].x, acc[c].x); acc[c].y = fmaf(w[u], x[u][c].y, acc[c].y);
                            acc[c].z = fmaf(w[u], x[u][c].z, acc[c].z); acc[c].w = fmaf(w[u], x[u][c].w, acc[c].w);
                        }
                    }
                }
            }
        }
    }
}

template <int NK>
__device__ __forceinline__ void segment_store(const float4 (&acc)[NK], int64_t node, int s0, int nslots, int lane, int feat,
                                              float* __restrict__ gh, int64_t ldgh, const Epi& epi) {
    float* orow = gh + node * ldgh;
#pragma unroll
    for (int c = 0; c < NK; ++c) {
        const int sl = s0 + lane + 64 * c;
        if (sl < nslots) {
            const float4 y = epi_apply4(epi, acc[c], node, (int64_t)sl * 4, feat, orow);
            reinterpret_cast<float4*>(orow)[sl] = y;
        }
    }
}

constexpr int SEGMENT_LONG = 64;            // items from which a segment is shared by the workgroup (S = W)

template <bool GVEC, int NK, int W, int S>
__global__ __launch_bounds__(64 * W) void edge_segment_bwd_group_kernel(
    const float* __restrict__ h, int64_t ldh, const int64_t* __restrict__ seg_ptr,
    const int64_t* __restrict__ seg_node, int64_t n_seg, const int32_t* __restrict__ item_edge,
    const int32_t* __restrict__ item_other, int feat, const float* __restrict__ g, int64_t ldg,
    float* __restrict__ gh, int64_t ldgh, Epi epi) {
    static_assert(S == 1 || S == W, "one segment per workgroup, or one per wave");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t first = (int64_t)blockIdx.x * S;
    if (first >= n_seg) return;                                  // (the whole workgroup)
    const int64_t long_len = S == 1 ? -1 : SEGMENT_LONG;
    // the group's S + 1 bounds: one load, lane k holds bound k (clamped at the end of the list)
    const int64_t mine = seg_ptr[first + lane < n_seg ? first + lane : n_seg];
    auto bound = [&](int k) -> int64_t {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)mine, k);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)mine >> 32), k);
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    __shared__ float4 part[(W - 1) * NK * 64];
    const int nslots = feat >> 2;
    for (int s0 = 0; s0 < nslots; s0 += 64 * NK) {
        // t = -1: phase A (this wave's own segment, if short); t = 0 .. S-1: phase B (segment t of the group, if long, by the
        // whole workgroup).  One loop so that both phases share the registers of one copy of the summation.
        for (int t = S > 1 ? -1 : 0; t < S; ++t) {
            const bool coop = t >= 0;
            const int k = coop ? t : wave;
            const int64_t beg = bound(k), end = bound(k + 1);
            const bool is_long = end - beg > long_len;
            if (first + k >= n_seg || is_long != coop) continue;    // (phase B: uniform over the workgroup)
            int64_t lo = beg, hi = end;
            if (coop) {
                const int64_t q = (end - beg + W - 1) / W;
                lo = beg + wave * q < end ? beg + wave * q : end;
                hi = lo + q < end ? lo + q : end;
            }
            float4 acc[NK];
#pragma unroll
            for (int c = 0; c < NK; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            segment_sum<GVEC, NK, (S == 1 && NK == 1 ? 8 : 4)>(acc, lo, hi, s0, nslots, lane, h, ldh, item_edge, item_other, g, ldg);
            if (coop) {
                if (wave > 0) {
#pragma unroll
                    for (int c = 0; c < NK; ++c) part[((wave - 1) * NK + c) * 64 + lane] = acc[c];
                }
                __syncthreads();
                if (wave == 0) {
#pragma unroll
                    for (int wv = 1; wv < W; ++wv)
#pragma unroll
                        for (int c = 0; c < NK; ++c) {
                            const float4 v = part[((wv - 1) * NK + c) * 64 + lane];
                            acc[c].x += v.x; acc[c].y += v.y; acc[c].z += v.z; acc[c].w += v.w;
                        }
                }
            }
            if (!coop || wave == 0)
                segment_store<NK>(acc, seg_node ? seg_node[first + k] : first + k, s0, nslots, lane, feat, gh, ldgh, epi);
            if (coop) __syncthreads();                              // (the partial sums have been read)
        }
    }
}

// Few segments over a table that does not fit ONE XCD's L2 (ddi: 4 267 rows x 2 KB = 8.7 MB against 4 MB): in the forms above every
// row of h comes over the fabric like the rows of g -- 1.9 GB per launch where g alone is 1.07 GB, at the fabric's 7 TB/s.  Here the
// feature columns are cut into EIGHT slabs and workgroup b takes slab b % 8 of segment b / 8: consecutive workgroups go to
// consecutive XCDs, so every XCD reads one slab of h only (1.1 MB: L2-resident) and its eighth of every g row.  A slab is SL float4
// slots wide; the 64 / SL lane groups of the workgroup's four waves take contiguous shares of the segment's items in order, and the
// first SL lanes add the groups' partial sums in group order through LDS (a fixed association).
template <bool GVEC, int SL>
__global__ __launch_bounds__(256) void edge_segment_bwd_slab_kernel(
    const float* __restrict__ h, int64_t ldh, const int64_t* __restrict__ seg_ptr,
    const int64_t* __restrict__ seg_node, int64_t n_seg, const int32_t* __restrict__ item_edge,
    const int32_t* __restrict__ item_other, int feat, const float* __restrict__ g, int64_t ldg,
    float* __restrict__ gh, int64_t ldgh, Epi epi) {
    constexpr int GPW = 64 / SL;                                 // lane groups per wave
    constexpr int NG = 4 * GPW;                                  // ... per workgroup
    constexpr int U = 4;
    const int slab = blockIdx.x & 7;
    const int64_t seg = (int64_t)(blockIdx.x >> 3);
    if (seg >= n_seg) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % SL, grp = wave * GPW + lane / SL;
    const int64_t beg = seg_ptr[seg], end = seg_ptr[seg + 1];
    const int64_t q = (end - beg + NG - 1) / NG;
    const int64_t lo = beg + grp * q < end ? beg + grp * q : end;
    const int64_t hi = lo + q < end ? lo + q : end;
    const int col = (slab * SL + sub) * 4;                       // (feat == 32 SL: the launcher checks)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = lo; i < hi; i += U) {
        float4 x[U], gg[U];
        float w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i + u < hi) {
                const int ed = item_edge[i + u], o = item_other[i + u];
                x[u] = *reinterpret_cast<const float4*>(h + (int64_t)o * ldh + col);
                if constexpr (GVEC) gg[u] = *reinterpret_cast<const float4*>(g + (int64_t)ed * ldg + col);
                else w[u] = g[ed];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i + u < hi) {
                if constexpr (GVEC) {
                    acc.x = fmaf(gg[u].x, x[u].x, acc.x); acc.y = fmaf(gg[u].y, x[u].y, acc.y);
                    acc.z = fmaf(gg[u].z, x[u].z, acc.z); acc.w = fmaf(gg[u].w, x[u].w, acc.w);
                } else {
                    acc.x = fmaf(w[u], x[u].x, acc.x); acc.y = fmaf(w[u], x[u].y, acc.y);
                    acc.z = fmaf(w[u], x[u].z, acc.z); acc.w = fmaf(w[u], x[u].w, acc.w);
                }
            }
        }
    }
    __shared__ float4 part[NG * SL];
    part[grp * SL + sub] = acc;
    __syncthreads();
    if (threadIdx.x < SL) {
        float4 t = part[sub];
#pragma unroll
        for (int k = 1; k < NG; ++k) {
            const float4 v = part[k * SL + sub];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        const int64_t node = seg_node ? seg_node[seg] : seg;
        float* orow = gh + node * ldgh;
        const float4 y = epi_apply4(epi, t, node, (int64_t)col, feat, orow);
        *reinterpret_cast<float4*>(orow + col) = y;
    }
}

// measurement knob (plnlp_edge_segment_tuning): 0 = long segments shared by a workgroup (the rule in plnlp_edge_segment_bwd_f32),
// 1 = one wave per segment always (the round-5 form), 2 = many segments in groups of four waves / four segments (measured on
// citation2's step: 273 us against 250 for eight / eight, 613 for one wave per segment), 3 = few segments without the XCD-pinned
// column slabs
static int g_segment_form = 0;

static inline int pick_lpr(int64_t feat, bool vec) {
    const int64_t units = vec ? feat / 4 : feat;
    if (units <= 8) return 8;
    if (units <= 16) return 16;
    if (units <= 32) return 32;
    return 64;
}
static inline unsigned edge_grid(int64_t n_edges, int lpr) {
    const int64_t gpb = 256 / lpr;
    int64_t b = (n_edges + gpb - 1) / gpb;
    const int64_t cap = 256 * 16;
    return (unsigned)(b < cap ? (b > 0 ? b : 1) : cap);
}

}  // namespace plnlp

#define PLNLP_DISPATCH_LPR(lpr, MACRO) \
    switch (lpr) { case 8: MACRO(8); break; case 16: MACRO(16); break; case 32: MACRO(32); break; default: MACRO(64); }

extern "C" int plnlp_edge_dot_fwd_f32(const float* h, int64_t ldh, int64_t n_rows, const int64_t* src,
                                      const int64_t* dst, int64_t n_edges, int64_t feat, float* out,
                                      void* stream) {
    using namespace plnlp;
    if (n_edges < 0 || feat <= 0 || ldh < feat || n_rows <= 0 || feat > (1 << 20)) return PLNLP_E_SHAPE;
    if (n_edges == 0) return 0;
    if (!h || !out) return PLNLP_E_NULL;
    if (!src || !dst) return PLNLP_E_NULL;
    const bool vec = feat % 4 == 0 && ldh % 4 == 0 && (uintptr_t)h % 16 == 0;
    const int lpr = pick_lpr(feat, vec);
    dim3 grid(edge_grid(n_edges, lpr));
    hipStream_t s = (hipStream_t)stream;
#define M(L)                                                                                                   \
    if (vec) hipLaunchKernelGGL((edge_dot_fwd_kernel<L, true>), grid, dim3(256), 0, s, h, ldh, n_rows, src, dst, \
                                n_edges, (int)feat, out);                                                      \
    else hipLaunchKernelGGL((edge_dot_fwd_kernel<L, false>), grid, dim3(256), 0, s, h, ldh, n_rows, src, dst,   \
                            n_edges, (int)feat, out)
    PLNLP_DISPATCH_LPR(lpr, M)
#undef M
    return launch_status();
}

extern "C" int plnlp_edge_hadamard_fwd_f32(const float* h, int64_t ldh, int64_t n_rows, const int64_t* src,
                                           const int64_t* dst, int64_t n_edges, int64_t feat, float* out,
                                           int64_t ldo, void* stream) {
    using namespace plnlp;
    if (n_edges < 0 || feat <= 0 || ldh < feat || ldo < feat || n_rows <= 0 || feat > (1 << 20)) return PLNLP_E_SHAPE;
    if (n_edges == 0) return 0;
    if (!h || !out) return PLNLP_E_NULL;
    if (!src || !dst) return PLNLP_E_NULL;
    const bool vec = feat % 4 == 0 && ldh % 4 == 0 && ldo % 4 == 0 && (uintptr_t)h % 16 == 0 &&
                     (uintptr_t)out % 16 == 0;
    const int lpr = pick_lpr(feat, vec);
    dim3 grid(edge_grid(n_edges, lpr));
    hipStream_t s = (hipStream_t)stream;
    // the table beyond one XCD's L2 (4 MB) but an eighth of it well inside: XCD-pinned column slabs
    if (vec && g_segment_form != 3 && (feat == 512 || feat == 256) && n_rows * feat * 4 > (4ll << 20) &&
        n_rows * feat * 4 <= (24ll << 20) && n_edges >= 4096) {
        const int gpb = feat == 512 ? 16 : 32;
        int64_t nwg = (n_edges + gpb - 1) / gpb;
        nwg = nwg < 2048 ? nwg : 2048;
        const dim3 g8((unsigned)(nwg * 8));
        if (feat == 512) hipLaunchKernelGGL((edge_hadamard_fwd_slab_kernel<16>), g8, dim3(256), 0, s, h, ldh, n_rows, src, dst, n_edges,
                                            out, ldo);
        else hipLaunchKernelGGL((edge_hadamard_fwd_slab_kernel<8>), g8, dim3(256), 0, s, h, ldh, n_rows, src, dst, n_edges, out, ldo);
        return launch_status();
    }
#define M(L)                                                                                                     \
    if (vec) hipLaunchKernelGGL((edge_hadamard_fwd_kernel<L, true>), grid, dim3(256), 0, s, h, ldh, n_rows, src,  \
                                dst, n_edges, (int)feat, out, ldo);                                              \
    else hipLaunchKernelGGL((edge_hadamard_fwd_kernel<L, false>), grid, dim3(256), 0, s, h, ldh, n_rows, src,     \
                            dst, n_edges, (int)feat, out, ldo)
    PLNLP_DISPATCH_LPR(lpr, M)
#undef M
    return launch_status();
}

extern "C" int plnlp_edge_scatter_bwd_f32(const float* h, int64_t ldh, const int64_t* src, const int64_t* dst,
                                          int64_t n_edges, int64_t feat, const float* g, int64_t ldg,
                                          int g_is_vector, float* gh, int64_t ldgh, void* stream) {
    using namespace plnlp;
    if (n_edges < 0 || feat <= 0 || ldh < feat || ldgh < feat || (g_is_vector && ldg < feat)) return PLNLP_E_SHAPE;
    if (n_edges == 0) return 0;
    if (!h || !g || !gh) return PLNLP_E_NULL;
    if (!src || !dst) return PLNLP_E_NULL;
    const int lpr = pick_lpr(feat, false);
    dim3 grid(edge_grid(n_edges, lpr));
    hipStream_t s = (hipStream_t)stream;
#define M(L)                                                                                                    \
    if (g_is_vector) hipLaunchKernelGGL((edge_scatter_bwd_kernel<L, true>), grid, dim3(256), 0, s, h, ldh, src,  \
                                        dst, n_edges, (int)feat, g, ldg, gh, ldgh);                              \
    else hipLaunchKernelGGL((edge_scatter_bwd_kernel<L, false>), grid, dim3(256), 0, s, h, ldh, src, dst,        \
                            n_edges, (int)feat, g, ldg, gh, ldgh)
    PLNLP_DISPATCH_LPR(lpr, M)
#undef M
    return launch_status();
}

extern "C" void plnlp_edge_segment_tuning(int form) { plnlp::g_segment_form = form; }

extern "C" int plnlp_edge_segment_bwd_f32(const float* h, int64_t ldh, const int64_t* seg_ptr,
                                          const int64_t* seg_node, int64_t n_seg, const int32_t* item_edge,
                                          const int32_t* item_other, int64_t feat, const float* g, int64_t ldg,
                                          int g_is_vector, float* gh, int64_t ldgh, const plnlp_epilogue* epi,
                                          void* stream) {
    using namespace plnlp;
    if (n_seg < 0 || feat <= 0 || ldh < feat || ldgh < feat || (g_is_vector && ldg < feat) || feat > (1 << 20))
        return PLNLP_E_SHAPE;
    if (n_seg == 0) return 0;
    if (!h || !g || !gh || !seg_ptr) return PLNLP_E_NULL;
    if (!item_edge || !item_other) return PLNLP_E_NULL;
    Epi e;
    if (int rc = make_epi(epi, &e)) return rc;
    const bool vec = feat % 4 == 0 && ldh % 4 == 0 && ldgh % 4 == 0 && (uintptr_t)h % 16 == 0 &&
                     (uintptr_t)gh % 16 == 0 && (!g_is_vector || (ldg % 4 == 0 && (uintptr_t)g % 16 == 0));
    dim3 grid((unsigned)((n_seg + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    if (vec && g_segment_form != 1) {
        // fewer segments than the chip holds waves: a workgroup of four waves per segment; many segments: eight per workgroup of
        // eight waves, its long ones (> SEGMENT_LONG items) shared by the eight
        const bool few = n_seg < 8192;
        // few segments, a table beyond one XCD's L2 (4 MB), rows of 256 / 512 floats: column slabs pinned to the XCDs
        if (few && g_segment_form != 3 && n_seg * feat * 4 > (4ll << 20) && (feat == 512 || feat == 256)) {
            const dim3 g3((unsigned)(n_seg * 8));
#define V3(GV, SL)                                                                                                             \
    hipLaunchKernelGGL((edge_segment_bwd_slab_kernel<GV, SL>), g3, dim3(256), 0, s, h, ldh, seg_ptr, seg_node, n_seg, item_edge, \
                       item_other, (int)feat, g, ldg, gh, ldgh, e)
            if (g_is_vector) { if (feat == 512) V3(true, 16); else V3(true, 8); }
            else             { if (feat == 512) V3(false, 16); else V3(false, 8); }
#undef V3
            return launch_status();
        }
#define V2(GV, NK)                                                                                                            \
    do {                                                                                                                      \
        if (few) hipLaunchKernelGGL((edge_segment_bwd_group_kernel<GV, NK, 4, 1>), dim3((unsigned)n_seg), dim3(256), 0, s, h, ldh, \
                                    seg_ptr, seg_node, n_seg, item_edge, item_other, (int)feat, g, ldg, gh, ldgh, e);         \
        else if (g_segment_form == 2)                                                                                         \
            hipLaunchKernelGGL((edge_segment_bwd_group_kernel<GV, NK, 4, 4>), dim3((unsigned)((n_seg + 3) / 4)), dim3(256), 0, s,  \
                               h, ldh, seg_ptr, seg_node, n_seg, item_edge, item_other, (int)feat, g, ldg, gh, ldgh, e);      \
        else hipLaunchKernelGGL((edge_segment_bwd_group_kernel<GV, NK, 8, 8>), dim3((unsigned)((n_seg + 7) / 8)), dim3(512), 0, s, \
                                h, ldh, seg_ptr, seg_node, n_seg, item_edge, item_other, (int)feat, g, ldg, gh, ldgh, e);     \
    } while (0)
        if (g_is_vector) { if (feat <= 256) V2(true, 1); else V2(true, 2); }
        else             { if (feat <= 256) V2(false, 1); else V2(false, 2); }
#undef V2
        return launch_status();
    }
#define L(GV, V, NK)                                                                                                  \
    hipLaunchKernelGGL((edge_segment_bwd_kernel<GV, V, NK>), grid, dim3(256), 0, s, h, ldh, seg_ptr, seg_node, n_seg, \
                       item_edge, item_other, (int)feat, g, ldg, gh, ldgh, e)
    // (wider rows keep two slots and one pass: at F = 512 -- ddi, long segments, bandwidth-bound -- one slot and two passes
    // measured the same 0.29 ms)
    const bool narrow = feat <= 256;
    if (g_is_vector) { if (vec) { if (narrow) L(true, true, 1); else L(true, true, 2); } else L(true, false, 2); }
    else             { if (vec) { if (narrow) L(false, true, 1); else L(false, true, 2); } else L(false, false, 2); }
#undef L
    return launch_status();
}
