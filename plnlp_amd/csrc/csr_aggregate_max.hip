// K1 with reduce = max, and its backward -- torch_sparse.matmul(adj_t, x, reduce='max') semantics:
//
//   out[r, f] = max_{e in row r} w_e * x[col[e], f]          (0 for a row without entries)
//   arg[r, f] = row-relative position of the FIRST entry attaining the maximum (-1 for an empty row)
//
// backward (gradient goes to the arg-max source only):
//   gx[j, f]  = sum_{e: col[e] = j} [arg[row(e), f] == pos(e)] * w_e * gy[row(e), f]
// evaluated as a GATHER over the transposed CSR (row j lists the rows i it feeds, with the position of
// that entry in row i), so there are no atomics and the sum order is the transposed CSR order:
// bit-reproducible like the sum/mean kernels.
//
// One wave64 per output row, 16 bytes per lane per neighbour row, LPR lanes per row so that narrow
// feature rows pack 64/LPR neighbours into one wave instruction; the lane groups' candidates are folded
// with a wavefront-shuffle (max, arg) tree.  Rows longer than the split threshold go through the same
// chunk / finalize passes as the sum kernel (plnlp_row_split), carrying (max, arg) pairs.
#include "common.hip.h"

namespace plnlp {

__device__ __forceinline__ void take_max(float& best, int& arg, float v, int a) {
    if (v > best) { best = v; arg = a; }           // strict: the first maximal entry wins
}
// candidate (v, a) from another lane group / chunk: larger value, or equal value at an earlier position
__device__ __forceinline__ void merge_max(float& best, int& arg, float v, int a) {
    if (a >= 0 && (arg < 0 || v > best || (v == best && a < arg))) { best = v; arg = a; }
}

template <int VPL, int LPR, bool WEIGHTED>
__device__ __forceinline__ void max_range(float4 (&best)[VPL], int4 (&arg)[VPL], int64_t row_beg, int64_t beg,
                                          int64_t end, const int32_t* __restrict__ col,
                                          const float* __restrict__ val, const float* __restrict__ x, int64_t ldx,
                                          int lane, int sub, int grp, int nslots) {
    constexpr int NG = 64 / LPR;
    constexpr int CH = (VPL >= 4) ? 2 : 4;
    for (int64_t e0 = beg; e0 < end; e0 += 64) {
        const int n = (int)((end - e0) < 64 ? (end - e0) : 64);
        int cvec = 0;
        float wvec = 1.f;
        if (lane < n) {
            cvec = col[e0 + lane];
            if constexpr (WEIGHTED) wvec = val[e0 + lane];
        }
        for (int j0 = 0; j0 < n; j0 += NG * CH) {
            float4 v[CH][VPL];
            float w[CH];
            bool ok[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int jj = j0 + u * NG + grp;
                ok[u] = jj < n;
                const int idx = __shfl(cvec, ok[u] ? jj : 0, 64);
                w[u] = WEIGHTED ? __shfl(wvec, ok[u] ? jj : 0, 64) : 1.f;
                const float4* p = reinterpret_cast<const float4*>(x + (int64_t)idx * ldx);
#pragma unroll
                for (int k = 0; k < VPL; ++k) {
                    const int s = sub + k * LPR;
                    v[u][k] = (s < nslots) ? p[s] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (!ok[u]) continue;
                const int a = (int)(e0 + j0 + u * NG + grp - row_beg);
#pragma unroll
                for (int k = 0; k < VPL; ++k) {
                    take_max(best[k].x, arg[k].x, w[u] * v[u][k].x, a);
                    take_max(best[k].y, arg[k].y, w[u] * v[u][k].y, a);
                    take_max(best[k].z, arg[k].z, w[u] * v[u][k].z, a);
                    take_max(best[k].w, arg[k].w, w[u] * v[u][k].w, a);
                }
            }
        }
    }
}

template <int VPL, int LPR>
__device__ __forceinline__ void fold_max(float4 (&best)[VPL], int4 (&arg)[VPL]) {
    if constexpr (LPR < 64) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                merge_max(best[k].x, arg[k].x, __shfl_xor(best[k].x, o, 64), __shfl_xor(arg[k].x, o, 64));
                merge_max(best[k].y, arg[k].y, __shfl_xor(best[k].y, o, 64), __shfl_xor(arg[k].y, o, 64));
                merge_max(best[k].z, arg[k].z, __shfl_xor(best[k].z, o, 64), __shfl_xor(arg[k].z, o, 64));
                merge_max(best[k].w, arg[k].w, __shfl_xor(best[k].w, o, 64), __shfl_xor(arg[k].w, o, 64));
            }
        }
    }
}

template <int VPL>
__device__ __forceinline__ void init_max(float4 (&best)[VPL], int4 (&arg)[VPL]) {
    const float ninf = -__builtin_huge_valf();
#pragma unroll
    for (int k = 0; k < VPL; ++k) { best[k] = make_float4(ninf, ninf, ninf, ninf); arg[k] = make_int4(-1, -1, -1, -1); }
}

struct MaxSplit {
    int64_t threshold, n_long, n_chunks;
    const int64_t* long_rows;
    const int64_t* chunk_beg;
    const int32_t* chunk_cnt;
    const int32_t* chunk_long;
    float* ws;          // [n_chunks, feat]
    int32_t* ws_arg;    // [n_chunks, feat]
};

// main pass: one wave per row; rows longer than skip_above (> 0) are left to the split passes
template <int VPL, int LPR, bool WEIGHTED>
__global__ __launch_bounds__(256) void csr_max_kernel(const int64_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col,
                                                      const float* __restrict__ val, const float* __restrict__ x,
                                                      int64_t ldx, float* __restrict__ out, int64_t ldo,
                                                      int32_t* __restrict__ arg_out, int64_t lda, int64_t n_rows,
                                                      int feat, int64_t skip_above, int64_t row_base) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t r = row_base + (int64_t)blockIdx.x * 4 + wave;
    if (r >= n_rows) return;
    const int sub = lane % LPR, grp = lane / LPR;
    const int nslots = feat >> 2;
    const int64_t beg = rowptr[r], end = rowptr[r + 1];
    if (skip_above > 0 && end - beg > skip_above) return;
    float4 best[VPL];
    int4 arg[VPL];
    init_max<VPL>(best, arg);
    max_range<VPL, LPR, WEIGHTED>(best, arg, beg, beg, end, col, val, x, ldx, lane, sub, grp, nslots);
    fold_max<VPL, LPR>(best, arg);
    if (LPR < 64 && grp != 0) return;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int s = sub + k * LPR;
        if (s < nslots) {
            float4 y = best[k];
            if (arg[k].x < 0) y.x = 0.f;
            if (arg[k].y < 0) y.y = 0.f;
            if (arg[k].z < 0) y.z = 0.f;
            if (arg[k].w < 0) y.w = 0.f;
            *reinterpret_cast<float4*>(out + r * ldo + s * 4) = y;
            *reinterpret_cast<int4*>(arg_out + r * lda + s * 4) = arg[k];
        }
    }
}

// split pass 1: one wave per chunk of a long row -> (max, arg) partials in the workspace
template <int VPL, int LPR, bool WEIGHTED>
__global__ __launch_bounds__(256) void csr_max_chunk_kernel(const int64_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            const float* __restrict__ val,
                                                            const float* __restrict__ x, int64_t ldx, int feat,
                                                            MaxSplit sp) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t c = (int64_t)blockIdx.x * 4 + wave;
    if (c >= sp.n_chunks) return;
    const int l = sp.chunk_long[c];
    if (l < 0) return;
    const int64_t r = sp.long_rows[l];
    if (r < 0) return;
    const int64_t j = c - sp.chunk_beg[l];
    const int64_t rb = rowptr[r], re = rowptr[r + 1];
    const int64_t beg = rb + j * sp.threshold;
    const int64_t end = (beg + sp.threshold) < re ? (beg + sp.threshold) : re;
    const int sub = lane % LPR, grp = lane / LPR;
    const int nslots = feat >> 2;
    float4 best[VPL];
    int4 arg[VPL];
    init_max<VPL>(best, arg);
    max_range<VPL, LPR, WEIGHTED>(best, arg, rb, beg, end, col, val, x, ldx, lane, sub, grp, nslots);
    fold_max<VPL, LPR>(best, arg);
    if (LPR < 64 && grp != 0) return;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int s = sub + k * LPR;
        if (s < nslots) {
            *reinterpret_cast<float4*>(sp.ws + c * (int64_t)feat + s * 4) = best[k];
            *reinterpret_cast<int4*>(sp.ws_arg + c * (int64_t)feat + s * 4) = arg[k];
        }
    }
}

// split pass 2: one wave per long row merges its chunk partials in chunk order
__global__ __launch_bounds__(64) void csr_max_finalize_kernel(int feat, MaxSplit sp, float* __restrict__ out,
                                                              int64_t ldo, int32_t* __restrict__ arg_out,
                                                              int64_t lda) {
    const int lane = threadIdx.x;
    const int64_t l = blockIdx.x;
    if (l >= sp.n_long) return;
    const int64_t r = sp.long_rows[l];
    if (r < 0) return;
    const int64_t c0 = sp.chunk_beg[l], c1 = c0 + sp.chunk_cnt[l];
    const int nslots = feat >> 2;
    const float ninf = -__builtin_huge_valf();
    for (int s = lane; s < nslots; s += 64) {
        float4 best = make_float4(ninf, ninf, ninf, ninf);
        int4 arg = make_int4(-1, -1, -1, -1);
        for (int64_t c = c0; c < c1; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(sp.ws + c * (int64_t)feat + s * 4);
            const int4 a = *reinterpret_cast<const int4*>(sp.ws_arg + c * (int64_t)feat + s * 4);
            merge_max(best.x, arg.x, v.x, a.x); merge_max(best.y, arg.y, v.y, a.y);
            merge_max(best.z, arg.z, v.z, a.z); merge_max(best.w, arg.w, v.w, a.w);
        }
        if (arg.x < 0) best.x = 0.f;
        if (arg.y < 0) best.y = 0.f;
        if (arg.z < 0) best.z = 0.f;
        if (arg.w < 0) best.w = 0.f;
        *reinterpret_cast<float4*>(out + r * ldo + s * 4) = best;
        *reinterpret_cast<int4*>(arg_out + r * lda + s * 4) = arg;
    }
}

// scalar path: any feat / alignment.  Lane l owns columns l, l+64, ...
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void csr_max_scalar_kernel(const int64_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col,
                                                             const float* __restrict__ val,
                                                             const float* __restrict__ x, int64_t ldx,
                                                             float* __restrict__ out, int64_t ldo,
                                                             int32_t* __restrict__ arg_out, int64_t lda,
                                                             int64_t n_rows, int feat, int64_t row_base) {
    const int lane = threadIdx.x & 63;
    const int64_t r = row_base + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int64_t beg = rowptr[r], end = rowptr[r + 1];
    for (int f = lane; f < feat; f += 64) {
        float best = -__builtin_huge_valf();
        int arg = -1;
        for (int64_t e = beg; e < end; ++e) {
            const float w = WEIGHTED ? val[e] : 1.f;
            take_max(best, arg, w * x[(int64_t)col[e] * ldx + f], (int)(e - beg));
        }
        out[r * ldo + f] = arg < 0 ? 0.f : best;
        arg_out[r * lda + f] = arg;
    }
}

// backward: one wave per SOURCE row j; transposed entries name the fed row i and the position of the
// (i <- j) entry in row i.  Lane l owns float4 slots l, l+64, ... (scalar columns when !VEC).
template <bool VEC, bool WEIGHTED>
__global__ __launch_bounds__(256) void csr_max_bwd_kernel(const int64_t* __restrict__ rowptr_t,
                                                          const int32_t* __restrict__ col_t,
                                                          const int32_t* __restrict__ pos_t,
                                                          const float* __restrict__ val_t,
                                                          const float* __restrict__ gy, int64_t ldg,
                                                          const int32_t* __restrict__ arg, int64_t lda,
                                                          float* __restrict__ gx, int64_t ldgx, int64_t n_src,
                                                          int feat, int64_t row_base) {
    const int lane = threadIdx.x & 63;
    const int64_t j = row_base + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_src) return;
    const int64_t beg = rowptr_t[j], end = rowptr_t[j + 1];
    if constexpr (VEC) {
        const int nslots = feat >> 2;
        for (int s = lane; s < nslots; s += 64) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            int64_t e = beg;
            for (; e + 2 <= end; e += 2) {           // two fed rows in flight
                const int i0 = col_t[e], i1 = col_t[e + 1];
                const int p0 = pos_t[e], p1 = pos_t[e + 1];
                const float w0 = WEIGHTED ? val_t[e] : 1.f, w1 = WEIGHTED ? val_t[e + 1] : 1.f;
                const int4 a0 = *reinterpret_cast<const int4*>(arg + (int64_t)i0 * lda + s * 4);
                const int4 a1 = *reinterpret_cast<const int4*>(arg + (int64_t)i1 * lda + s * 4);
                const float4 g0 = *reinterpret_cast<const float4*>(gy + (int64_t)i0 * ldg + s * 4);
                const float4 g1 = *reinterpret_cast<const float4*>(gy + (int64_t)i1 * ldg + s * 4);
                if (a0.x == p0) acc.x += w0 * g0.x;
                if (a0.y == p0) acc.y += w0 * g0.y;
                if (a0.z == p0) acc.z += w0 * g0.z;
                if (a0.w == p0) acc.w += w0 * g0.w;
                if (a1.x == p1) acc.x += w1 * g1.x;
                if (a1.y == p1) acc.y += w1 * g1.y;
                if (a1.z == p1) acc.z += w1 * g1.z;
                if (a1.w == p1) acc.w += w1 * g1.w;
            }
            for (; e < end; ++e) {
                const int i0 = col_t[e], p0 = pos_t[e];
                const float w0 = WEIGHTED ? val_t[e] : 1.f;
                const int4 a0 = *reinterpret_cast<const int4*>(arg + (int64_t)i0 * lda + s * 4);
                const float4 g0 = *reinterpret_cast<const float4*>(gy + (int64_t)i0 * ldg + s * 4);
                if (a0.x == p0) acc.x += w0 * g0.x;
                if (a0.y == p0) acc.y += w0 * g0.y;
                if (a0.z == p0) acc.z += w0 * g0.z;
                if (a0.w == p0) acc.w += w0 * g0.w;
            }
            *reinterpret_cast<float4*>(gx + j * ldgx + s * 4) = acc;
        }
    } else {
        for (int f = lane; f < feat; f += 64) {
            float acc = 0.f;
            for (int64_t e = beg; e < end; ++e) {
                const int i = col_t[e];
                if (arg[(int64_t)i * lda + f] == pos_t[e]) acc += (WEIGHTED ? val_t[e] : 1.f) * gy[(int64_t)i * ldg + f];
            }
            gx[j * ldgx + f] = acc;
        }
    }
}

template <int VPL, int LPR>
static int launch_max(bool weighted, hipStream_t s, const int64_t* rowptr, const int32_t* col, const float* val,
                      const float* x, int64_t ldx, float* out, int64_t ldo, int32_t* arg, int64_t lda, int64_t n_rows,
                      int feat, const MaxSplit* sp) {
    const int64_t skip = sp ? sp->threshold : 0;
    constexpr int64_t MAX_BLOCKS = (int64_t)1 << 22;
    const int64_t blocks = (n_rows + 3) / 4;
    for (int64_t b0 = 0; b0 < blocks; b0 += MAX_BLOCKS) {
        const dim3 g((unsigned)((blocks - b0) < MAX_BLOCKS ? (blocks - b0) : MAX_BLOCKS));
        if (weighted)
            hipLaunchKernelGGL((csr_max_kernel<VPL, LPR, true>), g, dim3(256), 0, s, rowptr, col, val, x, ldx, out, ldo,
                               arg, lda, n_rows, feat, skip, b0 * 4);
        else
            hipLaunchKernelGGL((csr_max_kernel<VPL, LPR, false>), g, dim3(256), 0, s, rowptr, col, val, x, ldx, out, ldo,
                               arg, lda, n_rows, feat, skip, b0 * 4);
        if (int rc = launch_status()) return rc;
    }
    if (!sp) return 0;
    dim3 cgrid((unsigned)((sp->n_chunks + 3) / 4));
    if (weighted)
        hipLaunchKernelGGL((csr_max_chunk_kernel<VPL, LPR, true>), cgrid, dim3(256), 0, s, rowptr, col, val, x, ldx, feat, *sp);
    else
        hipLaunchKernelGGL((csr_max_chunk_kernel<VPL, LPR, false>), cgrid, dim3(256), 0, s, rowptr, col, val, x, ldx, feat, *sp);
    if (int rc = launch_status()) return rc;
    hipLaunchKernelGGL(csr_max_finalize_kernel, dim3((unsigned)sp->n_long), dim3(64), 0, s, feat, *sp, out, ldo, arg, lda);
    return launch_status();
}

}  // namespace plnlp

extern "C" int plnlp_csr_aggregate_max_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                                           const float* x, int64_t ldx, float* out, int64_t ldo, int32_t* arg,
                                           int64_t ld_arg, int64_t n_rows, int64_t feat,
                                           const plnlp_row_split* split, int32_t* arg_workspace, void* stream) {
    using namespace plnlp;
    if (n_rows < 0 || feat <= 0 || ldx < feat || ldo < feat || ld_arg < feat || feat > (1 << 20)) return PLNLP_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!rowptr || !x || !out || !arg) return PLNLP_E_NULL;
    if (!col) return PLNLP_E_NULL;
    if (n_rows > (int64_t)4 * 0x7FFFFFFF) return PLNLP_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const bool weighted = val != nullptr;
    const bool vec_ok = (feat % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && (ld_arg % 4 == 0) &&
                        ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)arg % 16 == 0) &&
                        feat <= 1024;
    MaxSplit ms{};
    const MaxSplit* sp = nullptr;
    if (split && split->n_long > 0 && split->n_chunks > 0) {
        if (!vec_ok) return PLNLP_E_UNSUPPORTED;
        if (split->threshold < 64) return PLNLP_E_SHAPE;
        if (split->seg_beg) return PLNLP_E_UNSUPPORTED;      // explicit chunks: the sum kernel only
        if (!split->long_rows || !split->chunk_beg || !split->chunk_cnt || !split->chunk_long || !split->workspace ||
            !arg_workspace) return PLNLP_E_NULL;
        if (split->workspace_floats < split->n_chunks * feat) return PLNLP_E_WORKSPACE;
        if ((uintptr_t)split->workspace % 16 != 0 || (uintptr_t)arg_workspace % 16 != 0) return PLNLP_E_ALIGN;
        ms.threshold = split->threshold; ms.n_long = split->n_long; ms.n_chunks = split->n_chunks;
        ms.long_rows = split->long_rows; ms.chunk_beg = split->chunk_beg; ms.chunk_cnt = split->chunk_cnt;
        ms.chunk_long = split->chunk_long; ms.ws = split->workspace; ms.ws_arg = arg_workspace;
        sp = &ms;
    }
    if (!vec_ok) {
        constexpr int64_t MAX_BLOCKS = (int64_t)1 << 22;
        const int64_t blocks = (n_rows + 3) / 4;
        for (int64_t b0 = 0; b0 < blocks; b0 += MAX_BLOCKS) {
            const dim3 g((unsigned)((blocks - b0) < MAX_BLOCKS ? (blocks - b0) : MAX_BLOCKS));
            if (weighted)
                hipLaunchKernelGGL((csr_max_scalar_kernel<true>), g, dim3(256), 0, s, rowptr, col, val, x, ldx, out, ldo,
                                   arg, ld_arg, n_rows, (int)feat, b0 * 4);
            else
                hipLaunchKernelGGL((csr_max_scalar_kernel<false>), g, dim3(256), 0, s, rowptr, col, val, x, ldx, out, ldo,
                                   arg, ld_arg, n_rows, (int)feat, b0 * 4);
            if (int rc = launch_status()) return rc;
        }
        return 0;
    }
    const int nslots = (int)(feat / 4);
#define PLNLP_MAX(VPL, LPR) \
    return launch_max<VPL, LPR>(weighted, s, rowptr, col, val, x, ldx, out, ldo, arg, ld_arg, n_rows, (int)feat, sp)
    if (nslots <= 8) PLNLP_MAX(1, 8);
    if (nslots <= 16) PLNLP_MAX(1, 16);
    if (nslots <= 32) PLNLP_MAX(1, 32);
    if (nslots <= 64) PLNLP_MAX(1, 64);
    if (nslots <= 128) PLNLP_MAX(2, 64);
    PLNLP_MAX(4, 64);
#undef PLNLP_MAX
}

extern "C" int plnlp_csr_aggregate_max_bwd_f32(const int64_t* rowptr_t, const int32_t* col_t, const int32_t* pos_t,
                                               const float* val_t, const float* gy, int64_t ldg, const int32_t* arg,
                                               int64_t ld_arg, float* gx, int64_t ldgx, int64_t n_src, int64_t feat,
                                               void* stream) {
    using namespace plnlp;
    if (n_src < 0 || feat <= 0 || ldg < feat || ld_arg < feat || ldgx < feat || feat > (1 << 20)) return PLNLP_E_SHAPE;
    if (n_src == 0) return 0;
    if (!rowptr_t || !gy || !arg || !gx) return PLNLP_E_NULL;
    if (!col_t || !pos_t) return PLNLP_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    const bool weighted = val_t != nullptr;
    const bool vec_ok = (feat % 4 == 0) && (ldg % 4 == 0) && (ld_arg % 4 == 0) && (ldgx % 4 == 0) &&
                        ((uintptr_t)gy % 16 == 0) && ((uintptr_t)arg % 16 == 0) && ((uintptr_t)gx % 16 == 0);
    constexpr int64_t MAX_BLOCKS = (int64_t)1 << 22;
    const int64_t blocks = (n_src + 3) / 4;
    for (int64_t b0 = 0; b0 < blocks; b0 += MAX_BLOCKS) {
        const dim3 g((unsigned)((blocks - b0) < MAX_BLOCKS ? (blocks - b0) : MAX_BLOCKS));
#define PLNLP_MAXB(V, W) hipLaunchKernelGGL((csr_max_bwd_kernel<V, W>), g, dim3(256), 0, s, rowptr_t, col_t, pos_t, \
                                            val_t, gy, ldg, arg, ld_arg, gx, ldgx, n_src, (int)feat, b0 * 4)
        if (vec_ok) { if (weighted) PLNLP_MAXB(true, true); else PLNLP_MAXB(true, false); }
        else        { if (weighted) PLNLP_MAXB(false, true); else PLNLP_MAXB(false, false); }
#undef PLNLP_MAXB
        if (int rc = launch_status()) return rc;
    }
    return 0;
}
