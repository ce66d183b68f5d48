// K1/K2 -- CSR neighbour gather-and-reduce for gfx950 (MI355X).
//
//   out[r,:] = EPI( red_{e in row r} w_e * x[col[e], :] )
//
// HBM/L2-bound: 0.5 FLOP per byte, so the design is about bytes in flight, not
// math.  One wave64 owns one output row.  A feature row is read with 16 B per
// lane (1 KiB per wave-instruction); the row's column indices arrive 64 at a
// time through ONE coalesced load and are broadcast lane->SGPR with
// v_readlane, so every feature-row address is scalar-base + lane offset.  Up
// to 8 independent row loads are in flight per wave before the first FMA
// (>= 8 KiB per wave, 16 waves per CU -> >= 128 KiB per CU against the ~32 KiB
// per CU that saturates HBM).  No atomics: the reduction order inside a row is
// the CSR order, so results are bit-reproducible run to run.
//
// Narrow feature rows (feat <= 128) pack 2/4/8 neighbours into one wave
// instruction (LPR lanes per row) and fold the partial sums with DPP/bpermute
// shuffles at the end.
#include "common.hip.h"

#define PLNLP_AGG_LDS_BUDGET (152 * 1024)   // bytes of the 160 KiB LDS the staged slab may take

namespace plnlp {

template <int LPR>  // lanes per feature row: 64, 32, 16, 8
struct AggGeom {
    static constexpr int NG = 64 / LPR;  // neighbours per wave instruction
};

// gather up to CH neighbour rows (m of them valid, wave-uniform) then accumulate.
typedef float f32x4n __attribute__((ext_vector_type(4)));
// a gathered feature row is used once by this wave: with NT the load carries the streaming hint (the line
// need not stay in L2 / the Infinity Cache: right when the source matrix is far larger than both)
template <bool NT>
__device__ __forceinline__ float4 load_row16(const float4* p) {
    if constexpr (NT) {
        const f32x4n v = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}

template <int VPL, int LPR, int CH, bool WEIGHTED, bool NT = false>
__device__ __forceinline__ void agg_chunk(float4 (&acc)[VPL], const float* __restrict__ x, int64_t ldx,
                                          int cvec, float wvec, int j0, int m, int sub, int grp,
                                          int nslots) {
    constexpr int NG = 64 / LPR;
    float4 v[CH][VPL];
    float  w[CH];
    // the row loads need only the column indices: issue them all BEFORE touching the weights, whose
    // own gather (val[val_index[e]]) may still be in flight -- it then overlaps the row gather instead
    // of preceding it (one dependent round trip less per row)
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        if (u < m) {  // wave-uniform
            int idx;
            if constexpr (NG == 1) idx = __builtin_amdgcn_readlane(cvec, j0 + u);
            else                   idx = __shfl(cvec, j0 + u * NG + grp, 64);
            const float4* p = reinterpret_cast<const float4*>(x + (int64_t)idx * ldx);
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                int s = sub + k * LPR;
                v[u][k] = (s < nslots) ? load_row16<NT>(p + s) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    if constexpr (WEIGHTED) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (u < m) {
                if constexpr (NG == 1) w[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wvec), j0 + u));
                else                   w[u] = __shfl(wvec, j0 + u * NG + grp, 64);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        if (u < m) {
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                if constexpr (WEIGHTED) {
                    acc[k].x = fmaf(w[u], v[u][k].x, acc[k].x);
                    acc[k].y = fmaf(w[u], v[u][k].y, acc[k].y);
                    acc[k].z = fmaf(w[u], v[u][k].z, acc[k].z);
                    acc[k].w = fmaf(w[u], v[u][k].w, acc[k].w);
                } else {
                    acc[k].x += v[u][k].x; acc[k].y += v[u][k].y;
                    acc[k].z += v[u][k].z; acc[k].w += v[u][k].w;
                }
            }
        }
    }
}

// accumulate edges [beg, end) of one row into acc (wave-cooperative).
// The entries go in batches of 64 (one index per lane).  A batch is a chain of dependent round trips: index ->
// (source map ->) gathered rows; a long row or chunk repeats it per batch.  The NEXT batch's indices (and weight
// positions) are therefore requested before this batch's rows -- they arrive while the rows are in flight -- and the
// weights that hang off the indices (val[val_index[e]], src_scale[col]) are requested at the head of the batch, ahead
// of the rows, so that the wave waits for them with the rows still outstanding: one round trip per batch instead of
// two to four (measured on the collab hub pass, profiles/r03_agg_hub_pass.md).
template <int VPL, int LPR, bool WEIGHTED, int CHX = 0, bool NT = false>
__device__ __forceinline__ void agg_range(float4 (&acc)[VPL], int64_t beg, int64_t end,
                                          const int32_t* __restrict__ col, const float* __restrict__ val,
                                          const int32_t* __restrict__ val_index,
                                          const float* __restrict__ src_scale, const int32_t* __restrict__ src_map, const float* __restrict__ x,
                                          int64_t ldx, int lane, int sub, int grp, int nslots) {
    constexpr int NG = 64 / LPR;
    constexpr int CH = CHX > 0 ? CHX : ((VPL >= 4) ? 4 : 8);     // neighbour rows in flight per wave
    if (beg >= end) return;
    int n = (int)((end - beg) < 64 ? (end - beg) : 64);
    int cvec = 0, vidx = 0;
    float wraw = 1.f;
    if (lane < n) {
        cvec = col[beg + lane];
        if constexpr (WEIGHTED) {
            if (val && val_index) vidx = val_index[beg + lane];
            else if (val) wraw = val[beg + lane];
        }
    }
    for (int64_t e0 = beg; e0 < end; e0 += 64) {
        // ---- the next batch's indices: requested now, consumed at the bottom
        const int64_t e1 = e0 + 64;
        int n_next = 0, c_next = 0, vidx_next = 0;
        float wraw_next = 1.f;
        if (e1 < end) {
            n_next = (int)((end - e1) < 64 ? (end - e1) : 64);
            if (lane < n_next) {
                c_next = col[e1 + lane];
                if constexpr (WEIGHTED) {
                    if (val && val_index) vidx_next = val_index[e1 + lane];
                    else if (val) wraw_next = val[e1 + lane];
                }
            }
        }
        // ---- this batch's weights (they hang off indices that have arrived)
        float wvec = 0.f;
        if constexpr (WEIGHTED) {
            if (lane < n) {
                wvec = (val && val_index) ? val[vidx] : wraw;
                if (src_scale) wvec *= src_scale[cvec];
            }
        }
        bool any = true;
        if (src_map) {   // wave-uniform
            // x holds only the mapped source rows: translate the 64 indices and squeeze the entries
            // without a row to the back with one lane permutation (order of the rest kept, so the
            // sum order -- and every bit of the result -- is unchanged); the loops below then
            // simply see a shorter batch
            const int cm = (lane < n) ? src_map[cvec] : -1;
            const bool ok = cm >= 0;
            const uint64_t mask = __ballot(ok);
            const int before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
            const int nv = __popcll(mask);
            const int dest = ok ? before : nv + (lane - before);
            cvec = __builtin_amdgcn_ds_permute(dest << 2, cm);
            cvec = cvec < 0 ? 0 : cvec;      // lanes past nv: a loadable row (ragged lane groups still issue the load)
            if constexpr (WEIGHTED) wvec = __int_as_float(__builtin_amdgcn_ds_permute(dest << 2, __float_as_int(wvec)));
            n = nv;
            any = n > 0;
        }
        if (any) {
            const int ngroups = (n + NG - 1) / NG;  // wave instructions needed
            for (int j = 0; j < ngroups; j += CH) {
                const int m = (ngroups - j) < CH ? (ngroups - j) : CH;
                if constexpr (NG == 1) {
                    agg_chunk<VPL, LPR, CH, WEIGHTED, NT>(acc, x, ldx, cvec, wvec, j, m, sub, grp, nslots);
                } else {
                    // a lane group whose neighbour index runs past n must contribute nothing
                    // (not even 0*x: x may hold inf): complete groups first, the ragged one under a select.
                    const int full = (n / NG);
                    if (j + m <= full) {
                        agg_chunk<VPL, LPR, CH, WEIGHTED>(acc, x, ldx, cvec, wvec, j * NG, m, sub, grp, nslots);
                    } else {
                        const int mfull = full - j > 0 ? full - j : 0;
                        if (mfull > 0)
                            agg_chunk<VPL, LPR, CH, WEIGHTED>(acc, x, ldx, cvec, wvec, j * NG, mfull, sub, grp, nslots);
                        const int jl = (j + mfull) * NG;  // first neighbour of the ragged group
                        if (jl < n) {
                            float4 part[VPL];
#pragma unroll
                            for (int k = 0; k < VPL; ++k) part[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                            agg_chunk<VPL, LPR, 1, WEIGHTED>(part, x, ldx, cvec, wvec, jl, 1, sub, grp, nslots);
                            const bool ok = (jl + grp) < n;
#pragma unroll
                            for (int k = 0; k < VPL; ++k) {
                                acc[k].x += ok ? part[k].x : 0.f; acc[k].y += ok ? part[k].y : 0.f;
                                acc[k].z += ok ? part[k].z : 0.f; acc[k].w += ok ? part[k].w : 0.f;
                            }
                        }
                    }
                }
            }
        }
        n = n_next; cvec = c_next; vidx = vidx_next; wraw = wraw_next;
    }
}

template <int VPL, int LPR>
__device__ __forceinline__ void fold_groups(float4 (&acc)[VPL]) {
    if constexpr (LPR < 64) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                acc[k].x += __shfl_xor(acc[k].x, o, 64); acc[k].y += __shfl_xor(acc[k].y, o, 64);
                acc[k].z += __shfl_xor(acc[k].z, o, 64); acc[k].w += __shfl_xor(acc[k].w, o, 64);
            }
        }
    }
}

// Epilogue operands that do not depend on the row's sum (gate row, accumulate row, addend row):
// a short row is one chain of dependent round trips (rowptr -> col -> rows -> store), and loading
// these only after the sum appended two more (index -> operand row).  They are fetched at the START of
// the row instead and ride along with the index loads.
template <int VPL>
struct EpiPre {
    float4 g[VPL], p[VPL], a[VPL];
    bool   valid, has_add;
};

template <int VPL, int LPR>
__device__ __forceinline__ void epi_prefetch(EpiPre<VPL>& pre, const Epi& epi, int64_t r, int nslots, int sub,
                                             const float* __restrict__ out, int64_t ldo) {
    pre.valid = epi.vec4 && (epi.flags & (PLNLP_EPI_GATE | PLNLP_EPI_ACCUM | PLNLP_EPI_ADDEND)) && VPL <= 2;
    pre.has_add = false;
    if (!pre.valid) return;
    const float* grow = nullptr;
    const float* arow = nullptr;
    if (epi.flags & PLNLP_EPI_GATE)
        grow = epi.gate + (epi.gate_index ? (int64_t)epi.gate_index[r] : r) * epi.ld_gate;
    if (epi.flags & PLNLP_EPI_ADDEND) {
        const int64_t ai = epi.addend_index ? (int64_t)epi.addend_index[r] : r;
        if (ai >= 0) { arow = epi.addend + ai * epi.ld_addend; pre.has_add = true; }
    }
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int s = sub + k * LPR;
        if (s < nslots) {
            if (grow) pre.g[k] = *reinterpret_cast<const float4*>(grow + s * 4);
            if (arow) pre.a[k] = *reinterpret_cast<const float4*>(arow + s * 4);
            if (epi.flags & PLNLP_EPI_ACCUM) pre.p[k] = *reinterpret_cast<const float4*>(out + r * ldo + s * 4);
        }
    }
}

template <int VPL, int LPR>
__device__ __forceinline__ void finish_row(float4 (&acc)[VPL], int64_t r, int64_t deg, int mean, int feat,
                                           int nslots, int sub, float* __restrict__ out, int64_t ldo,
                                           const Epi& epi, const EpiPre<VPL>* pre = nullptr) {
    if (mean) {
        const float d = (float)(deg > 0 ? deg : 1);
#pragma unroll
        for (int k = 0; k < VPL; ++k) { acc[k].x /= d; acc[k].y /= d; acc[k].z /= d; acc[k].w /= d; }
    }
    float* orow = out + r * ldo;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        int s = sub + k * LPR;
        if (s < nslots) {
            float4 y;
            if (pre && pre->valid) {
                float4 b4 = zero4;
                if (epi.flags & PLNLP_EPI_BIAS) b4 = *reinterpret_cast<const float4*>(epi.bias + s * 4);
                y = epi_apply4_pre(epi, acc[k], r, (int64_t)s * 4, feat, b4, pre->g[k], pre->p[k], pre->a[k],
                                   pre->has_add);
            } else {
                y = epi_apply4(epi, acc[k], r, (int64_t)s * 4, feat, orow);
            }
            // streaming hint on the result rows: they are not read again by this launch, the gathered table is
            // (measured, profiles/r02_agg_nt_store.txt: -5 % on the collab launch whose 230 MiB table competes with
            // the 230 MiB result for the 256 MiB Infinity Cache, -1 % where the table is far larger than the cache)
            if (epi.flags & PLNLP_EPI_ADAM) { epi_adam4(epi, y, orow + s * 4, r * ldo + (int64_t)s * 4); continue; }
            const f32x4n yv = {y.x, y.y, y.z, y.w};
            __builtin_nontemporal_store(yv, reinterpret_cast<f32x4n*>(orow + s * 4));
        }
    }
}

// main pass: one wave per row; rows longer than skip_above (> 0) are left to the split passes
// (bx, slab: the workgroup's position -- blockIdx.x / .y in the stand-alone launch, an offset block id in the fused one)
template <int VPL, int LPR, bool WEIGHTED, int CHX = 0, bool NT = false>
__device__ __forceinline__ void vec_body(
    int64_t bx, int slab,
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const int32_t* __restrict__ val_index,
    const float* __restrict__ src_scale, const int32_t* __restrict__ src_map,
    const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t ldo,
    int64_t n_rows, int feat, int mean, int64_t skip_above, Epi epi, int64_t row_base, int slab_feat,
    const int32_t* __restrict__ row_index, int xcd_slabs, const int32_t* __restrict__ out_map) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (xcd_slabs > 0) {
        // slabs PINNED to the XCDs: consecutive workgroups go round-robin to the 8 XCDs, so with the slab as the fastest
        // index XCD k only ever gathers columns [k * slab_feat, (k + 1) * slab_feat) of the source rows -- its 4 MiB L2
        // then serves an eighth of the matrix (8x the source rows stay resident), and a source row piece crosses the
        // fabric once per launch at best instead of once per XCD that meets it
        slab = (int)(bx % xcd_slabs);
        bx /= xcd_slabs;
    }
    const int64_t r = row_base + bx * 4 + wave;
    if (r >= n_rows) return;
    const int64_t rg = row_index ? (int64_t)row_index[r] : r;      // CSR row behind result row r
    const int sub = lane % LPR, grp = lane / LPR;
    if (slab_feat > 0) {
        // feature slabs (blockIdx.y): this wave owns columns [c0, c0 + slab_feat) of its row.  A row of a
        // source matrix far beyond the caches is one chain of dependent round trips (rowptr -> col ->
        // batches of neighbour rows); a narrow slab puts 64/LPR neighbours into every wave instruction, so
        // a ~20-neighbour row is fetched in one or two batches instead of three, and there are
        // feat/slab_feat times as many independent chains in flight (the index lists are re-read per
        // slab: 4 bytes per 512).  No dropout epilogue here (its counter is the full-width column).
        const int c0 = slab * slab_feat;
        x += c0;
        out += c0;
        if (epi.bias) epi.bias += c0;
        if (epi.gate) epi.gate += c0;
        if (epi.addend) epi.addend += c0;
        if (epi.adam_m) { epi.adam_m += c0; epi.adam_v += c0; }
        feat = (feat - c0) < slab_feat ? (feat - c0) : slab_feat;
    }
    const int nslots = feat >> 2;
    const int64_t beg = rowptr[rg], end = rowptr[rg + 1];
    if (skip_above > 0 && end - beg > skip_above) {
        // a long row belongs to the split pass, which writes the ONE result row out_map names.  A row-indexed launch
        // may name the CSR row again (CompactIncidence pads its row list to a multiple of 32 with row id 0): those
        // result rows are defined as zero, never left unwritten -- the weight-gradient GEMM reduces over them
        // (against a zero gradient row, and 0 x stale memory is not always 0)
        if (row_index && out_map && (int64_t)out_map[rg] != r) {
            const f32x4n zero = {0.f, 0.f, 0.f, 0.f};
            for (int c = lane; c < nslots; c += 64) reinterpret_cast<f32x4n*>(out + r * ldo)[c] = zero;
        }
        return;
    }
    // (the narrow forms run without the epilogue-operand prefetch: its twelve registers cost them waves of occupancy, and
    // their launches are bound by rows in flight -- see the dispatch of the weighted <1, 16> form)
    constexpr bool PREFETCH = LPR > 16;
    EpiPre<VPL> pre;
    pre.valid = false;
    if constexpr (PREFETCH) { if (LPR == 64 || grp == 0) epi_prefetch<VPL, LPR>(pre, epi, r, nslots, sub, out, ldo); }

    float4 acc[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    agg_range<VPL, LPR, WEIGHTED, CHX, NT>(acc, beg, end, col, val, val_index, src_scale, src_map, x, ldx, lane, sub, grp, nslots);
    fold_groups<VPL, LPR>(acc);
    if (LPR < 64 && grp != 0) return;
    finish_row<VPL, LPR>(acc, r, end - beg, mean, feat, nslots, sub, out, ldo, epi, PREFETCH ? &pre : nullptr);
}

template <int VPL, int LPR, bool WEIGHTED, int CHX = 0, bool NT = false>
__global__ __launch_bounds__(256) void csr_agg_vec_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const int32_t* __restrict__ val_index,
    const float* __restrict__ src_scale, const int32_t* __restrict__ src_map,
    const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t ldo,
    int64_t n_rows, int feat, int mean, int64_t skip_above, Epi epi, int64_t row_base, int slab_feat,
    const int32_t* __restrict__ row_index, int xcd_slabs, const int32_t* __restrict__ out_map) {
    vec_body<VPL, LPR, WEIGHTED, CHX, NT>((int64_t)blockIdx.x, (int)blockIdx.y, rowptr, col, val, val_index, src_scale, src_map,
                                          x, ldx, out, ldo, n_rows, feat, mean, skip_above, epi, row_base, slab_feat,
                                          row_index, xcd_slabs, out_map);
}

// LDS-staged form for SMALL, DENSE graphs (ogbl-ddi: 4 267 nodes, ~500 neighbours per row).  The
// whole source matrix cannot live in LDS, but a feature SLAB of it can: a workgroup stages
// x[:, s0:s0+S] for every source row (n_src * S * 4 bytes <= ~150 KiB, S = 8 floats for ddi) with
// coalesced reads, then serves every neighbour gather of its row range from LDS (ds_read_b128)
// instead of L2: the same 4.4 GB of gathers per layer pass now come from the ~100 TB/s aggregate LDS
// instead of the ~15 TB/s the cache hierarchy gave.  64/L neighbours are fetched per wave
// instruction (L = S/4 lanes per neighbour); the partial sums of the lane groups are folded with
// a wavefront shuffle tree once per row.  Grid = feature slabs x row blocks.
constexpr int LDS_THREADS = 1024;   // 16 waves share one staged slab: the neighbour loop is a chain of
                                    // dependent index loads, so it needs many waves to hide them
template <int S, bool WEIGHTED>
__global__ __launch_bounds__(LDS_THREADS) void csr_agg_lds_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const int32_t* __restrict__ val_index,
    const float* __restrict__ src_scale, const int32_t* __restrict__ src_map, const float* __restrict__ x, int64_t ldx,
    float* __restrict__ out, int64_t ldo, int64_t n_rows, int64_t n_src, int feat, int mean,
    int64_t rows_per_block, Epi epi) {
    extern __shared__ __attribute__((aligned(16))) float4 slab[];      // [n_src][L]
    constexpr int L = S / 4;          // float4 parts per source row
    constexpr int NPW = 64 / L;       // neighbours per wave instruction
    constexpr int U = 16;             // wave instructions in flight (a ~500-neighbour row in one batch at S = 8)
    constexpr int NW = LDS_THREADS / 64;
    const int s0 = blockIdx.x * S;
    // ---- stage the slab: consecutive threads read consecutive 16-byte parts of a row
    for (int64_t i = threadIdx.x; i < n_src * L; i += LDS_THREADS) {
        const int64_t row = i / L;
        const int f = s0 + (int)(i % L) * 4;
        slab[i] = f < feat ? *reinterpret_cast<const float4*>(x + row * ldx + f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nb = lane / L, part = lane % L;
    const int64_t r_end = ((int64_t)blockIdx.y + 1) * rows_per_block < n_rows
                              ? ((int64_t)blockIdx.y + 1) * rows_per_block : n_rows;
    for (int64_t r = (int64_t)blockIdx.y * rows_per_block + wave; r < r_end; r += NW) {
        const int64_t beg = rowptr[r], end = rowptr[r + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t e0 = beg; e0 < end; e0 += NPW * U) {
            int c[U];
            float w[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t e = e0 + u * NPW + nb;
                ok[u] = e < end;
                c[u] = ok[u] ? col[e] : 0;
                w[u] = 1.f;
                if constexpr (WEIGHTED) {
                    if (ok[u]) {
                        w[u] = val ? val[val_index ? (int64_t)val_index[e] : e] : 1.f;
                        if (src_scale) w[u] *= src_scale[c[u]];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 v = slab[(int64_t)c[u] * L + part];
                if (ok[u]) {          // never 0 * x
                    acc.x = fmaf(w[u], v.x, acc.x); acc.y = fmaf(w[u], v.y, acc.y);
                    acc.z = fmaf(w[u], v.z, acc.z); acc.w = fmaf(w[u], v.w, acc.w);
                }
            }
        }
#pragma unroll
        for (int o = L; o < 64; o <<= 1) {       // fold the NPW lane groups (fixed order)
            acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
            acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
        }
        const int f = s0 + part * 4;
        if (nb == 0 && f < feat) {
            if (mean) {
                const float d = (float)((end - beg) > 0 ? (end - beg) : 1);
                acc.x /= d; acc.y /= d; acc.z /= d; acc.w /= d;
            }
            float* orow = out + r * ldo;
            const float4 y = epi_apply4(epi, acc, r, f, feat, orow);
            if (epi.flags & PLNLP_EPI_ADAM) epi_adam4(epi, y, orow + f, r * ldo + f);
            else *reinterpret_cast<float4*>(orow + f) = y;
        }
    }
}

template <int S>
static int launch_lds(bool weighted, hipStream_t s, const int64_t* rowptr, const int32_t* col, const float* val,
                      const int32_t* val_index, const float* src_scale, const int32_t* src_map, const float* x, int64_t ldx, float* out,
                      int64_t ldo, int64_t n_rows, int64_t n_src, int feat, int mean, const Epi& e) {
    const size_t lds_bytes = (size_t)n_src * S * 4;
    const int slabs = (feat + S - 1) / S;
    int64_t rb = (256 + slabs - 1) / slabs;      // one 16-wave workgroup per CU, one round
    if (rb < 1) rb = 1;
    if (rb > (n_rows + 15) / 16) rb = (n_rows + 15) / 16;
    const int64_t rows_per_block = (n_rows + rb - 1) / rb;
    dim3 grid((unsigned)slabs, (unsigned)rb);
    hipError_t err;
    if (weighted) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(&csr_agg_lds_kernel<S, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (err != hipSuccess) return (int)err;
        hipLaunchKernelGGL((csr_agg_lds_kernel<S, true>), grid, dim3(LDS_THREADS), lds_bytes, s, rowptr, col, val, val_index,
                           src_scale, src_map, x, ldx, out, ldo, n_rows, n_src, feat, mean, rows_per_block, e);
    } else {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(&csr_agg_lds_kernel<S, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (err != hipSuccess) return (int)err;
        hipLaunchKernelGGL((csr_agg_lds_kernel<S, false>), grid, dim3(LDS_THREADS), lds_bytes, s, rowptr, col, val,
                           val_index, src_scale, src_map, x, ldx, out, ldo, n_rows, n_src, feat, mean, rows_per_block, e);
    }
    count_launch(LK_AGG_LDS);
    return launch_status();
}

// split pass 1: one wave per chunk (<= threshold edges) of a long row; raw weighted sums to the workspace
struct SplitArgs {
    int64_t threshold;
    int64_t n_long;
    const int64_t* long_rows;
    const int64_t* chunk_beg;
    const int32_t* chunk_cnt;
    int64_t n_chunks;
    const int32_t* chunk_long;
    float* ws;
    // explicit chunks (plnlp_row_split.seg_*): chunk c = entries [seg_beg[c], seg_beg[c] + seg_len[c]), partial sum to
    // workspace slot seg_slot[c]; processed in table order (the caller's order: by source range)
    const int64_t* seg_beg;
    const int32_t* seg_len;
    const int32_t* seg_slot;
};

template <int VPL, int LPR, bool WEIGHTED>
__device__ __forceinline__ void chunk_body(
    int64_t bx, int slab,
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const int32_t* __restrict__ val_index,
    const float* __restrict__ src_scale, const int32_t* __restrict__ src_map,
    const float* __restrict__ x, int64_t ldx, int feat, SplitArgs sp, int slab_feat, int xcd_slabs,
    int64_t chunk_base) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (xcd_slabs > 0) { slab = (int)(bx % xcd_slabs); bx /= xcd_slabs; }   // see vec_body
    const int64_t c = chunk_base + bx * 4 + wave;
    if (c >= sp.n_chunks) return;
    const int feat_full = feat;
    int c0 = 0;
    if (slab_feat > 0) {          // feature slab of this block (see csr_agg_vec_kernel): columns [c0, c0 + slab_feat)
        c0 = slab * slab_feat;
        x += c0;
        feat = (feat - c0) < slab_feat ? (feat - c0) : slab_feat;
    }
    int64_t beg, end, slot = c;
    if (sp.seg_beg) {
        beg = sp.seg_beg[c];
        end = beg + sp.seg_len[c];
        slot = sp.seg_slot[c];
        if (slot < 0) return;
    } else {
        const int l = sp.chunk_long[c];
        if (l < 0) return;                 // idle chunk slot
        const int64_t r = sp.long_rows[l];
        if (r < 0) return;
        const int64_t j = c - sp.chunk_beg[l];
        const int64_t rb = rowptr[r], re = rowptr[r + 1];
        beg = rb + j * sp.threshold;
        end = (beg + sp.threshold) < re ? (beg + sp.threshold) : re;
    }
    const int sub = lane % LPR, grp = lane / LPR;
    const int nslots = feat >> 2;
    float4 acc[VPL];
#pragma unroll
    for (int k = 0; k < VPL; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    // (the weighted narrow form with 4 groups in flight, as in the main pass: 83 -> ~55 VGPRs, 5 -> 7 waves per SIMD)
    constexpr int CHC = (WEIGHTED && VPL == 1 && LPR == 16) ? 4 : 0;
    agg_range<VPL, LPR, WEIGHTED, CHC>(acc, beg, end, col, val, val_index, src_scale, src_map, x, ldx, lane, sub, grp, nslots);
    fold_groups<VPL, LPR>(acc);
    if (LPR < 64 && grp != 0) return;
    float* w = sp.ws + slot * (int64_t)feat_full + c0;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        int s = sub + k * LPR;
        if (s < nslots) *reinterpret_cast<float4*>(w + s * 4) = acc[k];
    }
}

template <int VPL, int LPR, bool WEIGHTED>
__global__ __launch_bounds__(256) void csr_agg_chunk_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const int32_t* __restrict__ val_index,
    const float* __restrict__ src_scale, const int32_t* __restrict__ src_map,
    const float* __restrict__ x, int64_t ldx, int feat, SplitArgs sp, int slab_feat, int xcd_slabs,
    int64_t chunk_base) {
    chunk_body<VPL, LPR, WEIGHTED>((int64_t)blockIdx.x, (int)blockIdx.y, rowptr, col, val, val_index, src_scale, src_map, x,
                                   ldx, feat, sp, slab_feat, xcd_slabs, chunk_base);
}

// main pass and chunk pass in ONE launch (PLNLP_AGG_FUSED_PASSES): the first chunk_blocks workgroups take the long
// rows' chunks, the rest one short row per wave.  The two passes write disjoint memory (result rows / workspace) and
// are bound by different things -- the short rows by the bytes they pull through the fabric, the chunks by their chain
// of dependent round trips -- so run back to back each leaves the other's resource idle; co-resident, the chunk
// waves' latency hides behind the row waves' traffic.  Chunks first: they are the long poles.  The finalize pass (one
// workgroup per long row) stays a launch of its own: it needs every partial sum.
template <int VPL, int LPR, int CVPL, int CLPR, bool WEIGHTED>
__global__ __launch_bounds__(256) void csr_agg_fused_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const int32_t* __restrict__ val_index,
    const float* __restrict__ src_scale, const int32_t* __restrict__ src_map,
    const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t ldo,
    int64_t n_rows, int feat, int mean, int64_t skip_above, Epi epi, const int32_t* __restrict__ row_index,
    SplitArgs sp, int chunk_slab_feat, int chunk_xcd_slabs, int64_t chunk_blocks, const int32_t* __restrict__ out_map) {
    const int64_t bx = blockIdx.x;
    if (bx < chunk_blocks) {       // block-uniform
        chunk_body<CVPL, CLPR, WEIGHTED>(bx, 0, rowptr, col, val, val_index, src_scale, src_map, x, ldx, feat, sp,
                                         chunk_slab_feat, chunk_xcd_slabs, 0);
        return;
    }
    vec_body<VPL, LPR, WEIGHTED>(bx - chunk_blocks, 0, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo,
                                 n_rows, feat, mean, skip_above, epi, 0, 0, row_index, 0, out_map);
}

// split pass 2: one BLOCK per long row.  Its 4 waves take the row's chunk sums round-robin (wave w:
// chunks w, w+4, ...), each adding in increasing chunk order, and wave 0 then adds the 4 wave sums in
// wave order -- a fixed order, so the result is reproducible; a hub with thousands of chunks no
// longer serialises on one wave.  Then mean / epilogue / store.
__global__ __launch_bounds__(256) void csr_agg_finalize_kernel(const int64_t* __restrict__ rowptr, int feat,
                                                               int mean, SplitArgs sp, float* __restrict__ out,
                                                               int64_t ldo, Epi epi,
                                                               const int32_t* __restrict__ out_map) {
    __shared__ float4 part[3][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t l = blockIdx.x;
    if (l >= sp.n_long) return;
    const int64_t rg = sp.long_rows[l];
    if (rg < 0) return;                                  // idle slot (block-uniform)
    const int64_t r = out_map ? (int64_t)out_map[rg] : rg;   // result row of this CSR row (row-indexed launches)
    if (r < 0) return;                                   // not among the produced rows
    const int64_t c0 = sp.chunk_beg[l], c1 = c0 + sp.chunk_cnt[l];
    const int64_t deg = rowptr[rg + 1] - rowptr[rg];
    const int nslots = feat >> 2;
    for (int s0 = 0; s0 < nslots; s0 += 64) {
        const int s = s0 + lane;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < nslots) {
            int64_t c = c0 + wave;
            for (; c + 12 < c1; c += 16) {               // 4 loads in flight
                const float4 v0 = *reinterpret_cast<const float4*>(sp.ws + (c + 0) * (int64_t)feat + s * 4);
                const float4 v1 = *reinterpret_cast<const float4*>(sp.ws + (c + 4) * (int64_t)feat + s * 4);
                const float4 v2 = *reinterpret_cast<const float4*>(sp.ws + (c + 8) * (int64_t)feat + s * 4);
                const float4 v3 = *reinterpret_cast<const float4*>(sp.ws + (c + 12) * (int64_t)feat + s * 4);
                acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
                acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
                acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
                acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
            }
            for (; c < c1; c += 4) {
                const float4 v = *reinterpret_cast<const float4*>(sp.ws + c * (int64_t)feat + s * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        if (wave > 0) part[wave - 1][lane] = acc;
        __syncthreads();
        if (wave == 0 && s < nslots) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float4 v = part[w][lane];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            if (mean) {
                const float d = (float)(deg > 0 ? deg : 1);
                acc.x /= d; acc.y /= d; acc.z /= d; acc.w /= d;
            }
            float* orow = out + r * ldo;
            const float4 y = epi_apply4(epi, acc, r, (int64_t)s * 4, feat, orow);
            if (epi.flags & PLNLP_EPI_ADAM) epi_adam4(epi, y, orow + s * 4, r * ldo + (int64_t)s * 4);
            else *reinterpret_cast<float4*>(orow + s * 4) = y;
        }
        __syncthreads();
    }
}

// scalar path: any feat / alignment.  Lane l owns columns l, l+64, ...
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void csr_agg_scalar_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, const int32_t* __restrict__ val_index,
    const float* __restrict__ src_scale, const int32_t* __restrict__ src_map,
    const float* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t ldo,
    int64_t n_rows, int feat, int mean, Epi epi, int64_t row_base, const int32_t* __restrict__ row_index) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t r = row_base + (int64_t)blockIdx.x * 4 + wave;
    if (r >= n_rows) return;
    const int64_t rg = row_index ? (int64_t)row_index[r] : r;
    const int64_t beg = rowptr[rg], end = rowptr[rg + 1];
    for (int f0 = 0; f0 < feat; f0 += 64 * 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int64_t e = beg; e < end; ++e) {
            int c = col[e];
            float w = 1.f;
            if constexpr (WEIGHTED) {
                w = val ? val[val_index ? (int64_t)val_index[e] : e] : 1.f;
                if (src_scale) w *= src_scale[c];
            }
            if (src_map) { c = src_map[c]; if (c < 0) continue; }
            const float* p = x + (int64_t)c * ldx;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int f = f0 + lane + 64 * k;
                if (f < feat) acc[k] = WEIGHTED ? fmaf(w, p[f], acc[k]) : acc[k] + p[f];
            }
        }
        const float d = (float)((end - beg) > 0 ? (end - beg) : 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int f = f0 + lane + 64 * k;
            if (f < feat) {
                float v = mean ? acc[k] / d : acc[k];
                float prev = (epi.flags & PLNLP_EPI_ACCUM) ? out[r * ldo + f] : 0.f;
                out[r * ldo + f] = epi_apply(epi, v, r, f, feat, prev);
            }
        }
    }
}

template <int VPL, int LPR>
static int launch_split(bool weighted, hipStream_t s, const int64_t* rowptr, const int32_t* col, const float* val,
                        const int32_t* val_index, const float* src_scale, const int32_t* src_map, const float* x, int64_t ldx, float* out,
                        int64_t ldo, int feat, int mean, const Epi& e, const SplitArgs* sp,
                        const int32_t* out_map = nullptr, int slab_feat = 0, int xcd_slabs = 0);

template <int VPL, int LPR, int CHX = 0, bool NT = false>
static int launch_vec(bool weighted, dim3 grid, hipStream_t s, const int64_t* rowptr, const int32_t* col,
                      const float* val, const int32_t* val_index, const float* src_scale, const int32_t* src_map, const float* x,
                      int64_t ldx, float* out,
                      int64_t ldo, int64_t n_rows, int feat, int mean, const Epi& e, const SplitArgs* sp,
                      int slab_feat = 0, const int32_t* row_index = nullptr, const int32_t* out_map = nullptr,
                      int xcd_slabs = 0, bool hub_xcd = false, bool fused = false) {
    const int64_t skip = sp ? sp->threshold : 0;
    if constexpr (LPR == 64 && CHX == 0 && !NT) {
        // PLNLP_AGG_FUSED_PASSES: the chunk pass rides in the main pass's launch (csr_agg_fused_kernel)
        const bool pin = hub_xcd && (feat == 256 || feat == 512 || feat == 1024);
        if (fused && sp && sp->n_long > 0 && sp->n_chunks > 0 && slab_feat == 0) {
            const int64_t cblocks = ((sp->n_chunks + 3) / 4) * (pin ? 8 : 1);
            if (cblocks + (int64_t)grid.x <= ((int64_t)1 << 22)) {
                const dim3 g((unsigned)(cblocks + (int64_t)grid.x));
#define PLNLP_FUSED(CV, CL, W, CSLAB, CXCD) \
    hipLaunchKernelGGL((csr_agg_fused_kernel<VPL, LPR, CV, CL, W>), g, dim3(256), 0, s, rowptr, col, val, val_index, \
                       src_scale, src_map, x, ldx, out, ldo, n_rows, feat, mean, skip, e, row_index, *sp, CSLAB, CXCD, cblocks, out_map)
                if (!pin) { if (weighted) PLNLP_FUSED(VPL, LPR, true, 0, 0); else PLNLP_FUSED(VPL, LPR, false, 0, 0); }
                else if (feat == 256) { if (weighted) PLNLP_FUSED(1, 8, true, 32, 8); else PLNLP_FUSED(1, 8, false, 32, 8); }
                else if (feat == 512) { if (weighted) PLNLP_FUSED(1, 16, true, 64, 8); else PLNLP_FUSED(1, 16, false, 64, 8); }
                else { if (weighted) PLNLP_FUSED(1, 32, true, 128, 8); else PLNLP_FUSED(1, 32, false, 128, 8); }
#undef PLNLP_FUSED
                count_launch(pin ? LK_AGG_FUSED_HUB_XCD : LK_AGG_FUSED);
                count_launch(LK_AGG_FINALIZE);
                if (int rc = launch_status()) return rc;
                hipLaunchKernelGGL(csr_agg_finalize_kernel, dim3((unsigned)sp->n_long), dim3(256), 0, s, rowptr,
                                   feat, mean, *sp, out, ldo, e, out_map);
                return launch_status();
            }
        }
    }
    // a launch may not exceed 2^32 threads: beyond 2^22 blocks (16 Mi rows) the rows go in slices
    const int n_slabs = slab_feat > 0 ? (feat + slab_feat - 1) / slab_feat : 1;
    const int64_t MAX_BLOCKS = ((int64_t)1 << 22) / n_slabs;
    const int64_t blocks = grid.x;
    for (int64_t b0 = 0; b0 < blocks; b0 += MAX_BLOCKS) {
        const int64_t nb = (blocks - b0) < MAX_BLOCKS ? (blocks - b0) : MAX_BLOCKS;
        const dim3 g = xcd_slabs > 0 ? dim3((unsigned)(nb * xcd_slabs)) : dim3((unsigned)nb, (unsigned)n_slabs);
        if (weighted)
            hipLaunchKernelGGL((csr_agg_vec_kernel<VPL, LPR, true, CHX, NT>), g, dim3(256), 0, s, rowptr, col, val,
                               val_index, src_scale, src_map, x, ldx, out, ldo, n_rows, feat, mean, skip, e, b0 * 4,
                               slab_feat, row_index, xcd_slabs, out_map);
        else
            hipLaunchKernelGGL((csr_agg_vec_kernel<VPL, LPR, false, CHX, NT>), g, dim3(256), 0, s, rowptr, col, val,
                               val_index, src_scale, src_map, x, ldx, out, ldo, n_rows, feat, mean, skip, e, b0 * 4,
                               slab_feat, row_index, xcd_slabs, out_map);
        count_launch(xcd_slabs > 0 ? LK_AGG_VEC_XCD : (slab_feat > 0 ? LK_AGG_VEC_SLABS : LK_AGG_VEC));
        if (int rc = launch_status()) return rc;
    }
    if (hub_xcd && slab_feat == 0 && (feat == 256 || feat == 512 || feat == 1024)) {
        // full-width main pass, but the long rows' chunks in eight slabs pinned to the XCDs (PLNLP_AGG_HUB_XCD)
        if (feat == 256)
            return launch_split<1, 8>(weighted, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo, feat,
                                      mean, e, sp, out_map, 32, 8);
        if (feat == 512)
            return launch_split<1, 16>(weighted, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo, feat,
                                       mean, e, sp, out_map, 64, 8);
        return launch_split<1, 32>(weighted, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo, feat,
                                   mean, e, sp, out_map, 128, 8);
    }
    if (slab_feat > 0)        // the long rows' chunks run per slab too (same geometry); the finalize pass is full width
        return launch_split<VPL, LPR>(weighted, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo, feat, mean,
                                      e, sp, out_map, slab_feat, xcd_slabs);
    return launch_split<VPL, LPR>(weighted, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo, feat, mean,
                                  e, sp, out_map);
}

template <int VPL, int LPR>
static int launch_split(bool weighted, hipStream_t s, const int64_t* rowptr, const int32_t* col, const float* val,
                        const int32_t* val_index, const float* src_scale, const int32_t* src_map, const float* x, int64_t ldx, float* out,
                        int64_t ldo, int feat, int mean, const Epi& e, const SplitArgs* sp, const int32_t* out_map,
                        int slab_feat, int xcd_slabs) {
    if (!sp || sp->n_long == 0 || sp->n_chunks == 0) return 0;
    const int n_slabs = slab_feat > 0 ? (feat + slab_feat - 1) / slab_feat : 1;
    // (a launch may not exceed 2^32 threads: the chunks go in slices of 2^22 workgroups over all slabs)
    const int64_t cblocks = (sp->n_chunks + 3) / 4;
    const int64_t MAX_BLOCKS = ((int64_t)1 << 22) / (xcd_slabs > 0 ? xcd_slabs : n_slabs);
    for (int64_t b0 = 0; b0 < cblocks; b0 += MAX_BLOCKS) {
        const int64_t nb = (cblocks - b0) < MAX_BLOCKS ? (cblocks - b0) : MAX_BLOCKS;
        const dim3 cgrid = xcd_slabs > 0 ? dim3((unsigned)(nb * xcd_slabs)) : dim3((unsigned)nb, (unsigned)n_slabs);
        if (weighted)
            hipLaunchKernelGGL((csr_agg_chunk_kernel<VPL, LPR, true>), cgrid, dim3(256), 0, s, rowptr, col, val,
                               val_index, src_scale, src_map, x, ldx, feat, *sp, slab_feat, xcd_slabs, b0 * 4);
        else
            hipLaunchKernelGGL((csr_agg_chunk_kernel<VPL, LPR, false>), cgrid, dim3(256), 0, s, rowptr, col, val,
                               val_index, src_scale, src_map, x, ldx, feat, *sp, slab_feat, xcd_slabs, b0 * 4);
        count_launch(xcd_slabs > 0 ? LK_AGG_CHUNK_XCD : LK_AGG_CHUNK);
        if (int rc = launch_status()) return rc;
    }
    count_launch(LK_AGG_FINALIZE);
    hipLaunchKernelGGL(csr_agg_finalize_kernel, dim3((unsigned)sp->n_long), dim3(256), 0, s, rowptr,
                       feat, mean, *sp, out, ldo, e, out_map);
    return launch_status();
}

}  // namespace plnlp

extern "C" int plnlp_csr_aggregate_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                                       const int32_t* val_index, const float* src_scale, const int32_t* src_map,
                                       const int32_t* row_index, const int32_t* split_out_map, const float* x,
                                       int64_t ldx, float* out,
                                       int64_t ldo, int64_t n_rows, int64_t n_src, int64_t feat, int reduce,
                                       int flags, const plnlp_epilogue* epi, const plnlp_row_split* split,
                                       void* stream) {
    using namespace plnlp;
    if (n_rows < 0 || feat <= 0 || ldx < feat || ldo < feat || feat > (1 << 20)) return PLNLP_E_SHAPE;
    if (reduce != PLNLP_REDUCE_SUM && reduce != PLNLP_REDUCE_MEAN) return PLNLP_E_UNSUPPORTED;
    // nothing to produce: an empty result has no storage, so its pointer may be NULL (a rank of a row-sharded step whose
    // block no edge of the batch touches -- found by the world-2 run of tests/test_hip_multirank.py)
    if (n_rows == 0) return 0;
    if (!rowptr || !x || !out) return PLNLP_E_NULL;
    if (!col) return PLNLP_E_NULL;
    Epi e;
    if (int rc = make_epi(epi, &e, /*allow_adam=*/true)) return rc;
    if (n_rows > (int64_t)4 * 0x7FFFFFFF) return PLNLP_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((n_rows + 3) / 4));
    const bool weighted = (val != nullptr) || (src_scale != nullptr);
    const int mean = reduce == PLNLP_REDUCE_MEAN;
    const bool vec_ok = (feat % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) &&
                        ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) && feat <= 1024;
    SplitArgs sa{};
    const SplitArgs* sp = nullptr;
    if (split && split->n_long > 0 && split->n_chunks > 0) {
        if (!vec_ok) return PLNLP_E_UNSUPPORTED;   // long-row splitting is implemented on the vector path
        if (split->threshold < 64) return PLNLP_E_SHAPE;
        if (!split->long_rows || !split->chunk_beg || !split->chunk_cnt || !split->chunk_long ||
            !split->workspace) return PLNLP_E_NULL;
        if (split->workspace_floats < split->n_chunks * feat) return PLNLP_E_WORKSPACE;
        if ((uintptr_t)split->workspace % 16 != 0) return PLNLP_E_ALIGN;
        sa.threshold = split->threshold; sa.n_long = split->n_long; sa.long_rows = split->long_rows;
        sa.chunk_beg = split->chunk_beg; sa.chunk_cnt = split->chunk_cnt; sa.n_chunks = split->n_chunks;
        sa.chunk_long = split->chunk_long;
        sa.ws = split->workspace;
        if (split->seg_beg) {
            if (!split->seg_len || !split->seg_slot) return PLNLP_E_NULL;
            sa.seg_beg = split->seg_beg; sa.seg_len = split->seg_len; sa.seg_slot = split->seg_slot;
        }
        sp = &sa;
    }
    // the Adam epilogue updates every result row exactly once: the vector path over ALL rows (no row subset)
    if ((e.flags & PLNLP_EPI_ADAM) && (!vec_ok || row_index)) return PLNLP_E_UNSUPPORTED;
    if (!vec_ok) {
        constexpr int64_t MAX_BLOCKS = (int64_t)1 << 22;          // < 2^32 threads per launch
        for (int64_t b0 = 0; b0 < (int64_t)grid.x; b0 += MAX_BLOCKS) {
            const dim3 g((unsigned)(((int64_t)grid.x - b0) < MAX_BLOCKS ? ((int64_t)grid.x - b0) : MAX_BLOCKS));
            if (weighted)
                hipLaunchKernelGGL((csr_agg_scalar_kernel<true>), g, dim3(256), 0, s, rowptr, col, val, val_index,
                                   src_scale, src_map, x, ldx, out, ldo, n_rows, (int)feat, mean, e, b0 * 4, row_index);
            else
                hipLaunchKernelGGL((csr_agg_scalar_kernel<false>), g, dim3(256), 0, s, rowptr, col, val, val_index,
                                   src_scale, src_map, x, ldx, out, ldo, n_rows, (int)feat, mean, e, b0 * 4, row_index);
            count_launch(LK_AGG_SCALAR);
            if (int rc = launch_status()) return rc;
        }
        return 0;
    }
    const int nslots = (int)(feat / 4);
    if (src_map) flags &= ~PLNLP_AGG_LDS_STAGE;   // the mapped gather exists on the one-row-per-wave forms only
    if (row_index) {
        flags &= ~PLNLP_AGG_LDS_STAGE;     // row-indexed launches: one-row-per-wave forms
        if (sp && !split_out_map) return PLNLP_E_NULL;
    }
    if ((flags & PLNLP_AGG_LDS_STAGE) && n_src > 0 && n_src * 16 <= PLNLP_AGG_LDS_BUDGET) {
        // widest slab that fits the LDS budget
        if (n_src * 128 <= PLNLP_AGG_LDS_BUDGET && feat >= 32)
            return launch_lds<32>(weighted, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo, n_rows, n_src, (int)feat, mean, e);
        if (n_src * 64 <= PLNLP_AGG_LDS_BUDGET && feat >= 16)
            return launch_lds<16>(weighted, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo, n_rows, n_src, (int)feat, mean, e);
        if (n_src * 32 <= PLNLP_AGG_LDS_BUDGET && feat >= 8)
            return launch_lds<8>(weighted, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo, n_rows, n_src, (int)feat, mean, e);
        return launch_lds<4>(weighted, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo, n_rows, n_src, (int)feat, mean, e);
    }
    const bool hub_xcd = (flags & PLNLP_AGG_HUB_XCD) != 0, fused = (flags & PLNLP_AGG_FUSED_PASSES) != 0;
#define PLNLP_AGG(VPL, LPR) \
    return launch_vec<VPL, LPR>(weighted, grid, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo, n_rows, (int)feat, mean, e, sp, 0, row_index, split_out_map, 0, hub_xcd, fused)
#define PLNLP_AGGX(VPL, LPR, CHX, NT) \
    return launch_vec<VPL, LPR, CHX, NT>(weighted, grid, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo, n_rows, (int)feat, mean, e, sp, 0, row_index, split_out_map)
    // eight feature slabs pinned to the eight XCDs (see csr_agg_vec_kernel): F = 256 / 512 / 1024
    if ((flags & PLNLP_AGG_SLABS_XCD) && !(e.flags & PLNLP_EPI_DROPOUT) && grid.x <= (1u << 22) / 8) {
#define PLNLP_AGG_XCD(LPR) \
    return launch_vec<1, LPR>(weighted, grid, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo, \
                              n_rows, (int)feat, mean, e, sp, (int)feat / 8, row_index, split_out_map, 8)
        if (feat == 256) PLNLP_AGG_XCD(8);
        if (feat == 512) PLNLP_AGG_XCD(16);
        if (feat == 1024) PLNLP_AGG_XCD(32);
#undef PLNLP_AGG_XCD
    }
    // feature slabs: 128 or 256 columns per wave (see csr_agg_vec_kernel); rows and slabs fill the grid
    if ((flags & (PLNLP_AGG_SLABS_128 | PLNLP_AGG_SLABS_256)) && nslots > 32 && !(e.flags & PLNLP_EPI_DROPOUT)) {
        if ((flags & PLNLP_AGG_SLABS_128))
            return launch_vec<1, 32>(weighted, grid, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo,
                                     n_rows, (int)feat, mean, e, sp, 128, row_index, split_out_map);
        if (nslots > 64)
            return launch_vec<1, 64>(weighted, grid, s, rowptr, col, val, val_index, src_scale, src_map, x, ldx, out, ldo,
                                     n_rows, (int)feat, mean, e, sp, 256, row_index, split_out_map);
    }
    // tuning variants of the full-wave forms (flags): streaming-hint row loads, fewer rows in flight
    if (nslots > 32 && nslots <= 64 && (flags & PLNLP_AGG_NT_LOADS)) PLNLP_AGGX(1, 64, 0, true);
    if (nslots > 64 && nslots <= 128) {
        const bool nt = flags & PLNLP_AGG_NT_LOADS, few = flags & PLNLP_AGG_FEW_IN_FLIGHT;
        if (nt && few) PLNLP_AGGX(2, 64, 4, true);
        if (nt) PLNLP_AGGX(2, 64, 0, true);
        if (few) PLNLP_AGGX(2, 64, 4, false);
    }
    // the weighted narrow form with 4 instead of 8 neighbour groups in flight: 53 instead of 84 VGPRs, 7 instead of 5 waves per
    // SIMD -- these launches (citation2's 52-wide embedding block, forward and transposed) are bound by rows in flight:
    // 3.54 ms at 4 waves (97 VGPRs), 2.99 at 5, 2.91 at 6 (two spills), 2.80 here; same sums in the same order
    if (weighted && nslots > 8 && nslots <= 16) PLNLP_AGGX(1, 16, 4, false);
    if (nslots <= 8) PLNLP_AGG(1, 8);
    if (nslots <= 16) PLNLP_AGG(1, 16);
    if (nslots <= 32) PLNLP_AGG(1, 32);
    if (nslots <= 64) PLNLP_AGG(1, 64);
    if (nslots <= 128) PLNLP_AGG(2, 64);
    PLNLP_AGG(4, 64);
#undef PLNLP_AGG
#undef PLNLP_AGGX
}
