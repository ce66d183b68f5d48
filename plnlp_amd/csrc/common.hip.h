// Shared device helpers for libplnlp_hip.so (gfx950 only; wave64 hard-coded).
#pragma once
#include <cmath>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/plnlp_hip.h"

#define PLNLP_WAVE 64

namespace plnlp {

// ---- counter-based dropout mask ---------------------------------------------
// keep(element) for logical element index idx = row * n_cols + col under a 64-bit
// per-call seed.  Elements are hashed in GROUPS of four (group = idx >> 2): two
// rounds of a 32-bit mixer over (group, seed) give one word, a third round a second
// word, and each element takes a 16-bit field of them -- 3 rounds per 4 elements
// instead of 8 (the hash was ~10 % of the forward GEMM).  P(drop) = thresh / 65536.
// oracle/reference_path.py::dropout_keep_mask restates this in numpy; the two must
// stay bit-identical.
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
// plain two-round counter hash of a 64-bit index (random walks)
__device__ __forceinline__ uint32_t counter_hash(uint64_t idx, uint32_t seed_lo, uint32_t seed_hi) {
    uint32_t h = lowbias32((uint32_t)idx ^ seed_lo);
    return lowbias32(h + (uint32_t)(idx >> 32) * 0x9E3779B9u + seed_hi);
}
__device__ __forceinline__ void dropout_words(uint64_t group, uint32_t seed_lo, uint32_t seed_hi,
                                              uint32_t& a, uint32_t& b) {
    a = counter_hash(group, seed_lo, seed_hi);
    b = lowbias32(a ^ 0x85EBCA6Bu);
}
__device__ __forceinline__ bool dropout_keep(uint64_t idx, uint32_t seed_lo, uint32_t seed_hi,
                                             uint32_t thresh) {
    uint32_t a, b;
    dropout_words(idx >> 2, seed_lo, seed_hi, a, b);
    const uint32_t word = (idx & 2) ? b : a;
    return ((word >> (((uint32_t)idx & 1u) * 16u)) & 0xFFFFu) >= thresh;
}
// the four elements i0 .. i0+3 scaled or zeroed; one group hash when i0 is a multiple of 4
__device__ __forceinline__ float4 dropout_apply4(float4 v, uint64_t i0, uint32_t seed_lo, uint32_t seed_hi,
                                                 uint32_t thresh, float keep_scale) {
    if ((i0 & 3) == 0) {
        uint32_t a, b;
        dropout_words(i0 >> 2, seed_lo, seed_hi, a, b);
        v.x = (a & 0xFFFFu) >= thresh ? v.x * keep_scale : 0.f;
        v.y = (a >> 16) >= thresh ? v.y * keep_scale : 0.f;
        v.z = (b & 0xFFFFu) >= thresh ? v.z * keep_scale : 0.f;
        v.w = (b >> 16) >= thresh ? v.w * keep_scale : 0.f;
        return v;
    }
    v.x = dropout_keep(i0 + 0, seed_lo, seed_hi, thresh) ? v.x * keep_scale : 0.f;
    v.y = dropout_keep(i0 + 1, seed_lo, seed_hi, thresh) ? v.y * keep_scale : 0.f;
    v.z = dropout_keep(i0 + 2, seed_lo, seed_hi, thresh) ? v.z * keep_scale : 0.f;
    v.w = dropout_keep(i0 + 3, seed_lo, seed_hi, thresh) ? v.w * keep_scale : 0.f;
    return v;
}
__host__ __device__ inline uint32_t dropout_thresh(float p) {   // 16-bit: P(drop) = thresh / 65536
    double t = (double)p * 65536.0 + 0.5;
    if (t < 0.0) t = 0.0;
    if (t > 65535.0) t = 65535.0;
    return (uint32_t)t;
}

// ---- epilogue, device-side copy of plnlp_epilogue ------------------------------
struct Epi {
    uint32_t     flags;
    uint32_t     thresh;        // dropout threshold
    uint32_t     seed_lo, seed_hi;
    const uint32_t* seed_ptr;   // nullable: the 64-bit dropout seed lives in DEVICE memory (lo word, hi word) -- a step
                                // captured in a hipGraph draws a fresh mask per replay without a new kernel argument
    float        keep_scale;    // 1/(1-p)
    float        gate_scale;
    const float* bias;
    const float* gate;
    int64_t      ld_gate;
    const int32_t* gate_index;  // gate row of result row r (nullable: r itself)
    const float* addend;        // PLNLP_EPI_ADDEND
    int64_t      ld_addend;
    const int32_t* addend_index;
    const int32_t* drop_row;    // dropout counter row of result row r (nullable: r itself)
    int          vec4;          // bias / gate / addend may be read 16 bytes at a time (f % 4 == 0 callers only)
    // PLNLP_EPI_ADAM
    float*       adam_m;
    float*       adam_v;
    float        adam_lr, adam_b1, adam_b2, adam_eps, adam_bc1, adam_bc2_sqrt;
    const float* adam_scalars;  // nullable: {lr, 1 - beta1^t, sqrt(1 - beta2^t)} of THIS step in device memory
                                // (plnlp_adam_step_scalars); overrides adam_lr / adam_bc1 / adam_bc2_sqrt
    // PLNLP_EPI_ROWDOT (gemm_x3s only)
    const float* rowdot_w;
    float*       rowdot_out;
    int64_t      rowdot_ld;
};

// the dropout seed of this launch: from device memory when the caller put it there
__device__ __forceinline__ void epi_seed(const Epi& e, uint32_t& lo, uint32_t& hi) {
    if (e.seed_ptr) { lo = e.seed_ptr[0]; hi = e.seed_ptr[1]; }
    else { lo = e.seed_lo; hi = e.seed_hi; }
}

// bias corrections of step t, as plnlp_adam_multi_f32 forms them
inline void adam_bias_corrections(float beta1, float beta2, int64_t step, float* bc1, float* bc2_sqrt) {
    *bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    *bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
}

// one element of torch.optim.Adam (amsgrad=False) given the clipped / decayed gradient -- the ONE place this
// arithmetic lives: the optimiser kernels and the PLNLP_EPI_ADAM epilogue must produce the same bits
__device__ __forceinline__ void adam_update(float& pi, float gi, float& mi, float& vi, float step, float b1,
                                            float b2, float eps, float bc2_sqrt) {
    mi = mi + (1.f - b1) * (gi - mi);
    vi = fmaf(1.f - b2, gi * gi, b2 * vi);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step * (mi / denom);
}

inline int make_epi(const plnlp_epilogue* e, Epi* out, bool allow_adam = false, bool allow_rowdot = false) {
    Epi d{};
    d.keep_scale = 1.f;
    d.gate_scale = 1.f;
    if (e) {
        d.flags = e->flags;
        if (d.flags & PLNLP_EPI_ROWDOT) {
            if (!allow_rowdot) return PLNLP_E_UNSUPPORTED;
            if (!e->rowdot_w || !e->rowdot_out) return PLNLP_E_NULL;
            if ((uintptr_t)e->rowdot_w % 16) return PLNLP_E_ALIGN;
            d.rowdot_w = e->rowdot_w; d.rowdot_out = e->rowdot_out; d.rowdot_ld = e->rowdot_ld;
        }
        if (d.flags & PLNLP_EPI_ADAM) {
            if (!allow_adam || (d.flags & PLNLP_EPI_ACCUM)) return PLNLP_E_UNSUPPORTED;
            if (!e->adam_m || !e->adam_v) return PLNLP_E_NULL;
            if ((!e->adam_scalars && e->adam_step < 1) || ((uintptr_t)e->adam_m % 16) || ((uintptr_t)e->adam_v % 16))
                return PLNLP_E_SHAPE;
            d.adam_m = e->adam_m; d.adam_v = e->adam_v;
            d.adam_lr = e->adam_lr; d.adam_b1 = e->adam_beta1; d.adam_b2 = e->adam_beta2; d.adam_eps = e->adam_eps;
            d.adam_scalars = e->adam_scalars;
            d.adam_bc1 = d.adam_bc2_sqrt = 1.f;
            if (!e->adam_scalars)
                adam_bias_corrections(e->adam_beta1, e->adam_beta2, e->adam_step, &d.adam_bc1, &d.adam_bc2_sqrt);
        }
        if (d.flags & PLNLP_EPI_BIAS) { if (!e->bias) return PLNLP_E_NULL; d.bias = e->bias; }
        if (d.flags & PLNLP_EPI_GATE) {
            if (!e->gate) return PLNLP_E_NULL;
            d.gate = e->gate; d.ld_gate = e->ld_gate; d.gate_scale = e->gate_scale;
            d.gate_index = e->gate_index;
        }
        if (d.flags & PLNLP_EPI_ADDEND) {
            if (!e->addend) return PLNLP_E_NULL;
            d.addend = e->addend; d.ld_addend = e->ld_addend; d.addend_index = e->addend_index;
        }
        if (d.flags & PLNLP_EPI_DROPOUT) {
            if (!(e->dropout_p >= 0.f) || e->dropout_p >= 1.f) return PLNLP_E_SHAPE;
            if (e->dropout_p == 0.f) d.flags &= ~PLNLP_EPI_DROPOUT;
            d.thresh = dropout_thresh(e->dropout_p);
            d.seed_lo = (uint32_t)e->dropout_seed;
            d.seed_hi = (uint32_t)(e->dropout_seed >> 32);
            d.seed_ptr = reinterpret_cast<const uint32_t*>(e->dropout_seed_ptr);
            d.keep_scale = 1.f / (1.f - e->dropout_p);
            d.drop_row = e->dropout_row_index;
        }
    }
    d.vec4 = (!(d.flags & PLNLP_EPI_BIAS) || ((uintptr_t)d.bias % 16 == 0)) &&
             (!(d.flags & PLNLP_EPI_GATE) || (((uintptr_t)d.gate % 16 == 0) && (d.ld_gate % 4 == 0))) &&
             (!(d.flags & PLNLP_EPI_ADDEND) || (((uintptr_t)d.addend % 16 == 0) && (d.ld_addend % 4 == 0)));
    *out = d;
    return 0;
}

// apply to one value at (row r, column f) of an [*, n_cols] result
__device__ __forceinline__ float epi_apply(const Epi& e, float v, int64_t r, int64_t f,
                                           int64_t n_cols, float prev) {
    if (e.flags & PLNLP_EPI_BIAS) v += e.bias[f];
    if (e.flags & PLNLP_EPI_RELU) v = fmaxf(v, 0.f);
    if (e.flags & PLNLP_EPI_DROPOUT) {
        const uint64_t dr = e.drop_row ? (uint64_t)e.drop_row[r] : (uint64_t)r;
        uint32_t s_lo, s_hi;
        epi_seed(e, s_lo, s_hi);
        v = dropout_keep(dr * (uint64_t)n_cols + (uint64_t)f, s_lo, s_hi, e.thresh) ? v * e.keep_scale : 0.f;
    }
    if (e.flags & PLNLP_EPI_ACCUM) v += prev;
    if (e.flags & PLNLP_EPI_ADDEND) {
        const int64_t a = e.addend_index ? (int64_t)e.addend_index[r] : r;
        if (a >= 0) v += e.addend[a * e.ld_addend + f];
    }
    if (e.flags & PLNLP_EPI_GATE) {
        const int64_t gr = e.gate_index ? (int64_t)e.gate_index[r] : r;
        v = e.gate[gr * e.ld_gate + f] > 0.f ? v * e.gate_scale : 0.f;
    }
    return v;
}

__device__ __forceinline__ float4 epi_apply4(const Epi& e, float4 v, int64_t r, int64_t f,
                                             int64_t n_cols, const float* out_row) {
    if (e.flags == 0) return v;
    float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e.flags & PLNLP_EPI_ACCUM) prev = *reinterpret_cast<const float4*>(out_row + f);
    if (e.vec4) {   // bias / gate rows are 16-byte loadable: one wide load each instead of four
        if (e.flags & PLNLP_EPI_BIAS) {
            const float4 b = *reinterpret_cast<const float4*>(e.bias + f);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (e.flags & PLNLP_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (e.flags & PLNLP_EPI_DROPOUT) {
            const uint64_t dr = e.drop_row ? (uint64_t)e.drop_row[r] : (uint64_t)r;
            uint32_t s_lo, s_hi;
            epi_seed(e, s_lo, s_hi);
            v = dropout_apply4(v, dr * (uint64_t)n_cols + (uint64_t)f, s_lo, s_hi, e.thresh, e.keep_scale);
        }
        if (e.flags & PLNLP_EPI_ACCUM) { v.x += prev.x; v.y += prev.y; v.z += prev.z; v.w += prev.w; }
        if (e.flags & PLNLP_EPI_ADDEND) {
            const int64_t a = e.addend_index ? (int64_t)e.addend_index[r] : r;
            if (a >= 0) {
                const float4 y = *reinterpret_cast<const float4*>(e.addend + a * e.ld_addend + f);
                v.x += y.x; v.y += y.y; v.z += y.z; v.w += y.w;
            }
        }
        if (e.flags & PLNLP_EPI_GATE) {
            const int64_t gr = e.gate_index ? (int64_t)e.gate_index[r] : r;
            const float4 y = *reinterpret_cast<const float4*>(e.gate + gr * e.ld_gate + f);
            v.x = y.x > 0.f ? v.x * e.gate_scale : 0.f; v.y = y.y > 0.f ? v.y * e.gate_scale : 0.f;
            v.z = y.z > 0.f ? v.z * e.gate_scale : 0.f; v.w = y.w > 0.f ? v.w * e.gate_scale : 0.f;
        }
        return v;
    }
    v.x = epi_apply(e, v.x, r, f + 0, n_cols, prev.x);
    v.y = epi_apply(e, v.y, r, f + 1, n_cols, prev.y);
    v.z = epi_apply(e, v.z, r, f + 2, n_cols, prev.z);
    v.w = epi_apply(e, v.w, r, f + 3, n_cols, prev.w);
    return v;
}

// vec4 form with the operands already in registers (the GEMM epilogue loads bias once per thread and
// gate / accumulate operands ahead of the loop that consumes them, so their latency is not exposed
// between the LDS read and the store of every output row).  `add` is the addend row when has_add.
__device__ __forceinline__ float4 epi_apply4_pre(const Epi& e, float4 v, int64_t r, int64_t f, int64_t n_cols,
                                                 float4 b, float4 y, float4 prev,
                                                 float4 add = make_float4(0.f, 0.f, 0.f, 0.f), bool has_add = false) {
    if (e.flags & PLNLP_EPI_BIAS) { v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    if (e.flags & PLNLP_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (e.flags & PLNLP_EPI_DROPOUT) {
        const uint64_t dr = e.drop_row ? (uint64_t)e.drop_row[r] : (uint64_t)r;
        uint32_t s_lo, s_hi;
        epi_seed(e, s_lo, s_hi);
        v = dropout_apply4(v, dr * (uint64_t)n_cols + (uint64_t)f, s_lo, s_hi, e.thresh, e.keep_scale);
    }
    if (e.flags & PLNLP_EPI_ACCUM) { v.x += prev.x; v.y += prev.y; v.z += prev.z; v.w += prev.w; }
    if (has_add) { v.x += add.x; v.y += add.y; v.z += add.z; v.w += add.w; }
    if (e.flags & PLNLP_EPI_GATE) {
        v.x = y.x > 0.f ? v.x * e.gate_scale : 0.f; v.y = y.y > 0.f ? v.y * e.gate_scale : 0.f;
        v.z = y.z > 0.f ? v.z * e.gate_scale : 0.f; v.w = y.w > 0.f ? v.w * e.gate_scale : 0.f;
    }
    return v;
}

// PLNLP_EPI_ADAM: g = the finished gradient of out[r, f .. f+3]; p = &out[r, f], off = r * ldo + f (m, v share
// out's leading dimension).  The moments are touched once per step: streaming hints on them.
__device__ __forceinline__ void epi_adam4(const Epi& e, float4 g, float* __restrict__ p, int64_t off) {
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    float4 p4 = *reinterpret_cast<float4*>(p);
    const f32x4n mn = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(e.adam_m + off));
    const f32x4n vn = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(e.adam_v + off));
    float4 m4 = make_float4(mn.x, mn.y, mn.z, mn.w), v4 = make_float4(vn.x, vn.y, vn.z, vn.w);
    float lr = e.adam_lr, bc1 = e.adam_bc1, bc2_sqrt = e.adam_bc2_sqrt;
    if (e.adam_scalars) { lr = e.adam_scalars[0]; bc1 = e.adam_scalars[1]; bc2_sqrt = e.adam_scalars[2]; }
    const float step = lr / bc1;
    adam_update(p4.x, g.x, m4.x, v4.x, step, e.adam_b1, e.adam_b2, e.adam_eps, bc2_sqrt);
    adam_update(p4.y, g.y, m4.y, v4.y, step, e.adam_b1, e.adam_b2, e.adam_eps, bc2_sqrt);
    adam_update(p4.z, g.z, m4.z, v4.z, step, e.adam_b1, e.adam_b2, e.adam_eps, bc2_sqrt);
    adam_update(p4.w, g.w, m4.w, v4.w, step, e.adam_b1, e.adam_b2, e.adam_eps, bc2_sqrt);
    *reinterpret_cast<float4*>(p) = p4;
    const f32x4n mo = {m4.x, m4.y, m4.z, m4.w}, vo = {v4.x, v4.y, v4.z, v4.w};
    __builtin_nontemporal_store(mo, reinterpret_cast<f32x4n*>(e.adam_m + off));
    __builtin_nontemporal_store(vo, reinterpret_cast<f32x4n*>(e.adam_v + off));
}

// True in exactly ONE workgroup of the launch -- the last to arrive -- once every workgroup's earlier global writes are
// visible to it: the "last block finishes the reduction" pattern that saves the follow-up launch of a one-block
// kernel (every dependent launch costs 5-9 us on this part whatever it does).  *counter must be 0 at launch; the
// winner puts it back to 0, so a caller keeps ONE persistent zeroed word per use site and stream.
// (release: each thread's writes -> __syncthreads -> thread 0's device-scope fence, which on a multi-XCD part writes
// this XCD's L2 back; acquire: the winner's fence invalidates what its own L2 / L1 hold of other XCDs' lines.)
__device__ __forceinline__ bool last_workgroup(unsigned int* counter, unsigned int n_groups) {
    __shared__ int last_flag;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned int ticket = atomicAdd(counter, 1u);
        last_flag = ticket == n_groups - 1u;
        if (last_flag) *counter = 0u;
    }
    __syncthreads();
    const bool last = last_flag != 0;
    if (last) __threadfence();
    return last;
}

// ---- which kernel families this process has launched (plnlp_launch_counts; csrc/launch_log.hip).  Diagnostics only: the
// trained-regime tests assert with them that the forms the benchmark runs are the forms they exercised.
enum LaunchKind : int {
    LK_GEMM_X3S = 0,          // split-bf16, stationary pre-split weights (gemm_x3s.hip)
    LK_GEMM_TILE_X3,          // split-bf16, 128 x 128 tile kernel
    LK_GEMM_TILE_F32,         // f32-input MFMA, 128 x 128 tile kernel
    LK_GEMM_SPLITK_REDUCE,    // the split-K slices' reduction
    LK_AGG_VEC,               // aggregation: one wave per row, full width
    LK_AGG_VEC_SLABS,         //   one wave per (row, 128- / 256-column slab)
    LK_AGG_VEC_XCD,           //   one wave per (row, F/8-column slab pinned to an XCD)
    LK_AGG_FUSED,             //   main pass + the long rows' chunk pass in ONE launch
    LK_AGG_FUSED_HUB_XCD,     //   ... with the chunks in eight XCD-pinned column slabs
    LK_AGG_CHUNK,             //   the long rows' chunk pass as its own launch
    LK_AGG_CHUNK_XCD,         //   ... in XCD-pinned slabs
    LK_AGG_FINALIZE,          //   the long rows' partial sums -> result rows
    LK_AGG_LDS,               //   LDS-staged feature slabs (small dense graphs)
    LK_AGG_SCALAR,            //   any width / alignment
    LK_GEMM_WGRAD_WIDE,       // split-bf16 weight gradient, the whole (<= 224 x 224) result per workgroup (gemm_wgw.hip)
    LK_GEMM_X3B,              // split-bf16, stationary pre-split weights, a 256-row block per workgroup (gemm_x3b.hip)
    LK_AGG_DENSE,             // aggregation of a dense graph as a bf16-counts x split-bf16 product (aggregate_dense.hip)
    LK_COUNT
};
void count_launch(int kind);

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace plnlp
