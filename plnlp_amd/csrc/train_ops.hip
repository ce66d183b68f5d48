// K5 pairwise loss (forward+backward fused), optimiser step, column sums and the
// small element-wise helpers.  All reductions are two-stage with a fixed order
// (block partials, then one block adds the partials in index order) so a step is
// bit-reproducible; nothing here syncs with the host.
#include "common.hip.h"

namespace plnlp {

// block-wide sum in a fixed order: wave shuffle tree, then wave 0 adds the 4 wave sums.
__device__ __forceinline__ float block_sum_256(float v, float* smem4) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) smem4[wave] = v;
    __syncthreads();
    float r = smem4[0] + smem4[1] + smem4[2] + smem4[3];
    __syncthreads();
    return r;
}

// ---------------- pairwise loss ---------------------------------------------------
__device__ __forceinline__ void pair_term(int kind, float d, float w, float& l, float& dl) {
    // returns loss term l and dl/dd
    switch (kind) {
        case PLNLP_LOSS_AUC: { float t = 1.f - d; l = t * t; dl = -2.f * t; } break;
        case PLNLP_LOSS_HINGE_AUC: { float t = fmaxf(1.f - d, 0.f); l = t * t; dl = -2.f * t; } break;
        case PLNLP_LOSS_WEIGHTED_AUC: { float t = 1.f - d; l = w * (t * t); dl = -2.f * w * t; } break;
        case PLNLP_LOSS_ADAPTIVE_AUC: { float t = w - d; l = t * t; dl = -2.f * t; } break;
        case PLNLP_LOSS_WEIGHTED_HINGE_AUC: { float t = fmaxf(w - d, 0.f); l = w * (t * t); dl = -2.f * w * t; } break;
        case PLNLP_LOSS_ADAPTIVE_HINGE_AUC: { float t = fmaxf(w - d, 0.f); l = t * t; dl = -2.f * t; } break;
        default: {  // LOG_RANK: -log(sigmoid(d) + 1e-15)
            float s = 1.f / (1.f + expf(-d));
            l = -logf(s + 1e-15f);
            dl = -(s * (1.f - s)) / (s + 1e-15f);
        } break;
    }
}

__global__ __launch_bounds__(256) void pairwise_loss_kernel(int kind, const float* __restrict__ pos,
                                                            const float* __restrict__ neg,
                                                            const float* __restrict__ weight, int64_t batch,
                                                            int num_neg, float term_scale, float grad_scale,
                                                            float* __restrict__ gpos, float* __restrict__ gneg,
                                                            float* __restrict__ partial,
                                                            unsigned int* __restrict__ counter,
                                                            float* __restrict__ loss, double* __restrict__ loss_acc,
                                                            double acc_weight) {
    __shared__ float sm[4];
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float lsum = 0.f;
    if (b < batch) {
        const float p = pos[b];
        const float w = weight ? weight[b] : 1.f;
        float gp = 0.f;
        for (int n = 0; n < num_neg; ++n) {
            const float q = neg[b * num_neg + n];
            float l, dl;
            pair_term(kind, p - q, w, l, dl);
            lsum += l;
            gp += dl;
            gneg[b * num_neg + n] = -dl * term_scale * grad_scale;
        }
        gpos[b] = gp * term_scale * grad_scale;
    }
    const float tot = block_sum_256(lsum, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot * term_scale;
    if (!counter) return;                   // two-launch form: sum_partials_kernel follows
    // the last workgroup adds the partials exactly as sum_partials_kernel does (same order, same bits) ...
    if (!last_workgroup(counter, gridDim.x)) return;
    float v = 0.f;
    for (int64_t i = threadIdx.x; i < (int64_t)gridDim.x; i += 256) v += partial[i];
    const float total = block_sum_256(v, sm);
    if (threadIdx.x == 0) {
        loss[0] = total;
        // ... and feeds the epoch's running sum (model.py:169: total_loss += loss.item() * num_examples, in double)
        if (loss_acc) loss_acc[0] += (double)total * acc_weight;
    }
}

// one block: out = (accumulate ? out : 0) + sum_i partial[i], fixed order
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int64_t n,
                                                           float* __restrict__ out, int accumulate) {
    __shared__ float sm[4];
    float v = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) v += partial[i];
    const float tot = block_sum_256(v, sm);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + tot : tot;
}

// ---------------- sum of squares --------------------------------------------------
constexpr int64_t SQ_CHUNK = 256 * 4 * 16;  // elements per block
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, int64_t n,
                                                     float* __restrict__ partial) {
    __shared__ float sm[4];
    const int64_t beg = (int64_t)blockIdx.x * SQ_CHUNK;
    const int64_t end = beg + SQ_CHUNK < n ? beg + SQ_CHUNK : n;
    float v = 0.f;
    if (((uintptr_t)g % 16) == 0) {
        int64_t i = beg + (int64_t)threadIdx.x * 4;
        // (four loads in flight per thread: a 16-iteration chain of dependent round trips was most of this kernel's
        //  12 us on the 131 K-element weight set; the order of the fused multiply-adds is unchanged)
        for (; i + 3 + 3 * 1024 < end; i += 4 * 1024) {
            const float4 x0 = *reinterpret_cast<const float4*>(g + i);
            const float4 x1 = *reinterpret_cast<const float4*>(g + i + 1024);
            const float4 x2 = *reinterpret_cast<const float4*>(g + i + 2048);
            const float4 x3 = *reinterpret_cast<const float4*>(g + i + 3072);
            v = fmaf(x0.x, x0.x, v); v = fmaf(x0.y, x0.y, v); v = fmaf(x0.z, x0.z, v); v = fmaf(x0.w, x0.w, v);
            v = fmaf(x1.x, x1.x, v); v = fmaf(x1.y, x1.y, v); v = fmaf(x1.z, x1.z, v); v = fmaf(x1.w, x1.w, v);
            v = fmaf(x2.x, x2.x, v); v = fmaf(x2.y, x2.y, v); v = fmaf(x2.z, x2.z, v); v = fmaf(x2.w, x2.w, v);
            v = fmaf(x3.x, x3.x, v); v = fmaf(x3.y, x3.y, v); v = fmaf(x3.z, x3.z, v); v = fmaf(x3.w, x3.w, v);
        }
        for (; i + 3 < end; i += 1024) {
            const float4 x = *reinterpret_cast<const float4*>(g + i);
            v = fmaf(x.x, x.x, v); v = fmaf(x.y, x.y, v); v = fmaf(x.z, x.z, v); v = fmaf(x.w, x.w, v);
        }
        for (; i < end; ++i) { if (i < end) v = fmaf(g[i], g[i], v); }
    } else {
        for (int64_t i = beg + threadIdx.x; i < end; i += 256) v = fmaf(g[i], g[i], v);
    }
    const float tot = block_sum_256(v, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// ---------------- Adam -------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr, float b1, float b2, float eps, float wd,
                                                   int decoupled, float bc1, float bc2_sqrt,
                                                   const float* __restrict__ sqnorm, float max_norm,
                                                   float grad_scale) {
    float coef = grad_scale;
    if (sqnorm) {
        const float c = max_norm / (sqrtf(sqnorm[0]) + 1e-6f);
        coef *= c < 1.f ? c : 1.f;
    }
    const float step = lr / bc1;
    auto update = [&](float& pi, float gi, float& mi, float& vi) {
        gi *= coef;
        if (wd != 0.f) { if (decoupled) pi *= (1.f - lr * wd); else gi = fmaf(wd, pi, gi); }
        adam_update(pi, gi, mi, vi, step, b1, b2, eps, bc2_sqrt);      // (common.hip.h: shared with PLNLP_EPI_ADAM)
    };
    int64_t done = 0;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16) == 0) {   // 16 bytes per lane
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            float4 p4 = reinterpret_cast<float4*>(p)[i];
            const float4 g4 = reinterpret_cast<const float4*>(g)[i];
            float4 m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
            update(p4.x, g4.x, m4.x, v4.x); update(p4.y, g4.y, m4.y, v4.y);
            update(p4.z, g4.z, m4.z, v4.z); update(p4.w, g4.w, m4.w, v4.w);
            reinterpret_cast<float4*>(p)[i] = p4; reinterpret_cast<float4*>(m)[i] = m4; reinterpret_cast<float4*>(v)[i] = v4;
        }
        done = n4 << 2;
    }
    for (int64_t i = done + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float pi = p[i], mi = m[i], vi = v[i];
        update(pi, g[i], mi, vi);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// ---------------- the same over a list of tensors, one launch ---------------------------
struct MultiSq {
    const float* g[PLNLP_MULTI_MAX];
    int64_t      n[PLNLP_MULTI_MAX];
    int          first_block[PLNLP_MULTI_MAX + 1];   // block range of tensor i = its range of partials
    int          count;
};
__global__ __launch_bounds__(256) void sqnorm_multi_kernel(MultiSq a, float* __restrict__ partial,
                                                           unsigned int* __restrict__ counter,
                                                           float* __restrict__ out) {
    __shared__ float sm[4];
    int ti = 0;
#pragma unroll
    for (int i = 1; i < PLNLP_MULTI_MAX; ++i) ti += (i < a.count && (int)blockIdx.x >= a.first_block[i]) ? 1 : 0;
    const float* g = a.g[0];
    int64_t n = a.n[0];
    int fb = 0;
#pragma unroll
    for (int i = 1; i < PLNLP_MULTI_MAX; ++i) if (i == ti) { g = a.g[i]; n = a.n[i]; fb = a.first_block[i]; }
    const int64_t beg = (int64_t)((int)blockIdx.x - fb) * SQ_CHUNK;
    const int64_t end = beg + SQ_CHUNK < n ? beg + SQ_CHUNK : n;
    float v = 0.f;
    if (((uintptr_t)g % 16) == 0) {
        int64_t i = beg + (int64_t)threadIdx.x * 4;
        // (four loads in flight per thread: a 16-iteration chain of dependent round trips was most of this kernel's
        //  12 us on the 131 K-element weight set; the order of the fused multiply-adds is unchanged)
        for (; i + 3 + 3 * 1024 < end; i += 4 * 1024) {
            const float4 x0 = *reinterpret_cast<const float4*>(g + i);
            const float4 x1 = *reinterpret_cast<const float4*>(g + i + 1024);
            const float4 x2 = *reinterpret_cast<const float4*>(g + i + 2048);
            const float4 x3 = *reinterpret_cast<const float4*>(g + i + 3072);
            v = fmaf(x0.x, x0.x, v); v = fmaf(x0.y, x0.y, v); v = fmaf(x0.z, x0.z, v); v = fmaf(x0.w, x0.w, v);
            v = fmaf(x1.x, x1.x, v); v = fmaf(x1.y, x1.y, v); v = fmaf(x1.z, x1.z, v); v = fmaf(x1.w, x1.w, v);
            v = fmaf(x2.x, x2.x, v); v = fmaf(x2.y, x2.y, v); v = fmaf(x2.z, x2.z, v); v = fmaf(x2.w, x2.w, v);
            v = fmaf(x3.x, x3.x, v); v = fmaf(x3.y, x3.y, v); v = fmaf(x3.z, x3.z, v); v = fmaf(x3.w, x3.w, v);
        }
        for (; i + 3 < end; i += 1024) {
            const float4 x = *reinterpret_cast<const float4*>(g + i);
            v = fmaf(x.x, x.x, v); v = fmaf(x.y, x.y, v); v = fmaf(x.z, x.z, v); v = fmaf(x.w, x.w, v);
        }
        for (; i < end; ++i) { if (i < end) v = fmaf(g[i], g[i], v); }
    } else {
        for (int64_t i = beg + threadIdx.x; i < end; i += 256) v = fmaf(g[i], g[i], v);
    }
    const float tot = block_sum_256(v, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
    if (!counter) return;
    if (!last_workgroup(counter, gridDim.x)) return;        // (as in pairwise_loss_kernel: sum_partials_kernel's sum)
    float w = 0.f;
    for (int64_t i = threadIdx.x; i < (int64_t)gridDim.x; i += 256) w += partial[i];
    const float total = block_sum_256(w, sm);
    if (threadIdx.x == 0) out[0] = total;
}

struct MultiAdam {
    float*       p[PLNLP_MULTI_MAX];
    const float* g[PLNLP_MULTI_MAX];
    float*       m[PLNLP_MULTI_MAX];
    float*       v[PLNLP_MULTI_MAX];
    int64_t      n[PLNLP_MULTI_MAX];
    const float* sqnorm[PLNLP_MULTI_MAX];
    float        max_norm[PLNLP_MULTI_MAX];
    float        bc1[PLNLP_MULTI_MAX], bc2_sqrt[PLNLP_MULTI_MAX];
    const float* scal[PLNLP_MULTI_MAX];      // nullable: {lr, bc1, bc2_sqrt} of this step in device memory
    int          first_block[PLNLP_MULTI_MAX + 1];
    int          count;
};
__global__ __launch_bounds__(256) void adam_multi_kernel(MultiAdam a, float lr, float b1, float b2, float eps,
                                                         float wd, int decoupled, float grad_scale) {
    int ti = 0;
#pragma unroll
    for (int i = 1; i < PLNLP_MULTI_MAX; ++i) ti += (i < a.count && (int)blockIdx.x >= a.first_block[i]) ? 1 : 0;
    float* p = a.p[0]; const float* g = a.g[0]; float* m = a.m[0]; float* v = a.v[0];
    int64_t n = a.n[0];
    const float* sqnorm = a.sqnorm[0];
    float max_norm = a.max_norm[0], bc1 = a.bc1[0], bc2_sqrt = a.bc2_sqrt[0];
    const float* scal = a.scal[0];
    int fb = 0, nb = a.first_block[1];
#pragma unroll
    for (int i = 1; i < PLNLP_MULTI_MAX; ++i)
        if (i == ti) {      // selects, not a runtime-indexed struct read (that would go through scratch)
            p = a.p[i]; g = a.g[i]; m = a.m[i]; v = a.v[i]; n = a.n[i]; sqnorm = a.sqnorm[i];
            max_norm = a.max_norm[i]; bc1 = a.bc1[i]; bc2_sqrt = a.bc2_sqrt[i]; scal = a.scal[i];
            fb = a.first_block[i]; nb = a.first_block[i + 1];
        }
    if (scal) { lr = scal[0]; bc1 = scal[1]; bc2_sqrt = scal[2]; }
    float coef = grad_scale;
    if (sqnorm) {
        const float c = max_norm / (sqrtf(sqnorm[0]) + 1e-6f);
        coef *= c < 1.f ? c : 1.f;
    }
    const float step = lr / bc1;
    auto update = [&](float& pi, float gi, float& mi, float& vi) {
        gi *= coef;
        if (wd != 0.f) { if (decoupled) pi *= (1.f - lr * wd); else gi = fmaf(wd, pi, gi); }
        adam_update(pi, gi, mi, vi, step, b1, b2, eps, bc2_sqrt);      // (common.hip.h: shared with PLNLP_EPI_ADAM)
    };
    const int64_t tid = (int64_t)((int)blockIdx.x - fb) * 256 + threadIdx.x;
    const int64_t nthreads = (int64_t)(nb - fb) * 256;
    int64_t done = 0;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16) == 0) {
        const int64_t n4 = n >> 2;
        // the gradient and the two moments are touched once per step: streaming hints on them, so that what the
        // caches keep after this kernel is the parameter itself -- the table the next step's first aggregation gathers
        typedef float f32x4n __attribute__((ext_vector_type(4)));
        for (int64_t i = tid; i < n4; i += nthreads) {
            float4 p4 = reinterpret_cast<float4*>(p)[i];
#ifndef PLNLP_ADAM_PLAIN_ACCESS
            const f32x4n gn = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(g) + i);
            const f32x4n mn = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(m) + i);
            const f32x4n vn = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(v) + i);
            const float4 g4 = make_float4(gn.x, gn.y, gn.z, gn.w);
            float4 m4 = make_float4(mn.x, mn.y, mn.z, mn.w), v4 = make_float4(vn.x, vn.y, vn.z, vn.w);
#else
            const float4 g4 = reinterpret_cast<const float4*>(g)[i];
            float4 m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i];
#endif
            update(p4.x, g4.x, m4.x, v4.x); update(p4.y, g4.y, m4.y, v4.y);
            update(p4.z, g4.z, m4.z, v4.z); update(p4.w, g4.w, m4.w, v4.w);
            reinterpret_cast<float4*>(p)[i] = p4;
#ifndef PLNLP_ADAM_PLAIN_ACCESS
            const f32x4n mo = {m4.x, m4.y, m4.z, m4.w}, vo = {v4.x, v4.y, v4.z, v4.w};
            __builtin_nontemporal_store(mo, reinterpret_cast<f32x4n*>(m) + i);
            __builtin_nontemporal_store(vo, reinterpret_cast<f32x4n*>(v) + i);
#else
            reinterpret_cast<float4*>(m)[i] = m4; reinterpret_cast<float4*>(v)[i] = v4;
#endif
        }
        done = n4 << 2;
    }
    for (int64_t i = done + tid; i < n; i += nthreads) {
        float pi = p[i], mi = m[i], vi = v[i];
        update(pi, g[i], mi, vi);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// scale a gradient in place by the clip coefficient (used when the optimiser is not ours)
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, int64_t n,
                                                         const float* __restrict__ sqnorm, float max_norm) {
    const float c = max_norm / (sqrtf(sqnorm[0]) + 1e-6f);
    const float coef = c < 1.f ? c : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) g[i] *= coef;
}

// ---------------- column sums -------------------------------------------------------
// stage 1: CS_BLOCKS blocks stride over the rows, thread = column (coalesced row reads);
// stage 2: one wave per column adds the block partials (shuffle tree: fixed order).
constexpr int CS_BLOCKS = 1024;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int64_t ldx,
                                                             int64_t n_rows, int feat,
                                                             const float* __restrict__ rw,
                                                             float* __restrict__ partial) {
    for (int f = threadIdx.x; f < feat; f += 256) {
        float acc = 0.f;
        for (int64_t r = blockIdx.x; r < n_rows; r += gridDim.x)
            acc = rw ? fmaf(rw[r], x[r * ldx + f], acc) : acc + x[r * ldx + f];
        partial[(int64_t)blockIdx.x * feat + f] = acc;
    }
}
// 16-byte variant (feat % 4 == 0, feat <= 1024, aligned): a row is covered by feat/4 lanes, so a
// 256-thread block streams 256/(feat/4) rows per step with dwordx4 loads; the row lanes are then
// folded through LDS in a fixed order.
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const float* __restrict__ x, int64_t ldx,
                                                                 int64_t n_rows, int feat,
                                                                 const float* __restrict__ rw,
                                                                 float* __restrict__ partial) {
    __shared__ float4 sm[256];
    const int q = feat >> 2;              // float4 per row
    const int lanes = 256 / q;            // rows per step; threads past lanes*q idle (e.g. feat = 200: 250 busy)
    const int c4 = threadIdx.x % q, rl = threadIdx.x / q;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r = (int64_t)blockIdx.x * lanes + rl; rl < lanes && r < n_rows; r += (int64_t)gridDim.x * lanes) {
        const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c4 * 4);
        const float w = rw ? rw[r] : 1.f;
        acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y);
        acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
    }
    if (rl < lanes) sm[rl * q + c4] = acc;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < lanes; ++k) {
            const float4 v = sm[k * q + c4];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * feat + c4 * 4) = acc;
    }
}
// feat == 1 (the gradient of a 1-output head's bias: the column sum of a [rows, 1] vector): colsum_partial_kernel leaves
// 255 of every 256 threads idle (81 us for 262 144 rows).  Same partial sums in the same order -- partial[b] = x[b] +
// x[b + blocks] + ... sequentially -- with thread = b: consecutive threads read consecutive addresses.
constexpr int COL1_U = 32;
__global__ __launch_bounds__(256) void colsum_partial_col1_kernel(const float* __restrict__ x, int64_t ldx, int64_t n_rows,
                                                                  const float* __restrict__ rw, int64_t n_blocks,
                                                                  float* __restrict__ partial) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blocks) return;
    float acc = 0.f;
    // (COL1_U loads in flight per thread, added in row order: the chain of dependent round trips -- 256 per thread for the ddi
    // scorer's 262 144 rows on four workgroups, ~3 us each -- was 98 us of the ddi step's main stream; eight in flight left 86)
    constexpr int U = COL1_U;
    for (int64_t r = b; r < n_rows; r += U * n_blocks) {
        float v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rr = r + u * n_blocks;
            const bool ok = rr < n_rows;
            v[u] = ok ? x[rr * ldx] : 0.f;
            w[u] = (ok && rw) ? rw[rr] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r + u * n_blocks < n_rows) acc = rw ? fmaf(w[u], v[u], acc) : acc + v[u];
    }
    partial[b] = acc;
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int64_t n_blocks,
                                                           int feat, float scale, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= feat) return;
    float acc = 0.f;
    for (int64_t b = lane; b < n_blocks; b += 64) acc += partial[b * feat + f];
    acc = wave_sum(acc);
    if (lane == 0) out[f] = acc * scale;
}

// ---------------- element-wise ------------------------------------------------------
__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ g, const float* __restrict__ gate,
                                                   float scale, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = gate[i] > 0.f ? g[i] * scale : 0.f;
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      int64_t n, uint32_t thresh, uint32_t seed_lo,
                                                      uint32_t seed_hi, float keep_scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = dropout_keep((uint64_t)i, seed_lo, seed_hi, thresh) ? x[i] * keep_scale : 0.f;
}
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, int64_t ldx,
                                                        float* __restrict__ y, int64_t ldy, int64_t n_rows,
                                                        int64_t n_cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < n_rows && c0 + tx < n_cols) tile[i][tx] = x[(r0 + i) * ldx + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < n_cols && r0 + tx < n_rows) y[(c0 + i) * ldy + r0 + tx] = tile[tx][i];
}

// ---------------- single-output linear head ----------------------------------------------
// one wave per row, 16 B per lane (float4) when aligned, shuffle tree reduction (fixed order)
template <bool VEC>
__global__ __launch_bounds__(256) void matvec_kernel(const float* __restrict__ x, int64_t ldx, int64_t n_rows,
                                                     int feat, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const float b = bias ? bias[0] : 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_rows; r += (int64_t)gridDim.x * 4) {
        const float* row = x + r * ldx;
        float acc = 0.f;
        if constexpr (VEC) {
            for (int s = lane; s < (feat >> 2); s += 64) {
                const float4 v = reinterpret_cast<const float4*>(row)[s];
                const float4 u = reinterpret_cast<const float4*>(w)[s];
                acc = fmaf(v.x, u.x, acc); acc = fmaf(v.y, u.y, acc);
                acc = fmaf(v.z, u.z, acc); acc = fmaf(v.w, u.w, acc);
            }
        } else {
            for (int f = lane; f < feat; f += 64) acc = fmaf(row[f], w[f], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) out[r] = acc + b;
    }
}

__global__ __launch_bounds__(256) void outer_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                    int64_t n_rows, int feat, float* __restrict__ dx,
                                                    int64_t lddx, Epi epi) {
    const int64_t total = n_rows * (int64_t)feat;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / feat;
        const int f = (int)(i - r * feat);
        float* p = dx + r * lddx + f;
        float v = g[r] * w[f];
        if (epi.flags) v = epi_apply(epi, v, r, f, feat, (epi.flags & PLNLP_EPI_ACCUM) ? *p : 0.f);
        *p = v;
    }
}
// 16-byte form: a wave covers one row per step (feat/4 lanes busy), w stays in registers
__global__ __launch_bounds__(256) void outer_vec_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                        int64_t n_rows, int feat, float* __restrict__ dx,
                                                        int64_t lddx, Epi epi) {
    const int lane = threadIdx.x & 63;
    const int nslots = feat >> 2;
    for (int s0 = 0; s0 < nslots; s0 += 64) {
        const int s = s0 + lane;
        const float4 wv = s < nslots ? reinterpret_cast<const float4*>(w)[s] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n_rows; r += (int64_t)gridDim.x * 4) {
            if (s >= nslots) continue;
            const float gr = g[r];
            float* orow = dx + r * lddx;
            float4 v = make_float4(gr * wv.x, gr * wv.y, gr * wv.z, gr * wv.w);
            v = epi_apply4(epi, v, r, (int64_t)s * 4, feat, orow);
            reinterpret_cast<float4*>(orow)[s] = v;
        }
    }
}

// ---------------- the backward of a 1-output head, in one pass -------------------------
// MLPPredictor's last linear has ONE output (layer.py:73, 86): score[r] = <a[r, :], w> + b, a = dropout(relu(z)) the
// hidden activation.  Given g[r] = d loss / d score[r], everything its backward needs from `a` comes out of ONE read of it:
//     dz[r, f]  = a[r, f] > 0 ? g[r] w[f] gate_scale : 0        gradient of the hidden pre-activation (outer product + gate)
//     dw[f]     = sum_r g[r] a[r, f]                             the head's weight gradient
//     dbp[f]    = sum_r dz[r, f]                                 the hidden layer's bias gradient
// instead of three passes (outer product, two column sums over [rows, feat]): 262 144 x 512 on ddi is 537 MB per pass.
// (The head's own bias gradient, sum_r g[r], stays plnlp_colsum_f32 on the [rows, 1] vector -- its exact value is ZERO for
// the pairwise losses, what comes out is round-off, and the trajectory fixtures are pinned to that kernel's association.)  Same row -> (block, lane) assignment and the same fold order as colsum_partial_vec_kernel, so dw and dbp are
// the bits plnlp_colsum_f32 gives; dz the bits of plnlp_outer_f32 with the gate epilogue.
__global__ __launch_bounds__(256) void mlp_head_bwd_kernel(const float* __restrict__ a, int64_t lda,
                                                           const float* __restrict__ g, const float* __restrict__ w,
                                                           float gate_scale, int64_t n_rows, int feat,
                                                           float* __restrict__ dz, int64_t lddz,
                                                           float* __restrict__ partial) {
    __shared__ float4 sm[2][256];
    const int q = feat >> 2;              // float4 per row
    const int lanes = 256 / q;            // rows per step
    const int c4 = threadIdx.x % q, rl = threadIdx.x / q;
    const int64_t stride = 2 * (int64_t)feat;
    float4 acc_w = make_float4(0.f, 0.f, 0.f, 0.f), acc_b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < lanes) {
        const float4 wv = *reinterpret_cast<const float4*>(w + c4 * 4);
        for (int64_t r = (int64_t)blockIdx.x * lanes + rl; r < n_rows; r += (int64_t)gridDim.x * lanes) {
            const float4 v = *reinterpret_cast<const float4*>(a + r * lda + c4 * 4);
            const float gr = g[r];
            float4 d = make_float4(gr * wv.x, gr * wv.y, gr * wv.z, gr * wv.w);
            d.x = v.x > 0.f ? d.x * gate_scale : 0.f; d.y = v.y > 0.f ? d.y * gate_scale : 0.f;
            d.z = v.z > 0.f ? d.z * gate_scale : 0.f; d.w = v.w > 0.f ? d.w * gate_scale : 0.f;
            *reinterpret_cast<float4*>(dz + r * lddz + c4 * 4) = d;
            acc_w.x = fmaf(gr, v.x, acc_w.x); acc_w.y = fmaf(gr, v.y, acc_w.y);
            acc_w.z = fmaf(gr, v.z, acc_w.z); acc_w.w = fmaf(gr, v.w, acc_w.w);
            acc_b.x += d.x; acc_b.y += d.y; acc_b.z += d.z; acc_b.w += d.w;
        }
        sm[0][rl * q + c4] = acc_w;
        sm[1][rl * q + c4] = acc_b;
    }
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < lanes; ++k) {
            const float4 v = sm[0][k * q + c4], u = sm[1][k * q + c4];
            acc_w.x += v.x; acc_w.y += v.y; acc_w.z += v.z; acc_w.w += v.w;
            acc_b.x += u.x; acc_b.y += u.y; acc_b.z += u.z; acc_b.w += u.w;
        }
        float* row = partial + (int64_t)blockIdx.x * stride;
        *reinterpret_cast<float4*>(row + c4 * 4) = acc_w;
        *reinterpret_cast<float4*>(row + feat + c4 * 4) = acc_b;
    }
}

// the tail of PLNLP_EPI_ROWDOT: out[r] = bias + sum of the column tiles' partial dot products, tiles in order
__global__ __launch_bounds__(256) void rowdot_finish_kernel(const float* __restrict__ partial, int64_t ld, int tiles,
                                                            int64_t n_rows, const float* __restrict__ bias,
                                                            float* __restrict__ out) {
    const float b = bias ? bias[0] : 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * 256) {
        float acc = partial[r];
        for (int t = 1; t < tiles; ++t) acc += partial[(int64_t)t * ld + r];
        out[r] = acc + b;
    }
}

static inline unsigned ew_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b < 2048 ? (b > 0 ? b : 1) : 2048);
}

}  // namespace plnlp

extern "C" int64_t plnlp_loss_workspace_floats(int64_t batch) { return (batch + 255) / 256 + 1; }

extern "C" int plnlp_pairwise_loss_tail_f32(int kind, const float* pos, const float* neg, const float* weight,
                                            int64_t batch, int64_t num_neg, float grad_scale, float* loss,
                                            float* gpos, float* gneg, float* workspace, int64_t workspace_floats,
                                            unsigned int* block_counter, double* loss_acc, double acc_weight,
                                            void* stream) {
    using namespace plnlp;
    if (!pos || !neg || !loss || !gpos || !gneg || !workspace) return PLNLP_E_NULL;
    if (loss_acc && !block_counter) return PLNLP_E_NULL;
    if (batch <= 0 || num_neg <= 0 || num_neg > 1 << 20) return PLNLP_E_SHAPE;
    if (kind < PLNLP_LOSS_AUC || kind > PLNLP_LOSS_LOG_RANK) return PLNLP_E_UNSUPPORTED;
    const bool needs_w = kind == PLNLP_LOSS_WEIGHTED_AUC || kind == PLNLP_LOSS_ADAPTIVE_AUC ||
                         kind == PLNLP_LOSS_WEIGHTED_HINGE_AUC || kind == PLNLP_LOSS_ADAPTIVE_HINGE_AUC;
    if (needs_w && !weight) return PLNLP_E_NULL;
    const int64_t blocks = (batch + 255) / 256;
    if (workspace_floats < blocks) return PLNLP_E_WORKSPACE;
    const float term_scale = kind == PLNLP_LOSS_LOG_RANK ? 1.f / (float)(batch * num_neg) : 1.f;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pairwise_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, s, kind, pos, neg,
                       needs_w ? weight : nullptr, batch, (int)num_neg, term_scale, grad_scale, gpos, gneg,
                       workspace, block_counter, loss, loss_acc, acc_weight);
    if (int rc = launch_status()) return rc;
    if (block_counter) return 0;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, workspace, blocks, loss, 0);
    return launch_status();
}

extern "C" int plnlp_pairwise_loss_f32(int kind, const float* pos, const float* neg, const float* weight, int64_t batch,
                                       int64_t num_neg, float grad_scale, float* loss, float* gpos, float* gneg,
                                       float* workspace, int64_t workspace_floats, void* stream) {
    return plnlp_pairwise_loss_tail_f32(kind, pos, neg, weight, batch, num_neg, grad_scale, loss, gpos, gneg, workspace,
                                        workspace_floats, nullptr, nullptr, 0.0, stream);
}

extern "C" int64_t plnlp_sqnorm_partials(int64_t n) { return n <= 0 ? 0 : (n + plnlp::SQ_CHUNK - 1) / plnlp::SQ_CHUNK; }

extern "C" int plnlp_sqnorm_f32(const float* g, int64_t n, float* partial, int64_t n_partial, void* stream) {
    using namespace plnlp;
    if (n < 0) return PLNLP_E_SHAPE;
    if (n == 0) return 0;
    if (!g || !partial) return PLNLP_E_NULL;
    const int64_t blocks = plnlp_sqnorm_partials(n);
    if (n_partial < blocks) return PLNLP_E_WORKSPACE;
    hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, partial);
    return launch_status();
}

extern "C" int plnlp_sum_partials_f32(const float* partial, int64_t n, float* out, int accumulate, void* stream) {
    using namespace plnlp;
    if (!out || (n > 0 && !partial)) return PLNLP_E_NULL;
    if (n < 0) return PLNLP_E_SHAPE;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, out, accumulate);
    return launch_status();
}

extern "C" int plnlp_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   float lr, float beta1, float beta2, float eps, float weight_decay,
                                   int decoupled_wd, int64_t step, const float* sqnorm, float max_norm,
                                   float grad_scale, void* stream) {
    using namespace plnlp;
    if (n < 0 || step < 1) return PLNLP_E_SHAPE;
    if (n == 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return PLNLP_E_NULL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, decoupled_wd, (float)bc1,
                       (float)sqrt(bc2), sqnorm, max_norm, grad_scale);
    return launch_status();
}

extern "C" int plnlp_sqnorm_multi_sum_f32(const float* const* grads, const int64_t* sizes, int n_tensors,
                                          float* partial, int64_t n_partial, float* out,
                                          unsigned int* block_counter, void* stream) {
    using namespace plnlp;
    if (block_counter && !out) return PLNLP_E_NULL;
    if (n_tensors < 0 || n_tensors > PLNLP_MULTI_MAX) return PLNLP_E_SHAPE;
    if (n_tensors == 0) return 0;
    if (!grads || !sizes || !partial) return PLNLP_E_NULL;
    MultiSq a{};
    int blocks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        if (sizes[i] < 0) return PLNLP_E_SHAPE;
        if (sizes[i] > 0 && !grads[i]) return PLNLP_E_NULL;
        a.g[i] = grads[i]; a.n[i] = sizes[i]; a.first_block[i] = blocks;
        const int64_t b = plnlp_sqnorm_partials(sizes[i]);
        if (blocks + b > 0x7FFFFFF0) return PLNLP_E_SHAPE;
        blocks += (int)b;
    }
    for (int i = n_tensors; i <= PLNLP_MULTI_MAX; ++i) a.first_block[i] = blocks;
    a.count = n_tensors;
    if (n_partial < blocks) return PLNLP_E_WORKSPACE;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(sqnorm_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, partial,
                       block_counter, out);
    return launch_status();
}

extern "C" int plnlp_sqnorm_multi_f32(const float* const* grads, const int64_t* sizes, int n_tensors, float* partial,
                                      int64_t n_partial, void* stream) {
    return plnlp_sqnorm_multi_sum_f32(grads, sizes, n_tensors, partial, n_partial, nullptr, nullptr, stream);
}

extern "C" int plnlp_adam_multi_f32(const plnlp_adam_tensor* tensors, int n_tensors, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int decoupled_wd, float grad_scale,
                                    void* stream) {
    using namespace plnlp;
    if (n_tensors < 0 || n_tensors > PLNLP_MULTI_MAX) return PLNLP_E_SHAPE;
    if (n_tensors == 0) return 0;
    if (!tensors) return PLNLP_E_NULL;
    MultiAdam a{};
    int blocks = 0;
    for (int i = 0; i < n_tensors; ++i) {
        const plnlp_adam_tensor& t = tensors[i];
        if (t.n < 0 || (!t.step_scalars && t.step < 1)) return PLNLP_E_SHAPE;
        if (t.n > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq)) return PLNLP_E_NULL;
        a.p[i] = t.param; a.g[i] = t.grad; a.m[i] = t.exp_avg; a.v[i] = t.exp_avg_sq; a.n[i] = t.n;
        a.sqnorm[i] = t.sqnorm; a.max_norm[i] = t.max_norm;
        a.scal[i] = t.step_scalars;
        a.bc1[i] = a.bc2_sqrt[i] = 1.f;
        if (!t.step_scalars) adam_bias_corrections(beta1, beta2, t.step, &a.bc1[i], &a.bc2_sqrt[i]);
        a.first_block[i] = blocks;
        blocks += (int)ew_grid((t.n + 3) / 4);          // 16 bytes per lane
    }
    for (int i = n_tensors; i <= PLNLP_MULTI_MAX; ++i) a.first_block[i] = blocks;
    a.count = n_tensors;
    if (blocks == 0) return 0;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, lr, beta1,
                       beta2, eps, weight_decay, decoupled_wd, grad_scale);
    return launch_status();
}

extern "C" int plnlp_adam_step_scalars(float lr, float beta1, float beta2, int64_t step, float* out) {
    if (!out) return PLNLP_E_NULL;
    if (step < 1) return PLNLP_E_SHAPE;
    out[0] = lr;
    plnlp::adam_bias_corrections(beta1, beta2, step, &out[1], &out[2]);
    return 0;
}

extern "C" int plnlp_clip_scale_f32(float* grad, int64_t n, const float* sqnorm, float max_norm, void* stream) {
    using namespace plnlp;
    if (n < 0) return PLNLP_E_SHAPE;
    if (n == 0) return 0;
    if (!grad || !sqnorm) return PLNLP_E_NULL;
    hipLaunchKernelGGL(clip_scale_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, grad, n, sqnorm,
                       max_norm);
    return launch_status();
}

static inline int64_t colsum_blocks(int64_t n_rows) {
    return n_rows < plnlp::CS_BLOCKS ? (n_rows > 0 ? n_rows : 1) : plnlp::CS_BLOCKS;
}
extern "C" int64_t plnlp_colsum_workspace_floats(int64_t n_rows, int64_t feat) {
    return colsum_blocks(n_rows) * feat;
}

extern "C" int plnlp_colsum_f32(const float* x, int64_t ldx, int64_t n_rows, int64_t feat,
                                const float* row_weight, float scale, float* out, float* workspace,
                                int64_t workspace_floats, void* stream) {
    using namespace plnlp;
    if (!x || !out || !workspace) return PLNLP_E_NULL;
    if (n_rows <= 0 || feat <= 0 || ldx < feat || feat > (1 << 24)) return PLNLP_E_SHAPE;
    const int64_t blocks = colsum_blocks(n_rows);
    if (workspace_floats < blocks * feat) return PLNLP_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t q = feat / 4;
    const bool vec = feat % 4 == 0 && q >= 1 && q <= 256 && ldx % 4 == 0 && (uintptr_t)x % 16 == 0 &&
                     (uintptr_t)workspace % 16 == 0;
    if (feat == 1)
        hipLaunchKernelGGL(colsum_partial_col1_kernel, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, s, x, ldx, n_rows,
                           row_weight, blocks, workspace);
    else if (vec)
        hipLaunchKernelGGL(colsum_partial_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, n_rows,
                           (int)feat, row_weight, workspace);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, n_rows,
                           (int)feat, row_weight, workspace);
    if (int rc = launch_status()) return rc;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((feat + 3) / 4)), dim3(256), 0, s, workspace, blocks,
                       (int)feat, scale, out);
    return launch_status();
}

extern "C" int64_t plnlp_mlp_head_backward_workspace_floats(int64_t n_rows, int64_t feat) {
    return colsum_blocks(n_rows) * 2 * feat;
}

extern "C" int plnlp_mlp_head_backward_f32(const float* a, int64_t lda, const float* g, const float* w, float gate_scale,
                                           int64_t n_rows, int64_t feat, float* dz, int64_t lddz, float* sums,
                                           float* workspace, int64_t workspace_floats, void* stream) {
    using namespace plnlp;
    if (n_rows <= 0 || feat <= 0 || lda < feat || lddz < feat) return PLNLP_E_SHAPE;
    if (!a || !g || !w || !dz || !sums || !workspace) return PLNLP_E_NULL;
    const int64_t q = feat / 4;
    if (feat % 4 != 0 || q > 256) return PLNLP_E_UNSUPPORTED;
    if (lda % 4 != 0 || lddz % 4 != 0 || (uintptr_t)a % 16 != 0 || (uintptr_t)dz % 16 != 0 || (uintptr_t)w % 16 != 0 ||
        (uintptr_t)workspace % 16 != 0) return PLNLP_E_ALIGN;
    const int64_t blocks = colsum_blocks(n_rows);
    if (workspace_floats < blocks * 2 * feat) return PLNLP_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mlp_head_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, lda, g, w, gate_scale, n_rows,
                       (int)feat, dz, lddz, workspace);
    if (int rc = launch_status()) return rc;
    const int64_t cols = 2 * feat;              // [dw | dbp]: one reduction over the blocks for both
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((cols + 3) / 4)), dim3(256), 0, s, workspace, blocks, (int)cols,
                       1.0f, sums);
    return launch_status();
}

extern "C" int plnlp_rowdot_finish_f32(const float* partial, int64_t ld, int tiles, int64_t n_rows, const float* bias,
                                       float* out, void* stream) {
    using namespace plnlp;
    if (n_rows < 0 || tiles < 1 || ld < n_rows) return PLNLP_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!partial || !out) return PLNLP_E_NULL;
    hipLaunchKernelGGL(rowdot_finish_kernel, dim3(ew_grid(n_rows)), dim3(256), 0, (hipStream_t)stream, partial, ld, tiles,
                       n_rows, bias, out);
    return launch_status();
}

extern "C" int plnlp_matvec_f32(const float* x, int64_t ldx, int64_t n_rows, int64_t feat, const float* w,
                                const float* bias, float* out, void* stream) {
    using namespace plnlp;
    if (n_rows < 0 || feat <= 0 || ldx < feat || feat > (1 << 24)) return PLNLP_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!x || !w || !out) return PLNLP_E_NULL;
    int64_t blocks = (n_rows + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const bool vec = feat % 4 == 0 && ldx % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0;
    if (vec)
        hipLaunchKernelGGL(matvec_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx,
                           n_rows, (int)feat, w, bias, out);
    else
        hipLaunchKernelGGL(matvec_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx,
                           n_rows, (int)feat, w, bias, out);
    return launch_status();
}

extern "C" int plnlp_outer_f32(const float* g, const float* w, int64_t n_rows, int64_t feat, float* dx,
                               int64_t lddx, const plnlp_epilogue* epi, void* stream) {
    using namespace plnlp;
    if (n_rows < 0 || feat <= 0 || lddx < feat || feat > (1 << 24)) return PLNLP_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!g || !w || !dx) return PLNLP_E_NULL;
    Epi e;
    if (int rc = make_epi(epi, &e)) return rc;
    const bool vec = feat % 4 == 0 && lddx % 4 == 0 && (uintptr_t)dx % 16 == 0 && (uintptr_t)w % 16 == 0;
    if (vec) {
        int64_t blocks = (n_rows + 3) / 4;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(outer_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, w, n_rows,
                           (int)feat, dx, lddx, e);
    } else {
        hipLaunchKernelGGL(outer_kernel, dim3(ew_grid(n_rows * feat)), dim3(256), 0, (hipStream_t)stream, g, w,
                           n_rows, (int)feat, dx, lddx, e);
    }
    return launch_status();
}

extern "C" int plnlp_gate_f32(const float* g, const float* gate, float scale, float* y, int64_t n, void* stream) {
    using namespace plnlp;
    if (n < 0) return PLNLP_E_SHAPE;
    if (n == 0) return 0;
    if (!g || !gate || !y) return PLNLP_E_NULL;
    hipLaunchKernelGGL(gate_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, g, gate, scale, y, n);
    return launch_status();
}

extern "C" int plnlp_dropout_f32(const float* x, float* y, int64_t n_rows, int64_t n_cols, float p, uint64_t seed,
                                 void* stream) {
    using namespace plnlp;
    if (n_rows < 0 || n_cols < 0 || !(p >= 0.f) || p >= 1.f) return PLNLP_E_SHAPE;
    const int64_t n = n_rows * n_cols;
    if (n == 0) return 0;
    if (!x || !y) return PLNLP_E_NULL;
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n,
                       dropout_thresh(p), (uint32_t)seed, (uint32_t)(seed >> 32), 1.f / (1.f - p));
    return launch_status();
}

extern "C" int plnlp_transpose_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t n_rows,
                                   int64_t n_cols, void* stream) {
    using namespace plnlp;
    if (n_rows < 0 || n_cols < 0 || ldx < n_cols || ldy < n_rows) return PLNLP_E_SHAPE;
    if (n_rows == 0 || n_cols == 0) return 0;
    if (!x || !y) return PLNLP_E_NULL;
    dim3 grid((unsigned)((n_cols + 31) / 32), (unsigned)((n_rows + 31) / 32));
    if (grid.y > 65535) return PLNLP_E_SHAPE;
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, n_rows, n_cols);
    return launch_status();
}

extern "C" int plnlp_abi_version(void) { return PLNLP_ABI_VERSION; }

extern "C" const char* plnlp_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case PLNLP_E_NULL: return "plnlp: required pointer is NULL";
        case PLNLP_E_SHAPE: return "plnlp: invalid or inconsistent size";
        case PLNLP_E_ALIGN: return "plnlp: misaligned pointer or leading dimension";
        case PLNLP_E_UNSUPPORTED: return "plnlp: unsupported flag or size";
        case PLNLP_E_WORKSPACE: return "plnlp: workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "plnlp: unknown error";
    }
}
