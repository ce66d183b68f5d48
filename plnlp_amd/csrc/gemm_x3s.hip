// K4s -- the split-bf16 GEMM with a STATIONARY, PRE-SPLIT weight operand (gfx950).
//
//   C[M,N] = EPI( sum_s A_s[M,K_s] * B_s[K_s,N] )       A_s K-contiguous (rows optionally gathered), up to 2 K-segments
//
// Same arithmetic as the split form of gemm_f32.hip (x = hi + mid + lo in bf16, six v_mfma_f32_32x32x16_bf16 per block,
// small terms first, f32 accumulation, K walked in steps of 16 in the same order) -- bit-identical results -- on a
// different division of labour.  Every GEMM of the path whose A operand is an activation matrix with 10^5 .. 10^6 rows
// multiplies it by WEIGHTS of at most 512 x 1024 (plnlp/layer.py:83,86: F.linear inside SAGEConv / GCNConv /
// MLPPredictor, and their data gradients).  The 128 x 128 kernel re-splits the weight tile in every one of its ~10^3 row
// panels and stages both operands through LDS with one barrier per 24 MFMAs per wave.  Here:
//   * the weights are split ONCE per launch by a small kernel into the exact LDS image of the main loop
//     ([n-tile][K-step][term 3][k-half 2][column] x 16 bytes = one lane's MFMA fragment per unit), whatever their
//     layout ([N,K], [K,N], or two buffers side by side along N) -- so the main kernel has one B path, fed by
//     global_load_lds (no registers, no VALU, no ds_write), and zero-padding of ragged K / N lives in the image;
//   * a wave owns 32 rows x (32 NB) columns: its A fragment never goes through LDS -- each lane loads the 8 consecutive
//     k of its row for the next K-step straight into registers (one K-step ahead) and splits them there, ONCE per
//     n-tile instead of once per 128-column tile and per wave pair: 44 VALU per 6 NB MFMAs (NB = 8: 0.9 per MFMA
//     against 3.7 in the 128 x 128 kernel), no LDS traffic for A at all;
//   * one barrier per K-step of 6 NB MFMAs per wave; the step's loads (A registers for step + 2, the B image of
//     step + 1) are issued at its top and waited for at its bottom.
// Block = 4 waves stacked along M (128 rows); two workgroups per CU (acc 16 NB registers per lane).
#include "gemm_x3s.hip.h"
#include <utility>

namespace plnlp {
namespace x3s {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// x = hi + mid + lo for two values at once (round to nearest each; every residual exact in f32): the split of
// gemm_f32.hip::split3, same bits
__device__ __forceinline__ void split3(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = pk_bf16(x0, x1);
    float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    mid = pk_bf16(r0, r1);
    r0 -= __uint_as_float(mid << 16);
    r1 -= __uint_as_float(mid & 0xffff0000u);
    lo = pk_bf16(r0, r1);
}
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, u32x4 (&t)[3]) {
    unsigned hi[4], mid[4], lo[4];
    split3(x0.x, x0.y, hi[0], mid[0], lo[0]);
    split3(x0.z, x0.w, hi[1], mid[1], lo[1]);
    split3(x1.x, x1.y, hi[2], mid[2], lo[2]);
    split3(x1.z, x1.w, hi[3], mid[3], lo[3]);
    t[0] = u32x4{hi[0], hi[1], hi[2], hi[3]};
    t[1] = u32x4{mid[0], mid[1], mid[2], mid[3]};
    t[2] = u32x4{lo[0], lo[1], lo[2], lo[3]};
}

// the same split in three stages per element pair (5 + 5 instructions, then the four lo conversions), so that the K loop
// can place a stage behind each pair of MFMAs
struct SplitStages {
    f32x2 r[4];
    unsigned hi[4], mid[4];
    static __device__ __forceinline__ unsigned pk(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
    template <int Q>
    __device__ __forceinline__ void first(const f32x4& x0, const f32x4& x1) {
        f32x2 v;
        if constexpr (Q == 0) { v.x = x0.x; v.y = x0.y; }
        else if constexpr (Q == 1) { v.x = x0.z; v.y = x0.w; }
        else if constexpr (Q == 2) { v.x = x1.x; v.y = x1.y; }
        else { v.x = x1.z; v.y = x1.w; }
        hi[Q] = pk(v);
        r[Q].x = v.x - __uint_as_float(hi[Q] << 16);
        r[Q].y = v.y - __uint_as_float(hi[Q] & 0xffff0000u);
    }
    template <int Q>
    __device__ __forceinline__ void second() {
        mid[Q] = pk(r[Q]);
        r[Q].x -= __uint_as_float(mid[Q] << 16);
        r[Q].y -= __uint_as_float(mid[Q] & 0xffff0000u);
    }
    __device__ __forceinline__ void third(u32x4 (&t)[3]) {
        t[0] = u32x4{hi[0], hi[1], hi[2], hi[3]};
        t[1] = u32x4{mid[0], mid[1], mid[2], mid[3]};
        t[2] = u32x4{pk(r[0]), pk(r[1]), pk(r[2]), pk(r[3])};
    }
};

// ---- the weight image: unit ((nt * KS + ks) * 6 + term * 2 + h) * WN + c  holds the bf16 `term` of
//      B[k = 16 kk + 8 h + 0..7][col = nt * WN + c]   (kk = the step inside its K-segment), zeros past K_s / N
__global__ __launch_bounds__(256) void split_b_kernel(SplitArgs g) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int wn = g.wn;
    u32x4* image = reinterpret_cast<u32x4*>(g.image);
    if (id >= (int64_t)g.gn * g.ks_total * 2 * wn) return;
    const int c = (int)(id % wn);
    const int h = (int)((id / wn) & 1);
    const int64_t st = id / (2 * wn);
    const int ks = (int)(st % g.ks_total), nt = (int)(st / g.ks_total);
    const int s = (g.nseg > 1 && ks >= g.ks0) ? 1 : 0;
    const int kk = ks - (s ? g.ks0 : 0);
    const int kdim = g.k[s];
    int col = nt * wn + c;
    const bool col_ok = col < g.n;
    const float* b = g.b[s];
    int64_t ldb = g.ldb[s];
    if (s == 0 && col >= g.nb_split && g.nb_split < g.n) { b = g.b2; ldb = g.ldb2; col -= g.nb_split; }
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 16 * kk + 8 * h + e;
        const bool ok = col_ok && k < kdim;
        const int64_t off = g.b_trans ? (int64_t)col * ldb + k : (int64_t)k * ldb + col;
        x[e] = ok ? b[ok ? off : 0] : 0.f;
    }
    u32x4 t[3];
    {
        const f32x4 y0 = {x[0], x[1], x[2], x[3]}, y1 = {x[4], x[5], x[6], x[7]};
        split8(y0, y1, t);
    }
    u32x4* u = image + (st * 6 + h) * wn + c;
    u[0] = t[0];
    u[2 * wn] = t[1];
    u[4 * wn] = t[2];
}

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// one lane's 8 consecutive k of its A row for K-step `ks` (zeros past the segment's K; k % 4 == 0)
template <bool RAGGED>
__device__ __forceinline__ void load_a(const Args& g, const float* p0, const float* p1, int ks, int h, f32x4& x0, f32x4& x1) {
    const bool s1 = g.nseg > 1 && ks >= g.ks0;
    const float* p = s1 ? p1 : p0;
    const int kk = ks - (s1 ? g.ks0 : 0);
    const int k0 = 16 * kk + 8 * h;
    if constexpr (RAGGED) {
        const int kdim = s1 ? g.k[1] : g.k[0];
        const bool v0 = k0 < kdim, v1 = k0 + 4 < kdim;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        const f32x4 y0 = *reinterpret_cast<const f32x4*>(p + (v0 ? k0 : 0));
        const f32x4 y1 = *reinterpret_cast<const f32x4*>(p + (v1 ? k0 + 4 : 0));
        x0 = v0 ? y0 : zero;
        x1 = v1 ? y1 : zero;
    } else {
        x0 = *reinterpret_cast<const f32x4*>(p + k0);
        x1 = *reinterpret_cast<const f32x4*>(p + k0 + 4);
    }
}

// (A K loop with TWO steps of load look-ahead was built and measured in round 4 -- three raw A sets through untracked inline-asm
// loads, three LDS stages, one counted s_waitcnt vmcnt + a bare s_barrier per step, the B fragments read one column block
// at a time to fit the registers: 2-4 % SLOWER than this one on every shape (profiles/r04_gemm_deep_loop.jsonl), removed.)
// ROWDOT (PLNLP_EPI_ROWDOT, NB = 4 / 7 / 8): the write-back also forms, per row, the dot product of this tile's stored values with
// rowdot_w -- a 1-output linear on top of this layer (MLPPredictor's head) without a second pass over the activation.
template <int NB, bool RAGGED, bool ROWDOT = false>
__global__ __launch_bounds__(256, NB >= 7 ? 2 : (NB == 4 ? 3 : 4)) void gemm_x3s_kernel(Args g, Epi epi) {
    constexpr int WN = 32 * NB;
    constexpr int STAGE_UNITS = 6 * WN;                // 16-byte units of one K-step of the image
    constexpr int STAGE_BYTES = STAGE_UNITS * 16;
    constexpr int GLDS = STAGE_UNITS / 256;            // global_load_lds instructions per wave per stage (WN % 128 == 0 ...)
    constexpr int GLDS_REM = STAGE_UNITS % 256;        // ... or a partial last round (WN = 224: 1344 units = 5 x 256 + 64)
    constexpr int NSTAGE = 2;
    // the C tile leaves through LDS in 64-column chunks, one private region per wave
    constexpr int CS = 64 + 4;
    constexpr int C_BYTES = 4 * 32 * CS * 4;
    constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES > C_BYTES ? NSTAGE * STAGE_BYTES : C_BYTES;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    // block -> (row panel, n-tile): the n-tiles of one row panel run back to back on ONE XCD (its A rows are fetched
    // from HBM once, the second n-tile finds them in that XCD's L2).  Pure performance.
    int64_t mt; int nt;
    {
        const int64_t id = blockIdx.x, gn = g.gn, total = g.gm * gn;
        const int64_t per_xcd = (total / (8 * gn)) * gn;
        const int64_t body = per_xcd * 8;
        if (id < body) {
            const int64_t tt = (id & 7) * per_xcd + (id >> 3);
            mt = tt / gn; nt = (int)(tt % gn);
        } else {
            mt = id / gn; nt = (int)(id % gn);
        }
    }
    const int64_t row_w = g.row_lo + mt * 128 + wave * 32;      // first row of this wave
    int64_t row = row_w + l31;
    row = row < g.m ? row : g.m - 1;                            // clamped rows feed result rows the store discards
    int64_t r0 = row, r1 = row;
    if (g.a_index[0]) r0 = g.a_index[0][row];                  // rows of A gathered in place (a conv at the touched rows)
    if (g.nseg > 1 && g.a_index[1]) r1 = g.a_index[1][row];
    const float* p0 = g.a[0] + r0 * g.lda[0];
    const float* p1 = g.nseg > 1 ? g.a[1] + r1 * g.lda[1] : p0;

    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

    const int KS = g.ks_total;
    const char* img = reinterpret_cast<const char*>(g.image) + (int64_t)nt * KS * STAGE_BYTES;
    auto stage_b = [&](int ks, int buf) {                      // K-step ks of the image -> LDS buffer buf
        const char* src = img + (int64_t)ks * STAGE_BYTES + lane * 16;
        char* dst = lds + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < GLDS; ++i) {
            const int u = (i * 4 + wave) * 1024;
            __builtin_amdgcn_global_load_lds((glb_void*)(src + u), (lds_void*)(dst + u), 16, 0, 0);
        }
        if constexpr (GLDS_REM != 0) {
            const int u = (GLDS * 4 + wave) * 1024;
            if (wave * 64 < GLDS_REM)
                __builtin_amdgcn_global_load_lds((glb_void*)(src + u), (lds_void*)(dst + u), 16, 0, 0);
        }
    };
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};       // hi lo, lo hi, mid mid, hi mid, mid hi, hi hi
    u32x4 cur[3], nxt[3];

    // ---- prologue: A of step 0 split, A of step 1 in flight, B of step 0 resident
    f32x4 ra[2][2];                                            // two sets of raw A registers, used alternately
    load_a<RAGGED>(g, p0, p1, 0, h, ra[0][0], ra[0][1]);
    stage_b(0, 0);
    split8(ra[0][0], ra[0][1], cur);
    load_a<RAGGED>(g, p0, p1, KS > 1 ? 1 : 0, h, ra[1][0], ra[1][1]);
    __syncthreads();

    constexpr int G = NB / 2 + (NB & 1);                       // column-block groups of 2 (the last may hold 1)
    constexpr int SLOTS = 6 * G;                               // one slot = the MFMAs of one term product of one group
    // one K-step: D = the parity of ks (which raw set holds step ks + 1, which LDS buffer holds this step's image).
    // The instruction order is pinned slot by slot (sched_barrier): behind the MFMAs of slot (group gi, product u) ride
    // one fragment read of group gi + 1 and a stage of the split of the next step's A terms -- the LDS latency and the
    // VALU work hide behind the matrix pipe instead of in front of it.
    auto step = [&](int ks, auto dc) {
        constexpr int D = decltype(dc)::value;
        // this step's loads FIRST, so that they have the whole step to land: A of step + 2 into the raw set that was
        // split one step ago, the image of step + 1 into the other buffer (clamped at the end: never read).  Both are
        // waited for at the barrier below.
        load_a<RAGGED>(g, p0, p1, ks + 2 < KS ? ks + 2 : KS - 1, h, ra[D][0], ra[D][1]);
        stage_b(ks + 1 < KS ? ks + 1 : ks, D ^ 1);
        const u32x4* bt = reinterpret_cast<const u32x4*>(lds + D * STAGE_BYTES) + h * WN + l31;
        bf16x8 fb[2][2][3];                                    // [group parity][column block of the group][term]
#pragma unroll
        for (int jj = 0; jj < (NB > 1 ? 2 : 1); ++jj)
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) fb[0][jj][t3] = __builtin_bit_cast(bf16x8, bt[t3 * 2 * WN + jj * 32]);
        SplitStages sp;
        const f32x4 x0 = ra[D ^ 1][0], x1 = ra[D ^ 1][1];      // the raw set requested one step ago
        __builtin_amdgcn_sched_barrier(0);
        static_for<SLOTS>([&](auto sc) {
            constexpr int slot = decltype(sc)::value;
            constexpr int gi = slot / 6, u = slot % 6;
            constexpr int j0 = 2 * gi;
            constexpr int NJ = (j0 + 1 < NB) ? 2 : 1;
            constexpr int NJN = (gi + 1 < G) ? ((j0 + 3 < NB) ? 2 : 1) : 0;       // column blocks of the next group
            if constexpr (u < 3 * NJN)                           // one fragment of the next group per slot
                fb[(gi + 1) & 1][u / 3][u % 3] = __builtin_bit_cast(bf16x8, bt[(u % 3) * 2 * WN + (j0 + 2 + u / 3) * 32]);
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
                acc[j0 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[TA[u]]),
                                                                        fb[gi & 1][jj][TB[u]], acc[j0 + jj], 0, 0, 0);
            // the nine stages of the split (first x 4, second x 4, third) spread over the slots
            static_for<9>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                if constexpr (q * SLOTS / 9 == slot) {
                    if constexpr (q < 4) sp.template first<q>(x0, x1);
                    else if constexpr (q < 8) sp.template second<q - 4>();
                    else sp.third(nxt);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
    };
    int ks = 0;
    for (; ks + 2 <= KS; ks += 2) {
        step(ks, std::integral_constant<int, 0>{});
        step(ks + 1, std::integral_constant<int, 1>{});
    }
    if (ks < KS) step(ks, std::integral_constant<int, 0>{});

    // ---- write back: 64 columns at a time through this wave's private LDS region (the MFMA C/D map -- col = lane & 31,
    // row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5) -- would leave as 4-byte scattered stores), then 16-byte row stores with
    // the epilogue on float4.  (The barrier that ended the K loop freed the image buffers.)
    float* cw = reinterpret_cast<float*>(lds) + wave * 32 * CS;
    const int n0 = nt * WN;
    float rd[8];                                               // ROWDOT: this lane's share of rows i * 4 + (lane >> 4)
#pragma unroll
    for (int i = 0; i < 8; ++i) rd[i] = 0.f;
    if constexpr (NB >= 7) {
        // The wide tiles: the chunks' 4-row batches as ONE sequence, the gate rows of batch t + 1 requested BEFORE the stores of
        // batch t.  The memory counter is in order: a gate load issued after a batch's stores is back only when the memory side
        // has acknowledged them -- eight such waits per panel, and a K loop of 13 steps (h = 200) is no longer than they are
        // (citation2's gated data gradient: 2.5 ms for 1.3 ms of matrix-pipe time).  Same arithmetic, same bits.
        constexpr int RB = 4, BPC = 8 / RB, NCH = NB / 2 + (NB & 1), NBATCH = NCH * BPC;
        const bool pre = epi.flags && epi.vec4 && !(epi.flags & PLNLP_EPI_ADDEND);
        const bool gate_on = pre && (epi.flags & PLNLP_EPI_GATE);
        const int c4 = (lane & 15) * 4;
        auto gate_load = [&](auto tt, float4 (&dst)[RB]) {
            constexpr int t = decltype(tt)::value, j0 = 2 * (t / BPC), i0 = (t % BPC) * RB, NJ = (j0 + 1 < NB) ? 2 : 1;
            const int col = n0 + j0 * 32 + c4;
            if (gate_on && c4 < NJ * 32 && col < g.n) {
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    int64_t orow = row_w + (i0 + i) * 4 + (lane >> 4);
                    orow = orow < g.m ? orow : g.m - 1;
                    const int64_t gr = epi.gate_index ? (int64_t)epi.gate_index[orow] : orow;
                    dst[i] = *reinterpret_cast<const float4*>(epi.gate + gr * epi.ld_gate + col);
                }
            }
        };
        // (a chunk's bias and row-dot weights likewise: those of chunk c + 1 are requested before the first store of chunk c)
        auto chunk_load = [&](auto cc, float4& b4, float4& r4) {
            constexpr int j0 = 2 * decltype(cc)::value, NJ = (j0 + 1 < NB) ? 2 : 1;
            const int col = n0 + j0 * 32 + c4;
            const bool col_ok = c4 < NJ * 32 && col < g.n;
            b4 = r4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col_ok && pre && (epi.flags & PLNLP_EPI_BIAS)) b4 = *reinterpret_cast<const float4*>(epi.bias + col);
            if constexpr (ROWDOT) { if (col_ok) r4 = *reinterpret_cast<const float4*>(epi.rowdot_w + col); }
        };
        float4 gq[2][RB], bq[2], rq[2];
        gate_load(std::integral_constant<int, 0>{}, gq[0]);
        chunk_load(std::integral_constant<int, 0>{}, bq[0], rq[0]);
        static_for<NBATCH>([&](auto tt) {
            constexpr int t = decltype(tt)::value, jc = t / BPC, j0 = 2 * jc, i0 = (t % BPC) * RB, NJ = (j0 + 1 < NB) ? 2 : 1;
            if constexpr (i0 == 0) {
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        cw[((q & 3) + 8 * (q >> 2) + 4 * h) * CS + jj * 32 + l31] = acc[j0 + jj][q];
                if constexpr (jc + 1 < NCH) chunk_load(std::integral_constant<int, jc + 1>{}, bq[(jc + 1) & 1], rq[(jc + 1) & 1]);
            }
            const int col = n0 + j0 * 32 + c4;
            const bool col_ok = c4 < NJ * 32 && col < g.n;
            const bool second = col >= g.n_split;
            const float4 bias4 = bq[jc & 1], rw4 = rq[jc & 1];
            if constexpr (t + 1 < NBATCH) gate_load(std::integral_constant<int, t + 1>{}, gq[(t + 1) & 1]);
            float4 p4[RB], v4[RB];
            if (col_ok && pre && (epi.flags & PLNLP_EPI_ACCUM)) {
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    int64_t orow = row_w + (i0 + i) * 4 + (lane >> 4);
                    orow = orow < g.m ? orow : g.m - 1;
                    const float* pp = second ? g.c2 + orow * g.ldc2 - g.n_split : g.c + orow * g.ldc;
                    p4[i] = *reinterpret_cast<const float4*>(pp + col);
                }
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) v4[i] = *reinterpret_cast<const float4*>(cw + ((i0 + i) * 4 + (lane >> 4)) * CS + c4);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int64_t orow = row_w + (i0 + i) * 4 + (lane >> 4);
                if (col_ok && orow < g.m) {
                    float* op = second ? g.c2 + orow * g.ldc2 - g.n_split : g.c + orow * g.ldc;
                    float4 y = v4[i];
                    if (pre) y = epi_apply4_pre(epi, y, orow, col, g.n, bias4, gq[t & 1][i], p4[i]);
                    else y = epi_apply4(epi, y, orow, col, g.n, op);
                    *reinterpret_cast<float4*>(op + col) = y;
                    if constexpr (ROWDOT) rd[i0 + i] = rowdot_acc(rd[i0 + i], y, rw4);
                }
            }
        });
    } else {
        static_for<NB / 2 + (NB & 1)>([&](auto jc) {
            constexpr int j0 = 2 * decltype(jc)::value;
            constexpr int NJ = (j0 + 1 < NB) ? 2 : 1;
    #pragma unroll
            for (int jj = 0; jj < NJ; ++jj)
    #pragma unroll
                for (int q = 0; q < 16; ++q)
                    cw[((q & 3) + 8 * (q >> 2) + 4 * h) * CS + jj * 32 + l31] = acc[j0 + jj][q];
            // rows i * 4 + (lane >> 4), 16-byte column group lane & 15.  The epilogue's operands (bias once per chunk; gate and
            // accumulate rows) are requested for all 8 rows BEFORE the first store: a load issued between stores waits for
            // them (the compiler must assume the result aliases the operand), 8 dependent round trips per chunk.
            const int c4 = (lane & 15) * 4;
            const int col = n0 + j0 * 32 + c4;
            const bool col_ok = c4 < NJ * 32 && col < g.n;
            const bool second = col >= g.n_split;
            const bool pre = epi.flags && epi.vec4 && !(epi.flags & PLNLP_EPI_ADDEND);
            float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col_ok && pre && (epi.flags & PLNLP_EPI_BIAS)) bias4 = *reinterpret_cast<const float4*>(epi.bias + col);
            float4 rw4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (ROWDOT) { if (col_ok) rw4 = *reinterpret_cast<const float4*>(epi.rowdot_w + col); }
            constexpr int RB = NB >= 7 ? 4 : 8;                    // rows per batch (the wide tiles have few registers to spare)
    #pragma unroll
            for (int i0 = 0; i0 < 8; i0 += RB) {
                float4 g4[RB], p4[RB], v4[RB];
                if (col_ok && pre && (epi.flags & (PLNLP_EPI_GATE | PLNLP_EPI_ACCUM))) {
    #pragma unroll
                    for (int i = 0; i < RB; ++i) {
                        int64_t orow = row_w + (i0 + i) * 4 + (lane >> 4);
                        orow = orow < g.m ? orow : g.m - 1;
                        if (epi.flags & PLNLP_EPI_GATE) {
                            const int64_t gr = epi.gate_index ? (int64_t)epi.gate_index[orow] : orow;
                            g4[i] = *reinterpret_cast<const float4*>(epi.gate + gr * epi.ld_gate + col);
                        }
                        if (epi.flags & PLNLP_EPI_ACCUM) {
                            const float* pp = second ? g.c2 + orow * g.ldc2 - g.n_split : g.c + orow * g.ldc;
                            p4[i] = *reinterpret_cast<const float4*>(pp + col);
                        }
                    }
                }
    #pragma unroll
                for (int i = 0; i < RB; ++i) v4[i] = *reinterpret_cast<const float4*>(cw + ((i0 + i) * 4 + (lane >> 4)) * CS + c4);
    #pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int64_t orow = row_w + (i0 + i) * 4 + (lane >> 4);
                    if (col_ok && orow < g.m) {
                        float* op = second ? g.c2 + orow * g.ldc2 - g.n_split : g.c + orow * g.ldc;
                        float4 y = v4[i];
                        if (pre) y = epi_apply4_pre(epi, y, orow, col, g.n, bias4, g4[i], p4[i]);
                        else y = epi_apply4(epi, y, orow, col, g.n, op);
                        *reinterpret_cast<float4*>(op + col) = y;
                        if constexpr (ROWDOT) rd[i0 + i] = rowdot_acc(rd[i0 + i], y, rw4);
                    }
                }
            }
        });
    }
    if constexpr (ROWDOT) {
        // the 16 lanes that share a row (lane & 15 = their 16-byte column group) fold their shares in a fixed tree; the
        // tile's partial goes to row nt of rowdot_out (plnlp_rowdot_finish_f32 adds the tiles in order)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = rd[i];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            const int64_t orow = row_w + i * 4 + (lane >> 4);
            if ((lane & 15) == 0 && orow < g.m) epi.rowdot_out[(int64_t)nt * epi.rowdot_ld + orow] = v;
        }
    }
}

template <int NB>
static int launch_nb(const Args& a, const Epi& e, bool ragged, hipStream_t s) {
    dim3 grid((unsigned)(a.gm * a.gn));
    if (e.flags & PLNLP_EPI_ROWDOT) {
        if constexpr (NB == 4 || NB == 7 || NB == 8) {
            if (e.rowdot_ld < a.m) return PLNLP_E_SHAPE;
            if (ragged) hipLaunchKernelGGL((gemm_x3s_kernel<NB, true, true>), grid, dim3(256), 0, s, a, e);
            else        hipLaunchKernelGGL((gemm_x3s_kernel<NB, false, true>), grid, dim3(256), 0, s, a, e);
            return launch_status();
        } else {
            return PLNLP_E_UNSUPPORTED;
        }
    }
    if (ragged) hipLaunchKernelGGL((gemm_x3s_kernel<NB, true>), grid, dim3(256), 0, s, a, e);
    else        hipLaunchKernelGGL((gemm_x3s_kernel<NB, false>), grid, dim3(256), 0, s, a, e);
    return launch_status();
}

// workgroups the chip holds at once for a tile width (registers: 2 / 3 / 4 waves per SIMD)
static int slots_of(int nb) { return 256 * (nb >= 7 ? 2 : nb == 4 ? 3 : 4); }

// measurement knob (plnlp_gemm_stationary_tuning): process-global, for A/B runs only
static int g_force_nb = 0;
static int g_min_rows = 16384;      // rows of A from which the form applies (gemm_impl); the knob's second argument, > 0, changes it
void set_tuning(int nb, int min_rows) { g_force_nb = nb; if (min_rows > 0) g_min_rows = min_rows; }
int min_rows() { return g_min_rows; }

// n-tile width (in 32-column blocks) of a launch.  A launch is rounds of slots_of(nb) workgroups; measured on MI355X
// (profiles/r04_gemm_tile_width.jsonl) a round of 256-column tiles takes 1.2 .. 1.3 x a round of 128-column tiles and
// does twice the work, so wide tiles win -- unless the row panels leave their last round nearly empty (the step's forward
// GEMM: 1 033 panels on 512 slots = 2.02 rounds -> three; on 768 slots of narrow tiles 2.69 -> three SHORT ones).
int pick_nb(int64_t m, int64_t n) {
    if (g_force_nb == 1 || g_force_nb == 2 || g_force_nb == 4 || g_force_nb == 7 || g_force_nb == 8) return g_force_nb;
    if (n <= 32) return 1;
    if (n <= 64) return 2;
    if (n <= 128) return 4;
    if (n > 192 && n <= 224) return 7;        // h = 200 in one tile (without the whole-block kernel gemm_impl sends only ragged K here)
    // whole 256-column tiles and enough rows: the whole-block kernel (gemm_x3b.hip), whose last round goes out in half blocks
    if (n % 256 == 0 && x3b::applies(m, 8)) return 8;
    const int64_t panels = (m + 127) / 128;
    auto rounds = [&](int nb) {
        const int64_t blocks = panels * ((n + 32 * nb - 1) / (32 * nb)), slots = slots_of(nb);
        return (double)((blocks + slots - 1) / slots);
    };
    return 1.2 * rounds(8) <= rounds(4) ? 8 : 4;
}

int64_t image_bytes(int64_t n, const int64_t* k, int nseg, int nb) {
    int64_t ks = 0;
    for (int s = 0; s < nseg; ++s) ks += (k[s] + 15) / 16;
    const int64_t wn = 32 * nb;
    return (n + wn - 1) / wn * wn * ks * 6 * 16;      // columns padded to whole tiles
}

static int launch_kernel(const Args& a, int nb, bool ragged, const Epi& e, hipStream_t s) {
    switch (nb) {
        case 8: return launch_nb<8>(a, e, ragged, s);
        case 7: return launch_nb<7>(a, e, ragged, s);
        case 4: return launch_nb<4>(a, e, ragged, s);
        case 2: return launch_nb<2>(a, e, ragged, s);
        case 1: return launch_nb<1>(a, e, ragged, s);
    }
    return PLNLP_E_UNSUPPORTED;
}

int launch(const SplitArgs& sp_in, const Args& a_in, int nb, const Epi& e, hipStream_t s) {
    SplitArgs sp = sp_in;
    Args a = a_in;
    const int wn = 32 * nb;
    sp.wn = wn;
    sp.gn = (int)((sp.n + wn - 1) / wn);
    const int64_t total = (int64_t)sp.gn * sp.ks_total * 2 * wn;
    hipLaunchKernelGGL(split_b_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, sp);
    if (int rc = launch_status()) return rc;
    bool ragged = false;
    for (int q = 0; q < a.nseg; ++q) ragged |= (a.k[q] % 16) != 0;
    a.gn = sp.gn;
    a.gm = (a.m + 127) / 128;
    a.row_lo = 0;
    a.image = sp.image;
    if (x3b::takes(a.m, a.n, nb, ragged, a.ks_total, e)) return x3b::launch(a, nb, ragged, e, s);
    return launch_kernel(a, nb, ragged, e, s);
}

}  // namespace x3s
}  // namespace plnlp
