// K4 -- fp32 GEMM on the f32-input MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   C[M,N] = EPI( sum_s op(A_s) * op(B_s) )          up to 2 K-segments (concat-K)
//
// Exact f32: the MFMA is bit-for-bit an fmaf chain (no TF32/xf32 on gfx950), so
// results stay inside fp32 round-off of the reference's sgemm.  Peak for this
// instruction is the fp32 vector peak, 157.3 TFLOP/s.
//
// Block = 256 threads = 4 waves (2x2), block tile 128x128, K-tile 32, wave tile
// 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator registers).  Global -> register
// -> LDS staging with the next K-tile's global loads in flight during the MFMAs,
// two LDS buffers, one barrier per K-tile.
//
// Operand layouts are kept as stored -- no transposes are materialised:
//   "K-contiguous" operand (reduction index fastest: A of x@W^T, W itself):
//       LDS tile [128][32+4]; a lane fetches 4 consecutive k with one
//       ds_read_b128 and feeds them to 4 successive MFMAs.  The MFMA's k slot
//       (lane>>5) is re-labelled -- half h owns k = 8q+4h+{0..3} -- which is
//       legal because A and B use the same labelling and the sum over k is
//       order-free across the two halves.  Row stride 36 floats makes the 16
//       rows of a ds_read_b128 lane group start on 16 distinct bank quads.
//   "row-contiguous" operand (reduction index slowest: dY and X in wgrad, W in
//       dgrad): LDS tile [32][128]; lanes read consecutive floats (ds_read_b32),
//       conflict-free.
// Split-K (wgrad reduces over hundreds of thousands of rows into a 256x256
// result): grid.z slices write raw partial tiles to a workspace and a second
// kernel adds them in slice order -- deterministic, no atomics.
#include "common.hip.h"
#include "gemm_x3s.hip.h"     // the stationary-weights form (its own translation unit, gemm_x3s.hip)
#include "gemm_wgw.hip.h"     // the weight gradients with whole 224- / 256-wide blocks of the result per workgroup (gemm_wgw.hip)
#include <utility>

namespace plnlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// staging registers use the native vector type: HIP's float4 struct is copied with memcpy, which
// kept the staged tile in scratch memory instead of VGPRs
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// This file is compiled TWICE (plnlp_amd/build.py): with K-tile depth 32 -- the translation unit that also
// holds the host entry points -- and with depth 16 (-DPLNLP_GEMM_BK=16: kernels + their launcher only, in
// their own namespace).  Depth 16 halves the LDS and staging-register footprint, so THREE workgroups share a
// CU instead of two: measured (profiles/r02_gemm_bk16_ab.txt) +2..5 % on the forward / data-gradient shapes,
// +15 % on the short reductions of citation2 (K = 180 / 200), -3 % on the weight gradients, which keep 32.
#ifndef PLNLP_GEMM_BK
#define PLNLP_GEMM_BK 32
#endif
// ... and a THIRD time with -DPLNLP_GEMM_BK=16 -DPLNLP_GEMM_X3=1: the same tiling on the bf16 MFMA
// (v_mfma_f32_32x32x16_bf16, 16x the f32 MFMA rate) with every f32 operand element split, on its way into
// LDS, into three bf16 terms  x = hi + mid + lo  (round-to-nearest each, the residuals exact in f32, so the
// three carry 24+ significant bits) and the product formed from the six terms that matter:
//     a b ~= hi hi + (hi mid + mid hi) + (mid mid + hi lo + lo hi),     dropped: mid lo, lo mid, lo lo
// bf16 rounds to 8 significant bits (unit round-off 2^-8): |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|, and what the three
// terms leave of x is <= 2^-24 |x|.  WORST case per product: the two dropped cross terms 2 x 2^-24, the two operand
// residuals 2 x 2^-24, lo lo 2^-32 -- about 2^-22 |a b|, four times the one rounding (2^-24) the f32 MFMA applies to
// the product; for operands that do not sit at the rounding boundaries the terms are a factor 4-16 smaller and of
// random sign, and the error MEASURED against fp64 is at or below the f32 MFMA's on every shape of the path
// (profiles/r02_gemm_bf16x3_error.jsonl; the test bound is 2^-22 and 1.5 x the f32 MFMA's own error).
// The sum is accumulated in f32 like before.  Six bf16 MFMAs of K = 16 replace eight f32
// MFMAs of K = 2 per 32x32x16 block: 6 x 32 = 192 cycles instead of 512.  Which form a launch uses is the
// caller's choice (plnlp_gemm_operand.math); tests/test_hip_round2.py measures both against fp64.
#ifndef PLNLP_GEMM_X3
#define PLNLP_GEMM_X3 0
#endif
#if PLNLP_GEMM_X3
#if PLNLP_GEMM_BK != 16
#error "the split-bf16 form is built at K-tile depth 16"
#endif
#define GEMM_NS x16
#elif PLNLP_GEMM_BK == 16
#define GEMM_NS g16
#else
#define GEMM_NS g32
#endif
constexpr int BM = 128, BN = 128;
struct Seg {
    const float* a; int64_t lda;
    const float* b; int64_t ldb;
    int k;
    int a_vec, b_vec;  // 16-byte path usable
    const int32_t* b_index;   // B's row for reduction index j (row-contiguous B only), nullable
    const int32_t* a_index;   // A's row for result row i (K-contiguous A only), nullable
    const int32_t* a_index2;  // with a_index: the row is the elementwise product of rows a_index[i], a_index2[i]
    const int32_t* b_index2;  // with b_index: likewise for B's row of reduction index j
};

struct GemmArgs {
    Seg seg[2];
    int nseg;
    int tiles0;       // K-tiles in segment 0
    int tiles_total;
    float* c; int64_t ldc;
    int64_t m; int n;
    int split_k;
    int64_t ws_stride;  // floats per split slice (m*n)
    int64_t gm; int gn; // tile grid of this launch (a rectangular region of the full tile grid)
    int64_t mt0; int nt0; // first row / column tile of the region
    int z0;               // first split-K slice this launch writes
    float* c2; int64_t ldc2; int n_split;   // columns >= n_split go to c2[:, col - n_split] (n_split = n: unused)
    const float* b2; int64_t ldb2; int nb_split;   // B columns >= nb_split come from b2 (segment 0; nb_split = n: unused)
    int bidx_mask;        // b_index applies to: bit 0 = the first B buffer, bit 1 = b2
    int vec_store;        // output rows are 16-byte storable (n, ldc, n_split, pointers all 4-float aligned)
};

namespace GEMM_NS {
constexpr int BK = PLNLP_GEMM_BK;     // 32 or 16
static_assert(BK == 32 || BK == 16, "K-tile depth");
constexpr int NP = BK / 8;                 // 16-byte loads per thread per operand per K-tile
constexpr int KQ = BK / 4;                 // 16-byte groups per K-contiguous tile row
constexpr int KQ_SHIFT = BK == 32 ? 3 : 2;
constexpr int KC_ROWS = 256 / KQ;          // K-contiguous tile rows covered by one pass of the 256 threads
constexpr int LDK = BK + 4;   // K-contiguous tile row stride (floats)
constexpr int LDR = 128;      // row-contiguous tile row stride (floats)
constexpr bool X3 = PLNLP_GEMM_X3 != 0;
// split-bf16 LDS image of one operand tile (128 x 16), in 16-byte units:
//   K-contiguous operand: [term 3][k-group 2][row 128] x (8 bf16 = k 8g .. 8g+7 of that row) -- exactly one lane's
//       MFMA fragment per unit (ds_read_b128, lanes 0-31 / 32-63 each on 512 contiguous bytes); the k-group
//       stride is 132 units = 2112 bytes = 64 (mod 128): LDS WRITES are banked modulo 32 dwords, and the
//       even / odd lanes of a ds_write_b128 (k-group 0 / 1 of the same rows) must fall on different halves
//       of that 128-byte row (at 136 units, = 0 mod 128, every write was a 2-way conflict:
//       SQ_LDS_BANK_CONFLICT = a third of the LDS-busy cycles, profiles/r02_gemm_x3_pmc.json)
//   row-contiguous operand: [term 3][k-pair 8][128 rows] x (one dword = bf16 k 2p, bf16 k 2p+1 of that row):
//       a lane's fragment is 4 ds_read_b32 (consecutive rows on consecutive banks), a staging thread's
//       4 rows x 2 k are one ds_write_b128
constexpr int X3_KG = 132, X3_TERM = 2 * X3_KG;
constexpr int TILE_FLOATS = X3 ? 3 * X3_TERM * 4 : 128 * LDK;  // floats per operand tile buffer

// staging thread t -> tile element of its p-th 16-byte load.  f32 form: K-contiguous rows (t >> KQ_SHIFT) + KC_ROWS p
// at k-quad (t & (KQ-1)); row-contiguous k rows (t >> 5) + 8 p.  Split form: a thread owns 8 consecutive k of ONE
// row (one LDS unit) resp. two consecutive k rows (one bf16 pair per column).
__device__ __forceinline__ int kc_row(int t, int p) { return X3 ? (t >> 1) : (t >> KQ_SHIFT) + KC_ROWS * p; }
__device__ __forceinline__ int kc_k(int t, int p) { return X3 ? (t & 1) * 8 + 4 * p : (t & (KQ - 1)) * 4; }
__device__ __forceinline__ int rc_k(int t, int p) { return X3 ? 2 * (t >> 5) + p : (t >> 5) + 8 * p; }


// ---- global -> registers ------------------------------------------------------
// K-contiguous operand: rows [row0, row0+128) x k [k0, k0+32); thread t loads
// float4 at (row0 + (t>>3) + 32p, k0 + 4*(t&7)), p = 0..3.
// FAST: 16-byte loads, no branches.  RAGGED (the segment's last K-tile may be partial; needs kdim % 4 == 0):
// a float4 whose k lies past kdim is loaded from the last valid group instead and zeroed by a select.
// rsel (nullable): the operand's rows for this thread's 4 tile rows, already looked up (rows gathered in
// the loader: the ids do not depend on the K-tile, so the kernel fetches them once)
struct RowSel { int r[NP]; };      // a thread's gathered rows, by VALUE (a pointer to a private array pinned it in scratch)
template <bool FAST, bool RAGGED = false, bool SEL = false>
__device__ __forceinline__ void load_kc(f32x4 (&r)[NP], const float* __restrict__ base, int64_t ld,
                                        int64_t row0, int64_t nrows, int k0, int kdim, int vec, int t,
                                        RowSel rsel = RowSel{}) {
    if constexpr (FAST) {
        // 4 independent 16-byte loads, no guards, nothing the compiler must wait on between them.
        // Rows past the matrix edge are CLAMPED to the last row: they read valid memory and only
        // feed output rows that the (guarded) store discards.
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int kq = kc_k(t, p) + k0;
            const bool kin = !RAGGED || kq < kdim;
            const int kc = (!RAGGED || kq < kdim) ? kq : kdim - 4;
            int64_t row = row0 + kc_row(t, p);
            row = row < nrows ? row : nrows - 1;
            if constexpr (SEL) row = rsel.r[p];
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + row * ld + kc);
            r[p] = kin ? v : zero;
        }
        return;
    }
    // guarded form, branch-free: every address is clamped into the matrix (always loadable) and
    // out-of-range elements are zeroed by selects, so the 16 loads of a thread are all in flight
    // together instead of waiting on each other across divergent branches
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int kq = kc_k(t, p) + k0;
        const int64_t row = row0 + kc_row(t, p);
        const bool rok = row < nrows;
        int64_t rr = rok ? row : nrows - 1;
        if constexpr (SEL) rr = rsel.r[p];
        const float* q = base + rr * ld;
        f32x4 v;
        if (vec && kq + 3 < kdim) {            // whole 16-byte group inside: one wide load
            v = *reinterpret_cast<const f32x4*>(q + kq);
        } else {
            const int kl = kdim - 1;
            v.x = q[kq + 0 < kdim ? kq + 0 : kl]; v.y = q[kq + 1 < kdim ? kq + 1 : kl];
            v.z = q[kq + 2 < kdim ? kq + 2 : kl]; v.w = q[kq + 3 < kdim ? kq + 3 : kl];
            v.x = kq + 0 < kdim ? v.x : 0.f; v.y = kq + 1 < kdim ? v.y : 0.f;
            v.z = kq + 2 < kdim ? v.z : 0.f; v.w = kq + 3 < kdim ? v.w : 0.f;
        }
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        r[p] = rok ? v : zero;
    }
}
// row-contiguous operand stored [kdim][nrows]: k [k0,k0+32) x rows [row0,row0+128);
// thread t loads float4 at (k0 + (t>>5) + 8p, row0 + 4*(t&31)).
// INDEXED: the operand's row for reduction index k is kidx[k] (rows gathered in place).
template <bool FAST, bool INDEXED = false, bool RAGGED = false>
__device__ __forceinline__ void load_rc(f32x4 (&r)[NP], const float* __restrict__ base, int64_t ld,
                                        int64_t row0, int64_t nrows, int k0, int kdim, int vec, int t,
                                        const int32_t* __restrict__ kidx = nullptr) {
    const int64_t rq = row0 + (t & 31) * 4;
    if constexpr (FAST) {   // needs nrows % 4 == 0 (checked on the host): clamp whole float4 groups
        const int64_t rc = rq < nrows ? rq : nrows - 4;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        int ks[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int k = k0 + rc_k(t, p);
            ks[p] = (!RAGGED || k < kdim) ? k : kdim - 1;            // a k row past the end: reload the last one
        }
        if constexpr (INDEXED) {
            if (kidx) {                      // block-uniform: the index may apply to one of two B buffers only
#pragma unroll
                for (int p = 0; p < NP; ++p) ks[p] = kidx[ks[p]];
            }
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (int64_t)ks[p] * ld + rc);
            r[p] = (!RAGGED || k0 + rc_k(t, p) < kdim) ? v : zero;
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int k = k0 + rc_k(t, p);
        const bool kok = k < kdim;
        const int kc = kok ? k : kdim - 1;
        const float* q = base + (int64_t)((INDEXED && kidx) ? kidx[kc] : kc) * ld;
        f32x4 v;
        if (vec && rq + 3 < nrows) {
            v = *reinterpret_cast<const f32x4*>(q + rq);
        } else {
            const int64_t rl = nrows - 1;
            v.x = q[rq + 0 < nrows ? rq + 0 : rl]; v.y = q[rq + 1 < nrows ? rq + 1 : rl];
            v.z = q[rq + 2 < nrows ? rq + 2 : rl]; v.w = q[rq + 3 < nrows ? rq + 3 : rl];
            v.x = rq + 0 < nrows ? v.x : 0.f; v.y = rq + 1 < nrows ? v.y : 0.f;
            v.z = rq + 2 < nrows ? v.z : 0.f; v.w = rq + 3 < nrows ? v.w : 0.f;
        }
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        r[p] = kok ? v : zero;
    }
}
__device__ __forceinline__ void store_kc(float* __restrict__ tile, const f32x4 (&r)[NP], int t) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
        *reinterpret_cast<f32x4*>(tile + kc_row(t, p) * LDK + kc_k(t, p)) = r[p];
}
__device__ __forceinline__ void store_rc(float* __restrict__ tile, const f32x4 (&r)[NP], int t) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
        *reinterpret_cast<f32x4*>(tile + rc_k(t, p) * LDR + (t & 31) * 4) = r[p];
}

// ---- the split-bf16 form: registers -> three bf16 terms -> LDS -----------------------------------------
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {      // (bf16(lo), bf16(hi)), round to nearest even
    typedef float f32x2 __attribute__((ext_vector_type(2)));      // selects v_cvt_pk_bf16_f32 (no inline asm: the
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));    // scheduler must see a VALU instruction)
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// elementwise product that stays a rounded f32 product (never contracted into a consumer's subtraction)
__device__ __forceinline__ f32x4 mul_rounded(f32x4 a, f32x4 b) {
    f32x4 r = a * b;
    asm("" : "+v"(r));          // opaque to the optimiser: what leaves here is the rounded value
    return r;
}
// x = hi + mid + lo for two values at once; each residual is exact in f32 (the rounded term shares its leading bits)
__device__ __forceinline__ void split3(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = cvt_pk_bf16(x0, x1);
    float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    mid = cvt_pk_bf16(r0, r1);
    r0 -= __uint_as_float(mid << 16);
    r1 -= __uint_as_float(mid & 0xffff0000u);
    lo = cvt_pk_bf16(r0, r1);
}
// K-contiguous: r[0], r[1] = k 8g .. 8g+7 of row t >> 1 (g = t & 1)
__device__ __forceinline__ void store_kc_x3(float* __restrict__ tile, const f32x4 (&r)[NP], int t) {
    static_assert(!X3 || NP == 2, "one LDS unit per thread");
    unsigned h[4], m[4], l[4];
    split3(r[0].x, r[0].y, h[0], m[0], l[0]);
    split3(r[0].z, r[0].w, h[1], m[1], l[1]);
    split3(r[1 % NP].x, r[1 % NP].y, h[2], m[2], l[2]);
    split3(r[1 % NP].z, r[1 % NP].w, h[3], m[3], l[3]);
    const u32x4 hi = {h[0], h[1], h[2], h[3]}, mid = {m[0], m[1], m[2], m[3]}, lo = {l[0], l[1], l[2], l[3]};
    u32x4* u = reinterpret_cast<u32x4*>(tile) + (t & 1) * X3_KG + (t >> 1);
    u[0] = hi; u[X3_TERM] = mid; u[2 * X3_TERM] = lo;
}
// row-contiguous: r[0] = k row 2P, r[1] = k row 2P+1 (P = t >> 5), rows 4 (t & 31) .. +3
__device__ __forceinline__ void store_rc_x3(float* __restrict__ tile, const f32x4 (&r)[NP], int t) {
    unsigned h[4], m[4], l[4];
    split3(r[0].x, r[1 % NP].x, h[0], m[0], l[0]);
    split3(r[0].y, r[1 % NP].y, h[1], m[1], l[1]);
    split3(r[0].z, r[1 % NP].z, h[2], m[2], l[2]);
    split3(r[0].w, r[1 % NP].w, h[3], m[3], l[3]);
    const u32x4 hi = {h[0], h[1], h[2], h[3]}, mid = {m[0], m[1], m[2], m[3]}, lo = {l[0], l[1], l[2], l[3]};
    u32x4* u = reinterpret_cast<u32x4*>(tile) + (t >> 5) * 32 + (t & 31);
    u[0] = hi; u[X3_TERM] = mid; u[2 * X3_TERM] = lo;
}
// the same split in three stages per element pair (5 + 5 + 1 instructions), so that the K loop can place one
// stage behind each MFMA.  ROWC: the operand is row-contiguous (pairs run down two k rows) or K-contiguous
struct X3Split {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    f32x2 r[4];                      // the pair's running residual (kept as the 2-vector the conversion consumes)
    unsigned hi[4], mid[4], lo[4];
    static __device__ __forceinline__ unsigned pk(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
    template <bool ROWC, int Q, bool PAIR = false>
    __device__ __forceinline__ void first(const f32x4 (&x)[NP], const f32x4 (*y)[NP] = nullptr) {
        // no contraction: with PAIR, x * y - hi as one fma would split the EXACT product instead of the f32 one
        // the materialised Hadamard holds (a different, if no worse, rounding than the unfused path)
#pragma clang fp contract(off)
        f32x2 v;
        if constexpr (ROWC) { v.x = x[0][Q]; v.y = x[1 % NP][Q]; }
        else { v.x = x[(Q >> 1) % NP][2 * (Q & 1)]; v.y = x[(Q >> 1) % NP][2 * (Q & 1) + 1]; }
        if constexpr (PAIR) {          // the operand element is the product of the two gathered rows' elements
            if constexpr (ROWC) { v.x *= (*y)[0][Q]; v.y *= (*y)[1 % NP][Q]; }
            else { v.x *= (*y)[(Q >> 1) % NP][2 * (Q & 1)]; v.y *= (*y)[(Q >> 1) % NP][2 * (Q & 1) + 1]; }
        }
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
    __device__ __forceinline__ void third() {
        lo[0] = pk(r[0]); lo[1] = pk(r[1]); lo[2] = pk(r[2]); lo[3] = pk(r[3]);
    }
    template <bool ROWC>
    __device__ __forceinline__ void store(float* __restrict__ tile, int t) const {
        u32x4* u = reinterpret_cast<u32x4*>(tile) + (ROWC ? (t >> 5) * 32 + (t & 31) : (t & 1) * X3_KG + (t >> 1));
        const u32x4 h4 = {hi[0], hi[1], hi[2], hi[3]}, m4 = {mid[0], mid[1], mid[2], mid[3]},
                    l4 = {lo[0], lo[1], lo[2], lo[3]};
        u[0] = h4; u[X3_TERM] = m4; u[2 * X3_TERM] = l4;
    }
};
// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// one lane's fragment (8 bf16: k 8h .. 8h+7 of `row`) of term s
template <bool ROWC>
__device__ __forceinline__ bf16x8 frag_x3(const float* __restrict__ tile, int s, int row, int h) {
    u32x4 v;
    if constexpr (ROWC) {
        const unsigned* d = reinterpret_cast<const unsigned*>(tile) + s * (X3_TERM * 4) + (4 * h) * 128 + row;
        v.x = d[0]; v.y = d[128]; v.z = d[256]; v.w = d[384];
    } else {
        v = reinterpret_cast<const u32x4*>(tile)[s * X3_TERM + h * X3_KG + row];
    }
    return __builtin_bit_cast(bf16x8, v);
}
// one K-tile (16) of the split product: 6 bf16 MFMAs per 32x32 block, small terms first
template <bool A_T, bool B_T>
__device__ __forceinline__ void mma_tile_x3(f32x16 (&acc)[2][2], const float* __restrict__ at,
                                            const float* __restrict__ bt, int wm, int wn, int l31, int h) {
    bf16x8 a[2][3], b[2][3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            a[i][s] = frag_x3<A_T>(at, s, wm * 64 + i * 32 + l31, h);
            b[i][s] = frag_x3<!B_T>(bt, s, wn * 64 + i * 32 + l31, h);
        }
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][TA[u]], b[j][TB[u]], acc[i][j], 0, 0, 0);
}

// one K-tile of both operands, global -> registers.  The segment is picked with selects (no
// runtime-indexed struct access: that sent the staging registers to scratch).
// MODE 0: guarded loads everywhere (unaligned operands); 1: fast loads, every K-tile full;
// 2: fast loads, a segment's last K-tile may be partial (k % 4 == 0): out-of-range groups zeroed by selects
// arow0 / arow1 (AIDX): this thread's gathered A rows for segment 0 / 1, looked up once per block
// (entries are the plain row ids where a segment has no index)
// PAIR: the indexed operand's row is the elementwise product of TWO gathered rows; the second factor goes to r2
// (A: its row ids arrive in arow1 -- one segment only, so the slot is free; B: seg[0].b_index2)
template <bool A_T, bool B_T, int MODE, bool BIDX, bool AIDX = false, bool PAIR = false>
__device__ __forceinline__ void load_tile(const GemmArgs& g, int tile, f32x4 (&ra)[NP], f32x4 (&rb)[NP],
                                          int64_t m0, int n0, int t, const int (&arow0)[NP],
                                          const int (&arow1)[NP], f32x4 (*r2)[NP] = nullptr) {
    static_assert(!PAIR || (AIDX != BIDX), "the pair form belongs to exactly one gathered operand");
    static_assert(!AIDX || !A_T, "gathered A rows exist for the K-contiguous layout only");
    static_assert(!BIDX || !B_T, "gathered B rows exist for the row-contiguous layout only");
    const bool s1 = (g.nseg > 1) && (tile >= g.tiles0);
    const float* a = s1 ? g.seg[1].a : g.seg[0].a;
    const float* b = s1 ? g.seg[1].b : g.seg[0].b;   // may be redirected to b2 below
    const int64_t lda = s1 ? g.seg[1].lda : g.seg[0].lda;
    int64_t ldb = s1 ? g.seg[1].ldb : g.seg[0].ldb;
    const int kdim = s1 ? g.seg[1].k : g.seg[0].k;
    const int avec = s1 ? g.seg[1].a_vec : g.seg[0].a_vec;
    const int bvec = s1 ? g.seg[1].b_vec : g.seg[0].b_vec;
    const int32_t* bidx = s1 ? g.seg[1].b_index : g.seg[0].b_index;
    const int32_t* bidx2 = g.seg[0].b_index2;
    RowSel aidx{}, aidx2{};
    if constexpr (AIDX) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            aidx.r[p] = (s1 && !PAIR) ? arow1[p] : arow0[p];
            aidx2.r[p] = arow1[p];
        }
    }
    const int k0 = (tile - (s1 ? g.tiles0 : 0)) * BK;
    int nb = g.n;                         // extent of the B operand along N as seen by this tile
    if (BIDX && !(g.bidx_mask & 1) && (g.nb_split >= g.n || n0 < g.nb_split)) bidx = nullptr;
    if (!bidx) bidx2 = nullptr;
    if (g.nb_split < g.n) {               // N-concatenated B: [b | b2], tiles never straddle the seam
        const bool second = n0 >= g.nb_split;
        if (BIDX && second && !(g.bidx_mask & 2)) { bidx = nullptr; bidx2 = nullptr; }
        b = second ? g.b2 : b;
        ldb = second ? g.ldb2 : ldb;
        nb = second ? g.n - g.nb_split : g.nb_split;
        n0 = second ? n0 - g.nb_split : n0;
    }
    // MODE 1 / 2: the fast loaders on every tile (MODE 2 zeroes the k positions past a ragged segment end
    // with selects) -- no branch decides which loads are issued, so the compiler's counted vmcnt waits
    // stay exact across the look-ahead
    if constexpr (MODE != 0) {
        constexpr bool RG = MODE == 2;
        if constexpr (A_T) load_rc<true, false, RG>(ra, a, lda, m0, g.m, k0, kdim, avec, t);
        else               load_kc<true, RG, AIDX>(ra, a, lda, m0, g.m, k0, kdim, avec, t, aidx);
        if constexpr (B_T) load_kc<true, RG>(rb, b, ldb, n0, nb, k0, kdim, bvec, t);
        else               load_rc<true, BIDX, RG>(rb, b, ldb, n0, nb, k0, kdim, bvec, t, bidx);
        if constexpr (PAIR && AIDX) load_kc<true, RG, true>(*r2, a, lda, m0, g.m, k0, kdim, avec, t, aidx2);
        if constexpr (PAIR && BIDX) load_rc<true, true, RG>(*r2, b, ldb, n0, nb, k0, kdim, bvec, t, bidx2);
    } else {
        if constexpr (A_T) load_rc<false>(ra, a, lda, m0, g.m, k0, kdim, avec, t);
        else               load_kc<false, false, AIDX>(ra, a, lda, m0, g.m, k0, kdim, avec, t, aidx);
        if constexpr (B_T) load_kc<false>(rb, b, ldb, n0, nb, k0, kdim, bvec, t);
        else               load_rc<false, BIDX>(rb, b, ldb, n0, nb, k0, kdim, bvec, t, bidx);
        if constexpr (PAIR && AIDX) load_kc<false, false, true>(*r2, a, lda, m0, g.m, k0, kdim, avec, t, aidx2);
        if constexpr (PAIR && BIDX) load_rc<false, true>(*r2, b, ldb, n0, nb, k0, kdim, bvec, t, bidx2);
    }
}

// one K-tile of MFMAs from the LDS buffers (at, bt)
template <bool A_T, bool B_T>
__device__ __forceinline__ void mma_tile(f32x16 (&acc)[2][2], const float* __restrict__ at,
                                         const float* __restrict__ bt, int wm, int wn, int l31, int h) {
    if constexpr (X3) { mma_tile_x3<A_T, B_T>(acc, at, bt, wm, wn, l31, h); return; }
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + l31;
            if constexpr (A_T) {
#pragma unroll
                for (int s = 0; s < 4; ++s) a[i][s] = at[(8 * q + 4 * h + s) * LDR + row];
            } else {
                const f32x4 v = *reinterpret_cast<const f32x4*>(at + row * LDK + 8 * q + 4 * h);
                a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wn * 64 + j * 32 + l31;
            if constexpr (B_T) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(bt + row * LDK + 8 * q + 4 * h);
                b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) b[j][s] = bt[(8 * q + 4 * h + s) * LDR + row];
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
}

template <bool A_T, bool B_T>
__device__ __forceinline__ void stage_tile(float* __restrict__ lds, int buf, const f32x4 (&ra)[NP],
                                           const f32x4 (&rb)[NP], int t) {
    float* at = lds + buf * TILE_FLOATS;
    float* bt = lds + (2 + buf) * TILE_FLOATS;
    if constexpr (X3) {
        if constexpr (A_T) store_rc_x3(at, ra, t); else store_kc_x3(at, ra, t);
        if constexpr (B_T) store_kc_x3(bt, rb, t); else store_rc_x3(bt, rb, t);
        return;
    }
    if constexpr (A_T) store_rc(at, ra, t); else store_kc(at, ra, t);
    if constexpr (B_T) store_kc(bt, rb, t); else store_rc(bt, rb, t);
}

// K loop.  Global -> register -> LDS staging, two LDS buffers, one barrier per K-tile, and the global
// loads run PF tiles ahead of the MFMAs (PF register sets used round-robin): under full load an HBM
// round trip is longer than one tile of MFMAs (64 x 64 cycles), so one tile of look-ahead left
// every wave waiting on vmcnt at the top of each iteration.
#ifndef PLNLP_GEMM_PF
#define PLNLP_GEMM_PF 2
#endif
template <bool A_T, bool B_T, int MODE, bool BIDX, bool AIDX = false>
__device__ __forceinline__ void k_loop(const GemmArgs& g, f32x16 (&acc)[2][2], float* __restrict__ lds,
                                       int64_t m0, int n0, int tb, int te, int t, int wm, int wn, int l31,
                                       int h) {
    constexpr int PF = PLNLP_GEMM_PF;
    f32x4 ra[PF][NP], rb[PF][NP];
    int arow0[NP], arow1[NP];
    if constexpr (AIDX) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            int64_t row = m0 + kc_row(t, p);
            row = row < g.m ? row : g.m - 1;
            arow0[p] = g.seg[0].a_index ? g.seg[0].a_index[row] : (int)row;
            arow1[p] = (g.nseg > 1 && g.seg[1].a_index) ? g.seg[1].a_index[row] : (int)row;
        }
    }
    int base = tb;
    // steady state: entered with all PF sets loaded and every tile of a round reloading its set,
    // UNCONDITIONALLY.  With any of those loads under a branch (`if (tile + PF < te)`, or a guarded
    // first fill) the compiler must assume the younger sets may never have been issued and waits for
    // vmcnt(0) before staging the oldest one -- i.e. for the younger sets too, which halves the
    // look-ahead.  Branch-free, it knows PF-1 sets are still in flight and waits for vmcnt(8 (PF-1)).
    if (te - tb >= 2 * PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) load_tile<A_T, B_T, MODE, BIDX, AIDX>(g, tb + d, ra[d], rb[d], m0, n0, t, arow0, arow1);
        do {
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                const int tile = base + d;
                const int buf = (tile - tb) & 1;
                stage_tile<A_T, B_T>(lds, buf, ra[d], rb[d], t);
                __syncthreads();
                load_tile<A_T, B_T, MODE, BIDX, AIDX>(g, tile + PF, ra[d], rb[d], m0, n0, t, arow0, arow1);
                mma_tile<A_T, B_T>(acc, lds + buf * TILE_FLOATS, lds + (2 + buf) * TILE_FLOATS, wm, wn, l31, h);
            }
            base += PF;
        } while (base + 2 * PF <= te);
    } else {
#pragma unroll
        for (int d = 0; d < PF; ++d)
            if (tb + d < te) load_tile<A_T, B_T, MODE, BIDX, AIDX>(g, tb + d, ra[d], rb[d], m0, n0, t, arow0, arow1);
    }
    // drain: the last < 2 PF tiles
    for (; base < te; base += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int tile = base + d;
            if (tile < te) {                                   // block-uniform
                const int buf = (tile - tb) & 1;
                stage_tile<A_T, B_T>(lds, buf, ra[d], rb[d], t);
                __syncthreads();
                if (tile + PF < te) load_tile<A_T, B_T, MODE, BIDX, AIDX>(g, tile + PF, ra[d], rb[d], m0, n0, t, arow0, arow1);
                mma_tile<A_T, B_T>(acc, lds + buf * TILE_FLOATS, lds + (2 + buf) * TILE_FLOATS, wm, wn, l31, h);
            }
        }
    }
}

// K loop of the split-bf16 form.  A K-tile is only 24 MFMAs (768 cycles) next to ~110 VALU instructions of
// operand splitting and addressing, so the two must overlap INSIDE a wave: the registers of tile i+1 (loaded one
// iteration earlier) are split and written to the other LDS buffer while the MFMAs of tile i run -- one basic
// block per tile (every look-ahead load unconditional, its tile index clamped to the slice's last tile; the junk
// it stages past the end is never read), the instruction mix pinned with sched_group_barrier: MFMA, then a few
// VALU, ... (a bf16 32x32x16 MFMA leaves ~7 issue slots before the next one can start).  One barrier per tile.
#if PLNLP_GEMM_X3
#define PLNLP_X3_PIPELINED 1
template <bool A_T, bool B_T, int MODE, bool BIDX, bool AIDX, bool PAIR, int D>
__device__ __forceinline__ void x3_step(const GemmArgs& g, f32x16 (&acc)[2][2], float* __restrict__ lds,
                                        f32x4 (&ra)[2][NP], f32x4 (&rb)[2][NP], f32x4 (&r2)[2][NP], int next_tile,
                                        int64_t m0, int n0, int t, int wm, int wn, int l31, int h, const int (&arow0)[NP],
                                        const int (&arow1)[NP]) {
    const float* at = lds + D * TILE_FLOATS;
    const float* bt = lds + (2 + D) * TILE_FLOATS;
    // fragments: both 32-column halves of B (3 terms each) and the first 32-row half of A; the second half of A
    // replaces the first after its 12 MFMAs (all twelve at once would not fit three waves per SIMD)
    bf16x8 a[3], b[2][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        a[s] = frag_x3<A_T>(at, s, wm * 64 + l31, h);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j][s] = frag_x3<!B_T>(bt, s, wn * 64 + j * 32 + l31, h);
    }
    // the set staged one step ago is free: fetch the tile two ahead into it
    load_tile<A_T, B_T, MODE, BIDX, AIDX, PAIR>(g, next_tile, ra[D], rb[D], m0, n0, t, arow0, arow1, &r2[D]);
    // the other set holds the next tile: it is split into the other LDS buffer BETWEEN the MFMAs -- region q of
    // the block = MFMA q + one 5-instruction stage of the split of one element pair (nothing crosses a
    // sched_barrier, so the wave always has VALU work to issue while the matrix pipe runs the MFMA)
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
    constexpr int TA2[6] = {2, 1, 1, 0, 0, 0}, TB2[6] = {0, 1, 0, 2, 1, 0};
    X3Split sa, sb;
    float* nat = lds + (D ^ 1) * TILE_FLOATS;
    float* nbt = lds + (2 + (D ^ 1)) * TILE_FLOATS;
    __builtin_amdgcn_sched_barrier(0);
    static_for<24>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int i = q / 12, u = (q % 12) >> 1, j = q & 1;
        // the A fragment of the second 32 rows replaces the first term by term, each as soon as the first half
        // has issued its last MFMA on that term (lo after q = 3, mid after q = 9, hi after q = 11); the second
        // half runs its six products in the order that needs them in that sequence
        if constexpr (q == 4)  a[2] = frag_x3<A_T>(at, 2, wm * 64 + 32 + l31, h);
        if constexpr (q == 10) a[1] = frag_x3<A_T>(at, 1, wm * 64 + 32 + l31, h);
        if constexpr (q == 12) a[0] = frag_x3<A_T>(at, 0, wm * 64 + 32 + l31, h);
        constexpr int ta = i == 0 ? TA[u] : TA2[u], tb = i == 0 ? TB[u] : TB2[u];
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta], b[j][tb], acc[i][j], 0, 0, 0);
        if constexpr (q < 4)        sa.template first<A_T, q & 3, PAIR && AIDX>(ra[D ^ 1], &r2[D ^ 1]);
        else if constexpr (q < 8)   sa.template second<q & 3>();
        else if constexpr (q == 8)  { sa.third(); sa.template store<A_T>(nat, t); }
        else if constexpr (q < 13)  sb.template first<!B_T, (q - 9) & 3, PAIR && BIDX>(rb[D ^ 1], &r2[D ^ 1]);
        else if constexpr (q < 17)  sb.template second<(q - 13) & 3>();
        else if constexpr (q == 17) { sb.third(); sb.template store<!B_T>(nbt, t); }
        __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
}
template <bool A_T, bool B_T, int MODE, bool BIDX, bool AIDX = false, bool PAIR = false>
__device__ __forceinline__ void k_loop_x3(const GemmArgs& g, f32x16 (&acc)[2][2], float* __restrict__ lds,
                                          int64_t m0, int n0, int tb, int te, int t, int wm, int wn, int l31,
                                          int h) {
    if (te <= tb) return;                       // an empty split-K slice (block-uniform)
    f32x4 ra[2][NP], rb[2][NP];
    f32x4 r2[PAIR ? 2 : 1][NP];                 // PAIR: the second factor of the gathered operand's rows
    int arow0[NP], arow1[NP];
    if constexpr (AIDX) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            int64_t row = m0 + kc_row(t, p);
            row = row < g.m ? row : g.m - 1;
            arow0[p] = g.seg[0].a_index ? g.seg[0].a_index[row] : (int)row;
            if constexpr (PAIR) arow1[p] = g.seg[0].a_index2[row];         // (one segment: see load_tile)
            else arow1[p] = (g.nseg > 1 && g.seg[1].a_index) ? g.seg[1].a_index[row] : (int)row;
        }
    }
    const int last = te - 1;
    f32x4 (&q2)[2][NP] = reinterpret_cast<f32x4 (&)[2][NP]>(r2);     // (only indexed past 0 when PAIR)
    load_tile<A_T, B_T, MODE, BIDX, AIDX, PAIR>(g, tb, ra[0], rb[0], m0, n0, t, arow0, arow1, &q2[0]);
    load_tile<A_T, B_T, MODE, BIDX, AIDX, PAIR>(g, tb + 1 < last ? tb + 1 : last, ra[1], rb[1], m0, n0, t, arow0, arow1,
                                                &q2[PAIR ? 1 : 0]);
    if constexpr (PAIR) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {           // (the split below must see the f32 product: see X3Split::first)
            if constexpr (AIDX) ra[0][p] = mul_rounded(ra[0][p], q2[0][p]);
            else rb[0][p] = mul_rounded(rb[0][p], q2[0][p]);
        }
    }
    stage_tile<A_T, B_T>(lds, 0, ra[0], rb[0], t);
    __syncthreads();
    int i = tb;
    for (; i + 2 <= te; i += 2) {
        x3_step<A_T, B_T, MODE, BIDX, AIDX, PAIR, 0>(g, acc, lds, ra, rb, q2, i + 2 < last ? i + 2 : last, m0, n0, t, wm,
                                                     wn, l31, h, arow0, arow1);
        x3_step<A_T, B_T, MODE, BIDX, AIDX, PAIR, 1>(g, acc, lds, ra, rb, q2, i + 3 < last ? i + 3 : last, m0, n0, t, wm,
                                                     wn, l31, h, arow0, arow1);
    }
    if (i < te)
        x3_step<A_T, B_T, MODE, BIDX, AIDX, PAIR, 0>(g, acc, lds, ra, rb, q2, last, m0, n0, t, wm, wn, l31, h, arow0,
                                                     arow1);
}
#endif

// MODE (see load_tile): separate kernels so the hot loop of the aligned case carries no guarded
// code at all (pure dwordx4 loads, nothing between their issue and the MFMAs).
// workgroups per CU: 3 at K-tile depth 16 (<= 168 registers), except the split-bf16 kernels whose guarded or
// ragged loaders with a row-contiguous operand, or the second factor of a pair-gathered operand, do not fit
// that budget without spilling
#ifndef PLNLP_X3_WGRAD_WGS
#define PLNLP_X3_WGRAD_WGS 3      // (A/B switch: workgroups per CU of the split-bf16 weight-gradient kernels)
#endif
template <bool A_T, bool B_T, int MODE, bool BIDX = false, bool AIDX = false, bool PAIR = false>
__global__ __launch_bounds__(256, (BK == 16 && !(X3 && (PAIR || MODE == 0 || (MODE == 2 && (A_T || !B_T))))) ? ((X3 && A_T && !B_T) ? PLNLP_X3_WGRAD_WGS : 3) : 2)
void gemm_f32_kernel(GemmArgs g, Epi epi) {
    static_assert(!PAIR || X3, "pair-gathered operands exist in the split-bf16 build only");
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE_FLOATS];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    // block -> tile: the gn column tiles of one row panel run back to back on ONE XCD (blocks
    // are dealt to XCDs round-robin), so the A panel they share is fetched from HBM once
    // and served from that XCD's L2 afterwards.  Pure performance: any mapping is correct.
    int64_t mt; int nt;
    {
        const int64_t id = blockIdx.x, gn = g.gn, total = g.gm * gn;
        const int64_t per_xcd = (total / (8 * gn)) * gn;     // whole row panels per XCD
        const int64_t body = per_xcd * 8;
        if (id < body) {
            const int64_t xcd = id & 7, local = id >> 3;
            const int64_t t = xcd * per_xcd + local;
            mt = t / gn; nt = (int)(t % gn);
        } else {
            mt = id / gn; nt = (int)(id % gn);
        }
    }
    const int64_t m0 = (mt + g.mt0) * BM;
    const int n0 = (nt + g.nt0) * BN;

    // K-tile range of this split slice
    const int z = blockIdx.z;
    const int per = (g.tiles_total + g.split_k - 1) / g.split_k;
    const int tb = z * per;
    const int te = (tb + per) < g.tiles_total ? (tb + per) : g.tiles_total;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

#ifdef PLNLP_X3_PIPELINED
    k_loop_x3<A_T, B_T, MODE, BIDX, AIDX, PAIR>(g, acc, lds, m0, n0, tb, te, t, wm, wn, l31, h);
#else
    k_loop<A_T, B_T, MODE, BIDX, AIDX>(g, acc, lds, m0, n0, tb, te, t, wm, wn, l31, h);
#endif

    // ---- write back.  The MFMA C/D map (col = lane&31, row = (q&3) + 8*(q>>2) + 4*(lane>>5)) would
    // give 64 scattered 4-byte stores per lane; instead the block tile is transposed through LDS
    // (free after the last K-tile) and leaves as 16-byte row stores, with the epilogue applied on
    // float4 (one wide load each for bias / gate / accumulate operands).
    constexpr int CS = BN + 4;                       // LDS row stride of the staged C tile (floats)
    // the whole 128-row tile when the operand buffers are large enough (BK = 32), else in two 64-row halves
    constexpr bool SPLIT_C = BM * CS > 4 * TILE_FLOATS;
    static_assert((BM / 2) * CS <= 4 * TILE_FLOATS, "half a C tile must fit the operand buffers");
    float* cbase = g.c;
    int64_t ldc = g.ldc;
    const bool raw = g.split_k > 1;
    if (raw) { cbase = g.c + (int64_t)(z + g.z0) * g.ws_stride; ldc = g.n; }
    if constexpr (!SPLIT_C) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    lds[(wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * CS + wn * 64 + j * 32 + l31] = acc[i][j][q];
        __syncthreads();
    }
    const bool vec_out = g.vec_store;
    // thread t owns column group c4 of rows rl0 + 8u: the column is the same for all 16 rows
    const int c4 = (t & 31) * 4, rl0 = t >> 5;
    const int col = n0 + c4;
    const bool col_live = col < g.n;                 // (no early return: the split form has barriers below)
    if (!SPLIT_C && !col_live) return;
    const bool colvec = vec_out && col + 3 < g.n;
    // epilogue operands as float4 registers: bias once per thread, gate / accumulate rows 8 at a time
    // ahead of the loop that uses them (their global-load latency used to sit between the LDS read
    // and the store of every row: +11 % on the forward GEMM)
    const bool pre = !raw && epi.flags && epi.vec4 && colvec && !(epi.flags & PLNLP_EPI_ADDEND);
    const bool second_col = !raw && col >= g.n_split;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pre && (epi.flags & PLNLP_EPI_BIAS)) bias4 = *reinterpret_cast<const float4*>(epi.bias + col);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if constexpr (SPLIT_C) {                     // rows [64 half, 64 half + 64): the waves wm == half own them
            __syncthreads();
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 16; ++q)
                            lds[(i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * CS + wn * 64 + j * 32 + l31] = acc[i][j][q];
            }
            __syncthreads();
            if (!col_live) continue;
        }
        // gate / accumulate operands are fetched EPF rows ahead of the loop that uses them (fewer at BK = 16,
        // whose register budget is 168 for three waves per SIMD)
        constexpr int EPF = BK == 16 ? 4 : 8;
#pragma unroll
        for (int u0 = 0; u0 < 8; u0 += EPF) {
        float4 g4[EPF], p4[EPF];
        if (pre && (epi.flags & (PLNLP_EPI_GATE | PLNLP_EPI_ACCUM))) {
#pragma unroll
            for (int ue = 0; ue < EPF; ++ue) {
                const int uu = u0 + ue;
                int64_t row = m0 + rl0 + 8 * (half * 8 + uu);
                row = row < g.m ? row : g.m - 1;
                if (epi.flags & PLNLP_EPI_GATE) {
                    const int64_t gr = epi.gate_index ? (int64_t)epi.gate_index[row] : row;
                    g4[ue] = *reinterpret_cast<const float4*>(epi.gate + gr * epi.ld_gate + col);
                }
                if (epi.flags & PLNLP_EPI_ACCUM) {
                    const float* prow = second_col ? g.c2 + row * g.ldc2 - g.n_split : cbase + row * ldc;
                    p4[ue] = *reinterpret_cast<const float4*>(prow + col);
                }
            }
        }
#pragma unroll
        for (int ue = 0; ue < EPF; ++ue) {
            const int uu = u0 + ue;
            const int rl = rl0 + 8 * (half * 8 + uu);
            const int64_t row = m0 + rl;
            if (row >= g.m) continue;
            float4 v = *reinterpret_cast<const float4*>(lds + (SPLIT_C ? rl - 64 * half : rl) * CS + c4);
            float* orow = second_col ? g.c2 + row * g.ldc2 - g.n_split : cbase + row * ldc;   // index with the global column
            if (colvec) {
                if (pre) v = epi_apply4_pre(epi, v, row, col, g.n, bias4, g4[ue], p4[ue]);
                else if (!raw) v = epi_apply4(epi, v, row, col, g.n, orow);
                *reinterpret_cast<float4*>(orow + col) = v;
            } else {
                const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (col + c >= g.n) break;
                    const bool sec = !raw && (col + c) >= g.n_split;
                    float* p = sec ? g.c2 + row * g.ldc2 + (col + c - g.n_split) : cbase + row * ldc + col + c;
                    float x = e4[c];
                    if (!raw && epi.flags) x = epi_apply(epi, x, row, col + c, g.n, (epi.flags & PLNLP_EPI_ACCUM) ? *p : 0.f);
                    *p = x;
                }
            }
        }
        }
    }
}

// every kernel variant of this K-tile depth behind one function (called from gemm_impl)
int launch_kernels(const GemmArgs& ga, int md, dim3 grid, int a_trans, int b_trans, hipStream_t s, const Epi& e) {
#define PLNLP_GEMM_M(AT, BT)                                                                              \
    switch (md) {                                                                                        \
        case 1: hipLaunchKernelGGL((gemm_f32_kernel<AT, BT, 1>), grid, dim3(256), 0, s, ga, e); break;       \
        case 2: hipLaunchKernelGGL((gemm_f32_kernel<AT, BT, 2>), grid, dim3(256), 0, s, ga, e); break;       \
        default: hipLaunchKernelGGL((gemm_f32_kernel<AT, BT, 0>), grid, dim3(256), 0, s, ga, e); break;      \
    }
#if PLNLP_GEMM_X3 && defined(PLNLP_X3_PIPELINED)
    if (ga.seg[0].a_index2) {         // rows of A = products of two gathered rows ((!a_trans, b_trans, one segment) checked by the caller)
        switch (md) {
            case 1: hipLaunchKernelGGL((gemm_f32_kernel<false, true, 1, false, true, true>), grid, dim3(256), 0, s, ga, e); break;
            case 2: hipLaunchKernelGGL((gemm_f32_kernel<false, true, 2, false, true, true>), grid, dim3(256), 0, s, ga, e); break;
            default: hipLaunchKernelGGL((gemm_f32_kernel<false, true, 0, false, true, true>), grid, dim3(256), 0, s, ga, e); break;
        }
        return launch_status();
    }
    if (ga.seg[0].b_index2) {         // rows of B likewise ((a_trans, !b_trans, one segment) checked by the caller)
        switch (md) {
            case 1: hipLaunchKernelGGL((gemm_f32_kernel<true, false, 1, true, false, true>), grid, dim3(256), 0, s, ga, e); break;
            case 2: hipLaunchKernelGGL((gemm_f32_kernel<true, false, 2, true, false, true>), grid, dim3(256), 0, s, ga, e); break;
            default: hipLaunchKernelGGL((gemm_f32_kernel<true, false, 0, true, false, true>), grid, dim3(256), 0, s, ga, e); break;
        }
        return launch_status();
    }
#else
    if (ga.seg[0].a_index2 || ga.seg[0].b_index2) return PLNLP_E_UNSUPPORTED;
#endif
    if (ga.seg[0].a_index || (ga.nseg > 1 && ga.seg[1].a_index)) {     // (!a_trans, b_trans) checked by the caller
        switch (md) {
            case 1: hipLaunchKernelGGL((gemm_f32_kernel<false, true, 1, false, true>), grid, dim3(256), 0, s, ga, e); break;
            case 2: hipLaunchKernelGGL((gemm_f32_kernel<false, true, 2, false, true>), grid, dim3(256), 0, s, ga, e); break;
            default: hipLaunchKernelGGL((gemm_f32_kernel<false, true, 0, false, true>), grid, dim3(256), 0, s, ga, e); break;
        }
        return launch_status();
    }
    if (ga.seg[0].b_index) {      // (a_trans, !b_trans) checked by the caller
        switch (md) {
            case 1: hipLaunchKernelGGL((gemm_f32_kernel<true, false, 1, true>), grid, dim3(256), 0, s, ga, e); break;
            case 2: hipLaunchKernelGGL((gemm_f32_kernel<true, false, 2, true>), grid, dim3(256), 0, s, ga, e); break;
            default: hipLaunchKernelGGL((gemm_f32_kernel<true, false, 0, true>), grid, dim3(256), 0, s, ga, e); break;
        }
        return launch_status();
    }
    if (a_trans) { if (b_trans) { PLNLP_GEMM_M(true, true) } else { PLNLP_GEMM_M(true, false) } }
    else         { if (b_trans) { PLNLP_GEMM_M(false, true) } else { PLNLP_GEMM_M(false, false) } }
#undef PLNLP_GEMM_M
    return launch_status();
}
}  // namespace GEMM_NS

#if PLNLP_GEMM_BK != 32
}  // namespace plnlp
#else
namespace g16 {      // the depth-16 translation unit
int launch_kernels(const GemmArgs& ga, int md, dim3 grid, int a_trans, int b_trans, hipStream_t s, const Epi& e);
}
namespace x16 {      // the split-bf16 translation unit
int launch_kernels(const GemmArgs& ga, int md, dim3 grid, int a_trans, int b_trans, hipStream_t s, const Epi& e);
}

// sum split-K slices in slice order, apply the epilogue.  16 bytes per thread, 8 slices in flight.
// c2 != nullptr: result columns >= n_split go to c2[:, col - n_split] (the pair form; no epilogue there)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int split_k,
                                                            int64_t stride, float* __restrict__ c, int64_t ldc,
                                                            int64_t m, int n, Epi epi, float* __restrict__ c2,
                                                            int64_t ldc2, int n_split, int64_t row0) {
    // row0: the workspace holds rows [row0, row0 + m) of the result (0 for every current caller)
    const int64_t total = m * (int64_t)n;
    const bool vec = (n % 4 == 0) && (stride % 4 == 0) && (ldc % 4 == 0) && ((uintptr_t)ws % 16 == 0) &&
                     ((uintptr_t)c % 16 == 0) &&
                     (!c2 || ((n_split % 4 == 0) && (ldc2 % 4 == 0) && ((uintptr_t)c2 % 16 == 0)));
    if (vec) {
        const int64_t total4 = total >> 2;
        for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (int64_t)gridDim.x * 256) {
            const float* p = ws + i4 * 4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            int zz = 0;
            for (; zz + 8 <= split_k; zz += 8) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (int64_t)(zz + u) * stride);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; zz < split_k; ++zz) acc += *reinterpret_cast<const f32x4*>(p + (int64_t)zz * stride);
            const int64_t i = i4 * 4;
            const int64_t row = row0 + i / n;
            const int col = (int)(i - (i / n) * n);
            float4 y = make_float4(acc.x, acc.y, acc.z, acc.w);
            if (c2 && col >= n_split) {
                *reinterpret_cast<float4*>(c2 + row * ldc2 + (col - n_split)) = y;
                continue;
            }
            float* o = c + row * ldc + col;
            y = epi_apply4(epi, y, row, col, n, c + row * ldc);
            *reinterpret_cast<float4*>(o) = y;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int zz = 0; zz < split_k; ++zz) v += ws[(int64_t)zz * stride + i];
        const int64_t row = row0 + i / n;
        const int col = (int)(i - (i / n) * n);
        if (c2 && col >= n_split) { c2[row * ldc2 + (col - n_split)] = v; continue; }
        float* p = c + row * ldc + col;
        if (epi.flags) {
            const float prev = (epi.flags & PLNLP_EPI_ACCUM) ? *p : 0.f;
            v = epi_apply(epi, v, row, col, n, prev);
        }
        *p = v;
    }
}

}  // namespace plnlp

// Does this launch run the stationary-weights form (given a large enough b_terms buffer)?  Returns the column-tile width
// (x 32 columns) it would use, 0 when the tile kernels take it; ks[] = the K of each segment.  Everything the decision
// depends on is an argument of the launch -- the Python host asks the same question through plnlp_gemm_stationary_applies
// BEFORE it decides about split-K and lends the buffer, so the two sides cannot disagree.
static int stationary_form(const plnlp_gemm_operand* segs, int n_seg, int a_trans, int b_trans, const float* c, int64_t ldc,
                           int64_t m, int64_t n, const float* c2, int64_t ldc2, int64_t n_split, int64_t (&ks)[2]) {
    using namespace plnlp;
    ks[0] = ks[1] = 0;
    if (!segs || n_seg < 1 || n_seg > 2 || segs[0].math != PLNLP_GEMM_MATH_BF16X3 || a_trans) return 0;
    if (m < x3s::min_rows() || n % 4 != 0 || n < 16) return 0;
    const bool vec_store = (ldc % 4 == 0) && ((uintptr_t)c % 16 == 0) &&
                           (!c2 || ((n_split % 4 == 0) && (ldc2 % 4 == 0) && ((uintptr_t)c2 % 16 == 0)));
    if (!vec_store) return 0;
    bool ragged_k = false;
    for (int si = 0; si < n_seg; ++si) {
        const plnlp_gemm_operand& o = segs[si];
        const bool a_vec = ((uintptr_t)o.a % 16 == 0) && (o.lda % 4 == 0);
        if (!a_vec || o.k <= 0 || o.k % 4 != 0 || o.b_index || o.a_index2 || o.b_index2 || (o.a_index && !b_trans)) return 0;
        ks[si] = o.k;
        ragged_k |= (o.k % 16) != 0;
    }
    // h = 200 (citation2): measured both ways on MI355X (profiles/r04_gemm_tile_width.jsonl) -- K a multiple of 16
    // (the first layer's padded 192): the tile kernel 1.85 ms against 2.17 (its aligned loaders are at their best,
    // the 224-column tile wastes 11 % of its MFMAs); ragged K (200): the tile kernel's select-zeroed loaders 2.22 ms
    // against 1.94 here, where the padding lives in the weight image
    // -- unless the launch is large enough for the whole-block kernel (gemm_x3b.hip), which takes both
    if (n > 192 && n <= 224 && !ragged_k && !x3b::applies(m, 7)) return 0;
    return x3s::pick_nb(m, n);
}

// The wide weight-gradient form (gemm_wgw.hip) for this launch: its arguments (everything but slices / ws), and the K slices it
// wants -- 0 where it does not apply: one K-segment, A^T B with A as stored, split-bf16 math, and wgw::slices_for's shape
// rule (B in one or two buffers, its rows gathered or not).  The launch takes the form exactly when the caller cuts K into
// this many slices.
static int wide_wgrad_slices(const plnlp_gemm_operand* segs, int n_seg, int a_trans, int b_trans, int64_t m, int64_t n,
                             const float* b2, int64_t ldb2, int64_t nb_split, int bidx_mask, plnlp::wgw::Args* out) {
    if (!segs || n_seg != 1 || !a_trans || b_trans || segs[0].math != PLNLP_GEMM_MATH_BF16X3) return 0;
    const plnlp_gemm_operand& o = segs[0];
    if (o.a_index || o.a_index2 || o.b_index2 || !o.a || !o.b || m > 0x7FFFFFFF || n > 0x7FFFFFFF) return 0;
    plnlp::wgw::Args w{};
    w.a = o.a; w.lda = o.lda; w.b = o.b; w.ldb = o.ldb; w.m = (int)m; w.n = (int)n; w.k = o.k;
    w.b2 = b2; w.ldb2 = ldb2; w.nb_split = b2 ? (int)nb_split : (int)n;
    w.b_index = o.b_index; w.bidx_mask = o.b_index ? bidx_mask & (b2 ? 3 : 1) : 0;
    if (w.b_index && !w.bidx_mask) w.b_index = nullptr;
    if (out) *out = w;
    return plnlp::wgw::slices_for(w);
}
extern "C" int plnlp_gemm_wide_wgrad_slices(const plnlp_gemm_operand* segs, int n_seg, int a_trans, int b_trans,
                                            int64_t m, int64_t n, const float* b2, int64_t ldb2, int64_t nb_split,
                                            int b_index_on) {
    return wide_wgrad_slices(segs, n_seg, a_trans, b_trans, m, n, b2, ldb2, nb_split, b_index_on, nullptr);
}
extern "C" int plnlp_gemm_stationary_applies(const plnlp_gemm_operand* segs, int n_seg, int a_trans, int b_trans,
                                             const float* c, int64_t ldc, int64_t m, int64_t n, const float* c2,
                                             int64_t ldc2, int64_t n_split) {
    int64_t ks[2];
    return stationary_form(segs, n_seg, a_trans, b_trans, c, ldc, m, n, c2, ldc2, n_split, ks) > 0 ? 1 : 0;
}

static int gemm_impl(const plnlp_gemm_operand* segs, int n_seg, int a_trans, int b_trans, float* c, int64_t ldc,
                     int64_t m, int64_t n, float* c2, int64_t ldc2, int64_t n_split, const plnlp_epilogue* epi,
                     int split_k, float* workspace, int64_t workspace_floats, void* stream,
                     const float* b2 = nullptr, int64_t ldb2 = 0, int64_t nb_split = -1, int bidx_mask = 3);

extern "C" int plnlp_gemm_concat_b_f32(const plnlp_gemm_operand* seg, const float* b2, int64_t ldb2,
                                       int64_t nb_split, int a_trans, int b_trans, float* c, int64_t ldc,
                                       int64_t m, int64_t n, const plnlp_epilogue* epi, int split_k,
                                       float* workspace, int64_t workspace_floats, void* stream) {
    if (!seg || !b2) return PLNLP_E_NULL;
    if (nb_split <= 0 || nb_split >= n || nb_split % 128 != 0 || ldc < n) return PLNLP_E_SHAPE;
    if (ldb2 < (b_trans ? seg->k : n - nb_split)) return PLNLP_E_SHAPE;
    return gemm_impl(seg, 1, a_trans, b_trans, c, ldc, m, n, nullptr, 0, n, epi, split_k, workspace,
                     workspace_floats, stream, b2, ldb2, nb_split);
}

extern "C" int plnlp_gemm_pair_f32(const plnlp_gemm_operand* seg, const float* b2, int64_t ldb2, int64_t nb_split,
                                   int b_index_on, int a_trans, int b_trans, float* c, int64_t ldc, float* c2, int64_t ldc2,
                                   int64_t n_split, int64_t m, int64_t n, int split_k, float* workspace,
                                   int64_t workspace_floats, void* stream) {
    if (!seg || !c2) return PLNLP_E_NULL;
    if (n_split <= 0 || n_split >= n || ldc < n_split || ldc2 < n - n_split) return PLNLP_E_SHAPE;
    if (b2) {
        if (nb_split <= 0 || nb_split >= n || nb_split % 128 != 0) return PLNLP_E_SHAPE;
        if (ldb2 < (b_trans ? seg->k : n - nb_split)) return PLNLP_E_SHAPE;
    }
    if (b_index_on < 1 || b_index_on > 3) return PLNLP_E_SHAPE;
    return gemm_impl(seg, 1, a_trans, b_trans, c, ldc, m, n, c2, ldc2, n_split, nullptr, split_k, workspace,
                     workspace_floats, stream, b2, ldb2, b2 ? nb_split : -1, b_index_on);
}

extern "C" int plnlp_gemm_f32(const plnlp_gemm_operand* segs, int n_seg, int a_trans, int b_trans, float* c,
                              int64_t ldc, int64_t m, int64_t n, const plnlp_epilogue* epi, int split_k,
                              float* workspace, int64_t workspace_floats, void* stream) {
    if (ldc < n) return PLNLP_E_SHAPE;
    return gemm_impl(segs, n_seg, a_trans, b_trans, c, ldc, m, n, nullptr, 0, n, epi, split_k, workspace,
                     workspace_floats, stream);
}

extern "C" void plnlp_gemm_stationary_tuning(int nb, int min_rows) { plnlp::x3s::set_tuning(nb, min_rows); }
extern "C" void plnlp_gemm_block_tuning(int mode) { plnlp::x3b::set_mode(mode); }

extern "C" int plnlp_gemm_rowdot_tiles(int64_t m, int64_t n) {
    if (m <= 0 || n <= 0) return 0;
    const int nb = plnlp::x3s::pick_nb(m, n);
    if (nb != 4 && nb != 7 && nb != 8) return 0;
    return (int)((n + 32 * nb - 1) / (32 * nb));
}

extern "C" int64_t plnlp_gemm_b_terms_bytes(int64_t m, int64_t n, int64_t k0, int64_t k1) {
    if (m <= 0 || n <= 0 || k0 <= 0 || k1 < 0) return 0;
    const int64_t k[2] = {k0, k1};
    int64_t need = 0;                       // whatever tile width the launch picks (or a measurement forces)
    for (int nb : {8, 7, 4, 2, 1}) {
        const int64_t b = plnlp::x3s::image_bytes(n, k, k1 > 0 ? 2 : 1, nb);
        need = b > need ? b : need;
    }
    (void)m;
    return need;
}

extern "C" int plnlp_gemm_split_out_f32(const plnlp_gemm_operand* segs, int n_seg, int a_trans, int b_trans,
                                        float* c, int64_t ldc, float* c2, int64_t ldc2, int64_t n_split,
                                        int64_t m, int64_t n, const plnlp_epilogue* epi, void* stream) {
    if (!c2) return PLNLP_E_NULL;
    if (n_split <= 0 || n_split >= n || ldc < n_split || ldc2 < n - n_split) return PLNLP_E_SHAPE;
    if (epi && epi->flags) return PLNLP_E_UNSUPPORTED;     // plain product only
    return gemm_impl(segs, n_seg, a_trans, b_trans, c, ldc, m, n, c2, ldc2, n_split, nullptr, 1, nullptr, 0, stream);
}

static int gemm_impl(const plnlp_gemm_operand* segs, int n_seg, int a_trans, int b_trans, float* c, int64_t ldc,
                     int64_t m, int64_t n, float* c2, int64_t ldc2, int64_t n_split, const plnlp_epilogue* epi,
                     int split_k, float* workspace, int64_t workspace_floats, void* stream, const float* b2,
                     int64_t ldb2, int64_t nb_split, int bidx_mask) {
    using namespace plnlp;
    if (n_seg < 1 || n_seg > 2 || m < 0 || n < 0 || n > 0x7FFFFFF0) return PLNLP_E_SHAPE;
    if (m == 0 || n == 0) return 0;           // (an empty result has no storage: its pointer may be NULL)
    if (!segs || !c) return PLNLP_E_NULL;
    if (split_k < 1) split_k = 1;
    // K-tile depth of this launch: the weight gradients (row-contiguous A, a reduction over 10^5 .. 10^6 rows cut
    // along K) keep 32; everything else runs the depth-16 kernels, three workgroups per CU
    const int math = segs[0].math;
    if (math != PLNLP_GEMM_MATH_F32 && math != PLNLP_GEMM_MATH_BF16X3) return PLNLP_E_UNSUPPORTED;
    const int BK = (math == PLNLP_GEMM_MATH_BF16X3) ? 16 : (a_trans ? 32 : 16);
    GemmArgs g{};
    g.nseg = n_seg;
    int tiles[2] = {0, 0};
    for (int s = 0; s < n_seg; ++s) {
        const plnlp_gemm_operand& o = segs[s];
        if (!o.a || !o.b) return PLNLP_E_NULL;
        if (o.k <= 0 || o.k > 0x7FFFFF00) return PLNLP_E_SHAPE;
        const int64_t a_inner = a_trans ? m : o.k, b_inner = b_trans ? o.k : (b2 ? nb_split : n);
        if (o.lda < a_inner || o.ldb < b_inner) return PLNLP_E_SHAPE;
        Seg& d = g.seg[s];
        d.a = o.a; d.lda = o.lda; d.b = o.b; d.ldb = o.ldb; d.k = (int)o.k;
        d.b_index = o.b_index;
        d.a_index = o.a_index;
        d.a_index2 = o.a_index2;
        d.b_index2 = o.b_index2;
        if ((o.a_index2 && (!o.a_index || n_seg != 1)) || (o.b_index2 && (!o.b_index || n_seg != 1 || b2)))
            return PLNLP_E_UNSUPPORTED;
        if ((o.a_index2 || o.b_index2) && segs[0].math != PLNLP_GEMM_MATH_BF16X3) return PLNLP_E_UNSUPPORTED;
        if (o.b_index && !(a_trans && !b_trans && n_seg == 1)) return PLNLP_E_UNSUPPORTED;
        if (o.a_index && !(!a_trans && b_trans)) return PLNLP_E_UNSUPPORTED;
        d.a_vec = ((uintptr_t)o.a % 16 == 0) && (o.lda % 4 == 0);
        d.b_vec = ((uintptr_t)o.b % 16 == 0) && (o.ldb % 4 == 0);
        tiles[s] = (int)((o.k + BK - 1) / BK);
    }
    g.tiles0 = tiles[0];
    g.tiles_total = tiles[0] + tiles[1];
    if (split_k > g.tiles_total) split_k = g.tiles_total;
    g.m = m; g.n = (int)n; g.split_k = split_k; g.ws_stride = m * n;
    g.c2 = c2; g.ldc2 = ldc2; g.n_split = (int)n_split;
    g.b2 = b2; g.ldb2 = ldb2; g.nb_split = b2 ? (int)nb_split : (int)n;
    g.bidx_mask = bidx_mask;
    g.vec_store = (n % 4 == 0) && (ldc % 4 == 0) && ((uintptr_t)c % 16 == 0) &&
                  (!c2 || ((n_split % 4 == 0) && (ldc2 % 4 == 0) && ((uintptr_t)c2 % 16 == 0))) &&
                  (split_k <= 1 || ((uintptr_t)workspace % 16 == 0));
    Epi e;
    if (int rc = make_epi(epi, &e, /*allow_adam=*/false, /*allow_rowdot=*/true)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int64_t gm = (m + BM - 1) / BM, gn = (n + BN - 1) / BN;
    if (gm > 0x7FFFFFFF || gn > 65535 || split_k > 65535) return PLNLP_E_SHAPE;
    if (split_k > 1) {
        if (!workspace) return PLNLP_E_NULL;
        if (workspace_floats < (int64_t)split_k * m * n) return PLNLP_E_WORKSPACE;
        g.c = workspace; g.ldc = n;
    } else {
        g.c = c; g.ldc = ldc;
    }
    if (gm * gn > 0x7FFFFFFF) return PLNLP_E_SHAPE;
    // aligned operands take the fast loaders (edge rows clamped).
    bool aligned = true, ragged = false;
    for (int si = 0; si < n_seg; ++si) {
        aligned = aligned && g.seg[si].a_vec && g.seg[si].b_vec;
        if (g.seg[si].k % BK != 0) {
            ragged = true;
            // a K-contiguous operand is cut in whole 16-byte groups: the ragged form needs k % 4 == 0
            if ((!a_trans || b_trans) && g.seg[si].k % 4 != 0) aligned = false;
        }
    }
    if (b2) aligned = aligned && ((uintptr_t)b2 % 16 == 0) && (ldb2 % 4 == 0) && ((n - nb_split) % 4 == 0);
    if (a_trans) aligned = aligned && (m % 4 == 0) && m >= 4;    // row-contiguous operands move 4 rows per load
    if (!b_trans) aligned = aligned && (n % 4 == 0) && n >= 4;
    int mode = !aligned ? 0 : (ragged ? 2 : 1);
    // ---- the stationary-weights form (gemm_x3s.hip): A an activation matrix with K-contiguous rows, B the weights, the
    // caller lent a buffer for B's pre-split image.  Same bits as the kernels below (same split, same six products in the
    // same order per K-step of 16), so which one runs is a pure speed choice (stationary_form: ONE rule, also behind
    // plnlp_gemm_stationary_applies).
    if (segs[0].b_terms && ((uintptr_t)segs[0].b_terms % 16 == 0)) {
        int64_t ks[2] = {0, 0};
        const int nb = stationary_form(segs, n_seg, a_trans, b_trans, c, ldc, m, n, c2, ldc2, n_split, ks);
        if (nb > 0 && segs[0].b_terms_bytes >= x3s::image_bytes(n, ks, n_seg, nb)) {
            x3s::SplitArgs sp{};
            x3s::Args xa{};
            for (int si = 0; si < n_seg; ++si) {
                const Seg& d = g.seg[si];
                sp.b[si] = d.b; sp.ldb[si] = d.ldb; sp.k[si] = d.k;
                xa.a[si] = d.a; xa.lda[si] = d.lda; xa.k[si] = d.k; xa.a_index[si] = d.a_index;
            }
            sp.b2 = b2; sp.ldb2 = ldb2; sp.nb_split = b2 ? (int)nb_split : (int)n;
            sp.b_trans = b_trans; sp.nseg = n_seg; sp.n = (int)n; sp.image = segs[0].b_terms;
            sp.ks0 = (int)((ks[0] + 15) / 16);
            sp.ks_total = sp.ks0 + (n_seg > 1 ? (int)((ks[1] + 15) / 16) : 0);
            xa.nseg = n_seg; xa.ks0 = sp.ks0; xa.ks_total = sp.ks_total; xa.image = segs[0].b_terms;
            xa.c = c; xa.ldc = ldc; xa.c2 = c2; xa.ldc2 = ldc2; xa.n_split = c2 ? (int)n_split : (int)n;
            xa.m = m; xa.n = (int)n;
            bool ragged_k = false;
            int k_steps = 0;
            for (int si = 0; si < n_seg; ++si) { ragged_k |= (ks[si] % 16) != 0; k_steps += (int)((ks[si] + 15) / 16); }
            count_launch(x3b::takes(m, n, nb, ragged_k, k_steps, e) ? LK_GEMM_X3B : LK_GEMM_X3S);      // (x3s::launch asks the same question)
            return x3s::launch(sp, xa, nb, e, s);
        }
    }
    if (e.flags & PLNLP_EPI_ROWDOT) return PLNLP_E_UNSUPPORTED;      // the row-dot epilogue lives in the stationary kernel only
    const int reduce_slices = split_k;
    // ---- the wide weight gradient (gemm_wgw.hip): whole 224- / 256-wide blocks of the result per workgroup and K slice; taken
    // when the caller ASKS for it (PLNLP_GEMM_FLAG_WIDE_WGRAD) and cut K into exactly the slices plnlp_gemm_wide_wgrad_slices
    // names for this launch (its workspace is then the right size)
    wgw::Args w{};
    if ((segs[0].flags & PLNLP_GEMM_FLAG_WIDE_WGRAD) && split_k > 1 &&
        split_k == wide_wgrad_slices(segs, n_seg, a_trans, b_trans, m, n, b2, ldb2, nb_split, bidx_mask, &w)) {
        w.slices = split_k; w.ws = workspace;
        float* colsum = segs[0].a_colsum;
        const bool vec = ((uintptr_t)c % 16 == 0) && (ldc % 4 == 0) && ((uintptr_t)workspace % 16 == 0) &&
                         (!c2 || (((uintptr_t)c2 % 16 == 0) && (ldc2 % 4 == 0) && (n_split % 4 == 0)));
        if (colsum) {             // the bias gradient rides along: each slice's column sums of A behind the partial products
            if (!vec || ((uintptr_t)colsum % 16) || (m % 4)) return PLNLP_E_ALIGN;
            if (workspace_floats < (int64_t)split_k * (m * n + m)) return PLNLP_E_WORKSPACE;
            w.colsum_ws = workspace + (int64_t)split_k * m * n;
        }
        count_launch(LK_GEMM_WGRAD_WIDE);
        if (int rc = wgw::launch(w, s)) return rc;
        count_launch(LK_GEMM_SPLITK_REDUCE);
        if (vec) return wgw::reduce(w, c, ldc, e, c2, ldc2, (int)n_split, colsum, s);
        int64_t blocks = (m * n / 4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, workspace, reduce_slices,
                           g.ws_stride, c, ldc, m, (int)n, e, c2, ldc2, (int)n_split, (int64_t)0);
        return launch_status();
    }
    if (segs[0].a_colsum) return PLNLP_E_UNSUPPORTED;     // (column sums of A come out of the wide weight-gradient kernel only)
    g.mt0 = 0; g.nt0 = 0; g.gm = gm; g.gn = (int)gn; g.z0 = 0;
    auto launch_grid = [&](const GemmArgs& ga, int md, dim3 grid) -> int {
        if (math == PLNLP_GEMM_MATH_BF16X3) return x16::launch_kernels(ga, md, grid, a_trans, b_trans, s, e);
        return BK == 16 ? g16::launch_kernels(ga, md, grid, a_trans, b_trans, s, e)
                        : g32::launch_kernels(ga, md, grid, a_trans, b_trans, s, e);
    };
    auto launch = [&](const GemmArgs& ga, int md, int slices) -> int {
        dim3 grid((unsigned)(gm * gn), 1, (unsigned)slices);
        return launch_grid(ga, md, grid);
    };
    count_launch(math == PLNLP_GEMM_MATH_BF16X3 ? LK_GEMM_TILE_X3 : LK_GEMM_TILE_F32);
    if (int rc = launch(g, mode, split_k)) return rc;
    if (split_k > 1) {
        count_launch(LK_GEMM_SPLITK_REDUCE);
        const int64_t total = m * n;
        int64_t blocks = (total / 4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, workspace, reduce_slices,
                           g.ws_stride, c, ldc, m, (int)n, e, c2, ldc2, (int)n_split, (int64_t)0);
        return launch_status();
    }
    return 0;
}
#endif   // PLNLP_GEMM_BK == 32
