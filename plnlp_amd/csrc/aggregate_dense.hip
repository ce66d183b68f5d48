// K1d -- the neighbour aggregation of a DENSE graph as a product on the matrix cores (gfx950).
//
//   out[r, :] = EPI( row_scale[r] * sum_k counts[r, k] * (src_scale[k] * x[k, :]) )
//
// torch_sparse.matmul under SAGEConv (plnlp/layer.py:30-36) on a graph like ogbl-ddi: 4 267 nodes, 2.1 M entries -- 11.7 % of
// all node pairs are edges, a row has ~500 entries, EVERY row is a "long row" of the CSR kernels (csr_aggregate.hip), whose gather
// then moves 4.4 GB out of L2 per launch to add up what is, as a matrix, 4 267 x 4 267 small integers times an 8.7 MB operand.
// Here the adjacency is a matrix of COUNTS in bf16 (exact up to 256 parallel edges; built once per static graph by the host,
// 36 MB) and the feature matrix is split into three bf16 terms as in the GEMMs (x = hi + mid + lo, round-to-nearest each, the
// residuals exact in f32: what is dropped is below 2^-24 |x|).  One operand is exact, so a 32 x 32 x 16 block is THREE MFMAs
// (counts x lo, x mid, x hi; small terms first) with f32 accumulation -- the CSR kernels' f32 sums in another order (k-block by
// k-block instead of entry by entry); measured against float64 the rms error is 0.75 x the CSR kernels' (1.14e-7 against 1.52e-7 of the
// rms value on the ddi graph).  A fourth term was built and measured: same error (the f32 accumulation is what is left), +18 % time.  north_star keeps the matrix cores for the dense linears "as evidenced by rocprof": the evidence for this exception is
// profiles/r06_ddi_dense_agg.txt (same box, the step's four aggregation launches).
//
// Division of labour: 256 threads = 4 waves, a workgroup owns 128 rows x 128 columns of one K slice; wave w owns rows 32 w ..
// 32 w + 31.  Its A fragment is a plain 16-byte load (lane (row, k-half) reads the 8 consecutive bf16 counts of its row), four
// K-steps ahead; the K-step's image of x (12 KB) goes global -> registers -> LDS two steps ahead (plain loads only: counted waits).
// The problem is small for the chip (56 GFLOP executed: 27 us of matrix-pipe time), so K is cut into slices to put ~2 workgroups
// on every CU; the slices' raw partials go to the caller's scratch and dense_reduce_kernel adds them in order, applies the row
// scale (the mean's 1 / deg) and the aggregation's epilogue (bias / relu / dropout / accumulate / addend / gate, or the table's Adam step).
#include "common.hip.h"
#include <utility>

namespace plnlp {
namespace aggd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NB = 4, WN = 32 * NB;                 // column blocks per wave / columns per workgroup
constexpr int TERMS = 3;
constexpr int STAGE_UNITS = 2 * TERMS * WN;         // 16-byte units of one K-step of the image ([term][k-half 2][column])
constexpr int PER = STAGE_UNITS / 256;              // units a thread moves per step (3)
constexpr int CS = 64 + 4;                          // the write-back's LDS row stride (floats)
constexpr int LDS_BYTES = 2 * STAGE_UNITS * 16 > 4 * 32 * CS * 4 ? 2 * STAGE_UNITS * 16 : 4 * 32 * CS * 4;

struct Args {
    const uint16_t* counts; int64_t ld_counts;      // bf16 [n_rows, ld_counts], ld_counts a multiple of 16 >= n_src, zero beyond n_src
    const void* image;                              // x split into its three bf16 terms (split_x_kernel)
    float* ws;                                      // [slices][n_rows][feat] raw partials
    int64_t n_rows; int feat;
    int ks_total, slices, gn;                       // K-steps of 16 in all, K slices, column tiles
    int64_t panels;                                 // row panels of 128
};

// the image of x: unit ((nt * KS + ks) * 2 TERMS + term * 2 + h) * WN + c  holds term `term` of
//   scale[k] * x[k = 16 ks + 8 h + 0..7][column nt * WN + c],   zeros past n_src / feat
struct SplitArgs { const float* x; int64_t ldx; const float* k_scale; int64_t n_src; int feat, ks_total, gn; void* image; };
__device__ __forceinline__ unsigned pk2(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__global__ __launch_bounds__(256) void split_x_kernel(SplitArgs g) {
#pragma clang fp contract(off)
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)g.gn * g.ks_total * 2 * WN) return;
    const int c = (int)(id % WN), h = (int)((id / WN) & 1);
    const int64_t st = id / (2 * WN);
    const int ks = (int)(st % g.ks_total), nt = (int)(st / g.ks_total);
    const int col = nt * WN + c;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t k = 16 * (int64_t)ks + 8 * h + e;
        const bool ok = col < g.feat && k < g.n_src;
        v[e] = ok ? g.x[(ok ? k : 0) * g.ldx + (ok ? col : 0)] : 0.f;
        if (g.k_scale) v[e] *= g.k_scale[ok ? k : 0];
    }
    u32x4* u = reinterpret_cast<u32x4*>(g.image) + (st * 2 * TERMS + h) * WN + c;
#pragma unroll
    for (int term = 0; term < TERMS; ++term) {
        u32x4 t;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned p = pk2(v[2 * q], v[2 * q + 1]);
            t[q] = p;
            v[2 * q] -= __uint_as_float(p << 16);                  // (exact: the residual of a round-to-nearest bf16)
            v[2 * q + 1] -= __uint_as_float(p & 0xffff0000u);
        }
        u[term * 2 * WN] = t;
    }
}

__global__ __launch_bounds__(256, 2) void dense_agg_kernel(Args g) {
    __shared__ __attribute__((aligned(16))) char lds_raw[LDS_BYTES];
    u32x4* lds = reinterpret_cast<u32x4*>(lds_raw);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    // workgroup -> (K slice, row panel, column tile): the tiles of one panel and slice back to back (they share the panel's counts)
    int64_t id = blockIdx.x;
    const int nt = (int)(id % g.gn); id /= g.gn;
    const int64_t mt = id % g.panels;
    const int z = (int)(id / g.panels);
    const int per = (g.ks_total + g.slices - 1) / g.slices;
    const int k0 = z * per, k1 = k0 + per < g.ks_total ? k0 + per : g.ks_total;
    const int64_t row_w = mt * 128 + wave * 32;
    int64_t row = row_w + l31;
    row = row < g.n_rows ? row : g.n_rows - 1;                      // clamped rows feed result rows the store discards
    const u32x4* arow = reinterpret_cast<const u32x4*>(g.counts + row * g.ld_counts) + h;      // unit = 8 bf16; a K-step = 2 units
    const u32x4* img = reinterpret_cast<const u32x4*>(g.image) + (int64_t)nt * g.ks_total * STAGE_UNITS;

    // Two levels of accumulation: the MFMAs add into `acc` for four K-steps, then `acc` is added to `tot` by plain f32 adds
    // (round-to-nearest-even) and cleared.  The matrix pipe's own f32 accumulation carries a small one-sided rounding bias
    // and its error grows with the length of the chain: accumulated in one go over a slice's 67 K-steps the full-size ddi step's
    // scorer-bias gradient -- a sum of 786 432 nearly cancelling terms -- sat 7e-4 of its scale from float64 (tests/test_hip_round4.py:
    // bound 1e-4; the CSR kernels: inside); with the short chains the aggregation's rms error is below the CSR kernels' and the
    // step is inside the bound.  Costs 64 registers (two workgroups per CU instead of three: 0.085 -> 0.093 ms per launch).
    f32x16 acc[NB], tot[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = tot[j][q] = 0.f;
    if (k0 < k1) {
        auto at_most = [&](int ks) { return ks < k1 ? ks : k1 - 1; };
        u32x4 a[4], ib[2][PER];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = arow[2 * (int64_t)at_most(k0 + i)];
#pragma unroll
        for (int i = 0; i < PER; ++i) { ib[0][i] = img[(int64_t)k0 * STAGE_UNITS + t + 256 * i]; ib[1][i] = img[(int64_t)at_most(k0 + 1) * STAGE_UNITS + t + 256 * i]; }
#pragma unroll
        for (int i = 0; i < PER; ++i) lds[t + 256 * i] = ib[0][i];
#pragma unroll
        for (int i = 0; i < PER; ++i) ib[0][i] = img[(int64_t)at_most(k0 + 2) * STAGE_UNITS + t + 256 * i];
        __syncthreads();
        // step ks: buffer (ks - k0) & 1 holds its image, a[(ks - k0) & 3] its counts; ib[(ks - k0 + 1) & 1] holds the image of
        // ks + 1 (stored into the other buffer now), ib[(ks - k0) & 1] that of ks + 2 (in flight)
        auto step = [&](int ks, auto pc) {
            constexpr int Pq = decltype(pc)::value;               // (ks - k0) & 3
            constexpr int D = Pq & 1;
            const u32x4* bt = lds + D * STAGE_UNITS + h * WN + l31;
            u32x4* nbuf = lds + (D ^ 1) * STAGE_UNITS;
            const bf16x8 av = __builtin_bit_cast(bf16x8, a[Pq]);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                bf16x8 b[TERMS];
#pragma unroll
                for (int s = 0; s < TERMS; ++s) b[s] = __builtin_bit_cast(bf16x8, bt[s * 2 * WN + j * 32]);
#pragma unroll
                for (int s = TERMS - 1; s >= 0; --s)             // counts x the smallest term first
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b[s], acc[j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) nbuf[t + 256 * i] = ib[D ^ 1][i];
#pragma unroll
            for (int i = 0; i < PER; ++i) ib[D ^ 1][i] = img[(int64_t)at_most(ks + 3) * STAGE_UNITS + t + 256 * i];
            a[Pq] = arow[2 * (int64_t)at_most(ks + 4)];
            if constexpr (Pq == 3) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    tot[j] += acc[j];
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
                }
            }
            __syncthreads();
        };
        int ks = k0;
        for (; ks + 4 <= k1; ks += 4) {
            step(ks, std::integral_constant<int, 0>{});
            step(ks + 1, std::integral_constant<int, 1>{});
            step(ks + 2, std::integral_constant<int, 2>{});
            step(ks + 3, std::integral_constant<int, 3>{});
        }
        if (ks < k1) step(ks++, std::integral_constant<int, 0>{});
        if (ks < k1) step(ks++, std::integral_constant<int, 1>{});
        if (ks < k1) step(ks++, std::integral_constant<int, 2>{});
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] += tot[j];
    // raw partial of this slice, 64 columns at a time through the wave's private LDS region (the MFMA C/D map -- column = lane & 31,
    // row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5) -- would leave as 4-byte scattered stores)
    float* cw = reinterpret_cast<float*>(lds_raw) + wave * 32 * CS;
    float* out = g.ws + (int64_t)z * g.n_rows * g.feat;
    const int c4 = (lane & 15) * 4, sub = lane >> 4;
#pragma unroll
    for (int j0 = 0; j0 < NB; j0 += 2) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                cw[((q & 3) + 8 * (q >> 2) + 4 * h) * CS + jj * 32 + l31] = acc[j0 + jj][q];
        const int col = nt * WN + j0 * 32 + c4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t orow = row_w + 4 * i + sub;
            const f32x4 v = *reinterpret_cast<const f32x4*>(cw + (4 * i + sub) * CS + c4);
            if (orow < g.n_rows && col < g.feat) *reinterpret_cast<f32x4*>(out + orow * g.feat + col) = v;
        }
    }
}

// out[r, c .. c+3] = EPI( row_scale[r] * sum_z ws[z][r][c .. c+3] ), z ascending; the aggregation's epilogue as csr_aggregate.hip applies it
__global__ __launch_bounds__(256) void dense_reduce_kernel(const float* __restrict__ ws, int slices, int64_t n_rows, int feat,
                                                           const float* __restrict__ row_scale, float* __restrict__ out, int64_t ldo,
                                                           Epi epi) {
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x, total4 = (n_rows * feat) >> 2;
    if (i4 >= total4) return;
    const int64_t i = i4 * 4, r = i / feat;
    const int f = (int)(i - r * feat);
    f32x4 acc = *reinterpret_cast<const f32x4*>(ws + i);
    for (int z = 1; z < slices; ++z) acc += *reinterpret_cast<const f32x4*>(ws + (int64_t)z * n_rows * feat + i);
    if (row_scale) acc *= row_scale[r];
    float* orow = out + r * ldo;
    const float4 y = epi_apply4(epi, make_float4(acc.x, acc.y, acc.z, acc.w), r, f, feat, orow);
    if (epi.flags & PLNLP_EPI_ADAM) { epi_adam4(epi, y, orow + f, r * ldo + f); return; }
    *reinterpret_cast<float4*>(orow + f) = y;
}

}  // namespace aggd
}  // namespace plnlp

using namespace plnlp;

// K slices of a launch.  The chip holds 512 workgroups of this kernel at once (two per CU: the two-level accumulation's registers), a
// launch is ROUNDS of that many, and a workgroup's time is its K-steps plus a fixed part (prologue, the write of its partial tile:
// ~10 K-steps' worth); the reduce reads every slice (~3 K-steps' worth each).  ddi (34 x 4 tiles, 267 K-steps): four slices -- the
// first rule, "about two workgroups per CU" -- are 544 workgroups = one full round and a second for 32 of them (77 us); three are
// one round of 408.  At least 8 K-steps per slice.
static int g_force_slices = 0;       // measurement knob (plnlp_dense_aggregate_tuning)
static int dense_slices(int64_t n_rows, int64_t n_src, int64_t feat) {
    const int64_t tiles = ((n_rows + 127) / 128) * ((feat + aggd::WN - 1) / aggd::WN), ks = (n_src + 15) / 16;
    int64_t smax = ks / 8;
    smax = smax < 1 ? 1 : (smax > 64 ? 64 : smax);
    if (g_force_slices > 0) return (int)(g_force_slices < smax ? g_force_slices : smax);
    int64_t best = 1, best_cost = -1;
    for (int64_t s = 1; s <= smax; ++s) {
        const int64_t rounds = (tiles * s + 511) / 512;
        const int64_t cost = rounds * ((ks + s - 1) / s + 10) + 3 * s;
        if (best_cost < 0 || cost < best_cost) { best = s; best_cost = cost; }
    }
    return (int)best;
}

static int64_t dense_image_bytes(int64_t n_src, int64_t feat) {
    return (feat + aggd::WN - 1) / aggd::WN * ((n_src + 15) / 16) * aggd::STAGE_UNITS * 16;
}

extern "C" void plnlp_dense_aggregate_tuning(int slices) { g_force_slices = slices; }

extern "C" int64_t plnlp_dense_aggregate_scratch_bytes(int64_t n_rows, int64_t n_src, int64_t feat) {
    if (n_rows <= 0 || n_src <= 0 || feat <= 0) return 0;
    return dense_image_bytes(n_src, feat) + (int64_t)dense_slices(n_rows, n_src, feat) * n_rows * feat * 4;
}

extern "C" int plnlp_dense_aggregate_f32(const void* counts, int64_t ld_counts, const float* src_scale, const float* row_scale,
                                         const float* x, int64_t ldx, float* out, int64_t ldo, int64_t n_rows, int64_t n_src,
                                         int64_t feat, const plnlp_epilogue* epi, void* scratch, int64_t scratch_bytes, void* stream) {
    if (n_rows < 0 || n_src <= 0 || feat <= 0 || feat > 0x7FFFFFF0) return PLNLP_E_SHAPE;
    if (n_rows == 0) return 0;
    if (!counts || !x || !out || !scratch) return PLNLP_E_NULL;
    if (ld_counts < (n_src + 15) / 16 * 16 || ld_counts % 16 || feat % 4 || ldx < feat || ldo < feat || ldx % 4 || ldo % 4) return PLNLP_E_SHAPE;
    if (((uintptr_t)counts % 16) || ((uintptr_t)x % 16) || ((uintptr_t)out % 16) || ((uintptr_t)scratch % 16)) return PLNLP_E_ALIGN;
    if (scratch_bytes < plnlp_dense_aggregate_scratch_bytes(n_rows, n_src, feat)) return PLNLP_E_WORKSPACE;
    Epi e;
    if (int rc = make_epi(epi, &e, /*allow_adam=*/true)) return rc;
    if (e.flags && !e.vec4) return PLNLP_E_ALIGN;
    if ((e.flags & PLNLP_EPI_ADAM) && ((uintptr_t)e.adam_m % 16 || (uintptr_t)e.adam_v % 16)) return PLNLP_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    // x -> its three bf16 terms, in the K-step image of the main loop (zeros past n_src and past feat)
    aggd::SplitArgs sp{};
    sp.x = x; sp.ldx = ldx; sp.k_scale = src_scale; sp.n_src = n_src; sp.feat = (int)feat;
    sp.ks_total = (int)((n_src + 15) / 16); sp.gn = (int)((feat + aggd::WN - 1) / aggd::WN); sp.image = scratch;
    const int64_t units = (int64_t)sp.gn * sp.ks_total * 2 * aggd::WN;
    hipLaunchKernelGGL(aggd::split_x_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, sp);
    if (int rc = launch_status()) return rc;
    aggd::Args g{};
    g.counts = reinterpret_cast<const uint16_t*>(counts); g.ld_counts = ld_counts; g.image = scratch;
    g.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + dense_image_bytes(n_src, feat));
    g.n_rows = n_rows; g.feat = (int)feat; g.ks_total = sp.ks_total; g.slices = dense_slices(n_rows, n_src, feat);
    g.gn = sp.gn; g.panels = (n_rows + 127) / 128;
    const int64_t blocks = g.panels * g.gn * g.slices;
    if (blocks > 0x7FFFFFFF) return PLNLP_E_SHAPE;
    count_launch(LK_AGG_DENSE);
    hipLaunchKernelGGL(aggd::dense_agg_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g);
    if (int rc = launch_status()) return rc;
    const int64_t total4 = (n_rows * feat) >> 2;
    hipLaunchKernelGGL(aggd::dense_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, g.ws, g.slices, n_rows,
                       (int)feat, row_scale, out, ldo, e);
    return launch_status();
}
