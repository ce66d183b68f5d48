// K4b -- the split-bf16 GEMM with stationary pre-split weights, a whole 256-row x (32 NB)-column block of the result per
// workgroup (gfx950).  Same product as gemm_x3s.hip, bit for bit (same split, the same six MFMAs per 32 x 32 x 16 block in the
// same order, K walked in steps of 16 in the same order, the same write-back) -- on K4w's division of labour (gemm_wgw.hip):
//
//   C[M,N] = EPI( sum_s A_s[M,K_s] * B_s[K_s,N] )       A_s K-contiguous (rows optionally gathered), up to 2 K-segments
//
//   * 512 threads = 8 waves, ONE persistent workgroup per CU that walks its share of the rows (struct Walk); wave w owns rows
//     32 w .. 32 w + 31 of a block and all its columns (16 NB accumulator registers).  Its A fragment never touches LDS: lane
//     (row l31, k-half h) loads the 8 consecutive k of its row straight into registers TWO K-steps ahead and splits them there,
//     one 5-instruction stage behind an MFMA at a time;
//   * the weight image of a K-step (split_b_kernel's [term 3][k-half 2][column] x 16 bytes, 21 / 24 KB) is fetched ONCE per 256
//     rows -- gemm_x3s streams it once per 128 rows through global_load_lds, and with an LDS-DMA in flight the compiler ends every
//     step in s_waitcnt vmcnt(0), so that every load had one step to land and none was hidden (profiles/
//     r05_gemm_x3s_ablation.txt).  Here every load of the K loop is a PLAIN global load into registers (the image: three 16-byte
//     units per thread, stored to LDS with ds_write_b128 two steps later); the compiler's own counted s_waitcnt vmcnt(n) is all
//     the waiting there is;
//   * the step is one basic block of 6 NB slots (slot = one MFMA + what is pinned behind it with sched_barrier); its ONE barrier
//     sits 10 MFMAs before its end, after this wave's last read of the current image buffer and its last store into the other --
//     behind those 10 MFMAs the next step's first B fragments are fetched, so the next step starts on its MFMAs at once;
//   * the pipeline runs THROUGH the block boundaries: the last three steps of a block request (and its last step splits and
//     stores) the first three steps of the next one, so that a block's write-back is the only time the matrix pipe waits --
//     and no load sits behind the write-back's stores (the memory counter is in order: a load issued after a store is not
//     back before the memory side has acknowledged the store);
//   * the rows are dealt in half blocks of 128 (waves 4 .. 7 then only stage operands: a SIMD carries one MFMA wave instead of
//     two and the block takes half the time), every CU the same share to half a block, and odd workgroups start with a half
//     block so that only half of the CUs write back at any time (struct Walk).
// Measured (profiles/r06_gemm_x3b.txt): at 224-column tiles (h = 200) 21 % faster than the 128 x 128 tile kernel that used to run
// K = 192, equal to gemm_x3s at K = 200; at 256-column tiles equal to gemm_x3s to a few per cent either way -- both hold the matrix
// pipe 67 % busy at the 1.85 GHz the power limit leaves -- so those stay on gemm_x3s unless plnlp_gemm_block_tuning says otherwise.
#include "gemm_x3s.hip.h"
#include <utility>

namespace plnlp {
namespace x3b {

using x3s::Args;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 512;                 // threads
constexpr int BR = 256;                 // rows of a full block
constexpr int CS = 64 + 4;              // the write-back's LDS row stride (floats)

template <int NB>
struct Geo {
    static constexpr int WN = 32 * NB;
    static constexpr int STAGE_UNITS = 6 * WN;                 // 16-byte units of one K-step of the image
    static constexpr int STAGE_BYTES = STAGE_UNITS * 16;
    static constexpr int PER = (STAGE_UNITS + NT - 1) / NT;    // units a thread moves per step (3)
    static constexpr int LAST_LIVE = STAGE_UNITS - (PER - 1) * NT;     // threads that own a third unit (512 / 320)
    static constexpr int C_BYTES = 8 * 32 * CS * 4;            // the write-back's region, behind the two image buffers
    static_assert(PER == 3 && LAST_LIVE % 64 == 0, "image units per thread");
};

__device__ __forceinline__ unsigned pk(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }

// x = hi + mid + lo of one lane's 8 consecutive k (gemm_x3s.hip::split3's bits) in nine stages that ride behind MFMAs:
// 0-3 the hi term of k-pair q and its residual, 4-7 the mid term, 8 the lo terms.  The terms are assembled in place in `t`.
struct SplitA {
    f32x2 r[4];
    template <int I>
    __device__ __forceinline__ void stage(const f32x4& x0, const f32x4& x1, u32x4 (&t)[3]) {
#pragma clang fp contract(off)
        if constexpr (I < 4) {
            f32x2 v;
            if constexpr (I == 0) { v.x = x0.x; v.y = x0.y; }
            else if constexpr (I == 1) { v.x = x0.z; v.y = x0.w; }
            else if constexpr (I == 2) { v.x = x1.x; v.y = x1.y; }
            else { v.x = x1.z; v.y = x1.w; }
            const unsigned hi = pk(v);
            t[0][I] = hi;
            r[I].x = v.x - __uint_as_float(hi << 16);
            r[I].y = v.y - __uint_as_float(hi & 0xffff0000u);
        } else if constexpr (I < 8) {
            constexpr int Q = I - 4;
            const unsigned mid = pk(r[Q]);
            t[1][Q] = mid;
            r[Q].x -= __uint_as_float(mid << 16);
            r[Q].y -= __uint_as_float(mid & 0xffff0000u);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[2][q] = pk(r[q]);
        }
    }
};

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

// this thread's units of image step `ks`: t, t + 512 and t + 1024 -- where the image has no such unit (NB = 7: 1 344 units) the
// thread moves its second unit once more (same bytes to the same place: no branch in the step)
template <int NB>
__device__ __forceinline__ int third_unit(int t) {
    typedef Geo<NB> G;
    if constexpr (G::LAST_LIVE == NT) return t + 2 * NT;
    else return t < G::LAST_LIVE ? t + 2 * NT : t + NT;
}
template <int NB>
__device__ __forceinline__ void load_image(const u32x4* __restrict__ img, int ks, int t, u32x4 (&ib)[3]) {
    const u32x4* src = img + (int64_t)ks * Geo<NB>::STAGE_UNITS;
    ib[0] = src[t];
    ib[1] = src[t + NT];
    ib[2] = src[third_unit<NB>(t)];
}
template <int NB, int I>
__device__ __forceinline__ void store_image(u32x4* __restrict__ buf, int t, const u32x4 (&ib)[3]) {
    if constexpr (I < 2) buf[t + I * NT] = ib[I];
    else buf[third_unit<NB>(t)] = ib[2];
}

constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};    // hi lo, lo hi, mid mid, hi mid, mid hi, hi hi

constexpr int stage_slot(int i, int slots) { return 6 + (i * (slots - 10)) / 9; }      // the slot behind which stage i of the A split rides
constexpr int split_stage_at(int q, int slots) {
    for (int i = 0; i < 9; ++i)
        if (stage_slot(i, slots) == q) return i;
    return -1;
}

// where a step's look-ahead loads come from: K-step `ks` of a block (its two A row pointers of this lane, its n-tile's image)
struct Target { const float* p0; const float* p1; const u32x4* img; int ks; };

// One pipelined K-step of an MFMA wave.  D = the step's parity: buffer `cbuf` holds this step's image, at[D] this step's A terms;
// ra[D ^ 1] / ib[D ^ 1] hold the raw A rows / image units of step + 1 (requested two steps ago), ra[D] / ib[D] those of step + 2
// (requested one step ago).  This step splits ra[D ^ 1] into at[D ^ 1], stores ib[D ^ 1] into `nbuf` and refills both sets with
// step + 3 (`tg`: the last three steps of a block request the first three of the NEXT one -- the pipeline runs through).
// P: which of the two B fragment sets block 0 uses (alternates when NB is odd).
// Hazards: every read of cbuf lies before this step's barrier, every write of nbuf too; nbuf was last read before the PREVIOUS
// step's barrier (as that step's cbuf) and is read again only after this one.
template <int NB, int D, int P, bool RAGGED>
__device__ __forceinline__ void step_mfma(f32x16 (&acc)[NB], const u32x4* __restrict__ cbuf, u32x4* __restrict__ nbuf, const Args& g,
                                          const Target& tg, u32x4 (&at)[2][3], f32x4 (&ra)[2][2], u32x4 (&ib)[2][3],
                                          bf16x8 (&B)[2][3], SplitA& sp, int t, int l31, int h) {
    typedef Geo<NB> G;
    const u32x4* bt = cbuf + h * G::WN + l31;
    constexpr int SLOTS = 6 * NB, BS = SLOTS - 10;
    static_assert(stage_slot(8, SLOTS) < SLOTS && stage_slot(0, SLOTS) > 5, "slot plan");
    __builtin_amdgcn_sched_barrier(0);
    static_for<SLOTS>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int j = q / 6, u = q % 6;
        if constexpr (q == BS) __syncthreads();
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, at[D][TA[u]]), B[(P + j) & 1][TB[u]], acc[j], 0, 0, 0);
        if constexpr (u == 0 && j + 1 < NB) {
#pragma unroll
            for (int s = 0; s < 3; ++s) B[(P + j + 1) & 1][s] = __builtin_bit_cast(bf16x8, bt[s * 2 * G::WN + (j + 1) * 32]);
        }
        if constexpr (q == BS + 4) {                 // the next step's first B fragments, into the set block NB - 2 has just left
#pragma unroll
            for (int s = 0; s < 3; ++s) B[(P + NB) & 1][s] = __builtin_bit_cast(bf16x8, nbuf[s * 2 * G::WN + h * G::WN + l31]);
        }
        if constexpr (q >= 2 && q <= 4) store_image<NB, q - 2>(nbuf, t, ib[D ^ 1]);       // each store in a slot of its own
        if constexpr (q == 5) load_image<NB>(tg.img, tg.ks, t, ib[D ^ 1]);
        constexpr int st = split_stage_at(q, SLOTS);
        if constexpr (st >= 0) sp.template stage<st>(ra[D ^ 1][0], ra[D ^ 1][1], at[D ^ 1]);
        if constexpr (q == stage_slot(3, SLOTS) + 1) load_a<RAGGED>(g, tg.p0, tg.p1, tg.ks, h, ra[D ^ 1][0], ra[D ^ 1][1]);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// the same step of a wave without rows in this block (waves 4 .. 7 of a half block): everything but the MFMAs -- its share of the
// image, and the A terms too (the NEXT block may have rows for it, and its first steps are prepared during this block's last)
template <int NB, int D, bool RAGGED>
__device__ __forceinline__ void step_idle(u32x4* __restrict__ nbuf, const Args& g, const Target& tg, u32x4 (&at)[2][3],
                                          f32x4 (&ra)[2][2], u32x4 (&ib)[2][3], SplitA& sp, int t, int h) {
    store_image<NB, 0>(nbuf, t, ib[D ^ 1]);
    store_image<NB, 1>(nbuf, t, ib[D ^ 1]);
    store_image<NB, 2>(nbuf, t, ib[D ^ 1]);
    load_image<NB>(tg.img, tg.ks, t, ib[D ^ 1]);
    static_for<9>([&](auto ic) { sp.template stage<decltype(ic)::value>(ra[D ^ 1][0], ra[D ^ 1][1], at[D ^ 1]); });
    load_a<RAGGED>(g, tg.p0, tg.p1, tg.ks, h, ra[D ^ 1][0], ra[D ^ 1][1]);
    __syncthreads();
}

// ---- the write-back of one block: 64 columns at a time through this wave's private LDS region (the MFMA C/D map -- col =
// lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5) -- would leave as 4-byte scattered stores), then 16-byte row stores with
// the epilogue on float4: gemm_x3s.hip's write-back, same bits.  No global load sits between its stores: the memory counter is in order, a load behind a store waits until the memory side has acknowledged
// the store -- four such waits per block (the bias of each chunk) cost 38 000 cycles where the whole K loop takes 109 000.  The
// bias and ROWDOT's weights therefore come from an LDS copy made once per launch (`stash`: [bias n floats | rowdot_w n floats]), the
// dropout row map from the pad column of the LDS tile, and epilogues with row-dependent operands (gate / accumulate) are not taken
// (takes()).
// ROWDOT (PLNLP_EPI_ROWDOT): also forms, per row, the dot product of this tile's stored values with rowdot_w.
constexpr int STASH_N = 2048;               // widest result whose bias / row-dot weights the stash holds (the launcher checks)
template <int NB, bool ROWDOT>
__device__ __forceinline__ void write_back(const f32x16 (&acc)[NB], float* __restrict__ cw, const float* __restrict__ stash,
                                           const Args& g, const Epi& epi, uint32_t seed_lo, uint32_t seed_hi, int64_t row_w,
                                           int nt, int lane, int l31, int h) {
    constexpr int NCH = NB / 2 + (NB & 1);                     // 64-column chunks
    const int n0 = nt * 32 * NB;
    const int c4 = (lane & 15) * 4, sub = lane >> 4;           // this lane: 16-byte column group c4 of rows sub + 4 i
    const int64_t row0 = row_w + sub;
    const int rows_here = g.m - row_w < 32 ? (int)(g.m - row_w) : 32;
    unsigned live = 0;                                         // bit i: row sub + 4 i exists
#pragma unroll
    for (int i = 0; i < 8; ++i) live |= sub + 4 * i < rows_here ? 1u << i : 0u;
    // the dropout counter row of each of the wave's 32 rows (common.hip.h: drop_row) sits in the pad column 64 of its row of the
    // wave's LDS region for the whole write-back -- one coalesced load here instead of one load per row between the stores
    const bool mapped = (epi.flags & PLNLP_EPI_DROPOUT) && epi.drop_row;
    if (mapped && lane < 32) {
        const int64_t r = lane < rows_here ? row_w + lane : row_w;
        reinterpret_cast<int*>(cw)[lane * CS + 64] = epi.drop_row[r];
    }
    float rd[8];                                               // ROWDOT: this lane's share of rows sub + 4 i
#pragma unroll
    for (int i = 0; i < 8; ++i) rd[i] = 0.f;
    static_for<NCH>([&](auto jc) {
        constexpr int ch = decltype(jc)::value;
        constexpr int j0 = 2 * ch;
        constexpr int NJ = (j0 + 1 < NB) ? 2 : 1;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                cw[((q & 3) + 8 * (q >> 2) + 4 * h) * CS + jj * 32 + l31] = acc[j0 + jj][q];
        const int col = n0 + j0 * 32 + c4;
        const bool col_ok = c4 < NJ * 32 && col < g.n;
        const bool second = col >= g.n_split;
        float* const base = (second ? g.c2 + row0 * g.ldc2 - g.n_split : g.c + row0 * g.ldc) + col;
        const int64_t stride = 4 * (second ? g.ldc2 : g.ldc);
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), rw4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col_ok && (epi.flags & PLNLP_EPI_BIAS)) bias4 = *reinterpret_cast<const float4*>(stash + col);
        if constexpr (ROWDOT) { if (col_ok) rw4 = *reinterpret_cast<const float4*>(stash + STASH_N + col); }
        constexpr int RB = 2;                                  // rows per batch
#pragma unroll
        for (int i0 = 0; i0 < 8; i0 += RB) {
            float4 v4[RB];
            int dr[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                v4[i] = *reinterpret_cast<const float4*>(cw + ((i0 + i) * 4 + sub) * CS + c4);
                dr[i] = mapped ? reinterpret_cast<const int*>(cw)[((i0 + i) * 4 + sub) * CS + 64] : (int)(row0 + 4 * (i0 + i));
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if (col_ok && ((live >> (i0 + i)) & 1)) {
                    // common.hip.h::epi_apply4_pre's arithmetic in its order (bias, relu, dropout)
                    float4 y = v4[i];
                    if (epi.flags & PLNLP_EPI_BIAS) { y.x += bias4.x; y.y += bias4.y; y.z += bias4.z; y.w += bias4.w; }
                    if (epi.flags & PLNLP_EPI_RELU) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
                    if (epi.flags & PLNLP_EPI_DROPOUT)
                        y = dropout_apply4(y, (uint64_t)(unsigned)dr[i] * (uint64_t)g.n + (uint64_t)col, seed_lo, seed_hi, epi.thresh, epi.keep_scale);
                    *reinterpret_cast<float4*>(base + (i0 + i) * stride) = y;
                    if constexpr (ROWDOT) rd[i0 + i] = x3s::rowdot_acc(rd[i0 + i], y, rw4);
                }
            }
        }
    });
    if constexpr (ROWDOT) {
        // the 16 lanes that share a row (lane & 15 = their 16-byte column group) fold their shares in a fixed tree; the
        // tile's partial goes to row nt of rowdot_out (plnlp_rowdot_finish_f32 adds the tiles in order)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = rd[i];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if ((lane & 15) == 0 && ((live >> i) & 1)) epi.rowdot_out[(int64_t)nt * epi.rowdot_ld + row0 + 4 * i] = v;
        }
    }
}

// A workgroup's walk over its share of the rows.  The rows are dealt in half blocks of 128: workgroup w of G takes H / G of the
// H half blocks (the first H % G workgroups one more), a contiguous range, as full blocks of 256 rows plus at most two half blocks
// (waves 4 .. 7 then only help with the image; a SIMD carries one MFMA wave instead of two and the block takes half the time).
// Every CU so gets the same work to half a block -- there are no rounds of workgroups whose last one runs nearly empty -- and ODD
// workgroups take a half block FIRST (`lead`): their K loops then run half a block beside the even workgroups', so that only half
// of the CUs write their results at any time.  (With every workgroup in step the write-back is a burst of 64 MB at the HBM's
// write rate -- 22 000 cycles per block, a fifth of the K loop at K = 512 -- during which no MFMA runs.)
// Each row block is walked through its n-tiles before the next one (its A rows come out of L2 the second time).
struct Walk {
    int64_t pos, rem;             // next half block, half blocks left
    int nt, gn;
    bool lead, first;
    __device__ __forceinline__ int halves() const { return ((first && lead) || rem == 1) ? 1 : 2; }
    __device__ __forceinline__ bool done() const { return rem <= 0; }
    __device__ __forceinline__ void advance() {
        if (++nt == gn) { nt = 0; const int hv = halves(); pos += hv; rem -= hv; first = false; }
    }
};

// what one wave needs of a block: where its rows start, whether it has any, its lanes' two A row pointers
struct Block {
    int64_t row_w;
    const float* p0; const float* p1;
    const u32x4* img;
    int nt;
    bool active;
};
template <int NB>
__device__ __forceinline__ Block block_of(const Args& g, const Walk& w, int wave, int l31) {
    Block b;
    const int hv = w.halves();
    const int64_t row_b = w.pos * (BR / 2);
    b.row_w = row_b + wave * 32;
    b.active = wave < 4 * hv && b.row_w < g.m;                  // (wave-uniform)
    b.nt = w.nt;
    b.img = reinterpret_cast<const u32x4*>(g.image) + (int64_t)w.nt * g.ks_total * Geo<NB>::STAGE_UNITS;
    // a wave without rows still splits "its" A rows (the step is the same for every wave): all lanes on the block's first row;
    // rows past the end feed result rows the store discards
    int64_t row = b.active ? b.row_w + l31 : row_b;
    row = row < g.m ? row : g.m - 1;
    int64_t r0 = row, r1 = row;
    if (g.a_index[0]) r0 = g.a_index[0][row];                  // rows of A gathered in place (a conv at the touched rows)
    if (g.nseg > 1 && g.a_index[1]) r1 = g.a_index[1][row];
    b.p0 = g.a[0] + r0 * g.lda[0];
    b.p1 = g.nseg > 1 ? g.a[1] + r1 * g.lda[1] : b.p0;
    return b;
}

template <typename T>
__device__ __forceinline__ void swap_regs(T& a, T& b) { const T c = a; a = b; b = c; }

template <int NB, bool RAGGED, bool ROWDOT>
__global__ __launch_bounds__(NT, 1) void gemm_x3b_kernel(Args g, Epi epi) {
    typedef Geo<NB> G;
    constexpr int WN = G::WN;
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];        // two image buffers, then the write-back's region

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    float* cw = reinterpret_cast<float*>(lds + 2 * G::STAGE_UNITS) + wave * 32 * CS;
    float* stash = reinterpret_cast<float*>(lds + 2 * G::STAGE_UNITS) + 8 * 32 * CS;    // [bias | rowdot_w], see write_back
    u32x4* b0 = lds;                       // the image buffer of the even steps ...
    u32x4* b1 = lds + G::STAGE_UNITS;      // ... and of the odd ones (they trade places after a block with an odd number of steps)
    Walk cur;
    {
        const int64_t halves = (g.m + BR / 2 - 1) / (BR / 2), G_ = gridDim.x, w = blockIdx.x;
        const int64_t base = halves / G_, extra = halves % G_;
        cur.pos = w * base + (w < extra ? w : extra);
        cur.rem = base + (w < extra ? 1 : 0);
        cur.nt = 0; cur.gn = g.gn;
        cur.first = true;
        cur.lead = g.lead_half && (w & 1) && cur.rem >= 2;
    }
    if (cur.done()) return;
    const int KS = g.ks_total;            // >= 3 (the launcher)
    uint32_t seed_lo, seed_hi;
    epi_seed(epi, seed_lo, seed_hi);

    f32x16 acc[NB];
    u32x4 at[2][3], ib[2][3];
    f32x4 ra[2][2];
    bf16x8 B[2][3];
    SplitA sp;
    Block blk = block_of<NB>(g, cur, wave, l31);
    for (int c = 4 * t; c < g.n; c += 4 * NT) {      // (n % 4 == 0, n <= STASH_N; visible to every wave after the prologue's barrier)
        if (epi.flags & PLNLP_EPI_BIAS) *reinterpret_cast<float4*>(stash + c) = *reinterpret_cast<const float4*>(epi.bias + c);
        if constexpr (ROWDOT) *reinterpret_cast<float4*>(stash + STASH_N + c) = *reinterpret_cast<const float4*>(epi.rowdot_w + c);
    }
    // ---- the walk's prologue: A of step 0 split, image of step 0 resident, steps 1 and 2 of both requested.  Never again: the
    // last three steps of every block request (and its last step splits / stores) the first three of the next one.
    load_a<RAGGED>(g, blk.p0, blk.p1, 0, h, ra[0][0], ra[0][1]);
    load_image<NB>(blk.img, 0, t, ib[0]);
    load_a<RAGGED>(g, blk.p0, blk.p1, 1, h, ra[1][0], ra[1][1]);
    load_image<NB>(blk.img, 1, t, ib[1]);
    static_for<9>([&](auto ic) { sp.template stage<decltype(ic)::value>(ra[0][0], ra[0][1], at[0]); });
    store_image<NB, 0>(b0, t, ib[0]);
    store_image<NB, 1>(b0, t, ib[0]);
    store_image<NB, 2>(b0, t, ib[0]);
    load_a<RAGGED>(g, blk.p0, blk.p1, 2, h, ra[0][0], ra[0][1]);
    load_image<NB>(blk.img, 2, t, ib[0]);
    __syncthreads();
    while (true) {
        Walk nxt = cur;
        nxt.advance();
        const bool more = !nxt.done();
        const Block nb_ = more ? block_of<NB>(g, nxt, wave, l31) : blk;
        // step `tks` counted from this block's first: past its end, the next block's (no next block: the last step once more --
        // what is staged then is never multiplied)
        auto target = [&](int tks) {
            const bool wrap = tks >= KS;
            Target tg;
            tg.p0 = wrap ? nb_.p0 : blk.p0;
            tg.p1 = wrap ? nb_.p1 : blk.p1;
            tg.img = wrap ? nb_.img : blk.img;
            tg.ks = wrap ? (more ? tks - KS : KS - 1) : tks;
            return tg;
        };
        if (blk.active) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) B[0][s] = __builtin_bit_cast(bf16x8, b0[s * 2 * WN + h * WN + l31]);
            int ks = 0;
            for (; ks + 2 <= KS; ks += 2) {
                step_mfma<NB, 0, 0, RAGGED>(acc, b0, b1, g, target(ks + 3), at, ra, ib, B, sp, t, l31, h);
                step_mfma<NB, 1, NB & 1, RAGGED>(acc, b1, b0, g, target(ks + 4), at, ra, ib, B, sp, t, l31, h);
            }
            if (ks < KS) step_mfma<NB, 0, 0, RAGGED>(acc, b0, b1, g, target(ks + 3), at, ra, ib, B, sp, t, l31, h);
        } else {
            int ks = 0;
            for (; ks + 2 <= KS; ks += 2) {
                step_idle<NB, 0, RAGGED>(b1, g, target(ks + 3), at, ra, ib, sp, t, h);
                step_idle<NB, 1, RAGGED>(b0, g, target(ks + 4), at, ra, ib, sp, t, h);
            }
            if (ks < KS) step_idle<NB, 0, RAGGED>(b1, g, target(ks + 3), at, ra, ib, sp, t, h);
        }
        if (KS & 1) {             // an odd number of steps: the two register sets and the two buffers have traded roles
#pragma unroll
            for (int s = 0; s < 3; ++s) { at[0][s] = at[1][s]; swap_regs(ib[0][s], ib[1][s]); }
            swap_regs(ra[0][0], ra[1][0]);
            swap_regs(ra[0][1], ra[1][1]);
            swap_regs(b0, b1);
        }
        if (blk.active) write_back<NB, ROWDOT>(acc, cw, stash, g, epi, seed_lo, seed_hi, blk.row_w, blk.nt, lane, l31, h);
        if (!more) break;
        cur = nxt;
        blk = nb_;
    }
}

// measurement knob (plnlp_gemm_block_tuning), bits: 1 = never (gemm_x3s / the tile kernels everywhere), 2 = no leading half
// blocks, 4 = also the 256-column tiles (measured: same time as gemm_x3s to -2 .. +5 %, profiles/r06_gemm_x3b.txt -- both run at the
// power limit there; the default is the 224-column tile, where this kernel is 21 % faster than what ran before)
static int g_mode = 0;
void set_mode(int mode) { g_mode = mode; }

// Could a launch of `m` rows at tile width nb (x3s::pick_nb) take this kernel?  Enough rows to give every CU a half block.
bool applies(int64_t m, int nb) {
    if ((g_mode & 1) || m < 32768) return false;
    return nb == 7 || (nb == 8 && (g_mode & 4));
}
// ... and this epilogue?  Bias / gate rows must load 16 bytes at a time (they do wherever the operands are aligned), no indexed
// addend (no GEMM of the path carries one), the row-dot head on 256-wide tiles with whole K-steps only.
bool takes(int64_t m, int64_t n, int nb, bool ragged, int k_steps, const Epi& e) {
    if (!applies(m, nb) || k_steps < 3) return false;
    if ((e.flags & (PLNLP_EPI_BIAS | PLNLP_EPI_ROWDOT)) && n > STASH_N) return false;     // (their LDS copy, see write_back)
    if (e.flags && (!e.vec4 || (e.flags & PLNLP_EPI_ADDEND))) return false;
    // no row-dependent operands (gate / accumulate): their loads sit between the write-back's stores and wait for the stores'
    // acknowledgements (the memory counter is in order) -- with ONE workgroup per CU nothing covers that; gemm_x3s, two per CU, does
    // (measured on citation2's gated data gradient: 2.94 ms here against 2.60 there)
    if (e.flags & (PLNLP_EPI_GATE | PLNLP_EPI_ACCUM)) return false;
    if ((e.flags & PLNLP_EPI_ROWDOT) && (nb != 8 || ragged)) return false;
    return true;
}

static int cu_count() {           // (per device; a process may hold several)
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

template <int NB, bool RAGGED, bool ROWDOT>
static int launch_as(const Args& a, const Epi& e, unsigned grid, hipStream_t s) {
    auto kernel = gemm_x3b_kernel<NB, RAGGED, ROWDOT>;
    constexpr int LDS = 2 * Geo<NB>::STAGE_BYTES + Geo<NB>::C_BYTES + 2 * STASH_N * 4;
    // (once per device and instantiation: the attribute belongs to the device's copy of the function)
    static bool armed[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return PLNLP_E_UNSUPPORTED;
    if (dev < 0 || dev >= 64 || !armed[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return PLNLP_E_UNSUPPORTED;
        if (dev >= 0 && dev < 64) armed[dev] = true;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(NT), LDS, s, a, e);
    return launch_status();
}

template <int NB>
static int launch_nb(const Args& a, const Epi& e, bool ragged, unsigned grid, hipStream_t s) {
    if (e.flags & PLNLP_EPI_ROWDOT) {
        if constexpr (NB == 8) {
            if (e.rowdot_ld < a.m) return PLNLP_E_SHAPE;
            if (!ragged) return launch_as<NB, false, true>(a, e, grid, s);
        }
        return PLNLP_E_UNSUPPORTED;           // (takes() keeps such launches away)
    }
    return ragged ? launch_as<NB, true, false>(a, e, grid, s) : launch_as<NB, false, false>(a, e, grid, s);
}

// the main kernel of a launch whose image x3s::launch has just queued (a.image, a.gn, a.m .. filled in): one workgroup per CU
int launch(const Args& a_in, int nb, bool ragged, const Epi& e, hipStream_t s) {
    Args a = a_in;
    const int64_t halves = (a.m + BR / 2 - 1) / (BR / 2);
    const int cus = cu_count();
    const unsigned grid = (unsigned)(halves < cus ? halves : cus);
    a.lead_half = (g_mode & 2) ? 0 : 1;
    a.row_lo = 0;
    switch (nb) {
        case 8: return launch_nb<8>(a, e, ragged, grid, s);
        case 7: return launch_nb<7>(a, e, ragged, grid, s);
    }
    return PLNLP_E_UNSUPPORTED;
}

}  // namespace x3b
}  // namespace plnlp
