// K4w -- the split-bf16 weight gradient with a whole W x W block of the result per workgroup:  C[m, n] = A^T B,  A [k, m],
// B [k, n], k = the rows of the graph or of the batch (10^5 .. 10^7).  gfx950.  Two geometries:
//   W = 224 (7 waves): a layer 129 .. 224 wide in ONE block (citation2's h = 200);
//   W = 256 (8 waves): m and n multiples of 256 (collab 256 x 512, ddi 512 x 512), the result cut into 256 x 256 blocks;
//       B may be two buffers side by side along n ([agg | x], plnlp_gemm_pair_f32) and its rows may be gathered (b_index).
// The text below describes W = 224; the other geometry differs only in the constants.
//
// Why a kernel of its own: on the 128 x 128 kernels (gemm_f32.hip) a 200 x 200 result is four tiles of which 39 % is
// padding, every operand panel is fetched by two workgroups, and each loaded element -- split into its three bf16 terms
// on the way into LDS, 11 VALU instructions per pair -- feeds 16 MFMA blocks per 256 loaded elements.  Here ONE workgroup
// holds the whole result: 448 threads = 7 waves, wave w owns the 32-row strip w of C and walks its seven 32 x 32 column
// blocks (112 accumulator registers); a K-step loads 16 rows of A and of B once (224 columns each), splits them
// once, and feeds 49 MFMA blocks per 448 loaded elements -- 1.75 x the MFMA work per split.  The reduction over k
// is cut into `slices` contiguous ranges, one workgroup each (one per CU: the two LDS buffers are 84 KB), whose raw
// partials wide_reduce_kernel (below) adds in a fixed order: deterministic, no atomics.
//
// Same arithmetic as the tile kernels (same split x = hi + mid + lo, round-to-nearest each; the same six products per
// 32 x 32 x 16 block, small terms first); the slices differ, so the f32 sums associate differently -- inside the
// contract every split-K launch already has (tests: against float64, bound 2^-22 sum |a||b|).
//
// LDS image of one operand K-step (16 x 224), in 16-byte units: [term 3][k-group 2][position 224], one unit = the 8 bf16
// k 8g .. 8g+7 of one column = exactly one lane's MFMA fragment: ONE ds_read_b128 (256 B/clk; the first version of this
// kernel kept the tile kernels' row-contiguous image -- a fragment = four ds_read_b32 at 128 B/clk, 172 KB of reads per
// K-step: 5 % slower).  To write whole units a staging thread owns 8 consecutive k of TWO adjacent columns
// (eight 8-byte loads, lanes side by side: a wave reads 512 contiguous bytes of a row); the two columns go to positions
// cp and w/2 + cp (w = the operand's width), so that both of a wave's stores cover consecutive units (conflict-free) and
// the live positions are 0 .. w-1.  Position p therefore holds column 2p (p < w/2) or 2 (p - w/2) + 1: the MFMAs work on
// positions, the write-back maps them to rows / columns of C.  Threads whose column pair lies past the width re-read the
// last pair and park it at the dead positions w .. 223 (products nobody stores).
#include "gemm_wgw.hip.h"
#include <utility>

namespace plnlp {
namespace wgw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KT = 16;                  // k per step
template <int NW>
struct Geo {                            // NW waves = NW 32-row strips of the block
    static constexpr int W = 32 * NW;           // rows / columns of the block
    static constexpr int CP = W / 2;            // column pairs per operand row
    static constexpr int NT = 64 * NW;          // threads: 2 operands x 2 k-groups x CP column pairs
    static constexpr int TERMU = 2 * W;         // 16-byte units per term
    static constexpr int OPERU = 3 * TERMU;     // units per operand
    static constexpr int BUFU = 2 * OPERU;      // units per buffer (A then B)
    static constexpr int LDS_BYTES = 2 * BUFU * 16;     // 86 016 / 98 304
};

__device__ __forceinline__ unsigned pk(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }

// one staging thread's share of a K-step: 8 consecutive k x 2 adjacent columns of one operand
struct Raw { f32x2 x[8]; };

// where a staging thread reads and writes (fixed for the whole launch)
struct Role {
    const float* src;     // its operand at (row 0, its column pair)
    int64_t ld;
    int unit0, unit1;     // LDS units (within a buffer, term 0) of its two columns
    int row8;             // its first row within a K-step (0 or 8)
    const int32_t* index; // GATHER kernels: the launch's row list (always readable) ...
    bool gathered;        // ... and whether THIS thread's operand reads row index[j] for reduction index j, or row j itself
};

// the split of one staging thread's share (8 k x 2 columns) in 18 stages of ~5 VALU instructions that ride between MFMAs:
// per column, stage 0-3 = hi term of k-pair q (and the residual), 4-7 = mid term, 8 = lo terms; a term's 16-byte unit is
// stored as soon as its four dwords exist (stages 3, 7, 8), so a wave's six stores are spread over the step -- three
// back-to-back ds_write_b128 from all the waves at once filled the LDS store queue and stalled the waves at issue
// COLSUM: the thread also keeps the running sums of the raw values of its two columns (`cs`, in program order; `live` = 1.f
// for a real tile, 0.f for the look-ahead tiles past the slice's end, which re-read its last tile) -- the column sums of A = dz are
// the layer's bias gradient, which used to be a second pass over dz beside this kernel (colsum_partial_vec_kernel)
struct Split {
    f32x2 r[4];
    unsigned t[4];        // the term being assembled
    template <int I, int TERMU, bool COLSUM>
    __device__ __forceinline__ void stage(const Raw& w, u32x4* __restrict__ buf, const Role& ro, float live, f32x2& cs) {
#pragma clang fp contract(off)
        constexpr int C = I / 9, K = I % 9;
        u32x4* dst = buf + (C ? ro.unit1 : ro.unit0);
        if constexpr (K < 4) {
            f32x2 v = {w.x[2 * K][C], w.x[2 * K + 1][C]};
            if constexpr (COLSUM) {
                cs[C] = __builtin_fmaf(v.x, live, cs[C]);
                cs[C] = __builtin_fmaf(v.y, live, cs[C]);
            }
            t[K] = pk(v);
            r[K].x = v.x - __uint_as_float(t[K] << 16);
            r[K].y = v.y - __uint_as_float(t[K] & 0xffff0000u);
        } else if constexpr (K < 8) {
            constexpr int Q = K - 4;
            t[Q] = pk(r[Q]);
            r[Q].x -= __uint_as_float(t[Q] << 16);
            r[Q].y -= __uint_as_float(t[Q] & 0xffff0000u);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = pk(r[q]);
        }
        if constexpr (K == 3 || K == 7 || K == 8) {
            const u32x4 v = {t[0], t[1], t[2], t[3]};
            dst[(K == 3 ? 0 : K == 7 ? 1 : 2) * TERMU] = v;
        }
    }
};

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// the whole split of one thread's share, not interleaved with anything (prologue, the partial last step)
template <int TERMU, bool COLSUM>
__device__ __forceinline__ void split_store(u32x4* __restrict__ buf, const Role& ro, const Raw& w, f32x2& cs) {
    Split sp;
    static_for<18>([&](auto ic) { sp.template stage<decltype(ic)::value, TERMU, COLSUM>(w, buf, ro, 1.f, cs); });
}

// GATHER: the operand rows of the 8 reduction indices a thread stages in one K-step (wave-uniform addresses: broadcast loads)
struct Rows { i32x4 lo, hi; };
__device__ __forceinline__ void load_rows(const Role& ro, Rows& ix, int64_t step) {
    const int32_t* p = ro.index + step * KT + ro.row8;
    ix.lo = *reinterpret_cast<const i32x4*>(p);
    ix.hi = *reinterpret_cast<const i32x4*>(p + 4);
}

// a FULL K-step (no reduction index past the end): plain 8-byte loads, rows I0 .. I1-1 of the thread's 8.  GATHER: the rows
// come from `ix` (loaded a step earlier) where the thread's operand is gathered
template <bool GATHER, int I0 = 0, int I1 = 8>
__device__ __forceinline__ void load_full(const Role& ro, Raw& w, int64_t step, const Rows& ix) {
    const int64_t r0 = step * KT + ro.row8;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        int64_t r = r0 + i;
        if constexpr (GATHER) r = ro.gathered ? (int64_t)(i < 4 ? ix.lo[i & 3] : ix.hi[i & 3]) : r;
        w.x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(ro.src + r * ro.ld));
    }
}

// one lane's fragment of term s: k 8h .. 8h+7 of position `pos`
template <int NW>
__device__ __forceinline__ bf16x8 frag(const u32x4* __restrict__ oper, int s, int pos, int h) {
    return __builtin_bit_cast(bf16x8, oper[s * Geo<NW>::TERMU + h * Geo<NW>::W + pos]);
}

constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};    // the six products, small terms first

// the MFMAs of one K-step from buffer `buf`, nothing else (the reduction's partial last step)
template <int NW, int NBLK>
__device__ __forceinline__ void mma_plain(f32x16 (&acc)[NBLK], const u32x4* __restrict__ buf, int wave, int l31, int h) {
    bf16x8 a[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) a[s] = frag<NW>(buf, s, 32 * wave + l31, h);
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        bf16x8 b[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) b[s] = frag<NW>(buf + Geo<NW>::OPERU, s, 32 * j + l31, h);
#pragma unroll
        for (int u = 0; u < 6; ++u) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[u]], b[TB[u]], acc[j], 0, 0, 0);
    }
}

// One pipelined K-step.  The workgroup is alone on its CU and its 7 waves meet at a barrier every step, so everything that is
// not an MFMA must ride BEHIND MFMAs inside each wave -- a step is one basic block of 6 NBLK slots (slot = one MFMA + what is
// pinned behind it with sched_barrier; nothing crosses a slot):
//   * buffer D holds this step's terms (tile t); the B fragments of column block j + 1 are fetched behind the first MFMA of
//     block j;
//   * the step's ONE barrier sits at slot BS = 6 NBLK - 10: after this wave's last read of buffer D and its last store into
//     buffer D^1, with 10 MFMAs still to issue.  Behind those, the next step's A fragments and first B fragments are
//     fetched from buffer D^1 (complete as of the barrier): the next step starts on its MFMAs at once.  (With the barrier at
//     the end of the step, store drain + barrier + fragment latency left the matrix pipe idle 30 % of the time.)
//   * the split of a tile runs from one barrier to the next: stages 0-4 of tile t + 2 (register set D, into buffer D --
//     free once barrier(t) has passed) behind the last slots of this step, stages 5-17 of tile t + 1 (register set D^1,
//     into buffer D^1) spread over the slots before BS - 7 -- one stage every other slot, each store in a slot of its own;
//     then set D^1 is refilled with tile t + 3 (its loads have a step to land).
// Hazards: every read of buffer D lies before barrier(t); writes of buffer D (tile t + 2) come after it.  Buffer D^1 is
// written before barrier(t) and read after it; its previous reads (step t - 1) lie before barrier(t - 1).
// P: which of the two B fragment sets this step's block 0 uses (it alternates per step when NBLK is odd).
template <int HEAD>
constexpr int head_stage_at(int q) {            // the split stage (5 .. 17) pinned behind slot q of the head, -1: none
    for (int i = 5; i < 18; ++i)
        if (((i - 5) * HEAD) / 13 == q) return i;
    return -1;
}
// live1 / live2 (COLSUM): 1.f when the tile whose split rides in this step -- tile + 1 (stages 5 .. 17) / tile + 2 (stages 0 .. 4) --
// is a real tile of the slice
template <int NW, int NBLK, int D, int P, bool GATHER, bool COLSUM>
__device__ __forceinline__ void step_pipelined(f32x16 (&acc)[NBLK], u32x4* __restrict__ lds, Raw (&raw)[2], Split (&S)[2],
                                               const Role& ro, int64_t next, int64_t next_rows, Rows& ix, bf16x8 (&A)[2][3],
                                               bf16x8 (&B)[2][3], int wave, int l31, int h, float live1, float live2, f32x2& cs) {
    typedef Geo<NW> G;
    u32x4* cbuf = lds + D * G::BUFU;
    const u32x4* bt = cbuf + G::OPERU;
    u32x4* nbuf = lds + (D ^ 1) * G::BUFU;
    constexpr int SLOTS = 6 * NBLK, BS = SLOTS - 10, HEAD = BS - 7, LD = BS - 6;
    static_assert(HEAD >= 13 && BS == 6 * (NBLK - 2) + 2, "slot plan");
    __builtin_amdgcn_sched_barrier(0);
    static_for<SLOTS>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr int j = q / 6, u = q % 6;
        if constexpr (q == BS) __syncthreads();
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[D][TA[u]], B[(P + j) & 1][TB[u]], acc[j], 0, 0, 0);
        if constexpr (u == 0 && j + 1 < NBLK) {
#pragma unroll
            for (int s = 0; s < 3; ++s) B[(P + j + 1) & 1][s] = frag<NW>(bt, s, 32 * (j + 1) + l31, h);
        }
        if constexpr (q == BS) {                     // the next step's A fragments (the other set: this step still multiplies A[D])
#pragma unroll
            for (int s = 0; s < 3; ++s) A[D ^ 1][s] = frag<NW>(nbuf, s, 32 * wave + l31, h);
        }
        if constexpr (q == BS + 4) {                 // ... and its first B fragments, into the set block NBLK - 2 has just left
#pragma unroll
            for (int s = 0; s < 3; ++s) B[(P + NBLK) & 1][s] = frag<NW>(nbuf + G::OPERU, s, l31, h);
        }
        if constexpr (q < HEAD) {
            constexpr int st = head_stage_at<HEAD>(q);
            if constexpr (st >= 0) S[D ^ 1].template stage<st, G::TERMU, COLSUM>(raw[D ^ 1], nbuf, ro, live1, cs);
        }
        if constexpr (q == LD)     load_full<GATHER, 0, 4>(ro, raw[D ^ 1], next, ix);
        if constexpr (q == LD + 1) {
            load_full<GATHER, 4, 8>(ro, raw[D ^ 1], next, ix);
            if constexpr (GATHER) load_rows(ro, ix, next_rows);      // (the rows of the NEXT step's refill; every thread: no branch)
        }
        if constexpr (q > BS && ((q - BS) & 1)) S[D].template stage<(q - BS) / 2, G::TERMU, COLSUM>(raw[D], cbuf, ro, live2, cs);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// position -> column of an operand block `width` wide (see the header)
__device__ __forceinline__ int column_of(int pos, int width) {
    const int half = width >> 1;
    return pos < half ? 2 * pos : 2 * (pos - half) + 1;
}

// NBLK: 32-position blocks of B with a live position -- compile-time, so that a K-step is ONE basic block
template <int NW, int NBLK, bool GATHER, bool COLSUM>
__global__ __launch_bounds__(64 * NW, 1) void wgrad_wide_kernel(Args g) {
    typedef Geo<NW> G;
    extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    // workgroup -> (K slice z, result block (mt, nt)): the blocks of one slice sit on ONE XCD (workgroups are dealt to the
    // XCDs round-robin), so the A rows they share come out of that XCD's L2 the second time
    int64_t z; int mt, nt;
    {
        const int tiles = g.tiles_m * g.tiles_n;
        const int64_t id = blockIdx.x, q = id >> 3;
        z = (q / tiles) * 8 + (id & 7);
        const int tile = (int)(q % tiles);
        mt = tile / g.tiles_n; nt = tile % g.tiles_n;
    }
    const int wa = g.m - mt * G::W < G::W ? g.m - mt * G::W : G::W;        // live rows / columns of this block
    const int wb = g.n - nt * G::W < G::W ? g.n - nt * G::W : G::W;
    Role ro;
    const int o = t >= 2 * G::CP, kg = (t - 2 * G::CP * o) >= G::CP, cp = t - 2 * G::CP * o - G::CP * kg;    // operand, k-group, column pair
    {
        const int width = o ? wb : wa, half = width >> 1;
        const bool live = cp < half;
        const int col = live ? 2 * cp : width - 2;
        const int p0 = live ? cp : width + 2 * (cp - half);
        const int p1 = live ? half + cp : p0 + 1;
        // B's columns from nb_split on live in the second buffer (the pair form: [agg | x])
        const bool second = o && g.b2 && nt * G::W >= g.nb_split;
        const float* base = !o ? g.a + mt * G::W : (second ? g.b2 + (nt * G::W - g.nb_split) : g.b + nt * G::W);
        ro.ld = !o ? g.lda : (second ? g.ldb2 : g.ldb);
        ro.row8 = 8 * kg;
        ro.src = base + col;
        ro.unit0 = o * G::OPERU + kg * G::W + p0;
        ro.unit1 = o * G::OPERU + kg * G::W + p1;
        ro.index = g.b_index;
        ro.gathered = GATHER && o && (g.bidx_mask & (second ? 2 : 1));
    }

    // K-steps of this slice: [sb, sf) full ones, then (last slice only) the reduction's partial last step
    const int64_t steps = (g.k + KT - 1) / KT, full = g.k / KT;
    const int64_t per = (steps + g.slices - 1) / g.slices;
    const int64_t sb = z * per;
    const int64_t se = sb + per < steps ? sb + per : steps;
    const int64_t sf = se < full ? se : full;

    f32x16 acc[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    f32x2 cs = {0.f, 0.f};            // COLSUM: this thread's running column sums (two columns of its operand)

    if (sb < sf) {
        const int64_t last = sf - 1;
        auto at = [&](int64_t step) { return step < last ? step : last; };
        Raw raw[2];
        Split S[2];
        Rows ix{};
        bf16x8 A[2][3], B[2][3];
        {   // (GATHER: the row lists of the first three steps in ONE round trip, then the data that depends on them)
            Rows i0{}, i1{}, i2{};
            if constexpr (GATHER) { load_rows(ro, i0, sb); load_rows(ro, i1, at(sb + 1)); load_rows(ro, i2, at(sb + 2)); }
            load_full<GATHER>(ro, raw[0], sb, i0);
            load_full<GATHER>(ro, raw[1], at(sb + 1), i1);
            if constexpr (GATHER) load_rows(ro, ix, at(sb + 3));       // for the first step's refill
            split_store<G::TERMU, COLSUM>(lds, ro, raw[0], cs);
            load_full<GATHER>(ro, raw[0], at(sb + 2), i2);
        }
        {
            const float live1 = sb + 1 <= last ? 1.f : 0.f;
            static_for<5>([&](auto ic) { S[1].template stage<decltype(ic)::value, G::TERMU, COLSUM>(raw[1], lds + G::BUFU, ro, live1, cs); });
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            A[0][s] = frag<NW>(lds, s, 32 * wave + l31, h);
            B[0][s] = frag<NW>(lds + G::OPERU, s, l31, h);
        }
        int64_t i = sb;
        // (look-ahead loads past the slice's last full step re-read that step: what they stage is never multiplied)
        auto real = [&](int64_t tile) { return tile <= last ? 1.f : 0.f; };
        for (; i + 2 <= sf; i += 2) {
            step_pipelined<NW, NBLK, 0, 0, GATHER, COLSUM>(acc, lds, raw, S, ro, at(i + 3), at(i + 4), ix, A, B, wave, l31, h, real(i + 1), real(i + 2), cs);
            step_pipelined<NW, NBLK, 1, NBLK & 1, GATHER, COLSUM>(acc, lds, raw, S, ro, at(i + 4), at(i + 5), ix, A, B, wave, l31, h, real(i + 2), real(i + 3), cs);
        }
        if (i < sf) step_pipelined<NW, NBLK, 0, 0, GATHER, COLSUM>(acc, lds, raw, S, ro, last, last, ix, A, B, wave, l31, h, 0.f, 0.f, cs);
        __syncthreads();              // (the last step's look-ahead reads are behind us before anything re-uses the buffers)
    }
    if (se > sf && sb <= sf) {        // the reduction's partial last step: indices past the end contribute zeros
        Raw w;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t r = sf * KT + ro.row8 + i, rc = r < g.k ? r : g.k - 1;
            int64_t row = rc;
            if constexpr (GATHER) { if (ro.gathered) row = ro.index[rc]; }
            const f32x2 v = *reinterpret_cast<const f32x2*>(ro.src + row * ro.ld);
            const f32x2 zero = {0.f, 0.f};
            w.x[i] = r < g.k ? v : zero;
        }
        split_store<G::TERMU, COLSUM>(lds, ro, w, cs);
        __syncthreads();
        mma_plain<NW, NBLK>(acc, lds, wave, l31, h);
    }
    if constexpr (COLSUM) {
        // this slice's column sums of A: the two k-groups of a column pair meet in LDS, the first n-tile's workgroup writes them
        float* cl = reinterpret_cast<float*>(lds);
        __syncthreads();
        const f32x2 tot = cs;
        const bool mine = !o && cp < (wa >> 1);
        if (mine) *reinterpret_cast<f32x2*>(cl + kg * G::W + 2 * cp) = tot;
        __syncthreads();
        if (mine && !kg && nt == 0 && g.colsum_ws) {
            const f32x2 lo = *reinterpret_cast<const f32x2*>(cl + 2 * cp), hi = *reinterpret_cast<const f32x2*>(cl + G::W + 2 * cp);
            *reinterpret_cast<f32x2*>(g.colsum_ws + z * (int64_t)g.m + mt * G::W + 2 * cp) = lo + hi;
        }
    }

    // raw partial of this slice.  MFMA C/D map: position of B = lane & 31, position of A = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
    // within the block; once per workgroup (tens to hundreds of K-steps), so plain 4-byte stores
    float* out = g.ws + z * (int64_t)g.m * g.n + (int64_t)mt * G::W * g.n + nt * G::W;
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        const int pb = 32 * j + l31;
        if (pb >= wb) continue;
        const int col = column_of(pb, wb);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int pa = 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (pa < wa) out[(int64_t)column_of(pa, wa) * g.n + col] = acc[j][q];
        }
    }
}

// The rule, in one place.  One block (W = 224): wider than one 128-tile in BOTH directions (else the tile kernels waste nothing
// worth a second kernel).  Blocks of 256 x 256: both extents whole blocks, at most 8 of them.  Whole 16-byte column groups,
// aligned operands, a reduction long enough to give every CU a slice of a few dozen K-steps.
int slices_for(const Args& g) {
    if (g.k < MIN_K || !g.a || !g.b) return 0;
    if (((uintptr_t)g.a % 16) || ((uintptr_t)g.b % 16) || (g.lda % 4) || (g.ldb % 4) || g.lda < g.m) return 0;
    if (g.b2 && (((uintptr_t)g.b2 % 16) || (g.ldb2 % 4) || g.nb_split <= 0 || g.nb_split >= g.n)) return 0;
    if (g.b_index && ((uintptr_t)g.b_index % 16)) return 0;
    if (g.m % 256 == 0 && g.n % 256 == 0 && (!g.b2 || g.nb_split % 256 == 0)) {
        const int tiles = (g.m / 256) * (g.n / 256);
        if (tiles > 8 || (!g.b2 && g.ldb < g.n) || (g.b2 && (g.ldb < g.nb_split || g.ldb2 < g.n - g.nb_split))) return 0;
        return tiles <= 2 ? 256 / tiles : (tiles <= 4 ? 64 : 32);          // (a multiple of 8: the slices are dealt over the XCDs)
    }
    if (g.b2 || g.b_index) return 0;
    if (g.m <= 128 || g.n <= 128 || g.m > W || g.n > W || (g.m % 4) || (g.n % 4) || g.ldb < g.n) return 0;
    return 256;                   // one workgroup per CU
}

// the slices' sum: 8 threads per 16-byte group of the result, thread g adds slices [g S/8, (g+1) S/8) in order (all its loads
// in flight at once), the 8 partial sums are added in order of g.  The generic reduce of gemm_f32.hip walks ALL slices per
// thread: 256 slices = 32 dependent rounds of 8 loads on 40 workgroups, 0.1 ms for a 200 x 200 result.
constexpr int RG = 8, RE = 32;          // slice groups x result groups per workgroup (256 threads)
__global__ __launch_bounds__(RG * RE) void wide_reduce_kernel(const float* __restrict__ ws, int slices, int64_t stride,
                                                              float* __restrict__ c, int64_t ldc, int64_t m, int n, Epi epi,
                                                              float* __restrict__ c2, int64_t ldc2, int n_split,
                                                              const float* __restrict__ colsum_ws, float* __restrict__ colsum,
                                                              unsigned first_colsum_block) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ f32x4 part[RG][RE];
    const int e = threadIdx.x % RE, gidx = threadIdx.x / RE;
    // the launch's last blocks add the slices' column sums of A (m floats per slice) the same way
    const bool cs = blockIdx.x >= first_colsum_block;
    if (cs) { ws = colsum_ws; stride = m; }
    const int64_t i4 = (int64_t)(cs ? blockIdx.x - first_colsum_block : blockIdx.x) * RE + e, total4 = cs ? m >> 2 : (m * n) >> 2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (i4 < total4) {
        const int per = (slices + RG - 1) / RG;
        const int z0 = gidx * per, z1 = z0 + per < slices ? z0 + per : slices;
        const float* p = ws + i4 * 4;
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (int64_t)(z + u) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; z < z1; ++z) acc += *reinterpret_cast<const f32x4*>(p + (int64_t)z * stride);
    }
    part[gidx][e] = acc;
    __syncthreads();
    if (gidx != 0 || i4 >= total4) return;
#pragma unroll
    for (int g2 = 1; g2 < RG; ++g2) acc += part[g2][e];
    if (cs) {
        *reinterpret_cast<f32x4*>(colsum + i4 * 4) = acc;
        return;
    }
    const int64_t i = i4 * 4, row = i / n;
    const int col = (int)(i - row * n);
    float4 y = make_float4(acc.x, acc.y, acc.z, acc.w);
    if (c2 && col >= n_split) {                   // the pair form's second result (no epilogue there)
        *reinterpret_cast<float4*>(c2 + row * ldc2 + (col - n_split)) = y;
        return;
    }
    y = epi_apply4(epi, y, row, col, n, c + row * ldc);
    *reinterpret_cast<float4*>(c + row * ldc + col) = y;
}

template <int NW, int NBLK, bool GATHER, bool COLSUM>
static int launch_as2(const Args& g, hipStream_t s) {
    auto kernel = wgrad_wide_kernel<NW, NBLK, GATHER, COLSUM>;
    // (once per device and instantiation: the attribute belongs to the device's copy of the function, and a process may
    // hold several devices)
    static bool armed[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return PLNLP_E_UNSUPPORTED;
    if (dev < 0 || dev >= 64 || !armed[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                Geo<NW>::LDS_BYTES) != hipSuccess)
            return PLNLP_E_UNSUPPORTED;
        if (dev >= 0 && dev < 64) armed[dev] = true;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)(g.slices * g.tiles_m * g.tiles_n)), dim3(Geo<NW>::NT), Geo<NW>::LDS_BYTES, s, g);
    return launch_status();
}

template <int NW, int NBLK, bool GATHER>
static int launch_as(const Args& g, hipStream_t s) {
    return g.colsum_ws ? launch_as2<NW, NBLK, GATHER, true>(g, s) : launch_as2<NW, NBLK, GATHER, false>(g, s);
}

int launch(const Args& g_in, hipStream_t s) {
    Args g = g_in;
    if (g.slices % 8) return PLNLP_E_SHAPE;
    if (g.m % 256 == 0 && g.n % 256 == 0) {
        g.tiles_m = g.m / 256; g.tiles_n = g.n / 256;
        return (g.b_index && g.bidx_mask) ? launch_as<8, 8, true>(g, s) : launch_as<8, 8, false>(g, s);
    }
    g.tiles_m = g.tiles_n = 1;
    switch ((g.n + 31) / 32) {
        case 5: return launch_as<7, 5, false>(g, s);
        case 6: return launch_as<7, 6, false>(g, s);
        case 7: return launch_as<7, 7, false>(g, s);
    }
    return PLNLP_E_SHAPE;
}

// c[m, n] (leading dimension ldc) = epilogue(sum of the slices); c2 != nullptr: columns from n_split on go to c2[:, col - n_split]
// (the pair form).  Everything 16-byte aligned, leading dimensions and n_split multiples of 4 (the caller checks).
int reduce(const Args& g, float* c, int64_t ldc, const Epi& e, float* c2, int64_t ldc2, int n_split, float* colsum, hipStream_t s) {
    const int64_t total4 = ((int64_t)g.m * g.n) >> 2;
    const unsigned blocks = (unsigned)((total4 + RE - 1) / RE);
    const unsigned cs_blocks = (colsum && g.colsum_ws) ? (unsigned)(((g.m >> 2) + RE - 1) / RE) : 0u;
    hipLaunchKernelGGL(wide_reduce_kernel, dim3(blocks + cs_blocks), dim3(RG * RE), 0, s, g.ws, g.slices,
                       (int64_t)g.m * g.n, c, ldc, (int64_t)g.m, g.n, e, c2, ldc2, n_split, g.colsum_ws, colsum, blocks);
    return launch_status();
}

}  // namespace wgw
}  // namespace plnlp
