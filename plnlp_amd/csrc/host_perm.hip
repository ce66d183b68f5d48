// Host side of the hot loop's index stream: the DataLoader batch permutation (plnlp/model.py:147), produced
// INCREMENTALLY.  No device code in this file.
//
// `DataLoader(range(E), B, shuffle=True)` permutes with torch.randperm(E, generator=Generator().manual_seed(seed))
// on the CPU: a forward Fisher-Yates shuffle driven by MT19937 (ATen randperm_cpu, the n < 2^32 / 20 branch):
//     for i in 0 .. n-2:  z = mt() % (n - i);  swap(perm[i], perm[i + z])
// After iteration i the entries perm[0 .. i] are FINAL.  The reference (and round 2 of this repo) runs the whole
// shuffle before the first step: at the collab recipe's 23 M random-walk pairs that is 0.8 s of host time in front of
// a 0.6 s GPU epoch.  These entry points reproduce the same permutation bit for bit in caller-owned memory and in
// slices, so a host thread shuffles a few batches ahead of the GPU (plnlp_amd/utils.py::StreamedPermutation) --
// and, with the draws of 32 iterations ahead used as prefetch hints for the random side of each swap, in about a
// third of the time of the loop it restates.
#include <stdint.h>
#include "../../include/plnlp_hip.h"

namespace {

struct MT {          // at::mt19937: state[624] then the read index, exactly the caller's uint32 [625]
    uint32_t s[624];
    uint32_t idx;
};

inline void mt_refill(MT* m) {
    uint32_t* s = m->s;
    for (int k = 0; k < 624; ++k) {
        const uint32_t y = (s[k] & 0x80000000u) | (s[(k + 1) % 624] & 0x7fffffffu);
        s[k] = s[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    m->idx = 0;
}

inline uint32_t mt_next(MT* m) {
    if (m->idx >= 624) mt_refill(m);
    uint32_t y = m->s[m->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

}  // namespace

extern "C" int plnlp_host_randperm_init(uint64_t seed, int64_t n, int64_t* perm, uint32_t* mt_state) {
    if (n < 0 || n >= (int64_t)(0xFFFFFFFFu / 20u)) return n < 0 ? PLNLP_E_SHAPE : PLNLP_E_UNSUPPORTED;
    if (!mt_state || (n > 0 && !perm)) return PLNLP_E_NULL;
    MT* m = reinterpret_cast<MT*>(mt_state);
    m->s[0] = (uint32_t)seed;                     // at::mt19937(seed): the low 32 bits seed the generator
    for (uint32_t i = 1; i < 624; ++i) m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + i;
    m->idx = 624;
    for (int64_t i = 0; i < n; ++i) perm[i] = i;
    return 0;
}

extern "C" int plnlp_host_randperm_advance(int64_t n, int64_t* perm, uint32_t* mt_state, int64_t from, int64_t to) {
    if (n < 0 || from < 0 || to < from || to > n) return PLNLP_E_SHAPE;
    if (n >= (int64_t)(0xFFFFFFFFu / 20u)) return PLNLP_E_UNSUPPORTED;
    if (!mt_state || (n > 0 && !perm)) return PLNLP_E_NULL;
    MT* m = reinterpret_cast<MT*>(mt_state);
    if (to > n - 1) to = n - 1;                   // the last entry needs no draw
    constexpr int LA = 32;                        // draws taken ahead of their swaps (the draws do not depend on them)
    uint32_t z[LA];
    int64_t i = from;
    while (i < to) {
        const int cnt = (to - i) < LA ? (int)(to - i) : LA;
        for (int u = 0; u < cnt; ++u) {
            z[u] = mt_next(m) % (uint32_t)(n - (i + u));
            __builtin_prefetch(&perm[i + u + z[u]], 1, 0);
        }
        for (int u = 0; u < cnt; ++u) {
            const int64_t a = i + u, b = a + z[u];
            const int64_t sav = perm[a];
            perm[a] = perm[b];
            perm[b] = sav;
        }
        i += cnt;
    }
    return 0;
}
