// The running K-best list of a wave and its update with a tile of 64 candidates.
//
// A candidate is the 64-bit key (dist_bits << 32 | idx); keys are unique and their unsigned
// order is the canonical (dist, idx) order.  Lane l < K holds the l-th smallest key seen so far
// (INF = ~0 when there are fewer), `thr` is the K-th (wave-uniform).  A tile contributes the
// keys that beat `thr` ("winners", typically all 64 for the first tile and a handful later).
//
// Update = MERGE BY RANK, not one insertion per winner: every element of (best U winners) gets
// its rank in the union,
//     rank(best[l])  = l + #{winners < best[l]}
//     rank(winner x) = #{best < x} + #{winners < x}
// and the elements of rank < K are scattered to their lane through a 512-byte LDS slot of the
// wave (write INF everywhere, write the ranked keys, read back: LDS operations of one wave
// complete in order, no barrier).  The two winner counts come from one loop over the winners
// (2 readlanes + 2 compares each), the best count from a loop over the K list entries.  One
// insertion costs a chain of ~250 cycles (readlane, ballot, shift, readlane); the merge costs
// ~12 instructions per winner plus ~4 per list entry plus four LDS operations, whatever the
// number of winners -- 5x fewer cycles on the first tiles, where every candidate is a winner.
#pragma once
#include "tpg_common.hpp"

__device__ __forceinline__ void tpg_knn_merge(tpg_u64 &best, tpg_u64 &thr, tpg_u64 key, int K, int lane,
                                              tpg_u64 *__restrict__ slot /* LDS, 64 entries, this wave's */) {
    const tpg_u64 INF = ~0ull;
    tpg_u64 mask = __ballot(key < thr);
    if (mask == 0) return;
    if ((mask & (mask - 1)) == 0) {       // a single winner: plain insertion is cheaper
        const int src = __builtin_amdgcn_readfirstlane(__builtin_ctzll(mask));
        const tpg_u64 x = tpg_readlane_u64(key, src);
        const int pos = __popcll(__ballot(best < x));
        const tpg_u64 up = tpg_wave_shr1_u64(best);
        best = lane < pos ? best : (lane == pos ? x : up);
        thr = tpg_readlane_u64(best, K - 1);
        return;
    }
    const bool win = (mask >> lane) & 1ull;
    int cb = 0, ck = 0;                   // winners below best[lane] / below this lane's key
    for (tpg_u64 m = mask; m; m &= m - 1) {
        const int s = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
        const tpg_u64 xs = tpg_readlane_u64(key, s);
        cb += xs < best ? 1 : 0;
        ck += xs < key ? 1 : 0;
    }
    int kb = 0;                           // list entries below this lane's key
    for (int l = 0; l < K; ++l) kb += tpg_readlane_u64(best, l) < key ? 1 : 0;
    slot[lane] = INF;
    const int rb = lane + cb, rk = kb + ck;
    if (lane < K && best != INF && rb < K) slot[rb] = best;
    if (win && rk < K) slot[rk] = key;
    best = slot[lane];
    thr = tpg_readlane_u64(best, K - 1);
}
