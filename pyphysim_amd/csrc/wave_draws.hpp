// wave_draws.hpp -- wave-cooperative use of the mcle-philox-v1 streams for the kernels in which ONE wavefront
// walks the symbol columns of a realization (k_run_ia, k_run_bd, k_run_mimo_flat).
//
// The contract fixes which Philox block a draw comes from: symbol n is byte n & 15 of DATA block n >> 4,
// complex normal i is the word pair i & 1 of its stream's block i >> 1.  A lane that takes one column per pass
// spends a full Philox evaluation (40 quarter-rate multiplies) on one byte and one on each half-used noise
// block.  Here a pass covers 128 columns, lane l owning columns t0 + 2l and t0 + 2l + 1:
//   * noise: for an even row stride the two columns are the two samples of ONE block (cn_pair) -- every word used;
//   * symbols: the (at most) 9 blocks that cover a stream's 128 positions are evaluated by 9 different lanes in
//     the SAME Philox call as the blocks of six other streams (7 x 9 = 63 lanes), and each lane fetches the word
//     holding its two bytes with four ds_bpermute.
// Values are identical to symbol_at / cn_sample position by position (tests/test_gpu_pipelines.py compare the
// kernels with the oracle's draws); only who computes them changes.  Needs an even number of columns per row.
#pragma once
#include "philox.hpp"

namespace mcle {

constexpr int kPairCols = 128;            // columns per wave pass
constexpr int kBlocksPerRun = 9;          // 128 bytes at any alignment touch <= 9 sixteen-byte blocks
constexpr int kStreamsPerRound = 64 / kBlocksPerRun;   // 7

// Symbols of `n_streams` rows (row j at positions j * stride + column) for columns t0 + 2 * lane (+ 1).
// stride even, t0 a multiple of 128.  s0[j], s1[j] valid for j < n_streams.
template <int MAXS>
__device__ __forceinline__ void wave_symbol_pairs(const Rng& rng, int n_streams, uint32_t stride, uint32_t t0,
                                                  uint32_t mask, int lane, int (&s0)[MAXS], int (&s1)[MAXS]) {
    const int jj = lane / kBlocksPerRun, bi = lane - jj * kBlocksPerRun;
#pragma unroll
    for (int j0 = 0; j0 < MAXS; j0 += kStreamsPerRound) {
        if (j0 >= n_streams) break;
        // producer side: lane -> (stream j0 + jj, block bi of that stream's run)
        const uint32_t qb = (uint32_t)(j0 + jj) * stride + t0;
        const Words4 blk = rng.block(STREAM_DATA, (qb >> 4) + (uint32_t)bi);
#pragma unroll
        for (int k = 0; k < kStreamsPerRound; ++k) {
            const int j = j0 + k;
            if (j >= MAXS || j >= n_streams) break;
            const uint32_t base = (uint32_t)j * stride + t0;
            const uint32_t q = base + 2u * (uint32_t)lane;
            const int src = k * kBlocksPerRun + (int)((q >> 4) - (base >> 4));
            const uint32_t w0 = (uint32_t)__shfl((int)blk.w[0], src, 64), w1 = (uint32_t)__shfl((int)blk.w[1], src, 64);
            const uint32_t w2 = (uint32_t)__shfl((int)blk.w[2], src, 64), w3 = (uint32_t)__shfl((int)blk.w[3], src, 64);
            const uint32_t sel = (q >> 2) & 3u;
            const uint32_t word = sel == 0 ? w0 : (sel == 1 ? w1 : (sel == 2 ? w2 : w3));
            const uint32_t sh = (q & 3u) * 8u;          // q even: bytes sh and sh + 8 of the same word
            s0[j] = (int)((word >> sh) & mask);
            s1[j] = (int)((word >> (sh + 8u)) & mask);
        }
    }
}

}  // namespace mcle
