// pipe_common.hpp -- pieces shared by the fused pipeline translation units.
#pragma once
#include "common.hpp"
#include "modem.hpp"

namespace mcle {

constexpr int kPipeBlock = 256;
constexpr int kMaxTable = 256;  // Philox symbols are bytes

// workgroup-wide sum of two unsigned values; result valid in thread 0
__device__ __forceinline__ void block_sum2(unsigned& a, unsigned& b, unsigned* s_red) {
    a = wave_sum_u32(a);
    b = wave_sum_u32(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
        s_red[2 * wave] = a;
        s_red[2 * wave + 1] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0;
        b = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
            a += s_red[2 * w];
            b += s_red[2 * w + 1];
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------
// Grid of a persistent kernel whose workgroups each take an equal share of `units` (realizations, passes): up to eight
// times the resident set, as long as a workgroup keeps >= 8 units to spread its set-up over.  The queued workgroups start
// as the first ones finish, which evens out per-workgroup speed differences and shortens the tail -- measured on the
// matrix-core kernels (MCLE_OPT_GRID_OVERSUB = 1 / 2 / 4 / 8 / 16 / 32): config 4 1.516 / 1.458 / 1.415 / 1.401 / 1.411 / 2.02 ms
// per 65 536 realizations, config 3 1.602 / 1.575 / 1.555 / 1.541 / 1.550 / 1.608 ms per 131 072.
// Round 4: the planar config-4 kernels take up to SIXTEEN times the resident set (10.75 -> 10.66 ms complex128, 4.77 -> 4.70 ms
// complex64 per 262 144 realizations); the wavefront kernels of config 3 need >= 12 units (48 realizations) per workgroup -- their
// set-up (tables, twiddles, one barrier) is heavier: 2.17 -> 1.94 ms per 262 144, 1.20 -> 0.88 ms per 104 860
// (scripts/experiments/r04_oversub_probe.py).
inline uint64_t oversubscribed_grid(const mcle_ctx* ctx, uint64_t resident, uint64_t units, uint64_t min_units = 8,
                                    uint64_t max_f = 8) {
    uint64_t f = max_f;
    if (ctx->opt[MCLE_OPT_GRID_OVERSUB] > 0) {
        f = (uint64_t)ctx->opt[MCLE_OPT_GRID_OVERSUB];
    } else {
        while (f > 1 && units < resident * f * min_units) f >>= 1;
    }
    const uint64_t g = resident * f;
    return units < g ? units : g;
}

// What a workgroup must do per launch for its ONE flush of the counters to disappear: the flush is six atomics on one cache line,
// ~9 ns each and serialized chip-wide, so `grid` workgroups cost grid x 54 ns whatever else runs -- at 32 768 one-wavefront workgroups
// of eight realizations each that was 1.8 of the 2.5 ms of a (256, 1 x 2) complex64 launch (profiles/r06/planar_grid_ab.log: 1.06e8
// realizations/s at the old grid, 3.2e8 with one workgroup per resident slot).  A realization of `samples` = fft_size x receive
// antennas takes ~ps_per_sample picoseconds of the chip; the workgroup gets enough of them for the flush to stay below ~10 %.
inline uint64_t flush_min_units(uint64_t base, uint64_t k_samples, uint64_t samples) {
    const uint64_t m = k_samples / (samples > 0 ? samples : 1);
    return m > base ? m : base;
}

template <typename T> ModemParams<T> pipe_modem(const mcle_ctx* ctx, int method) {
    ModemParams<T> p;
    // the candidate grid serves the complex128 kernels too since round 3 (cell from the float-rounded point, the literal f64
    // metric on the listed points: decision-identical to the sweep, modem.hpp) -- the f64 sweep over 64 points doubled their time
    p.grid = context_grid<T>(ctx, method, true);
    if (sizeof(T) == 8)
        p.g_table = reinterpret_cast<const cx<T>*>(ctx->d_table_f64);
    else
        p.g_table = reinterpret_cast<const cx<T>*>(ctx->d_table_f32);
    p.M = ctx->M;
    p.bits = ctx->bits;
    p.method = method;
    p.qam_scale = (T)ctx->qam_scale;
    p.qam_L = ctx->qam_L;
    p.half_bits = ctx->bits / 2;
    modem_fill_cert(ctx, method, p);
    return p;
}

inline int check_pipe(const mcle_ctx* ctx, int dtype, int method, const void* cfg) {
    MCLE_REQUIRE(ctx != nullptr && cfg != nullptr, "null argument");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(ctx->M > 0, "no constellation set (mcle_set_constellation)");
    MCLE_REQUIRE(ctx->M <= kMaxTable, "fused pipelines draw symbols as bytes: M <= %d", kMaxTable);
    MCLE_REQUIRE(method == MCLE_DEMOD_MINDIST || method == MCLE_DEMOD_QAM_SLICER, "bad demodulation method");
    MCLE_REQUIRE(method != MCLE_DEMOD_QAM_SLICER || ctx->kind == MCLE_CONST_QAM,
                 "the slicer needs a square Gray QAM constellation (kind MCLE_CONST_QAM)");
    return MCLE_OK;
}

}  // namespace mcle
