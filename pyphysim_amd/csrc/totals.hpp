// totals.hpp -- per-workgroup running totals of the fused pipelines and their flush into the
// exact-integer mcle_counters block (what Result.update / merge accumulate, reference
// simulations/results.py:469-623).
#pragma once
#include "common.hpp"

namespace mcle {

// Running totals of one workgroup (held by thread 0), flushed with six atomics at kernel end:
// the counter block is exact integer sums, so the order of the atomics is irrelevant.
// Lives in LDS (a `__shared__ WgTotals`): thread 0 touches it once per realization, and keeping six
// 64-bit accumulators in VGPRs across the whole kernel made the allocator spill them to scratch.
struct WgTotals {
    unsigned long long se, se2, be, be2, ok, skip;
};
__device__ __forceinline__ void wg_zero(WgTotals& t) {
    t.se = t.se2 = t.be = t.be2 = t.ok = t.skip = 0ull;
}
__device__ __forceinline__ void wg_account(WgTotals& t, unsigned se, unsigned be, bool skipped, uint64_t rl,
                                           uint32_t* __restrict__ sym_out, uint32_t* __restrict__ bit_out) {
    if (sym_out) sym_out[rl] = skipped ? 0xFFFFFFFFu : se;
    if (bit_out) bit_out[rl] = skipped ? 0xFFFFFFFFu : be;
    if (skipped) {
        ++t.skip;
    } else {
        ++t.ok;
        t.se += se;
        t.se2 += (unsigned long long)se * se;
        t.be += be;
        t.be2 += (unsigned long long)be * be;
    }
}
__device__ __forceinline__ void wg_flush(const WgTotals& t, mcle_counters* counters, unsigned long long n_sym,
                                         unsigned long long n_bits) {
#if defined(MCLE_EXPERIMENTS) && defined(MCLE_NO_FLUSH)
    return;                 // (timing bound: no flush at all -- wrong counters)
#endif
    if (!counters) return;
    // (a zero is not added: the six words share a cache line and its atomics are serialized chip-wide, ~9 ns each -- `skip` is zero
    //  outside the iterative solvers' outage cases, the error sums at high SNR)
    if (t.se) atomicAdd((unsigned long long*)&counters->sym_errors, t.se);
    if (t.se2) atomicAdd((unsigned long long*)&counters->sym_errors_sq, t.se2);
    if (t.be) atomicAdd((unsigned long long*)&counters->bit_errors, t.be);
    if (t.be2) atomicAdd((unsigned long long*)&counters->bit_errors_sq, t.be2);
    if (t.ok) atomicAdd((unsigned long long*)&counters->n_realizations, t.ok);
    if (t.skip) atomicAdd((unsigned long long*)&counters->n_skipped, t.skip);
    if (blockIdx.x == 0) {
        counters->n_symbols = n_sym;
        counters->n_bits = n_bits;
    }
}

// A workgroup of NWV independent wavefronts with per-wavefront totals (t[w], written by each wavefront's lane 0): ONE flush.
// Six global atomics per WAVEFRONT put tens of thousands of atomics per launch on the counters' one cache line (~9 ns each, and a
// wavefront's slot is held until they are acknowledged): a third of a short kernel's time (round 6, profiles/r06/walk_grid_sweep.log).
// Call from every thread of the workgroup at kernel end.
template <int NWV>
__device__ __forceinline__ void wg_flush_waves(WgTotals (&t)[NWV], mcle_counters* counters, unsigned long long n_sym,
                                               unsigned long long n_bits) {
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < NWV; ++i) {
            t[0].se += t[i].se;
            t[0].se2 += t[i].se2;
            t[0].be += t[i].be;
            t[0].be2 += t[i].be2;
            t[0].ok += t[i].ok;
            t[0].skip += t[i].skip;
        }
        wg_flush(t[0], counters, n_sym, n_bits);
    }
}

}  // namespace mcle
