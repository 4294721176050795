// pipeline_siso_tdl_wave_f64.hip -- the one-realization-per-wavefront kernels of config 3 (siso_tdl_wave.hpp) in complex128
#include "siso_tdl_wave.hpp"

namespace mcle {

int run_siso_tdl_wave_f64(mcle_ctx* ctx, int fft_size, const SisoTdlParams& pp, int method, uint64_t seed, uint64_t first, uint64_t count,
                          mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    switch (fft_size) {
        case 256: return run_siso_tdl_wave<double, 256>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
        case 512: return run_siso_tdl_wave<double, 512>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
        case 1024: return run_siso_tdl_wave<double, 1024>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
        case 2048: return run_siso_tdl_wave<double, 2048>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
        default: return MCLE_E_UNSUPPORTED;
    }
}

}  // namespace mcle
