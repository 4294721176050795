// pipeline_siso_tdl_wave_f32.hip -- the wavefront kernels of config 3 (siso_tdl_wave.hpp: one realization per wavefront; siso_tdl_hw.hpp: two
// wavefronts per realization at 2048 points) in complex64
#include "siso_tdl_hw.hpp"
#include "siso_tdl_wave.hpp"

namespace mcle {

int run_siso_tdl_wave_f32(mcle_ctx* ctx, int fft_size, const SisoTdlParams& pp, int method, uint64_t seed, uint64_t first, uint64_t count,
                          mcle_counters* d_counters, uint32_t* d_sym, uint32_t* d_bit) {
    switch (fft_size) {
        case 256: return run_siso_tdl_wave<float, 256>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
        case 512: return run_siso_tdl_wave<float, 512>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
        case 1024: return run_siso_tdl_wave<float, 1024>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
        case 2048: {                 // two wavefronts per realization (siso_tdl_hw.hpp, round 6; one realization per workgroup (six per CU; 5.72 against 5.67e7 with two));
                                     // MCLE_OPT_TDL_KERNEL = 3: the one-wavefront kernel (A/B), which also serves what is outside the envelope
            if (ctx->opt[MCLE_OPT_TDL_KERNEL] != 3) {
                const int rc = run_siso_tdl_hw<float, 3, 1>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
                if (rc != MCLE_E_UNSUPPORTED) return rc;
            }
            return run_siso_tdl_wave<float, 2048>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
        }
        default: return MCLE_E_UNSUPPORTED;
    }
}

}  // namespace mcle
