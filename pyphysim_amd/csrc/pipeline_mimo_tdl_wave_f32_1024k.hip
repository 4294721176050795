// pipeline_mimo_tdl_wave_f32_1024k.hip -- the one-receive-antenna-per-wavefront kernels of the frequency-selective MIMO-OFDM link (mimo_tdl_wave.hpp) in
// complex64: fft_size 1024, polynomial order 2 parked in registers (the benchmark's); every 1 <= Nt <= Nr <= 4
#include "mimo_tdl_wave.hpp"

namespace mcle {

MCLE_MIMO_TDL_WAVE_TU(run_mimo_tdl_wave_f32_1024k, float, 1024, mimo_tdl_wave_kf<float>())

#ifdef MCLE_EXPERIMENTS
// timing experiments on the benchmark geometry (4 x 4; option mimo_tdl_kernel = 16 + code): stage ablations (wrong results by
// construction) and register / work-item shapes (correct results).  Never part of the product build.
int run_mimo_tdl_wave_f32_experiment(int code, MCLE_MIMO_TDL_WAVE_ARGS) {
    if (nt != 4 || nr != 4 || pp.K != 2) return MCLE_E_UNSUPPORTED;
#define MCLE_EXP(CODE_, BQ_, WPS_, ABL_)                                                                                      \
    if (code == CODE_)                                                                                                        \
        return launch_mimo_tdl_wave<float, 1024, 4, 4, 2, BQ_, WPS_, ABL_>(ctx, pp, method, seed, first, count, d_counters, d_sym, d_bit);
    MCLE_EXP(1, 2, 3, 1) MCLE_EXP(2, 2, 3, 2) MCLE_EXP(4, 2, 3, 4) MCLE_EXP(8, 2, 3, 8) MCLE_EXP(16, 2, 3, 16) MCLE_EXP(31, 2, 3, 31)
    MCLE_EXP(40, 2, 2, 0) MCLE_EXP(41, 1, 3, 0) MCLE_EXP(42, 2, 4, 0) MCLE_EXP(43, 2, 3, 0) MCLE_EXP(44, 2, 3, 128) MCLE_EXP(45, 2, 2, 128) MCLE_EXP(64, 2, 3, 64)
    MCLE_EXP(46, 2, 3, 512) MCLE_EXP(47, 2, 3, 1024) MCLE_EXP(48, 2, 3, 2048) MCLE_EXP(49, 2, 3, 3584) MCLE_EXP(50, 2, 3, 3584 + 8) MCLE_EXP(51, 2, 3, 3584 + 8 + 4 + 16)
#undef MCLE_EXP
    return MCLE_E_UNSUPPORTED;
}
#endif

}  // namespace mcle
