// pipeline_mimo_tdl_wave_f32_1024k.hip -- the one-receive-antenna-per-wavefront kernels of the frequency-selective MIMO-OFDM link (mimo_tdl_wave.hpp) in
// complex64: fft_size 1024, polynomial order 2 parked in registers (the benchmark's); every 1 <= Nt <= Nr <= 4
#include "mimo_tdl_wave.hpp"

namespace mcle {

MCLE_MIMO_TDL_WAVE_TU(run_mimo_tdl_wave_f32_1024k, float, 1024, mimo_tdl_wave_kf<float>())

}  // namespace mcle
