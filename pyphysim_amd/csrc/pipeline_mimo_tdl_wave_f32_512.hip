// pipeline_mimo_tdl_wave_f32_512.hip -- the one-receive-antenna-per-wavefront kernels of the frequency-selective MIMO-OFDM link (mimo_tdl_wave.hpp) in
// complex64: fft_size 512, run-time polynomial order; every 1 <= Nt <= Nr <= 4
#include "mimo_tdl_wave.hpp"

namespace mcle {

MCLE_MIMO_TDL_WAVE_TU(run_mimo_tdl_wave_f32_512, float, 512, 0)

}  // namespace mcle
