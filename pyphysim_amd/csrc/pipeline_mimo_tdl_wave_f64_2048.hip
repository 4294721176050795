// pipeline_mimo_tdl_wave_f64_2048.hip -- the one-receive-antenna-per-wavefront kernels of the frequency-selective MIMO-OFDM link (mimo_tdl_wave.hpp) in
// complex128: fft_size 2048, run-time polynomial order; every 1 <= Nt <= Nr <= 4
#include "mimo_tdl_wave.hpp"

namespace mcle {

MCLE_MIMO_TDL_WAVE_TU(run_mimo_tdl_wave_f64_2048, double, 2048, 0)

}  // namespace mcle
