// pipeline_mimo_tdl_wave_f64_512.hip -- the one-receive-antenna-per-wavefront kernels of the frequency-selective MIMO-OFDM link (mimo_tdl_wave.hpp) in
// complex128: fft_size 512, run-time polynomial order; every 1 <= Nt <= Nr <= 4
#include "mimo_tdl_wave.hpp"

namespace mcle {

MCLE_MIMO_TDL_WAVE_TU(run_mimo_tdl_wave_f64_512, double, 512, 0)

}  // namespace mcle
