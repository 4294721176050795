// pipeline_mimo_tdl_wave_f32_2048k.hip -- the one-receive-antenna-per-wavefront kernels of the frequency-selective MIMO-OFDM link (mimo_tdl_wave.hpp) in
// complex64: fft_size 2048, polynomial order 2 parked in registers (the benchmark's Doppler per symbol; round 6: until then only the 1024
// kernels had this form, and the run-time-order kernels of the other sizes issued 1.4 x the vector and 5 x the scalar instructions
// per subcarrier -- profiles/r06/f1_pmc.log); every 1 <= Nt <= Nr <= 4
#include "mimo_tdl_wave.hpp"

namespace mcle {

MCLE_MIMO_TDL_WAVE_TU(run_mimo_tdl_wave_f32_2048k, float, 2048, mimo_tdl_wave_kf<float>())

}  // namespace mcle
