// pipeline_mimo_tdl_wave_f64_512k.hip -- the one-receive-antenna-per-wavefront kernels of the frequency-selective MIMO-OFDM link (mimo_tdl_wave.hpp) in
// complex128: fft_size 512, polynomial order 5 parked in registers (the benchmark's Doppler per symbol; round 6: until then only the 1024
// kernels had this form, and the run-time-order kernels of the other sizes issued 1.4 x the vector and 5 x the scalar instructions
// per subcarrier -- profiles/r06/f1_pmc.log); every 1 <= Nt <= Nr <= 4
#include "mimo_tdl_wave.hpp"

namespace mcle {

MCLE_MIMO_TDL_WAVE_TU(run_mimo_tdl_wave_f64_512k, double, 512, mimo_tdl_wave_kf<double>())

}  // namespace mcle
