/*
 * mcle.h -- C ABI of libmcle: the MI355X (gfx950) Monte Carlo link-level engine.
 *
 * The reference (darcamo/pyphysim v0.7.2) is pure Python/NumPy and has no FFI; its
 * "operator interface" for the hot path is a set of Python method signatures.  Each entry
 * point below names the reference method it stands in for (paths relative to the reference).
 * A maintainer-side ctypes binding is sketched in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (MCLE_E_*); mcle_last_error() gives the
 *     message for the calling thread.  No exception crosses the ABI.
 *   - `d_` arguments are DEVICE pointers (hipMalloc'ed by anyone: mcle_malloc, PyTorch, ...);
 *     everything else is host memory or passed by value.
 *   - complex arrays are interleaved (re, im) of the context dtype: MCLE_F32 -> float,
 *     MCLE_F64 -> double (the parity instantiation; reference arithmetic is complex128).
 *   - symbol indices are int32 on the device (the reference uses int64 on the host).
 *   - all work is enqueued on the context's stream; only mcle_ctx_sync / mcle_memcpy_d2h and
 *     the *_host helpers block.  A context is not thread-safe; distinct contexts are.
 */
#ifndef MCLE_H
#define MCLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCLE_VERSION 1

enum { MCLE_OK = 0, MCLE_E_INVAL = -1, MCLE_E_HIP = -2, MCLE_E_NOMEM = -3, MCLE_E_STATE = -4,
       MCLE_E_UNSUPPORTED = -5 /* valid request outside a fused kernel's envelope and no generic kernel behind the same entry
                                  point: returned by mcle_run_mimo_ofdm_tdl only (compose the staged operators); every other
                                  mcle_run_* falls back internally or reports its envelope as MCLE_E_INVAL */ };
enum { MCLE_F32 = 0, MCLE_F64 = 1 };
/* demodulation method: exhaustive minimum distance over the constellation held in LDS
 * (any constellation), or the per-axis slicer (square Gray QAM only; same decisions). */
enum { MCLE_DEMOD_MINDIST = 0, MCLE_DEMOD_QAM_SLICER = 1 };
/* constellation kinds (tell the library what structure it may exploit) */
enum { MCLE_CONST_GENERIC = 0, MCLE_CONST_QAM = 1, MCLE_CONST_BPSK = 2 };

typedef struct mcle_ctx mcle_ctx;

/* Integer result block of one parameter variation.  Every field is an exact integer sum over
 * realizations, so reductions are order-independent and identical for any GPU count.  It carries
 * what Result.update()/merge() accumulate (simulations/results.py:469-623): value = sum of
 * errors, total = n_realizations * units per realization, sum / sum-of-squares of the
 * per-realization ratio = {sum, sum_sq} / units^k. */
typedef struct mcle_counters {
    uint64_t n_realizations;
    uint64_t n_skipped;        /* SkipThisOne analogue (runner.py:151-185): singular filter etc. */
    uint64_t sym_errors;       /* sum_r e_r          */
    uint64_t sym_errors_sq;    /* sum_r e_r^2        */
    uint64_t bit_errors;
    uint64_t bit_errors_sq;
    uint64_t n_symbols;        /* symbols per realization (constant)  */
    uint64_t n_bits;           /* bits per realization (constant)     */
} mcle_counters;

/* ---- context, stream, memory -------------------------------------------------------- */
const char* mcle_last_error(void);
int mcle_version(void);
int mcle_device_count(int* count);
int mcle_ctx_create(int device_id, mcle_ctx** out);
int mcle_ctx_destroy(mcle_ctx* ctx);
/* adopt an external hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = own */
int mcle_ctx_set_stream(mcle_ctx* ctx, void* hip_stream);
int mcle_ctx_get_stream(mcle_ctx* ctx, void** hip_stream);
int mcle_ctx_sync(mcle_ctx* ctx);
/* Releases the context's scratch buffer (fading / filter records of the two-launch pipelines: up to 138 MiB for the
 * complex128 MIMO-OFDM family, 256 MiB + 25 % for mcle_run_mimo_ofdm_tdl; it otherwise lives until mcle_ctx_destroy).
 * Waits for the stream first.  The next call that needs it allocates again. */
int mcle_ctx_trim_scratch(mcle_ctx* ctx);
int mcle_ctx_device_info(mcle_ctx* ctx, int* n_cu, int* lds_bytes, char* name, int name_len);
int mcle_malloc(mcle_ctx* ctx, size_t bytes, void** d_ptr);
int mcle_free(mcle_ctx* ctx, void* d_ptr);
int mcle_memset(mcle_ctx* ctx, void* d_ptr, int value, size_t bytes);
int mcle_memcpy_h2d(mcle_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int mcle_memcpy_d2h(mcle_ctx* ctx, void* dst, const void* d_src, size_t bytes); /* blocks */
/* device-to-device strided row copy (rows of row_bytes; pitches in bytes), enqueued on the stream */
int mcle_memcpy_2d(mcle_ctx* ctx, void* d_dst, size_t dst_pitch, const void* d_src, size_t src_pitch,
                   size_t row_bytes, size_t rows);
/* Per-context kernel-selection options (round 3: they used to be environment variables read inside the launch
 * path, which made a context's behaviour depend on the caller's environment).  Every option is an integer, 0 is the
 * default, and none of them changes a result beyond the rounding-level differences documented per kernel:
 * they exist for A/B measurements and for tests that compare the matrix-core kernels with the VALU kernels they
 * replace.  set: MCLE_E_INVAL for an unknown option or a value outside its range. */
enum {
    MCLE_OPT_NO_MFMA = 0,          /* 1: every fused pipeline / operator runs its VALU kernel (no matrix-core form) */
    MCLE_OPT_MFMA_VARIANT = 1,     /* config-4 matrix-core kernel variant: 0 (= 36), 36, 32, 30, 21 */
    MCLE_OPT_GRID_OVERSUB = 2,     /* persistent grids = this multiple of the resident set; 0: automatic (<= 8) */
    MCLE_OPT_FLAT_WGS_PER_CU = 3,  /* single-carrier kernels: workgroups started per CU; 0: 64 */
    MCLE_OPT_SINGLE_TDL = 4,       /* 1: config 3 on the one-realization-per-workgroup kernel */
    MCLE_OPT_TDL_MFMA_WAVES = 5,   /* config-3 matrix-core kernel: 0 / 2 = two waves per SIMD, 3 = three, 32 = three with two
                                      realizations per pass instead of four */
    MCLE_OPT_JAKES_DIRECT = 6,     /* 1: one sincos per ray and sample (jakes_generate: k_jakes; complex128 flat-fading pipeline: no rotation recurrence) */
    MCLE_OPT_F64_GENERIC = 7,      /* 1: config 4 on the generic radix-4 kernel instead of the planar family (either arithmetic) */
    MCLE_OPT_F64_THREADS = 8,      /* complex128 config-4 kernel at (1024, 4x4): 0 = the quarter-wave kernel (round 6,
                                      csrc/pipeline_mimo_qw.hip: a wavefront owns one time class n mod 4 of all antennas, samples in
                                      registers between radix-16 passes, three workgroups per CU, the channel contraction on
                                      v_mfma_f64_4x4x4) wherever its envelope holds -- full band, even cyclic prefix, decisions by
                                      slicer or certificate -- and the planar radix-16 form elsewhere (default); 260 = the same,
                                      explicit; 262 = quarter-wave bounded for two wavefronts per SIMD; 261 = the planar radix-16
                                      form (one transform per wavefront, the channel fused with the span-1 butterflies: the default
                                      of rounds 4-5); 257 = that with a separate channel stage; 258 = 261 with every layer-1 twiddle
                                      from the table;
                                      512 = radix-4 stages, two antennas per thread, 512 threads; 256 = radix-4 stages, four antennas
                                      per thread, 256 threads (all of them: same results contract; A/B times in DESIGN.md 5.5 / 5.10).
                                      complex64 (the planar family on planes of floats): 0 = radix-16 passes with a SEPARATE channel
                                      stage at a four-wavefront register bound (default); 257 = the same at a three-wavefront
                                      bound; 259 = the fused channel stage; 512 / 256 as above.
                                      fft_size 2048 with four receive antennas and Nt >= 3: 512 = four antennas per thread (512 threads,
                                      nothing spilled; the default where it is the faster form), 1024 = two antennas per thread.
                                      complex128 at (256, 4x4) and (512, 4x4), same envelope: 0 / 260 = the full-wave kernel (256:
                                      csrc/pipeline_mimo_fw.hip, one realization per wavefront) / the half-wave kernel (512:
                                      csrc/pipeline_mimo_pw.hip, two wavefronts per realization), both with channel AND decode on
                                      v_mfma_f64_4x4x4; 262 = bounded for two wavefronts per SIMD; 261 = the planar radix-4 form of
                                      rounds 3-5; (256, 2x2): the full-wave kernel with two realizations per wavefront; (2048, 4x4): the
                                      eighth-wave kernel (512 threads).  (1024, 4x4): since the end of round 6 the DEFAULT (0) is the
                                      quarter-wave decomposition with the decode on the matrix cores as well (pipeline_mimo_pw.hip,
                                      NW = 4; 263 = the same, explicit; 264 = two wavefronts per SIMD); 260 / 262 select the first
                                      quarter-wave kernel (pipeline_mimo_qw.hip, VALU decode) */
    MCLE_OPT_BD_RUNTIME_SOLVE = 9, /* 1: the block-diagonalisation pipeline solves with the run-time-sized routine (private
                                      arrays in scratch) also where the compile-time-sized one (K nr <= 6) applies */
    MCLE_OPT_DEMOD_NOCERT = 10,    /* 1: min-distance decisions of a square Gray QAM always through the table search (candidate
                                      grid / sweep); 0: through the margin certificate of modem.hpp (demod_qam_cert: the
                                      closed-form nearest level per axis, accepted when the received point is farther than
                                      2^-30 (complex64: 2^-15) of a level spacing from every decision boundary -- and, in
                                      complex64, no farther than 16 spacings outside the outermost level -- the table search
                                      otherwise: no table gathers, and the decisions of the exhaustive sweep by a margin argument
                                      that holds for |re|, |im| up to ~2^11 level spacings in complex128 (beyond that -- an
                                      equaliser output in a fade of -66 dB -- the identity rests on test coverage: the chance that
                                      such a point also sits within 2^-30 of a boundary is ~1e-14 per symbol)); likewise for a
                                      four-point constellation with one point per quadrant at (+-a, +-b) -- QPSK -- decided by the
                                      signs (demod_quad_cert: certified for 2^-30 min(a, b) <= |re|, |im| <= 2^8 max(a, b)), and for
                                      8- / 16-PSK inside the table search itself (demod_psk_cert: the sector by sign masks and one
                                      compare per sector boundary of an octant, certified at an angular margin of 2^-28
                                      (complex64: 2^-12) inside a magnitude window of 2^-8 .. 2^8 (1/8 .. 8) radii) */
    MCLE_OPT_F64_VARIANT = 11,     /* (builds with -DMCLE_EXPERIMENTS only; the product library accepts 0 and refuses anything else)
                                      complex128 config-4 kernel (1024, 4x4), TIMING BOUNDS ONLY on its 512-thread radix-4 form --
                                      results are wrong by construction: bit 0 = the LDS stores of the last transmit stage and of
                                      the channel stage dropped, bit 1 = the two workgroup barriers around the channel stage
                                      dropped */
    MCLE_OPT_F32_MFMA = 12,        /* complex64 config 4 at (1024, 4x4): 1 = the matrix-core kernel k_run_mimo_ofdm_mfma (the default of
                                      rounds 2-3) instead of the planar VALU family (k_run_mimo_ofdm_planar<float>: the complex128
                                      kernels on planes of floats, 10-20 % faster at this geometry and the only fast complex64 kernel
                                      at every other one; default since round 4) */
    MCLE_OPT_TDL_KERNEL = 13,      /* config 3 at fft_size 256 / 512 / 1024 / 2048 with <= 8 taps reaching <= 256 samples back (inside the cyclic prefix or,
                                      since round 6, beyond it; polynomial order <= 8): 0 = one realization per WAVEFRONT (k_run_ofdm_tdl_wave: no
                                      workgroup barrier in the loop; radix-16 register passes at 1024, radix-4 stages otherwise) WHERE IT
                                      IS THE FASTER KERNEL (1024; 2048 in complex64; 256 / 512 in complex128 -- default since round 4),
                                      1 = the batched kernels of rounds 1-3 everywhere (four / two realizations per workgroup pass;
                                      complex64 at 1024: matrix cores), 2 = the wavefront kernel wherever it exists, 3 = the same, but at
                                      2048 points the one-wavefront kernel instead of the round-6 default there, TWO wavefronts per
                                      realization (k_run_ofdm_tdl_hw: every delay inside the prefix, orders 2 .. 5; A/B), 4 = as 2 with
                                      the complex64 registers at 1024 bounded for four wavefronts per SIMD instead of three (A/B) */
    MCLE_OPT_MIMO_TDL_KERNEL = 14, /* frequency-selective MIMO-OFDM (mcle_run_mimo_ofdm_tdl) at fft_size 256 / 512 / 1024 / 2048 with <= 8 taps
                                      reaching <= min(256, fft_size / 2) samples back (inside the cyclic prefix or, since round 6, beyond it): 0 = one receive antenna per WAVEFRONT
                                      (k_run_mimo_ofdm_tdl_wave, every 1 <= Nt <= Nr <= 4; default since round 5), 1 = the
                                      workgroup-cooperative kernel of rounds 1-4 (Nt = Nr in {2, 4}), 2 = the wavefront kernels with the
                                      tap polynomials' order at run time also where the parked-coefficient kernel applies (A/B) */
    MCLE_OPT_WALK_LEGACY = 15,     /* symbol walks of mcle_run_ia / mcle_run_bd (either arithmetic) with an even number of columns >= 128 (and, for
                                      block diagonalisation, two or three users of <= 2 antennas): 0 = the packed walk of round 6
                                      (csrc/walk_f64.hpp: the lane pairs of a chunk of realizations as one index space, decision form
                                      fixed at compile time, records in LDS), 1 = the per-realization walks of rounds 2-5 (A/B and
                                      kernel-vs-kernel tests: identical counters) */
    MCLE_OPT_COUNT = 16
};
int mcle_ctx_set_option(mcle_ctx* ctx, int option, long long value);
int mcle_ctx_get_option(mcle_ctx* ctx, int option, long long* value);

/* kernel timing on the context stream with HIP events (used by bench.py's roofline leg) */
int mcle_timer_start(mcle_ctx* ctx);
int mcle_timer_stop_ms(mcle_ctx* ctx, float* ms);                               /* blocks */

/* What the box's HBM delivers to a streaming kernel, measured by the library (csrc/kernels_hbm.hip): `reps` launches of a
 * 16-byte-per-access grid-stride kernel over arrays of `bytes` bytes each (in the context's scratch), n_cu * blocks_per_cu
 * workgroups of 256 threads; *gbps = bytes moved (read + written) per second / 1e9.  The denominator of the
 * "fraction of the achievable HBM rate" figures of bench.py (both the 8 TB/s specification and the rate
 * measured on the box are quoted).  Nothing in the reference corresponds to it. */
enum { MCLE_HBM_COPY = 0, MCLE_HBM_READ = 1, MCLE_HBM_TRIAD = 2, MCLE_HBM_WRITE = 3,
       MCLE_HBM_NONTEMPORAL = 4 /* | : non-temporal loads / stores */, MCLE_HBM_UNROLL8 = 8 /* | : eight accesses in flight per thread instead of four */ };
int mcle_hbm_stream_rate(mcle_ctx* ctx, int kind, size_t bytes, int reps, int blocks_per_cu, double* gbps);

/* ---- multi-GPU: realization sharding + ONE all-reduce of the integer counters.
 *      The reference's only multi-process mechanism is ipyparallel, one parameter variation per engine
 *      (simulations/runner.py:1836-1846 simulate_in_parallel); here every rank takes a slice of every
 *      variation (realizations are addressed by index) and the exact integer sums are reduced over RCCL / xGMI.
 *      One process and one context per GPU.  RCCL is bound at run time (dlopen), so a host without it can still
 *      load the library; mcle_comm_load names the librccl to use (NULL: librccl.so.1 as the loader finds it --
 *      the copy PyTorch has loaded, if any). ------------------------------------------------------------------ */
int mcle_comm_load(const char* rccl_path);
/* rank 0 draws the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means (a socket, a file,
 * MPI, torch.distributed ...); every rank then joins.  world == 1 is allowed (all-reduces become no-ops).
 * Draw an id only to use it: RCCL starts a bootstrap listener per id that ends when all ranks have joined. */
int mcle_comm_unique_id(void* id_out, size_t bytes);
int mcle_comm_init(mcle_ctx* ctx, const void* unique_id, int rank, int world);
int mcle_comm_destroy(mcle_ctx* ctx);
int mcle_comm_info(mcle_ctx* ctx, int* rank, int* world);
/* in place, on the context stream: words 0..5 of each of the n blocks are summed over the ranks
 * (ncclUint64 / ncclSum -- exact, order independent), n_symbols / n_bits take the maximum (they are
 * per-realization constants, zero on a rank whose shard was empty).  No-op without a communicator. */
int mcle_counters_allreduce(mcle_ctx* ctx, mcle_counters* d_counters, int n);
/* sum of n doubles over the ranks, in place (the 'sum_capacity' style side results of the IA application) */
int mcle_allreduce_f64(mcle_ctx* ctx, double* d_values, size_t n);

/* ---- constellation (a1: modulators/fundamental.py:131-146 setConstellation, :396-448 PSK,
 *      :659-777 QAM; the table itself is built by the host mirror) ----------------------- */
int mcle_set_constellation(mcle_ctx* ctx, const double* re_im, int M, int kind);

/* Host only (no device needed): the candidate grid the f32 min-distance kernels search instead of all M points
 * (DESIGN.md section 5.1).  cells [32*32] receives G*G words: byte 0 = number of candidates (0xFF: search
 * everything), bytes 1..7 = candidate indices in ascending order; cell (ix, iy) covers
 * [x0 + ix*h, x0 + (ix+1)*h) x [y0 + iy*h, ...), border cells extending to infinity.  2 <= M <= 256. */
int mcle_build_demod_grid(const double* re_im, int M, int* G, double* x0, double* y0, double* h,
                          unsigned long long* cells);

/* ---- a2/a3/a4: Modulator.modulate / demodulate (fundamental.py:175-248), count_bit_errors
 *      (util/misc.py:519-566) and the symbol-error count of user code --------------------- */
int mcle_modulate(mcle_ctx* ctx, int dtype, const int32_t* d_idx, void* d_out, size_t n);
int mcle_demodulate(mcle_ctx* ctx, int dtype, int method, const void* d_rx, int32_t* d_idx,
                    size_t n);
/* n_real blocks of n_per_real indices each; adds into d_counters[0] (may be NULL) and, when
 * non-NULL, writes per-realization counts d_sym_err[n_real] / d_bit_err[n_real]. */
int mcle_count_errors(mcle_ctx* ctx, const int32_t* d_tx_idx, const int32_t* d_rx_idx,
                      size_t n_per_real, size_t n_real, int bits_per_symbol,
                      mcle_counters* d_counters, uint32_t* d_sym_err, uint32_t* d_bit_err);
/* fused demodulate + count (no index array written) */
int mcle_demod_count(mcle_ctx* ctx, int dtype, int method, const void* d_rx,
                     const int32_t* d_tx_idx, size_t n_per_real, size_t n_real,
                     mcle_counters* d_counters, uint32_t* d_sym_err, uint32_t* d_bit_err);
/* the same against byte labels (mcle_rand_modulate_batch_u8) */
int mcle_demod_count_u8(mcle_ctx* ctx, int dtype, int method, const void* d_rx,
                        const uint8_t* d_tx_idx, size_t n_per_real, size_t n_real,
                        mcle_counters* d_counters, uint32_t* d_sym_err, uint32_t* d_bit_err);

/* ---- a5: randn_c (util/misc.py:327-355) under the mcle-philox-v1 contract, and AWGN ----- */
/* out[i] = sqrt(variance) * CN(0,1) sample (first_sample + i) of (seed, realization, stream) */
int mcle_randn_c(mcle_ctx* ctx, int dtype, uint64_t seed, uint64_t realization, uint32_t stream,
                 uint64_t first_sample, double variance, void* d_out, size_t n);
int mcle_rand_symbols(mcle_ctx* ctx, uint64_t seed, uint64_t realization, uint64_t first_symbol,
                      int M, int32_t* d_idx, size_t n);
/* y = x + sqrt(noise_var) * noise (injected noise array: parity against golden vectors) */
int mcle_awgn_add(mcle_ctx* ctx, int dtype, const void* d_x, const void* d_noise,
                  double noise_var, void* d_y, size_t n);

/* ---- a6/a9: JakesSampleGenerator (channels/fading_generators.py:289-553) and
 *      TdlChannel.corrupt_data (channels/fading.py:1046-1124) --------------------------- */
/* h[s, n] = L^-1/2 sum_l exp(j(2 pi Fd cos(phi[l,s]) t_n + psi[l,s])) * sqrt(tap_power[s]),
 * t_n = t0 + n*dt.  phi/psi: host doubles [L, n_streams]; tap_power: host [n_streams] or NULL. */
int mcle_jakes_generate(mcle_ctx* ctx, int dtype, const double* phi, const double* psi, int L,
                        int n_streams, double Fd, double t0, double dt, const double* tap_power,
                        void* d_h, size_t n_samples);
/* same, at explicit sample times (block-static use of TdlChannel.corrupt_data_in_freq_domain) */
int mcle_jakes_generate_at(mcle_ctx* ctx, int dtype, const double* phi, const double* psi, int L,
                           int n_streams, double Fd, const double* times, const double* tap_power,
                           void* d_h, size_t n_samples);
/* SISO time-varying sparse convolution: y[d_i + n] += g[i, n] x[n]; y has n + max_delay */
int mcle_tdl_apply(mcle_ctx* ctx, int dtype, const void* d_x, const void* d_taps,
                   const int32_t* delays, int n_taps, void* d_y, size_t n);
/* MIMO branch of corrupt_data (fading.py:1107-1117): x [nt][n], taps [n_taps][nr][nt][n] ->
 * y [nr][n + max_delay] */
int mcle_tdl_apply_mimo(mcle_ctx* ctx, int dtype, const void* d_x, const void* d_taps,
                        const int32_t* delays, int n_taps, int nr, int nt, void* d_y, size_t n,
                        size_t batch);
/* per-OFDM-symbol mean frequency response on the used subcarriers for n_links parallel links:
 * taps [n_taps][n_links][n_sym*(fft+cp)] -> H [n_sym][num_used][n_links]
 * (TdlImpulseResponse.get_freq_response fading.py:513-536 averaged like ofdm.py:545-547).
 * num_used = -g (g >= 1): all fft_size bins in natural order, averaging groups of g samples
 * (taps [n_taps][n_links][n_sym*g]); g = 1 is the per-block response of corrupt_data_in_freq_domain. */
int mcle_tdl_mean_freq_response(mcle_ctx* ctx, int dtype, const void* d_taps, const int32_t* delays,
                                int n_taps, int n_links, size_t n_sym, int fft_size, int cp_size,
                                int num_used, void* d_H, size_t batch);
/* batched Jakes taps with the phases drawn on-chip (PHASE stream of realization first + r, draw order of
 * fading_generators.py:421-425): taps [count][n_streams][n]; stream_amp = sqrt(tap power / L) per stream */
int mcle_jakes_taps_philox(mcle_ctx* ctx, int dtype, uint64_t seed, uint64_t first, uint64_t count,
                           int L, int n_streams, double Fd, double t0, double dt,
                           const double* stream_amp, void* d_taps, size_t n_samples);
/* y[r][i] = x[r][i] + sqrt(noise_var) * CN(0,1) sample i of (seed, first + r, NOISE); rows of row_len */
int mcle_awgn_philox(mcle_ctx* ctx, int dtype, const void* d_x, uint64_t seed, uint64_t first,
                     uint64_t count, size_t row_len, double noise_var, void* d_y);
/* out[r][i] = sqrt(variance) * CN(0,1) sample i of (seed, first + r, stream): mcle_randn_c for a batch of realizations
 * (e.g. the flat channel matrices H = randn_c(Nr, Nt) of count realizations from the CHAN stream,
 * apps/mimo/simulate_mimo.py:78) */
int mcle_randn_c_batch(mcle_ctx* ctx, int dtype, uint64_t seed, uint64_t first, uint64_t count,
                       uint32_t stream, size_t row_len, double variance, void* d_out);
/* d_idx[r][i] = symbol i of realization first_realization + r (DATA stream) */
int mcle_rand_symbols_batch(mcle_ctx* ctx, uint64_t seed, uint64_t first_realization, uint64_t count,
                            int M, int32_t* d_idx, size_t n);
/* mcle_rand_symbols_batch + mcle_modulate in one pass (the "gen + modulate" operator of SURVEY 8(d)'s staged model, one
 * write of the labels and one of the samples): d_idx[r][i] as above for the bound constellation's M, d_sym[r][i] =
 * table[d_idx[r][i]] (randint + Modulator.modulate, modulators/fundamental.py:175-199) */
int mcle_rand_modulate_batch(mcle_ctx* ctx, int dtype, uint64_t seed, uint64_t first_realization, uint64_t count,
                             int32_t* d_idx, void* d_sym, size_t n);
/* the same with BYTE labels (M <= 256): SURVEY 8(d)'s staged model counts an index as one byte; the int32 form above moves
 * four.  Consumed by mcle_demod_count_u8. */
int mcle_rand_modulate_batch_u8(mcle_ctx* ctx, int dtype, uint64_t seed, uint64_t first_realization, uint64_t count,
                                uint8_t* d_idx, void* d_sym, size_t n);
/* element-wise complex product (frequency-domain channel application, fading.py:1259) */
int mcle_cmul(mcle_ctx* ctx, int dtype, const void* d_a, const void* d_b, void* d_out, size_t n);
/* element-wise complex divide (flat-fading equalisation y / h of the C2 template) */
int mcle_cdiv(mcle_ctx* ctx, int dtype, const void* d_num, const void* d_den, void* d_out,
              size_t n);

/* ---- a11/a10: OFDM.modulate / demodulate (modulators/ofdm.py:394-466) and
 *      OfdmOneTapEqualizer.equalize_data (ofdm.py:515-552) ------------------------------- */
/* n_in data symbols (zero padded to n_sym*num_used) -> n_sym*(fft+cp) samples; batch rows */
int mcle_ofdm_modulate(mcle_ctx* ctx, int dtype, const void* d_in, size_t n_in, int fft_size,
                       int cp_size, int num_used, void* d_out, size_t batch);
/* n_sym*(fft+cp) samples -> n_sym*num_used symbols; batch rows */
int mcle_ofdm_demodulate(mcle_ctx* ctx, int dtype, const void* d_in, size_t n_sym, int fft_size,
                         int cp_size, int num_used, void* d_out, size_t batch);
/* d_taps [n_taps, n_sym*(fft+cp)] sparse taps of the samples the OFDM symbols rode on */
int mcle_onetap_equalize(mcle_ctx* ctx, int dtype, const void* d_data, const void* d_taps,
                         const int32_t* delays, int n_taps, size_t n_sym, int fft_size,
                         int cp_size, int num_used, void* d_out);

/* ---- a12: Blast (mimo/mimo.py:465-660) --------------------------------------------------- */
/* x[n] -> X[a, c] = x[c*nt + a] / sqrt(nt)   (encode, :639-641); batch rows of n */
int mcle_blast_encode(mcle_ctx* ctx, int dtype, const void* d_x, int nt, size_t n, void* d_X,
                      size_t batch);
/* G = sqrt(nt) * solve(H^H H + noise_var I, H^H) per batch item (noise_var = 0: ZF / pinv for
 * full column rank); d_H [batch, nr, nt] row-major, d_G [batch, nt, nr]; d_skipped[batch]
 * flags singular systems (may be NULL). */
int mcle_blast_filter(mcle_ctx* ctx, int dtype, const void* d_H, int nr, int nt,
                      double noise_var, void* d_G, uint32_t* d_skipped, size_t batch);
/* est[c*nt + a] = sum_r G[a, r] Y[r, c]      (decode, :658-660) */
int mcle_blast_decode(mcle_ctx* ctx, int dtype, const void* d_G, const void* d_Y, int nr, int nt,
                      size_t ns, void* d_est, size_t batch);
/* one receive filter per column (per-subcarrier MMSE): G [b][ns][nt][nr], Y [b][nr][ns] -> est[b][c*nt + a] */
int mcle_blast_decode_per_subcarrier(mcle_ctx* ctx, int dtype, const void* d_G, const void* d_Y,
                                     int nr, int nt, size_t ns, void* d_est, size_t batch);
/* Y = H X  (+ sqrt(noise_var) * noise when d_noise != NULL): apps/mimo/simulate_mimo.py:96-98 */
int mcle_mimo_channel(mcle_ctx* ctx, int dtype, const void* d_H, const void* d_X,
                      const void* d_noise, double noise_var, int nr, int nt, size_t ns, void* d_Y,
                      size_t batch);
/* the same with the noise drawn on-chip: Y[b][r][c] = sum_a H[b][r][a] X[b][a][c] + sqrt(noise_var) * CN(0,1)
 * sample r*ns + c of (seed, first + b, NOISE) -- the (Nr, Ns) row-major randn_c of simulate_mimo.py:97 under the
 * mcle-philox-v1 contract; one pass over X and Y (the "H T + awgn" operator of SURVEY.md 8(d)'s staged model) */
int mcle_mimo_channel_philox(mcle_ctx* ctx, int dtype, const void* d_H, const void* d_X, uint64_t seed,
                             uint64_t first, double noise_var, int nr, int nt, size_t ns, void* d_Y,
                             size_t batch);

/* ---- a13: further MIMO schemes of mimo/mimo.py ---------------------------------------------- */
/* Alamouti (mimo.py:1168-1269): x [batch][n] -> X [batch][2][n] (n even); decode with H [batch][nr][2] */
int mcle_alamouti_encode(mcle_ctx* ctx, int dtype, const void* d_x, size_t n, void* d_X,
                         size_t batch);
int mcle_alamouti_decode(mcle_ctx* ctx, int dtype, const void* d_H, const void* d_Y, int nr, size_t n,
                         void* d_out, size_t batch);
/* MRT (mimo.py:666-783), MISO h [batch][nt]: X = exp(-1j angle(h)).T / sqrt(nt) * x; decode y * sqrt(nt)/sum|h| */
int mcle_mrt_encode(mcle_ctx* ctx, int dtype, const void* d_h, const void* d_x, int nt, size_t n,
                    void* d_X, size_t batch);
int mcle_mrt_decode(mcle_ctx* ctx, int dtype, const void* d_h, const void* d_y, int nt, size_t n,
                    void* d_out, size_t batch);
/* SVDMimo (mimo.py:833-946), square H [batch][n][n]: W = V/sqrt(n) (precoder), G = diag(1/S) U^H sqrt(n)
 * (receive filter), S descending (d_S may be NULL).  encode = W @ x.reshape(n,-1), decode = G @ Y via
 * mcle_mimo_channel. */
int mcle_svd_filters(mcle_ctx* ctx, int dtype, const void* d_H, int n_ant, void* d_W, void* d_G,
                     double* d_S, size_t batch);

/* GMDMimo (mimo.py:952-1067) + util.misc.gmd (misc.py:18-159), square H: W = P/sqrt(n), G = Blast filter
 * (ZF for noise_var = 0, else MMSE) on the equivalent channel Q R; d_R [batch][n][n] (real upper
 * triangular with constant diagonal; may be NULL). */
int mcle_gmd_filters(mcle_ctx* ctx, int dtype, const void* d_H, int n_ant, double noise_var, void* d_W,
                     void* d_G, double* d_R, uint32_t* d_skipped, size_t batch);
/* calc_post_processing_linear_SINRs (mimo/mimo.py:62-118; MimoBase.calc_linear_SINRs :311-345): complex128
 * H [batch][nr][nt], precoder W [batch][nt][ns], receive filter G_H [batch][ns][nr] -> linear SINR [batch][ns]:
 * |E_ii|^2 / (|sum_{j != i} E_ij|^2 + noise_var ||G_H row i||^2) with E = G_H H W (the reference's formula,
 * modulus of the summed off-diagonal entries included). */
int mcle_post_processing_sinrs(mcle_ctx* ctx, const void* d_H, const void* d_W, const void* d_G, double noise_var,
                               int nr, int nt, int ns, double* d_sinr, size_t batch);

/* ---- K-user interference channel: covariance matrices and post-filter SINRs ------------------
 * MultiUserChannelMatrix / MultiUserChannelMatrixExtInt (channels/multiuser.py): calc_Q / calc_JP_Q (:1314-1450,
 * :2530-2634), _calc_Bkl_cov_matrix_all_l and its JP form (:1452-1826, :2676-2742), calc_SINR / calc_JP_SINR
 * (:1828-2008, :2636-2807), calc_cov_matrix_extint_plus_noise (:2469-2520), path loss on the block matrix (:1264-1312).
 * complex128.  d_bigH [batch][sum nr][sum nt + n_ext]; d_pathloss (may be NULL) [sum nr][sum nt + n_ext] LINEAR power
 * ratio per entry (the reference's _pathloss_big_matrix; shared by the batch); d_F [batch][K][16][4]: user j's
 * precoder in the top-left corner (nt[j] x ns[j], or (sum nt) x ns[j] when `joint`); d_U [batch][K][4][4]
 * (nr[k] x ns[k], may be NULL).  Outputs (each may be NULL), zero padded: d_Q [batch][K][4][4] interference from
 * the other users + Re_k; d_Re [batch][K][4][4] = pe ext ext^H + noise_var I; d_B [batch][K][4][4][4] (stream l:
 * everything received minus stream l's own covariance); d_sinr [batch][K][4] (linear). */
typedef struct mcle_mu_stats_cfg {
    int32_t K;                  /* users (with receive antennas), <= 4 */
    int32_t n_ext;              /* external interferer antennas: trailing columns of big_H, <= 8 */
    int32_t joint;              /* 0: per-link precoders (calc_Q, calc_SINR); 1: joint processing (calc_JP_*) */
    int32_t reserved;
    int32_t nr[4], nt[4], ns[4];
    double noise_var;           /* 0: no noise term */
    double pe;                  /* power of the external interference */
} mcle_mu_stats_cfg;
int mcle_mu_link_stats(mcle_ctx* ctx, const mcle_mu_stats_cfg* cfg, const void* d_bigH, const double* d_pathloss,
                       const void* d_F, const void* d_U, void* d_Q, void* d_Re, void* d_B, double* d_sinr,
                       size_t batch);

/* ---- fused pipelines: whole realizations on-chip (randomness: mcle-philox-v1) ----------- */
typedef struct mcle_awgn_cfg {          /* C1: apps/awgn_modulators/simulate_psk.py:51-115 */
    int32_t n_symbols;
    int32_t demod_method;
    double noise_var;
} mcle_awgn_cfg;

typedef struct mcle_flat_cfg {          /* C2: flat Jakes fading, y = h s + n, equalise y/h */
    int32_t n_symbols;
    int32_t demod_method;
    double noise_var;
    double Fd, Ts;
    int32_t L;
    int32_t rayleigh_iid;               /* 1: h ~ randn_c per sample (RayleighSampleGenerator) */
} mcle_flat_cfg;

#define MCLE_MAX_TAPS 24
typedef struct mcle_ofdm_tdl_cfg {      /* C3: notebooks/TDL_and_OFDM.ipynb OfdmTdlSimulator */
    int32_t fft_size, cp_size, num_used, n_ofdm_sym;
    int32_t demod_method;
    int32_t n_taps;
    int32_t L;
    int32_t reserved;
    double noise_var;
    double Fd, Ts;
    double tap_power[MCLE_MAX_TAPS];    /* linear, discretised profile (sum 1) */
    int32_t tap_delay[MCLE_MAX_TAPS];   /* sample indexes, increasing */
} mcle_ofdm_tdl_cfg;

typedef struct mcle_mimo_ofdm_cfg {     /* C4: apps/mimo/simulate_mimo.py:68-142 + OFDM */
    int32_t nt, nr;
    int32_t fft_size, cp_size, num_used, n_ofdm_sym;
    int32_t demod_method;
    int32_t mmse;                       /* 1: Blast.set_noise_var(noise_var); 0: zero forcing */
    double noise_var;
} mcle_mimo_ofdm_cfg;

typedef struct mcle_mimo_ofdm_tdl_cfg { /* TdlMimoChannel (fading.py:1290-1333) + per-antenna OFDM +
                                         * one Blast filter per used subcarrier (mimo.py:577-607) */
    int32_t nt, nr;                     /* fused: every 1 <= nt <= nr <= 4 at fft_size 256 .. 2048 with the last delay <= min(256, fft_size / 2)
                                         * (inside the prefix or beyond it); nt == nr in {2, 4} at 64 .. 2048 otherwise */
    int32_t fft_size, cp_size, num_used, n_ofdm_sym;
    int32_t demod_method;
    int32_t mmse;                       /* 1: MMSE with noise_var; 0: zero forcing */
    int32_t n_taps, L;                  /* discretised profile taps; Jakes rays per fading process */
    double noise_var, Fd, Ts;
    double tap_power[MCLE_MAX_TAPS];    /* linear */
    int32_t tap_delay[MCLE_MAX_TAPS];   /* samples, ascending, < fft_size */
} mcle_mimo_ofdm_tdl_cfg;

enum { MCLE_MIMO_BLAST = 0,     /* mimo.Blast     mimo/mimo.py:463-660   (Nt <= Nr <= 4) */
       MCLE_MIMO_MRC = 1,       /* mimo.MRC       mimo/mimo.py:789-830   (Nt = 1)        */
       MCLE_MIMO_MRT = 2,       /* mimo.MRT       mimo/mimo.py:666-783   (Nr = 1)        */
       MCLE_MIMO_ALAMOUTI = 3,  /* mimo.Alamouti  mimo/mimo.py:1073-1287 (Nt = 2)        */
       MCLE_MIMO_SVD = 4,       /* mimo.SVDMimo   mimo/mimo.py:833-946   (square, 2..4)  */
       MCLE_MIMO_GMD = 5 };     /* mimo.GMDMimo   mimo/mimo.py:952-1067  (square, 2..4)  */

typedef struct mcle_mimo_flat_cfg {     /* apps/mimo/simulate_mimo.py:68-142: flat H = randn_c(Nr, Nt), single carrier */
    int32_t scheme;                     /* MCLE_MIMO_* */
    int32_t nt, nr;
    int32_t n_symbols;                  /* NSymbs per layer (layers: Nt for Blast / MRC / SVD / GMD, 1 otherwise) */
    int32_t demod_method;
    int32_t mmse;                       /* Blast / MRC / GMD: 1 = set_noise_var(noise_var); 0 = zero forcing (the app) */
    double noise_var;
} mcle_mimo_flat_cfg;

enum { MCLE_IA_CLOSED_FORM = 0,  /* ClosedFormIASolver      ia/algorithms.py:42-265    */
       MCLE_IA_ALT_MIN = 1,      /* AlternatingMinIASolver  ia/algorithms.py:885-1129  */
       MCLE_IA_MIN_LEAKAGE = 2,  /* MinLeakageIASolver      ia/algorithms.py:1132-1240 */
       MCLE_IA_MAX_SINR = 3,     /* MaxSinrIASolver         ia/algorithms.py:1243-1507 */
       MCLE_IA_MMSE = 4 };       /* MMSEIASolver            ia/algorithms.py:1510-1850 */

enum { MCLE_IA_INIT_GIVEN = 0,        /* 'random' (pipelines: drawn on-chip) or 'fix' (operator: injected) */
       MCLE_IA_INIT_CLOSED_FORM = 1,  /* 'closed_form': F and W of ClosedFormIASolver                      */
       MCLE_IA_INIT_ALT_MIN = 2,      /* 'alt_min': AlternatingMinIASolver run first, same max_iterations */
       MCLE_IA_INIT_SVD = 3 };        /* 'svd': dominant right singular vector of each direct channel (phase: ours) */

typedef struct mcle_ia_cfg {            /* C5: apps/ia/simulate_ia.py:94-245, ClosedFormIASolver */
    int32_t K, nr, nt, ns;              /* supported: K = 3, nr = nt = 2, ns = 1 */
    int32_t n_symbols;                  /* NSymbs per stream */
    int32_t demod_method;
    double noise_var;
    int32_t solver;                     /* MCLE_IA_*: closed form or an iterative solver (initialize_with='random') */
    int32_t max_iterations;             /* iterative solvers: IterativeIASolverBaseClass.max_iterations */
    double relative_factor;             /* ... and .relative_factor (algorithms.py:316-322) */
    int32_t initialize_with;            /* MCLE_IA_INIT_*: 'random' / 'closed_form' / 'alt_min' / 'svd' (algorithms.py:633-663) */
    int32_t reserved;
} mcle_ia_cfg;

/* Each run_* processes realizations [first, first+count) of `seed`, ADDS into d_counters[0]
 * and, when non-NULL, writes per-realization d_sym_err / d_bit_err [count]. */
int mcle_run_awgn(mcle_ctx* ctx, int dtype, const mcle_awgn_cfg* cfg, uint64_t seed,
                  uint64_t first, uint64_t count, mcle_counters* d_counters,
                  uint32_t* d_sym_err, uint32_t* d_bit_err);
/* Jakes fading (JakesSampleGenerator, fading_generators.py:427-493) in the complex128 instantiation: by default the ray
 * phasors of a thread's run of 16 symbols advance by a rotation recurrence (one exact sincos per ray and run, then <= 15
 * complex products: <= 3e-15 relative drift against evaluating sin / cos at every sample, which is what NumPy does).  The
 * per-realization error counts equal the oracle's on every tested case incl. the full 10^5-symbol config 2, but that is by
 * test coverage, not by construction: a symbol within 3e-15 of a decision boundary may flip.  MCLE_OPT_JAKES_DIRECT = 1
 * evaluates every sample (the literal parity statement, 3.3 x slower). */
int mcle_run_flat_fading(mcle_ctx* ctx, int dtype, const mcle_flat_cfg* cfg, uint64_t seed,
                         uint64_t first, uint64_t count, mcle_counters* d_counters,
                         uint32_t* d_sym_err, uint32_t* d_bit_err);
int mcle_run_ofdm_tdl(mcle_ctx* ctx, int dtype, const mcle_ofdm_tdl_cfg* cfg, uint64_t seed,
                      uint64_t first, uint64_t count, mcle_counters* d_counters,
                      uint32_t* d_sym_err, uint32_t* d_bit_err);
/* Envelope: 1 <= Nt <= Nr <= 4, either arithmetic: fft_size 256 / 512 / 1024 / 2048 with every Nt <= Nr (Blast takes any Nr x Nt,
 * mimo/mimo.py:264-309) on the planar kernel family (pipeline_mimo_planar.hip; in complex128 2048 with 4 receive antennas exists
 * there only: 148 KiB of LDS); 2x2 / 4x4 at 64 and 128 on the generic kernel.  complex64 at (1024, 4x4): the planar radix-16
 * kernel by default, the matrix-core kernel with MCLE_OPT_F32_MFMA = 1.  Anything else: MCLE_E_INVAL. */
int mcle_run_mimo_ofdm(mcle_ctx* ctx, int dtype, const mcle_mimo_ofdm_cfg* cfg, uint64_t seed,
                       uint64_t first, uint64_t count, mcle_counters* d_counters,
                       uint32_t* d_sym_err, uint32_t* d_bit_err);

/* The reference's MIMO application, any of its six schemes, fused. */
int mcle_run_mimo_flat(mcle_ctx* ctx, int dtype, const mcle_mimo_flat_cfg* cfg, uint64_t seed,
                       uint64_t first, uint64_t count, mcle_counters* d_counters,
                       uint32_t* d_sym_err, uint32_t* d_bit_err);

/* Fused frequency-selective MIMO-OFDM (round 5: one receive antenna per wavefront, csrc/mimo_tdl_wave.hpp; the workgroup-cooperative
 * kernel of rounds 1-4 behind MCLE_OPT_MIMO_TDL_KERNEL = 1 and for square geometries outside the wavefront kernel's envelope).
 * Returns MCLE_E_UNSUPPORTED (and touches nothing) when the Doppler phase across half an OFDM symbol is beyond the kernels'
 * polynomial tap model, or for a rectangular geometry outside the envelope named at mcle_mimo_ofdm_tdl_cfg: run the staged operators
 * then.  Device memory: the context's scratch buffer grows to hold the fading records of one launch slice -- 2.5 KiB per
 * realization and symbol in complex64 at five taps of 4 x 4, 7.5 KiB in complex128, at most 4 GiB (+ 25 %) per slice
 * (mcle_run_ofdm_tdl: at most 2 GiB) -- and is kept until the context is destroyed.
 * complex64 (this function and mcle_run_ofdm_tdl): while the largest Doppler phase of the run, Fd x (Ts + dt x samples of all symbols),
 * stays below a quarter turn the rays' Doppler frequencies are evaluated in float (v_cos_f32: < 4e-7 turns of phase, below the float
 * phasor's own rounding); beyond that in double as in complex128. */
int mcle_run_mimo_ofdm_tdl(mcle_ctx* ctx, int dtype, const mcle_mimo_ofdm_tdl_cfg* cfg, uint64_t seed,
                           uint64_t first, uint64_t count, mcle_counters* d_counters,
                           uint32_t* d_sym_err, uint32_t* d_bit_err);

/* d_sum_capacity (may be NULL): per-realization sum_k log2(1 + SINR_k) of the chosen solution */
/* d_iterations (may be NULL): per-realization runned_iterations of an iterative solver.  Iterative solvers
 * draw their random initial precoders (randomizeF, iabase.py:538-540) from stream 3 of the realization. */
int mcle_run_ia(mcle_ctx* ctx, int dtype, const mcle_ia_cfg* cfg, uint64_t seed, uint64_t first,
                uint64_t count, mcle_counters* d_counters, uint32_t* d_sym_err, uint32_t* d_bit_err,
                double* d_sum_capacity, uint32_t* d_iterations);

/* ---- a15: ClosedFormIASolver.solve (ia/algorithms.py:194-265) on injected channels, f64 only:
 *      d_bigH [batch][6][6] (MultiUserChannelMatrix.big_H, K = 3, 2x2) -> precoders d_F
 *      [batch][3][2] (full_F, P = 1), receive filters d_U [batch][3][2] (full_W_H), per-user SINR
 *      d_sinr [batch][3] and sum capacity d_capacity [batch] (both may be NULL) ------------------ */
int mcle_ia_closed_form(mcle_ctx* ctx, const void* d_bigH, double noise_var, void* d_F, void* d_U,
                        double* d_sinr, double* d_capacity, uint32_t* d_skipped, size_t batch);
/* Iterative solvers (solver = MCLE_IA_ALT_MIN / MIN_LEAKAGE / MAX_SINR / MMSE) from injected initial precoders
 * d_F_init [batch][3][2] (unit norm): the starting precoders for MCLE_IA_INIT_GIVEN ('fix' / a captured
 * 'random' start), the alternating-minimisation solver's start for MCLE_IA_INIT_ALT_MIN, ignored for
 * MCLE_IA_INIT_CLOSED_FORM.  IterativeIASolverBaseClass.solve (algorithms.py:802-883).  d_iterations [batch]
 * (may be NULL) = runned_iterations.
 * Reproducibility against the reference (complex128): the device eigen-solver is a cyclic Jacobi sweep, the reference's is
 * LAPACK (numpy.linalg.eig / eigh); both are backward stable, so precoders, filters and SINRs agree to <= 1e-9 relative
 * where the iteration is well conditioned, and the decisions of a link built on them are then identical.  Where the
 * iteration ends badly conditioned -- a stream in outage, an eighth or more of a realization's symbols wrong -- the
 * rounding-level difference is amplified over the iterations and decisions at near-ties may differ: bound asserted by
 * tests/test_gpu_fuzz.py on such realizations: |symbol errors - reference's| <= max(2, 2 %), |bit errors| <= max(8, 4 %);
 * every other realization is exact.  The closed-form solver (mcle_ia_closed_form) has no such caveat. */
int mcle_ia_iterative(mcle_ctx* ctx, int solver, int initialize_with, const void* d_bigH,
                      const void* d_F_init, double noise_var, int max_iterations, double relative_factor,
                      void* d_F, void* d_U, double* d_sinr, double* d_capacity, uint32_t* d_iterations,
                      uint32_t* d_skipped, size_t batch);

/* ---- block diagonalisation with external interference (comm/blockdiagonalization.py:666-1469): WhiteningBD
 *      (:722-836: whiten every user's rows with the interference-plus-noise covariance, block-diagonalise, receive
 *      filter = pinv(newH) x whitening filter) and EnhancedBD (:839-1469: BD directions, then per user a stream
 *      reduction onto the directions least hit by the external interference; number of streams fixed ('naive',
 *      'fixed') or chosen per user by a metric).  The channel is MultiUserChannelMatrixExtInt.big_H
 *      (channels/multiuser.py:2011-2520): d_bigH [batch][K r][K r + n_ext], the last n_ext columns being the
 *      interferers' antennas; covariance = pe H_ext H_ext^H + noise_var I (:2469-2520).
 *      Outputs, zero padded: d_Ms [batch][K][K r][r] (user k's precoder MsPk, K r x Ns), d_W [batch][K][r][r]
 *      (receive filter, Ns x r), d_ns [batch][K]; d_cand_sinr [batch][K][r][r] (metric 4 only: row ns-1 = the
 *      post-filter SINRs with ns streams, EnhancedBD._calc_linear_SINRs :1101-1138, for caller-side metrics such
 *      as 'effective_throughput').  Singular-vector phases are ours (see mcle_block_diagonalize). */
typedef struct mcle_bd_extint_cfg {
    int32_t num_users, n_ant_per_user, n_ext;
    int32_t method;             /* 0: WhiteningBD, 1: EnhancedBD */
    int32_t metric;             /* EnhancedBD: 0 None, 1 'naive', 2 'fixed', 3 'capacity', 4 report candidate SINRs,
                                   5 stream counts given per user (ns_user) */
    int32_t num_streams;        /* 'naive' / 'fixed' */
    int32_t ns_user[4];
    double iPu, noise_var, pe;
} mcle_bd_extint_cfg;
int mcle_bd_extint(mcle_ctx* ctx, const mcle_bd_extint_cfg* cfg, const void* d_bigH, void* d_Ms, void* d_W,
                   int32_t* d_ns, double* d_cand_sinr, uint32_t* d_skipped, size_t batch);

/* ---- iterative interference alignment for general geometries (SURVEY 8(f).3 tail): K <= 4 users with
 *      Nr x Nt <= 6 x 6 antennas each (the reference's own application runs K = 3, Nr = 5, Nt = 3, Ns = 2,
 *      apps/ia/IA_Results_NrxNt(Ns).py:130-133) and per-user stream counts; AlternatingMinIASolver / MinLeakageIASolver /
 *      MaxSinrIASolver .solve (ia/algorithms.py:802-883, 885-1507) from injected precoders ('fix') or the 'svd'
 *      start (:503-547, Nr == Nt), optionally inside GreedStreamIASolver.solve (:1905-2010: drop the worst stream
 *      while the sum capacity grows) or BruteForceStreamIASolver.solve (:2147-2260: every stream combination up to
 *      ns[], 'svd' start, best sum capacity).  One lane per channel realization, f64.
 *      Padded arrays are D x D with D = 4 when max(Nr, Nt) <= 4 and D = 6 otherwise (two instantiations of the solver).
 *      d_bigH [batch][K nr][K nt]; d_F_init [batch][4][D][D] = user k's nt x ns[k] start in the top-left corner
 *      (ignored for 'svd' and for brute force).  Outputs, same padded layout: d_F (nt x ns, unit Frobenius norm =
 *      full_F for P = 1), d_U (full_W_H, ns x nr), d_sinr [batch][4][D] (linear, per stream), d_capacity [batch],
 *      d_iterations [batch] (all runs of a selection wrapper added up), d_ns [batch][4] (streams kept per user),
 *      d_skipped [batch] (a singular system met on the way), d_every_capacity [batch][256] (brute force only: the sum
 *      capacity of every stream combination in itertools.product order, last user fastest --
 *      BruteForceStreamIASolver.every_sum_capacity :2135-2145); every output but d_F / d_U may be NULL.
 *      Eigenvector phases are ours, not LAPACK's: SINRs, capacity and decisions do not depend on them. */
typedef struct mcle_ia_general_cfg {
    int32_t K, nr, nt;
    int32_t ns[4];
    int32_t solver;             /* MCLE_IA_ALT_MIN / MCLE_IA_MIN_LEAKAGE / MCLE_IA_MAX_SINR */
    int32_t initialize_with;    /* MCLE_IA_INIT_GIVEN or MCLE_IA_INIT_SVD */
    int32_t max_iterations;
    int32_t stream_selection;   /* 0: none, 1: greedy, 2: brute force */
    int32_t reserved;
    double noise_var, relative_factor;
} mcle_ia_general_cfg;
int mcle_ia_solve_general(mcle_ctx* ctx, const mcle_ia_general_cfg* cfg, const void* d_bigH, const void* d_F_init,
                          void* d_F, void* d_U, double* d_sinr, double* d_capacity, uint32_t* d_iterations,
                          int32_t* d_ns, uint32_t* d_skipped, double* d_every_capacity, size_t batch);

/* ---- block diagonalisation of a multi-user downlink (SURVEY 8(f).3 tail) --------------------
 * comm/waterfilling.py:15-92 doWF: d_gains [batch][n] channel POWER gains -> optimum powers
 * d_powers [batch][n] (same order) and the water level d_mu [batch] (may be NULL); n <= 64. */
int mcle_waterfilling(mcle_ctx* ctx, const double* d_gains, int n, double total_power, double noise_var,
                      double* d_powers, double* d_mu, size_t batch);
/* comm/blockdiagonalization.py:466-508 BlockDiagonalizer.block_diagonalize (waterfilling = 1: water-filling
 * over all streams normalised to the strongest user block, :403-464) or :510-566
 * block_diagonalize_no_waterfilling (waterfilling = 0), plus :568-585 calc_receive_filter, f64 only, on
 * injected channels d_H [batch][n][n] (row-major, n = num_users * n_rx_per_user <= 8, square: as many
 * transmit as receive antennas).  Outputs (each may be NULL): precoder d_Ms [batch][n][n], d_newH = H Ms
 * [batch][n][n], zero-forcing filter d_W = pinv(newH) [batch][n][n], singular values of the users'
 * equivalent channels d_sigma [batch][n] (ascending per user, the reference's Sigma), d_skipped [batch] =
 * 1 for a numerically singular channel.  Singular vectors are unique up to one phase per stream; this
 * entry point returns the representative whose largest entry per Ms column is real positive. */
int mcle_block_diagonalize(mcle_ctx* ctx, const void* d_H, int num_users, int n_rx_per_user, double iPu,
                           double noise_var, int waterfilling, void* d_Ms, void* d_newH, void* d_W,
                           double* d_sigma, uint32_t* d_skipped, size_t batch);

/* np.linalg.pinv (the receive filter of blockdiagonalization.py:568-585 for any newH): d_A [batch][m][n] ->
 * d_out [batch][n][m], f64, m, n <= 8; singular values <= rcond * max are dropped (numpy's default 1e-15). */
int mcle_pinv(mcle_ctx* ctx, const void* d_A, int m, int n, double rcond, void* d_out, size_t batch);

typedef struct mcle_bd_cfg {            /* apps/comp_BD/simulate_comp_simple.py:95-140 (no external interference) */
    int32_t K, nr;                      /* K cells/users of nr x nr antennas: channel (K nr) x (K nr), K nr <= 8 */
    int32_t n_symbols;                  /* NSymbs per stream (K nr streams) */
    int32_t demod_method;
    int32_t waterfilling;               /* 1: block_diagonalize, 0: block_diagonalize_no_waterfilling */
    int32_t has_pathloss;               /* 1: block (rx k, tx l) scaled by sqrt(pathloss[k*K + l]) (multiuser.py:256-292) */
    double iPu;                         /* power per user */
    double noise_var;                   /* channel noise variance */
    double bd_noise_var;                /* noise variance handed to the water-filling (the app passes 1e-50) */
    double pathloss[16];
} mcle_bd_cfg;
/* Fused: channel draw, block diagonalisation, precoding, channel + noise, zero forcing, demodulation, counts. */
int mcle_run_bd(mcle_ctx* ctx, int dtype, const mcle_bd_cfg* cfg, uint64_t seed, uint64_t first,
                uint64_t count, mcle_counters* d_counters, uint32_t* d_sym_err, uint32_t* d_bit_err);

/* ---- same-seed parity mode: NumPy's legacy global RandomState replayed on the device -------
 * Realization r receives exactly what the reference draws after np.random.seed(seed_base + r)
 * (legacy MT19937; util/misc.py:327-355 randn_c = randn real block then imag block, and
 * np.random.randint of apps/awgn_modulators/simulate_psk.py:65).  A program is a sequence of
 * segments: kind 0 = randint(0, range, n) with `range` a power of two -> int32 outputs,
 * kind 1 = randn(n) -> float64 outputs (the polar method's cached value carries over between
 * segments like NumPy's), kind 2 = rand(n) -> float64 uniforms in [0, 1).  Rows: d_int [count][n_int], d_dbl [count][n_dbl]; d_status[count] is
 * set to 1 if the word budget of a realization was exhausted (never observed; may be NULL). */
typedef struct mcle_legacy_seg {
    int32_t kind;
    int32_t n;
    uint32_t range;
    uint32_t reserved;
} mcle_legacy_seg;
int mcle_legacy_draws(mcle_ctx* ctx, const mcle_legacy_seg* segs, int n_segs, uint32_t seed_base,
                      uint64_t first, uint64_t count, int32_t* d_int, size_t n_int, double* d_dbl,
                      size_t n_dbl, uint32_t* d_status);
/* out = scale * (re + 1j*im): assembles randn_c from the two randn blocks */
int mcle_complex_from_parts(mcle_ctx* ctx, int dtype, const double* d_re, const double* d_im,
                            double scale, void* d_out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* MCLE_H */
