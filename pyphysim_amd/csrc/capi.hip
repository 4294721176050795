// capi.hip -- context, stream, memory and constellation management of libmcle.
#include <cmath>
#include <cstdarg>

#include "common.hpp"
#include "modem.hpp"

namespace mcle {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace mcle

using namespace mcle;

int mcle_ctx::bind() const {
    MCLE_HIP(hipSetDevice(device));
    return MCLE_OK;
}

int mcle_ctx::scratch(size_t bytes, void** d_ptr) {
    if (bytes > scratch_bytes) {
        if (d_scratch) MCLE_HIP(hipFree(d_scratch));  // hipFree synchronises with in-flight work
        d_scratch = nullptr;
        scratch_bytes = 0;
        // 25 % headroom against slowly growing requests -- for SMALL buffers only: on top of a multi-GiB record slice it is what
        // tipped a shared device over; and a request that fails WITH the headroom is retried without it
        size_t want = bytes < (1u << 20) ? (1u << 20) : (bytes < ((size_t)256 << 20) ? bytes + bytes / 4 : bytes);
        hipError_t e = hipMalloc(&d_scratch, want);
        if (e == hipErrorOutOfMemory && want > bytes) {
            (void)hipGetLastError();
            want = bytes;
            e = hipMalloc(&d_scratch, want);
        }
        if (e != hipSuccess) {
            d_scratch = nullptr;
            MCLE_HIP(e);
        }
        scratch_bytes = want;
    }
    *d_ptr = d_scratch;
    return MCLE_OK;
}

int mcle_ctx::scratch_upto(size_t want, size_t floor_bytes, void** d_ptr, size_t* got) {
    if (want <= scratch_bytes) {
        *d_ptr = d_scratch;
        *got = want;
        return MCLE_OK;
    }
    size_t free_b = 0, total_b = 0;
    // never ask for more than what is free now plus what this context already holds, less a margin for the other allocations of
    // the call (workspace, counters): saves the failed hipMalloc round trips on a crowded device
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t avail = free_b + scratch_bytes;
        const size_t margin = (size_t)256 << 20;
        if (avail > margin && want > avail - margin) want = avail - margin;
    } else {
        (void)hipGetLastError();
    }
    if (want < floor_bytes) want = floor_bytes;
    for (;;) {
        const int rc = scratch(want, d_ptr);
        if (rc == MCLE_OK) {
            *got = want;
            return MCLE_OK;
        }
        if (want <= floor_bytes) return rc;             // (the message of the failed hipMalloc stays in mcle_last_error)
        (void)hipGetLastError();
        want = want / 2 < floor_bytes ? floor_bytes : want / 2;
    }
}

// w[k] = exp(-2 pi i k / n) computed in double on the host (then rounded once for f32)
int mcle_ctx::get_twiddles(int n, int dtype, void** d_tw) {
    TwiddleKey key{n, dtype};
    auto it = twiddles.find(key);
    if (it != twiddles.end()) {
        *d_tw = it->second;
        return MCLE_OK;
    }
    const double two_pi = 6.283185307179586476925286766559;
    void* d = nullptr;
    if (dtype == MCLE_F64) {
        std::vector<double2> h(n);
        for (int k = 0; k < n; ++k) {
            h[k].x = std::cos(two_pi * k / n);
            h[k].y = -std::sin(two_pi * k / n);
        }
        MCLE_HIP(hipMalloc(&d, n * sizeof(double2)));
        MCLE_HIP(hipMemcpy(d, h.data(), n * sizeof(double2), hipMemcpyHostToDevice));
    } else {
        std::vector<float2> h(n);
        for (int k = 0; k < n; ++k) {
            h[k].x = (float)std::cos(two_pi * k / n);
            h[k].y = (float)-std::sin(two_pi * k / n);
        }
        MCLE_HIP(hipMalloc(&d, n * sizeof(float2)));
        MCLE_HIP(hipMemcpy(d, h.data(), n * sizeof(float2), hipMemcpyHostToDevice));
    }
    twiddles[key] = d;
    *d_tw = d;
    return MCLE_OK;
}

extern "C" {

const char* mcle_last_error(void) { return g_err; }

int mcle_version(void) { return MCLE_VERSION; }

int mcle_device_count(int* count) {
    MCLE_REQUIRE(count != nullptr, "null output pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return MCLE_OK;
}

int mcle_ctx_create(int device_id, mcle_ctx** out) {
    MCLE_REQUIRE(out != nullptr, "null output pointer");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (hipGetDeviceCount: %s); libmcle has no CPU fallback",
                  e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return MCLE_E_HIP;
    }
    MCLE_REQUIRE(device_id >= 0 && device_id < n, "device_id %d out of range [0, %d)", device_id, n);
    MCLE_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    MCLE_HIP(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libmcle is built for gfx950 (MI355X) only", device_id, prop.gcnArchName);
        return MCLE_E_HIP;
    }
    mcle_ctx* ctx = new (std::nothrow) mcle_ctx();
    if (!ctx) {
        set_error("out of host memory");
        return MCLE_E_NOMEM;
    }
    ctx->device = device_id;
    ctx->n_cu = prop.multiProcessorCount;
    ctx->lds_bytes = (int)prop.sharedMemPerBlock;
    ctx->name = prop.name;
    hipError_t es = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (es != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(es));
        delete ctx;
        return MCLE_E_HIP;
    }
    ctx->own_stream = true;
    (void)hipEventCreate(&ctx->ev0);
    (void)hipEventCreate(&ctx->ev1);
    (void)hipEventCreate(&ctx->ev_probe0);
    (void)hipEventCreate(&ctx->ev_probe1);
    *out = ctx;
    return MCLE_OK;
}

int mcle_ctx_destroy(mcle_ctx* ctx) {
    if (!ctx) return MCLE_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)mcle_comm_destroy(ctx);
    if (ctx->d_comm_buf) (void)hipFree(ctx->d_comm_buf);
    for (auto& kv : ctx->twiddles) (void)hipFree(kv.second);
    if (ctx->d_table_f32) (void)hipFree(ctx->d_table_f32);
    if (ctx->d_table_f64) (void)hipFree(ctx->d_table_f64);
    if (ctx->d_grid) (void)hipFree(ctx->d_grid);
    if (ctx->d_psk) (void)hipFree(ctx->d_psk);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_probe0) (void)hipEventDestroy(ctx->ev_probe0);
    if (ctx->ev_probe1) (void)hipEventDestroy(ctx->ev_probe1);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MCLE_OK;
}

int mcle_ctx_set_stream(mcle_ctx* ctx, void* hip_stream) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_HIP(hipSetDevice(ctx->device));
    MCLE_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) MCLE_HIP(hipStreamDestroy(ctx->stream));
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->own_stream = false;
    } else {
        MCLE_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return MCLE_OK;
}

int mcle_ctx_get_stream(mcle_ctx* ctx, void** hip_stream) {
    MCLE_REQUIRE(ctx != nullptr && hip_stream != nullptr, "null argument");
    *hip_stream = (void*)ctx->stream;
    return MCLE_OK;
}

int mcle_ctx_set_option(mcle_ctx* ctx, int option, long long value) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(option >= 0 && option < MCLE_OPT_COUNT, "unknown option %d", option);
    bool ok = false;
    switch (option) {
        case MCLE_OPT_NO_MFMA: case MCLE_OPT_SINGLE_TDL: case MCLE_OPT_JAKES_DIRECT: case MCLE_OPT_F64_GENERIC: case MCLE_OPT_BD_RUNTIME_SOLVE: case MCLE_OPT_DEMOD_NOCERT: case MCLE_OPT_F32_MFMA: case MCLE_OPT_WALK_LEGACY: ok = value == 0 || value == 1; break;
        case MCLE_OPT_TDL_KERNEL: ok = value >= 0 && value <= 4; break;
#ifdef MCLE_EXPERIMENTS
        case MCLE_OPT_MIMO_TDL_KERNEL: ok = value >= 0 && value <= 255; break;
#else
        case MCLE_OPT_MIMO_TDL_KERNEL: ok = value >= 0 && value <= 2; break;
#endif
#ifdef MCLE_EXPERIMENTS
        case MCLE_OPT_F64_VARIANT: ok = value >= 0 && value <= 2047; break;
#else
        case MCLE_OPT_F64_VARIANT:      // the timing-bound kernels exist in -DMCLE_EXPERIMENTS builds only (wrong counters by construction)
            MCLE_REQUIRE(value == 0, "option MCLE_OPT_F64_VARIANT: the timing-bound variants are compiled with -DMCLE_EXPERIMENTS only");
            ok = true;
            break;
#endif
        case MCLE_OPT_MFMA_VARIANT: ok = value == 0 || value == 36 || value == 32 || value == 30 || value == 21; break;
        case MCLE_OPT_GRID_OVERSUB: ok = value >= 0 && value <= 64; break;
        case MCLE_OPT_FLAT_WGS_PER_CU: ok = value >= 0 && value <= 4096; break;
        case MCLE_OPT_TDL_MFMA_WAVES: ok = value == 0 || value == 2 || value == 3 || value == 32; break;
        case MCLE_OPT_F64_THREADS: ok = value == 0 || (value >= 256 && value <= 264) || value == 512 || value == 1024; break;
    }
    MCLE_REQUIRE(ok, "option %d: value %lld out of range", option, value);
    ctx->opt[option] = value;
    return MCLE_OK;
}

int mcle_ctx_trim_scratch(mcle_ctx* ctx) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_HIP(hipSetDevice(ctx->device));
    MCLE_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->d_scratch) MCLE_HIP(hipFree(ctx->d_scratch));
    ctx->d_scratch = nullptr;
    ctx->scratch_bytes = 0;
    return MCLE_OK;
}

int mcle_ctx_get_option(mcle_ctx* ctx, int option, long long* value) {
    MCLE_REQUIRE(ctx != nullptr && value != nullptr, "null argument");
    MCLE_REQUIRE(option >= 0 && option < MCLE_OPT_COUNT, "unknown option %d", option);
    *value = ctx->opt[option];
    return MCLE_OK;
}

int mcle_ctx_sync(mcle_ctx* ctx) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_HIP(hipSetDevice(ctx->device));
    MCLE_HIP(hipStreamSynchronize(ctx->stream));
    return MCLE_OK;
}

int mcle_ctx_device_info(mcle_ctx* ctx, int* n_cu, int* lds_bytes, char* name, int name_len) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    if (n_cu) *n_cu = ctx->n_cu;
    if (lds_bytes) *lds_bytes = ctx->lds_bytes;
    if (name && name_len > 0) {
        std::strncpy(name, ctx->name.c_str(), name_len - 1);
        name[name_len - 1] = 0;
    }
    return MCLE_OK;
}

int mcle_malloc(mcle_ctx* ctx, size_t bytes, void** d_ptr) {
    MCLE_REQUIRE(ctx != nullptr && d_ptr != nullptr, "null argument");
    MCLE_HIP(hipSetDevice(ctx->device));
    *d_ptr = nullptr;
    if (bytes == 0) return MCLE_OK;
    hipError_t e = hipMalloc(d_ptr, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return MCLE_E_NOMEM;
    }
    return MCLE_OK;
}

int mcle_free(mcle_ctx* ctx, void* d_ptr) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    if (!d_ptr) return MCLE_OK;
    MCLE_HIP(hipSetDevice(ctx->device));
    MCLE_HIP(hipFree(d_ptr));
    return MCLE_OK;
}

int mcle_memset(mcle_ctx* ctx, void* d_ptr, int value, size_t bytes) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    if (bytes == 0) return MCLE_OK;
    MCLE_HIP(hipSetDevice(ctx->device));
    MCLE_HIP(hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
    return MCLE_OK;
}

int mcle_memcpy_h2d(mcle_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    if (bytes == 0) return MCLE_OK;
    MCLE_HIP(hipSetDevice(ctx->device));
    // pageable source: hipMemcpyAsync stages and returns once the source is consumed
    MCLE_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    MCLE_HIP(hipStreamSynchronize(ctx->stream));
    return MCLE_OK;
}

int mcle_memcpy_d2h(mcle_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    if (bytes == 0) return MCLE_OK;
    MCLE_HIP(hipSetDevice(ctx->device));
    MCLE_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MCLE_HIP(hipStreamSynchronize(ctx->stream));
    return MCLE_OK;
}

int mcle_memcpy_2d(mcle_ctx* ctx, void* d_dst, size_t dst_pitch, const void* d_src, size_t src_pitch,
                   size_t row_bytes, size_t rows) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(row_bytes <= dst_pitch && row_bytes <= src_pitch, "row_bytes exceeds a pitch");
    if (row_bytes == 0 || rows == 0) return MCLE_OK;
    MCLE_HIP(hipSetDevice(ctx->device));
    MCLE_HIP(hipMemcpy2DAsync(d_dst, dst_pitch, d_src, src_pitch, row_bytes, rows, hipMemcpyDeviceToDevice,
                              ctx->stream));
    return MCLE_OK;
}

int mcle_timer_start(mcle_ctx* ctx) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_HIP(hipSetDevice(ctx->device));
    MCLE_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return MCLE_OK;
}

int mcle_timer_stop_ms(mcle_ctx* ctx, float* ms) {
    MCLE_REQUIRE(ctx != nullptr && ms != nullptr, "null argument");
    MCLE_HIP(hipSetDevice(ctx->device));
    MCLE_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    MCLE_HIP(hipEventSynchronize(ctx->ev1));
    MCLE_HIP(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return MCLE_OK;
}

// a1: the table comes from the host mirror (QAM / PSK / BPSK classes or setConstellation).  For
// kind == MCLE_CONST_QAM the library re-derives the square Gray QAM layout of the reference
// (modulators/fundamental.py:697-777) and refuses tables that do not match it, because the slicer
// fast path relies on that structure.
// Candidate grid of the pruned min-distance search (modem.hpp: DemodGrid), built in f64.
// For every cell: c* = nearest point to the (clamped) cell centre; a point c stays on the list unless c* is
// closer than c by more than `margin` EVERYWHERE in the cell.  f(p) = |p - c*|^2 - |p - c|^2 is linear in p, so
// its supremum over the (possibly unbounded) rectangle sits at a corner or is +inf.  Any point that can win
// the exhaustive search somewhere in the cell must beat c* there, hence is on the list.
int mcle_build_demod_grid(const double* re_im, int M, int* G_out, double* x0_out, double* y0_out, double* h_out,
                          unsigned long long* cells) {
    MCLE_REQUIRE(re_im != nullptr && G_out != nullptr && x0_out != nullptr && y0_out != nullptr && h_out != nullptr &&
                     cells != nullptr, "null argument");
    MCLE_REQUIRE(M >= 2 && M <= 256, "the candidate grid covers 2 <= M <= 256 (byte indices)");
    int G = 8;
    while (G < 32 && G * G < 4 * M) G *= 2;              // about 2 sqrt(M) cells per axis, in {8, 16, 32}
    double maxabs = 0.0;
    for (int m = 0; m < 2 * M; ++m) maxabs = std::max(maxabs, std::fabs(re_im[m]));
    MCLE_REQUIRE(maxabs > 0.0, "degenerate constellation");
    const double half = 1.2345 * maxabs;                  // off the Voronoi edges of the regular constellations
    const double h = 2.0 * half / G, x0 = -half, y0 = -half;
    const double R = 1.5 * (half + maxabs);
    const double margin = 1e-5 * R * R;                   // >> f32 rounding of either metric inside the box
    for (int iy = 0; iy < G; ++iy)
        for (int ix = 0; ix < G; ++ix) {
            const double xl = x0 + ix * h, xh = xl + h, yl = y0 + iy * h, yh = yl + h;
            const double cx_ = 0.5 * (xl + xh), cy_ = 0.5 * (yl + yh);
            int best = 0;
            double bd = 1e300;
            for (int m = 0; m < M; ++m) {
                const double dx = cx_ - re_im[2 * m], dy = cy_ - re_im[2 * m + 1];
                const double d = dx * dx + dy * dy;
                if (d < bd) {
                    bd = d;
                    best = m;
                }
            }
            const double sx = re_im[2 * best], sy = re_im[2 * best + 1];
            unsigned long long word = 0;
            int n = 0;
            for (int m = 0; m < M && n <= 7; ++m) {
                bool keep = (m == best);
                if (!keep) {
                    const double px = re_im[2 * m], py = re_im[2 * m + 1];
                    const double bx = 2.0 * (px - sx), by = 2.0 * (py - sy);   // f(p) = K + b . p
                    const double K = (sx * sx + sy * sy) - (px * px + py * py);
                    const bool inf_x = (bx > 0.0 && ix == G - 1) || (bx < 0.0 && ix == 0);
                    const bool inf_y = (by > 0.0 && iy == G - 1) || (by < 0.0 && iy == 0);
                    if (inf_x || inf_y) {
                        keep = true;
                    } else {
                        const double sup = K + bx * (bx > 0.0 ? xh : xl) + by * (by > 0.0 ? yh : yl);
                        keep = sup >= -margin;
                    }
                }
                if (keep) {
                    if (n < 7) word |= (unsigned long long)m << (8 * (n + 1));
                    ++n;
                }
            }
            cells[iy * G + ix] = n > 7 ? 0xFFull : (word | (unsigned long long)n);
        }
    *G_out = G;
    *x0_out = x0;
    *y0_out = y0;
    *h_out = h;
    return MCLE_OK;
}

int mcle_set_constellation(mcle_ctx* ctx, const double* re_im, int M, int kind) {
    MCLE_REQUIRE(ctx != nullptr && re_im != nullptr, "null argument");
    MCLE_REQUIRE(M >= 2 && M <= 1024 && (M & (M - 1)) == 0, "M must be a power of two in [2, 1024] (got %d)", M);
    MCLE_REQUIRE(kind == MCLE_CONST_GENERIC || kind == MCLE_CONST_QAM || kind == MCLE_CONST_BPSK,
                 "unknown constellation kind %d", kind);
    int bits = 0;
    while ((1 << bits) < M) ++bits;
    double scale = 0.0;
    int L = 0;
    if (kind == MCLE_CONST_QAM) {
        MCLE_REQUIRE(bits % 2 == 0, "M must be a square power of 2");
        L = 1 << (bits / 2);
        scale = std::sqrt((M - 1) * 2.0 / 3.0);
        for (int r = 0; r < L; ++r)
            for (int c = 0; c < L; ++c) {
                const int label = (r << (bits / 2)) | c;
                const int gi = r ^ (r >> 1), gj = c ^ (c >> 1);
                const double re = (-(L - 1) + 2 * gj) / scale, im = ((L - 1) - 2 * gi) / scale;
                MCLE_REQUIRE(std::fabs(re_im[2 * label] - re) < 1e-12 && std::fabs(re_im[2 * label + 1] - im) < 1e-12,
                             "table is not the reference's square Gray %d-QAM (label %d)", M, label);
            }
    }
    MCLE_HIP(hipSetDevice(ctx->device));
    MCLE_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->d_table_f32) MCLE_HIP(hipFree(ctx->d_table_f32));
    if (ctx->d_table_f64) MCLE_HIP(hipFree(ctx->d_table_f64));
    ctx->d_table_f32 = nullptr;
    ctx->d_table_f64 = nullptr;
    ctx->M = 0;
    std::vector<float2> h32(M);
    for (int m = 0; m < M; ++m) {
        h32[m].x = (float)re_im[2 * m];
        h32[m].y = (float)re_im[2 * m + 1];
    }
    MCLE_HIP(hipMalloc((void**)&ctx->d_table_f32, M * sizeof(float2)));
    MCLE_HIP(hipMalloc((void**)&ctx->d_table_f64, M * sizeof(double2)));
    MCLE_HIP(hipMemcpy(ctx->d_table_f32, h32.data(), M * sizeof(float2), hipMemcpyHostToDevice));
    MCLE_HIP(hipMemcpy(ctx->d_table_f64, re_im, M * sizeof(double2), hipMemcpyHostToDevice));
    ctx->grid_G = 0;
    if (M <= 256) {
        std::vector<unsigned long long> cells(32 * 32);
        int G = 0;
        double x0 = 0, y0 = 0, h = 0;
        int rc = mcle_build_demod_grid(re_im, M, &G, &x0, &y0, &h, cells.data());
        if (rc) return rc;
        if (!ctx->d_grid) MCLE_HIP(hipMalloc((void**)&ctx->d_grid, 32 * 32 * sizeof(unsigned long long)));
        MCLE_HIP(hipMemcpy(ctx->d_grid, cells.data(), (size_t)G * G * sizeof(unsigned long long), hipMemcpyHostToDevice));
        ctx->grid_G = G;
        ctx->grid_x0 = (float)x0;
        ctx->grid_y0 = (float)y0;
        ctx->grid_inv_h = (float)(1.0 / h);
    }
    ctx->M = M;
    ctx->bits = bits;
    ctx->kind = kind;
    ctx->qam_scale = scale;
    ctx->qam_L = L;
    ctx->quad_ok = 0;
    if (M == 4) {                               // one point per quadrant at (+-a, +-b): decisions by the signs (demod_quad_cert)
        const double a = std::fabs(re_im[0]), b = std::fabs(re_im[1]);
        unsigned lut = 0, seen = 0;
        bool ok = a > 0.0 && b > 0.0;
        for (int m = 0; m < 4 && ok; ++m) {
            const double re = re_im[2 * m], im = re_im[2 * m + 1];
            ok = std::fabs(std::fabs(re) - a) <= 1e-12 * a && std::fabs(std::fabs(im) - b) <= 1e-12 * b;
            const unsigned q = (re < 0.0 ? 1u : 0u) | (im < 0.0 ? 2u : 0u);
            seen |= 1u << q;
            lut |= (unsigned)m << (8 * q);
        }
        if (ok && seen == 0xFu) {
            ctx->quad_ok = 1;
            ctx->quad_lut = lut;
            ctx->quad_min = a < b ? a : b;
            ctx->quad_max = a < b ? b : a;
        }
    }
    ctx->axis_ok = 0;
    if (M == 4 && !ctx->quad_ok) {              // (+-a, 0), (0, +-a): decisions by the signs of re - im and re + im (demod_axis4_cert)
        double a = 0.0;
        for (int i = 0; i < 8; ++i) a = std::fabs(re_im[i]) > a ? std::fabs(re_im[i]) : a;
        unsigned lut = 0, seen = 0;
        bool ok = a > 0.0;
        for (int m = 0; m < 4 && ok; ++m) {
            const double re = re_im[2 * m], im = re_im[2 * m + 1];
            const double big = std::fabs(re) > std::fabs(im) ? std::fabs(re) : std::fabs(im);
            const double small = std::fabs(re) > std::fabs(im) ? std::fabs(im) : std::fabs(re);
            ok = std::fabs(big - a) <= 1e-12 * a && small <= 1e-12 * a;
            const unsigned q = (re - im < 0.0 ? 1u : 0u) | (re + im < 0.0 ? 2u : 0u);
            seen |= 1u << q;
            lut |= (unsigned)m << (8 * q);
        }
        if (ok && seen == 0xFu) {
            ctx->axis_ok = 1;
            ctx->axis_lut = lut;
            ctx->axis_a = a;
        }
    }
    // M-PSK beyond four points (modulators/fundamental.py:396-448: exp(j (2 pi m / M + phaseOffset)) in Gray order): equal radii,
    // angles on the grid 2 pi k / M + phi0 with every k taken once -> the sector certificate (modem.hpp demod_psk_cert)
    ctx->psk_ok = 0;
    if ((M == 8 || M == 16) && kind != MCLE_CONST_QAM) {
        const double two_pi = 6.283185307179586476925286766559;
        const double rad = std::hypot(re_im[0], re_im[1]);
        bool ok = rad > 0.0;
        double phi0 = 0.0;
        {   // the offset: the smallest angle in [0, 2 pi / M) that some point sits on (mod 2 pi / M)
            const double a0 = std::atan2(re_im[1], re_im[0]);
            phi0 = std::fmod(a0, two_pi / M);
            if (phi0 < 0.0) phi0 += two_pi / M;
            if (two_pi / M - phi0 < 1e-9) phi0 = 0.0;
        }
        unsigned long long lut = 0, seen = 0;
        const int fw = 64 / M;                                    // label of sector k in bits [k fw, (k + 1) fw)
        for (int m = 0; m < M && ok; ++m) {
            const double re = re_im[2 * m], im = re_im[2 * m + 1];
            ok = std::fabs(std::hypot(re, im) - rad) <= 1e-12 * rad;
            const double kk = (std::atan2(im, re) - phi0) / (two_pi / M);
            const long k = std::lround(kk);
            ok = ok && std::fabs(kk - (double)k) <= 1e-9;
            const int kq = (int)(((k % M) + M) % M);
            seen |= 1ull << kq;
            lut |= (unsigned long long)m << (fw * kq);
        }
        if (ok && seen == ((1ull << M) - 1)) {
            ctx->psk_ok = 1;
            ctx->psk_lut[0] = (unsigned)lut;
            ctx->psk_lut[1] = (unsigned)(lut >> 32);
            ctx->psk_rot[0] = std::cos(phi0);
            ctx->psk_rot[1] = -std::sin(phi0);
            ctx->psk_radius = rad;
            mcle::PskCert h{};
            h.lut[0] = ctx->psk_lut[0];
            h.lut[1] = ctx->psk_lut[1];
            for (int j = 0; j < 2; ++j) {
                const double th = (2 * j + 1) * 3.14159265358979323846 / (double)M;
                h.cb_d[j] = std::cos(th);
                h.sb_d[j] = std::sin(th);
                h.cb_f[j] = (float)h.cb_d[j];
                h.sb_f[j] = (float)h.sb_d[j];
                h.rot_d[j] = ctx->psk_rot[j];
                h.rot_f[j] = (float)ctx->psk_rot[j];
            }
            // hi = max(|re|, |im|) >= |u| / sqrt 2: the window on |u| in units of the radius, a factor sqrt 2 inside on the low side
            h.lo_d = rad * 0x1p-8;
            h.hi_d = rad * 0x1p+8;
            h.lo_f = (float)(rad * 0.125);
            h.hi_f = (float)(rad * 8.0);
            if (!ctx->d_psk) MCLE_HIP(hipMalloc(&ctx->d_psk, sizeof(mcle::PskCert)));
            MCLE_HIP(hipMemcpy(ctx->d_psk, &h, sizeof(mcle::PskCert), hipMemcpyHostToDevice));
        }
    }
    return MCLE_OK;
}

}  // extern "C"
