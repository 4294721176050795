// kernels_modem.hip -- per-operator kernels: modulate, demodulate, error counting, AWGN, RNG.
// HBM-streaming kernels (one pass over the symbol stream, 16 B/lane where the layout allows);
// the constellation lives in LDS.  Reference operators: modulators/fundamental.py:175-248,
// util/misc.py:327-355,519-566.
#include "modem.hpp"
#include "philox.hpp"

namespace mcle {

constexpr int kBlock = 256;
constexpr int kMaxM = 1024;

template <typename T> ModemParams<T> modem_params(const mcle_ctx* ctx, int method);
template <> ModemParams<float> modem_params<float>(const mcle_ctx* ctx, int method) {
    ModemParams<float> p;
    p.grid = context_grid<float>(ctx, method);
    p.g_table = ctx->d_table_f32;
    p.M = ctx->M;
    p.bits = ctx->bits;
    p.method = method;
    p.qam_scale = (float)ctx->qam_scale;
    p.qam_L = ctx->qam_L;
    p.half_bits = ctx->bits / 2;
    modem_fill_cert(ctx, method, p);
    return p;
}
template <> ModemParams<double> modem_params<double>(const mcle_ctx* ctx, int method) {
    ModemParams<double> p;
    p.grid = context_grid<double>(ctx, method, true);
    p.g_table = ctx->d_table_f64;
    p.M = ctx->M;
    p.bits = ctx->bits;
    p.method = method;
    p.qam_scale = ctx->qam_scale;
    p.qam_L = ctx->qam_L;
    p.half_bits = ctx->bits / 2;
    modem_fill_cert(ctx, method, p);
    return p;
}

// ---- modulate: out[i] = table[idx[i]] (negative indices wrap like NumPy; idx >= M flags) -----
template <typename T>
__global__ __launch_bounds__(kBlock) void k_modulate(ModemParams<T> mp, const int32_t* __restrict__ idx,
                                                     cx<T>* __restrict__ out, size_t n,
                                                     unsigned* __restrict__ status, int vec) {
    __shared__ cx<T> s_table[kMaxM];
    load_table(mp, s_table);
    __syncthreads();
    bool bad = false;
    auto lookup = [&](int v) -> cx<T> {
        if (v < 0) v += mp.M;
        if (v < 0 || v >= mp.M) {
            bad = true;
            v = 0;
        }
        return s_table[v];
    };
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    bool done = false;
    if constexpr (sizeof(T) == 4) {
        if (vec) {                       // two symbols per thread: 8-byte index load, 16-byte sample store
            const int2* idx2 = reinterpret_cast<const int2*>(idx);
            float4* out4 = reinterpret_cast<float4*>(out);
            for (size_t p = tid; p < n / 2; p += stride) {
                const int2 v = idx2[p];
                const float2 a = lookup(v.x), b = lookup(v.y);
                out4[p] = make_float4(a.x, a.y, b.x, b.y);
            }
            if ((n & 1) && tid == 0) out[n - 1] = lookup(idx[n - 1]);
            done = true;
        }
    }
    if (!done)
        for (size_t i = tid; i < n; i += stride) out[i] = lookup(idx[i]);
    if (bad) atomicOr(status, 1u);
}

// ---- demodulate ----------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_demodulate(ModemParams<T> mp, const cx<T>* __restrict__ rx,
                                                       int32_t* __restrict__ idx, size_t n, int vec) {
    __shared__ cx<T> s_table[kMaxM];
    __shared__ unsigned long long s_grid[kMaxGridCells];
    load_table(mp, s_table);
    load_grid(mp, s_grid);
    __syncthreads();
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if constexpr (sizeof(T) == 4) {
        if (vec) {                       // two symbols per thread: 16-byte sample load, 8-byte index store
            const float4* rx4 = reinterpret_cast<const float4*>(rx);
            int2* idx2 = reinterpret_cast<int2*>(idx);
            for (size_t p = tid; p < n / 2; p += stride) {
                const float4 v = rx4[p];
                idx2[p] = make_int2(demod_one(mp, s_table, s_grid, make_float2(v.x, v.y)),
                                    demod_one(mp, s_table, s_grid, make_float2(v.z, v.w)));
            }
            if ((n & 1) && tid == 0) idx[n - 1] = demod_one(mp, s_table, s_grid, rx[n - 1]);
            return;
        }
    }
    for (size_t i = tid; i < n; i += stride) idx[i] = demod_one(mp, s_table, s_grid, rx[i]);
}

// ---- error counting --------------------------------------------------------------------------
// grid = (chunks, realizations-in-flight); per-realization partials land in ws[r] = {sym, bit}.
// L: label type of the transmitted / received index arrays (int32_t: the operators' default; uint8_t: the byte labels of
// the staged chains, SURVEY 8(d)'s I = 1 B -- a quarter of the label traffic)
template <typename T, bool DEMOD, typename L = int32_t>
__global__ __launch_bounds__(kBlock) void k_count(ModemParams<T> mp, const cx<T>* __restrict__ rx,
                                                  const L* __restrict__ a, const L* __restrict__ b,
                                                  size_t n_per_real, size_t n_real, unsigned* __restrict__ ws) {
    __shared__ cx<T> s_table[DEMOD ? kMaxM : 1];
    __shared__ unsigned long long s_grid[DEMOD ? kMaxGridCells : 1];
    __shared__ unsigned s_part[2 * (kBlock / 64)];
    if (DEMOD) {
        load_table(mp, s_table);
        load_grid(mp, s_grid);
        __syncthreads();
    }
    for (size_t r = blockIdx.y; r < n_real; r += gridDim.y) {
        unsigned se = 0, be = 0;
        const size_t base = r * n_per_real;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_per_real;
             i += (size_t)gridDim.x * blockDim.x) {
            const int tx = (int)a[base + i];
            const int dec = DEMOD ? demod_one(mp, s_table, s_grid, rx[base + i]) : (int)b[base + i];
            const unsigned x = (unsigned)(tx ^ dec);
            se += (x != 0u);
            be += __popc(x);
        }
        se = wave_sum_u32(se);
        be = wave_sum_u32(be);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) {
            s_part[2 * wave] = se;
            s_part[2 * wave + 1] = be;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned s = 0, t = 0;
            for (int w = 0; w < kBlock / 64; ++w) {
                s += s_part[2 * w];
                t += s_part[2 * w + 1];
            }
            if (s | t) {
                atomicAdd(&ws[2 * r], s);
                atomicAdd(&ws[2 * r + 1], t);
            }
        }
        __syncthreads();
    }
}

// fold per-realization {sym, bit} into the counter block (exact integer sums) and publish them
__global__ __launch_bounds__(kBlock) void k_count_finalize(const unsigned* __restrict__ ws, size_t n_real,
                                                           size_t n_per_real, int bits, mcle_counters* counters,
                                                           uint32_t* __restrict__ sym_out,
                                                           uint32_t* __restrict__ bit_out) {
    unsigned long long se = 0, se2 = 0, be = 0, be2 = 0;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_real; r += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long s = ws[2 * r], t = ws[2 * r + 1];
        if (sym_out) sym_out[r] = (uint32_t)s;
        if (bit_out) bit_out[r] = (uint32_t)t;
        se += s;
        se2 += s * s;
        be += t;
        be2 += t * t;
    }
    if (!counters) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        se += __shfl_xor(se, off, 64);
        se2 += __shfl_xor(se2, off, 64);
        be += __shfl_xor(be, off, 64);
        be2 += __shfl_xor(be2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long*)&counters->sym_errors, se);
        atomicAdd((unsigned long long*)&counters->sym_errors_sq, se2);
        atomicAdd((unsigned long long*)&counters->bit_errors, be);
        atomicAdd((unsigned long long*)&counters->bit_errors_sq, be2);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd((unsigned long long*)&counters->n_realizations, (unsigned long long)n_real);
        counters->n_symbols = n_per_real;
        counters->n_bits = n_per_real * (unsigned long long)bits;
    }
}

// ---- AWGN / element-wise -----------------------------------------------------------------------
// Element-wise binary operators on complex streams.  f32: two samples per thread through 16-byte loads and
// stores when the three pointers are 16-byte aligned (`vec`); the odd tail and unaligned buffers go one by one.
struct OpAwgn {
    template <typename T> __device__ __forceinline__ cx<T> operator()(cx<T> a, cx<T> b, T sigma) const {
        return mk<T>(a.x + sigma * b.x, a.y + sigma * b.y);
    }
};
struct OpDiv {
    template <typename T> __device__ __forceinline__ cx<T> operator()(cx<T> a, cx<T> b, T) const { return cdivide(a, b); }
};
struct OpMul {
    template <typename T> __device__ __forceinline__ cx<T> operator()(cx<T> a, cx<T> b, T) const { return cmul(a, b); }
};

template <typename T, typename Op>
__global__ __launch_bounds__(kBlock) void k_binary(const cx<T>* __restrict__ x, const cx<T>* __restrict__ z, T param,
                                                   cx<T>* __restrict__ y, size_t n, int vec) {
    const Op op{};
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if constexpr (sizeof(T) == 4) {
        if (vec) {
            const size_t pairs = n / 2;
            const float4* x4 = reinterpret_cast<const float4*>(x);
            const float4* z4 = reinterpret_cast<const float4*>(z);
            float4* y4 = reinterpret_cast<float4*>(y);
            for (size_t p = tid; p < pairs; p += stride) {
                const float4 a = x4[p], b = z4[p];
                const float2 r0 = op(make_float2(a.x, a.y), make_float2(b.x, b.y), param);
                const float2 r1 = op(make_float2(a.z, a.w), make_float2(b.z, b.w), param);
                y4[p] = make_float4(r0.x, r0.y, r1.x, r1.y);
            }
            if ((n & 1) && tid == 0) y[n - 1] = op(x[n - 1], z[n - 1], param);
            return;
        }
    }
    for (size_t i = tid; i < n; i += stride) y[i] = op(x[i], z[i], param);
}

template <typename Op>
int launch_binary(mcle_ctx* ctx, int dtype, const void* d_a, const void* d_b, double param, void* d_out, size_t n) {
    const int vec = ((((uintptr_t)d_a) | ((uintptr_t)d_b) | ((uintptr_t)d_out)) & 15u) == 0;
    const int grid = grid_for(ctx, dtype == MCLE_F32 && vec ? (n + 1) / 2 : n, kBlock);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL((k_binary<float, Op>), dim3(grid), dim3(kBlock), 0, ctx->stream, (const float2*)d_a,
                           (const float2*)d_b, (float)param, (float2*)d_out, n, vec);
    else
        hipLaunchKernelGGL((k_binary<double, Op>), dim3(grid), dim3(kBlock), 0, ctx->stream, (const double2*)d_a,
                           (const double2*)d_b, param, (double2*)d_out, n, 0);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

// One thread per Philox block, every word used: block b of the stream holds the complex normals 2b and 2b + 1
// (philox.hpp), so a thread owns the pair and writes it as one 16-byte store where both fall inside [first, first + n)
// and the output is aligned (a sample-per-thread form evaluated every block twice).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_randn_c(Rng rng, uint32_t stream, uint64_t first, T sigma,
                                                    cx<T>* __restrict__ out, size_t n) {
    const uint64_t b_lo = first >> 1, b_hi = (first + n - 1) >> 1;          // blocks touched
    const bool vec = sizeof(T) == 4 && ((first & 1) == 0) && (((uintptr_t)out & 15) == 0);
    for (uint64_t b = b_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= b_hi;
         b += (uint64_t)gridDim.x * blockDim.x) {
        cx<T> z0, z1;
        cn_pair<T>(rng, stream, (uint32_t)b, sigma, z0, z1);
        const uint64_t a0 = 2 * b;                                          // absolute index of z0
        const bool in0 = a0 >= first, in1 = a0 + 1 < first + n;
        if constexpr (sizeof(T) == 4) {
            if (vec && in0 && in1) {
                *reinterpret_cast<float4*>(out + (a0 - first)) = make_float4(z0.x, z0.y, z1.x, z1.y);
                continue;
            }
        }
        if (in0) out[a0 - first] = z0;
        if (in1) out[a0 + 1 - first] = z1;
    }
}

// One thread per DATA block: its sixteen bytes are symbols 16 b .. 16 b + 15 (symbol_at); four 16-byte stores.
__device__ __forceinline__ void symbols_of_block(const Rng& rng, uint64_t b, uint64_t first, uint64_t end, uint32_t mask,
                                                 int32_t* __restrict__ out) {
    const Words4 w = rng.block(STREAM_DATA, (uint32_t)b);
    const uint64_t a0 = 16 * b;
    const bool whole = a0 >= first && a0 + 16 <= end && ((((uintptr_t)(out + (a0 - first))) & 15) == 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int4 v = make_int4((int)(w.w[q] & mask), (int)((w.w[q] >> 8) & mask), (int)((w.w[q] >> 16) & mask),
                                 (int)((w.w[q] >> 24) & mask));
        if (whole) {
            *reinterpret_cast<int4*>(out + (a0 - first) + 4 * q) = v;
        } else {
            const int e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t a = a0 + 4 * q + k;
                if (a >= first && a < end) out[a - first] = e[k];
            }
        }
    }
}
__global__ __launch_bounds__(kBlock) void k_rand_symbols(Rng rng, uint64_t first, uint32_t mask,
                                                         int32_t* __restrict__ out, size_t n) {
    const uint64_t b_lo = first >> 4, b_hi = (first + n - 1) >> 4;
    for (uint64_t b = b_lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= b_hi;
         b += (uint64_t)gridDim.x * blockDim.x)
        symbols_of_block(rng, b, first, first + n, mask, out);
}

// out[r][i] = symbol i of realization first_real + r
__global__ __launch_bounds__(kBlock) void k_rand_symbols_batch(uint64_t seed, uint64_t first_real, uint32_t mask,
                                                               int32_t* __restrict__ out, size_t n) {
    const Rng rng(seed, first_real + blockIdx.y);
    int32_t* row = out + (size_t)blockIdx.y * n;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= (uint64_t)((n - 1) >> 4);
         b += (uint64_t)gridDim.x * blockDim.x)
        symbols_of_block(rng, b, 0, n, mask, row);
}

// "gen + modulate" of the staged chains in one pass (W N (I + S)).  A thread evaluates one DATA block (sixteen
// label bytes); the wavefront parks its 1 024 bytes in LDS and walks them back two symbols per lane, so that every store
// instruction covers a contiguous run (512 B of labels, 1 KiB of complex64 samples) -- sixteen labels and samples stored per
// lane straight from the block (64 B / 128 B strides across the wave) ran at a third of the rate.
template <typename T, typename L = int32_t>
__global__ __launch_bounds__(kBlock) void k_rand_modulate_batch(ModemParams<T> mp, uint64_t seed, uint64_t first_real,
                                                                uint32_t mask, L* __restrict__ idx_out,
                                                                cx<T>* __restrict__ sym_out, size_t n, int vec_ok) {
    __shared__ cx<T> s_table[kMaxM];
    __shared__ __attribute__((aligned(16))) uint32_t s_lab[kBlock * 4];
    load_table(mp, s_table);
    __syncthreads();
    const Rng rng(seed, first_real + blockIdx.y);
    L* irow = idx_out + (size_t)blockIdx.y * n;
    constexpr bool kByte = sizeof(L) == 1;         // byte labels: a thread's sixteen labels ARE its masked Philox block
    cx<T>* srow = sym_out + (size_t)blockIdx.y * n;
    const bool vec = vec_ok != 0;                  // even rows on 8-byte (labels) / 16-byte (samples) boundaries (host check)
    const uint32_t mask4 = mask * 0x01010101u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* bytes = reinterpret_cast<const unsigned char*>(s_lab) + wave * 1024;
    const uint64_t last_block = (uint64_t)((n - 1) >> 4);
    for (uint64_t b0 = (uint64_t)blockIdx.x * blockDim.x; b0 <= last_block; b0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = b0 + threadIdx.x;
        uint4 lab = make_uint4(0u, 0u, 0u, 0u);
        if (b <= last_block) {
            const Words4 w = rng.block(STREAM_DATA, (uint32_t)b);
            lab = make_uint4(w.w[0] & mask4, w.w[1] & mask4, w.w[2] & mask4, w.w[3] & mask4);
        }
        reinterpret_cast<uint4*>(s_lab)[threadIdx.x] = lab;
        if constexpr (kByte) {                     // one 16-byte store per lane, contiguous across the wavefront
            if (b <= last_block) {
                const uint64_t a0 = 16 * b;
                if (vec && a0 + 16 <= n) {
                    *reinterpret_cast<uint4*>(irow + a0) = lab;
                } else {
                    const unsigned char* lb = reinterpret_cast<const unsigned char*>(&lab);
                    for (int e = 0; e < 16 && a0 + e < n; ++e) irow[a0 + e] = (L)lb[e];
                }
            }
        }
        wave_lds_sync();                           // a wavefront reads back only its own kilobyte
        const uint64_t s_wave = 16 * (b0 + 64u * (uint64_t)wave);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int o = 128 * it + 2 * lane;
            const uint64_t a = s_wave + (uint64_t)o;
            const unsigned pair = *reinterpret_cast<const unsigned short*>(bytes + o);
            const int e0 = (int)(pair & 0xFFu), e1 = (int)(pair >> 8);
            if (vec && a + 2 <= n) {
                if constexpr (!kByte) *reinterpret_cast<int2*>(irow + a) = make_int2(e0, e1);
                const cx<T> c0 = s_table[e0], c1 = s_table[e1];
                if constexpr (sizeof(T) == 4) {
                    *reinterpret_cast<float4*>(srow + a) = make_float4(c0.x, c0.y, c1.x, c1.y);
                } else {
                    srow[a] = c0;
                    srow[a + 1] = c1;
                }
            } else {
                if (a < n) {
                    if constexpr (!kByte) irow[a] = (L)e0;
                    srow[a] = s_table[e0];
                }
                if (a + 1 < n) {
                    if constexpr (!kByte) irow[a + 1] = (L)e1;
                    srow[a + 1] = s_table[e1];
                }
            }
        }
        wave_lds_sync();                           // the next round overwrites the kilobyte
    }
}

int check_modem(const mcle_ctx* ctx, int dtype, int method) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(ctx->M > 0, "no constellation set (mcle_set_constellation)");
    MCLE_REQUIRE(ctx->M <= kMaxM, "constellation too large for the LDS table (%d > %d)", ctx->M, kMaxM);
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(method == MCLE_DEMOD_MINDIST || method == MCLE_DEMOD_QAM_SLICER, "bad demodulation method");
    MCLE_REQUIRE(method != MCLE_DEMOD_QAM_SLICER || ctx->kind == MCLE_CONST_QAM,
                 "the slicer needs a square Gray QAM constellation (kind MCLE_CONST_QAM)");
    return MCLE_OK;
}

template <typename T, bool DEMOD, typename L = int32_t>
int count_impl(mcle_ctx* ctx, int method, const void* d_rx, const L* d_a, const L* d_b,
               size_t n_per_real, size_t n_real, int bits, mcle_counters* d_counters, uint32_t* d_sym,
               uint32_t* d_bit) {
    if (n_real == 0) return MCLE_OK;
    void* ws = nullptr;
    int rc = ctx->scratch(n_real * 2 * sizeof(unsigned), &ws);
    if (rc) return rc;
    MCLE_HIP(hipMemsetAsync(ws, 0, n_real * 2 * sizeof(unsigned), ctx->stream));
    ModemParams<T> mp = modem_params<T>(ctx, method);
    const size_t chunks_needed = (n_per_real + kBlock - 1) / kBlock;
    size_t cap = (size_t)ctx->n_cu * 8;
    size_t gy = n_real < cap ? n_real : cap;
    size_t gx = cap / gy;
    if (gx < 1) gx = 1;
    if (gx > chunks_needed) gx = chunks_needed;
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)gy);
    hipLaunchKernelGGL((k_count<T, DEMOD, L>), grid, dim3(kBlock), 0, ctx->stream, mp, (const cx<T>*)d_rx, d_a, d_b,
                       n_per_real, n_real, (unsigned*)ws);
    MCLE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_count_finalize, dim3(grid_for(ctx, n_real, kBlock, 1)), dim3(kBlock), 0, ctx->stream,
                       (const unsigned*)ws, n_real, n_per_real, bits, d_counters, d_sym, d_bit);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

}  // namespace mcle

using namespace mcle;

extern "C" {

int mcle_modulate(mcle_ctx* ctx, int dtype, const int32_t* d_idx, void* d_out, size_t n) {
    int rc = check_modem(ctx, dtype, MCLE_DEMOD_MINDIST);
    if (rc) return rc;
    MCLE_REQUIRE(ctx->M <= kMaxM, "constellation too large for the LDS table (%d > %d)", ctx->M, kMaxM);
    if (n == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    void* st = nullptr;
    if ((rc = ctx->scratch(sizeof(unsigned), &st))) return rc;
    MCLE_HIP(hipMemsetAsync(st, 0, sizeof(unsigned), ctx->stream));
    const int vec = dtype == MCLE_F32 && (((uintptr_t)d_idx & 7u) | ((uintptr_t)d_out & 15u)) == 0;
    const int grid = grid_for(ctx, vec ? (n + 1) / 2 : n, kBlock);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_modulate<float>, dim3(grid), dim3(kBlock), 0, ctx->stream, modem_params<float>(ctx, 0),
                           d_idx, (float2*)d_out, n, (unsigned*)st, vec);
    else
        hipLaunchKernelGGL(k_modulate<double>, dim3(grid), dim3(kBlock), 0, ctx->stream, modem_params<double>(ctx, 0),
                           d_idx, (double2*)d_out, n, (unsigned*)st, 0);
    MCLE_LAUNCH_CHECK();
    unsigned flag = 0;
    MCLE_HIP(hipMemcpyAsync(&flag, st, sizeof(flag), hipMemcpyDeviceToHost, ctx->stream));
    MCLE_HIP(hipStreamSynchronize(ctx->stream));
    // reference: IndexError -> ValueError("Input data must be between 0 and 2^M") fundamental.py:196-199
    MCLE_REQUIRE(flag == 0, "Input data must be between 0 and 2^M");
    return MCLE_OK;
}

int mcle_demodulate(mcle_ctx* ctx, int dtype, int method, const void* d_rx, int32_t* d_idx, size_t n) {
    int rc = check_modem(ctx, dtype, method);
    if (rc) return rc;
    if (n == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    const int vec = dtype == MCLE_F32 && (((uintptr_t)d_rx & 15u) | ((uintptr_t)d_idx & 7u)) == 0;
    const int grid = grid_for(ctx, vec ? (n + 1) / 2 : n, kBlock);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_demodulate<float>, dim3(grid), dim3(kBlock), 0, ctx->stream,
                           modem_params<float>(ctx, method), (const float2*)d_rx, d_idx, n, vec);
    else
        hipLaunchKernelGGL(k_demodulate<double>, dim3(grid), dim3(kBlock), 0, ctx->stream,
                           modem_params<double>(ctx, method), (const double2*)d_rx, d_idx, n, 0);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_count_errors(mcle_ctx* ctx, const int32_t* d_tx_idx, const int32_t* d_rx_idx, size_t n_per_real,
                      size_t n_real, int bits_per_symbol, mcle_counters* d_counters, uint32_t* d_sym_err,
                      uint32_t* d_bit_err) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(bits_per_symbol >= 1 && bits_per_symbol <= 31, "bits_per_symbol out of range");
    int rc = ctx->bind();
    if (rc) return rc;
    ModemParams<float> dummy{};  // unused on the index-vs-index path (no constellation needed)
    if (n_real == 0) return MCLE_OK;
    void* ws = nullptr;
    if ((rc = ctx->scratch(n_real * 2 * sizeof(unsigned), &ws))) return rc;
    MCLE_HIP(hipMemsetAsync(ws, 0, n_real * 2 * sizeof(unsigned), ctx->stream));
    const size_t chunks_needed = (n_per_real + kBlock - 1) / kBlock;
    size_t cap = (size_t)ctx->n_cu * 8;
    size_t gy = n_real < cap ? n_real : cap;
    size_t gx = cap / gy;
    if (gx > chunks_needed) gx = chunks_needed;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((k_count<float, false>), dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, ctx->stream,
                       dummy, (const float2*)nullptr, d_tx_idx, d_rx_idx, n_per_real, n_real, (unsigned*)ws);
    MCLE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_count_finalize, dim3(grid_for(ctx, n_real, kBlock, 1)), dim3(kBlock), 0, ctx->stream,
                       (const unsigned*)ws, n_real, n_per_real, bits_per_symbol, d_counters, d_sym_err, d_bit_err);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_demod_count(mcle_ctx* ctx, int dtype, int method, const void* d_rx, const int32_t* d_tx_idx,
                     size_t n_per_real, size_t n_real, mcle_counters* d_counters, uint32_t* d_sym_err,
                     uint32_t* d_bit_err) {
    int rc = check_modem(ctx, dtype, method);
    if (rc) return rc;
    if ((rc = ctx->bind())) return rc;
    if (dtype == MCLE_F32)
        return count_impl<float, true, int32_t>(ctx, method, d_rx, d_tx_idx, (const int32_t*)nullptr, n_per_real, n_real,
                                                ctx->bits, d_counters, d_sym_err, d_bit_err);
    return count_impl<double, true, int32_t>(ctx, method, d_rx, d_tx_idx, (const int32_t*)nullptr, n_per_real, n_real,
                                             ctx->bits, d_counters, d_sym_err, d_bit_err);
}

int mcle_demod_count_u8(mcle_ctx* ctx, int dtype, int method, const void* d_rx, const uint8_t* d_tx_idx,
                        size_t n_per_real, size_t n_real, mcle_counters* d_counters, uint32_t* d_sym_err,
                        uint32_t* d_bit_err) {
    int rc = check_modem(ctx, dtype, method);
    if (rc) return rc;
    MCLE_REQUIRE(ctx->M <= 256, "byte labels: the bound constellation must have M <= 256 (got %d)", ctx->M);
    if ((rc = ctx->bind())) return rc;
    const uint8_t* none = nullptr;
    if (dtype == MCLE_F32)
        return count_impl<float, true, uint8_t>(ctx, method, d_rx, d_tx_idx, none, n_per_real, n_real, ctx->bits,
                                                d_counters, d_sym_err, d_bit_err);
    return count_impl<double, true, uint8_t>(ctx, method, d_rx, d_tx_idx, none, n_per_real, n_real, ctx->bits, d_counters,
                                             d_sym_err, d_bit_err);
}

int mcle_randn_c(mcle_ctx* ctx, int dtype, uint64_t seed, uint64_t realization, uint32_t stream,
                 uint64_t first_sample, double variance, void* d_out, size_t n) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(variance >= 0.0, "variance must be non-negative");
    if (n == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    Rng rng(seed, realization);
    const int grid = grid_for(ctx, n / 2 + 1, kBlock);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_randn_c<float>, dim3(grid), dim3(kBlock), 0, ctx->stream, rng, stream, first_sample,
                           (float)sqrt(variance), (float2*)d_out, n);
    else
        hipLaunchKernelGGL(k_randn_c<double>, dim3(grid), dim3(kBlock), 0, ctx->stream, rng, stream, first_sample,
                           sqrt(variance), (double2*)d_out, n);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_rand_symbols(mcle_ctx* ctx, uint64_t seed, uint64_t realization, uint64_t first_symbol, int M,
                      int32_t* d_idx, size_t n) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(M >= 2 && M <= 256 && (M & (M - 1)) == 0, "M must be a power of two in [2, 256]");
    if (n == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    hipLaunchKernelGGL(k_rand_symbols, dim3(grid_for(ctx, n / 16 + 1, kBlock)), dim3(kBlock), 0, ctx->stream,
                       Rng(seed, realization), first_symbol, (uint32_t)(M - 1), d_idx, n);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_rand_symbols_batch(mcle_ctx* ctx, uint64_t seed, uint64_t first_realization, uint64_t count, int M,
                            int32_t* d_idx, size_t n) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(M >= 2 && M <= 256 && (M & (M - 1)) == 0, "M must be a power of two in [2, 256]");
    MCLE_REQUIRE(count <= 65535, "at most 65535 realizations per call");
    if (n == 0 || count == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    dim3 grid((unsigned)grid_for(ctx, n / 16 + 1, kBlock, 2), (unsigned)count);
    hipLaunchKernelGGL(k_rand_symbols_batch, grid, dim3(kBlock), 0, ctx->stream, seed, first_realization,
                       (uint32_t)(M - 1), d_idx, n);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_rand_modulate_batch(mcle_ctx* ctx, int dtype, uint64_t seed, uint64_t first_realization, uint64_t count,
                             int32_t* d_idx, void* d_sym, size_t n) {
    int rc = check_modem(ctx, dtype, MCLE_DEMOD_MINDIST);
    if (rc) return rc;
    MCLE_REQUIRE((ctx->M & (ctx->M - 1)) == 0 && ctx->M >= 2 && ctx->M <= 256,
                 "the bound constellation's size must be a power of two in [2, 256] (labels are Philox bytes, as in "
                 "mcle_rand_symbols_batch; got %d)", ctx->M);
    MCLE_REQUIRE(count <= 65535, "at most 65535 realizations per call");
    MCLE_REQUIRE(d_idx != nullptr && d_sym != nullptr, "null output");
    if (n == 0 || count == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    dim3 grid((unsigned)grid_for(ctx, n / 16 + 1, kBlock, 2), (unsigned)count);
    const int vec_ok = (n & 1) == 0 && ((uintptr_t)d_idx & 7u) == 0 && ((uintptr_t)d_sym & 15u) == 0;
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_rand_modulate_batch<float>, grid, dim3(kBlock), 0, ctx->stream,
                           modem_params<float>(ctx, MCLE_DEMOD_MINDIST), seed, first_realization, (uint32_t)(ctx->M - 1),
                           d_idx, (float2*)d_sym, n, vec_ok);
    else
        hipLaunchKernelGGL(k_rand_modulate_batch<double>, grid, dim3(kBlock), 0, ctx->stream,
                           modem_params<double>(ctx, MCLE_DEMOD_MINDIST), seed, first_realization, (uint32_t)(ctx->M - 1),
                           d_idx, (double2*)d_sym, n, vec_ok);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_rand_modulate_batch_u8(mcle_ctx* ctx, int dtype, uint64_t seed, uint64_t first_realization, uint64_t count,
                                uint8_t* d_idx, void* d_sym, size_t n) {
    int rc = check_modem(ctx, dtype, MCLE_DEMOD_MINDIST);
    if (rc) return rc;
    MCLE_REQUIRE((ctx->M & (ctx->M - 1)) == 0 && ctx->M >= 2 && ctx->M <= 256,
                 "the bound constellation's size must be a power of two in [2, 256] (labels are Philox bytes; got %d)", ctx->M);
    MCLE_REQUIRE(count <= 65535, "at most 65535 realizations per call");
    MCLE_REQUIRE(d_idx != nullptr && d_sym != nullptr, "null output");
    if (n == 0 || count == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    dim3 grid((unsigned)grid_for(ctx, n / 16 + 1, kBlock, 2), (unsigned)count);
    // vector stores: rows of whole 16-label groups on 16-byte boundaries (labels and samples)
    const int vec_ok = (n & 15) == 0 && ((uintptr_t)d_idx & 15u) == 0 && ((uintptr_t)d_sym & 15u) == 0;
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL((k_rand_modulate_batch<float, uint8_t>), grid, dim3(kBlock), 0, ctx->stream,
                           modem_params<float>(ctx, MCLE_DEMOD_MINDIST), seed, first_realization, (uint32_t)(ctx->M - 1),
                           d_idx, (float2*)d_sym, n, vec_ok);
    else
        hipLaunchKernelGGL((k_rand_modulate_batch<double, uint8_t>), grid, dim3(kBlock), 0, ctx->stream,
                           modem_params<double>(ctx, MCLE_DEMOD_MINDIST), seed, first_realization, (uint32_t)(ctx->M - 1),
                           d_idx, (double2*)d_sym, n, vec_ok);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_awgn_add(mcle_ctx* ctx, int dtype, const void* d_x, const void* d_noise, double noise_var, void* d_y,
                  size_t n) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(noise_var >= 0.0, "noise variance must be non-negative");
    if (n == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    return launch_binary<OpAwgn>(ctx, dtype, d_x, d_noise, sqrt(noise_var), d_y, n);
}

int mcle_cmul(mcle_ctx* ctx, int dtype, const void* d_a, const void* d_b, void* d_out, size_t n) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    if (n == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    return launch_binary<OpMul>(ctx, dtype, d_a, d_b, 0.0, d_out, n);
}

int mcle_cdiv(mcle_ctx* ctx, int dtype, const void* d_num, const void* d_den, void* d_out, size_t n) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    if (n == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    return launch_binary<OpDiv>(ctx, dtype, d_num, d_den, 0.0, d_out, n);
}

}  // extern "C"
