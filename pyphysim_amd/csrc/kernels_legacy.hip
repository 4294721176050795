// kernels_legacy.hip -- NumPy's legacy global RandomState on the GPU ("same-seed" parity mode).
//
// The reference draws everything from np.random's legacy MT19937 stream:
//   np.random.seed(s)           init_genrand(s): mt[0] = s, mt[i] = 1812433253 (mt[i-1] ^ mt[i-1]>>30) + i
//   randint(0, 2^k, n)          n tempered 32-bit outputs, each & (2^k - 1)
//   rand() / random_sample()    two outputs a, b:  ((a >> 5) * 2^26 + (b >> 6)) / 2^53
//   randn()                     Marsaglia polar method on pairs of such doubles (x = 2u - 1), rejecting
//                               r2 >= 1 or r2 == 0; returns f*x2 first and caches f*x1 (f = sqrt(-2 ln r2 / r2));
//                               the cache survives randint calls.
// One wavefront per realization replays that stream for `np.random.seed(seed_base + r)` and a
// short program of randint / randn segments, writing the draws to device arrays that the
// per-operator kernels then consume.  The rejection step makes the normal stream's word
// positions data dependent; it is resolved 64 candidates at a time with a ballot + rank.
#include "common.hpp"

namespace mcle {

constexpr int kMtN = 624, kMtM = 397;
constexpr int kMaxSegs = 8;

struct LegacySeg {
    int kind;  // 0: randint with power-of-two range (mask), 1: randn (doubles), 2: rand (doubles)
    int n;
    uint32_t mask;
    int offset;  // into the realization's int or double output row
};
struct LegacyProgram {
    int nseg;
    LegacySeg seg[kMaxSegs];
    int n_int, n_dbl;        // row lengths of the outputs
    int words_cap;           // multiple of 624
};

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__global__ __launch_bounds__(64) void k_legacy_draws(LegacyProgram prog, uint32_t seed_base, uint64_t first,
                                                     uint64_t count, uint32_t* __restrict__ g_words,
                                                     int32_t* __restrict__ g_int, double* __restrict__ g_dbl,
                                                     uint32_t* __restrict__ g_status) {
    __shared__ uint32_t mt[kMtN];
    const int lane = threadIdx.x;
    for (uint64_t rl = blockIdx.x; rl < count; rl += gridDim.x) {
        uint32_t* words = g_words + rl * (uint64_t)prog.words_cap;
        int32_t* out_i = g_int + rl * (uint64_t)prog.n_int;
        double* out_d = g_dbl + rl * (uint64_t)prog.n_dbl;
        __syncthreads();
        // ---- init_genrand(seed): inherently serial ----
        if (lane == 0) {
            uint32_t v = seed_base + (uint32_t)(first + rl);
            mt[0] = v;
            for (int i = 1; i < kMtN; ++i) {
                v = 1812433253u * (v ^ (v >> 30)) + (uint32_t)i;
                mt[i] = v;
            }
        }
        __syncthreads();
        // ---- phase A: all the words this realization can need, tempered, to global memory ----
        for (int blk = 0; blk < prog.words_cap / kMtN; ++blk) {
            // twist: lanes walk kk = lane, lane+64, ... in order; a single in-order wavefront thereby
            // reads exactly the old / new values the sequential algorithm would
            for (int it = 0; it < (kMtN + 63) / 64; ++it) {
                const int kk = lane + 64 * it;
                uint32_t nv = 0;
                if (kk < kMtN) {
                    const uint32_t y = (mt[kk] & 0x80000000u) | (mt[(kk + 1) % kMtN] & 0x7fffffffu);
                    nv = mt[(kk + kMtM) % kMtN] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                }
                __syncthreads();
                if (kk < kMtN) mt[kk] = nv;
                __syncthreads();
            }
            for (int i = lane; i < kMtN; i += 64) words[blk * kMtN + i] = mt_temper(mt[i]);
        }
        __syncthreads();
        // ---- phase B: replay the draw program ----
        int p = 0;                 // word position
        bool overflow = false;
        bool has_cached = false;   // legacy gauss cache
        double cached = 0.0;
        for (int sgi = 0; sgi < prog.nseg; ++sgi) {
            const LegacySeg sg = prog.seg[sgi];
            if (sg.kind == 0) {
                if (p + sg.n > prog.words_cap) {
                    overflow = true;
                    break;
                }
                for (int i = lane; i < sg.n; i += 64) out_i[sg.offset + i] = (int32_t)(words[p + i] & sg.mask);
                p += sg.n;
                continue;
            }
            if (sg.kind == 2) {  // rand(): 53-bit doubles from two words each
                if (p + 2 * sg.n > prog.words_cap) {
                    overflow = true;
                    break;
                }
                for (int i = lane; i < sg.n; i += 64) {
                    const uint32_t a = words[p + 2 * i], b = words[p + 2 * i + 1];
                    out_d[sg.offset + i] = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
                }
                p += 2 * sg.n;
                continue;
            }
            int produced = 0;
            if (has_cached && sg.n > 0) {
                if (lane == 0) out_d[sg.offset] = cached;
                produced = 1;
                has_cached = false;
            }
            while (produced < sg.n) {
                if (p + 256 > prog.words_cap) {
                    overflow = true;
                    break;
                }
                const uint32_t a0 = words[p + 4 * lane], b0 = words[p + 4 * lane + 1];
                const uint32_t a1 = words[p + 4 * lane + 2], b1 = words[p + 4 * lane + 3];
                const double u1 = ((double)(a0 >> 5) * 67108864.0 + (double)(b0 >> 6)) / 9007199254740992.0;
                const double u2 = ((double)(a1 >> 5) * 67108864.0 + (double)(b1 >> 6)) / 9007199254740992.0;
                const double x1 = 2.0 * u1 - 1.0, x2 = 2.0 * u2 - 1.0;
                const double r2 = __dadd_rn(__dmul_rn(x1, x1), __dmul_rn(x2, x2));
                const bool acc = !(r2 >= 1.0 || r2 == 0.0);
                const unsigned long long mask = __ballot(acc);
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                const int need_pairs = (sg.n - produced + 1) / 2;   // pairs still to take from this chunk
                const int got = __popcll(mask);
                const int take = got < need_pairs ? got : need_pairs;
                if (acc && rank < take) {
                    const double f = sqrt(-2.0 * log(r2) / r2);
                    const int o = produced + 2 * rank;
                    out_d[sg.offset + o] = f * x2;                       // returned first
                    if (o + 1 < sg.n)
                        out_d[sg.offset + o + 1] = f * x1;               // the cached one, returned next
                }
                if (take == need_pairs) {
                    // the chunk satisfies the segment: advance to just after the last candidate used
                    // lane index of the take-th accepted candidate
                    unsigned long long m = mask;
                    for (int k = 1; k < take; ++k) m &= m - 1ull;        // drop the lowest take-1 set bits
                    const int last = __ffsll((long long)m) - 1;
                    const int total = produced + 2 * take;
                    if (total > sg.n) {
                        // odd request: the pair's second value stays cached for the next randn
                        const int src = last;
                        const double f = sqrt(-2.0 * log(r2) / r2);
                        cached = __shfl(f * x1, src, 64);
                        has_cached = true;
                    }
                    produced = sg.n;
                    p += 4 * (last + 1);
                } else {
                    produced += 2 * take;
                    p += 256;
                }
            }
            if (overflow) break;
        }
        if (lane == 0 && g_status) g_status[rl] = overflow ? 1u : 0u;
    }
}

// out[i] = scale * (re[i] + 1j * im[i])  -- randn_c's (1/sqrt(2)) * (randn + 1j*randn), misc.py:354-355
template <typename T>
__global__ __launch_bounds__(256) void k_complex_from_parts(const double* __restrict__ re, const double* __restrict__ im,
                                                            double scale, cx<T>* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = mk<T>((T)__dmul_rn(scale, re[i]), (T)__dmul_rn(scale, im[i]));
}

}  // namespace mcle

using namespace mcle;

extern "C" {

int mcle_complex_from_parts(mcle_ctx* ctx, int dtype, const double* d_re, const double* d_im, double scale,
                            void* d_out, size_t n) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    if (n == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    const int grid = grid_for(ctx, n, 256);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_complex_from_parts<float>, dim3(grid), dim3(256), 0, ctx->stream, d_re, d_im, scale,
                           (float2*)d_out, n);
    else
        hipLaunchKernelGGL(k_complex_from_parts<double>, dim3(grid), dim3(256), 0, ctx->stream, d_re, d_im, scale,
                           (double2*)d_out, n);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_legacy_draws(mcle_ctx* ctx, const mcle_legacy_seg* segs, int n_segs, uint32_t seed_base, uint64_t first,
                      uint64_t count, int32_t* d_int, size_t n_int, double* d_dbl, size_t n_dbl,
                      uint32_t* d_status) {
    MCLE_REQUIRE(ctx != nullptr && segs != nullptr, "null argument");
    MCLE_REQUIRE(n_segs >= 1 && n_segs <= kMaxSegs, "between 1 and %d draw segments", kMaxSegs);
    LegacyProgram prog;
    prog.nseg = n_segs;
    size_t ni = 0, nd = 0, words = 0;
    for (int i = 0; i < n_segs; ++i) {
        MCLE_REQUIRE(segs[i].n >= 0, "negative draw count");
        prog.seg[i].kind = segs[i].kind;
        prog.seg[i].n = segs[i].n;
        if (segs[i].kind == 0) {
            const uint32_t r = segs[i].range;
            MCLE_REQUIRE(r >= 2 && (r & (r - 1)) == 0, "randint range must be a power of two (got %u)", r);
            prog.seg[i].mask = r - 1;
            prog.seg[i].offset = (int)ni;
            ni += segs[i].n;
            words += segs[i].n;
        } else if (segs[i].kind == 1) {
            prog.seg[i].mask = 0;
            prog.seg[i].offset = (int)nd;
            nd += segs[i].n;
            // 4 words per candidate pair, acceptance pi/4; 40 % head room + one chunk per segment
            words += (size_t)((segs[i].n / 2 + 1) * 4 * 1.8) + 512;
        } else if (segs[i].kind == 2) {
            prog.seg[i].mask = 0;
            prog.seg[i].offset = (int)nd;
            nd += segs[i].n;
            words += 2 * (size_t)segs[i].n;
        } else {
            set_error("unknown segment kind %d", segs[i].kind);
            return MCLE_E_INVAL;
        }
    }
    MCLE_REQUIRE(ni <= n_int && nd <= n_dbl, "output rows too short (need %zu ints, %zu doubles)", ni, nd);
    MCLE_REQUIRE(count <= 1u << 20, "at most 2^20 realizations per call in the legacy mode");
    if (count == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    prog.n_int = (int)n_int;
    prog.n_dbl = (int)n_dbl;
    prog.words_cap = (int)(((words + kMtN - 1) / kMtN) * kMtN);
    void* d_words = nullptr;
    if ((rc = ctx->scratch((size_t)count * prog.words_cap * sizeof(uint32_t), &d_words))) return rc;
    const uint64_t cap = (uint64_t)ctx->n_cu * 16;
    hipLaunchKernelGGL(k_legacy_draws, dim3((unsigned)(count < cap ? count : cap)), dim3(64), 0, ctx->stream, prog,
                       seed_base, first, count, (uint32_t*)d_words, d_int, d_dbl, d_status);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

}  // extern "C"
