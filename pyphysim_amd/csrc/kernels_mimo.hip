// kernels_mimo.hip -- Blast encode / receive filter / decode and the flat MIMO channel, batched.
// Reference: mimo/mimo.py:609-660 (encode / decode, Fortran-order (de)interleave), :264-309 and
// :597-607 (filters), apps/mimo/simulate_mimo.py:96-98 (received = H @ X + noise).
#include "mimo.hpp"
#include "mimo_svd.hpp"
#include "philox.hpp"

namespace mcle {

constexpr int kMimoBlock = 256;

// X[b][a][c] = x[b][c*nt + a] / sqrt(nt)
template <typename T>
__global__ __launch_bounds__(kMimoBlock) void k_blast_encode(const cx<T>* __restrict__ x, int nt, size_t ns,
                                                             T inv_root_nt, cx<T>* __restrict__ X) {
    const size_t b = blockIdx.y;
    const size_t n = ns * nt;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t a = i / ns, c = i - a * ns;  // output-major: coalesced writes
        X[b * n + i] = cscale(x[b * n + c * nt + a], inv_root_nt);
    }
}

// nt = 2 / 4, ns even, aligned rows: a thread takes two columns -- 2 nt consecutive inputs in 16-byte (32-byte) reads,
// one pair per antenna row out -- instead of one strided 8-byte read per output.
template <typename T, int NT>
__global__ __launch_bounds__(kMimoBlock) void k_blast_encode_pairs(const cx<T>* __restrict__ x, size_t ns, T inv_root_nt,
                                                                   cx<T>* __restrict__ X) {
    struct alignas(2 * sizeof(cx<T>)) Pair {
        cx<T> a, b;
    };
    const size_t b = blockIdx.y, n = ns * NT;
    const Pair* xin = reinterpret_cast<const Pair*>(x + b * n);
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < ns / 2; p += (size_t)gridDim.x * blockDim.x) {
        Pair v[NT];                                   // columns 2p and 2p + 1: inputs (2p) NT .. (2p + 2) NT - 1
#pragma unroll
        for (int k = 0; k < NT; ++k) v[k] = xin[p * NT + k];
        const cx<T>* e = reinterpret_cast<const cx<T>*>(v);
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            Pair o;
            o.a = cscale(e[a], inv_root_nt);
            o.b = cscale(e[NT + a], inv_root_nt);
            reinterpret_cast<Pair*>(X + b * n + (size_t)a * ns)[p] = o;
        }
    }
}

template <typename T, int NT, int NR>
__global__ __launch_bounds__(64) void k_blast_filter(const cx<T>* __restrict__ Hg, double nv,
                                                     cx<T>* __restrict__ Gg, uint32_t* __restrict__ skipped,
                                                     size_t batch) {
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (size_t)gridDim.x * blockDim.x) {
        double2 H[NR][NT], G[NT][NR];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const cx<T> v = Hg[(b * NR + r) * NT + a];
                H[r][a] = mk<double>((double)v.x, (double)v.y);
            }
        const bool ok = blast_filter<NT, NR>(H, nv, G);
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int r = 0; r < NR; ++r) Gg[(b * NT + a) * NR + r] = mk<T>((T)G[a][r].x, (T)G[a][r].y);
        if (skipped) skipped[b] = ok ? 0u : 1u;
    }
}

// The same filters with coalesced traffic: a wavefront's 64 matrices are one contiguous run of 64 E complex values;
// the run crosses LDS in 16-byte chunks (matrix stride padded by 4 dwords: lane-per-matrix b128 accesses then hit all
// 32 banks once per 8 lanes), so every global access is a full 1 KiB wave access instead of 64 strided 8-byte ones.
// Needs an even number of 4-byte words per matrix pair (E even, or complex128) and 16-byte aligned arrays.
template <typename T, int NT, int NR>
__global__ __launch_bounds__(64) void k_blast_filter_staged(const cx<T>* __restrict__ Hg, double nv,
                                                            cx<T>* __restrict__ Gg, uint32_t* __restrict__ skipped,
                                                            size_t batch) {
    constexpr int E = NT * NR;
    constexpr int DW = E * (int)sizeof(cx<T>) / 4;        // dwords per matrix
    constexpr int CH = DW / 4;                              // 16-byte chunks per matrix
    constexpr int STRIDE = DW + 4;
    static_assert(DW % 4 == 0, "matrix size must be a multiple of 16 bytes");
    __shared__ __attribute__((aligned(16))) float s_m[64 * STRIDE];
    const int lane = threadIdx.x;
    for (size_t b0 = (size_t)blockIdx.x * 64; b0 < batch; b0 += (size_t)gridDim.x * 64) {
        const size_t left = batch - b0 < 64 ? batch - b0 : 64;          // matrices of this step
        const float4* src = reinterpret_cast<const float4*>(Hg + b0 * E);
        float4 in[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int q = k * 64 + lane;
            in[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((size_t)q < left * CH) in[k] = src[q];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int q = k * 64 + lane;
            *reinterpret_cast<float4*>(s_m + (q / CH) * STRIDE + 4 * (q % CH)) = in[k];
        }
        __syncthreads();
        cx<T> hm[E];
        {
            float4* d = reinterpret_cast<float4*>(hm);
#pragma unroll
            for (int k = 0; k < CH; ++k) d[k] = *reinterpret_cast<const float4*>(s_m + lane * STRIDE + 4 * k);
        }
        double2 H[NR][NT], G[NT][NR];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int a = 0; a < NT; ++a) H[r][a] = mk<double>((double)hm[r * NT + a].x, (double)hm[r * NT + a].y);
        const bool ok = blast_filter<NT, NR>(H, nv, G);
        cx<T> gm[E];
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int r = 0; r < NR; ++r) gm[a * NR + r] = mk<T>((T)G[a][r].x, (T)G[a][r].y);
        __syncthreads();
        {
            const float4* d = reinterpret_cast<const float4*>(gm);
#pragma unroll
            for (int k = 0; k < CH; ++k) *reinterpret_cast<float4*>(s_m + lane * STRIDE + 4 * k) = d[k];
        }
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(Gg + b0 * E);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int q = k * 64 + lane;
            if ((size_t)q < left * CH) dst[q] = *reinterpret_cast<const float4*>(s_m + (q / CH) * STRIDE + 4 * (q % CH));
        }
        if (skipped && (size_t)lane < left) skipped[b0 + lane] = ok ? 0u : 1u;
    }
}

// est[b][c*nt + a] = sum_r G[b][a][r] Y[b][r][c]
template <typename T>
__global__ __launch_bounds__(kMimoBlock) void k_blast_decode(const cx<T>* __restrict__ G, const cx<T>* __restrict__ Y,
                                                             int nr, int nt, size_t ns, cx<T>* __restrict__ est) {
    const size_t b = blockIdx.y;
    const cx<T>* Gb = G + b * (size_t)nt * nr;
    const cx<T>* Yb = Y + b * (size_t)nr * ns;
    cx<T>* eb = est + b * (size_t)nt * ns;
    // one thread per column; the nt interleaved results of a column are contiguous in memory and leave as
    // 16-byte stores (four strided 8-byte streams per wave were the bottleneck of the first version)
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ns; c += (size_t)gridDim.x * blockDim.x) {
        if (nt == 4 && nr == 4) {
            cx<T> y[4], e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = Yb[(size_t)r * ns + c];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                cx<T> acc = mk<T>(0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = cfma(Gb[a * 4 + r], y[r], acc);
                e[a] = acc;
            }
            if constexpr (sizeof(T) == 4) {
                float4* o4 = reinterpret_cast<float4*>(eb + c * 4);
                o4[0] = make_float4(e[0].x, e[0].y, e[1].x, e[1].y);
                o4[1] = make_float4(e[2].x, e[2].y, e[3].x, e[3].y);
            } else {
#pragma unroll
                for (int a = 0; a < 4; ++a) eb[c * 4 + a] = e[a];
            }
        } else {
            for (int a = 0; a < nt; ++a) {
                cx<T> acc = mk<T>(0, 0);
                for (int r = 0; r < nr; ++r) acc = cfma(Gb[a * nr + r], Yb[(size_t)r * ns + c], acc);
                eb[c * nt + a] = acc;
            }
        }
    }
}

// per-subcarrier decode: est[c*nt + a] = sum_r G[c][a][r] Y[r][c]  (one receive filter per column c)
template <typename T>
__global__ __launch_bounds__(kMimoBlock) void k_blast_decode_persc(const cx<T>* __restrict__ G,
                                                                   const cx<T>* __restrict__ Y, int nr, int nt,
                                                                   size_t ns, cx<T>* __restrict__ est) {
    G += (size_t)blockIdx.y * ns * nt * nr;    // batch item: G [b][ns][nt][nr], Y [b][nr][ns], est [b][ns*nt]
    Y += (size_t)blockIdx.y * nr * ns;
    est += (size_t)blockIdx.y * ns * nt;
    // one thread per output (c, a): its filter row G[c][a][:] is contiguous and consecutive threads read
    // consecutive rows, so the G stream (the bulk of the traffic) is fully coalesced
    const size_t total = ns * (size_t)nt;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const size_t c = o / nt;
        const cx<T>* Ga = G + o * (size_t)nr;
        cx<T> acc = mk<T>(0, 0);
        for (int r = 0; r < nr; ++r) acc = cfma(Ga[r], Y[(size_t)r * ns + c], acc);
        est[o] = acc;
    }
}

// Y[b][r][c] = sum_a H[b][r][a] X[b][a][c] (+ sigma * noise[b][r][c])
template <typename T>
__global__ __launch_bounds__(kMimoBlock) void k_mimo_channel(const cx<T>* __restrict__ H, const cx<T>* __restrict__ X,
                                                             const cx<T>* __restrict__ nz, T sigma, int nr, int nt,
                                                             size_t ns, cx<T>* __restrict__ Y, int vec) {
    const size_t b = blockIdx.y;
    const cx<T>* Hb = H + b * (size_t)nr * nt;
    const cx<T>* Xb = X + b * (size_t)nt * ns;
    if constexpr (sizeof(T) == 4) {
        if (vec) {   // 4x4, even ns, 16-byte aligned rows: two columns per thread, every stream in 16-byte accesses
            float2 Hr[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < 4; ++a) Hr[r][a] = Hb[r * 4 + a];
            for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < ns / 2; p += (size_t)gridDim.x * blockDim.x) {
                float4 x[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) x[a] = reinterpret_cast<const float4*>(Xb + (size_t)a * ns)[p];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float2 a0 = make_float2(0.f, 0.f), a1 = make_float2(0.f, 0.f);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        a0 = cfma(Hr[r][a], make_float2(x[a].x, x[a].y), a0);
                        a1 = cfma(Hr[r][a], make_float2(x[a].z, x[a].w), a1);
                    }
                    const size_t row = (b * 4 + r) * ns;
                    if (nz) {
                        const float4 w = reinterpret_cast<const float4*>(nz + row)[p];
                        a0.x += sigma * w.x;
                        a0.y += sigma * w.y;
                        a1.x += sigma * w.z;
                        a1.y += sigma * w.w;
                    }
                    reinterpret_cast<float4*>(Y + row)[p] = make_float4(a0.x, a0.y, a1.x, a1.y);
                }
            }
            return;
        }
    }
    if (nr == 4 && nt == 4) {   // H in registers, every column of X read once (the loop below reads it once per output row)
        cx<T> Hr[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < 4; ++a) Hr[r][a] = Hb[r * 4 + a];
        for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ns; c += (size_t)gridDim.x * blockDim.x) {
            cx<T> x[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) x[a] = Xb[(size_t)a * ns + c];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cx<T> acc = mk<T>(0, 0);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc = cfma(Hr[r][a], x[a], acc);
                const size_t o = (b * 4 + r) * ns + c;
                if (nz) {
                    const cx<T> w = nz[o];
                    acc.x += sigma * w.x;
                    acc.y += sigma * w.y;
                }
                Y[o] = acc;
            }
        }
        return;
    }
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ns; c += (size_t)gridDim.x * blockDim.x) {
        for (int r = 0; r < nr; ++r) {
            cx<T> acc = mk<T>(0, 0);
            for (int a = 0; a < nt; ++a) acc = cfma(Hb[r * nt + a], Xb[(size_t)a * ns + c], acc);
            const size_t o = (b * nr + r) * ns + c;
            if (nz) {
                const cx<T> w = nz[o];
                acc.x += sigma * w.x;
                acc.y += sigma * w.y;
            }
            Y[o] = acc;
        }
    }
}

// Y[b][r][c] = sum_a H[b][r][a] X[b][a][c] + sigma * CN(0,1) sample r*ns + c of (seed, first + b, NOISE): two columns per
// thread = one Philox block per (row, thread) when r*ns is even (every word used); 4x4 complex64 in 16-byte accesses
template <typename T>
__global__ __launch_bounds__(kMimoBlock) void k_mimo_channel_philox(const cx<T>* __restrict__ H, const cx<T>* __restrict__ X,
                                                                    uint64_t seed, uint64_t first, T sigma, int nr, int nt,
                                                                    size_t ns, cx<T>* __restrict__ Y, int vec) {
    const size_t b = blockIdx.y;
    const Rng rng(seed, first + b);
    const cx<T>* Hb = H + b * (size_t)nr * nt;
    const cx<T>* Xb = X + b * (size_t)nt * ns;
    if constexpr (sizeof(T) == 4) {
        if (vec) {
            float2 Hr[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < 4; ++a) Hr[r][a] = Hb[r * 4 + a];
            for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < ns / 2; p += (size_t)gridDim.x * blockDim.x) {
                float4 x[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) x[a] = reinterpret_cast<const float4*>(Xb + (size_t)a * ns)[p];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float2 a0, a1;
                    cn_pair<float>(rng, STREAM_NOISE, (uint32_t)(((size_t)r * ns) / 2 + p), sigma, a0, a1);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        a0 = cfma(Hr[r][a], make_float2(x[a].x, x[a].y), a0);
                        a1 = cfma(Hr[r][a], make_float2(x[a].z, x[a].w), a1);
                    }
                    reinterpret_cast<float4*>(Y + (b * 4 + r) * ns)[p] = make_float4(a0.x, a0.y, a1.x, a1.y);
                }
            }
            return;
        }
    }
    if constexpr (sizeof(T) == 8) {
        // complex128, 4 x 4, even ns: the same shape as the complex64 form above -- the two columns of all four transmit rows are
        // loaded ONCE (round 6: the generic loop below re-read them per receive row and moved 313 kB per realization of config 4
        // where the operator's own traffic is 133 kB: the 19 % surplus of the complex128 staged chain)
        if (vec) {
            // (the Box-Muller tables from the workgroup's LDS copy, as in the fused pipelines: 8 320 samples x 3 table reads per
            //  realization went to L2 as scattered 16-byte requests)
            __shared__ double s_bm[kBmLdsDoubles];
            bm_tables_to_lds(s_bm, (int)threadIdx.x, (int)blockDim.x);
            __syncthreads();
            double2 Hr[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int a = 0; a < 4; ++a) Hr[r][a] = Hb[r * 4 + a];
            // ONE column per lane: a wavefront's load of a transmit row is 1 KiB of consecutive bytes (two columns per lane left every
            // load instruction using half of each cache line it touched, and the counter showed the rows fetched 2.8 times).  The
            // columns 2 p and 2 p + 1 of a row are one Philox block: of the lane pair (2 p, 2 p + 1) the even lane evaluates the blocks
            // of rows 0 and 1, the odd lane those of rows 2 and 3, and a lane exchange hands over the two words the partner needs.
            const size_t cols = ns;
            const size_t stride = (size_t)gridDim.x * blockDim.x;                 // even (blockDim.x is)
            for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ((cols + stride - 1) / stride) * stride; c += stride) {
                const bool live = c < cols;                                       // (the exchange needs both lanes of a pair in the loop)
                const size_t cc = live ? c : cols - 2 + (c & 1);
                const int odd = (int)(cc & 1);
                double2 x[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) x[a] = Xb[(size_t)a * ns + cc];
                uint32_t w[4][2];                                                 // [row][2 words] of THIS lane's sample
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int r_mine = 2 * odd + k;                               // the row whose block this lane evaluates
                    const Words4 blk = rng.block(STREAM_NOISE, (uint32_t)(((size_t)r_mine * ns) / 2 + cc / 2));
                    const uint32_t keep0 = odd ? blk.w[2] : blk.w[0], keep1 = odd ? blk.w[3] : blk.w[1];
                    const uint32_t give0 = odd ? blk.w[0] : blk.w[2], give1 = odd ? blk.w[1] : blk.w[3];
                    const uint32_t got0 = (uint32_t)__shfl_xor((int)give0, 1, 64), got1 = (uint32_t)__shfl_xor((int)give1, 1, 64);
                    // rows 2 odd + k are mine; the partner evaluated rows 2 (1 - odd) + k and gave me my words of those
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if ((r & 1) == k) {                                       // r = k or k + 2 (compile-time pair, run-time choice)
                            const bool mine = (r >> 1) == odd;
                            w[r][0] = mine ? keep0 : got0;
                            w[r][1] = mine ? keep1 : got1;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 z = cn_from_words_lds(w[r][0], w[r][1], sigma, s_bm);
                    double2 s0 = mk<double>(0, 0);
#pragma unroll
                    for (int a = 0; a < 4; ++a) s0 = cfma(Hr[r][a], x[a], s0);
                    if (live) Y[(b * 4 + r) * ns + cc] = cadd(s0, z);            // (sum first, then the noise: the generic loop's association)
                }
            }
            return;
        }
    }
    const size_t pairs = (ns + 1) / 2;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < pairs; p += (size_t)gridDim.x * blockDim.x) {
        const size_t c0 = 2 * p;
        const bool two = c0 + 1 < ns;
        for (int r = 0; r < nr; ++r) {
            const uint64_t i0 = (uint64_t)r * ns + c0;
            cx<T> z0, z1;
            if ((i0 & 1) == 0) {
                cn_pair<T>(rng, STREAM_NOISE, (uint32_t)(i0 >> 1), sigma, z0, z1);
            } else {
                z0 = cn_sample<T>(rng, STREAM_NOISE, i0, sigma);
                z1 = two ? cn_sample<T>(rng, STREAM_NOISE, i0 + 1, sigma) : z0;
            }
            cx<T> acc0 = mk<T>(0, 0), acc1 = mk<T>(0, 0);
            for (int a = 0; a < nt; ++a) {
                const cx<T> h = Hb[r * nt + a];
                acc0 = cfma(h, Xb[(size_t)a * ns + c0], acc0);
                if (two) acc1 = cfma(h, Xb[(size_t)a * ns + c0 + 1], acc1);
            }
            const size_t o = (b * nr + r) * ns + c0;
            Y[o] = cadd(acc0, z0);
            if (two) Y[o + 1] = cadd(acc1, z1);
        }
    }
}

// ---- Alamouti (mimo.py:1168-1269): x[2i], x[2i+1] -> [[x0, -x1*], [x1, x0*]] / sqrt(2) --------------
template <typename T>
__global__ __launch_bounds__(kMimoBlock) void k_alamouti_encode(const cx<T>* __restrict__ x, size_t n, T scale,
                                                                cx<T>* __restrict__ X) {
    const size_t b = blockIdx.y;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * blockDim.x) {
        const cx<T> s0 = x[b * n + 2 * i], s1 = x[b * n + 2 * i + 1];
        cx<T>* r0 = X + (b * 2) * n;
        cx<T>* r1 = r0 + n;
        r0[2 * i] = cscale(s0, scale);
        r0[2 * i + 1] = cscale(mk<T>(-s1.x, s1.y), scale);
        r1[2 * i] = cscale(s1, scale);
        r1[2 * i + 1] = cscale(cconj(s0), scale);
    }
}

template <typename T> struct alignas(2 * sizeof(cx<T>)) CxPair {
    cx<T> a, b;
};

// d[2i] = h0^H y[:,2i] + h1^T conj(y[:,2i+1]);  d[2i+1] = h1^H y[:,2i] - h0^T conj(y[:,2i+1]);  / |H|_F^2 * sqrt(2)
template <typename T>
__global__ __launch_bounds__(kMimoBlock) void k_alamouti_decode(const cx<T>* __restrict__ H, const cx<T>* __restrict__ Y,
                                                                int nr, size_t n, T root2, cx<T>* __restrict__ out,
                                                                bool vec) {
    const size_t b = blockIdx.y;
    const cx<T>* Hb = H + b * (size_t)nr * 2;
    const cx<T>* Yb = Y + b * (size_t)nr * n;
    // numpy: norm(H, 'fro')**2 = (sqrt(sum |h|^2))^2
    T acc = 0;
    for (int r = 0; r < nr; ++r)
        for (int a = 0; a < 2; ++a) acc += Hb[r * 2 + a].x * Hb[r * 2 + a].x + Hb[r * 2 + a].y * Hb[r * 2 + a].y;
    const T nrm = sqrt(acc);
    const T fro2 = nrm * nrm;
    const T g = root2 / fro2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * blockDim.x) {
        cx<T> a0 = mk<T>(0, 0), a1 = mk<T>(0, 0), b0 = mk<T>(0, 0), b1 = mk<T>(0, 0);
        for (int r = 0; r < nr; ++r) {
            const cx<T> h0 = Hb[r * 2], h1 = Hb[r * 2 + 1];
            cx<T> y0, y1c;
            if (vec) {     // the pair in one 16-byte (32-byte) access: n is even, so a row's pairs are aligned with the base
                const CxPair<T> yy = reinterpret_cast<const CxPair<T>*>(Yb + (size_t)r * n)[i];
                y0 = yy.a;
                y1c = cconj(yy.b);
            } else {
                y0 = Yb[(size_t)r * n + 2 * i];
                y1c = cconj(Yb[(size_t)r * n + 2 * i + 1]);
            }
            a0 = cadd(a0, cmul(cconj(h0), y0));
            a1 = cadd(a1, cmul(h1, y1c));
            b0 = cadd(b0, cmul(cconj(h1), y0));
            b1 = cadd(b1, cmul(mk<T>(-h0.x, -h0.y), y1c));
        }
        const cx<T> d0 = cadd(a0, a1), d1 = cadd(b0, b1);
        CxPair<T> o;
        if (sizeof(T) == 8) {       // the reference's order of operations: / |H|_F^2, then * sqrt(2)
            o.a = mk<T>(d0.x / fro2 * root2, d0.y / fro2 * root2);
            o.b = mk<T>(d1.x / fro2 * root2, d1.y / fro2 * root2);
        } else {
            o.a = mk<T>(d0.x * g, d0.y * g);
            o.b = mk<T>(d1.x * g, d1.y * g);
        }
        if (vec) {
            reinterpret_cast<CxPair<T>*>(out + b * n)[i] = o;
        } else {
            out[b * n + 2 * i] = o.a;
            out[b * n + 2 * i + 1] = o.b;
        }
    }
}

// ---- MRT (mimo.py:666-783), MISO 1 x Nt: W = exp(-1j angle(h)).T / sqrt(Nt); G = sqrt(Nt) / sum|h| ------
template <typename T>
__global__ __launch_bounds__(kMimoBlock) void k_mrt_encode(const cx<T>* __restrict__ h, const cx<T>* __restrict__ x,
                                                           int nt, size_t n, cx<T>* __restrict__ X) {
    __shared__ cx<T> s_w[64];                               // nt <= 64 (check_mimo): the precoder, once per workgroup
    const size_t b = blockIdx.y;
    if ((int)threadIdx.x < nt) {
        const double inv = 1.0 / sqrt((double)nt);
        const cx<T> ha = h[b * nt + threadIdx.x];
        double sn, cs;
        sincos(-atan2((double)ha.y, (double)ha.x), &sn, &cs);
        s_w[threadIdx.x] = mk<T>((T)(cs * inv), (T)(sn * inv));
    }
    __syncthreads();
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (size_t)gridDim.x * blockDim.x) {
        const cx<T> xv = x[b * n + c];                      // read once, written to every antenna's row
        for (int a = 0; a < nt; ++a) X[(b * nt + a) * n + c] = cmul(s_w[a], xv);
    }
}
template <typename T>
__global__ __launch_bounds__(kMimoBlock) void k_mrt_decode(const cx<T>* __restrict__ h, const cx<T>* __restrict__ y,
                                                           int nt, size_t n, cx<T>* __restrict__ out) {
    const size_t b = blockIdx.y;
    T sum = 0;
    for (int a = 0; a < nt; ++a) sum += sqrt(h[b * nt + a].x * h[b * nt + a].x + h[b * nt + a].y * h[b * nt + a].y);
    const T g = (T)sqrt((double)nt) / sum;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[b * n + i] = cscale(y[b * n + i], g);
}

// SVDMimo (mimo.py:833-946): W = V / sqrt(Nt), G = diag(1/S) U^H sqrt(Nt) for square H.
template <typename T, int NA>
__global__ __launch_bounds__(64) void k_svd_filters(const cx<T>* __restrict__ Hg, cx<T>* __restrict__ Wg,
                                                    cx<T>* __restrict__ Gg, double* __restrict__ Sg, size_t batch) {
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (size_t)gridDim.x * blockDim.x) {
        double2 A[NA][NA], W[NA][NA], G[NA][NA];
        double S[NA];
#pragma unroll
        for (int r = 0; r < NA; ++r)
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                const cx<T> v = Hg[(b * NA + r) * NA + c];
                A[r][c] = mk<double>((double)v.x, (double)v.y);
            }
        svd_filters_dev<NA>(A, W, G, S);
#pragma unroll
        for (int r = 0; r < NA; ++r) {
            if (Sg) Sg[b * NA + r] = S[r];
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                Wg[(b * NA + r) * NA + c] = mk<T>((T)W[r][c].x, (T)W[r][c].y);
                Gg[(b * NA + r) * NA + c] = mk<T>((T)G[r][c].x, (T)G[r][c].y);
            }
        }
    }
}

// GMDMimo (mimo.py:952-1067): see mimo_svd.hpp
template <typename T, int NA>
__global__ __launch_bounds__(64) void k_gmd_filters(const cx<T>* __restrict__ Hg, double nv, cx<T>* __restrict__ Wg,
                                                    cx<T>* __restrict__ Gg, double* __restrict__ Rg,
                                                    uint32_t* __restrict__ skipped, size_t batch) {
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < batch; b += (size_t)gridDim.x * blockDim.x) {
        double2 H[NA][NA], W[NA][NA], G[NA][NA];
        double R[NA][NA];
#pragma unroll
        for (int r = 0; r < NA; ++r)
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                const cx<T> v = Hg[(b * NA + r) * NA + c];
                H[r][c] = mk<double>((double)v.x, (double)v.y);
            }
        const bool ok = gmd_filters_dev<NA>(H, nv, W, G, R);
#pragma unroll
        for (int r = 0; r < NA; ++r)
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                Wg[(b * NA + r) * NA + c] = mk<T>((T)W[r][c].x, (T)W[r][c].y);
                Gg[(b * NA + r) * NA + c] = mk<T>((T)G[r][c].x, (T)G[r][c].y);
                if (Rg) Rg[(b * NA + r) * NA + c] = R[r][c];
            }
        if (skipped) skipped[b] = ok ? 0u : 1u;
    }
}

template <typename T, int NT, int NR>
int launch_filter(mcle_ctx* ctx, const void* d_H, double nv, void* d_G, uint32_t* d_skipped, size_t batch) {
    if constexpr ((NT * NR * sizeof(cx<T>)) % 16 == 0) {
        if ((((uintptr_t)d_H | (uintptr_t)d_G) & 15) == 0 && batch >= 64) {
            hipLaunchKernelGGL((k_blast_filter_staged<T, NT, NR>), dim3(grid_for(ctx, batch, 64, 16)), dim3(64), 0,
                               ctx->stream, (const cx<T>*)d_H, nv, (cx<T>*)d_G, d_skipped, batch);
            MCLE_LAUNCH_CHECK();
            return MCLE_OK;
        }
    }
    hipLaunchKernelGGL((k_blast_filter<T, NT, NR>), dim3(grid_for(ctx, batch, 64, 16)), dim3(64), 0, ctx->stream,
                       (const cx<T>*)d_H, nv, (cx<T>*)d_G, d_skipped, batch);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

template <typename T>
int dispatch_filter(mcle_ctx* ctx, int nr, int nt, const void* d_H, double nv, void* d_G, uint32_t* d_skipped,
                    size_t batch) {
#define MCLE_F(NR_, NT_) \
    if (nr == NR_ && nt == NT_) return launch_filter<T, NT_, NR_>(ctx, d_H, nv, d_G, d_skipped, batch);
    MCLE_F(1, 1) MCLE_F(2, 1) MCLE_F(2, 2) MCLE_F(3, 1) MCLE_F(3, 2) MCLE_F(3, 3) MCLE_F(4, 1) MCLE_F(4, 2)
    MCLE_F(4, 3) MCLE_F(4, 4)
#undef MCLE_F
    set_error("unsupported antenna configuration nr=%d nt=%d (need 1 <= nt <= nr <= 4)", nr, nt);
    return MCLE_E_INVAL;
}

int check_mimo(const mcle_ctx* ctx, int dtype, int nr, int nt, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(nt >= 1 && nr >= 1 && nt <= 64 && nr <= 64, "bad antenna counts nr=%d nt=%d", nr, nt);
    MCLE_REQUIRE(batch <= 65535, "batch too large (%zu > 65535)", batch);
    return MCLE_OK;
}

}  // namespace mcle

namespace mcle {
// calc_post_processing_linear_SINRs (reference mimo/mimo.py:62-118): per channel of the batch, E = G_H H W (ns x ns);
// stream i: S = |E_ii|^2, I = |sum_{j != i} E_ij|^2 (the reference takes the modulus of the SUM of the off-diagonal
// row entries), N = noise_var * ||row i of G_H||^2; SINR = S / (I + N).  One thread per (channel, stream), f64.
__global__ __launch_bounds__(256) void k_post_sinr(const double2* __restrict__ H, const double2* __restrict__ W,
                                                   const double2* __restrict__ G, double noise_var, int nr, int nt,
                                                   int ns, double* __restrict__ sinr, size_t batch) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * (size_t)ns) return;
    const size_t b = idx / ns;
    const int i = (int)(idx - b * ns);
    const double2* Hb = H + b * nr * nt;
    const double2* Wb = W + b * nt * ns;
    const double2* Gi = G + (b * ns + i) * nr;
    double2 diag = mk<double>(0, 0), row = mk<double>(0, 0);
    double gn = 0.0;
    for (int r = 0; r < nr; ++r) gn += Gi[r].x * Gi[r].x + Gi[r].y * Gi[r].y;
    for (int j = 0; j < ns; ++j) {
        double2 e = mk<double>(0, 0);
        for (int a = 0; a < nt; ++a) {
            double2 hw = mk<double>(0, 0);            // (G_H H)_{i a}
            for (int r = 0; r < nr; ++r) hw = cadd(hw, cmul(Gi[r], Hb[r * nt + a]));
            e = cadd(e, cmul(hw, Wb[a * ns + j]));
        }
        row = cadd(row, e);
        if (j == i) diag = e;
    }
    const double2 off = csub(row, diag);
    const double S = diag.x * diag.x + diag.y * diag.y, I = off.x * off.x + off.y * off.y;
    sinr[idx] = S / (I + noise_var * gn);
}
}  // namespace mcle

using namespace mcle;

extern "C" {

int mcle_blast_encode(mcle_ctx* ctx, int dtype, const void* d_x, int nt, size_t n, void* d_X, size_t batch) {
    int rc = check_mimo(ctx, dtype, 1, nt, batch);
    if (rc) return rc;
    // mimo.py:633-637
    MCLE_REQUIRE(n % (size_t)nt == 0,
                 "Input array number of elements must be a multiple of the number of transmit antennas.");
    if (n == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    dim3 grid((unsigned)grid_for(ctx, n, kMimoBlock, 4), (unsigned)batch);
    const double s = 1.0 / std::sqrt((double)nt);
    const size_t ns_ = n / nt, pair_bytes = dtype == MCLE_F32 ? 16 : 32;
    if ((nt == 2 || nt == 4) && ns_ % 2 == 0 && ((uintptr_t)d_x % pair_bytes) == 0 && ((uintptr_t)d_X % pair_bytes) == 0) {
        dim3 gridp((unsigned)grid_for(ctx, ns_ / 2, kMimoBlock, 4), (unsigned)batch);
        if (dtype == MCLE_F32 && nt == 2)
            hipLaunchKernelGGL((k_blast_encode_pairs<float, 2>), gridp, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_x,
                               ns_, (float)s, (float2*)d_X);
        else if (dtype == MCLE_F32)
            hipLaunchKernelGGL((k_blast_encode_pairs<float, 4>), gridp, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_x,
                               ns_, (float)s, (float2*)d_X);
        else if (nt == 2)
            hipLaunchKernelGGL((k_blast_encode_pairs<double, 2>), gridp, dim3(kMimoBlock), 0, ctx->stream,
                               (const double2*)d_x, ns_, s, (double2*)d_X);
        else
            hipLaunchKernelGGL((k_blast_encode_pairs<double, 4>), gridp, dim3(kMimoBlock), 0, ctx->stream,
                               (const double2*)d_x, ns_, s, (double2*)d_X);
        MCLE_LAUNCH_CHECK();
        return MCLE_OK;
    }
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_blast_encode<float>, grid, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_x, nt,
                           n / nt, (float)s, (float2*)d_X);
    else
        hipLaunchKernelGGL(k_blast_encode<double>, grid, dim3(kMimoBlock), 0, ctx->stream, (const double2*)d_x, nt,
                           n / nt, s, (double2*)d_X);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_blast_filter(mcle_ctx* ctx, int dtype, const void* d_H, int nr, int nt, double noise_var, void* d_G,
                      uint32_t* d_skipped, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    // mimo.py:553
    MCLE_REQUIRE(noise_var >= 0.0, "Noise variance must be a non-negative value.");
    MCLE_REQUIRE(nt <= nr, "Blast needs at least as many receive as transmit antennas (nr=%d nt=%d)", nr, nt);
    if (batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    if (dtype == MCLE_F32) return dispatch_filter<float>(ctx, nr, nt, d_H, noise_var, d_G, d_skipped, batch);
    return dispatch_filter<double>(ctx, nr, nt, d_H, noise_var, d_G, d_skipped, batch);
}

int mcle_blast_decode(mcle_ctx* ctx, int dtype, const void* d_G, const void* d_Y, int nr, int nt, size_t ns,
                      void* d_est, size_t batch) {
    int rc = check_mimo(ctx, dtype, nr, nt, batch);
    if (rc) return rc;
    if (ns == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    dim3 grid((unsigned)grid_for(ctx, ns, kMimoBlock, 4), (unsigned)batch);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_blast_decode<float>, grid, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_G,
                           (const float2*)d_Y, nr, nt, ns, (float2*)d_est);
    else
        hipLaunchKernelGGL(k_blast_decode<double>, grid, dim3(kMimoBlock), 0, ctx->stream, (const double2*)d_G,
                           (const double2*)d_Y, nr, nt, ns, (double2*)d_est);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_mimo_channel(mcle_ctx* ctx, int dtype, const void* d_H, const void* d_X, const void* d_noise,
                      double noise_var, int nr, int nt, size_t ns, void* d_Y, size_t batch) {
    int rc = check_mimo(ctx, dtype, nr, nt, batch);
    if (rc) return rc;
    MCLE_REQUIRE(noise_var >= 0.0, "noise variance must be non-negative");
    if (ns == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    const int vec = nr == 4 && nt == 4 && ns % 2 == 0 &&
                    ((((uintptr_t)d_X) | ((uintptr_t)d_Y) | ((uintptr_t)d_noise)) & 15u) == 0;
    dim3 grid((unsigned)grid_for(ctx, vec ? ns / 2 : ns, kMimoBlock, 4), (unsigned)batch);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_mimo_channel<float>, grid, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_H,
                           (const float2*)d_X, (const float2*)d_noise, (float)std::sqrt(noise_var), nr, nt, ns,
                           (float2*)d_Y, vec);
    else
        hipLaunchKernelGGL(k_mimo_channel<double>, grid, dim3(kMimoBlock), 0, ctx->stream, (const double2*)d_H,
                           (const double2*)d_X, (const double2*)d_noise, std::sqrt(noise_var), nr, nt, ns,
                           (double2*)d_Y, 0);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_mimo_channel_philox(mcle_ctx* ctx, int dtype, const void* d_H, const void* d_X, uint64_t seed, uint64_t first,
                             double noise_var, int nr, int nt, size_t ns, void* d_Y, size_t batch) {
    int rc = check_mimo(ctx, dtype, nr, nt, batch);
    if (rc) return rc;
    MCLE_REQUIRE(noise_var >= 0.0, "noise variance must be non-negative");
    MCLE_REQUIRE(batch <= 65535, "at most 65535 realizations per call");
    MCLE_REQUIRE((uint64_t)nr * ns < (1ull << 33), "more than 2^33 noise samples per realization");
    if (ns == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    const int vec = dtype == MCLE_F32 && nr == 4 && nt == 4 && ns % 2 == 0 &&
                    ((((uintptr_t)d_X) | ((uintptr_t)d_Y)) & 15u) == 0;
    // (complex128 4 x 4: one column per lane; everything else two)
    dim3 grid((unsigned)grid_for(ctx, (dtype == MCLE_F64 && vec) ? ns : (ns + 1) / 2, kMimoBlock, 4), (unsigned)batch);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_mimo_channel_philox<float>, grid, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_H,
                           (const float2*)d_X, seed, first, (float)std::sqrt(noise_var), nr, nt, ns, (float2*)d_Y, vec);
    else
        hipLaunchKernelGGL(k_mimo_channel_philox<double>, grid, dim3(kMimoBlock), 0, ctx->stream, (const double2*)d_H,
                           (const double2*)d_X, seed, first, std::sqrt(noise_var), nr, nt, ns, (double2*)d_Y, vec);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_alamouti_encode(mcle_ctx* ctx, int dtype, const void* d_x, size_t n, void* d_X, size_t batch) {
    int rc = check_mimo(ctx, dtype, 1, 2, batch);
    if (rc) return rc;
    MCLE_REQUIRE(n % 2 == 0, "Alamouti needs an even number of symbols");
    if (n == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    dim3 grid((unsigned)grid_for(ctx, n / 2, kMimoBlock, 4), (unsigned)batch);
    const double s = 1.0 / std::sqrt(2.0);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_alamouti_encode<float>, grid, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_x, n,
                           (float)s, (float2*)d_X);
    else
        hipLaunchKernelGGL(k_alamouti_encode<double>, grid, dim3(kMimoBlock), 0, ctx->stream, (const double2*)d_x, n, s,
                           (double2*)d_X);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_alamouti_decode(mcle_ctx* ctx, int dtype, const void* d_H, const void* d_Y, int nr, size_t n, void* d_out,
                         size_t batch) {
    int rc = check_mimo(ctx, dtype, nr, 2, batch);
    if (rc) return rc;
    MCLE_REQUIRE(n % 2 == 0, "Alamouti needs an even number of symbols");
    if (n == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    dim3 grid((unsigned)grid_for(ctx, n / 2, kMimoBlock, 4), (unsigned)batch);
    const size_t pair = dtype == MCLE_F32 ? 16 : 32;
    const bool vec = ((uintptr_t)d_Y % pair) == 0 && ((uintptr_t)d_out % pair) == 0;
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_alamouti_decode<float>, grid, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_H,
                           (const float2*)d_Y, nr, n, (float)std::sqrt(2.0), (float2*)d_out, vec);
    else
        hipLaunchKernelGGL(k_alamouti_decode<double>, grid, dim3(kMimoBlock), 0, ctx->stream, (const double2*)d_H,
                           (const double2*)d_Y, nr, n, std::sqrt(2.0), (double2*)d_out, vec);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_mrt_encode(mcle_ctx* ctx, int dtype, const void* d_h, const void* d_x, int nt, size_t n, void* d_X,
                    size_t batch) {
    int rc = check_mimo(ctx, dtype, 1, nt, batch);
    if (rc) return rc;
    if (n == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    dim3 grid((unsigned)grid_for(ctx, n, kMimoBlock, 4), (unsigned)batch);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_mrt_encode<float>, grid, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_h,
                           (const float2*)d_x, nt, n, (float2*)d_X);
    else
        hipLaunchKernelGGL(k_mrt_encode<double>, grid, dim3(kMimoBlock), 0, ctx->stream, (const double2*)d_h,
                           (const double2*)d_x, nt, n, (double2*)d_X);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_mrt_decode(mcle_ctx* ctx, int dtype, const void* d_h, const void* d_y, int nt, size_t n, void* d_out,
                    size_t batch) {
    int rc = check_mimo(ctx, dtype, 1, nt, batch);
    if (rc) return rc;
    if (n == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    dim3 grid((unsigned)grid_for(ctx, n, kMimoBlock, 4), (unsigned)batch);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_mrt_decode<float>, grid, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_h,
                           (const float2*)d_y, nt, n, (float2*)d_out);
    else
        hipLaunchKernelGGL(k_mrt_decode<double>, grid, dim3(kMimoBlock), 0, ctx->stream, (const double2*)d_h,
                           (const double2*)d_y, nt, n, (double2*)d_out);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_svd_filters(mcle_ctx* ctx, int dtype, const void* d_H, int n_ant, void* d_W, void* d_G, double* d_S,
                     size_t batch) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(n_ant >= 2 && n_ant <= 4, "SVD filters support square channels with 2 <= N <= 4 (got %d)", n_ant);
    if (batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    const dim3 grid(grid_for(ctx, batch, 64, 16));
#define MCLE_SVD(T_, N_)                                                                                       \
    hipLaunchKernelGGL((k_svd_filters<T_, N_>), grid, dim3(64), 0, ctx->stream, (const cx<T_>*)d_H, (cx<T_>*)d_W, \
                       (cx<T_>*)d_G, d_S, batch)
    if (dtype == MCLE_F32) {
        if (n_ant == 2) MCLE_SVD(float, 2);
        else if (n_ant == 3) MCLE_SVD(float, 3);
        else MCLE_SVD(float, 4);
    } else {
        if (n_ant == 2) MCLE_SVD(double, 2);
        else if (n_ant == 3) MCLE_SVD(double, 3);
        else MCLE_SVD(double, 4);
    }
#undef MCLE_SVD
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_blast_decode_per_subcarrier(mcle_ctx* ctx, int dtype, const void* d_G, const void* d_Y, int nr, int nt,
                                     size_t ns, void* d_est, size_t batch) {
    int rc = check_mimo(ctx, dtype, nr, nt, 1);
    if (rc) return rc;
    MCLE_REQUIRE(batch <= 65535, "batch too large (%zu > 65535)", batch);
    if (ns == 0 || batch == 0) return MCLE_OK;
    if ((rc = ctx->bind())) return rc;
    const dim3 grid(grid_for(ctx, ns * (size_t)nt, kMimoBlock, 4), (unsigned)batch);
    if (dtype == MCLE_F32)
        hipLaunchKernelGGL(k_blast_decode_persc<float>, grid, dim3(kMimoBlock), 0, ctx->stream, (const float2*)d_G,
                           (const float2*)d_Y, nr, nt, ns, (float2*)d_est);
    else
        hipLaunchKernelGGL(k_blast_decode_persc<double>, grid, dim3(kMimoBlock), 0, ctx->stream, (const double2*)d_G,
                           (const double2*)d_Y, nr, nt, ns, (double2*)d_est);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_gmd_filters(mcle_ctx* ctx, int dtype, const void* d_H, int n_ant, double noise_var, void* d_W, void* d_G,
                     double* d_R, uint32_t* d_skipped, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr, "null context");
    MCLE_REQUIRE(dtype == MCLE_F32 || dtype == MCLE_F64, "dtype must be MCLE_F32 or MCLE_F64");
    MCLE_REQUIRE(n_ant >= 2 && n_ant <= 4, "GMD filters support square channels with 2 <= N <= 4 (got %d)", n_ant);
    MCLE_REQUIRE(noise_var >= 0.0, "Noise variance must be a non-negative value.");
    if (batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    const dim3 grid(grid_for(ctx, batch, 64, 16));
#define MCLE_GMD(T_, N_)                                                                                          \
    hipLaunchKernelGGL((k_gmd_filters<T_, N_>), grid, dim3(64), 0, ctx->stream, (const cx<T_>*)d_H, noise_var,    \
                       (cx<T_>*)d_W, (cx<T_>*)d_G, d_R, d_skipped, batch)
    if (dtype == MCLE_F32) {
        if (n_ant == 2) MCLE_GMD(float, 2);
        else if (n_ant == 3) MCLE_GMD(float, 3);
        else MCLE_GMD(float, 4);
    } else {
        if (n_ant == 2) MCLE_GMD(double, 2);
        else if (n_ant == 3) MCLE_GMD(double, 3);
        else MCLE_GMD(double, 4);
    }
#undef MCLE_GMD
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

int mcle_post_processing_sinrs(mcle_ctx* ctx, const void* d_H, const void* d_W, const void* d_G, double noise_var,
                               int nr, int nt, int ns, double* d_sinr, size_t batch) {
    MCLE_REQUIRE(ctx != nullptr && d_H != nullptr && d_W != nullptr && d_G != nullptr && d_sinr != nullptr,
                 "null argument");
    MCLE_REQUIRE(nr >= 1 && nt >= 1 && ns >= 1 && nr <= 64 && nt <= 64 && ns <= 64, "dimensions must be in [1, 64]");
    MCLE_REQUIRE(noise_var >= 0.0, "Noise variance must be a non-negative value.");
    if (batch == 0) return MCLE_OK;
    int rc = ctx->bind();
    if (rc) return rc;
    const size_t items = batch * (size_t)ns;
    hipLaunchKernelGGL(k_post_sinr, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double2*)d_H, (const double2*)d_W, (const double2*)d_G, noise_var, nr, nt, ns, d_sinr,
                       batch);
    MCLE_LAUNCH_CHECK();
    return MCLE_OK;
}

}  // extern "C"
