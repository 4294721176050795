// modem.hpp -- device-side constellation helpers shared by operator kernels and pipelines.
// Stands in for Modulator.modulate / demodulate (reference modulators/fundamental.py:175-248)
// and util/misc.py:449-566 (count_bits / count_bit_errors).
#pragma once
#include "common.hpp"

namespace mcle {

// Pruned exhaustive search (f32 instantiations).  The plane is cut into G x G cells (border cells extend to
// infinity); each cell lists, in ascending index order, every constellation point that can be the nearest one
// -- within a rounding margin -- somewhere in the cell (built on the host in f64: capi.hip build_demod_grid).
// Searching the list with the same metric and the same strict '<' gives the decision of the full sweep over
// all M points, tie rule included, at a few candidates per symbol.  Cell word: byte 0 = count (0xFF: sweep
// everything), bytes 1..7 = candidates.
constexpr int kMaxGridCells = 32 * 32;
struct PskCert;
struct DemodGrid {
    const unsigned long long* cells;   // device [G*G]
    int G;                             // 0: no grid (M > 256 or f64)
    float x0, y0, inv_h;
    // 8- / 16-PSK: the sector certificate's constants (device memory, one block per context, written by mcle_set_constellation);
    // null otherwise.  EVERY grid search below tries that certificate first (demod_psk_cert) -- it is part of the table search, not
    // of the first-line certificates of demod_cert_any: inlined there it cost the QAM kernels registers (the complex64 headline
    // kernel 25 -> 33 spilled registers, -5 %: scripts/experiments/r05_calls.txt [call 20]), here a QAM launch never reaches it.
    const PskCert* psk;
};
// label of sector k (64 / M bits each, sector 0 in the low bits), e^{-j phi0}, cos / sin of the M / 8 sector boundaries inside the
// first octant, the magnitude window of the certificate -- in both arithmetics
struct PskCert {
    unsigned lut[2];
    float rot_f[2], cb_f[2], sb_f[2], lo_f, hi_f;
    double rot_d[2], cb_d[2], sb_d[2], lo_d, hi_d;
};
template <typename T> struct PskView;
template <> struct PskView<float> {
    const PskCert* p;
    __device__ __forceinline__ float rot(int i) const { return p->rot_f[i]; }
    __device__ __forceinline__ float cb(int i) const { return p->cb_f[i]; }
    __device__ __forceinline__ float sb(int i) const { return p->sb_f[i]; }
    __device__ __forceinline__ float lo() const { return p->lo_f; }
    __device__ __forceinline__ float hi() const { return p->hi_f; }
};
template <> struct PskView<double> {
    const PskCert* p;
    __device__ __forceinline__ double rot(int i) const { return p->rot_d[i]; }
    __device__ __forceinline__ double cb(int i) const { return p->cb_d[i]; }
    __device__ __forceinline__ double sb(int i) const { return p->sb_d[i]; }
    __device__ __forceinline__ double lo() const { return p->lo_d; }
    __device__ __forceinline__ double hi() const { return p->hi_d; }
};

// Min-distance decision of an M-PSK (M = 8, 16: M points of one radius at angles 2 pi k / M + phi0) WITHOUT touching the
// table: the regions are the M sectors.  u = r e^{-j phi0}; fold into the first octant (hi = max(|re|, |im|), lo = the other); the
// M / 8 sector boundaries inside the octant sit at theta_j = (2 j + 1) pi / M, and lo cos(theta_j) - hi sin(theta_j) =
// |u| sin(theta - theta_j) says on which side of boundary j the point lies -- no arctangent, no division.  p = boundaries passed =
// point index inside the octant; un-fold: swap -> M / 4 - p, re < 0 -> M / 2 - k, im < 0 -> -k (mod M); label = lut[k].  The folds
// are consistent ON their own borders (the 45-degree line and the axes are point directions, both sides give the same k), so only
// the sector boundaries need a margin: `sure` iff every |lo cos - hi sin| >= eps hi, i.e. an angular distance >= eps / sqrt 2 from
// every boundary, and lo_bound <= hi <= hi_bound.  Two neighbouring candidates then differ by >= 2 |r| rho sin(pi / M) eps in
// squared distance (rho = the radius) against a rounding of ~ (|r|^2 + rho^2) ulp of either metric: eps = 2^-28 inside
// [2^-8, 2^8] rho in complex128, 2^-12 inside [1/8, 8] rho in complex64 -- the sweep, first-minimum rule included, returns this
// very label.  Elsewhere (probability ~ M eps / 4 per symbol, or a deep-fade equaliser output) the caller searches the table.
// (Written without selects: a v_cndmask on a freshly compared mask costs several issue slots on gfx950 (profiles/r04/f32_rates.txt),
//  and the first form of this function -- five selects and a select tree for the label -- was SLOWER than the candidate-grid search
//  it replaces (scripts/experiments/r05_psk_rates.py).  The folds are sign masks: m = (sign bit of a difference) >> 31 is 0 or -1,
//  and "M / 4 - p if swapped" is (p ^ m) + ((M / 4 + 1) & m); the label comes out of a 64-bit word of M fields by one shift.)
__device__ __forceinline__ int sign_mask(float v) { return __float_as_int(v) >> 31; }
__device__ __forceinline__ int sign_mask(double v) { return __double2hiint(v) >> 31; }
template <typename T>
__device__ __forceinline__ int demod_psk_cert(cx<T> r, const PskCert* psk, int M, bool& sure) {
    const PskView<T> c{psk};                                   // wave-uniform loads (scalar cache)
    const T ux = r.x * c.rot(0) - r.y * c.rot(1), uy = r.x * c.rot(1) + r.y * c.rot(0);
    const T ax = fabs(ux), ay = fabs(uy);
    const T hi = fmax(ax, ay), lo = fmin(ax, ay);
    constexpr T eps = sizeof(T) == 8 ? (T)0x1p-28 : (T)0x1p-12;
    const T tol = eps * hi;
    const int nb = M >> 3;                                     // boundaries inside an octant: 1 (8-PSK), 2 (16-PSK)
    int p = nb;
    bool ok = hi >= c.lo() && hi <= c.hi();                       // NaN: not sure
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (j < nb) {
            const T d = lo * c.cb(j) - hi * c.sb(j);              // |u| sin(theta - theta_j)
            p += sign_mask(d);                                    // nb - (boundaries NOT passed)
            ok = ok && fabs(d) >= tol;
        }
    const int ms = sign_mask(ax - ay), mx = sign_mask(ux), my = sign_mask(uy);
    int k = (p ^ ms) + (((M >> 2) + 1) & ms);                  // swapped: M / 4 - p
    k = (k ^ mx) + (((M >> 1) + 1) & mx);                      // re < 0: M / 2 - k
    k = (k ^ my) - my;                                            // im < 0: -k
    k &= M - 1;
    sure = ok;
    const unsigned long long lut = ((unsigned long long)psk->lut[1] << 32) | psk->lut[0];   // M fields of 64 / M bits
    const int fw = M == 8 ? 8 : 4;
    return (int)((lut >> (k * fw)) & (M == 8 ? 0xFFull : 0xFull));
}


template <typename T> struct ModemParams {
    DemodGrid grid;
    const cx<T>* g_table;  // global constellation table [M]
    int M;
    int bits;        // log2(M)
    int method;      // MCLE_DEMOD_*
    T qam_scale;     // sqrt(2(M-1)/3)
    int qam_L;       // sqrt(M)
    int half_bits;   // bits/2
    int cert;        // 1: min-distance decisions of a square Gray QAM through the margin certificate (demod_qam_cert);
                     // 2: of a four-point one-per-quadrant constellation (QPSK) through the quadrant certificate (demod_quad_cert)
    unsigned quad_lut;   // cert == 2: label of quadrant (re < 0) | (im < 0) << 1, a byte each
    T quad_lo, quad_hi;  // cert == 2: the certificate holds for lo <= |re|, |im| <= hi
};
// exhaustive minimum distance over a table held in LDS; strict '<' keeps the FIRST minimum,
// which is numpy.argmin's tie rule (fundamental.py:245).  The reference compares |c - r|; we
// compare |c - r|^2 (same ordering away from rounding-level ties).
template <typename T>
__device__ __forceinline__ int demod_mindist(const cx<T>* __restrict__ s_table, int M, cx<T> r) {
    T best = (r.x - s_table[0].x) * (r.x - s_table[0].x) + (r.y - s_table[0].y) * (r.y - s_table[0].y);
    int idx = 0;
#pragma unroll 8
    for (int m = 1; m < M; ++m) {
        const cx<T> c = s_table[m];
        const T dx = r.x - c.x, dy = r.y - c.y;
        const T d = dx * dx + dy * dy;
        if (d < best) {
            best = d;
            idx = m;
        }
    }
    return idx;
}

// K symbols against the same table in one sweep (one LDS read per candidate serves all K).  For f32
// the ordering metric is |c|^2/2 - Re(r conj(c)) (two FMAs per candidate; argmin-equivalent to
// |c - r|^2 since |r|^2 is common), read from a float4 table {c.re, c.im, |c|^2/2, 0}; f64 keeps the
// literal |c - r|^2 of the parity path.  First minimum wins, as in demod_mindist.
template <int K>
__device__ __forceinline__ void demod_mindist_multi(const float4* __restrict__ s_tab4, int M, const float2 (&r)[K],
                                                    int (&idx)[K]) {
    float best[K];
    {
        const float4 c = s_tab4[0];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            best[k] = fmaf(-r[k].y, c.y, fmaf(-r[k].x, c.x, c.z));
            idx[k] = 0;
        }
    }
#pragma unroll 4
    for (int m = 1; m < M; ++m) {
        const float4 c = s_tab4[m];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float d = fmaf(-r[k].y, c.y, fmaf(-r[k].x, c.x, c.z));
            if (d < best[k]) {
                best[k] = d;
                idx[k] = m;
            }
        }
    }
}
template <int K>
__device__ __forceinline__ void demod_mindist_multi(const double2* __restrict__ s_table, int M, const double2 (&r)[K],
                                                    int (&idx)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) idx[k] = demod_mindist<double>(s_table, M, r[k]);
}

__device__ __forceinline__ unsigned long long grid_cell(const unsigned long long* __restrict__ s_grid, const DemodGrid& g,
                                                        float x, float y) {
    int ix = (int)floorf((x - g.x0) * g.inv_h), iy = (int)floorf((y - g.y0) * g.inv_h);
    ix = ix < 0 ? 0 : (ix >= g.G ? g.G - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= g.G ? g.G - 1 : iy);
    return s_grid[iy * g.G + ix];
}
// literal |c - r|^2 metric on the plain table (operator kernels); same decisions as demod_mindist<float>
__device__ __forceinline__ int demod_grid_search(const float2* __restrict__ s_table, const unsigned long long* __restrict__ s_grid,
                                                 const DemodGrid& g, int M, float2 r) {
    unsigned long long w = grid_cell(s_grid, g, r.x, r.y);
    const unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
    const int n = (int)(lo & 0xFFu);
    if (n == 0xFF) return demod_mindist<float>(s_table, M, r);
    // first four candidates straight-line, two at a time (see the complex128 form below)
    const int idx0 = (int)((lo >> 8) & 0xFFu);
    int idx = idx0;
    const int m1 = n > 1 ? (int)((lo >> 16) & 0xFFu) : idx0;
    const float2 c0 = s_table[idx0], c1 = s_table[m1];
    float best = (r.x - c0.x) * (r.x - c0.x) + (r.y - c0.y) * (r.y - c0.y);
    const float d1 = (r.x - c1.x) * (r.x - c1.x) + (r.y - c1.y) * (r.y - c1.y);
    if (d1 < best) {
        best = d1;
        idx = m1;
    }
    if (n > 2) {
        const int m2 = (int)(lo >> 24), m3 = n > 3 ? (int)(hi & 0xFFu) : idx0;
        const float2 c2 = s_table[m2], c3 = s_table[m3];
        const float d2 = (r.x - c2.x) * (r.x - c2.x) + (r.y - c2.y) * (r.y - c2.y);
        const float d3 = (r.x - c3.x) * (r.x - c3.x) + (r.y - c3.y) * (r.y - c3.y);
        if (d2 < best) {
            best = d2;
            idx = m2;
        }
        if (d3 < best) {
            best = d3;
            idx = m3;
        }
    }
    w >>= 32;
    for (int j = 4; j < n; ++j) {
        w >>= 8;
        const int m = (int)(w & 0xFFull);
        const float2 c = s_table[m];
        const float dx = r.x - c.x, dy = r.y - c.y;
        const float d = dx * dx + dy * dy;
        if (d < best) {
            best = d;
            idx = m;
        }
    }
    return idx;
}
// complex128 operator kernels: the cell is looked up from the float-rounded point (the lists carry a margin far above
// f32 rounding, so a point a rounding step across a cell edge still finds its nearest neighbour and every near-tie in
// the list), the metric is the literal f64 |c - r|^2 of demod_mindist<double>: same decisions, first minimum included
__device__ __forceinline__ int demod_grid_search(const double2* __restrict__ s_table, const unsigned long long* __restrict__ s_grid,
                                                 const DemodGrid& g, int M, double2 r) {
    unsigned long long w = grid_cell(s_grid, g, (float)r.x, (float)r.y);
    const unsigned lo = (unsigned)w, hi = (unsigned)(w >> 32);
    const int n = (int)(lo & 0xFFu);
    if (n == 0xFF) return demod_mindist<double>(s_table, M, r);
    // the first four candidates straight-line, two at a time: their table entries are fetched together (one or two LDS round
    // trips instead of up to three dependent ones; a wave walks its longest list anyway), short lists padded with the first
    // candidate, which never beats itself under the strict comparison -- same order, same first minimum as the loop
    const int idx0 = (int)((lo >> 8) & 0xFFu);
    int idx = idx0;
    const int m1 = n > 1 ? (int)((lo >> 16) & 0xFFu) : idx0;
    const double2 c0 = s_table[idx0], c1 = s_table[m1];
    double best = (r.x - c0.x) * (r.x - c0.x) + (r.y - c0.y) * (r.y - c0.y);
    const double d1 = (r.x - c1.x) * (r.x - c1.x) + (r.y - c1.y) * (r.y - c1.y);
    if (d1 < best) {
        best = d1;
        idx = m1;
    }
    if (n > 2) {                           // small constellations rarely get here
        const int m2 = (int)(lo >> 24), m3 = n > 3 ? (int)(hi & 0xFFu) : idx0;
        const double2 c2 = s_table[m2], c3 = s_table[m3];
        const double d2 = (r.x - c2.x) * (r.x - c2.x) + (r.y - c2.y) * (r.y - c2.y);
        const double d3 = (r.x - c3.x) * (r.x - c3.x) + (r.y - c3.y) * (r.y - c3.y);
        if (d2 < best) {
            best = d2;
            idx = m2;
        }
        if (d3 < best) {
            best = d3;
            idx = m3;
        }
    }
    w >>= 32;
    for (int j = 4; j < n; ++j) {          // lists of five to seven: rare (cells at a corner of four regions of a dense set)
        w >>= 8;
        const int m = (int)(w & 0xFFull);
        const double2 c = s_table[m];
        const double dx = r.x - c.x, dy = r.y - c.y;
        const double d = dx * dx + dy * dy;
        if (d < best) {
            best = d;
            idx = m;
        }
    }
    return idx;
}
// two-FMA metric on the {re, im, |c|^2/2} table (fused pipelines); same decisions as demod_mindist_multi<float>
__device__ __forceinline__ int demod_grid4_search(const float4* __restrict__ s_tab4, const unsigned long long* __restrict__ s_grid,
                                                  const DemodGrid& g, int M, float2 r) {
    unsigned long long w = grid_cell(s_grid, g, r.x, r.y);
    int n = (int)(w & 0xFFull);
    if (n == 0xFF) {
        const float2 rr[1] = {r};
        int out[1];
        demod_mindist_multi<1>(s_tab4, M, rr, out);
        return out[0];
    }
    w >>= 8;
    int idx = (int)(w & 0xFFull);
    float best;
    {
        const float4 c = s_tab4[idx];
        best = fmaf(-r.y, c.y, fmaf(-r.x, c.x, c.z));
    }
    for (int j = 1; j < n; ++j) {
        w >>= 8;
        const int m = (int)(w & 0xFFull);
        const float4 c = s_tab4[m];
        const float d = fmaf(-r.y, c.y, fmaf(-r.x, c.x, c.z));
        if (d < best) {
            best = d;
            idx = m;
        }
    }
    return idx;
}
// K symbols through the candidate grid in lockstep: the K cell words are fetched together, then up to four candidates
// per symbol are evaluated as K independent LDS round trips per step (a symbol with fewer candidates re-evaluates its
// current best, which the strict '<' ignores).  Cells with more than four candidates, or the 0xFF "sweep
// everything" marker, take demod_grid4's loop.  Same decisions as demod_grid4 / demod_mindist_multi.
template <int K>
__device__ __forceinline__ void demod_grid4_multi_search(const float4* __restrict__ s_tab4,
                                                  const unsigned long long* __restrict__ s_grid, const DemodGrid& g, int M,
                                                  const float2 (&r)[K], int (&idx)[K]) {
    unsigned long long w[K];
    int n[K];
    float best[K];
    bool slow = false;
#pragma unroll
    for (int k = 0; k < K; ++k) w[k] = grid_cell(s_grid, g, r[k].x, r[k].y);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        n[k] = (int)(w[k] & 0xFFull);
        slow = slow || n[k] > 4;
        idx[k] = (int)((w[k] >> 8) & 0xFFull);
        const float4 c = s_tab4[idx[k]];
        best[k] = fmaf(-r[k].y, c.y, fmaf(-r[k].x, c.x, c.z));
    }
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        int m[K];
        float4 c[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            m[k] = j < n[k] ? (int)((w[k] >> (8 * (j + 1))) & 0xFFull) : idx[k];
            c[k] = s_tab4[m[k]];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float d = fmaf(-r[k].y, c[k].y, fmaf(-r[k].x, c[k].x, c[k].z));
            if (d < best[k]) {
                best[k] = d;
                idx[k] = m[k];
            }
        }
    }
    if (slow) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (n[k] > 4) idx[k] = demod_grid4_search(s_tab4, s_grid, g, M, r[k]);
    }
}

// The complex128 form of the lockstep search: the literal |c - r|^2 metric of demod_mindist<double> on the plain table, K
// symbols at a time (the K cell words, then up to four candidates per symbol as K independent LDS round trips per step).
// Same decisions as demod_grid(double) / demod_mindist<double>, first minimum included.
template <int K>
__device__ __forceinline__ void demod_grid_multi_search(const double2* __restrict__ s_table,
                                                 const unsigned long long* __restrict__ s_grid, const DemodGrid& g, int M,
                                                 const double2 (&r)[K], int (&idx)[K]) {
    unsigned long long w[K];
    int n[K];
    double best[K];
    bool slow = false;
#pragma unroll
    for (int k = 0; k < K; ++k) w[k] = grid_cell(s_grid, g, (float)r[k].x, (float)r[k].y);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        n[k] = (int)(w[k] & 0xFFull);
        slow = slow || n[k] > 4;
        idx[k] = n[k] == 0xFF ? 0 : (int)((w[k] >> 8) & 0xFFull);
        const double2 c = s_table[idx[k]];
        const double dx = r[k].x - c.x, dy = r[k].y - c.y;
        best[k] = dx * dx + dy * dy;
    }
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        int m[K];
        double2 c[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            m[k] = (j < n[k] && n[k] != 0xFF) ? (int)((w[k] >> (8 * (j + 1))) & 0xFFull) : idx[k];
            c[k] = s_table[m[k]];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double dx = r[k].x - c[k].x, dy = r[k].y - c[k].y;
            const double d = dx * dx + dy * dy;
            if (d < best[k]) {
                best[k] = d;
                idx[k] = m[k];
            }
        }
    }
    if (slow) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (n[k] > 4) idx[k] = demod_grid(s_table, s_grid, g, M, r[k]);     // long lists and the 0xFF "sweep everything" marker
    }
}

// ---- the table searches as callers see them: the sector certificate of an 8- / 16-PSK first (g.psk, wave-uniform), the candidate
//      grid for whatever it does not certify -- identical decisions (tests/test_demod_cert.py) ----
__device__ __forceinline__ int demod_grid(const float2* __restrict__ s_table, const unsigned long long* __restrict__ s_grid,
                                          const DemodGrid& g, int M, float2 r) {
    if (g.psk != nullptr) {
        bool sure;
        const int lab = demod_psk_cert<float>(r, g.psk, M, sure);
        if (sure) return lab;
    }
    return demod_grid_search(s_table, s_grid, g, M, r);
}
__device__ __forceinline__ int demod_grid(const double2* __restrict__ s_table, const unsigned long long* __restrict__ s_grid,
                                          const DemodGrid& g, int M, double2 r) {
    if (g.psk != nullptr) {
        bool sure;
        const int lab = demod_psk_cert<double>(r, g.psk, M, sure);
        if (sure) return lab;
    }
    return demod_grid_search(s_table, s_grid, g, M, r);
}
__device__ __forceinline__ int demod_grid4(const float4* __restrict__ s_tab4, const unsigned long long* __restrict__ s_grid,
                                           const DemodGrid& g, int M, float2 r) {
    if (g.psk != nullptr) {
        bool sure;
        const int lab = demod_psk_cert<float>(r, g.psk, M, sure);
        if (sure) return lab;
    }
    return demod_grid4_search(s_tab4, s_grid, g, M, r);
}
template <int K>
__device__ __forceinline__ void demod_grid4_multi(const float4* __restrict__ s_tab4, const unsigned long long* __restrict__ s_grid,
                                                  const DemodGrid& g, int M, const float2 (&r)[K], int (&idx)[K]) {
    if (g.psk != nullptr) {
        bool all = true;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            bool sure;
            idx[k] = demod_psk_cert<float>(r[k], g.psk, M, sure);
            all = all && sure;
        }
        if (all) return;
    }
    demod_grid4_multi_search<K>(s_tab4, s_grid, g, M, r, idx);
}
template <int K>
__device__ __forceinline__ void demod_grid_multi(const double2* __restrict__ s_table, const unsigned long long* __restrict__ s_grid,
                                                 const DemodGrid& g, int M, const double2 (&r)[K], int (&idx)[K]) {
    if (g.psk != nullptr) {
        bool all = true;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            bool sure;
            idx[k] = demod_psk_cert<double>(r[k], g.psk, M, sure);
            all = all && sure;
        }
        if (all) return;
    }
    demod_grid_multi_search<K>(s_table, s_grid, g, M, r, idx);
}

template <typename T>
__device__ __forceinline__ void load_grid(const ModemParams<T>& mp, unsigned long long* s_grid) {
    const int cells = mp.grid.G * mp.grid.G;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) s_grid[i] = mp.grid.cells[i];
}

__device__ __forceinline__ int gray2binary8(int g) {
    g ^= g >> 4;
    g ^= g >> 2;
    g ^= g >> 1;
    return g;
}

// square Gray QAM: label (gray^-1(row) << k/2) | gray^-1(col); row 0 = +max imag, col 0 = -max real
// (fundamental.py:697-777).  Decision-identical to demod_mindist away from exact ties.
template <typename T>
__device__ __forceinline__ int demod_qam_slicer(cx<T> r, T scale, int L, int half_bits) {
    const T lm1 = (T)(L - 1);
    T fj = floor((r.x * scale + lm1) * (T)0.5 + (T)0.5);
    T fi = floor((lm1 - r.y * scale) * (T)0.5 + (T)0.5);
    fj = fj < (T)0 ? (T)0 : (fj > lm1 ? lm1 : fj);
    fi = fi < (T)0 ? (T)0 : (fi > lm1 ? lm1 : fi);
    return (gray2binary8((int)fi) << half_bits) | gray2binary8((int)fj);
}

// Min-distance decision of a square Gray QAM WITHOUT touching the table, with a certificate.  The constellation is the
// closed form (-(L-1) + 2 j) / scale, ((L-1) - 2 i) / scale that mcle_set_constellation verified entry by entry (1e-12), so
// the nearest point per axis is the nearest level: t = level coordinate of the received value, k = rint(clamp(t)), and
// f = clamp(t) - k is the offset from that level in units of the level spacing.  With |f| <= 1/2 - eps on both axes the
// nearest point beats every other by >= 2 eps (spacing)^2 in squared distance -- orders of magnitude above what the
// rounding of t, of the table entries or of either metric (|c - r|^2 here, hypot in numpy.abs, fundamental.py:241-246)
// can move -- so the exhaustive sweep, its first-minimum tie rule included, returns this very label: `sure`.  (That margin
// argument needs |re|, |im| within ~20 (complex64) / ~2^11 (complex128) level spacings: complex64 checks it, complex128 does NOT
// and is identical to the sweep BY CONSTRUCTION only inside that range; beyond it the identity is a probability bound,
// <= 1e-14 per symbol, see the note inside the function and include/mcle.h MCLE_OPT_DEMOD_NOCERT.)  Otherwise
// (a point within eps of a decision boundary: probability ~ 4 eps per symbol) the caller runs the table search it
// always ran.  eps = 2^-30 for complex128 (t <= 16 carries an error of 4e-15), 2^-15 for complex64 (t <= 32 carries an error
// of <= 4e-6: eight times below; a wavefront pass of 256 symbols then takes the table search in 3 % of its passes -- 2^-12,
// the first setting, sent 40 % of them there and left the complex64 min-distance rate 15 % under the slicer's).
// What this buys: no data-dependent LDS gathers (cell word + 1..4 table entries per symbol: they were the bank conflicts
// of the min-distance kernels, conflict fraction 0.41-0.64) and ~25 instead of ~45 instructions per symbol.
template <typename T>
__device__ __forceinline__ int demod_qam_cert(cx<T> r, T scale, int L, int half_bits, bool& sure) {
    constexpr T lim = sizeof(T) == 8 ? (T)(0.5 - 0x1p-30) : (T)(0.5 - 0x1p-15);
    const T lm1 = (T)(L - 1), hs = scale * (T)0.5, hl = lm1 * (T)0.5;
    T tj = r.x * hs + hl, ti = hl - r.y * hs;                      // level coordinates (col from -max real, row from +max imag)
    tj = fmin(fmax(tj, (T)0), lm1);                               // beyond the outer levels: certain (f = 0); NaN -> 0
    ti = fmin(fmax(ti, (T)0), lm1);
    const T kj = rint(tj), ki = rint(ti);
    sure = fabs(tj - kj) <= lim && fabs(ti - ki) <= lim;
    // Beyond the outer levels the clamp makes f = 0 on that axis, but the margin on the OTHER axis (2 eps spacing^2) only dominates
    // the rounding of the metrics (~ |r|^2 ulp) while |r| stays below ~20 level spacings in complex64 and ~2^11 in complex128
    // (ADVICE r04).  complex64: `sure` also needs |re|, |im| within 16 spacings of the outermost level's centre -- a zero-forcing
    // output in a deep fade goes farther, and the two compares are free there (scripts/experiments/r05_calls.txt [call 21]).  complex128: NOT
    // checked -- two f64 compares per symbol cost the headline kernel 2.3 % (11.01 against 10.76 ms per 262 144 realizations), and a
    // point beyond 2^11 spacings (an equaliser output in a fade of -66 dB) that ALSO sits within 2^-30 of a boundary on its other
    // axis has probability ~1e-14 per symbol: there the identity with the sweep rests on the margin argument up to 2^11 spacings
    // and on test coverage beyond (include/mcle.h says so).
    if constexpr (sizeof(T) == 4) {
        const T rmax = ((T)16 + hl) / hs;                         // wave-uniform
        sure = sure && fabs(r.x) <= rmax && fabs(r.y) <= rmax;
    }
    unsigned v = ((unsigned)(int)ki << 8) | (unsigned)(int)kj;     // both Gray decodes at once, a byte each (levels < 2^8)
    v ^= (v >> 4) & 0x0F0Fu;
    v ^= (v >> 2) & 0x3F3Fu;
    v ^= (v >> 1) & 0x7F7Fu;
    return (int)(((v >> 8) << half_bits) | (v & 0xFFu));
}

// Min-distance decision of four points (+-a, +-b), one per quadrant, WITHOUT touching the table: the regions are the quadrants.
// Two candidates mirrored in an axis differ by 4 a |re| (4 b |im|) in squared distance, by >= a |re| / |d| in distance (NumPy
// compares |c - r|); with lo = 2^-30 min(a, b) <= |re|, |im| <= hi = 2^8 max(a, b) that is >= 2^-39 a against a rounding of
// <= 2^-43 a of either metric in complex128 -- the sweep, first-minimum rule included, returns this very label: `sure`.
// Outside (a point on an axis to 1e-9, or an equaliser output beyond 256 a in a deep fade: both rare) the caller searches the
// table as before.  complex64: lo = 2^-15 min(a, b), the same figure as the QAM certificate's.
template <typename T>
__device__ __forceinline__ int demod_quad_cert(cx<T> r, unsigned lut, T lo, T hi, bool& sure) {
    const T ax = fabs(r.x), ay = fabs(r.y);
    sure = ax >= lo && ay >= lo && ax <= hi && ay <= hi;             // NaN: not sure
    const unsigned q = (r.x < (T)0 ? 8u : 0u) | (r.y < (T)0 ? 16u : 0u);
    return (int)((lut >> q) & 0xFFu);
}
// The same for four points ON the axes, (+-a, 0) and (0, +-a) (the reference's PSK(4)): the regions are bounded by the diagonals.
// With u = re - im, v = re + im (their signs are exact in floating point) the nearest point is (a, 0) for u, v > 0, (0, a) for
// u < 0 < v, (0, -a) for v < 0 < u, (-a, 0) for u, v < 0, and the runner-up is farther by 2 a min(|u|, |v|) in squared distance:
// with lo = 2^-30 a <= |u|, |v| and |re|, |im| <= hi = 2^8 a that is >= 2^-29 a^2 against a rounding of <= 2^-35 a^2 of either
// metric in complex128 (|c - r|^2 here, numpy.abs there) -- the sweep, first-minimum rule included, returns this very label: `sure`.
// lut: label of (u < 0) | (v < 0) << 1, a byte each.  Used by the complex128 symbol walks (walk_f64.hpp, WDEC_AXIS4_CERT).
template <typename T>
__device__ __forceinline__ int demod_axis4_cert(cx<T> r, unsigned lut, T lo, T hi, bool& sure) {
    const T u = r.x - r.y, v = r.x + r.y;
    sure = fabs(u) >= lo && fabs(v) >= lo && fabs(r.x) <= hi && fabs(r.y) <= hi;      // NaN: not sure
    const unsigned q = (u < (T)0 ? 8u : 0u) | (v < (T)0 ? 16u : 0u);
    return (int)((lut >> q) & 0xFFu);
}
template <typename T> __device__ __forceinline__ int demod_cert_any(const ModemParams<T>& mp, cx<T> r, bool& sure) {
    if (mp.cert == 2) return demod_quad_cert<T>(r, mp.quad_lut, mp.quad_lo, mp.quad_hi, sure);
    return demod_qam_cert<T>(r, mp.qam_scale, mp.qam_L, mp.half_bits, sure);
}

template <typename T>
__device__ __forceinline__ int demod_one(const ModemParams<T>& mp, const cx<T>* s_table, cx<T> r) {
    if (mp.method == MCLE_DEMOD_QAM_SLICER) return demod_qam_slicer<T>(r, mp.qam_scale, mp.qam_L, mp.half_bits);
    if (mp.cert) {
        bool sure;
        const int dec = demod_cert_any<T>(mp, r, sure);
        if (sure) return dec;
    }
    return demod_mindist<T>(s_table, mp.M, r);
}
// with the candidate grid in LDS (s_grid may be anything when mp.grid.G == 0 or T = double)
template <typename T>
__device__ __forceinline__ int demod_one(const ModemParams<T>& mp, const cx<T>* s_table,
                                         const unsigned long long* s_grid, cx<T> r) {
    if (mp.method == MCLE_DEMOD_QAM_SLICER) return demod_qam_slicer<T>(r, mp.qam_scale, mp.qam_L, mp.half_bits);
    if (mp.cert) {
        bool sure;
        const int dec = demod_cert_any<T>(mp, r, sure);
        if (sure) return dec;
    }
    if (mp.grid.G > 0) return demod_grid(s_table, s_grid, mp.grid, mp.M, r);   // G == 0: no grid bound to this launch
    return demod_mindist<T>(s_table, mp.M, r);
}

// K symbols: all K certificates first (straight-line), the table search only in lanes that hold an uncertified symbol.
// `search(idx)` is the caller's lockstep search over all K (demod_grid_multi / demod_grid4_multi / demod_mindist_multi);
// it gives the same labels as the certificate wherever that one is sure, so overwriting all K in such a lane is harmless.
template <typename T, int K, typename Search>
__device__ __forceinline__ void demod_multi_cert(const ModemParams<T>& mp, const cx<T> (&r)[K], int (&idx)[K], Search&& search) {
    if (mp.cert) {
        bool all = true;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            bool sure;
            idx[k] = demod_cert_any<T>(mp, r[k], sure);
            all = all && sure;
        }
        if (all) return;
    }
    search(idx);
}

// host: candidate grid of the context for the f32 instantiation; for f64 only where the caller asks for it (the operator
// kernels: same decisions as the sweep at a fifth of the cost; the fused f64 parity pipelines sweep everything)
template <typename T> inline DemodGrid context_grid(const mcle_ctx* ctx, int method = MCLE_DEMOD_MINDIST, bool f64_too = false) {
    DemodGrid g;
    g.cells = ctx->d_grid;
    g.G = ((sizeof(T) == 4 || f64_too) && ctx->d_grid != nullptr && method == MCLE_DEMOD_MINDIST) ? ctx->grid_G : 0;   // the slicer needs none
    g.x0 = ctx->grid_x0;
    g.y0 = ctx->grid_y0;
    g.inv_h = ctx->grid_inv_h;
    g.psk = (ctx->psk_ok && g.G > 0 && !ctx->opt[MCLE_OPT_DEMOD_NOCERT]) ? reinterpret_cast<const PskCert*>(ctx->d_psk) : nullptr;
    return g;
}

// host: does a launch with this method decide through the margin certificate (demod_qam_cert)?
inline int modem_cert(const mcle_ctx* ctx, int method) {
    if (method != MCLE_DEMOD_MINDIST || ctx->opt[MCLE_OPT_DEMOD_NOCERT]) return 0;
    if (ctx->kind == MCLE_CONST_QAM && ctx->qam_L >= 2 && ctx->qam_L <= 256) return 1;
    return ctx->quad_ok ? 2 : 0;      // (an 8- / 16-PSK is certified inside the table searches: DemodGrid::psk)
}

// cooperative copy of the constellation into LDS (call before a __syncthreads())
// host: the certificate fields of a launch's modem parameters
template <typename T> inline void modem_fill_cert(const mcle_ctx* ctx, int method, ModemParams<T>& p) {
    p.cert = modem_cert(ctx, method);
    p.quad_lut = ctx->quad_lut;
    p.quad_lo = (T)(ctx->quad_min * (sizeof(T) == 8 ? 0x1p-30 : 0x1p-15));
    p.quad_hi = (T)(ctx->quad_max * 256.0);
}

template <typename T>
__device__ __forceinline__ void load_table(const ModemParams<T>& mp, cx<T>* s_table) {
    for (int m = threadIdx.x; m < mp.M; m += blockDim.x) s_table[m] = mp.g_table[m];
}

}  // namespace mcle
