// Grouped strided conv (the discriminator's k41 / stride-4 / 4-channels-per-group layers,
// reference discriminator/full.py:15-18) on the fp32 matrix cores.
//
// Per group the layer is a small GEMM: M = outputs per group (16, or 4 for the 256-group layer),
// K = 4 input channels x 41 taps, N = B x Lout.  v_mfma_f32_16x16x4_f32 fits it exactly: the MFMA
// k index is the input channel (4), one MFMA per tap.  A wave keeps the 41 weight fragments of its
// group in registers and sweeps 16-output time tiles; the input rows are staged in LDS once per
// workgroup in a phase-split layout ([ci][u mod stride][u / stride]) so that the stride-4 gather
// of tap j becomes a unit-stride, conflict-free ds_read for every lane.
#include "ms_common.h"
#include "gconv_mfma.h"
#include <stdint.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GK = 41, GS = 4, GCG = 4;      // taps, stride, input channels per group

// Batched staging: every global load of a fill is issued before the first LDS store that consumes
// one (clamped address + mask, fixed trip count).  A `for (idx = tid; idx < n; idx += 256)` fill
// waits for each load before its store: one memory round trip per iteration, which made these
// HBM-sized layers latency-bound (41 us for a 33 MB layer).
template <int MAXIT>
__device__ inline void stage_weights(const float* __restrict__ w, float* __restrict__ ws, int n, int tid) {
    float v[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int idx = tid + it * 256;
        v[it] = w[idx < n ? idx : 0];
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int idx = tid + it * 256;
        if (idx < n) ws[idx] = v[it];
    }
}

// ---------------------------------------------------------------- forward
// Work unit of a WAVE: 64 consecutive outputs of one (batch row, group).  The wave stages the
// 4 x 293 inputs the unit reads into its private LDS region (phase-split: [ci][u mod 4][u / 4],
// so the stride-4 gather of tap j is a unit-stride conflict-free ds_read), runs 4 independent
// 16-output MFMA chains of 41 taps, and prefetches the next unit's inputs into registers while
// the matrix cores work.  No workgroup barrier after the weight fill: short rows (L = 9 .. 65 at
// the coarse scales) keep all four waves busy on different batch rows.
//
// (16-byte loads / an LDS-transposed 16-byte store path were built and measured SLOWER -- 95 vs 80 us
// at B = 128: the kernel is bound by vector-instruction issue beside the matrix pipe, not by memory
// instructions -- and removed.)
constexpr int UT = 64;                               // outputs per unit
constexpr int USPAN = (UT - 1) * GS + GK;            // 293 inputs per channel
constexpr int UK = (USPAN + 63) / 64;                // 5 dword loads per lane and channel
constexpr int UPS = 76;                              // phase-row pitch: >= 74, == 4 (mod 8)
constexpr int UWS = GCG * GS * UPS;                  // floats per wave region

template <int NTT>
__device__ inline void gfwd_tiles(const float (&a)[GK], const float* __restrict__ xb, f32x4 (&acc)[4]) {
#pragma unroll
    for (int j = 0; j < GK; ++j) {
        float bv[NTT];
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) bv[tt] = xb[(j & (GS - 1)) * UPS + tt * 16 + (j >> 2)];
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bv[tt], acc[tt], 0, 0, 0);
    }
}

// Lean scalar bookkeeping (PMC: the first version spent 553 vector + 212 scalar instructions per
// 164 MFMAs of a unit, mostly 64-bit address arithmetic and per-unit divisions, and the matrix pipe
// sat at 37 %): wave-uniform 64-bit bases + 32-bit lane offsets, unit -> (b, tile) advanced
// incrementally, LDS / output offsets hoisted out of the unit loop, activation as a template flag.
template <bool LRELU>
__global__ __launch_bounds__(256, 3) void k_gconv_mfma_fwd(ConvP p, int tiles, int nunits,
                                                       const float* __restrict__ x,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ y) {
    __shared__ float xs[4 * UWS];
    __shared__ float ws[16 * GCG * GK];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = blockIdx.y;
    const int m = lane & 15, ci = lane >> 4;
    float* xw = xs + wid * UWS;
    const float* xb = xw + ci * (GS * UPS) + m;
    const int wstride = gridDim.x * 4;
    const int db = wstride / tiles, dt = wstride - db * tiles;      // unit += wstride, in (b, tile) form

    // loop-invariant lane offsets
    int lo[UK];                                  // LDS offset of input u = lane + 64k (phase-split)
#pragma unroll
    for (int k = 0; k < UK; ++k) {
        const int u = lane + 64 * k;
        lo[k] = (u & (GS - 1)) * UPS + (u >> 2);
    }
    unsigned ro[4];                              // output offset of accumulator row r (relative to the unit)
    bool rok[4];
    float bq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int mo = ci * 4 + r;
        rok[r] = mo < p.Og;
        ro[r] = (unsigned)(mo * p.Lout + m);
        bq[r] = bias ? bias[g * p.Og + (rok[r] ? mo : 0)] : 0.f;
    }

    float xr[GCG][UK];
    auto gload = [&](int b, int ti) {
        const int u0 = ti * (UT * GS) - p.pad;
        const float* xg = x + ((size_t)b * p.Cin + (size_t)g * GCG) * p.Lin;      // wave-uniform base
#pragma unroll
        for (int k = 0; k < UK; ++k) {
            const int u = lane + 64 * k, sidx = u0 + u;
            const bool ok = u < USPAN && sidx >= 0 && sidx < p.Lin;
            const unsigned so = ok ? (unsigned)sidx : 0u;
#pragma unroll
            for (int c = 0; c < GCG; ++c) xr[c][k] = xg[so + (unsigned)(c * p.Lin)];
        }
    };   // (masking happens at the LDS store, so the loads stay in flight across the MFMA loop)
    int unit = blockIdx.x * 4 + wid;
    int b = unit / tiles, ti = unit - b * tiles;
    if (unit < nunits) gload(b, ti);     // in flight while the weights are staged
    const int nwf = p.Og * GCG * GK;
    stage_weights<(16 * GCG * GK + 255) / 256>(w + (size_t)g * nwf, ws, nwf, tid);
    __syncthreads();
    // weight fragments: lane (m = lane&15, ci = lane>>4) holds w[g*Og+m][ci][j] for every tap j
    float a[GK];
    {
        const bool ok = m < p.Og;
        const float* wr = ws + ((ok ? m : 0) * GCG + ci) * GK;   // lane stride 41: conflict-free
#pragma unroll
        for (int j = 0; j < GK; ++j) {
            const float v = wr[j];
            a[j] = ok ? v : 0.f;
        }
    }

    for (; unit < nunits; unit += wstride) {
        const int t0 = ti * UT, u0 = t0 * GS - p.pad;
#pragma unroll
        for (int k = 0; k < UK; ++k) {
            const int u = lane + 64 * k, sidx = u0 + u;
            const bool ok = u < USPAN && sidx >= 0 && sidx < p.Lin;
            if (u < GS * UPS) {
#pragma unroll
                for (int c = 0; c < GCG; ++c) xw[c * (GS * UPS) + lo[k]] = ok ? xr[c][k] : 0.f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float* yg = y + ((size_t)b * p.Cout + (size_t)g * p.Og) * p.Lout + t0;     // wave-uniform base
        int nb = b + db, nti = ti + dt;
        if (nti >= tiles) { nti -= tiles; ++nb; }
        if (unit + wstride < nunits) gload(nb, nti);
        const int ntt = min(4, (p.Lout - t0 + 15) >> 4);          // wave-uniform
        f32x4 acc[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ntt == 4) gfwd_tiles<4>(a, xb, acc);
        else if (ntt == 3) gfwd_tiles<3>(a, xb, acc);
        else if (ntt == 2) gfwd_tiles<2>(a, xb, acc);
        else gfwd_tiles<1>(a, xb, acc);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            if (tt < ntt && t0 + tt * 16 + m < p.Lout) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[tt][r] + bq[r];
                    if (LRELU) v = v > 0.f ? v : v * p.slope;
                    else v = ms_apply_act(v, p.act, p.slope);
                    if (rok[r]) yg[ro[r] + tt * 16] = v;
                }
            }
        }
        b = nb; ti = nti;
    }
}


// ---------------------------------------------------------------- backward data
// gx[b, g*4+ci, 4q+r] = gx_add + sum_{co,jj} w[co, ci, r + 4jj] * gp[b, co, q + 5 - jj]
// (pad = 20 == 0 mod 4, so output phase r only sees taps k = r + 4jj).  GEMM per group and q-tile:
// M = (ci, r) = 16 rows, N = 16 consecutive q, K = co (4 per MFMA) x jj (11).  A lane's 4
// accumulator rows are the 4 phases r of one (ci, q): 4 consecutive samples of gx.
constexpr int JJ = (GK + GS - 1) / GS;        // 11 taps per phase
constexpr int TQ = 64;                        // q's per unit (4 MFMA column tiles)
constexpr int BSP = TQ + 2 * (JJ - 1);        // 84 gradient columns a unit reads per output channel
constexpr int BGR = 112;                      // LDS row pitch: >= 84, == 16 (mod 32)

template <int OQ, int NTQ>
__device__ inline void gbwd_tiles(const float (&a)[OQ * JJ], const float* __restrict__ gb, f32x4 (&acc)[4]) {
#pragma unroll
    for (int cq = 0; cq < OQ; ++cq)
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            float bv[NTQ];
#pragma unroll
            for (int tq = 0; tq < NTQ; ++tq) bv[tq] = gb[cq * 4 * BGR + tq * 16 - jj];
#pragma unroll
            for (int tq = 0; tq < NTQ; ++tq)
                acc[tq] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cq * JJ + jj], bv[tq], acc[tq], 0, 0, 0);
        }
}

// Work unit of a WAVE: 64 consecutive q (256 samples of the 4 input-channel rows) of one (batch row,
// group): private LDS tile of the Og x 84 gradient values it needs (activation derivative applied),
// 4 independent MFMA chains, next unit prefetched into registers, 16-byte stores where the rows allow.
template <int OQ, bool VOUT>   // OQ = Og / 4 co-quads (4 for Og = 16, 1 for Og = 4)
__global__ __launch_bounds__(256, 2) void k_gconv_mfma_bwd_data(ConvP p, int tiles, int nunits,
                                                               const float* __restrict__ gy,
                                                               const float* __restrict__ y_act,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ gx_add,
                                                               float* __restrict__ gx) {
    constexpr int ROWS = OQ * 4;
    constexpr int NLD = (ROWS * BSP + 63) / 64;       // 21 (Og = 16) / 6 (Og = 4) loads per lane and tensor
    __shared__ float gsm[4 * ROWS * BGR];
    __shared__ float ws[ROWS * GCG * GK];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = blockIdx.y;
    float* gsw = gsm + wid * ROWS * BGR;
    const float* ya = y_act ? y_act : gy;
    const int kind = y_act ? p.act : MS_ACT_NONE;
    const int wstride = gridDim.x * 4;
    const int db = wstride / tiles, dt = wstride - db * tiles;

    // lane-invariant: element idx = lane + 64 i -> (row co, column tt) of the tile
    unsigned loff[NLD];                                // global offset relative to (b, g, q0 - 5): co*Lout + tt
    int lds_o[NLD], ltt[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = lane + 64 * i;
        const int co = idx / BSP, tt = idx - co * BSP;    // compile-time divisor
        ltt[i] = co < ROWS ? tt : -1000000;
        loff[i] = (unsigned)((co < ROWS ? co : 0) * p.Lout);
        lds_o[i] = (co < ROWS ? co : 0) * BGR + tt;
    }
    float gv[NLD], av[NLD];
    auto gload = [&](int b, int ti) {
        const int tlo = ti * TQ + 5 - (JJ - 1);            // first gradient index of the unit (may be < 0)
        const size_t base = ((size_t)b * p.Cout + (size_t)g * p.Og) * p.Lout;
        const float* gb_ = gy + base;
        const float* ab_ = ya + base;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int t = tlo + ltt[i];
            const bool ok = t >= 0 && t < p.Lout;         // (ltt = -1e6 for lanes past the tile)
            const unsigned o = ok ? loff[i] + (unsigned)t : 0u;
            gv[i] = gb_[o];
            av[i] = ab_[o];
        }
    };
    int unit = blockIdx.x * 4 + wid;
    int b = unit / tiles, ti = unit - b * tiles;
    if (unit < nunits) gload(b, ti);
    stage_weights<(ROWS * GCG * GK + 255) / 256>(w + (size_t)g * p.Og * GCG * GK, ws, ROWS * GCG * GK, tid);
    __syncthreads();
    // weight fragments: lane (m = (ci, r) = lane&15, k = lane>>4): w[g*Og + 4cq + k][ci][r + 4jj]
    const int mrow = lane & 15, ci = mrow >> 2, r = mrow & 3, kq = lane >> 4;
    float a[OQ * JJ];
#pragma unroll
    for (int cq = 0; cq < OQ; ++cq)
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            const int k = r + 4 * jj;
            const bool ok = k < GK;
            const float v = ws[((cq * 4 + kq) * GCG + ci) * GK + (ok ? k : 0)];
            a[cq * JJ + jj] = ok ? v : 0.f;
        }
    // gp index for (q, jj): q + 5 - jj  ->  LDS column (q - q0) + (JJ-1) - jj
    const float* gb = gsw + kq * BGR + (lane & 15) + (JJ - 1);
    const int Lq = (p.Lin + GS - 1) / GS;

    for (; unit < nunits; unit += wstride) {
        const int q0 = ti * TQ;
        const int tlo = q0 + 5 - (JJ - 1);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int t = tlo + ltt[i];
            const bool ok = t >= 0 && t < p.Lout;
            if (ltt[i] >= 0) gsw[lds_o[i]] = ok ? ms_act_grad(gv[i], av[i], kind, p.slope) : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // D[row][col]: col = lane&15 = q, row = (lane>>4)*4 + reg = ci*4 + r  ->  ci = lane>>4, r = reg
        const size_t rowoff = ((size_t)b * p.Cin + (size_t)g * GCG + (lane >> 4)) * p.Lin;
        const int ntq = min(4, (Lq - q0 + 15) >> 4);       // wave-uniform
        float4 addv[4];
        if (VOUT && gx_add) {                              // residual gradient: 16-byte loads, batched
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                const int sidx = (q0 + tq * 16 + (lane & 15)) * GS;
                addv[tq] = *reinterpret_cast<const float4*>(gx_add + (sidx < p.Lin ? rowoff + sidx : 0));
            }
        }
        int nb = b + db, nti = ti + dt;
        if (nti >= tiles) { nti -= tiles; ++nb; }
        if (unit + wstride < nunits) gload(nb, nti);
        f32x4 acc[4];
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) acc[tq] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ntq == 4) gbwd_tiles<OQ, 4>(a, gb, acc);
        else if (ntq == 3) gbwd_tiles<OQ, 3>(a, gb, acc);
        else if (ntq == 2) gbwd_tiles<OQ, 2>(a, gb, acc);
        else gbwd_tiles<OQ, 1>(a, gb, acc);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
            const int sidx = (q0 + tq * 16 + (lane & 15)) * GS;
            if (tq >= ntq || sidx >= p.Lin) continue;
            if (VOUT) {                                    // Lin % 4 == 0: the 4 samples exist together
                float4 v = make_float4(acc[tq][0], acc[tq][1], acc[tq][2], acc[tq][3]);
                if (gx_add) { v.x += addv[tq].x; v.y += addv[tq].y; v.z += addv[tq].z; v.w += addv[tq].w; }
                *reinterpret_cast<float4*>(gx + rowoff + sidx) = v;
            } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    if (sidx + rr < p.Lin) {
                        float v = acc[tq][rr];
                        if (gx_add) v += gx_add[rowoff + sidx + rr];
                        gx[rowoff + sidx + rr] = v;
                    }
                }
            }
        }
        b = nb; ti = nti;
    }
}


// --------------------------------------------------------------- backward weight
// gw[co, ci, k] = sum_{b,t} gp[b, co, t] * x[b, g*4+ci, 4t + k - 20] per group: M = co (16),
// N = (ci, k) = 164 columns in 11 MFMA tiles, MFMA k index = 4 consecutive t.  Every wave sums
// its own quarter of each 256-output chunk into registers (11 x f32x4) and writes ONE slab per
// wave at the end; slabs are combined by the deterministic split-K reduce (k_reduce_slabs).
// Work unit of a WAVE (as in the forward kernel): 64 consecutive outputs of one batch row.  The wave
// stages the 16 x 64 gradient values and the 4 x 296 inputs of the unit in its private LDS region,
// runs 16 k-steps x 11 column tiles, and prefetches the next unit into registers meanwhile.  No
// workgroup barriers: at the coarse scales (L = 9 .. 65) all four waves work on different batch rows
// instead of three of them idling behind a 256-output chunk.
constexpr int WU = 64;                        // outputs per unit
constexpr int RSA = WU + 2;                   // == 2 (mod 32): (co, k) fragment reads conflict-free
constexpr int XSPAN = (WU - 1) * GS + GK;     // 293 inputs per channel and unit
constexpr int XW = 300;                       // LDS pitch of an input row (reads reach 292 + 3)
constexpr int NT = (GCG * GK + 15) / 16;      // 11 column tiles
constexpr int WGS = 16 * RSA, WXS = GCG * XW; // floats per wave: gradient / input region

__global__ __launch_bounds__(256) void k_gconv_mfma_wgrad(ConvP p, const float* __restrict__ x,
                                                         const float* __restrict__ gy,
                                                         const float* __restrict__ y_act,
                                                         float* __restrict__ partial,
                                                         size_t partial_stride) {
    __shared__ float lds[4 * (WGS + WXS)];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = blockIdx.y;
    const float* ya = y_act ? y_act : gy;
    const int kind = y_act ? p.act : MS_ACT_NONE;
    float* gs = lds + wid * (WGS + WXS);
    float* xs = gs + WGS;

    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    // B-fragment offsets of this lane: column nn = tile*16 + (lane&15) -> (ci, kk); k = lane>>4
    int xo[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        int nn = i * 16 + (lane & 15);
        if (nn >= GCG * GK) nn = 0;           // padded columns: computed, never stored
        const int ci = nn / GK, kk = nn - ci * GK;
        xo[i] = ci * XW + kk + 4 * (lane >> 4);
    }
    const int ao = (lane & 15) * RSA + (lane >> 4);

    const int tiles = (p.Lout + WU - 1) / WU;
    const int nunits = p.B * tiles;
    const int ustride = gridDim.x * 4;
    const int db = ustride / tiles, dt = ustride - db * tiles;     // unit += ustride, in (b, tile) form
    constexpr int NXI = (XSPAN + 63) / 64;             // 5 input loads per lane and channel
    float gv[16], ga[16], xv[GCG][NXI];
    // wave-uniform 64-bit bases + 32-bit lane offsets (the address arithmetic was most of this kernel)
    auto gload = [&](int b, int ti) {
        const int t0 = ti * WU, u0 = t0 * GS - p.pad;
        const bool tok = t0 + lane < p.Lout;
        const size_t gbase = ((size_t)b * p.Cout + (size_t)g * p.Og) * p.Lout + t0;
        const float* gyb = gy + gbase;
        const float* yab = ya + gbase;
#pragma unroll
        for (int co = 0; co < 16; ++co) {              // lane = output t, co = row
            const unsigned off = (tok && co < p.Og) ? (unsigned)(co * p.Lout + lane) : 0u;
            gv[co] = gyb[off];
            ga[co] = yab[off];
        }
        const float* xg = x + ((size_t)b * p.Cin + (size_t)g * GCG) * p.Lin;
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int u = lane + 64 * k, sidx = u0 + u;
            const bool ok = u < XSPAN && sidx >= 0 && sidx < p.Lin;
            const unsigned so = ok ? (unsigned)sidx : 0u;
#pragma unroll
            for (int c = 0; c < GCG; ++c) xv[c][k] = xg[so + (unsigned)(c * p.Lin)];
        }
    };
    int unit = blockIdx.x * 4 + wid;
    int b = unit / tiles, ti = unit - b * tiles;
    if (unit < nunits) gload(b, ti);
    for (; unit < nunits; unit += ustride) {
        const int t0 = ti * WU;
        const int u0 = t0 * GS - p.pad;
        const bool tok = t0 + lane < p.Lout;
#pragma unroll
        for (int co = 0; co < 16; ++co)
            gs[co * RSA + lane] = (tok && co < p.Og) ? ms_act_grad(gv[co], ga[co], kind, p.slope) : 0.f;
#pragma unroll
        for (int k = 0; k < NXI; ++k) {
            const int u = lane + 64 * k, sidx = u0 + u;
            const bool ok = u < XSPAN && sidx >= 0 && sidx < p.Lin;
            if (u < XW) {
#pragma unroll
                for (int c = 0; c < GCG; ++c) xs[c * XW + u] = ok ? xv[c][k] : 0.f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int nb = b + db, nti = ti + dt;
        if (nti >= tiles) { nti -= tiles; ++nb; }
        if (unit + ustride < nunits) gload(nb, nti);
        const int tvalid = min(WU, p.Lout - t0);
        const int isteps = (tvalid + 3) >> 2;          // wave-uniform; outputs past tvalid are zero in gs
#pragma unroll 4
        for (int i = 0; i < isteps; ++i) {             // MFMA k-steps of 4 outputs
            const float a = gs[ao + 4 * i];
            bsum += a;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float bv = xs[xo[n] + 16 * i];
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[n], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        b = nb; ti = nti;
    }

    float* part = partial + (size_t)(blockIdx.x * 4 + wid) * partial_stride;
    const int J = GCG * GK;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int nn = n * 16 + (lane & 15);
        if (nn >= J) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = (lane >> 4) * 4 + r;
            if (m < p.Og) part[(size_t)(g * p.Og + m) * J + nn] = acc[n][r];
        }
    }
    // bias grad: lane (co, k) summed the outputs with t % 4 == k; combine the 4 k lanes
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (lane < 16 && lane < p.Og) part[(size_t)p.Cout * J + g * p.Og + lane] = bsum;
}

__global__ __launch_bounds__(256) void k_reduce_slabs(const float* __restrict__ partial,
                                                     size_t partial_stride, int nsplit,
                                                     size_t wsize, int nbias,
                                                     float* __restrict__ gw,
                                                     float* __restrict__ gb, float beta) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    const bool ok = i < wsize + (size_t)nbias;
    float s0 = 0.f, s1 = 0.f;
    if (ok) {
        int z = wv;
        for (; z + 4 < nsplit; z += 8) {
            s0 += partial[(size_t)z * partial_stride + i];
            s1 += partial[(size_t)(z + 4) * partial_stride + i];
        }
        for (; z < nsplit; z += 4) s0 += partial[(size_t)z * partial_stride + i];
    }
    red[wv][lane] = s0 + s1;
    __syncthreads();
    if (wv == 0 && ok) {
        const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (i < wsize) gw[i] = (beta != 0.f ? beta * gw[i] : 0.f) + s;
        else if (gb) gb[i - wsize] = (beta != 0.f ? beta * gb[i - wsize] : 0.f) + s;
    }
}

}  // namespace

bool msg_fwd_applicable(const ConvP& p) {
    return p.K == GK && p.stride == GS && p.Cg == GCG && p.dil == 1 && p.pad_mode == MS_PAD_ZERO &&
           p.Og <= 16 && p.groups <= 65535 && p.B <= 65535;
}

const char* msg_fwd_name(const ConvP&) { return "k_gconv_mfma_fwd"; }

int msg_conv1d_fwd(const ConvP& p, const float* x, const float* w, const float* bias, float* y,
                   hipStream_t s) {
    const int tiles = ms_ceil_div(p.Lout, UT), nunits = p.B * tiles;
    // 4 wave units per workgroup pass; ~2048 workgroups at most, each wave then loops over units
    // ~2 waves per SIMD (2048 waves): a wave loops over `upw` units so that its loads, MFMA chains
    // and stores of consecutive units overlap
    const int target_waves = 2048;
    const long long total_units = (long long)nunits * p.groups;
    const int upw = (int)((total_units + target_waves - 1) / target_waves);
    const int gx = ms_ceil_div(nunits, 4 * (upw > 0 ? upw : 1));
    // (16-byte loads / stores were measured SLOWER here -- 95 vs 80 us at B=128: the kernel is bound by
    // vector-instruction issue beside the matrix pipe, not by memory instructions)
    const dim3 grid(gx, p.groups);
    if (p.act == MS_ACT_LRELU) hipLaunchKernelGGL((k_gconv_mfma_fwd<true>), grid, dim3(256), 0, s, p, tiles, nunits, x, w, bias, y);
    else hipLaunchKernelGGL((k_gconv_mfma_fwd<false>), grid, dim3(256), 0, s, p, tiles, nunits, x, w, bias, y);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

bool msg_bwd_data_applicable(const ConvP& p) {
    return p.K == GK && p.stride == GS && p.Cg == GCG && p.dil == 1 && p.pad == 20 &&
           p.pad_mode == MS_PAD_ZERO && (p.Og == 16 || p.Og == 4) && p.groups <= 65535 && p.B <= 65535;
}

const char* msg_bwd_data_name(const ConvP& p) {
    return p.Og == 16 ? "k_gconv_mfma_bwd_data<4>" : "k_gconv_mfma_bwd_data<1>";
}

int msg_conv1d_bwd_data(const ConvP& p, const float* gy, const float* y_act, const float* w,
                        const float* gx_add, float* gx, hipStream_t s) {
    const int Lq = ms_ceil_div(p.Lin, GS);
    const int tiles = ms_ceil_div(Lq, TQ), nunits = p.B * tiles;
    // ~2 waves per SIMD; a wave loops over `upw` units (loads / MFMAs / stores of consecutive units overlap)
    const long long total_units = (long long)nunits * p.groups;
    const int upw = (int)((total_units + 2047) / 2048);
    const int gxn = ms_ceil_div(nunits, 4 * (upw > 0 ? upw : 1));
    const dim3 grid(gxn, p.groups);
    const bool vout = p.Lin % 4 == 0 && (((uintptr_t)gx) & 15) == 0 && (!gx_add || (((uintptr_t)gx_add) & 15) == 0);
#define MS_GBWD(OQV)                                                                                         \
    do {                                                                                                     \
        if (vout) hipLaunchKernelGGL((k_gconv_mfma_bwd_data<OQV, true>), grid, dim3(256), 0, s, p, tiles,    \
                                     nunits, gy, y_act, w, gx_add, gx);                                      \
        else hipLaunchKernelGGL((k_gconv_mfma_bwd_data<OQV, false>), grid, dim3(256), 0, s, p, tiles,        \
                                nunits, gy, y_act, w, gx_add, gx);                                           \
    } while (0)
    if (p.Og == 16) MS_GBWD(4);
    else MS_GBWD(1);
#undef MS_GBWD
    MS_CHECK_LAUNCH();
    return MS_OK;
}

bool msg_bwd_weight_applicable(const ConvP& p) {
    return p.K == GK && p.stride == GS && p.Cg == GCG && p.dil == 1 && p.pad_mode == MS_PAD_ZERO &&
           p.Og <= 16 && p.groups <= 65535;
}

int msg_wgrad_gridx(const ConvP& p) {
    const size_t slab = ((size_t)p.Cout * GCG * GK + p.Cout) * sizeof(float);
    const int nchunks = ms_ceil_div(p.B * ms_ceil_div(p.Lout, WU), 4);   // 4 wave units per workgroup pass
    int gx = ms_ceil_div(512, p.groups);                 // ~2 workgroups per CU
    const size_t cap = (size_t)32 << 20;                 // slabs (4 per workgroup column) <= 32 MiB
    while (gx > 1 && (size_t)gx * 4 * slab > cap) --gx;
    if (gx > nchunks) gx = nchunks;
    return gx < 1 ? 1 : gx;
}

size_t msg_bwd_weight_ws(const ConvP& p) {
    return (size_t)msg_wgrad_gridx(p) * 4 * ((size_t)p.Cout * GCG * GK + p.Cout) * sizeof(float);
}

const char* msg_bwd_weight_name(const ConvP&) { return "k_gconv_mfma_wgrad"; }

int msg_conv1d_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act,
                          float* gw, float* gb, float beta, void* ws, size_t ws_bytes,
                          hipStream_t s) {
    if (!ws || ws_bytes < msg_bwd_weight_ws(p)) return MS_ERR_WORKSPACE;
    const int gxn = msg_wgrad_gridx(p);
    const size_t stride = (size_t)p.Cout * GCG * GK + p.Cout;
    float* partial = (float*)ws;
    hipLaunchKernelGGL(k_gconv_mfma_wgrad, dim3(gxn, p.groups), dim3(256), 0, s, p, x, gy, y_act,
                       partial, stride);
    MS_CHECK_LAUNCH();
    return msg_reduce_slabs(p, partial, stride, gxn * 4, gw, gb, beta, s);
}

int msg_reduce_slabs(const ConvP& p, const float* partial, size_t stride, int nslabs, float* gw, float* gb,
                     float beta, hipStream_t s) {
    const size_t wsize = (size_t)p.Cout * GCG * GK;
    const size_t total = wsize + p.Cout;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, partial,
                       stride, nslabs, wsize, p.Cout, gw, gb, beta);
    MS_CHECK_LAUNCH();
    return MS_OK;
}
