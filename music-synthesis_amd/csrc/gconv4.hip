// The discriminator's 256-group layer  Conv1d(1024, 1024, 41, stride=4, padding=20, groups=256) + LeakyReLU  (reference
// discriminator/full.py:18) over all scales in one launch each way: forward, backward data, weight gradient.
//
// A group is 4 input x 4 output channels: 656 weights and 164 multiply-adds per output.  On the matrix pipe a 16-row tile
// carries 4 useful rows (gconv_split.hip ran this layer at 49 us forward / 95 us weight gradient against ~15 us of HBM
// time).  Here it runs in plain fp32 -- exact operands, no split:
//   * item = (batch row, output position j): a lane owns one item, consecutive lanes consecutive positions, rows flattened
//     so that the short rows of the coarse scales (17, 9 outputs) fill the wave;
//   * the four output channels of a tap are ONE v_mfma_f32_4x4x1 (16 independent 4 x 4 outer products, the weights broadcast
//     from one block of a register): fp32 in, one rounding per product -- an fmaf chain at twice the vector pipe's rate;
//   * the 44-sample input window of an item is 11 ds_read_b128 per channel from a zero-padded LDS copy of the rows
//     (row ends and the padding are resolved once, in the staging pass);
//   * backward data is the same shape transposed: the item is the input QUAD 4j .. 4j+3, its window 11 gradient samples
//     per output channel, 16 accumulators (4 channels x 4 phases), tap k = 4m + r meets output j + 5 - m;
//   * weight gradient: one workgroup per group; the sum over items is the MFMA's own accumulation: per item three
//     instructions, block = one quad of the item's window (44 quads = 4 channels x 11), A = the item's four gradient
//     samples, so D[co] of lane (quad, j) IS gw[co][ci][4 quad + j] -- no cross-lane reduction, no slabs, no finish launch,
//     a fixed summation order.  Rows travel global -> registers -> LDS one chunk ahead of the arithmetic.
// 656 FMAs per item either way: 0.62 GMAC per pass at B = 64 x 3 scales = 16 us of the vector pipe at full rate.
#include "ms_common.h"
#include "gconv_mfma.h"
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned OOB = 0xF0000000u;
constexpr int NP = MS_CONV_PARTS_MAX;
constexpr int K4 = 41, PAD4 = 20, CG = 4, WPG = CG * CG * K4;      // 656 weights per group, [co][ci][k]
constexpr int NMIN = 8, NMAX = 64;                                 // outputs per row the kernels take (Lin 29 .. 256)
constexpr int ITEMS = 256;                                         // items per workgroup (forward, backward) / per chunk (wgrad)
constexpr int XS = 2600 * CG;                                      // floats: largest nrows * (4n + 44) over n = 8 .. 64, times the 4 channels
constexpr int NQX = (XS / 4 + 255) / 256;                          // quads of the x lines a thread stages (11)
constexpr int NEB = (ITEMS * CG + 8 * NMAX + 255) / 256;           // gradient samples a thread stages in the backward kernel (6)

// Table of parts.  n = Lout = ceil(Lin / 4); wg0: prefix sum of workgroups (chunks) per part.
struct G4Parts {
    int count, wg0[NP + 1];
    int B[NP], L[NP], n[NP];
    const float* a[NP];
    const float* b[NP];
    const float* c[NP];
    float* o[NP];
};
struct G4Part {
    int B, L, n, wg;
    const float* a;
    const float* b;
    const float* c;
    float* o;
};
// (compile-time indices into the by-value tables: conv5_img.hip)
__device__ __forceinline__ G4Part pick_part(const G4Parts& q, int wg) {
    G4Part p{q.B[0], q.L[0], q.n[0], wg, q.a[0], q.b[0], q.c[0], q.o[0]};
#pragma unroll
    for (int k = 1; k < NP; ++k)
        if (k < q.count && wg >= q.wg0[k]) p = G4Part{q.B[k], q.L[k], q.n[k], wg - q.wg0[k], q.a[k], q.b[k], q.c[k], q.o[k]};
    return p;
}

// 4 consecutive samples t .. t+3 of the row at element `row_elems` (length L): one 16-byte load at any 4-byte aligned
// address; samples outside [0, L) read 0.0.  t is a multiple of 4 (disc_parts.hip).
__device__ __forceinline__ f32x4 load_row4(__amdgpu_buffer_rsrc_t rs, unsigned row_elems, int t, int L) {
    const bool any = t >= 0 && t < L;
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, any ? (row_elems + (unsigned)t) * 4u : OOB, 0, 0));
#pragma unroll
    for (int e = 1; e < 4; ++e) v[e] = t + e < L ? v[e] : 0.f;
    return v;
}

extern __shared__ __attribute__((aligned(16))) float g4_smem[];

// rows [r_lo, r_lo + nrows) of the items [e0, e0 + ITEMS) of a part
__device__ __forceinline__ void item_rows(int e0, int total, int n, int& r_lo, int& nrows) {
    r_lo = e0 / n;
    const int e_hi = e0 + ITEMS - 1 < total - 1 ? e0 + ITEMS - 1 : total - 1;
    nrows = e_hi / n - r_lo + 1;
}

// v_mfma_f32_4x4x1_16B_f32 is 16 independent 4 x 4 outer products: D[r] of lane l += A[lane 4 (l / 4) + r] * B[lane l]
// (observed: tools/scratch/mfma4).  With the A operand broadcast from ONE block (cbsz = 4, abid = t: every block takes
// lanes 4t .. 4t+3) a lane's four accumulators are  acc[r] += w_r * x_lane:  the four FMAs of one tap for the lane's own item,
// in one instruction at twice the rate of four v_fmac, each product rounded once into fp32 -- bitwise an fmaf chain.
// The 656 weights of a group live in 11 registers: lane 4t + r of register R holds weight (r, pair 16 R + t), pair = 41 a + k
// with (r, a) = (output, input) channel in the forward kernel and (input, output) channel in the backward one.
// (Earlier forms: scalar loads + v_fmac v, s, v -- the compiler hoisted all 656 loads and spilled 430 scalar registers to
// vector lanes; v_fmac_f32_dpp row_newbcast from 41 registers -- correct, 66 us forward against 49 us of the matrix-pipe
// kernel it was to replace.)
constexpr int NPAIR = CG * K4;                   // 164 (channel, tap) pairs
constexpr int NWV = (NPAIR + 15) / 16;           // 11 weight registers

template <int T>
__device__ __forceinline__ void mfma_bc(f32x4& acc, float a, float b) {
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 4, T, 0);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// forward: wv[R] of lane 4t + r = w[4g + r][a][k], 41 a + k = 16 R + t;  backward (transposed): = w[4g + a][r][k]
template <bool BWD>
__device__ __forceinline__ void load_group_weights(const float* __restrict__ w, int g, float (&wv)[NWV]) {
    const int lane = threadIdx.x & 63, r = lane & 3, t = lane >> 2;
    const float* wg = w + (size_t)g * WPG;
#pragma unroll
    for (int R = 0; R < NWV; ++R) {
        const int pair = 16 * R + t, a = pair / K4, k = pair - a * K4;
        wv[R] = pair < NPAIR ? wg[((BWD ? a : r) * CG + (BWD ? r : a)) * K4 + k] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------ forward
// y[b, 4g + co, j] = act(bias + sum_{ci, k} w[4g + co, ci, k] x[b, 4g + ci, 4j + k - 20]).
// LDS: line (row, ci) = x[-20 .. 4n + 24) of that row, WL = 4n + 44 floats.
__global__ __launch_bounds__(256) void k_g4_fwd(G4Parts q, int C, const float* __restrict__ w, const float* __restrict__ bias,
                                               int act, float slope) {
    const G4Part p = pick_part(q, blockIdx.x);
    const int g = blockIdx.y, tid = threadIdx.x;
    const int n = p.n, L = p.L, total = p.B * n, e0 = p.wg * ITEMS;
    const int WL = 4 * n + 44, QW = n + 11;
    int r_lo, nrows;
    item_rows(e0, total, n, r_lo, nrows);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, 4u * (unsigned)(p.B * C * L), 0x00020000);
    float wv[NWV];
    load_group_weights<false>(w, g, wv);
    // every load of the staging pass is in flight before the first LDS store (a loop of load -> store per line paid the
    // memory latency once per line: 64 us for the layer); quad i of the flat index sits at float 4 i of the lines
    {
        const unsigned magic = ((1u << 20) + QW - 1) / QW;           // i / QW for i < 2^20 / QW
        const int nquads = nrows * CG * QW;
        f32x4 xq[NQX];
#pragma unroll
        for (int u = 0; u < NQX; ++u) {
            const int i = tid + 256 * u;
            const int line = (int)(((unsigned)i * magic) >> 20), qi = i - line * QW;
            const unsigned row_elems = (unsigned)(((r_lo + (line >> 2)) * C + CG * g + (line & 3)) * L);
            xq[u] = i < nquads ? load_row4(rs, row_elems, 4 * qi - PAD4, L) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < NQX; ++u) {
            const int i = tid + 256 * u;
            if (i < nquads) *reinterpret_cast<f32x4*>(g4_smem + 4 * i) = xq[u];
        }
    }
    __syncthreads();
    const bool valid = e0 + tid < total;
    const int e = valid ? e0 + tid : total - 1;
    const int row = e / n, j = e - row * n;
    // four accumulators in rotation (a dependent 4x4x1 MFMA costs two idle cycles), the bias in the first; the next channel's
    // window is read before this channel's 41 MFMAs
    f32x4 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int co = 0; co < CG; ++co) acc[a][co] = (a == 0 && bias) ? bias[CG * g + co] : 0.f;
    const float* xw = g4_smem + (row - r_lo) * CG * WL + 4 * j;
    f32x4 win[2][11];
    auto fetch = [&](int ci, f32x4 (&wd)[11]) {
#pragma unroll
        for (int i = 0; i < 11; ++i) wd[i] = *reinterpret_cast<const f32x4*>(xw + ci * WL + 4 * i);
    };
    fetch(0, win[0]);
    static_for<0, CG>([&](auto CI) {
        constexpr int ci = decltype(CI)::value, cur = ci & 1;
        if constexpr (ci + 1 < CG) fetch(ci + 1, win[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, K4>([&](auto K) {
            constexpr int k = decltype(K)::value, pair = ci * K4 + k;
            mfma_bc<pair & 15>(acc[k & 3], wv[pair >> 4], win[cur][k >> 2][k & 3]);
        });
        __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int co = 0; co < CG; ++co) acc[0][co] = (acc[0][co] + acc[1][co]) + (acc[2][co] + acc[3][co]);
    if (!valid) return;
#pragma unroll
    for (int co = 0; co < CG; ++co)
        p.o[((size_t)row * C + CG * g + co) * n + j] = ms_apply_act(acc[0][co], act, slope);
}

// ------------------------------------------------------------------------------------------------------ backward data
// gx[b, 4g + ci, 4j + r] = sum_{co, m} w[4g + co, ci, 4m + r] gp[b, 4g + co, j + 5 - m]  (+ gx_add),  gp = gy act'(y).
// LDS: line (row, co) = gp[-5 .. n + 5) of that row, PL = n + 11 floats.
__global__ __launch_bounds__(256) void k_g4_bwd_data(G4Parts q, int C, const float* __restrict__ w, int act, float slope) {
    const G4Part p = pick_part(q, blockIdx.x);
    const int g = blockIdx.y, tid = threadIdx.x;
    const int n = p.n, L = p.L, total = p.B * n, e0 = p.wg * ITEMS;
    const int PL = n + 11;
    int r_lo, nrows;
    item_rows(e0, total, n, r_lo, nrows);
    float wv[NWV];
    load_group_weights<true>(w, g, wv);
    {   // all loads first (see the forward kernel); sample i of the flat index = (line, position) = (i / n, i % n)
        const unsigned magic = ((1u << 20) + n - 1) / n;
        const int count = nrows * CG * n;
        float gv[NEB], yv[NEB];
#pragma unroll
        for (int u = 0; u < NEB; ++u) {
            const int i = tid + 256 * u;
            const int line = (int)(((unsigned)i * magic) >> 20), pp = i - line * n;
            const size_t at = ((size_t)(r_lo + (line >> 2)) * C + CG * g + (line & 3)) * n + pp;
            gv[u] = i < count ? p.a[at] : 0.f;
            yv[u] = (i < count && act != MS_ACT_NONE) ? p.b[at] : 1.f;
        }
        for (int i = tid; i < nrows * CG * 11; i += 256) {            // the 5 + 6 halo samples of every line
            const int line = i / 11, h = i - 11 * line;
            g4_smem[line * PL + (h < 5 ? h : n + h)] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < NEB; ++u) {
            const int i = tid + 256 * u;
            const int line = (int)(((unsigned)i * magic) >> 20), pp = i - line * n;
            if (i < count) g4_smem[line * PL + 5 + pp] = act != MS_ACT_NONE ? ms_act_grad(gv[u], yv[u], act, slope) : gv[u];
        }
    }
    __syncthreads();
    const bool valid = e0 + tid < total;
    const int e = valid ? e0 + tid : total - 1;
    const int row = e / n, j = e - row * n;
    f32x4 acc[4];                                                    // acc[phase r][ci]
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* gw_ = g4_smem + (row - r_lo) * CG * PL + j;
    float win[CG][11];                                               // gp[co][j - 5 .. j + 5]: all reads before the MFMAs
#pragma unroll
    for (int co = 0; co < CG; ++co)
#pragma unroll
        for (int i = 0; i < 11; ++i) win[co][i] = gw_[co * PL + i];
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, CG>([&](auto CO) {
        constexpr int co = decltype(CO)::value;
        static_for<0, K4>([&](auto K) {
            constexpr int k = decltype(K)::value, m = k >> 2, r = k & 3, pair = co * K4 + k;
            mfma_bc<pair & 15>(acc[r], wv[pair >> 4], win[co][10 - m]);
        });
    });
    if (!valid) return;
    const int t0 = 4 * j;
#pragma unroll
    for (int ci = 0; ci < CG; ++ci) {
        const size_t off = ((size_t)row * C + CG * g + ci) * L + t0;
        if (t0 + 3 < L) {
            f32x4 v = {acc[0][ci], acc[1][ci], acc[2][ci], acc[3][ci]};
            if (p.c) v += *reinterpret_cast<const f32x4u*>(p.c + off);
            *reinterpret_cast<f32x4u*>(p.o + off) = v;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (t0 + r < L) p.o[off + r] = acc[r][ci] + (p.c ? p.c[off + r] : 0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------------- weight gradient
// gw[4g + co, ci, k] = beta gw + sum_{parts, b, j} gp[b, 4g + co, j] x[b, 4g + ci, 4j + k - 20];  gb[4g + co] = beta gb + sum gp.
// One workgroup per group; chunk = ITEMS items of a part, a wave takes 64 of them.  LDS per buffer (two buffers):
//   x lines as in the forward pass (XS floats), gp[item][co] (4 ITEMS floats), window offset of the item (ITEMS ints).
constexpr int WBUF = XS + 4 * ITEMS + ITEMS;     // floats per buffer
// (An earlier vector-pipe form of the weight gradient -- lane = item, 164 accumulators per lane, one wave reduction per
//  accumulator at the end -- exposed a hazard worth recording: the compiler packed its scalar FMAs into
//  `v_pk_fma_f32 ... op_sel:[0,1,0]` (the HIGH half of a register pair broadcast to both lanes), and exactly those products
//  (taps 4i + 1, even output channels) differed from run to run whenever the kernel shared the chip with the backward-data
//  chain on another stream; alone, or built with -fno-slp-vectorize, or with the packed FMA written without operand selection,
//  it was bit-stable.  tests/test_gpu_dp.py caught it.)
// PMC on the first form: 13.8 k vector instructions per wave beside 3 k MFMAs -- address arithmetic, row-end masks and
// register shuffling of the staging pass, on one wave per SIMD.  What does not change from chunk to chunk of a part now lives
// in a per-thread PLAN (rebuilt three times per launch); a chunk costs one add per load.
constexpr unsigned XINV = 0x80000000u;           // "no load": beyond every tensor (< 2 GB), and + a row step it does not wrap

struct G4Plan {
    unsigned xoff[NQX];                          // byte offset of the thread's quads for a chunk that starts at row 0, or XINV
    int keep[NQX];                               // samples of the quad inside the row (row ends of lengths not a multiple of 4)
    int row, j;                                  // the thread's item of the part's next chunk
    int r_lo, rem;                               // first row of that chunk, and its first item's position in it (uniform)
    int dq, dr;                                  // ITEMS / n, ITEMS % n (uniform)
};

struct G4Stage {                                 // a chunk on its way to LDS, in registers: RAW loads -- nothing here may
    f32x4 xq[NQX];                               // depend on the loaded values before g4_stage_store, or the loads' latency
    f32x4 gy, ya;                                // is paid in front of the arithmetic instead of behind it
    int woff;
};

__device__ __forceinline__ void g4_plan(const G4Part& p, int C, int g, G4Plan& pl) {
    const int tid = threadIdx.x;
    const int n = p.n, L = p.L, QW = n + 11;
    const int rows_max = ITEMS / n + 2 < p.B ? ITEMS / n + 2 : p.B;
    const unsigned magic = ((1u << 20) + QW - 1) / QW;               // i / QW for i < 2^20 / QW
#pragma unroll
    for (int u = 0; u < NQX; ++u) {
        const int i = tid + 256 * u;
        const int line = (int)(((unsigned)i * magic) >> 20), qi = i - line * QW;
        const int t = 4 * qi - PAD4;                                 // a multiple of 4: in front of the row, or starting inside it
        const bool ok = line < rows_max * CG && t >= 0 && t < L;
        pl.xoff[u] = ok ? 4u * (unsigned)(((line >> 2) * C + CG * g + (line & 3)) * L + t) : XINV;
        pl.keep[u] = ok ? (L - t < 4 ? L - t : 4) : 0;
    }
    pl.row = tid / n; pl.j = tid - pl.row * n;
    pl.r_lo = 0; pl.rem = 0;
    pl.dq = ITEMS / n; pl.dr = ITEMS - pl.dq * n;
}

// issues the loads of the part's next chunk and moves the plan on by one chunk
__device__ __forceinline__ void g4_stage_load(const G4Part& p, int C, int g, int act, G4Plan& pl, G4Stage& st) {
    const int n = p.n, L = p.L, WL = 4 * n + 44;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, 4u * (unsigned)(p.B * C * L), 0x00020000);
    const unsigned rstep = 4u * (unsigned)(pl.r_lo * C * L);         // (rows beyond the chunk's own but inside the tensor are read too)
#pragma unroll
    for (int u = 0; u < NQX; ++u)
        st.xq[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, pl.xoff[u] + rstep, 0, 0));
    const bool valid = pl.row < p.B;
    st.woff = valid ? (pl.row - pl.r_lo) * CG * WL + 4 * pl.j : 0;
    const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, 4u * (unsigned)(p.B * C * n), 0x00020000);
    const auto ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(act != MS_ACT_NONE ? p.c : p.b), 0, 4u * (unsigned)(p.B * C * n), 0x00020000);
    const unsigned off0 = valid ? 4u * (unsigned)((pl.row * C + CG * g) * n + pl.j) : XINV;       // (an invalid item reads 0.0)
#pragma unroll
    for (int co = 0; co < CG; ++co) {
        st.gy[co] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, off0 + 4u * (unsigned)(co * n), 0, 0));
        st.ya[co] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, off0 + 4u * (unsigned)(co * n), 0, 0));
    }
    pl.row += pl.dq; pl.j += pl.dr;
    if (pl.j >= n) { pl.j -= n; ++pl.row; }
    pl.r_lo += pl.dq; pl.rem += pl.dr;
    if (pl.rem >= n) { pl.rem -= n; ++pl.r_lo; }
}

__device__ __forceinline__ void g4_stage_store(const G4Stage& st, const G4Plan& pl, bool ragged, int act, float slope, float* buf) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < NQX; ++u) {
        const int i = tid + 256 * u;                                 // quad qi of line l sits at l WL + 4 qi = 4 i
        f32x4 v = st.xq[u];
        if (ragged) {                                                // (uniform: rows whose length is not a multiple of 4)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = e < pl.keep[u] ? v[e] : 0.f;
        }
        if (i < XS / 4) *reinterpret_cast<f32x4*>(buf + 4 * i) = v;
    }
    f32x4 gp = st.gy;
    if (act != MS_ACT_NONE) {
#pragma unroll
        for (int co = 0; co < CG; ++co) gp[co] = ms_act_grad(st.gy[co], st.ya[co], act, slope);
    }
    *reinterpret_cast<f32x4*>(buf + XS + 4 * tid) = gp;
    reinterpret_cast<int*>(buf + XS + 4 * ITEMS)[tid] = 4 * st.woff;     // (in bytes)
}

__global__ __launch_bounds__(256) void k_g4_wgrad(G4Parts q, int C, int act, float slope, float beta, float* __restrict__ gw,
                                                 float* __restrict__ gb) {
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nch = q.wg0[q.count];
    // block 16 u + b of the three MFMAs of an item = window quad (ci, qq) = ((16 u + b) / 11, (16 u + b) % 11); blocks 44 .. 47 idle
    const int b = lane >> 2, jj = lane & 3;
    int bci[3], bqq[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int Bq = 16 * u + b;
        bci[u] = Bq < 44 ? Bq / 11 : 0;
        bqq[u] = Bq < 44 ? Bq - 11 * bci[u] : 0;
    }
    f32x4 acc[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float gsum = 0.f;                               // lane 4t + r: sum of gp[co = r] over the items this lane held in its A register
    // Chunks travel global -> registers -> LDS one chunk ahead of the arithmetic (two chunks ahead measured the same: the
    // kernel was bound by its vector instructions, not by the loads).
    G4Plan pl;
    G4Stage st;
    {
        const G4Part p0 = pick_part(q, 0);
        g4_plan(p0, C, g, pl);
        g4_stage_load(p0, C, g, act, pl, st);
        g4_stage_store(st, pl, (p0.L & 3) != 0, act, slope, g4_smem);
    }
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        const float* buf = g4_smem + (c & 1) * WBUF;
        const G4Part pn = pick_part(q, c + 1 < nch ? c + 1 : c);
        if (c + 1 < nch) {
            if (pn.wg == 0) g4_plan(pn, C, g, pl);                  // the next chunk opens a part
            g4_stage_load(pn, C, g, act, pl, st);
        }
        const int WL = 4 * pick_part(q, c).n + 44;
        unsigned off[3];                            // LDS byte address of this lane's window sample for an item at float 0
#pragma unroll
        for (int u = 0; u < 3; ++u) off[u] = (unsigned)(uintptr_t)buf + 4u * (unsigned)(bci[u] * WL + 4 * bqq[u] + jj);
        // 16 items at a time: their 48 window samples are read one group AHEAD of the 48 MFMAs that use them (left to itself the
        // compiler reused one register: read -> wait -> MFMA, an LDS latency per instruction, 184 us for the layer)
        float G[2], xv[2][16][3];
        auto fetch = [&](int t16, float& Gd, float (&xd)[16][3]) {
            const int it0 = 64 * wid + 16 * t16;
            Gd = buf[XS + 4 * it0 + lane];                          // lane 4t + r: gp[item it0 + t][co = r]
            const int W = reinterpret_cast<const int*>(buf + XS + 4 * ITEMS)[it0 + (lane & 15)];
            static_for<0, 16>([&](auto T) {
                constexpr int t = decltype(T)::value;
                const unsigned base = (unsigned)__builtin_amdgcn_readlane(W, t);
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    xd[t][u] = *reinterpret_cast<const float __attribute__((address_space(3)))*>(off[u] + base);
            });
        };
        fetch(0, G[0], xv[0]);
        static_for<0, ITEMS / 64>([&](auto T16) {
            constexpr int t16 = decltype(T16)::value, cur = t16 & 1;
            if constexpr (t16 + 1 < ITEMS / 64) fetch(t16 + 1, G[cur ^ 1], xv[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
            gsum += G[cur];
            static_for<0, 16>([&](auto T) {
                constexpr int t = decltype(T)::value;
#pragma unroll
                for (int u = 0; u < 3; ++u) mfma_bc<t>(acc[u], G[cur], xv[cur][t][u]);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        if (c + 1 < nch) g4_stage_store(st, pl, (pn.L & 3) != 0, act, slope, g4_smem + ((c + 1) & 1) * WBUF);
        __syncthreads();
    }
    // the four waves' sums, added in a fixed order
    float* red = g4_smem;                           // [wave][13][64]
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) gsum += __shfl_xor(gsum, o, 64);   // over the 16 items a register holds: lane r = co r
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wid * 13 + 4 * u + r) * 64 + lane] = acc[u][r];
    red[(wid * 13 + 12) * 64 + lane] = gsum;
    __syncthreads();
    const int r = wid;                              // wave r writes output channel r
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        float v = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) v += red[(w4 * 13 + 4 * u + r) * 64 + lane];
        const int Bq = 16 * u + b, k = 4 * bqq[u] + jj;
        if (Bq < 44 && k < K4) {
            float* dst = gw + ((size_t)(CG * g + r) * CG + bci[u]) * K4 + k;
            *dst = beta != 0.f ? beta * *dst + v : v;
        }
    }
    if (gb && lane == 0) {
        float v = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) v += red[(w4 * 13 + 12) * 64 + r];
        gb[CG * g + r] = beta != 0.f ? beta * gb[CG * g + r] + v : v;
    }
}

bool g4_enabled() {
    const char* sw = getenv("MSYNTH_G4");        // tuning / test switch (0: the matrix-pipe kernels of gconv_split.hip)
    return !(sw && atoi(sw) == 0);
}

bool g4_layer(const ConvP& c) {
    return c.groups >= 1 && c.Cg == CG && c.Og == CG && c.K == K4 && c.stride == 4 && c.pad == PAD4 && c.dil == 1 &&
           c.pad_mode == MS_PAD_ZERO && !c.in_act && (c.act == MS_ACT_NONE || c.act == MS_ACT_LRELU);
}

// fills the table; per_item_wgs: workgroups (chunks) of ITEMS items per part
bool g4_table(const ConvP& c, const ms_conv1d_parts* parts, G4Parts* q) {
    if (!parts || parts->count < 1 || parts->count > NP || !g4_layer(c) || !g4_enabled()) return false;
    q->count = parts->count;
    q->wg0[0] = 0;
    for (int i = 0; i < NP; ++i) {
        if (i < parts->count) {
            const int B = parts->B[i], L = parts->Lin[i], n = (L + 3) / 4;
            if (B < 1 || n < NMIN || n > NMAX) return false;
            if ((long long)B * c.Cin * L * 4 >= (1ll << 31)) return false;
            q->B[i] = B; q->L[i] = L; q->n[i] = n;
            q->wg0[i + 1] = q->wg0[i] + ms_ceil_div(B * n, ITEMS);
        } else {
            q->B[i] = q->L[i] = q->n[i] = 0;
            q->wg0[i + 1] = q->wg0[i];
        }
        q->a[i] = q->b[i] = q->c[i] = nullptr; q->o[i] = nullptr;
    }
    return true;
}

size_t g4_rows_lds(const G4Parts& q, int per_row_extra) {            // bytes: lines of the largest workgroup
    size_t m = 0;
    for (int i = 0; i < q.count; ++i) {
        const int rows = ITEMS / q.n[i] + 2 < q.B[i] ? ITEMS / q.n[i] + 2 : q.B[i];
        const size_t b = (size_t)rows * CG * (per_row_extra == 44 ? 4 * q.n[i] + 44 : q.n[i] + 11) * sizeof(float);
        if (b > m) m = b;
    }
    return m;
}

}  // namespace

bool msg4_parts_applicable(const ConvP& c, const ms_conv1d_parts* parts) {
    G4Parts q;
    return g4_table(c, parts, &q);
}

int msg4_parts_fwd(const ConvP& c, const ms_conv1d_parts* parts, const float* w, const float* bias, hipStream_t s) {
    G4Parts q;
    if (!g4_table(c, parts, &q)) return MS_ERR_UNSUPPORTED;
    if (!w) return MS_ERR_INVALID_ARG;
    for (int i = 0; i < q.count; ++i) {
        if (!parts->x[i] || !parts->y[i]) return MS_ERR_INVALID_ARG;
        q.a[i] = parts->x[i]; q.o[i] = parts->y[i];
    }
    ms_note_kernel(0, "k_g4_fwd");
    hipLaunchKernelGGL(k_g4_fwd, dim3(q.wg0[q.count], c.groups), dim3(256), g4_rows_lds(q, 44), s, q, c.Cin, w, bias, c.act, c.slope);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int msg4_parts_bwd_data(const ConvP& c, const ms_conv1d_parts* parts, const float* w, hipStream_t s) {
    G4Parts q;
    if (!g4_table(c, parts, &q)) return MS_ERR_UNSUPPORTED;
    if (!w) return MS_ERR_INVALID_ARG;
    for (int i = 0; i < q.count; ++i) {
        if (!parts->gy[i] || !parts->gx[i] || (c.act != MS_ACT_NONE && !parts->y_act[i])) return MS_ERR_INVALID_ARG;
        q.a[i] = parts->gy[i]; q.b[i] = c.act != MS_ACT_NONE ? parts->y_act[i] : nullptr; q.c[i] = parts->gx_add[i];
        q.o[i] = parts->gx[i];
    }
    ms_note_kernel(0, "k_g4_bwd_data");
    hipLaunchKernelGGL(k_g4_bwd_data, dim3(q.wg0[q.count], c.groups), dim3(256), g4_rows_lds(q, 11), s, q, c.Cin, w, c.act, c.slope);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int msg4_parts_bwd_weight(const ConvP& c, const ms_conv1d_parts* parts, float* gw, float* gb, float beta, hipStream_t s) {
    G4Parts q;
    if (!g4_table(c, parts, &q)) return MS_ERR_UNSUPPORTED;
    if (!gw) return MS_ERR_INVALID_ARG;
    for (int i = 0; i < q.count; ++i) {
        if (!parts->x[i] || !parts->gy[i] || (c.act != MS_ACT_NONE && !parts->y_act[i])) return MS_ERR_INVALID_ARG;
        q.a[i] = parts->x[i]; q.b[i] = parts->gy[i]; q.c[i] = c.act != MS_ACT_NONE ? parts->y_act[i] : nullptr;
    }
    const size_t lds = (size_t)2 * WBUF * sizeof(float);
    static unsigned long long attr_set = 0;
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_g4_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ms_done_on_device(attr_set);
    }
    ms_note_kernel(0, "k_g4_wgrad");
    hipLaunchKernelGGL(k_g4_wgrad, dim3(c.groups), dim3(256), lds, s, q, c.Cin, c.act, c.slope, beta, gw, gb);
    MS_CHECK_LAUNCH();
    return MS_OK;
}
