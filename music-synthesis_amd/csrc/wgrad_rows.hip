// Weight gradient of the dense stride-1 "same" convolutions on the fp32 matrix cores, row-tile form.
//
//   gw[m, c, j] = sum_{b,t} G'[b, m, t] * X'[b, c, t + j*dil - pad]      gb[m] = sum_{b,t} G'[b, m, t]
//   G' = gy * act'(y)   (LeakyReLU / tanh derivative from the saved output),  X' = x or LeakyReLU(x)
//
// GEMM view: M = output channels, N = (c, j), contraction over (b, t).  The im2col form
// (k_igemm_wgrad*, conv_mfma.hip) loads the X operand once per TAP from L2 and is bound by L2
// bandwidth (16 FLOP per byte moved into LDS).  Here a workgroup owns 64*TM output channels x 64
// input channels x ALL K taps: per chunk of <= 64 time steps it stages the gradient rows and the
// input rows (with their (K-1)*dil halo) ONCE, and the B fragment of tap j is the same LDS row read
// at a +j*dil column offset -- 2.4x fewer bytes per FLOP.  Short rows (the discriminator's
// 1024 -> 1024 k5 conv at L = 32 / 17 / 9) pack R batch rows per chunk, each in its own segment of
// SS = L + halo columns; the gradient tile is zero in the halo columns, so the contraction simply
// runs over the padded columns.
//
// Split-K over chunks (grid.z) into per-slice slabs, summed in slice order by k_wgrad_reduce.
#include "ms_common.h"
#include "conv_mfma.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CB = 64;          // input channels per workgroup
constexpr int KMAX = 64;        // contraction columns per chunk

struct WrP {
    int B, CK, L, M, dil, pad, g_kind, x_kind;
    int Lt, R, SS, RSZ, kcols, tiles_per_row, nchunks, cps;
    int PG, PX;
    float slope;
};

// Several weight gradients of identical geometry (the six k3 convs of a ResidualStack: same channels and
// length, dilations 1/3/9) in ONE launch: grid.y = problem x M-tile.  A single layer leaves each workgroup
// only 4-8 chunks of contraction (prologue, slab write and the reduce launch cost as much as the MFMAs);
// batched, the same 256/512 workgroup slots get 6x longer K loops and 6x fewer slab bytes.
constexpr int WR_MULTI_MAX = 8;
struct WrMulti {
    int n, tiles_m;
    size_t slab_stride;                 // floats between the problems' regions inside one slice
    const float* x[WR_MULTI_MAX];
    const float* g[WR_MULTI_MAX];
    const float* gact[WR_MULTI_MAX];
    int dil[WR_MULTI_MAX], pad[WR_MULTI_MAX];
    const float* xmax[WR_MULTI_MAX];    // per problem: MS_ATOM_AMAX_N bounds of |x| / |g| (nullptr: none) -- k_wgrad_rows3<., 2>
    int signs;                          // gact[] hold SIGN WORDS of the activations (atom_fused.hip, MASK), not the fp32 tensors
    const float* gmax[WR_MULTI_MAX];
};

// Activation handling is a template parameter: a runtime `kind` compiles to scalar branches around
// every store piece, which cuts the chunk into hundreds of basic blocks and defeats the MFMA /
// store / load interleaving.  AK 1: LeakyReLU derivative on the gradient (the hot layers);
// AK 2: LeakyReLU applied to the input (pre-activation blocks); AK 0: runtime kinds (generic);
// AK 4: both of those AND reflection padding (the dilated conv of a weight-normed ResnetBlock): an
// aligned input vector that lies wholly in the padding is loaded from its mirror image (4 consecutive
// samples, generally not 16-byte aligned) and stored in reverse order.
template <int AK>
__device__ __forceinline__ float wr_gact(float g, float ya, int kind, float slope) {
    if (AK == 1 || AK == 4) return ya > 0.f ? g : g * slope;
    if (AK == 2 || AK == 3) return g;
    if (AK == 5) return g > 0.f ? g : g * slope;   // transposed conv with LeakyReLU in front: G is its INPUT
    return ms_act_grad(g, ya, kind, slope);
}
template <int AK>
__device__ __forceinline__ float wr_xact(float v, int kind, float slope) {
    if (AK == 1 || AK == 3 || AK == 5) return v;
    if (AK == 2 || AK == 4) return v > 0.f ? v : v * slope;
    return kind == MS_MOD_LRELU_FWD ? (v > 0.f ? v : v * slope) : v;
}

// Software pipeline (one workgroup per CU, one wave per SIMD, so nothing else hides a stall):
//   chunk c multiplies out of LDS buffer c&1 in KPI fully unrolled double-steps; between the MFMAs
//   of double-step `it` the wave stores piece `it` of chunk c+1 (registers -> the other buffer) and
//   then issues the global load of the same piece of chunk c+2 into the registers just freed.
//   One barrier per chunk.  All pieces are branch-free (invalid lanes store to a scratch slot,
//   masked loads read element 0) so that the whole chunk is one basic block the scheduler can
//   interleave.
template <int K, int TM, bool VEC, int KPI, int AK, int XS = 1>
__global__ __launch_bounds__(256) void k_wgrad_rows(WrP p, const float* __restrict__ X_,
                                                   const float* __restrict__ Xact_,
                                                   const float* __restrict__ G_,
                                                   const float* __restrict__ Gact_,
                                                   float* __restrict__ partial, size_t pstride, WrMulti mp) {
    const float* __restrict__ X = X_;
    const float* __restrict__ Xact = Xact_;
    const float* __restrict__ G = G_;
    const float* __restrict__ Gact = Gact_;
    int by = blockIdx.y;
    if (mp.n > 0) {                                  // batched launch: this workgroup's problem
        const int prob = by / mp.tiles_m;
        by -= prob * mp.tiles_m;
        X = mp.x[prob]; G = mp.g[prob]; Gact = mp.gact[prob]; Xact = nullptr;
        p.dil = mp.dil[prob]; p.pad = mp.pad[prob];
        p.SS = p.Lt + (K - 1) * p.dil;
        p.RSZ = p.R * p.SS;
        partial += (size_t)prob * mp.slab_stride;
    }
    constexpr int BM = 2 * TM * 32;
    constexpr int NGQ = VEC ? TM * 4 : TM * 16;     // G pieces (one load per thread each)
    // XS > 1 (transposed-conv weight grads): the 64 X rows are the XS phases of 64/XS channels of a
    // stride-XS signal; a piece is a 16-byte vector of the phase-interleaved span of one channel and its
    // 4 elements scatter to (phase row, column); the activation derivative sits on this operand (AK 3)
    constexpr int NXQ = XS > 1 ? 5 : (VEC ? 8 : 32); // X pieces
    constexpr int NP = NGQ + NXQ;
    constexpr int PPI = (NP + KPI - 1) / KPI;       // pieces per double-step
    extern __shared__ float smem[];
    const int tile_floats = BM * p.PG + CB * p.PX;
    float* scratch = smem + 2 * tile_floats;        // 256 floats: sink for out-of-tile lanes
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = by * BM, c0 = blockIdx.x * CB;
    const int NG = p.CK * K;
    const float* Gq = Gact ? Gact : G;
    const int g_kind = Gact ? p.g_kind : MS_ACT_NONE;

    // XS > 1: a transposed conv with K = 2*XS, pad = XS/2 has exactly two live taps per phase (d in {0,+1}
    // for the low phases, {-1,0} for the high ones).  The LDS rows are ordered [low phases | high phases]
    // so a wave's 32 rows share their tap pair, and only those two taps are multiplied and written.
    constexpr int KT = XS > 1 ? 2 : K;
    constexpr int XH = XS > 1 ? XS / 2 : 1;       // phases per half
    const int jlo = XS > 1 ? 1 - wn : 0;
    f32x16 acc[TM][KT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float asum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) asum[i] = 0.f;


    // ---- chunk-invariant loader descriptors
    // VEC: G thread = (vector v = tid&15 of the R*Lt/4 per row, rows (tid>>4) + 16q);
    //      X thread = (aligned vector v = tid&31 of R*NVS per row, rows (tid>>5) + 8q)
    // scalar: G thread = (column tid&63, rows (tid>>6) + 4q); X thread = (column tid&127, rows (tid>>7) + 2q)
    const int sh = (4 - (p.pad & 3)) & 3;           // row start of the X window within its 16-byte vector
    int g_seg, g_t, x_seg, x_u;
    bool g_cv, x_cv;
    if (VEC) {
        const int LV = p.Lt >> 2, v = tid & 15;
        g_seg = v / LV; g_t = 4 * (v - g_seg * LV); g_cv = g_seg < p.R;
        const int NVS = (p.SS + 6) >> 2, w = tid & 31;
        x_seg = w / NVS; x_u = 4 * (w - x_seg * NVS) - sh; x_cv = x_seg < p.R;
    } else {
        const int k = tid & 63;
        g_seg = k / p.SS; g_t = k - g_seg * p.SS; g_cv = g_seg < p.R && g_t < p.Lt;
        const int c = tid & 127;
        x_seg = c / p.SS; x_u = c - x_seg * p.SS; x_cv = x_seg < p.R && x_u < p.SS;
    }
    constexpr int g_rstep = VEC ? 16 : 4, x_rstep = VEC ? 8 : 2;
    const int g_row0 = VEC ? tid >> 4 : tid >> 6;
    const int x_row0 = VEC ? tid >> 5 : tid >> 7;
    // element offsets (fit in 31 bits, checked on the host) relative to the chunk origin
    const int g_off0 = (g_seg * p.M + m0 + g_row0) * p.L + g_t;
    const int x_off0 = (x_seg * p.CK + c0 + x_row0) * p.L + x_u - p.pad;
    // LDS offsets inside a buffer; lanes outside the tile write to their scratch slot instead
    const int g_lds0 = g_row0 * p.PG + g_seg * p.SS + g_t;
    const int x_lds0 = BM * p.PG + x_row0 * p.PX + x_seg * p.SS + x_u;
    bool x_ev[4];                                    // VEC: which of the 4 elements fall inside the segment
    unsigned x_evm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x_ev[i] = x_cv && x_u + i >= 0 && x_u + i < p.SS;
        x_evm |= x_ev[i] ? 1u << i : 0u;
    }

    int t_off[XS > 1 ? NXQ : 1], t_pos[XS > 1 ? NXQ : 1], t_lds[XS > 1 ? NXQ : 1][4];
    bool t_in[XS > 1 ? NXQ : 1];
    if (XS > 1) {
        const int span = p.SS * XS;                  // phase-interleaved floats of one channel and chunk
        const int NVco = (span + 6) >> 2;
        const int shx = (4 - ((p.pad * XS) & 3)) & 3;
#pragma unroll
        for (int q = 0; q < NXQ; ++q) {
            const int idx = tid + 256 * q;
            const int corow = idx / NVco, v = idx - corow * NVco;
            t_in[q] = corow < CB / XS && c0 / XS + corow < p.CK / XS;
            t_pos[q] = 4 * v - shx - p.pad * XS;     // position of the vector relative to t0*XS (multiple of 4)
            t_off[q] = t_in[q] ? (c0 / XS + corow) * p.L * XS + t_pos[q] : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pos = 4 * v + e - shx;
                const int r = pos % XS, u = pos / XS;
                const int xrow = (r / XH) * 32 + corow * XH + r % XH;
                t_lds[q][e] = (t_in[q] && pos >= 0 && pos < span) ? BM * p.PG + xrow * p.PX + u : -1;
            }
        }
    }

    float4 gv4[VEC ? NGQ : 1], ga4[VEC ? NGQ : 1], xv4[VEC ? NXQ : 1], xa4[(VEC && XS > 1) ? NXQ : 1];
    float gv1[VEC ? 1 : NGQ], ga1[VEC ? 1 : NGQ], xv1[VEC ? 1 : NXQ];

    // per-chunk scalars of the chunk whose pieces are being loaded / stored
    struct Cs { int gbase, xbase, t0x, xdelta; bool gok, xok, xrev; };
    auto chunk_state = [&](int ch, int c_end) {
        Cs s;
        int b0, t0;
        const int bq = ch / p.tiles_per_row;              // (R > 1: tiles_per_row == 1)
        b0 = bq * p.R;
        t0 = (ch - bq * p.tiles_per_row) * p.Lt;
        const bool live = ch < c_end;
        s.gbase = b0 * p.M * p.L + t0;
        s.xbase = b0 * p.CK * p.L + t0 * XS;
        s.t0x = t0 * XS;
        const int tg = t0 + g_t, tx = t0 - p.pad + x_u;       // VEC: tx % 4 == 0, vector all in or all out
        s.gok = live && g_cv && b0 + g_seg < p.B && tg < p.L;
        s.xok = XS > 1 ? (live && b0 < p.B) : (live && x_cv && b0 + x_seg < p.B && tx >= 0 && tx < p.L);
        s.xdelta = 0; s.xrev = false;
        if (AK == 4) {                                    // mirror image of a vector in the padding
            int src = tx;
            if (tx < 0) { src = -tx - 3; s.xrev = true; }
            else if (tx >= p.L) { src = 2 * p.L - 5 - tx; s.xrev = true; }
            s.xdelta = src - tx;
            s.xok = live && x_cv && b0 + x_seg < p.B && src >= 0 && src + 3 < p.L;
        }
        return s;
    };
    auto load_piece = [&](int pi, const Cs& s) {
        if (pi < NGQ) {
            const int q = pi;
            const bool ok = s.gok && m0 + g_row0 + g_rstep * q < p.M;
            const int o = ok ? s.gbase + g_off0 + q * g_rstep * p.L : 0;
            if (VEC) {
                gv4[q] = *reinterpret_cast<const float4*>(G + o);
                if (AK != 2 && AK != 3 && AK != 5) ga4[q] = *reinterpret_cast<const float4*>(Gq + o);
            } else {
                gv1[q] = G[o];
                if (AK != 2 && AK != 3 && AK != 5) ga1[q] = Gq[o];
            }
        } else {
            const int q = pi - NGQ;
            if (XS > 1) {
                const int gp = s.t0x + t_pos[q];         // all 4 samples inside [0, L*XS) or none
                const bool ok = s.xok && t_in[q] && gp >= 0 && gp < p.L * XS;
                const int o = ok ? s.xbase + t_off[q] : 0;
                xv4[q] = *reinterpret_cast<const float4*>(X + o);
                if (AK == 3) xa4[q] = *reinterpret_cast<const float4*>(Xact + o);
            } else {
                const bool ok = s.xok && c0 + x_row0 + x_rstep * q < p.CK;
                const int o = ok ? s.xbase + x_off0 + q * x_rstep * p.L + (AK == 4 ? s.xdelta : 0) : 0;
                if (VEC) xv4[q] = *reinterpret_cast<const float4*>(X + o);
                else xv1[q] = X[o];
            }
        }
    };
    auto store_piece = [&](int pi, const Cs& s, float* buf) {
        if (pi < NGQ) {
            const int q = pi;
            const bool ok = s.gok && m0 + g_row0 + g_rstep * q < p.M;
            float* d = g_cv ? buf + g_lds0 + q * g_rstep * p.PG : scratch + tid;
            if (VEC) {
                const float v0 = ok ? wr_gact<AK>(gv4[q].x, ga4[q].x, g_kind, p.slope) : 0.f;
                const float v1 = ok ? wr_gact<AK>(gv4[q].y, ga4[q].y, g_kind, p.slope) : 0.f;
                const float v2 = ok ? wr_gact<AK>(gv4[q].z, ga4[q].z, g_kind, p.slope) : 0.f;
                const float v3 = ok ? wr_gact<AK>(gv4[q].w, ga4[q].w, g_kind, p.slope) : 0.f;
                d[0] = v0; d[g_cv ? 1 : 0] = v1; d[g_cv ? 2 : 0] = v2; d[g_cv ? 3 : 0] = v3;
            } else {
                d[0] = ok ? wr_gact<AK>(gv1[q], ga1[q], g_kind, p.slope) : 0.f;
            }
        } else {
            const int q = pi - NGQ;
            if (XS > 1) {
                const int gp = s.t0x + t_pos[q];
                const bool ok = s.xok && t_in[q] && gp >= 0 && gp < p.L * XS;
                float e[4] = {xv4[q].x, xv4[q].y, xv4[q].z, xv4[q].w};
                if (AK == 3) {
                    const float a[4] = {xa4[q].x, xa4[q].y, xa4[q].z, xa4[q].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) e[i] = a[i] > 0.f ? e[i] : e[i] * p.slope;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float* di = t_lds[q][i] >= 0 ? buf + t_lds[q][i] : scratch + tid;
                    *di = ok ? e[i] : 0.f;
                }
                return;
            }
            const bool ok = s.xok && c0 + x_row0 + x_rstep * q < p.CK;
            float* d = buf + x_lds0 + q * x_rstep * p.PX;
            if (VEC) {
                const float e[4] = {xv4[q].x, xv4[q].y, xv4[q].z, xv4[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float* di;
                    if (AK == 4) {
                        const int j = s.xrev ? 3 - i : i;
                        di = (x_evm >> j) & 1u ? d + j : scratch + tid;
                    } else {
                        di = x_ev[i] ? d + i : scratch + tid;
                    }
                    *di = ok ? wr_xact<AK>(e[i], p.x_kind, p.slope) : 0.f;
                }
            } else {
                float* di = x_cv ? d : scratch + tid;
                *di = ok ? wr_xact<AK>(xv1[q], p.x_kind, p.slope) : 0.f;
            }
        }
    };

    const int c_begin = blockIdx.z * p.cps;
    const int c_end = min(c_begin + p.cps, p.nchunks);
    // prologue: chunk c_begin -> buffer 0, chunk c_begin + 1 -> registers
    Cs s_cur = chunk_state(c_begin, c_end);
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) load_piece(pi, s_cur);
    // the halo columns of the gradient tiles (and the columns past the last segment) stay zero
    for (int i = tid; i < 2 * tile_floats + 256; i += 256) smem[i] = 0.f;
    __syncthreads();                                 // zero fill complete
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) store_piece(pi, s_cur, smem);
    Cs s_next = chunk_state(c_begin + 1, c_end);
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) load_piece(pi, s_next);
    __syncthreads();

    const int a_off = (wm * TM * 32 + (lane & 31)) * p.PG + h;
    const int b_off = BM * p.PG + (wn * 32 + (lane & 31)) * p.PX + h;
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int cur = (ch - c_begin) & 1;
        const float* ap = smem + cur * tile_floats + a_off;
        const float* bp = smem + cur * tile_floats + b_off;
        float* nbuf = smem + (cur ^ 1) * tile_floats;
        const Cs s_after = chunk_state(ch + 2, c_end);
        float a0[TM], b0[KT], a1[TM], b1[KT];
        auto frag = [&](int kk, float (&a)[TM], float (&b)[KT]) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = ap[i * 32 * p.PG + 2 * kk];
#pragma unroll
            for (int j = 0; j < KT; ++j) b[j] = bp[2 * kk + (jlo + j) * p.dil];
        };
        auto mma = [&](const float (&a)[TM], const float (&b)[KT]) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                asum[i] += a[i];
#pragma unroll
                for (int j = 0; j < KT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        };
        const bool wave_live = m0 + wm * TM * 32 < p.M;     // rows of this wave exist (M = 32 under a 64-row tile)
        frag(0, a0, b0);
#pragma unroll
        for (int it = 0; it < KPI; ++it) {
            frag(2 * it + 1, a1, b1);
            if (wave_live) mma(a0, b0);
#pragma unroll
            for (int pp = 0; pp < PPI; ++pp) {
                const int pi = it * PPI + pp;
                if (pi < NP) {
                    store_piece(pi, s_next, nbuf);      // chunk ch+1: registers -> the other buffer
                    load_piece(pi, s_after);            // chunk ch+2: into the registers just freed
                }
            }
            frag(2 * it + 2, a0, b0);                   // (past the last step: zeroed pad columns, unused)
            if (wave_live) mma(a1, b1);
        }
        s_next = s_after;
        __syncthreads();
    }

    // D[row][col]: col = lane&31 (input channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (output channel)
    float* part = partial + (size_t)blockIdx.z * pstride;
    int c = c0 + wn * 32 + (lane & 31);
    if (XS > 1) {       // undo the [low | high] phase ordering of the LDS rows
        const int l = lane & 31;
        c = c0 + (l / XH) * XS + wn * XH + l % XH;
    }
    if (c < p.CK) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.M) {
#pragma unroll
                    for (int j = 0; j < KT; ++j) part[(size_t)m * NG + (size_t)c * K + jlo + j] = acc[i][j][r];
                }
            }
        }
    }
    // bias grad = row sums of the gradient tile: lanes (m, h=0) and (m, h=1) each saw half of the columns
    if (blockIdx.x == 0 && wn == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float s = asum[i] + __shfl_xor(asum[i], 32, 64);
            const int m = m0 + wm * TM * 32 + i * 32 + (lane & 31);
            if (lane < 32 && m < p.M) part[(size_t)p.M * NG + m] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same weight gradient on the bf16 matrix pipe with fp32-exact operands (see conv_rows3.hip for the
// scheme: every fp32 value = the exact sum of three bf16 pieces, six partial products per multiply, fp32
// accumulate) for the generator's k3 atom convs: K = 3, pad = dil in {1, 3, 9}, 16-byte aligned rows of at
// least 64 samples, LeakyReLU derivative on the gradient operand, 64 | Cin, 64*TM | Cout.
//
// The contraction runs over TIME, so an MFMA k-step is 16 consecutive samples: both operands keep their
// natural row-major layout in LDS ([row][piece][time] bf16, no transposes).  The input rows are staged ONCE
// with their halo; the B fragment of tap j starts j*dil - pad samples off the 16-byte grid, so it is read as
// the two aligned 16-byte vectors around it and funnelled together in registers (v_alignbit for odd shifts;
// the shift is a compile-time constant per dilation and tap).
//
// 8 waves = two groups of four.  Both groups own the SAME output tile and split the contraction: of every
// 64-sample chunk of the plan, group g takes samples [32g, 32g + 32).  The groups alternate between the
// matrix pipe and the vector work of the staging slot by slot (as k_conv_rows3p), each with its own
// single-buffered LDS tiles; at the end group 1's accumulators are added to group 0's through LDS and
// the tile goes to the split-K slab in the layout of k_wgrad_rows (same reduce kernels).
typedef __bf16 w3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 w3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float w3_f32x2 __attribute__((ext_vector_type(2)));
typedef float w3_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void w3_split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    const w3_f32x2 v = {a, b};
    const w3_bf16x2 hi = __builtin_convertvector(v, w3_bf16x2);
    const w3_f32x2 r1 = v - __builtin_convertvector(hi, w3_f32x2);
    const w3_bf16x2 mi = __builtin_convertvector(r1, w3_bf16x2);
    const w3_f32x2 r2 = r1 - __builtin_convertvector(mi, w3_f32x2);
    const w3_bf16x2 lo = __builtin_convertvector(r2, w3_bf16x2);
    h = __builtin_bit_cast(unsigned, hi);
    m = __builtin_bit_cast(unsigned, mi);
    l = __builtin_bit_cast(unsigned, lo);
}

// 4 consecutive samples -> one 8-byte group per piece
__device__ __forceinline__ void w3_split_quad(const float (&e)[4], uint2 (&o)[3]) {
    unsigned h0, m0, l0, h1, m1, l1;
    w3_split_pair(e[0], e[1], h0, m0, l0);
    w3_split_pair(e[2], e[3], h1, m1, l1);
    o[0] = make_uint2(h0, h1);
    o[1] = make_uint2(m0, m1);
    o[2] = make_uint2(l0, l1);
}

// the 8 16-bit elements that start E elements into the 16 elements of (lo, hi)
template <int E>
__device__ __forceinline__ uint4 w3_funnel(const uint4& lo, const uint4& hi) {
    const unsigned d[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint4 o;
    if (E % 2 == 0) {
        o = make_uint4(d[E / 2], d[E / 2 + 1], d[E / 2 + 2], d[E / 2 + 3]);
    } else {
        o.x = __builtin_amdgcn_alignbit(d[(E + 1) / 2], d[(E - 1) / 2], 16);
        o.y = __builtin_amdgcn_alignbit(d[(E + 1) / 2 + 1], d[(E - 1) / 2 + 1], 16);
        o.z = __builtin_amdgcn_alignbit(d[(E + 1) / 2 + 2], d[(E - 1) / 2 + 2], 16);
        o.w = __builtin_amdgcn_alignbit(d[(E + 1) / 2 + 3], d[(E - 1) / 2 + 3], 16);
    }
    return o;
}

// r04: the same kernel on block-scaled two-piece fp16 operands (NP = 2, three products per multiply into one fp32
// accumulator: atom_fused.hip) when the caller hands in upper bounds of both tensors' magnitudes (the fused atom kernels
// publish them): one power-of-two scale per tensor puts its largest magnitude at 2^14, the slab is written unscaled.
typedef _Float16 w3_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 w3_f16x8 __attribute__((ext_vector_type(8)));

template <int NP>
__device__ __forceinline__ void w3_split_quad_np(const float (&e)[4], uint2 (&o)[NP]) {
    if constexpr (NP == 3) {
        w3_split_quad(e, o);
    } else {
        const w3_f32x2 v0 = {e[0], e[1]}, v1 = {e[2], e[3]};
        const w3_f16x2 h0 = __builtin_convertvector(v0, w3_f16x2), h1 = __builtin_convertvector(v1, w3_f16x2);
        const w3_f16x2 l0 = __builtin_convertvector(v0 - __builtin_convertvector(h0, w3_f32x2), w3_f16x2);
        const w3_f16x2 l1 = __builtin_convertvector(v1 - __builtin_convertvector(h1, w3_f32x2), w3_f16x2);
        o[0] = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
        o[1] = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
    }
}

// S = 2^k with m S in [2^14, 2^15) and 1 / S; 1 for a zero / denormal-range / non-finite bound (atom_fused.hip)
__device__ __forceinline__ void w3_block_scale(float m, float& S, float& invS) {
    const unsigned eb = (__builtin_bit_cast(unsigned, m) >> 23) & 0xFFu;
    const bool ok = eb >= 16u && eb <= 250u;
    S = ok ? __builtin_bit_cast(float, (268u - eb) << 23) : 1.f;
    invS = ok ? __builtin_bit_cast(float, (eb - 14u) << 23) : 1.f;
}

template <int NP> constexpr int w3_grs() { return NP * 64 + 16; }      // bytes per gradient row: NP pieces x 32 samples x 2 + 16
constexpr int W3_XPB = 128;                // bytes per input-row piece: 64 samples (32 + halo + funnel over-read)
template <int NP> constexpr int w3_xrs() { return NP * W3_XPB + 16; }  // 400 / 272: odd multiples of 16 bytes
template <int NP>
constexpr size_t w3_lds_bytes(int BM) {
    const size_t kloop = (size_t)2 * (BM * w3_grs<NP>() + 64 * w3_xrs<NP>());
    const size_t merge = (size_t)4 * (BM / 64) * 3 * 16 * 64 * sizeof(float) + (size_t)2 * BM * sizeof(float);
    return kloop > merge ? kloop : merge;
}

// GM: the activation whose derivative multiplies the gradient arrives as sign words -- one 16-bit word per (batch row,
// 32-channel block, lane half, column), bit 15 - r = "positive" of channel 32 blk + (r & 3) + 8 (r >> 2) + 4 h
// (atom_fused.hip, MASK): a thread's four columns are ONE 8-byte load where the fp32 tensor cost 16 bytes per column quad x 4.
// SOLO (r05, the NP = 2 launches): ONE group of four waves per workgroup (256 threads) that walks both halves of every chunk
// itself -- stage, barrier, multiply, barrier -- into one set of accumulators (no merge): up to three such workgroups share a
// CU and drift apart, where the two groups of the eight-wave form (kept for NP = 3, whose accumulators need its registers)
// alternate in lockstep behind two barriers per slot.
template <int TM, int NP, bool GM = false, bool SOLO = false>
__global__ __launch_bounds__(SOLO ? 256 : 512, SOLO ? 3 : 2) void k_wgrad_rows3(WrP p, const float* __restrict__ X_,
                                                    const float* __restrict__ G_,
                                                    const float* __restrict__ Gact_,
                                                    float* __restrict__ partial, size_t pstride, WrMulti mp) {
    const float* __restrict__ X = X_;
    const float* __restrict__ G = G_;
    const float* __restrict__ Gact = Gact_;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), while the
    // tiles that read the same gradient / input rows at the same time -- all (input-channel block, row tile) pairs
    // of one (problem, slice) -- are neighbours in the plain order.  Re-number so that neighbours share an XCD:
    // hardware id h -> logical id (h % 8) * (T / 8) + h / 8 (a bijection when 8 | T).  Speed only, never
    // correctness: every workgroup is independent.
    unsigned bx = blockIdx.x, by_ = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned T = gridDim.x * gridDim.y * gridDim.z;
        if (T % 8 == 0) {
            const unsigned hid = bx + gridDim.x * (by_ + gridDim.y * bz);
            const unsigned lid = (hid % 8) * (T / 8) + hid / 8;
            bx = lid % gridDim.x;
            by_ = (lid / gridDim.x) % gridDim.y;
            bz = lid / (gridDim.x * gridDim.y);
        }
    }
    int by = (int)by_;
    const float* xmax = nullptr;
    const float* gmax = nullptr;
    if (mp.n > 0) {                                  // batched launch: this workgroup's problem
        const int prob = by / mp.tiles_m;
        by -= prob * mp.tiles_m;
        X = mp.x[prob]; G = mp.g[prob]; Gact = mp.gact[prob];
        xmax = mp.xmax[prob]; gmax = mp.gmax[prob];
        p.dil = mp.dil[prob]; p.pad = mp.pad[prob];
        partial += (size_t)prob * mp.slab_stride;
    }
    constexpr int K = 3, BM = 64 * TM, NGU = BM / 32;
    constexpr int W3_GRS = w3_grs<NP>(), W3_XRS = w3_xrs<NP>();
    constexpr bool SC = NP == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5;
    const int g = SOLO ? 0 : __builtin_amdgcn_readfirstlane(wid >> 2), gt = tid & 255, gw = wid & 3;
    const int wm = gw >> 1, wn = gw & 1;
    unsigned char* const Gs = smem3 + g * (BM * W3_GRS + 64 * W3_XRS);
    unsigned char* const Xs = Gs + BM * W3_GRS;
    const int m0 = by * BM, c0 = (int)bx * CB;
    const int NG = p.CK * K;
    const int PADA = (p.pad + 3) & ~3;               // halo in front of the row tile, rounded to whole vectors

    constexpr unsigned OOB = 0xF0000000u;
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G), 0, 0x80000000u, 0x00020000);
    const auto rsGa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Gact), 0, 0x80000000u, 0x00020000);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0x80000000u, 0x00020000);

    // NP = 2: one power-of-two scale per tensor from the published bounds (1024 entries each: two per thread, folded over the
    // eight waves through LDS; everyone ends up with the same two scalars)
    float Sx = 1.f, Sg = 1.f, kfin = 1.f;
    if (SC) {
        float mx = fmaxf(xmax[tid], xmax[tid + 512]), mg = fmaxf(gmax[tid], gmax[tid + 512]);
        if (SOLO) {
            mx = fmaxf(mx, fmaxf(xmax[tid + 256], xmax[tid + 768]));
            mg = fmaxf(mg, fmaxf(gmax[tid + 256], gmax[tid + 768]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            mg = fmaxf(mg, __shfl_xor(mg, o, 64));
        }
        float* red = reinterpret_cast<float*>(smem3);
        if (lane == 0) { red[wid] = mx; red[8 + wid] = mg; }
        __syncthreads();
        mx = red[0]; mg = red[8];
#pragma unroll
        for (int w = 1; w < (SOLO ? 4 : 8); ++w) { mx = fmaxf(mx, red[w]); mg = fmaxf(mg, red[8 + w]); }
        float ix, ig;
        w3_block_scale(mx, Sx, ix);
        w3_block_scale(mg, Sg, ig);
        kfin = ix * ig;
        __syncthreads();                             // (the staging buffers reuse this LDS)
    }

    // staging units of this thread: gradient (row (gt >> 3) + 32 q, vector gt & 7), input (row (gt >> 4) + 16 q,
    // vector gt & 15 of the 64-sample window that starts PADA samples in front of the tile)
    const int g_row = gt >> 3, g_t = 4 * (gt & 7);
    const int x_row = gt >> 4, x_u = 4 * (gt & 15);
    w3_f32x4 gv[NGU], ga[GM ? 1 : NGU], xv[4];
    uint2 gm[GM ? NGU : 1];                          // (GM) sign words of the vector's four columns
    // (GM) row g_row of every 32-row block: lane half (g_row >> 2) & 1, register (g_row & 3) + 4 (g_row >> 3)
    const int gm_h = (g_row >> 2) & 1, gm_sh = 15 - ((g_row & 3) + 4 * (g_row >> 3));
    float bs[NGU];
#pragma unroll
    for (int q = 0; q < NGU; ++q) bs[q] = 0.f;

    auto load_chunk = [&](int ch, int c_end, int half = 0) {
        const int b = ch / p.tiles_per_row;
        const int t0 = (ch - b * p.tiles_per_row) * p.Lt + 32 * (SOLO ? half : g);
        const bool live = ch < c_end && b < p.B;
        const int tg = t0 + g_t, tx = t0 - PADA + x_u;        // multiples of 4: a vector is all in or all out
        const unsigned go = (live && tg < p.L) ? 4u * (unsigned)((b * p.M + m0 + g_row) * p.L + tg) : OOB;
        const unsigned xo = (live && tx >= 0 && tx < p.L) ? 4u * (unsigned)((b * p.CK + c0 + x_row) * p.L + tx) : OOB;
        const unsigned mo = (live && tg < p.L) ? 2u * (unsigned)(((b * (p.M / 32) + m0 / 32) * 2 + gm_h) * p.L + tg) : OOB;
#pragma unroll
        for (int q = 0; q < NGU; ++q) {
            gv[q] = __builtin_bit_cast(w3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, go, q * 32 * 4 * p.L, 0));
            if constexpr (GM) gm[q] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rsGa, mo, q * 4 * p.L, 0));
            else ga[q] = __builtin_bit_cast(w3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsGa, go, q * 32 * 4 * p.L, 0));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            xv[q] = __builtin_bit_cast(w3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, xo, q * 16 * 4 * p.L, 0));
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < NGU; ++q) {
            float e[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bool pos;
                if constexpr (GM) {
                    const unsigned pair = (i >> 1) ? gm[q].y : gm[q].x;
                    pos = ((((i & 1) ? pair >> 16 : pair) >> gm_sh) & 1u) != 0;
                } else {
                    pos = ga[q][i] > 0.f;
                }
                e[i] = pos ? gv[q][i] : gv[q][i] * p.slope;
            }
            bs[q] += (e[0] + e[1]) + (e[2] + e[3]);
            if (SC) {
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] *= Sg;
            }
            uint2 o3[NP];
            w3_split_quad_np<NP>(e, o3);
            unsigned char* d = Gs + (g_row + 32 * q) * W3_GRS + g_t * 2;
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) *reinterpret_cast<uint2*>(d + pp * 64) = o3[pp];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float e[4] = {xv[q][0], xv[q][1], xv[q][2], xv[q][3]};
            if (SC) {
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] *= Sx;
            }
            uint2 o3[NP];
            w3_split_quad_np<NP>(e, o3);
            unsigned char* d = Xs + (x_row + 16 * q) * W3_XRS + x_u * 2;
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) *reinterpret_cast<uint2*>(d + pp * W3_XPB) = o3[pp];
        }
    };

    f32x16 acc[TM][K];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned char* const ap = Gs + (wm * TM * 32 + (lane & 31)) * W3_GRS + h * 16;
    const unsigned char* const bp = Xs + (wn * 32 + (lane & 31)) * W3_XRS + h * 16;
    auto compute = [&](auto dilc) __attribute__((always_inline)) {
        constexpr int DIL = decltype(dilc)::value;
        constexpr int PA_ = (DIL + 3) & ~3;
        constexpr int O0 = PA_ - DIL, O2 = PA_ + DIL;       // first sample of taps 0 / 2 in window coordinates (tap 1: PA_)
        constexpr int V0 = O0 / 8, NV = (O2 + 7) / 8 - V0 + 1;   // aligned 8-sample vectors the three taps touch
        constexpr int NPR = NP == 3 ? 6 : 3;
        constexpr int PAI[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0}, PBI[6] = {NP == 3 ? 2 : 1, 0, NP == 3 ? 1 : 0, 1, 0, 0};
        // fragments of k-step s: the A rows and, per piece, the NV aligned vectors that hold all three taps'
        // windows (2 for dilations 1 / 3, 4 for dilation 9) -- each tap is funnelled out of two neighbours
        uint4 a[2][TM][NP];
        uint4 v[2][NP][NV];
        auto frag = [&](int s, uint4 (&af)[TM][NP], uint4 (&vf)[NP][NV]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pp = 0; pp < NP; ++pp)
                    af[i][pp] = *reinterpret_cast<const uint4*>(ap + i * 32 * W3_GRS + pp * 64 + s * 32);
#pragma unroll
            for (int pp = 0; pp < NP; ++pp)
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    vf[pp][k] = *reinterpret_cast<const uint4*>(bp + pp * W3_XPB + (s * 16 + (V0 + k) * 8) * 2);
        };
        frag(0, a[0], v[0]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 0) frag(1, a[1], v[1]);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int o = j * DIL - DIL + PA_;
                const int k = o / 8 - V0;
                uint4 b[NP];
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    const uint4 lo = v[s][pp][k];
                    const uint4 hi = v[s][pp][k + 1 < NV ? k + 1 : k];
                    switch (o & 7) {
                        case 0: b[pp] = w3_funnel<0>(lo, hi); break;
                        case 1: b[pp] = w3_funnel<1>(lo, hi); break;
                        case 2: b[pp] = w3_funnel<2>(lo, hi); break;
                        case 3: b[pp] = w3_funnel<3>(lo, hi); break;
                        case 4: b[pp] = w3_funnel<4>(lo, hi); break;
                        case 5: b[pp] = w3_funnel<5>(lo, hi); break;
                        case 6: b[pp] = w3_funnel<6>(lo, hi); break;
                        default: b[pp] = w3_funnel<7>(lo, hi); break;
                    }
                }
                // (NP = 3: six products, smallest first; NP = 2: a_h b_l, a_l b_h, a_h b_h)
#pragma unroll
                for (int t = 0; t < NPR; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        if constexpr (NP == 3)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(w3_bf16x8, a[s][i][PAI[t]]),
                                                                               __builtin_bit_cast(w3_bf16x8, b[PBI[t]]), acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(w3_f16x8, a[s][i][PAI[t]]),
                                                                              __builtin_bit_cast(w3_f16x8, b[PBI[t]]), acc[i][j], 0, 0, 0);
                    }
            }
        }
    };
    const int c_begin = (int)bz * p.cps;
    const int c_end = min(c_begin + p.cps, p.nchunks);
    // Slot 2i: group 0 multiplies its half of chunk i while group 1 stages its half and at once fetches the next;
    // slot 2i + 1: the roles swap.  A fetch so has the rest of its staging slot plus the whole multiply slot to
    // land (one slot did not cover the HBM latency under load).  One straight-line loop per group (the group is
    // wave-uniform): the compiler then waits for a fetch where it is first used, not at a merged back edge.
    // The dilation is dispatched ONCE, around the whole loop: with the three compute bodies inside one loop the register
    // allocator gave each its own accumulator registers and moved all 48 of them through the accumulation file on every
    // iteration (48 v_accvgpr_read + 48 v_accvgpr_write per 18 MFMAs: a third of the kernel's vector instructions, r05).
    const int n = c_end - c_begin;
    auto main_loop = [&](auto dilc) __attribute__((always_inline)) {
        if constexpr (SOLO) {
            load_chunk(c_begin, c_end, 0);
            for (int u = 0; u < 2 * n; ++u) {
                stage();                                 // (waits for unit u's registers)
                load_chunk(c_begin + ((u + 1) >> 1), c_end, (u + 1) & 1);
                __syncthreads();                         // tiles staged
                compute(dilc);
                __syncthreads();                         // tiles free
            }
        } else {
            load_chunk(c_begin, c_end);
            if (g == 0) {
                stage();
                load_chunk(c_begin + 1, c_end);
                __syncthreads();
                for (int i = 0; i < n; ++i) {
                    compute(dilc);
                    __syncthreads();
                    stage();
                    load_chunk(c_begin + i + 2, c_end);
                    __syncthreads();
                }
            } else {
                __syncthreads();
                for (int i = 0; i < n; ++i) {
                    stage();
                    load_chunk(c_begin + i + 1, c_end);
                    __syncthreads();
                    compute(dilc);
                    __syncthreads();
                }
            }
        }
    };
    if (p.dil == 1) main_loop(std::integral_constant<int, 1>());
    else if (p.dil == 3) main_loop(std::integral_constant<int, 3>());
    else main_loop(std::integral_constant<int, 9>());

    // ---- merge the two groups and write the slab (layout of k_wgrad_rows)
    float* const mrg = reinterpret_cast<float*>(smem3);                 // [wave][tile][reg][lane]
    float* const bsum = SOLO ? mrg : mrg + 4 * TM * K * 16 * 64;        // [group][row] (SOLO: no merge area, one group)
#pragma unroll
    for (int q = 0; q < NGU; ++q) {     // row sums of the gradient tile: the 8 lanes that share a row
        float v = bs[q];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
        bs[q] = v;
    }
    if (!SOLO && g == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mrg[((gw * TM * K + i * K + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    if ((gt & 7) == 0) {
#pragma unroll
        for (int q = 0; q < NGU; ++q) bsum[g * BM + g_row + 32 * q] = bs[q];
    }
    __syncthreads();
    float* part = partial + (size_t)bz * pstride;
    if (g == 0) {
        const int c = c0 + wn * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float other = SOLO ? 0.f : mrg[((gw * TM * K + i * K + j) * 16 + r) * 64 + lane];
                    part[(size_t)m * NG + (size_t)c * K + j] = SC ? (acc[i][j][r] + other) * kfin : acc[i][j][r] + other;
                }
            }
    }
    if (SOLO ? (bx == 0 && gt < BM) : (g == 1 && bx == 0 && gt < BM)) {
        part[(size_t)p.M * NG + m0 + gt] = SOLO ? bsum[gt] : bsum[gt] + bsum[BM + gt];
    }
}

// Split-K slice count: fill the resident workgroup slots without a mostly empty extra round
// (160 tiles x 4 slices = 640 workgroups on 512 slots ran 63 % longer than 128 x 4).
int pick_slices(int tiles, int slots, int ns_max) {
    int ns = 1;
    double best = 0.0;
    if (ns_max > 2 * slots) ns_max = 2 * slots;
    for (int c = 1; c <= ns_max; ++c) {
        const int tot = tiles * c, rounds = ms_ceil_div(tot, slots);
        if (rounds > 2) break;
        const double eff = (double)tot / (rounds * (double)slots);
        if (eff > best + 1e-9) { best = eff; ns = c; }
    }
    return ns;
}

struct WrPlan {
    bool ok;
    WrP p;
    int tm, nsplit;
    bool vec, refl = false;
    dim3 grid;
    size_t lds, stride_floats;
    WrMulti mp = {};                    // n == 0: single problem
};

WrPlan plan_wrows(const ConvP& c) {
    WrPlan q;
    q.ok = false;
    const int K = c.K;
    if (c.groups != 1 || c.stride != 1 || c.Lout != c.Lin) return q;
    // reflection padding: only the pre-activation dilated k3 conv of the weight-normed ResnetBlock
    // (LeakyReLU in front and behind), on the 16-byte path with one row per chunk
    const bool refl = c.pad_mode == MS_PAD_REFLECT;
    if (refl && !(c.K == 3 && c.in_act && c.act == MS_ACT_LRELU && c.Lin % 4 == 0 && c.Lin >= KMAX && c.pad + 4 < c.Lin)) return q;
    if (!(K == 1 || K == 3 || K == 5 || K == 7)) return q;
    if (c.Cout < 64 || c.Cin < 32) return q;      // (32-channel layers: measured slower than the im2col kernel)
    const int H = (K - 1) * c.dil;
    if (H > 24 || c.pad > H) return q;
    if ((long long)c.B * c.Cout * c.Lin >= (1LL << 31) || (long long)c.B * c.Cin * c.Lin >= (1LL << 31)) return q;
    WrP& p = q.p;
    p.B = c.B; p.CK = c.Cin; p.L = c.Lin; p.M = c.Cout; p.dil = c.dil; p.pad = c.pad;
    p.g_kind = c.act; p.x_kind = c.in_act ? MS_MOD_LRELU_FWD : MS_ACT_NONE; p.slope = c.slope;
    const int LT = KMAX;
    if (p.L >= KMAX) { p.Lt = LT; p.R = 1; p.tiles_per_row = ms_ceil_div(p.L, LT); p.nchunks = p.B * p.tiles_per_row; }
    else {
        p.Lt = p.L;
        p.R = (KMAX + H) / (p.L + H);
        if (p.R < 1) p.R = 1;
        p.tiles_per_row = 1;
        p.nchunks = ms_ceil_div(p.B, p.R);
    }
    p.SS = p.Lt + H;
    p.RSZ = p.R * p.SS;
    p.kcols = p.RSZ - H <= 32 ? 32 : 64;            // 8 or 16 unrolled double-steps (pad columns are zero)
    q.vec = p.L % 4 == 0 && p.R * (p.Lt / 4) <= 16 && p.R * ((p.SS + 6) / 4) <= 32;
    q.refl = refl;
    if (refl && !q.vec) return q;
    if (!q.vec && (p.kcols > 64 || p.RSZ > 128)) return q;
    p.PG = (p.kcols + 2) | 1;
    p.PX = (p.kcols + H + 2) | 1;
    q.tm = (K <= 3 && c.Cout >= 128) ? 2 : 1;
    const int BM = 64 * q.tm;
    q.lds = (size_t)(2 * (BM * p.PG + CB * p.PX) + 256) * sizeof(float);   // two buffers + scratch
    if (q.lds > 150 * 1024) return q;
    const int tiles = ms_ceil_div(p.M, BM) * ms_ceil_div(p.CK, CB);
    q.stride_floats = (size_t)p.M * p.CK * K + p.M;
    // split-K slices: fill the resident workgroup slots (1 per CU above 80 KiB of LDS, else 2) without a
    // mostly empty extra round; >= 4 chunks per slice; slabs <= 48 MiB
    const int slots = 256 * (q.lds > 80 * 1024 ? 1 : 2);
    const int max_by_work = p.nchunks / 4 > 0 ? p.nchunks / 4 : 1;
    const size_t max_by_bytes = ((size_t)48 << 20) / (q.stride_floats * 4);
    int ns_max = max_by_work;
    if ((size_t)ns_max > max_by_bytes) ns_max = max_by_bytes > 0 ? (int)max_by_bytes : 1;
    const int ns = pick_slices(tiles, slots, ns_max);
    p.cps = ms_ceil_div(p.nchunks, ns);
    q.nsplit = ms_ceil_div(p.nchunks, p.cps);
    q.grid = dim3((unsigned)ms_ceil_div(p.CK, CB), (unsigned)ms_ceil_div(p.M, BM), (unsigned)q.nsplit);
    q.ok = true;
    return q;
}

template <int K, int TM, bool VEC, int KPI, int AK>
void launch_wrows_ak(const WrPlan& q, const float* x, const float* gy, const float* y_act,
                     float* partial, hipStream_t s) {
    static unsigned long long attr_set = 0;                    // > 64 KiB of dynamic LDS needs the opt-in once
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_rows<K, TM, VEC, KPI, AK>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        ms_done_on_device(attr_set);
    }
    hipLaunchKernelGGL((k_wgrad_rows<K, TM, VEC, KPI, AK>), q.grid, dim3(256), q.lds, s, q.p, x,
                       (const float*)nullptr, gy, y_act, partial, q.stride_floats, q.mp);
}

// the split-bf16 kernel takes: the k3 atom convs (pad = dil in {1, 3, 9}), aligned rows of >= 64 samples, whole tiles
bool wrows3_ok(const WrPlan& q, int K, int TM, bool vec, bool has_yact) {
    const char* sw = getenv("MSYNTH_WROWS3");        // tuning / test switch (0: fp32-MFMA kernel)
    if (sw && atoi(sw) == 0) return false;
    const WrP& p = q.p;
    if (K != 3 || !vec || q.refl || !has_yact || p.g_kind != MS_ACT_LRELU || p.x_kind != MS_ACT_NONE) return false;
    if (p.R != 1 || p.Lt != KMAX || p.L % 64 || p.CK % CB || p.M % 64) return false;
    if (q.mp.n > 0) {
        for (int i = 0; i < q.mp.n; ++i)
            if (q.mp.pad[i] != q.mp.dil[i] || !(q.mp.dil[i] == 1 || q.mp.dil[i] == 3 || q.mp.dil[i] == 9)) return false;
    } else if (p.pad != p.dil || !(p.dil == 1 || p.dil == 3 || p.dil == 9)) {
        return false;
    }
    return true;
}

// NP = 3: always 64 output channels per workgroup (TM = 1): with 128 the kernel needs more than the 256 registers a wave
// of a 512-thread workgroup may hold (96 accumulators + fragments + the chunk in flight) and spills.  The plan's grid /
// batching descriptor are re-derived for 64-row tiles; the slab layout does not depend on it.
// NP = 2 (every problem of a batched launch carries bounds of both operands): block-scaled fp16 x 2, three products.
template <int TM, int NP, bool GM = false, bool SOLO = false>
void launch_wrows3_np(const WrPlan& q, const float* x, const float* gy, const float* y_act, float* partial, hipStream_t s) {
    const size_t lds = SOLO ? (size_t)(64 * TM) * w3_grs<NP>() + 64 * w3_xrs<NP>() + 2 * 64 * TM * sizeof(float) : w3_lds_bytes<NP>(64 * TM);
    static unsigned long long attr_set = 0;
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_rows3<TM, NP, GM, SOLO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        ms_done_on_device(attr_set);
    }
    WrMulti mp = q.mp;
    const int tiles_m = q.p.M / (64 * TM);
    if (mp.n > 0) mp.tiles_m = tiles_m;
    const dim3 grid(q.grid.x, (unsigned)((mp.n > 0 ? mp.n : 1) * tiles_m), q.grid.z);
    ms_note_kernel(NP == 2 ? 3 : 6, "k_wgrad_rows3<%d, %d, %s, %s>", TM, NP, GM ? "true" : "false", SOLO ? "true" : "false");
    hipLaunchKernelGGL((k_wgrad_rows3<TM, NP, GM, SOLO>), grid, dim3(SOLO ? 256 : 512), lds, s, q.p, x, gy, y_act, partial, q.stride_floats, mp);
}

bool wrows3_scaled(const WrPlan& q) {
    static const int sw = getenv("MSYNTH_WROWS3_NP") ? atoi(getenv("MSYNTH_WROWS3_NP")) : 2;      // tuning / test switch (3: bf16 x 3)
    if (sw == 3 || q.mp.n <= 0) return false;
    for (int i = 0; i < q.mp.n; ++i)
        if (!q.mp.xmax[i] || !q.mp.gmax[i]) return false;
    return true;
}

void launch_wrows3(const WrPlan& q, const float* x, const float* gy, const float* y_act, float* partial, hipStream_t s) {
    // NP = 2 runs the four-wave (SOLO) form: 168 registers and 27 KB of LDS per workgroup instead of 252 / 53 KB -- alone it
    // takes the same time as the eight-wave form (C = 256: -8 %, 128 / 64: +3 %), but it shares the chip better with the
    // backward-data chain it runs beside: train step -1.3 % per call over four interleaved A/B pairs (r05)
    if (q.mp.n > 0 && q.mp.signs) {                  // (sign words only travel with the two-piece scheme: checked by the caller)
        launch_wrows3_np<1, 2, true, true>(q, x, gy, y_act, partial, s);
        return;
    }
    if (wrows3_scaled(q)) launch_wrows3_np<1, 2, false, true>(q, x, gy, y_act, partial, s);
    else launch_wrows3_np<1, 3>(q, x, gy, y_act, partial, s);
}

template <int K, int TM, bool VEC, int KPI>
void launch_wrows_inst(const WrPlan& q, const float* x, const float* gy, const float* y_act,
                       float* partial, hipStream_t s) {
    if (wrows3_ok(q, K, TM, VEC, y_act != nullptr)) {
        launch_wrows3(q, x, gy, y_act, partial, s);
        return;
    }
    const int gk = y_act ? q.p.g_kind : MS_ACT_NONE;
    if (q.refl) {
        if constexpr (VEC && K == 3) launch_wrows_ak<3, TM, true, KPI, 4>(q, x, gy, y_act, partial, s);
        return;
    }
    if (gk == MS_ACT_LRELU && q.p.x_kind == MS_ACT_NONE) launch_wrows_ak<K, TM, VEC, KPI, 1>(q, x, gy, y_act, partial, s);
    else if (gk == MS_ACT_NONE && q.p.x_kind == MS_MOD_LRELU_FWD) launch_wrows_ak<K, TM, VEC, KPI, 2>(q, x, gy, y_act, partial, s);
    else launch_wrows_ak<K, TM, VEC, KPI, 0>(q, x, gy, y_act, partial, s);
}

template <int K, int TM>
void launch_wrows_tm(const WrPlan& q, bool vec, const float* x, const float* gy, const float* y_act,
                     float* partial, hipStream_t s) {
    const bool k16 = q.p.kcols == 64;
    if (vec) {
        if (k16) launch_wrows_inst<K, TM, true, 16>(q, x, gy, y_act, partial, s);
        else launch_wrows_inst<K, TM, true, 8>(q, x, gy, y_act, partial, s);
    } else {
        if (k16) launch_wrows_inst<K, TM, false, 16>(q, x, gy, y_act, partial, s);
        else launch_wrows_inst<K, TM, false, 8>(q, x, gy, y_act, partial, s);
    }
}

template <int K>
void launch_wrows(const WrPlan& q, bool vec, const float* x, const float* gy, const float* y_act,
                  float* partial, hipStream_t s) {
    if (K <= 3 && q.tm == 2) launch_wrows_tm<K, K <= 3 ? 2 : 1>(q, vec, x, gy, y_act, partial, s);
    else launch_wrows_tm<K, 1>(q, vec, x, gy, y_act, partial, s);
}

}  // namespace

bool msw_bwd_weight_applicable(const ConvP& p) {
    const char* e = getenv("MSYNTH_WROWS");          // tuning / test switch (0 disables)
    if (e && atoi(e) == 0) return false;
    return plan_wrows(p).ok;
}

size_t msw_bwd_weight_ws(const ConvP& p) {
    const WrPlan q = plan_wrows(p);
    return (size_t)q.nsplit * q.stride_floats * sizeof(float);
}

const char* msw_bwd_weight_name(const ConvP& p) {
    static thread_local char buf[64];
    const WrPlan q = plan_wrows(p);
    if (wrows3_ok(q, p.K, q.tm, q.vec, p.act == MS_ACT_LRELU)) {
        snprintf(buf, sizeof(buf), "k_wgrad_rows3<1, 3>");
        return buf;
    }
    snprintf(buf, sizeof(buf), "k_wgrad_rows<%d, %d, %s, %d>", p.K, q.tm, q.vec ? "true" : "false", q.p.kcols / 4);
    return buf;
}

// ---- transposed-conv weight gradient through the phase-split identity: M = Cin_T (rows of x_T),
// X rows = the S phases of the Cout_T gradient channels, 3 taps (d = -1, 0, +1), contraction over
// (b, q).  Produces dWq[Cin_T][(co, r), d] (the caller un-packs it into (Cin_T, Cout_T, K)).
// c = mirrored conv of the transposed conv: Cin_T = c.Cout, Cout_T = c.Cin, Lin_T = c.Lout.
namespace {

WrPlan plan_wrows_t(const ConvP& c) {
    WrPlan q;
    q.ok = false;
    const int S = c.stride, CinT = c.Cout, CoutT = c.Cin, LinT = c.Lout;
    if (!(S == 2 || S == 8) || c.K != 2 * S || 2 * c.pad != S || c.dil != 1 || c.groups != 1) return q;
    if (CinT < 64 || (CoutT * S) % CB || LinT % 4) return q;
    if ((long long)c.B * CinT * LinT >= (1LL << 31) || (long long)c.B * CoutT * S * LinT >= (1LL << 31)) return q;
    WrP& p = q.p;
    p.B = c.B; p.CK = CoutT * S; p.L = LinT; p.M = CinT; p.dil = 1; p.pad = 1;
    p.g_kind = MS_ACT_NONE; p.x_kind = MS_ACT_NONE; p.slope = c.slope;
    if (p.L >= KMAX) { p.Lt = KMAX; p.R = 1; p.tiles_per_row = ms_ceil_div(p.L, KMAX); }
    else { p.Lt = p.L; p.R = 1; p.tiles_per_row = 1; }
    p.nchunks = p.B * p.tiles_per_row;
    p.SS = p.Lt + 2;
    p.RSZ = p.SS;
    p.kcols = p.Lt <= 32 ? 32 : 64;
    if (p.Lt != 32 && p.Lt != 64) return q;
    if ((CB / S) * ((p.SS * S + 6) / 4) > 5 * 256) return q;
    p.PG = (p.kcols + 2) | 1;
    p.PX = (p.kcols + 2 + 2) | 1;
    q.vec = true;
    q.tm = CinT >= 128 ? 2 : 1;
    const int BM = 64 * q.tm;
    q.lds = (size_t)(2 * (BM * p.PG + CB * p.PX) + 256) * sizeof(float);
    const int tiles = ms_ceil_div(p.M, BM) * (p.CK / CB);
    q.stride_floats = (size_t)p.M * p.CK * 3 + p.M;
    const int slots = 256 * (q.lds > 80 * 1024 ? 1 : 2);
    int ns_max = p.nchunks / 2 > 0 ? p.nchunks / 2 : 1;
    const size_t max_by_bytes = ((size_t)48 << 20) / (q.stride_floats * 4);
    if ((size_t)ns_max > max_by_bytes) ns_max = max_by_bytes > 0 ? (int)max_by_bytes : 1;
    const int ns = pick_slices(tiles, slots, ns_max);
    p.cps = ms_ceil_div(p.nchunks, ns);
    q.nsplit = ms_ceil_div(p.nchunks, p.cps);
    q.grid = dim3((unsigned)(p.CK / CB), (unsigned)ms_ceil_div(p.M, BM), (unsigned)q.nsplit);
    q.ok = true;
    return q;
}

template <int TM, int KPI, int AK, int XS>
void launch_wrows_t(const WrPlan& q, const float* gy, const float* y_act, const float* x, float* partial,
                    hipStream_t s) {
    static unsigned long long attr_set = 0;
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_rows<3, TM, true, KPI, AK, XS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        ms_done_on_device(attr_set);
    }
    hipLaunchKernelGGL((k_wgrad_rows<3, TM, true, KPI, AK, XS>), q.grid, dim3(256), q.lds, s, q.p, gy, y_act, x,
                       (const float*)nullptr, partial, q.stride_floats, q.mp);
}

template <int AK, int XS>
void launch_wrows_t2(const WrPlan& q, const float* gy, const float* y_act, const float* x, float* partial,
                     hipStream_t s) {
    if (q.tm == 2) {
        if (q.p.kcols == 64) launch_wrows_t<2, 16, AK, XS>(q, gy, y_act, x, partial, s);
        else launch_wrows_t<2, 8, AK, XS>(q, gy, y_act, x, partial, s);
    } else {
        if (q.p.kcols == 64) launch_wrows_t<1, 16, AK, XS>(q, gy, y_act, x, partial, s);
        else launch_wrows_t<1, 8, AK, XS>(q, gy, y_act, x, partial, s);
    }
}

}  // namespace

size_t msw_convt_ws(const ConvP& c) {
    const char* e = getenv("MSYNTH_WROWS");
    if (e && atoi(e) == 0) return 0;
    const WrPlan q = plan_wrows_t(c);
    return q.ok ? (size_t)q.nsplit * q.stride_floats * sizeof(float) : 0;
}

int msw_convt_dwq(const ConvP& c, const float* x, const float* gy, const float* y_act, float* dwq,
                  void* ws, size_t ws_bytes, hipStream_t s) {
    const char* e = getenv("MSYNTH_WROWS");
    if (e && atoi(e) == 0) return MS_ERR_UNSUPPORTED;
    const WrPlan q = plan_wrows_t(c);
    if (!q.ok) return MS_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < (size_t)q.nsplit * q.stride_floats * sizeof(float)) return MS_ERR_UNSUPPORTED;
    if (((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)(y_act ? y_act : gy))) & 15) != 0) return MS_ERR_UNSUPPORTED;
    if (y_act && c.act != MS_ACT_LRELU) return MS_ERR_UNSUPPORTED;
    if (c.in_act && y_act) return MS_ERR_UNSUPPORTED;
    float* partial = (float*)ws;
    const int S = c.stride;
    if (c.in_act) { if (S == 8) launch_wrows_t2<5, 8>(q, gy, y_act, x, partial, s); else launch_wrows_t2<5, 2>(q, gy, y_act, x, partial, s); }
    else if (y_act) { if (S == 8) launch_wrows_t2<3, 8>(q, gy, y_act, x, partial, s); else launch_wrows_t2<3, 2>(q, gy, y_act, x, partial, s); }
    else { if (S == 8) launch_wrows_t2<0, 8>(q, gy, y_act, x, partial, s); else launch_wrows_t2<0, 2>(q, gy, y_act, x, partial, s); }
    MS_CHECK_LAUNCH();
    return msm_wgrad_reduce(partial, q.stride_floats, q.nsplit, (size_t)q.p.M * q.p.CK * 3, q.p.M, dwq,
                            (float*)nullptr, 0.f, s);
}

int msw_conv1d_bwd_weight(const ConvP& c, const float* x, const float* gy, const float* y_act,
                          float* gw, float* gb, float beta, void* ws, size_t ws_bytes,
                          hipStream_t s) {
    const WrPlan q = plan_wrows(c);
    if (!q.ok) return MS_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < (size_t)q.nsplit * q.stride_floats * sizeof(float)) return MS_ERR_WORKSPACE;
    float* partial = (float*)ws;
    const bool vec = q.vec && ((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)(y_act ? y_act : gy))) & 15) == 0;
    if (!vec && (q.p.kcols > 64 || q.p.RSZ > 128)) return MS_ERR_UNSUPPORTED;
    if (q.refl && (!vec || !y_act)) return MS_ERR_UNSUPPORTED;
    if (c.K == 1) launch_wrows<1>(q, vec, x, gy, y_act, partial, s);
    else if (c.K == 3) launch_wrows<3>(q, vec, x, gy, y_act, partial, s);
    else if (c.K == 5) launch_wrows<5>(q, vec, x, gy, y_act, partial, s);
    else launch_wrows<7>(q, vec, x, gy, y_act, partial, s);
    MS_CHECK_LAUNCH();
    return msm_wgrad_reduce(partial, q.stride_floats, q.nsplit, (size_t)c.Cout * c.Cin * c.K, c.Cout, gw,
                            gb, beta, s);
}

// ------------------------------------------------------------------ batched launch (see WrMulti)
namespace {

struct WrReduceMulti {
    float* gw[WR_MULTI_MAX];
    float* gb[WR_MULTI_MAX];
    float beta[WR_MULTI_MAX];
};

// out[prob] = beta*out + sum over the slices' slabs of problem blockIdx.y (16-byte form of k_wgrad_reduce_v4)
__global__ __launch_bounds__(256) void k_wgrad_reduce_multi(const float* __restrict__ partial,
                                                           size_t slice_stride, size_t prob_stride,
                                                           int nsplit, size_t wsize, size_t total,
                                                           WrReduceMulti o) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int prob = blockIdx.y;
    partial += (size_t)prob * prob_stride;
    const size_t i = ((size_t)blockIdx.x * 64 + lane) * 4;
    const bool ok = i < total;                       // total % 4 == 0
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (ok) {
        int z = wv;
        for (; z + 4 < nsplit; z += 8) {
            const float4 a = *reinterpret_cast<const float4*>(partial + (size_t)z * slice_stride + i);
            const float4 b = *reinterpret_cast<const float4*>(partial + (size_t)(z + 4) * slice_stride + i);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        for (; z < nsplit; z += 4) {
            const float4 a = *reinterpret_cast<const float4*>(partial + (size_t)z * slice_stride + i);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    red[wv][lane] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (wv == 0 && ok) {
        float4 r;
        r.x = (red[0][lane].x + red[1][lane].x) + (red[2][lane].x + red[3][lane].x);
        r.y = (red[0][lane].y + red[1][lane].y) + (red[2][lane].y + red[3][lane].y);
        r.z = (red[0][lane].z + red[1][lane].z) + (red[2][lane].z + red[3][lane].z);
        r.w = (red[0][lane].w + red[1][lane].w) + (red[2][lane].w + red[3][lane].w);
        float* outb = o.gb[prob];
        if (i >= wsize && !outb) return;
        float4* dst = i < wsize ? reinterpret_cast<float4*>(o.gw[prob] + i) : reinterpret_cast<float4*>(outb + (i - wsize));
        const float beta = o.beta[prob];
        if (beta != 0.f) {
            const float4 pv = *dst;
            r.x += beta * pv.x; r.y += beta * pv.y; r.z += beta * pv.z; r.w += beta * pv.w;
        }
        *dst = r;
    }
}

// plan of the batched launch; ok only for n identical k3 LeakyReLU convs (dilation / padding may differ)
WrPlan plan_wrows_multi(const ConvP* cs, int n) {
    WrPlan q;
    q.ok = false;
    if (n < 2 || n > WR_MULTI_MAX) return q;
    int imax = 0;
    for (int i = 0; i < n; ++i) {
        const ConvP& c = cs[i];
        if (c.B != cs[0].B || c.Cin != cs[0].Cin || c.Lin != cs[0].Lin || c.Cout != cs[0].Cout || c.K != 3 ||
            c.act != MS_ACT_LRELU || c.in_act || c.pad_mode != MS_PAD_ZERO || c.pad != c.dil || c.slope != cs[0].slope)
            return q;
        if (c.dil > cs[imax].dil) imax = i;
    }
    q = plan_wrows(cs[imax]);                        // LDS pitches sized for the widest halo
    if (!q.ok || !q.vec || q.refl || q.p.R != 1 || (q.p.M % 4)) { q.ok = false; return q; }
    WrP& p = q.p;
    const int BM = 64 * q.tm;
    const int tiles_m = ms_ceil_div(p.M, BM), tiles_c = ms_ceil_div(p.CK, CB);
    const size_t stride_one = (size_t)p.M * p.CK * 3 + p.M;
    const int tiles = n * tiles_m * tiles_c;
    const int slots = 256 * (q.lds > 80 * 1024 ? 1 : 2);
    int ns_max = p.nchunks / 4 > 0 ? p.nchunks / 4 : 1;
    const size_t max_by_bytes = ((size_t)48 << 20) / (stride_one * n * 4);
    if ((size_t)ns_max > max_by_bytes) ns_max = max_by_bytes > 0 ? (int)max_by_bytes : 1;
    const int ns = pick_slices(tiles, slots, ns_max);
    p.cps = ms_ceil_div(p.nchunks, ns);
    q.nsplit = ms_ceil_div(p.nchunks, p.cps);
    q.grid = dim3((unsigned)tiles_c, (unsigned)(n * tiles_m), (unsigned)q.nsplit);
    q.stride_floats = stride_one * n;                // slice stride
    q.mp.n = n;
    q.mp.tiles_m = tiles_m;
    q.mp.slab_stride = stride_one;
    for (int i = 0; i < n; ++i) { q.mp.dil[i] = cs[i].dil; q.mp.pad[i] = cs[i].pad; }
    return q;
}

}  // namespace

size_t msw_multi_ws(const ConvP* cs, int n) {
    const char* e = getenv("MSYNTH_WMULTI");         // tuning / test switch (0 disables the batched launch)
    if (e && atoi(e) == 0) return 0;
    const WrPlan q = plan_wrows_multi(cs, n);
    return q.ok ? (size_t)q.nsplit * q.stride_floats * sizeof(float) : 0;
}

// does the batched split kernel take these convs with SIGN WORDS in place of the activations (k_wgrad_rows3<., 2, true>)?
bool msw_multi_takes_signs(const ConvP* cs, int n) {
    if (msw_multi_ws(cs, n) == 0) return false;
    static const int sw = getenv("MSYNTH_WROWS3_NP") ? atoi(getenv("MSYNTH_WROWS3_NP")) : 2;
    if (sw == 3) return false;
    const WrPlan q = plan_wrows_multi(cs, n);
    return q.ok && wrows3_ok(q, 3, q.tm, true, true);
}

int msw_conv1d_bwd_weight_multi(const ConvP* cs, int n, const float* const* x, const float* const* gy,
                                const float* const* y_act, float* const* gw, float* const* gb,
                                const float* beta, const float* const* xmax, const float* const* gmax, int signs, void* ws,
                                size_t ws_bytes, hipStream_t s) {
    if (msw_multi_ws(cs, n) == 0) return MS_ERR_UNSUPPORTED;
    if (signs) {
        if (!msw_multi_takes_signs(cs, n) || !xmax || !gmax) return MS_ERR_UNSUPPORTED;
        for (int i = 0; i < n; ++i)
            if (!xmax[i] || !gmax[i]) return MS_ERR_UNSUPPORTED;
    }
    WrPlan q = plan_wrows_multi(cs, n);
    q.mp.signs = signs ? 1 : 0;
    if (!ws || ws_bytes < (size_t)q.nsplit * q.stride_floats * sizeof(float) || (((uintptr_t)ws) & 15)) return MS_ERR_UNSUPPORTED;
    WrReduceMulti o;
    for (int i = 0; i < n; ++i) {
        if (!x[i] || !gy[i] || !y_act[i] || !gw[i] || !gb[i]) return MS_ERR_UNSUPPORTED;
        if (((((uintptr_t)x[i]) | ((uintptr_t)gy[i]) | ((uintptr_t)y_act[i]) | ((uintptr_t)gw[i]) | ((uintptr_t)gb[i])) & 15) != 0)
            return MS_ERR_UNSUPPORTED;
        if (beta[i] != 0.f && beta[i] != 1.f) return MS_ERR_INVALID_ARG;
        q.mp.x[i] = x[i]; q.mp.g[i] = gy[i]; q.mp.gact[i] = y_act[i];
        q.mp.xmax[i] = xmax ? xmax[i] : nullptr; q.mp.gmax[i] = gmax ? gmax[i] : nullptr;
        o.gw[i] = gw[i]; o.gb[i] = gb[i]; o.beta[i] = beta[i];
    }
    float* partial = (float*)ws;
    launch_wrows<3>(q, true, x[0], gy[0], y_act[0], partial, s);
    MS_CHECK_LAUNCH();
    const size_t wsize = (size_t)q.p.M * q.p.CK * 3, total = wsize + (size_t)q.p.M;
    hipLaunchKernelGGL(k_wgrad_reduce_multi, dim3((unsigned)((total / 4 + 63) / 64), (unsigned)n), dim3(256), 0, s,
                       partial, q.stride_floats, q.mp.slab_stride, q.nsplit, wsize, total, o);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

// ------------------------------------------------------------------ 32-channel layers
// The 32 -> 32 k3 atoms at L = 8192 move 100 MB for 1.6 GFLOP: the weight gradient is bound by HBM,
// and the whole 32 x (32 x 3) result is ONE MFMA tile row.  Work unit of a WAVE (as in the grouped
// convs): 64 time steps of one batch row -- the wave stages the 32 x 64 gradient tile (activation
// derivative applied) and the 32 x (64 + halo) input tile in its private LDS region with 16-byte loads,
// multiplies 32 k-pair steps x 3 taps, and prefetches the next unit meanwhile.  No workgroup barrier
// until the end, where the 4 waves' accumulators are summed through LDS into one slab per workgroup.
namespace {

constexpr int W32_PG = 65;                 // gradient tile pitch (odd)
constexpr int W32_PX = 89;                 // input tile pitch (odd, >= 64 + 18 + 3 + 3)
constexpr int W32_WF = 32 * W32_PG + 32 * W32_PX;   // floats per wave region (4928)

struct W32P {
    int B, L, dil, pad, tiles, nunits;
    float slope;
};

// several 32-channel weight gradients in one launch: workgroups [prob*g_per, (prob+1)*g_per) own problem prob
struct W32Multi {
    int n, g_per;
    const float* x[WR_MULTI_MAX];
    const float* g[WR_MULTI_MAX];
    const float* gact[WR_MULTI_MAX];
    int dil[WR_MULTI_MAX], pad[WR_MULTI_MAX];
};

// AK: 1: LeakyReLU derivative on the gradient (y_act given); 0: plain gradient.  GM: y_act arrives as sign words (k_wgrad_rows3)
template <int AK, bool GM = false>
__global__ __launch_bounds__(256, 2) void k_wgrad32(W32P p, const float* __restrict__ X_,
                                                   const float* __restrict__ G_,
                                                   const float* __restrict__ Gact_,
                                                   float* __restrict__ partial, size_t pstride, W32Multi mp) {
    const float* __restrict__ X = X_;
    const float* __restrict__ G = G_;
    const float* __restrict__ Gact = Gact_;
    int bx = blockIdx.x, gx = gridDim.x;
    if (mp.n > 0) {
        const int prob = bx / mp.g_per;
        bx -= prob * mp.g_per;
        gx = mp.g_per;
        X = mp.x[prob]; G = mp.g[prob]; Gact = mp.gact[prob];
        p.dil = mp.dil[prob]; p.pad = mp.pad[prob];
    }
    constexpr int K = 3, NGQ = 8, NXQ = 11;
    extern __shared__ __attribute__((aligned(16))) float lds[];       // 4 * W32_WF floats
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5;
    float* Gs = lds + wid * W32_WF;
    float* Xs = Gs + 32 * W32_PG;
    const int H = 2 * p.dil;
    const int SS = 64 + H;
    const int NVS = (SS + 6) >> 2;                         // aligned vectors covering one input row
    const int sh = (4 - (p.pad & 3)) & 3;

    // lane-invariant piece descriptors
    int g_off[NGQ], g_lds[NGQ];
#pragma unroll
    for (int q = 0; q < NGQ; ++q) {
        const int idx = lane + 64 * q, row = idx >> 4, v = idx & 15;
        g_off[q] = row * p.L + 4 * v;
        g_lds[q] = row * W32_PG + 4 * v;
    }
    int x_off[NXQ], x_lds[NXQ], x_u[NXQ];
    bool x_in[NXQ];
#pragma unroll
    for (int q = 0; q < NXQ; ++q) {
        const int idx = lane + 64 * q, row = idx / NVS, sv = idx - row * NVS;
        x_in[q] = row < 32;
        x_u[q] = 4 * sv - sh;                              // column of the vector's first element
        x_off[q] = (x_in[q] ? row : 0) * p.L + x_u[q] - p.pad;
        x_lds[q] = (x_in[q] ? row : 0) * W32_PX + x_u[q];
    }

    f32x16 acc[K];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float asum = 0.f;

    float4 gv[NGQ], ga[AK && !GM ? NGQ : 1], xv[NXQ];
    uint2 gs[GM ? NGQ : 1];                           // (GM) sign words of the piece's four columns
    auto gload = [&](int b, int ti) {
        const int t0 = ti * 64;
        const float* gb = G + (size_t)b * 32 * p.L + t0;
        const float* ab = Gact + (size_t)b * 32 * p.L + t0;
        const unsigned short* sb = reinterpret_cast<const unsigned short*>(Gact) + (size_t)b * 2 * p.L + t0;
        const float* xb = X + (size_t)b * 32 * p.L + t0;
#pragma unroll
        for (int q = 0; q < NGQ; ++q) {
            const bool ok = t0 + 4 * ((lane + 64 * q) & 15) < p.L;       // L % 4 == 0: all in or all out
            const int o = ok ? g_off[q] : 0;
            gv[q] = *reinterpret_cast<const float4*>(gb + o);
            // (GM) row (lane >> 4) + 4 q: lane half q & 1, register (lane >> 4) + 4 (q >> 1)
            if constexpr (GM) gs[q] = *reinterpret_cast<const uint2*>(sb + (ok ? (q & 1) * p.L + 4 * ((lane + 64 * q) & 15) : 0));
            else if (AK) ga[q] = *reinterpret_cast<const float4*>(ab + o);
        }
#pragma unroll
        for (int q = 0; q < NXQ; ++q) {
            const int t = t0 - p.pad + x_u[q];                            // multiple of 4
            const bool ok = x_in[q] && t >= 0 && t < p.L;
            xv[q] = *reinterpret_cast<const float4*>(ok ? xb + x_off[q] : X);
        }
    };

    const int wstride = gx * 4;
    const int db = wstride / p.tiles, dt = wstride - db * p.tiles;
    int unit = bx * 4 + wid;
    int b = unit / p.tiles, ti = unit - b * p.tiles;
    if (unit < p.nunits) gload(b, ti);
    const float* ap = Gs + (lane & 31) * W32_PG + h;
    const float* bp = Xs + (lane & 31) * W32_PX + h;
    for (; unit < p.nunits; unit += wstride) {
        const int t0 = ti * 64;
#pragma unroll
        for (int q = 0; q < NGQ; ++q) {
            const bool ok = t0 + 4 * ((lane + 64 * q) & 15) < p.L;
            float e[4] = {gv[q].x, gv[q].y, gv[q].z, gv[q].w};
            if constexpr (GM) {
                const int shf = 15 - ((lane >> 4) + 4 * (q >> 1));
                const unsigned w4[4] = {gs[q].x, gs[q].x >> 16, gs[q].y, gs[q].y >> 16};
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = ((w4[i] >> shf) & 1u) ? e[i] : e[i] * p.slope;
            } else if (AK) {
                const float a[4] = {ga[q].x, ga[q].y, ga[q].z, ga[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = a[i] > 0.f ? e[i] : e[i] * p.slope;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) Gs[g_lds[q] + i] = ok ? e[i] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NXQ; ++q) {
            const int t = t0 - p.pad + x_u[q];
            const bool ok = x_in[q] && t >= 0 && t < p.L;
            const float e[4] = {xv[q].x, xv[q].y, xv[q].z, xv[q].w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (x_in[q] && x_u[q] + i >= 0 && x_u[q] + i < SS) Xs[x_lds[q] + i] = ok ? e[i] : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int nb = b + db, nti = ti + dt;
        if (nti >= p.tiles) { nti -= p.tiles; ++nb; }
        if (unit + wstride < p.nunits) gload(nb, nti);
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const float a = ap[2 * kk];
            asum += a;
#pragma unroll
            for (int j = 0; j < K; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[2 * kk + j * p.dil], acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        b = nb; ti = nti;
    }

    // workgroup reduction of the 4 waves' tiles through LDS, then one slab per workgroup
    __syncthreads();
    float* mine = lds + wid * W32_WF;                     // [j][reg][lane]: 3 * 16 * 64 floats
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(j * 16 + r) * 64 + lane] = acc[j][r];
    const float bsum = asum + __shfl_xor(asum, 32, 64);
    if (lane < 32) mine[K * 16 * 64 + lane] = bsum;
    __syncthreads();
    float* part = partial + (size_t)blockIdx.x * pstride;
    for (int e = tid; e < K * 16 * 64 + 32; e += 256) {
        const float v = (lds[e] + lds[W32_WF + e]) + (lds[2 * W32_WF + e] + lds[3 * W32_WF + e]);
        if (e < K * 16 * 64) {
            // D[row][col]: col = lane&31 (input channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
            const int l = e & 63, rr = (e >> 6) & 15, j = e >> 10;
            const int m = (rr & 3) + 8 * (rr >> 2) + 4 * (l >> 5), c = l & 31;
            part[(size_t)m * (32 * K) + c * K + j] = v;
        } else {
            part[(size_t)32 * 32 * K + (e - K * 16 * 64)] = v;
        }
    }
}

}  // namespace

bool msw32_applicable(const ConvP& c) {
    const char* e = getenv("MSYNTH_WROWS");
    if (e && atoi(e) == 0) return false;
    return c.groups == 1 && c.stride == 1 && c.Lout == c.Lin && c.pad_mode == MS_PAD_ZERO && c.K == 3 &&
           c.Cout == 32 && c.Cin == 32 && c.dil >= 1 && c.dil <= 9 && c.pad == c.dil && !c.in_act &&
           c.Lin % 4 == 0 && c.Lin >= 64 && (c.act == MS_ACT_LRELU || c.act == MS_ACT_NONE) &&
           (long long)c.B * 32 * c.Lin < (1LL << 31);
}

static int w32_grid(const ConvP& c) {
    const int units = c.B * ms_ceil_div(c.Lin, 64);
    int g = ms_ceil_div(units, 4);
    return g > 512 ? 512 : g;
}

size_t msw32_ws(const ConvP& c) { return (size_t)w32_grid(c) * (32 * 32 * 3 + 32) * sizeof(float); }

int msw32_bwd_weight(const ConvP& c, const float* x, const float* gy, const float* y_act, float* gw,
                     float* gb, float beta, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < msw32_ws(c)) return MS_ERR_WORKSPACE;
    if (((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)(y_act ? y_act : gy))) & 15) != 0) return MS_ERR_UNSUPPORTED;
    W32P p;
    p.B = c.B; p.L = c.Lin; p.dil = c.dil; p.pad = c.pad; p.slope = c.slope;
    p.tiles = ms_ceil_div(c.Lin, 64); p.nunits = c.B * p.tiles;
    const int g = w32_grid(c);
    const size_t stride = 32 * 32 * 3 + 32;
    float* partial = (float*)ws;
    const size_t lds = (size_t)4 * W32_WF * sizeof(float);
    static unsigned long long attr_set = 0;                    // > 64 KiB of dynamic LDS needs the opt-in once
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad32<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad32<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        ms_done_on_device(attr_set);
    }
    if (y_act && c.act == MS_ACT_LRELU)
        hipLaunchKernelGGL(k_wgrad32<1>, dim3(g), dim3(256), lds, s, p, x, gy, y_act, partial, stride, W32Multi{});
    else
        hipLaunchKernelGGL(k_wgrad32<0>, dim3(g), dim3(256), lds, s, p, x, gy, gy, partial, stride, W32Multi{});
    MS_CHECK_LAUNCH();
    return msm_wgrad_reduce(partial, stride, g, (size_t)32 * 32 * 3, 32, gw, gb, beta, s);
}

// n 32 -> 32 k3 LeakyReLU weight gradients of one length (a ResidualStack's six) in one launch pair:
// each problem gets 512/n workgroups, i.e. n times more units per wave and the same slab bytes in total
static bool w32_multi_ok(const ConvP* cs, int n) {
    const char* e = getenv("MSYNTH_WMULTI");
    if (e && atoi(e) == 0) return false;
    if (n < 2 || n > WR_MULTI_MAX) return false;
    for (int i = 0; i < n; ++i) {
        if (!msw32_applicable(cs[i]) || cs[i].B != cs[0].B || cs[i].Lin != cs[0].Lin || cs[i].act != MS_ACT_LRELU ||
            cs[i].slope != cs[0].slope)
            return false;
    }
    return true;
}

size_t msw32_multi_ws(const ConvP* cs, int n) {
    if (!w32_multi_ok(cs, n)) return 0;
    return (size_t)n * (512 / n) * (32 * 32 * 3 + 32) * sizeof(float);
}

bool msw32_multi_takes_signs(const ConvP* cs, int n) { return w32_multi_ok(cs, n); }

int msw32_bwd_weight_multi(const ConvP* cs, int n, const float* const* x, const float* const* gy,
                           const float* const* y_act, float* const* gw, float* const* gb, const float* beta, int signs,
                           void* ws, size_t ws_bytes, hipStream_t s) {
    if (!w32_multi_ok(cs, n)) return MS_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < msw32_multi_ws(cs, n) || (((uintptr_t)ws) & 15)) return MS_ERR_UNSUPPORTED;
    W32Multi mp;
    WrReduceMulti o;
    mp.n = n;
    const ConvP& c = cs[0];
    const int units = c.B * ms_ceil_div(c.Lin, 64);
    int g_per = 512 / n;
    if (g_per > ms_ceil_div(units, 4)) g_per = ms_ceil_div(units, 4);
    mp.g_per = g_per;
    for (int i = 0; i < n; ++i) {
        if (!x[i] || !gy[i] || !y_act[i] || !gw[i] || !gb[i]) return MS_ERR_UNSUPPORTED;
        if (((((uintptr_t)x[i]) | ((uintptr_t)gy[i]) | ((uintptr_t)y_act[i]) | ((uintptr_t)gw[i]) | ((uintptr_t)gb[i])) & 15) != 0)
            return MS_ERR_UNSUPPORTED;
        if (beta[i] != 0.f && beta[i] != 1.f) return MS_ERR_INVALID_ARG;
        mp.x[i] = x[i]; mp.g[i] = gy[i]; mp.gact[i] = y_act[i];
        mp.dil[i] = cs[i].dil; mp.pad[i] = cs[i].pad;
        o.gw[i] = gw[i]; o.gb[i] = gb[i]; o.beta[i] = beta[i];
    }
    W32P p;
    p.B = c.B; p.L = c.Lin; p.dil = c.dil; p.pad = c.pad; p.slope = c.slope;
    p.tiles = ms_ceil_div(c.Lin, 64); p.nunits = c.B * p.tiles;
    const size_t stride = 32 * 32 * 3 + 32;
    float* partial = (float*)ws;
    const size_t lds = (size_t)4 * W32_WF * sizeof(float);
    static unsigned long long attr_set = 0;
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad32<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad32<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        ms_done_on_device(attr_set);
    }
    if (signs) hipLaunchKernelGGL((k_wgrad32<1, true>), dim3(n * g_per), dim3(256), lds, s, p, x[0], gy[0], y_act[0], partial, stride, mp);
    else hipLaunchKernelGGL(k_wgrad32<1>, dim3(n * g_per), dim3(256), lds, s, p, x[0], gy[0], y_act[0], partial, stride, mp);
    MS_CHECK_LAUNCH();
    // slabs of problem i: workgroups [i*g_per, (i+1)*g_per) -> slice stride = one slab, problem stride = g_per slabs
    const size_t wsize = 32 * 32 * 3, total = wsize + 32;
    hipLaunchKernelGGL(k_wgrad_reduce_multi, dim3((unsigned)((total / 4 + 63) / 64), (unsigned)n), dim3(256), 0, s,
                       partial, stride, (size_t)g_per * stride, g_per, wsize, total, o);
    MS_CHECK_LAUNCH();
    return MS_OK;
}
