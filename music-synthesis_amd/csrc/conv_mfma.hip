// f32-MFMA implicit-GEMM convolution kernels for gfx950 (CDNA4).
//
// The dense stride-1 convolutions of the hot path (the 24 dilated k3 convs of the generator's
// residual atoms, its k7 input conv, the discriminator's 1024->1024 k5 conv) carry ~3/4 of the
// train step's FLOPs at 48..250 FLOP/B: in exact fp32 they are bound by the matrix pipe, not by
// HBM.  They are lowered to an im2col x weight contraction executed with v_mfma_f32_32x32x2_f32
// (fp32 in / fp32 accumulate, bit-identical to an fmaf chain, 64 FLOP/clk/SIMD):
//
//   forward        Y[m, (b,t)]  = sum_{(c,j)}  W[m, c, j]        * X[b, c, t + j*dil - pad]
//   backward data  dX[m, (b,t)] = sum_{(c,j')} W[c, m, K-1-j']   * dY'[b, c, t + j'*dil - pad]
//   backward wgt   dW[m, (c,j)] = sum_{(b,t)}  dY'[b, m, t]      * X[b, c, t + j*dil - pad]
//
// (dY' = dY * act'(Y) is formed in the operand loader.)  A 256-thread workgroup (4 waves of 64)
// owns a BM x BN output tile; each wave a TM x TN grid of 32x32 MFMA tiles.  Per K-chunk the A
// (weights or dY') and B (im2col of the activations, built on the fly: nothing is materialised
// in HBM) tiles are staged in LDS; time is the lane-fast index of every global access, so
// activation traffic is coalesced along contiguous audio frames.  Global loads of chunk c+1 are
// in flight while the MFMAs of chunk c run.
#include <cstdlib>
#include "conv_mfma.h"
#include "conv_rows2.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 16;        // GEMM-K elements staged per chunk
constexpr int KCP = KC + 1;   // odd LDS row stride: conflict-free 32-lane column reads

struct IgP {
    int B, CK, L, M, dil, off0, pad_mode, act, in_act, N, KG;
    float slope;
    int in_s;   // weight-grad B operand: channel c' = co*in_s + r is phase r of a stride-in_s signal
};

// branch-free source index: every load below is issued unconditionally from a clamped (always
// valid) address and masked afterwards, so hipcc batches the loads of a chunk instead of waiting
// for each one (a load under a run-time branch costs a full memory round trip per element)
__device__ __forceinline__ int src_index_sel(int t, int L, int reflect) {
    const int r = t < 0 ? -t : (t >= L ? 2 * (L - 1) - t : t);
    return reflect ? r : t;
}

// ------------------------------------------------------------ forward / backward data
template <int WGM, int WGN, int TM, int TN, int K, bool TRANS>
__global__ __launch_bounds__(256) void k_igemm_conv(IgP p, const float* __restrict__ X,
                                                   const float* __restrict__ Xact,
                                                   const float* __restrict__ W,
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ res,
                                                   float* __restrict__ Y,
                                                   float* __restrict__ Yact) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int RA = BM * KC / 256, RB = BN * KC / 256;
    constexpr int NCOL = BN > 256 ? BN / 256 : 1;         // columns a loader thread owns
    constexpr int ROWSTEP = BN > 256 ? 1 : 256 / BN;      // B-tile rows covered per pass
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(RB % NCOL == 0, "loader mapping");
    __shared__ float As[BM * KCP];
    __shared__ float Bs[KC * BN];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // B loader: fixed column(s) per thread
    int cb[NCOL], ct[NCOL];
    bool cvalid[NCOL];
#pragma unroll
    for (int q = 0; q < NCOL; ++q) {
        const int n = n0 + (tid % BN) + q * 256;
        cvalid[q] = n < p.N;
        const int nn = cvalid[q] ? n : 0;
        cb[q] = nn / p.L;
        ct[q] = nn - cb[q] * p.L;
    }
    const int brow0 = BN > 256 ? 0 : tid / BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // absent activation operand: alias the data (second load hits L1) with a pass-through kind
    const float* Xa = Xact ? Xact : X;
    const int in_act = Xact ? p.in_act : MS_ACT_NONE;
    float ra[RA], rb[RB];
    auto gload = [&](int kc0) {
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const int e = r * 256 + tid;
            bool ok;
            size_t off;
            if (!TRANS) {
                const int kk = e % KC, m = e / KC;
                ok = m0 + m < p.M && kc0 + kk < p.KG;
                off = (size_t)(m0 + m) * p.KG + kc0 + kk;
            } else {
                const int m = e % BM, kk = e / BM;
                const int kg = kc0 + kk;
                const int c = kg / K, j = kg - c * K;
                ok = m0 + m < p.M && kg < p.KG;
                off = ((size_t)c * p.M + m0 + m) * K + (K - 1 - j);
            }
            const float v = W[ok ? off : 0];
            ra[r] = ok ? v : 0.f;
        }
        float xv[RB], av[RB];
        bool ok[RB];
#pragma unroll
        for (int r = 0; r < RB / NCOL; ++r) {
            const int kg = kc0 + brow0 + r * ROWSTEP;
            const int c = kg / K, j = kg - c * K;
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                const int sidx = src_index_sel(ct[q] + j * p.dil + p.off0, p.L, p.pad_mode == MS_PAD_REFLECT);
                const bool v = cvalid[q] && kg < p.KG && (unsigned)sidx < (unsigned)p.L;
                const size_t off = v ? ((size_t)cb[q] * p.CK + c) * p.L + sidx : 0;
                ok[r * NCOL + q] = v;
                xv[r * NCOL + q] = X[off];
                av[r * NCOL + q] = Xa[off];
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) rb[r] = ok[r] ? ms_act_grad(xv[r], av[r], in_act, p.slope) : 0.f;
    };
    auto lstore = [&]() {
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const int e = r * 256 + tid;
            if (!TRANS) As[(e / KC) * KCP + (e % KC)] = ra[r];
            else As[(e % BM) * KCP + (e / BM)] = ra[r];
        }
#pragma unroll
        for (int r = 0; r < RB / NCOL; ++r) {
            const int kk = brow0 + r * ROWSTEP;
#pragma unroll
            for (int q = 0; q < NCOL; ++q) Bs[kk * BN + (tid % BN) + q * 256] = rb[r * NCOL + q];
        }
    };

    const int nchunks = (p.KG + KC - 1) / KC;
    gload(0);
    lstore();
    __syncthreads();
    const int arow = (wm * TM * 32 + (lane & 31)) * KCP + (lane >> 5);
    const int bcol = (lane >> 5) * BN + wn * TN * 32 + (lane & 31);
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) gload((c + 1) * KC);
#pragma unroll
        for (int k2 = 0; k2 < KC / 2; ++k2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[arow + i * 32 * KCP + 2 * k2];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[bcol + 2 * k2 * BN + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            lstore();
            __syncthreads();
        }
    }

    // epilogue: D[row][col]: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); bias and
    // residual loads of a tile are batched ahead of its stores (no loads under per-element branches)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        if (n >= p.N) continue;
        const int b = n / p.L, t = n - b * p.L;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * TM * 32 + i * 32 + 4 * (lane >> 5);
            const size_t obase = ((size_t)b * p.M + mb) * p.L + t;
            float bv[16], rv[16];
            bool ok[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = (r & 3) + 8 * (r >> 2);
                ok[r] = mb + dm < p.M;
                bv[r] = bias ? bias[ok[r] ? mb + dm : 0] : 0.f;
            }
            if (res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    rv[r] = res[ok[r] ? obase + (size_t)dm * p.L : 0];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = (r & 3) + 8 * (r >> 2);
                if (!ok[r]) continue;
                const size_t o = obase + (size_t)dm * p.L;
                float v = ms_apply_act(acc[i][j][r] + bv[r], p.act, p.slope);
                if (Yact) Yact[o] = v;
                if (res) v += rv[r];
                Y[o] = v;
            }
        }
    }
}

// ------------------------------------------------ forward / backward data, row-tile form
// Same contraction as k_igemm_conv, but the activation tile is staged ONCE per channel chunk
// as raw rows with their dilation halo (no im2col duplication, no per-element index math in the
// K loop): a workgroup owns either a BN-long segment of one (b) row (L >= BN) or R = BN / L whole
// rows (short L: the discriminator's 1024->1024 k5 conv at L = 32 / 17 / 9).  The B fragment of
// tap j is the same LDS row read at a +j*dil column offset.  Weights arrive as 16-byte loads of
// the contiguous (c, j) run of each output-channel row; for backward-data they are first
// re-laid-out (transposed + tap-flipped) into the workspace by k_transpose_flip_w.
struct RowP {
    int B, CK, L, M, dil, off0, pad_mode, act, in_act, KG;
    int Lt, R, SS, RSZ, tiles_per_row;   // segment length, rows per tile, LDS segment/row strides
    float slope;
    int CKs;                             // input channels per split-K slice (== CK: no split)
    long long zstride;                   // floats between the partial-output slabs of two slices
    const float* Wfwd;                   // backward data: the forward-layout weights (pipelined kernel reads them directly)
};

template <int WGM, int WGN, int TM, int TN, int K, int CC, bool HAS_ACT, int EPI_S, int IN_S>
__global__ __launch_bounds__(256) void k_conv_mfma_rows(RowP p, const float* __restrict__ X,
                                                       const float* __restrict__ Xact,
                                                       const float* __restrict__ W,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ res,
                                                       float* __restrict__ Y,
                                                       float* __restrict__ Yact) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int KK = CC * K;               // GEMM-K elements per chunk
    constexpr int AS = KK + 1;               // odd LDS stride of the weight tile
    constexpr int A4 = BM * (KK / 4);        // float4 loads per weight tile
    constexpr int RA4 = (A4 + 255) / 256;
    constexpr int MAXCOL = 2;                // LDS columns a loader thread owns (RSZ <= 512)
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(KK % 4 == 0 && KK % 2 == 0, "chunk must be float4- and k-pair-sized");
    extern __shared__ float smem[];
    float* As = smem;                        // [BM][AS]
    float* Xs = smem + BM * AS;              // [CC][RSZ]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5;
    const int wm = wid / WGN, wn = wid % WGN;
    const int m0 = blockIdx.y * BM;
    int b0, t0;
    if (p.R == 1) { b0 = blockIdx.x / p.tiles_per_row; t0 = (blockIdx.x - b0 * p.tiles_per_row) * BN; }
    else { b0 = blockIdx.x * p.R; t0 = 0; }

    // X loader: per-thread LDS columns -> global offset of channel 0 (or -1: zero padding)
    long long goff[MAXCOL];
#pragma unroll
    for (int q = 0; q < MAXCOL; ++q) {
        const int col = tid + q * 256;
        goff[q] = -1;
        if (col < p.RSZ) {
            const int r = col / p.SS, pos = col - r * p.SS;
            const int b = b0 + r;
            if (b < p.B) {
                const int s = ms_src_index(t0 + pos + p.off0, p.L, p.pad_mode);
                // IN_S > 1: channel c' = co*IN_S + r is phase r of a stride-IN_S signal (row length
                // L*IN_S) -- the input of a transposed conv's backward passes
                if (s >= 0) goff[q] = (long long)b * p.CK * p.L + (long long)s * IN_S;
            }
        }
    }
    const bool has_col1 = p.RSZ > 256;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // B-fragment base column of this lane for each N sub-tile (output column -> LDS column)
    int bbase[TN];
    bool nvalid[TN];
    int ob[TN], ot[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * TN * 32 + j * 32 + (lane & 31);
        const int r = nl / p.Lt, tc = nl - r * p.Lt;
        nvalid[j] = r < p.R && b0 + r < p.B && t0 + tc < p.L;
        bbase[j] = nvalid[j] ? r * p.SS + tc : 0;
        ob[j] = b0 + r;
        ot[j] = t0 + tc;
    }

    float4 ra[RA4];
    float rx[MAXCOL][CC];
    auto gload = [&](int c0) {
#pragma unroll
        for (int i = 0; i < RA4; ++i) {
            const int e = i * 256 + tid;
            const int row = e / (KK / 4), q4 = e - row * (KK / 4);
            const bool ok = e < A4 && m0 + row < p.M;
            const size_t off = ok ? (size_t)(m0 + row) * p.KG + (size_t)c0 * K + q4 * 4 : 0;
            float4 v = *reinterpret_cast<const float4*>(W + off);
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            ra[i] = v;
        }
#pragma unroll
        for (int q = 0; q < MAXCOL; ++q) {
            if (q == 1 && !has_col1) break;
            const bool ok = goff[q] >= 0;
            const size_t base = ok ? (size_t)goff[q] + (size_t)(c0 / IN_S) * p.L * IN_S : 0;
            const size_t step = ok ? (size_t)p.L * IN_S : 0;
            float av[CC];
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const size_t off = base + (c / IN_S) * step + (ok ? (c % IN_S) : 0);
                rx[q][c] = X[off];
                if (HAS_ACT) av[c] = Xact[off];
            }
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                float v = rx[q][c];
                if (HAS_ACT) v = ms_act_grad(v, av[c], p.in_act, p.slope);
                rx[q][c] = ok ? v : 0.f;
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < RA4; ++i) {
            const int e = i * 256 + tid;
            if (e < A4) {
                const int row = e / (KK / 4), q4 = e - row * (KK / 4);
                float* d = As + row * AS + q4 * 4;
                d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
            }
        }
#pragma unroll
        for (int q = 0; q < MAXCOL; ++q) {
            if (q == 1 && !has_col1) break;
            const int col = tid + q * 256;
            if (col < p.RSZ) {
#pragma unroll
                for (int c = 0; c < CC; ++c) Xs[c * p.RSZ + col] = rx[q][c];
            }
        }
    };

    // split-K: slice z contracts channels [z*CKs, min((z+1)*CKs, CK)) into its own output slab
    const int cbeg = blockIdx.z * p.CKs;
    const int nchunks = ((cbeg + p.CKs < p.CK ? cbeg + p.CKs : p.CK) - cbeg) / CC;
    Y += (size_t)blockIdx.z * p.zstride;
    gload(cbeg);
    lstore();
    __syncthreads();
    const int arow = (wm * TM * 32 + (lane & 31)) * AS + h;
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        if (more) gload(cbeg + (ch + 1) * CC);
#pragma unroll
        for (int q = 0; q < KK / 2; ++q) {
            constexpr int dummy = 0; (void)dummy;
            const int kk0 = 2 * q, kk1 = 2 * q + 1;
            const int off_lo = (kk0 / K) * p.RSZ + (kk0 % K) * p.dil;
            const int off_hi = (kk1 / K) * p.RSZ + (kk1 % K) * p.dil;
            const int off = h ? off_hi : off_lo;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[arow + i * 32 * AS + 2 * q];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Xs[off + bbase[j]];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            lstore();
            __syncthreads();
        }
    }

    if (EPI_S == 0) {
        // All loads of a 16-row accumulator tile (bias, residual) are issued as one batch from
        // clamped addresses before any store: loads under per-element branches serialise into one
        // memory round trip each (measured: 40 us of a 109 us launch).
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!nvalid[j]) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + wm * TM * 32 + i * 32 + 4 * h;
                const size_t obase = ((size_t)ob[j] * p.M + mb) * p.L + ot[j];
                float bv[16], rv[16];
                bool ok[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    ok[r] = mb + dm < p.M;
                    bv[r] = bias ? bias[ok[r] ? mb + dm : 0] : 0.f;
                }
                if (res) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dm = (r & 3) + 8 * (r >> 2);
                        rv[r] = res[ok[r] ? obase + (size_t)dm * p.L : 0];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = (r & 3) + 8 * (r >> 2);
                    if (!ok[r]) continue;
                    const size_t o = obase + (size_t)dm * p.L;
                    float v = ms_apply_act(acc[i][j][r] + bv[r], p.act, p.slope);
                    if (Yact) Yact[o] = v;
                    if (res) v += rv[r];
                    Y[o] = v;
                }
            }
        }
    } else {
        // transposed conv: GEMM row m' = co*S + r is output phase r of channel co; a lane's 4
        // consecutive accumulator rows are 4 consecutive output samples (S = 8) or 2 x 2 (S = 2),
        // written as one 16-byte / two 8-byte stores along contiguous audio frames
        constexpr int ES = EPI_S > 0 ? EPI_S : 1;     // (EPI_S == 0 never reaches this branch)
        const int Cout = p.M / ES;
        const size_t Lo = (size_t)p.L * ES;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!nvalid[j]) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float bq[4][2];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {   // bias loads batched ahead of the stores
                    const int mb = m0 + wm * TM * 32 + i * 32 + 8 * rg + 4 * h;
                    const int co = (mb < p.M ? mb : 0) / ES;
                    bq[rg][0] = bias ? bias[co] : 0.f;
                    bq[rg][1] = (bias && EPI_S < 4) ? bias[co + 1 < Cout ? co + 1 : co] : 0.f;
                }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int mb = m0 + wm * TM * 32 + i * 32 + 8 * rg + 4 * h;
                    if (mb >= p.M) continue;
                    if (EPI_S >= 4) {
                        const int co = mb / ES, ph = mb - co * ES;
                        const float bv = bq[rg][0];
                        float4 v;
                        v.x = ms_apply_act(acc[i][j][4 * rg + 0] + bv, p.act, p.slope);
                        v.y = ms_apply_act(acc[i][j][4 * rg + 1] + bv, p.act, p.slope);
                        v.z = ms_apply_act(acc[i][j][4 * rg + 2] + bv, p.act, p.slope);
                        v.w = ms_apply_act(acc[i][j][4 * rg + 3] + bv, p.act, p.slope);
                        *reinterpret_cast<float4*>(Y + ((size_t)ob[j] * Cout + co) * Lo + (size_t)ot[j] * ES + ph) = v;
                    } else {
                        const int co = mb / 2;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const float bv = bq[rg][u];
                            float2 v;
                            v.x = ms_apply_act(acc[i][j][4 * rg + 2 * u + 0] + bv, p.act, p.slope);
                            v.y = ms_apply_act(acc[i][j][4 * rg + 2 * u + 1] + bv, p.act, p.slope);
                            *reinterpret_cast<float2*>(Y + ((size_t)ob[j] * Cout + co + u) * Lo + (size_t)ot[j] * 2) = v;
                        }
                    }
                }
            }
        }
    }
}

// Epilogue of a split-K row-tile launch: sums the slices' slabs in slice order (deterministic) and
// applies what the fused epilogue would have: bias (channel = (i / rowlen) % nch), activation,
// the pre-residual copy and the residual add.
__global__ __launch_bounds__(256) void k_rows_split_finish(const float* __restrict__ slabs, int ns,
                                                          size_t zstride,
                                                          const float* __restrict__ bias, int nch,
                                                          int rowlen, int act, float slope,
                                                          const float* __restrict__ res,
                                                          float* __restrict__ Y,
                                                          float* __restrict__ Yact, size_t total) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        float v = slabs[i];
        for (int z = 1; z < ns; ++z) v += slabs[(size_t)z * zstride + i];
        if (bias) v += bias[(i / rowlen) % nch];
        v = ms_apply_act(v, act, slope);
        if (Yact) Yact[i] = v;
        if (res) v += res[i];
        Y[i] = v;
    }
}

// Packed weights of the 3-tap polyphase form of ConvTranspose1d(K = 2S, padding = S/2):
//   y[b, co, q*S + r] = sum_ci sum_{d=-1..1} W[ci, co, r + pad - d*S] * x[b, ci, q + d]
// Wp[(co*S + r)][ci*3 + (d+1)], zero where the tap index falls outside [0, K).
__global__ __launch_bounds__(256) void k_pack_convt_w(const float* __restrict__ W,
                                                     float* __restrict__ Wp, int Cin, int Cout,
                                                     int K, int S, int pad) {
    const size_t total = (size_t)Cout * S * Cin * 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int d = (int)(i % 3) - 1;
        size_t r0 = i / 3;
        const int ci = (int)(r0 % Cin);
        r0 /= Cin;
        const int r = (int)(r0 % S);
        const int co = (int)(r0 / S);
        const int k = r + pad - d * S;
        Wp[i] = (k >= 0 && k < K) ? W[((size_t)ci * Cout + co) * K + k] : 0.f;
    }
}

// Two-live-tap form of the same weights (K = 2S, pad = S/2: d in {-1,0} for r < S/2, {0,+1} above), rows
// in groups of 64 = [32 low-phase | 32 high-phase] rows of the same 64/S output channels:
//   Wp2[g*64 + half*32 + co_l*(S/2) + rh][ci*2 + jj] = W[ci, co, r + pad - (half + jj - 1)*S]
//   with co = g*(64/S) + co_l, r = half*(S/2) + rh
__global__ __launch_bounds__(256) void k_pack_convt_w2(const float* __restrict__ W,
                                                      float* __restrict__ Wp, int Cin, int Cout,
                                                      int K, int S, int pad) {
    const size_t total = (size_t)Cout * S * Cin * 2;
    const int SH = S / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int jj = (int)(i % 2);
        size_t r0 = i / 2;
        const int ci = (int)(r0 % Cin);
        const int m = (int)(r0 / Cin);
        const int g = m / 64, half = (m % 64) / 32, l = m % 32;
        const int co = g * (64 / S) + l / SH, r = half * SH + l % SH;
        const int k = r + pad - (half + jj - 1) * S;
        Wp[i] = (k >= 0 && k < K) ? W[((size_t)ci * Cout + co) * K + k] : 0.f;
    }
}

// Wt[ci][co*K + j'] = W[co][ci][K-1-j']  (weights of the conv that computes backward-data)
__global__ __launch_bounds__(256) void k_transpose_flip_w(const float* __restrict__ W,
                                                         float* __restrict__ Wt, int Co, int Ci,
                                                         int K) {
    const size_t total = (size_t)Co * Ci * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int jp = (int)(i % K);
        const size_t r = i / K;
        const int co = (int)(r % Co);
        const int ci = (int)(r / Co);
        Wt[i] = W[((size_t)co * Ci + ci) * K + (K - 1 - jp)];
    }
}

// Weights of the 3-tap phase-split form of ConvTranspose1d's backward-data:
//   gx[b, ci, q] = sum_{co, r, d} W[ci, co, r + pad + d*S] * G'[b, (co, r), q + d],  G'[b,(co,r),q] = gp[b,co,qS+r]
// Wq[ci][(co*S + r)*3 + (d+1)], zero where the tap index falls outside [0, K).
__global__ __launch_bounds__(256) void k_pack_convt_bwd_w(const float* __restrict__ W,
                                                         float* __restrict__ Wq, int Cin, int Cout,
                                                         int K, int S, int pad) {
    const size_t total = (size_t)Cin * Cout * S * 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int d = (int)(i % 3) - 1;
        size_t r0 = i / 3;
        const int r = (int)(r0 % S);
        r0 /= S;
        const int co = (int)(r0 % Cout);
        const int ci = (int)(r0 / Cout);
        const int k = r + pad + d * S;
        Wq[i] = (k >= 0 && k < K) ? W[((size_t)ci * Cout + co) * K + k] : 0.f;
    }
}

// The same weights with only the two live taps of each phase (K = 2S, pad = S/2: d in {0,+1} for
// r < S/2, {-1,0} above):  Wq2[ci][(co*S + r)*2 + jj] = W[ci, co, r + pad + (jb(r) + jj - 1)*S]
__global__ __launch_bounds__(256) void k_pack_convt_bwd_w2(const float* __restrict__ W,
                                                          float* __restrict__ Wq, int Cin, int Cout,
                                                          int K, int S, int pad) {
    const size_t total = (size_t)Cin * Cout * S * 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int jj = (int)(i % 2);
        size_t r0 = i / 2;
        const int r = (int)(r0 % S);
        r0 /= S;
        const int co = (int)(r0 % Cout);
        const int ci = (int)(r0 / Cout);
        const int d = (r < S / 2 ? 1 : 0) + jj - 1;
        const int k = r + pad + d * S;
        Wq[i] = (k >= 0 && k < K) ? W[((size_t)ci * Cout + co) * K + k] : 0.f;
    }
}

// gw[ci, co, k] = beta*gw + dWq[ci][(co*S + r)*3 + (d+1)] with k = r + pad + d*S
__global__ __launch_bounds__(256) void k_unpack_convt_gw(const float* __restrict__ dWq,
                                                        float* __restrict__ gw, int Cin, int Cout,
                                                        int K, int S, int pad, float beta) {
    const size_t total = (size_t)Cin * Cout * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i % K);
        const size_t cc = i / K;               // ci*Cout + co
        const int kp = k - pad;
        int d = kp >= 0 ? kp / S : -((-kp + S - 1) / S);
        const int r = kp - d * S;
        const float v = dWq[(cc * S + r) * 3 + (d + 1)];
        gw[i] = (beta != 0.f ? beta * gw[i] : 0.f) + v;
    }
}

// ------------------------------------------------------------------ backward weight
// partial[z][m][n] = sum over this block's kk = (b,t) range of A[m,kk] * Bm[kk,n]
//   A[m, kk] = dY'[b, m, t],  Bm[kk, n=(c,j)] = X[b, c, t + j*dil + off0]
// Both operands are contiguous along kk (time), so both LDS tiles are [row][kk].
template <int WGM, int WGN, int TM, int TN, int K>
__global__ __launch_bounds__(256) void k_igemm_wgrad(IgP p, int chunks_per_split,
                                                    const float* __restrict__ X,
                                                    const float* __restrict__ Xact,
                                                    const float* __restrict__ G,
                                                    const float* __restrict__ Gact, int g_act,
                                                    float* __restrict__ partial,
                                                    size_t partial_stride) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int RA = BM * KC / 256, RB = BN * KC / 256;
    static_assert(WGM * WGN == 4, "4 waves");
    __shared__ float As[BM * KCP];
    __shared__ float Bs[BN * KCP];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int NG = p.CK * K;                 // GEMM-N = (c, j) pairs
    const int KT = p.B * p.L;                // GEMM-K = (b, t) pairs
    const int kk_l = tid % KC, row0 = tid / KC;   // loader: fixed kk lane, rows row0 + 16*r

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float asum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) asum[i] = 0.f;

    // per-thread row descriptors are chunk-invariant: A rows m (offset m*L), B rows n = (c, j)
    // (offset c*L, tap shift j*dil + off0 for the padding test)
    int aoff[RA], boff[RB], bsh[RB];
    bool avalid[RA], bvalid[RB];
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int m = m0 + row0 + 16 * r;
        avalid[r] = m < p.M;
        aoff[r] = (avalid[r] ? m : 0) * p.L;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int n = n0 + row0 + 16 * r;
        bvalid[r] = n < NG;
        const int nn = bvalid[r] ? n : 0;
        const int c = nn / K, j = nn - c * K;
        bsh[r] = j * p.dil + p.off0;
        boff[r] = (c / p.in_s) * p.L * p.in_s + (c % p.in_s);
    }

    const int c_begin = blockIdx.z * chunks_per_split;
    int c_end = c_begin + chunks_per_split;
    const int nchunks = (KT + KC - 1) / KC;
    if (c_end > nchunks) c_end = nchunks;

    // absent activation operands alias the data with a pass-through kind (keeps loads branch-free)
    const float* Gq = Gact ? Gact : G;
    const float* Xq = Xact ? Xact : X;
    const int g_kind = Gact ? g_act : MS_ACT_NONE;
    const int x_kind = Xact ? p.in_act : MS_ACT_NONE;
    float ra[RA], rb[RB];
    auto gload = [&](int chunk) {
        const int kg = chunk * KC + kk_l;
        const bool kv = kg < KT;
        const int b = kv ? kg / p.L : 0;
        const int t = kv ? kg - b * p.L : 0;
        const float* Ga = G + (size_t)b * p.M * p.L + t;
        const float* Gy = Gq + (size_t)b * p.M * p.L + t;
        const float* Xb = X + (size_t)b * p.CK * p.L;
        const float* Xy = Xq + (size_t)b * p.CK * p.L;
        float gv[RA], ga[RA], xv[RB], xa[RB];
        bool xok[RB];
#pragma unroll
        for (int r = 0; r < RA; ++r) {   // rows past M read row 0 and are masked below
            gv[r] = Ga[aoff[r]];
            ga[r] = Gy[aoff[r]];
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int sidx = src_index_sel(t + bsh[r], p.L, p.pad_mode == MS_PAD_REFLECT);
            xok[r] = kv && bvalid[r] && (unsigned)sidx < (unsigned)p.L;
            const int off = xok[r] ? boff[r] + sidx * p.in_s : 0;
            xv[r] = Xb[off];
            xa[r] = Xy[off];
        }
#pragma unroll
        for (int r = 0; r < RA; ++r)
            ra[r] = (kv && avalid[r]) ? ms_act_grad(gv[r], ga[r], g_kind, p.slope) : 0.f;
#pragma unroll
        for (int r = 0; r < RB; ++r) rb[r] = xok[r] ? ms_act_grad(xv[r], xa[r], x_kind, p.slope) : 0.f;
    };
    auto lstore = [&]() {
#pragma unroll
        for (int r = 0; r < RA; ++r) As[(row0 + 16 * r) * KCP + kk_l] = ra[r];
#pragma unroll
        for (int r = 0; r < RB; ++r) Bs[(row0 + 16 * r) * KCP + kk_l] = rb[r];
    };

    const int arow = (wm * TM * 32 + (lane & 31)) * KCP + (lane >> 5);
    const int brow = (wn * TN * 32 + (lane & 31)) * KCP + (lane >> 5);
    if (c_begin < c_end) {
        gload(c_begin);
        lstore();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = c + 1 < c_end;
        if (more) gload(c + 1);
#pragma unroll
        for (int k2 = 0; k2 < KC / 2; ++k2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a[i] = As[arow + i * 32 * KCP + 2 * k2];
                asum[i] += a[i];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[brow + j * 32 * KCP + 2 * k2];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            lstore();
            __syncthreads();
        }
    }

    float* part = partial + (size_t)blockIdx.z * partial_stride;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        if (n >= NG) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) part[(size_t)m * NG + n] = acc[i][j][r];
            }
        }
    }
    // bias grad = row sums of A: lanes (i, k=0) and (i, k=1) each saw half of the kk's
    if (blockIdx.x == 0 && wn == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float s = asum[i] + __shfl_xor(asum[i], 32, 64);
            const int m = m0 + wm * TM * 32 + i * 32 + (lane & 31);
            if (lane < 32 && m < p.M) part[(size_t)p.M * NG + m] = s;
        }
    }
}

template <int WGM, int WGN, int TM, int TN, int K>
__global__ __launch_bounds__(256) void k_igemm_wgrad_v4(IgP p, int chunks_per_split,
                                                    const float* __restrict__ X,
                                                    const float* __restrict__ Xact,
                                                    const float* __restrict__ G,
                                                    const float* __restrict__ Gact, int g_act,
                                                    float* __restrict__ partial,
                                                    size_t partial_stride) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    static_assert(WGM * WGN == 4, "4 waves");
    __shared__ float As[BM * KCP];
    __shared__ float Bs[BN * KCP];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int NG = p.CK * K;                 // GEMM-N = (c, j) pairs
    const int KT = p.B * p.L;                // GEMM-K = (b, t) pairs
    // loader: a thread owns 4 consecutive kk (= 4 consecutive time steps of one batch row; the
    // host guarantees L % 4 == 0) of rows row0 + 64*r: 16-byte loads along contiguous audio frames
    const int kq = tid & 3, row0 = tid >> 2;
    constexpr int RA4 = (BM + 63) / 64, RB4 = (BN + 63) / 64;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float asum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) asum[i] = 0.f;

    int aoff[RA4], boff[RB4], bsh[RB4];
    bool avalid[RA4], bvalid[RB4];
#pragma unroll
    for (int r = 0; r < RA4; ++r) {
        const int row = row0 + 64 * r;
        const int m = m0 + row;
        avalid[r] = row < BM && m < p.M;
        aoff[r] = (avalid[r] ? m : 0) * p.L;
    }
#pragma unroll
    for (int r = 0; r < RB4; ++r) {
        const int row = row0 + 64 * r;
        const int n = n0 + row;
        bvalid[r] = row < BN && n < NG;
        const int nn = bvalid[r] ? n : 0;
        const int c = nn / K, j = nn - c * K;
        bsh[r] = j * p.dil + p.off0;
        boff[r] = c * p.L;
    }

    const int c_begin = blockIdx.z * chunks_per_split;
    int c_end = c_begin + chunks_per_split;
    const int nchunks = (KT + KC - 1) / KC;
    if (c_end > nchunks) c_end = nchunks;

    // absent activation operands alias the data with a pass-through kind (keeps loads branch-free)
    const float* Gq = Gact ? Gact : G;
    const float* Xq = Xact ? Xact : X;
    const int g_kind = Gact ? g_act : MS_ACT_NONE;
    const int x_kind = Xact ? p.in_act : MS_ACT_NONE;
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    float4 ra[RA4], rb[RB4];
    auto gload = [&](int chunk) {
        const int kg = chunk * KC + 4 * kq;
        const bool kv = kg < KT;                  // KT % 4 == 0: all 4 or none
        const int b = kv ? kg / p.L : 0;
        const int t = kv ? kg - b * p.L : 0;
        const float* Ga = G + (size_t)b * p.M * p.L + t;
        const float* Gy = Gq + (size_t)b * p.M * p.L + t;
        const float* Xb = X + (size_t)b * p.CK * p.L;
        const float* Xy = Xq + (size_t)b * p.CK * p.L;
        float4 gv[RA4], ga[RA4];
        f4u xv[RB4], xa[RB4];
        int s0[RB4];
#pragma unroll
        for (int r = 0; r < RA4; ++r) {
            gv[r] = *reinterpret_cast<const float4*>(Ga + aoff[r]);
            ga[r] = *reinterpret_cast<const float4*>(Gy + aoff[r]);
        }
#pragma unroll
        for (int r = 0; r < RB4; ++r) {
            s0[r] = t + bsh[r];
            // the 4 taps sit in one row; a window hanging over a row end is re-read element-wise below
            const bool inside = s0[r] >= 0 && s0[r] + 3 < p.L;
            const int off = inside ? boff[r] + s0[r] : boff[r];
            xv[r] = *reinterpret_cast<const f4u*>(Xb + off);
            xa[r] = *reinterpret_cast<const f4u*>(Xy + off);
        }
#pragma unroll
        for (int r = 0; r < RB4; ++r) {
            const bool inside = s0[r] >= 0 && s0[r] + 3 < p.L;
            if (!inside && kv && bvalid[r]) {      // rare: row edges only
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int sidx = s0[r] + i;
                    const bool ok = (unsigned)sidx < (unsigned)p.L;
                    xv[r][i] = ok ? Xb[boff[r] + sidx] : 0.f;
                    xa[r][i] = ok ? Xy[boff[r] + sidx] : 1.f;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RA4; ++r) {
            const bool ok = kv && avalid[r];
            ra[r].x = ok ? ms_act_grad(gv[r].x, ga[r].x, g_kind, p.slope) : 0.f;
            ra[r].y = ok ? ms_act_grad(gv[r].y, ga[r].y, g_kind, p.slope) : 0.f;
            ra[r].z = ok ? ms_act_grad(gv[r].z, ga[r].z, g_kind, p.slope) : 0.f;
            ra[r].w = ok ? ms_act_grad(gv[r].w, ga[r].w, g_kind, p.slope) : 0.f;
        }
#pragma unroll
        for (int r = 0; r < RB4; ++r) {
            const bool ok = kv && bvalid[r];
            rb[r].x = ok ? ms_act_grad(xv[r][0], xa[r][0], x_kind, p.slope) : 0.f;
            rb[r].y = ok ? ms_act_grad(xv[r][1], xa[r][1], x_kind, p.slope) : 0.f;
            rb[r].z = ok ? ms_act_grad(xv[r][2], xa[r][2], x_kind, p.slope) : 0.f;
            rb[r].w = ok ? ms_act_grad(xv[r][3], xa[r][3], x_kind, p.slope) : 0.f;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int r = 0; r < RA4; ++r) {
            const int row = row0 + 64 * r;
            if (row < BM) {
                float* d = As + row * KCP + 4 * kq;
                d[0] = ra[r].x; d[1] = ra[r].y; d[2] = ra[r].z; d[3] = ra[r].w;
            }
        }
#pragma unroll
        for (int r = 0; r < RB4; ++r) {
            const int row = row0 + 64 * r;
            if (row < BN) {
                float* d = Bs + row * KCP + 4 * kq;
                d[0] = rb[r].x; d[1] = rb[r].y; d[2] = rb[r].z; d[3] = rb[r].w;
            }
        }
    };

    const int arow = (wm * TM * 32 + (lane & 31)) * KCP + (lane >> 5);
    const int brow = (wn * TN * 32 + (lane & 31)) * KCP + (lane >> 5);
    if (c_begin < c_end) {
        gload(c_begin);
        lstore();
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = c + 1 < c_end;
        if (more) gload(c + 1);
#pragma unroll
        for (int k2 = 0; k2 < KC / 2; ++k2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a[i] = As[arow + i * 32 * KCP + 2 * k2];
                asum[i] += a[i];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[brow + j * 32 * KCP + 2 * k2];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            lstore();
            __syncthreads();
        }
    }

    float* part = partial + (size_t)blockIdx.z * partial_stride;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        if (n >= NG) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) part[(size_t)m * NG + n] = acc[i][j][r];
            }
        }
    }
    // bias grad = row sums of A: lanes (i, k=0) and (i, k=1) each saw half of the kk's
    if (blockIdx.x == 0 && wn == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float s = asum[i] + __shfl_xor(asum[i], 32, 64);
            const int m = m0 + wm * TM * 32 + i * 32 + (lane & 31);
            if (lane < 32 && m < p.M) part[(size_t)p.M * NG + m] = s;
        }
    }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partial,
                                                     size_t partial_stride, int nsplit,
                                                     size_t wsize, int nbias,
                                                     float* __restrict__ gw,
                                                     float* __restrict__ gb, float beta) {
    // 64 consecutive outputs per workgroup (coalesced 256-byte rows of every slab); the slabs are
    // dealt round-robin to the 4 waves, whose partial sums meet in LDS: fixed order, deterministic
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    const bool ok = i < wsize + (size_t)nbias;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (ok) {
        int z = wv;
        for (; z + 12 < nsplit; z += 16) {
            s0 += partial[(size_t)z * partial_stride + i];
            s1 += partial[(size_t)(z + 4) * partial_stride + i];
            s2 += partial[(size_t)(z + 8) * partial_stride + i];
            s3 += partial[(size_t)(z + 12) * partial_stride + i];
        }
        for (; z < nsplit; z += 4) s0 += partial[(size_t)z * partial_stride + i];
    }
    red[wv][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wv == 0 && ok) {
        const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (i < wsize) gw[i] = (beta != 0.f ? beta * gw[i] : 0.f) + s;
        else if (gb) gb[i - wsize] = (beta != 0.f ? beta * gb[i - wsize] : 0.f) + s;
    }
}

// ------------------------------------------------------------------------ host side
bool dense_same(const ConvP& p) {
    return p.groups == 1 && p.stride == 1 && p.Lout == p.Lin && (p.K == 1 || p.K == 3 || p.K == 5 || p.K == 7) &&
           (long long)p.B * p.Lin < (1LL << 31) && (long long)p.Cin * p.K < (1 << 30);
}

enum Cfg { CFG_128x128, CFG_64x64, CFG_32x256, CFG_32x128 };

Cfg pick_cfg(int M, long long N) {
    if (M <= 32) return CFG_32x256;
    if (M >= 128 && (N / 128) * (M / 128) >= 384) return CFG_128x128;
    return CFG_64x64;
}

void cfg_tile(Cfg c, int* bm, int* bn) {
    if (c == CFG_128x128) { *bm = 128; *bn = 128; }
    else if (c == CFG_64x64) { *bm = 64; *bn = 64; }
    else if (c == CFG_32x128) { *bm = 32; *bn = 128; }
    else { *bm = 32; *bn = 256; }
}

template <int K, bool TRANS>
int launch_conv_k(Cfg cfg, const IgP& p, const float* X, const float* Xact, const float* W,
                  const float* bias, const float* res, float* Y, float* Yact, hipStream_t s) {
    int bm, bn;
    cfg_tile(cfg, &bm, &bn);
    dim3 grid((unsigned)((p.N + bn - 1) / bn), (unsigned)((p.M + bm - 1) / bm));
    if (cfg == CFG_128x128)
        hipLaunchKernelGGL((k_igemm_conv<2, 2, 2, 2, K, TRANS>), grid, dim3(256), 0, s, p, X, Xact, W, bias, res, Y, Yact);
    else if (cfg == CFG_64x64)
        hipLaunchKernelGGL((k_igemm_conv<2, 2, 1, 1, K, TRANS>), grid, dim3(256), 0, s, p, X, Xact, W, bias, res, Y, Yact);
    else
        hipLaunchKernelGGL((k_igemm_conv<1, 4, 1, 2, K, TRANS>), grid, dim3(256), 0, s, p, X, Xact, W, bias, res, Y, Yact);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

template <bool TRANS>
int launch_conv(int K, Cfg cfg, const IgP& p, const float* X, const float* Xact, const float* W,
                const float* bias, const float* res, float* Y, float* Yact, hipStream_t s) {
    if (K == 1) return launch_conv_k<1, TRANS>(cfg, p, X, Xact, W, bias, res, Y, Yact, s);
    if (K == 3) return launch_conv_k<3, TRANS>(cfg, p, X, Xact, W, bias, res, Y, Yact, s);
    if (K == 5) return launch_conv_k<5, TRANS>(cfg, p, X, Xact, W, bias, res, Y, Yact, s);
    if (K == 7) return launch_conv_k<7, TRANS>(cfg, p, X, Xact, W, bias, res, Y, Yact, s);
    return MS_ERR_UNSUPPORTED;
}

struct WgradPlan {
    Cfg cfg;
    int bm, bn, nsplit, cps;
    size_t stride_floats;
};

WgradPlan plan_wgrad(const ConvP& p) {
    WgradPlan q;
    const int NG = p.Cin * p.K;
    const int t128 = ms_ceil_div(p.Cout, 128) * ms_ceil_div(NG, 128);
    (void)t128;
    if (p.Cout <= 32) q.cfg = NG <= 128 ? CFG_32x128 : CFG_32x256;
    else if (p.Cout >= 128 && NG >= 128) q.cfg = CFG_128x128;   // 4 MFMAs per 4 LDS fragment reads
    else q.cfg = CFG_64x64;
    cfg_tile(q.cfg, &q.bm, &q.bn);
    const int tiles = ms_ceil_div(p.Cout, q.bm) * ms_ceil_div(NG, q.bn);
    const long long KT = (long long)p.B * p.Lin;
    const int nchunks = (int)((KT + KC - 1) / KC);
    q.stride_floats = (size_t)p.Cout * NG + p.Cout;
    // split-K: ~2 workgroups per CU, >= 16 chunks per split, partial slabs <= 12 MiB in total
    int ns = ms_ceil_div(512, tiles);
    const int max_by_work = nchunks / 16 > 0 ? nchunks / 16 : 1;
    if (ns > max_by_work) ns = max_by_work;
    const size_t cap = (size_t)24 << 20;
    const size_t max_by_bytes = cap / (q.stride_floats * 4);
    if ((size_t)ns > max_by_bytes) ns = max_by_bytes > 0 ? (int)max_by_bytes : 1;
    if (ns < 1) ns = 1;
    q.cps = ms_ceil_div(nchunks, ns);
    q.nsplit = ms_ceil_div(nchunks, q.cps);
    return q;
}


// ---- row-tile kernel dispatch
enum RowCfg { ROW_128x128, ROW_64x128, ROW_64x64, ROW_32x256, ROW_64x256 };

const char* row_tile_str(RowCfg c) {
    switch (c) {
        case ROW_128x128: return "2, 2, 2, 2";
        case ROW_64x128: return "2, 2, 1, 2";
        case ROW_64x64: return "2, 2, 1, 1";
        case ROW_64x256: return "1, 4, 2, 2";
        default: return "1, 4, 1, 2";
    }
}

constexpr int row_cc(int K) { return K == 7 ? 4 : (K == 1 ? 32 : 8); }

bool rows_applicable(int M, int CK, int K, int L) {
    if (!(K == 1 || K == 3 || K == 5 || K == 7)) return false;
    if (CK % row_cc(K)) return false;
    if (M < 32 && CK < 256) return false;   // (few outputs, deep contraction: the judge conv 1024 -> 1 with split-K)
    return L >= 1;
}

// dense = a plain stride-1 conv with 3 or 5 taps, i.e. a launch the split-bf16 kernel (conv_rows3.hip) may take:
// its 128x128 tile needs 112 KiB of LDS (one workgroup per CU, every latency exposed); two co-resident 64x128
// workgroups measured faster (C = 128, L = 2048, B = 32: 58 vs 68 us)
RowCfg pick_row_cfg(int M, int B, int L, bool dense = false) {
    if (M <= 32) return ROW_32x256;
    if (L < 128) return (M >= 512 && (long long)B * L >= 512) ? ROW_64x128 : ROW_64x64;   // short rows: R = 128 / L rows per tile
    const long long N = (long long)B * L;
    const char* r3 = getenv("MSYNTH_ROWS3");
    const bool split = dense && L % 4 == 0 && !(r3 && atoi(r3) == 0);
    if (M >= 128 && (N / 128) * (M / 128) >= 384 && !split) return ROW_128x128;
    if ((N / 128) * (M / 64) < 192) return ROW_64x64;   // small batches: more, smaller workgroups
    return ROW_64x128;
}

// tile shape of the transposed-conv forward: short rows (the stage-1 2-D transposed convs as lines: L = 16 .. 64,
// thousands of lines) take the 128-column tiling with R = 128 / L rows per tile even for M < 512, so that the paired
// split-bf16 kernel (which owns 128-column tiles) applies
RowCfg convt_fwd_cfg(int M, int B, int L) {
    const RowCfg c = pick_row_cfg(M, B, L);
    if (c == ROW_64x64 && L < 128 && L % 4 == 0 && 128 % L == 0 && M % 64 == 0 && (long long)B * L >= 128 * 256)
        return ROW_64x128;
    return c;
}

void row_tile(RowCfg c, int* bm, int* bn) {
    switch (c) {
        case ROW_128x128: *bm = 128; *bn = 128; break;
        case ROW_64x128: *bm = 64; *bn = 128; break;
        case ROW_64x64: *bm = 64; *bn = 64; break;
        case ROW_64x256: *bm = 64; *bn = 256; break;
        default: *bm = 32; *bn = 256; break;
    }
}

bool make_rowp(RowP* q, RowCfg cfg, int B, int CK, int L, int M, int K, int dil, int off0,
               int pad_mode, int act, int in_act, float slope) {
    int bm, bn;
    row_tile(cfg, &bm, &bn);
    q->B = B; q->CK = CK; q->L = L; q->M = M; q->dil = dil; q->off0 = off0; q->pad_mode = pad_mode;
    q->act = act; q->in_act = in_act; q->KG = CK * K; q->slope = slope;
    const int H = (K - 1) * dil;
    if (L >= bn) { q->Lt = bn; q->R = 1; q->tiles_per_row = (L + bn - 1) / bn; }
    else { q->Lt = L; q->R = bn / L; q->tiles_per_row = 1; }
    q->SS = q->Lt + H;
    q->RSZ = q->R * q->SS;
    q->CKs = CK; q->zstride = 0; q->Wfwd = nullptr;
    return q->RSZ <= 512;
}

// deep contractions (>= 128 input channels) stage two channel chunks per barrier pair
bool row_deep(int K, int CK) { return K != 7 && K != 1 && CK >= 128 && CK % (2 * row_cc(K)) == 0; }

// Does a row-tile launch go to the pipelined kernel of conv_rows2.hip?  (16-byte aligned operands,
// zero padding, plain input rows, activation handling it knows; backward data additionally needs
// the forward-layout weights, which it reads directly.)
bool rows2_pick(RowCfg cfg, int K, int CC, bool has_act, int epi_s, int in_s, const RowP& p, const float* X,
                const float* Xact, const float* W, const float* res, const float* Y, const float* Yact,
                Row2P* q, int* tile, int* am, int* in_s_out = nullptr) {
    if (in_s_out) *in_s_out = in_s;
    if (p.pad_mode != MS_PAD_ZERO || cfg == ROW_64x256) return false;
    if (has_act && p.in_act != MS_ACT_LRELU && p.in_act != MS_ACT_NONE && p.in_act != MS_MOD_LRELU_FWD) return false;
    *am = (has_act && p.in_act == MS_ACT_LRELU) ? (in_s == 1 ? 1 : 2) : 0;
    if (has_act && p.in_act == MS_MOD_LRELU_FWD) *am = 3;      // LeakyReLU in front of the conv, on load
    if (in_s != 1 && *am != 2 && *am != 0) return false;
    const float* Wuse = *am == 1 ? p.Wfwd : W;
    if (!Wuse || (*am == 1 && p.M % 4)) return false;
    if (((((uintptr_t)X) | ((uintptr_t)Wuse) | ((uintptr_t)(Xact ? Xact : X)) | ((uintptr_t)Y) |
          ((uintptr_t)(Yact ? Yact : Y)) | ((uintptr_t)(res ? res : X))) & 15) != 0)
        return false;
    if ((long long)p.B * p.CK * p.L >= (1LL << 31) || (long long)p.M * p.KG >= (1LL << 31)) return false;
    q->B = p.B; q->CK = p.CK; q->L = p.L; q->M = p.M; q->dil = p.dil; q->off0 = p.off0; q->act = p.act;
    q->KG = p.KG; q->Lt = p.Lt; q->R = p.R; q->SS = p.SS; q->RSZ = p.RSZ; q->tiles_per_row = p.tiles_per_row;
    q->PX = p.RSZ; q->CKs = p.CKs; q->zstride = p.zstride; q->slope = p.slope; q->scratch_off = 0;
    *tile = cfg == ROW_128x128 ? MSR2_128x128 : (cfg == ROW_64x128 ? MSR2_64x128 :
            (cfg == ROW_64x64 ? MSR2_64x64 : MSR2_32x256));
    if (in_s == 1 && msr3_supported(*tile, K, *am, epi_s, *q, 1)) return true;   // split-bf16 kernel: rows of any length
    // short rows of a length that is not a multiple of 4 (k5 conv at L = 17 / 9): whole-row staging
    if (in_s == 1 && in_s_out && K == 5 && p.tiles_per_row == 1 && p.Lt == p.L) {
        int bm, bn;
        row_tile(cfg, &bm, &bn);
        if (p.L < bn && p.L % 4 != 0 && msr2_supported(*tile, K, CC, *am, epi_s, *q, 0)) {
            *in_s_out = 0;
            return true;
        }
    }
    return msr3_supported(*tile, K, *am, epi_s, *q, in_s) || msr2_supported(*tile, K, CC, *am, epi_s, *q, in_s);
}

// Paired eight-wave split-bf16 kernel (k_conv_rows3p): rows per workgroup (128 / 64) when the launch should take
// it -- supported and its grid (half as many, twice as wide workgroups) still fills most of the 256 CUs -- else 0.
int rows3p_bm(const Row2P& q, int bn, int K, int am, int epi_s, int in_s, unsigned gz) {
    if (bn != 128) return 0;                     // (q's tiling must be the 128-column one the kernel's groups own)
    constexpr int min_wgs = 128;
    const long long ntiles = q.R == 1 ? (long long)q.B * q.tiles_per_row : (q.B + q.R - 1) / q.R;
    // (C = 256 at L = 256, B = 32: 64 workgroups of 128 rows or 128 of 64 rows both measured slower than the
    //  four-wave kernel's 256 workgroups: 61 / 44 vs 30 us -- no fallback to narrower workgroups)
    const int bm = (q.M >= 128 && K == 3) ? 128 : (q.M <= 32 ? 32 : 64);
    if (!msr3p_supported(bm, K, am, epi_s, q, in_s)) return 0;
    const long long wgs = ((ntiles + 1) / 2) * ((q.M + bm - 1) / bm) * gz;
    return wgs >= min_wgs ? bm : 0;
}

// ... and for the transposed-conv forward (two-tap form, conv_rows3.hip HS): 128 rows where that still fills the chip
int rows3p_convt_bm(const Row2P& q, int bn, int S, unsigned gz) {
    if (bn != 128) return 0;
    constexpr int min_wgs = 128;
    const long long ntiles = q.R == 1 ? (long long)q.B * q.tiles_per_row : (q.B + q.R - 1) / q.R;
    for (int bm = 128; bm >= 64; bm >>= 1) {
        if (q.M < bm || !msr3p_convt_supported(bm, S, q)) continue;
        if (((ntiles + 1) / 2) * ((q.M + bm - 1) / bm) * gz >= min_wgs) return bm;
    }
    return 0;
}

Row2P rows3p_retile(const Row2P& q, int K) {
    Row2P r = q;
    r.Lt = 128;
    r.tiles_per_row = (q.L + 127) / 128;
    r.SS = 128 + (K - 1) * q.dil;
    r.RSZ = r.SS;
    r.PX = r.SS;
    return r;
}

template <int K, bool HAS_ACT, int EPI_S = 0, int IN_S = 1, int CCMUL = 1>
int launch_rows_k(RowCfg cfg, const RowP& p, const float* X, const float* Xact, const float* W,
                  const float* bias, const float* res, float* Y, float* Yact, hipStream_t s) {
    constexpr int CC = row_cc(K) * CCMUL;
    int bm, bn;
    row_tile(cfg, &bm, &bn);
    const unsigned gx = p.R == 1 ? (unsigned)(p.B * p.tiles_per_row) : (unsigned)((p.B + p.R - 1) / p.R);
    dim3 grid(gx, (unsigned)((p.M + bm - 1) / bm), (unsigned)((p.CK + p.CKs - 1) / p.CKs));
    // pipelined second-generation kernel where its requirements hold (conv_rows2.hip)
    {
        Row2P q;
        int tile, am, in_s_eff;
        if (rows2_pick(cfg, K, CC, HAS_ACT, EPI_S, IN_S, p, X, Xact, W, res, Y, Yact, &q, &tile, &am, &in_s_eff)) {
            if (const int bmp = rows3p_bm(q, bn, K, am, EPI_S, in_s_eff, grid.z))
                return msr3p_launch(bmp, K, am, q, X, Xact, am == 1 ? p.Wfwd : W, bias, res, Y, Yact, grid.z, s);
            if (cfg == ROW_32x256 && q.R == 1 && q.Lt == 256) {      // 32-channel layers: the same rows as 128-column tiles
                Row2P q2 = rows3p_retile(q, K);
                if (const int bmp = rows3p_bm(q2, 128, K, am, EPI_S, in_s_eff, grid.z))
                    return msr3p_launch(bmp, K, am, q2, X, Xact, am == 1 ? p.Wfwd : W, bias, res, Y, Yact, grid.z, s);
            }
            if (msr3_supported(tile, K, am, EPI_S, q, in_s_eff))      // split-bf16 matrix pipe (conv_rows3.hip)
                return msr3_launch(tile, K, am, q, X, Xact, am == 1 ? p.Wfwd : W, bias, res, Y, Yact, grid.x, grid.y,
                                   grid.z, s);
            return msr2_launch(tile, K, CC, am, EPI_S, q, X, Xact, am == 1 ? p.Wfwd : W, bias, res, Y, Yact, grid.x,
                               grid.y, grid.z, s, in_s_eff);
        }
    }
    const size_t lds = (size_t)(bm * (CC * K + 1) + CC * p.RSZ) * sizeof(float);
    if (lds > 64 * 1024) return MS_ERR_UNSUPPORTED;
    switch (cfg) {
        case ROW_128x128:
            hipLaunchKernelGGL((k_conv_mfma_rows<2, 2, 2, 2, K, CC, HAS_ACT, EPI_S, IN_S>), grid, dim3(256), lds, s, p, X, Xact, W, bias, res, Y, Yact);
            break;
        case ROW_64x128:
            hipLaunchKernelGGL((k_conv_mfma_rows<2, 2, 1, 2, K, CC, HAS_ACT, EPI_S, IN_S>), grid, dim3(256), lds, s, p, X, Xact, W, bias, res, Y, Yact);
            break;
        case ROW_64x64:
            hipLaunchKernelGGL((k_conv_mfma_rows<2, 2, 1, 1, K, CC, HAS_ACT, EPI_S, IN_S>), grid, dim3(256), lds, s, p, X, Xact, W, bias, res, Y, Yact);
            break;
        case ROW_64x256:
            hipLaunchKernelGGL((k_conv_mfma_rows<1, 4, 2, 2, K, CC, HAS_ACT, EPI_S, IN_S>), grid, dim3(256), lds, s, p, X, Xact, W, bias, res, Y, Yact);
            break;
        default:
            hipLaunchKernelGGL((k_conv_mfma_rows<1, 4, 1, 2, K, CC, HAS_ACT, EPI_S, IN_S>), grid, dim3(256), lds, s, p, X, Xact, W, bias, res, Y, Yact);
            break;
    }
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int launch_rows(int K, RowCfg cfg, const RowP& p, const float* X, const float* Xact, const float* W,
                const float* bias, const float* res, float* Y, float* Yact, hipStream_t s) {
    const bool deep = row_deep(K, p.CK);
#define MS_ROWS(KK, ACT)                                                                              \
    (deep ? launch_rows_k<KK, ACT, 0, 1, 2>(cfg, p, X, Xact, W, bias, res, Y, Yact, s)                \
          : launch_rows_k<KK, ACT, 0, 1, 1>(cfg, p, X, Xact, W, bias, res, Y, Yact, s))
    if (Xact) {
        if (K == 1) return launch_rows_k<1, true>(cfg, p, X, Xact, W, bias, res, Y, Yact, s);
        if (K == 3) return MS_ROWS(3, true);
        if (K == 5) return MS_ROWS(5, true);
        if (K == 7) return launch_rows_k<7, true>(cfg, p, X, Xact, W, bias, res, Y, Yact, s);
    } else {
        if (K == 1) return launch_rows_k<1, false>(cfg, p, X, Xact, W, bias, res, Y, Yact, s);
        if (K == 3) return MS_ROWS(3, false);
        if (K == 5) return MS_ROWS(5, false);
        if (K == 7) return launch_rows_k<7, false>(cfg, p, X, Xact, W, bias, res, Y, Yact, s);
    }
#undef MS_ROWS
    return MS_ERR_UNSUPPORTED;
}

// name of the kernel a row-tile launch resolves to (16-byte aligned tensors assumed): the pipelined
// second generation where its requirements hold (conv_rows2.hip), else the first
const char* row_kname(RowCfg c, int K, bool act, int CK, int L = 0, int R = 1, int SS = 0, int pad_mode = MS_PAD_ZERO,
                      int in_act = MS_ACT_LRELU, int epi_s = 0, int B = 0, int M = 0) {
    static thread_local char buf[96];
    const char* tile = row_tile_str(c);
    const int CC = epi_s ? row_cc(K) : row_cc(K) * (row_deep(K, CK) ? 2 : 1);
    const int am = (act && in_act == MS_ACT_LRELU) ? 1 : 0;
    if (L > 0 && c != ROW_64x256 && pad_mode == MS_PAD_ZERO && (!act || in_act == MS_ACT_LRELU || in_act == MS_ACT_NONE)) {
        Row2P q;
        q.L = L; q.R = R; q.SS = SS;
        const int t2 = c == ROW_128x128 ? MSR2_128x128 : (c == ROW_64x128 ? MSR2_64x128 : (c == ROW_64x64 ? MSR2_64x64 : MSR2_32x256));
        int bm, bn;
        row_tile(c, &bm, &bn);
        {
            Row2P h = q;
            h.B = B; h.M = M > 0 ? M : 64; h.CK = CK; h.CKs = CK; h.PX = R * SS; h.Lt = L >= bn ? bn : L;
            h.tiles_per_row = L >= bn ? (L + bn - 1) / bn : 1;
            h.dil = K > 1 ? (SS - h.Lt) / (K - 1) : 1;        // SS = Lt + (K - 1) dil (rows3p_retile rebuilds it)
            if (h.dil < 1) h.dil = 1;
            if (const int bmp = B > 0 ? rows3p_bm(h, bn, K, am, epi_s, 1, 1) : 0) {
                snprintf(buf, sizeof(buf), "k_conv_rows3p<%s, %d, %d>", bmp == 128 ? "2, 2, 2" : (bmp == 64 ? "2, 1, 2" : "1, 1, 1"), K, am);
                return buf;
            }
            if (c == ROW_32x256 && R == 1 && L >= 256 && B > 0) {
                Row2P h2 = h;
                h2.Lt = 256;
                h2 = rows3p_retile(h2, K);
                if (const int bmp = rows3p_bm(h2, 128, K, am, epi_s, 1, 1)) {
                    snprintf(buf, sizeof(buf), "k_conv_rows3p<%s, %d, %d>", bmp == 128 ? "2, 2, 2" : (bmp == 64 ? "2, 1, 2" : "1, 1, 1"), K, am);
                    return buf;
                }
            }
            if (msr3_supported(t2, K, am, epi_s, h)) {
                snprintf(buf, sizeof(buf), "k_conv_rows3<%s, %d, %d, %s>", tile, K, am,
                         (L % 4 == 0 && h.Lt % 4 == 0) ? "true" : "false");
                return buf;
            }
        }
        if (K == 5 && CC == 16 && epi_s == 0 && am == 1 && L < bn && L % 4 != 0 && c != ROW_32x256) {   // short-row mode
            snprintf(buf, sizeof(buf), "k_conv_rows2<%s, 5, 16, %d, 0, 0>", tile, am);
            return buf;
        }
        if (msr2_supported(t2, K, CC, am, epi_s, q)) {
            snprintf(buf, sizeof(buf), "k_conv_rows2<%s, %d, %d, %d, %d>", tile, K, CC, am, epi_s);
            return buf;
        }
    }
    if (epi_s) snprintf(buf, sizeof(buf), "k_conv_mfma_rows<%s, 3, 8, false, %d, 1>", tile, epi_s);
    else snprintf(buf, sizeof(buf), "k_conv_mfma_rows<%s, %d, %d, %s, 0, 1>", tile, K, CC, act ? "true" : "false");
    return buf;
}

// ---- split-K for row-tile launches whose (M, N) tiling alone leaves most of the 256 CUs idle
// (the discriminator's 1024->1024 k5 conv at L = 32 / 17 / 9, the generator's first transposed
// conv): the contraction is cut into ns channel slices (grid.z), each slice writes a raw partial
// slab into the workspace and k_rows_split_finish sums them in slice order and applies the epilogue.
struct RowSplit {
    int ns, cks;
    size_t out_floats;
};

int rows_wgs(RowCfg cfg, const RowP& p) {
    int bm, bn;
    row_tile(cfg, &bm, &bn);
    const int gx = p.R == 1 ? p.B * p.tiles_per_row : (p.B + p.R - 1) / p.R;
    return gx * ((p.M + bm - 1) / bm);
}

int split_max_wgs() {
    const char* e = getenv("MSYNTH_SPLIT_WGS");   // tuning / test switch (0 disables split-K)
    return e ? atoi(e) : 192;
}

RowSplit plan_rows_split(RowCfg cfg, const RowP& p, int CC) {
    RowSplit q;
    q.ns = 1; q.cks = p.CK; q.out_floats = (size_t)p.B * p.M * p.L;
    const int wgs = rows_wgs(cfg, p);
    const int nchunks = p.CK / CC;
    // (one workgroup per CU is still under-filled when the contraction is long enough to amortise the
    // slab pass: the 1024 -> 1024 k5 conv at B*L = 2048 has 256 tiles and 64 chunks)
    const int lim = nchunks >= 32 ? (split_max_wgs() * 4) / 3 : split_max_wgs();
    if (wgs > lim || nchunks < 8) return q;
    // slices so that tiles * slices fills the 512 resident workgroup slots (2 per CU) without spilling
    // into a mostly empty second round: 160 tiles x 4 slices = 640 workgroups ran 63 % longer than
    // 128 x 4 = 512 (k5 conv at L = 17 vs 16); 160 x 3 = 480 fits
    int ns = 1;
    double best = 0.0;
    const int ns_max = nchunks / 4 < 16 ? nchunks / 4 : 16;
    for (int c = 1; c <= ns_max; ++c) {
        const int tot = wgs * c, rounds = ms_ceil_div(tot, 512);
        const double eff = (double)tot / (rounds * 512.0);
        if (eff > best + 1e-9) { best = eff; ns = c; }
    }
    while (ns > 1 && (size_t)ns * q.out_floats * sizeof(float) > ((size_t)64 << 20)) --ns;
    if (ns <= 1) return q;
    q.cks = ms_ceil_div(nchunks, ns) * CC;
    if (p.CK % 16 == 0 && q.cks % 16) q.cks += 16 - q.cks % 16;   // slices in whole 16-channel chunks (conv_rows3.hip)
    q.ns = ms_ceil_div(p.CK, q.cks);
    return q;
}

size_t rows_split_ws(RowCfg cfg, const RowP& p, int CC) {
    const RowSplit q = plan_rows_split(cfg, p, CC);
    return q.ns > 1 ? (size_t)q.ns * q.out_floats * sizeof(float) : 0;
}

// launch(rowp, bias, res, Y, Yact) runs the row-tile kernel; nch / rowlen describe the bias index
template <class F>
int rows_maybe_split(RowCfg cfg, const RowP& r, int CC, int nch, int rowlen, const float* bias,
                     const float* res, float* Y, float* Yact, void* slab_ws, size_t slab_bytes,
                     hipStream_t s, F launch) {
    const RowSplit q = plan_rows_split(cfg, r, CC);
    if (q.ns <= 1 || !slab_ws || slab_bytes < (size_t)q.ns * q.out_floats * sizeof(float))
        return launch(r, bias, res, Y, Yact);
    RowP z = r;
    z.CKs = q.cks; z.zstride = (long long)q.out_floats; z.act = MS_ACT_NONE;
    float* slabs = (float*)slab_ws;
    const int rc = launch(z, (const float*)nullptr, (const float*)nullptr, slabs, (float*)nullptr);
    if (rc != MS_OK) return rc;
    unsigned nb = (unsigned)((q.out_floats + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_rows_split_finish, dim3(nb), dim3(256), 0, s, slabs, q.ns, q.out_floats,
                       bias, nch, rowlen, r.act, r.slope, res, Y, Yact, q.out_floats);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int rows_cc_eff(int K, int CK) { return row_cc(K) * (row_deep(K, CK) ? 2 : 1); }

size_t align16(size_t n) { return (n + 15) & ~(size_t)15; }

bool rows_ok(const ConvP& p, bool bwd) {
    const int M = bwd ? p.Cin : p.Cout, CK = bwd ? p.Cout : p.Cin;
    if (!rows_applicable(M, CK, p.K, p.Lin)) return false;
    if (((size_t)CK * p.K) % 4) return false;
    RowP q;
    return make_rowp(&q, pick_row_cfg(M, p.B, p.Lin), p.B, CK, p.Lin, M, p.K, p.dil, 0, 0, 0, 0, 0.f);
}

const char* kname(const char* kernel, Cfg c, int K, const char* tail) {
    static thread_local char buf[96];
    const char* tile = c == CFG_128x128 ? "2, 2, 2, 2" : (c == CFG_64x64 ? "2, 2, 1, 1" :
                       (c == CFG_32x128 ? "1, 4, 1, 1" : "1, 4, 1, 2"));
    snprintf(buf, sizeof(buf), "%s<%s, %d%s>", kernel, tile, K, tail);
    return buf;
}

}  // namespace

bool msm_fwd_applicable(const ConvP& p) {
    return dense_same(p) && (p.Cout >= 32 || p.Cin >= 256) && p.Cin * p.K >= 32;
}
bool msm_bwd_data_applicable(const ConvP& p) {
    return dense_same(p) && p.pad_mode == MS_PAD_ZERO && p.Cin >= 32 && p.Cout * p.K >= 32;
}
bool msm_bwd_weight_applicable(const ConvP& p) {
    const bool kok = p.K == 1 || p.K == 3 || p.K == 5 || p.K == 7 || p.K == 15;
    return p.groups == 1 && p.stride == 1 && p.Lout == p.Lin && kok &&
           (long long)p.B * p.Lin < (1LL << 31) && (long long)p.B * p.Cout * p.Lin < (1LL << 31) &&
           (long long)p.B * p.Cin * p.Lin < (1LL << 31) && p.Cout >= 16 && p.Cin * p.K >= 15;
}
// p = mirrored conv of the transposed conv: Cin_T = p.Cout, Cout_T = p.Cin, Lin_T = p.Lout
bool msm_convt_fwd_applicable(const ConvP& p) {
    const int S = p.stride;
    if (!(S == 2 || S == 8) || p.K != 2 * S || 2 * p.pad != S || p.dil != 1 || p.groups != 1) return false;
    if (p.Cout % 8 || p.Cin < 32 || (p.Cin * S) % 32) return false;
    if ((long long)p.B * p.Lin >= (1LL << 31)) return false;
    RowP q;
    return make_rowp(&q, convt_fwd_cfg(p.Cin * S, p.B, p.Lout), p.B, p.Cout, p.Lout, p.Cin * S, 3, 1, -1, 0, 0, 0, 0.f);
}
size_t msm_fwd_ws(const ConvP& p) {
    if (!rows_ok(p, false)) return 0;
    RowP r;
    const RowCfg cfg = pick_row_cfg(p.Cout, p.B, p.Lin, p.K == 3 || p.K == 5);
    make_rowp(&r, cfg, p.B, p.Cin, p.Lin, p.Cout, p.K, p.dil, -p.pad, p.pad_mode, 0, 0, 0.f);
    return rows_split_ws(cfg, r, rows_cc_eff(p.K, p.Cin));
}
size_t msm_bwd_data_ws(const ConvP& p) {
    if (!rows_ok(p, true)) return 0;
    RowP r;
    const RowCfg cfg = pick_row_cfg(p.Cin, p.B, p.Lin, p.K == 3 || p.K == 5);
    make_rowp(&r, cfg, p.B, p.Cout, p.Lin, p.Cin, p.K, p.dil, 0, 0, 0, 0, 0.f);
    return align16((size_t)p.Cin * p.Cout * p.K * sizeof(float)) + rows_split_ws(cfg, r, rows_cc_eff(p.K, p.Cout));
}
size_t msm_bwd_weight_ws(const ConvP& p) {
    const WgradPlan q = plan_wgrad(p);
    return (size_t)q.nsplit * q.stride_floats * sizeof(float);
}
size_t msm_convt_fwd_ws(const ConvP& p) {
    RowP r;
    const RowCfg cfg = convt_fwd_cfg(p.Cin * p.stride, p.B, p.Lout);
    make_rowp(&r, cfg, p.B, p.Cout, p.Lout, p.Cin * p.stride, 3, 1, -1, 0, 0, 0, 0.f);
    return align16((size_t)p.Cin * p.stride * p.Cout * 3 * sizeof(float)) + rows_split_ws(cfg, r, row_cc(3));
}

const char* msm_fwd_name(const ConvP& p) {
    if (rows_ok(p, false)) {
        RowP r;
        const RowCfg cfg = pick_row_cfg(p.Cout, p.B, p.Lin, p.K == 3 || p.K == 5);
        make_rowp(&r, cfg, p.B, p.Cin, p.Lin, p.Cout, p.K, p.dil, -p.pad, p.pad_mode, 0, 0, 0.f);
        return row_kname(cfg, p.K, p.in_act != 0, p.Cin, p.Lin, r.R, r.SS, p.pad_mode, p.in_act ? MS_MOD_LRELU_FWD : 0, 0,
                         p.B, p.Cout);
    }
    return kname("k_igemm_conv", pick_cfg(p.Cout, (long long)p.B * p.Lin), p.K, ", false");
}
const char* msm_bwd_data_name(const ConvP& p) {
    if (rows_ok(p, true)) {
        RowP r;
        const RowCfg cfg = pick_row_cfg(p.Cin, p.B, p.Lin, p.K == 3 || p.K == 5);
        make_rowp(&r, cfg, p.B, p.Cout, p.Lin, p.Cin, p.K, p.dil, 0, 0, 0, 0, 0.f);
        return row_kname(cfg, p.K, p.act != MS_ACT_NONE, p.Cout, p.Lin, r.R, r.SS, MS_PAD_ZERO, p.act, 0, p.B, p.Cin);
    }
    return kname("k_igemm_conv", pick_cfg(p.Cin, (long long)p.B * p.Lin), p.K, ", true");
}
const char* msm_bwd_weight_name(const ConvP& p) {
    const bool v4 = (p.Lin % 4 == 0) && p.pad_mode == MS_PAD_ZERO;
    return kname(v4 ? "k_igemm_wgrad_v4" : "k_igemm_wgrad", plan_wgrad(p).cfg, p.K, "");
}
const char* msm_convt_fwd_name(const ConvP& p) {
    static thread_local char buf[96];
    const RowCfg c = convt_fwd_cfg(p.Cin * p.stride, p.B, p.Lout);
    RowP r;
    make_rowp(&r, c, p.B, p.Cout, p.Lout, p.Cin * p.stride, 3, 1, -1, 0, 0, 0, 0.f);
    {   // the paired split-bf16 kernel, decided exactly as msm_convt1d_fwd does (dummy 16-byte aligned pointers)
        RowP r2 = r;
        if (p.in_act) r2.in_act = MS_MOD_LRELU_FWD;
        const float* D = reinterpret_cast<const float*>(uintptr_t(64));
        Row2P q;
        int tile = 0, am = 0, bm = 0, bn = 0;
        if (rows2_pick(c, 2, 8, p.in_act != 0, p.stride, 1, r2, D, p.in_act ? D : nullptr, D, nullptr, D, nullptr, &q, &tile, &am)) {
            q.KG = r.CK * 2;
            const RowSplit sp = plan_rows_split(c, r, row_cc(3));
            if (sp.ns > 1) q.CKs = sp.cks;
            const unsigned gz = (unsigned)((q.CK + q.CKs - 1) / q.CKs);
            row_tile(c, &bm, &bn);
            int b3 = rows3p_convt_bm(q, bn, p.stride, gz);
            if (!b3 && c == ROW_32x256 && q.R == 1 && q.Lt == 256) b3 = rows3p_convt_bm(rows3p_retile(q, 3), 128, p.stride, gz);
            if (b3) {
                snprintf(buf, sizeof(buf), "k_conv_rows3p<2, %d, 2, 3, 0, %d, %s>", b3 == 128 ? 2 : 1, p.stride,
                         p.in_act ? "true" : "false");
                return buf;
            }
        }
    }
    if (c == ROW_128x128 || c == ROW_64x128) {
        Row2P q;
        q.L = p.Lout; q.R = r.R; q.SS = r.SS; q.M = p.Cin * p.stride;
        const int am = p.in_act ? 3 : 0;
        if (msr2_supported(c == ROW_128x128 ? MSR2_128x128 : MSR2_64x128, 2, 8, am, p.stride, q)) {
            snprintf(buf, sizeof(buf), "k_conv_rows2<%s, 2, 8, %d, %d>", c == ROW_128x128 ? "2, 2, 2, 2" : "1, 4, 2, 1", am, p.stride);
            return buf;
        }
    }
    return row_kname(c, 3, p.in_act != 0, p.Cout, p.Lout, r.R, r.SS, MS_PAD_ZERO, p.in_act ? MS_MOD_LRELU_FWD : 0, p.stride);
}

int msm_conv1d_fwd(const ConvP& p, const float* x, const float* x_act, int x_act_kind,
                   const float* w, const float* bias, const float* residual, float* y,
                   float* y_act, void* ws, size_t ws_bytes, hipStream_t s) {
    if (rows_ok(p, false) && (((uintptr_t)w) & 15) == 0) {
        RowP r;
        const RowCfg cfg = pick_row_cfg(p.Cout, p.B, p.Lin, p.K == 3 || p.K == 5);
        make_rowp(&r, cfg, p.B, p.Cin, p.Lin, p.Cout, p.K, p.dil, -p.pad, p.pad_mode, p.act,
                  x_act_kind, p.slope);
        const int K = p.K;
        return rows_maybe_split(cfg, r, rows_cc_eff(K, p.Cin), p.Cout, p.Lin, bias, residual, y, y_act,
                                ws, ws_bytes, s,
                                [&](const RowP& rp, const float* b_, const float* r_, float* y_, float* ya_) {
                                    return launch_rows(K, cfg, rp, x, x_act, w, b_, r_, y_, ya_, s);
                                });
    }
    IgP q;
    q.B = p.B; q.CK = p.Cin; q.L = p.Lin; q.M = p.Cout; q.dil = p.dil; q.off0 = -p.pad;
    q.pad_mode = p.pad_mode; q.act = p.act; q.in_act = x_act_kind; q.slope = p.slope;
    q.N = p.B * p.Lin; q.KG = p.Cin * p.K; q.in_s = 1;
    return launch_conv<false>(p.K, pick_cfg(q.M, q.N), q, x, x_act, w, bias, residual, y, y_act, s);
}

int msm_conv1d_bwd_data(const ConvP& p, const float* gy, const float* y_act, const float* w,
                        const float* gx_add, float* gx, void* ws, size_t ws_bytes, hipStream_t s) {
    if (rows_ok(p, true) && ws && ws_bytes >= msm_bwd_data_ws(p) && (((uintptr_t)ws) & 15) == 0) {
        float* wt = (float*)ws;
        const size_t total = (size_t)p.Cin * p.Cout * p.K;
        const size_t wbytes = align16(total * sizeof(float));
        RowP r;
        const RowCfg cfg = pick_row_cfg(p.Cin, p.B, p.Lin, p.K == 3 || p.K == 5);
        make_rowp(&r, cfg, p.B, p.Cout, p.Lin, p.Cin, p.K, p.dil, p.pad - (p.K - 1) * p.dil,
                  MS_PAD_ZERO, MS_ACT_NONE, p.act, p.slope);
        r.Wfwd = w;
        {
            Row2P q2;
            int tile2, am2;
            int ins2;
            const bool direct = y_act && rows2_pick(cfg, p.K, rows_cc_eff(p.K, p.Cout), true, 0, 1, r, gy, y_act, wt,
                                                    gx_add, gx, nullptr, &q2, &tile2, &am2, &ins2) && am2 == 1;
            if (!direct) {     // first-generation kernel: weights re-laid-out (transposed + tap-flipped)
                unsigned nb = (unsigned)((total + 255) / 256);
                if (nb > 2048) nb = 2048;
                hipLaunchKernelGGL(k_transpose_flip_w, dim3(nb), dim3(256), 0, s, w, wt, p.Cout, p.Cin, p.K);
                MS_CHECK_LAUNCH();
            }
        }
        const int K = p.K;
        return rows_maybe_split(cfg, r, rows_cc_eff(K, p.Cout), p.Cin, p.Lin, nullptr, gx_add, gx, nullptr,
                                (char*)ws + wbytes, ws_bytes - wbytes, s,
                                [&](const RowP& rp, const float* b_, const float* r_, float* y_, float* ya_) {
                                    return launch_rows(K, cfg, rp, gy, y_act, wt, b_, r_, y_, ya_, s);
                                });
    }
    IgP q;
    q.B = p.B; q.CK = p.Cout; q.L = p.Lin; q.M = p.Cin; q.dil = p.dil;
    q.off0 = p.pad - (p.K - 1) * p.dil;   // flipped taps: j' = K-1-j
    q.pad_mode = MS_PAD_ZERO; q.act = MS_ACT_NONE; q.in_act = p.act; q.slope = p.slope;
    q.N = p.B * p.Lin; q.KG = p.Cout * p.K; q.in_s = 1;
    return launch_conv<true>(p.K, pick_cfg(q.M, q.N), q, gy, y_act, w, nullptr, gx_add, gx, nullptr, s);
}

int msm_conv1d_bwd_weight(const ConvP& p, const float* x, const float* x_act, int x_act_kind,
                          const float* gy, const float* y_act, int y_act_kind, float* gw,
                          float* gb, float beta, void* ws, size_t ws_bytes, hipStream_t s) {
    const WgradPlan pl = plan_wgrad(p);
    const size_t need = (size_t)pl.nsplit * pl.stride_floats * sizeof(float);
    if (!ws || ws_bytes < need) return MS_ERR_WORKSPACE;
    IgP q;
    q.B = p.B; q.CK = p.Cin; q.L = p.Lin; q.M = p.Cout; q.dil = p.dil; q.off0 = -p.pad;
    q.pad_mode = p.pad_mode; q.act = MS_ACT_NONE; q.in_act = x_act_kind; q.slope = p.slope;
    q.N = p.B * p.Lin; q.KG = p.Cin * p.K; q.in_s = 1;
    const int NG = p.Cin * p.K;
    dim3 grid((unsigned)ms_ceil_div(NG, pl.bn), (unsigned)ms_ceil_div(p.Cout, pl.bm), (unsigned)pl.nsplit);
    float* partial = (float*)ws;
    const bool v4 = (p.Lin % 4 == 0) && p.pad_mode == MS_PAD_ZERO &&
                    ((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)(y_act ? y_act : gy))) & 15) == 0;
#define MS_WG(WGM, WGN, TM, TN, KK)                                                                 \
    do {                                                                                            \
        if (v4)                                                                                     \
            hipLaunchKernelGGL((k_igemm_wgrad_v4<WGM, WGN, TM, TN, KK>), grid, dim3(256), 0, s, q,  \
                               pl.cps, x, x_act, gy, y_act, y_act_kind, partial, pl.stride_floats); \
        else                                                                                        \
            hipLaunchKernelGGL((k_igemm_wgrad<WGM, WGN, TM, TN, KK>), grid, dim3(256), 0, s, q,     \
                               pl.cps, x, x_act, gy, y_act, y_act_kind, partial, pl.stride_floats); \
    } while (0)
#define MS_WG_K(KK)                                                                                 \
    do {                                                                                            \
        if (pl.cfg == CFG_128x128) MS_WG(2, 2, 2, 2, KK);                                           \
        else if (pl.cfg == CFG_64x64) MS_WG(2, 2, 1, 1, KK);                                        \
        else if (pl.cfg == CFG_32x128) MS_WG(1, 4, 1, 1, KK);                                       \
        else MS_WG(1, 4, 1, 2, KK);                                                                 \
    } while (0)
    if (p.K == 1) MS_WG_K(1);
    else if (p.K == 3) MS_WG_K(3);
    else if (p.K == 5) MS_WG_K(5);
    else if (p.K == 7) MS_WG_K(7);
    else if (p.K == 15) MS_WG_K(15);
    else return MS_ERR_UNSUPPORTED;
#undef MS_WG_K
#undef MS_WG
    MS_CHECK_LAUNCH();
    return msm_wgrad_reduce(partial, pl.stride_floats, pl.nsplit, (size_t)p.Cout * NG, p.Cout, gw, gb, beta, s);
}

int msm_convt1d_fwd(const ConvP& p, const float* x, const float* w, const float* bias, float* y,
                    void* ws, size_t ws_bytes, hipStream_t s) {
    const int S = p.stride, CinT = p.Cout, CoutT = p.Cin, LinT = p.Lout;
    if (!ws || ws_bytes < msm_convt_fwd_ws(p) || (((uintptr_t)ws) & 15)) return MS_ERR_WORKSPACE;
    float* wp = (float*)ws;
    const size_t total = (size_t)CoutT * S * CinT * 3;
    unsigned nb = (unsigned)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    RowP r;
    const RowCfg cfg = convt_fwd_cfg(CoutT * S, p.B, LinT);
    make_rowp(&r, cfg, p.B, CinT, LinT, CoutT * S, 3, 1, -1, MS_PAD_ZERO, p.act, MS_ACT_NONE, p.slope);
    const size_t wbytes = align16(total * sizeof(float));
    const bool ia = p.in_act != 0;   // LeakyReLU in front of the transposed conv: applied to x on load
    if (ia) r.in_act = MS_MOD_LRELU_FWD;
    // pipelined kernel: the 3-tap window stays, but only the two live taps of each phase are multiplied
    Row2P q2;
    int tile2 = 0, am2 = 0;
    const bool two = rows2_pick(cfg, 2, 8, ia, S, 1, r, x, ia ? x : nullptr, wp, nullptr, y, nullptr, &q2, &tile2, &am2);
    if (two)
        hipLaunchKernelGGL(k_pack_convt_w2, dim3(nb), dim3(256), 0, s, w, wp, CinT, CoutT, p.K, S, p.pad);
    else
        hipLaunchKernelGGL(k_pack_convt_w, dim3(nb), dim3(256), 0, s, w, wp, CinT, CoutT, p.K, S, p.pad);
    MS_CHECK_LAUNCH();
    return rows_maybe_split(cfg, r, row_cc(3), CoutT, LinT * S, bias, nullptr, y, nullptr,
                            (char*)ws + wbytes, ws_bytes - wbytes, s,
                            [&](const RowP& rp, const float* b_, const float* r_, float* y_, float* ya_) {
                                if (two) {
                                    Row2P q;
                                    int tile, am, bm, bn;
                                    if (!rows2_pick(cfg, 2, 8, ia, S, 1, rp, x, ia ? x : nullptr, wp, r_, y_, ya_, &q, &tile, &am))
                                        return (int)MS_ERR_UNSUPPORTED;
                                    q.KG = rp.CK * 2;
                                    row_tile(cfg, &bm, &bn);
                                    const unsigned gz_ = (unsigned)((rp.CK + rp.CKs - 1) / rp.CKs);
                                    if (!r_ && !ya_) {     // paired split-bf16 kernel where its grid fills the chip
                                        if (const int b3 = rows3p_convt_bm(q, bn, S, gz_))
                                            return msr3p_convt_launch(b3, S, ia, q, x, wp, b_, y_, gz_, s);
                                        if (cfg == ROW_32x256 && q.R == 1 && q.Lt == 256) {
                                            Row2P q3 = rows3p_retile(q, 3);
                                            if (const int b3 = rows3p_convt_bm(q3, 128, S, gz_))
                                                return msr3p_convt_launch(b3, S, ia, q3, x, wp, b_, y_, gz_, s);
                                        }
                                    }
                                    const unsigned gx_ = rp.R == 1 ? (unsigned)(rp.B * rp.tiles_per_row)
                                                                   : (unsigned)((rp.B + rp.R - 1) / rp.R);
                                    return msr2_launch(tile, 2, 8, am, S, q, x, nullptr, wp, b_, r_, y_, ya_, gx_,
                                                       (unsigned)((rp.M + bm - 1) / bm),
                                                       (unsigned)((rp.CK + rp.CKs - 1) / rp.CKs), s, 1);
                                }
                                if (ia) {
                                    if (S == 8) return launch_rows_k<3, true, 8>(cfg, rp, x, x, wp, b_, r_, y_, ya_, s);
                                    return launch_rows_k<3, true, 2>(cfg, rp, x, x, wp, b_, r_, y_, ya_, s);
                                }
                                if (S == 8) return launch_rows_k<3, false, 8>(cfg, rp, x, nullptr, wp, b_, r_, y_, ya_, s);
                                return launch_rows_k<3, false, 2>(cfg, rp, x, nullptr, wp, b_, r_, y_, ya_, s);
                            });
}

// ---- ConvTranspose1d backward (p = mirrored conv: Cin_T = p.Cout, Cout_T = p.Cin, Lin_T = p.Lout)
bool msm_convt_bwd_applicable(const ConvP& p) {
    const int S = p.stride;
    if (!(S == 2 || S == 8) || p.K != 2 * S || 2 * p.pad != S || p.dil != 1 || p.groups != 1) return false;
    if (p.Cout < 32 || p.Cin % 8 || (p.Cin * S) % 8) return false;
    if ((long long)p.B * p.Cin * p.Lin >= (1LL << 31) || (long long)p.B * p.Cout * p.Lout >= (1LL << 31)) return false;
    RowP q;
    return make_rowp(&q, pick_row_cfg(p.Cout, p.B, p.Lout), p.B, p.Cin * S, p.Lout, p.Cout, 3, 1, -1, 0, 0, 0, 0.f);
}

size_t msm_convt_bwd_data_ws(const ConvP& p) {
    RowP r;
    const RowCfg cfg = pick_row_cfg(p.Cout, p.B, p.Lout);
    make_rowp(&r, cfg, p.B, p.Cin * p.stride, p.Lout, p.Cout, 3, 1, -1, 0, 0, 0, 0.f);
    return align16((size_t)p.Cout * p.Cin * p.stride * 3 * sizeof(float)) + rows_split_ws(cfg, r, row_cc(3));
}

const char* msm_convt_bwd_data_name(const ConvP& p) {
    static thread_local char buf[96];
    const RowCfg c = pick_row_cfg(p.Cout, p.B, p.Lout);
    const char* tile = row_tile_str(c);
    RowP r;
    make_rowp(&r, c, p.B, p.Cin * p.stride, p.Lout, p.Cout, 3, 1, -1, 0, 0, 0, 0.f);
    Row2P q;
    q.L = p.Lout; q.R = r.R; q.SS = r.SS;
    const int t2 = c == ROW_128x128 ? MSR2_128x128 : (c == ROW_64x128 ? MSR2_64x128 : (c == ROW_64x64 ? MSR2_64x64 : MSR2_32x256));
    if (c != ROW_64x256 && p.act == MS_ACT_LRELU && msr2_supported(t2, 2, 8, 2, 0, q, p.stride))
        snprintf(buf, sizeof(buf), "k_conv_rows2<%s, 2, 8, 2, 0, %d>", tile, p.stride);
    else
        snprintf(buf, sizeof(buf), "k_conv_mfma_rows<%s, 3, 8, true, 0, %d>", tile, p.stride);
    return buf;
}

// gx_T = conv over the phase-split gradient: M = Cin_T, channels (co, r), 3 taps
int msm_convt1d_bwd_data(const ConvP& p, const float* gy, const float* y_act, const float* w,
                         float* gx, void* ws, size_t ws_bytes, hipStream_t s) {
    const int S = p.stride, CinT = p.Cout, CoutT = p.Cin, LinT = p.Lout;
    if (!ws || ws_bytes < msm_convt_bwd_data_ws(p) || (((uintptr_t)ws) & 15)) return MS_ERR_WORKSPACE;
    float* wq = (float*)ws;
    const size_t total = (size_t)CinT * CoutT * S * 3;
    unsigned nb = (unsigned)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    RowP r;
    const RowCfg cfg = pick_row_cfg(CinT, p.B, LinT);
    make_rowp(&r, cfg, p.B, CoutT * S, LinT, CinT, 3, 1, -1, MS_PAD_ZERO, MS_ACT_NONE, p.act, p.slope);
    // the activation operand aliases the data when absent (pass-through kind)
    const float* ya = y_act ? y_act : gy;
    if (!y_act) r.in_act = MS_ACT_NONE;
    const size_t wbytes = align16(total * sizeof(float));
    // pipelined kernel: the 3-tap window stays, but only the two live taps of each phase are multiplied
    Row2P q2;
    int tile2 = 0, am2 = 0;
    const bool two = rows2_pick(cfg, 2, 8, true, 0, S, r, gy, ya, wq, nullptr, gx, nullptr, &q2, &tile2, &am2);
    if (two)
        hipLaunchKernelGGL(k_pack_convt_bwd_w2, dim3(nb), dim3(256), 0, s, w, wq, CinT, CoutT, p.K, S, p.pad);
    else
        hipLaunchKernelGGL(k_pack_convt_bwd_w, dim3(nb), dim3(256), 0, s, w, wq, CinT, CoutT, p.K, S, p.pad);
    MS_CHECK_LAUNCH();
    return rows_maybe_split(cfg, r, row_cc(3), CinT, LinT, nullptr, nullptr, gx, nullptr,
                            (char*)ws + wbytes, ws_bytes - wbytes, s,
                            [&](const RowP& rp, const float* b_, const float* r_, float* y_, float* ya_) {
                                if (two) {
                                    Row2P q;
                                    int tile, am, bm, bn;
                                    if (!rows2_pick(cfg, 2, 8, true, 0, S, rp, gy, ya, wq, r_, y_, ya_, &q, &tile, &am))
                                        return (int)MS_ERR_UNSUPPORTED;
                                    q.KG = rp.CK * 2;
                                    row_tile(cfg, &bm, &bn);
                                    const unsigned gx_ = rp.R == 1 ? (unsigned)(rp.B * rp.tiles_per_row)
                                                                   : (unsigned)((rp.B + rp.R - 1) / rp.R);
                                    return msr2_launch(tile, 2, 8, am, 0, q, gy, ya, wq, b_, r_, y_, ya_, gx_,
                                                       (unsigned)((rp.M + bm - 1) / bm),
                                                       (unsigned)((rp.CK + rp.CKs - 1) / rp.CKs), s, S);
                                }
                                if (S == 8) return launch_rows_k<3, true, 0, 8>(cfg, rp, gy, ya, wq, b_, r_, y_, ya_, s);
                                return launch_rows_k<3, true, 0, 2>(cfg, rp, gy, ya, wq, b_, r_, y_, ya_, s);
                            });
}

struct ConvtWgradPlan {
    WgradPlan w;
    size_t dwq_floats;
};

static ConvtWgradPlan plan_convt_wgrad(const ConvP& p) {
    // weight-grad GEMM: M = Cin_T, N = Cout_T * S * 3, K = B * Lin_T
    ConvP m = p;
    m.Cout = p.Cout;                 // rows: x_T channels
    m.Cin = p.Cin * p.stride;        // (co, r) phase channels
    m.K = 3; m.Lin = p.Lout; m.Lout = p.Lout; m.stride = 1; m.pad = 1; m.dil = 1;
    ConvtWgradPlan q;
    q.w = plan_wgrad(m);
    q.dwq_floats = (size_t)p.Cout * m.Cin * 3 + p.Cout;
    return q;
}

// workspace: [dWq | split-K slabs of whichever kernel takes the launch]
size_t msm_convt_bwd_weight_ws(const ConvP& p) {
    const ConvtWgradPlan q = plan_convt_wgrad(p);
    size_t slabs = (size_t)q.w.nsplit * q.w.stride_floats * sizeof(float);
    const size_t rows = msw_convt_ws(p);
    if (rows > slabs) slabs = rows;
    return align16(q.dwq_floats * sizeof(float)) + slabs;
}

const char* msm_convt_bwd_weight_name(const ConvP& p) {
    if (msw_convt_ws(p) > 0 && p.act != MS_ACT_TANH) {
        static thread_local char buf[64];
        snprintf(buf, sizeof(buf), "k_wgrad_rows<3, %d, true, %d, 3, %d>", p.Cout >= 128 ? 2 : 1, p.Lout <= 32 ? 8 : 16, p.stride);
        return buf;
    }
    return kname("k_igemm_wgrad", plan_convt_wgrad(p).w.cfg, 3, "");
}

// gw_T via the phase-split weight-grad GEMM, then un-packed into (Cin_T, Cout_T, K)
int msm_convt1d_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act,
                           float* gw, float beta, void* ws, size_t ws_bytes, hipStream_t s) {
    const int S = p.stride, CinT = p.Cout, CoutT = p.Cin, LinT = p.Lout;
    const ConvtWgradPlan pl = plan_convt_wgrad(p);
    if (!ws || ws_bytes < msm_convt_bwd_weight_ws(p)) return MS_ERR_WORKSPACE;
    float* dwq = (float*)ws;
    const size_t dbytes = align16(pl.dwq_floats * sizeof(float));
    float* partial = (float*)((char*)ws + dbytes);
    const size_t gtotal0 = (size_t)CinT * CoutT * p.K;
    // row-tile form (wgrad_rows.hip) where it applies; the im2col form below otherwise
    if (msw_convt_dwq(p, x, gy, y_act, dwq, partial, ws_bytes - dbytes, s) == MS_OK) {
        unsigned nb0 = (unsigned)((gtotal0 + 255) / 256);
        if (nb0 > 4096) nb0 = 4096;
        hipLaunchKernelGGL(k_unpack_convt_gw, dim3(nb0), dim3(256), 0, s, dwq, gw, CinT, CoutT, p.K, S, p.pad, beta);
        MS_CHECK_LAUNCH();
        return MS_OK;
    }
    IgP q;
    q.B = p.B; q.CK = CoutT * S; q.L = LinT; q.M = CinT; q.dil = 1; q.off0 = -1;
    q.pad_mode = MS_PAD_ZERO; q.act = MS_ACT_NONE; q.in_act = p.act; q.slope = p.slope;
    q.N = p.B * LinT; q.KG = q.CK * 3; q.in_s = S;
    const int NG = q.CK * 3;
    dim3 grid((unsigned)ms_ceil_div(NG, pl.w.bn), (unsigned)ms_ceil_div(CinT, pl.w.bm), (unsigned)pl.w.nsplit);
    // A operand = x_T (no activation), B operand = phase-split gy with act'(y_T)
#define MS_TWG(WGM, WGN, TM, TN)                                                                     \
    hipLaunchKernelGGL((k_igemm_wgrad<WGM, WGN, TM, TN, 3>), grid, dim3(256), 0, s, q, pl.w.cps, gy, \
                       y_act, x, p.in_act ? x : (const float*)nullptr,                               \
                       p.in_act ? MS_MOD_LRELU_FWD : 0, partial, pl.w.stride_floats)
    if (pl.w.cfg == CFG_128x128) MS_TWG(2, 2, 2, 2);
    else if (pl.w.cfg == CFG_64x64) MS_TWG(2, 2, 1, 1);
    else if (pl.w.cfg == CFG_32x128) MS_TWG(1, 4, 1, 1);
    else MS_TWG(1, 4, 1, 2);
#undef MS_TWG
    MS_CHECK_LAUNCH();
    {
        const int rc = msm_wgrad_reduce(partial, pl.w.stride_floats, pl.w.nsplit, (size_t)CinT * NG, CinT, dwq,
                                        (float*)nullptr, 0.f, s);
        if (rc != MS_OK) return rc;
    }
    const size_t gtotal = (size_t)CinT * CoutT * p.K;
    unsigned nb = (unsigned)((gtotal + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_unpack_convt_gw, dim3(nb), dim3(256), 0, s, dwq, gw, CinT, CoutT, p.K, S, p.pad, beta);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

// 16-byte variant: a lane owns 4 consecutive outputs, 8 slabs in flight per lane (the dword version is
// latency-bound: 30 dependent-ish rounds of 4 loads for the 121 slabs of a 128-channel layer).
__global__ __launch_bounds__(256) void k_wgrad_reduce_v4(const float* __restrict__ partial,
                                                        size_t partial_stride, int nsplit,
                                                        size_t wsize, size_t total,
                                                        float* __restrict__ out,
                                                        float* __restrict__ outb, float beta) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t i = ((size_t)blockIdx.x * 64 + lane) * 4;
    const bool ok = i < total;                       // total % 4 == 0
    float4 s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
        int z = wv;
        for (; z + 28 < nsplit; z += 32) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = *reinterpret_cast<const float4*>(partial + (size_t)(z + 4 * u) * partial_stride + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s[u].x += v[u].x; s[u].y += v[u].y; s[u].z += v[u].z; s[u].w += v[u].w; }
        }
        for (; z < nsplit; z += 4) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)z * partial_stride + i);
            s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
        }
    }
    float4 t;
    t.x = ((s[0].x + s[1].x) + (s[2].x + s[3].x)) + ((s[4].x + s[5].x) + (s[6].x + s[7].x));
    t.y = ((s[0].y + s[1].y) + (s[2].y + s[3].y)) + ((s[4].y + s[5].y) + (s[6].y + s[7].y));
    t.z = ((s[0].z + s[1].z) + (s[2].z + s[3].z)) + ((s[4].z + s[5].z) + (s[6].z + s[7].z));
    t.w = ((s[0].w + s[1].w) + (s[2].w + s[3].w)) + ((s[4].w + s[5].w) + (s[6].w + s[7].w));
    red[wv][lane] = t;
    __syncthreads();
    if (wv == 0 && ok) {
        float4 r;
        r.x = (red[0][lane].x + red[1][lane].x) + (red[2][lane].x + red[3][lane].x);
        r.y = (red[0][lane].y + red[1][lane].y) + (red[2][lane].y + red[3][lane].y);
        r.z = (red[0][lane].z + red[1][lane].z) + (red[2][lane].z + red[3][lane].z);
        r.w = (red[0][lane].w + red[1][lane].w) + (red[2][lane].w + red[3][lane].w);
        // (wsize % 4 == 0: a vector is all weights or all bias; the bias entries follow the weights in a slab)
        if (i >= wsize && !outb) return;
        float4* o = i < wsize ? reinterpret_cast<float4*>(out + i) : reinterpret_cast<float4*>(outb + (i - wsize));
        if (beta != 0.f) {
            const float4 p = *o;
            r.x += beta * p.x; r.y += beta * p.y; r.z += beta * p.z; r.w += beta * p.w;
        }
        *o = r;
    }
}

int msm_wgrad_reduce(const float* partial, size_t stride_floats, int nsplit, size_t wsize, int nbias,
                     float* gw, float* gb, float beta, hipStream_t s) {
    const size_t total = wsize + (size_t)nbias;
    // weights with 16-byte accesses where the layout allows it, the bias tail with the dword kernel
    if (wsize % 4 == 0 && stride_floats % 4 == 0 &&
        ((((uintptr_t)partial) | ((uintptr_t)gw)) & 15) == 0) {
        // the bias tail rides along when it is 16-byte sized and aligned too (one launch instead of two)
        const bool with_bias = nbias > 0 && gb && nbias % 4 == 0 && (((uintptr_t)gb) & 15) == 0;
        const size_t tot4 = with_bias ? wsize + (size_t)nbias : wsize;
        hipLaunchKernelGGL(k_wgrad_reduce_v4, dim3((unsigned)((tot4 / 4 + 63) / 64)), dim3(256), 0, s, partial,
                           stride_floats, nsplit, wsize, tot4, gw, with_bias ? gb : (float*)nullptr, beta);
        MS_CHECK_LAUNCH();
        if (nbias > 0 && gb && !with_bias) {
            hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((nbias + 63) / 64)), dim3(256), 0, s,
                               partial + wsize, stride_floats, nsplit, (size_t)0, nbias, gw, gb, beta);
            MS_CHECK_LAUNCH();
        }
        return MS_OK;
    }
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, partial,
                       stride_floats, nsplit, wsize, nbias, gw, gb, beta);
    MS_CHECK_LAUNCH();
    return MS_OK;
}
