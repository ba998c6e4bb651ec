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

__device__ __forceinline__ float wr_xact(float v, int kind, float slope) {
    return kind == MS_MOD_LRELU_FWD ? (v > 0.f ? v : v * slope) : v;
}

template <int K, int TM, bool VEC>
__global__ __launch_bounds__(256) void k_wgrad_rows(WrP p, const float* __restrict__ X,
                                                   const float* __restrict__ G,
                                                   const float* __restrict__ Gact,
                                                   float* __restrict__ partial, size_t pstride) {
    constexpr int BM = 2 * TM * 32;
    constexpr int NGQ = VEC ? TM * 4 : TM * 16;     // G loads per thread and chunk
    constexpr int NXQ = VEC ? 8 : 32;               // X loads per thread and chunk
    extern __shared__ float smem[];
    float* Gs = smem;                               // [BM][PG]
    float* Xs = smem + BM * p.PG;                   // [CB][PX]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * BM, c0 = blockIdx.x * CB;
    const int NG = p.CK * K;
    const float* Gq = Gact ? Gact : G;
    const int g_kind = Gact ? p.g_kind : MS_ACT_NONE;

    f32x16 acc[TM][K];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float asum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) asum[i] = 0.f;

    // the halo columns of the gradient tile (and the column past the last segment) stay zero
    for (int i = tid; i < BM * p.PG + CB * p.PX + 8; i += 256) smem[i] = 0.f;

    // ---- chunk-invariant loader descriptors
    // VEC: G thread = (vector v = tid&15 of the R*Lt/4 per row, rows (tid>>4) + 16q);
    //      X thread = (aligned vector v = tid&31 of R*NVS per row, rows (tid>>5) + 8q)
    // scalar: G thread = (column tid&63, rows (tid>>6) + 4q); X thread = (column tid&127, rows (tid>>7) + 2q)
    const int sh = (4 - (p.pad & 3)) & 3;           // row start of the X window within its 16-byte vector
    int g_seg, g_t, x_seg, x_u;                      // segment / position of this thread's column
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
        x_seg = c / p.SS; x_u = c - x_seg * p.SS; x_cv = x_seg < p.R;
    }
    const int g_row0 = VEC ? tid >> 4 : tid >> 6, g_rstep = VEC ? 16 : 4;
    const int x_row0 = VEC ? tid >> 5 : tid >> 7, x_rstep = VEC ? 8 : 2;

    float4 gv4[VEC ? NGQ : 1], ga4[VEC ? NGQ : 1], xv4[VEC ? NXQ : 1];
    float gv1[VEC ? 1 : NGQ], ga1[VEC ? 1 : NGQ], xv1[VEC ? 1 : NXQ];
    auto chunk_origin = [&](int ch, int& b0, int& t0) {
        if (p.R == 1) { b0 = ch / p.tiles_per_row; t0 = (ch - b0 * p.tiles_per_row) * p.Lt; }
        else { b0 = ch * p.R; t0 = 0; }
    };
    auto gload = [&](int ch) {
        int b0, t0;
        chunk_origin(ch, b0, t0);
        {   // gradient rows
            const int b = b0 + g_seg, t = t0 + g_t;
            const bool cok = g_cv && b < p.B && t < p.L;
#pragma unroll
            for (int q = 0; q < NGQ; ++q) {
                const int m = m0 + g_row0 + g_rstep * q;
                const bool ok = cok && m < p.M;
                const size_t o = ok ? ((size_t)b * p.M + m) * p.L + t : 0;
                if (VEC) {
                    gv4[q] = *reinterpret_cast<const float4*>(G + o);
                    ga4[q] = *reinterpret_cast<const float4*>(Gq + o);
                } else {
                    gv1[q] = G[o];
                    ga1[q] = Gq[o];
                }
            }
        }
        {   // input rows with halo
            const int b = b0 + x_seg, t = t0 - p.pad + x_u;      // VEC: t % 4 == 0, all in or all out
            const bool cok = x_cv && b < p.B && t >= 0 && t < p.L && (VEC || x_u < p.SS);
#pragma unroll
            for (int q = 0; q < NXQ; ++q) {
                const int c = c0 + x_row0 + x_rstep * q;
                const bool ok = cok && c < p.CK;
                const size_t o = ok ? ((size_t)b * p.CK + c) * p.L + t : 0;
                if (VEC) xv4[q] = *reinterpret_cast<const float4*>(X + o);
                else xv1[q] = X[o];
            }
        }
    };
    auto lstore = [&](int ch) {
        int b0, t0;
        chunk_origin(ch, b0, t0);
        {
            const int b = b0 + g_seg, t = t0 + g_t;
            const bool cok = g_cv && b < p.B && t < p.L;
            float* d = Gs + g_seg * p.SS + g_t;
            if (g_cv) {
#pragma unroll
                for (int q = 0; q < NGQ; ++q) {
                    const int row = g_row0 + g_rstep * q;
                    const bool ok = cok && m0 + row < p.M;
                    if (VEC) {
                        d[row * p.PG + 0] = ok ? ms_act_grad(gv4[q].x, ga4[q].x, g_kind, p.slope) : 0.f;
                        d[row * p.PG + 1] = ok ? ms_act_grad(gv4[q].y, ga4[q].y, g_kind, p.slope) : 0.f;
                        d[row * p.PG + 2] = ok ? ms_act_grad(gv4[q].z, ga4[q].z, g_kind, p.slope) : 0.f;
                        d[row * p.PG + 3] = ok ? ms_act_grad(gv4[q].w, ga4[q].w, g_kind, p.slope) : 0.f;
                    } else {
                        d[row * p.PG] = ok ? ms_act_grad(gv1[q], ga1[q], g_kind, p.slope) : 0.f;
                    }
                }
            }
        }
        {
            const int b = b0 + x_seg, t = t0 - p.pad + x_u;
            const bool cok = x_cv && b < p.B && t >= 0 && t < p.L;
            float* d = Xs + x_seg * p.SS + x_u;
            if (x_cv) {
#pragma unroll
                for (int q = 0; q < NXQ; ++q) {
                    const int row = x_row0 + x_rstep * q;
                    const bool ok = cok && c0 + row < p.CK;
                    if (VEC) {
                        const float e[4] = {xv4[q].x, xv4[q].y, xv4[q].z, xv4[q].w};
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (x_u + i >= 0 && x_u + i < p.SS)
                                d[row * p.PX + i] = ok ? wr_xact(e[i], p.x_kind, p.slope) : 0.f;
                    } else if (x_u < p.SS) {
                        d[row * p.PX] = ok ? wr_xact(xv1[q], p.x_kind, p.slope) : 0.f;
                    }
                }
            }
        }
    };

    const int c_begin = blockIdx.z * p.cps;
    const int c_end = min(c_begin + p.cps, p.nchunks);
    if (c_begin < c_end) gload(c_begin);
    __syncthreads();                                 // zero fill complete
    if (c_begin < c_end) lstore(c_begin);
    __syncthreads();

    const float* ap = Gs + (wm * TM * 32 + (lane & 31)) * p.PG + h;
    const float* bp = Xs + (wn * 32 + (lane & 31)) * p.PX + h;
    const int KP = p.kcols >> 1;                     // even (kcols % 4 == 0)
    float a0[TM], b0[K], a1[TM], b1[K];
    auto frag = [&](int kk, float (&a)[TM], float (&b)[K]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = ap[i * 32 * p.PG + 2 * kk];
#pragma unroll
        for (int j = 0; j < K; ++j) b[j] = bp[2 * kk + j * p.dil];
    };
    auto mma = [&](const float (&a)[TM], const float (&b)[K]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            asum[i] += a[i];
#pragma unroll
            for (int j = 0; j < K; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    for (int ch = c_begin; ch < c_end; ++ch) {
        const bool more = ch + 1 < c_end;
        if (more) gload(ch + 1);
        // two-stage register pipeline: the fragments of step kk+1 are read while step kk multiplies
        // (the read past the last step lands in the zeroed pad columns and is not used)
        frag(0, a0, b0);
        for (int kk = 0; kk < KP; kk += 2) {
            frag(kk + 1, a1, b1);
            mma(a0, b0);
            frag(kk + 2, a0, b0);
            mma(a1, b1);
        }
        __syncthreads();
        if (more) {
            lstore(ch + 1);
            __syncthreads();
        }
    }

    // D[row][col]: col = lane&31 (input channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (output channel)
    float* part = partial + (size_t)blockIdx.z * pstride;
    const int c = c0 + wn * 32 + (lane & 31);
    if (c < p.CK) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.M) {
#pragma unroll
                    for (int j = 0; j < K; ++j) part[(size_t)m * NG + (size_t)c * K + j] = acc[i][j][r];
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

struct WrPlan {
    bool ok;
    WrP p;
    int tm, nsplit;
    bool vec;
    dim3 grid;
    size_t lds, stride_floats;
};

WrPlan plan_wrows(const ConvP& c) {
    WrPlan q;
    q.ok = false;
    const int K = c.K;
    if (c.groups != 1 || c.stride != 1 || c.Lout != c.Lin || c.pad_mode != MS_PAD_ZERO) return q;
    if (!(K == 1 || K == 3 || K == 5 || K == 7)) return q;
    if (c.Cout < 64 || c.Cin < 32) return q;
    const int H = (K - 1) * c.dil;
    if (H > 24 || c.pad > H) return q;
    if ((long long)c.B * c.Cout * c.Lin >= (1LL << 31) || (long long)c.B * c.Cin * c.Lin >= (1LL << 31)) return q;
    WrP& p = q.p;
    p.B = c.B; p.CK = c.Cin; p.L = c.Lin; p.M = c.Cout; p.dil = c.dil; p.pad = c.pad;
    p.g_kind = c.act; p.x_kind = c.in_act ? MS_MOD_LRELU_FWD : MS_ACT_NONE; p.slope = c.slope;
    if (p.L >= KMAX) { p.Lt = KMAX; p.R = 1; p.tiles_per_row = ms_ceil_div(p.L, KMAX); p.nchunks = p.B * p.tiles_per_row; }
    else {
        p.Lt = p.L;
        p.R = (KMAX + H) / (p.L + H);
        if (p.R < 1) p.R = 1;
        p.tiles_per_row = 1;
        p.nchunks = ms_ceil_div(p.B, p.R);
    }
    p.SS = p.Lt + H;
    p.RSZ = p.R * p.SS;
    p.kcols = (p.RSZ - H + 3) & ~3;
    q.vec = p.L % 4 == 0 && p.R * (p.Lt / 4) <= 16 && p.R * ((p.SS + 6) / 4) <= 32;
    if (!q.vec && (p.kcols > 64 || p.RSZ > 128)) return q;
    p.PG = (p.kcols + 2) | 1;
    p.PX = (p.kcols + H + 2) | 1;
    q.tm = (K <= 3 && c.Cout >= 128) ? 2 : 1;
    const int BM = 64 * q.tm;
    q.lds = (size_t)(BM * p.PG + CB * p.PX + 8) * sizeof(float);
    if (q.lds > 64 * 1024) return q;
    const int tiles = ms_ceil_div(p.M, BM) * ms_ceil_div(p.CK, CB);
    q.stride_floats = (size_t)p.M * p.CK * K + p.M;
    int ns = ms_ceil_div(512, tiles);
    const int max_by_work = p.nchunks / 4 > 0 ? p.nchunks / 4 : 1;
    if (ns > max_by_work) ns = max_by_work;
    const size_t cap = (size_t)24 << 20;
    const size_t max_by_bytes = cap / (q.stride_floats * 4);
    if ((size_t)ns > max_by_bytes) ns = max_by_bytes > 0 ? (int)max_by_bytes : 1;
    if (ns < 1) ns = 1;
    p.cps = ms_ceil_div(p.nchunks, ns);
    q.nsplit = ms_ceil_div(p.nchunks, p.cps);
    q.grid = dim3((unsigned)ms_ceil_div(p.CK, CB), (unsigned)ms_ceil_div(p.M, BM), (unsigned)q.nsplit);
    q.ok = true;
    return q;
}

template <int K>
void launch_wrows(const WrPlan& q, bool vec, const float* x, const float* gy, const float* y_act,
                  float* partial, hipStream_t s) {
    if (q.tm == 2) {
        if (vec) hipLaunchKernelGGL((k_wgrad_rows<K, 2, true>), q.grid, dim3(256), q.lds, s, q.p, x, gy, y_act, partial, q.stride_floats);
        else hipLaunchKernelGGL((k_wgrad_rows<K, 2, false>), q.grid, dim3(256), q.lds, s, q.p, x, gy, y_act, partial, q.stride_floats);
    } else {
        if (vec) hipLaunchKernelGGL((k_wgrad_rows<K, 1, true>), q.grid, dim3(256), q.lds, s, q.p, x, gy, y_act, partial, q.stride_floats);
        else hipLaunchKernelGGL((k_wgrad_rows<K, 1, false>), q.grid, dim3(256), q.lds, s, q.p, x, gy, y_act, partial, q.stride_floats);
    }
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
    snprintf(buf, sizeof(buf), "k_wgrad_rows<%d, %d, %s>", p.K, q.tm, q.vec ? "true" : "false");
    return buf;
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
    if (c.K == 1) launch_wrows<1>(q, vec, x, gy, y_act, partial, s);
    else if (c.K == 3) launch_wrows<3>(q, vec, x, gy, y_act, partial, s);
    else if (c.K == 5) launch_wrows<5>(q, vec, x, gy, y_act, partial, s);
    else launch_wrows<7>(q, vec, x, gy, y_act, partial, s);
    MS_CHECK_LAUNCH();
    return msm_wgrad_reduce(partial, q.stride_floats, q.nsplit, (size_t)c.Cout * c.Cin * c.K, c.Cout, gw,
                            gb, beta, s);
}
