// The generator at INFERENCE batch sizes (BASELINE config 2: B = 1, 32 frames; generator/full.py:23-40): the first conv
// (nn.Conv1d(80, 512, 7) behind ReflectionPad1d(3)) and the upsampling layers on long rows (nn.ConvTranspose1d 256 -> 128 /
// k16 s8 on 256 positions, 128 -> 64 and 64 -> 32 / k4 s2 on 2048 and 4096), each + LeakyReLU.  With one batch row these are
// 18 / 134 / 67 / 34 MFLOP: nothing for the vector pipe, but 10 + 15 + 13 + 11 us on the row-tile matrix kernels, whose
// tiles (64+ rows x 128+ columns per workgroup, operands split for the 16-bit matrix pipe) give such a layer a few dozen
// workgroups and a launch-latency-sized critical path.  Here: plain fp32 FMA (exact fp32 products), ~256 small workgroups,
// no weight image, no pack launch, one launch per layer.  Measured at B = 1 (tools/gfwd_b1.py --list, us): 10.1 -> 4.4,
// 15.0 -> 11.5, 13.1 -> 5.6, 11.2 -> 1.7; the 30-layer forward 237 -> 219 us.  Dispatched only while the whole layer is at
// most two workgroups per CU (conv: four tiles of 32 positions): the train step (B = 32) never comes here.
//
// k_convt_long<S, R> ConvTranspose1d, kernel 2 S / stride S / padding S / 2, rows of >= 64 positions:
//     y[b, co, j S + k - S/2] = sum_ci x[b, ci, j] w[ci, co, k].  A workgroup owns 64 input positions j x R output channels
//     (R 2 S = 32 weights per input channel, CONTIGUOUS in w[ci, co, k]) and writes the S outputs j S .. j S + S - 1 of each:
//     output j S + r takes taps (j, r + S/2) and (j - 1, r + S/2 + S) for r < S/2, (j + 1, r - S/2) and (j, r + S/2)
//     otherwise -- three window values per channel and lane for 32 FMAs.  Input channels in rounds of 64 through LDS (window
//     rows of 66, the 64 x 32 weights; wave-uniform 16-byte weight reads are broadcasts); inside a round wave v walks channels
//     [16 v, 16 v + 16).  The four waves' partial sums meet in LDS; wave v finishes a quarter of the 16 accumulators, adding in
//     wave order (deterministic), + bias + activation.  1024 waves at B = 1 for each of the three layers.
// k_conv_small<K>    Conv1d, stride 1, dilation 1, one group, zero or reflection padding, tiles of 32 output positions:
//     thread = (half of the input channels, output channel of 4, position of 32); window and weights from LDS; 128 workgroups
//     x 9 KB of weights for the 80 -> 512 layer.
#include "ms_common.h"
#include "conv_thin.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct CtsP {
    int B, Cin, Cout, Lin, Lout;
    int act, in_act;
    float slope;
};

// Every staging loop below is "all loads of a round into registers, then the LDS stores": a load -> store loop leaves ONE load
// in flight per thread and the round takes (elements per thread) x (memory latency) -- measured on the first form of
// k_conv_small: 9.8 us, 4.5 us since.  The NEXT round's loads are issued before the current round's arithmetic.
constexpr int CL_CR = 64;         // input channels per LDS round
constexpr int CL_XS = 66;         // 64 positions + one either side

template <int S, int R>
__global__ __launch_bounds__(256) void k_convt_long(const CtsP p, const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ y) {
    constexpr int K = 2 * S, NA = R * S;
    static_assert(R * K == 32 && NA == 16, "32 weights per input channel, 16 accumulators");
    constexpr int NX = CL_CR * CL_XS;                  // 4224 window values per round
    constexpr int XPT = (NX + 255) / 256;              // 17
    constexpr int WPT = CL_CR * 8 / 256;               // 16-byte pieces of a round's weights per thread
    __shared__ float xl[NX];
    __shared__ __attribute__((aligned(16))) float wl[CL_CR * 32];
    __shared__ float red[4 * NA * 64];
    const int jt = blockIdx.x, co0 = blockIdx.y * R, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = jt * 64 + lane;
    const bool in = j < p.Lin;
    float acc[R][S];
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
        for (int r = 0; r < S; ++r) acc[q][r] = 0.f;
    const float* xb = x + (size_t)b * p.Cin * p.Lin;
    float xv[XPT];
    f32x4 wq[WPT];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int e = tid + 256 * u;
            const int c = e / CL_XS, m = e - c * CL_XS;
            const int jj = jt * 64 - 1 + m;            // column m of a row holds x[jt 64 - 1 + m]; zero outside the row
            xv[u] = (e < NX && c0 + c < p.Cin && jj >= 0 && jj < p.Lin) ? xb[(size_t)(c0 + c) * p.Lin + jj] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < WPT; ++u) {                 // channel x 8 pieces of 16 bytes: the R output channels' 2 S taps
            const int c = (tid + 256 * u) >> 3, part = tid & 7;
            wq[u] = (c0 + c < p.Cin) ? *reinterpret_cast<const f32x4*>(w + ((size_t)(c0 + c) * p.Cout + co0) * K + part * 4)
                                     : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < p.Cin; c0 += CL_CR) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int e = tid + 256 * u;
            if (e < NX) {
                float v = xv[u];
                if (p.in_act == MS_ACT_LRELU) v = v > 0.f ? v : v * p.slope;
                xl[e] = v;
            }
        }
#pragma unroll
        for (int u = 0; u < WPT; ++u) *reinterpret_cast<f32x4*>(wl + (tid + 256 * u) * 4) = wq[u];
        __syncthreads();
        if (c0 + CL_CR < p.Cin) fetch(c0 + CL_CR);
        const int cb = wv * (CL_CR / 4);
#pragma unroll 2
        for (int c = cb; c < cb + CL_CR / 4; ++c) {
            const float xm = xl[c * CL_XS + lane], x0 = xl[c * CL_XS + lane + 1], x1 = xl[c * CL_XS + lane + 2];
            float wr[32];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(wl + c * 32 + i * 4);      // wave-uniform address: a broadcast
                wr[4 * i] = t.x; wr[4 * i + 1] = t.y; wr[4 * i + 2] = t.z; wr[4 * i + 3] = t.w;
            }
#pragma unroll
            for (int q = 0; q < R; ++q) {
#pragma unroll
                for (int r = 0; r < S; ++r) {
                    if (r < S / 2) acc[q][r] = fmaf(wr[q * K + r + S / 2], x0, fmaf(wr[q * K + r + S / 2 + S], xm, acc[q][r]));
                    else acc[q][r] = fmaf(wr[q * K + r - S / 2], x1, fmaf(wr[q * K + r + S / 2], x0, acc[q][r]));
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < R; ++q)
#pragma unroll
        for (int r = 0; r < S; ++r) red[(wv * NA + q * S + r) * 64 + lane] = acc[q][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NA / 4; ++i) {
        const int a = wv * (NA / 4) + i;
        float v = red[a * 64 + lane];
        v += red[(NA + a) * 64 + lane];
        v += red[(2 * NA + a) * 64 + lane];
        v += red[(3 * NA + a) * 64 + lane];
        const int q = a / S, r = a % S;
        if (in)
            y[((size_t)b * p.Cout + co0 + q) * p.Lout + (size_t)j * S + r] = ms_apply_act(v + (bias ? bias[co0 + q] : 0.f), p.act, p.slope);
    }
}

constexpr int CS_TL = 32;         // output positions per workgroup
constexpr int CS_CO = 4;          // output channels per workgroup
constexpr int CS_WMAX = 1024;     // Cin * K per output channel in LDS
constexpr int CS_XMAX = 8192;     // Cin * (CS_TL + K - 1)

template <int K>
__global__ __launch_bounds__(256) void k_conv_small(const ConvP p, const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ y) {
    constexpr int XS = CS_TL + K - 1;
    constexpr int XPT = CS_XMAX / 256, WPT = CS_CO * CS_WMAX / 256;
    __shared__ float xl[CS_XMAX];
    __shared__ float wl[CS_CO * CS_WMAX];
    __shared__ float red[128];
    const int t0 = blockIdx.x * CS_TL, co0 = blockIdx.y * CS_CO, b = blockIdx.z;
    const int tid = threadIdx.x;
    const int CK = p.Cin * K;
    const int nx = p.Cin * XS, nw = CS_CO * CK;         // the four output channels' weights are CONTIGUOUS in w
    const float* xb = x + (size_t)b * p.Cin * p.Lin;
    const float* ws = w + (size_t)co0 * CK;
    float xv[XPT], wv[WPT];
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        if (256 * u < nw) {
            const int e = tid + 256 * u;
            wv[u] = e < nw ? ws[e] : 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
        if (256 * u < nx) {
            const int e = tid + 256 * u;
            const int c = e / XS, m = e - c * XS;
            const int src = e < nx ? ms_src_index(t0 + m - p.pad, p.Lin, p.pad_mode) : -1;
            xv[u] = src >= 0 ? xb[(size_t)c * p.Lin + src] : 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        if (256 * u < nw) {
            const int e = tid + 256 * u;
            if (e < nw) wl[e] = wv[u];
        }
    }
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
        if (256 * u < nx) {
            const int e = tid + 256 * u;
            if (e < nx) xl[e] = xv[u];
        }
    }
    __syncthreads();
    const int tl = tid & 31, col = (tid >> 5) & 3, half = tid >> 7;
    const int ch = (p.Cin + 1) / 2;
    const int c_lo = half * ch, c_hi = min(p.Cin, c_lo + ch);
    const float* wr = wl + col * CK;
    float acc = 0.f;
#pragma unroll 2
    for (int c = c_lo; c < c_hi; ++c) {
#pragma unroll
        for (int k = 0; k < K; ++k) acc = fmaf(wr[c * K + k], xl[c * XS + tl + k], acc);
    }
    if (half == 1) red[tid - 128] = acc;
    __syncthreads();
    if (half == 0) {
        const int t = t0 + tl;
        if (t < p.Lout) {
            const float v = acc + red[tid] + (bias ? bias[co0 + col] : 0.f);
            y[((size_t)b * p.Cout + co0 + col) * p.Lout + t] = ms_apply_act(v, p.act, p.slope);
        }
    }
}

// (MSYNTH_SMALLROWS=0: the row-tile matrix kernels take these layers, as before r05 -- the A/B of DESIGN section 8)
bool small_rows_on() {
    static const int on = [] { const char* e = getenv("MSYNTH_SMALLROWS"); return (e && atoi(e) == 0) ? 0 : 1; }();
    return on != 0;
}

// 64 positions x (32 / 2 S) output channels per workgroup, while that is at most two workgroups per CU
bool ctl_plan(const ms_convt1d_desc* d, CtsP* q) {
    if (!d || !small_rows_on()) return false;
    const int S = d->stride;
    if (S != 8 && S != 2) return false;
    if (d->K != 2 * S || d->pad != S / 2) return false;
    if (d->in_act != MS_ACT_NONE && d->in_act != MS_ACT_LRELU) return false;
    if (d->act < MS_ACT_NONE || d->act > MS_ACT_TANH) return false;
    if (d->B <= 0 || d->B > 65535 || d->Cin < 16 || d->Cout <= 0) return false;
    // rows shorter than a tile leave lanes idle and the layer is then all weights (512 -> 256 on 32 positions: 8.4 MB; 22.9 us
    // here against 20.4 us on the row-tile kernel; a one-workgroup-per-output-channel form took 39 us -- DESIGN section 8)
    if (d->Lin < 64) return false;
    const int R = 16 / S;
    if (d->Cout % R) return false;
    const long long groups = (long long)ms_ceil_div(d->Lin, 64) * (d->Cout / R) * d->B;
    if (groups > 512 || d->Cout / R > 65535) return false;                            // beyond that the matrix kernels
    if ((long long)d->B * d->Cout * d->Lin * S >= (1ll << 31)) return false;
    q->B = d->B; q->Cin = d->Cin; q->Cout = d->Cout; q->Lin = d->Lin; q->Lout = d->Lin * S;
    q->act = d->act; q->in_act = d->in_act; q->slope = d->slope;
    return true;
}

}  // namespace

bool mss_convt_applicable(const ms_convt1d_desc* d) {
    CtsP q;
    return ctl_plan(d, &q);
}

const char* mss_convt_name(const ms_convt1d_desc* d) { return d->stride == 8 ? "k_convt_long<8, 2>" : "k_convt_long<2, 8>"; }

int mss_convt_fwd(const ms_convt1d_desc* d, const float* x, const float* w, const float* bias, float* y, hipStream_t s) {
    CtsP q;
    if (!ctl_plan(d, &q)) return MS_ERR_UNSUPPORTED;
    if (((uintptr_t)w) & 15) return MS_ERR_UNSUPPORTED;          // (16-byte weight loads: the caller falls through to the row-tile kernels)
    ms_note_kernel(0, "%s", mss_convt_name(d));
    const dim3 grid(ms_ceil_div(q.Lin, 64), q.Cout / (16 / d->stride), q.B);
    if (d->stride == 8) hipLaunchKernelGGL((k_convt_long<8, 2>), grid, dim3(256), 0, s, q, x, w, bias, y);
    else hipLaunchKernelGGL((k_convt_long<2, 8>), grid, dim3(256), 0, s, q, x, w, bias, y);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

bool mss_conv_applicable(const ConvP& p) {
    if (!small_rows_on()) return false;
    if (p.K != 7 || p.stride != 1 || p.dil != 1 || p.groups != 1 || p.in_act) return false;
    if (p.Cout % CS_CO || p.Cout < 128) return false;
    if (p.Cin * p.K > CS_WMAX || p.Cin * (CS_TL + p.K - 1) > CS_XMAX) return false;
    if (p.Lout != p.Lin + 2 * p.pad - (p.K - 1) || p.Lout <= 0) return false;
    if (p.pad_mode == MS_PAD_REFLECT && p.pad >= p.Lin) return false;
    const long long tiles = (long long)p.B * ms_ceil_div(p.Lout, CS_TL);
    return tiles <= 4 && p.B <= 65535;
}

const char* mss_conv_name(const ConvP&) { return "k_conv_small<7>"; }

int mss_conv_fwd(const ConvP& p, const float* x, const float* w, const float* bias, float* y, hipStream_t s) {
    if (!mss_conv_applicable(p)) return MS_ERR_UNSUPPORTED;
    const dim3 grid(ms_ceil_div(p.Lout, CS_TL), p.Cout / CS_CO, p.B);
    ms_note_kernel(0, "k_conv_small<7>");
    hipLaunchKernelGGL(k_conv_small<7>, grid, dim3(256), 0, s, p, x, w, bias, y);
    MS_CHECK_LAUNCH();
    return MS_OK;
}
