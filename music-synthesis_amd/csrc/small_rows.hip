// The generator at INFERENCE batch sizes (BASELINE config 2: B = 1, 32 frames; generator/full.py:23-40): the first conv
// (nn.Conv1d(80, 512, 7) behind ReflectionPad1d(3)) and the four upsampling layers (nn.ConvTranspose1d 512 -> 256 and 256 -> 128
// / k16 s8 on 32 and 256 positions, 128 -> 64 and 64 -> 32 / k4 s2 on 2048 and 4096), each + LeakyReLU.  With one batch row these
// are 18 / 67 / 134 / 67 / 34 MFLOP: nothing for the vector pipe, but 10 + 20 + 15 + 13 + 11 us on the row-tile matrix kernels,
// whose tiles (64+ rows x 128+ columns per workgroup, operands split for the 16-bit matrix pipe) give such a layer a few dozen
// workgroups and a launch-latency-sized critical path.  Here: plain fp32 FMA (exact fp32 products), up to 256 small
// workgroups, no weight image, no pack launch, one launch per layer.  Measured at B = 1 (tools/gfwd_b1.py --list, us):
// 10.1 -> 5.0, 20.4 -> 20.5, 15.0 -> 10.4, 13.1 -> 4.5, 11.2 -> 1.4; the 30-layer forward 237.5 -> 214.0 us.  Dispatched only while
// the whole layer is at most two workgroups per CU (conv: four tiles of 32 positions): the train step (B = 32) never comes here.
//
// k_convt_lanes<S>   ConvTranspose1d, kernel 2 S / stride S / padding S / 2:  y[b, co, j S + k - S/2] = sum_ci x[b, ci, j] w[ci, co, k].
//     A workgroup owns 16 input positions x 64 / S output channels; a wave's LANES are the (output channel, output phase r)
//     pairs and every lane keeps the 16 outputs j S + r of its pair: output j S + r takes taps (j, r + S/2) and
//     (j - 1, r + S/2 + S) for r < S/2, (j + 1, r - S/2) and (j, r + S/2) otherwise -- per input channel a lane reads its OWN two
//     weights and the 17 window values as wave broadcasts (two addresses per read: the two phase halves look one position apart)
//     for 32 FMAs.  (A first form with lanes = positions read all 32 weights of a channel as 16-byte broadcasts: 64 LDS cycles
//     per wave and channel, x 4 waves per CU = twice the FMA time.)  Input channels in rounds of 64 through LDS (64 x 128
//     weights, CONTIGUOUS 512-byte runs of w[ci, co, k]; window rows of 18); inside a round wave v walks channels
//     [16 v, 16 v + 16).  The four waves' partial sums meet in LDS; wave v finishes four of the 16 positions, adding in wave order
//     (deterministic), + bias + activation; lanes of one channel store 4 S contiguous bytes.  Partial channel groups, partial
//     rounds and tile tails are masked.
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
constexpr int CL_JT = 16;         // input positions per workgroup
constexpr int CL_XS = 18;         // window columns per channel: positions jt 16 - 1 .. jt 16 + 16

template <int S>
__global__ __launch_bounds__(256) void k_convt_lanes(const CtsP p, const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y) {
    constexpr int K = 2 * S, CG = 64 / S;              // a wave's lanes = CG output channels x S output phases
    constexpr int WROW = CG * K;                       // 128 weights per input channel, contiguous in w[ci, co, k]
    constexpr int NX = CL_CR * CL_XS;                  // 1152 window values per round
    constexpr int XPT = (NX + 255) / 256;              // 5
    constexpr int WPT = CL_CR * WROW / 4 / 256;        // 8 pieces of 16 bytes per thread and round
    __shared__ float xl[NX + 2];
    __shared__ __attribute__((aligned(16))) float wl[CL_CR * WROW];
    __shared__ float red[4 * CL_JT * 64];
    const int jt = blockIdx.x, co0 = blockIdx.y * CG, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane / S, r = lane % S;
    const int hi = r >= S / 2 ? 1 : 0;
    // output j S + r of position j: taps (j, r + S/2) and (j - 1, r + S/2 + S) for r < S/2, (j + 1, r - S/2) and (j, r + S/2) otherwise
    const int ka = hi ? r - S / 2 : r + S / 2, kb = hi ? r + S / 2 : r + S / 2 + S;
    float acc[CL_JT];
#pragma unroll
    for (int i = 0; i < CL_JT; ++i) acc[i] = 0.f;
    const float* xb = x + (size_t)b * p.Cin * p.Lin;
    float xv[XPT];
    f32x4 wq[WPT];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int e = tid + 256 * u;
            const int c = e / CL_XS, m = e - c * CL_XS;
            const int jj = jt * CL_JT - 1 + m;         // column m of a row holds x[jt 16 - 1 + m]; zero outside the row
            xv[u] = (e < NX && c0 + c < p.Cin && jj >= 0 && jj < p.Lin) ? xb[(size_t)(c0 + c) * p.Lin + jj] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int e = tid + 256 * u;
            const int c = e / (WROW / 4), part = e % (WROW / 4);
            const bool ok = c0 + c < p.Cin && co0 + part / (K / 4) < p.Cout;
            wq[u] = ok ? *reinterpret_cast<const f32x4*>(w + ((size_t)(c0 + c) * p.Cout + co0) * K + part * 4)
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
            const float wa = wl[c * WROW + col * K + ka], wb = wl[c * WROW + col * K + kb];
            const float* xr = xl + c * CL_XS + hi;      // two addresses per wave: a broadcast each
            float prev = xr[0];
#pragma unroll
            for (int i = 0; i < CL_JT; ++i) {
                const float cur = xr[i + 1];
                acc[i] = fmaf(wa, cur, fmaf(wb, prev, acc[i]));
                prev = cur;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < CL_JT; ++i) red[(wv * CL_JT + i) * 64 + lane] = acc[i];
    __syncthreads();
    const int co = co0 + col;
    const float bv = (bias && co < p.Cout) ? bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < CL_JT / 4; ++i) {
        const int jl = wv * (CL_JT / 4) + i;
        float v = red[jl * 64 + lane];
        v += red[(CL_JT + jl) * 64 + lane];
        v += red[(2 * CL_JT + jl) * 64 + lane];
        v += red[(3 * CL_JT + jl) * 64 + lane];
        const int j = jt * CL_JT + jl;
        if (j < p.Lin && co < p.Cout) y[((size_t)b * p.Cout + co) * p.Lout + (size_t)j * S + r] = ms_apply_act(v + bv, p.act, p.slope);
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

// 16 positions x (64 / S) output channels per workgroup, while that is at most two workgroups per CU
bool ctl_plan(const ms_convt1d_desc* d, CtsP* q) {
    if (!d || !small_rows_on()) return false;
    const int S = d->stride;
    if (S != 8 && S != 2) return false;
    if (d->K != 2 * S || d->pad != S / 2) return false;
    if (d->in_act != MS_ACT_NONE && d->in_act != MS_ACT_LRELU) return false;
    if (d->act < MS_ACT_NONE || d->act > MS_ACT_TANH) return false;
    if (d->B <= 0 || d->B > 65535 || d->Cin < 16 || d->Cout <= 0 || d->Lin <= 0) return false;
    const int CG = 64 / S;
    const long long groups = (long long)ms_ceil_div(d->Lin, CL_JT) * ms_ceil_div(d->Cout, CG) * d->B;
    if (groups > 512 || ms_ceil_div(d->Cout, CG) > 65535) return false;               // beyond that the matrix kernels
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

const char* mss_convt_name(const ms_convt1d_desc* d) { return d->stride == 8 ? "k_convt_lanes<8>" : "k_convt_lanes<2>"; }

int mss_convt_fwd(const ms_convt1d_desc* d, const float* x, const float* w, const float* bias, float* y, hipStream_t s) {
    CtsP q;
    if (!ctl_plan(d, &q)) return MS_ERR_UNSUPPORTED;
    if (((uintptr_t)w) & 15) return MS_ERR_UNSUPPORTED;          // (16-byte weight loads: the caller falls through to the row-tile kernels)
    ms_note_kernel(0, "%s", mss_convt_name(d));
    const dim3 grid(ms_ceil_div(q.Lin, CL_JT), ms_ceil_div(q.Cout, 64 / d->stride), q.B);
    if (d->stride == 8) hipLaunchKernelGGL(k_convt_lanes<8>, grid, dim3(256), 0, s, q, x, w, bias, y);
    else hipLaunchKernelGGL(k_convt_lanes<2>, grid, dim3(256), 0, s, q, x, w, bias, y);
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
