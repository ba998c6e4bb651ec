// "Thin" convolutions: one side of the layer has a single channel, so the layer is a stream over
// the many-channel tensor and bound by HBM, not by the matrix cores:
//   * the generator's last conv   Conv1d(32, 1, 7, padding=3) + tanh  (reference generator/full.py:43-44)
//   * the discriminator's first   Conv1d(1, 16, 15, padding=7)        (reference discriminator/full.py:14)
// Three kernels cover their passes (stride 1, dilation 1, zero padding, Lout == Lin):
//   k_thin_reduce  many -> 1 : forward of Cout == 1, backward-data of Cin == 1
//   k_thin_expand  1 -> many : backward-data of Cout == 1, forward of Cin == 1
//   k_thin_wgrad            : weight/bias gradients of both
// Every thread owns 4 consecutive samples; the one-channel operand ("thin") is read as a window of
// K + 3 samples, the many-channel operand ("stream") as one 16-byte access per channel.
#include "ms_common.h"
#include "conv_thin.h"
#include <stdlib.h>

namespace {

constexpr int TS = 1024;                 // stream samples per workgroup chunk (weight-grad kernel)

// ------------------------------------------------------------------ many -> 1
// out[b, 0, t] = act(bias + res + sum_c sum_j wt[c, j] * S'[b, c, t + j - off]),  S' = S * act'(Sact)
// (forward: wt = w[0, c, j];  backward-data of Cin == 1: wt = w[c, 0, K-1-j];  off = (K-1)/2 both ways)
// A workgroup owns 256 consecutive samples of one batch row; its 4 waves each sum a quarter of the
// channels (CW channels, all loads of a wave issued as one batch) and are combined through LDS.
template <int K, bool FLIP, bool VEC, int CW>
__global__ __launch_bounds__(256) void k_thin_reduce(int B, int C, int L, int s_kind, int act,
                                                    float slope, const float* __restrict__ S,
                                                    const float* __restrict__ Sact,
                                                    const float* __restrict__ w,
                                                    const float* __restrict__ bias,
                                                    const float* __restrict__ res,
                                                    float* __restrict__ out) {
    constexpr int OFF = (K - 1) / 2;
    constexpr int HV = (OFF + 3) / 4;                   // 16-byte vectors on each side of the centre one
    constexpr int NV = 2 * HV + 1;
    constexpr int NW = VEC ? NV * 4 : K + 3;            // window floats per channel
    constexpr int W0 = VEC ? 4 * HV : OFF;              // window index of S'[t]
    __shared__ float red[4][256 + 4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tiles = (L + 255) / 256;
    const int b = blockIdx.x / tiles, t = (blockIdx.x - b * tiles) * 256 + lane * 4;
    const float* Sq = Sact ? Sact : S;
    const int kind = Sact ? s_kind : MS_ACT_NONE;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float win[CW][NW];
#pragma unroll
    for (int cc = 0; cc < CW; ++cc) {
        const int c = wid * CW + cc;
        const size_t row = ((size_t)b * C + (c < C ? c : 0)) * L;
        if (VEC) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int tq = t + 4 * (q - HV);
                const bool ok = tq >= 0 && tq < L;     // L % 4 == 0: a vector is all in or all out
                const float4 v = *reinterpret_cast<const float4*>(S + row + (ok ? tq : 0));
                win[cc][4 * q + 0] = v.x; win[cc][4 * q + 1] = v.y; win[cc][4 * q + 2] = v.z; win[cc][4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < K + 3; ++i) {
                const int ti = t + i - OFF;
                const bool ok = ti >= 0 && ti < L;
                win[cc][i] = S[row + (ok ? ti : 0)];
            }
        }
    }
    if (Sact) {       // activation-derivative operand (wave-uniform branch; second batch of loads)
        float wa[CW][NW];
#pragma unroll
        for (int cc = 0; cc < CW; ++cc) {
            const int c = wid * CW + cc;
            const size_t row = ((size_t)b * C + (c < C ? c : 0)) * L;
            if (VEC) {
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const int tq = t + 4 * (q - HV);
                    const bool ok = tq >= 0 && tq < L;
                    const float4 v = *reinterpret_cast<const float4*>(Sq + row + (ok ? tq : 0));
                    wa[cc][4 * q + 0] = v.x; wa[cc][4 * q + 1] = v.y; wa[cc][4 * q + 2] = v.z; wa[cc][4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < K + 3; ++i) {
                    const int ti = t + i - OFF;
                    const bool ok = ti >= 0 && ti < L;
                    wa[cc][i] = Sq[row + (ok ? ti : 0)];
                }
            }
        }
#pragma unroll
        for (int cc = 0; cc < CW; ++cc)
#pragma unroll
            for (int i = 0; i < NW; ++i) win[cc][i] = ms_act_grad(win[cc][i], wa[cc][i], kind, slope);
    }
#pragma unroll
    for (int cc = 0; cc < CW; ++cc) {
        const int c = wid * CW + cc;
        if (c >= C) continue;                         // wave-uniform
#pragma unroll
        for (int i = 0; i < NW; ++i) {                // zero padding outside the row
            const int ti = t + i - W0;
            if (ti < 0 || ti >= L) win[cc][i] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float wv = w[c * K + (FLIP ? K - 1 - j : j)];     // wave-uniform: scalar load
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(wv, win[cc][W0 + e + j - OFF], acc[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wid][lane * 4 + e] = acc[e];
    __syncthreads();
    const int to = (blockIdx.x - b * tiles) * 256 + threadIdx.x;
    if (to < L) {
        const int i = threadIdx.x;
        float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        v = ms_apply_act(v + (bias ? bias[0] : 0.f), act, slope);
        const size_t o = (size_t)b * L + to;
        if (res) v += res[o];
        out[o] = v;
    }
}

// ------------------------------------------------------------------ 1 -> many
// out[b, c, s] = act(bias[c] + sum_j wt[c, j] * T'[b, 0, s + j - off]) + add,  T' = T * act'(Tact)
// (backward-data of Cout == 1: wt = w[0, c, K-1-j];  forward of Cin == 1: wt = w[c, 0, j])
// Same tiling: every wave reads the (tiny) window itself and writes a quarter of the channels.
template <int K, bool FLIP, bool VEC, int CW>
__global__ __launch_bounds__(256) void k_thin_expand(int B, int C, int L, int t_kind, int act,
                                                    float slope, const float* __restrict__ T,
                                                    const float* __restrict__ Tact,
                                                    const float* __restrict__ w,
                                                    const float* __restrict__ bias,
                                                    const float* __restrict__ add,
                                                    float* __restrict__ out) {
    constexpr int OFF = (K - 1) / 2;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tiles = (L + 255) / 256;
    const int b = blockIdx.x / tiles, s = (blockIdx.x - b * tiles) * 256 + lane * 4;
    if (s >= L) return;
    const float* Tq = Tact ? Tact : T;
    const int kind = Tact ? t_kind : MS_ACT_NONE;
    float v[K + 3], a[K + 3], win[K + 3];
#pragma unroll
    for (int i = 0; i < K + 3; ++i) {
        const int ti = s + i - OFF;
        const bool ok = ti >= 0 && ti < L;
        v[i] = T[(size_t)b * L + (ok ? ti : 0)];
        a[i] = Tq[(size_t)b * L + (ok ? ti : 0)];
    }
#pragma unroll
    for (int i = 0; i < K + 3; ++i) {
        const int ti = s + i - OFF;
        const bool ok = ti >= 0 && ti < L;
        win[i] = ok ? ms_act_grad(v[i], a[i], kind, slope) : 0.f;
    }
    float4 av[CW];
    if (add && VEC) {
#pragma unroll
        for (int cc = 0; cc < CW; ++cc) {
            const int c = wid * CW + cc;
            av[cc] = *reinterpret_cast<const float4*>(add + ((size_t)b * C + (c < C ? c : 0)) * L + s);
        }
    }
#pragma unroll
    for (int cc = 0; cc < CW; ++cc) {
        const int c = wid * CW + cc;
        if (c >= C) continue;                         // wave-uniform
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float wv = w[c * K + (FLIP ? K - 1 - j : j)];     // wave-uniform: scalar load
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(wv, win[e + j], acc[e]);
        }
        const float bv = bias ? bias[c] : 0.f;
        const size_t o = ((size_t)b * C + c) * L + s;
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = ms_apply_act(acc[e] + bv, act, slope);
        if (VEC) {
            if (add) { r[0] += av[cc].x; r[1] += av[cc].y; r[2] += av[cc].z; r[3] += av[cc].w; }
            *reinterpret_cast<float4*>(out + o) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (s + e < L) out[o + e] = r[e] + (add ? add[o + e] : 0.f);
        }
    }
}

// ------------------------------------------------------------------ weight / bias gradients
// ROLE 0 (Cout == 1): stream = x (C = Cin), thin = gy * act'(y):  gw[0, c, j] = sum S[b,c,s] T'[b, s - j + pad]
// ROLE 1 (Cin  == 1): stream = gy * act'(y) (C = Cout), thin = x: gw[c, 0, j] = sum S'[b,c,s] T[b, s + j - pad]
// A workgroup owns one (batch row, 1024-sample chunk): the thin window goes to LDS once, wave w
// accumulates channels [w*CPW, (w+1)*CPW) over the chunk in registers (CPW*K accumulators per lane),
// and the lanes are combined once at the end.  One partial row per workgroup; k_reduce_partials_wave
// sums the rows.
template <int K, int ROLE, int CPW, bool VEC>
__global__ __launch_bounds__(256) void k_thin_wgrad(int B, int C, int L, int pad, int act, float slope,
                                                   int schunks, const float* __restrict__ S,
                                                   const float* __restrict__ Sact,
                                                   const float* __restrict__ T,
                                                   const float* __restrict__ Tact,
                                                   float* __restrict__ partial, size_t pstride) {
    constexpr int TW = TS + K - 1;
    constexpr int NT = (TW + 255) / 256;
    __shared__ float Tl[NT * 256];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x / schunks, s0 = (blockIdx.x - b * schunks) * TS;
    const int lo = ROLE == 0 ? s0 - (K - 1) + pad : s0 - pad;
    const float* Tq = Tact ? Tact : T;
    const int t_kind = Tact ? act : MS_ACT_NONE;
    const float* Sq = Sact ? Sact : S;
    const int s_kind = Sact ? act : MS_ACT_NONE;
    float bs = 0.f;
    {
        float tv[NT], ta[NT];
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const int i = tid + it * 256, t = lo + i;
            const bool ok = i < TW && t >= 0 && t < L;
            tv[it] = T[(size_t)b * L + (ok ? t : 0)];
            ta[it] = Tq[(size_t)b * L + (ok ? t : 0)];
        }
#pragma unroll
        for (int it = 0; it < NT; ++it) {
            const int i = tid + it * 256, t = lo + i;
            const bool ok = i < TW && t >= 0 && t < L;
            const float v = ok ? ms_act_grad(tv[it], ta[it], t_kind, slope) : 0.f;
            Tl[i] = v;
            if (ROLE == 0 && ok && t >= s0 && t < s0 + TS) bs += v;     // bias grad: each sample once
        }
    }
    __syncthreads();
    float acc[CPW][K];
    float bsum[CPW];
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
        bsum[cc] = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) acc[cc][j] = 0.f;
    }
    const int c0 = wid * CPW;
#pragma unroll 1
    for (int pass = 0; pass < TS / 256; ++pass) {
        const int sl = pass * 256 + lane * 4, s = s0 + sl;
        float win[K + 3];
#pragma unroll
        for (int i = 0; i < K + 3; ++i) win[i] = Tl[sl + i];
        float4 sv[CPW], sa[CPW];
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            const int c = c0 + cc;
            const bool ok = c < C && s < L;
            const size_t o = ok ? ((size_t)b * C + c) * L + s : 0;
            if (VEC) {                           // L % 4 == 0: 4 samples all in or all out
                sv[cc] = *reinterpret_cast<const float4*>(S + o);
                sa[cc] = *reinterpret_cast<const float4*>(Sq + o);
            } else {
                const size_t o1 = ok && s + 1 < L ? o + 1 : o, o2 = ok && s + 2 < L ? o + 2 : o,
                             o3 = ok && s + 3 < L ? o + 3 : o;
                sv[cc] = make_float4(S[o], S[o1], S[o2], S[o3]);
                sa[cc] = make_float4(Sq[o], Sq[o1], Sq[o2], Sq[o3]);
            }
        }
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            const int c = c0 + cc;
            const float xv[4] = {sv[cc].x, sv[cc].y, sv[cc].z, sv[cc].w};
            const float xa[4] = {sa[cc].x, sa[cc].y, sa[cc].z, sa[cc].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = c < C && s + e < L;
                const float v = ok ? ms_act_grad(xv[e], xa[e], s_kind, slope) : 0.f;
                bsum[cc] += v;
#pragma unroll
                for (int j = 0; j < K; ++j)
                    acc[cc][j] = fmaf(v, win[ROLE == 0 ? e + K - 1 - j : e + j], acc[cc][j]);
            }
        }
    }
    float* prow = partial + (size_t)blockIdx.x * pstride;
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
        const int c = c0 + cc;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float r = ms_wave_sum(acc[cc][j]);
            if (lane == 0 && c < C) prow[c * K + j] = r;
        }
        if (ROLE == 1) {
            const float r = ms_wave_sum(bsum[cc]);
            if (lane == 0 && c < C) prow[(size_t)C * K + c] = r;
        }
    }
    if (ROLE == 0) {
        const float tot = ms_block_sum(bs, red);
        if (tid == 0) prow[(size_t)C * K] = tot;
    }
}

// ------------------------------------------------------------------ deep one-output conv on short rows
// The judge conv (1024 -> 1, k3, L = 32 / 17 / 9; discriminator/full.py:22): 12 MFLOP per launch -- as a
// row-tile GEMM with split-K it cost 31 us + a finish launch.  Here a workgroup owns one batch row: thread
// i walks channels i, i+256, ... over the whole (short) row keeping L accumulators, then the block sums.
template <int LT>
__global__ __launch_bounds__(256) void k_thin_short_fwd(int C, int L, int act, float slope,
                                                       const float* __restrict__ x,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ y) {
    __shared__ float red[4][LT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, b = blockIdx.x;
    float acc[LT];
#pragma unroll
    for (int t = 0; t < LT; ++t) acc[t] = 0.f;
    for (int c = tid; c < C; c += 256) {
        const float w0 = w[c * 3], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2];
        const float* row = x + ((size_t)b * C + c) * L;
        float v[LT + 2];
        v[0] = 0.f;
#pragma unroll
        for (int t = 0; t < LT; ++t) v[t + 1] = t < L ? row[t < L ? t : 0] : 0.f;
        v[LT + 1] = 0.f;
#pragma unroll
        for (int t = 0; t < LT; ++t) acc[t] += w0 * v[t] + w1 * v[t + 1] + w2 * v[t + 2];
    }
#pragma unroll
    for (int t = 0; t < LT; ++t) {
        const float sred = ms_wave_sum(acc[t]);
        if (lane == 0) red[wid][t] = sred;
    }
    __syncthreads();
    if (tid < L) {
        const float tot = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]) + (bias ? bias[0] : 0.f);
        y[(size_t)b * L + tid] = ms_apply_act(tot, act, slope);
    }
}

bool thin_common(const ConvP& p) {
    return p.groups == 1 && p.stride == 1 && p.dil == 1 && p.pad_mode == MS_PAD_ZERO &&
           p.Lout == p.Lin && !p.in_act && 2 * p.pad == p.K - 1 && p.B <= 65535;
}
bool role0(const ConvP& p) { return thin_common(p) && p.Cout == 1 && p.K == 7 && p.Cin <= 32 && p.Cin >= 4; }
bool role1(const ConvP& p) { return thin_common(p) && p.Cin == 1 && p.K == 15 && p.Cout <= 16 && p.Cout >= 4; }
bool aligned16(const void* a) { return (((uintptr_t)a) & 15) == 0; }
// deep contraction, one output channel, short rows (the judge conv)
bool role_short(const ConvP& p) {
    return thin_common(p) && p.Cout == 1 && p.K == 3 && p.Cin >= 256 && p.Lin <= 64;
}

}  // namespace

bool mst_fwd_applicable(const ConvP& p) { return role0(p) || role_short(p); }
bool mst_fwd_short_applicable(const ConvP& p) { return role_short(p); }
bool mst_bwd_data_applicable(const ConvP& p) { return role0(p) || role1(p); }
bool mst_bwd_weight_applicable(const ConvP& p) { return role0(p) || role1(p); }

const char* mst_fwd_name(const ConvP& p) { return role_short(p) ? "k_thin_short_fwd" : "k_thin_reduce<7, false>"; }
const char* mst_bwd_data_name(const ConvP& p) { return role0(p) ? "k_thin_expand<7, true>" : "k_thin_reduce<15, true>"; }
const char* mst_bwd_weight_name(const ConvP& p) { return role0(p) ? "k_thin_wgrad<7, 0, 8>" : "k_thin_wgrad<15, 1, 4>"; }

static unsigned thin_grid(const ConvP& p) { return (unsigned)(p.B * ms_ceil_div(p.Lin, 256)); }

int mst_conv1d_fwd(const ConvP& p, const float* x, const float* w, const float* bias,
                   const float* residual, float* y, hipStream_t s) {
    if (role_short(p)) {
        if (residual) return MS_ERR_UNSUPPORTED;
        const dim3 g((unsigned)p.B);
        if (p.Lin <= 16)
            hipLaunchKernelGGL(k_thin_short_fwd<16>, g, dim3(256), 0, s, p.Cin, p.Lin, p.act, p.slope, x, w, bias, y);
        else if (p.Lin <= 32)
            hipLaunchKernelGGL(k_thin_short_fwd<32>, g, dim3(256), 0, s, p.Cin, p.Lin, p.act, p.slope, x, w, bias, y);
        else
            hipLaunchKernelGGL(k_thin_short_fwd<64>, g, dim3(256), 0, s, p.Cin, p.Lin, p.act, p.slope, x, w, bias, y);
        MS_CHECK_LAUNCH();
        return MS_OK;
    }
    const bool vec = p.Lin % 4 == 0 && aligned16(x);
    const dim3 grid(thin_grid(p));
    if (vec)
        hipLaunchKernelGGL((k_thin_reduce<7, false, true, 8>), grid, dim3(256), 0, s, p.B, p.Cin, p.Lin,
                           MS_ACT_NONE, p.act, p.slope, x, (const float*)nullptr, w, bias, residual, y);
    else
        hipLaunchKernelGGL((k_thin_reduce<7, false, false, 8>), grid, dim3(256), 0, s, p.B, p.Cin, p.Lin,
                           MS_ACT_NONE, p.act, p.slope, x, (const float*)nullptr, w, bias, residual, y);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int mst_conv1d_bwd_data(const ConvP& p, const float* gy, const float* y_act, const float* w,
                        const float* gx_add, float* gx, hipStream_t s) {
    const dim3 grid(thin_grid(p));
    if (role0(p)) {      // 1 -> Cin
        const bool vec = p.Lin % 4 == 0 && aligned16(gx) && (!gx_add || aligned16(gx_add));
        if (vec)
            hipLaunchKernelGGL((k_thin_expand<7, true, true, 8>), grid, dim3(256), 0, s, p.B, p.Cin, p.Lin,
                               p.act, MS_ACT_NONE, p.slope, gy, y_act, w, (const float*)nullptr, gx_add, gx);
        else
            hipLaunchKernelGGL((k_thin_expand<7, true, false, 8>), grid, dim3(256), 0, s, p.B, p.Cin, p.Lin,
                               p.act, MS_ACT_NONE, p.slope, gy, y_act, w, (const float*)nullptr, gx_add, gx);
    } else {             // Cout -> 1
        const bool vec = p.Lin % 4 == 0 && aligned16(gy) && (!y_act || aligned16(y_act));
        if (vec)
            hipLaunchKernelGGL((k_thin_reduce<15, true, true, 4>), grid, dim3(256), 0, s, p.B, p.Cout, p.Lin,
                               p.act, MS_ACT_NONE, p.slope, gy, y_act, w, (const float*)nullptr, gx_add, gx);
        else
            hipLaunchKernelGGL((k_thin_reduce<15, true, false, 4>), grid, dim3(256), 0, s, p.B, p.Cout, p.Lin,
                               p.act, MS_ACT_NONE, p.slope, gy, y_act, w, (const float*)nullptr, gx_add, gx);
    }
    MS_CHECK_LAUNCH();
    return MS_OK;
}

static int thin_nchunks(const ConvP& p) { return p.B * ms_ceil_div(p.Lin, TS); }
static size_t thin_pstride(const ConvP& p) { return (size_t)p.Cout * p.Cin * p.K + p.Cout; }

size_t mst_bwd_weight_ws(const ConvP& p) {
    return (size_t)thin_nchunks(p) * thin_pstride(p) * sizeof(float);
}

int mst_conv1d_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act,
                          float* gw, float* gb, float beta, void* ws, size_t ws_bytes,
                          hipStream_t s) {
    if (!ws || ws_bytes < mst_bwd_weight_ws(p)) return MS_ERR_WORKSPACE;
    float* partial = (float*)ws;
    const int nch = thin_nchunks(p), sch = ms_ceil_div(p.Lin, TS);
    const size_t ps = thin_pstride(p);
    const dim3 grid(nch);
    if (role0(p)) {
        const bool vec = p.Lin % 4 == 0 && aligned16(x);
        if (vec)
            hipLaunchKernelGGL((k_thin_wgrad<7, 0, 8, true>), grid, dim3(256), 0, s, p.B, p.Cin, p.Lin, p.pad,
                               p.act, p.slope, sch, x, (const float*)nullptr, gy, y_act, partial, ps);
        else
            hipLaunchKernelGGL((k_thin_wgrad<7, 0, 8, false>), grid, dim3(256), 0, s, p.B, p.Cin, p.Lin, p.pad,
                               p.act, p.slope, sch, x, (const float*)nullptr, gy, y_act, partial, ps);
    } else {
        const bool vec = p.Lin % 4 == 0 && aligned16(gy) && (!y_act || aligned16(y_act));
        if (vec)
            hipLaunchKernelGGL((k_thin_wgrad<15, 1, 4, true>), grid, dim3(256), 0, s, p.B, p.Cout, p.Lin, p.pad,
                               p.act, p.slope, sch, gy, y_act, x, (const float*)nullptr, partial, ps);
        else
            hipLaunchKernelGGL((k_thin_wgrad<15, 1, 4, false>), grid, dim3(256), 0, s, p.B, p.Cout, p.Lin, p.pad,
                               p.act, p.slope, sch, gy, y_act, x, (const float*)nullptr, partial, ps);
    }
    MS_CHECK_LAUNCH();
    return msk_reduce_partials(partial, ps, nch, (size_t)p.Cout * p.Cin * p.K, p.Cout, gw, gb, beta, s);
}

// ---------------------------------------------------------------- ConvTranspose1d with ONE output channel
// (stride 2, kernel 4, padding 1: the last layer of the stage-1 2-D generator as a transposed conv over lines,
// reference featuregenerator/upscale.py:77-112).  HBM-bound streams: the forward reads C x L inputs per line and
// writes 2 L outputs, the weight gradient reads the same inputs and reduces them to C x 4 numbers.
//   y[2q]   = b + sum_ci w[ci][1] x[ci][q] + w[ci][3] x[ci][q-1]
//   y[2q+1] = b + sum_ci w[ci][2] x[ci][q] + w[ci][0] x[ci][q+1]
// p = mirrored conv: Cin_T = p.Cout, Cout_T = p.Cin = 1, Lin_T = p.Lout.
namespace {

// one thread: 4 consecutive input positions of one line -> 8 outputs; lanes run along the line (coalesced rows)
template <bool INA>
__global__ __launch_bounds__(256) void k_convt1_fwd(int B, int C, int L, int act, float slope,
                                                   const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ y) {
    const int nq = (L + 3) / 4;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)B * nq) return;
    const int b = (int)(gid / nq), q0 = (int)(gid - (long long)b * nq) * 4;
    const float* xb = x + (size_t)b * C * L;
    float ev[4] = {0.f, 0.f, 0.f, 0.f}, od[4] = {0.f, 0.f, 0.f, 0.f};
    const bool full = q0 + 3 < L && (L % 4 == 0);
    for (int ci = 0; ci < C; ++ci) {
        const float* xr = xb + (size_t)ci * L;
        float v[6];                                     // x[q0 - 1 .. q0 + 4]
        if (full) {
            const float4 c4 = *reinterpret_cast<const float4*>(xr + q0);
            v[1] = c4.x; v[2] = c4.y; v[3] = c4.z; v[4] = c4.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[1 + e] = q0 + e < L ? xr[q0 + e] : 0.f;
        }
        v[0] = q0 > 0 ? xr[q0 - 1] : 0.f;
        v[5] = q0 + 4 < L ? xr[q0 + 4] : 0.f;
        if (INA) {
#pragma unroll
            for (int e = 0; e < 6; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        }
        const float w0 = w[ci * 4], w1 = w[ci * 4 + 1], w2 = w[ci * 4 + 2], w3 = w[ci * 4 + 3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ev[e] = fmaf(w1, v[1 + e], fmaf(w3, v[e], ev[e]));
            od[e] = fmaf(w2, v[1 + e], fmaf(w0, v[2 + e], od[e]));
        }
    }
    const float bv = bias ? bias[0] : 0.f;
    float* yr = y + (size_t)b * 2 * L + 2 * q0;
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[2 * e] = ms_apply_act(ev[e] + bv, act, slope);
        o[2 * e + 1] = ms_apply_act(od[e] + bv, act, slope);
    }
    if (full && (((uintptr_t)yr) & 15) == 0) {
        *reinterpret_cast<float4*>(yr) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(yr + 4) = make_float4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (q0 + e / 2 < L) yr[e] = o[e];
    }
}

// weight gradient: gw[ci][k] = sum_{b,q} x[ci][q] gp[2q + k - 1].  A workgroup owns a slice of lines.  Channels are
// walked in groups of 8: a thread accumulates its positions of ALL the workgroup's lines into 8 x 4 registers and the
// group is reduced over the workgroup once (reducing per line cost 24 shuffles per channel and line: slower than the
// direct kernel it replaced).  One slab [C][4] per workgroup, deterministic reduce after.
template <bool INA>
__global__ __launch_bounds__(256) void k_convt1_wgrad(int B, int C, int L, int act, float slope, int lines_per_wg,
                                                     const float* __restrict__ x, const float* __restrict__ gy,
                                                     const float* __restrict__ y_act, float* __restrict__ partial) {
    constexpr int CG = 8;
    __shared__ float wsum[4][CG * 4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b_beg = blockIdx.x * lines_per_wg, b_end = min(B, b_beg + lines_per_wg);
    const int kind = y_act ? act : MS_ACT_NONE;
    for (int cg = 0; cg < C; cg += CG) {
        float acc[CG][4];
#pragma unroll
        for (int c = 0; c < CG; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[c][k] = 0.f;
        for (int b = b_beg; b < b_end; ++b) {
            const float* gr = gy + (size_t)b * 2 * L;
            const float* ar = y_act ? y_act + (size_t)b * 2 * L : gr;
            for (int q = tid; q < L; q += 256) {
                float g[4];                             // gp[2q - 1 .. 2q + 2]
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = 2 * q + k - 1;
                    g[k] = (t >= 0 && t < 2 * L) ? ms_act_grad(gr[t], ar[t], kind, slope) : 0.f;
                }
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    float v = cg + c < C ? x[((size_t)b * C + cg + c) * L + q] : 0.f;
                    if (INA) v = v > 0.f ? v : v * slope;
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[c][k] = fmaf(v, g[k], acc[c][k]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CG; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float r = acc[c][k];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) r += __shfl_xor(r, o, 64);
                if (lane == 0) wsum[wv][c * 4 + k] = r;
            }
        __syncthreads();
        if (tid < CG * 4 && cg + tid / 4 < C)
            partial[(size_t)blockIdx.x * C * 4 + (size_t)cg * 4 + tid] =
                (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]);
        __syncthreads();
    }
}

// The same with 16-byte loads (L % 4 == 0): a thread owns FOUR consecutive positions q .. q + 3 of a line (one float4 of x per
// channel, gradient samples 2 q - 1 .. 2 q + 8), a workgroup walks its lines 256 / (L / 4) at a time -- four times the bytes in
// flight per wave of the dword form above, which ran at 1.2 TB/s (341 us for the stage-1 generator's last layer: 403 MB).
template <bool INA>
__global__ __launch_bounds__(256) void k_convt1_wgrad_v4(int B, int C, int L, int act, float slope, int lines_per_wg,
                                                        const float* __restrict__ x, const float* __restrict__ gy,
                                                        const float* __restrict__ y_act, float* __restrict__ partial) {
    constexpr int CG = 8;
    __shared__ float wsum[4][CG * 4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b_beg = blockIdx.x * lines_per_wg, b_end = min(B, b_beg + lines_per_wg);
    const int kind = y_act ? act : MS_ACT_NONE;
    const int L4 = L / 4;                               // vectors per line
    const int lines_par = 256 / L4 > 0 ? 256 / L4 : 1;  // lines a pass of the workgroup covers (L4 <= 256)
    const int my_line = tid / L4, my_v = tid - my_line * L4;
    const bool active = my_line < lines_par;
    for (int cg = 0; cg < C; cg += CG) {
        float acc[CG][4];
#pragma unroll
        for (int c = 0; c < CG; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[c][k] = 0.f;
        for (int b0 = b_beg; b0 < b_end; b0 += lines_par) {
            const int b = b0 + my_line;
            if (!active || b >= b_end) continue;
            const int q = 4 * my_v;
            const float* gr = gy + (size_t)b * 2 * L + 2 * q;
            const float* ar = y_act ? y_act + (size_t)b * 2 * L + 2 * q : gr;
            float g[10];                                // gp[2q - 1 .. 2q + 8]
            const float4 g0 = *reinterpret_cast<const float4*>(gr), g1 = *reinterpret_cast<const float4*>(gr + 4);
            const float4 a0 = *reinterpret_cast<const float4*>(ar), a1 = *reinterpret_cast<const float4*>(ar + 4);
            g[1] = ms_act_grad(g0.x, a0.x, kind, slope); g[2] = ms_act_grad(g0.y, a0.y, kind, slope);
            g[3] = ms_act_grad(g0.z, a0.z, kind, slope); g[4] = ms_act_grad(g0.w, a0.w, kind, slope);
            g[5] = ms_act_grad(g1.x, a1.x, kind, slope); g[6] = ms_act_grad(g1.y, a1.y, kind, slope);
            g[7] = ms_act_grad(g1.z, a1.z, kind, slope); g[8] = ms_act_grad(g1.w, a1.w, kind, slope);
            g[0] = q > 0 ? ms_act_grad(gr[-1], ar[-1], kind, slope) : 0.f;
            g[9] = q + 4 < L ? ms_act_grad(gr[8], ar[8], kind, slope) : 0.f;
            float4 xv[CG];
#pragma unroll
            for (int c = 0; c < CG; ++c)
                xv[c] = cg + c < C ? *reinterpret_cast<const float4*>(x + ((size_t)b * C + cg + c) * L + q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                float v[4] = {xv[c].x, xv[c].y, xv[c].z, xv[c].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (INA) v[j] = v[j] > 0.f ? v[j] : v[j] * slope;
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[c][k] = fmaf(v[j], g[2 * j + k], acc[c][k]);      // sample 2 (q + j) + k - 1
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CG; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float r = acc[c][k];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) r += __shfl_xor(r, o, 64);
                if (lane == 0) wsum[wv][c * 4 + k] = r;
            }
        __syncthreads();
        if (tid < CG * 4 && cg + tid / 4 < C)
            partial[(size_t)blockIdx.x * C * 4 + (size_t)cg * 4 + tid] =
                (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]);
        __syncthreads();
    }
}

bool convt1_geom(const ConvP& p) {
    return p.Cin == 1 && p.groups == 1 && p.stride == 2 && p.K == 4 && p.pad == 1 && p.dil == 1 &&
           p.Lin == 2 * p.Lout && p.Cout >= 1 && p.Cout <= 512 && (long long)p.B * ((p.Lout + 3) / 4) < (1ll << 31);
}
int convt1_lines_per_wg(const ConvP& p) { return ms_ceil_div(p.B, 512); }

}  // namespace

bool mst_convt1_applicable(const ConvP& p) {
    const char* e = getenv("MSYNTH_CONVT1");          // tuning / test switch (0: direct kernels)
    if (e && atoi(e) == 0) return false;
    return convt1_geom(p);
}
const char* mst_convt1_fwd_name() { return "k_convt1_fwd"; }
const char* mst_convt1_wgrad_name() { return "k_convt1_wgrad"; }

int mst_convt1_fwd(const ConvP& p, const float* x, const float* w, const float* bias, float* y, hipStream_t s) {
    const long long n = (long long)p.B * ((p.Lout + 3) / 4);
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (p.in_act) hipLaunchKernelGGL((k_convt1_fwd<true>), dim3(nb), dim3(256), 0, s, p.B, p.Cout, p.Lout, p.act, p.slope, x, w, bias, y);
    else hipLaunchKernelGGL((k_convt1_fwd<false>), dim3(nb), dim3(256), 0, s, p.B, p.Cout, p.Lout, p.act, p.slope, x, w, bias, y);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

size_t mst_convt1_wgrad_ws(const ConvP& p) {
    return (size_t)ms_ceil_div(p.B, convt1_lines_per_wg(p)) * p.Cout * 4 * sizeof(float);
}

int mst_convt1_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act, float* gw, float beta,
                          void* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < mst_convt1_wgrad_ws(p)) return MS_ERR_WORKSPACE;
    const int lpw = convt1_lines_per_wg(p), nwg = ms_ceil_div(p.B, lpw);
    float* partial = (float*)ws;
    const bool v4 = p.Lout % 4 == 0 && p.Lout / 4 <= 256 &&
                    ((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)(y_act ? y_act : gy))) & 15) == 0;
    if (v4 && p.in_act) hipLaunchKernelGGL((k_convt1_wgrad_v4<true>), dim3(nwg), dim3(256), 0, s, p.B, p.Cout, p.Lout, p.act, p.slope, lpw, x, gy, y_act, partial);
    else if (v4) hipLaunchKernelGGL((k_convt1_wgrad_v4<false>), dim3(nwg), dim3(256), 0, s, p.B, p.Cout, p.Lout, p.act, p.slope, lpw, x, gy, y_act, partial);
    else if (p.in_act) hipLaunchKernelGGL((k_convt1_wgrad<true>), dim3(nwg), dim3(256), 0, s, p.B, p.Cout, p.Lout, p.act, p.slope, lpw, x, gy, y_act, partial);
    else hipLaunchKernelGGL((k_convt1_wgrad<false>), dim3(nwg), dim3(256), 0, s, p.B, p.Cout, p.Lout, p.act, p.slope, lpw, x, gy, y_act, partial);
    MS_CHECK_LAUNCH();
    return msk_reduce_partials(partial, (size_t)p.Cout * 4, nwg, (size_t)p.Cout * 4, 0, gw, nullptr, beta, s);
}
