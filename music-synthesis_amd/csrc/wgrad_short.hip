// Weight gradient of a reflection-padded dense conv on SHORT rows: the generator's first layer, ReflectionPad1d(3) +
// Conv1d(80, 512, 7) on 32 mel frames (reference generator/full.py:28-30).  At B = 32 the whole problem is a 512 x 560 x 1024
// GEMM (0.6 GFLOP) that sits at the very END of the G-step's weight-gradient chain -- its gradient is the last one the backward
// pass produces -- so its duration is the step's: the im2col kernel ran it in 63 us (9 TFLOP/s).
//
// Plain fp32 on v_mfma_f32_32x32x2_f32 (exact operands), one workgroup per 32 x 32 output tile (32 output channels x 4 input
// channels x 8 tap slots), its four waves splitting the contraction; no LDS staging of operands:
//   * the contraction index (batch row, position) is ORDERED so that the two k-slots of an MFMA are positions s and 16 + s of a
//     32-sample block: a lane then needs 16 CONSECUTIVE samples of its gradient row and of its input row per block --
//     four 16-byte loads each, the input's at a lane-dependent 4-byte offset (tap k), which the memory pipe takes;
//   * the rows of x are padded once into the workspace (k_ws_pad_rows: reflection or zeros, a few us for 0.4 MB), so the
//     window loads need no edge logic;
//   * the activation's derivative is applied to the gradient samples in registers; the bias gradient falls out of the same
//     samples in the waves of the first input-channel tile;
//   * loads of block i + 1 are issued before the 16 MFMAs of block i.
#include "ms_common.h"
#include "conv_mfma.h"
#include <stdint.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WsP {
    int B, Cin, Cout, L, K, pad, Lq, act, reflect;
    float slope;
};

// xr[b][ci][t] = x[b][ci][source of (t - pad)] for t < L + 2 pad, 0 behind
__global__ __launch_bounds__(256) void k_ws_pad_rows(WsP p, const float* __restrict__ x, float* __restrict__ xr) {
    const size_t n = (size_t)p.B * p.Cin * p.Lq, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t r = i / p.Lq;
    const int t = (int)(i - r * p.Lq);
    float v = 0.f;
    if (t < p.L + 2 * p.pad) {
        const int u = ms_src_index(t - p.pad, p.L, p.reflect ? MS_PAD_REFLECT : MS_PAD_ZERO);
        if (u >= 0) v = x[r * p.L + u];
    }
    xr[i] = v;
}

struct WsRegs {
    f32x4 g[4], y[4];
    f32x4u x[4];
};

__device__ __forceinline__ void ws_load(const WsP& p, const float* __restrict__ gy, const float* __restrict__ ya,
                                        const float* __restrict__ xr, int it, int nlb, int co, int ci, int k, int kk, WsRegs& r) {
    const int b = it / nlb, lb = it - b * nlb;
    const size_t go = ((size_t)b * p.Cout + co) * p.L + 32 * lb + 16 * kk;
    const size_t xo = ((size_t)b * p.Cin + ci) * p.Lq + 32 * lb + 16 * kk + k;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        r.g[q] = *reinterpret_cast<const f32x4*>(gy + go + 4 * q);
        if (p.act != MS_ACT_NONE) r.y[q] = *reinterpret_cast<const f32x4*>(ya + go + 4 * q);
        r.x[q] = *reinterpret_cast<const f32x4u*>(xr + xo + 4 * q);
    }
}

// grid (Cout / 32, Cin / 4), 256 threads: the four waves of a tile take every fourth (batch row, block) step -- four times the
// loads in flight per tile (one wave per tile left the 1 us of a load exposed behind 0.4 us of MFMAs: 55 us) -- and their
// sums are added in wave order.
__global__ __launch_bounds__(256) void k_wgrad_short(WsP p, const float* __restrict__ gy, const float* __restrict__ ya,
                                                    const float* __restrict__ xr, float beta, float* __restrict__ gw,
                                                    float* __restrict__ gb) {
    __shared__ float red[3][17][64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, m = lane & 31, kk = lane >> 5;
    const int co = 32 * blockIdx.x + m;                                 // this lane's gradient row (A operand)
    const int ci = 4 * blockIdx.y + (m >> 3), k = m & 7;                // this lane's column (B operand): (input channel, tap slot)
    const int nlb = p.L / 32, nit = p.B * nlb;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float bsum = 0.f;
    WsRegs cur, nxt;
    if (wid < nit) ws_load(p, gy, ya, xr, wid, nlb, co, ci, k, kk, cur);
    for (int it = wid; it < nit; it += 4) {
        if (it + 4 < nit) ws_load(p, gy, ya, xr, it + 4, nlb, co, ci, k, kk, nxt);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float g = cur.g[q][e];
                if (p.act != MS_ACT_NONE) g = ms_act_grad(g, cur.y[q][e], p.act, p.slope);
                bsum += g;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g, cur.x[q][e], acc, 0, 0, 0);
            }
        if (it + 4 < nit) cur = nxt;
    }
    if (wid > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wid - 1][i][lane] = acc[i];
        red[wid - 1][16][lane] = bsum;
    }
    __syncthreads();
    if (wid > 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += red[w][i][lane];
        bsum += red[w][16][lane];
    }
    // D[i] of lane: column n = lane % 32 = this lane's (ci, k); row 8 (i / 4) + 4 (lane / 32) + i % 4
    if (k < p.K) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 32 * blockIdx.x + 8 * (i >> 2) + 4 * kk + (i & 3);
            float* dst = gw + ((size_t)row * p.Cin + ci) * p.K + k;
            *dst = beta != 0.f ? beta * *dst + acc[i] : acc[i];
        }
    }
    if (gb && blockIdx.y == 0) {
        bsum += __shfl_xor(bsum, 32, 64);                               // the two k-slots of the row
        if (kk == 0) gb[co] = beta != 0.f ? beta * gb[co] + bsum : bsum;
    }
}

bool ws_enabled() {
    const char* sw = getenv("MSYNTH_WSHORT");     // tuning / test switch (0: the im2col weight-gradient kernel)
    return !(sw && atoi(sw) == 0);
}

bool ws_plan(const ConvP& c, WsP* p) {
    if (!ws_enabled() || c.pad_mode != MS_PAD_REFLECT) return false;
    if (c.groups != 1 || c.stride != 1 || c.dil != 1 || c.in_act || c.Lout != c.Lin) return false;
    if (c.K < 1 || c.K > 8 || 2 * c.pad != c.K - 1 || c.pad >= c.Lin) return false;
    if (c.Lin % 32 || c.Lin > 64 || c.Cout % 32 || c.Cin % 4 || (long long)c.B * c.Lin < 256) return false;
    p->B = c.B; p->Cin = c.Cin; p->Cout = c.Cout; p->L = c.Lin; p->K = c.K; p->pad = c.pad;
    p->Lq = c.Lin + 8;                            // >= L + 7 samples are read behind a window's first; a multiple of 4
    p->act = c.act; p->slope = c.slope; p->reflect = 1;
    return true;
}

}  // namespace

bool msws_applicable(const ConvP& c) {
    WsP p;
    return ws_plan(c, &p);
}

size_t msws_ws(const ConvP& c) {
    WsP p;
    return ws_plan(c, &p) ? (size_t)p.B * p.Cin * p.Lq * sizeof(float) : 0;
}

int msws_bwd_weight(const ConvP& c, const float* x, const float* gy, const float* y_act, float* gw, float* gb, float beta,
                    void* ws, size_t ws_bytes, hipStream_t s) {
    WsP p;
    if (!ws_plan(c, &p)) return MS_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < msws_ws(c)) return MS_ERR_WORKSPACE;
    if ((((uintptr_t)gy) & 15) || (y_act && (((uintptr_t)y_act) & 15)) || (((uintptr_t)ws) & 15)) return MS_ERR_UNSUPPORTED;
    if (p.act != MS_ACT_NONE && !y_act) return MS_ERR_INVALID_ARG;
    float* xr = (float*)ws;
    const size_t n = (size_t)p.B * p.Cin * p.Lq;
    ms_note_kernel(0, "k_wgrad_short");
    hipLaunchKernelGGL(k_ws_pad_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, x, xr);
    MS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_wgrad_short, dim3(p.Cout / 32, p.Cin / 4), dim3(256), 0, s, p, gy, y_act, xr, beta, gw, gb);
    MS_CHECK_LAUNCH();
    return MS_OK;
}
