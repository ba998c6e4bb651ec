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

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GK = 41, GS = 4, GCG = 4;      // taps, stride, input channels per group
constexpr int TT = 64;                        // outputs per wave  (4 MFMA column tiles)
constexpr int WTT = 4 * TT;                   // outputs per workgroup
constexpr int PS = (WTT * GS + GK - 1 + GS - 1) / GS + 1;   // entries per phase row
constexpr int PSP = ((PS + 7) / 8) * 8 + 4;   // == 4 (mod 8): the 4 ci rows land 16 banks apart

__global__ __launch_bounds__(256) void k_gconv_mfma_fwd(ConvP p, const float* __restrict__ x,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ y) {
    __shared__ float xs[GCG * GS * PSP];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = blockIdx.y, b = blockIdx.z;
    const int T0 = blockIdx.x * WTT;                 // first output of this workgroup
    const int u0 = T0 * GS - p.pad;                  // input index of padded position 0
    const int span = WTT * GS + GK - 1;

    // stage x[b, g*4 + ci, u0 .. u0+span) de-interleaved by phase
    for (int idx = tid; idx < GCG * span; idx += 256) {
        const int ci = idx / span, u = idx - ci * span;
        const int s = u0 + u;
        const bool ok = s >= 0 && s < p.Lin;
        const float v = x[ok ? ((size_t)b * p.Cin + (size_t)g * GCG + ci) * p.Lin + s : 0];
        xs[ci * (GS * PSP) + (u & (GS - 1)) * PSP + (u >> 2)] = ok ? v : 0.f;
    }

    // weight fragments: lane (m = lane&15, ci = lane>>4) holds w[g*Og+m][ci][j] for every tap j
    const int m = lane & 15, ci = lane >> 4;
    float a[GK];
    {
        const bool ok = m < p.Og;
        const float* wr = w + ((size_t)(g * p.Og + (ok ? m : 0)) * GCG + ci) * GK;
#pragma unroll
        for (int j = 0; j < GK; ++j) {
            const float v = wr[j];
            a[j] = ok ? v : 0.f;
        }
    }
    __syncthreads();

    const float* xb = xs + ci * (GS * PSP) + wid * TT + (lane & 15);
#pragma unroll
    for (int tt = 0; tt < TT / 16; ++tt) {
        const int tbase = T0 + wid * TT + tt * 16;
        if (tbase >= p.Lout) break;                  // wave-uniform
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < GK; ++j) {
            const float bv = xb[(j & (GS - 1)) * PSP + tt * 16 + (j >> 2)];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bv, acc, 0, 0, 0);
        }
        const int t = tbase + (lane & 15);
        if (t < p.Lout) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mo = (lane >> 4) * 4 + r;
                if (mo < p.Og) {
                    const int co = g * p.Og + mo;
                    const float v = acc[r] + (bias ? bias[co] : 0.f);
                    y[((size_t)b * p.Cout + co) * p.Lout + t] = ms_apply_act(v, p.act, p.slope);
                }
            }
        }
    }
}


// ---------------------------------------------------------------- backward data
// gx[b, g*4+ci, 4q+r] = gx_add + sum_{co,jj} w[co, ci, r + 4jj] * gp[b, co, q + 5 - jj]
// (pad = 20 == 0 mod 4, so output phase r only sees taps k = r + 4jj).  GEMM per group and q-tile:
// M = (ci, r) = 16 rows, N = 16 consecutive q, K = co (4 per MFMA) x jj (11).  A lane's 4
// accumulator rows are the 4 phases r of one (ci, q): 4 consecutive samples of gx.
constexpr int JJ = (GK + GS - 1) / GS;        // 11 taps per phase
constexpr int TQ = 64;                        // q's per wave
constexpr int WQ = 4 * TQ;                    // q's per workgroup
constexpr int GRS = ((WQ + 2 * (JJ - 1) + 31) / 32) * 32 + 16;   // == 16 (mod 32)

template <int OQ>   // OQ = Og / 4 co-quads (4 for Og = 16, 1 for Og = 4)
__global__ __launch_bounds__(256) void k_gconv_mfma_bwd_data(ConvP p, const float* __restrict__ gy,
                                                            const float* __restrict__ y_act,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ gx_add,
                                                            float* __restrict__ gx) {
    __shared__ float gs[OQ * 4 * GRS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = blockIdx.y, b = blockIdx.z;
    const int Q0 = blockIdx.x * WQ;
    const int tlo = Q0 + 5 - (JJ - 1);               // first gp index staged (may be < 0)
    const int span = WQ + 2 * (JJ - 1);
    const float* ya = y_act ? y_act : gy;
    const int kind = y_act ? p.act : MS_ACT_NONE;
    for (int idx = tid; idx < OQ * 4 * span; idx += 256) {
        const int co = idx / span, tt = idx - co * span;
        const int t = tlo + tt;
        const bool ok = t >= 0 && t < p.Lout;
        const size_t off = ok ? ((size_t)b * p.Cout + (size_t)g * p.Og + co) * p.Lout + t : 0;
        const float v = gy[off], a = ya[off];
        gs[co * GRS + tt] = ok ? ms_act_grad(v, a, kind, p.slope) : 0.f;
    }
    // weight fragments: lane (m = (ci, r) = lane&15, k = lane>>4): w[g*Og + 4cq + k][ci][r + 4jj]
    const int mrow = lane & 15, ci = mrow >> 2, r = mrow & 3, kq = lane >> 4;
    float a[OQ * JJ];
#pragma unroll
    for (int cq = 0; cq < OQ; ++cq)
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            const int k = r + 4 * jj;
            const bool ok = k < GK;
            const float v = w[((size_t)(g * p.Og + cq * 4 + kq) * GCG + ci) * GK + (ok ? k : 0)];
            a[cq * JJ + jj] = ok ? v : 0.f;
        }
    __syncthreads();

#pragma unroll
    for (int tq = 0; tq < TQ / 16; ++tq) {
        const int qbase = Q0 + wid * TQ + tq * 16;
        if (qbase * GS >= p.Lin) break;              // wave-uniform
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // gp index for (q, jj): q + 5 - jj  ->  LDS column (q - Q0) + (JJ-1) - jj
        const float* gb = gs + kq * GRS + wid * TQ + tq * 16 + (lane & 15) + (JJ - 1);
#pragma unroll
        for (int cq = 0; cq < OQ; ++cq)
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) {
                const float bv = gb[cq * 4 * GRS - jj];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cq * JJ + jj], bv, acc, 0, 0, 0);
            }
        // D[row][col]: col = lane&15 = q, row = (lane>>4)*4 + reg = ci*4 + r  ->  ci = lane>>4, r = reg
        const int q = qbase + (lane & 15);
        const size_t rowoff = ((size_t)b * p.Cin + (size_t)g * GCG + (lane >> 4)) * p.Lin;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int sidx = q * GS + rr;
            if (sidx < p.Lin) {
                float v = acc[rr];
                if (gx_add) v += gx_add[rowoff + sidx];
                gx[rowoff + sidx] = v;
            }
        }
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
    dim3 grid(ms_ceil_div(p.Lout, WTT), p.groups, p.B);
    hipLaunchKernelGGL(k_gconv_mfma_fwd, grid, dim3(256), 0, s, p, x, w, bias, y);
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
    dim3 grid(ms_ceil_div(Lq, WQ), p.groups, p.B);
    if (p.Og == 16)
        hipLaunchKernelGGL(k_gconv_mfma_bwd_data<4>, grid, dim3(256), 0, s, p, gy, y_act, w, gx_add, gx);
    else
        hipLaunchKernelGGL(k_gconv_mfma_bwd_data<1>, grid, dim3(256), 0, s, p, gy, y_act, w, gx_add, gx);
    MS_CHECK_LAUNCH();
    return MS_OK;
}
