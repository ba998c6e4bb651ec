// HBM-bound elementwise / reduction kernels: avg-pool, activation backward, the GAN losses
// (wavefront-shuffle reductions), fused Adam.  All streaming, coalesced along contiguous
// audio frames, grid-strided over <= 2048 workgroups.
#include "ms_common.h"

namespace {

constexpr int kMaxBlocks = 2048;

inline int grid_for(int64_t n, int per_thread = 4) {
    int64_t b = (n + 256LL * per_thread - 1) / (256LL * per_thread);
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

// ---- avg_pool1d(k=4, s=2, p=2), zeros counted in the divisor
__global__ __launch_bounds__(256) void k_pool_fwd(const float* __restrict__ x,
                                                 float* __restrict__ y, int64_t rows, int Lin,
                                                 int Lout) {
    const int64_t total = rows * Lout;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / Lout;
        const int o = (int)(i - r * Lout);
        const float* xr = x + r * Lin;
        const int s = o * 2 - 2;
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = s + j;
            if (q >= 0 && q < Lin) a += xr[q];
        }
        y[i] = a * 0.25f;
    }
}

// gx[i] = gx_add[i] + 0.25 * sum of gy[o] over windows o that cover i (o*2-2 <= i <= o*2+1)
__global__ __launch_bounds__(256) void k_pool_bwd(const float* __restrict__ gy,
                                                 const float* __restrict__ gx_add,
                                                 float* __restrict__ gx, int64_t rows, int Lin,
                                                 int Lout) {
    const int64_t total = rows * Lin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / Lin;
        const int p = (int)(i - r * Lin);
        const float* gr = gy + r * Lout;
        // o in [ceil((p-1)/2), floor((p+2)/2)]
        const int o_lo = p >> 1;  // ceil((p-1)/2) == p/2 for p >= 0 (p=0 -> 0)
        const int o_hi = (p + 2) >> 1;
        float a = 0.f;
        for (int o = o_lo; o <= o_hi; ++o)
            if (o < Lout) a += gr[o];
        a *= 0.25f;
        if (gx_add) a += gx_add[i];
        gx[i] = a;
    }
}


// ---- nn.AvgPool1d(4, 2, padding=1, count_include_pad=False): divisor = in-range samples
__global__ __launch_bounds__(256) void k_pool421_fwd(const float* __restrict__ x,
                                                    float* __restrict__ y, int64_t rows, int Lin,
                                                    int Lout) {
    const int64_t total = rows * Lout;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / Lout;
        const int o = (int)(i - r * Lout);
        const float* xr = x + r * Lin;
        const int s = o * 2 - 1;
        float a = 0.f;
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = s + j;
            const bool ok = q >= 0 && q < Lin;
            a += ok ? xr[q] : 0.f;
            cnt += ok ? 1 : 0;
        }
        y[i] = a / (float)cnt;
    }
}

// ---- F.avg_pool1d(x, k): window = stride = k, no padding, Lout = Lin / k (the conditioning branch of the
// weight-normed MelGAN's discriminators pools the mel features down to the feature map's rate,
// experiment/realmelgan.py:150-151)
__global__ __launch_bounds__(256) void k_poolk_fwd(const float* __restrict__ x, float* __restrict__ y,
                                                  int64_t rows, int Lin, int Lout, int k) {
    const int64_t total = rows * Lout;
    const float inv = 1.f / (float)k;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / Lout;
        const int o = (int)(i - r * Lout);
        const float* xr = x + r * Lin + (int64_t)o * k;
        float a = 0.f;
        for (int j = 0; j < k; ++j) a += xr[j];
        y[i] = a * inv;
    }
}

__global__ __launch_bounds__(256) void k_poolk_bwd(const float* __restrict__ gy, float* __restrict__ gx,
                                                  int64_t rows, int Lin, int Lout, int k) {
    const int64_t total = rows * Lin;
    const float inv = 1.f / (float)k;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / Lin;
        const int p = (int)(i - r * Lin);
        const int o = p / k;
        gx[i] = o < Lout ? gy[r * Lout + o] * inv : 0.f;     // (samples behind the last full window get no gradient)
    }
}

__global__ __launch_bounds__(256) void k_pool421_bwd(const float* __restrict__ gy,
                                                    const float* __restrict__ gx_add,
                                                    float* __restrict__ gx, int64_t rows, int Lin,
                                                    int Lout) {
    const int64_t total = rows * Lin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / Lin;
        const int p = (int)(i - r * Lin);
        const float* gr = gy + r * Lout;
        // windows o with 2o-1 <= p <= 2o+2  ->  o in [ceil((p-2)/2), floor((p+1)/2)]
        const int o_lo = p >= 2 ? (p - 1) >> 1 : 0;
        const int o_hi = (p + 1) >> 1;
        float a = 0.f;
        for (int o = o_lo; o <= o_hi; ++o) {
            if (o >= Lout) continue;
            const int s = o * 2 - 1;
            const int lo = s < 0 ? 0 : s, hi = s + 3 >= Lin ? Lin - 1 : s + 3;
            a += gr[o] / (float)(hi - lo + 1);
        }
        if (gx_add) a += gx_add[i];
        gx[i] = a;
    }
}

// ---- weight normalisation: one workgroup per row of the (rows, cols) parameter view
__global__ __launch_bounds__(256) void k_weight_norm_fwd(const float* __restrict__ v,
                                                        const float* __restrict__ g,
                                                        float* __restrict__ w, int cols) {
    __shared__ float red[4];
    __shared__ float scale;
    const int r = blockIdx.x;
    const float* vr = v + (size_t)r * cols;
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) s += vr[c] * vr[c];
    const float tot = ms_block_sum(s, red);
    if (threadIdx.x == 0) scale = g[r] / sqrtf(tot);
    __syncthreads();
    const float sc = scale;
    for (int c = threadIdx.x; c < cols; c += 256) w[(size_t)r * cols + c] = vr[c] * sc;
}

__global__ __launch_bounds__(256) void k_weight_norm_bwd(const float* __restrict__ v,
                                                        const float* __restrict__ g,
                                                        const float* __restrict__ gw,
                                                        float* __restrict__ gv,
                                                        float* __restrict__ gg, int cols,
                                                        float beta) {
    __shared__ float red[4];
    __shared__ float sh[2];
    const int r = blockIdx.x;
    const float* vr = v + (size_t)r * cols;
    const float* gr = gw + (size_t)r * cols;
    float s2 = 0.f, dot = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        s2 += vr[c] * vr[c];
        dot += gr[c] * vr[c];
    }
    const float t2 = ms_block_sum(s2, red);
    const float td = ms_block_sum(dot, red);
    if (threadIdx.x == 0) { sh[0] = t2; sh[1] = td; }
    __syncthreads();
    const float nrm = sqrtf(sh[0]);
    const float gdot = sh[1] / nrm;              // gw . v^
    const float gs = g[r] / nrm;
    if (threadIdx.x == 0) gg[r] = (beta != 0.f ? beta * gg[r] : 0.f) + gdot;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float val = gs * (gr[c] - gdot * vr[c] / nrm);
        const size_t o = (size_t)r * cols + c;
        gv[o] = (beta != 0.f ? beta * gv[o] : 0.f) + val;
    }
}

// multi-tensor forms: workgroup b owns row b of the concatenated row list
__device__ __forceinline__ int wn_find(const ms_wn_multi_desc& d, int b, int* row) {
    int t = 0;
    while (t + 1 < d.count && b >= d.rows[t]) { b -= d.rows[t]; ++t; }
    *row = b;
    return t;
}

__global__ __launch_bounds__(256) void k_weight_norm_multi_fwd(ms_wn_multi_desc d) {
    __shared__ float red[4];
    __shared__ float scale;
    int r;
    const int t = wn_find(d, blockIdx.x, &r);
    const int cols = d.cols[t];
    const float* vr = d.v[t] + (size_t)r * cols;
    float* wr = d.out[t] + (size_t)r * cols;
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) s += vr[c] * vr[c];
    const float tot = ms_block_sum(s, red);
    if (threadIdx.x == 0) scale = d.g[t][r] / sqrtf(tot);
    __syncthreads();
    const float sc = scale;
    for (int c = threadIdx.x; c < cols; c += 256) wr[c] = vr[c] * sc;
}

__global__ __launch_bounds__(256) void k_weight_norm_multi_bwd(ms_wn_multi_desc d, float beta) {
    __shared__ float red[4];
    __shared__ float sh[2];
    int r;
    const int t = wn_find(d, blockIdx.x, &r);
    const int cols = d.cols[t];
    const float* vr = d.v[t] + (size_t)r * cols;
    const float* gr = d.out[t] + (size_t)r * cols;
    float* gvr = d.gv[t] + (size_t)r * cols;
    float s2 = 0.f, dot = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        s2 += vr[c] * vr[c];
        dot += gr[c] * vr[c];
    }
    const float t2 = ms_block_sum(s2, red);
    const float td = ms_block_sum(dot, red);
    if (threadIdx.x == 0) { sh[0] = t2; sh[1] = td; }
    __syncthreads();
    const float nrm = sqrtf(sh[0]);
    const float gdot = sh[1] / nrm;
    const float gs = d.g[t][r] / nrm;
    if (threadIdx.x == 0) d.gg[t][r] = (beta != 0.f ? beta * d.gg[t][r] : 0.f) + gdot;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float val = gs * (gr[c] - gdot * vr[c] / nrm);
        gvr[c] = (beta != 0.f ? beta * gvr[c] : 0.f) + val;
    }
}

__global__ __launch_bounds__(256) void k_act_bwd(const float* __restrict__ ya,
                                                const float* __restrict__ gy,
                                                float* __restrict__ out, int64_t n, int act,
                                                float slope) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = ms_act_grad(gy[i], ya[i], act, slope);
}

__global__ __launch_bounds__(256) void k_add(const float* __restrict__ a,
                                            const float* __restrict__ b, float* __restrict__ out,
                                            int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = a[i] + b[i];
}

// out = act(a + b): the residual DilatedStack layer (activation OVER the skip sum, util/modules.py:131-134)
__global__ __launch_bounds__(256) void k_add_act(const float* __restrict__ a, const float* __restrict__ b,
                                                float* __restrict__ out, int64_t n4, int64_t n, int act,
                                                float slope) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
        float4 o;
        o.x = ms_apply_act(x.x + y.x, act, slope); o.y = ms_apply_act(x.y + y.y, act, slope);
        o.z = ms_apply_act(x.z + y.z, act, slope); o.w = ms_apply_act(x.w + y.w, act, slope);
        reinterpret_cast<float4*>(out)[i] = o;
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        out[i] = ms_apply_act(a[i] + b[i], act, slope);
}

// ---- reductions: per-element term selected by MODE, two deterministic stages
enum { R_HINGE_D = 0, R_NEG = 1, R_L1 = 2, R_LS_G = 3, R_LS_D = 4 };

template <int MODE>
__device__ __forceinline__ float term(const float* __restrict__ a, const float* __restrict__ b,
                                      int64_t i) {
    if (MODE == R_HINGE_D) return fmaxf(1.f - a[i], 0.f) + fmaxf(1.f + b[i], 0.f);
    if (MODE == R_NEG) return -a[i];
    if (MODE == R_L1) return fabsf(a[i] - b[i]);
    if (MODE == R_LS_G) { const float d = a[i] - 1.f; return 0.5f * d * d; }
    { const float d = a[i] - 1.f; return 0.5f * (d * d + b[i] * b[i]); }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_reduce_stage1(const float* __restrict__ a,
                                                      const float* __restrict__ b, int64_t n,
                                                      float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        s += term<MODE>(a, b, i);
    const float tot = ms_block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_reduce_stage2(const float* __restrict__ partial, int np,
                                                      float inv_n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
    const float tot = ms_block_sum(s, red);
    if (threadIdx.x == 0) out[0] = tot * inv_n;
}

template <int MODE>
int reduce_mean(const float* a, const float* b, int64_t n, float* out, void* ws, size_t ws_bytes,
                hipStream_t s) {
    if (!a || !out || n <= 0) return MS_ERR_INVALID_ARG;
    const int nb = grid_for(n, 8);
    if (!ws || ws_bytes < (size_t)nb * sizeof(float)) return MS_ERR_WORKSPACE;
    float* partial = (float*)ws;
    hipLaunchKernelGGL((k_reduce_stage1<MODE>), dim3(nb), dim3(256), 0, s, a, b, n, partial);
    MS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_reduce_stage2, dim3(1), dim3(256), 0, s, partial, nb, 1.0f / (float)n, out);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

__global__ __launch_bounds__(256) void k_hinge_d_bwd(const float* __restrict__ r,
                                                    const float* __restrict__ f, int64_t n,
                                                    const float* __restrict__ gout, float scale,
                                                    float* __restrict__ gr,
                                                    float* __restrict__ gf) {
    const float g = gout[0] * scale / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (gr) gr[i] = (1.f - r[i] > 0.f) ? -g : 0.f;
        if (gf) gf[i] = (1.f + f[i] > 0.f) ? g : 0.f;
    }
}

// all judgement tensors of a loss in one launch (ms_judge_multi_desc by value)
__global__ __launch_bounds__(256) void k_judge_multi_fwd(ms_judge_multi_desc d, float* __restrict__ out) {
    __shared__ float red[4];
    float total = 0.f;
    for (int i = 0; i < d.count; ++i) {
        const float* r = d.r[i];
        const float* f = d.f[i];
        const int64_t n = d.n[i];
        float s = 0.f;
        for (int64_t e = threadIdx.x; e < n; e += 256)
            s += d.kind == MS_JUDGE_HINGE_D ? fmaxf(1.f - r[e], 0.f) + fmaxf(1.f + f[e], 0.f) : -f[e];
        const float t = ms_block_sum(s, red);
        if (threadIdx.x == 0) total += t / (float)n;
    }
    if (threadIdx.x == 0) out[0] = total;
}

__global__ __launch_bounds__(256) void k_judge_multi_bwd(ms_judge_multi_desc d, const float* __restrict__ gout,
                                                        float scale) {
    const int i = blockIdx.y;
    const int64_t n = d.n[i];
    const float g = gout[0] * scale / (float)n;
    const float* r = d.r[i];
    const float* f = d.f[i];
    float* gr = d.gr[i];
    float* gf = d.gf[i];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        if (d.kind == MS_JUDGE_HINGE_D) {
            if (gr) gr[e] = (1.f - r[e] > 0.f) ? -g : 0.f;
            if (gf) gf[e] = (1.f + f[e] > 0.f) ? g : 0.f;
        } else if (gf) {
            gf[e] = -g;
        }
    }
}

__global__ __launch_bounds__(256) void k_fill_scaled(int64_t n, const float* __restrict__ gout,
                                                    float scale, float* __restrict__ out) {
    const float g = gout[0] * scale / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = g;
}

__global__ __launch_bounds__(256) void k_l1_bwd(const float* __restrict__ r,
                                               const float* __restrict__ f, int64_t n,
                                               const float* __restrict__ gout, float scale,
                                               float* __restrict__ gf, int accumulate) {
    const float g = gout[0] * scale / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = f[i] - r[i];
        float v = d > 0.f ? g : (d < 0.f ? -g : 0.f);
        if (accumulate) v += gf[i];
        gf[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_ls_bwd(const float* __restrict__ r,
                                               const float* __restrict__ f, int64_t n,
                                               const float* __restrict__ gout, float scale,
                                               float* __restrict__ gr, float* __restrict__ gf) {
    const float g = gout[0] * scale / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (gr) gr[i] = g * (r[i] - 1.f);
        if (gf) gf[i] = g * f[i];
    }
}

__global__ void k_weighted_sum(const float* __restrict__ terms, const float* __restrict__ coef,
                               int n, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += coef[i] * terms[i];
        out[0] = s;
    }
}

// ---- Adam
__global__ void k_adam_tick(int32_t* step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1;
}

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                             float* __restrict__ m, float* __restrict__ v,
                                             int64_t n, float lr, float b1, float b2, float eps,
                                             float gscale, const int32_t* __restrict__ step) {
    // bias corrections in double, once per thread (torch computes them on the host in double)
    const double t = (double)step[0];
    const double bc1 = 1.0 - pow((double)b1, t);
    const double bc2 = 1.0 - pow((double)b2, t);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
    const int64_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gi = ga[j] * gscale;
            ma[j] = b1 * ma[j] + omb1 * gi;
            va[j] = b2 * va[j] + omb2 * gi * gi;
            const float denom = sqrtf(va[j]) * inv_bc2_sqrt + eps;
            pa[j] = pa[j] - step_size * (ma[j] / denom);
        }
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    // tail (n not a multiple of 4)
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + omb1 * gi;
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + eps));
    }
}

}  // namespace


// ---------------------------------------------------------------- multi-tensor L1 (feature matching)
// One launch covers all the feature maps: block -> map through the prefix table in the by-value
// descriptor.  2048 elements per block in the forward (partial sum per block, folded per map in a
// fixed order by one block), 2048 per block in the backward.
namespace {
constexpr int L1M_FWD_EPB = 16384;       // elements per forward block (the backward keeps 2048)
struct L1MultiK {
    ms_l1_multi_desc d;
    int blk0[MS_L1_MULTI_MAX + 1];       // first block of map i
};

__device__ __forceinline__ int l1m_map(const L1MultiK& k, int blk) {
    int m = 0;
#pragma unroll 1
    for (int i = 1; i < k.d.count; ++i) m = blk >= k.blk0[i] ? i : m;
    return m;
}

__global__ __launch_bounds__(256) void k_l1_multi_fwd(L1MultiK k, float* __restrict__ partials) {
    __shared__ float red[4];
    const int m = l1m_map(k, blockIdx.x);
    const float* r = k.d.r[m];
    const float* f = k.d.f[m];
    const int64_t n = k.d.n[m];
    // L1M_FWD_EPB elements per block in 2048-element rounds: 8x fewer partials for the single-block fold
    const int64_t base0 = (int64_t)(blockIdx.x - k.blk0[m]) * L1M_FWD_EPB;
    float s = 0.f;
#pragma unroll 1
    for (int rnd = 0; rnd < L1M_FWD_EPB / 2048; ++rnd) {
        const int64_t base = base0 + (int64_t)rnd * 2048;
        if (base >= n) break;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t e = base + threadIdx.x + 256 * i;
            const int64_t ec = e < n ? e : 0;
            const float d = f[ec] - r[ec];
            v[i] = e < n ? fabsf(d) : 0.f;
        }
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    const float tot = ms_block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_l1_multi_final(L1MultiK k, const float* __restrict__ partials,
                                                       float* __restrict__ out) {
    __shared__ float red[4];
    __shared__ float per_map[MS_L1_MULTI_MAX];
    // map i is summed by wave-sized strides of thread group i (fixed order: deterministic)
    for (int m = threadIdx.x >> 3; m < k.d.count; m += 32) {
        const int sub = threadIdx.x & 7;
        float s = 0.f;
        for (int b = k.blk0[m] + sub; b < k.blk0[m + 1]; b += 8) s += partials[b];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (sub == 0) per_map[m] = s * k.d.w[m] / (float)k.d.n[m];
    }
    __syncthreads();
    float t = threadIdx.x < k.d.count ? per_map[threadIdx.x] : 0.f;
    const float tot = ms_block_sum(t, red);
    if (threadIdx.x == 0) out[0] = tot;
}

__global__ __launch_bounds__(256) void k_l1_multi_bwd(L1MultiK k, const float* __restrict__ gout,
                                                     float scale) {
    const int m = l1m_map(k, blockIdx.x);
    float* gf = k.d.gf[m];
    if (!gf) return;
    const float* r = k.d.r[m];
    const float* f = k.d.f[m];
    const int64_t n = k.d.n[m];
    const float g = gout[0] * scale * k.d.w[m] / (float)n;
    const int64_t base = (int64_t)(blockIdx.x - k.blk0[m]) * 2048;
    float d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t e = base + threadIdx.x + 256 * i;
        const int64_t ec = e < n ? e : 0;
        d[i] = f[ec] - r[ec];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t e = base + threadIdx.x + 256 * i;
        if (e < n) gf[e] = d[i] > 0.f ? g : (d[i] < 0.f ? -g : 0.f);
    }
}

// Forward and backward in ONE pass over the maps (r04): a train step's loss is the root of the backward pass, so the upstream
// gradient of every L1 term is a constant the host knows (gconst) and sign(f - r) can be written while |f - r| is summed:
// r and f are read once instead of twice (266 MB less per generator step at B = 32).  16-byte accesses (every map is a
// multiple of 4 elements and 16-byte aligned: checked by the caller), 8192 elements per block.
constexpr int L1M_FB_EPB = 8192;
__global__ __launch_bounds__(256) void k_l1_multi_fwd_bwd(L1MultiK k, float* __restrict__ partials, float gconst) {
    __shared__ float red[4];
    const int m = l1m_map(k, blockIdx.x);
    const float4* r = reinterpret_cast<const float4*>(k.d.r[m]);
    const float4* f = reinterpret_cast<const float4*>(k.d.f[m]);
    float4* gf = reinterpret_cast<float4*>(k.d.gf[m]);
    const int64_t n4 = k.d.n[m] / 4;
    const float g = gconst * k.d.w[m] / (float)k.d.n[m];
    const int64_t base = (int64_t)(blockIdx.x - k.blk0[m]) * (L1M_FB_EPB / 4);
    float4 rv[8], fv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t e = base + threadIdx.x + 256 * i;
        const int64_t ec = e < n4 ? e : 0;
        rv[i] = r[ec]; fv[i] = f[ec];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t e = base + threadIdx.x + 256 * i;
        const float d0 = fv[i].x - rv[i].x, d1 = fv[i].y - rv[i].y, d2 = fv[i].z - rv[i].z, d3 = fv[i].w - rv[i].w;
        if (e < n4) {
            s += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
            if (gf) gf[e] = make_float4(d0 > 0.f ? g : (d0 < 0.f ? -g : 0.f), d1 > 0.f ? g : (d1 < 0.f ? -g : 0.f),
                                        d2 > 0.f ? g : (d2 < 0.f ? -g : 0.f), d3 > 0.f ? g : (d3 < 0.f ? -g : 0.f));
        }
    }
    const float tot = ms_block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

bool l1m_plan(const ms_l1_multi_desc* d, L1MultiK* k, int epb = 2048) {
    if (!d || d->count <= 0 || d->count > MS_L1_MULTI_MAX) return false;
    k->d = *d;
    long long blk = 0;
    for (int i = 0; i < d->count; ++i) {
        if (!d->r[i] || !d->f[i] || d->n[i] <= 0) return false;
        k->blk0[i] = (int)blk;
        blk += (d->n[i] + epb - 1) / epb;
        if (blk > (1 << 30)) return false;
    }
    for (int i = d->count; i <= MS_L1_MULTI_MAX; ++i) k->blk0[i] = (int)blk;
    return true;
}
}  // namespace

extern "C" {

int ms_avg_pool1d_4_2_2_fwd(const float* x, float* y, int64_t rows, int32_t Lin, ms_stream_t stream) {
    if (!x || !y || rows <= 0 || Lin <= 0) return MS_ERR_INVALID_ARG;
    const int Lout = (Lin + 4 - 4) / 2 + 1;
    hipLaunchKernelGGL(k_pool_fwd, dim3(grid_for(rows * Lout)), dim3(256), 0, (hipStream_t)stream,
                       x, y, rows, Lin, Lout);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_avg_pool1d_4_2_2_bwd(const float* gy, const float* gx_add, float* gx, int64_t rows,
                            int32_t Lin, ms_stream_t stream) {
    if (!gy || !gx || rows <= 0 || Lin <= 0) return MS_ERR_INVALID_ARG;
    const int Lout = (Lin + 4 - 4) / 2 + 1;
    hipLaunchKernelGGL(k_pool_bwd, dim3(grid_for(rows * Lin)), dim3(256), 0, (hipStream_t)stream,
                       gy, gx_add, gx, rows, Lin, Lout);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_avg_pool1d_4_2_1_fwd(const float* x, float* y, int64_t rows, int32_t Lin, ms_stream_t stream) {
    if (!x || !y || rows <= 0 || Lin < 2) return MS_ERR_INVALID_ARG;
    const int Lout = (Lin + 2 - 4) / 2 + 1;
    hipLaunchKernelGGL(k_pool421_fwd, dim3(grid_for(rows * Lout)), dim3(256), 0, (hipStream_t)stream,
                       x, y, rows, Lin, Lout);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_avg_pool1d_4_2_1_bwd(const float* gy, const float* gx_add, float* gx, int64_t rows,
                            int32_t Lin, ms_stream_t stream) {
    if (!gy || !gx || rows <= 0 || Lin < 2) return MS_ERR_INVALID_ARG;
    const int Lout = (Lin + 2 - 4) / 2 + 1;
    hipLaunchKernelGGL(k_pool421_bwd, dim3(grid_for(rows * Lin)), dim3(256), 0, (hipStream_t)stream,
                       gy, gx_add, gx, rows, Lin, Lout);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_avg_pool1d_k_fwd(const float* x, float* y, int64_t rows, int32_t Lin, int32_t k, ms_stream_t stream) {
    if (!x || !y || rows <= 0 || k < 1 || Lin < k) return MS_ERR_INVALID_ARG;
    const int Lout = Lin / k;
    hipLaunchKernelGGL(k_poolk_fwd, dim3(grid_for(rows * Lout)), dim3(256), 0, (hipStream_t)stream, x, y, rows, Lin,
                       Lout, k);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_avg_pool1d_k_bwd(const float* gy, float* gx, int64_t rows, int32_t Lin, int32_t k, ms_stream_t stream) {
    if (!gy || !gx || rows <= 0 || k < 1 || Lin < k) return MS_ERR_INVALID_ARG;
    const int Lout = Lin / k;
    hipLaunchKernelGGL(k_poolk_bwd, dim3(grid_for(rows * Lin)), dim3(256), 0, (hipStream_t)stream, gy, gx, rows, Lin,
                       Lout, k);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_weight_norm_fwd(const float* v, const float* g, float* w, int32_t rows, int32_t cols,
                       ms_stream_t stream) {
    if (!v || !g || !w || rows <= 0 || cols <= 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_weight_norm_fwd, dim3(rows), dim3(256), 0, (hipStream_t)stream, v, g, w, cols);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_weight_norm_bwd(const float* v, const float* g, const float* gw, float* gv, float* gg,
                       int32_t rows, int32_t cols, float beta, ms_stream_t stream) {
    if (!v || !g || !gw || !gv || !gg || rows <= 0 || cols <= 0) return MS_ERR_INVALID_ARG;
    if (beta != 0.f && beta != 1.f) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_weight_norm_bwd, dim3(rows), dim3(256), 0, (hipStream_t)stream, v, g, gw, gv,
                       gg, cols, beta);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

static int wn_multi_rows(const ms_wn_multi_desc* d, bool bwd) {
    if (!d || d->count <= 0 || d->count > MS_WN_MULTI_MAX) return -1;
    long long rows = 0;
    for (int i = 0; i < d->count; ++i) {
        if (!d->v[i] || !d->g[i] || !d->out[i] || d->rows[i] <= 0 || d->cols[i] <= 0) return -1;
        if (bwd && (!d->gv[i] || !d->gg[i])) return -1;
        rows += d->rows[i];
    }
    return rows < (1LL << 31) ? (int)rows : -1;
}

int ms_weight_norm_multi_fwd(const ms_wn_multi_desc* d, ms_stream_t stream) {
    const int rows = wn_multi_rows(d, false);
    if (rows <= 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_weight_norm_multi_fwd, dim3(rows), dim3(256), 0, (hipStream_t)stream, *d);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_weight_norm_multi_bwd(const ms_wn_multi_desc* d, float beta, ms_stream_t stream) {
    const int rows = wn_multi_rows(d, true);
    if (rows <= 0 || (beta != 0.f && beta != 1.f)) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_weight_norm_multi_bwd, dim3(rows), dim3(256), 0, (hipStream_t)stream, *d, beta);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_act_bwd(const float* y_act, const float* gy, float* gpre, int64_t n, int32_t act,
               float slope, ms_stream_t stream) {
    if (!y_act || !gy || !gpre || n <= 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_act_bwd, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y_act, gy,
                       gpre, n, act, slope);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_add(const float* a, const float* b, float* out, int64_t n, ms_stream_t stream) {
    if (!a || !b || !out || n <= 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_add, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_add_act(const float* a, const float* b, float* out, int64_t n, int32_t act, float slope,
               ms_stream_t stream) {
    if (!a || !b || !out || n <= 0) return MS_ERR_INVALID_ARG;
    const bool al = ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15) == 0;
    const int64_t n4 = al ? n / 4 : 0;
    hipLaunchKernelGGL(k_add_act, dim3(grid_for(n4 > 0 ? n4 : n)), dim3(256), 0, (hipStream_t)stream, a, b, out, n4,
                       n, act, slope);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

size_t ms_reduce_workspace_bytes(int64_t n) { return (size_t)grid_for(n, 8) * sizeof(float); }

int ms_hinge_d_fwd(const float* r, const float* f, int64_t n, float* out, void* ws, size_t wsb,
                   ms_stream_t stream) {
    if (!f) return MS_ERR_INVALID_ARG;
    return reduce_mean<R_HINGE_D>(r, f, n, out, ws, wsb, (hipStream_t)stream);
}

int ms_hinge_d_bwd(const float* r, const float* f, int64_t n, const float* gout, float scale,
                   float* gr, float* gf, ms_stream_t stream) {
    if (!r || !f || !gout || n <= 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_hinge_d_bwd, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, r, f, n,
                       gout, scale, gr, gf);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

static bool judge_multi_ok(const ms_judge_multi_desc* d, int64_t* nmax) {
    if (!d || d->count <= 0 || d->count > MS_JUDGE_MULTI_MAX) return false;
    if (d->kind != MS_JUDGE_HINGE_D && d->kind != MS_JUDGE_NEG_MEAN) return false;
    *nmax = 0;
    for (int i = 0; i < d->count; ++i) {
        if (!d->f[i] || (d->kind == MS_JUDGE_HINGE_D && !d->r[i])) return false;
        if (d->n[i] <= 0 || d->n[i] > MS_JUDGE_MULTI_NMAX) return false;
        if (d->n[i] > *nmax) *nmax = d->n[i];
    }
    return true;
}

int ms_judge_loss_multi_fwd(const ms_judge_multi_desc* d, float* out, ms_stream_t stream) {
    int64_t nmax;
    if (!out || !judge_multi_ok(d, &nmax)) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_judge_multi_fwd, dim3(1), dim3(256), 0, (hipStream_t)stream, *d, out);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_judge_loss_multi_bwd(const ms_judge_multi_desc* d, const float* gout, float scale, ms_stream_t stream) {
    int64_t nmax;
    if (!gout || !judge_multi_ok(d, &nmax)) return MS_ERR_INVALID_ARG;
    unsigned gx = (unsigned)((nmax + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_judge_multi_bwd, dim3(gx, (unsigned)d->count), dim3(256), 0, (hipStream_t)stream, *d, gout,
                       scale);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_neg_mean_fwd(const float* f, int64_t n, float* out, void* ws, size_t wsb, ms_stream_t stream) {
    return reduce_mean<R_NEG>(f, nullptr, n, out, ws, wsb, (hipStream_t)stream);
}

int ms_neg_mean_bwd(int64_t n, const float* gout, float scale, float* gf, ms_stream_t stream) {
    if (!gout || !gf || n <= 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_fill_scaled, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, gout,
                       -scale, gf);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_l1_mean_fwd(const float* r, const float* f, int64_t n, float* out, void* ws, size_t wsb,
                   ms_stream_t stream) {
    if (!f) return MS_ERR_INVALID_ARG;
    return reduce_mean<R_L1>(r, f, n, out, ws, wsb, (hipStream_t)stream);
}

int ms_l1_mean_bwd(const float* r, const float* f, int64_t n, const float* gout, float scale,
                   float* gf, int32_t accumulate, ms_stream_t stream) {
    if (!r || !f || !gout || !gf || n <= 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_l1_bwd, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, r, f, n, gout,
                       scale, gf, accumulate);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

size_t ms_l1_mean_multi_workspace_bytes(const ms_l1_multi_desc* d) {
    L1MultiK k;
    if (!l1m_plan(d, &k, L1M_FWD_EPB)) return 0;
    return (size_t)k.blk0[MS_L1_MULTI_MAX] * sizeof(float);
}

int ms_l1_mean_multi_fwd(const ms_l1_multi_desc* d, float* out, void* ws, size_t wsb,
                         ms_stream_t stream) {
    L1MultiK k;
    if (!out || !l1m_plan(d, &k, L1M_FWD_EPB)) return MS_ERR_INVALID_ARG;
    const int nblk = k.blk0[MS_L1_MULTI_MAX];
    if (!ws || wsb < (size_t)nblk * sizeof(float)) return MS_ERR_WORKSPACE;
    hipLaunchKernelGGL(k_l1_multi_fwd, dim3(nblk), dim3(256), 0, (hipStream_t)stream, k, (float*)ws);
    MS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_l1_multi_final, dim3(1), dim3(256), 0, (hipStream_t)stream, k, (const float*)ws, out);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

static bool l1m_vec_ok(const ms_l1_multi_desc* d) {
    for (int i = 0; i < d->count; ++i)
        if ((d->n[i] & 3) || (((uintptr_t)d->r[i] | (uintptr_t)d->f[i] | (uintptr_t)d->gf[i]) & 15)) return false;
    return true;
}

size_t ms_l1_mean_multi_fwd_bwd_workspace_bytes(const ms_l1_multi_desc* d) {
    L1MultiK k;
    if (!l1m_plan(d, &k, L1M_FB_EPB) || !l1m_vec_ok(d)) return 0;
    return (size_t)k.blk0[MS_L1_MULTI_MAX] * sizeof(float);
}

int ms_l1_mean_multi_fwd_bwd(const ms_l1_multi_desc* d, float* out, float gconst, void* ws, size_t wsb, ms_stream_t stream) {
    L1MultiK k;
    if (!out || !l1m_plan(d, &k, L1M_FB_EPB)) return MS_ERR_INVALID_ARG;
    if (!l1m_vec_ok(d)) return MS_ERR_UNSUPPORTED;
    const int nblk = k.blk0[MS_L1_MULTI_MAX];
    if (!ws || wsb < (size_t)nblk * sizeof(float)) return MS_ERR_WORKSPACE;
    hipLaunchKernelGGL(k_l1_multi_fwd_bwd, dim3(nblk), dim3(256), 0, (hipStream_t)stream, k, (float*)ws, gconst);
    MS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_l1_multi_final, dim3(1), dim3(256), 0, (hipStream_t)stream, k, (const float*)ws, out);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_l1_mean_multi_bwd(const ms_l1_multi_desc* d, const float* gout, float scale,
                         ms_stream_t stream) {
    L1MultiK k;
    if (!gout || !l1m_plan(d, &k)) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_l1_multi_bwd, dim3(k.blk0[MS_L1_MULTI_MAX]), dim3(256), 0, (hipStream_t)stream, k, gout,
                       scale);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_ls_g_fwd(const float* j, int64_t n, float* out, void* ws, size_t wsb, ms_stream_t stream) {
    return reduce_mean<R_LS_G>(j, nullptr, n, out, ws, wsb, (hipStream_t)stream);
}

int ms_ls_g_bwd(const float* j, int64_t n, const float* gout, float scale, float* gj,
                ms_stream_t stream) {
    if (!j || !gout || !gj || n <= 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_ls_bwd, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, j, j, n, gout,
                       scale, gj, (float*)nullptr);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_ls_d_fwd(const float* r, const float* f, int64_t n, float* out, void* ws, size_t wsb,
                ms_stream_t stream) {
    if (!f) return MS_ERR_INVALID_ARG;
    return reduce_mean<R_LS_D>(r, f, n, out, ws, wsb, (hipStream_t)stream);
}

int ms_ls_d_bwd(const float* r, const float* f, int64_t n, const float* gout, float scale,
                float* gr, float* gf, ms_stream_t stream) {
    if (!r || !f || !gout || n <= 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_ls_bwd, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, r, f, n, gout,
                       scale, gr, gf);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_weighted_sum(const float* terms, const float* coef, int32_t n, float* out, ms_stream_t stream) {
    if (!terms || !coef || !out || n <= 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_weighted_sum, dim3(1), dim3(64), 0, (hipStream_t)stream, terms, coef, n, out);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, float grad_scale, int32_t* step, ms_stream_t stream) {
    if (!p || !g || !m || !v || !step || n <= 0) return MS_ERR_INVALID_ARG;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(64), 0, (hipStream_t)stream, step);
    MS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_adam, dim3(grid_for(n, 16)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       lr, beta1, beta2, eps, grad_scale, step);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

}  // extern "C"
