// Generic LDS-tiled direct 1-D convolution kernels (fp32 vector FMA) for gfx950.
//
// These cover EVERY conv geometry on the hot path (any groups / stride / dilation / padding,
// reflect padding, odd lengths): they are the always-correct baseline and the production path
// for the layers that are too thin for the matrix cores (1 or 4 channels per group, K = 41
// grouped stride-4 convs, the 1-channel first/last convs).  The dense C >= 32 stride-1 convs
// are routed to the f32-MFMA implicit-GEMM kernels in conv_mfma.hip when those apply.
//
// Tiling: a 256-thread workgroup produces CT channels x TT time steps; each thread keeps an
// RC x RT register tile.  The input rows (with their dilation / stride halo) and the weight
// slice of the current channel chunk are staged in LDS; time is the lane-fast index so global
// loads/stores of activations are coalesced along contiguous audio frames.
#include "ms_common.h"

namespace {

constexpr int kLdsBudgetBytes = 48 * 1024;

// ------------------------------------------------------------------ forward
template <int CT, int TT, int RC, int RT>
__global__ __launch_bounds__(256) void k_conv1d_fwd_direct(
    ConvP p, int cic, int span, const float* __restrict__ x, const float* __restrict__ x_act,
    int x_act_kind, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ res, float* __restrict__ y, float* __restrict__ y_act) {
    constexpr int NTX = TT / RT, NTY = CT / RC, CTP = CT + 1;
    static_assert(NTX * NTY == 256, "tile must map onto 256 threads");
    extern __shared__ float smem[];
    float* xs = smem;               // [cic][span]
    float* ws = smem + cic * span;  // [cic*K][CTP]
    const int tid = threadIdx.x;
    const int tx = tid % NTX, cy = tid / NTX;
    const int t0 = blockIdx.x * TT;
    const int tiles_per_group = (p.Og + CT - 1) / CT;
    const int g = blockIdx.y / tiles_per_group;
    const int co0 = (blockIdx.y % tiles_per_group) * CT;
    const int b = blockIdx.z;
    const int K = p.K;

    float acc[RC][RT];
#pragma unroll
    for (int i = 0; i < RC; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j) acc[i][j] = 0.f;

    // only the columns that valid outputs of this tile read are filled (short rows: L << TT)
    const int tvalid = min(TT, p.Lout - t0);
    const int span_used = min(span, (tvalid - 1) * p.stride + (K - 1) * p.dil + 1);
    for (int ci0 = 0; ci0 < p.Cg; ci0 += cic) {
        for (int idx = tid; idx < cic * span_used; idx += 256) {
            const int ci = idx / span_used, pos = idx - ci * span_used;
            float v = 0.f;
            if (ci0 + ci < p.Cg) {
                const int s = ms_src_index(t0 * p.stride - p.pad + pos, p.Lin, p.pad_mode);
                if (s >= 0) {
                    const size_t off = ((size_t)b * p.Cin + (size_t)g * p.Cg + ci0 + ci) * p.Lin + s;
                    v = x[off];
                    if (x_act) v = ms_act_grad(v, x_act[off], x_act_kind, p.slope);
                }
            }
            xs[ci * span + pos] = v;
        }
        const int rk = cic * K;
        for (int idx = tid; idx < rk * CT; idx += 256) {
            const int co = idx / rk, r = idx - co * rk;
            const int ci = r / K;
            float v = 0.f;
            if (co0 + co < p.Og && ci0 + ci < p.Cg)
                v = w[((size_t)(g * p.Og + co0 + co) * p.Cg + ci0) * K + r];
            ws[r * CTP + co] = v;
        }
        __syncthreads();
        for (int ci = 0; ci < cic; ++ci) {
            const float* xr = xs + ci * span + tx * p.stride;
            const float* wr = ws + (ci * K) * CTP + cy * RC;
            for (int k = 0; k < K; ++k) {
                float wv[RC], xv[RT];
#pragma unroll
                for (int i = 0; i < RC; ++i) wv[i] = wr[k * CTP + i];
#pragma unroll
                for (int j = 0; j < RT; ++j)   // columns past span_used belong to masked outputs
                    xv[j] = (tx + j * NTX < tvalid) ? xr[j * NTX * p.stride + k * p.dil] : 0.f;
#pragma unroll
                for (int i = 0; i < RC; ++i)
#pragma unroll
                    for (int j = 0; j < RT; ++j) acc[i][j] = fmaf(wv[i], xv[j], acc[i][j]);
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < RC; ++i) {
        const int col = co0 + cy * RC + i;
        if (col >= p.Og) continue;
        const int co = g * p.Og + col;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int j = 0; j < RT; ++j) {
            const int t = t0 + tx + j * NTX;
            if (t >= p.Lout) continue;
            const size_t o = ((size_t)b * p.Cout + co) * p.Lout + t;
            float v = ms_apply_act(acc[i][j] + bv, p.act, p.slope);
            if (y_act) y_act[o] = v;
            if (res) v = res[o] + v;
            y[o] = v;
        }
    }
}


// ---------------------------------------------- forward, few output channels, many inputs
// The discriminator's judge conv (1024 -> 1, k3) has 1024*3 MACs per output and only B*L outputs:
// it is a reduction, not a tiling problem.  32 time lanes x 8 channel slices per workgroup, each
// slice sums its channels (coalesced row reads), slices are combined through LDS.
__global__ __launch_bounds__(256) void k_conv1d_fwd_cred(ConvP p, const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ y) {
    __shared__ float red[8][33];
    const int tl = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int t = blockIdx.x * 32 + tl;
    const int co = blockIdx.y, b = blockIdx.z;
    const int g = co / p.Og;
    const int cper = (p.Cg + 7) / 8;
    const int c_lo = part * cper, c_hi = min(p.Cg, c_lo + cper);
    const float* xb = x + ((size_t)b * p.Cin + (size_t)g * p.Cg) * p.Lin;
    const float* wb = w + (size_t)co * p.Cg * p.K;
    float acc = 0.f;
    // branch-free taps: clamped address + mask, 4 channels in flight per iteration
    for (int k = 0; k < p.K; ++k) {
        const int sidx = ms_src_index(t * p.stride + k * p.dil - p.pad, p.Lin, p.pad_mode);
        const bool ok = t < p.Lout && sidx >= 0;
        const int so = ok ? sidx : 0;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int c = c_lo;
        for (; c + 3 < c_hi; c += 4) {
            const float x0 = xb[(size_t)(c + 0) * p.Lin + so], x1 = xb[(size_t)(c + 1) * p.Lin + so];
            const float x2 = xb[(size_t)(c + 2) * p.Lin + so], x3 = xb[(size_t)(c + 3) * p.Lin + so];
            a0 = fmaf(wb[(c + 0) * p.K + k], x0, a0);
            a1 = fmaf(wb[(c + 1) * p.K + k], x1, a1);
            a2 = fmaf(wb[(c + 2) * p.K + k], x2, a2);
            a3 = fmaf(wb[(c + 3) * p.K + k], x3, a3);
        }
        for (; c < c_hi; ++c) a0 = fmaf(wb[c * p.K + k], xb[(size_t)c * p.Lin + so], a0);
        if (ok) acc += (a0 + a1) + (a2 + a3);
    }
    red[part][tl] = acc;
    __syncthreads();
    if (part == 0 && t < p.Lout) {
        float v = bias ? bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) v += red[i][tl];
        y[((size_t)b * p.Cout + co) * p.Lout + t] = ms_apply_act(v, p.act, p.slope);
    }
}

// ------------------------------------------------------------ backward data
// gx[b, g*Cg+ci, s] = gx_add + out_act(bias + sum_{co,k} w[co,ci,k] * gp[b, co, (s+pad-k*dil)/stride])
// over the k with (s+pad-k*dil) divisible by stride.  Requires stride == 1 || dil == 1, and
// NTX % stride == 0 so that all RT positions of a thread share the same tap phase.
template <int CT, int TT, int RC, int RT>
__global__ __launch_bounds__(256) void k_conv1d_bwd_data_direct(
    ConvP p, int coc, int tspan, const float* __restrict__ gy, const float* __restrict__ y_act,
    const float* __restrict__ w, const float* __restrict__ bias, int out_act,
    const float* __restrict__ gx_add, float* __restrict__ gx) {
    constexpr int NTX = TT / RT, NTY = CT / RC, CTP = CT + 1;
    static_assert(NTX * NTY == 256, "tile must map onto 256 threads");
    extern __shared__ float smem[];
    float* gs = smem;                // [coc][tspan]
    float* ws = smem + coc * tspan;  // [coc*K][CTP]
    const int tid = threadIdx.x;
    const int tx = tid % NTX, cy = tid / NTX;
    const int s0 = blockIdx.x * TT;
    const int tiles_per_group = (p.Cg + CT - 1) / CT;
    const int g = blockIdx.y / tiles_per_group;
    const int ci0 = (blockIdx.y % tiles_per_group) * CT;
    const int b = blockIdx.z;
    const int K = p.K;
    // first output index whose receptive field can touch this tile (floor division, may be < 0)
    int num_lo = s0 + p.pad - (K - 1) * p.dil;
    int t_lo = num_lo / p.stride;
    if ((num_lo % p.stride) && num_lo < 0) --t_lo;

    float acc[RC][RT];
#pragma unroll
    for (int i = 0; i < RC; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j) acc[i][j] = 0.f;

    const int sA = s0 + tx;                    // this thread's first position
    const int k0 = (sA + p.pad) % p.stride;    // tap phase (sA + pad >= 0)
    const int tstep = NTX / p.stride;          // t increment between the thread's RT positions

    for (int co0 = 0; co0 < p.Og; co0 += coc) {
        for (int idx = tid; idx < coc * tspan; idx += 256) {
            const int co = idx / tspan, tt = idx - co * tspan;
            const int t = t_lo + tt;
            float v = 0.f;
            if (co0 + co < p.Og && t >= 0 && t < p.Lout) {
                const size_t off = ((size_t)b * p.Cout + (size_t)g * p.Og + co0 + co) * p.Lout + t;
                v = gy[off];
                if (y_act) v = ms_act_grad(v, y_act[off], p.act, p.slope);
            }
            gs[idx] = v;
        }
        const int qn = CT * K;
        for (int idx = tid; idx < coc * qn; idx += 256) {
            const int co = idx / qn, q = idx - co * qn;
            const int ci = q / K, k = q - ci * K;
            float v = 0.f;
            if (co0 + co < p.Og && ci0 + ci < p.Cg)
                v = w[((size_t)(g * p.Og + co0 + co) * p.Cg + ci0) * K + q];
            ws[(co * K + k) * CTP + ci] = v;
        }
        __syncthreads();
        for (int co = 0; co < coc; ++co) {
            const float* gr = gs + co * tspan;
            const float* wr = ws + (co * K) * CTP + cy * RC;
            for (int k = k0; k < K; k += p.stride) {
                const int base = (sA + p.pad - k * p.dil) / p.stride - t_lo;  // exact division
                float wv[RC], xv[RT];
#pragma unroll
                for (int i = 0; i < RC; ++i) wv[i] = wr[k * CTP + i];
#pragma unroll
                for (int j = 0; j < RT; ++j) xv[j] = gr[base + j * tstep];
#pragma unroll
                for (int i = 0; i < RC; ++i)
#pragma unroll
                    for (int j = 0; j < RT; ++j) acc[i][j] = fmaf(wv[i], xv[j], acc[i][j]);
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < RC; ++i) {
        const int cil = ci0 + cy * RC + i;
        if (cil >= p.Cg) continue;
        const int c = g * p.Cg + cil;
        const float bv = bias ? bias[c] : 0.f;
#pragma unroll
        for (int j = 0; j < RT; ++j) {
            const int s = sA + j * NTX;
            if (s >= p.Lin) continue;
            const size_t o = ((size_t)b * p.Cin + c) * p.Lin + s;
            float v = ms_apply_act(acc[i][j] + bv, out_act, p.slope);
            if (gx_add) v += gx_add[o];
            gx[o] = v;
        }
    }
}


// ------------------------------------------------- reflection-pad fold (backward data)
// A conv behind nn.ReflectionPad1d(pad) reads mirrored samples at the row ends.  The zero-padded
// backward-data pass covers the taps that read in-range samples; this kernel adds, for every padded
// border position u (u < pad or u >= L + pad), the gradient that reached it,
//   gpad[u] = sum_{co,j} w[co,ci,j] * gp[b,co,u - j*dil]      (0 <= u - j*dil < Lout),
// onto its mirror source s = reflect(u - pad).  2*pad positions per (b, ci) row: tiny.
__global__ __launch_bounds__(256) void k_reflect_fold_bwd(ConvP p, const float* __restrict__ gy,
                                                         const float* __restrict__ y_act,
                                                         const float* __restrict__ w,
                                                         float* __restrict__ gx) {
    // thread = (b, edge position e, input channel ci) with ci the lane-fast index: a wave shares
    // (b, e), so the gradient samples it reads are the same address for every lane (one cache line)
    // and the weights are read at a stride of K floats; 4 independent chains over the output channels
    // (the first version walked Cout x K dependent FMAs per thread with e lane-fast: 66 us).
    const int nb = 2 * p.pad;
    const long long total = (long long)p.B * p.Cin * nb;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ci = (int)(i % p.Cin);
    const int e = (int)((i / p.Cin) % nb);
    const int b = (int)(i / ((long long)nb * p.Cin));
    const int u = e < p.pad ? e : p.Lin + e;            // padded index (length Lin + 2*pad)
    const int sidx = ms_src_index(u - p.pad, p.Lin, MS_PAD_REFLECT);
    if (sidx < 0) return;
    const float* ya = y_act ? y_act : gy;
    const int kind = y_act ? p.act : MS_ACT_NONE;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int j = 0; j < p.K; ++j) {
        const int t = u - j * p.dil;
        if (t < 0 || t >= p.Lout) continue;             // uniform over the wave
        const float* wj = w + (size_t)ci * p.K + j;
        const size_t gbase = (size_t)b * p.Cout * p.Lout + t;
        int co = 0;
        for (; co + 3 < p.Cout; co += 4) {
            const size_t o0 = gbase + (size_t)co * p.Lout, o1 = o0 + p.Lout, o2 = o1 + p.Lout, o3 = o2 + p.Lout;
            const float g0 = gy[o0], g1 = gy[o1], g2 = gy[o2], g3 = gy[o3];
            const float y0 = ya[o0], y1 = ya[o1], y2 = ya[o2], y3 = ya[o3];
            const size_t ws = (size_t)p.Cin * p.K;
            const float w0 = wj[(size_t)co * ws], w1 = wj[(size_t)(co + 1) * ws], w2 = wj[(size_t)(co + 2) * ws],
                        w3 = wj[(size_t)(co + 3) * ws];
            a0 = fmaf(w0, ms_act_grad(g0, y0, kind, p.slope), a0);
            a1 = fmaf(w1, ms_act_grad(g1, y1, kind, p.slope), a1);
            a2 = fmaf(w2, ms_act_grad(g2, y2, kind, p.slope), a2);
            a3 = fmaf(w3, ms_act_grad(g3, y3, kind, p.slope), a3);
        }
        for (; co < p.Cout; ++co) {
            const size_t off = gbase + (size_t)co * p.Lout;
            a0 = fmaf(wj[(size_t)co * p.Cin * p.K], ms_act_grad(gy[off], ya[off], kind, p.slope), a0);
        }
    }
    atomicAdd(&gx[((size_t)b * p.Cin + ci) * p.Lin + sidx], (a0 + a1) + (a2 + a3));
}

// ---------------------------------------------------------- backward weight
// partial[z][co][j] = sum over this block's (b, t-chunk) slices of gp[b,co,t] * x[b, ci(j), t*stride + k(j)*dil - pad]
// with j = ci*K + k inside the group.  Partials (and the bias column) are summed by k_reduce_partials.
template <int COT, int JT>
__global__ __launch_bounds__(256) void k_conv1d_bwd_weight_direct(
    ConvP p, int TC, int nci, int xspan, int nsplit, const float* __restrict__ x,
    const float* __restrict__ x_act, int x_act_kind, const float* __restrict__ gy,
    const float* __restrict__ y_act, int y_act_kind, float* __restrict__ partial,
    size_t partial_stride) {
    constexpr int NCO = COT / 4, NJ = JT / 4, COTP = COT + 1;
    static_assert(NCO * NJ == 256, "tile must map onto 256 threads");
    extern __shared__ float smem[];
    float* gs = smem;              // [TC][COTP]
    float* xs = smem + TC * COTP;  // [nci][xspan]
    const int tid = threadIdx.x;
    const int jg = tid % NJ, cog = tid / NJ;
    const int K = p.K;
    const int J = p.Cg * K;
    const int j0 = blockIdx.x * JT;
    const int co_tiles = (p.Og + COT - 1) / COT;
    const int g = blockIdx.y / co_tiles;
    const int co0 = (blockIdx.y % co_tiles) * COT;
    const int ci_start = j0 / K;

    int xoff[4];
    bool jvalid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + jg * 4 + r;
        jvalid[r] = j < J;
        const int jj = jvalid[r] ? j : j0;
        const int ci = jj / K, k = jj - ci * K;
        xoff[r] = (ci - ci_start) * xspan + k * p.dil;
    }
    float acc[4][4];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;

    const int tchunks = (p.Lout + TC - 1) / TC;
    const int nchunks = p.B * tchunks;
    for (int ch = blockIdx.z; ch < nchunks; ch += nsplit) {
        const int b = ch / tchunks, tc0 = (ch - b * tchunks) * TC;
        for (int idx = tid; idx < COT * TC; idx += 256) {
            const int co = idx / TC, t = idx - co * TC;
            float v = 0.f;
            if (co0 + co < p.Og && tc0 + t < p.Lout) {
                const size_t off = ((size_t)b * p.Cout + (size_t)g * p.Og + co0 + co) * p.Lout + tc0 + t;
                v = gy[off];
                if (y_act) v = ms_act_grad(v, y_act[off], y_act_kind, p.slope);
            }
            gs[t * COTP + co] = v;
        }
        for (int idx = tid; idx < nci * xspan; idx += 256) {
            const int ci = idx / xspan, pos = idx - ci * xspan;
            float v = 0.f;
            if (ci_start + ci < p.Cg) {
                const int s = ms_src_index(tc0 * p.stride - p.pad + pos, p.Lin, p.pad_mode);
                if (s >= 0) {
                    const size_t off = ((size_t)b * p.Cin + (size_t)g * p.Cg + ci_start + ci) * p.Lin + s;
                    v = x[off];
                    if (x_act) v = ms_act_grad(v, x_act[off], x_act_kind, p.slope);
                }
            }
            xs[idx] = v;
        }
        __syncthreads();
        for (int t = 0; t < TC; ++t) {
            float gv[4], xv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) gv[c] = gs[t * COTP + cog * 4 + c];
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[r] = xs[xoff[r] + t * p.stride];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bsum[c] += gv[c];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[c][r] = fmaf(gv[c], xv[r], acc[c][r]);
            }
        }
        __syncthreads();
    }

    float* part = partial + (size_t)blockIdx.z * partial_stride;
    const size_t wsize = (size_t)p.Cout * J;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = co0 + cog * 4 + c;
        if (col >= p.Og) continue;
        const int co = g * p.Og + col;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (jvalid[r]) part[(size_t)co * J + j0 + jg * 4 + r] = acc[c][r];
        if (blockIdx.x == 0 && jg == 0) part[wsize + co] = bsum[c];
    }
}

// out[i] = beta*out[i] + sum_z partial[z][i]; the trailing Cout entries of a slab are the bias grads
__global__ __launch_bounds__(256) void k_reduce_partials(const float* __restrict__ partial,
                                                        size_t partial_stride, int nsplit,
                                                        size_t wsize, int nbias,
                                                        float* __restrict__ gw,
                                                        float* __restrict__ gb, float beta) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= wsize + (size_t)nbias) return;
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += partial[(size_t)z * partial_stride + i];
    if (i < wsize) {
        gw[i] = (beta != 0.f ? beta * gw[i] : 0.f) + s;
    } else if (gb) {
        const size_t c = i - wsize;
        gb[c] = (beta != 0.f ? beta * gb[c] : 0.f) + s;
    }
}

// Same reduction for FEW outputs and MANY partial slabs (the 1-output-channel convs: 225 outputs,
// up to 2048 slabs): one wave per output, lanes stride over the slabs (independent loads), fixed
// shuffle tree.  The one-thread-per-output loop above is a serial chain of nsplit dependent
// iterations -- 125 us for 225 outputs.
__global__ __launch_bounds__(256) void k_reduce_partials_wave(const float* __restrict__ partial,
                                                             size_t partial_stride, int nsplit,
                                                             size_t wsize, int nbias,
                                                             float* __restrict__ gw,
                                                             float* __restrict__ gb, float beta) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 4 + wv;
    if (i >= wsize + (size_t)nbias) return;          // wave-uniform
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = lane;
    for (; z + 192 < nsplit; z += 256) {
        const float a = partial[(size_t)z * partial_stride + i];
        const float b = partial[(size_t)(z + 64) * partial_stride + i];
        const float c = partial[(size_t)(z + 128) * partial_stride + i];
        const float d = partial[(size_t)(z + 192) * partial_stride + i];
        s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; z < nsplit; z += 64) s0 += partial[(size_t)z * partial_stride + i];
    float s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        if (i < wsize) gw[i] = (beta != 0.f ? beta * gw[i] : 0.f) + s;
        else if (gb) gb[i - wsize] = (beta != 0.f ? beta * gb[i - wsize] : 0.f) + s;
    }
}

// out[c] = beta*out[c] + sum_{b,t} g[b,c,t] * act'(y_act[b,c,t]); one workgroup per (channel, batch
// slice) writes a partial, the last stage folds the slices in a fixed order (deterministic)
constexpr int kChanSlices = 32;

__global__ __launch_bounds__(256) void k_channel_sum_partial(const float* __restrict__ gsrc,
                                                            const float* __restrict__ y_act,
                                                            int act, float slope, int B, int C,
                                                            int L, float* __restrict__ partial) {
    __shared__ float red[4];
    const int c = blockIdx.x, sl = blockIdx.y;
    const float* ya = y_act ? y_act : gsrc;
    const int kind = y_act ? act : MS_ACT_NONE;
    float s = 0.f;
    for (int b = sl; b < B; b += kChanSlices) {
        const size_t base = ((size_t)b * C + c) * L;
        for (int t = threadIdx.x; t < L; t += 256) s += ms_act_grad(gsrc[base + t], ya[base + t], kind, slope);
    }
    const float tot = ms_block_sum(s, red);
    if (threadIdx.x == 0) partial[(size_t)sl * C + c] = tot;
}

__global__ __launch_bounds__(256) void k_channel_sum_final(const float* __restrict__ partial, int C,
                                                          float* __restrict__ out, float beta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int i = 0; i < kChanSlices; ++i) s += partial[(size_t)i * C + c];
    out[c] = (beta != 0.f ? beta * out[c] : 0.f) + s;
}

template <int CT>
int pick_chunk(int per_unit_floats_a, int per_unit_floats_b, int limit) {
    int c = (kLdsBudgetBytes / 4) / (per_unit_floats_a + per_unit_floats_b);
    if (c < 1) c = 1;
    if (c > limit) c = limit;
    return c;
}

template <int CT, int TT, int RC, int RT>
int launch_fwd(const ConvP& p, const float* x, const float* x_act, int x_act_kind, const float* w,
               const float* bias, const float* res, float* y, float* y_act, hipStream_t s) {
    const int span = (TT - 1) * p.stride + (p.K - 1) * p.dil + 1;
    const int cic = pick_chunk<CT>(span, p.K * (CT + 1), p.Cg);
    const size_t lds = (size_t)(cic * span + cic * p.K * (CT + 1)) * sizeof(float);
    if (lds > 64 * 1024) return MS_ERR_UNSUPPORTED;
    dim3 grid(ms_ceil_div(p.Lout, TT), p.groups * ms_ceil_div(p.Og, CT), p.B);
    hipLaunchKernelGGL((k_conv1d_fwd_direct<CT, TT, RC, RT>), grid, dim3(256), lds, s, p, cic, span,
                       x, x_act, x_act_kind, w, bias, res, y, y_act);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

template <int CT, int TT, int RC, int RT>
int launch_bwd_data(const ConvP& p, const float* gy, const float* y_act, const float* w,
                    const float* bias, int out_act, const float* gx_add, float* gx,
                    hipStream_t s) {
    constexpr int NTX = TT / RT;
    if (NTX % p.stride) return MS_ERR_UNSUPPORTED;
    const int tspan = (TT - 1 + (p.K - 1) * p.dil) / p.stride + 2;
    const int coc = pick_chunk<CT>(tspan, p.K * (CT + 1), p.Og);
    const size_t lds = (size_t)(coc * tspan + coc * p.K * (CT + 1)) * sizeof(float);
    if (lds > 64 * 1024) return MS_ERR_UNSUPPORTED;
    dim3 grid(ms_ceil_div(p.Lin, TT), p.groups * ms_ceil_div(p.Cg, CT), p.B);
    hipLaunchKernelGGL((k_conv1d_bwd_data_direct<CT, TT, RC, RT>), grid, dim3(256), lds, s, p, coc,
                       tspan, gy, y_act, w, bias, out_act, gx_add, gx);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

struct WgPlan {
    int cot, jt, TC, nci, xspan, nsplit, jtiles, cotiles;
    size_t stride_floats, lds;
};

WgPlan plan_bwd_weight(const ConvP& p) {
    WgPlan q;
    if (p.Og >= 64) { q.cot = 64; q.jt = 64; }
    else if (p.Og >= 32) { q.cot = 32; q.jt = 128; }
    else if (p.Og >= 8) { q.cot = 16; q.jt = 256; }
    else { q.cot = 4; q.jt = 1024; }
    const int J = p.Cg * p.K;
    int nci = q.jt / p.K + 2;
    if (nci > p.Cg) nci = p.Cg;
    q.nci = nci;
    // time chunk per LDS fill: as long as fits the LDS budget (wide J tiles of many-channel,
    // few-output layers such as the 1024->1 judge conv need a short chunk)
    q.TC = 64;
    for (;;) {
        q.xspan = (q.TC - 1) * p.stride + (p.K - 1) * p.dil + 1;
        const size_t lds = (size_t)(q.TC * (q.cot + 1) + q.nci * q.xspan) * sizeof(float);
        if (lds <= (size_t)kLdsBudgetBytes || q.TC <= 4) break;
        q.TC >>= 1;
    }
    q.jtiles = ms_ceil_div(J, q.jt);
    q.cotiles = ms_ceil_div(p.Og, q.cot);
    const int tiles = q.jtiles * q.cotiles * p.groups;
    const int nchunks = p.B * ms_ceil_div(p.Lout, q.TC);
    int ns = ms_ceil_div(2048, tiles);
    if (ns > nchunks) ns = nchunks;
    q.stride_floats = (size_t)p.Cout * J + p.Cout;
    const size_t cap = (size_t)64 << 20;  // keep the partial slabs <= 64 MiB
    while (ns > 1 && (size_t)ns * q.stride_floats * 4 > cap) --ns;
    if (ns < 1) ns = 1;
    q.nsplit = ns;
    q.lds = (size_t)(q.TC * (q.cot + 1) + q.nci * q.xspan) * sizeof(float);
    return q;
}

}  // namespace

const char* msk_conv1d_fwd_direct_name(const ConvP& p) {
    if (p.Cout <= 4 && p.Cg >= 256) return "k_conv1d_fwd_cred";
    if (p.Og >= 32) return "k_conv1d_fwd_direct<64, 64, 4, 4>";
    if (p.Og >= 8) return "k_conv1d_fwd_direct<16, 256, 4, 4>";
    return "k_conv1d_fwd_direct<4, 256, 1, 4>";
}

const char* msk_conv1d_bwd_data_direct_name(const ConvP& p) {
    if (p.Cg >= 32) return "k_conv1d_bwd_data_direct<64, 64, 4, 4>";
    if (p.Cg >= 8) return "k_conv1d_bwd_data_direct<16, 256, 4, 4>";
    return "k_conv1d_bwd_data_direct<4, 256, 1, 4>";
}

const char* msk_conv1d_bwd_weight_direct_name(const ConvP& p) {
    const WgPlan q = plan_bwd_weight(p);
    if (q.cot == 64) return "k_conv1d_bwd_weight_direct<64, 64>";
    if (q.cot == 32) return "k_conv1d_bwd_weight_direct<32, 128>";
    if (q.cot == 16) return "k_conv1d_bwd_weight_direct<16, 256>";
    return "k_conv1d_bwd_weight_direct<4, 1024>";
}

static bool use_cred(const ConvP& p, const float* x_act, const float* residual, const float* y_act) {
    return p.Cout <= 4 && p.Cg >= 256 && !x_act && !residual && !y_act;
}

int msk_conv1d_fwd_direct(const ConvP& p, const float* x, const float* x_act, int x_act_kind,
                          const float* w, const float* bias, const float* residual, float* y,
                          float* y_act, hipStream_t s) {
    if (use_cred(p, x_act, residual, y_act)) {
        dim3 grid(ms_ceil_div(p.Lout, 32), p.Cout, p.B);
        hipLaunchKernelGGL(k_conv1d_fwd_cred, grid, dim3(256), 0, s, p, x, w, bias, y);
        MS_CHECK_LAUNCH();
        return MS_OK;
    }
    if (p.Og >= 32) return launch_fwd<64, 64, 4, 4>(p, x, x_act, x_act_kind, w, bias, residual, y, y_act, s);
    if (p.Og >= 8) return launch_fwd<16, 256, 4, 4>(p, x, x_act, x_act_kind, w, bias, residual, y, y_act, s);
    return launch_fwd<4, 256, 1, 4>(p, x, x_act, x_act_kind, w, bias, residual, y, y_act, s);
}

int msk_conv1d_bwd_data_direct(const ConvP& p, const float* gy, const float* y_act, const float* w,
                               const float* bias, int out_act, const float* gx_add, float* gx,
                               hipStream_t s) {
    if (p.pad_mode != MS_PAD_ZERO) return MS_ERR_UNSUPPORTED;
    if (p.stride != 1 && p.dil != 1) return MS_ERR_UNSUPPORTED;
    if (p.Cg >= 32) return launch_bwd_data<64, 64, 4, 4>(p, gy, y_act, w, bias, out_act, gx_add, gx, s);
    if (p.Cg >= 8) return launch_bwd_data<16, 256, 4, 4>(p, gy, y_act, w, bias, out_act, gx_add, gx, s);
    return launch_bwd_data<4, 256, 1, 4>(p, gy, y_act, w, bias, out_act, gx_add, gx, s);
}

size_t msk_conv1d_bwd_weight_ws(const ConvP& p) {
    const WgPlan q = plan_bwd_weight(p);
    return (size_t)q.nsplit * q.stride_floats * sizeof(float);
}

int msk_conv1d_bwd_weight_direct(const ConvP& p, const float* x, const float* x_act,
                                 int x_act_kind, const float* gy, const float* y_act,
                                 int y_act_kind, float* gw, float* gb, float beta, void* ws,
                                 size_t ws_bytes, hipStream_t s) {
    const WgPlan q = plan_bwd_weight(p);
    const size_t need = (size_t)q.nsplit * q.stride_floats * sizeof(float);
    if (!ws || ws_bytes < need) return MS_ERR_WORKSPACE;
    if (q.lds > 64 * 1024) return MS_ERR_UNSUPPORTED;
    float* partial = (float*)ws;
    dim3 grid(q.jtiles, p.groups * q.cotiles, q.nsplit);
#define MS_LAUNCH_WG(COT, JT)                                                                      \
    hipLaunchKernelGGL((k_conv1d_bwd_weight_direct<COT, JT>), grid, dim3(256), q.lds, s, p, q.TC,  \
                       q.nci, q.xspan, q.nsplit, x, x_act, x_act_kind, gy, y_act, y_act_kind,      \
                       partial, q.stride_floats)
    if (q.cot == 64) MS_LAUNCH_WG(64, 64);
    else if (q.cot == 32) MS_LAUNCH_WG(32, 128);
    else if (q.cot == 16) MS_LAUNCH_WG(16, 256);
    else MS_LAUNCH_WG(4, 1024);
#undef MS_LAUNCH_WG
    MS_CHECK_LAUNCH();
    const size_t wsize = (size_t)p.Cout * p.Cg * p.K;
    return msk_reduce_partials(partial, q.stride_floats, q.nsplit, wsize, p.Cout, gw, gb, beta, s);
}

int msk_reduce_partials(const float* partial, size_t partial_stride, int nsplit, size_t wsize,
                        int nbias, float* gw, float* gb, float beta, hipStream_t s) {
    const size_t total = wsize + (size_t)nbias;
    if (total <= 8192 && nsplit >= 64)
        hipLaunchKernelGGL(k_reduce_partials_wave, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s,
                           partial, partial_stride, nsplit, wsize, nbias, gw, gb, beta);
    else
        hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                           partial, partial_stride, nsplit, wsize, nbias, gw, gb, beta);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

size_t msk_channel_sum_ws(int C) { return (size_t)kChanSlices * C * sizeof(float); }

int msk_channel_sum(const float* g, const float* y_act, int act, float slope, int B, int C, int L,
                    float* out, float beta, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < msk_channel_sum_ws(C)) return MS_ERR_WORKSPACE;
    float* partial = (float*)ws;
    hipLaunchKernelGGL(k_channel_sum_partial, dim3(C, kChanSlices), dim3(256), 0, s, g, y_act, act,
                       slope, B, C, L, partial);
    MS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_channel_sum_final, dim3((C + 255) / 256), dim3(256), 0, s, partial, C, out, beta);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int msk_reflect_fold_bwd(const ConvP& p, const float* gy, const float* y_act, const float* w,
                         float* gx, hipStream_t s) {
    if (p.pad <= 0) return MS_OK;
    const long long total = (long long)p.B * p.Cin * 2 * p.pad;
    hipLaunchKernelGGL(k_reflect_fold_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, gy,
                       y_act, w, gx);
    MS_CHECK_LAUNCH();
    return MS_OK;
}
