// The discriminator's two one-channel-side layers over ALL scales in one launch each way (ms_conv1d_parts_*):
//   first conv   Conv1d(1, 16, 15, padding=7) + LeakyReLU      reference discriminator/full.py:14
//   judge conv   Conv1d(1024, 1, 3, padding=1)                  reference discriminator/full.py:22
// applied by the reference's MelGanDiscriminator to x, pool(x), pool(pool(x)) (discriminator/melgan.py:13-27).  Both are
// streams over the many-channel tensor and bound by HBM, not by arithmetic; per scale they were 3 x 6 launches of 7-27 us
// for 2-34 MB each.  Here a launch walks a table of parts (batch rows x row length per scale); rows of any length are read
// and written with 16-byte accesses that need not be 16-byte aligned (the memory pipe takes them), samples outside a row
// are cleared / skipped, and every tensor is read through a buffer descriptor of its true size.
//
// Arithmetic is plain fp32 FMA in a fixed order: results are deterministic; they differ from the per-scale kernels only by
// summation order (tests/test_gpu_parts.py: float64 reference 1e-6, per-scale launches 1e-6).
#include "ms_common.h"
#include "gconv_mfma.h"
#include <stdint.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr unsigned OOB = 0xF0000000u;
constexpr int NP = MS_CONV_PARTS_MAX;

// Table of parts.  wg0[i]: first workgroup of part i (a prefix sum; wg0[count] = grid size).
struct DParts {
    int count, wg0[NP + 1];
    int B[NP], L[NP], tiles[NP];
    const float* a[NP];
    const float* b[NP];
    const float* c[NP];
    float* o[NP];
};

// (compile-time indices into the by-value tables: a run-time index into kernel-argument pointer arrays crashes the
//  compiler's argument promotion, ROCm 7.2 -- conv5_img.hip)
struct DPart {
    int B, L, tiles, wg;              // wg: workgroup index within the part
    const float* a;
    const float* b;
    const float* c;
    float* o;
};
__device__ __forceinline__ DPart pick_part(const DParts& q, int wg) {
    DPart p{q.B[0], q.L[0], q.tiles[0], wg, q.a[0], q.b[0], q.c[0], q.o[0]};
#pragma unroll
    for (int k = 1; k < NP; ++k)
        if (k < q.count && wg >= q.wg0[k]) p = DPart{q.B[k], q.L[k], q.tiles[k], wg - q.wg0[k], q.a[k], q.b[k], q.c[k], q.o[k]};
    return p;
}

__device__ __forceinline__ float lrelu_grad(float g, float y, float slope) { return y > 0.f ? g : g * slope; }

// 4 consecutive samples t .. t+3 of the row that starts at element `row_elems` (length L) of a tensor read through rs: one
// (possibly unaligned) 16-byte load; samples outside [0, L) read 0.0.  t must be a multiple of 4: the vector then lies wholly
// in front of the row or starts inside it, and only the one across the row END needs clearing.
__device__ __forceinline__ f32x4 load_row4(__amdgpu_buffer_rsrc_t rs, unsigned row_elems, int t, int L) {
    const bool any = t >= 0 && t < L;
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, any ? (row_elems + (unsigned)t) * 4u : OOB, 0, 0));
#pragma unroll
    for (int e = 1; e < 4; ++e) v[e] = t + e < L ? v[e] : 0.f;
    return v;
}

// ------------------------------------------------------------------------------------------------ first conv, forward
// y[b, co, t] = lrelu(bias[co] + sum_k w[co, k] x[b, 0, t + k - 7]).  A workgroup owns 1024 samples of one (part, batch row):
// the input window goes to LDS once, every thread computes 4 consecutive samples of all 16 channels.
constexpr int FK = 15, FPAD = 7, FC = 16, FT = 1024;

// stages x[tile - 8 .. tile + 1032) of row b into xs (1040 floats); returns after the barrier
__device__ __forceinline__ void stage_thin_window(float* xs, const float* x, int B, int L, int b, int tile_t) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 4u * (unsigned)(B * L), 0x00020000);
    for (int i = threadIdx.x; i < (FT + 16) / 4; i += 256) {
        const int t = tile_t - 8 + 4 * i;            // a multiple of 4: wholly in front of the row, or starting inside it
        *reinterpret_cast<f32x4*>(xs + 4 * i) = load_row4(rs, (unsigned)(b * L), t, L);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_dfirst_fwd(DParts q, const float* __restrict__ w, const float* __restrict__ bias,
                                                   float slope) {
    __shared__ __attribute__((aligned(16))) float xs[FT + 16];
    const DPart p = pick_part(q, blockIdx.x);
    const int b = p.wg / p.tiles, tile_t = (p.wg - b * p.tiles) * FT;
    stage_thin_window(xs, p.a, p.B, p.L, b, tile_t);
    const int t0 = tile_t + 4 * threadIdx.x;
    if (t0 >= p.L) return;
    float win[20];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xs + 4 * threadIdx.x + 4 * i);
        win[4 * i] = v[0]; win[4 * i + 1] = v[1]; win[4 * i + 2] = v[2]; win[4 * i + 3] = v[3];
    }
    const bool whole = t0 + 3 < p.L;
#pragma unroll 4
    for (int co = 0; co < FC; ++co) {
        float v[4];
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = bv;
#pragma unroll
        for (int k = 0; k < FK; ++k) {
            const float wk = w[co * FK + k];                       // wave-uniform: scalar load
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(wk, win[1 + e + k], v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);       // LeakyReLU, slope in [0, 1]
        float* yr = p.o + ((size_t)b * FC + co) * p.L + t0;
        if (whole) {
            *reinterpret_cast<f32x4u*>(yr) = (f32x4u){v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (t0 + e < p.L) yr[e] = v[e];
        }
    }
}

// ------------------------------------------------------------------------------------------------ first conv, backward data
// gx[b, 0, t] = sum_co sum_k w[co, k] gp[b, co, t + 7 - k],  gp = gy * lrelu'(y).   (G-step only: the gradient that reaches
// the generator.)  Workgroup = 256 samples of one (part, batch row); wave w sums channels 4w .. 4w+3, LDS combines the waves.
__global__ __launch_bounds__(256) void k_dfirst_bwd_data(DParts q, const float* __restrict__ w, float slope) {
    __shared__ float red[4][256 + 4];
    const DPart p = pick_part(q, blockIdx.x);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int b = p.wg / p.tiles, tile_t = (p.wg - b * p.tiles) * 256;
    const int t0 = tile_t + 4 * lane;
    const unsigned bytes = 4u * (unsigned)(p.B * FC * p.L);
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, bytes, 0x00020000);
    const auto rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, bytes, 0x00020000);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int co = wid * 4 + cc;
        const unsigned row = (unsigned)((b * FC + co) * p.L);
        float win[20];                                  // gp[t0 - 8 .. t0 + 12)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const f32x4 g = load_row4(rsG, row, t0 - 8 + 4 * i, p.L);
            const f32x4 y = load_row4(rsY, row, t0 - 8 + 4 * i, p.L);
#pragma unroll
            for (int e = 0; e < 4; ++e) win[4 * i + e] = lrelu_grad(g[e], y[e], slope);
        }
#pragma unroll
        for (int k = 0; k < FK; ++k) {
            const float wk = w[co * FK + k];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(wk, win[8 + e + FPAD - k], acc[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wid][4 * lane + e] = acc[e];
    __syncthreads();
    const int to = tile_t + threadIdx.x;
    if (to < p.L) {
        const int i = threadIdx.x;
        float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        if (p.c) v += p.c[(size_t)b * p.L + to];
        p.o[(size_t)b * p.L + to] = v;
    }
}

// ------------------------------------------------------------------------------------------------ first conv, weight gradient
// gw[co, 0, k] = sum_{b, t} gp[b, co, t] x[b, 0, t + k - 7],  gb[co] = sum gp.  Persistent workgroups walk (part, batch row,
// 1024-sample tile) units: the x window in LDS, wave w accumulates channels 4w .. 4w+3 in registers over ALL its units and
// the lanes are combined once at the end -- one partial row per workgroup (FC * FK weights, then FC biases), summed in
// slab order by k_dparts_reduce.
__global__ __launch_bounds__(256) void k_dfirst_wgrad(DParts q, float slope, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float xs[FT + 16];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc[4][FK], bsum[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        bsum[cc] = 0.f;
#pragma unroll
        for (int k = 0; k < FK; ++k) acc[cc][k] = 0.f;
    }
    const int nunits = q.wg0[q.count];
    for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        const DPart p = pick_part(q, unit);
        const int b = p.wg / p.tiles, tile_t = (p.wg - b * p.tiles) * FT;
        __syncthreads();                               // (the previous unit's window is no longer read)
        stage_thin_window(xs, p.a, p.B, p.L, b, tile_t);
        const unsigned bytes = 4u * (unsigned)(p.B * FC * p.L);
        const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b), 0, bytes, 0x00020000);
        const auto rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.c), 0, bytes, 0x00020000);
#pragma unroll 1
        for (int pass = 0; pass < FT / 256; ++pass) {
            const int sl = pass * 256 + 4 * lane, t0 = tile_t + sl;
            if (tile_t + pass * 256 >= p.L) break;      // wave-uniform
            float win[20];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xs + sl + 4 * i);
                win[4 * i] = v[0]; win[4 * i + 1] = v[1]; win[4 * i + 2] = v[2]; win[4 * i + 3] = v[3];
            }
            f32x4 g[4], y[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const unsigned row = (unsigned)((b * FC + wid * 4 + cc) * p.L);
                g[cc] = load_row4(rsG, row, t0, p.L);
                y[cc] = load_row4(rsY, row, t0, p.L);
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = lrelu_grad(g[cc][e], y[cc][e], slope);      // 0 outside the row (g is 0 there)
                    bsum[cc] += v;
#pragma unroll
                    for (int k = 0; k < FK; ++k) acc[cc][k] = fmaf(v, win[1 + e + k], acc[cc][k]);
                }
        }
    }
    float* prow = partial + (size_t)blockIdx.x * (FC * FK + FC);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int co = wid * 4 + cc;
#pragma unroll
        for (int k = 0; k < FK; ++k) {
            const float r = ms_wave_sum(acc[cc][k]);
            if (lane == 0) prow[co * FK + k] = r;
        }
        const float r = ms_wave_sum(bsum[cc]);
        if (lane == 0) prow[FC * FK + co] = r;
    }
}

// out[i] = beta * out[i] + sum_z partial[z][i] (z in order): one wave per output, lanes stride over the slabs.
__global__ __launch_bounds__(256) void k_dparts_reduce(const float* __restrict__ partial, int nslabs, int stride, int nw,
                                                      int nb, float* __restrict__ gw, float* __restrict__ gb, float beta) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nw + nb) return;
    float s = 0.f;
    for (int z = lane; z < nslabs; z += 64) s += partial[(size_t)z * stride + i];
    s = ms_wave_sum(s);
    if (lane == 0 && (i < nw || gb)) {
        float* dst = i < nw ? gw + i : gb + (i - nw);
        *dst = (beta != 0.f ? beta * *dst : 0.f) + s;
    }
}

// ------------------------------------------------------------------------------------------------ judge conv
// Rows of L <= 32 samples, C = 1024 channels (a multiple of 256): the C * L floats of a batch row are contiguous and start
// 16-byte aligned for every L, so they are read as a flat stream of aligned vectors and transposed through LDS into
// [channel][sample] rows of an odd pitch (conflict-free row reads).
constexpr int JL = 32, JP = 33;            // longest row, LDS pitch

// 256 channels x L samples starting at src (16-byte aligned, contiguous) -> lds[256][JP]
__device__ __forceinline__ void stage_rows(float* lds, const float* __restrict__ src, int L) {
    const int nv = 64 * L;                                   // 16-byte vectors
    const float invL = 1.f / (float)L;
    for (int v = threadIdx.x; v < nv; v += 256) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(src + 4 * v);
        int r = (int)(((float)(4 * v) + 0.5f) * invL);       // 4 v / L (exact: 4 v < 2^14)
        int t = 4 * v - r * L;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lds[r * JP + t] = x[e];
            if (++t == L) { t = 0; ++r; }
        }
    }
}

// forward: y[b, 0, t] = bias + sum_ci sum_k w[ci, k] x[b, ci, t + k - 1].  Workgroup = one (part, batch row): four chunks of
// 256 channels, thread = one channel of the chunk, JL accumulators; the block is summed at the end.
__global__ __launch_bounds__(256) void k_djudge_fwd(DParts q, int C, const float* __restrict__ w, const float* __restrict__ bias) {
    __shared__ __attribute__((aligned(16))) float rows[256 * JP];
    __shared__ float red[4][JL];
    const DPart p = pick_part(q, blockIdx.x);
    const int b = p.wg, L = p.L, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float acc[JL];
#pragma unroll
    for (int t = 0; t < JL; ++t) acc[t] = 0.f;
    for (int c0 = 0; c0 < C; c0 += 256) {
        __syncthreads();
        stage_rows(rows, p.a + ((size_t)b * C + c0) * L, L);
        __syncthreads();
        const float w0 = w[(c0 + tid) * 3], w1 = w[(c0 + tid) * 3 + 1], w2 = w[(c0 + tid) * 3 + 2];
        float v[JL + 2];
        v[0] = 0.f;
#pragma unroll
        for (int t = 0; t < JL; ++t) v[t + 1] = t < L ? rows[tid * JP + t] : 0.f;
        v[JL + 1] = 0.f;
#pragma unroll
        for (int t = 0; t < JL; ++t) acc[t] = fmaf(w2, v[t + 2], fmaf(w1, v[t + 1], fmaf(w0, v[t], acc[t])));
    }
#pragma unroll
    for (int t = 0; t < JL; ++t) {
        const float s = ms_wave_sum(acc[t]);
        if (lane == 0) red[wid][t] = s;
    }
    __syncthreads();
    if (tid < L) p.o[(size_t)b * L + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]) + (bias ? bias[0] : 0.f);
}

// backward data: gx[b, ci, t] = sum_k w[ci, k] gj[b, t + 1 - k] (+ gx_add).  Workgroup = (part, batch row, chunk of 256
// channels); the output block is a flat aligned stream.
__global__ __launch_bounds__(256) void k_djudge_bwd_data(DParts q, int C, const float* __restrict__ w) {
    __shared__ float gj[JL + 2];
    __shared__ float ws[256 * 3];
    const DPart p = pick_part(q, blockIdx.x);
    const int chunks = C / 256, L = p.L, tid = threadIdx.x;
    const int b = p.wg / chunks, c0 = (p.wg - b * chunks) * 256;
    if (tid < JL + 2) {
        const int t = tid - 1;
        gj[tid] = (t >= 0 && t < L) ? p.a[(size_t)b * L + t] : 0.f;
    }
    for (int i = tid; i < 256 * 3; i += 256) ws[i] = w[c0 * 3 + i];
    __syncthreads();
    const size_t base = ((size_t)b * C + c0) * L;
    const float invL = 1.f / (float)L;
    for (int v = tid; v < 64 * L; v += 256) {
        int r = (int)(((float)(4 * v) + 0.5f) * invL);
        int t = 4 * v - r * L;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // gj[] is shifted by one: gj[t + 1] = sample t
            o[e] = fmaf(ws[r * 3], gj[t + 2], fmaf(ws[r * 3 + 1], gj[t + 1], ws[r * 3 + 2] * gj[t]));
            if (++t == L) { t = 0; ++r; }
        }
        if (p.c) o += *reinterpret_cast<const f32x4*>(p.c + base + 4 * v);
        *reinterpret_cast<f32x4*>(p.o + base + 4 * v) = o;
    }
}

// weight gradient: gw[0, ci, k] = sum_{b, t} gj[b, t] x[b, ci, t + k - 1],  gb = sum gj.  Workgroup = (chunk of 256 channels,
// slab of the (part, batch row) list): thread = channel, three accumulators over the slab's rows; one partial row per
// workgroup ([slab][C * 3 + 1]: every chunk writes its 768 weights, chunk 0 the bias), summed by k_dparts_reduce.
__global__ __launch_bounds__(256) void k_djudge_wgrad(DParts q, int C, int nslabs, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float rows[256 * JP];
    __shared__ float gj[JL + 2];
    __shared__ float red[4];
    const int chunks = C / 256, tid = threadIdx.x;
    const int chunk = blockIdx.x % chunks, slab = blockIdx.x / chunks;
    const int c0 = chunk * 256;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, bs = 0.f;
    const int nrows = q.wg0[q.count];                 // (part, batch row) pairs: tiles == 1
    for (int row = slab; row < nrows; row += nslabs) {
        const DPart p = pick_part(q, row);
        const int b = p.wg, L = p.L;
        __syncthreads();
        stage_rows(rows, p.a + ((size_t)b * C + c0) * L, L);
        if (tid < JL + 2) {
            const int t = tid - 1;
            gj[tid] = (t >= 0 && t < L) ? p.b[(size_t)b * L + t] : 0.f;
        }
        __syncthreads();
        for (int t = 0; t < L; ++t) {
            const float x = rows[tid * JP + t];
            // x[t] meets gj[t + 1 - k] for tap k (gj[] shifted by one)
            a0 = fmaf(x, gj[t + 2], a0);
            a1 = fmaf(x, gj[t + 1], a1);
            a2 = fmaf(x, gj[t], a2);
        }
        if (chunk == 0 && tid < L) bs += gj[tid + 1];
    }
    float* prow = partial + (size_t)slab * (C * 3 + 1);
    prow[(c0 + tid) * 3] = a0; prow[(c0 + tid) * 3 + 1] = a1; prow[(c0 + tid) * 3 + 2] = a2;
    if (chunk == 0) {
        const float tot = ms_block_sum(bs, red);
        if (tid == 0) prow[C * 3] = tot;
    }
}

// ------------------------------------------------------------------------------------------------ host side
bool is_first(const ConvP& c) {
    return c.groups == 1 && c.stride == 1 && c.dil == 1 && c.pad_mode == MS_PAD_ZERO && !c.in_act && c.Cin == 1 && c.Cout == FC &&
           c.K == FK && c.pad == FPAD && c.act == MS_ACT_LRELU && c.slope >= 0.f && c.slope <= 1.f;
}
bool is_judge(const ConvP& c) {
    return c.groups == 1 && c.stride == 1 && c.dil == 1 && c.pad_mode == MS_PAD_ZERO && !c.in_act && c.Cout == 1 && c.K == 3 &&
           c.pad == 1 && c.act == MS_ACT_NONE && c.Cin % 256 == 0 && c.Cin >= 256 && c.Cin <= 4096;
}

// tile: samples per workgroup unit (0: one unit per batch row)
bool table(const ConvP& c, const ms_conv1d_parts* parts, int tile, int chan, DParts* q) {
    if (!parts || parts->count < 1 || parts->count > NP) return false;
    q->count = parts->count;
    q->wg0[0] = 0;
    for (int i = 0; i < NP; ++i) {
        const bool on = i < parts->count;
        if (on && (parts->B[i] <= 0 || parts->Lin[i] <= 0)) return false;
        if (on && (long long)parts->B[i] * chan * parts->Lin[i] * 4 >= (1ll << 31)) return false;
        q->B[i] = on ? parts->B[i] : 0; q->L[i] = on ? parts->Lin[i] : 1;
        q->tiles[i] = on ? (tile ? ms_ceil_div(parts->Lin[i], tile) : 1) : 1;
        q->wg0[i + 1] = q->wg0[i] + q->B[i] * q->tiles[i];
        q->a[i] = q->b[i] = q->c[i] = nullptr; q->o[i] = nullptr;
    }
    return true;
}

constexpr int FIRST_WGS = 512;           // persistent workgroups of the first conv's weight gradient
constexpr int JUDGE_SLABS = 48;          // batch slabs of the judge conv's weight gradient (x C / 256 workgroups)

}  // namespace

bool msd_parts_applicable(const ConvP& c, const ms_conv1d_parts* parts, int which) {
    const char* e = getenv("MSYNTH_DTHIN");                 // tuning / test switch (0: the per-scale kernels, part by part)
    if (e && atoi(e) == 0) return false;
    if (!parts || parts->count < 1 || parts->count > NP) return false;
    if (is_first(c)) return true;
    if (is_judge(c)) {
        for (int i = 0; i < parts->count; ++i)
            if (parts->Lin[i] > JL) return false;
        return true;
    }
    (void)which;
    return false;
}

size_t msd_parts_bwd_weight_ws(const ConvP& c, const ms_conv1d_parts* parts) {
    (void)parts;
    if (is_first(c)) return (size_t)FIRST_WGS * (FC * FK + FC) * sizeof(float);
    return (size_t)JUDGE_SLABS * (c.Cin * 3 + 1) * sizeof(float);
}

int msd_parts_fwd(const ConvP& c, const ms_conv1d_parts* parts, const float* w, const float* bias, hipStream_t s) {
    DParts q;
    const bool first = is_first(c);
    if (!table(c, parts, first ? FT : 0, first ? FC : c.Cin, &q)) return MS_ERR_INVALID_ARG;
    for (int i = 0; i < q.count; ++i) {
        if (!parts->x[i] || !parts->y[i]) return MS_ERR_INVALID_ARG;
        if (!first && (((uintptr_t)parts->x[i]) & 15)) return MS_ERR_UNSUPPORTED;
        q.a[i] = parts->x[i]; q.o[i] = parts->y[i];
    }
    if (first) {
        ms_note_kernel(0, "k_dfirst_fwd");
        hipLaunchKernelGGL(k_dfirst_fwd, dim3(q.wg0[q.count]), dim3(256), 0, s, q, w, bias, c.slope);
    } else {
        ms_note_kernel(0, "k_djudge_fwd");
        hipLaunchKernelGGL(k_djudge_fwd, dim3(q.wg0[q.count]), dim3(256), 0, s, q, c.Cin, w, bias);
    }
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int msd_parts_bwd_data(const ConvP& c, const ms_conv1d_parts* parts, const float* w, hipStream_t s) {
    DParts q;
    const bool first = is_first(c);
    if (!table(c, parts, first ? 256 : 0, first ? FC : c.Cin, &q)) return MS_ERR_INVALID_ARG;
    for (int i = 0; i < q.count; ++i) {
        if (!parts->gy[i] || !parts->gx[i] || (first && !parts->y_act[i])) return MS_ERR_INVALID_ARG;
        if (!first && ((((uintptr_t)parts->gx[i]) & 15) || (parts->gx_add[i] && (((uintptr_t)parts->gx_add[i]) & 15))))
            return MS_ERR_UNSUPPORTED;
        q.a[i] = parts->gy[i]; q.b[i] = parts->y_act[i]; q.c[i] = parts->gx_add[i]; q.o[i] = parts->gx[i];
    }
    if (first) {
        ms_note_kernel(0, "k_dfirst_bwd_data");
        hipLaunchKernelGGL(k_dfirst_bwd_data, dim3(q.wg0[q.count]), dim3(256), 0, s, q, w, c.slope);
    } else {
        // workgroups: (part, batch row, chunk of 256 channels)
        const int chunks = c.Cin / 256;
        for (int i = 0; i < NP; ++i) q.wg0[i + 1] = q.wg0[i] + q.B[i] * chunks;
        ms_note_kernel(0, "k_djudge_bwd_data");
        hipLaunchKernelGGL(k_djudge_bwd_data, dim3(q.wg0[q.count]), dim3(256), 0, s, q, c.Cin, w);
    }
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int msd_parts_bwd_weight(const ConvP& c, const ms_conv1d_parts* parts, float* gw, float* gb, float beta, void* ws,
                         size_t ws_bytes, hipStream_t s) {
    DParts q;
    const bool first = is_first(c);
    if (!ws || ws_bytes < msd_parts_bwd_weight_ws(c, parts)) return MS_ERR_WORKSPACE;
    if (!table(c, parts, first ? FT : 0, first ? FC : c.Cin, &q)) return MS_ERR_INVALID_ARG;
    float* partial = (float*)ws;
    if (first) {
        for (int i = 0; i < q.count; ++i) {
            if (!parts->x[i] || !parts->gy[i] || !parts->y_act[i]) return MS_ERR_INVALID_ARG;
            q.a[i] = parts->x[i]; q.b[i] = parts->gy[i]; q.c[i] = parts->y_act[i];
        }
        const int units = q.wg0[q.count];
        const int grid = units < FIRST_WGS ? units : FIRST_WGS;
        ms_note_kernel(0, "k_dfirst_wgrad");
        hipLaunchKernelGGL(k_dfirst_wgrad, dim3(grid), dim3(256), 0, s, q, c.slope, partial);
        MS_CHECK_LAUNCH();
        const int nw = FC * FK, nb = FC;
        hipLaunchKernelGGL(k_dparts_reduce, dim3(ms_ceil_div(nw + nb, 4)), dim3(256), 0, s, partial, grid, nw + nb, nw, nb, gw, gb,
                           beta);
    } else {
        for (int i = 0; i < q.count; ++i) {
            if (!parts->x[i] || !parts->gy[i]) return MS_ERR_INVALID_ARG;
            if (((uintptr_t)parts->x[i]) & 15) return MS_ERR_UNSUPPORTED;
            q.a[i] = parts->x[i]; q.b[i] = parts->gy[i];
        }
        const int rows = q.wg0[q.count], chunks = c.Cin / 256;
        const int slabs = rows < JUDGE_SLABS ? rows : JUDGE_SLABS;
        ms_note_kernel(0, "k_djudge_wgrad");
        hipLaunchKernelGGL(k_djudge_wgrad, dim3(slabs * chunks), dim3(256), 0, s, q, c.Cin, slabs, partial);
        MS_CHECK_LAUNCH();
        const int nw = c.Cin * 3, nb = 1;
        hipLaunchKernelGGL(k_dparts_reduce, dim3(ms_ceil_div(nw + nb, 4)), dim3(256), 0, s, partial, slabs, nw + nb, nw, nb, gw, gb,
                           beta);
    }
    MS_CHECK_LAUNCH();
    return MS_OK;
}
