// Weight gradient of the stride-2 / kernel-4 / padding-1 transposed conv on SHORT rows (4 .. 32 input positions): the first
// line convolutions of the stage-1 generator (reference featuregenerator/upscale.py:85-91 through util/modules.py:
// HipConvTranspose2d), (rows, channels) = (128, 2048 -> 512), (256, 1024 -> 256), (512, 512 -> 128), (1024, 256 -> 128).
//
//   gw[ci][co][k] = sum_{b, l} x[b][ci][l] * g'[b][co][2 l + k - 1],      g' = gy * act'(y)
// Few positions, many channels: a GEMM  C[ci][(co, k)] = A[ci][pos] . B[(co, k)][pos]  with M = Cin, N = 4 Cout and the
// contraction over the positions pos = (b, l) -- both operands contraction-contiguous (an "NT" GEMM).  These shapes ran on
// the first-generation implicit-GEMM kernel at ~20 TFLOP/s (k_igemm_wgrad): it walks rows of >= 32 positions.
// Here: 128 x 128 output tiles, 8 waves (2 x 4, 64 rows x 32 columns each), 32 positions per step staged through LDS as
// three bf16 pieces [row][piece][32 positions] (208-byte rows: conflict-free 16-byte fragment reads); the loader reads
// x[b][ci][l..l+3] with the lanes running over ci and gy[b][co][..] with the lanes over co (both contiguous), forms the four
// tap operands of a gradient row in registers (tap k at position l is sample 2 l + k - 1: even / odd samples, shifted by one
// with zeros at the row ends), splits and stores 8-byte groups; the next step's global loads are in flight during the MFMAs;
// split-K over batch rows fills the chip, slabs are summed in slice order (msm_wgrad_reduce, deterministic).
// Arithmetic: exact 3-piece bf16 split, six products per multiply, fp32 accumulation.
#include "ms_common.h"
#include "conv_mfma.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 32;      // output tile, positions per step
constexpr int RS = 208;                          // LDS bytes per tile row: 3 pieces x 64 + 16
constexpr int NT = 512;

__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2 v = {a, b};
    const bf16x2 hi = __builtin_convertvector(v, bf16x2);
    const f32x2 r1 = v - __builtin_convertvector(hi, f32x2);
    const bf16x2 mi = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(mi, f32x2);
    const bf16x2 lo = __builtin_convertvector(r2, bf16x2);
    h = __builtin_bit_cast(unsigned, hi);
    m = __builtin_bit_cast(unsigned, mi);
    l = __builtin_bit_cast(unsigned, lo);
}

__device__ __forceinline__ void split_quad(const float (&e)[4], uint2 (&o)[3]) {
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair(e[0], e[1], h0, m0, l0);
    split_pair(e[2], e[3], h1, m1, l1);
    o[0] = make_uint2(h0, h1);
    o[1] = make_uint2(m0, m1);
    o[2] = make_uint2(l0, l1);
}

struct W2P {
    int B, CI, CO, W;         // batch rows, input channels (GEMM rows), gradient channels (GEMM columns / 4), positions per row
    int wsh;                  // log2(W)
    int nsteps, sps;          // steps of 32 positions in all (B * W / 32), steps per split-K slice
    int masked;
    float slope;
    long long zstride;        // floats per slab (CI * CO * 4)
};

__global__ __launch_bounds__(NT, 2) void k_wgrad_convt2_short(W2P p, const float* __restrict__ X, const float* __restrict__ GY,
                                                             const float* __restrict__ YA, float* __restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) unsigned char sA[BM * RS];
    __shared__ __attribute__((aligned(16))) unsigned char sB[BN * RS];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;                   // 2 x 4 waves: 64 rows x 32 columns each
    const int W = p.W, W2 = 2 * W;
    const int m0 = blockIdx.x * BM, co0 = blockIdx.y * (BN / 4);
    const int sbeg = blockIdx.z * p.sps;
    const int send = sbeg + p.sps < p.nsteps ? sbeg + p.sps : p.nsteps;
    const bool masked = p.masked != 0;
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 4u * (unsigned)(p.B * p.CI * W), 0x00020000);
    const unsigned g_bytes = 4u * (unsigned)(p.B * p.CO * W2);
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GY), 0, g_bytes, 0x00020000);
    const auto rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(masked ? YA : GY), 0, g_bytes, 0x00020000);

    // ---- A units (two per thread): (position group v of the step's 8, row ci): one float4 x[b][ci][l4 .. l4 + 3];
    //      lanes run over ci (contiguous blocks of W floats)
    unsigned a_goff[2];
    int a_lds[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + NT * k;
        const int ci = u & (BM - 1), v = u >> 7;                        // v in 0 .. 7
        const int pos = 4 * v, rl = pos >> p.wsh, l4 = pos & (W - 1);   // batch row within the step, first position
        a_goff[k] = 4u * (unsigned)((rl * p.CI + m0 + ci) * W + l4);    // + step * (32 / W) * CI * W * 4 (scalar)
        a_lds[k] = ci * RS + 8 * v;
    }
    // ---- B unit (threads 0 .. 255): (position group v, channel co): gradient samples 8 g - 1 .. 8 g + 8 of the row that
    //      holds positions 4 g .. 4 g + 3 (g = position group within the row); lanes run over co
    const bool b_on = tid < 256;
    const int b_co = tid & 31, b_v = (tid >> 5) & 7;
    const int b_pos = 4 * b_v, b_rl = b_pos >> p.wsh, b_l4 = b_pos & (W - 1);
    const unsigned b_goff = 4u * (unsigned)((b_rl * p.CO + co0 + b_co) * W2 + 2 * b_l4);        // sample 2 l4 of the row
    const bool b_first = b_l4 == 0, b_last = b_l4 + 4 == W;
    const int b_lds = (4 * b_co) * RS + 8 * b_v;

    f32x4 ra[2], rg[2], ry[2];
    float rg_m, rg_p, ry_m, ry_p;                                      // samples 2 l4 - 1 and 2 l4 + 8
    auto load = [&](int step, bool live) {
        const int rows_per_step = BK >> p.wsh;
        const int so_x = live ? 4 * step * rows_per_step * p.CI * W : 0;
        const int so_g = live ? 4 * step * rows_per_step * p.CO * W2 : 0;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            ra[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, a_goff[k], so_x, 0));
        if (b_on) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                rg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, b_goff + 16 * k, so_g, 0));
                ry[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, b_goff + 16 * k, so_g, 0));
            }
            // (the neighbours across the group: inside the row unless the group is its first / last)
            rg_m = b_first ? 0.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsG, b_goff - 4, so_g, 0));
            ry_m = b_first ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsY, b_goff - 4, so_g, 0));
            rg_p = b_last ? 0.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsG, b_goff + 32, so_g, 0));
            ry_p = b_last ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsY, b_goff + 32, so_g, 0));
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float e[4] = {ra[k][0], ra[k][1], ra[k][2], ra[k][3]};
            uint2 o3[3];
            split_quad(e, o3);
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<uint2*>(sA + a_lds[k] + pp * 64) = o3[pp];
        }
        if (b_on) {
            float s[10];                                               // samples 2 l4 - 1 .. 2 l4 + 8
            s[0] = rg_m; s[9] = rg_p;
#pragma unroll
            for (int i = 0; i < 4; ++i) { s[1 + i] = rg[0][i]; s[5 + i] = rg[1][i]; }
            if (masked) {
                s[0] = ry_m > 0.f ? s[0] : s[0] * p.slope;
                s[9] = ry_p > 0.f ? s[9] : s[9] * p.slope;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s[1 + i] = ry[0][i] > 0.f ? s[1 + i] : s[1 + i] * p.slope;
                    s[5 + i] = ry[1][i] > 0.f ? s[5 + i] : s[5 + i] * p.slope;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {                              // tap k at position l4 + j: sample 2 j + k - 1 -> s[2 j + k]
                const float e[4] = {s[k], s[2 + k], s[4 + k], s[6 + k]};
                uint2 o3[3];
                split_quad(e, o3);
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<uint2*>(sB + b_lds + k * RS + pp * 64) = o3[pp];
            }
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned char* fA = sA + (wm * 64 + l31) * RS + h * 16;
    const unsigned char* fB = sB + (wn * 32 + l31) * RS + h * 16;
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};

    if (sbeg < send) load(sbeg, true);
#pragma unroll 1
    for (int st = sbeg; st < send; ++st) {
        __syncthreads();                                               // every wave is done with the previous step's tiles
        stage();
        load(st + 1, st + 1 < send);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[2][3], b[3];
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) {
                b[pp] = *reinterpret_cast<const bf16x8*>(fB + pp * 64 + ks * 32);
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i][pp] = *reinterpret_cast<const bf16x8*>(fA + i * 32 * RS + pp * 64 + ks * 32);
            }
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[s6]], b[PB[s6]], acc[i], 0, 0, 0);
        }
    }

    // ---- slab: C[ci][n], n = 4 (co - co0) + k  ==  gw[ci][co][k] (row length 4 CO)
    float* out = slabs + (size_t)blockIdx.z * p.zstride;
    const int N = 4 * p.CO, n = 4 * co0 + wn * 32 + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = acc[i][4 * g + q];
                out[(size_t)(m0 + wm * 64 + i * 32 + 8 * g + 4 * h + q) * N + n] = v;
            }
}

int pick_slices(const ConvP& p, int nsteps) {
    const int tiles = (p.Cout / BM) * (p.Cin * 4 / BN);
    int ns = 512 / tiles;
    if (ns < 1) ns = 1;
    if (ns > 64) ns = 64;
    while (ns > 1 && nsteps / ns < 8) --ns;
    const size_t slab = (size_t)p.Cout * p.Cin * 4 * sizeof(float);
    while (ns > 1 && (size_t)ns * slab > ((size_t)64 << 20)) --ns;
    return ns < 1 ? 1 : ns;
}

}  // namespace

// p = mirrored conv of the transposed conv: Cin_T = p.Cout, Cout_T = p.Cin, Lin_T = p.Lout, Lout_T = p.Lin
bool mswt2s_applicable(const ConvP& p) {
    const char* e = getenv("MSYNTH_WGRADT2S");        // tuning / test switch (0: the generic weight-gradient kernels)
    if (e && atoi(e) == 0) return false;
    const int W = p.Lout;
    return p.stride == 2 && p.K == 4 && p.pad == 1 && p.dil == 1 && p.groups == 1 && p.in_act == MS_ACT_NONE &&
           (p.act == MS_ACT_NONE || p.act == MS_ACT_LRELU) && p.Lin == 2 * W && (W == 4 || W == 8 || W == 16 || W == 32) &&
           (p.B * W) % BK == 0 && p.Cout % BM == 0 && (p.Cin * 4) % BN == 0 && p.B * W >= 256 &&
           (long long)p.B * p.Cin * p.Lin * 4 < (1ll << 31) && (long long)p.B * p.Cout * W * 4 < (1ll << 31);
}

size_t mswt2s_ws(const ConvP& p) {
    return (size_t)pick_slices(p, p.B * p.Lout / BK) * (size_t)p.Cout * p.Cin * 4 * sizeof(float);
}

const char* mswt2s_name(const ConvP&) { return "k_wgrad_convt2_short"; }

int mswt2s_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act, float* gw, float beta, void* ws,
                      size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < mswt2s_ws(p) || (((uintptr_t)ws) & 15)) return MS_ERR_WORKSPACE;
    if (((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)(y_act ? y_act : gy))) & 15) != 0) return MS_ERR_UNSUPPORTED;
    if (p.act != MS_ACT_NONE && !y_act) return MS_ERR_INVALID_ARG;
    W2P q;
    q.B = p.B; q.CI = p.Cout; q.CO = p.Cin; q.W = p.Lout;
    q.wsh = 0;
    while ((1 << q.wsh) < q.W) ++q.wsh;
    q.nsteps = p.B * q.W / BK;
    const int ns = pick_slices(p, q.nsteps);
    q.sps = (q.nsteps + ns - 1) / ns;
    q.masked = p.act == MS_ACT_LRELU ? 1 : 0;
    q.slope = p.slope;
    q.zstride = (long long)p.Cout * p.Cin * 4;
    const int nz = (q.nsteps + q.sps - 1) / q.sps;
    float* partial = (float*)ws;
    ms_note_kernel(6, "k_wgrad_convt2_short");
    hipLaunchKernelGGL(k_wgrad_convt2_short, dim3(p.Cout / BM, p.Cin * 4 / BN, nz), dim3(NT), 0, s, q, x, gy, y_act, partial);
    MS_CHECK_LAUNCH();
    return msm_wgrad_reduce(partial, (size_t)q.zstride, nz, (size_t)q.zstride, 0, gw, nullptr, beta, s);
}
