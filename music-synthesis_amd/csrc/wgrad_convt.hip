// Weight gradient of ConvTranspose1d with stride 8 / kernel 16 / padding 4 (the generator's first two upsampling
// layers, reference generator/full.py) on the bf16 matrix pipe with fp32-exact operands (three bf16 pieces, six
// partial products, fp32 accumulate: conv_rows3.hip).
//
//   gw[ci, co, k] = sum_{b, q} x[b, ci, q] * gp[b, co, 8 q + k - 4],      gp = gy * act'(y_act),  k = 0 .. 15
//
// GEMM per output channel co: M = ci, N = the 16 taps, contraction over (b, q).  v_mfma_f32_16x16x32_bf16 contracts
// 32 consecutive input positions q of one batch row per step:
//   A operand: x[ci][q0 + 8 kg .. +7]: one 16-byte read of the image [octet][ci][8 q] (as wgrad_k5.hip);
//   B operand: B[q][k] = gp[co][8 q + k - 4] is a Toeplitz view of the LINEAR gradient row whose rows sit 16 bytes
//     (8 bf16) apart -- gfx950's transposing LDS read (ds_read_b64_tr_b16: 16 lanes supply 4 row addresses x 4 column
//     quads, receive column-major data) delivers it from a plain bf16 copy of the row, as in gconv_split.hip's weight
//     gradient; no phase-split image.
// Workgroup = 128 ci x 16 co (8 waves; wave w owns 16 ci and all 16 co: 16 accumulator tiles), the (b, q) range is cut into
// `nsplit` slabs (deterministic reduce: msm_wgrad_reduce).  Staging as in wgrad_k5.hip: global loads of step s+1 in
// flight during the MFMAs of step s, two LDS images, one barrier per step.  The LeakyReLU in front of the transposed
// conv (in_act) is applied to x on its way into LDS.  The bias gradient is a separate channel sum (api.hip).
#include "ms_common.h"
#include "conv_mfma.h"
#include <stdint.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;

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

struct W8P {
    int B, CI, CO, L, spr, nsteps, sps, act, in_act;      // spr = L / 32 steps per batch row
    float slope;
    size_t stride;
};

constexpr int TS = 8, TK = 16;                 // stride, taps
constexpr int TCI = 128, TCO = 16;             // workgroup tile: 8 waves x 16 ci, every wave all 16 co
constexpr int NT = 4 * TCI;                    // threads: one (ci, octet) input item each
constexpr int QS = 32;                         // input positions per step
constexpr int A_PIECE = 4 * TCI * 16;          // [octet slot][ci][8 q] bf16
constexpr int GSPAN = QS * TS + TK;            // 272 gradient samples per (co, step)
constexpr int GQ = GSPAN / 4;                  // 68 quads
constexpr int GROW = GSPAN * 2 + 16;           // bytes per LDS gradient row (16-byte multiple)
constexpr int B_PIECE = TCO * GROW;
constexpr int IMG = 3 * (A_PIECE + B_PIECE);
constexpr int NGI = (TCO * GQ + NT - 1) / NT;  // gradient quads per thread and step

__global__ __launch_bounds__(NT) void k_wgrad_convt8_split(W8P p, const float* __restrict__ x,
                                                              const float* __restrict__ gy,
                                                              const float* __restrict__ y_act,
                                                              float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c0 = blockIdx.x * TCI, o0 = blockIdx.y * TCO, z = blockIdx.z;
    const int kind = y_act ? p.act : MS_ACT_NONE;
    const int r64 = tid % TCI, oc = tid / TCI;            // input item: channel r64 of octet slot oc
    const int Lg = p.L * TS;
    constexpr unsigned OOB = 0xF0000000u;
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x80000000u, 0x00020000);
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, 0x80000000u, 0x00020000);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y_act ? y_act : gy), 0, 0x80000000u, 0x00020000);

    // gradient items of this thread: quad Q of row co
    int g_co[NGI], g_q[NGI];
#pragma unroll
    for (int i = 0; i < NGI; ++i) {
        const int idx = tid + NT * i;
        const int co = idx / GQ;
        g_co[i] = co < TCO ? co : -1;
        g_q[i] = idx - co * GQ;
    }
    f32x4 xv[2], gv[NGI], ga[NGI];
    auto gload = [&](int step) {
        const int b = step / p.spr, q0 = (step - b * p.spr) * QS;
        const unsigned xo = (unsigned)((b * p.CI + c0 + r64) * p.L + q0 + 8 * oc);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            xv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (xo + 4u * j) * 4u, 0, 0));
        const int g0 = TS * q0 - TK / 4;                       // first staged gradient sample (multiple of 4)
#pragma unroll
        for (int i = 0; i < NGI; ++i) {
            const int g = g0 + 4 * g_q[i];
            const bool ok = g_co[i] >= 0 && g >= 0 && g < Lg;
            const unsigned vo = ok ? (unsigned)((b * p.CO + o0 + g_co[i]) * Lg + g) * 4u : OOB;
            gv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, vo, 0, 0));
            ga[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, 0, 0));
        }
    };
    auto stage = [&](unsigned char* img) {
        float e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = xv[k >> 2][k & 3];
            e[k] = p.in_act ? (v > 0.f ? v : v * p.slope) : v;
        }
        u32x4 h, m, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned hh, mm, ll;
            split_pair(e[2 * q], e[2 * q + 1], hh, mm, ll);
            h[q] = hh; m[q] = mm; l[q] = ll;
        }
        unsigned char* a = img + (oc * TCI + r64) * 16;
        *reinterpret_cast<u32x4*>(a) = h;
        *reinterpret_cast<u32x4*>(a + A_PIECE) = m;
        *reinterpret_cast<u32x4*>(a + 2 * A_PIECE) = l;
#pragma unroll
        for (int i = 0; i < NGI; ++i) {
            if (g_co[i] < 0) continue;
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = ms_act_grad(gv[i][k], ga[i][k], kind, p.slope);
            unsigned h0, m0, l0, h1, m1, l1;
            split_pair(d[0], d[1], h0, m0, l0);
            split_pair(d[2], d[3], h1, m1, l1);
            unsigned char* g = img + 3 * A_PIECE + g_co[i] * GROW + g_q[i] * 8;
            *reinterpret_cast<u32x2*>(g) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2*>(g + B_PIECE) = (u32x2){m0, m1};
            *reinterpret_cast<u32x2*>(g + 2 * B_PIECE) = (u32x2){l0, l1};
        }
    };

    f32x4 acc[TCO];
#pragma unroll
    for (int c = 0; c < TCO; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int n = lane & 15, kg = lane >> 4;
    const int a_rd = (kg * TCI + 16 * wid + n) * 16;
    // transposing read: lane 4 qq + pp of a 16-lane group supplies row (8 kg + 4 rd + qq), column quad pp
    const int b_rd = 3 * A_PIECE + 128 * kg + 16 * ((lane & 15) >> 2) + 8 * (lane & 3);
    auto compute = [&](const unsigned char* img) {
        bf16x8 A[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) A[pc] = *reinterpret_cast<const bf16x8*>(img + a_rd + pc * A_PIECE);
        const unsigned bb = (unsigned)(uintptr_t)img + b_rd;
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int c4 = 0; c4 < TCO; c4 += 4) {             // four output channels at a time: independent MFMA chains
            bf16x8 Bf[4][3];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    const unsigned ad = bb + pc * B_PIECE + (c4 + c) * GROW;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(ad));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(ad + 64));
                    Bf[c][pc] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[c4 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[PA[i]], Bf[c][PB[i]], acc[c4 + c], 0, 0, 0);
        }
    };

    const int s_beg = z * p.sps, s_end = min(p.nsteps, s_beg + p.sps);
    unsigned char* img0 = smem8;
    unsigned char* img1 = smem8 + IMG;
    if (s_beg < s_end) {
        gload(s_beg);
        stage(img0);
    }
    __syncthreads();
    for (int s = s_beg; s < s_end; ++s) {
        const bool more = s + 1 < s_end;
        unsigned char* cur = ((s - s_beg) & 1) ? img1 : img0;
        unsigned char* nxt = ((s - s_beg) & 1) ? img0 : img1;
        if (more) gload(s + 1);
        compute(cur);
        if (more) stage(nxt);
        __syncthreads();
    }

    // ---- slab: D[ci][tap] per output channel: lane (tap = lane & 15, ci quad = lane >> 4), rows r
    float* part = partial + (size_t)z * p.stride;
#pragma unroll
    for (int c = 0; c < TCO; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = c0 + 16 * wid + 4 * kg + r;
            part[((size_t)ci * p.CO + o0 + c) * TK + n] = acc[c][r];
        }
}

int pick_nsplit8(const ConvP& p, int nsteps) {
    const int tiles = (p.Cout / TCI) * (p.Cin / TCO);
    int ns = (256 + tiles - 1) / tiles;                     // one 8-wave workgroup per CU
    if (ns > nsteps / 2) ns = nsteps / 2;
    const size_t slab = (size_t)p.Cout * p.Cin * TK * sizeof(float);
    while (ns > 1 && (size_t)ns * slab > ((size_t)64 << 20)) --ns;
    return ns < 1 ? 1 : ns;
}

}  // namespace

// p = mirrored conv of the transposed conv: Cin_T = p.Cout, Cout_T = p.Cin, Lin_T = p.Lout, Lout_T = p.Lin
bool mswt8_applicable(const ConvP& p) {
    const char* e = getenv("MSYNTH_WGRADT8");         // tuning / test switch (0: fp32-MFMA phase-split kernel)
    if (e && atoi(e) == 0) return false;
    return p.stride == TS && p.K == TK && p.pad == TK / 4 && p.dil == 1 && p.groups == 1 && p.Lin == p.Lout * TS &&
           p.Lout % QS == 0 && p.Cout % TCI == 0 && p.Cin % TCO == 0 &&
           (long long)p.B * p.Cin * p.Lin * 4 < (1ll << 31) && (long long)p.B * p.Cout * p.Lout * 4 < (1ll << 31);
}

size_t mswt8_ws(const ConvP& p) {
    const int nsteps = p.B * (p.Lout / QS);
    return (size_t)pick_nsplit8(p, nsteps) * (size_t)p.Cout * p.Cin * TK * sizeof(float);
}

const char* mswt8_name(const ConvP&) { return "k_wgrad_convt8_split"; }

int mswt8_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act, float* gw, float beta,
                     void* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < mswt8_ws(p) || (((uintptr_t)ws) & 15)) return MS_ERR_WORKSPACE;
    if (((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)(y_act ? y_act : gy))) & 15) != 0) return MS_ERR_UNSUPPORTED;
    W8P q;
    q.B = p.B; q.CI = p.Cout; q.CO = p.Cin; q.L = p.Lout; q.spr = p.Lout / QS;
    q.nsteps = p.B * q.spr;
    const int ns = pick_nsplit8(p, q.nsteps);
    q.sps = (q.nsteps + ns - 1) / ns;
    q.act = p.act; q.in_act = p.in_act ? 1 : 0; q.slope = p.slope;
    q.stride = (size_t)p.Cout * p.Cin * TK;
    const int nz = (q.nsteps + q.sps - 1) / q.sps;
    static unsigned long long attr_set = 0;
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_convt8_split),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG);
        ms_done_on_device(attr_set);
    }
    float* partial = (float*)ws;
    ms_note_kernel(6, "k_wgrad_convt8_split");
    hipLaunchKernelGGL(k_wgrad_convt8_split, dim3(p.Cout / TCI, p.Cin / TCO, nz), dim3(NT), 2 * IMG, s, q, x, gy,
                       y_act, partial);
    MS_CHECK_LAUNCH();
    return msm_wgrad_reduce(partial, q.stride, nz, q.stride, 0, gw, nullptr, beta, s);
}
