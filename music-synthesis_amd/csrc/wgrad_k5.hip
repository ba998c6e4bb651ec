// Weight gradient of the dense k5 conv with many channels and short rows (the discriminator's 1024 -> 1024 k5 layer
// at L = 32 / 17 / 9, reference discriminator/full.py:19) on the bf16 matrix pipe with fp32-exact operands
// (x = x1 + x2 + x3 in bf16 pieces, the six partial products with i + j <= 4 accumulated in fp32: conv_rows3.hip).
//
//   gw[co, ci, k] = sum_{b, t} gp[b, co, t] * x[b, ci, t + k - 2],      gp = gy * act'(y_act)
//
// GEMM: M = co, N = (ci, k), contraction over (b, t).  v_mfma_f32_16x16x32_bf16 contracts 4 lane groups x 8
// elements: a lane group is one OCTET of 8 consecutive outputs of one batch row (rows are cut into ceil(L / 8)
// octets, zero-padded), so a step contracts 4 octets -- from one row (L = 32) or from consecutive rows.
//   A operand: gp[co][t0 .. t0+7]: one 16-byte LDS read of the image [octet][co][8 t].
//   B operand of tap k: x[ci][t0 + k - 2 .. t0 + k + 5] -- the same 12 staged samples shifted by k - 2 elements.
//     The lane reads the 6 dwords D0..D5 (samples t0-2 .. t0+9) once per piece and funnels the five operands in
//     registers: taps 0 / 2 / 4 are the dword windows D0-3 / D1-4 / D2-5, taps 1 / 3 are v_alignbit by 16 bits of
//     neighbouring dwords -- ~12 vector instructions per piece for 5 x 4 x 6 = 120 MFMAs.
// Workgroup = 64 co x 64 ci (wave w owns 16 ci, all four share the gp image), 20 accumulator tiles per wave held
// across all steps; the batch is cut into `nsplit` slabs (deterministic reduce: msm_wgrad_reduce).  Staging: every
// thread owns one (co, octet) and one (ci, octet) item per step -- global loads of step s+1 are in flight during the
// MFMAs of step s, two LDS images, one barrier per step.
#include "ms_common.h"
#include "conv_mfma.h"
#include <stdint.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

// The contraction runs over OCTETS of 8 consecutive outputs.  The octets of up to MS_CONV_PARTS_MAX inputs ("parts": the
// shared discriminator's three scales, reference discriminator/melgan.py:13-27) form one index space -- part i owns octets
// [o0[i], o0[i + 1]), NO[i] per batch row -- so that the weight gradient of the shared layer over all scales is ONE launch
// and one slab reduction.
struct W5P {
    int M, C, nsteps, sps, act;               // nsteps = ceil(total octets / 4), sps steps per slab
    float slope;
    size_t stride;                            // floats per slab
    int count, o0[MS_CONV_PARTS_MAX + 1];
    int B[MS_CONV_PARTS_MAX], L[MS_CONV_PARTS_MAX], NO[MS_CONV_PARTS_MAX];
    const float* x[MS_CONV_PARTS_MAX];
    const float* gy[MS_CONV_PARTS_MAX];
    const float* ya[MS_CONV_PARTS_MAX];       // saved outputs (activation derivative) or nullptr
};

constexpr int TCO = 64, TCI = 64;
constexpr int A_PIECE = 4 * TCO * 16;        // [octet slot][co][8 t] bf16
constexpr int B_PIECE = 4 * TCI * 32;        // [octet slot][ci][16 samples bf16: 12 used, halves swizzled by slot]
constexpr int IMG = 3 * (A_PIECE + B_PIECE);
constexpr int EPI_FLOATS = 32 * TCI * 5;     // half a tile staged for the coalesced slab write
static_assert(2 * IMG >= EPI_FLOATS * 4, "epilogue staging fits the images");

// VEC: one part whose rows are 16-byte shaped (aligned vectors, all-or-nothing); otherwise rows of any length through
// unaligned 16- / 8-byte loads whose out-of-row samples are cleared (the conditions are wave-uniform: a wave stages one octet
// slot per step)
template <bool VEC>
__global__ __launch_bounds__(256, 2) void k_wgrad_k5_split(W5P p, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem5[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m0 = blockIdx.x * TCO, c0 = blockIdx.y * TCI, z = blockIdx.z;
    const int kind = p.ya[0] ? p.act : MS_ACT_NONE;
    const int r64 = tid & 63, oc = __builtin_amdgcn_readfirstlane(tid >> 6);   // staging item: row (co / ci) r64 of octet slot oc
    constexpr unsigned OOB = 0xF0000000u;

    float gv[8], ga[8], xv[12];
    auto gload = [&](int step) {
        const int o = 4 * step + oc;
        // the octet's part (compile-time indices into the by-value tables: see conv5_img.hip)
        int ob = 0, Bp = p.B[0], L = p.L[0], NO = p.NO[0];
        const float* xp = p.x[0];
        const float* gp = p.gy[0];
        const float* ap = p.ya[0];
#pragma unroll
        for (int k = 1; k < MS_CONV_PARTS_MAX; ++k)
            if (k < p.count && o >= p.o0[k]) {
                ob = p.o0[k]; Bp = p.B[k]; L = p.L[k]; NO = p.NO[k];
                xp = p.x[k]; gp = p.gy[k]; ap = p.ya[k];
            }
        // (true sizes: the unaligned loads across the end of a tensor's last row read 0.0 behind it)
        const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xp), 0, 4u * (unsigned)(Bp * p.C * L), 0x00020000);
        const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gp), 0, 4u * (unsigned)(Bp * p.M * L), 0x00020000);
        const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ap ? ap : gp), 0, 4u * (unsigned)(Bp * p.M * L), 0x00020000);
        const int ol = o - ob;
        const int b = ol / NO, t0 = (ol - b * NO) * 8;
        const bool ov = b < Bp;
        const unsigned grow = (unsigned)((b * p.M + m0 + r64) * L);
        const unsigned xrow = (unsigned)((b * p.C + c0 + r64) * L);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vo = (ov && t0 + 4 * j < L) ? (grow + (unsigned)(t0 + 4 * j)) * 4u : OOB;
            const f32x4 g4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, vo, 0, 0));
            const f32x4 a4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, 0, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = VEC || t0 + 4 * j + e < L;
                gv[4 * j + e] = in ? g4[e] : 0.f; ga[4 * j + e] = in ? a4[e] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int t = t0 - 2 + 2 * j;                  // (even: a pair lies wholly in front of the row or starts inside it)
            const unsigned vo = (ov && t >= 0 && t < L) ? (xrow + (unsigned)t) * 4u : OOB;
            const f32x2 v2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsX, vo, 0, 0));
            xv[2 * j] = v2[0];
            xv[2 * j + 1] = (VEC || t + 1 < L) ? v2[1] : 0.f;
        }
    };
    float bsum = 0.f;
    auto stage = [&](unsigned char* img) {
        // gradient octet: activation derivative, split, 16 bytes per piece
        float e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { e[k] = ms_act_grad(gv[k], ga[k], kind, p.slope); bsum += e[k]; }
        u32x4 h, m, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned hh, mm, ll;
            split_pair(e[2 * q], e[2 * q + 1], hh, mm, ll);
            h[q] = hh; m[q] = mm; l[q] = ll;
        }
        unsigned char* a = img + (oc * TCO + r64) * 16;
        *reinterpret_cast<u32x4*>(a) = h;
        *reinterpret_cast<u32x4*>(a + A_PIECE) = m;
        *reinterpret_cast<u32x4*>(a + 2 * A_PIECE) = l;
        // input samples t0-2 .. t0+9: dwords D0..D5, the two 16-byte halves swapped for odd slots (bank spread)
        unsigned dh[6], dm[6], dl[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) split_pair(xv[2 * q], xv[2 * q + 1], dh[q], dm[q], dl[q]);
        unsigned char* bq = img + 3 * A_PIECE + (oc * TCI + r64) * 32;
        const int h0 = (oc & 1) * 16, h1 = 16 - h0;
        *reinterpret_cast<u32x4*>(bq + h0) = (u32x4){dh[0], dh[1], dh[2], dh[3]};
        *reinterpret_cast<u32x2*>(bq + h1) = (u32x2){dh[4], dh[5]};
        *reinterpret_cast<u32x4*>(bq + B_PIECE + h0) = (u32x4){dm[0], dm[1], dm[2], dm[3]};
        *reinterpret_cast<u32x2*>(bq + B_PIECE + h1) = (u32x2){dm[4], dm[5]};
        *reinterpret_cast<u32x4*>(bq + 2 * B_PIECE + h0) = (u32x4){dl[0], dl[1], dl[2], dl[3]};
        *reinterpret_cast<u32x2*>(bq + 2 * B_PIECE + h1) = (u32x2){dl[4], dl[5]};
    };

    f32x4 acc[4][5];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[mt][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int n = lane & 15, kg = lane >> 4;
    const int a_rd = (kg * TCO + n) * 16;                                  // + mt * 256 + piece
    const int b_rd = 3 * A_PIECE + (kg * TCI + 16 * wid + n) * 32;         // + piece
    const int bh0 = (kg & 1) * 16, bh1 = 16 - bh0;
    auto compute = [&](const unsigned char* img) {
        bf16x8 A[4][3];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                A[mt][pc] = *reinterpret_cast<const bf16x8*>(img + a_rd + mt * 256 + pc * A_PIECE);
        unsigned D[3][6];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            const u32x4 lo = *reinterpret_cast<const u32x4*>(img + b_rd + pc * B_PIECE + bh0);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(img + b_rd + pc * B_PIECE + bh1);
            D[pc][0] = lo[0]; D[pc][1] = lo[1]; D[pc][2] = lo[2]; D[pc][3] = lo[3]; D[pc][4] = hi[0]; D[pc][5] = hi[1];
        }
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            bf16x8 Bo[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                u32x4 v;
                if ((k & 1) == 0) {
                    v = (u32x4){D[pc][k / 2], D[pc][k / 2 + 1], D[pc][k / 2 + 2], D[pc][k / 2 + 3]};
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        v[q] = __builtin_amdgcn_alignbit(D[pc][k / 2 + q + 1], D[pc][k / 2 + q], 16);
                }
                Bo[pc] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[mt][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[mt][PA[i]], Bo[PB[i]], acc[mt][k], 0, 0, 0);
        }
    };

    const int s_beg = z * p.sps, s_end = min(p.nsteps, s_beg + p.sps);
    unsigned char* img0 = smem5;
    unsigned char* img1 = smem5 + IMG;
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

    // ---- slab: the tile is staged through LDS in two halves of 32 co so that rows of 64 ci x 5 taps (1280
    // contiguous bytes of gw) leave as 16-byte vectors
    float* part = partial + (size_t)z * p.stride;
    float* ts = reinterpret_cast<float*>(smem5);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ml = 0; ml < 2; ++ml)
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ts[((ml * 16 + 4 * kg + r) * TCI + 16 * wid + n) * 5 + k] = acc[2 * half + ml][k][r];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EPI_FLOATS / 4 / 256; ++q) {
            const int idx = tid + 256 * q;                 // 16-byte vector of the half tile: row = idx / 80
            const int row = idx / 80, c4 = idx - row * 80;
            const f32x4 v = *reinterpret_cast<const f32x4*>(ts + row * 320 + 4 * c4);
            *reinterpret_cast<f32x4*>(part + ((size_t)(m0 + 32 * half + row) * p.C + c0) * 5 + 4 * c4) = v;
        }
        __syncthreads();
    }
    // bias gradient (ci tile 0 only): the four octet slots of a co sit in the four waves
    if (blockIdx.y == 0) {
        ts[oc * TCO + r64] = bsum;
        __syncthreads();
        if (tid < TCO)
            part[(size_t)p.M * p.C * 5 + m0 + tid] = (ts[tid] + ts[TCO + tid]) + (ts[2 * TCO + tid] + ts[3 * TCO + tid]);
    }
}

// ================================================================================================================
// r05: the same contraction on PRE-SPLIT operands (block-scaled two-piece fp16, three products per multiply).
// Both operands are data here, and every element is used by 64 x 5 (gradient) / 64 (input) output tiles: splitting them in the
// loader made the kernel spend more vector work than matrix work (the loader of k_wgrad_k5_split: 20 loads, 8 derivative
// selects, 30 three-piece splits per thread and step for 120 MFMAs per wave).  Here two small passes run first:
//   k_w5_maxima   largest magnitude of x and of gy per 16-channel chunk over ALL batch rows and parts (the raw gradient bounds
//                 the masked one) -> one power-of-two scale per chunk: the contraction runs over (batch row, sample), so a
//                 scale may depend on the channel but not on the batch row
//   k_w5_split    gp = gy * lrelu'(y) and x, scaled, as fp16 piece planes in the order the loader wants them: rows padded to
//                 whole octets ([row][co][8 NO] / [row][ci][8 NO + 8], the input plane shifted by the conv's left padding of
//                 2), plus the per-row bias sums
// and the contraction kernel stages an octet with three 16-byte and one 8-byte copy per piece -- no vector arithmetic.
// Accuracy: an element within 2^16 of its chunk's largest magnitude over the batch keeps 22 significand bits, smaller ones an
// absolute error below 2^-39 of that maximum: an fp32 FMA chain's accuracy for the sums (tests/test_gpu_parts.py: float64 2e-6).
struct W5Q {
    int M, C, nsteps, sps;
    size_t stride;
    int count, o0[MS_CONV_PARTS_MAX + 1], row0[MS_CONV_PARTS_MAX + 1];
    int B[MS_CONV_PARTS_MAX], L[MS_CONV_PARTS_MAX], NO[MS_CONV_PARTS_MAX];
    const float* x[MS_CONV_PARTS_MAX];
    const float* gy[MS_CONV_PARTS_MAX];
    const float* ya[MS_CONV_PARTS_MAX];
    unsigned short* gh[MS_CONV_PARTS_MAX];     // gradient planes (high, low): [row][co][8 NO] fp16
    unsigned short* gl[MS_CONV_PARTS_MAX];
    unsigned short* xh[MS_CONV_PARTS_MAX];     // input planes: [row][ci][8 NO + 8], sample t at element t + 2
    unsigned short* xl[MS_CONV_PARTS_MAX];
    unsigned* mx;                              // [C / 16 + M / 16] chunk maxima (float bits): inputs, then gradients
    float* tb;                                 // [rows][M] per-row bias sums
    float slope;
    int act;
};

struct W5Part { int B, L, NO, row0, o0; const float *x, *gy, *ya; unsigned short *gh, *gl, *xh, *xl; };
__device__ __forceinline__ W5Part w5_part(const W5Q& q, int key, bool by_octet) {
    W5Part p{q.B[0], q.L[0], q.NO[0], 0, 0, q.x[0], q.gy[0], q.ya[0], q.gh[0], q.gl[0], q.xh[0], q.xl[0]};
#pragma unroll
    for (int k = 1; k < MS_CONV_PARTS_MAX; ++k)
        if (k < q.count && key >= (by_octet ? q.o0[k] : q.row0[k]))
            p = W5Part{q.B[k], q.L[k], q.NO[k], q.row0[k], q.o0[k], q.x[k], q.gy[k], q.ya[k], q.gh[k], q.gl[k], q.xh[k], q.xl[k]};
    return p;
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// S = 2^k with m S in [2^12, 2^13) (room for the pieces of every batch row), and 1 / S; 1 for zero / non-finite maxima
__device__ __forceinline__ void chunk_scale(float m, float& S, float& invS) {
    const unsigned eb = (__builtin_bit_cast(unsigned, m) >> 23) & 0xFFu;
    const bool ok = eb >= 16u && eb <= 250u;
    S = ok ? __builtin_bit_cast(float, (266u - eb) << 23) : 1.f;
    invS = ok ? __builtin_bit_cast(float, (eb - 12u) << 23) : 1.f;
}

// Pre-pass workgroups = (row of a part, block of 128 channels); blocks [0, C / 128) belong to the input, the rest to the gradient.
constexpr int WB_CH = 128;

// per-chunk maxima of the block's 8 chunks -> pmx[row][chunk] (inputs' chunks first, then the gradients')
__global__ __launch_bounds__(256) void k_w5_maxima(W5Q q, float* __restrict__ pmx) {
    __shared__ unsigned mx[WB_CH / 16];
    const W5Part p = w5_part(q, blockIdx.x, false);
    const int b = (int)blockIdx.x - p.row0, tid = threadIdx.x;
    const int nbx = q.C / WB_CH, ncx = q.C / 16, ncg = q.M / 16;
    const bool isx = (int)blockIdx.y < nbx;
    const int blk = isx ? blockIdx.y : blockIdx.y - nbx;
    if (tid < WB_CH / 16) mx[tid] = 0u;
    __syncthreads();
    // the block's 128 * L floats are contiguous and start 16-byte aligned
    const float* src = (isx ? p.x + (size_t)b * q.C * p.L : p.gy + (size_t)b * q.M * p.L) + (size_t)blk * WB_CH * p.L;
    const int cv = 4 * p.L;                                   // 16-byte vectors per chunk
    for (int v = tid; v < WB_CH * p.L / 4; v += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + 4 * v);
        atomicMax(&mx[v / cv], __builtin_bit_cast(unsigned, fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3])))));
    }
    __syncthreads();
    if (tid < WB_CH / 16)
        pmx[(size_t)blockIdx.x * (ncx + ncg) + (isx ? 0 : ncx) + blk * (WB_CH / 16) + tid] = __builtin_bit_cast(float, mx[tid]);
}

__global__ __launch_bounds__(256) void k_w5_split(W5Q q, const float* __restrict__ pmx) {
    __shared__ float cm[WB_CH / 16];
    const W5Part p = w5_part(q, blockIdx.x, false);
    const int b = (int)blockIdx.x - p.row0, tid = threadIdx.x, L = p.L, NO = p.NO;
    const int row = blockIdx.x, rows = q.row0[q.count];
    const int nbx = q.C / WB_CH, ncx = q.C / 16, ncg = q.M / 16, nct = ncx + ncg;
    const bool isx = (int)blockIdx.y < nbx;
    const int blk = isx ? blockIdx.y : blockIdx.y - nbx;
    const int chunk0 = (isx ? 0 : ncx) + blk * (WB_CH / 16);          // first chunk of this block in the tables
    // the chunks' largest magnitudes over ALL rows: 8 chunks x 32 lanes, fixed-order maximum
    {
        const int c = tid >> 5, l32 = tid & 31;
        float m = 0.f;
        for (int r = l32; r < rows; r += 32) m = fmaxf(m, pmx[(size_t)r * nct + chunk0 + c]);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (l32 == 0) {
            cm[c] = m;
            if (row == 0) q.mx[chunk0 + c] = __builtin_bit_cast(unsigned, m);      // for the contraction kernel's epilogue
        }
    }
    __syncthreads();
    if (!isx) {
        // ---- gradient planes + bias row sums.  items = (channel, octet slot): NOP = NO rounded up to a power of two slots per
        // channel, so that a channel's octets are NOP consecutive lanes of ONE wave and the bias row sum is a fixed-order shuffle tree
        const bool masked = p.ya != nullptr && q.act == MS_ACT_LRELU;
        const float* gr = p.gy + ((size_t)b * q.M + (size_t)blk * WB_CH) * L;
        const float* ar = (masked ? p.ya : p.gy) + ((size_t)b * q.M + (size_t)blk * WB_CH) * L;
        int NOP = 1;
        while (NOP < NO) NOP <<= 1;
        for (int i = tid; i < WB_CH * NOP; i += 256) {
            const int cl = i / NOP, oc = i - cl * NOP, co = blk * WB_CH + cl;
            const bool live = oc < NO;
            float S, invS;
            chunk_scale(cm[cl >> 4], S, invS);
            unsigned hh[4], ll[4];
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float e[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int t = 8 * oc + 2 * k + j;
                    const bool in = live && t < L;
                    float v = in ? gr[(size_t)cl * L + t] : 0.f;
                    if (masked && in) v = ar[(size_t)cl * L + t] > 0.f ? v : v * q.slope;
                    e[j] = v;
                    sum += v;
                }
                const f32x2 s2 = {e[0] * S, e[1] * S};
                const f16x2 hi = __builtin_convertvector(s2, f16x2);
                const f16x2 lo = __builtin_convertvector(s2 - __builtin_convertvector(hi, f32x2), f16x2);
                hh[k] = __builtin_bit_cast(unsigned, hi); ll[k] = __builtin_bit_cast(unsigned, lo);
            }
            if (live) {
                const size_t o = ((size_t)b * q.M + co) * (8 * NO) + 8 * oc;
                *reinterpret_cast<u32x4*>(p.gh + o) = (u32x4){hh[0], hh[1], hh[2], hh[3]};
                *reinterpret_cast<u32x4*>(p.gl + o) = (u32x4){ll[0], ll[1], ll[2], ll[3]};
            }
            for (int d = 1; d < NOP; d <<= 1) sum += __shfl_xor(sum, d, 64);
            if (oc == 0) q.tb[(size_t)row * q.M + co] = sum;
        }
        return;
    }
    // ---- input planes: vector j of a row covers samples 8 j - 2 .. 8 j + 5 (element t + 2 holds sample t)
    const float* xr = p.x + ((size_t)b * q.C + (size_t)blk * WB_CH) * L;
    const int NV = NO + 1;
    for (int i = tid; i < WB_CH * NV; i += 256) {
        const int cl = i / NV, j = i - cl * NV, ci = blk * WB_CH + cl;
        float S, invS;
        chunk_scale(cm[cl >> 4], S, invS);
        unsigned hh[4], ll[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float e[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int t = 8 * j - 2 + 2 * k + jj;
                e[jj] = (t >= 0 && t < L) ? xr[(size_t)cl * L + t] : 0.f;
            }
            const f32x2 s2 = {e[0] * S, e[1] * S};
            const f16x2 hi = __builtin_convertvector(s2, f16x2);
            const f16x2 lo = __builtin_convertvector(s2 - __builtin_convertvector(hi, f32x2), f16x2);
            hh[k] = __builtin_bit_cast(unsigned, hi); ll[k] = __builtin_bit_cast(unsigned, lo);
        }
        const size_t o = ((size_t)b * q.C + ci) * (8 * NO + 8) + 8 * j;
        *reinterpret_cast<u32x4*>(p.xh + o) = (u32x4){hh[0], hh[1], hh[2], hh[3]};
        *reinterpret_cast<u32x4*>(p.xl + o) = (u32x4){ll[0], ll[1], ll[2], ll[3]};
    }
}

constexpr int A2_PIECE = 4 * TCO * 16;       // [octet slot][co][8 t] fp16
constexpr int B2_PIECE = 4 * TCI * 32;       // [octet slot][ci][16 samples fp16: 12 used, halves swizzled by slot]
constexpr int IMG2 = 2 * (A2_PIECE + B2_PIECE);
static_assert(2 * IMG2 >= EPI_FLOATS * 4, "epilogue staging fits the images");

__global__ __launch_bounds__(256, 2) void k_wgrad_k5_pre(W5Q p, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem5[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m0 = blockIdx.x * TCO, c0 = blockIdx.y * TCI, z = blockIdx.z;
    const int r64 = tid & 63, oc = __builtin_amdgcn_readfirstlane(tid >> 6);

    u32x4 gvh, gvl, xh4, xl4;
    u32x2 xh2, xl2;
    auto gload = [&](int step) {
        const int o = 4 * step + oc;
        const W5Part q = w5_part(p, o, true);
        const int ol = o - q.o0;
        const int b = ol / q.NO, oct = ol - b * q.NO;
        const bool ov = b < q.B;
        const size_t go = ((size_t)b * p.M + m0 + r64) * (8 * q.NO) + 8 * oct;
        const size_t xo = ((size_t)b * p.C + c0 + r64) * (8 * q.NO + 8) + 8 * oct;
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        const u32x2 z2 = {0u, 0u};
        gvh = ov ? *reinterpret_cast<const u32x4*>(q.gh + go) : z4;
        gvl = ov ? *reinterpret_cast<const u32x4*>(q.gl + go) : z4;
        xh4 = ov ? *reinterpret_cast<const u32x4*>(q.xh + xo) : z4;
        xl4 = ov ? *reinterpret_cast<const u32x4*>(q.xl + xo) : z4;
        xh2 = ov ? *reinterpret_cast<const u32x2*>(q.xh + xo + 8) : z2;
        xl2 = ov ? *reinterpret_cast<const u32x2*>(q.xl + xo + 8) : z2;
    };
    auto stage = [&](unsigned char* img) {
        unsigned char* a = img + (oc * TCO + r64) * 16;
        *reinterpret_cast<u32x4*>(a) = gvh;
        *reinterpret_cast<u32x4*>(a + A2_PIECE) = gvl;
        unsigned char* bq = img + 2 * A2_PIECE + (oc * TCI + r64) * 32;
        const int h0 = (oc & 1) * 16, h1 = 16 - h0;
        *reinterpret_cast<u32x4*>(bq + h0) = xh4;
        *reinterpret_cast<u32x2*>(bq + h1) = xh2;
        *reinterpret_cast<u32x4*>(bq + B2_PIECE + h0) = xl4;
        *reinterpret_cast<u32x2*>(bq + B2_PIECE + h1) = xl2;
    };

    f32x4 acc[4][5];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[mt][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int n = lane & 15, kg = lane >> 4;
    const int a_rd = (kg * TCO + n) * 16;
    const int b_rd = 2 * A2_PIECE + (kg * TCI + 16 * wid + n) * 32;
    const int bh0 = (kg & 1) * 16, bh1 = 16 - bh0;
    auto compute = [&](const unsigned char* img) {
        f16x8 A[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
                A[mt][pc] = *reinterpret_cast<const f16x8*>(img + a_rd + mt * 256 + pc * A2_PIECE);
        unsigned D[2][6];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const u32x4 lo = *reinterpret_cast<const u32x4*>(img + b_rd + pc * B2_PIECE + bh0);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(img + b_rd + pc * B2_PIECE + bh1);
            D[pc][0] = lo[0]; D[pc][1] = lo[1]; D[pc][2] = lo[2]; D[pc][3] = lo[3]; D[pc][4] = hi[0]; D[pc][5] = hi[1];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            f16x8 Bo[2];
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                u32x4 v;
                if ((k & 1) == 0) {
                    v = (u32x4){D[pc][k / 2], D[pc][k / 2 + 1], D[pc][k / 2 + 2], D[pc][k / 2 + 3]};
                } else {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        v[qq] = __builtin_amdgcn_alignbit(D[pc][k / 2 + qq + 1], D[pc][k / 2 + qq], 16);
                }
                Bo[pc] = __builtin_bit_cast(f16x8, v);
            }
            // three partial products, smallest first; the four m-tiles' chains are interleaved
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[mt][0], Bo[1], acc[mt][k], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[mt][1], Bo[0], acc[mt][k], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[mt][0], Bo[0], acc[mt][k], 0, 0, 0);
        }
    };

    const int s_beg = z * p.sps, s_end = min(p.nsteps, s_beg + p.sps);
    unsigned char* img0 = smem5;
    unsigned char* img1 = smem5 + IMG2;
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

    // ---- undo the chunk scales: an m-tile of 16 co is one gradient chunk, a wave's 16 ci one input chunk
    const int ncx = p.C / 16;
    float sx, isx;
    chunk_scale(__builtin_bit_cast(float, p.mx[c0 / 16 + wid]), sx, isx);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float sg, isg;
        chunk_scale(__builtin_bit_cast(float, p.mx[ncx + m0 / 16 + mt]), sg, isg);
        const float f = isx * isg;
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[mt][k] *= f;
    }
    // ---- slab (as k_wgrad_k5_split)
    float* part = partial + (size_t)z * p.stride;
    float* ts = reinterpret_cast<float*>(smem5);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ml = 0; ml < 2; ++ml)
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ts[((ml * 16 + 4 * kg + r) * TCI + 16 * wid + n) * 5 + k] = acc[2 * half + ml][k][r];
        __syncthreads();
#pragma unroll
        for (int qq = 0; qq < EPI_FLOATS / 4 / 256; ++qq) {
            const int idx = tid + 256 * qq;
            const int row = idx / 80, c4 = idx - row * 80;
            const f32x4 v = *reinterpret_cast<const f32x4*>(ts + row * 320 + 4 * c4);
            *reinterpret_cast<f32x4*>(part + ((size_t)(m0 + 32 * half + row) * p.C + c0) * 5 + 4 * c4) = v;
        }
        __syncthreads();
    }
    // bias gradient: the per-row sums of k_w5_split, added up by slab 0's ci-tile-0 workgroups (zeros in the other slabs)
    if (blockIdx.y == 0 && tid < TCO) {
        float v = 0.f;
        if (z == 0) {
            const int rows = p.row0[p.count];
            for (int r = 0; r < rows; ++r) v += p.tb[(size_t)r * p.M + m0 + tid];
        }
        part[(size_t)p.M * p.C * 5 + m0 + tid] = v;
    }
}

int pick_nsplit(const ConvP& p, int nsteps) {
    const int tiles = (p.Cout / TCO) * (p.Cin / TCI);
    int ns = (512 + tiles - 1) / tiles;                     // ~2 workgroups per CU
    if (ns > nsteps / 4) ns = nsteps / 4;                   // at least 4 steps per slab
    const size_t slab = ((size_t)p.Cout * p.Cin * 5 + p.Cout) * sizeof(float);
    while (ns > 1 && (size_t)ns * slab > ((size_t)96 << 20)) --ns;
    return ns < 1 ? 1 : ns;
}

bool w5_geometry(const ConvP& p) {
    return p.groups == 1 && p.stride == 1 && p.dil == 1 && p.K == 5 && p.pad == 2 && p.Lout == p.Lin &&
           p.pad_mode == MS_PAD_ZERO && !p.in_act && p.Cout % TCO == 0 && p.Cin % TCI == 0 && p.Cout >= 256 &&
           p.Cin >= 256 && p.Lin <= 64 && (long long)p.B * p.Cout * p.Lin * 4 < (1ll << 31) &&
           (long long)p.B * p.Cin * p.Lin * 4 < (1ll << 31);
}

bool w5_enabled() {
    const char* e = getenv("MSYNTH_WGRAD5");          // tuning / test switch (0: fp32-MFMA row-tile kernel)
    return !(e && atoi(e) == 0);
}

// table of `n` parts (B[i], L[i]) of the layer c; pointers filled by the caller
void w5_table(const ConvP& c, int n, const int* B, const int* L, W5P* q) {
    q->M = c.Cout; q->C = c.Cin; q->act = c.act; q->slope = c.slope;
    q->stride = (size_t)c.Cout * c.Cin * 5 + c.Cout;
    q->count = n;
    q->o0[0] = 0;
    for (int i = 0; i < MS_CONV_PARTS_MAX; ++i) {
        const bool on = i < n;
        q->B[i] = on ? B[i] : 0; q->L[i] = on ? L[i] : 1; q->NO[i] = on ? (L[i] + 7) / 8 : 1;
        q->o0[i + 1] = q->o0[i] + q->B[i] * q->NO[i];
        q->x[i] = q->gy[i] = q->ya[i] = nullptr;
    }
    q->nsteps = (q->o0[n] + 3) / 4;
    const int ns = pick_nsplit(c, q->nsteps);
    q->sps = (q->nsteps + ns - 1) / ns;
}

int w5_launch(const ConvP& c, const W5P& q, bool vec, float* gw, float* gb, float beta, void* ws, size_t ws_bytes, hipStream_t s) {
    const int nz = (q.nsteps + q.sps - 1) / q.sps;
    if (!ws || ws_bytes < (size_t)nz * q.stride * sizeof(float) || (((uintptr_t)ws) & 15)) return MS_ERR_WORKSPACE;
    const dim3 grid(c.Cout / TCO, c.Cin / TCI, nz);
    static unsigned long long attr_set = 0;
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_k5_split<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_k5_split<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG);
        ms_done_on_device(attr_set);
    }
    float* partial = (float*)ws;
    ms_note_kernel(6, "k_wgrad_k5_split<%s>", vec ? "true" : "false");
    if (vec) hipLaunchKernelGGL((k_wgrad_k5_split<true>), grid, dim3(256), 2 * IMG, s, q, partial);
    else hipLaunchKernelGGL((k_wgrad_k5_split<false>), grid, dim3(256), 2 * IMG, s, q, partial);
    MS_CHECK_LAUNCH();
    return msm_wgrad_reduce(partial, q.stride, nz, (size_t)c.Cout * c.Cin * 5, c.Cout, gw, gb, beta, s);
}

// ---- host side of the pre-split path
bool w5_pre_enabled() {
    const char* e = getenv("MSYNTH_W5_NP");            // tuning / test switch (3: the exact three-piece bf16 kernel, operands split
    return !(e && atoi(e) == 3);                       //  in the loader)
}

size_t al256(size_t n) { return (n + 255) & ~(size_t)255; }

struct W5Layout { size_t slabs, mx, pmx, tb, gh[MS_CONV_PARTS_MAX], gl[MS_CONV_PARTS_MAX], xh[MS_CONV_PARTS_MAX], xl[MS_CONV_PARTS_MAX], total; };

void w5q_table(const ConvP& c, int n, const int* B, const int* L, W5Q* q, W5Layout* lay) {
    q->M = c.Cout; q->C = c.Cin; q->act = c.act; q->slope = c.slope;
    q->stride = (size_t)c.Cout * c.Cin * 5 + c.Cout;
    q->count = n;
    q->o0[0] = q->row0[0] = 0;
    for (int i = 0; i < MS_CONV_PARTS_MAX; ++i) {
        const bool on = i < n;
        q->B[i] = on ? B[i] : 0; q->L[i] = on ? L[i] : 1; q->NO[i] = on ? (L[i] + 7) / 8 : 1;
        q->o0[i + 1] = q->o0[i] + q->B[i] * q->NO[i];
        q->row0[i + 1] = q->row0[i] + q->B[i];
        q->x[i] = q->gy[i] = q->ya[i] = nullptr;
        q->gh[i] = q->gl[i] = q->xh[i] = q->xl[i] = nullptr;
    }
    q->nsteps = (q->o0[n] + 3) / 4;
    const int ns = pick_nsplit(c, q->nsteps);
    q->sps = (q->nsteps + ns - 1) / ns;
    const int nz = (q->nsteps + q->sps - 1) / q->sps;
    size_t o = 0;
    lay->slabs = o; o += al256((size_t)nz * q->stride * sizeof(float));
    lay->mx = o; o += al256((size_t)(c.Cin / 16 + c.Cout / 16) * sizeof(unsigned));
    lay->pmx = o; o += al256((size_t)q->row0[n] * (c.Cin / 16 + c.Cout / 16) * sizeof(float));
    lay->tb = o; o += al256((size_t)q->row0[n] * c.Cout * sizeof(float));
    for (int i = 0; i < n; ++i) {
        const size_t g = al256((size_t)q->B[i] * c.Cout * 8 * q->NO[i] * 2), x = al256((size_t)q->B[i] * c.Cin * (8 * q->NO[i] + 8) * 2);
        lay->gh[i] = o; o += g; lay->gl[i] = o; o += g;
        lay->xh[i] = o; o += x; lay->xl[i] = o; o += x;
    }
    lay->total = o;
}

int w5_pre_launch(const ConvP& c, W5Q& q, const W5Layout& lay, float* gw, float* gb, float beta, void* ws, size_t ws_bytes,
                  hipStream_t s) {
    if (!ws || ws_bytes < lay.total || (((uintptr_t)ws) & 255)) return MS_ERR_WORKSPACE;
    char* w8 = (char*)ws;
    for (int i = 0; i < q.count; ++i) {
        q.gh[i] = (unsigned short*)(w8 + lay.gh[i]); q.gl[i] = (unsigned short*)(w8 + lay.gl[i]);
        q.xh[i] = (unsigned short*)(w8 + lay.xh[i]); q.xl[i] = (unsigned short*)(w8 + lay.xl[i]);
    }
    q.mx = (unsigned*)(w8 + lay.mx);
    q.tb = (float*)(w8 + lay.tb);
    const int nz = (q.nsteps + q.sps - 1) / q.sps, rows = q.row0[q.count];
    float* pmx = (float*)(w8 + lay.pmx);
    const dim3 pgrid((unsigned)rows, (unsigned)((c.Cin + c.Cout) / WB_CH));
    hipLaunchKernelGGL(k_w5_maxima, pgrid, dim3(256), 0, s, q, pmx);
    MS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_w5_split, pgrid, dim3(256), 0, s, q, pmx);
    MS_CHECK_LAUNCH();
    static unsigned long long attr_set = 0;
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_k5_pre), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * IMG2);
        ms_done_on_device(attr_set);
    }
    float* partial = (float*)(w8 + lay.slabs);
    ms_note_kernel(3, "k_wgrad_k5_pre");
    hipLaunchKernelGGL(k_wgrad_k5_pre, dim3(c.Cout / TCO, c.Cin / TCI, nz), dim3(256), 2 * IMG2, s, q, partial);
    MS_CHECK_LAUNCH();
    return msm_wgrad_reduce(partial, q.stride, nz, (size_t)c.Cout * c.Cin * 5, c.Cout, gw, gb, beta, s);
}

bool w5_pre_ok(const ConvP& c, int n, const float* const* x, const float* const* gy, const float* const* ya) {
    if (!w5_pre_enabled() || c.Cin % WB_CH || c.Cout % WB_CH) return false;
    if (c.act != MS_ACT_NONE && c.act != MS_ACT_LRELU) return false;
    for (int i = 0; i < n; ++i)
        if ((((uintptr_t)x[i]) & 15) || (((uintptr_t)gy[i]) & 15)) return false;
    (void)ya;
    return true;
}

}  // namespace

bool msw5_applicable(const ConvP& p) { return w5_enabled() && w5_geometry(p); }

size_t msw5_ws(const ConvP& p) {
    W5P q;
    w5_table(p, 1, &p.B, &p.Lin, &q);
    size_t n = (size_t)((q.nsteps + q.sps - 1) / q.sps) * q.stride * sizeof(float);
    if (w5_pre_enabled()) {
        W5Q q2;
        W5Layout lay;
        w5q_table(p, 1, &p.B, &p.Lin, &q2, &lay);
        if (lay.total > n) n = lay.total;
    }
    return n;
}

const char* msw5_name(const ConvP&) { return "k_wgrad_k5_split"; }

int msw5_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act, float* gw, float* gb,
                    float beta, void* ws, size_t ws_bytes, hipStream_t s) {
    if (w5_pre_ok(p, 1, &x, &gy, &y_act) && ws && (((uintptr_t)ws) & 255) == 0) {
        W5Q q2;
        W5Layout lay;
        w5q_table(p, 1, &p.B, &p.Lin, &q2, &lay);
        if (ws_bytes >= lay.total) {
            q2.x[0] = x; q2.gy[0] = gy; q2.ya[0] = p.act == MS_ACT_NONE ? nullptr : y_act;
            return w5_pre_launch(p, q2, lay, gw, gb, beta, ws, ws_bytes, s);
        }
    }
    W5P q;
    w5_table(p, 1, &p.B, &p.Lin, &q);
    q.x[0] = x; q.gy[0] = gy; q.ya[0] = y_act;
    const bool vec = p.Lin % 4 == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)gy) & 15) == 0 &&
                     (!y_act || (((uintptr_t)y_act) & 15) == 0);
    return w5_launch(p, q, vec, gw, gb, beta, ws, ws_bytes, s);
}

// ---- the layer's weight gradient over all parts in one launch (api.hip: ms_conv1d_parts_bwd_weight)
bool msw5_parts_applicable(const ConvP& c, const ms_conv1d_parts* parts) {
    if (!w5_enabled() || !parts || parts->count < 2 || parts->count > MS_CONV_PARTS_MAX) return false;
    for (int i = 0; i < parts->count; ++i) {
        ConvP p = c;
        p.B = parts->B[i]; p.Lin = p.Lout = parts->Lin[i];
        if (p.B <= 0 || p.Lin <= 0 || !w5_geometry(p)) return false;
    }
    return true;
}

size_t msw5_parts_ws(const ConvP& c, const ms_conv1d_parts* parts) {
    W5P q;
    w5_table(c, parts->count, parts->B, parts->Lin, &q);
    size_t n = (size_t)((q.nsteps + q.sps - 1) / q.sps) * q.stride * sizeof(float);
    if (w5_pre_enabled()) {
        W5Q q2;
        W5Layout lay;
        w5q_table(c, parts->count, parts->B, parts->Lin, &q2, &lay);
        if (lay.total > n) n = lay.total;
    }
    return n;
}

int msw5_parts_bwd_weight(const ConvP& c, const ms_conv1d_parts* parts, float* gw, float* gb, float beta, void* ws,
                          size_t ws_bytes, hipStream_t s) {
    for (int i = 0; i < parts->count; ++i)
        if (!parts->x[i] || !parts->gy[i]) return MS_ERR_INVALID_ARG;
    if (w5_pre_ok(c, parts->count, parts->x, parts->gy, parts->y_act) && ws && (((uintptr_t)ws) & 255) == 0) {
        W5Q q2;
        W5Layout lay;
        w5q_table(c, parts->count, parts->B, parts->Lin, &q2, &lay);
        if (ws_bytes >= lay.total) {
            for (int i = 0; i < parts->count; ++i) {
                q2.x[i] = parts->x[i]; q2.gy[i] = parts->gy[i]; q2.ya[i] = c.act == MS_ACT_NONE ? nullptr : parts->y_act[i];
            }
            return w5_pre_launch(c, q2, lay, gw, gb, beta, ws, ws_bytes, s);
        }
    }
    W5P q;
    w5_table(c, parts->count, parts->B, parts->Lin, &q);
    for (int i = 0; i < parts->count; ++i) {
        if (!parts->x[i] || !parts->gy[i]) return MS_ERR_INVALID_ARG;
        q.x[i] = parts->x[i]; q.gy[i] = parts->gy[i]; q.ya[i] = c.act == MS_ACT_NONE ? nullptr : parts->y_act[i];
    }
    return w5_launch(c, q, false, gw, gb, beta, ws, ws_bytes, s);
}
