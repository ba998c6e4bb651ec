// Dense k5 / stride-1 / pad-2 conv on short rows with PRE-SPLIT weight images: the discriminator's 1024 -> 1024 layer
// (reference discriminator/full.py:19; rows of 32 / 17 / 9 samples at the three scales), forward and backward data.
//
// It is 70 % of the discriminator's FLOPs and ran on the generic split-bf16 row kernels (conv_rows3.hip) at 95-220 TFLOP/s:
// there every workgroup stages and SPLITS its 64 x 16 x 5 weight slab per chunk (5120 elements against 4224 activation
// elements: more than half of the staging work, PMC: 7445 vector instructions per 1920 MFMAs per wave) and keeps it in
// LDS twice (63 KB: one workgroup per CU).  Here, as in atom_fused.hip:
//   * the weights are split ONCE per step (k_conv5_pack) into fragment-ordered images -- one 1 KiB block per (32 rows,
//     16-channel chunk, tap, piece) = the A operand of one v_mfma_f32_32x32x16_bf16 -- and stream from L2 straight into
//     registers, a whole chunk (15 blocks) ahead: no LDS, no vector work for weights;
//   * LDS holds only the activation tile, double-buffered (2 x <= 39 KB: two workgroups per CU); all 8 waves run MFMAs
//     all the time (2 x 4 waves of 32 rows x 64 columns); the activation chunk c+1 is split and stored while chunk c is
//     multiplied, one barrier per chunk;
//   * a tile is 64 rows x R whole batch rows (R = 256 / L: 8 / 15 / 26 rows), each row with its own 2-column zero halo in
//     LDS, so rows of any length need no padding pass (api.hip pad4) and no dword loader;
//   * split-K over channel slices fills the chip (tiles x slices ~ 512 workgroups); slices write raw slabs, a small finish
//     kernel sums them in slice order (deterministic) and applies bias + LeakyReLU / the gradient add.
// Arithmetic (template NP, as in atom_fused.hip):
//   NP = 3  the exact 3-piece bf16 split and six products per multiply of conv_rows3.hip, fp32 accumulation;
//   NP = 2  (r04, default) block-scaled two-piece fp16 operands, THREE products per multiply on v_mfma_f32_32x32x16_f16.  The
//           block here is (batch row, 16-channel chunk): all five taps of an output column read that row's LDS segment, so
//           one power-of-two scale per segment factors out of the chunk's partial sum.  The staging lanes of a row (16-64
//           consecutive lanes of one wave) find the row's largest magnitude by lane shuffles -- no barrier --, scale, split,
//           and leave 1 / scale in LDS.  A row's scale is sticky across the chunks (it moves only when the row's chunk maximum
//           times the scale leaves [2^8, 2^15)), so the MFMA waves keep accumulating under it and fold their partial sums
//           into the running fp32 sums only when a scale has moved (rare).
// The accumulation order differs from the row kernels' (other chunk / slice grouping): results agree to ~1e-7, not bitwise.
#include "ms_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NP> __host__ __device__ constexpr int xrs() { return NP * 32 + 16; }   // bytes per LDS column of a 16-channel chunk
constexpr int K5 = 5;
constexpr int PX_MAX = 350;              // LDS columns per buffer (39.2 / 28 KB; two buffers)
constexpr int R_MAX = 64;                // batch rows per tile (scale slots)
constexpr unsigned OOB = 0xF0000000u;
// NP = 2: weights are packed as the fp16 pieces of S_w w, S_w a power of two taken from the tensor's largest magnitude (it goes
// to [2^12, 2^13): weights of any magnitude; r04 packed 64 w and overflowed to inf from |w| >= 2^9).  A pack is two launches:
// W_NPART partial maxima into the image's tail (the allocation is sized for three pieces), then the pack proper, whose
// threads reduce the partials and whose first thread leaves 1 / S_w in the tail for the consuming kernels.
constexpr int W_NPART = 256;          // partial maxima of a weight tensor (one workgroup each)
// (called by ALL 256 threads of a pack workgroup, before any of them returns: the first wave reduces the partials, LDS broadcasts)
__device__ __forceinline__ void weight_scale(const float* __restrict__ pm, float& S, float& invS) {
    __shared__ float wmax_s;
    if (threadIdx.x < 64) {
        float m = fmaxf(fmaxf(pm[threadIdx.x], pm[threadIdx.x + 64]), fmaxf(pm[threadIdx.x + 128], pm[threadIdx.x + 192]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (threadIdx.x == 0) wmax_s = m;
    }
    __syncthreads();
    const float m = wmax_s;
    const unsigned eb = (__builtin_bit_cast(unsigned, m) >> 23) & 0xFFu;
    const bool ok = eb >= 16u && eb <= 250u;
    S = ok ? __builtin_bit_cast(float, (266u - eb) << 23) : 1.f;
    invS = ok ? __builtin_bit_cast(float, (eb - 12u) << 23) : 1.f;
}
__host__ __device__ inline size_t c5_tail_u4(int M, int CK) { return (size_t)(M / 32) * (CK / 16) * 5 * 2 * 64; }   // end of the NP = 2 data

__global__ __launch_bounds__(256) void k_conv5_wmax(const float* __restrict__ W, size_t n, float* __restrict__ pm) {
    __shared__ float red[4];
    const size_t per = (n + W_NPART - 1) / W_NPART, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    float m = 0.f;
    // 16 bytes per lane at any 4-byte aligned address (a view into a flat parameter bucket): a 21 MB tensor in ~7 us, not 21
    typedef float f32x4w __attribute__((ext_vector_type(4), aligned(4)));
    size_t i = lo + 4 * (size_t)threadIdx.x;
    for (; i + 3 < hi; i += 1024) {
        const f32x4w v = *reinterpret_cast<const f32x4w*>(W + i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    for (; i < hi; ++i) m = fmaxf(m, fabsf(W[i]));                     // (the one thread whose quad crosses the part's end)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) pm[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// (a, b) -> NP packed 16-bit pairs: NP = 3 exact bf16 pieces; NP = 2 fp16 pieces a = o[0] + o[1] (22 bits; the caller has
// scaled a so that its block's largest magnitude sits in [2^8, 2^15): atom_fused.hip)
template <int NP>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&o)[NP]) {
    const f32x2 v = {a, b};
    if constexpr (NP == 3) {
        const bf16x2 hi = __builtin_convertvector(v, bf16x2);
        const f32x2 r1 = v - __builtin_convertvector(hi, f32x2);
        const bf16x2 mi = __builtin_convertvector(r1, bf16x2);
        const f32x2 r2 = r1 - __builtin_convertvector(mi, f32x2);
        const bf16x2 lo = __builtin_convertvector(r2, bf16x2);
        o[0] = __builtin_bit_cast(unsigned, hi);
        o[1] = __builtin_bit_cast(unsigned, mi);
        o[2] = __builtin_bit_cast(unsigned, lo);
    } else {
        const f16x2 hi = __builtin_convertvector(v, f16x2);
        const f16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x2), f16x2);
        o[0] = __builtin_bit_cast(unsigned, hi);
        o[1] = __builtin_bit_cast(unsigned, lo);
    }
}

template <int NP>
__device__ __forceinline__ void split_quad(const float (&e)[4], uint2 (&o)[NP]) {
    unsigned a[NP], b[NP];
    split_pair<NP>(e[0], e[1], a);
    split_pair<NP>(e[2], e[3], b);
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) o[pp] = make_uint2(a[pp], b[pp]);
}

// block scale of values whose largest magnitude is m: S = 2^k with m S in [2^14, 2^15), and 1 / S; 1 for a zero /
// denormal-range / non-finite maximum (atom_fused.hip)
__device__ __forceinline__ void block_scale(float m, float& S, float& invS) {
    const unsigned eb = (__builtin_bit_cast(unsigned, m) >> 23) & 0xFFu;
    const bool ok = eb >= 16u && eb <= 250u;
    S = ok ? __builtin_bit_cast(float, (268u - eb) << 23) : 1.f;
    invS = ok ? __builtin_bit_cast(float, (eb - 14u) << 23) : 1.f;
}

// image[ms][chunk][tap][piece][lane] (16 B): rows ms*32 + (lane & 31), contraction channels chunk*16 + 8*(lane >> 5) + 0..7
//   forward:        A[row = co][k = ci][tap] = W[co][ci][tap]
//   backward data:  A[row = ci][k = co][tap] = W[co][ci][4 - tap]
// pm: the tensor's partial maxima (k_conv5_wmax); null = in this image's own tail.  (The two images of a layer -- forward,
// backward data -- are packed from one maxima pass: ms_conv1d_img_pack2.)
__global__ __launch_bounds__(256) void k_conv5_pack(const float* __restrict__ W, u32x4* __restrict__ img, int M, int CK,
                                                   int backward, int np, const float* __restrict__ pm) {
    const int NC = CK / 16;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over ms x chunk x tap x lane
    const size_t total = (size_t)(M / 32) * NC * K5 * 64;
    float WS = 1.f, iWS = 1.f;
    if (np == 2) {
        float* tail = reinterpret_cast<float*>(img + c5_tail_u4(M, CK));
        weight_scale(pm ? pm : tail, WS, iWS);
        if (idx == 0) tail[W_NPART] = iWS;
    }
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    size_t r = idx >> 6;
    const int tap = (int)(r % K5); r /= K5;
    const int chunk = (int)(r % NC);
    const int ms = (int)(r / NC);
    const int row = ms * 32 + (lane & 31), k0 = chunk * 16 + 8 * (lane >> 5);
    unsigned pc[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float a, b;
        if (!backward) {          // W is (Cout = M, Cin = CK, 5)
            a = W[((size_t)row * CK + k0 + 2 * q) * K5 + tap];
            b = W[((size_t)row * CK + k0 + 2 * q + 1) * K5 + tap];
        } else {                  // W is (Cout = CK, Cin = M, 5)
            a = W[((size_t)(k0 + 2 * q) * M + row) * K5 + (K5 - 1 - tap)];
            b = W[((size_t)(k0 + 2 * q + 1) * M + row) * K5 + (K5 - 1 - tap)];
        }
        if (np == 3) {
            unsigned o[3];
            split_pair<3>(a, b, o);
            pc[0][q] = o[0]; pc[1][q] = o[1]; pc[2][q] = o[2];
        } else {
            unsigned o[2];
            split_pair<2>(a * WS, b * WS, o);
            pc[0][q] = o[0]; pc[1][q] = o[1]; pc[2][q] = 0u;
        }
    }
    u32x4* dst = img + ((size_t)((ms * NC + chunk) * K5 + tap) * np) * 64 + lane;
    for (int pp = 0; pp < np; ++pp) dst[pp * 64] = u32x4{pc[pp][0], pc[pp][1], pc[pp][2], pc[pp][3]};
}

struct C5P {
    int B, M, CK, L;          // GEMM rows (output channels of this pass), contraction channels, row length
    int R, SS, PX;            // batch rows per tile, LDS columns per row (L + 4), LDS columns per buffer (R * SS)
    int NV, NVG;              // 4-sample vectors per row, groups of 4 vectors
    int cks, nsplit;          // channels per split-K slice (multiple of 16), slices
    int act;                  // fused epilogue when nsplit == 1: bias + act (forward) / + add (backward data)
    float slope;
    long long zstride;        // floats per slab
};

extern __shared__ __attribute__((aligned(16))) unsigned char smem5[];

// MODE 0: forward (X = x); MODE 1: backward data (X = gy, multiplied on load by act'(Xact) when Xact != nullptr)
// bx: the workgroup's batch-row tile (blockIdx.x in a single launch, its offset within the part's range in a parts launch)
// PRE (NP = 2, parts launches): the activations arrive ALREADY split -- k_conv5_presplit wrote them once as fp16 pieces in
// exactly the order an LDS column holds them (P: [batch row][chunk][sample][piece][16 channels], 64 bytes per column), with the
// rows' sticky chunk scales beside them (Pinv: [batch row][chunk]) -- so staging a chunk is two 16-byte copies per thread
// instead of eight loads, a lane-shuffle maximum, 32 multiplies and 16 splits: the operand split is out of the K loop.
template <int MODE, int NP, bool PRE = false>
__device__ __forceinline__ void conv5_body(const C5P& p, const float* __restrict__ X, const float* __restrict__ Xact,
                                           const u32x4* __restrict__ IMG, const float* __restrict__ bias,
                                           const float* __restrict__ add, float* __restrict__ Y,
                                           float* __restrict__ slabs, int bx, const u32x4* __restrict__ P = nullptr,
                                           const float* __restrict__ Pinv = nullptr) {
    constexpr int XRS = xrs<NP>();
    constexpr bool SC = NP == 2;
    static_assert(!PRE || NP == 2, "pre-split operands are fp16 pieces");
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;                   // 2 x 4 waves: 32 rows x 64 columns each
    const int L = p.L, b0 = bx * p.R, m0 = blockIdx.y * 64;
    const int cbeg = blockIdx.z * p.cks;
    const int cend = cbeg + p.cks < p.CK ? cbeg + p.cks : p.CK;
    const int nchunks = (cend - cbeg) / 16;
    const int buf_bytes = p.PX * XRS;
    const bool masked = MODE == 1 && Xact != nullptr;
    float* sinv = reinterpret_cast<float*>(smem5 + 2 * buf_bytes);       // NP = 2: [2 buffers][R_MAX] 1 / scale of a row's chunk

    // (true sizes: a 4-sample vector that starts inside the last row may reach past the tensor -- those dwords read 0.0)
    const unsigned x_bytes = 4u * (unsigned)(p.B * p.CK * L);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, x_bytes, 0x00020000);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(masked ? Xact : X), 0, x_bytes, 0x00020000);
    const auto rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(IMG), 0, 0x80000000u, 0x00020000);

    // ---- staging units (two rounds of 512): unit = (batch row r, channel quad cq, 4-sample vector v); 16 consecutive
    // lanes = 4 quads x 4 consecutive vectors; a row = NVG (a power of two for NP = 2) consecutive 16-lane groups of ONE
    // wave.  Rows are contiguous runs of L floats, 4-byte aligned (any L).
    unsigned u_goff[2];
    int u_l[2], u_lbase[2], u_r[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + 512 * k;
        const int g16 = u >> 4, r = g16 / p.NVG, vg = g16 - r * p.NVG;
        const int cq = (u >> 2) & 3, v = vg * 4 + (u & 3);
        const bool ok = r < p.R && v < p.NV && b0 + r < p.B;
        u_goff[k] = ok ? 4u * (unsigned)((((b0 + r) * p.CK) + 4 * cq) * L + 4 * v) : OOB;       // + (c0 + cc) * L * 4 (scalar)
        u_l[k] = ok ? 4 * v : (1 << 20);
        u_lbase[k] = (r * p.SS + 2 + 4 * v) * XRS + cq * 8;
        u_r[k] = (r < p.R && (u & (16 * p.NVG - 1)) == 0) ? r : -1;           // the lane that publishes the row's 1 / scale
    }
    // PRE: copy items (two rounds of 512): item = (column of the tile, 16-byte quarter of its 64 pre-split bytes)
    const int NCH = p.CK / 16;
    unsigned q_goff[2];
    int q_lds[2];
    u32x4 rp[2];
    float rinv = 1.f;
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(PRE ? P : IMG), 0,
                                                        PRE ? 64u * (unsigned)(p.B * NCH * L) : 16u, 0x00020000);
    if (PRE) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + 512 * k, col = i >> 2, qq = i & 3;
            const int r = col / L, l = col - r * L;
            const bool ok = r < p.R && b0 + r < p.B;
            q_goff[k] = ok ? 64u * (unsigned)(((b0 + r) * NCH) * L + l) + 16u * qq : OOB;        // + chunk * L * 64 (scalar)
            q_lds[k] = ok ? (r * p.SS + 2 + l) * XRS + qq * 16 : -1;
        }
    }
    f32x4 rx[2][4], rxa[MODE == 1 ? 2 : 1][4];
    auto load_x = [&](int c0, bool live) {
        if (PRE) {
            const int so = live ? (c0 / 16) * L * 64 : 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) rp[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, q_goff[k], so, 0));
            if (tid < p.R) rinv = (live && b0 + tid < p.B) ? Pinv[(size_t)(b0 + tid) * NCH + c0 / 16] : 1.f;
            return;
        }
        const int so = live ? 4 * c0 * L : 0;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                rx[k][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, u_goff[k], so + cc * 4 * L, 0));
                if (MODE == 1)
                    rxa[k][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, u_goff[k], so + cc * 4 * L, 0));
            }
    };
    // NP = 2: the scale of a row is STICKY across the chunks of the K loop: it moves only when the row's largest magnitude
    // in a chunk, times the current scale, leaves [2^8, 2^15) (no overflow; fp16's normal range still holds every element
    // down to 2^-22 of that maximum with 11 + 11 bits).  The MFMA waves can then keep accumulating across chunks and fold
    // their partial sums into the running sums only where a scale has moved.
    float Scur[2] = {0.f, 0.f};
    auto store_x = [&](unsigned char* buf, float* inv_out) {
        if (PRE) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (q_lds[k] >= 0) *reinterpret_cast<u32x4*>(buf + q_lds[k]) = rp[k];
            if (tid < p.R) inv_out[tid] = rinv;
            return;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (SC) {
                // the row's largest magnitude in this chunk (the raw gradient bounds the masked one).  The row's lanes are
                // 16 NVG consecutive lanes of this wave; lanes without a unit hold zeros (out-of-range loads)
                float m = 0.f;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (u_l[k] + e < L) m = fmaxf(m, fabsf(rx[k][cc][e]));      // (samples behind the row belong to a neighbour: not staged, not in the scale)
                // 16 lanes by DPP (quad swaps, half-row mirror, row mirror), wider spans by lane shuffles
                m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));
                m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true)));
                m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x141, 0xF, 0xF, true)));
                m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x140, 0xF, 0xF, true)));
                for (int o = 16; o < 16 * p.NVG; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                const float ms = m * Scur[k];
                if (!(ms >= 256.f && ms < 32768.f) && m > 0.f) {         // (also the first chunk: Scur = 0)
                    float inv;
                    block_scale(m * 4.f, Scur[k], inv);                  // largest magnitude at 2^12: room to grow and to shrink
                }
                if (Scur[k] == 0.f) Scur[k] = 1.f;                       // an all-zero row so far
                if (u_r[k] >= 0) inv_out[u_r[k]] = 1.f / Scur[k];        // (a power of two: exact)
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (u_l[k] + e >= L) continue;                        // behind the row: the halo stays zero
                float c4[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    c4[cc] = rx[k][cc][e];
                    if (masked) c4[cc] = rxa[k][cc][e] > 0.f ? c4[cc] : c4[cc] * p.slope;
                    if (SC) c4[cc] *= Scur[k];
                }
                uint2 o3[NP];
                split_quad<NP>(c4, o3);
                unsigned char* dst = buf + u_lbase[k] + e * XRS;
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) *reinterpret_cast<uint2*>(dst + pp * 32) = o3[pp];
            }
        }
    };

    // ---- A fragments: one chunk (5 taps x NP pieces) in registers, refilled tap by tap for the next chunk
    u32x4 fa[K5][NP];
    const int NC = p.CK / 16;
    const int a_voff = lane * 16;
    const int a_row = ((blockIdx.y * 2 + wm) * NC) * (K5 * NP * 1024);          // byte offset of (ms, chunk 0)
    auto load_a_tap = [&](int chunk_abs, int t) {
        const int so = a_row + chunk_abs * (K5 * NP * 1024) + t * NP * 1024;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
            fa[t][pp] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsI, a_voff, so + pp * 1024, 0));
    };

    // ---- B fragment bases: MFMA column n = wn*64 + j*32 + l31 -> (row r, sample l)
    int bbase[2], brow[2];
    unsigned o_lane[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = wn * 64 + j * 32 + l31;
        const int r = n / L, l = n - r * L;
        const bool ok = r < p.R && b0 + r < p.B;
        bbase[j] = ok ? (r * p.SS + l) * XRS + h * 16 : h * 16;
        brow[j] = r < p.R ? r : 0;
        o_lane[j] = ok ? 4u * (unsigned)((r * p.M + 4 * h) * L + l) : OOB;        // + ((b0*M + m) * L) * 4 (scalar)
    }

    // ---- zero both LDS buffers (the halos are never written), stage chunk 0, prefetch chunk 1
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int i = tid * 16; i < 2 * buf_bytes; i += 512 * 16) *reinterpret_cast<u32x4*>(smem5 + i) = z;
        if (SC && tid < 2 * R_MAX) sinv[tid] = 1.f;
    }
    load_x(cbeg, true);
#pragma unroll
    for (int t = 0; t < K5; ++t) load_a_tap(cbeg / 16, t);
    __syncthreads();
    store_x(smem5, sinv);
    load_x(cbeg + 16, nchunks > 1);
    __syncthreads();

    f32x16 acc[2];                                   // the running sums (NP = 2: fp32, unscaled except for the weight scale S_w)
    f32x16 cm[SC ? 2 : 1];                           // NP = 2: partial sums under the rows' current scales
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[j][r] = 0.f;
            if (SC) cm[j][r] = 0.f;
        }
    float inv_cur[2] = {1.f, 1.f};
    auto fold = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[j][r] = fmaf(cm[j][r], inv_cur[j], acc[j][r]);
                cm[j][r] = 0.f;
            }
    };

#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
        const unsigned char* Xs = smem5 + (ch & 1) * buf_bytes;
        unsigned char* Xn = smem5 + ((ch & 1) ^ 1) * buf_bytes;
        const bool more = ch + 1 < nchunks;
        const int a_next = (cbeg / 16) + (more ? ch + 1 : ch);
        u32x4 fb[2][2][NP];
        auto fragb = [&](int t, u32x4 (&dst)[2][NP]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pp = 0; pp < NP; ++pp)
                    dst[j][pp] = *reinterpret_cast<const u32x4*>(Xs + bbase[j] + t * XRS + pp * 32);
        };
        if (SC) {
            // a row's scale moved: what was accumulated under the old scale goes to the running sums first (rare)
            const float i0 = sinv[(ch & 1) * R_MAX + brow[0]], i1 = sinv[(ch & 1) * R_MAX + brow[1]];
            if (__any(i0 != inv_cur[0] || i1 != inv_cur[1])) {
                if (ch > 0) fold();
                inv_cur[0] = i0; inv_cur[1] = i1;
            }
        }
        fragb(0, fb[0]);
#pragma unroll
        for (int t = 0; t < K5; ++t) {
            if (t + 1 < K5) fragb(t + 1, fb[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NP == 3) {
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                for (int s = 0; s < 6; ++s)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[t][PA[s]]),
                                                                         __builtin_bit_cast(bf16x8, fb[t & 1][j][PB[s]]), acc[j], 0, 0, 0);
            } else {
                const f16x8 ah = __builtin_bit_cast(f16x8, fa[t][0]), al = __builtin_bit_cast(f16x8, fa[t][1]);
                const f16x8 bh0 = __builtin_bit_cast(f16x8, fb[t & 1][0][0]), bl0 = __builtin_bit_cast(f16x8, fb[t & 1][0][1]);
                const f16x8 bh1 = __builtin_bit_cast(f16x8, fb[t & 1][1][0]), bl1 = __builtin_bit_cast(f16x8, fb[t & 1][1][1]);
                // (the two column tiles' accumulator chains interleaved: dependent MFMAs sit one issue apart)
                cm[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl0, cm[0], 0, 0, 0);
                cm[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl1, cm[1], 0, 0, 0);
                cm[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh0, cm[0], 0, 0, 0);
                cm[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh1, cm[1], 0, 0, 0);
                cm[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh0, cm[0], 0, 0, 0);
                cm[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh1, cm[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifndef MS_C5_PROBE
#define MS_C5_PROBE 0
#endif
            // (timing probes, never in a shipped build: 1 = no weight-fragment loads, 2 = no activation staging, 4 = no barrier)
            if (!(MS_C5_PROBE & 1)) load_a_tap(a_next, t);             // this tap's registers are free: the next chunk's tap
            if (!(MS_C5_PROBE & 2)) {
                if (t == 1 && more) store_x(Xn, sinv + ((ch & 1) ^ 1) * R_MAX);   // chunk ch+1: registers -> the other buffer
                if (t == 2) load_x(cbeg + (ch + 2) * 16, ch + 2 < nchunks);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(MS_C5_PROBE & 4)) __syncthreads();
    }
    if (SC) fold();

    // ---- epilogue: dword stores straight from the accumulators (32 lanes = 32 consecutive samples of one or two rows)
    const float winv = SC ? reinterpret_cast<const float*>(IMG + c5_tail_u4(p.M, p.CK))[W_NPART] : 1.f;     // 1 / S_w (k_conv5_pack)
    int L4;
    asm volatile("s_mov_b32 %0, %1" : "=s"(L4) : "s"(4 * L));
    const bool fused = p.nsplit == 1;
    float* out = fused ? Y : slabs + (size_t)blockIdx.z * p.zstride;
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x80000000u, 0x00020000);
    const auto rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(add ? add : X), 0, 0x80000000u, 0x00020000);
    const int base = b0 * p.M * L4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int chs = m0 + wm * 32 + 8 * g;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (fused && MODE == 0 && bias) bv = *reinterpret_cast<const f32x4*>(bias + chs + 4 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = acc[j][4 * g + q];
                if (SC) v *= winv;
                if (fused) {
                    if (MODE == 0) {
                        v += bv[q];
                        if (p.act == MS_ACT_LRELU) v = v > 0.f ? v : v * p.slope;
                    } else if (add) {
                        v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, o_lane[j], base + (chs + q) * L4, 0));
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsO, o_lane[j], base + (chs + q) * L4, 0);
            }
        }
}

template <int MODE, int NP>
__global__ __launch_bounds__(512, 2) void k_conv5_img(C5P p, const float* __restrict__ X, const float* __restrict__ Xact,
                                                     const u32x4* __restrict__ IMG, const float* __restrict__ bias,
                                                     const float* __restrict__ add, float* __restrict__ Y,
                                                     float* __restrict__ slabs) {
    conv5_body<MODE, NP>(p, X, Xact, IMG, bias, add, Y, slabs, blockIdx.x);
}

// The layer over up to three inputs of different batch size / row length in ONE launch (the shared discriminator's three
// scales: rows of 32 / 17 / 9 samples, reference discriminator/melgan.py:13-27).  Batch-row tiles [bx0[i], bx0[i + 1]) belong
// to part i and run exactly as a launch of their own would, with that part's tile geometry; no split-K (the parts together
// fill the chip: 16 x (8 + 5 + 3) = 256 workgroups at B = 64), so every tile applies its epilogue itself and there is no
// slab traffic and no finish kernel.
struct C5Parts {
    int count, bx0[MS_CONV_PARTS_MAX + 1];
    int B[MS_CONV_PARTS_MAX], L[MS_CONV_PARTS_MAX], R[MS_CONV_PARTS_MAX], SS[MS_CONV_PARTS_MAX], PX[MS_CONV_PARTS_MAX],
        NV[MS_CONV_PARTS_MAX], NVG[MS_CONV_PARTS_MAX];
    const float* X[MS_CONV_PARTS_MAX];
    const float* Xact[MS_CONV_PARTS_MAX];
    const float* add[MS_CONV_PARTS_MAX];
    float* Y[MS_CONV_PARTS_MAX];
    u32x4* P[MS_CONV_PARTS_MAX];          // pre-split activations of the part (k_conv5_presplit), PRE launches
    float* Pinv[MS_CONV_PARTS_MAX];       // 1 / scale per (batch row, chunk)
};

// Two small passes over the layer's input ahead of a PRE launch; workgroup = (batch row of a part, block of PB_CH channels):
//   k_conv5_rowmax    the block's largest magnitude (raw gradient for MODE 1: it bounds the masked one) -> pm[row][block]
//   k_conv5_presplit  the row's power-of-two scale (its largest magnitude at 2^12: ONE scale per batch row -- every output
//                     column sums over one batch row only, so the scale factors out of the whole contraction and the K
//                     loop never folds), then the values (MODE 1: times the LeakyReLU derivative at Xact), scaled and split
//                     into fp16 pieces, 64 bytes per (16-channel chunk, sample)
// An element within 2^16 of its batch row's largest magnitude keeps 22 significand bits; smaller ones an absolute error below
// 2^-37 of that maximum.
constexpr int PB_CH = 128;                 // channels per pre-pass workgroup

struct C5Pre { int row0, B, L; const float* X; const float* Xact; u32x4* P; float* Pinv; };
__device__ __forceinline__ C5Pre c5_pre_part(const C5Parts& q, int row) {
    C5Pre r{0, q.B[0], q.L[0], q.X[0], q.Xact[0], q.P[0], q.Pinv[0]};
    int wg0 = 0;
#pragma unroll
    for (int k = 1; k < MS_CONV_PARTS_MAX; ++k) {
        wg0 += q.B[k - 1];
        if (k < q.count && row >= wg0) r = C5Pre{wg0, q.B[k], q.L[k], q.X[k], q.Xact[k], q.P[k], q.Pinv[k]};
    }
    return r;
}

__global__ __launch_bounds__(256) void k_conv5_rowmax(C5P p, C5Parts q, float* __restrict__ pm) {
    __shared__ float red[4];
    const C5Pre r = c5_pre_part(q, blockIdx.x);
    const int b = (int)blockIdx.x - r.row0, tid = threadIdx.x;
    // the block's PB_CH * L floats are contiguous and start 16-byte aligned
    const float* src = r.X + ((size_t)b * p.CK + (size_t)blockIdx.y * PB_CH) * r.L;
    float m = 0.f;
    for (int v = tid; v < PB_CH * r.L / 4; v += 256) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(src + 4 * v);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) pm[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <int MODE>
__global__ __launch_bounds__(256) void k_conv5_presplit(C5P p, C5Parts q, const float* __restrict__ pm, float slope) {
    const C5Pre r = c5_pre_part(q, blockIdx.x);
    const int b = (int)blockIdx.x - r.row0, tid = threadIdx.x, L = r.L;
    const int NC = p.CK / 16, c0 = blockIdx.y * (PB_CH / 16);
    float m = 0.f;
    for (int k = 0; k < (int)gridDim.y; ++k) m = fmaxf(m, pm[(size_t)blockIdx.x * gridDim.y + k]);
    float S, inv;
    block_scale(m * 4.f, S, inv);
    if (tid < PB_CH / 16) r.Pinv[(size_t)b * NC + c0 + tid] = inv;
    const bool masked = MODE == 1 && r.Xact != nullptr;
    const float* row = r.X + (size_t)b * p.CK * L;
    const float* rowa = (masked ? r.Xact : r.X) + (size_t)b * p.CK * L;
    for (int i = tid; i < (PB_CH / 16) * L; i += 256) {
        const int cl = i / L, l = i - cl * L, c = c0 + cl;
        const float* xr = row + (size_t)(16 * c) * L + l;
        const float* ar = rowa + (size_t)(16 * c) * L + l;
        float xv[16], av[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) xv[k] = xr[(size_t)k * L];
        if (masked) {
#pragma unroll
            for (int k = 0; k < 16; ++k) av[k] = ar[(size_t)k * L];
#pragma unroll
            for (int k = 0; k < 16; ++k) xv[k] = av[k] > 0.f ? xv[k] : xv[k] * slope;
        }
        unsigned hh[8], ll[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned o[2];
            split_pair<2>(xv[2 * k] * S, xv[2 * k + 1] * S, o);
            hh[k] = o[0]; ll[k] = o[1];
        }
        u32x4* dst = r.P + ((size_t)(b * NC + c) * L + l) * 4;
        dst[0] = u32x4{hh[0], hh[1], hh[2], hh[3]};
        dst[1] = u32x4{hh[4], hh[5], hh[6], hh[7]};
        dst[2] = u32x4{ll[0], ll[1], ll[2], ll[3]};
        dst[3] = u32x4{ll[4], ll[5], ll[6], ll[7]};
    }
}

template <int MODE, int NP, bool PRE>
__global__ __launch_bounds__(512, 2) void k_conv5_img_parts(C5P p, C5Parts q, const u32x4* __restrict__ IMG,
                                                           const float* __restrict__ bias) {
    // (the part's fields are picked with compile-time indices: a run-time index into the pointer arrays of a by-value kernel
    //  argument crashes this compiler's kernel-argument promotion)
    int bx0 = 0;
    const float* X = q.X[0];
    const float* Xact = q.Xact[0];
    const float* add = q.add[0];
    float* Y = q.Y[0];
    const u32x4* P = q.P[0];
    const float* Pinv = q.Pinv[0];
    p.B = q.B[0]; p.L = q.L[0]; p.R = q.R[0]; p.SS = q.SS[0]; p.PX = q.PX[0]; p.NV = q.NV[0]; p.NVG = q.NVG[0];
#pragma unroll
    for (int k = 1; k < MS_CONV_PARTS_MAX; ++k)
        if (k < q.count && (int)blockIdx.x >= q.bx0[k]) {
            bx0 = q.bx0[k];
            X = q.X[k]; Xact = q.Xact[k]; add = q.add[k]; Y = q.Y[k]; P = q.P[k]; Pinv = q.Pinv[k];
            p.B = q.B[k]; p.L = q.L[k]; p.R = q.R[k]; p.SS = q.SS[k]; p.PX = q.PX[k]; p.NV = q.NV[k]; p.NVG = q.NVG[k];
        }
    // (nsplit == 1: the slab pointer is never used; a literal nullptr there crashes the compiler's inliner, ROCm 7.2)
    conv5_body<MODE, NP, PRE>(p, X, PRE ? nullptr : Xact, IMG, bias, add, Y, Y, (int)blockIdx.x - bx0, P, Pinv);
}

// y = act(sum_z slab_z + bias[channel]) (+ add), slabs summed in slice order
__global__ __launch_bounds__(256) void k_conv5_finish(const float* __restrict__ slabs, int ns, long long zstride,
                                                     const float* __restrict__ bias, int M, int L, int act, float slope,
                                                     const float* __restrict__ add, float* __restrict__ Y, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float v = slabs[i];
        for (int z = 1; z < ns; ++z) v += slabs[(long long)z * zstride + i];
        if (bias) v += bias[(i / L) % M];
        if (act == MS_ACT_LRELU) v = v > 0.f ? v : v * slope;
        if (add) v += add[i];
        Y[i] = v;
    }
}

// pieces per operand element: 2 (block-scaled fp16 x 2, three products) unless MSYNTH_C5_NP=3 (bf16 x 3, six products)
int c5_np() {
    static const int np = (getenv("MSYNTH_C5_NP") && atoi(getenv("MSYNTH_C5_NP")) == 3) ? 3 : 2;
    return np;
}

bool c5_geometry(const ConvP& c, bool backward, C5P* p) {
    if (c.K != 5 || c.stride != 1 || c.dil != 1 || c.pad != 2 || c.groups != 1 || c.pad_mode != MS_PAD_ZERO || c.in_act) return false;
    if (c.Lout != c.Lin || c.Lin < 1 || c.Lin > 64) return false;
    if (c.act != MS_ACT_NONE && c.act != MS_ACT_LRELU) return false;
    const int M = backward ? c.Cin : c.Cout, CK = backward ? c.Cout : c.Cin;
    if (M % 64 || CK % 16 || M < 256 || CK < 256) return false;
    if ((long long)c.B * M * c.Lin * 4 >= (1ll << 31) || (long long)c.B * CK * c.Lin * 4 >= (1ll << 31)) return false;
    if ((long long)(M / 32) * (CK / 16) * (K5 * 3 * 1024) >= (1ll << 31)) return false;
    p->B = c.B; p->M = M; p->CK = CK; p->L = c.Lin;
    p->SS = c.Lin + 4;
    int R = 256 / c.Lin;
    if (R * p->SS > PX_MAX) R = PX_MAX / p->SS;
    if (R < 1) return false;
    p->R = R;
    p->PX = R * p->SS;
    p->NV = (c.Lin + 3) / 4;
    p->NVG = (p->NV + 3) / 4;
    if (c5_np() == 2) {                                      // a row's staging lanes = a power-of-two span inside one wave
        while (p->NVG & (p->NVG - 1)) ++p->NVG;
        if (p->NVG > 4 || R > R_MAX) return false;
    }
    while (R > 1 && R * p->NVG * 16 > 1024) --R;            // two staging rounds of 512 units
    if (R * p->NVG * 16 > 1024) return false;
    p->R = R;
    p->PX = R * p->SS;
    p->act = c.act; p->slope = c.slope;
    // split-K: tiles x slices ~ the 512 resident workgroups, at least 8 chunks per slice, at most 8 slices / 64 MiB of slabs
    const int tiles = (M / 64) * ((c.B + R - 1) / R);
    const int nchunks = CK / 16;
    int ns = 512 / tiles;
    if (ns < 1) ns = 1;
    if (ns > 8) ns = 8;
    while (ns > 1 && nchunks / ns < 8) --ns;
    p->zstride = (long long)c.B * M * c.Lin;
    while (ns > 1 && (size_t)ns * p->zstride * 4 > ((size_t)64 << 20)) --ns;
    int cks = ((nchunks + ns - 1) / ns) * 16;
    p->cks = cks;
    p->nsplit = (CK + cks - 1) / cks;
    return true;
}

bool c5_enabled() {
    const char* sw = getenv("MSYNTH_CONV5IMG");               // tuning / test switch (0: the generic row kernels)
    return !(sw && atoi(sw) == 0);
}

template <int MODE, int NP>
int c5_launch_np(const C5P& p, const float* X, const float* Xact, const void* image, const float* bias, const float* add, float* Y,
              void* ws, size_t ws_bytes, hipStream_t s) {
    float* slabs = nullptr;
    if (p.nsplit > 1) {
        const size_t need = (size_t)p.nsplit * p.zstride * sizeof(float);
        if (!ws || ws_bytes < need || (((uintptr_t)ws) & 15)) return MS_ERR_WORKSPACE;
        slabs = (float*)ws;
    }
    const size_t lds = (size_t)2 * p.PX * xrs<NP>() + 2 * R_MAX * sizeof(float);
    static unsigned long long attr_set = 0;
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv5_img<MODE, NP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * PX_MAX * xrs<NP>() + 2 * R_MAX * sizeof(float));
        ms_done_on_device(attr_set);
    }
    const dim3 grid((unsigned)((p.B + p.R - 1) / p.R), (unsigned)(p.M / 64), (unsigned)p.nsplit);
    ms_note_kernel(NP == 2 ? 3 : 6, "k_conv5_img<%d, %d>", MODE, NP);
    hipLaunchKernelGGL((k_conv5_img<MODE, NP>), grid, dim3(512), lds, s, p, X, Xact, (const u32x4*)image, bias, add, Y, slabs);
    MS_CHECK_LAUNCH();
    if (p.nsplit > 1) {
        const long long total = p.zstride;
        unsigned nb = (unsigned)((total + 255) / 256);
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(k_conv5_finish, dim3(nb), dim3(256), 0, s, slabs, p.nsplit, p.zstride, MODE == 0 ? bias : nullptr, p.M,
                           p.L, MODE == 0 ? p.act : MS_ACT_NONE, p.slope, MODE == 1 ? add : nullptr, Y, total);
        MS_CHECK_LAUNCH();
    }
    return MS_OK;
}

template <int MODE>
int c5_launch(const C5P& p, const float* X, const float* Xact, const void* image, const float* bias, const float* add, float* Y,
              void* ws, size_t ws_bytes, hipStream_t s) {
    if (c5_np() == 3) return c5_launch_np<MODE, 3>(p, X, Xact, image, bias, add, Y, ws, ws_bytes, s);
    return c5_launch_np<MODE, 2>(p, X, Xact, image, bias, add, Y, ws, ws_bytes, s);
}

bool c5_parts_geometry(const ConvP& c, const ms_conv1d_parts* parts, bool backward, C5P* p, C5Parts* q) {
    if (!parts || parts->count < 2 || parts->count > MS_CONV_PARTS_MAX) return false;
    q->count = parts->count;
    q->bx0[0] = 0;
    for (int i = 0; i < parts->count; ++i) {
        ConvP ci = c;
        ci.B = parts->B[i]; ci.Lin = ci.Lout = parts->Lin[i];
        C5P pi;
        if (ci.B <= 0 || ci.Lin <= 0 || !c5_geometry(ci, backward, &pi)) return false;
        if (i == 0) *p = pi;
        q->B[i] = pi.B; q->L[i] = pi.L; q->R[i] = pi.R; q->SS[i] = pi.SS; q->PX[i] = pi.PX; q->NV[i] = pi.NV; q->NVG[i] = pi.NVG;
        q->bx0[i + 1] = q->bx0[i] + (pi.B + pi.R - 1) / pi.R;
    }
    for (int i = parts->count; i < MS_CONV_PARTS_MAX; ++i) {
        q->B[i] = q->L[i] = q->R[i] = q->SS[i] = q->PX[i] = q->NV[i] = q->NVG[i] = 0;
        q->bx0[i + 1] = q->bx0[parts->count];
        q->X[i] = q->Xact[i] = q->add[i] = nullptr; q->Y[i] = nullptr;
    }
    for (int i = 0; i < MS_CONV_PARTS_MAX; ++i) { q->P[i] = nullptr; q->Pinv[i] = nullptr; }
    p->nsplit = 1;
    p->cks = p->CK;
    // without split-K the tiles alone must occupy the chip: at least one workgroup for every second CU
    return q->bx0[parts->count] * (p->M / 64) >= 128;
}

// workspace of a PRE launch: the parts' pre-split activations and scale tables (256-byte aligned blocks)
size_t c5_pre_bytes(const C5P& p, const C5Parts& q, size_t* offP, size_t* offS, size_t* offM = nullptr) {
    size_t o = 0;
    int rows = 0;
    for (int i = 0; i < q.count; ++i) rows += q.B[i];
    if (offM) *offM = o;
    o += ((size_t)rows * (p.CK / PB_CH) * sizeof(float) + 255) & ~(size_t)255;
    for (int i = 0; i < q.count; ++i) {
        if (offP) offP[i] = o;
        o += ((size_t)q.B[i] * (p.CK / 16) * q.L[i] * 64 + 255) & ~(size_t)255;
    }
    for (int i = 0; i < q.count; ++i) {
        if (offS) offS[i] = o;
        o += ((size_t)q.B[i] * (p.CK / 16) * sizeof(float) + 255) & ~(size_t)255;
    }
    return o;
}

bool c5_pre_enabled() {
    const char* sw = getenv("MSYNTH_C5_PRE");                 // tuning / test switch (0: operands split inside the K loop)
    return c5_np() == 2 && !(sw && atoi(sw) == 0);
}

template <int MODE, int NP, bool PRE>
int c5_parts_launch_np(const C5P& p, C5Parts& q, const void* image, const float* bias, void* ws, size_t ws_bytes, hipStream_t s) {
    int pxmax = 0;
    for (int i = 0; i < q.count; ++i) pxmax = q.PX[i] > pxmax ? q.PX[i] : pxmax;
    const size_t lds = (size_t)2 * pxmax * xrs<NP>() + 2 * R_MAX * sizeof(float);
    static unsigned long long attr_set = 0;
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv5_img_parts<MODE, NP, PRE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PX_MAX * xrs<NP>() + 2 * R_MAX * sizeof(float));
        ms_done_on_device(attr_set);
    }
    if (PRE) {
        size_t offP[MS_CONV_PARTS_MAX], offS[MS_CONV_PARTS_MAX], offM;
        const size_t need = c5_pre_bytes(p, q, offP, offS, &offM);
        if (!ws || ws_bytes < need || (((uintptr_t)ws) & 255)) return MS_ERR_WORKSPACE;
        int rows = 0;
        for (int i = 0; i < q.count; ++i) {
            q.P[i] = (u32x4*)((char*)ws + offP[i]);
            q.Pinv[i] = (float*)((char*)ws + offS[i]);
            rows += q.B[i];
        }
        float* pm = (float*)((char*)ws + offM);
        const dim3 pgrid((unsigned)rows, (unsigned)(p.CK / PB_CH));
        hipLaunchKernelGGL(k_conv5_rowmax, pgrid, dim3(256), 0, s, p, q, pm);
        MS_CHECK_LAUNCH();
        hipLaunchKernelGGL((k_conv5_presplit<MODE>), pgrid, dim3(256), 0, s, p, q, pm, p.slope);
        MS_CHECK_LAUNCH();
    }
    const dim3 grid((unsigned)q.bx0[q.count], (unsigned)(p.M / 64), 1);
    ms_note_kernel(NP == 2 ? 3 : 6, "k_conv5_img_parts<%d, %d, %s>", MODE, NP, PRE ? "true" : "false");
    hipLaunchKernelGGL((k_conv5_img_parts<MODE, NP, PRE>), grid, dim3(512), lds, s, p, q, (const u32x4*)image, bias);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

template <int MODE>
int c5_parts_launch(const C5P& p, C5Parts& q, const void* image, const float* bias, void* ws, size_t ws_bytes, hipStream_t s) {
    if (c5_np() == 3) return c5_parts_launch_np<MODE, 3, false>(p, q, image, bias, ws, ws_bytes, s);
    if (c5_pre_enabled() && p.CK % PB_CH == 0) return c5_parts_launch_np<MODE, 2, true>(p, q, image, bias, ws, ws_bytes, s);
    return c5_parts_launch_np<MODE, 2, false>(p, q, image, bias, ws, ws_bytes, s);
}

bool to_convp(const ms_conv1d_desc* d, ConvP* p) {
    if (!d || d->B <= 0 || d->Cin <= 0 || d->Lin <= 0 || d->Cout <= 0 || d->K <= 0 || d->stride <= 0 || d->pad < 0 || d->dil <= 0 ||
        d->groups <= 0)
        return false;
    p->B = d->B; p->Cin = d->Cin; p->Lin = d->Lin; p->Cout = d->Cout; p->K = d->K; p->stride = d->stride; p->pad = d->pad;
    p->dil = d->dil; p->groups = d->groups; p->Cg = d->Cin / d->groups; p->Og = d->Cout / d->groups;
    p->Lout = (d->Lin + 2 * d->pad - d->dil * (d->K - 1) - 1) / d->stride + 1;
    p->pad_mode = d->pad_mode; p->act = d->act; p->slope = d->slope; p->in_act = d->in_act;
    return true;
}

}  // namespace

// parts launches of the k5 layer (api.hip: ms_conv1d_parts_*); c: the layer with any B / Lin
bool ms5_parts_applicable(const ConvP& c, const ms_conv1d_parts* parts, bool backward) {
    C5P p;
    C5Parts q;
    return c5_enabled() && c5_parts_geometry(c, parts, backward, &p, &q);
}

size_t ms5_parts_ws(const ConvP& c, const ms_conv1d_parts* parts, bool backward) {
    C5P p;
    C5Parts q;
    if (!c5_parts_geometry(c, parts, backward, &p, &q) || !c5_pre_enabled() || p.CK % PB_CH) return 0;
    return c5_pre_bytes(p, q, nullptr, nullptr);
}

int ms5_parts_fwd(const ConvP& c, const ms_conv1d_parts* parts, const void* image, const float* bias, void* ws, size_t ws_bytes,
                  hipStream_t s) {
    C5P p;
    C5Parts q;
    if (!c5_parts_geometry(c, parts, false, &p, &q)) return MS_ERR_UNSUPPORTED;
    if (!image || (((uintptr_t)image) & 15) || (bias && (((uintptr_t)bias) & 15))) return MS_ERR_INVALID_ARG;
    for (int i = 0; i < q.count; ++i) {
        if (!parts->x[i] || !parts->y[i]) return MS_ERR_INVALID_ARG;
        q.X[i] = parts->x[i]; q.Xact[i] = nullptr; q.add[i] = nullptr; q.Y[i] = parts->y[i];
    }
    return c5_parts_launch<0>(p, q, image, bias, ws, ws_bytes, s);
}

int ms5_parts_bwd_data(const ConvP& c, const ms_conv1d_parts* parts, const void* image_bwd, void* ws, size_t ws_bytes,
                       hipStream_t s) {
    C5P p;
    C5Parts q;
    if (!c5_parts_geometry(c, parts, true, &p, &q)) return MS_ERR_UNSUPPORTED;
    if (!image_bwd || (((uintptr_t)image_bwd) & 15)) return MS_ERR_INVALID_ARG;
    for (int i = 0; i < q.count; ++i) {
        if (!parts->gy[i] || !parts->gx[i]) return MS_ERR_INVALID_ARG;
        q.X[i] = parts->gy[i]; q.Xact[i] = c.act == MS_ACT_NONE ? nullptr : parts->y_act[i]; q.add[i] = parts->gx_add[i];
        q.Y[i] = parts->gx[i];
    }
    return c5_parts_launch<1>(p, q, image_bwd, nullptr, ws, ws_bytes, s);
}

extern "C" {

size_t ms_conv1d_img_bytes(const ms_conv1d_desc* d) {
    ConvP c;
    C5P p;
    if (!to_convp(d, &c) || !c5_enabled() || !c5_geometry(c, false, &p)) return 0;
    return (size_t)(p.M / 32) * (p.CK / 16) * K5 * 3 * 1024;
}

size_t ms_conv1d_img_workspace_bytes(const ms_conv1d_desc* d, int which) {
    ConvP c;
    C5P p;
    if (!to_convp(d, &c) || !c5_geometry(c, which == 1, &p)) return 0;
    return p.nsplit > 1 ? (size_t)p.nsplit * p.zstride * sizeof(float) : 0;
}

int ms_conv1d_img_pack(const ms_conv1d_desc* d, const float* w, int backward, void* image, ms_stream_t stream) {
    ConvP c;
    C5P p;
    if (!to_convp(d, &c) || !w || !image || (((uintptr_t)image) & 15)) return MS_ERR_INVALID_ARG;
    if (!c5_geometry(c, backward != 0, &p)) return MS_ERR_UNSUPPORTED;
    const size_t total = (size_t)(p.M / 32) * (p.CK / 16) * K5 * 64;
    if (c5_np() == 2) {
        hipLaunchKernelGGL(k_conv5_wmax, dim3(W_NPART), dim3(256), 0, (hipStream_t)stream, w, (size_t)p.M * p.CK * K5,
                           reinterpret_cast<float*>((u32x4*)image + c5_tail_u4(p.M, p.CK)));
        MS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_conv5_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (u32x4*)image, p.M,
                       p.CK, backward ? 1 : 0, c5_np(), (const float*)nullptr);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_conv1d_img_pack2(const ms_conv1d_desc* d, const float* w, void* image_fwd, void* image_bwd, ms_stream_t stream) {
    ConvP c;
    C5P pf, pb;
    if (!to_convp(d, &c) || !w || !image_fwd || !image_bwd || (((uintptr_t)image_fwd) & 15) || (((uintptr_t)image_bwd) & 15))
        return MS_ERR_INVALID_ARG;
    if (!c5_geometry(c, false, &pf) || !c5_geometry(c, true, &pb)) return MS_ERR_UNSUPPORTED;
    const float* pm = nullptr;
    if (c5_np() == 2) {
        float* tail = reinterpret_cast<float*>((u32x4*)image_fwd + c5_tail_u4(pf.M, pf.CK));
        hipLaunchKernelGGL(k_conv5_wmax, dim3(W_NPART), dim3(256), 0, (hipStream_t)stream, w, (size_t)pf.M * pf.CK * K5, tail);
        MS_CHECK_LAUNCH();
        pm = tail;
    }
    const size_t tf = (size_t)(pf.M / 32) * (pf.CK / 16) * K5 * 64, tb = (size_t)(pb.M / 32) * (pb.CK / 16) * K5 * 64;
    hipLaunchKernelGGL(k_conv5_pack, dim3((unsigned)((tf + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (u32x4*)image_fwd, pf.M,
                       pf.CK, 0, c5_np(), pm);
    MS_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_conv5_pack, dim3((unsigned)((tb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (u32x4*)image_bwd, pb.M,
                       pb.CK, 1, c5_np(), pm);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_conv1d_img_fwd(const ms_conv1d_desc* d, const float* x, const void* image, const float* bias, float* y, void* workspace,
                      size_t workspace_bytes, ms_stream_t stream) {
    ConvP c;
    C5P p;
    if (!to_convp(d, &c) || !x || !image || !y || (((uintptr_t)image) & 15) || (bias && (((uintptr_t)bias) & 15)))
        return MS_ERR_INVALID_ARG;
    if (!c5_geometry(c, false, &p)) return MS_ERR_UNSUPPORTED;
    return c5_launch<0>(p, x, nullptr, image, bias, nullptr, y, workspace, workspace_bytes, (hipStream_t)stream);
}

int ms_conv1d_img_bwd_data(const ms_conv1d_desc* d, const float* gy, const float* y_act, const void* image_bwd,
                           const float* gx_add, float* gx, void* workspace, size_t workspace_bytes, ms_stream_t stream) {
    ConvP c;
    C5P p;
    if (!to_convp(d, &c) || !gy || !image_bwd || !gx || (((uintptr_t)image_bwd) & 15)) return MS_ERR_INVALID_ARG;
    if (!c5_geometry(c, true, &p)) return MS_ERR_UNSUPPORTED;
    if (c.act == MS_ACT_NONE) y_act = nullptr;
    return c5_launch<1>(p, gy, y_act, image_bwd, nullptr, gx_add, gx, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
