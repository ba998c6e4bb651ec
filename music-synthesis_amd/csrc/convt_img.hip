// ConvTranspose1d forward (kernel 2S, stride S, padding S/2; S = 2 or 8) with pre-split weight images and an
// LDS-resident input window: the four upsampling layers of the generator (reference generator/full.py:27-40).
//
//   y[b, co, q S + r] = bias[co] + sum_ci sum_d W[ci, co, r + S/2 - d S] x[b, ci, q + d],   d in {-1, 0} (r < S/2), {0, +1} (else)
// is a GEMM with Cout * S rows (co, r), the input positions q as columns and (ci, live tap) as the contraction: every phase
// has exactly TWO live taps of the 3-column window.  Rows are ordered in sub-tiles of 32 = one phase half (low / high) of
// 64 / S channels, so that a sub-tile's rows share their tap pair; a wave owns the low and the high sub-tile of one channel
// group and stores their accumulators together: 4 consecutive output samples per lane (S = 8) or 2 (S = 2).
// Same recipe as atom_fused.hip: the input window of the tile (all input channels, NTP + 2 columns) is split once into LDS,
// there is no staging and no barrier inside the K loop, the weights are split ONCE per pass (k_convt_pack) into
// fragment-ordered images that stream from L2 into registers one chunk ahead, workgroups are persistent and prefetch the
// next tile's window under the GEMM, results leave through buffer stores with scalar row offsets.  Unlike the atom there is
// no halo to recompute: a tile of NTP input columns yields NTP * S output samples.
// Arithmetic (r04): block-scaled two-piece fp16 operands, three products per multiply into one fp32 accumulator
// (atom_fused.hip): the tile's input window is scaled by the power of two that puts its largest magnitude at 2^14 (the
// maximum of the NEXT window is published under the current tile, as in the atom kernel), the weights are packed as
// pieces of S_w w (S_w: a power of two from the largest weight).  The short-row K-loop kernel (convt_fwd_short.hip) shares the pack kernel and keeps the exact
// three-piece bf16 split (np = 3 images): agreement between the schemes ~3e-7.
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
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int XRS = 80;                  // bytes per LDS column of a 16-channel chunk: 2 fp16 pieces x 32 + 16
constexpr unsigned OOB = 0xF0000000u;
// NP = 2: the weights are packed as the fp16 pieces of S_w w, S_w a power of two from the tensor's largest magnitude (it goes to
// [2^12, 2^13): weights of any magnitude; see conv5_img.hip): partial maxima -> image tail -> the pack reduces them and leaves
// 1 / S_w there for the kernel.
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
// end of the NP = 2 data of an image (the allocation is sized for three pieces), in 16-byte units
__host__ __device__ inline size_t ct_tail_u4(int Cin, int Cout, int S) { return (size_t)(Cout * S / 32) * (Cin / 16) * 2 * 2 * 64; }

__global__ __launch_bounds__(256) void k_convt_wmax(const float* __restrict__ W, size_t n, float* __restrict__ pm) {
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

// (a, b), scaled into fp16's range by the caller -> a = h.lo + l.lo to 22 significand bits (atom_fused.hip)
__device__ __forceinline__ void split_pair2(float a, float b, unsigned& h, unsigned& l) {
    const f32x2 v = {a, b};
    const f16x2 hi = __builtin_convertvector(v, f16x2);
    const f16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x2), f16x2);
    h = __builtin_bit_cast(unsigned, hi);
    l = __builtin_bit_cast(unsigned, lo);
}

__device__ __forceinline__ void block_scale(float m, float& S, float& invS) {
    const unsigned eb = (__builtin_bit_cast(unsigned, m) >> 23) & 0xFFu;
    const bool ok = eb >= 16u && eb <= 250u;
    S = ok ? __builtin_bit_cast(float, (268u - eb) << 23) : 1.f;
    invS = ok ? __builtin_bit_cast(float, (eb - 14u) << 23) : 1.f;
}

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

// image[cg][half][chunk][tap jj][piece][lane] (16 B): sub-tile (cg, half) row i = lane & 31 is
//   S = 8: channel cg*8 + i/4, phase half*4 + i%4;      S = 2: channel cg*32 + i, phase half
// contraction channels chunk*16 + 8*(lane >> 5) + 0..7, tap jj -> window offset d = half + jj - 1:
//   A = W[ci][co][phase + S/2 - d*S]          (W is (Cin, Cout, 2S))
__global__ __launch_bounds__(256) void k_convt_pack(const float* __restrict__ W, u32x4* __restrict__ img, int Cin, int Cout, int S,
                                                   int np) {
    const int NC = Cin / 16, NCG = Cout * S / 64;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over cg x half x chunk x jj x lane
    const size_t total = (size_t)NCG * 2 * NC * 2 * 64;
    float WS = 1.f, iWS = 1.f;
    if (np == 2) {
        float* tail = reinterpret_cast<float*>(img + ct_tail_u4(Cin, Cout, S));
        weight_scale(tail, WS, iWS);
        if (idx == 0) tail[W_NPART] = iWS;
    }
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    size_t r = idx >> 6;
    const int jj = (int)(r & 1); r >>= 1;
    const int chunk = (int)(r % NC); r /= NC;
    const int half = (int)(r & 1);
    const int cg = (int)(r >> 1);
    const int i = lane & 31;
    const int co = S == 8 ? cg * 8 + i / 4 : cg * 32 + i;
    const int phase = S == 8 ? half * 4 + i % 4 : half;
    const int k = phase + S / 2 - (half + jj - 1) * S;
    const int ci0 = chunk * 16 + 8 * (lane >> 5);
    const int K = 2 * S;
    unsigned pc[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = W[((size_t)(ci0 + 2 * q) * Cout + co) * K + k];
        const float b = W[((size_t)(ci0 + 2 * q + 1) * Cout + co) * K + k];
        if (np == 3) split_pair(a, b, pc[0][q], pc[1][q], pc[2][q]);
        else { split_pair2(a * WS, b * WS, pc[0][q], pc[1][q]); pc[2][q] = 0u; }
    }
    u32x4* dst = img + ((size_t)((((cg * 2 + half) * NC + chunk) * 2 + jj) * np)) * 64 + lane;
    for (int pp = 0; pp < np; ++pp) dst[pp * 64] = u32x4{pc[pp][0], pc[pp][1], pc[pp][2], pc[pp][3]};
}

struct CtP {
    int B, Cin, Cout, L;            // L = input length
    int tiles_per_row, mtiles;      // column tiles per batch row, channel-group tiles
    int act;
    float slope;
};

// CIN input channels; S stride; NTP input columns per tile; WGM x WGN = 4 waves: a wave owns one channel group (both phase
// halves: TM = 2 sub-tiles) x TN column sub-tiles
template <int CIN, int S, int NTP, int WGM>
__global__ __launch_bounds__(256, 1) void k_convt_img(CtP p, const float* __restrict__ X, const u32x4* __restrict__ IMG,
                                                     const float* __restrict__ bias, float* __restrict__ Y) {
    constexpr int NC = CIN / 16, WGN = 4 / WGM, TN = NTP / 32 / WGN, NT = 256;
    constexpr int NXA = NTP + 8;                      // LDS columns per chunk: window NTP + 2, vectors reach up to 3 further
    constexpr int XCS = NXA * XRS;
    constexpr int NV = (NTP + 2 + 3 + 3) / 4, NV16 = (NV + 3) / 4;       // window start sits 3 samples behind an aligned vector
    constexpr int ROUNDS = (NC * NV16 * 16 + NT - 1) / NT;
    constexpr int CPG = 64 / S;                       // channels per group
    static_assert(TN >= 1 && TN * WGN * 32 == NTP, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ct[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const int L = p.L;
    const int ntiles = p.B * p.tiles_per_row * p.mtiles;
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0x80000000u, 0x00020000);
    const auto rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(IMG), 0, 0x80000000u, 0x00020000);
    const float winv = reinterpret_cast<const float*>(IMG + ct_tail_u4(p.Cin, p.Cout, S))[W_NPART];      // 1 / S_w (k_convt_pack)
    const auto rsY = __builtin_amdgcn_make_buffer_rsrc(Y, 0, 0x80000000u, 0x00020000);

    // ---- staging units (tile-invariant): 4 channels x one aligned 4-sample vector; window column c <-> position q0 - 1 + c,
    // q0 a multiple of 32: the window start sits 3 samples behind the aligned vector at q0 - 4
    int u_goff[ROUNDS], u_t[ROUNDS], u_lcol[ROUNDS], u_lbase[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int u = tid + NT * r;
        const int grp = u >> 4, chunk = grp / NV16, vg = grp - chunk * NV16;
        const int cq = (u >> 2) & 3, v = vg * 4 + (u & 3);
        const bool in = chunk < NC && v < NV;
        u_t[r] = in ? 4 * v - 4 : (1 << 28);                              // position of the vector relative to q0
        u_goff[r] = 4 * ((chunk * 16 + 4 * cq) * L + (4 * v - 4));
        u_lcol[r] = in ? 4 * v - 3 : -1000;
        u_lbase[r] = chunk * XCS + cq * 8;
    }
    f32x4 rx[ROUNDS][4];
    auto tile_of = [&](int t, int& b, int& q0, int& mt) {
        mt = t % p.mtiles;
        const int nt = t / p.mtiles;
        b = nt / p.tiles_per_row;
        q0 = (nt - b * p.tiles_per_row) * NTP;
    };
    auto load_x = [&](int t) {
        int bb, qq, mt;
        tile_of(t, bb, qq, mt);
        const int base = 4 * bb * CIN * L;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int pos = qq + u_t[r];
            const unsigned goff = (pos >= 0 && pos < L) ? (unsigned)(u_goff[r] + 4 * qq) : OOB;      // (>= 0 where pos >= 0)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                rx[r][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, goff, base + cc * 4 * L, 0));
        }
    };
    float* smax = reinterpret_cast<float*>(smem_ct + NC * XCS);          // [4]: per-wave |max| of the NEXT tile's window
    auto publish_window_max = [&]() {
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(rx[r][cc][e]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) smax[wid] = m;
    };
    auto store_x = [&](float SC_) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = u_lcol[r] + e;
                if (i < 0 || i >= NXA) continue;
                unsigned h0, l0, h1, l1;
                split_pair2(rx[r][0][e] * SC_, rx[r][1][e] * SC_, h0, l0);
                split_pair2(rx[r][2][e] * SC_, rx[r][3][e] * SC_, h1, l1);
                unsigned char* dst = smem_ct + u_lbase[r] + i * XRS;
                *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(dst + 32) = make_uint2(l0, l1);
            }
        }
    };

    // ---- A fragments of one chunk: [half][tap jj][piece]; one chunk ahead
    u32x4 fa[2][2][2][2];
    const int a_voff = lane * 16;
    auto load_a = [&](int cg, int chunk, u32x4 (&dst)[2][2][2]) {
        const int so = ((cg * 2) * NC + chunk) * (2 * 2 * 1024);           // (cg, half 0, chunk); half 1 is NC chunks further
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
                    dst[hf][jj][pp] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rsI, a_voff, so + hf * NC * (2 * 2 * 1024) + (jj * 2 + pp) * 1024, 0));
    };

    int tile = blockIdx.x;
    if (tile < ntiles) load_x(tile);
    publish_window_max();                             // (waits for the first window)
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        int b, q0, mt;
        tile_of(tile, b, q0, mt);
        const int cg = mt * WGM + wm;
        load_a(cg, 0, fa[0]);
        float SCL, iS;
        block_scale(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])), SCL, iS);
        store_x(SCL);
        const int nxt = tile + gridDim.x;
        if (nxt < ntiles) load_x(nxt);
        f32x16 acc[2][TN];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[hf][j][r] = 0.f;
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                              // window staged
        const unsigned char* Bs = smem_ct + ((wn * TN) * 32 + l31) * XRS + h * 16;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            if (ch + 1 < NC) load_a(cg, ch + 1, fa[(ch + 1) & 1]);
            // window column offsets 0, 1, 2 (low half: taps at 0, 1; high half: 1, 2)
            u32x4 fb[3][TN][2];
#pragma unroll
            for (int o = 0; o < 3; ++o)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
                        fb[o][j][pp] = *reinterpret_cast<const u32x4*>(Bs + ch * XCS + (j * 32 + o) * XRS + pp * 32);
            __builtin_amdgcn_sched_barrier(0);
            // three products per multiply, smallest first: a_h b_l, a_l b_h, a_h b_h
            constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[hf][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[ch & 1][hf][jj][PA[t]]),
                                                                               __builtin_bit_cast(f16x8, fb[hf + jj][j][PB[t]]),
                                                                               acc[hf][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue: bias + activation, the two halves of a channel group stored together
        int LS4;                                       // 4 * L * S (bytes per output channel row), opaque: see atom_fused.hip
        asm volatile("s_mov_b32 %0, %1" : "=s"(LS4) : "s"(4 * L * S));
        const int obase = b * p.Cout * LS4;
        const float kscale = iS * winv;               // undoes the window's and the weights' scales
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = (wn * TN + j) * 32 + l31, q = q0 + n;
            const bool ok = q < L;
            if (S == 8) {
                // row i = 4h + (r & 3) + 8g: channel cg*8 + h + 2g, phase (half) * 4 + (r & 3): 4 consecutive samples per (half, g)
                const unsigned ol = ok ? 4u * (unsigned)(q * 8 + h * (L * 8)) : OOB;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cs = cg * CPG + 2 * g;                       // + h in the lane part
                    const float bv = bias ? bias[cs + h] : 0.f;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = fmaf(acc[hf][j][4 * g + e], kscale, bv);
                            if (p.act == MS_ACT_LRELU) t = t > 0.f ? t : t * p.slope;
                            v[e] = t;
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsY, ol, obase + cs * LS4 + hf * 16, 0);
                    }
                }
            } else {
                // S = 2: row i = 4h + (r & 3) + 8g: channel cg*32 + i, phase = half: (low, high) -> samples 2q, 2q + 1
                const unsigned ol = ok ? 4u * (unsigned)(q * 2 + 4 * h * (L * 2)) : OOB;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cs = cg * CPG + (r & 3) + 8 * (r >> 2);     // + 4h in the lane part
                    const float bv = bias ? bias[cs + 4 * h] : 0.f;
                    f32x2 v;
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        float t = fmaf(acc[hf][j][r], kscale, bv);
                        if (p.act == MS_ACT_LRELU) t = t > 0.f ? t : t * p.slope;
                        v[hf] = t;
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rsY, ol, obase + cs * LS4, 0);
                }
            }
        }
        if (nxt < ntiles) publish_window_max();        // (waits for the next tile's window: issued before the GEMM)
        __syncthreads();                              // the window is dead: the next one may overwrite it
    }
}

template <int CIN, int S, int NTP, int WGM>
int launch_ct(CtP p, const float* x, const void* image, const float* bias, float* y, hipStream_t s) {
    constexpr int NC = CIN / 16;
    const size_t lds = (size_t)NC * (NTP + 8) * XRS + 4 * sizeof(float);      // window + the scale exchange
    if (lds > 158 * 1024) return MS_ERR_UNSUPPORTED;
    p.tiles_per_row = (p.L + NTP - 1) / NTP;
    const int ncg = p.Cout * S / 64;
    if (ncg % WGM) return MS_ERR_UNSUPPORTED;
    p.mtiles = ncg / WGM;
    const void* fn = reinterpret_cast<const void*>(&k_convt_img<CIN, S, NTP, WGM>);
    static int wgs_per_cu[64] = {}, n_cu[64] = {};                   // per device (ms_common.h: one-time launch setup)
    const int dev = ms_current_device();
    if (!__atomic_load_n(&wgs_per_cu[dev], __ATOMIC_ACQUIRE)) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        int nb = 0;
        hipDeviceProp_t prop;
        (void)hipGetDeviceProperties(&prop, dev);
        n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, lds) != hipSuccess || nb < 1) nb = 1;
        __atomic_store_n(&wgs_per_cu[dev], nb, __ATOMIC_RELEASE);
    }
    const long long ntiles = (long long)p.B * p.tiles_per_row * p.mtiles;
    const long long slots = (long long)n_cu[dev] * wgs_per_cu[dev];
    const dim3 grid((unsigned)(ntiles < slots ? ntiles : slots));
    ms_note_kernel(3, "k_convt_img<%d, %d, %d, %d>", CIN, S, NTP, WGM);
    hipLaunchKernelGGL((k_convt_img<CIN, S, NTP, WGM>), grid, dim3(256), lds, s, p, x, (const u32x4*)image, bias, y);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

bool ct_ok(const ms_convt1d_desc* d) {
    if (!d || d->B <= 0 || d->Lin <= 0) return false;
    if (!((d->stride == 8 && d->K == 16 && d->pad == 4) || (d->stride == 2 && d->K == 4 && d->pad == 1))) return false;
    if (d->in_act != MS_ACT_NONE || (d->act != MS_ACT_NONE && d->act != MS_ACT_LRELU)) return false;
    if (d->Lin % 4) return false;
    const bool shape = (d->Cin == 512 && d->Cout == 256 && d->stride == 8) || (d->Cin == 256 && d->Cout == 128 && d->stride == 8) ||
                       (d->Cin == 128 && d->Cout == 64 && d->stride == 2) || (d->Cin == 64 && d->Cout == 32 && d->stride == 2);
    if (!shape) return false;
    // measured (tools/scratch/microbench_convt_img.py, B = 32): stride 8: 69 -> 37 us (512 -> 256), 77 -> 64 us (256 -> 128); the two
    // stride-2 layers are HBM-bound and run 20 % FASTER on the paired row kernel (42 / 28 us), and at B = 1 the pack launch
    // costs more than the kernel saves: those stay on ms_convt1d_fwd's row-tile path
    // (r04, fp16 x 2 images, B = 32: stride 2 on this kernel 37.7 / 30.8 us against 40.0 / 28.0 on the row kernel -- a wash)
    if (d->stride != 8 || (long long)d->B * d->Lin < 1024) return false;
    if ((long long)d->B * d->Cin * d->Lin * 4 >= (1ll << 31) || (long long)d->B * d->Cout * d->Lin * d->stride * 4 >= (1ll << 31)) return false;
    const char* sw = getenv("MSYNTH_CONVTIMG");                 // tuning / test switch (0: the row-tile kernels)
    return !(sw && atoi(sw) == 0);
}

}  // namespace

// short rows, many channels (stride 2): the K-loop kernel of convt_fwd_short.hip on the same image
bool msct_short_ok(const ms_convt1d_desc* d);
size_t msct_short_ws(const ms_convt1d_desc* d);
int msct_short_fwd(const ms_convt1d_desc* d, const float* x, const void* image, const float* bias, float* y, void* ws,
                   size_t ws_bytes, hipStream_t s);

extern "C" {

size_t ms_convt1d_img_bytes(const ms_convt1d_desc* d) {
    if (!ct_ok(d) && !msct_short_ok(d)) return 0;
    return (size_t)(d->Cout * d->stride / 32) * (d->Cin / 16) * 2 * 3 * 1024;      // (the larger of the two piece schemes)
}

size_t ms_convt1d_img_workspace_bytes(const ms_convt1d_desc* d) { return msct_short_ok(d) ? msct_short_ws(d) : 0; }

int ms_convt1d_img_pack(const ms_convt1d_desc* d, const float* w, void* image, ms_stream_t stream) {
    if (!d || !w || !image || (((uintptr_t)image) & 15)) return MS_ERR_INVALID_ARG;
    if (!ct_ok(d) && !msct_short_ok(d)) return MS_ERR_UNSUPPORTED;
    const size_t total = (size_t)(d->Cout * d->stride / 64) * 2 * (d->Cin / 16) * 2 * 64;
    // the short-row K-loop kernel reads three bf16 pieces, the LDS-resident-window kernel two fp16 pieces of S_w w
    if (!msct_short_ok(d)) {
        hipLaunchKernelGGL(k_convt_wmax, dim3(W_NPART), dim3(256), 0, (hipStream_t)stream, w, (size_t)d->Cin * d->Cout * d->K,
                           reinterpret_cast<float*>((u32x4*)image + ct_tail_u4(d->Cin, d->Cout, d->stride)));
        MS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_convt_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (u32x4*)image,
                       d->Cin, d->Cout, d->stride, msct_short_ok(d) ? 3 : 2);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_convt1d_img_fwd(const ms_convt1d_desc* d, const float* x, const void* image, const float* bias, float* y,
                       void* workspace, size_t workspace_bytes, ms_stream_t stream) {
    if (!d || !x || !image || !y || (((uintptr_t)image) & 15)) return MS_ERR_INVALID_ARG;
    if (msct_short_ok(d)) return msct_short_fwd(d, x, image, bias, y, workspace, workspace_bytes, (hipStream_t)stream);
    if (!ct_ok(d)) return MS_ERR_UNSUPPORTED;
    CtP p;
    p.B = d->B; p.Cin = d->Cin; p.Cout = d->Cout; p.L = d->Lin; p.act = d->act; p.slope = d->slope;
    hipStream_t s = (hipStream_t)stream;
    if (d->Cin == 512) return launch_ct<512, 8, 32, 4>(p, x, image, bias, y, s);
    if (d->Cin == 256) return launch_ct<256, 8, 64, 4>(p, x, image, bias, y, s);
    if (d->Cin == 128) return launch_ct<128, 2, 64, 2>(p, x, image, bias, y, s);
    return launch_ct<64, 2, 128, 1>(p, x, image, bias, y, s);
}

}  // extern "C"
