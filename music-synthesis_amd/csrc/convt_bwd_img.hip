// ConvTranspose1d BACKWARD DATA (kernel 2S, stride S, padding S/2; S = 2 or 8) on rows of <= 256 input positions, with
// pre-split weight images: the generator's two stride-8 upsampling layers (reference generator/full.py:27-32) and the six
// stride-2 line convolutions of the stage-1 generator (featuregenerator/upscale.py:85-97 through util/modules.py:
// HipConvTranspose2d).  They were the last dense layers on the fp32-input MFMA (conv_rows2.hip, 50-80 TFLOP/s).
//
//   gx[b, ci, l] = sum_co sum_k g'[b, co, S l + k - S/2] W[ci, co, k],      g' = gy * act'(y)
// Cut g' into blocks of S samples, G[(co, r)][u] = g'[co][S u + r]:
//   r <  S/2 ("low"):  blocks u = l,     l + 1   through taps k = r + S/2,  r + S/2 + S
//   r >= S/2 ("high"): blocks u = l - 1, l       through taps k = r - S/2,  r - S/2 + S
// so it is a GEMM with rows ci, columns (b, l), contraction (co, r) and TWO taps m = 0, 1 -- a 3-block window in which the
// low and the high half of a block look one block apart.  An MFMA k-step (16 contraction elements) is a chunk of 16 / S
// channels x S phases laid out [low 8 | high 8]: the lanes that supply the low half read block column l + m, the other
// lanes block column l + m - 1, from an LDS tile [block column][piece][low | high] with the usual 112-byte column stride.
// The gradient row is contiguous in exactly this order (8 consecutive samples = one block = low 4 | high 4 for S = 8;
// 4 channels x 4 samples transposed in registers for S = 2): no gather anywhere.
// Same recipe as conv5_img.hip: weights split once per step into fragment-ordered images streamed L2 -> registers a chunk
// ahead, LDS holds only the double-buffered gradient tile, a tile is 64 (TM = 1) or 128 (TM = 2) rows x R WHOLE batch rows
// with zero halo blocks, split-K slices fill the chip, a finish kernel sums the slabs in slice order.
// Arithmetic: exact 3-piece bf16 split, six products per multiply, fp32 accumulation (other summation order than the row
// kernels: agreement ~1e-7).
#include "ms_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int XRS = 112;                 // bytes per LDS block column: 3 pieces x 32 + 16
constexpr int PX_MAX = 350;              // LDS columns per buffer (39.2 KB; two buffers, two workgroups per CU)
constexpr unsigned OOB = 0xF0000000u;

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

// image[ms][chunk][tap m][piece][lane] (16 B): row ci = ms*32 + (lane & 31); the lane's 8 contraction elements are half
// hh = lane >> 5 of the chunk: element e -> channel chunk*(16/S) + e / (S/2), phase r = hh*S/2 + e % (S/2),
// tap k = r + S/2 - hh*S + S*m  (= r + S/2 + S m for the low half, r - S/2 + S m for the high half).  W is (Cin, Cout, 2S).
__global__ __launch_bounds__(256) void k_convt_bwd_pack(const float* __restrict__ W, u32x4* __restrict__ img, int Cin, int Cout, int S) {
    const int NCH = Cout * S / 16, HS = S / 2, K = 2 * S;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over ms x chunk x m x lane
    const size_t total = (size_t)(Cin / 32) * NCH * 2 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    size_t r = idx >> 6;
    const int m = (int)(r & 1); r >>= 1;
    const int chunk = (int)(r % NCH);
    const int ms = (int)(r / NCH);
    const int ci = ms * 32 + (lane & 31), hh = lane >> 5;
    unsigned pc[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int e = 2 * q + t;
            const int co = chunk * (16 / S) + e / HS, ph = hh * HS + e % HS;
            const int k = ph + HS - hh * S + S * m;
            v[t] = W[((size_t)ci * Cout + co) * K + k];
        }
        split_pair(v[0], v[1], pc[0][q], pc[1][q], pc[2][q]);
    }
    u32x4* dst = img + ((size_t)((ms * NCH + chunk) * 2 + m) * 3) * 64 + lane;
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) dst[pp * 64] = u32x4{pc[pp][0], pc[pp][1], pc[pp][2], pc[pp][3]};
}

struct CbP {
    int B, M, Cout, L;        // batch rows, GEMM rows (Cin), gradient channels, input positions per row (power of two, 4 .. 256)
    int lsh;                  // log2(L)
    int R, SS, PX;            // batch rows per tile, LDS columns per row (L + 2), LDS columns per buffer (R * SS)
    int nchunks, cps, nsplit; // contraction chunks (Cout * S / 16), chunks per split-K slice, slices
    int masked;               // multiply the gradient by act'(y) on load
    float slope;
    long long zstride;        // floats per slab
};

// S: stride; TM: 32-row sub-tiles per wave (workgroup = 2 x 4 waves: 64 TM rows x 256 columns)
template <int S, int TM>
__global__ __launch_bounds__(512, 2) void k_convt_bwd_img(CbP p, const float* __restrict__ GY, const float* __restrict__ YA,
                                                         const u32x4* __restrict__ IMG, float* __restrict__ GX,
                                                         float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cb[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int L = p.L, Lout = L * S, b0 = blockIdx.x * p.R, m0 = blockIdx.y * (64 * TM);
    const int cbeg = blockIdx.z * p.cps;
    const int cend = cbeg + p.cps < p.nchunks ? cbeg + p.cps : p.nchunks;
    const int nch = cend - cbeg;
    const int buf_bytes = p.PX * XRS;
    const bool masked = p.masked != 0;
    constexpr int CPC = 16 / S;                                  // gradient channels per chunk
    constexpr int NLD = S == 8 ? 2 : 1, NCC = S == 8 ? 1 : 4;    // staging rounds; channels a unit loads

    const unsigned g_bytes = 4u * (unsigned)(p.B * p.Cout * Lout);
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GY), 0, g_bytes, 0x00020000);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(masked ? YA : GY), 0, g_bytes, 0x00020000);
    const auto rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(IMG), 0, 0x80000000u, 0x00020000);

    // ---- staging units: one aligned 4-sample vector of the gradient row (of one channel for S = 8, of four for S = 2)
    //   S = 8: unit = (row r, channel c' of the chunk's 2, vector f of the 2 L in the row): block f >> 1, half f & 1
    //   S = 2: unit = (row r, channel quad q of the chunk's 2, vector f of the L / 2): blocks 2 f, 2 f + 1, both halves
    unsigned u_goff[NLD];
    int u_lds[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int u = tid + 512 * k;
        int r, cl, f;
        if (S == 8) { f = u & (2 * L - 1); cl = (u >> (p.lsh + 1)) & 1; r = u >> (p.lsh + 2); }
        else        { f = u & (L / 2 - 1); cl = ((u >> (p.lsh - 1)) & 1) * 4; r = u >> p.lsh; }
        const bool ok = r < p.R && b0 + r < p.B;
        u_goff[k] = ok ? 4u * (unsigned)(((b0 + r) * p.Cout + cl) * Lout + 4 * f) : OOB;       // + chunk * CPC * Lout * 4 (scalar)
        u_lds[k] = S == 8 ? (r * p.SS + 1 + (f >> 1)) * XRS + (f & 1) * 16 + cl * 8
                          : (r * p.SS + 1 + 2 * f) * XRS + (cl >> 2) * 8;
        if (!ok) u_lds[k] = -1;
    }
    f32x4 rg[NLD][NCC], ra[NLD][NCC];
    auto load_g = [&](int chunk, bool live) {
        const int so = live ? 4 * chunk * CPC * Lout : 0;
#pragma unroll
        for (int k = 0; k < NLD; ++k)
#pragma unroll
            for (int cc = 0; cc < NCC; ++cc) {
                rg[k][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, u_goff[k], so + cc * 4 * Lout, 0));
                ra[k][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, u_goff[k], so + cc * 4 * Lout, 0));
            }
    };
    auto store_g = [&](unsigned char* buf) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            if (u_lds[k] < 0) continue;
            if (S == 8) {
                float e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e[i] = rg[k][0][i];
                    if (masked) e[i] = ra[k][0][i] > 0.f ? e[i] : e[i] * p.slope;
                }
                uint2 o3[3];
                split_quad(e, o3);
                unsigned char* dst = buf + u_lds[k];
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<uint2*>(dst + pp * 32) = o3[pp];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {                      // sample 4 f + i: block 2 f + (i >> 1), half i & 1
                    float e[4];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        e[cc] = rg[k][cc][i];
                        if (masked) e[cc] = ra[k][cc][i] > 0.f ? e[cc] : e[cc] * p.slope;
                    }
                    uint2 o3[3];
                    split_quad(e, o3);
                    unsigned char* dst = buf + u_lds[k] + (i >> 1) * XRS + (i & 1) * 16;
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<uint2*>(dst + pp * 32) = o3[pp];
                }
            }
        }
    };

    // ---- A fragments: one chunk (2 taps x 3 pieces per row sub-tile) in registers, refilled tap by tap for the next chunk
    bf16x8 fa[TM][2][3];
    const int a_voff = lane * 16;
    const int a_row = (blockIdx.y * 2 * TM + wm * TM) * p.nchunks * (2 * 3 * 1024);          // byte offset of (first ms, chunk 0)
    auto load_a_tap = [&](int chunk, int t) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int so = a_row + (i * p.nchunks + chunk) * (2 * 3 * 1024) + t * 3 * 1024;
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
                fa[i][t][pp] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsI, a_voff, so + pp * 1024, 0));
        }
    };

    // ---- B fragment bases: MFMA column n = wn*64 + j*32 + l31 -> (row r, position l); the low-half lanes (h = 0) read block
    // column l + m, the high-half lanes block column l + m - 1 (LDS column of block u is r*SS + 1 + u)
    int bbase[2];
    unsigned o_lane[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = wn * 64 + j * 32 + l31;
        const int r = n >> p.lsh, l = n & (L - 1);
        const bool ok = r < p.R && b0 + r < p.B;
        bbase[j] = ok ? (r * p.SS + 1 + l - h) * XRS + h * 16 : h * 16;
        o_lane[j] = ok ? 4u * (unsigned)((r * p.M + 4 * h) * L + l) : OOB;        // + ((b0*M + row) * L) * 4 (scalar)
    }

    // ---- zero both LDS buffers (the halo blocks are never written), stage the first chunk, prefetch the second
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int i = tid * 16; i < 2 * buf_bytes; i += 512 * 16) *reinterpret_cast<u32x4*>(smem_cb + i) = z;
    }
    load_g(cbeg, true);
#pragma unroll
    for (int t = 0; t < 2; ++t) load_a_tap(cbeg, t);
    __syncthreads();
    store_g(smem_cb);
    load_g(cbeg + 1, nch > 1);
    __syncthreads();

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
        const unsigned char* Xs = smem_cb + (ch & 1) * buf_bytes;
        unsigned char* Xn = smem_cb + ((ch & 1) ^ 1) * buf_bytes;
        const bool more = ch + 1 < nch;
        const int a_next = cbeg + (more ? ch + 1 : ch);
        bf16x8 fb[2][2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
                    fb[t][j][pp] = *reinterpret_cast<const bf16x8*>(Xs + bbase[j] + t * XRS + pp * 32);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s = 0; s < 6; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][t][PA[s]], fb[t][j][PB[s]], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_a_tap(a_next, t);                                     // this tap's registers are free: the next chunk's tap
            if (t == 0 && more) store_g(Xn);                           // chunk ch+1: registers -> the other buffer
            if (t == 1) load_g(cbeg + ch + 2, ch + 2 < nch);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue: dword stores straight from the accumulators (32 lanes = 32 consecutive positions of one or more rows)
    int L4;
    asm volatile("s_mov_b32 %0, %1" : "=s"(L4) : "s"(4 * L));
    float* out = p.nsplit == 1 ? GX : slabs + (size_t)blockIdx.z * p.zstride;
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x80000000u, 0x00020000);
    const int base = b0 * p.M * L4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int chs = m0 + (wm * TM + i) * 32 + 8 * g;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = acc[i][j][4 * g + q];      // (a copy: bit-casting the vector element expression itself stores element 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsO, o_lane[j], base + (chs + q) * L4, 0);
                }
            }
}

// gx = sum_z slab_z, slabs summed in slice order
__global__ __launch_bounds__(256) void k_convt_bwd_finish(const f32x4* __restrict__ slabs, int ns, long long zstride4,
                                                         f32x4* __restrict__ GX, long long total4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        f32x4 v = slabs[i];
        for (int z = 1; z < ns; ++z) v += slabs[(long long)z * zstride4 + i];
        GX[i] = v;
    }
}

bool cb_enabled() {
    const char* sw = getenv("MSYNTH_CONVTBWDIMG");            // tuning / test switch (0: the fp32 row-tile kernels)
    return !(sw && atoi(sw) == 0);
}

bool cb_geometry(const ms_convt1d_desc* d, CbP* p, int* tm) {
    if (!d || d->B <= 0 || d->Lin <= 0 || d->Cin <= 0 || d->Cout <= 0) return false;
    const int S = d->stride;
    if (!((S == 8 && d->K == 16 && d->pad == 4) || (S == 2 && d->K == 4 && d->pad == 1))) return false;
    if (d->in_act != MS_ACT_NONE || (d->act != MS_ACT_NONE && d->act != MS_ACT_LRELU)) return false;
    const int L = d->Lin;
    if (L < 4 || L > 256 || (L & (L - 1))) return false;                  // whole rows per tile, shifts for the index maths
    if (d->Cin % 64 || (d->Cout * S) % 16 || d->Cout < 16 / S) return false;
    if ((long long)d->B * L < 512) return false;                          // (B = 1 inference: nothing to fill the chip with)
    if ((long long)d->B * d->Cout * L * S * 4 >= (1ll << 31) || (long long)d->B * d->Cin * L * 4 >= (1ll << 31)) return false;
    if ((long long)(d->Cin / 32) * (d->Cout * S / 16) * (2 * 3 * 1024) >= (1ll << 31)) return false;
    *tm = d->Cin % 128 == 0 ? 2 : 1;
    p->B = d->B; p->M = d->Cin; p->Cout = d->Cout; p->L = L;
    int lsh = 0;
    while ((1 << lsh) < L) ++lsh;
    p->lsh = lsh;
    p->SS = L + 2;
    int R = 256 / L;
    if (R * p->SS > PX_MAX) R = PX_MAX / p->SS;
    p->R = R;
    p->PX = R * p->SS;
    p->nchunks = d->Cout * S / 16;
    p->masked = d->act == MS_ACT_LRELU ? 1 : 0;
    p->slope = d->slope;
    // split-K: tiles x slices ~ the 512 resident workgroups, at least 8 chunks per slice, at most 16 slices / 64 MiB of slabs
    const int tiles = (d->Cin / (64 * *tm)) * ((d->B + R - 1) / R);
    int ns = 512 / tiles;
    if (ns < 1) ns = 1;
    if (ns > 16) ns = 16;
    while (ns > 1 && p->nchunks / ns < 8) --ns;
    p->zstride = (long long)d->B * d->Cin * L;
    while (ns > 1 && (size_t)ns * p->zstride * 4 > ((size_t)64 << 20)) --ns;
    p->cps = (p->nchunks + ns - 1) / ns;
    p->nsplit = (p->nchunks + p->cps - 1) / p->cps;
    return true;
}

template <int S, int TM>
int cb_launch(const CbP& p, const float* gy, const float* ya, const void* image, float* gx, void* ws, size_t ws_bytes, hipStream_t s) {
    float* slabs = nullptr;
    if (p.nsplit > 1) {
        const size_t need = (size_t)p.nsplit * p.zstride * sizeof(float);
        if (!ws || ws_bytes < need || (((uintptr_t)ws) & 15)) return MS_ERR_WORKSPACE;
        slabs = (float*)ws;
    }
    const size_t lds = (size_t)2 * p.PX * XRS;
    static unsigned long long attr_set = 0;
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_convt_bwd_img<S, TM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * PX_MAX * XRS);
        ms_done_on_device(attr_set);
    }
    const dim3 grid((unsigned)((p.B + p.R - 1) / p.R), (unsigned)(p.M / (64 * TM)), (unsigned)p.nsplit);
    ms_note_kernel(6, "k_convt_bwd_img<%d, %d>", S, TM);
    hipLaunchKernelGGL((k_convt_bwd_img<S, TM>), grid, dim3(512), lds, s, p, gy, ya, (const u32x4*)image, gx, slabs);
    MS_CHECK_LAUNCH();
    if (p.nsplit > 1) {
        const long long total4 = p.zstride / 4;
        unsigned nb = (unsigned)((total4 + 255) / 256);
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(k_convt_bwd_finish, dim3(nb), dim3(256), 0, s, (const f32x4*)slabs, p.nsplit, p.zstride / 4, (f32x4*)gx, total4);
        MS_CHECK_LAUNCH();
    }
    return MS_OK;
}

}  // namespace

extern "C" {

size_t ms_convt1d_bwd_img_bytes(const ms_convt1d_desc* d) {
    CbP p;
    int tm;
    if (!cb_enabled() || !cb_geometry(d, &p, &tm)) return 0;
    return (size_t)(d->Cin / 32) * p.nchunks * 2 * 3 * 1024;
}

size_t ms_convt1d_bwd_img_workspace_bytes(const ms_convt1d_desc* d) {
    CbP p;
    int tm;
    if (!cb_geometry(d, &p, &tm)) return 0;
    return p.nsplit > 1 ? (size_t)p.nsplit * p.zstride * sizeof(float) : 0;
}

int ms_convt1d_bwd_img_pack(const ms_convt1d_desc* d, const float* w, void* image, ms_stream_t stream) {
    CbP p;
    int tm;
    if (!d || !w || !image || (((uintptr_t)image) & 15)) return MS_ERR_INVALID_ARG;
    if (!cb_geometry(d, &p, &tm)) return MS_ERR_UNSUPPORTED;
    const size_t total = (size_t)(d->Cin / 32) * p.nchunks * 2 * 64;
    hipLaunchKernelGGL(k_convt_bwd_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (u32x4*)image,
                       d->Cin, d->Cout, d->stride);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_convt1d_bwd_img_data(const ms_convt1d_desc* d, const float* gy, const float* y_act, const void* image, float* gx,
                            void* workspace, size_t workspace_bytes, ms_stream_t stream) {
    CbP p;
    int tm;
    if (!d || !gy || !image || !gx || (((uintptr_t)image) & 15) || (((uintptr_t)gx) & 15)) return MS_ERR_INVALID_ARG;
    if (!cb_geometry(d, &p, &tm)) return MS_ERR_UNSUPPORTED;
    if (p.masked && !y_act) return MS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (d->stride == 8)
        return tm == 2 ? cb_launch<8, 2>(p, gy, y_act, image, gx, workspace, workspace_bytes, s)
                       : cb_launch<8, 1>(p, gy, y_act, image, gx, workspace, workspace_bytes, s);
    return tm == 2 ? cb_launch<2, 2>(p, gy, y_act, image, gx, workspace, workspace_bytes, s)
                   : cb_launch<2, 1>(p, gy, y_act, image, gx, workspace, workspace_bytes, s);
}

}  // extern "C"
