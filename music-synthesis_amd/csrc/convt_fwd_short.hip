// ConvTranspose1d FORWARD (kernel 4, stride 2, padding 1) on SHORT rows (4 .. 16 input positions) with many channels: the first
// line convolutions of the stage-1 generator (reference featuregenerator/upscale.py:85-91 through util/modules.py:
// HipConvTranspose2d), (rows, channels) = (128, 2048 -> 512), (256, 1024 -> 256), (512, 512 -> 128).
// They ran on three different generic row kernels at 44-70 TFLOP/s (conv_mfma.hip / conv_rows2.hip / conv_rows3.hip).
//
// Same arithmetic and the SAME weight image as convt_img.hip (k_convt_pack: rows in sub-tiles of 32 = one phase half of 32
// channels, two live taps of the 3-column window per half), but as a K-loop kernel in the manner of conv5_img.hip /
// convt_bwd_img.hip, because here the contraction (up to 2048 input channels) does not fit in LDS while a whole tile of
// positions does: 128 GEMM rows (= 64 output channels x 2 phases; a wave owns the low and the high sub-tile of 32 channels) x
// 256 columns = R whole batch rows with a zero halo column on either side; per 16-channel chunk the activation tile is split
// into LDS (double-buffered, one barrier per chunk), the A fragments stream L2 -> registers a chunk ahead; split-K slices over
// the input channels fill the chip, a finish kernel sums the slabs in slice order and applies bias + LeakyReLU.
// Epilogue: a lane's low / high accumulators of one channel are output samples 2 q and 2 q + 1: 8-byte stores.
#include "ms_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int XRS = 112;
constexpr int PX_MAX = 350;
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

struct CsP {
    int B, Cin, Cout, L;      // batch rows, input channels (contraction), output channels, input positions per row (power of two)
    int lsh;
    int R, SS, PX;            // batch rows per tile, LDS columns per row (L + 2), LDS columns per buffer
    int nchunks, cps, nsplit; // 16-channel chunks, chunks per split-K slice, slices
    int act;
    float slope;
    long long zstride;        // floats per slab (B * Cout * 2 L)
};

__global__ __launch_bounds__(512, 2) void k_convt_fwd_short(CsP p, const float* __restrict__ X, const u32x4* __restrict__ IMG,
                                                           const float* __restrict__ bias, float* __restrict__ Y,
                                                           float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cs[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;                   // 2 channel groups x 4 column blocks of 64
    const int L = p.L, b0 = blockIdx.x * p.R;
    const int cg = blockIdx.y * 2 + wm;                      // channel group of 32 output channels (both phase halves)
    const int cbeg = blockIdx.z * p.cps;
    const int cend = cbeg + p.cps < p.nchunks ? cbeg + p.cps : p.nchunks;
    const int nch = cend - cbeg;
    const int buf_bytes = p.PX * XRS;
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 4u * (unsigned)(p.B * p.Cin * L), 0x00020000);
    const auto rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(IMG), 0, 0x80000000u, 0x00020000);

    // ---- staging units: (batch row r, channel quad cq of the chunk's 4, 4-position vector v of the row's L / 4)
    const int nv = L >> 2, vsh = p.lsh - 2;                   // vectors per row, log2
    unsigned u_goff[2];
    int u_lds[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + 512 * k;
        const int v = u & (nv - 1), cq = (u >> vsh) & 3, r = u >> (vsh + 2);
        const bool ok = r < p.R && b0 + r < p.B;
        u_goff[k] = ok ? 4u * (unsigned)(((b0 + r) * p.Cin + 4 * cq) * L + 4 * v) : OOB;       // + chunk * 16 * L * 4 (scalar)
        u_lds[k] = ok ? (r * p.SS + 1 + 4 * v) * XRS + cq * 8 : -1;
    }
    f32x4 rx[2][4];
    auto load_x = [&](int chunk, bool live) {
        const int so = live ? 4 * chunk * 16 * L : 0;
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                rx[k][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, u_goff[k], so + cc * 4 * L, 0));
    };
    auto store_x = [&](unsigned char* buf) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (u_lds[k] < 0) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float c4[4] = {rx[k][0][e], rx[k][1][e], rx[k][2][e], rx[k][3][e]};
                uint2 o3[3];
                split_quad(c4, o3);
                unsigned char* dst = buf + u_lds[k] + e * XRS;
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<uint2*>(dst + pp * 32) = o3[pp];
            }
        }
    };

    // ---- A fragments: image[cg][half][chunk][tap jj][piece][lane] (k_convt_pack, convt_img.hip), one chunk in registers
    bf16x8 fa[2][2][3];
    const int a_voff = lane * 16;
    auto load_a = [&](int chunk, int hf) {
        const int so = (((cg * 2 + hf) * p.nchunks + chunk) * 2) * (3 * 1024);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
                fa[hf][jj][pp] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsI, a_voff, so + (jj * 3 + pp) * 1024, 0));
    };

    // ---- B fragment bases: column n = wn*64 + j*32 + l31 -> (row r, position q); window column o (0 .. 2) is position q + o - 1
    int bbase[2];
    unsigned o_lane[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = wn * 64 + j * 32 + l31;
        const int r = n >> p.lsh, q = n & (L - 1);
        const bool ok = r < p.R && b0 + r < p.B;
        bbase[j] = ok ? (r * p.SS + q) * XRS + h * 16 : h * 16;
        o_lane[j] = ok ? 4u * (unsigned)((r * p.Cout + 4 * h) * (2 * L) + 2 * q) : OOB;       // + ((b0*Cout + channel) * 2 L) * 4 (scalar)
    }

    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (int i = tid * 16; i < 2 * buf_bytes; i += 512 * 16) *reinterpret_cast<u32x4*>(smem_cs + i) = z;
    }
    load_x(cbeg, true);
    load_a(cbeg, 0);
    load_a(cbeg, 1);
    __syncthreads();
    store_x(smem_cs);
    load_x(cbeg + 1, nch > 1);
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[hf][j][r] = 0.f;

    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
        const unsigned char* Xs = smem_cs + (ch & 1) * buf_bytes;
        unsigned char* Xn = smem_cs + ((ch & 1) ^ 1) * buf_bytes;
        const bool more = ch + 1 < nch;
        const int a_next = cbeg + (more ? ch + 1 : ch);
        bf16x8 fb[3][2][3];
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
                    fb[o][j][pp] = *reinterpret_cast<const bf16x8*>(Xs + bbase[j] + o * XRS + pp * 32);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int s = 0; s < 6; ++s)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[hf][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[hf][jj][PA[s]], fb[hf + jj][j][PB[s]], acc[hf][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_a(a_next, hf);                                        // this half's registers are free: the next chunk's
            if (hf == 0 && more) store_x(Xn);
            if (hf == 1) load_x(cbeg + ch + 2, ch + 2 < nch);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue: row i = 4h + (r & 3) + 8 (r >> 2) of the sub-tile is channel cg*32 + i; (low, high) -> samples 2q, 2q + 1
    int LS4;
    asm volatile("s_mov_b32 %0, %1" : "=s"(LS4) : "s"(8 * L));        // bytes per output channel row
    const bool fused = p.nsplit == 1;
    float* out = fused ? Y : slabs + (size_t)blockIdx.z * p.zstride;
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x80000000u, 0x00020000);
    const int obase = b0 * p.Cout * LS4;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cs = cg * 32 + (r & 3) + 8 * (r >> 2);           // + 4h in the lane part
            float bv = 0.f;
            if (fused && bias) bv = bias[cs + 4 * h];
            f32x2 v;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float t = acc[hf][j][r];
                if (fused) {
                    t += bv;
                    if (p.act == MS_ACT_LRELU) t = t > 0.f ? t : t * p.slope;
                }
                v[hf] = t;
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rsO, o_lane[j], obase + cs * LS4, 0);
        }
}

__global__ __launch_bounds__(256) void k_convt_fwd_short_finish(const float* __restrict__ slabs, int ns, long long zstride,
                                                               const float* __restrict__ bias, int Cout, int Lout, int act, float slope,
                                                               float* __restrict__ Y, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float v = slabs[i];
        for (int z = 1; z < ns; ++z) v += slabs[(long long)z * zstride + i];
        if (bias) v += bias[(i / Lout) % Cout];
        if (act == MS_ACT_LRELU) v = v > 0.f ? v : v * slope;
        Y[i] = v;
    }
}

bool cs_geometry(const ms_convt1d_desc* d, CsP* p) {
    if (!d || d->B <= 0 || d->Lin <= 0 || d->Cin <= 0 || d->Cout <= 0) return false;
    if (!(d->stride == 2 && d->K == 4 && d->pad == 1)) return false;
    if (d->in_act != MS_ACT_NONE || (d->act != MS_ACT_NONE && d->act != MS_ACT_LRELU)) return false;
    const int L = d->Lin;
    if (L < 4 || L > 16 || (L & (L - 1))) return false;        // (measured at 32 positions: 84 us here vs 62 us on the paired row kernel)
    if (d->Cin % 16 || d->Cout % 64 || d->Cin < 256) return false;
    if ((long long)d->B * L < 512) return false;
    if ((long long)d->B * d->Cin * L * 4 >= (1ll << 31) || (long long)d->B * d->Cout * L * 8 >= (1ll << 31)) return false;
    if ((long long)(d->Cout * 2 / 32) * (d->Cin / 16) * (2 * 3 * 1024) >= (1ll << 31)) return false;
    p->B = d->B; p->Cin = d->Cin; p->Cout = d->Cout; p->L = L;
    int lsh = 0;
    while ((1 << lsh) < L) ++lsh;
    p->lsh = lsh;
    p->SS = L + 2;
    int R = 256 / L;
    if (R * p->SS > PX_MAX) R = PX_MAX / p->SS;
    p->R = R;
    p->PX = R * p->SS;
    p->nchunks = d->Cin / 16;
    p->act = d->act; p->slope = d->slope;
    const int tiles = (d->Cout / 64) * ((d->B + R - 1) / R);
    int ns = 512 / tiles;
    if (ns < 1) ns = 1;
    if (ns > 16) ns = 16;
    while (ns > 1 && p->nchunks / ns < 8) --ns;
    p->zstride = (long long)d->B * d->Cout * 2 * L;
    while (ns > 1 && (size_t)ns * p->zstride * 4 > ((size_t)64 << 20)) --ns;
    p->cps = (p->nchunks + ns - 1) / ns;
    p->nsplit = (p->nchunks + p->cps - 1) / p->cps;
    return true;
}

}  // namespace

// internal interface used by convt_img.hip's ms_convt1d_img_* entry points
bool msct_short_ok(const ms_convt1d_desc* d) {
    const char* sw = getenv("MSYNTH_CONVTSHORT");               // tuning / test switch (0: the generic row kernels)
    if (sw && atoi(sw) == 0) return false;
    CsP p;
    return cs_geometry(d, &p);
}

size_t msct_short_ws(const ms_convt1d_desc* d) {
    CsP p;
    if (!cs_geometry(d, &p)) return 0;
    return p.nsplit > 1 ? (size_t)p.nsplit * p.zstride * sizeof(float) : 0;
}

int msct_short_fwd(const ms_convt1d_desc* d, const float* x, const void* image, const float* bias, float* y, void* ws,
                   size_t ws_bytes, hipStream_t s) {
    CsP p;
    if (!cs_geometry(d, &p)) return MS_ERR_UNSUPPORTED;
    float* slabs = nullptr;
    if (p.nsplit > 1) {
        const size_t need = (size_t)p.nsplit * p.zstride * sizeof(float);
        if (!ws || ws_bytes < need || (((uintptr_t)ws) & 15)) return MS_ERR_WORKSPACE;
        slabs = (float*)ws;
    }
    const size_t lds = (size_t)2 * p.PX * XRS;
    static unsigned long long attr_set = 0;
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_convt_fwd_short), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * PX_MAX * XRS);
        ms_done_on_device(attr_set);
    }
    const dim3 grid((unsigned)((p.B + p.R - 1) / p.R), (unsigned)(p.Cout / 64), (unsigned)p.nsplit);
    ms_note_kernel(6, "k_convt_fwd_short");
    hipLaunchKernelGGL(k_convt_fwd_short, grid, dim3(512), lds, s, p, x, (const u32x4*)image, bias, y, slabs);
    MS_CHECK_LAUNCH();
    if (p.nsplit > 1) {
        const long long total = p.zstride;
        unsigned nb = (unsigned)((total + 255) / 256);
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(k_convt_fwd_short_finish, dim3(nb), dim3(256), 0, s, slabs, p.nsplit, p.zstride, bias, p.Cout, 2 * p.L, p.act,
                           p.slope, y, total);
        MS_CHECK_LAUNCH();
    }
    return MS_OK;
}
