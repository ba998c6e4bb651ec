// Row-tile forward / backward-data convolution on the bf16 matrix pipe with fp32-exact operands
// (third generation of the dense row-tile kernel; same contraction, tiling, staging geometry and
// epilogue as k_conv_rows2 in conv_rows2.hip).
//
// Why: gfx950's fp32-input MFMA runs at the fp32 VECTOR rate (157 TFLOP/s peak, 1/16 of the bf16
// matrix rate) and the second-generation K loop already keeps it ~96 % busy -- the fp32 pipe itself is
// the ceiling.  Here every fp32 operand is split ON THE WAY INTO LDS into three bf16 pieces
//     x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
// (each subtraction is exact in fp32 and 3 x 8 significand bits cover fp32's 24, so the sum is EXACT),
// and a product a*b is accumulated in fp32 as the six partial products with i + j <= 4
//     a1 b3 + a3 b1 + a2 b2 + a1 b2 + a2 b1 + a1 b1
// on v_mfma_f32_32x32x16_bf16; the three dropped ones are below 2^-24 |a b|, i.e. below the rounding
// of the fp32 FMA chain this replaces (measured: rel-L2 error vs float64 1.3e-7..3e-7 against 5e-7
// for the fmaf chain, K = 768).  Six bf16 MFMAs cover 16 k-elements in 6 x 32 cycles where eight
// fp32 MFMAs (32x32x2) need 8 x 64: 2.67x fewer matrix-pipe cycles per FLOP at fp32-equivalent accuracy.
// It is not a reduced-precision path: no operand is rounded, only sub-ulp cross terms are dropped.
//
// Layout: a chunk is CC = 16 input channels x K taps; MFMA k-step s = tap s (k = the 16 channels).
//   A (weights)      LDS row m:      [tap j][piece p][16 channels] bf16, row stride K*96 + 16 bytes
//   X (activations)  LDS column u:   [piece p][16 channels] bf16, column stride 112 bytes
// (strides = odd multiples of 16 bytes: the 16-byte fragment reads of a lane group cover all banks).
// A lane's fragment of k-step s is ONE 16-byte read per piece: channels 8h..8h+7 of row / column
// (lane & 31) -- the dilation halo is a column offset s*dil, exactly as in the fp32 kernels.
// Staging transposes in registers: a thread takes 4 channels x 4 consecutive samples (4 float4 loads)
// or 4 channels x K taps of one weight row (K float4 loads), splits, and writes 8-byte [4 channels]
// groups.  Two LDS buffers, one barrier per chunk; stores of chunk c+1 and loads of chunk c+2 are
// spread between the MFMAs of chunk c.
//
// AM 0: forward (W = [M][CK][K]);  AM 1: backward data, LeakyReLU derivative from Xact applied on load,
// W read in the forward layout W[co][ci][K] (GEMM row = ci, taps flipped).
#include "conv_rows2.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int CC3 = 16;                     // channels per chunk = one k-step per tap
constexpr int XRS = 112;                    // bytes per LDS activation column: 3 pieces x 32 + 16
constexpr int a_row_bytes(int K) { return K * 96 + 16; }

// (a, b) -> three packed bf16 pairs with a = h.lo + m.lo + l.lo exactly (same for b in the high halves)
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

// 4 values (4 consecutive channels of one (row|column, tap)) -> one 8-byte group per piece
__device__ __forceinline__ void split_quad(const float (&e)[4], uint2 (&o)[3]) {
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair(e[0], e[1], h0, m0, l0);
    split_pair(e[2], e[3], h1, m1, l1);
    o[0] = make_uint2(h0, h1);
    o[1] = make_uint2(m0, m1);
    o[2] = make_uint2(l0, l1);
}

constexpr int msr3_nxu(int BN) { return (4 * (BN / 4 + 12) + 255) / 256; }   // activation units per thread

// VEC: rows are 16-byte aligned (L % 4 == 0): activations are staged with aligned float4 loads and the
// epilogue moves 16-byte vectors.  !VEC (any L: the 1024 -> 1024 k5 conv at L = 17 / 9): dword loads with a
// per-sample range check and dword stores -- those layers are small and L2-resident.
template <int WGM, int WGN, int TM, int TN, int K, int AM, bool VEC>
__global__ __launch_bounds__(256) void k_conv_rows3(Row2P p, const float* __restrict__ X,
                                                   const float* __restrict__ Xact,
                                                   const float* __restrict__ W,
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ res,
                                                   float* __restrict__ Y,
                                                   float* __restrict__ Yact) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int ARS = a_row_bytes(K);
    constexpr int NAU = (BM * 4 + 255) / 256;        // weight units (row, channel quad) per thread
    constexpr int NXU = msr3_nxu(BN);                // activation units (channel quad, 4-sample vector)
    constexpr int NU = NAU + NXU;
    static_assert(WGM * WGN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const int tile_bytes = BM * ARS + p.PX * XRS;
    unsigned char* scratch = smem3 + p.scratch_off;   // 256 x 8 bytes: sink for masked stores
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5;
    const int wm = wid / WGN, wn = wid % WGN;
    const int m0 = blockIdx.y * BM;
    int b0, t0;
    if (p.R == 1) { b0 = blockIdx.x / p.tiles_per_row; t0 = (blockIdx.x - b0 * p.tiles_per_row) * BN; }
    else { b0 = blockIdx.x * p.R; t0 = 0; }

    // Operands are read through buffer descriptors: a lane whose unit lies outside the tensor (rows >= M,
    // samples before / behind a row: the zero padding) carries an out-of-range byte offset and the hardware
    // range check returns 0.0f for it -- no per-element selects in the staging code.  The chunk's channel
    // offset travels in the (unchecked) scalar offset.
    constexpr unsigned OOB = 0xF0000000u;
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 0x80000000u, 0x00020000);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0x80000000u, 0x00020000);
    const auto rsXa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(AM == 1 ? Xact : X), 0, 0x80000000u, 0x00020000);

    // ---- chunk-invariant unit descriptors
    // weight unit i: row = e / 4, channel quad cq = e % 4
    unsigned a_goff[NAU];
    int a_loff[NAU];
#pragma unroll
    for (int i = 0; i < NAU; ++i) {
        const int e = i * 256 + tid;
        const int row = e >> 2, cq = e & 3;
        const bool in = row < BM;
        const bool ok = in && m0 + row < p.M;
        if (AM == 0) a_goff[i] = ok ? 4u * (unsigned)((m0 + row) * p.KG + 4 * cq * K) : OOB;       // + c0*K per chunk
        else a_goff[i] = ok ? 4u * (unsigned)((4 * cq * p.M + m0 + row) * K) : OOB;                // + c0*M*K per chunk
        a_loff[i] = in ? row * ARS + cq * 8 : -1;
    }
    // activation unit q: channel quad cq, 4-sample vector v of the tile's R segments (VEC: the ALIGNED vector)
    const int sh = VEC ? ((p.off0 % 4) + 4) % 4 : 0; // segment start within its 16-byte vector
    const int NVS = (p.SS + 3 + sh) >> 2;            // vectors covering one segment
    const int NVT = p.R * NVS;
    unsigned x_goff[NXU][VEC ? 1 : 4];
    int x_lcol[NXU];
    unsigned x_em[NXU];
    int x_cq8[NXU];
#pragma unroll
    for (int q = 0; q < NXU; ++q) {
        // 16 consecutive lanes = 4 channel quads x 4 consecutive vectors: their 8-byte LDS stores (column stride
        // 112 bytes, 4 columns apart) then fall into 16 different bank pairs (lanes that differ only in the
        // vector index would hit 4)
        const int i = tid + 256 * q;
        const int cq = (i >> 2) & 3, v = (i >> 4) * 4 + (i & 3);
        const int r = v / NVS, sv = v - r * NVS;
        const int u0 = 4 * sv - sh;
        const int t = t0 + p.off0 + u0;                // VEC: multiple of 4, the vector is all in or all out
        const bool in = v < NVT;
        const int base = ((b0 + r) * p.CK + 4 * cq) * p.L + t;
        if (VEC) {
            const bool ok = in && b0 + r < p.B && t >= 0 && t < p.L;
            x_goff[q][0] = ok ? 4u * (unsigned)base : OOB;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = in && b0 + r < p.B && t + e >= 0 && t + e < p.L && u0 + e < p.SS;
                x_goff[q][VEC ? 0 : e] = ok ? 4u * (unsigned)(base + e) : OOB;
            }
        }
        x_lcol[q] = r * p.SS + u0;
        x_cq8[q] = cq * 8;
        unsigned em = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (in && u0 + e >= 0 && u0 + e < p.SS) em |= 1u << e;
        x_em[q] = em;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int bbase[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * TN * 32 + j * 32 + (lane & 31);
        const int r = nl / p.Lt, tc = nl - r * p.Lt;
        const bool ok = r < p.R && b0 + r < p.B && t0 + tc < p.L;
        bbase[j] = ok ? r * p.SS + tc : 0;
    }

    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 ra[NAU][AM == 1 ? 1 : K];
    float rad[NAU][AM == 1 ? 4 * K : 1];
    f32x4 rx[NXU][4], rxa[AM == 1 ? NXU : 1][4];
    auto load_unit = [&](int u, int c0, bool live) {          // c0: first channel of the chunk
        if (u < NAU) {
            const int i = u;
            if (AM == 0) {
                const int so = live ? 4 * c0 * K : 0;
#pragma unroll
                for (int k4 = 0; k4 < K; ++k4)
                    ra[i][k4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, a_goff[i] + 16 * k4, so, 0));
            } else {
                const int qs = 4 * p.M * K;           // byte stride between the quad's 4 contraction channels
                const int so = live ? c0 * qs : 0;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                    for (int j = 0; j < K; ++j)
                        rad[i][qq * K + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsW, a_goff[i] + 4 * j, so + qq * qs, 0));
            }
        } else {
            const int q = u - NAU;
            const int cs = 4 * p.L;
            const int so = live ? c0 * cs : 0;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                if (VEC) {
                    rx[q][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, x_goff[q][0], so + cc * cs, 0));
                    if (AM == 1) rxa[q][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsXa, x_goff[q][0], so + cc * cs, 0));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        rx[q][cc][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, x_goff[q][VEC ? 0 : e], so + cc * cs, 0));
                        if (AM == 1) rxa[q][cc][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsXa, x_goff[q][VEC ? 0 : e], so + cc * cs, 0));
                    }
                }
            }
        }
    };
    // part t of unit u's registers -> LDS: weight units store tap t (t < K), activation units sample t (t < 4)
    auto store_part = [&](int u, int t, unsigned char* buf) {
        if (u < NAU) {
            if (t >= K) return;
            const int i = u, j = t;
            const bool in = a_loff[i] >= 0;
            unsigned char* base = in ? buf + a_loff[i] + j * 96 : scratch + tid * 8;
            const int pstep = in ? 32 : 0;
            float e[4];
            if (AM == 0) {
                // element (channel qq, tap j) of the unit = float index qq*K + j of its K float4s
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) e[qq] = ra[i][(qq * K + j) >> 2][(qq * K + j) & 3];
            } else {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) e[qq] = rad[i][qq * K + (K - 1 - j)];   // taps flipped
            }
            uint2 o3[3];
            split_quad(e, o3);
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<uint2*>(base + pp * pstep) = o3[pp];
        } else {
            if (t >= 4) return;
            const int q = u - NAU, e = t;
            const bool in = (x_em[q] >> e) & 1u;
            unsigned char* base = in ? buf + BM * ARS + (x_lcol[q] + e) * XRS + x_cq8[q] : scratch + tid * 8;
            const int pstep = in ? 32 : 0;
            float c4[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                c4[cc] = rx[q][cc][e];
                if (AM == 1) c4[cc] = rxa[q][cc][e] > 0.f ? c4[cc] : c4[cc] * p.slope;
            }
            uint2 o3[3];
            split_quad(c4, o3);
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<uint2*>(base + pp * pstep) = o3[pp];
        }
    };
    auto store_unit = [&](int u, unsigned char* buf) {
#pragma unroll
        for (int t = 0; t < 6; ++t) store_part(u, t, buf);
    };

    // split-K: slice z contracts channels [z*CKs, min((z+1)*CKs, CK)) into its own output slab
    const int cbeg = blockIdx.z * p.CKs;
    const int nchunks = ((cbeg + p.CKs < p.CK ? cbeg + p.CKs : p.CK) - cbeg) / CC3;
    Y += (size_t)blockIdx.z * p.zstride;

    // prologue: chunk 0 -> buffer 0, chunk 1 -> registers
#pragma unroll
    for (int u = 0; u < NU; ++u) load_unit(u, cbeg, true);
#pragma unroll
    for (int u = 0; u < NU; ++u) store_unit(u, smem3);
#pragma unroll
    for (int u = 0; u < NU; ++u) load_unit(u, cbeg + CC3, nchunks > 1);
    __syncthreads();

    const int arow = (wm * TM * 32 + (lane & 31)) * ARS + h * 16;
    static_assert(K == 3 || K == 5, "taps");
    for (int ch = 0; ch < nchunks; ++ch) {
        const unsigned char* As = smem3 + (ch & 1) * tile_bytes;
        const unsigned char* Xs = As + BM * ARS + h * 16;
        unsigned char* nbuf = smem3 + ((ch & 1) ^ 1) * tile_bytes;
        const bool live2 = ch + 2 < nchunks;
        const int c2 = cbeg + (ch + 2) * CC3;
        bf16x8 fa[2][TM][3], fb[2][TN][3];
        auto frag = [&](int s, bf16x8 (&a)[TM][3], bf16x8 (&b)[TN][3]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
                    a[i][pp] = *reinterpret_cast<const bf16x8*>(As + arow + i * 32 * ARS + s * 96 + pp * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
                    b[j][pp] = *reinterpret_cast<const bf16x8*>(Xs + (bbase[j] + s * p.dil) * XRS + pp * 32);
        };
        frag(0, fa[0], fb[0]);
#pragma unroll
        for (int s = 0; s < K; ++s) {
            if (s + 1 < K) frag(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
            // the six partial products with piece indices i + j <= 4 (1-based), smallest first; between them the
            // staging parts of the units that belong to this k-step (chunk ch+1: registers -> the other buffer;
            // chunk ch+2: loads into the registers just freed)
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][i][PA[t]], fb[s & 1][j][PB[t]], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int u = s; u < NU; u += K) {
                    store_part(u, t, nbuf);
                    if (t == 5) load_unit(u, c2, live2);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue (as k_conv_rows2, EPI_S == 0): the tile goes through LDS -- bias + activation on the way
    // in, then 16-byte rows out: residual loads and both output stores are 16 bytes per lane
    constexpr int TP = BN + 4;
    float* Ts = reinterpret_cast<float*>(smem3);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = wm * TM * 32 + i * 32 + 4 * h;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + mb + (r & 3) + 8 * (r >> 2);
            bv[r] = bias ? bias[m < p.M ? m : 0] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * TN * 32 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ts[(mb + (r & 3) + 8 * (r >> 2)) * TP + col] = ms_apply_act(acc[i][j][r] + bv[r], p.act, p.slope);
        }
    }
    __syncthreads();
    if (VEC) {
        constexpr int V4 = BN / 4;
        constexpr int NQ = BM * V4 / 256;
        float4 tv[NQ], rv[NQ];
        size_t go[NQ];
        bool ok[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = tid + 256 * q;
            const int row = idx / V4, c4 = idx - row * V4;
            const int nl = 4 * c4;
            const int r = nl / p.Lt, tc = nl - r * p.Lt;     // Lt % 4 == 0: the 4 samples share a row
            ok[q] = m0 + row < p.M && r < p.R && b0 + r < p.B && t0 + tc < p.L;
            go[q] = ok[q] ? ((size_t)(b0 + r) * p.M + m0 + row) * p.L + t0 + tc : 0;
            tv[q] = *reinterpret_cast<const float4*>(Ts + row * TP + nl);
            if (res) rv[q] = *reinterpret_cast<const float4*>(res + go[q]);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (!ok[q]) continue;
            if (Yact) *reinterpret_cast<float4*>(Yact + go[q]) = tv[q];
            float4 v = tv[q];
            if (res) { v.x += rv[q].x; v.y += rv[q].y; v.z += rv[q].z; v.w += rv[q].w; }
            *reinterpret_cast<float4*>(Y + go[q]) = v;
        }
    } else {
        // any row length: one dword per lane, 32 consecutive samples of a tile row per half-wave
#pragma unroll 4
        for (int idx = tid; idx < BM * BN; idx += 256) {
            const int row = idx / BN, nl = idx - row * BN;
            const int r = nl / p.Lt, tc = nl - r * p.Lt;
            if (!(m0 + row < p.M && r < p.R && b0 + r < p.B && t0 + tc < p.L)) continue;
            const size_t o = ((size_t)(b0 + r) * p.M + m0 + row) * p.L + t0 + tc;
            float v = Ts[row * TP + nl];
            if (Yact) Yact[o] = v;
            if (res) v += res[o];
            Y[o] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Paired form: 8 waves = two groups of four.  Group g owns output tile 2*blockIdx.x + g of the 128-column
// tiling (same rows m0.., same channel chunks: ONE weight tile in LDS serves both), and the groups alternate
// roles slot by slot:
//     slot 2c    group 0: MFMAs of chunk c        | group 1: split + store X1[c] and its half of A[c+1]
//     slot 2c+1  group 1: MFMAs of chunk c        | group 0: split + store X0[c+1] and its half of A[c+1]
// so on every SIMD one wave feeds the matrix pipe while its partner does the vector work of the staging
// (separate pipes, they co-issue), and a workgroup barrier ends each slot.  Against the four-wave kernel at
// 64x128 this halves the vector work per MFMA (each activation column meets 128 rows, each weight row 256
// columns) -- that kernel is bound by vector issue, not by the matrix pipe.  Weights are double-buffered
// (chunk c+1 is written while chunk c is read), each group's activation tile is single-buffered (written and
// read by the same group in consecutive slots).  Global loads are issued at the start of a group's MFMA
// slot and consumed in its next staging slot.
// Group layout: WGM x (4 / WGM) waves of TM x TN sub-tiles: 2 x 2 waves of TM x 2 (64 or 128 rows) or, for the
// 32-channel layers, 1 x 4 waves of 1 x 1 (32 rows); 128 columns per group either way.
//
// HS = S > 0: ConvTranspose1d forward (stride S = 2 / 8, kernel 2S, padding S/2) as a conv over the input rows with
// M = Cout * S GEMM rows packed by k_pack_convt_w2 in blocks of 64 = [32 low-phase | 32 high-phase] rows: every
// phase has TWO live taps of the 3-column window -- columns {0, 1} for the low phases, {1, 2} for the high ones --
// so a 32-row sub-tile multiplies two weight taps (A has K = 2) against the window columns offset by its half.
// The epilogue scatters row (co, phase) / column q to y[co][q S + phase]: four consecutive samples per store.
// INA: LeakyReLU applied to the input on its way into LDS (the activation in front of the transposed conv).
// DBG (tools/scratch/dbg_r3p.py only; 0 in the product): 1 = staging without the operand split (raw bits stored),
// 2 = no MFMAs, 3 = no staging at all (no loads, no LDS stores), 4 = no global loads inside the K loop, 5 = loads but no
// split / LDS stores -- timing probes, results are garbage.
template <int WGM, int TM, int TN, int K, int AM, int HS = 0, bool INA = false, int DBG = 0>
__global__ __launch_bounds__(512) void k_conv_rows3p(Row2P p, const float* __restrict__ X,
                                                    const float* __restrict__ Xact,
                                                    const float* __restrict__ W,
                                                    const float* __restrict__ bias,
                                                    const float* __restrict__ res,
                                                    float* __restrict__ Y,
                                                    float* __restrict__ Yact) {
    constexpr int WGN = 4 / WGM, BM = WGM * TM * 32, BN = WGN * TN * 32;
    static_assert(BN == 128, "a group owns one 128-column tile");
    constexpr int KA = HS ? 2 : K;                             // weight taps per row
    static_assert(!HS || (K == 3 && AM == 0 && BM >= 64), "transposed-conv form: 3-column window, forward weights");
    constexpr int ARS = a_row_bytes(KA);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const int a_bytes = BM * ARS, x_bytes = p.PX * XRS;
    unsigned char* const Abuf = smem3;                         // A[0] | A[1]
    unsigned char* scratch = smem3 + p.scratch_off;            // 512 x 8 bytes: sink for masked stores
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5;
    const int g = __builtin_amdgcn_readfirstlane(wid >> 2), gt = tid & 255, gw = wid & 3;
    const int wm = gw / WGN, wn = gw % WGN;
    unsigned char* const Xg = smem3 + 2 * a_bytes + g * x_bytes;
    const int m0 = blockIdx.y * BM;
    const int ti = 2 * blockIdx.x + g;
    int b0, t0;
    if (p.R == 1) { b0 = ti / p.tiles_per_row; t0 = (ti - b0 * p.tiles_per_row) * BN; }
    else { b0 = ti * p.R; t0 = 0; }

    constexpr unsigned OOB = 0xF0000000u;
    const auto rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 0x80000000u, 0x00020000);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0x80000000u, 0x00020000);
    const auto rsXa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(AM == 1 ? Xact : X), 0, 0x80000000u, 0x00020000);

    // weight unit of this thread: group g stages rows [g*BM/2, (g+1)*BM/2), unit = (row, channel quad)
    unsigned a_goff;
    int a_loff;
    {
        const int rl = gt >> 2, cq = gt & 3;
        const int row = g * (BM / 2) + rl;
        const bool in = rl < BM / 2;
        const bool ok = in && m0 + row < p.M;
        if (AM == 0) a_goff = ok ? 4u * (unsigned)((m0 + row) * p.KG + 4 * cq * KA) : OOB;
        else a_goff = ok ? 4u * (unsigned)((4 * cq * p.M + m0 + row) * K) : OOB;
        a_loff = in ? row * ARS + cq * 8 : -1;
    }
    // activation unit of this thread (of its group's tile): as k_conv_rows3
    const int sh = ((p.off0 % 4) + 4) % 4;
    const int NVS = (p.SS + 6) >> 2;
    const int NVT = p.R * NVS;
    unsigned x_goff, x_em;
    int x_lcol, x_cq8;
    {
        const int cq = (gt >> 2) & 3, v = (gt >> 4) * 4 + (gt & 3);
        const int r = v / NVS, sv = v - r * NVS;
        const int u0 = 4 * sv - sh;
        const int t = t0 + p.off0 + u0;
        const bool in = v < NVT;
        const bool ok = in && b0 + r < p.B && t >= 0 && t < p.L;
        x_goff = ok ? 4u * (unsigned)(((b0 + r) * p.CK + 4 * cq) * p.L + t) : OOB;
        x_lcol = r * p.SS + u0;
        x_cq8 = cq * 8;
        unsigned em = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (in && u0 + e >= 0 && u0 + e < p.SS) em |= 1u << e;
        x_em = em;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int bbase[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * TN * 32 + j * 32 + (lane & 31);
        const int r = nl / p.Lt, tc = nl - r * p.Lt;
        const bool ok = r < p.R && b0 + r < p.B && t0 + tc < p.L;
        bbase[j] = ok ? r * p.SS + tc : 0;
    }

    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 ra[AM == 1 ? 1 : KA];
    float rad[AM == 1 ? 4 * K : 1];
    f32x4 rx[4], rxa[AM == 1 ? 4 : 1];
    auto load_a = [&](int c0, bool live) {
        if (AM == 0) {
            const int so = live ? 4 * c0 * KA : 0;
#pragma unroll
            for (int k4 = 0; k4 < KA; ++k4)
                ra[k4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, a_goff + 16 * k4, so, 0));
        } else {
            const int qs = 4 * p.M * K;
            const int so = live ? c0 * qs : 0;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int j = 0; j < K; ++j)
                    rad[qq * K + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsW, a_goff + 4 * j, so + qq * qs, 0));
        }
    };
    auto load_x = [&](int c0, bool live) {
        const int cs = 4 * p.L;
        const int so = live ? c0 * cs : 0;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            rx[cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, x_goff, so + cc * cs, 0));
            if (AM == 1) rxa[cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsXa, x_goff, so + cc * cs, 0));
        }
    };
    auto store_a = [&](unsigned char* abuf) {
        const bool in = a_loff >= 0;
        unsigned char* base0 = in ? abuf + a_loff : scratch + tid * 8;
        const int jstep = in ? 96 : 0, pstep = in ? 32 : 0;
#pragma unroll
        for (int j = 0; j < KA; ++j) {
            float e[4];
            if (AM == 0) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) e[qq] = ra[(qq * KA + j) >> 2][(qq * KA + j) & 3];
            } else {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) e[qq] = rad[qq * K + (K - 1 - j)];   // taps flipped
            }
            uint2 o3[3];
            if (DBG == 1) {
                o3[0] = make_uint2(__float_as_uint(e[0]), __float_as_uint(e[1]));
                o3[1] = make_uint2(__float_as_uint(e[2]), __float_as_uint(e[3]));
                o3[2] = o3[0];
            } else split_quad(e, o3);
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<uint2*>(base0 + j * jstep + pp * pstep) = o3[pp];
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = (x_em >> e) & 1u;
            unsigned char* base = in ? Xg + (x_lcol + e) * XRS + x_cq8 : scratch + tid * 8;
            const int pstep = in ? 32 : 0;
            float c4[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                c4[cc] = rx[cc][e];
                if (AM == 1) c4[cc] = rxa[cc][e] > 0.f ? c4[cc] : c4[cc] * p.slope;
                if (INA) c4[cc] = c4[cc] > 0.f ? c4[cc] : c4[cc] * p.slope;
            }
            uint2 o3[3];
            if (DBG == 1) {
                o3[0] = make_uint2(__float_as_uint(c4[0]), __float_as_uint(c4[1]));
                o3[1] = make_uint2(__float_as_uint(c4[2]), __float_as_uint(c4[3]));
                o3[2] = o3[0];
            } else split_quad(c4, o3);
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<uint2*>(base + pp * pstep) = o3[pp];
        }
    };

    float dbg_sink = 0.f;
    auto sink_regs = [&]() {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) dbg_sink += rx[cc][0] + rx[cc][1] + rx[cc][2] + rx[cc][3];
        if (AM == 0) {
#pragma unroll
            for (int k4 = 0; k4 < KA; ++k4) dbg_sink += ra[k4][0] + ra[k4][1] + ra[k4][2] + ra[k4][3];
        }
    };
    const int cbeg = blockIdx.z * p.CKs;
    const int nchunks = ((cbeg + p.CKs < p.CK ? cbeg + p.CKs : p.CK) - cbeg) / CC3;
    Y += (size_t)blockIdx.z * p.zstride;

    const int arow = (wm * TM * 32 + (lane & 31)) * ARS + h * 16;
    auto compute = [&](const unsigned char* As) {
        if (DBG == 2) return;
        const unsigned char* Xs = Xg + h * 16;
        if constexpr (HS != 0) {
            // two weight taps; sub-tile i reads the window columns (tap + its half): TM == 1: the wave's 32 rows
            // are one half (wm), TM == 2: sub-tile 0 = low, 1 = high phases
            constexpr int NO = TM == 1 ? 2 : 3;
            const int hb = TM == 1 ? (wm & 1) : 0;
            bf16x8 fa[TM][2][3], fb[TN][NO][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int sa = 0; sa < 2; ++sa)
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp)
                        fa[i][sa][pp] = *reinterpret_cast<const bf16x8*>(As + arow + i * 32 * ARS + sa * 96 + pp * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int o = 0; o < NO; ++o)
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp)
                        fb[j][o][pp] = *reinterpret_cast<const bf16x8*>(Xs + (bbase[j] + hb + o) * XRS + pp * 32);
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int sa = 0; sa < 2; ++sa)
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                fa[i][sa][PA[t]], fb[j][sa + (TM == 1 ? 0 : (i & 1))][PB[t]], acc[i][j], 0, 0, 0);
            return;
        }
        bf16x8 fa[2][TM][3], fb[2][TN][3];
        auto frag = [&](int s, bf16x8 (&a)[TM][3], bf16x8 (&b)[TN][3]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
                    a[i][pp] = *reinterpret_cast<const bf16x8*>(As + arow + i * 32 * ARS + s * 96 + pp * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
                    b[j][pp] = *reinterpret_cast<const bf16x8*>(Xs + (bbase[j] + s * p.dil) * XRS + pp * 32);
        };
        frag(0, fa[0], fb[0]);
#pragma unroll
        for (int s = 0; s < K; ++s) {
            if (s + 1 < K) frag(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][i][PA[t]], fb[s & 1][j][PB[t]], acc[i][j], 0, 0, 0);
        }
    };

    // prologue: everybody stages its half of A[0]; group 0 its X0[0]; group 1 then holds X1[0] and its half of
    // A[1] in registers for slot 0
    if (DBG != 3) load_a(cbeg, true);
    if (g == 0 && DBG != 3) load_x(cbeg, true);
    if (DBG != 3) store_a(Abuf);
    if (g == 0 && DBG != 3) store_x();
    else if (DBG != 3) { load_x(cbeg, true); load_a(cbeg + CC3, nchunks > 1); }
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        const unsigned char* As = Abuf + (ch & 1) * a_bytes;
        unsigned char* An = Abuf + ((ch & 1) ^ 1) * a_bytes;
        // slot 2*ch
        if (g == 0) {
            if (DBG != 3 && DBG != 4) load_x(cbeg + (ch + 1) * CC3, ch + 1 < nchunks);     // staged in slot 2*ch + 1
            if (DBG != 3 && DBG != 4) load_a(cbeg + (ch + 1) * CC3, ch + 1 < nchunks);
            compute(As);
        } else {
            if (DBG == 5) sink_regs(); else if (DBG != 3) store_x();                                           // X1[ch]
            if (DBG != 3 && DBG != 5) store_a(An);                                         // its half of A[ch + 1]
        }
        __syncthreads();
        // slot 2*ch + 1
        if (g == 1) {
            if (DBG != 3 && DBG != 4) load_x(cbeg + (ch + 1) * CC3, ch + 1 < nchunks);     // staged in slot 2*ch + 2
            if (DBG != 3 && DBG != 4) load_a(cbeg + (ch + 2) * CC3, ch + 2 < nchunks);
            compute(As);
        } else {
            if (DBG == 5) sink_regs(); else if (DBG != 3) store_x();                                           // X0[ch + 1]
            if (DBG != 3 && DBG != 5) store_a(An);                                         // its half of A[ch + 1]
        }
        __syncthreads();
    }

    // ---- epilogue: each group transposes its tile through its own LDS region (the K-loop buffers are dead)
    constexpr int TP = BN + 4;
    float* Ts = reinterpret_cast<float*>(smem3) + g * BM * TP;
    if (DBG == 5 && dbg_sink == 1.2345e-30f) Y[0] = dbg_sink;       // (keeps the probe's loads alive)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int mb = wm * TM * 32 + i * 32 + 4 * h;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = m0 + mb + (r & 3) + 8 * (r >> 2);
            if (m >= p.M) m = 0;
            if (HS) m = (m >> 6) * (64 / (HS ? HS : 1)) + (m & 31) / ((HS ? HS : 2) / 2);    // GEMM row -> output channel
            bv[r] = bias ? bias[m] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * TN * 32 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ts[(mb + (r & 3) + 8 * (r >> 2)) * TP + col] = ms_apply_act(acc[i][j][r] + bv[r], p.act, p.slope);
        }
    }
    __syncthreads();
    constexpr int V4 = BN / 4;
    constexpr int NQ = BM * V4 / 256;
    static_assert(NQ >= 1, "tile too small");
    if constexpr (HS != 0) {
        constexpr int S = HS, SH = S / 2, CPB = 64 / S;            // channels per 64-row block
        constexpr int V4C = 32 * S;                                 // 16-byte vectors per channel of the tile
        const int Cout = p.M / S;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = gt + 256 * q;
            const int chl = idx / V4C, t4 = idx - chl * V4C;
            const int b64 = chl / CPB, cl = chl - b64 * CPB;
            const int nl = (4 * t4) / S, ph0 = (4 * t4) % S;
            const int r = nl / p.Lt, tc = nl - r * p.Lt;
            const int co = (m0 >> 6) * CPB + chl;
            const bool okq = co < Cout && r < p.R && b0 + r < p.B && t0 + tc < p.L;
            float4 v;
            if (S == 8) {
                const float* tr = Ts + (b64 * 64 + (ph0 / SH) * 32 + cl * SH) * TP + nl;
                v = make_float4(tr[0], tr[TP], tr[2 * TP], tr[3 * TP]);
            } else {
                const float* tr = Ts + (b64 * 64 + cl) * TP + nl;
                v = make_float4(tr[0], tr[32 * TP], tr[1], tr[32 * TP + 1]);
            }
            if (okq)
                *reinterpret_cast<float4*>(Y + ((size_t)(b0 + r) * Cout + co) * ((size_t)p.L * S) +
                                           (size_t)(t0 + tc) * S + ph0) = v;
        }
        return;
    }
    float4 tv[NQ], rv[NQ];
    size_t go[NQ];
    bool ok[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int idx = gt + 256 * q;
        const int row = idx / V4, c4 = idx - row * V4;
        const int nl = 4 * c4;
        const int r = nl / p.Lt, tc = nl - r * p.Lt;
        ok[q] = m0 + row < p.M && r < p.R && b0 + r < p.B && t0 + tc < p.L;
        go[q] = ok[q] ? ((size_t)(b0 + r) * p.M + m0 + row) * p.L + t0 + tc : 0;
        tv[q] = *reinterpret_cast<const float4*>(Ts + row * TP + nl);
        if (res) rv[q] = *reinterpret_cast<const float4*>(res + go[q]);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (!ok[q]) continue;
        if (Yact) *reinterpret_cast<float4*>(Yact + go[q]) = tv[q];
        float4 v = tv[q];
        if (res) { v.x += rv[q].x; v.y += rv[q].y; v.z += rv[q].z; v.w += rv[q].w; }
        *reinterpret_cast<float4*>(Y + go[q]) = v;
    }
}

template <int BM, int K>
size_t ldsp_bytes(const Row2P& p) {
    size_t by = (size_t)2 * BM * a_row_bytes(K) + (size_t)2 * p.PX * XRS;
    const size_t epi = (size_t)2 * BM * (128 + 4) * sizeof(float);
    return by < epi ? epi : by;
}

template <int WGM, int TM, int TN, int K, int AM, int HS = 0, bool INA = false, int DBG = 0>
int launch_pair(const Row2P& p, const float* X, const float* Xact, const float* W, const float* bias,
                const float* res, float* Y, float* Yact, dim3 grid, hipStream_t s) {
    const size_t by = ldsp_bytes<WGM * TM * 32, HS ? 2 : K>(p);
    const size_t lds = by + 512 * 8;
    if (lds > 158 * 1024) return MS_ERR_UNSUPPORTED;
    static unsigned long long attr_set = 0;
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_rows3p<WGM, TM, TN, K, AM, HS, INA, DBG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        ms_done_on_device(attr_set);
    }
    Row2P pp = p;
    pp.scratch_off = (int)by;
    ms_note_kernel(6, "k_conv_rows3p<%d, %d, %d, %d, %d, %d, %s>", WGM, TM, TN, K, AM, HS, INA ? "true" : "false");
    hipLaunchKernelGGL((k_conv_rows3p<WGM, TM, TN, K, AM, HS, INA, DBG>), grid, dim3(512), lds, s, pp, X, Xact, W, bias, res,
                       Y, Yact);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

template <int WGM, int WGN, int TM, int TN, int K, int AM, bool VEC = true>
size_t lds_bytes(const Row2P& p) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    size_t by = (size_t)2 * (BM * a_row_bytes(K) + p.PX * XRS);
    const size_t epi = (size_t)BM * (BN + 4) * sizeof(float);
    if (by < epi) by = epi;
    return by;
}

template <int WGM, int WGN, int TM, int TN, int K, int AM, bool VEC>
int launch_inst(const Row2P& p, const float* X, const float* Xact, const float* W, const float* bias,
                const float* res, float* Y, float* Yact, dim3 grid, hipStream_t s) {
    const size_t by = lds_bytes<WGM, WGN, TM, TN, K, AM>(p);
    const size_t lds = by + 256 * 8;
    if (lds > 156 * 1024) return MS_ERR_UNSUPPORTED;
    static unsigned long long attr_set = 0;                    // > 64 KiB of dynamic LDS needs the opt-in once
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_rows3<WGM, WGN, TM, TN, K, AM, VEC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        ms_done_on_device(attr_set);
    }
    Row2P pp = p;
    pp.scratch_off = (int)by;
    ms_note_kernel(6, "k_conv_rows3<%d, %d, %d, %d, %d, %d, %s>", WGM, WGN, TM, TN, K, AM, VEC ? "true" : "false");
    hipLaunchKernelGGL((k_conv_rows3<WGM, WGN, TM, TN, K, AM, VEC>), grid, dim3(256), lds, s, pp, X, Xact, W, bias,
                       res, Y, Yact);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

bool rows_vec(const Row2P& p) { return p.L % 4 == 0 && p.Lt % 4 == 0; }

template <int K, int AM>
int launch_tile(int tile, const Row2P& p, const float* X, const float* Xact, const float* W, const float* bias,
                const float* res, float* Y, float* Yact, dim3 grid, hipStream_t s) {
    if (rows_vec(p)) {
        switch (tile) {
            case MSR2_128x128: return launch_inst<2, 2, 2, 2, K, AM, true>(p, X, Xact, W, bias, res, Y, Yact, grid, s);
            case MSR2_64x128: return launch_inst<2, 2, 1, 2, K, AM, true>(p, X, Xact, W, bias, res, Y, Yact, grid, s);
            case MSR2_64x64: return launch_inst<2, 2, 1, 1, K, AM, true>(p, X, Xact, W, bias, res, Y, Yact, grid, s);
            default: return MS_ERR_UNSUPPORTED;
        }
    }
    switch (tile) {      // rows of any length (dword loader / epilogue)
        case MSR2_64x128: return launch_inst<2, 2, 1, 2, K, AM, false>(p, X, Xact, W, bias, res, Y, Yact, grid, s);
        case MSR2_64x64: return launch_inst<2, 2, 1, 1, K, AM, false>(p, X, Xact, W, bias, res, Y, Yact, grid, s);
        default: return MS_ERR_UNSUPPORTED;
    }
}

}  // namespace

// Same contract as msr2_supported / msr2_launch (conv_rows2.h) for the subset: stride-1 plain rows
// (in_s == 1), zero padding, L % 4 == 0, K in {3, 5}, chunks of 16 channels, act_mode 0 / 1, plain epilogue.
bool msr3_supported(int tile, int K, int act_mode, int epi_s, const Row2P& p, int in_s) {
    const char* sw = getenv("MSYNTH_ROWS3");         // tuning / test switch (0: stay on the fp32-MFMA kernels)
    if (sw && atoi(sw) == 0) return false;
    if (tile < 0 || tile >= MSR2_32x256) return false;      // (32 x 256: measured no faster than the fp32 kernel)
    if (in_s != 1 || epi_s != 0 || (act_mode != 0 && act_mode != 1)) return false;
    if (K != 3 && K != 5) return false;
    if (p.CK % CC3 || p.CKs % CC3) return false;
    if (!rows_vec(p) && tile != MSR2_64x128 && tile != MSR2_64x64) return false;
    if (act_mode == 1 && p.M % 4) return false;
    const int bn = tile == MSR2_32x256 ? 256 : (tile == MSR2_64x64 ? 64 : 128);
    const int bm = tile == MSR2_128x128 ? 128 : (tile == MSR2_32x256 ? 32 : 64);
    const int nvt = p.R * ((p.SS + 6) / 4);
    if (16 * ((nvt + 3) / 4) > 256 * msr3_nxu(bn)) return false;
    size_t by = (size_t)2 * (bm * a_row_bytes(K) + p.PX * XRS);
    const size_t epi = (size_t)bm * (bn + 4) * sizeof(float);
    if (by < epi) by = epi;
    return by + 256 * 8 <= 156 * 1024;
}

int msr3_launch(int tile, int K, int act_mode, const Row2P& p, const float* X, const float* Xact, const float* W,
                const float* bias, const float* res, float* Y, float* Yact, unsigned gx, unsigned gy, unsigned gz,
                hipStream_t s) {
    const dim3 grid(gx, gy, gz);
    if (K == 3 && act_mode == 0) return launch_tile<3, 0>(tile, p, X, Xact, W, bias, res, Y, Yact, grid, s);
    if (K == 3 && act_mode == 1) return launch_tile<3, 1>(tile, p, X, Xact, W, bias, res, Y, Yact, grid, s);
    if (K == 5 && act_mode == 0) return launch_tile<5, 0>(tile, p, X, Xact, W, bias, res, Y, Yact, grid, s);
    if (K == 5 && act_mode == 1) return launch_tile<5, 1>(tile, p, X, Xact, W, bias, res, Y, Yact, grid, s);
    return MS_ERR_UNSUPPORTED;
}

// Paired eight-wave form (k_conv_rows3p): bm = 128 (K = 3) or 64 rows x two adjacent 128-column tiles per
// workgroup.  The caller uses it where the grid still fills the chip: msr3p_grid gives the workgroup count.
bool msr3p_supported(int bm, int K, int act_mode, int epi_s, const Row2P& p, int in_s) {
    const char* sw = getenv("MSYNTH_ROWS3P");        // tuning / test switch (0: four-wave kernel only)
    if (sw && atoi(sw) == 0) return false;
    if (bm != 32 && bm != 64 && bm != 128) return false;
    if (!msr3_supported(bm == 128 ? MSR2_128x128 : MSR2_64x128, K, act_mode, epi_s, p, in_s)) return false;
    if (!rows_vec(p)) return false;
    if (bm == 32 && K != 3) return false;
    const size_t by = bm == 128 ? (K == 3 ? ldsp_bytes<128, 3>(p) : ldsp_bytes<128, 5>(p))
                    : bm == 64 ? (K == 3 ? ldsp_bytes<64, 3>(p) : ldsp_bytes<64, 5>(p)) : ldsp_bytes<32, 3>(p);
    return by + 512 * 8 <= 158 * 1024;
}

int msr3p_launch(int bm, int K, int act_mode, const Row2P& p, const float* X, const float* Xact, const float* W,
                 const float* bias, const float* res, float* Y, float* Yact, unsigned gz, hipStream_t s) {
    const unsigned ntiles = p.R == 1 ? (unsigned)(p.B * p.tiles_per_row) : (unsigned)((p.B + p.R - 1) / p.R);
    const dim3 grid((ntiles + 1) / 2, (unsigned)((p.M + bm - 1) / bm), gz);
#define MS3P(WGM_, TM_, TN_, K_, A_) return launch_pair<WGM_, TM_, TN_, K_, A_>(p, X, Xact, W, bias, res, Y, Yact, grid, s)
    if (bm == 128) {
        if (K == 3 && act_mode == 0) MS3P(2, 2, 2, 3, 0);
        if (K == 3 && act_mode == 1) MS3P(2, 2, 2, 3, 1);
        if (K == 5 && act_mode == 0) MS3P(2, 2, 2, 5, 0);
        if (K == 5 && act_mode == 1) MS3P(2, 2, 2, 5, 1);
    } else if (bm == 64) {
        if (K == 3 && act_mode == 0) MS3P(2, 1, 2, 3, 0);
        if (K == 3 && act_mode == 1) MS3P(2, 1, 2, 3, 1);
        if (K == 5 && act_mode == 0) MS3P(2, 1, 2, 5, 0);
        if (K == 5 && act_mode == 1) MS3P(2, 1, 2, 5, 1);
    } else {
        if (K == 3 && act_mode == 0) MS3P(1, 1, 1, 3, 0);
        if (K == 3 && act_mode == 1) MS3P(1, 1, 1, 3, 1);
    }
#undef MS3P
    return MS_ERR_UNSUPPORTED;
}

// Transposed-conv forward on the paired kernel (HS form).  p: the row description of the mirrored conv with the
// 3-column window (rows2_pick(.., K = 2, ..)), p.M = Cout * S GEMM rows, p.KG = 2 CK; W packed by k_pack_convt_w2.
bool msr3p_convt_supported(int bm, int S, const Row2P& p) {
    const char* sw = getenv("MSYNTH_ROWS3P");
    if (sw && atoi(sw) == 0) return false;
    const char* sc = getenv("MSYNTH_CONVT3");        // tuning / test switch (0: fp32-MFMA transposed-conv kernel)
    if (sc && atoi(sc) == 0) return false;
    if ((S != 2 && S != 8) || (bm != 64 && bm != 128)) return false;
    if (p.M % 64 || p.CK % CC3 || p.CKs % CC3 || p.dil != 1 || p.off0 != -1) return false;
    if (!rows_vec(p) || (long long)p.B * (p.M / S) * p.L * S >= (1ll << 31)) return false;
    const int nvt = p.R * ((p.SS + 6) / 4);
    if (16 * ((nvt + 3) / 4) > 256 * msr3_nxu(128)) return false;
    const size_t by = bm == 128 ? ldsp_bytes<128, 2>(p) : ldsp_bytes<64, 2>(p);
    return by + 512 * 8 <= 158 * 1024;
}

int msr3p_convt_launch(int bm, int S, bool in_act, const Row2P& p, const float* X, const float* W, const float* bias,
                       float* Y, unsigned gz, hipStream_t s) {
    const unsigned ntiles = p.R == 1 ? (unsigned)(p.B * p.tiles_per_row) : (unsigned)((p.B + p.R - 1) / p.R);
    const dim3 grid((ntiles + 1) / 2, (unsigned)((p.M + bm - 1) / bm), gz);
#define MS3T(WGM_, TM_, S_, IA_) \
    return launch_pair<WGM_, TM_, 2, 3, 0, S_, IA_>(p, X, nullptr, W, bias, nullptr, Y, nullptr, grid, s)
    if (bm == 128) {
        if (S == 8) { if (in_act) MS3T(2, 2, 8, true); MS3T(2, 2, 8, false); }
        if (in_act) MS3T(2, 2, 2, true);
        MS3T(2, 2, 2, false);
    }
    if (S == 8) { if (in_act) MS3T(2, 1, 8, true); MS3T(2, 1, 8, false); }
    if (in_act) MS3T(2, 1, 2, true);
    MS3T(2, 1, 2, false);
#undef MS3T
}
