// Fused ResidualAtom forward:  out = x + lrelu(conv1_k3(lrelu(conv_d_k3_dil(x) + b0)) + b1)  in ONE launch
// (reference util/modules.py:350-388; 24 of the generator's 30 convs are the two halves of such an atom).
//
// Same arithmetic as the two row-tile launches it replaces (conv_rows3.hip): fp32 operands split exactly into
// three bf16 pieces, six partial products per multiply on v_mfma_f32_32x32x16_bf16, fp32 accumulation, chunks of
// 16 input channels, taps inside a chunk -- so the intermediate t and the output are bitwise what the unfused
// pair computes from the same weights.  What changes is where the data lives:
//
//   * a workgroup (4 waves) owns NO = NTP - 4 output columns of one batch row for ALL C channels.  It stages the
//     input window x[:, c0-1-d .. c0+NTP+d) ONCE (all channels: the whole contraction is LDS-resident, there is
//     no K-loop staging and no barrier inside either GEMM), computes t on the NTP columns c0-1 .. c0+NTP-2
//     (one halo column each side for the k3 / dil 1 conv that follows), writes lrelu(t) -- split into its bf16
//     pieces straight from the accumulators -- over the dead x window, and runs the second GEMM from there.
//     t never travels to HBM (training additionally stores t and u = lrelu(conv1 + b1) for the backward pass);
//   * the weights are PRE-SPLIT once per step by k_atom_pack into fragment-linear images (one 1 KiB block per
//     (32 output rows, 16-channel chunk, tap, piece) in exactly the order the MFMA A operand wants them) and
//     stream from L2 straight into registers, one chunk ahead: no LDS traffic, no vector work for weights at all
//     (in the unfused kernels every workgroup re-splits every weight it stages: ~45 % of their staging work).
//   MFMA issue is then only interleaved with the B-fragment ds_reads; the vector work left is one split per
//   input element (prologue) and one per t element (between the GEMMs), hidden by the second workgroup on the CU.
//
// LDS: [chunk][column][3 pieces x 16 channels bf16 | 16 pad] = 112 B per (column, chunk) as in conv_rows3.hip
// (fragment reads and 8-byte staging stores conflict-free).  C = 64, d = 9: 4 x 150 x 112 = 67 KB -> 2 WGs / CU.
#include "ms_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int XRS = 112;          // bytes per LDS column of one 16-channel chunk: 3 pieces x 32 + 16

// (a, b) -> three packed bf16 pairs with a = h.lo + m.lo + l.lo exactly (conv_rows3.hip)
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

// ---- weight image ------------------------------------------------------------------------------------------
// image[conv][ms][chunk][tap][piece][lane] (16 B each): lane's A fragment of v_mfma_f32_32x32x16_bf16 for output
// rows ms*32 + (lane & 31), contraction channels chunk*16 + 8*(lane >> 5) + 0..7, tap `tap`.
__host__ __device__ constexpr size_t atom_conv_image_u4(int C) { return (size_t)(C / 32) * (C / 16) * 9 * 64; }

struct AtomPackJob {
    const float* w0;
    const float* w1;
    u32x4* image;
    int C;
    int first_block;     // prefix sum of blocks over the jobs
    int backward;        // 1: the images of the backward-data pass (rows = input channels, taps flipped, conv1 first)
};
constexpr int ATOM_PACK_MAX = 16;
struct AtomPackTable {
    int count;
    AtomPackJob job[ATOM_PACK_MAX];
};

// one thread = one lane's 16-byte fragment of all three pieces of one (conv, ms, chunk, tap)
__global__ __launch_bounds__(256) void k_atom_pack(AtomPackTable t) {
    int j = 0;
#pragma unroll 1
    for (int i = 1; i < t.count; ++i)
        if ((int)blockIdx.x >= t.job[i].first_block) j = i;
    const AtomPackJob jb = t.job[j];
    const int C = jb.C, MS = C / 32, NC = C / 16;
    const int idx = ((int)blockIdx.x - jb.first_block) * 256 + threadIdx.x;      // over conv x ms x chunk x tap x lane
    const int total = 2 * MS * NC * 3 * 64;
    if (idx >= total) return;
    const int lane = idx & 63;
    int r = idx >> 6;
    const int tap = r % 3; r /= 3;
    const int chunk = r % NC; r /= NC;
    const int ms = r % MS;
    const int conv = r / MS;
    // forward: GEMM 0 = the dilated conv (w0), GEMM 1 = the dilation-1 conv (w1), A[row = co][k = ci][tap]
    // backward data: GEMM 0 = conv1 transposed, GEMM 1 = the dilated conv transposed: A[row = ci][k = co][tap] = W[co][ci][2 - tap]
    const float* w = (conv != jb.backward) ? jb.w1 : jb.w0;
    const int row = ms * 32 + (lane & 31), k0 = chunk * 16 + 8 * (lane >> 5);
    unsigned pc[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float a, b;
        if (!jb.backward) {
            a = w[((size_t)row * C + k0 + 2 * q) * 3 + tap];
            b = w[((size_t)row * C + k0 + 2 * q + 1) * 3 + tap];
        } else {
            a = w[((size_t)(k0 + 2 * q) * C + row) * 3 + (2 - tap)];
            b = w[((size_t)(k0 + 2 * q + 1) * C + row) * 3 + (2 - tap)];
        }
        split_pair(a, b, pc[0][q], pc[1][q], pc[2][q]);
    }
    u32x4* dst = jb.image + (size_t)conv * atom_conv_image_u4(C) + ((size_t)((ms * NC + chunk) * 3 + tap) * 3) * 64 + lane;
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) dst[pp * 64] = u32x4{pc[pp][0], pc[pp][1], pc[pp][2], pc[pp][3]};
}

// ---- the fused kernel ---------------------------------------------------------------------------------------
struct AtomP {
    int B, C, L, dil, NO, tiles_per_row, NXA;      // NXA: allocated LDS columns per chunk (>= NTP + 2 dil + 3)
    float slope;
};

// NW waves per workgroup (4, or 8 for C = 256): WGM of them along the channels (32 rows each x TM), WGN along the columns
template <int C, int NTP, int NW>
struct AtomCfg {
    static constexpr int MS = C / 32, WGM = MS < NW ? MS : NW, TM = MS / WGM, WGN = NW / WGM, TN = NTP / 32 / WGN, NC = C / 16;
    static constexpr int NT = 64 * NW;
    static constexpr int NVMAX = (NTP + 18 + 3 + 3) / 4, NV16MAX = (NVMAX + 3) / 4;       // dilation <= 9
    static constexpr int ROUNDS = (NC * NV16MAX * 16 + NT - 1) / NT;
    static_assert(MS >= 1 && TN >= 1 && TM * WGM == MS && TN * WGN * 32 == NTP, "tile shape");
};

// DBG (tools/scratch/probe_atom.py only; 0 in the product) -- timing probes, results are garbage: 1 = weight fragments loaded
// once (no streaming), 2 = B fragments read once per GEMM (no LDS traffic in the K loop), 3 = no global stores,
// 4 = no x window loads / staging, 5 = no MFMAs.
//
// Persistent workgroups: the grid is the number of workgroups the chip holds at once; each walks the tiles
// blockIdx.x, + gridDim.x, ...  Per tile:   [x window(t) registers -> split -> LDS] [issue the x window loads of tile
// t + 1 into registers] [GEMM 1] [t tile -> LDS (+ stores of t)] [GEMM 2] [stores of y (and u)] -- the next tile's loads
// and this tile's stores travel under the two GEMMs, so HBM and the matrix pipe overlap inside ONE workgroup.
// MODE 0: forward, inference (nothing saved);  1: forward, training (also stores t and u);
// MODE 2: BACKWARD DATA of the atom.  With g = dL/dy:   gt = conv1^T(g * lrelu'(u)),   gx = g + conv_d^T(gt * lrelu'(t)).
//   Same two-GEMM structure with the roles mirrored: X = g, the window is multiplied by the LeakyReLU derivative taken from
//   u (U, read) on its way into LDS; GEMM 0 is the dilation-1 conv transposed (halo 1), its raw result gt is stored (T: the
//   weight gradient of the dilated conv needs it) and, multiplied by the derivative from t (Tm, read), becomes the LDS operand
//   of GEMM 1, the dilated conv transposed (halo d): a tile yields NO = NTP - 2 d output columns.
template <int C, int NTP, int NW, int MODE, int DBG = 0>
__global__ __launch_bounds__(64 * NW, 2) void k_atom_fwd(AtomP p, const float* __restrict__ X, const u32x4* __restrict__ IMG,
                                                 const float* __restrict__ b0, const float* __restrict__ b1,
                                                 float* __restrict__ Y, float* __restrict__ T, float* __restrict__ U,
                                                 const float* __restrict__ Tm) {
    constexpr bool SAVE = MODE == 1, BWD = MODE == 2;
    typedef AtomCfg<C, NTP, NW> Cfg;
    constexpr int TM = Cfg::TM, WGN = Cfg::WGN, TN = Cfg::TN, NC = Cfg::NC, ROUNDS = Cfg::ROUNDS, NT = Cfg::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_atom[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: everything derived from it is scalar
    const int wm = wid / WGN, wn = wid % WGN;
    const int d = p.dil, L = p.L;
    const int h1 = BWD ? 1 : d, h2 = BWD ? d : 1;    // halo (= tap step) of the first / second GEMM's conv
    const int sh = (((-1 - d) % 4) + 4) % 4;         // tile starts are multiples of 4: the window start c0 - 1 - d sits sh
                                                     // samples behind an aligned 16-byte vector, the same for every tile
    const int NX = NTP + 2 * h1;                     // window columns the first GEMM reads
    constexpr int NXA = NTP + 22;                    // LDS columns per chunk (dilation <= 9: NTP + 2 d + 3 used)
    constexpr int XCS = NXA * XRS;                   // chunk stride of the x window (compile-time: LDS offsets fold into the instructions)
    constexpr int TCS = XCS;                         // the t tile aliases the window, same strides (it is read up to column NTP - 1 + 2 h2)
    constexpr unsigned OOB = 0xF0000000u;
    const int ntiles = p.B * p.tiles_per_row;

    // Operands and results go through buffer descriptors: a lane that is out of the tensor carries an out-of-range
    // byte offset -- loads return 0.0 (the zero padding), stores are dropped -- and the per-row / per-channel part of an
    // address travels in the scalar offset: no 64-bit address arithmetic, no divergent branches in the epilogues.
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0x80000000u, 0x00020000);
    const auto rsY = __builtin_amdgcn_make_buffer_rsrc(Y, 0, 0x80000000u, 0x00020000);
    // (t tile column `col` is global column c0 - h2 + col: the descriptors of the tensors addressed by t-tile columns start
    //  h2 elements early, so that the lane part of an offset is 4 col >= 0 -- the part the hardware range-checks -- and the
    //  scalar part stays non-negative; lanes left of the row are masked out before they could touch those h2 elements)
    const auto rsT = __builtin_amdgcn_make_buffer_rsrc(MODE ? T - h2 : Y, 0, 0x80000000u, 0x00020000);
    const auto rsU = __builtin_amdgcn_make_buffer_rsrc(MODE ? U : Y, 0, 0x80000000u, 0x00020000);
    const auto rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BWD ? Tm - h2 : X), 0, 0x80000000u, 0x00020000);

    // ---- tile-invariant staging units.  Unit = 4 channels x one aligned 4-sample vector; 16 consecutive lanes = 4
    // channel quads x 4 consecutive vectors (their 8-byte LDS stores fall into 16 different bank pairs).
    const int NV = (NX + sh + 3) >> 2, NV16 = (NV + 3) >> 2;
    int u_goff[ROUNDS], u_t[ROUNDS], u_lcol[ROUNDS], u_lbase[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int u = tid + NT * r;
        const int grp = u >> 4, chunk = grp / NV16, vg = grp - chunk * NV16;
        const int cq = (u >> 2) & 3, v = vg * 4 + (u & 3);
        const bool in = chunk < NC && v < NV;
        u_t[r] = in ? 4 * v - sh - 1 - d : (1 << 28);              // global column of the vector, relative to c0
        u_goff[r] = 4 * ((chunk * 16 + 4 * cq) * L + (4 * v - sh - 1 - d));   // byte offset relative to the tile base
        u_lcol[r] = in ? 4 * v - sh : -1000;
        u_lbase[r] = chunk * XCS + cq * 8;
    }
    f32x4 rx[ROUNDS][4];
    auto load_x = [&](int tile) {
        const int bb = tile / p.tiles_per_row, cc0 = (tile - bb * p.tiles_per_row) * p.NO;
        const int base = 4 * bb * C * L;                            // (wave-uniform: scalar offset)
#pragma unroll
        for (int r = 0; r < (DBG == 4 ? 0 : ROUNDS); ++r) {
            const int t = cc0 + u_t[r];                              // multiple of 4: all inside the row or all outside
            // (the column base rides in the lane offset: the hardware range-checks THAT part, which must not go negative)
            const unsigned goff = (t >= 0 && t < L) ? (unsigned)(u_goff[r] + 4 * cc0) : OOB;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                rx[r][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, goff, base + cc * 4 * L, 0));
        }
    };
    // backward: the window of u (this tile; not prefetched: it would double the registers held across the GEMMs) gives
    // the LeakyReLU derivative the gradient window is multiplied by
    f32x4 ru[BWD ? ROUNDS : 1][4];
    auto load_u = [&](int tile) {
        const int bb = tile / p.tiles_per_row, cc0 = (tile - bb * p.tiles_per_row) * p.NO;
        const int base = 4 * bb * C * L;
#pragma unroll
        for (int r = 0; r < (BWD ? ROUNDS : 0); ++r) {
            const int t = cc0 + u_t[r];
            const unsigned goff = (t >= 0 && t < L) ? (unsigned)(u_goff[r] + 4 * cc0) : OOB;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                ru[r][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, goff, base + cc * 4 * L, 0));
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int r = 0; r < (DBG == 4 ? 0 : ROUNDS); ++r) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = u_lcol[r] + e;
                if (i < 0 || i >= NXA) continue;
                float c4[4] = {rx[r][0][e], rx[r][1][e], rx[r][2][e], rx[r][3][e]};
                if (BWD) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) c4[cc] = ru[r][cc][e] > 0.f ? c4[cc] : c4[cc] * p.slope;
                }
                uint2 o3[3];
                split_quad(c4, o3);
                unsigned char* dst = smem_atom + u_lbase[r] + i * XRS;
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<uint2*>(dst + pp * 32) = o3[pp];
            }
        }
    };

    // ---- A fragments: image -> registers, one chunk ahead (pinned in place: the machine scheduler would otherwise sink
    // every load to its first use and serialise the L2 latency into the MFMA stream)
    // (buffer loads: lane part = lane * 16 + the wave's row block, everything else is a literal scalar offset -- 64-bit
    //  per-fragment pointers would be hoisted out of the tile loop and eat ~100 registers)
    bf16x8 fa[2][TM][3][3];
    const auto rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(IMG), 0, 0x80000000u, 0x00020000);
    const int a_voff = lane * 16 + wm * TM * NC * 9 * 1024;
    auto load_a = [&](int conv, int chunk, bf16x8 (&dst)[TM][3][3]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
                    dst[i][s][pp] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                        rsI, a_voff, (int)(conv * atom_conv_image_u4(C) * 16) + ((i * NC + chunk) * 9 + s * 3 + pp) * 1024, 0));
    };

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    // one GEMM over the LDS-resident operand: B fragment of (chunk, tap s, column sub-tile j) = 16 bytes per piece at
    // column (col0 + 32 j + l31 + s * step), channels 8h .. 8h + 7 of the chunk.  next: what the last chunk prefetches
    // (0: conv 1's chunk 0 for the second GEMM, 1: conv 0's chunk 0 for the next tile, 2: nothing)
    auto gemm = [&](int conv, int cs, int step, int next) {
        const unsigned char* Bs = smem_atom + ((wn * TN) * 32 + l31) * XRS + h * 16;
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // piece pairs, smallest products first
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            if (DBG == 1) {
                if (ch == 0 && conv == 0) load_a(conv, 1, fa[1]);
            } else if (ch + 1 < NC) load_a(conv, ch + 1, fa[(ch + 1) & 1]);
            else if (next == 0) load_a(1, 0, fa[(ch + 1) & 1]);
            else if (next == 1) load_a(0, 0, fa[(ch + 1) & 1]);
            bf16x8 fb[2][TN][3];
            auto fragb = [&](int s, bf16x8 (&dst)[TN][3]) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp)
                        dst[j][pp] = *reinterpret_cast<const bf16x8*>(Bs + ch * cs + (j * 32 + s * step) * XRS + pp * 32);
            };
            // (C = 32: one fragment buffer -- the twelve registers of the second one are what keeps three waves per SIMD, and
            //  with three waves the other two cover the ds_read latency)
            constexpr bool FB2 = C != 32;
            if (FB2 && (DBG != 2 || ch == 0)) fragb(0, fb[0]);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (FB2 && s + 1 < 3 && (DBG != 2 || ch == 0)) fragb(s + 1, fb[(s + 1) & 1]);
                if (!FB2 && (DBG != 2 || (ch == 0 && s == 0))) fragb(s, fb[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (DBG == 5) continue;
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ch & 1][i][s][PA[t]], fb[s & 1][j][PB[t]],
                                                                                acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // per-lane parts of the result offsets (bytes): column n of sub-tile j, channel half h
    int o_lane[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) o_lane[j] = 4 * ((wn * TN + j) * 32 + l31 + 4 * h * L);

    int tile = blockIdx.x;
    if (tile < ntiles) load_x(tile);
    load_a(0, 0, fa[0]);
    // biases: once per workgroup into LDS behind the window (an epilogue then waits ~100 cycles for a ds_read, not for L2)
    float* sbias = reinterpret_cast<float*>(smem_atom + NC * XCS);
    if (!BWD) {
        for (int i = tid; i < 2 * C; i += NT) sbias[i] = i < C ? b0[i] : b1[i - C];
    }
    // PRE: the per-tile dependent loads of the epilogues (the residual; backward: t for the derivative) are issued a GEMM
    // ahead of their use where the registers allow (<= 64 channels): with 2-3 workgroups per CU every memory round trip a
    // tile waits for is throughput lost
    constexpr bool PRE = C <= 64;
    for (; tile < ntiles; tile += gridDim.x) {
        const int b = tile / p.tiles_per_row, c0 = (tile - b * p.tiles_per_row) * p.NO;
        const int base = 4 * (b * C * L + c0);                      // byte offset of (row b, channel 0, column c0)
        if (BWD) load_u(tile);
        int L4;                                                      // 4 L, opaque to the optimiser: the per-channel scalar offsets
        asm volatile("s_mov_b32 %0, %1" : "=s"(L4) : "s"(4 * L));    // are then formed where they are used (2 scalar ops), not hoisted
        constexpr bool PRE_T = BWD && C == 64;          // (C = 32: measured slower with the derivative operand hoisted)
        float tmp[PRE_T ? TM : 1][PRE_T ? TN : 1][16];
        if (PRE_T) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int gc = c0 - h2 + (wn * TN + j) * 32 + l31;
                    const unsigned o_m = (gc >= 0 && gc < L) ? (unsigned)o_lane[j] : OOB;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        tmp[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rsM, o_m, base + ((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2)) * L4, 0));
                }
        }
        store_x();                                                   // (waits for this tile's window)
        zero_acc();
        const int nxt = tile + gridDim.x;
        if (nxt < ntiles) load_x(nxt);                              // travels under both GEMMs
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                             // x window staged
        gemm(0, XCS, h1, 0);                                         // t (pre-activation) on columns c0-h2 .. c0-h2+NTP-1

        unsigned o_y[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = (wn * TN + j) * 32 + l31;
            o_y[j] = (n < p.NO && c0 + n < L) ? (unsigned)o_lane[j] : OOB;
        }

        __syncthreads();                                             // every wave is done with the x window: t overwrites it
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = (wn * TN + j) * 32 + l31;           // t tile column <-> global column c0 - h2 + col
                const int gc = c0 - h2 + col;
                const bool inrow = gc >= 0 && gc < L;                // outside the row t is the second conv's ZERO padding
                // the tile's own columns of t are stored (forward training: the saved activation; backward: the raw gt)
                const unsigned o_t = (MODE != 0 && DBG != 3 && inrow && col >= h2 && col < h2 + p.NO) ? (unsigned)o_lane[j] : OOB;
                float tm[BWD ? 16 : 1];                              // backward: t itself, for the derivative
                if (PRE_T) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) tm[r] = tmp[i][j][r];
                } else if (BWD) {
                    const unsigned o_m = inrow ? (unsigned)o_lane[j] : OOB;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        tm[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rsM, o_m, base + ((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2)) * L4, 0));
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int chs = (wm * TM + i) * 32 + 8 * g, ch0 = chs + 4 * h;
                    float e[4];
                    if (!BWD) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + ch0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float v = acc[i][j][4 * g + q] + bv[q];
                            e[q] = inrow ? (v > 0.f ? v : v * p.slope) : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) e[q] = acc[i][j][4 * g + q];
                    }
                    if (MODE != 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e[q]), rsT, o_t, base + (chs + q) * L4, 0);
                    }
                    if (BWD) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) e[q] = inrow ? (tm[4 * g + q] > 0.f ? e[q] : e[q] * p.slope) : 0.f;
                    }
                    uint2 o3[3];
                    split_quad(e, o3);
                    unsigned char* dst = smem_atom + (col * XRS + 8 * h) + ((chs >> 4) * TCS + (chs & 15) * 2);
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp) *reinterpret_cast<uint2*>(dst + pp * 32) = o3[pp];
                }
            }
        float xrp[PRE ? TM : 1][PRE ? TN : 1][16];                  // the residual x values of this lane's outputs (L2-warm)
        if (PRE) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        xrp[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rsX, o_y[j], base + ((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2)) * L4, 0));
        }
        zero_acc();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                             // t tile complete
        gemm(1, TCS, h2, nxt < ntiles ? 1 : 2);                      // output column n reads t tile columns n, n + h2, n + 2 h2

#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned oy = DBG == 3 ? OOB : o_y[j];
                float xr[16];
                if (PRE) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) xr[r] = xrp[i][j][r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        xr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rsX, o_y[j], base + ((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2)) * L4, 0));
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int chs = (wm * TM + i) * 32 + 8 * g, ch0 = chs + 4 * h;
                    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                    if (!BWD) bv = *reinterpret_cast<const f32x4*>(sbias + C + ch0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v = acc[i][j][4 * g + q] + bv[q];
                        if (!BWD) v = v > 0.f ? v : v * p.slope;
                        if (SAVE) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsU, oy, base + (chs + q) * L4, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + xr[4 * g + q]), rsY, oy,
                                                              base + (chs + q) * L4, 0);
                    }
                }
            }
        __syncthreads();                                             // the t tile is dead: the next window may overwrite it
    }
}

template <int C, int NTP, int NW, int MODE>
int launch_atom_mode(AtomP p, const float* x, const void* image, const float* b0, const float* b1, float* y, float* t,
                     float* u, const float* tm, hipStream_t s) {
    p.NO = MODE == 2 ? ((NTP - 2 * p.dil) & ~3) : NTP - 4;
    if (p.NO < 4) return MS_ERR_UNSUPPORTED;
    p.tiles_per_row = (p.L + p.NO - 1) / p.NO;
    p.NXA = NTP + 22;
    const size_t lds = (size_t)(C / 16) * p.NXA * XRS + 2 * C * sizeof(float);       // window + both biases
    if (lds > 158 * 1024) return MS_ERR_UNSUPPORTED;
    const void* fn = reinterpret_cast<const void*>(&k_atom_fwd<C, NTP, NW, MODE>);
    static int wgs_per_cu = 0, n_cu = 0;
    if (!wgs_per_cu) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        int nb = 0, dev = 0;
        hipDeviceProp_t prop;
        (void)hipGetDevice(&dev);
        (void)hipGetDeviceProperties(&prop, dev);
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * NW, lds) != hipSuccess || nb < 1) nb = 1;
        wgs_per_cu = nb;
    }
    const long long ntiles = (long long)p.B * p.tiles_per_row;
    const long long slots = (long long)n_cu * wgs_per_cu;
    const dim3 grid((unsigned)(ntiles < slots ? ntiles : slots));
    static const int dbg = getenv("MSYNTH_ATOM_DBG") ? atoi(getenv("MSYNTH_ATOM_DBG")) : 0;       // timing probes
    if (dbg && MODE == 1 && NW == 4 && (C == 64 || C == 32)) {
#define MS_ATOM_DBG(D_) if (dbg == D_) { \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_atom_fwd<C, NTP, NW, 1, D_>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024); \
            hipLaunchKernelGGL((k_atom_fwd<C, NTP, NW, 1, D_>), grid, dim3(64 * NW), lds, s, p, x, (const u32x4*)image, b0, b1, y, t, u, tm); MS_CHECK_LAUNCH(); return MS_OK; }
        MS_ATOM_DBG(1) MS_ATOM_DBG(2) MS_ATOM_DBG(3) MS_ATOM_DBG(4) MS_ATOM_DBG(5)
#undef MS_ATOM_DBG
    }
    ms_note_kernel("k_atom_fwd<%d, %d, %d, %d>", C, NTP, NW, MODE);
    hipLaunchKernelGGL((k_atom_fwd<C, NTP, NW, MODE>), grid, dim3(64 * NW), lds, s, p, x, (const u32x4*)image, b0, b1, y, t, u, tm);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

// mode 0 / 1: forward (t, u: the saved activations, both or neither);  mode 2: backward data (x = g, u and tm read, t = gt out)
template <int C, int NTP, int NW>
int launch_atom(int mode, const AtomP& p, const float* x, const void* image, const float* b0, const float* b1, float* y,
                float* t, float* u, const float* tm, hipStream_t s) {
    if (mode == 2) return launch_atom_mode<C, NTP, NW, 2>(p, x, image, b0, b1, y, t, u, tm, s);
    if (mode == 1) return launch_atom_mode<C, NTP, NW, 1>(p, x, image, b0, b1, y, t, u, tm, s);
    return launch_atom_mode<C, NTP, NW, 0>(p, x, image, b0, b1, y, t, u, tm, s);
}

int dispatch_atom(int mode, const ms_atom_desc* d, const float* x, const void* image, const float* b0, const float* b1,
                  float* y, float* t, float* u, const float* tm, hipStream_t s) {
    AtomP p;
    p.B = d->B; p.C = d->C; p.L = d->L; p.dil = d->dil; p.slope = d->slope;
    // tile width: the widest tile whose grid still spreads over the chip; narrow tiles when the whole problem is a few
    // dozen tiles (B = 1 inference: latency, not throughput)
    const long long cols = (long long)d->B * d->L;
    switch (d->C) {
        case 32: return launch_atom<32, 128, 4>(mode, p, x, image, b0, b1, y, t, u, tm, s);
        case 64:
            if (cols < 124 * 128) return launch_atom<64, 64, 4>(mode, p, x, image, b0, b1, y, t, u, tm, s);
            return launch_atom<64, 128, 4>(mode, p, x, image, b0, b1, y, t, u, tm, s);
        case 128:
            if (cols < 60 * 128 && mode != 2) return launch_atom<128, 32, 4>(mode, p, x, image, b0, b1, y, t, u, tm, s);
            return launch_atom<128, 64, 4>(mode, p, x, image, b0, b1, y, t, u, tm, s);
        case 256:
            if (cols < 60 * 64 && mode != 2) return launch_atom<256, 32, 8>(mode, p, x, image, b0, b1, y, t, u, tm, s);
            return launch_atom<256, 64, 8>(mode, p, x, image, b0, b1, y, t, u, tm, s);
        default: return MS_ERR_UNSUPPORTED;
    }
}

bool atom_ok(const ms_atom_desc* d) {
    if (!d || d->B <= 0 || d->L <= 0 || d->dil < 1 || d->dil > 9) return false;
    if (d->C != 32 && d->C != 64 && d->C != 128 && d->C != 256) return false;
    if (d->L % 4) return false;                                   // 16-byte aligned rows
    if ((long long)d->B * d->C * d->L * 4 >= (1ll << 31)) return false;   // 32-bit buffer offsets
    return true;
}

}  // namespace

extern "C" {

size_t ms_residual_atom_image_bytes(int32_t C) {
    if (C <= 0 || C % 32) return 0;
    return 2 * atom_conv_image_u4(C) * 16;
}

int ms_residual_atom_supported(const ms_atom_desc* d) {
    const char* sw = getenv("MSYNTH_ATOM");                      // tuning / test switch (0: the two row-tile launches)
    if (sw && atoi(sw) == 0) return 0;
    return atom_ok(d) ? 1 : 0;
}

int ms_residual_atom_pack_multi(const ms_atom_pack_desc* d, ms_stream_t stream) {
    if (!d || d->count <= 0 || d->count > MS_ATOM_PACK_MAX) return MS_ERR_INVALID_ARG;
    AtomPackTable t;
    t.count = d->count;
    int blocks = 0;
    for (int i = 0; i < d->count; ++i) {
        const int C = d->C[i];
        if (C <= 0 || C % 32 || !d->w0[i] || !d->w1[i] || !d->image[i] || (((uintptr_t)d->image[i]) & 15))
            return MS_ERR_INVALID_ARG;
        t.job[i].w0 = d->w0[i];
        t.job[i].w1 = d->w1[i];
        t.job[i].image = (u32x4*)d->image[i];
        t.job[i].C = C;
        t.job[i].first_block = blocks;
        t.job[i].backward = d->backward[i] ? 1 : 0;
        const int total = 2 * (C / 32) * (C / 16) * 3 * 64;
        blocks += (total + 255) / 256;
    }
    hipLaunchKernelGGL(k_atom_pack, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_residual_atom_fwd(const ms_atom_desc* d, const float* x, const void* image, const float* b0, const float* b1,
                         float* y, float* t, float* y_act, ms_stream_t stream) {
    if (!atom_ok(d)) return d ? MS_ERR_UNSUPPORTED : MS_ERR_INVALID_ARG;
    if (!x || !image || !b0 || !b1 || !y || ((t == nullptr) != (y_act == nullptr))) return MS_ERR_INVALID_ARG;
    if ((((uintptr_t)image) & 15) || (((uintptr_t)b0) & 15) || (((uintptr_t)b1) & 15)) return MS_ERR_INVALID_ARG;
    return dispatch_atom(t ? 1 : 0, d, x, image, b0, b1, y, t, y_act, nullptr, (hipStream_t)stream);
}

int ms_residual_atom_bwd_supported(const ms_atom_desc* d) {
    if (!ms_residual_atom_supported(d)) return 0;
    const char* sw = getenv("MSYNTH_ATOM_BWD");                  // tuning / test switch (0: the two backward-data launches)
    if (sw && atoi(sw) == 0) return 0;
    // a tile yields NTP - 2 dil output columns: with 64-column tiles (128 / 256 channels) dilation 9 would spend a third
    // of the first GEMM on halo -- those atoms keep the two launches
    if (d->C >= 128 && d->dil > 3) return 0;
    return 1;
}

int ms_residual_atom_bwd_data(const ms_atom_desc* d, const float* gy, const float* y_act, const float* t,
                              const void* image_bwd, float* gt, float* gx, ms_stream_t stream) {
    if (!atom_ok(d)) return d ? MS_ERR_UNSUPPORTED : MS_ERR_INVALID_ARG;
    if (!gy || !y_act || !t || !image_bwd || !gt || !gx || (((uintptr_t)image_bwd) & 15)) return MS_ERR_INVALID_ARG;
    return dispatch_atom(2, d, gy, image_bwd, nullptr, nullptr, gx, gt, const_cast<float*>(y_act), t, (hipStream_t)stream);
}

}  // extern "C"
