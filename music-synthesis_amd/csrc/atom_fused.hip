// Fused ResidualAtom forward:  out = x + lrelu(conv1_k3(lrelu(conv_d_k3_dil(x) + b0)) + b1)  in ONE launch
// (reference util/modules.py:350-388; 24 of the generator's 30 convs are the two halves of such an atom), and its
// backward data (MODE 2).
//
//   * a workgroup (4 waves; 8 at 256 channels) owns NO = NTP - 4 output columns of one batch row for ALL C channels.  It
//     stages the input window x[:, c0-1-d .. c0+NTP+d) ONCE (all channels: the whole contraction is LDS-resident, there is
//     no K-loop staging and no barrier inside either GEMM), computes t on the NTP columns c0-1 .. c0+NTP-2 (one halo
//     column each side for the k3 / dil 1 conv that follows), writes lrelu(t) -- split into its 16-bit pieces straight
//     from the accumulators -- over the dead x window, and runs the second GEMM from there.  t never travels to HBM
//     (training additionally stores t and u = lrelu(conv1 + b1) for the backward pass);
//   * the weights are PRE-SPLIT once per step by k_atom_pack into fragment-linear images (one 1 KiB block per
//     (32 output rows, 16-channel chunk, tap, piece) in exactly the order the MFMA A operand wants them) and
//     stream from L2 straight into registers through a ring of AD chunk buffers: no LDS traffic, no vector work for
//     weights at all.
//
// Two operand schemes on the 16-bit matrix pipe, fp32 in / fp32 accumulate / fp32 out either way (template NP):
//   NP = 3  every fp32 operand split EXACTLY into three bf16 pieces (8 + 8 + 8 significand bits), six partial products
//           per multiply on v_mfma_f32_32x32x16_bf16 (conv_rows3.hip's arithmetic, bitwise equal to the two row-tile
//           launches).  Ceiling 2500 / 6 TFLOP/s.
//   NP = 2  (r04) block-scaled two-piece fp16: per tile the window (and later the t tile) is multiplied by a power of two S
//           that puts its largest magnitude at 2^14, then x S = h + l with h = fp16(x S), l = fp16(x S - h): 11 + 11
//           significand bits for every element within 2^16 of the tile maximum, an absolute error below 2^-39 of that
//           maximum for the smaller ones (which cannot matter in a dot product with it).  Products h h' + h l' + l h' into
//           ONE fp32 accumulator (every product of two 11-bit numbers is exact in fp32), the dropped l l' is below 2^-22:
//           THREE products on v_mfma_f32_32x32x16_f16 instead of six, 4 instead of 6 bytes per operand
//           element in LDS and in the weight stream.  Power-of-two scalings are exact; the result differs from the fp32
//           FMA chain by what two fp32 summation orders differ by (measured against float64: tests/test_gpu_atom.py).
//           Weights are packed under a per-conv power-of-two scale taken from their largest magnitude (atom_common.h): any magnitude.
//           Ceiling 2500 / 3 TFLOP/s.
//
// LDS: [chunk][column][NP pieces x 16 channels x 2 B | 16 pad] = 112 / 80 B per (column, chunk): an odd multiple of 16 B, so
// fragment reads and the 8-byte staging stores are conflict-free.
#include "atom_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct AtomPackJob {
    const float* w0;
    const float* w1;
    u32x4* image;
    int C;
    int first_block;     // prefix sum of blocks over the jobs
    int backward;        // 1: the images of the backward-data pass (rows = input channels, taps flipped, conv1 first)
    int np;              // pieces per element (3: bf16 x 3, 2: fp16 x 2 of S_w w)
};
constexpr int ATOM_PACK_MAX = 16;
struct AtomPackTable {
    int count;
    AtomPackJob job[ATOM_PACK_MAX];
};

// tail of a job's NP = 2 image: partial maxima and 1 / S_w of its two convs (the allocation is sized for three pieces)
__device__ __forceinline__ float* atom_image_tail(const AtomPackJob& jb) {
    return reinterpret_cast<float*>(jb.image + 2 * atom_conv_image_u4(jb.C, 2));
}

// partial maxima of |w|: workgroup = (job, image conv slot, part)
__global__ __launch_bounds__(256) void k_atom_wmax(AtomPackTable t) {
    __shared__ float red[4];
    const int j = blockIdx.x / (2 * W_NPART), r = blockIdx.x - j * 2 * W_NPART;
    const int conv = r / W_NPART, part = r - conv * W_NPART;
    const AtomPackJob jb = t.job[j];
    const float* w = (conv != jb.backward) ? jb.w1 : jb.w0;          // (the tensor image slot `conv` is packed from: k_atom_pack)
    const int n = jb.C * jb.C * 3, per = (n + W_NPART - 1) / W_NPART;
    const int lo = part * per, hi = lo + per < n ? lo + per : n;
    float m = 0.f;
    typedef float f32x4w __attribute__((ext_vector_type(4), aligned(4)));      // (16 bytes per lane at any 4-byte aligned address)
    int i = lo + 4 * (int)threadIdx.x;
    for (; i + 3 < hi; i += 1024) {
        const f32x4w v = *reinterpret_cast<const f32x4w*>(w + i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    for (; i < hi; ++i) m = fmaxf(m, fabsf(w[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atom_image_tail(jb)[conv * W_NPART + part] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// one thread = one lane's 16-byte fragment of all pieces of one (conv, ms, chunk, tap)
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
    float WS = 1.f, iWS = 1.f;
    if (jb.np == 2) {
        float* tail = atom_image_tail(jb);
        weight_scale(tail + conv * W_NPART, WS, iWS);
        if (ms == 0 && chunk == 0 && tap == 0 && lane == 0) tail[2 * W_NPART + conv] = iWS;
    }
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
        if (jb.np == 3) {
            unsigned o[3];
            split_pair<3>(a, b, o);
            pc[0][q] = o[0]; pc[1][q] = o[1]; pc[2][q] = o[2];
        } else {
            unsigned o[2];
            split_pair<2>(a * WS, b * WS, o);
            pc[0][q] = o[0]; pc[1][q] = o[1]; pc[2][q] = 0u;
        }
    }
    u32x4* dst = jb.image + (size_t)conv * atom_conv_image_u4(C, jb.np) + ((size_t)((ms * NC + chunk) * 3 + tap) * jb.np) * 64 + lane;
    for (int pp = 0; pp < jb.np; ++pp) dst[pp * 64] = u32x4{pc[pp][0], pc[pp][1], pc[pp][2], pc[pp][3]};
}

// m = 2 m + (v > 0): the compare sets the carry the add consumes -- two vector instructions per sign bit
__device__ __forceinline__ void sign_push(unsigned& m, float v) {
    asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(v) : "vcc");
}

// ---- the fused kernel ---------------------------------------------------------------------------------------
struct AtomP {
    int B, C, L, dil, NO, tiles_per_row;           // NO: output columns per tile
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

// Persistent workgroups: the grid is the number of workgroups the chip holds at once; each walks the tiles
// blockIdx.x, + gridDim.x, ...  Per tile:   [x window(t) registers -> split -> LDS] [GEMM 1] [issue the x window loads of
// tile t + 1 into registers] [t tile -> LDS (+ stores of t)] [GEMM 2] [stores of y (and u)] -- the next tile's loads and this
// tile's stores travel under the GEMMs, so HBM and the matrix pipe overlap inside ONE workgroup.
// (A wave's vector-memory operations complete IN ORDER: a wait for a weight fragment also waits for every older load.  NP = 2
//  issues the next window's HBM loads BEHIND the first GEMM, so that the first wait that covers them sits a t epilogue and a
//  chunk later; the NP = 3 form keeps r03's placement in front of the first GEMM.)
// MODE 0: forward, inference (nothing saved);  1: forward, training (also stores t and u);
// MASK (r04, training forward and backward data): the backward pass uses u and -- in the data path -- t only through their
//   SIGNS (the LeakyReLU derivatives).  The training forward then stores, instead of the fp32 tensor u, one bit per element
//   of u and of t: a 16-bit word per (batch row, 32-channel block, lane half h, column) whose bit 15 - r is "value > 0" of
//   channel 32 blk + (r & 3) + 8 (r >> 2) + 4 h -- exactly the 16 accumulator registers a lane holds for that column, so the
//   words leave (and, in the backward t epilogue, arrive) as ONE 2-byte access per 32 x 32 accumulator tile; the staging side
//   of the backward reads four columns' words as one 8-byte vector.  t itself is still stored (the weight gradient of the
//   second conv multiplies by it).  Per atom the forward writes 3 + 1/16 instead of 4 tensors, the backward reads 1 + 1/16
//   instead of 3 (and the weight gradients, wgrad_rows.hip, 4 + 1/16 instead of 6).
// MODE 2: BACKWARD DATA of the atom.  With g = dL/dy:   gt = conv1^T(g * lrelu'(u)),   gx = g + conv_d^T(gt * lrelu'(t)).
//   Same two-GEMM structure with the roles mirrored: X = g, the window is multiplied by the LeakyReLU derivative taken from
//   u (U, read) on its way into LDS; GEMM 0 is the dilation-1 conv transposed (halo 1), its raw result gt is stored (T: the
//   weight gradient of the dilated conv needs it) and, multiplied by the derivative from t (Tm, read), becomes the LDS operand
//   of GEMM 1, the dilated conv transposed (halo d): a tile yields NTP - 2 d output columns.
// Measured and dropped in r04 (tools/scratch/probe_atom_np.py, B = 32): a ring of four fragment buffers (spills at 64 channels,
// no gain at 128: the kernel is not bound by the weight stream's latency); the second GEMM with swapped operands so that the
// epilogue moves 16 bytes per lane (one channel per lane: 64 scattered 16-byte pieces per instruction -- 5-15 % SLOWER than
// dword accesses that are 128 contiguous bytes per channel row); two tile widths dealt so that the persistent workgroups get
// equal column counts (-4 % at 128 channels, +8 % at 64 where the second GEMM body spills); a start stagger of the second
// resident workgroup and s_setprio around the GEMMs (0 to +5 % slower); y / u / residual as 16-byte row-major vectors through a
// per-wave LDS transpose (a quarter of the epilogue's vector-memory instructions: no change of the train step, +-1 %).
template <int C, int NTP, int NW, int MODE, int NP, bool MASK = false>
__global__ __launch_bounds__(64 * NW, 2) void k_atom_fwd(AtomP p, const float* __restrict__ X, const u32x4* __restrict__ IMG,
                                                 const float* __restrict__ b0, const float* __restrict__ b1,
                                                 float* __restrict__ Y, float* __restrict__ T, float* __restrict__ U,
                                                 const float* __restrict__ Tm, float* __restrict__ AM) {
    // (MASK: U is the sign-word array of u, Tm that of t -- written by MODE 1, read by MODE 2)
    constexpr bool SAVE = MODE == 1, BWD = MODE == 2;
    static_assert(!MASK || (MODE != 0 && NP == 2), "sign words: training forward / backward data of the two-piece scheme");
    typedef AtomCfg<C, NTP, NW> Cfg;
    constexpr int TM = Cfg::TM, WGN = Cfg::WGN, TN = Cfg::TN, NC = Cfg::NC, ROUNDS = Cfg::ROUNDS, NT = Cfg::NT;
    constexpr int XRS = xrs<NP>();
    constexpr bool SC = NP == 2;                     // block-scaled fp16 pieces
    constexpr bool XLATE = NP == 2;                  // next window's loads issued behind GEMM 1 (see above)
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
    // sign words (MASK): [b][C / 32][h][column] of 16 bits; the array addressed by t-tile columns starts h2 words early, as rsT
    typedef unsigned short u16;
    const auto rsSU = __builtin_amdgcn_make_buffer_rsrc(MASK ? reinterpret_cast<u16*>(U) : reinterpret_cast<u16*>(Y), 0, 0x80000000u, 0x00020000);
    const auto rsST = __builtin_amdgcn_make_buffer_rsrc(
        MASK ? reinterpret_cast<u16*>(const_cast<float*>(Tm)) - h2 : reinterpret_cast<u16*>(Y), 0, 0x80000000u, 0x00020000);

    // ---- tile-invariant staging units.  Unit = 4 channels x one aligned 4-sample vector; 16 consecutive lanes = 4
    // channel quads x 4 consecutive vectors (their 8-byte LDS stores fall into 16 different bank pairs).
    const int NV = (NX + sh + 3) >> 2, NV16 = (NV + 3) >> 2;
    int u_goff[ROUNDS], u_t[ROUNDS], u_lcol[ROUNDS], u_lbase[ROUNDS];
    int u_moff[MASK && BWD ? ROUNDS : 1], u_msh[MASK && BWD ? ROUNDS : 1];   // sign words of the unit's 4 channels x 4 columns
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
        if constexpr (MASK && BWD) {
            // channels chunk * 16 + 4 cq + 0..3: block chunk >> 1, half cq & 1, registers r0 .. r0 + 3 with r0 = 4 (2 (chunk & 1) + (cq >> 1))
            u_moff[r] = 2 * ((((chunk >> 1) * 2 + (cq & 1)) * L) + (4 * v - sh - 1 - d));
            u_msh[r] = 15 - 4 * (2 * (chunk & 1) + (cq >> 1));     // bit of channel + 0; channel + cc: one lower each
        }
    }
    f32x4 rx[ROUNDS][4];
    auto load_x = [&](int tile) {
        const int bb = tile / p.tiles_per_row, cc0 = (tile - bb * p.tiles_per_row) * p.NO;
        const int base = 4 * bb * C * L;                            // (wave-uniform: scalar offset)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
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
    f32x4 ru[BWD && !MASK ? ROUNDS : 1][4];
    u32x2 rum[BWD && MASK ? ROUNDS : 1];             // (MASK) four columns' sign words of the unit
    auto load_u = [&](int bb, int cc0) {
        const int base = 4 * bb * C * L;
        if constexpr (MASK) {
#pragma unroll
            for (int r = 0; r < (BWD ? ROUNDS : 0); ++r) {
                const int t = cc0 + u_t[r];
                const unsigned moff = (t >= 0 && t < L) ? (unsigned)(u_moff[r] + 2 * cc0) : OOB;
                rum[r] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsSU, moff, 2 * bb * (C / 32) * 2 * L, 0));
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < (BWD && !MASK ? ROUNDS : 0); ++r) {
            const int t = cc0 + u_t[r];
            const unsigned goff = (t >= 0 && t < L) ? (unsigned)(u_goff[r] + 4 * cc0) : OOB;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                ru[r][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsU, goff, base + cc * 4 * L, 0));
        }
    };
    // biases and the two block-scale exchange areas live behind the window
    float* sbias = reinterpret_cast<float*>(smem_atom + NC * XCS);
    float* smax1 = sbias + 2 * C;                                    // [NW]: per-wave |max| of the NEXT tile's window
    float* smax2 = smax1 + NW;                                       // [NW]: per-wave |max| of this tile's t operand
    // (the largest magnitude of the raw window bounds the operand: the backward pass multiplies it by 1 or the slope)
    auto publish_window_max = [&]() {
        if (!SC) return;
        float m = 0.f;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(rx[r][cc][e]));
        m = wave_max(m);
        if (lane == 0) smax1[wid] = m;
    };
    auto read_max = [&](const float* sm) {
        float m = sm[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, sm[w]);
        return m;
    };
    auto store_x = [&](float S) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = u_lcol[r] + e;
                if (i < 0 || i >= NXA) continue;
                float c4[4] = {rx[r][0][e], rx[r][1][e], rx[r][2][e], rx[r][3][e]};
                if constexpr (BWD && MASK) {
                    const unsigned pair = (e >> 1) ? rum[r].y : rum[r].x;
                    const unsigned wd = (e & 1) ? pair >> 16 : pair;           // column e's word (bits above 15: ignored)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) c4[cc] = ((wd >> (u_msh[r] - cc)) & 1u) ? c4[cc] : c4[cc] * p.slope;
                } else if constexpr (BWD) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) c4[cc] = ru[r][cc][e] > 0.f ? c4[cc] : c4[cc] * p.slope;
                }
                if (SC) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) c4[cc] *= S;
                }
                uint2 o3[NP];
                split_quad<NP>(c4, o3);
                unsigned char* dst = smem_atom + u_lbase[r] + i * XRS;
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) *reinterpret_cast<uint2*>(dst + pp * 32) = o3[pp];
            }
        }
    };

    // ---- A fragments: image -> registers, one chunk ahead through two buffers (pinned in place: the machine scheduler would
    // otherwise sink every load to its first use and serialise the L2 latency into the MFMA stream)
    // (buffer loads: lane part = lane * 16 + the wave's row block, everything else is a literal scalar offset -- 64-bit
    //  per-fragment pointers would be hoisted out of the tile loop and eat ~100 registers)
    u32x4 fa[2][TM][3][NP];
    const auto rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(IMG), 0, 0x80000000u, 0x00020000);
    // NP = 2: 1 / S_w of the two GEMMs' weight images (the pack left them in the image's tail)
    float winv0 = 1.f, winv1 = 1.f;
    if (SC) {
        const float* tail = reinterpret_cast<const float*>(IMG + 2 * atom_conv_image_u4(C, 2)) + 2 * W_NPART;
        winv0 = tail[0]; winv1 = tail[1];
    }
    const int a_voff = lane * 16 + wm * TM * NC * 3 * NP * 1024;
    // q: chunk counter inside a tile, 0 .. 2 NC - 1 (GEMM 0's chunks, then GEMM 1's); buffer q & 1
    auto load_a = [&](int q) {
        const int conv = q / NC, chunk = q % NC;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int pp = 0; pp < NP; ++pp)
                    fa[q & 1][i][s][pp] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rsI, a_voff, (int)(conv * atom_conv_image_u4(C, NP) * 16) + ((i * NC + chunk) * 3 * NP + s * NP + pp) * 1024, 0));
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
    // column (col0 + 32 j + l31 + s * step), channels 8h .. 8h + 7 of the chunk.  more: a next tile exists (its first chunk's
    // fragments are requested from the tail of the second GEMM)
    auto gemm = [&](auto gc, int step, bool more) {
        constexpr int g = decltype(gc)::value, TNE = TN;
        const unsigned char* Bs = smem_atom + ((wn * TN) * 32 + l31) * XRS + h * 16;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const int q = g * NC + ch, qn = q + 1;
            if (qn < 2 * NC) load_a(qn);
            else if (more) load_a(0);
            u32x4 fb[2][TNE][NP];
            auto fragb = [&](int s, u32x4 (&dst)[TNE][NP]) {
#pragma unroll
                for (int j = 0; j < TNE; ++j)
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp)
                        dst[j][pp] = *reinterpret_cast<const u32x4*>(Bs + ch * XCS + (j * 32 + s * step) * XRS + pp * 32);
            };
            // (C = 32: one fragment buffer -- the registers of the second one are what keeps three waves per SIMD, and
            //  with three waves the other two cover the ds_read latency)
            constexpr bool FB2 = C != 32;
            if (FB2) fragb(0, fb[0]);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (FB2 && s + 1 < 3) fragb(s + 1, fb[(s + 1) & 1]);
                if (!FB2) fragb(s, fb[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (NP == 3) {
                    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // piece pairs, smallest products first
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TNE; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                    __builtin_bit_cast(bf16x8, fa[q & 1][i][s][PA[t]]), __builtin_bit_cast(bf16x8, fb[s & 1][j][PB[t]]),
                                    acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TNE; ++j) {
                            const f16x8 ah = __builtin_bit_cast(f16x8, fa[q & 1][i][s][0]), al = __builtin_bit_cast(f16x8, fa[q & 1][i][s][1]);
                            const f16x8 bh = __builtin_bit_cast(f16x8, fb[s & 1][j][0]), bl = __builtin_bit_cast(f16x8, fb[s & 1][j][1]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);     // smallest products first
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    // per-lane parts of the result offsets (bytes): column n of sub-tile j, channel half h
    int o_lane[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) o_lane[j] = 4 * ((wn * TN + j) * 32 + l31 + 4 * h * L);
    int o_word[MASK ? TN : 1];                       // ... and of a sign-word offset: column, lane half
    if constexpr (MASK) {
#pragma unroll
        for (int j = 0; j < TN; ++j) o_word[j] = 2 * ((wn * TN + j) * 32 + l31 + h * L);
    }

    int tile = blockIdx.x;
    if (tile < ntiles) load_x(tile);
    load_a(0);
    // biases: once per workgroup into LDS behind the window (an epilogue then waits ~100 cycles for a ds_read, not for L2)
    if (!BWD) {
        for (int i = tid; i < 2 * C; i += NT) sbias[i] = i < C ? b0[i] : b1[i - C];
    }
    if (SC) {
        publish_window_max();                        // (waits for the first window)
        __syncthreads();
    }
    float run1 = 0.f, run2 = 0.f;                    // NP = 2: largest window / second-operand magnitude over this workgroup's tiles
    // PRE: the per-tile dependent loads of the epilogues (the residual; backward: t for the derivative) are issued a GEMM
    // ahead of their use where the registers allow: with 2-3 workgroups per CU every memory round trip a tile waits for is
    // throughput lost
    // (128 channels with the registers the single accumulator freed: measured 3 % SLOWER on the train step)
    constexpr bool PRE = C == 64 || (C == 32 && NP == 3);
    for (; tile < ntiles; tile += gridDim.x) {
        const int b = tile / p.tiles_per_row, c0 = (tile - b * p.tiles_per_row) * p.NO;
        const int no = p.NO;
        const int wcol = wn * TN * 32;                               // first tile column of this wave's sub-tiles
        const int base = 4 * (b * C * L + c0);                      // byte offset of (row b, channel 0, column c0)
        if (BWD) load_u(b, c0);
        int L4;                                                      // 4 L, opaque to the optimiser: the per-channel scalar offsets
        asm volatile("s_mov_b32 %0, %1" : "=s"(L4) : "s"(4 * L));    // are then formed where they are used (2 scalar ops), not hoisted
        const int wbase = 2 * (b * (C / 32) * 2 * L + c0);           // (MASK) byte offset of (row b, block 0, half 0, column c0)
        unsigned tmw[MASK && BWD ? TM : 1][MASK && BWD ? TN : 1];    // (MASK) t's sign words of this lane's columns
        if constexpr (MASK && BWD) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int gc = c0 - h2 + wcol + j * 32 + l31;
                    const unsigned o_w = (gc >= 0 && gc < L) ? (unsigned)o_word[j] : OOB;
                    tmw[i][j] = __builtin_amdgcn_raw_buffer_load_b16(rsST, o_w, wbase + (wm * TM + i) * 4 * L, 0);
                }
        }
        constexpr bool PRE_T = BWD && C == 64 && !MASK;  // (C = 32: measured slower with the derivative operand hoisted)
        float tmp[PRE_T ? TM : 1][PRE_T ? TN : 1][16];
        if (PRE_T) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int gc = c0 - h2 + wcol + j * 32 + l31;
                    const unsigned o_m = (gc >= 0 && gc < L) ? (unsigned)o_lane[j] : OOB;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        tmp[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rsM, o_m, base + ((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2)) * L4, 0));
                }
        }
        float S1 = 1.f, iS1 = 1.f;
        if (SC) {                                                    // (published behind the previous tile / in the prologue)
            const float m1 = read_max(smax1);
            run1 = fmaxf(run1, m1);
            block_scale(m1, S1, iS1);
        }
        store_x(S1);                                                 // (waits for this tile's window)
        zero_acc();
        const int nxt = tile + gridDim.x;
        if (!XLATE && nxt < ntiles) load_x(nxt);                    // travels under both GEMMs
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                             // x window staged
        gemm(I0{}, h1, nxt < ntiles);                                // t (pre-activation) on columns c0-h2 .. c0-h2+NTP-1
        if (XLATE && nxt < ntiles) load_x(nxt);                     // travels under the t epilogue and the second GEMM
        __builtin_amdgcn_sched_barrier(0);

        // ---- t epilogue, first half: accumulators -> the second GEMM's operand values, in place (fp32); stores of t
        const float k1 = SC ? iS1 * winv0 : 1.f;                    // undoes the window's and the weights' scales
        float tmax = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wcol + j * 32 + l31;                // t tile column <-> global column c0 - h2 + col
                const int gc = c0 - h2 + col;
                const bool inrow = gc >= 0 && gc < L;                // outside the row t is the second conv's ZERO padding
                // the tile's own columns of t are stored (forward training: the saved activation; backward: the raw gt)
                const unsigned o_t = (MODE != 0 && inrow && col >= h2 && col < h2 + no) ? (unsigned)o_lane[j] : OOB;
                float tm[BWD && !MASK ? 16 : 1];                     // backward: t itself, for the derivative
                unsigned sw = 0;                                     // (MASK, forward) sign word of this tile of t
                if constexpr (MASK && BWD) {
                } else if (PRE_T) {
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
                            float v = acc[i][j][4 * g + q];
                            if (SC) v *= k1;
                            v += bv[q];
                            e[q] = inrow ? (v > 0.f ? v : v * p.slope) : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v = acc[i][j][4 * g + q];
                            if (SC) v *= k1;
                            e[q] = v;
                        }
                    }
                    if (MODE != 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e[q]), rsT, o_t, base + (chs + q) * L4, 0);
                    }
                    if constexpr (MASK && SAVE) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) sign_push(sw, e[q]);
                    }
                    if (BWD) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            bool pos;
                            if constexpr (MASK) pos = ((tmw[i][j] >> (15 - (4 * g + q))) & 1u) != 0;
                            else pos = tm[4 * g + q] > 0.f;
                            e[q] = inrow ? (pos ? e[q] : e[q] * p.slope) : 0.f;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[i][j][4 * g + q] = e[q];
                        if (SC) tmax = fmaxf(tmax, fabsf(e[q]));
                    }
                }
                if constexpr (MASK && SAVE)
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)sw, rsST, o_t == OOB ? OOB : (unsigned)o_word[j],
                                                          wbase + (wm * TM + i) * 4 * L, 0);
            }
        if (SC) {
            tmax = wave_max(tmax);
            if (lane == 0) smax2[wid] = tmax;
        }
        __syncthreads();                                             // every wave is done with the x window: t overwrites it
        float S2 = 1.f, iS2 = 1.f;
        if (SC) {
            const float m2 = read_max(smax2);
            run2 = fmaxf(run2, m2);
            block_scale(m2, S2, iS2);
        }
        // ---- second half: split (x S2) -> LDS
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wcol + j * 32 + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int chs = (wm * TM + i) * 32 + 8 * g;
                    float e[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) e[q] = SC ? acc[i][j][4 * g + q] * S2 : acc[i][j][4 * g + q];
                    uint2 o3[NP];
                    split_quad<NP>(e, o3);
                    unsigned char* dst = smem_atom + (col * XRS + 8 * h) + ((chs >> 4) * TCS + (chs & 15) * 2);
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) *reinterpret_cast<uint2*>(dst + pp * 32) = o3[pp];
                }
            }
        unsigned o_y[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = wcol + j * 32 + l31;
            o_y[j] = (n < no && c0 + n < L) ? (unsigned)o_lane[j] : OOB;
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
        gemm(I1{}, h2, nxt < ntiles);                                // output column n reads t tile columns n, n + h2, n + 2 h2

        const float k2 = SC ? iS2 * winv1 : 1.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const unsigned oy = o_y[j];
                unsigned su = 0;                                     // (MASK) sign word of this tile of u
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
                        float v = acc[i][j][4 * g + q];
                        if (SC) v *= k2;
                        v += bv[q];
                        if (!BWD) v = v > 0.f ? v : v * p.slope;
                        if constexpr (SAVE && MASK) sign_push(su, v);
                        else if (SAVE) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsU, oy, base + (chs + q) * L4, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + xr[4 * g + q]), rsY, oy,
                                                              base + (chs + q) * L4, 0);
                    }
                }
                if constexpr (MASK && SAVE)
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)su, rsSU, oy == OOB ? OOB : (unsigned)o_word[j],
                                                          wbase + (wm * TM + i) * 4 * L, 0);
            }
        if (SC && nxt < ntiles) publish_window_max();               // (waits for the next tile's window: issued a GEMM ago)
        __syncthreads();                                             // the t tile is dead: the next window may overwrite it
    }
    // Per-workgroup maxima for the consumers of these tensors (the weight-gradient kernel takes its block scales from them:
    // wgrad_rows.hip): AM[0][b] = largest |first-GEMM operand| this workgroup saw (forward: x; backward: g, which bounds
    // g lrelu'(u)), AM[1][b] = largest |second-GEMM operand| (forward: t; backward: gt lrelu'(t)).  Windows overlap, so
    // a value may be reported by two workgroups -- the consumer takes the maximum over all MS_ATOM_AMAX_N entries, the ones
    // behind the grid are cleared by workgroup 0.
    if (SC && AM) {
        if (tid == 0) { AM[blockIdx.x] = run1; AM[MS_ATOM_AMAX_N + blockIdx.x] = run2; }
        if (blockIdx.x == 0)
            for (int i = gridDim.x + tid; i < MS_ATOM_AMAX_N; i += NT) { AM[i] = 0.f; AM[MS_ATOM_AMAX_N + i] = 0.f; }
    }
}

// pieces per operand element: 2 (block-scaled fp16 x 2, three products) unless MSYNTH_ATOM_NP=3 (bf16 x 3, six products:
// the r03 kernel, bitwise equal to the two row-tile launches)
int atom_np() {
    static const int np = (getenv("MSYNTH_ATOM_NP") && atoi(getenv("MSYNTH_ATOM_NP")) == 3) ? 3 : 2;
    return np;
}

constexpr int MAX_DEV = 64;            // per-device launch parameters (ms_common.h: one-time launch setup)

template <int C, int NTP, int NW, int MODE, int NP, bool MASK = false>
int launch_atom_np(AtomP p, const float* x, const void* image, const float* b0, const float* b1, float* y, float* t,
                   float* u, const float* tm, float* am, hipStream_t s) {
    p.NO = MODE == 2 ? ((NTP - 2 * p.dil) & ~3) : NTP - 4;
    if (p.NO < 4) return MS_ERR_UNSUPPORTED;
    p.tiles_per_row = (p.L + p.NO - 1) / p.NO;
    const size_t lds = (size_t)(C / 16) * (NTP + 22) * xrs<NP>() + (2 * C + 2 * NW) * sizeof(float);       // window + biases + scale exchange
    if (lds > 158 * 1024) return MS_ERR_UNSUPPORTED;
    const void* fn = reinterpret_cast<const void*>(&k_atom_fwd<C, NTP, NW, MODE, NP, MASK>);
    static int wgs_per_cu[MAX_DEV] = {}, n_cu[MAX_DEV] = {};
    const int dev = ms_current_device();
    if (!__atomic_load_n(&wgs_per_cu[dev], __ATOMIC_ACQUIRE)) {      // (idempotent: racing first calls compute the same values)
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
        int nb = 0;
        hipDeviceProp_t prop;
        (void)hipGetDeviceProperties(&prop, dev);
        n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * NW, lds) != hipSuccess || nb < 1) nb = 1;
        __atomic_store_n(&wgs_per_cu[dev], nb, __ATOMIC_RELEASE);
    }
    const long long slots = (long long)n_cu[dev] * wgs_per_cu[dev];
    const long long ntiles = (long long)p.B * p.tiles_per_row;
    // a launch that publishes operand maxima has one slot per workgroup in the caller's table: the persistent grid (whose tile
    // loop works with any workgroup count) is clamped to it, whatever the occupancy query and the CU count say
    long long nwg = ntiles < slots ? ntiles : slots;
    if (am && nwg > MS_ATOM_AMAX_N) nwg = MS_ATOM_AMAX_N;
    const dim3 grid((unsigned)nwg);
    ms_note_kernel(NP == 2 ? 3 : 6, "k_atom_fwd<%d, %d, %d, %d, %d, %s>", C, NTP, NW, MODE, NP, MASK ? "true" : "false");
    hipLaunchKernelGGL((k_atom_fwd<C, NTP, NW, MODE, NP, MASK>), grid, dim3(64 * NW), lds, s, p, x, (const u32x4*)image, b0, b1, y, t, u, tm, am);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

template <int C, int NTP, int NW, int MODE>
int launch_atom_mode(bool mask, const AtomP& p, const float* x, const void* image, const float* b0, const float* b1, float* y, float* t,
                     float* u, const float* tm, float* am, hipStream_t s) {
    if (atom_np() == 3) return mask ? MS_ERR_UNSUPPORTED : launch_atom_np<C, NTP, NW, MODE, 3>(p, x, image, b0, b1, y, t, u, tm, nullptr, s);
    if constexpr (MODE != 0) {
        if (mask) return launch_atom_np<C, NTP, NW, MODE, 2, true>(p, x, image, b0, b1, y, t, u, tm, am, s);
    }
    return launch_atom_np<C, NTP, NW, MODE, 2>(p, x, image, b0, b1, y, t, u, tm, am, s);
}

// mode 0 / 1: forward (t, u: the saved activations, both or neither);  mode 2: backward data (x = g, u and tm read, t = gt out)
// mode & 4: sign words instead of the fp32 tensors u (and, backward, t) -- MASK
template <int C, int NTP, int NW>
int launch_atom(int mode, const AtomP& p, const float* x, const void* image, const float* b0, const float* b1, float* y,
                float* t, float* u, const float* tm, float* am, hipStream_t s) {
    const bool mask = (mode & 4) != 0;
    mode &= 3;
    if (mode == 2) return launch_atom_mode<C, NTP, NW, 2>(mask, p, x, image, b0, b1, y, t, u, tm, am, s);
    if (mode == 1) return launch_atom_mode<C, NTP, NW, 1>(mask, p, x, image, b0, b1, y, t, u, tm, am, s);
    return launch_atom_mode<C, NTP, NW, 0>(false, p, x, image, b0, b1, y, t, u, tm, am, s);
}

int dispatch_atom(int mode, const ms_atom_desc* d, const float* x, const void* image, const float* b0, const float* b1,
                  float* y, float* t, float* u, const float* tm, float* am, hipStream_t s) {
    AtomP p = {};
    p.B = d->B; p.C = d->C; p.L = d->L; p.dil = d->dil; p.slope = d->slope;
    // tile width: the widest tile whose grid still spreads over the chip; narrow tiles when the whole problem is a few
    // dozen tiles (B = 1 inference: latency, not throughput)
    const long long cols = (long long)d->B * d->L;
    switch (d->C) {
        case 32: return launch_atom<32, 128, 4>(mode, p, x, image, b0, b1, y, t, u, tm, am, s);
        case 64:
            if (cols < 124 * 128) return launch_atom<64, 64, 4>(mode, p, x, image, b0, b1, y, t, u, tm, am, s);
            return launch_atom<64, 128, 4>(mode, p, x, image, b0, b1, y, t, u, tm, am, s);
        case 128:
            if (cols < 60 * 128 && (mode & 3) != 2) return launch_atom<128, 32, 4>(mode, p, x, image, b0, b1, y, t, u, tm, am, s);
            // forward at full batch: 96-column tiles (92 outputs per pass over the 393 KB weight image instead of 60, three
            // accumulator tiles per wave, the rounds of the persistent grid fuller): 6 % faster than 64 columns in an interleaved
            // A/B at B = 32 (tools/scratch/probe_atom_cfg2.py); the backward pass, whose halo is the dilation, is 5-10 % slower on them
            if (cols >= 512 * 92 && (mode & 3) != 2) return launch_atom<128, 96, 4>(mode, p, x, image, b0, b1, y, t, u, tm, am, s);
            return launch_atom<128, 64, 4>(mode, p, x, image, b0, b1, y, t, u, tm, am, s);
        case 256:
            if (cols < 60 * 64 && (mode & 3) != 2) return launch_atom<256, 32, 8>(mode, p, x, image, b0, b1, y, t, u, tm, am, s);
            return launch_atom<256, 64, 8>(mode, p, x, image, b0, b1, y, t, u, tm, am, s);
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
    return 2 * atom_conv_image_u4(C, 3) * 16;                    // (the larger of the two piece schemes)
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
        t.job[i].np = atom_np();
        const int total = 2 * (C / 32) * (C / 16) * 3 * 64;
        blocks += (total + 255) / 256;
    }
    if (atom_np() == 2) {
        hipLaunchKernelGGL(k_atom_wmax, dim3((unsigned)(d->count * 2 * W_NPART)), dim3(256), 0, (hipStream_t)stream, t);
        MS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_atom_pack, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_residual_atom_publishes_amax(void) { return atom_np() == 2 ? 1 : 0; }

int ms_residual_atom_fwd(const ms_atom_desc* d, const float* x, const void* image, const float* b0, const float* b1,
                         float* y, float* t, float* y_act, float* amax, ms_stream_t stream) {
    if (!atom_ok(d)) return d ? MS_ERR_UNSUPPORTED : MS_ERR_INVALID_ARG;
    if (!x || !image || !b0 || !b1 || !y || ((t == nullptr) != (y_act == nullptr))) return MS_ERR_INVALID_ARG;
    if ((((uintptr_t)image) & 15) || (((uintptr_t)b0) & 15) || (((uintptr_t)b1) & 15)) return MS_ERR_INVALID_ARG;
    return dispatch_atom(t ? 1 : 0, d, x, image, b0, b1, y, t, y_act, nullptr, amax, (hipStream_t)stream);
}

size_t ms_residual_atom_sign_words(const ms_atom_desc* d) {
    if (!atom_ok(d) || atom_np() != 2) return 0;
    return (size_t)d->B * (d->C / 32) * 2 * d->L;
}

int ms_residual_atom_fwd_signs(const ms_atom_desc* d, const float* x, const void* image, const float* b0, const float* b1,
                               float* y, float* t, uint16_t* t_signs, uint16_t* y_signs, float* amax, ms_stream_t stream) {
    if (!ms_residual_atom_sign_words(d)) return d ? MS_ERR_UNSUPPORTED : MS_ERR_INVALID_ARG;
    if (!x || !image || !b0 || !b1 || !y || !t || !t_signs || !y_signs) return MS_ERR_INVALID_ARG;
    if ((((uintptr_t)image) & 15) || (((uintptr_t)b0) & 15) || (((uintptr_t)b1) & 15) || (((uintptr_t)t_signs) & 15) ||
        (((uintptr_t)y_signs) & 15))
        return MS_ERR_INVALID_ARG;
    return dispatch_atom(1 | 4, d, x, image, b0, b1, y, t, reinterpret_cast<float*>(y_signs), reinterpret_cast<const float*>(t_signs),
                         amax, (hipStream_t)stream);
}

int ms_residual_atom_bwd_data_signs(const ms_atom_desc* d, const float* gy, const uint16_t* y_signs, const uint16_t* t_signs,
                                    const void* image_bwd, float* gt, float* gx, float* amax, ms_stream_t stream) {
    if (!ms_residual_atom_sign_words(d) || !ms_residual_atom_bwd_supported(d)) return d ? MS_ERR_UNSUPPORTED : MS_ERR_INVALID_ARG;
    if (!gy || !y_signs || !t_signs || !image_bwd || !gt || !gx || (((uintptr_t)image_bwd) & 15) ||
        (((uintptr_t)t_signs) & 15) || (((uintptr_t)y_signs) & 15))
        return MS_ERR_INVALID_ARG;
    return dispatch_atom(2 | 4, d, gy, image_bwd, nullptr, nullptr, gx, gt,
                         reinterpret_cast<float*>(const_cast<uint16_t*>(y_signs)), reinterpret_cast<const float*>(t_signs), amax,
                         (hipStream_t)stream);
}

int ms_residual_atom_bwd_supported(const ms_atom_desc* d) {
    if (!ms_residual_atom_supported(d)) return 0;
    const char* sw = getenv("MSYNTH_ATOM_BWD");                  // tuning / test switch (0: the two backward-data launches)
    if (sw && atoi(sw) == 0) return 0;
    // (a tile yields NTP - 2 dil output columns: with 64-column tiles (128 / 256 channels) dilation 9 spends 28 % of both
    //  GEMMs on halo.  With six products per multiply that lost against the two row-tile launches; with three it wins:
    //  B = 32: 89 vs 114 us at 128 channels, 41 vs 69 us at 256 -- r04 takes every atom of the generator.)
    return 1;
}

int ms_residual_atom_bwd_data(const ms_atom_desc* d, const float* gy, const float* y_act, const float* t,
                              const void* image_bwd, float* gt, float* gx, float* amax, ms_stream_t stream) {
    if (!atom_ok(d)) return d ? MS_ERR_UNSUPPORTED : MS_ERR_INVALID_ARG;
    if (!gy || !y_act || !t || !image_bwd || !gt || !gx || (((uintptr_t)image_bwd) & 15)) return MS_ERR_INVALID_ARG;
    return dispatch_atom(2, d, gy, image_bwd, nullptr, nullptr, gx, gt, const_cast<float*>(y_act), t, amax, (hipStream_t)stream);
}

}  // extern "C"
