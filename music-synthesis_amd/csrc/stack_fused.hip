// Fused ResidualStack forward, inference: up to three ResidualAtoms back to back (reference util/modules.py:391-405, the
// generator's stacks with dilations 1, 3, 9) in ONE launch.  Between the atoms nothing travels to HBM: a workgroup reads the
// stack's input window once and writes the stack's output columns once -- a third of the traffic of three atom launches
// (atom_fused.hip), which is what bounds the 32- and 64-channel stacks.  Used where nothing has to be saved for a backward
// pass: the D-step's generator forward and plain inference (BASELINE config 2).
//
// Frame: a tile works on NTP columns f = 0 .. NTP - 1 <-> global column c0 - 16 + f, the SAME columns in every stage, so a
// lane's accumulator registers hold the same (channel, column) positions throughout and the residual of stage s + 1 is
// simply the lane's own stage-s output (exact fp32, in registers).  Each atom's two k = 3 convs read d + 1 columns either
// side, so the columns that are right shrink by d + 1 per side and stage: with dilations 1, 3, 9 the last stage is right on
// f = 16 .. NTP - 17 -- the NO = NTP - 32 columns the tile stores (75 % of both GEMMs at NTP = 128; the price of never
// exchanging halos between workgroups).  Columns outside are computed from incomplete (but finite) inputs and never
// stored; columns outside the row are forced to zero after every conv (each conv pads ITS input with zeros).
// LDS column of frame column f is f + 12: both GEMMs address their taps as f + 12 + (s - 1) step, the twelve columns either
// side are zero / stale-but-finite margins.
// Arithmetic: atom_fused.hip's NP = 2 scheme (block-scaled two-piece fp16, three products per multiply, fp32 accumulate),
// same weight images (ms_residual_atom_pack_multi, forward), one block scale per tile and GEMM operand.
#include "atom_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int NP = 2;
constexpr int SM = 12;          // LDS column of frame column 0
constexpr int HT = 16;          // frame column of the first stored column

struct StackP {
    int B, C, L, NO, tiles_per_row, nst;
    int d0, d1, d2;
    float slope;
    const u32x4* img[3];
    const float* b0[3];
    const float* b1[3];
};

template <int C, int NTP, int NW>
struct StackCfg {
    static constexpr int MS = C / 32, WGM = MS < NW ? MS : NW, TM = MS / WGM, WGN = NW / WGM, TN = NTP / 32 / WGN, NC = C / 16;
    static constexpr int NT = 64 * NW;
    static constexpr int NV = (NTP + 8) / 4, NV16 = (NV + 3) / 4;        // window = frame columns -4 .. NTP + 3
    static constexpr int ROUNDS = (NC * NV16 * 16 + NT - 1) / NT;
    static constexpr int NXA = NTP + 24, XRS = xrs<NP>(), XCS = NXA * XRS;
    static constexpr size_t LDS = (size_t)NC * XCS + (6 * C + 2 * NW) * sizeof(float);
    static_assert(MS >= 1 && TN >= 1 && TM * WGM == MS && TN * WGN * 32 == NTP, "tile shape");
};

template <int C, int NTP, int NW>
__global__ __launch_bounds__(64 * NW, C >= 128 ? 1 : 2) void k_stack_fwd(StackP p, const float* __restrict__ X, float* __restrict__ Y) {
    typedef StackCfg<C, NTP, NW> Cfg;
    constexpr int TM = Cfg::TM, WGN = Cfg::WGN, TN = Cfg::TN, NC = Cfg::NC, ROUNDS = Cfg::ROUNDS, NT = Cfg::NT;
    constexpr int XRS = Cfg::XRS, XCS = Cfg::XCS, NV = Cfg::NV, NV16 = Cfg::NV16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_stack[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const int L = p.L;
    constexpr unsigned OOB = 0xF0000000u;
    const int ntiles = p.B * p.tiles_per_row;

    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0x80000000u, 0x00020000);
    // (tensors addressed by frame columns: the descriptor starts HT elements early so that the lane part of an offset --
    //  the part the hardware range-checks -- is 4 f >= 0; lanes left of the row are masked before they could touch them)
    const auto rsXf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X) - HT, 0, 0x80000000u, 0x00020000);
    const auto rsYf = __builtin_amdgcn_make_buffer_rsrc(Y - HT, 0, 0x80000000u, 0x00020000);

    // ---- staging units of the input window (atom_fused.hip): 4 channels x one aligned 4-sample vector
    int u_goff[ROUNDS], u_t[ROUNDS], u_lcol[ROUNDS], u_lbase[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int u = tid + NT * r;
        const int grp = u >> 4, chunk = grp / NV16, vg = grp - chunk * NV16;
        const int cq = (u >> 2) & 3, v = vg * 4 + (u & 3);
        const bool in = chunk < NC && v < NV;
        u_t[r] = in ? 4 * v - 4 - HT : (1 << 28);                     // global column of the vector, relative to c0
        u_goff[r] = 4 * ((chunk * 16 + 4 * cq) * L + (4 * v - 4 - HT));
        u_lcol[r] = in ? SM - 4 + 4 * v : -1000;
        u_lbase[r] = chunk * XCS + cq * 8;
    }
    f32x4 rx[ROUNDS][4];
    auto load_x = [&](int tile) {
        const int bb = tile / p.tiles_per_row, cc0 = (tile - bb * p.tiles_per_row) * p.NO;
        const int base = 4 * bb * C * L;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int t = cc0 + u_t[r];
            const unsigned goff = (t >= 0 && t < L) ? (unsigned)(u_goff[r] + 4 * cc0) : OOB;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                rx[r][cc] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, goff, base + cc * 4 * L, 0));
        }
    };
    float* sbias = reinterpret_cast<float*>(smem_stack + NC * XCS);       // [stage][conv][C]
    float* smax1 = sbias + 6 * C;                                         // [NW]: |max| of the next first-GEMM operand
    float* smax2 = smax1 + NW;                                            // [NW]: |max| of this stage's second-GEMM operand
    auto publish_window_max = [&]() {
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
                if (i < 0) continue;
                float c4[4] = {rx[r][0][e] * S, rx[r][1][e] * S, rx[r][2][e] * S, rx[r][3][e] * S};
                uint2 o2[NP];
                split_quad<NP>(c4, o2);
                unsigned char* dst = smem_stack + u_lbase[r] + i * XRS;
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) *reinterpret_cast<uint2*>(dst + pp * 32) = o2[pp];
            }
        }
    };

    // ---- A fragments: image -> registers, one chunk ahead through two buffers (atom_fused.hip); the image of the stage
    // travels in the descriptor
    u32x4 fa[2][TM][3][NP];
    const int a_voff = lane * 16 + wm * TM * NC * 3 * NP * 1024;
    auto image_of = [&](int st) {
        const u32x4* im = st == 0 ? p.img[0] : (st == 1 ? p.img[1] : p.img[2]);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(im), 0, 0x80000000u, 0x00020000);
    };
    auto load_a = [&](int q, auto rs) {
        const int conv = q / NC, chunk = q % NC;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int pp = 0; pp < NP; ++pp)
                    fa[q & 1][i][s][pp] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rs, a_voff, (int)(conv * atom_conv_image_u4(C, NP) * 16) + ((i * NC + chunk) * 3 * NP + s * NP + pp) * 1024, 0));
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
    // one GEMM over the LDS-resident operand: output frame column n reads LDS columns n + SM + (s - 1) step.  rs_cur: this
    // stage's image; rs_next: the image the chunk behind this stage's last one comes from (more: there is one)
    auto gemm = [&](auto gc, int step, bool more, auto rs_cur, auto rs_next) {
        constexpr int g = decltype(gc)::value;
        const unsigned char* Bs = smem_stack + ((wn * TN) * 32 + l31 + SM - step) * XRS + h * 16;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const int q = g * NC + ch, qn = q + 1;
            if (qn < 2 * NC) load_a(qn, rs_cur);
            else if (more) load_a(0, rs_next);
            u32x4 fb[2][TN][NP];
            auto fragb = [&](int s, u32x4 (&dst)[TN][NP]) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp)
                        dst[j][pp] = *reinterpret_cast<const u32x4*>(Bs + ch * XCS + (j * 32 + s * step) * XRS + pp * 32);
            };
            constexpr bool FB2 = C != 32;
            if (FB2) fragb(0, fb[0]);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (FB2 && s + 1 < 3) fragb(s + 1, fb[(s + 1) & 1]);
                if (!FB2) fragb(s, fb[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const f16x8 ah = __builtin_bit_cast(f16x8, fa[q & 1][i][s][0]), al = __builtin_bit_cast(f16x8, fa[q & 1][i][s][1]);
                        const f16x8 bh = __builtin_bit_cast(f16x8, fb[s & 1][j][0]), bl = __builtin_bit_cast(f16x8, fb[s & 1][j][1]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);     // smallest products first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    int o_lane[TN];                                  // per-lane part of a result offset (bytes): frame column, channel half h
#pragma unroll
    for (int j = 0; j < TN; ++j) o_lane[j] = 4 * ((wn * TN + j) * 32 + l31 + 4 * h * L);

    // one-time: margins (and everything else) of the window to zero, biases into LDS
    for (int i = tid; i < NC * XCS / 16; i += NT) reinterpret_cast<u32x4*>(smem_stack)[i] = u32x4{0u, 0u, 0u, 0u};
    for (int i = tid; i < 6 * C; i += NT) {
        const int st = i / (2 * C), k = i - st * 2 * C;
        float v = 0.f;
        if (st < p.nst) v = k < C ? (st == 0 ? p.b0[0] : st == 1 ? p.b0[1] : p.b0[2])[k] : (st == 0 ? p.b1[0] : st == 1 ? p.b1[1] : p.b1[2])[k - C];
        sbias[i] = v;
    }
    int tile = blockIdx.x;
    if (tile < ntiles) load_x(tile);
    load_a(0, image_of(0));
    publish_window_max();
    __syncthreads();

    float yres[TM][TN][16];                          // this lane's positions of the running stack value (the residual)
    for (; tile < ntiles; tile += gridDim.x) {
        const int b = tile / p.tiles_per_row, c0 = (tile - b * p.tiles_per_row) * p.NO;
        const int no = p.NO;
        const int wcol = wn * TN * 32;
        const int base = 4 * (b * C * L + c0);
        int L4;
        asm volatile("s_mov_b32 %0, %1" : "=s"(L4) : "s"(4 * L));
        const int nxt = tile + gridDim.x;
        float S1, iS1;
        block_scale(read_max(smax1), S1, iS1);
        store_x(S1);
        zero_acc();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                             // window staged
#pragma unroll 1
        for (int st = 0; st < p.nst; ++st) {
            const int d = st == 0 ? p.d0 : (st == 1 ? p.d1 : p.d2);
            const bool last = st + 1 == p.nst;
            const auto rs_cur = image_of(st), rs_next = image_of(last ? 0 : st + 1);
            const float* bias = sbias + st * 2 * C;
            // 1 / S_w of the stage's two weight images (atom_common.h: the pack leaves them in the image's tail)
            const float* wtail = reinterpret_cast<const float*>((st == 0 ? p.img[0] : (st == 1 ? p.img[1] : p.img[2])) +
                                                                2 * atom_conv_image_u4(C, 2)) + 2 * W_NPART;
            const float winv0 = wtail[0], winv1 = wtail[1];
            gemm(I0{}, d, true, rs_cur, rs_cur);
            if (last && nxt < ntiles) load_x(nxt);                   // travels under the t epilogue and the last GEMM
            __builtin_amdgcn_sched_barrier(0);

            // ---- t epilogue: accumulators -> lrelu(conv_d + b0), zero outside the row, in place; block maximum
            const float k1 = iS1 * winv0;
            float tmax = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int gc = c0 - HT + wcol + j * 32 + l31;
                    const bool inrow = gc >= 0 && gc < L;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int ch0 = (wm * TM + i) * 32 + 8 * g + 4 * h;
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + ch0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v = acc[i][j][4 * g + q] * k1 + bv[q];
                            v = inrow ? (v > 0.f ? v : v * p.slope) : 0.f;
                            acc[i][j][4 * g + q] = v;
                            tmax = fmaxf(tmax, fabsf(v));
                        }
                    }
                }
            tmax = wave_max(tmax);
            if (lane == 0) smax2[wid] = tmax;
            __syncthreads();                                         // every wave is done with the window: t overwrites it
            float S2, iS2;
            block_scale(read_max(smax2), S2, iS2);
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
                        for (int q = 0; q < 4; ++q) e[q] = acc[i][j][4 * g + q] * S2;
                        uint2 o2[NP];
                        split_quad<NP>(e, o2);
                        unsigned char* dst = smem_stack + ((col + SM) * XRS + 8 * h) + ((chs >> 4) * XCS + (chs & 15) * 2);
#pragma unroll
                        for (int pp = 0; pp < NP; ++pp) *reinterpret_cast<uint2*>(dst + pp * 32) = o2[pp];
                    }
                }
            if (st == 0) {                                           // the stack's input at this lane's positions (L2-warm)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int gc = c0 - HT + wcol + j * 32 + l31;
                        const unsigned o_x = (gc >= 0 && gc < L) ? (unsigned)o_lane[j] : OOB;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            yres[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                rsXf, o_x, base + ((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2)) * L4, 0));
                    }
            }
            zero_acc();
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();                                         // t tile complete
            gemm(I1{}, 1, !last || nxt < ntiles, rs_cur, rs_next);

            // ---- y = residual + lrelu(conv1 + b1)
            const float k2 = iS2 * winv1;
            float ymax = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = wcol + j * 32 + l31, gc = c0 - HT + col;
                    const bool inrow = gc >= 0 && gc < L;
                    const unsigned oy = (last && inrow && col >= HT && col < HT + no) ? (unsigned)o_lane[j] : OOB;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int chs = (wm * TM + i) * 32 + 8 * g, ch0 = chs + 4 * h;
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + C + ch0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v = acc[i][j][4 * g + q] * k2 + bv[q];
                            v = v > 0.f ? v : v * p.slope;
                            v = inrow ? v + yres[i][j][4 * g + q] : 0.f;      // (outside the row: the next conv's zero padding)
                            yres[i][j][4 * g + q] = v;
                            ymax = fmaxf(ymax, fabsf(v));
                            if (last) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsYf, oy, base + (chs + q) * L4, 0);
                        }
                    }
                }
            if (!last) {
                ymax = wave_max(ymax);
                if (lane == 0) smax1[wid] = ymax;
                __syncthreads();                                     // every wave is done with the t tile: y overwrites it
                block_scale(read_max(smax1), S1, iS1);
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
                            for (int q = 0; q < 4; ++q) e[q] = yres[i][j][4 * g + q] * S1;
                            uint2 o2[NP];
                            split_quad<NP>(e, o2);
                            unsigned char* dst = smem_stack + ((col + SM) * XRS + 8 * h) + ((chs >> 4) * XCS + (chs & 15) * 2);
#pragma unroll
                            for (int pp = 0; pp < NP; ++pp) *reinterpret_cast<uint2*>(dst + pp * 32) = o2[pp];
                        }
                    }
                zero_acc();
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                                     // next stage's operand staged
            }
        }
        if (nxt < ntiles) publish_window_max();                      // (waits for the next tile's window)
        __syncthreads();                                             // the t tile is dead: the next window may overwrite it
    }
}

constexpr int MAX_DEV = 64;

template <int C, int NTP, int NW>
int launch_stack(StackP p, const float* x, float* y, hipStream_t s) {
    typedef StackCfg<C, NTP, NW> Cfg;
    p.NO = NTP - 2 * HT;
    p.tiles_per_row = (p.L + p.NO - 1) / p.NO;
    const size_t lds = Cfg::LDS;
    if (lds > 158 * 1024) return MS_ERR_UNSUPPORTED;
    const void* fn = reinterpret_cast<const void*>(&k_stack_fwd<C, NTP, NW>);
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
    const dim3 grid((unsigned)(ntiles < slots ? ntiles : slots));
    ms_note_kernel(3, "k_stack_fwd<%d, %d, %d>", C, NTP, NW);
    hipLaunchKernelGGL((k_stack_fwd<C, NTP, NW>), grid, dim3(64 * NW), lds, s, p, x, y);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

bool stack_ok(const ms_stack_desc* d) {
    if (!d || d->B <= 0 || d->L <= 0 || d->count < 1 || d->count > MS_STACK_MAX) return false;
    // (128 channels, one eight-wave workgroup per CU: measured r04 at B = 32, L = 2048: 169 us against 157 for the three atom
    //  launches, 45 against 37 at B = 1 -- with 96 KiB of window per workgroup nothing overlaps the stages; not dispatched)
    if (d->C != 32 && d->C != 64) return false;
    if (d->L % 4) return false;
    if ((long long)d->B * d->C * d->L * 4 >= (1ll << 31)) return false;
    static const bool np3 = getenv("MSYNTH_ATOM_NP") && atoi(getenv("MSYNTH_ATOM_NP")) == 3;     // (three-piece images: atom_fused.hip)
    if (np3) return false;
    int halo = 0;
    for (int i = 0; i < d->count; ++i) {
        if (d->dil[i] < 1 || d->dil[i] > 9) return false;
        halo += d->dil[i] + 1;
    }
    return halo <= HT && d->dil[0] <= 4;            // (the window loaded from memory reaches 4 columns beyond the frame)
}

}  // namespace

extern "C" {

int ms_residual_stack_supported(const ms_stack_desc* d) {
    const char* sw = getenv("MSYNTH_STACK");                     // tuning / test switch (0: one launch per atom)
    if (sw && atoi(sw) == 0) return 0;
    return stack_ok(d) ? 1 : 0;
}

int ms_residual_stack_fwd(const ms_stack_desc* d, const float* x, const void* const* images, const float* const* b0,
                          const float* const* b1, float* y, ms_stream_t stream) {
    if (!stack_ok(d)) return d ? MS_ERR_UNSUPPORTED : MS_ERR_INVALID_ARG;
    if (!x || !y || !images || !b0 || !b1 || x == y) return MS_ERR_INVALID_ARG;
    StackP p = {};
    p.B = d->B; p.C = d->C; p.L = d->L; p.nst = d->count; p.slope = d->slope;
    p.d0 = d->dil[0]; p.d1 = d->count > 1 ? d->dil[1] : 1; p.d2 = d->count > 2 ? d->dil[2] : 1;
    for (int i = 0; i < 3; ++i) {
        const int k = i < d->count ? i : 0;
        if (!images[k] || !b0[k] || !b1[k] || (((uintptr_t)images[k]) & 15) || (((uintptr_t)b0[k]) & 15) || (((uintptr_t)b1[k]) & 15))
            return MS_ERR_INVALID_ARG;
        p.img[i] = (const u32x4*)images[k]; p.b0[i] = b0[k]; p.b1[i] = b1[k];
    }
    hipStream_t s = (hipStream_t)stream;
    switch (d->C) {
        case 32: return launch_stack<32, 128, 4>(p, x, y, s);
        case 64: return launch_stack<64, 128, 4>(p, x, y, s);
        default: return MS_ERR_UNSUPPORTED;
    }
}

}  // extern "C"
