// Grouped strided conv (the discriminator's k41 / stride-4 / 4-channels-per-group layers, reference
// discriminator/full.py:15-18) on the bf16 matrix pipe with fp32-exact operands: second generation
// of gconv_mfma.hip, same operand split as conv_rows3.hip (x = x1 + x2 + x3 in bf16 pieces, the six
// partial products with i + j <= 4 accumulated in fp32).
//
// Forward.  Per group the layer is a GEMM  y[co, t] = sum_{ci, j} w[co, ci, j] x[ci, 4t + j - pad],
// M = 16 outputs (4 for the 256-group layer), contraction = 4 channels x 41 taps.
// v_mfma_f32_16x16x32_bf16 contracts 4 lane groups x 8 elements: lane group = input channel ci, the
// 8 elements = 8 CONSECUTIVE taps j = 8G .. 8G+7 (6 tap groups, taps 41..47 carry zero weights), so a
// lane's B fragment is 8 consecutive input samples starting at 4t + 8G.
// The 64 outputs of a unit are four INTERLEAVED column tiles, t = 4n + s (tile s, column n):
//   * fragment start = 16n + 4s + 8G: always a multiple of 4 samples, so with the inputs kept in LDS as
//     8-byte quads (4 samples x bf16) a fragment is two aligned ds_read_b64 -- no phase-split gather;
//   * tiles s and s+2 read the same quads one tap group apart: 14 fragment reads feed 24 (tile, G) steps;
//   * a lane's accumulator row r of the four tiles is y[co][4n .. 4n+3]: one 16-byte store per row.
// LDS image of a wave (private, no workgroup barrier): [piece][row = (ci, segment)][Q mod 4][Q div 4]
// quads; the 16 lanes of a column tile read consecutive quads, the rows sit 16 / 8 / 4 quads apart
// modulo the 32-quad bank window: conflict-free for ds_read_b64.
// Short rows (the coarse scales: Lout = 9 .. 65) are covered by RB = 2 / 4 segments of 32 / 16 outputs
// per unit, taken from consecutive (batch row, segment) pairs.
#include "ms_common.h"
#include "gconv_mfma.h"
#include <stdint.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // a 16-byte access at any 4-byte aligned address
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int GK = 41, GS = 4, GCG = 4;
// The weights of a wave's group are split in registers under the wave's own power-of-two scale S_w (their largest magnitude goes to
// [2^12, 2^13): weights of any magnitude; r04 took the pieces of 64 w and overflowed to inf from |w| >= 2^9).

// Block-scaled two-piece fp16 split (r04, atom_fused.hip): (a, b) scaled into fp16's range by the caller ->
// a = h.lo + l.lo to 22 significand bits (block maximum in [2^8, 2^15)); products h h' + h l' + l h' into one fp32
// accumulator: three MFMAs instead of six.
__device__ __forceinline__ void split_pair2(float a, float b, unsigned& h, unsigned& l) {
    const f32x2 v = {a, b};
    const f16x2 hi = __builtin_convertvector(v, f16x2);
    const f16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x2), f16x2);
    h = __builtin_bit_cast(unsigned, hi);
    l = __builtin_bit_cast(unsigned, lo);
}

// S = 2^k with m S in [2^14, 2^15), and 1 / S (exact powers of two); 1 for a zero / denormal-range / non-finite maximum
__device__ __forceinline__ void block_scale(float m, float& S, float& invS) {
    const unsigned eb = (__builtin_bit_cast(unsigned, m) >> 23) & 0xFFu;
    const bool ok = eb >= 16u && eb <= 250u;
    S = ok ? __builtin_bit_cast(float, (268u - eb) << 23) : 1.f;
    invS = ok ? __builtin_bit_cast(float, (eb - 14u) << 23) : 1.f;
}

// largest value over the wave, wave-uniform: 16 lanes by DPP (quad swaps, half-row mirror, row mirror), the four rows by readlane
__device__ __forceinline__ float wave_max_dpp(float m) {
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0xB1, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x4E, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x141, 0xF, 0xF, true)));
    m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), 0x140, 0xF, 0xF, true)));
    const int b = __builtin_bit_cast(int, m);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
constexpr int NG = 6;                         // tap groups of 8

// S = 2^k with m S in [2^12, 2^13), and 1 / S; 1 for a zero / denormal-range / non-finite maximum
__device__ __forceinline__ void weight_scale(float m, float& S, float& invS) {
    const unsigned eb = (__builtin_bit_cast(unsigned, m) >> 23) & 0xFFu;
    const bool ok = eb >= 16u && eb <= 250u;
    S = ok ? __builtin_bit_cast(float, (266u - eb) << 23) : 1.f;
    invS = ok ? __builtin_bit_cast(float, (eb - 12u) << 23) : 1.f;
}

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

template <int RB>
struct GF {
    static constexpr int NR = 16 / RB;               // MFMA columns per segment
    static constexpr int TS = 64 / RB;               // outputs per segment
    static constexpr int SUBP = NR + 3;              // quads per (row, Q mod 4) sub-row
    static constexpr int NQR = 4 * SUBP;             // quads staged per (segment, channel) row
    static constexpr int RP = RB == 1 ? 80 : (RB == 2 ? 56 : 28);   // row pitch in quads (see header)
    static constexpr int ROWS = 4 * RB;
    static constexpr int PIECE_BYTES = ROWS * RP * 8;
    static constexpr int WAVE_BYTES = 2 * PIECE_BYTES;               // forward: two fp16 pieces
    static constexpr int NIT = (ROWS * NQR + 63) / 64;              // staged quads per lane
    static_assert(NQR <= RP, "row pitch");
    static_assert(NQR < RP || (ROWS * NQR) % 64 == 0, "a spare quad for the lanes without an item");
};

// Per unit: loads (issued one unit ahead) -> split -> LDS image -> 7 tap-group steps of MFMAs -> stores.
// PMC on the first version: wave cycles = MFMA cycles + vector-ALU cycles (an MFMA leaves room for ~2 vector
// instructions in its 16 cycles, nothing more overlaps), so the per-unit vector work is kept small: interior
// units load through lane-constant offsets + a scalar unit base, fragments arrive as whole operands
// (ds_read2_b64), LeakyReLU is mul + max.  (A software-pipelined variant -- next unit split and written to a
// second LDS image between the tap-group steps, two register sets of loads in flight -- measured 5-10 %
// SLOWER at 4-8 units per wave: its longer prologue costs more than the in-wave overlap returns; removed.)
// All kernels of this file take their LDS from ONE dynamic array: a "parts" kernel (below) runs the bodies of up to three
// instantiations in one launch, one per workgroup range, and they must not each own a static allocation.
extern __shared__ __attribute__((aligned(16))) unsigned char g3_smem[];

// bx / nbx: this workgroup's index and the number of workgroups that share the (part of the) problem -- blockIdx.x and
// gridDim.x in a single launch, offsets into a workgroup range in a parts launch.
template <int RB, bool VEC, bool VOUT>
__device__ __forceinline__ void gconv_fwd_body(const ConvP& p, int spr, int nseg, int nunits,
                                               const float* __restrict__ x,
                                               const float* __restrict__ w,
                                               const float* __restrict__ bias,
                                               float* __restrict__ y, int bx, int nbx) {
    using C = GF<RB>;
    unsigned char* const lds = g3_smem;               // 4 * C::WAVE_BYTES
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = blockIdx.y;
    const int n = lane & 15, kg = lane >> 4;
    const int sgl = n / C::NR, np = n % C::NR;
    const unsigned wv_off = wid * C::WAVE_BYTES;                                  // the wave's image within lds[]
    const unsigned lw = (unsigned)(uintptr_t)lds + wv_off;                        // ... as an LDS byte address
    const unsigned rb0 = lw + ((kg * RB + sgl) * C::RP + np) * 8;               // this lane's fragment base, piece 0

    // ---- staging items of this lane: quad Q of row (segment sg, channel ci)
    int it_lds[C::NIT];            // LDS byte offset of the quad in piece 0
    unsigned it_x[C::NIT];         // ci * Lin
    int it_qs[C::NIT];             // 4 Q | sg  (< 0: no item)
    unsigned it_c[C::NIT];         // interior units: byte offset of the quad relative to the unit's first sample
#pragma unroll
    for (int i = 0; i < C::NIT; ++i) {
        const int idx = lane + 64 * i;
        const int row = idx / C::NQR, Q = idx - row * C::NQR;
        const int sg = row >> 2, ci = row & 3;
        const bool act = row < C::ROWS;
        // lanes past the last item stage zeros into a spare quad of row 0 (no divergent branch in the pipeline)
        it_lds[i] = act ? ((ci * RB + sg) * C::RP + (Q & 3) * C::SUBP + (Q >> 2)) * 8 : C::NQR * 8;
        it_x[i] = (unsigned)(ci * p.Lin);
        it_qs[i] = act ? 4 * Q + sg : -4;
        it_c[i] = act ? (unsigned)(ci * p.Lin + 4 * Q) * 4u : 0xF0000000u;
    }
    const int wstride = nbx * 4;
    const int sstride = wstride * RB;
    const int dsb = sstride / spr, dst = sstride - dsb * spr;     // segment index += sstride, in (b, ts) form
    struct Cur { int seg, b, ts; };
    auto advance = [&](Cur& c) {
        c.seg += sstride; c.b += dsb; c.ts += dst;
        if (c.ts >= spr) { c.ts -= spr; ++c.b; }
    };

    // Inputs are read through a buffer descriptor: a quad that lies outside its row (the zero padding) or
    // behind the last segment carries an out-of-range offset and the hardware range check returns 0.0f.
    constexpr unsigned OOB = 0xF0000000u;
    // (true size: the unaligned quad across the END of the tensor's last row must not touch memory behind the tensor -- the
    //  range check returns 0.0 for those dwords)
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 4u * (unsigned)(p.B * p.Cin * p.Lin), 0x00020000);
    auto gload = [&](auto& xr, const Cur& c) {
        if (RB == 1) {
            // interior unit (every quad inside its row): lane-constant offsets + scalar unit base, no vector math (rows of
            // any length: a quad that is not 16-byte aligned is one unaligned 16-byte access, which the memory pipe takes)
            const int s0 = c.ts * (C::TS * GS) - p.pad;
            if (c.seg < nseg && s0 >= 0 && s0 + 4 * C::NQR <= p.Lin) {
                const unsigned ub = (unsigned)__builtin_amdgcn_readfirstlane(((c.b * p.Cin + g * GCG) * p.Lin + s0) * 4);
#pragma unroll
                for (int i = 0; i < C::NIT; ++i)
                    xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, it_c[i], ub, 0));
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < C::NIT; ++i) {
            const int sg = RB == 1 ? 0 : (it_qs[i] & 3), Q4 = it_qs[i] & ~3;
            int ts = c.ts + sg, b = c.b;
            if (spr == 1) { b += ts; ts = 0; }               // whole rows: segment index = batch row
            else if (ts >= spr) { ts -= spr; ++b; }           // spr >= RB otherwise (pick_rb): one wrap at most
            const bool segok = c.seg + sg < nseg && it_qs[i] >= 0;
            const int sidx = ts * (C::TS * GS) - p.pad + Q4;          // first sample of the quad within its row
            const unsigned rowoff = (unsigned)((b * p.Cin + g * GCG) * p.Lin) + it_x[i];
            if (VEC) {
                const bool ok = segok && sidx >= 0 && sidx < p.Lin;
                xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rsX, ok ? (rowoff + (unsigned)sidx) * 4u : OOB, 0, 0));
            } else {
                // rows whose length is not a multiple of 4: the quad is still ONE (unaligned) 16-byte load; sidx is a multiple of
                // 4, so a quad is wholly in front of its row or starts inside it, and only the quad across the row END holds
                // samples of the next row, which are cleared
                const bool ok = segok && sidx >= 0 && sidx < p.Lin;
                xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rsX, ok ? (rowoff + (unsigned)sidx) * 4u : OOB, 0, 0));
#pragma unroll
                for (int e = 1; e < 4; ++e) xr[i][e] = sidx + e < p.Lin ? xr[i][e] : 0.f;
            }
        }
    };
    // registers -> LDS: scaled by the unit's power-of-two block scale, split into two fp16 pieces, one 8-byte quad per piece
    auto stage_item = [&](const f32x4& v, int lds_off, float S) {
        unsigned h0, l0, h1, l1;
        split_pair2(v[0] * S, v[1] * S, h0, l0);
        split_pair2(v[2] * S, v[3] * S, h1, l1);
        unsigned char* d = lds + wv_off + lds_off;
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(d + C::PIECE_BYTES) = make_uint2(l0, l1);
    };

    int unit = bx * 4 + wid;
    Cur ld, ep;
    ld.seg = unit * RB; ld.b = ld.seg / spr; ld.ts = ld.seg - ld.b * spr;
    ep = ld;
    f32x4 xp[C::NIT];
    gload(xp, ld); advance(ld);

    // ---- weight fragments: lane (co = n, ci = kg) holds S_w w[g*Og + co][ci][8G .. 8G+7] in two fp16 pieces
    f16x8 A[NG][2];
    float WS = 1.f, iWS = 1.f;
    {
        const bool ok = n < p.Og;
        const float* wr = w + ((size_t)(g * p.Og + (ok ? n : 0)) * GCG + kg) * GK;
        float wv[NG * 8];
        float wm = 0.f;
#pragma unroll
        for (int j = 0; j < NG * 8; ++j) {
            wv[j] = (j < GK && ok) ? wr[j] : 0.f;
            wm = fmaxf(wm, fabsf(wv[j]));
        }
        weight_scale(wave_max_dpp(wm), WS, iWS);
#pragma unroll
        for (int j = 0; j < NG * 8; ++j) wv[j] *= WS;
#pragma unroll
        for (int G = 0; G < NG; ++G) {
            u32x4 h, l;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned hh, ll;
                split_pair2(wv[8 * G + 2 * q], wv[8 * G + 2 * q + 1], hh, ll);
                h[q] = hh; l[q] = ll;
            }
            A[G][0] = __builtin_bit_cast(f16x8, h);
            A[G][1] = __builtin_bit_cast(f16x8, l);
        }
    }
    const float eslope = p.act == MS_ACT_LRELU ? p.slope : 1.f;      // in [0, 1] (msg3_fwd_applicable)
    float bq[4];
    bool rok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rok[r] = kg * 4 + r < p.Og;
        bq[r] = (bias && rok[r]) ? bias[g * p.Og + kg * 4 + r] : 0.f;
    }

    auto unit_mfma_store = [&](float kscale) {
        unsigned rbo[2];
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) rbo[pc] = rb0 + pc * C::PIECE_BYTES;
        f32x4 acc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Fragment reads are hand-issued ds_read2_b64 (two quads -> one 4-register operand; the compiler's own
        // pairing of the 8-byte reads put the halves in unrelated registers and paid ~140 v_mov per unit).  The
        // reads of step H+1 are in flight during the MFMAs of step H: counted wait on the older four.
        u32x4 f[2][2][2];
#define MS_G3_READ(H_, buf)                                                                                     \
    _Pragma("unroll") for (int par = 0; par < 2; ++par) {                                                       \
        const int c0 = 2 * (H_) + par, c1 = c0 + 1;                                                             \
        _Pragma("unroll") for (int pc = 0; pc < 2; ++pc)                                                        \
            asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3"                                            \
                         : "=v"(f[buf][par][pc])                                                                \
                         : "v"(rbo[pc]), "n"((c0 & 3) * C::SUBP + (c0 >> 2)), "n"((c1 & 3) * C::SUBP + (c1 >> 2))); \
    }
        MS_G3_READ(0, 0);
#pragma unroll
        for (int H = 0; H <= NG; ++H) {
            const int cur = H & 1;
            if (H < NG) {
                MS_G3_READ(H + 1, cur ^ 1);
                asm volatile("s_waitcnt lgkmcnt(4)"
                             : "+v"(f[cur][0][0]), "+v"(f[cur][0][1]), "+v"(f[cur][1][0]), "+v"(f[cur][1][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(f[cur][0][0]), "+v"(f[cur][0][1]), "+v"(f[cur][1][0]), "+v"(f[cur][1][1]));
            }
            // three partial products per tile, smallest first: (a_h b_l), (a_h b_h) ... (a_l b_h); the tiles' accumulator chains
            // are interleaved so that dependent MFMAs sit 2-4 issues apart
            const f16x8 f0h = __builtin_bit_cast(f16x8, f[cur][0][0]), f0l = __builtin_bit_cast(f16x8, f[cur][0][1]);
            const f16x8 f1h = __builtin_bit_cast(f16x8, f[cur][1][0]), f1l = __builtin_bit_cast(f16x8, f[cur][1][1]);
            if (H < NG) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[H][0], f0l, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[H][0], f1l, acc[1], 0, 0, 0);
            }
            if (H >= 1) {
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[H - 1][0], f0l, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[H - 1][0], f1l, acc[3], 0, 0, 0);
            }
            if (H < NG) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[H][0], f0h, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[H][0], f1h, acc[1], 0, 0, 0);
            }
            if (H >= 1) {
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[H - 1][0], f0h, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[H - 1][0], f1h, acc[3], 0, 0, 0);
            }
            if (H < NG) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[H][1], f0h, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[H][1], f1h, acc[1], 0, 0, 0);
            }
            if (H >= 1) {
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[H - 1][1], f0h, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[H - 1][1], f1h, acc[3], 0, 0, 0);
            }
        }
#undef MS_G3_READ

        // ---- epilogue: row r of the four tiles = 4 consecutive outputs
        int ets = ep.ts + sgl, eb = ep.b;
        if (spr == 1) { eb += ets; ets = 0; }
        else if (ets >= spr) { ets -= spr; ++eb; }
        const bool eok = ep.seg + sgl < nseg;
        const int t = ets * C::TS + 4 * np;
        if (eok && t < p.Lout) {
            float* yr = y + ((size_t)eb * p.Cout + (size_t)g * p.Og + kg * 4) * p.Lout + t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!rok[r]) continue;
                float v[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float pre = fmaf(acc[s][r], kscale, bq[r]);     // undo the block and weight scales
                    v[s] = fmaxf(pre, pre * eslope);                // LeakyReLU (slope in [0, 1]) / identity
                }
                if (VOUT) {
                    *reinterpret_cast<float4*>(yr + (size_t)r * p.Lout) = make_float4(v[0], v[1], v[2], v[3]);
                } else if (t + 3 < p.Lout) {          // any row length: one unaligned 16-byte store inside the row
                    *reinterpret_cast<f32x4u*>(yr + (size_t)r * p.Lout) = (f32x4u){v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        if (t + s < p.Lout) yr[(size_t)r * p.Lout + s] = v[s];
                }
            }
        }
        advance(ep);
    };

    for (; unit < nunits; unit += wstride) {
        // the unit's block scale: its largest input magnitude goes to 2^14 (a wave owns the whole unit: no barrier)
        float um = 0.f;
#pragma unroll
        for (int i = 0; i < C::NIT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) um = fmaxf(um, fabsf(xp[i][e]));
        float S, invS;
        block_scale(wave_max_dpp(um), S, invS);
#pragma unroll
        for (int i = 0; i < C::NIT; ++i) stage_item(xp[i], it_lds[i], S);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (unit + wstride < nunits) gload(xp, ld);     // next unit's inputs: in flight across the MFMA loop
        advance(ld);
        unit_mfma_store(invS * iWS);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int RB, bool VEC, bool VOUT>
__global__ __launch_bounds__(256, 2) void k_gconv_split_fwd(ConvP p, int spr, int nseg, int nunits,
                                                           const float* __restrict__ x,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ y) {
    gconv_fwd_body<RB, VEC, VOUT>(p, spr, nseg, nunits, x, w, bias, y, blockIdx.x, gridDim.x);
}

// The SAME layer (weights, bias) over up to three inputs of different batch size / length in one launch: the shared
// discriminator on x, pool(x), pool(pool(x)) (reference discriminator/melgan.py:13-27).  Workgroups [bx0[i], bx0[i + 1]) run
// part i exactly as a launch of its own would.  Part 0 is the full-rate scale (16-byte shaped: checked by the launcher),
// the pooled parts have odd lengths and take the dword paths.
struct G3Parts {
    ConvP p;                         // the layer; B / Lin / Lout are taken per part
    int count, bx0[MS_CONV_PARTS_MAX + 1];
    int B[MS_CONV_PARTS_MAX], Lin[MS_CONV_PARTS_MAX], Lout[MS_CONV_PARTS_MAX];
    int u0[MS_CONV_PARTS_MAX], u1[MS_CONV_PARTS_MAX], u2[MS_CONV_PARTS_MAX];       // per-part launch integers of the pass
    const float* a[MS_CONV_PARTS_MAX];
    const float* b[MS_CONV_PARTS_MAX];
    const float* c[MS_CONV_PARTS_MAX];
    float* o[MS_CONV_PARTS_MAX];
};

__device__ __forceinline__ int g3_part_of(const G3Parts& q, int bx) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < MS_CONV_PARTS_MAX; ++k)
        if (k < q.count && bx >= q.bx0[k]) i = k;
    return __builtin_amdgcn_readfirstlane(i);
}

__device__ __forceinline__ ConvP g3_part_conv(const G3Parts& q, int i) {
    ConvP p = q.p;
    p.B = q.B[i]; p.Lin = q.Lin[i]; p.Lout = q.Lout[i];
    return p;
}

template <int RB0, int RB1, int RB2>
__global__ __launch_bounds__(256, 2) void k_gconv_split_fwd_parts(G3Parts q, const float* __restrict__ w,
                                                                 const float* __restrict__ bias) {
    const int bx = blockIdx.x, i = g3_part_of(q, bx);
    const ConvP p = g3_part_conv(q, i);
    const int rbx = bx - q.bx0[i], nbx = q.bx0[i + 1] - q.bx0[i];
    if (i == 0) gconv_fwd_body<RB0, true, true>(p, q.u0[0], q.u1[0], q.u2[0], q.a[0], w, bias, q.o[0], rbx, nbx);
    else if (i == 1) gconv_fwd_body<RB1, false, false>(p, q.u0[1], q.u1[1], q.u2[1], q.a[1], w, bias, q.o[1], rbx, nbx);
    else gconv_fwd_body<RB2, false, false>(p, q.u0[2], q.u1[2], q.u2[2], q.a[2], w, bias, q.o[2], rbx, nbx);
}

// ---------------------------------------------------------------- backward weight
// gw[co, ci, k] = sum_{b, t} gp[b, co, t] * x[b, g*4 + ci, 4t + k - pad]   (gp = gy * act'(y_act))
// Per group and input channel: M = co (16), N = tap k (3 tiles of 16: 41 taps + 7 discarded columns), the MFMA
// contraction runs over 32 consecutive outputs t.  The B operand B[t][k] = x[4t + k - pad] is a Toeplitz view
// of the LINEAR input row: row t of a 16-tap tile is 16 consecutive samples starting at 4t + k0 - pad, 8 bytes
// (4 bf16) further on for every t -- exactly what gfx950's transposing LDS read wants (ds_read_b64_tr_b16: 16
// lanes supply 4 row addresses x 4 column quads and receive column-major data), so the stride-4 gather of the
// fp32 kernel's phase-split image becomes two transposing reads per operand on a plain bf16 copy of the row.
// The A operand A[co][t] is read as 16-byte rows of the gradient image [t / 8][co][8 t] (a lane group's
// 8-output blocks sit 256 bytes apart: conflict-free).
// Work unit of a WAVE = 64 consecutive outputs of one (batch row, group), as in the fp32 kernel: private LDS
// image, next unit's loads in flight during the MFMAs; at the end the four waves' 12 accumulator tiles are summed
// through LDS into one slab per workgroup, slabs combined by the deterministic split-K reduce (k_reduce_slabs).
constexpr int WU = 64;                        // outputs per unit
constexpr int WXQ = 75;                       // input quads per (unit, channel): 4 * 63 + 48 = 300 samples
constexpr int WXROW = 608;                    // bytes per LDS input row (300 bf16 + pad, multiple of 8)
constexpr int WX_PIECE = GCG * WXROW;
constexpr int WG_PIECE = 8 * 16 * 16;         // gradient image of a piece: [t / 8][co][8 t] (16-bit elements)
constexpr int W_WAVE = 3 * (WG_PIECE + WX_PIECE);     // (sized for three pieces: the end-of-kernel reduction scratch needs it)
constexpr int WNP = 2;                        // pieces per operand element: block-scaled fp16 x 2 (r04)
constexpr int NKT = 3;                        // 16-tap column tiles

// slab: the workgroup's slab index (blockIdx.x in a single launch; the workgroup's index over ALL parts in a parts launch,
// whose slabs the one deterministic reduce then sums: the weight gradient of the shared layer over every scale)
template <bool VEC>
__device__ __forceinline__ void gconv_wgrad_body(const ConvP& p, const float* __restrict__ x,
                                                 const float* __restrict__ gy,
                                                 const float* __restrict__ y_act,
                                                 float* __restrict__ partial, size_t partial_stride, int bx, int nbx, int slab) {
    unsigned char* const lds = g3_smem;               // 4 * W_WAVE
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = blockIdx.y;
    const int kind = y_act ? p.act : MS_ACT_NONE;
    unsigned char* gimg = lds + wid * W_WAVE;          // gradient pieces, then input pieces
    unsigned char* ximg = gimg + WNP * WG_PIECE;

    // ---- staging items
    // gradient: 16 co x 16 quads of 4 outputs; item i of a lane: co = (lane >> 4) + 4 i, quad tq = lane & 15
    const int tq = lane & 15, co0 = lane >> 4;
    // (the co slot of 8-output block tb is XORed with tb: the 16 quads a lane group stores for one co would
    // otherwise sit 256 bytes apart -- one bank)
    const int g_tb = tq >> 1;
    // input: 4 ci x 75 quads of 4 samples, 5 items per lane
    int x_lds[5];
    unsigned x_c[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int idx = lane + 64 * i;
        const int ci = idx / WXQ, Q = idx - ci * WXQ;
        const bool act = ci < GCG;
        x_lds[i] = act ? ci * WXROW + Q * 8 : 3 * WXROW + WXQ * 8;        // spare quad behind the last row's data
        x_c[i] = act ? (unsigned)(ci * p.Lin + 4 * Q) : 0x3C000000u;      // (elements; x4 = out of range)
    }
    constexpr unsigned OOB = 0xF0000000u;
    // (true sizes: see the forward kernel)
    const unsigned g_bytes = 4u * (unsigned)(p.B * p.Cout * p.Lout);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 4u * (unsigned)(p.B * p.Cin * p.Lin), 0x00020000);
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, g_bytes, 0x00020000);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y_act ? y_act : gy), 0, g_bytes, 0x00020000);

    const int tiles = (p.Lout + WU - 1) / WU;
    const int nunits = p.B * tiles;
    const int ustride = nbx * 4;
    const int db = ustride / tiles, dt = ustride - db * tiles;     // unit += ustride, in (b, tile) form

    // interior units (all 64 outputs and all 300 inputs inside their rows): lane-constant byte offsets + a
    // scalar unit base -- no vector address math
    unsigned g_c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) g_c[i] = co0 + 4 * i < p.Og ? (unsigned)((co0 + 4 * i) * p.Lout + 4 * tq) * 4u : OOB;
    f32x4 gv[4], ga[4], xv[5];
    auto gload = [&](int b, int ti) {
        const int t0 = ti * WU, u0 = t0 * GS - p.pad;
        const unsigned gbase = (unsigned)((b * p.Cout + g * p.Og) * p.Lout + t0);
        const unsigned xbase = (unsigned)((b * p.Cin + g * GCG) * p.Lin);
        if (t0 + WU <= p.Lout && u0 >= 0 && u0 + 4 * WXQ <= p.Lin) {      // interior unit, rows of any length (unaligned 16-byte loads)
            const unsigned gs = (unsigned)__builtin_amdgcn_readfirstlane((int)(gbase * 4u));
            const unsigned xs = (unsigned)__builtin_amdgcn_readfirstlane((int)((xbase + (unsigned)u0) * 4u));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, g_c[i], gs, 0));
                ga[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, g_c[i], gs, 0));
            }
#pragma unroll
            for (int i = 0; i < 5; ++i)
                xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rsX, x_c[i] < 0x3C000000u ? x_c[i] * 4u : OOB, xs, 0));
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + 4 * i;
            const unsigned off = gbase + (unsigned)(co * p.Lout + 4 * tq);
            if (VEC) {
                const bool ok = co < p.Og && t0 + 4 * tq < p.Lout;
                const unsigned vo = ok ? off * 4u : OOB;
                gv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, vo, 0, 0));
                ga[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, 0, 0));
            } else {         // any row length: unaligned 16-byte loads, the samples behind the row end cleared
                const bool ok = co < p.Og && t0 + 4 * tq < p.Lout;
                const unsigned vo = ok ? off * 4u : OOB;
                gv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, vo, 0, 0));
                ga[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, 0, 0));
#pragma unroll
                for (int e = 1; e < 4; ++e)
                    if (t0 + 4 * tq + e >= p.Lout) { gv[i][e] = 0.f; ga[i][e] = 0.f; }
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int Q4 = 4 * ((lane + 64 * i) % WXQ);
            const int sidx = u0 + Q4;
            if (VEC) {
                const bool ok = x_c[i] < 0x3C000000u && sidx >= 0 && sidx < p.Lin;
                xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rsX, ok ? (xbase + x_c[i] + (unsigned)u0) * 4u : OOB, 0, 0));
            } else {         // (u0 is a multiple of 4: a quad lies wholly in front of the row or starts inside it)
                const bool ok = x_c[i] < 0x3C000000u && sidx >= 0 && sidx < p.Lin;
                xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rsX, ok ? (xbase + x_c[i] + (unsigned)u0) * 4u : OOB, 0, 0));
#pragma unroll
                for (int e = 1; e < 4; ++e) xv[i][e] = sidx + e < p.Lin ? xv[i][e] : 0.f;
            }
        }
    };

    // Accumulators live across all units of a wave.  Both operands are data here (no weights): the gradient tile and the input
    // tile each carry a wave-local power-of-two scale that is STICKY from unit to unit -- it moves only when the unit's largest
    // magnitude times the current scale leaves [2^8, 2^15) -- and when one moves, the accumulators are multiplied by the
    // ratio of the new to the old product scale (a power of two: exact), so they always hold sums under the CURRENT scales.
    f32x4 acc[GCG][NKT];
#pragma unroll
    for (int c = 0; c < GCG; ++c)
#pragma unroll
        for (int j = 0; j < NKT; ++j) acc[c][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float Sg = 0.f, Sx = 0.f;                          // current scales (0: none yet)
    float Sg_lo = 0.f, Sx_lo = 0.f;                    // smallest scales so far (the largest magnitudes seen): a scale never rises more
                                                       // than 2^30 above them, so the sums are re-expressed by at most 2^60 in total and
                                                       // stay finite; operands that much smaller than earlier ones are below the sums' resolution
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};

    // fragment addresses of this lane
    const int m = lane & 15, kg = lane >> 4;
    // A rows: block tb = 4 step + kg, slot m ^ tb = (m ^ kg) ^ 4 step
    const unsigned char* a_rd0 = gimg + (kg * 16 + (m ^ kg)) * 16;
    const unsigned char* a_rd1 = gimg + 1024 + (kg * 16 + (m ^ kg ^ 4)) * 16;
    // transposing read: lane 4q + p of a 16-lane group supplies row q (output t), column quad p (4 taps)
    const unsigned b_rd = (unsigned)(uintptr_t)ximg + 64 * kg + 8 * ((lane & 15) >> 2) + 8 * (lane & 3);

    int unit = bx * 4 + wid;
    int b = unit / tiles, ti = unit - b * tiles;
    if (unit < nunits) gload(b, ti);
    for (; unit < nunits; unit += ustride) {
        // ---- the unit's largest magnitudes -> sticky scales (the raw gradient bounds the masked one)
        {
            float mg = 0.f, mx = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) mg = fmaxf(mg, fabsf(gv[i][k]));
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) mx = fmaxf(mx, fabsf(xv[i][k]));
            mg = wave_max_dpp(mg); mx = wave_max_dpp(mx);
            float nSg = Sg, nSx = Sx, inv;
            if (mg > 0.f && !(mg * Sg >= 256.f && mg * Sg < 32768.f)) block_scale(mg * 4.f, nSg, inv);   // largest magnitude at 2^12
            if (mx > 0.f && !(mx * Sx >= 256.f && mx * Sx < 32768.f)) block_scale(mx * 4.f, nSx, inv);
            if (nSg == 0.f) nSg = 1.f;
            if (nSx == 0.f) nSx = 1.f;
            if (Sg_lo != 0.f) nSg = fminf(nSg, Sg_lo * 0x1p30f);
            if (Sx_lo != 0.f) nSx = fminf(nSx, Sx_lo * 0x1p30f);
            if (mg > 0.f) Sg_lo = Sg_lo == 0.f ? nSg : fminf(Sg_lo, nSg);      // (an all-zero unit sets no reference)
            if (mx > 0.f) Sx_lo = Sx_lo == 0.f ? nSx : fminf(Sx_lo, nSx);
            if ((nSg != Sg || nSx != Sx) && Sg != 0.f) {               // (wave-uniform) a scale moved: re-express the sums
                const float ratio = (nSg / Sg) * (nSx / Sx);
#pragma unroll
                for (int c = 0; c < GCG; ++c)
#pragma unroll
                    for (int j = 0; j < NKT; ++j) acc[c][j] *= ratio;
            }
            Sg = nSg; Sx = nSx;
        }
        // ---- registers -> LDS: activation derivative, scale, split, 8-byte quads
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = ms_act_grad(gv[i][k], ga[i][k], kind, p.slope);
            bsum[i] += (e[0] + e[1]) + (e[2] + e[3]);
            unsigned h0, l0, h1, l1;
            split_pair2(e[0] * Sg, e[1] * Sg, h0, l0);
            split_pair2(e[2] * Sg, e[3] * Sg, h1, l1);
            unsigned char* d = gimg + (g_tb * 16 + ((co0 + 4 * i) ^ g_tb)) * 16 + (tq & 1) * 8;
            *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(d + WG_PIECE) = make_uint2(l0, l1);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            unsigned h0, l0, h1, l1;
            split_pair2(xv[i][0] * Sx, xv[i][1] * Sx, h0, l0);
            split_pair2(xv[i][2] * Sx, xv[i][3] * Sx, h1, l1);
            unsigned char* d = ximg + x_lds[i];
            *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(d + WX_PIECE) = make_uint2(l0, l1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int tvalid = min(WU, p.Lout - ti * WU);
        int nb = b + db, nti = ti + dt;
        if (nti >= tiles) { nti -= tiles; ++nb; }
        if (unit + ustride < nunits) gload(nb, nti);

        // (step, channel) iterations, each in three read groups -- the lo / hi transposing reads of the three
        // column tiles for one piece, in the order the partial products need them: piece 3, piece 1, piece 2.
        // Reads are hand-issued two groups (12 reads; lgkmcnt counts to 15) ahead of the MFMAs they feed, with
        // counted waits.  Left to the compiler, the reads of a tile were shared with its neighbour (rows t+4.. of
        // tile j = rows t.. of tile j+1) at the price of ~70 v_mov per unit, and every MFMA group waited for its
        // own reads.
        const int niter = tvalid > 32 ? 2 * GCG : GCG;              // wave-uniform: outputs 32.. are all zero
        f16x8 A[2][WNP];
#pragma unroll
        for (int step = 0; step < 2; ++step)
#pragma unroll
            for (int pc = 0; pc < WNP; ++pc)
                A[step][pc] = *reinterpret_cast<const f16x8*>((step ? a_rd1 : a_rd0) + pc * WG_PIECE);
        u32x2 Bl[2][WNP][NKT], Bh[2][WNP][NKT];          // [iteration parity][piece][tile]
#define MS_W3_READ(it_, pc)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < NKT; ++j) {                                                         \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"                                                    \
                     : "=v"(Bl[(it_) & 1][pc][j])                                                             \
                     : "v"(b_rd), "n"((pc) * WX_PIECE + ((it_) & 3) * WXROW + ((it_) >> 2) * 256 + j * 32));  \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"                                                    \
                     : "=v"(Bh[(it_) & 1][pc][j])                                                             \
                     : "v"(b_rd), "n"((pc) * WX_PIECE + ((it_) & 3) * WXROW + ((it_) >> 2) * 256 + j * 32 + 32)); \
    }
#define MS_W3_WAIT(cnt, it_, pc)                                                                              \
    asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                \
                 : "+v"(Bl[(it_) & 1][pc][0]), "+v"(Bl[(it_) & 1][pc][1]), "+v"(Bl[(it_) & 1][pc][2]),        \
                   "+v"(Bh[(it_) & 1][pc][0]), "+v"(Bh[(it_) & 1][pc][1]), "+v"(Bh[(it_) & 1][pc][2]))
#define MS_W3_MMA(dst, it_, pa, pb)                                                                           \
    _Pragma("unroll") for (int j = 0; j < NKT; ++j) {                                                         \
        const u32x4 bv = {Bl[(it_) & 1][pb][j][0], Bl[(it_) & 1][pb][j][1], Bh[(it_) & 1][pb][j][0],          \
                          Bh[(it_) & 1][pb][j][1]};                                                           \
        dst[(it_) & 3][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                           \
            A[(it_) >> 2][pa], __builtin_bit_cast(f16x8, bv), dst[(it_) & 3][j], 0, 0, 0);                    \
    }
        // per (step, channel) iteration: (a_h b_l) and (a_l b_h) into the cross sums, (a_h b_h) into the main sums; the
        // transposing reads of a piece (lo / hi of the three column tiles: six reads) are hand-issued one group ahead of
        // the MFMAs they feed, with counted waits
#define MS_W3_ITER(it_)                                                                                       \
    if ((it_) < niter) {                                                                                      \
        MS_W3_WAIT(6, it_, 1);                                                                                \
        MS_W3_MMA(acc, it_, 0, 1);                                                                            \
        if ((it_) + 1 < niter) {                                                                              \
            MS_W3_READ((it_) + 1, 1);                                                                         \
            MS_W3_WAIT(6, it_, 0);                                                                            \
        } else {                                                                                              \
            MS_W3_WAIT(0, it_, 0);                                                                            \
        }                                                                                                     \
        MS_W3_MMA(acc, it_, 1, 0);                                                                            \
        MS_W3_MMA(acc, it_, 0, 0);                                                                            \
        if ((it_) + 1 < niter) { MS_W3_READ((it_) + 1, 0); }                                                  \
    }
        MS_W3_READ(0, 1);
        MS_W3_READ(0, 0);
        MS_W3_ITER(0) MS_W3_ITER(1) MS_W3_ITER(2) MS_W3_ITER(3)
        MS_W3_ITER(4) MS_W3_ITER(5) MS_W3_ITER(6) MS_W3_ITER(7)
#undef MS_W3_ITER
#undef MS_W3_MMA
#undef MS_W3_READ
#undef MS_W3_WAIT
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        b = nb; ti = nti;
    }

    // ---- one slab per WORKGROUP: the four waves' accumulators are summed through LDS (the staging images are
    // free now), wave c writes input channel c.  D[co][k]: lane (k = lane & 15, co quad = lane >> 4), rows r.
    {
        const float kfin = (Sg != 0.f) ? 1.f / (Sg * Sx) : 0.f;       // (powers of two: exact) undo the current scales
#pragma unroll
        for (int c = 0; c < GCG; ++c)
#pragma unroll
            for (int j = 0; j < NKT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[c][j][r] *= kfin;
    }
    __syncthreads();
    static_assert((4 * GCG * NKT * 4 * 64 + 64) * 4 <= 4 * W_WAVE, "reduction scratch fits the staging images");
    float* red = reinterpret_cast<float*>(lds);                   // [wave][c][tile][r][lane]
#pragma unroll
    for (int c = 0; c < GCG; ++c)
#pragma unroll
        for (int j = 0; j < NKT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((wid * GCG + c) * NKT + j) * 4 + r) * 64 + lane] = acc[c][j][r];
    float* redb = red + 4 * GCG * NKT * 4 * 64;                   // [wave][16 co]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = bsum[i];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if ((lane & 15) == 0) redb[wid * 16 + co0 + 4 * i] = v;
    }
    __syncthreads();
    float* part = partial + (size_t)slab * partial_stride;
    const int J = GCG * GK;
    {
        const int c = wid;
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            const int kk = j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = 0.f;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) v += red[(((wv * GCG + c) * NKT + j) * 4 + r) * 64 + lane];
                const int co = (lane >> 4) * 4 + r;
                if (kk < GK && co < p.Og) part[(size_t)(g * p.Og + co) * J + c * GK + kk] = v;
            }
        }
    }
    if (wid == 0 && lane < 16 && lane < p.Og)
        part[(size_t)p.Cout * J + g * p.Og + lane] = (redb[lane] + redb[16 + lane]) + (redb[32 + lane] + redb[48 + lane]);
}

template <bool VEC>
__global__ __launch_bounds__(256, 2) void k_gconv_split_wgrad(ConvP p, const float* __restrict__ x,
                                                             const float* __restrict__ gy,
                                                             const float* __restrict__ y_act,
                                                             float* __restrict__ partial, size_t partial_stride) {
    gconv_wgrad_body<VEC>(p, x, gy, y_act, partial, partial_stride, blockIdx.x, gridDim.x, blockIdx.x);
}

// parts: a = x, b = gy, c = y_act
__global__ __launch_bounds__(256, 2) void k_gconv_split_wgrad_parts(G3Parts q, float* __restrict__ partial, size_t partial_stride) {
    const int bx = blockIdx.x, i = g3_part_of(q, bx);
    const ConvP p = g3_part_conv(q, i);
    const int rbx = bx - q.bx0[i], nbx = q.bx0[i + 1] - q.bx0[i];
    if (i == 0) gconv_wgrad_body<true>(p, q.a[0], q.b[0], q.c[0], partial, partial_stride, rbx, nbx, bx);
    else gconv_wgrad_body<false>(p, q.a[i], q.b[i], q.c[i], partial, partial_stride, rbx, nbx, bx);
}

// ---------------------------------------------------------------- backward data (16 outputs per group)
// gx[b, g*4+ci, 4q + r] = gx_add + sum_{co, jj} w[co, ci, r + 4 jj] * gp[b, co, q + 5 - jj]     (pad = 20)
// GEMM per group and q-tile: M = (ci, r) = 16 rows, N = 16 consecutive q, contraction = 16 co x 12 taps jj (the
// 12th carries zero weights) in 6 MFMA steps of (2 taps x 16 co).  A lane's B fragment is 8 output channels at ONE
// gradient position, so the gradient tile is kept position-major in LDS: [co octet][position][8 co] bf16, 16 bytes
// per (octet, position).  Lane group kg = (octet = kg >> 1, tap parity = kg & 1): the two lane groups that share a
// ds_read_b128 bank group read the SAME octet at positions one apart -- consecutive 16-byte slots, conflict-free.
// Staging transposes in registers: a lane loads 4 co x 4 positions (4 + 4 float4 of gy / y_act), applies the
// activation derivative, and per position writes the 4 co as one 8-byte quad per piece (octet images 32 bytes
// off a 128-byte multiple apart: 2-way at worst).  A lane's 4 accumulator rows are the 4 phases r of one (ci, q):
// 4 consecutive samples of gx, one 16-byte store.  Work unit of a WAVE: 48 consecutive q (192 samples x 4 input
// channels) of one (batch row, group): a 64-position window = exactly one staging item per lane.
constexpr int BQ = 48;                        // q's per unit (3 MFMA column tiles)
constexpr int BNP = 64;                       // gradient positions staged per unit: q0 - 8 .. q0 + 55
constexpr int BOCT = BNP * 16 + 32;           // bytes per octet image (== 32 mod 128)
constexpr int B_PIECE = 2 * BOCT;
constexpr int B_WAVE = 2 * B_PIECE;         // two fp16 pieces (r04)
constexpr int BJ = 6;                         // MFMA steps: taps 2J, 2J+1

template <bool VEC, bool VOUT>
__device__ __forceinline__ void gconv_bwd_data_body(const ConvP& p, int tiles, int nunits,
                                                    const float* __restrict__ gy,
                                                    const float* __restrict__ y_act,
                                                    const float* __restrict__ w,
                                                    const float* __restrict__ gx_add,
                                                    float* __restrict__ gx, int bx, int nbx) {
    unsigned char* const lds = g3_smem;               // 4 * B_WAVE
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = blockIdx.y;
    const int kind = y_act ? p.act : MS_ACT_NONE;
    unsigned char* img = lds + wid * B_WAVE;
    const int n = lane & 15, kg = lane >> 4;
    const int oct = kg >> 1, tp = kg & 1;
    const int wstride = nbx * 4;
    const int db = wstride / tiles, dt = wstride - db * tiles;

    // staging item of this lane: co quad cq, position quad pq
    const int cq = lane & 3, pq = lane >> 2;
    unsigned char* wr = img + (cq >> 1) * BOCT + (4 * pq) * 16 + (cq & 1) * 8;       // + j * 16 (position), + piece
    constexpr unsigned OOB = 0xF0000000u;
    const unsigned g_bytes = 4u * (unsigned)(p.B * p.Cout * p.Lout);            // (true sizes: see the forward kernel)
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, g_bytes, 0x00020000);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y_act ? y_act : gy), 0, g_bytes, 0x00020000);
    unsigned g_c[4];                              // interior units: byte offset of row 4 cq + k, position quad pq
#pragma unroll
    for (int k = 0; k < 4; ++k) g_c[k] = (unsigned)((4 * cq + k) * p.Lout + 4 * pq) * 4u;
    f32x4 gv[4], ga[4];
    auto gload = [&](int b, int ti) {
        const int ws = ti * BQ - 8;                                   // first staged gradient position
        const unsigned base = (unsigned)((b * p.Cout + g * 16) * p.Lout);
        if (ws >= 0 && ws + BNP <= p.Lout) {                   // interior unit, rows of any length (unaligned 16-byte loads)
            const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)((base + (unsigned)ws) * 4u));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                gv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, g_c[k], sb, 0));
                ga[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, g_c[k], sb, 0));
            }
            return;
        }
        const int t = ws + 4 * pq;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned off = base + (unsigned)((4 * cq + k) * p.Lout);
            if (VEC) {
                const unsigned vo = (t >= 0 && t < p.Lout) ? (off + (unsigned)t) * 4u : OOB;
                gv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, vo, 0, 0));
                ga[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, 0, 0));
            } else {         // any row length (t is a multiple of 4: only the quad across the row end needs clearing)
                const unsigned vo = (t >= 0 && t < p.Lout) ? (off + (unsigned)t) * 4u : OOB;
                gv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsG, vo, 0, 0));
                ga[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, 0, 0));
#pragma unroll
                for (int e = 1; e < 4; ++e)
                    if (t + e >= p.Lout) { gv[k][e] = 0.f; ga[k][e] = 0.f; }
            }
        }
    };
    int unit = bx * 4 + wid;
    int b = unit / tiles, ti = unit - b * tiles;
    if (unit < nunits) gload(b, ti);

    // ---- weight fragments: lane (m = (ci, r), kg = (octet, tap parity)) holds S_w w[g*16 + 8 oct + e][ci][r + 4 (2J + tp)]
    // in two fp16 pieces
    f16x8 A[BJ][2];
    float iWS = 1.f;
    {
        const int ci = n >> 2, r = n & 3;
        float wall[BJ][8];
        float wm = 0.f;
#pragma unroll
        for (int J = 0; J < BJ; ++J) {
            const int tap = r + 4 * (2 * J + tp);
            const bool ok = tap < GK;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = w[((size_t)(g * 16 + 8 * oct + e) * GCG + ci) * GK + (ok ? tap : 0)];
                wall[J][e] = ok ? v : 0.f;
                wm = fmaxf(wm, fabsf(wall[J][e]));
            }
        }
        float WS;
        weight_scale(wave_max_dpp(wm), WS, iWS);              // the group's weights under the wave's own scale
#pragma unroll
        for (int J = 0; J < BJ; ++J) {
            u32x4 h, l;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned hh, ll;
                split_pair2(wall[J][2 * q] * WS, wall[J][2 * q + 1] * WS, hh, ll);
                h[q] = hh; l[q] = ll;
            }
            A[J][0] = __builtin_bit_cast(f16x8, h);
            A[J][1] = __builtin_bit_cast(f16x8, l);
        }
    }
    // B fragment of (tile tq, step J): position (q0 + 16 tq + n) + 5 - 2J - tp, window index = that - (q0 - 8)
    const unsigned rd = (unsigned)(uintptr_t)img + oct * BOCT + (n + 3 - tp) * 16;       // + tq * 256 + (5 - J) * 32 + piece
    const int Lq = (p.Lin + GS - 1) / GS;

    for (; unit < nunits; unit += wstride) {
        // ---- the unit's block scale (the raw gradient bounds the masked one; a wave owns the whole unit: no barrier)
        float um = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) um = fmaxf(um, fabsf(gv[k][j]));
        float S, invS;
        block_scale(wave_max_dpp(um), S, invS);
        // ---- registers -> LDS: activation derivative, 4 x 4 transpose, scale, split, one 8-byte co quad per position
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] = ms_act_grad(gv[k][j], ga[k][j], kind, p.slope) * S;
            unsigned h0, l0, h1, l1;
            split_pair2(e[0], e[1], h0, l0);
            split_pair2(e[2], e[3], h1, l1);
            *reinterpret_cast<uint2*>(wr + j * 16) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(wr + j * 16 + B_PIECE) = make_uint2(l0, l1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int q0 = ti * BQ;
        const size_t rowoff = ((size_t)b * p.Cin + (size_t)g * GCG + kg) * p.Lin;       // D rows 4 kg + r: ci = kg
        f32x4 addv[3];
        if (VOUT && gx_add) {
#pragma unroll
            for (int tq = 0; tq < 3; ++tq) {
                const int sidx = (q0 + tq * 16 + n) * GS;
                addv[tq] = *reinterpret_cast<const f32x4*>(gx_add + (sidx < p.Lin ? rowoff + sidx : 0));
            }
        }
        int nb = b + db, nti = ti + dt;
        if (nti >= tiles) { nti -= tiles; ++nb; }
        if (unit + wstride < nunits) gload(nb, nti);

        f32x4 acc[3];
#pragma unroll
        for (int tq = 0; tq < 3; ++tq) acc[tq] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // read groups (one piece, three tiles) one group ahead of the MFMAs they feed, counted waits; per step the low
        // gradient piece first (a_h b_l), then the high one (a_l b_h, a_h b_h)
        u32x4 Bf[2][2][3];                            // [step parity][piece][tile]
#define MS_B3_READ(J_, pc)                                                                                    \
    _Pragma("unroll") for (int tq = 0; tq < 3; ++tq)                                                          \
        asm volatile("ds_read_b128 %0, %1 offset:%2"                                                          \
                     : "=v"(Bf[(J_) & 1][pc][tq])                                                             \
                     : "v"(rd), "n"((pc) * B_PIECE + tq * 256 + (BJ - 1 - (J_)) * 32));
#define MS_B3_WAIT(cnt, J_, pc)                                                                               \
    asm volatile("s_waitcnt lgkmcnt(" #cnt ")"                                                                \
                 : "+v"(Bf[(J_) & 1][pc][0]), "+v"(Bf[(J_) & 1][pc][1]), "+v"(Bf[(J_) & 1][pc][2]))
#define MS_B3_MMA(dst, J_, pa, pb)                                                                            \
    _Pragma("unroll") for (int tq = 0; tq < 3; ++tq)                                                          \
        dst[tq] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                                                     \
            A[J_][pa], __builtin_bit_cast(f16x8, Bf[(J_) & 1][pb][tq]), dst[tq], 0, 0, 0);
#define MS_B3_STEP(J_, last)                                                                                  \
    MS_B3_WAIT(3, J_, 1);                                                                                     \
    MS_B3_MMA(acc, J_, 0, 1);                                                                                 \
    if (!(last)) { MS_B3_READ((J_) + 1, 1); MS_B3_WAIT(3, J_, 0); } else { MS_B3_WAIT(0, J_, 0); }            \
    MS_B3_MMA(acc, J_, 1, 0);                                                                                 \
    MS_B3_MMA(acc, J_, 0, 0);                                                                                 \
    if (!(last)) { MS_B3_READ((J_) + 1, 0); }
        MS_B3_READ(0, 1);
        MS_B3_READ(0, 0);
        MS_B3_STEP(0, false) MS_B3_STEP(1, false) MS_B3_STEP(2, false)
        MS_B3_STEP(3, false) MS_B3_STEP(4, false) MS_B3_STEP(5, true)
#undef MS_B3_STEP
#undef MS_B3_MMA
#undef MS_B3_WAIT
#undef MS_B3_READ
        const float kscale = invS * iWS;
#pragma unroll
        for (int tq = 0; tq < 3; ++tq)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) acc[tq][rr] *= kscale;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int tq = 0; tq < 3; ++tq) {
            const int sidx = (q0 + tq * 16 + n) * GS;
            if (q0 + tq * 16 + n >= Lq || sidx >= p.Lin) continue;
            if (VOUT) {                                    // Lin % 4 == 0: the 4 samples exist together
                f32x4 v = acc[tq];
                if (gx_add) v += addv[tq];
                *reinterpret_cast<f32x4*>(gx + rowoff + sidx) = v;
            } else if (sidx + 3 < p.Lin) {                 // any row length: unaligned 16-byte accesses inside the row
                f32x4 v = acc[tq];
                if (gx_add) {
                    const f32x4u a4 = *reinterpret_cast<const f32x4u*>(gx_add + rowoff + sidx);
                    v += (f32x4){a4[0], a4[1], a4[2], a4[3]};
                }
                *reinterpret_cast<f32x4u*>(gx + rowoff + sidx) = (f32x4u){v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    if (sidx + rr < p.Lin) {
                        float v = acc[tq][rr];
                        if (gx_add) v += gx_add[rowoff + sidx + rr];
                        gx[rowoff + sidx + rr] = v;
                    }
                }
            }
        }
        b = nb; ti = nti;
    }
}

template <bool VEC, bool VOUT>
__global__ __launch_bounds__(256, 2) void k_gconv_split_bwd_data(ConvP p, int tiles, int nunits,
                                                                const float* __restrict__ gy,
                                                                const float* __restrict__ y_act,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ gx_add,
                                                                float* __restrict__ gx) {
    gconv_bwd_data_body<VEC, VOUT>(p, tiles, nunits, gy, y_act, w, gx_add, gx, blockIdx.x, gridDim.x);
}

// parts: a = gy, b = y_act, c = gx_add, o = gx; u0 = tiles, u1 = nunits
__global__ __launch_bounds__(256, 2) void k_gconv_split_bwd_data_parts(G3Parts q, const float* __restrict__ w) {
    const int bx = blockIdx.x, i = g3_part_of(q, bx);
    const ConvP p = g3_part_conv(q, i);
    const int rbx = bx - q.bx0[i], nbx = q.bx0[i + 1] - q.bx0[i];
    if (i == 0) gconv_bwd_data_body<true, true>(p, q.u0[0], q.u1[0], q.a[0], q.b[0], w, q.c[0], q.o[0], rbx, nbx);
    else gconv_bwd_data_body<false, false>(p, q.u0[i], q.u1[i], q.a[i], q.b[i], w, q.c[i], q.o[i], rbx, nbx);
}

int pick_rb(int Lout) {
    if (Lout <= 16) return 4;
    if (Lout <= 32) return 2;
    if (Lout <= 64) return 1;
    // long rows: padded work of 64- / 32- / 16-output segments; prefer the widest within 6 %
    const int w64 = ms_ceil_div(Lout, 64) * 64, w32 = ms_ceil_div(Lout, 32) * 32, w16 = ms_ceil_div(Lout, 16) * 16;
    if (w64 * 100 <= w16 * 106) return 1;
    if (w32 * 100 <= w16 * 106) return 2;
    return 4;
}

}  // namespace

bool msg3_fwd_applicable(const ConvP& p) {
    const char* e = getenv("MSYNTH_GCONV3");          // read per call: the micro-benchmarks flip it
    const bool on = !e || atoi(e) != 0;
    return on && p.K == GK && p.stride == GS && p.Cg == GCG && p.dil == 1 && p.pad_mode == MS_PAD_ZERO &&
           p.Og <= 16 && p.groups <= 65535 &&
           (p.act == MS_ACT_NONE || (p.act == MS_ACT_LRELU && p.slope >= 0.f && p.slope <= 1.f)) &&
           (long long)p.B * p.Cin * p.Lin * 4 < (1ll << 31) &&
           (long long)p.B * p.Cout * p.Lout < (1ll << 31);
}

const char* msg3_fwd_name(const ConvP&) { return "k_gconv_split_fwd"; }

template <int RB>
constexpr size_t g3_fwd_lds() { return 4 * (size_t)GF<RB>::WAVE_BYTES; }
static size_t g3_fwd_lds_rt(int rb) { return rb == 1 ? g3_fwd_lds<1>() : (rb == 2 ? g3_fwd_lds<2>() : g3_fwd_lds<4>()); }

int msg3_conv1d_fwd(const ConvP& p, const float* x, const float* w, const float* bias, float* y,
                    hipStream_t s) {
    const int rb = pick_rb(p.Lout);
    const int ts = 64 / rb;
    const int spr = ms_ceil_div(p.Lout, ts), nseg = p.B * spr, nunits = ms_ceil_div(nseg, rb);
    const int target_waves = 2048;
    const long long total_units = (long long)nunits * p.groups;
    const int upw = (int)((total_units + target_waves - 1) / target_waves);
    const int gxn = ms_ceil_div(nunits, 4 * (upw > 0 ? upw : 1));
    const dim3 grid(gxn, p.groups);
    const bool vec = p.Lin % 4 == 0 && p.pad % 4 == 0 && (((uintptr_t)x) & 15) == 0;
    const bool vout = p.Lout % 4 == 0 && (((uintptr_t)y) & 15) == 0;
#define MS_G3(RBV, V, VO)                                                                                                    \
    do {                                                                                                                     \
        ms_note_kernel(3, "k_gconv_split_fwd<%d, %s, %s>", RBV, V ? "true" : "false", VO ? "true" : "false");               \
        hipLaunchKernelGGL((k_gconv_split_fwd<RBV, V, VO>), grid, dim3(256), g3_fwd_lds<RBV>(), s, p, spr, nseg, nunits, x, w, \
                           bias, y);                                                                                         \
    } while (0)
#define MS_G3V(RBV)                                   \
    do {                                              \
        if (vec && vout) MS_G3(RBV, true, true);      \
        else if (vec) MS_G3(RBV, true, false);        \
        else if (vout) MS_G3(RBV, false, true);       \
        else MS_G3(RBV, false, false);                \
    } while (0)
    if (rb == 1) MS_G3V(1);
    else if (rb == 2) MS_G3V(2);
    else MS_G3V(4);
#undef MS_G3V
#undef MS_G3
    MS_CHECK_LAUNCH();
    return MS_OK;
}

// ---- parts launches: one layer over the discriminator's scales (G3Parts)
namespace {

bool g3_parts_table(const ConvP& c, const ms_conv1d_parts* parts, G3Parts* q) {
    if (!parts || parts->count < 2 || parts->count > MS_CONV_PARTS_MAX) return false;
    q->p = c;
    q->count = parts->count;
    for (int i = 0; i < parts->count; ++i) {
        if (parts->B[i] <= 0 || parts->Lin[i] <= 0) return false;
        const int eff = parts->Lin[i] + 2 * c.pad - c.dil * (c.K - 1) - 1;
        if (eff < 0) return false;
        q->B[i] = parts->B[i]; q->Lin[i] = parts->Lin[i]; q->Lout[i] = eff / c.stride + 1;
    }
    for (int i = parts->count; i < MS_CONV_PARTS_MAX; ++i) {
        q->B[i] = q->Lin[i] = q->Lout[i] = q->u0[i] = q->u1[i] = q->u2[i] = 0;
        q->a[i] = q->b[i] = q->c[i] = nullptr; q->o[i] = nullptr;
    }
    return true;
}

ConvP g3_host_part(const ConvP& c, const G3Parts& q, int i) {
    ConvP p = c;
    p.B = q.B[i]; p.Lin = q.Lin[i]; p.Lout = q.Lout[i];
    return p;
}

bool al16(const void* a) { return (((uintptr_t)a) & 15) == 0; }

}  // namespace

// Forward: every part must be a geometry the single launch takes; part 0 16-byte shaped (the instantiations of a parts kernel
// are <RB0, vec, vec>, <RB1, dword, dword>, <RB2, dword, dword>); the RB triples are those of the three scales of an 8192-sample
// window (outputs 2048 / 1025 / 513 ... 32 / 17 / 9) -- anything else runs part by part.
bool msg3_parts_fwd_applicable(const ConvP& c, const ms_conv1d_parts* parts) {
    G3Parts q;
    if (!g3_parts_table(c, parts, &q) || q.count != 3) return false;
    for (int i = 0; i < q.count; ++i)
        if (!msg3_fwd_applicable(g3_host_part(c, q, i)) || !parts->x[i] || !parts->y[i]) return false;
    if (q.Lin[0] % 4 || c.pad % 4 || q.Lout[0] % 4 || !al16(parts->x[0]) || !al16(parts->y[0])) return false;
    const int r0 = pick_rb(q.Lout[0]), r1 = pick_rb(q.Lout[1]), r2 = pick_rb(q.Lout[2]);
    return (r0 == 1 && r1 == 1 && r2 == 2) || (r0 == 1 && r1 == 2 && r2 == 4) || (r0 == 1 && r1 == 4 && r2 == 1) ||
           (r0 == 2 && r1 == 2 && r2 == 4);
}

int msg3_parts_fwd(const ConvP& c, const ms_conv1d_parts* parts, const float* w, const float* bias, hipStream_t s) {
    G3Parts q;
    if (!g3_parts_table(c, parts, &q)) return MS_ERR_INVALID_ARG;
    int rb[MS_CONV_PARTS_MAX], nunits[MS_CONV_PARTS_MAX];
    long long total_units = 0;
    size_t lds = 0;
    for (int i = 0; i < q.count; ++i) {
        rb[i] = pick_rb(q.Lout[i]);
        const int ts = 64 / rb[i];
        const int spr = ms_ceil_div(q.Lout[i], ts), nseg = q.B[i] * spr;
        nunits[i] = ms_ceil_div(nseg, rb[i]);
        q.u0[i] = spr; q.u1[i] = nseg; q.u2[i] = nunits[i];
        q.a[i] = parts->x[i]; q.o[i] = parts->y[i];
        total_units += (long long)nunits[i] * c.groups;
        if (g3_fwd_lds_rt(rb[i]) > lds) lds = g3_fwd_lds_rt(rb[i]);
    }
    const int upw = (int)((total_units + 2047) / 2048);             // ~2 waves per SIMD, `upw` units per wave
    q.bx0[0] = 0;
    for (int i = 0; i < q.count; ++i) q.bx0[i + 1] = q.bx0[i] + ms_ceil_div(nunits[i], 4 * (upw > 0 ? upw : 1));
    for (int i = q.count; i < MS_CONV_PARTS_MAX; ++i) q.bx0[i + 1] = q.bx0[q.count];
    const dim3 grid(q.bx0[q.count], c.groups);
    ms_note_kernel(3, "k_gconv_split_fwd_parts<%d, %d, %d>", rb[0], rb[1], rb[2]);
#define MS_G3P(A, B_, C_) \
    hipLaunchKernelGGL((k_gconv_split_fwd_parts<A, B_, C_>), grid, dim3(256), lds, s, q, w, bias)
    if (rb[0] == 1 && rb[1] == 1 && rb[2] == 2) MS_G3P(1, 1, 2);              // 2048 / 1025 / 513 outputs
    else if (rb[0] == 1 && rb[1] == 2 && rb[2] == 4) MS_G3P(1, 2, 4);         // 512 / 257 / 129
    else if (rb[0] == 1 && rb[1] == 4 && rb[2] == 1) MS_G3P(1, 4, 1);         // 128 / 65 / 33
    else if (rb[0] == 2 && rb[1] == 2 && rb[2] == 4) MS_G3P(2, 2, 4);         // 32 / 17 / 9
    else return MS_ERR_UNSUPPORTED;
#undef MS_G3P
    MS_CHECK_LAUNCH();
    return MS_OK;
}

bool msg3_bwd_weight_applicable(const ConvP& p) {
    const char* e = getenv("MSYNTH_GCONV3");
    const bool on = !e || atoi(e) != 0;
    return on && p.K == GK && p.stride == GS && p.Cg == GCG && p.dil == 1 && p.pad_mode == MS_PAD_ZERO &&
           p.Og <= 16 && p.groups <= 65535 && (long long)p.B * p.Cin * p.Lin * 4 < (1ll << 30) &&
           (long long)p.B * p.Cout * p.Lout * 4 < (1ll << 31);
}

const char* msg3_bwd_weight_name(const ConvP&) { return "k_gconv_split_wgrad"; }

int msg3_conv1d_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act,
                           float* gw, float* gb, float beta, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < msg_bwd_weight_ws(p)) return MS_ERR_WORKSPACE;
    const int gxn = msg_wgrad_gridx(p);
    const size_t stride = (size_t)p.Cout * GCG * GK + p.Cout;
    float* partial = (float*)ws;
    const bool vec = p.Lin % 4 == 0 && p.pad % 4 == 0 && p.Lout % 4 == 0 && (((uintptr_t)x) & 15) == 0 &&
                     (((uintptr_t)gy) & 15) == 0 && (!y_act || (((uintptr_t)y_act) & 15) == 0);
    ms_note_kernel(3, "k_gconv_split_wgrad<%s>", vec ? "true" : "false");
    if (vec) hipLaunchKernelGGL((k_gconv_split_wgrad<true>), dim3(gxn, p.groups), dim3(256), 4 * W_WAVE, s, p, x, gy, y_act, partial, stride);
    else hipLaunchKernelGGL((k_gconv_split_wgrad<false>), dim3(gxn, p.groups), dim3(256), 4 * W_WAVE, s, p, x, gy, y_act, partial, stride);
    MS_CHECK_LAUNCH();
    return msg_reduce_slabs(p, partial, stride, gxn, gw, gb, beta, s);      // one slab per workgroup
}

// Weight gradient of the shared layer over all parts: workgroup columns are dealt to the parts in proportion to their
// wave units, every workgroup writes one slab, ONE reduce sums them all -- the sum over the scales that
// discriminator/melgan.py:13-27 implies for the shared parameters costs nothing extra.
namespace {
int g3_wgrad_parts_columns(const ConvP& c, const G3Parts& q, int* cols) {
    const size_t slab = ((size_t)c.Cout * GCG * GK + c.Cout) * sizeof(float);
    long long chunks[MS_CONV_PARTS_MAX], total = 0;
    for (int i = 0; i < q.count; ++i) {
        chunks[i] = ms_ceil_div(q.B[i] * ms_ceil_div(q.Lout[i], WU), 4);
        total += chunks[i];
    }
    int gx = ms_ceil_div(512, c.groups);                 // ~2 workgroups per CU over all parts
    const size_t cap = (size_t)32 << 20;
    while (gx > q.count && (size_t)gx * slab > cap) --gx;
    if (gx < q.count) gx = q.count;
    int sum = 0;
    for (int i = 0; i < q.count; ++i) {
        long long n = (chunks[i] * gx + total - 1) / total;
        if (n < 1) n = 1;
        if (n > chunks[i]) n = chunks[i];
        cols[i] = (int)n;
        sum += cols[i];
    }
    return sum;
}
}  // namespace

bool msg3_parts_bwd_weight_applicable(const ConvP& c, const ms_conv1d_parts* parts) {
    G3Parts q;
    if (!g3_parts_table(c, parts, &q)) return false;
    for (int i = 0; i < q.count; ++i)
        if (!msg3_bwd_weight_applicable(g3_host_part(c, q, i)) || !parts->x[i] || !parts->gy[i]) return false;
    return q.Lin[0] % 4 == 0 && c.pad % 4 == 0 && q.Lout[0] % 4 == 0 && al16(parts->x[0]) && al16(parts->gy[0]) &&
           (!parts->y_act[0] || al16(parts->y_act[0]));
}

size_t msg3_parts_bwd_weight_ws(const ConvP& c, const ms_conv1d_parts* parts) {
    G3Parts q;
    int cols[MS_CONV_PARTS_MAX];
    if (!g3_parts_table(c, parts, &q)) return 0;
    return (size_t)g3_wgrad_parts_columns(c, q, cols) * ((size_t)c.Cout * GCG * GK + c.Cout) * sizeof(float);
}

int msg3_parts_bwd_weight(const ConvP& c, const ms_conv1d_parts* parts, float* gw, float* gb, float beta, void* ws,
                          size_t ws_bytes, hipStream_t s) {
    G3Parts q;
    int cols[MS_CONV_PARTS_MAX];
    if (!g3_parts_table(c, parts, &q)) return MS_ERR_INVALID_ARG;
    const int gxn = g3_wgrad_parts_columns(c, q, cols);
    const size_t stride = (size_t)c.Cout * GCG * GK + c.Cout;
    if (!ws || ws_bytes < (size_t)gxn * stride * sizeof(float)) return MS_ERR_WORKSPACE;
    q.bx0[0] = 0;
    for (int i = 0; i < q.count; ++i) {
        q.bx0[i + 1] = q.bx0[i] + cols[i];
        q.a[i] = parts->x[i]; q.b[i] = parts->gy[i]; q.c[i] = c.act == MS_ACT_NONE ? nullptr : parts->y_act[i];
    }
    for (int i = q.count; i < MS_CONV_PARTS_MAX; ++i) q.bx0[i + 1] = q.bx0[q.count];
    ms_note_kernel(3, "k_gconv_split_wgrad_parts");
    hipLaunchKernelGGL(k_gconv_split_wgrad_parts, dim3(gxn, c.groups), dim3(256), 4 * W_WAVE, s, q, (float*)ws, stride);
    MS_CHECK_LAUNCH();
    return msg_reduce_slabs(c, (float*)ws, stride, gxn, gw, gb, beta, s);
}

bool msg3_bwd_data_applicable(const ConvP& p) {
    const char* e = getenv("MSYNTH_GCONV3");
    const bool on = !e || atoi(e) != 0;
    return on && p.K == GK && p.stride == GS && p.Cg == GCG && p.dil == 1 && p.pad == 20 &&
           p.pad_mode == MS_PAD_ZERO && p.Og == 16 && p.groups <= 65535 &&
           (long long)p.B * p.Cout * p.Lout * 4 < (1ll << 31);
}

const char* msg3_bwd_data_name(const ConvP&) { return "k_gconv_split_bwd_data"; }

int msg3_conv1d_bwd_data(const ConvP& p, const float* gy, const float* y_act, const float* w,
                         const float* gx_add, float* gx, hipStream_t s) {
    const int Lq = ms_ceil_div(p.Lin, GS);
    const int tiles = ms_ceil_div(Lq, BQ), nunits = p.B * tiles;
    const long long total_units = (long long)nunits * p.groups;
    const int upw = (int)((total_units + 2047) / 2048);             // ~2 waves per SIMD, `upw` units per wave
    const int gxn = ms_ceil_div(nunits, 4 * (upw > 0 ? upw : 1));
    const dim3 grid(gxn, p.groups);
    const bool vec = p.Lout % 4 == 0 && (((uintptr_t)gy) & 15) == 0 && (!y_act || (((uintptr_t)y_act) & 15) == 0);
    const bool vout = p.Lin % 4 == 0 && (((uintptr_t)gx) & 15) == 0 && (!gx_add || (((uintptr_t)gx_add) & 15) == 0);
#define MS_B3(V, VO)                                                                                                              \
    do {                                                                                                                          \
        ms_note_kernel(3, "k_gconv_split_bwd_data<%s, %s>", V ? "true" : "false", VO ? "true" : "false");                        \
        hipLaunchKernelGGL((k_gconv_split_bwd_data<V, VO>), grid, dim3(256), 4 * B_WAVE, s, p, tiles, nunits, gy, y_act, w, gx_add, \
                           gx);                                                                                                   \
    } while (0)
    if (vec && vout) MS_B3(true, true);
    else if (vec) MS_B3(true, false);
    else if (vout) MS_B3(false, true);
    else MS_B3(false, false);
#undef MS_B3
    MS_CHECK_LAUNCH();
    return MS_OK;
}

bool msg3_parts_bwd_data_applicable(const ConvP& c, const ms_conv1d_parts* parts) {
    G3Parts q;
    if (!g3_parts_table(c, parts, &q)) return false;
    for (int i = 0; i < q.count; ++i)
        if (!msg3_bwd_data_applicable(g3_host_part(c, q, i)) || !parts->gy[i] || !parts->gx[i]) return false;
    return q.Lout[0] % 4 == 0 && q.Lin[0] % 4 == 0 && al16(parts->gy[0]) && al16(parts->gx[0]) &&
           (!parts->y_act[0] || al16(parts->y_act[0])) && (!parts->gx_add[0] || al16(parts->gx_add[0]));
}

int msg3_parts_bwd_data(const ConvP& c, const ms_conv1d_parts* parts, const float* w, hipStream_t s) {
    G3Parts q;
    if (!g3_parts_table(c, parts, &q)) return MS_ERR_INVALID_ARG;
    int nunits[MS_CONV_PARTS_MAX];
    long long total_units = 0;
    for (int i = 0; i < q.count; ++i) {
        const int Lq = ms_ceil_div(q.Lin[i], GS);
        const int tiles = ms_ceil_div(Lq, BQ);
        nunits[i] = q.B[i] * tiles;
        q.u0[i] = tiles; q.u1[i] = nunits[i]; q.u2[i] = 0;
        q.a[i] = parts->gy[i]; q.b[i] = c.act == MS_ACT_NONE ? nullptr : parts->y_act[i]; q.c[i] = parts->gx_add[i];
        q.o[i] = parts->gx[i];
        total_units += (long long)nunits[i] * c.groups;
    }
    const int upw = (int)((total_units + 2047) / 2048);
    q.bx0[0] = 0;
    for (int i = 0; i < q.count; ++i) q.bx0[i + 1] = q.bx0[i] + ms_ceil_div(nunits[i], 4 * (upw > 0 ? upw : 1));
    for (int i = q.count; i < MS_CONV_PARTS_MAX; ++i) q.bx0[i + 1] = q.bx0[q.count];
    ms_note_kernel(3, "k_gconv_split_bwd_data_parts");
    hipLaunchKernelGGL(k_gconv_split_bwd_data_parts, dim3(q.bx0[q.count], c.groups), dim3(256), 4 * B_WAVE, s, q, w);
    MS_CHECK_LAUNCH();
    return MS_OK;
}
