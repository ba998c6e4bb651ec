// Shared pieces of the fused ResidualAtom kernels (atom_fused.hip: one atom per launch, forward / training / backward
// data; stack_fused.hip: a whole ResidualStack per launch, inference): vector types, the two operand splits, the block
// scale, the layout of the pre-split weight images.
#pragma once
#include "ms_common.h"

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

// NP = 2: weights are packed as the fp16 pieces of S_w w, S_w a power of two chosen PER CONV from the tensor's largest
// magnitude (it goes to [2^12, 2^13)): weights of any magnitude work (r04 packed 64 w: |w| >= 2^9 overflowed to inf), and an
// element within 2^16 of the tensor's largest weight keeps 22 significand bits.  A pack is two launches: partial maxima
// (W_NPART workgroups per conv, written into the image's tail), then the pack proper, whose threads reduce the partials and
// whose first thread leaves 1 / S_w in the tail for the consuming kernels.
constexpr int W_NPART = 16;               // partial maxima per conv
// tail of an NP = 2 image (floats): [conv][W_NPART] partial maxima, then [conv] 1 / S_w
__host__ __device__ constexpr int wtail_floats(int nconv) { return nconv * W_NPART + nconv; }

template <int NP> __host__ __device__ constexpr int xrs() { return NP * 32 + 16; }   // bytes per LDS column of one chunk

// (a, b) -> NP packed 16-bit pairs.  NP = 3: a = o[0].lo + o[1].lo + o[2].lo exactly (bf16, conv_rows3.hip);
// NP = 2: a = o[0].lo + o[1].lo to 22 bits (fp16; the caller has scaled a so that its block's largest magnitude sits near 2^14:
// the low piece of any element within 2^16 of that maximum is a normal or fully represented fp16 number, smaller elements
// keep an absolute error below 2^-25 -- 2^-39 of the maximum)
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

// block scale of a tile whose largest magnitude is m: S = 2^k with m S in [2^14, 2^15), and 1 / S (both exact powers of
// two); 1 for a zero / denormal-range / non-finite maximum
__device__ __forceinline__ void block_scale(float m, float& S, float& invS) {
    const unsigned eb = (__builtin_bit_cast(unsigned, m) >> 23) & 0xFFu;
    const bool ok = eb >= 16u && eb <= 250u;
    S = ok ? __builtin_bit_cast(float, (268u - eb) << 23) : 1.f;
    invS = ok ? __builtin_bit_cast(float, (eb - 14u) << 23) : 1.f;
}

// weight scale of a conv from its W_NPART partial maxima: S_w = 2^k with max S_w in [2^12, 2^13); 1 for an all-zero tensor
__device__ __forceinline__ void weight_scale(const float* __restrict__ pm, float& S, float& invS) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < W_NPART; ++i) m = fmaxf(m, pm[i]);
    const unsigned eb = (__builtin_bit_cast(unsigned, m) >> 23) & 0xFFu;
    const bool ok = eb >= 16u && eb <= 250u;
    S = ok ? __builtin_bit_cast(float, (266u - eb) << 23) : 1.f;
    invS = ok ? __builtin_bit_cast(float, (eb - 12u) << 23) : 1.f;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- weight image ------------------------------------------------------------------------------------------
// image[conv][ms][chunk][tap][piece][lane] (16 B each): lane's A fragment of the 32x32x16 MFMA for output rows
// ms*32 + (lane & 31), contraction channels chunk*16 + 8*(lane >> 5) + 0..7, tap `tap`.
__host__ __device__ constexpr size_t atom_conv_image_u4(int C, int NP) { return (size_t)(C / 32) * (C / 16) * 3 * NP * 64; }


}  // namespace
