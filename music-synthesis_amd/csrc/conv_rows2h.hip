// Two-half variant of the pipelined row-tile convolution (conv_rows2.hip) for the layers whose
// contraction is only 8 chunks long (the 128- and 64-channel k3 atoms, forward and backward data).
//
// There a launch is 8 chunks of K loop between a prologue and an output phase that nothing overlaps:
// all workgroups of the launch run in phase, so while the outputs stream out no MFMA runs anywhere.
// Here a workgroup owns TWO 64 x 128 output halves (256 consecutive samples of 64 output channels) and
// runs them back to back as one stream of 16 chunks: the first half's accumulators are parked in LDS
// (bias + activation applied) when its last chunk is done, and its residual loads and 16-byte output
// stores are issued piece by piece between the MFMAs of the second half's chunks, exactly like the
// staging pieces; the second half's first chunks prefetch under the first half's last ones.  Only the
// second half's output phase stays exposed.
//
// Requirements (the caller checks them and falls back to k_conv_rows2): K = 3, plain stride-1 input,
// zero padding, rows of L % 256 == 0 samples, M % 64 == 0 output rows, CK = 8 * CC channels, no split-K.
#include "conv_rows2.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int CC, int AM, bool HAS_RES, bool HAS_YACT>
__global__ __launch_bounds__(256) void k_conv_rows2h(Row2P p, const float* __restrict__ X,
                                                    const float* __restrict__ Xact,
                                                    const float* __restrict__ W,
                                                    const float* __restrict__ bias,
                                                    const float* __restrict__ res,
                                                    float* __restrict__ Y,
                                                    float* __restrict__ Yact) {
    constexpr int BM = 64, BN = 128, TN = 2, NCH = 8;   // one half; chunks per half
    constexpr int KK = CC * K, AS = KK + 1;
    constexpr int A4 = BM * (KK / 4), RA4 = (A4 + 255) / 256;
    constexpr int NXQ = msr2_nxq(CC, BN);
    constexpr int NP = RA4 + NXQ;
    constexpr int NSTEP = KK / 2;
    constexpr int PPS = (NP + NSTEP - 1) / NSTEP;
    constexpr int TP = BN + 4;
    constexpr int V4 = BN / 4;
    constexpr int NOUT = BM * V4 / 256;                 // output vectors per thread and half (8)
    static_assert(NOUT == NCH, "one output piece per chunk of the second half");
    static_assert(KK % 4 == 0, "chunk must be float4-sized");
    extern __shared__ float smem[];
    const int tile_floats = BM * AS + CC * p.PX;
    float* Ts = smem + 2 * tile_floats;                 // parked outputs of a half
    float* scratch = Ts + BM * TP;                      // 256 floats: sink for out-of-tile lanes
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * BM;
    const int tpr = p.L / (2 * BN);
    const int b0 = blockIdx.x / tpr, t0 = (blockIdx.x - b0 * tpr) * (2 * BN);

    // ---- chunk-invariant piece descriptors (see conv_rows2.hip)
    const int sh = ((p.off0 % 4) + 4) % 4;
    const int NVS = (p.SS + 6) >> 2;
    int a_goff[RA4], a_loff[RA4][AM == 1 ? 4 : 1];
    bool a_ok[RA4];
#pragma unroll
    for (int i = 0; i < RA4; ++i) {
        const int e = i * 256 + tid;
        const bool in = e < A4;
        if (AM != 1) {
            const int row = e / (KK / 4), q4 = e - row * (KK / 4);
            a_ok[i] = in;
            a_goff[i] = in ? (m0 + row) * p.KG + q4 * 4 : 0;
            a_loff[i][0] = in ? row * AS + q4 * 4 : -1;
        } else {
            constexpr int PER_CO = BM * K / 4;
            const int co_l = e / PER_CO, f4 = e - co_l * PER_CO;
            a_ok[i] = in;
            a_goff[i] = in ? (co_l * p.M + m0) * K + 4 * f4 : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = 4 * f4 + u, ci_l = f / K, j = f - ci_l * K;
                a_loff[i][u] = in ? ci_l * AS + co_l * K + (K - 1 - j) : -1;
            }
        }
    }
    int x_base[NXQ], x_loff[NXQ];
    bool x_ok0[NXQ], x_ok1[NXQ];
    unsigned x_em[NXQ];
#pragma unroll
    for (int q = 0; q < NXQ; ++q) {
        const int i = tid + 256 * q;
        const int c = i / NVS, sv = i - c * NVS;
        const int u0 = 4 * sv - sh;
        const int t = t0 + p.off0 + u0;                 // first half; the second is 128 samples on
        const bool in = c < CC;
        x_ok0[q] = in && t >= 0 && t < p.L;
        x_ok1[q] = in && t + BN >= 0 && t + BN < p.L;
        x_base[q] = in ? (b0 * p.CK + c) * p.L + t : 0;
        x_loff[q] = BM * AS + c * p.PX + u0;
        unsigned em = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (in && u0 + e >= 0 && u0 + e < p.SS) em |= 1u << e;
        x_em[q] = em;
    }

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int bbase[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bbase[j] = wn * TN * 32 + j * 32 + (lane & 31);

    constexpr bool ACTOP = AM == 1;
    float4 ra[RA4], rx[NXQ], rxa[ACTOP ? NXQ : 1];
    // g = chunk of the 16-chunk stream: half g / NCH, channel chunk g % NCH
    auto load_piece = [&](int pi, int g) {
        const bool live = g < 2 * NCH;
        const int half = g >= NCH ? 1 : 0;
        const int c0 = (g - half * NCH) * CC;
        if (pi < RA4) {
            const int i = pi;
            const int o = (live && a_ok[i]) ? a_goff[i] + c0 * (AM == 1 ? p.M * K : K) : 0;
            ra[i] = *reinterpret_cast<const float4*>(W + o);
        } else {
            const int q = pi - RA4;
            const bool ok = live && (half ? x_ok1[q] : x_ok0[q]);
            const int o = ok ? x_base[q] + half * BN + c0 * p.L : 0;
            rx[q] = *reinterpret_cast<const float4*>(X + o);
            if (ACTOP) rxa[q] = *reinterpret_cast<const float4*>(Xact + o);
        }
    };
    auto store_piece = [&](int pi, int g, float* buf) {
        const bool live = g < 2 * NCH;
        const int half = g >= NCH ? 1 : 0;
        if (pi < RA4) {
            const int i = pi;
            const bool ok = live && a_ok[i];
            const float e[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int lo = AM == 1 ? a_loff[i][u] : a_loff[i][0] + u;
                const bool in = (AM == 1 ? a_loff[i][u] : a_loff[i][0]) >= 0;
                float* d = in ? buf + lo : scratch + tid;
                *d = ok ? e[u] : 0.f;
            }
        } else {
            const int q = pi - RA4;
            const bool ok = live && (half ? x_ok1[q] : x_ok0[q]);
            float e[4] = {rx[q].x, rx[q].y, rx[q].z, rx[q].w};
            if (ACTOP) {
                const float a[4] = {rxa[q].x, rxa[q].y, rxa[q].z, rxa[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = a[i] > 0.f ? e[i] : e[i] * p.slope;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* d = (x_em[q] >> i) & 1u ? buf + x_loff[q] + i : scratch + tid;
                *d = ok ? e[i] : 0.f;
            }
        }
    };

    // outputs of a half: vector idx -> (row, 4 columns); half 0 starts at t0, half 1 at t0 + BN
    auto out_off = [&](int j, int half) {
        const int idx = tid + 256 * j;
        const int row = idx / V4, c4 = idx - row * V4;
        return ((size_t)b0 * p.M + m0 + row) * p.L + t0 + half * BN + 4 * c4;
    };
    auto ts_vec = [&](int j) {
        const int idx = tid + 256 * j;
        const int row = idx / V4, c4 = idx - row * V4;
        return *reinterpret_cast<const float4*>(Ts + row * TP + 4 * c4);
    };
    auto park = [&]() {                                 // accumulators (+ bias, activation) -> Ts
        const int mb = wm * 32 + 4 * h;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = bias ? bias[m0 + mb + (r & 3) + 8 * (r >> 2)] : 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * TN * 32 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                Ts[(mb + (r & 3) + 8 * (r >> 2)) * TP + col] = ms_apply_act(acc[j][r] + bv[r], p.act, p.slope);
                acc[j][r] = 0.f;
            }
        }
    };

    // prologue: chunk 0 -> buffer 0, chunk 1 -> registers
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) load_piece(pi, 0);
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) store_piece(pi, 0, smem);
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) load_piece(pi, 1);
    __syncthreads();

    const int arow = (wm * 32 + (lane & 31)) * AS + h;
    float4 rv_next = make_float4(0.f, 0.f, 0.f, 0.f);
    // one chunk: MFMAs out of buffer g & 1, staging pieces of g + 1 / g + 2 in between; in the second
    // half (SECOND) also output piece g - NCH of the parked first half
    auto chunk = [&](auto second_tag, int g) {
        constexpr bool SECOND = decltype(second_tag)::value;
        const float* As = smem + (g & 1) * tile_floats;
        const float* Xs = As + BM * AS;
        float* nbuf = smem + ((g & 1) ^ 1) * tile_floats;
        float a0, b0f[TN], a1, b1f[TN];
        auto frag = [&](int q, float& a, float (&b)[TN]) {
            const int kk0 = 2 * q, kk1 = 2 * q + 1;
            const int off_lo = (kk0 / K) * p.PX + (kk0 % K) * p.dil;
            const int off_hi = (kk1 / K) * p.PX + (kk1 % K) * p.dil;
            const int off = h ? off_hi : off_lo;
            a = As[arow + 2 * q];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Xs[off + bbase[j]];
        };
        auto mma = [&](float a, const float (&b)[TN]) {
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], acc[j], 0, 0, 0);
        };
        frag(0, a0, b0f);
#pragma unroll
        for (int q = 0; q < NSTEP; q += 2) {
            if (q + 1 < NSTEP) frag(q + 1, a1, b1f);
            mma(a0, b0f);
#pragma unroll
            for (int pp = 0; pp < 2 * PPS; ++pp) {
                const int pi = (q / 2) * (2 * PPS) + pp;
                if (pi < NP) {
                    store_piece(pi, g + 1, nbuf);
                    load_piece(pi, g + 2);
                }
            }
            if (SECOND && q == (NSTEP / 4) * 2) {       // one parked output vector of the first half
                const int j = g - NCH;
                const size_t go = out_off(j, 0);
                const float4 tv = ts_vec(j);
                if (HAS_YACT) *reinterpret_cast<float4*>(Yact + go) = tv;
                float4 v = tv;
                if (HAS_RES) {
                    v.x += rv_next.x; v.y += rv_next.y; v.z += rv_next.z; v.w += rv_next.w;
                    const int jn = j + 1 < NOUT ? j + 1 : j;
                    rv_next = *reinterpret_cast<const float4*>(res + out_off(jn, 0));
                }
                *reinterpret_cast<float4*>(Y + go) = v;
            }
            if (q + 2 < NSTEP) frag(q + 2, a0, b0f);
            if (q + 1 < NSTEP) mma(a1, b1f);
        }
        __syncthreads();
    };
    for (int g = 0; g < NCH; ++g) chunk(std::false_type{}, g);
    park();
    if (HAS_RES) rv_next = *reinterpret_cast<const float4*>(res + out_off(0, 0));
    __syncthreads();
    for (int g = NCH; g < 2 * NCH; ++g) chunk(std::true_type{}, g);

    // second half: exposed output phase
    park();
    __syncthreads();
    float4 tv[NOUT], rv[HAS_RES ? NOUT : 1];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        tv[j] = ts_vec(j);
        if (HAS_RES) rv[j] = *reinterpret_cast<const float4*>(res + out_off(j, 1));
    }
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        const size_t go = out_off(j, 1);
        if (HAS_YACT) *reinterpret_cast<float4*>(Yact + go) = tv[j];
        float4 v = tv[j];
        if (HAS_RES) { v.x += rv[j].x; v.y += rv[j].y; v.z += rv[j].z; v.w += rv[j].w; }
        *reinterpret_cast<float4*>(Y + go) = v;
    }
}

template <int K, int CC, int AM, bool HAS_RES, bool HAS_YACT>
int launch_h(const Row2P& p, const float* X, const float* Xact, const float* W, const float* bias,
             const float* res, float* Y, float* Yact, dim3 grid, hipStream_t s) {
    const size_t fl = (size_t)2 * (64 * (CC * K + 1) + CC * p.PX) + 64 * (128 + 4) + 256;
    const size_t lds = fl * sizeof(float);
    if (lds > 80 * 1024) return MS_ERR_UNSUPPORTED;     // two workgroups per CU or not at all
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_rows2h<K, CC, AM, HAS_RES, HAS_YACT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_conv_rows2h<K, CC, AM, HAS_RES, HAS_YACT>), grid, dim3(256), lds, s, p, X, Xact, W, bias,
                       res, Y, Yact);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

}  // namespace

bool msr2h_supported(int K, int CC, int act_mode, const Row2P& p, bool has_res, bool has_yact) {
    // tuning / test switch: 0 off (default), 1 on where the launch fills the chip, 2 on for every fitting shape.
    // Measured (tools/microbench_rows2h.py, B=32): 128 channels 81 -> 76 us (forward), 85 -> 78 us (backward
    // data); 64 channels 44 -> 40 us -- but +0.16 % on the whole train step (three same-box A/B pairs), so it
    // stays off until the output phase is worth more than that.
    const char* e = getenv("MSYNTH_ROWS2H");
    const int on = e ? atoi(e) : 0;
    if (!on) return false;
    if (K != 3 || !(CC == 8 || CC == 16) || !(act_mode == 0 || act_mode == 1)) return false;
    if (p.R != 1 || p.L % 256 || p.M % 64 || p.CK != 8 * CC || p.CKs != p.CK) return false;
    if (act_mode == 1 && has_yact) return false;
    if (has_yact && !has_res) return false;
    if (on != 2 && (long long)p.B * (p.L / 256) * (p.M / 64) < 384) return false;   // keep the launch >= 1.5 workgroups per CU
    const size_t fl = (size_t)2 * (64 * (CC * K + 1) + CC * p.PX) + 64 * 132 + 256;
    return fl * sizeof(float) <= 80 * 1024 && CC * ((p.SS + 6) / 4) <= 256 * msr2_nxq(CC, 128);
}

int msr2h_launch(int K, int CC, int act_mode, const Row2P& p, const float* X, const float* Xact, const float* W,
                 const float* bias, const float* res, float* Y, float* Yact, hipStream_t s) {
    const dim3 grid((unsigned)(p.B * (p.L / 256)), (unsigned)(p.M / 64), 1);
#define MSR2H(C, A, R, YA) return launch_h<3, C, A, R, YA>(p, X, Xact, W, bias, res, Y, Yact, grid, s)
    const bool r = res != nullptr, ya = Yact != nullptr;
    if (CC == 16) {
        if (act_mode == 0) { if (r && ya) MSR2H(16, 0, true, true); if (r) MSR2H(16, 0, true, false); if (!ya) MSR2H(16, 0, false, false); }
        else { if (r) MSR2H(16, 1, true, false); MSR2H(16, 1, false, false); }
    } else if (CC == 8) {
        if (act_mode == 0) { if (r && ya) MSR2H(8, 0, true, true); if (r) MSR2H(8, 0, true, false); if (!ya) MSR2H(8, 0, false, false); }
        else { if (r) MSR2H(8, 1, true, false); MSR2H(8, 1, false, false); }
    }
#undef MSR2H
    return MS_ERR_UNSUPPORTED;
}
