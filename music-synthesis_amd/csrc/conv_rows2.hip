// Row-tile forward / backward-data convolution, software-pipelined (second generation of
// k_conv_mfma_rows in conv_mfma.hip; same contraction, same tile shapes, same epilogues).
//
// What changed, and why (measured on the weight-gradient kernel first, wgrad_rows.hip):
//  * activations are staged with 16-byte loads of ALIGNED vectors of the raw rows (a row segment
//    with its dilation halo starts off a 16-byte boundary by sh = off0 mod 4; the four elements of
//    a vector just land sh columns earlier in LDS) instead of one dword load per (channel, column);
//  * two LDS buffers: the chunk being multiplied and the chunk being staged; ONE barrier per chunk;
//  * the chunk is fully unrolled and branch-free, with the LDS stores of chunk c+1 and the global
//    loads of chunk c+2 placed between the MFMAs of chunk c, so they issue in the matrix pipe's
//    shadow (one wave per SIMD: nothing else would hide them).
// Requirements (the caller falls back to k_conv_mfma_rows otherwise): zero padding, L % 4 == 0,
// 16-byte aligned tensors, plain stride-1 input (IN_S == 1), activation handling one of
//   AM 0: none (forward);  AM 1: LeakyReLU derivative from the saved output, weights read in the forward
//   layout (backward data);  AM 2: the same derivative with pre-packed weights and IN_S > 1: the CC
//   rows of a chunk are the IN_S phases of CC/IN_S channels of a stride-IN_S signal (transposed-conv
//   backward data), staged from contiguous 16-byte loads of the phase-interleaved spans.
#include "conv_rows2.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WGM, int WGN, int TM, int TN, int K, int CC, int AM, int EPI_S, int IN_S = 1>
__global__ __launch_bounds__(256) void k_conv_rows2(Row2P p, const float* __restrict__ X,
                                                   const float* __restrict__ Xact,
                                                   const float* __restrict__ W,
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ res,
                                                   float* __restrict__ Y,
                                                   float* __restrict__ Yact) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int KK = CC * K;               // GEMM-K elements per chunk
    constexpr int AS = KK + 1;               // odd LDS stride of the weight tile
    constexpr int A4 = BM * (KK / 4);        // 16-byte loads per weight tile
    constexpr int RA4 = (A4 + 255) / 256;    // weight pieces per thread
    // IN_S == 0: SHORT-ROW mode.  A tile packs R whole rows of length L < BN (any L, also odd): the CC*L
    // floats of one (batch row, channel chunk) are contiguous and start 16-byte aligned, so they are staged
    // as aligned vectors whose 4 elements scatter to (channel, column) -- no halo loads (the halo is zero
    // padding: LDS is cleared once), no alignment requirement on L.  The epilogue writes the BM*L
    // contiguous outputs of a segment the same way.
    constexpr bool SR = IN_S == 0;
    constexpr int NXQ = SR ? (CC * BN / 4 + 255) / 256 : msr2_nxq(CC, BN);    // activation pieces per thread
    constexpr int NP = RA4 + NXQ;
    constexpr int NSTEP = KK / 2;            // MFMA k-pair steps per chunk
    constexpr int PPS = (NP + NSTEP - 1) / NSTEP;
    // HALF (transposed-conv forward, kernel 2S, padding S/2): every output phase has two live taps of the
    // 3-tap window -- {-1,0} for the low phases, {0,+1} for the high ones.  GEMM rows come packed in groups
    // of 64 = [32 low-phase rows | 32 high-phase rows] of the same output channels, so a wave's two M
    // sub-tiles differ only in the window offset of their B fragments.
    constexpr bool HALF = K == 2 && EPI_S > 0 && IN_S == 1;
    constexpr int NBH = HALF ? 2 : 1;
    static_assert(!HALF || TM == 2, "HALF needs two M sub-tiles per wave");
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(KK % 4 == 0, "chunk must be float4-sized");
    extern __shared__ float smem[];
    const int tile_floats = BM * AS + CC * p.PX;
    float* scratch = smem + p.scratch_off;           // 256 floats: sink for out-of-tile lanes
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5;
    const int wm = wid / WGN, wn = wid % WGN;
    const int m0 = blockIdx.y * BM;
    int b0, t0;
    if (p.R == 1) { b0 = blockIdx.x / p.tiles_per_row; t0 = (blockIdx.x - b0 * p.tiles_per_row) * BN; }
    else { b0 = blockIdx.x * p.R; t0 = 0; }

    // ---- chunk-invariant piece descriptors (only the channel base moves from chunk to chunk)
    const int sh = ((p.off0 % 4) + 4) % 4;           // segment start within its 16-byte vector
    const int NVS = (p.SS + 6) >> 2;                 // aligned vectors covering one segment
    const int NVT = p.R * NVS;                       // ... one channel row of the tile
    // AM 0: W is [M][CK][K] (forward weights, or pre-transposed backward weights): a 16-byte piece is 4
    //       consecutive (c, j) of one output row.
    // AM 1: backward data straight from the forward layout W[co][ci][j] (no transpose pass): for one
    //       co of the chunk the BM*K floats of rows ci = m0.. are contiguous; a piece is 4 consecutive
    //       (ci, j) of one co, scattered to As[ci][co_l*K + (K-1-j)] (taps flipped).
    int a_goff[RA4], a_loff[RA4][AM == 1 ? 4 : 1];
    bool a_ok[RA4];
#pragma unroll
    for (int i = 0; i < RA4; ++i) {
        const int e = i * 256 + tid;
        const bool in = e < A4;
        if (AM != 1) {
            const int row = e / (KK / 4), q4 = e - row * (KK / 4);
            a_ok[i] = in && m0 + row < p.M;
            a_goff[i] = a_ok[i] ? (m0 + row) * p.KG + q4 * 4 : 0;
            a_loff[i][0] = in ? row * AS + q4 * 4 : -1;
        } else {
            constexpr int PER_CO = BM * K / 4;               // pieces per output channel of the chunk
            const int co_l = e / PER_CO, f4 = e - co_l * PER_CO;
            a_ok[i] = in && 4 * f4 < (p.M - m0) * K;          // (M, m0 multiples of 4: all 4 in or out)
            a_goff[i] = a_ok[i] ? (co_l * p.M + m0) * K + 4 * f4 : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int f = 4 * f4 + u, ci_l = f / K, j = f - ci_l * K;
                a_loff[i][u] = in ? ci_l * AS + co_l * K + (K - 1 - j) : -1;
            }
        }
    }
    int x_goff[NXQ], x_loff[NXQ];
    bool x_ok[NXQ];
    unsigned x_em[NXQ];                              // which of the 4 elements fall inside the segment
    int x_loff4[IN_S != 1 ? NXQ : 1][4];             // IN_S != 1: LDS offset of each element (-1: none)
    if (SR) {
        const int NV = (CC * p.L) >> 2;              // vectors of one (segment, chunk) span
#pragma unroll
        for (int q = 0; q < NXQ; ++q) {
            const int i = tid + 256 * q;
            const int r = i / NV, v = i - r * NV;
            const bool in = r < p.R;
            x_ok[q] = in && b0 + r < p.B;
            x_goff[q] = x_ok[q] ? (b0 + r) * p.CK * p.L + 4 * v : 0;
            x_loff[q] = 0; x_em[q] = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pos = 4 * v + e;
                const int c = pos / p.L, t = pos - c * p.L;
                x_loff4[q][e] = in ? BM * AS + c * p.PX + r * p.SS + t - p.off0 : -1;
            }
        }
    } else if (IN_S == 1) {
#pragma unroll
        for (int q = 0; q < NXQ; ++q) {
            const int i = tid + 256 * q;
            const int c = i / NVT, v = i - c * NVT;
            const int r = v / NVS, sv = v - r * NVS;
            const int u0 = 4 * sv - sh;
            const int t = t0 + p.off0 + u0;                // multiple of 4: the vector is all in or all out
            const bool in = c < CC;
            x_ok[q] = in && b0 + r < p.B && t >= 0 && t < p.L;
            x_goff[q] = x_ok[q] ? ((b0 + r) * p.CK + c) * p.L + t : 0;
            x_loff[q] = BM * AS + c * p.PX + r * p.SS + u0;
            unsigned em = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (in && u0 + e >= 0 && u0 + e < p.SS) em |= 1u << e;
            x_em[q] = em;
        }
    } else {
        // a piece = 16-byte vector v of the SS*IN_S phase-interleaved floats of (segment r, channel co_l)
        const int span = p.SS * IN_S;
        const int NVco = (span + 6) >> 2;
        const int shx = ((((p.off0 * IN_S) % 4) + 4) % 4);  // span start within its 16-byte vector
#pragma unroll
        for (int q = 0; q < NXQ; ++q) {
            const int i = tid + 256 * q;
            const int sc = i / NVco, v = i - sc * NVco;     // sc = (segment, channel) pair
            const int r = sc / (CC / IN_S), co_l = sc - r * (CC / IN_S);
            const bool in = r < p.R;
            const int gp = (t0 + p.off0) * IN_S - shx + 4 * v;   // multiple of 4: all in or all out of the row
            x_ok[q] = in && b0 + r < p.B && gp >= 0 && gp < p.L * IN_S;
            x_goff[q] = x_ok[q] ? ((b0 + r) * p.CK + co_l * IN_S) * p.L + gp : 0;
            x_loff[q] = 0; x_em[q] = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int pos = 4 * v + e - shx;
                const int rr = pos % IN_S, u = pos / IN_S;
                x_loff4[q][e] = (in && pos >= 0 && pos < span) ? BM * AS + (co_l * IN_S + rr) * p.PX + r * p.SS + u : -1;
            }
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // B-fragment base column of this lane for each N sub-tile (output column -> LDS column)
    int bbase[TN];
    bool nvalid[TN];
    int ob[TN], ot[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * TN * 32 + j * 32 + (lane & 31);
        const int r = nl / p.Lt, tc = nl - r * p.Lt;
        nvalid[j] = r < p.R && b0 + r < p.B && t0 + tc < p.L;
        bbase[j] = nvalid[j] ? r * p.SS + tc : 0;
        ob[j] = b0 + r;
        ot[j] = t0 + tc;
    }

    constexpr bool ACTOP = AM == 1 || AM == 2;       // a second operand carries the activation derivative
    float4 ra[RA4], rx[NXQ], rxa[ACTOP ? NXQ : 1];
    auto load_piece = [&](int pi, int c0, bool live) {        // c0: first channel of the chunk
        if (pi < RA4) {
            const int i = pi;
            const int o = (live && a_ok[i]) ? a_goff[i] + c0 * (AM == 1 ? p.M * K : K) : 0;
            ra[i] = *reinterpret_cast<const float4*>(W + o);
        } else {
            const int q = pi - RA4;
            const int o = (live && x_ok[q]) ? x_goff[q] + c0 * p.L : 0;
            rx[q] = *reinterpret_cast<const float4*>(X + o);
            if (ACTOP) rxa[q] = *reinterpret_cast<const float4*>(Xact + o);
        }
    };
    auto store_piece = [&](int pi, bool live, float* buf) {
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
            const bool ok = live && x_ok[q];
            float e[4] = {rx[q].x, rx[q].y, rx[q].z, rx[q].w};
            if (ACTOP) {
                const float a[4] = {rxa[q].x, rxa[q].y, rxa[q].z, rxa[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = a[i] > 0.f ? e[i] : e[i] * p.slope;
            } else if (AM == 3) {                    // LeakyReLU in front of the conv, applied on load
#pragma unroll
                for (int i = 0; i < 4; ++i) e[i] = e[i] > 0.f ? e[i] : e[i] * p.slope;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* d;
                if (IN_S != 1) d = x_loff4[q][i] >= 0 ? buf + x_loff4[q][i] : scratch + tid;
                else d = (x_em[q] >> i) & 1u ? buf + x_loff[q] + i : scratch + tid;
                *d = ok ? e[i] : 0.f;
            }
        }
    };

    // split-K: slice z contracts channels [z*CKs, min((z+1)*CKs, CK)) into its own output slab
    const int cbeg = blockIdx.z * p.CKs;
    const int nchunks = ((cbeg + p.CKs < p.CK ? cbeg + p.CKs : p.CK) - cbeg) / CC;
    Y += (size_t)blockIdx.z * p.zstride;

    // prologue: chunk 0 -> buffer 0, chunk 1 -> registers
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) load_piece(pi, cbeg, true);
    if (SR) {                                        // the halo columns are never written: clear both buffers
        for (int i = tid; i < 2 * tile_floats; i += 256) smem[i] = 0.f;
        __syncthreads();
    }
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) store_piece(pi, true, smem);
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) load_piece(pi, cbeg + CC, nchunks > 1);
    __syncthreads();

    const int arow = (wm * TM * 32 + (lane & 31)) * AS + h;
    for (int ch = 0; ch < nchunks; ++ch) {
        const float* As = smem + (ch & 1) * tile_floats;
        const float* Xs = As + BM * AS;
        float* nbuf = smem + ((ch & 1) ^ 1) * tile_floats;
        const bool live1 = ch + 1 < nchunks, live2 = ch + 2 < nchunks;
        const int c2 = cbeg + (ch + 2) * CC;
        float a0[TM], b0f[NBH][TN], a1[TM], b1f[NBH][TN];
        auto frag = [&](int q, float (&a)[TM], float (&b)[NBH][TN]) {
            // k-pair q: lane half h takes GEMM-k element 2q + h = (channel, tap) of the chunk
            // IN_S > 1 with K = 2 (transposed conv, kernel 2*IN_S, padding IN_S/2): of the 3-tap window a
            // phase channel has two live taps -- {0,+1} for the low phases, {-1,0} for the high ones
            const int kk0 = 2 * q, kk1 = 2 * q + 1;
            constexpr bool TWO = IN_S > 1 && K == 2;
            const int jb0 = TWO && ((kk0 / K) % IN_S) < IN_S / 2 ? 1 : 0;
            const int jb1 = TWO && ((kk1 / K) % IN_S) < IN_S / 2 ? 1 : 0;
            const int off_lo = (kk0 / K) * p.PX + (kk0 % K + jb0) * p.dil;
            const int off_hi = (kk1 / K) * p.PX + (kk1 % K + jb1) * p.dil;
            const int off = h ? off_hi : off_lo;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[arow + i * 32 * AS + 2 * q];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = Xs[off + bbase[j]];
            if (HALF) {         // the high-phase rows (M sub-tile 1) read the window one tap later
#pragma unroll
                for (int j = 0; j < TN; ++j) b[NBH - 1][j] = Xs[off + p.dil + bbase[j]];
            }
        };
        auto mma = [&](const float (&a)[TM], const float (&b)[NBH][TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[HALF ? i : 0][j], acc[i][j], 0, 0, 0);
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
                    store_piece(pi, live1, nbuf);        // chunk ch+1: registers -> the other buffer
                    load_piece(pi, c2, live2);           // chunk ch+2: into the registers just freed
                }
            }
            if (q + 2 < NSTEP) frag(q + 2, a0, b0f);
            if (q + 1 < NSTEP) mma(a1, b1f);
        }
        __syncthreads();
    }

    if (EPI_S == 0) {
        // Dword stores straight from the accumulators (32 consecutive samples x 2 channels per
        // instruction) are store-ISSUE bound: 19.6 us of an 80 us launch for the 128-channel layers.
        // The tile goes through LDS instead (free after the last chunk's barrier): bias + activation
        // on the way in, then 16-byte rows out -- residual loads and both output stores are 16 bytes
        // per lane, 512 contiguous bytes per 32 lanes.
        constexpr int TP = BN + 4;                       // == 4 (mod 32): the two lane halves miss each other
        float* Ts = smem;
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
        constexpr int V4 = BN / 4;                       // vectors per tile row
        constexpr int NQ = BM * V4 / 256;                // vectors per thread
        float4 tv[NQ], rv[NQ];
        size_t go[NQ];
        bool ok[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = tid + 256 * q;
            if (SR) {
                // vector v of the BM*L contiguous outputs of segment r (rows m0.., all L samples each)
                const int NVo = (BM * p.L) >> 2;
                const int r = idx / NVo, v = idx - r * NVo;
                const int rows_left = p.M - m0;              // (M % 4 == 0: a vector never straddles row M)
                ok[q] = r < p.R && b0 + r < p.B && (4 * v) / p.L < rows_left;
                go[q] = ok[q] ? ((size_t)(b0 + r) * p.M + m0) * p.L + 4 * v : 0;
                float e4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int pos = ok[q] ? 4 * v + e : 0;
                    const int row = pos / p.L, t = pos - row * p.L;
                    e4[e] = Ts[row * TP + (ok[q] ? r : 0) * p.Lt + t];
                }
                tv[q] = make_float4(e4[0], e4[1], e4[2], e4[3]);
            } else {
                const int row = idx / V4, c4 = idx - row * V4;
                const int nl = 4 * c4;                       // tile column of the vector's first sample
                const int r = nl / p.Lt, tc = nl - r * p.Lt; // Lt % 4 == 0: the 4 samples share a row
                ok[q] = m0 + row < p.M && r < p.R && b0 + r < p.B && t0 + tc < p.L;
                go[q] = ok[q] ? ((size_t)(b0 + r) * p.M + m0 + row) * p.L + t0 + tc : 0;
                tv[q] = *reinterpret_cast<const float4*>(Ts + row * TP + nl);
            }
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
        // transposed conv: GEMM row m' = co*S + r is output phase r of channel co; a lane's 4
        // consecutive accumulator rows are 4 consecutive output samples (S = 8) or 2 x 2 (S = 2)
        constexpr int ES = EPI_S > 0 ? EPI_S : 1;
        const int Cout = p.M / ES;
        const size_t Lo = (size_t)p.L * ES;
        if (HALF) {
            const int mg = m0 + wm * 64;                     // this wave's 64-row group
            if (mg < p.M) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!nvalid[j]) continue;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        if (EPI_S >= 4) {       // row-in-half 8rg+4h+u = (channel 2rg+h, phase u) for S = 8
                            const int co = mg / ES + (8 * rg + 4 * h) / (ES / 2);
                            const float bv = bias ? bias[co] : 0.f;
                            float* yp = Y + ((size_t)ob[j] * Cout + co) * Lo + (size_t)ot[j] * ES;
#pragma unroll
                            for (int i = 0; i < TM; ++i) {
                                float4 v;
                                v.x = ms_apply_act(acc[i][j][4 * rg + 0] + bv, p.act, p.slope);
                                v.y = ms_apply_act(acc[i][j][4 * rg + 1] + bv, p.act, p.slope);
                                v.z = ms_apply_act(acc[i][j][4 * rg + 2] + bv, p.act, p.slope);
                                v.w = ms_apply_act(acc[i][j][4 * rg + 3] + bv, p.act, p.slope);
                                *reinterpret_cast<float4*>(yp + (i % 2) * (ES / 2)) = v;
                            }
                        } else {                // S = 2: row-in-half = channel, sub-tile = phase
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int co = mg / 2 + 8 * rg + 4 * h + u;
                                const float bv = bias ? bias[co] : 0.f;
                                float2 v;
                                v.x = ms_apply_act(acc[0][j][4 * rg + u] + bv, p.act, p.slope);
                                v.y = ms_apply_act(acc[TM - 1][j][4 * rg + u] + bv, p.act, p.slope);
                                *reinterpret_cast<float2*>(Y + ((size_t)ob[j] * Cout + co) * Lo + (size_t)ot[j] * 2) = v;
                            }
                        }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (!nvalid[j]) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float bq[4][2];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int mb = m0 + wm * TM * 32 + i * 32 + 8 * rg + 4 * h;
                    const int co = (mb < p.M ? mb : 0) / ES;
                    bq[rg][0] = bias ? bias[co] : 0.f;
                    bq[rg][1] = (bias && EPI_S < 4) ? bias[co + 1 < Cout ? co + 1 : co] : 0.f;
                }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int mb = m0 + wm * TM * 32 + i * 32 + 8 * rg + 4 * h;
                    if (mb >= p.M) continue;
                    if (EPI_S >= 4) {
                        const int co = mb / ES, ph = mb - co * ES;
                        const float bv = bq[rg][0];
                        float4 v;
                        v.x = ms_apply_act(acc[i][j][4 * rg + 0] + bv, p.act, p.slope);
                        v.y = ms_apply_act(acc[i][j][4 * rg + 1] + bv, p.act, p.slope);
                        v.z = ms_apply_act(acc[i][j][4 * rg + 2] + bv, p.act, p.slope);
                        v.w = ms_apply_act(acc[i][j][4 * rg + 3] + bv, p.act, p.slope);
                        *reinterpret_cast<float4*>(Y + ((size_t)ob[j] * Cout + co) * Lo + (size_t)ot[j] * ES + ph) = v;
                    } else {
                        const int co = mb / 2;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const float bv = bq[rg][u];
                            float2 v;
                            v.x = ms_apply_act(acc[i][j][4 * rg + 2 * u + 0] + bv, p.act, p.slope);
                            v.y = ms_apply_act(acc[i][j][4 * rg + 2 * u + 1] + bv, p.act, p.slope);
                            *reinterpret_cast<float2*>(Y + ((size_t)ob[j] * Cout + co + u) * Lo + (size_t)ot[j] * 2) = v;
                        }
                    }
                }
            }
        }
    }
}

template <int WGM, int WGN, int TM, int TN, int K, int CC, int AM, int EPI_S, int IN_S = 1>
int launch_inst(const Row2P& p, const float* X, const float* Xact, const float* W, const float* bias,
                const float* res, float* Y, float* Yact, dim3 grid, hipStream_t s) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    size_t fl = (size_t)2 * (BM * (CC * K + 1) + CC * p.PX);
    if (EPI_S == 0 && fl < (size_t)BM * (BN + 4)) fl = (size_t)BM * (BN + 4);   // output transpose tile
    const size_t lds = (fl + 256) * sizeof(float);
    if (lds > 150 * 1024) return MS_ERR_UNSUPPORTED;
    static unsigned long long attr_set = 0;                    // > 64 KiB of dynamic LDS needs the opt-in once
    if (ms_first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_rows2<WGM, WGN, TM, TN, K, CC, AM, EPI_S, IN_S>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        ms_done_on_device(attr_set);
    }
    Row2P pp = p;
    pp.scratch_off = (int)fl;
    ms_note_kernel(0, "k_conv_rows2<%d, %d, %d, %d, %d, %d, %d, %d, %d>", WGM, WGN, TM, TN, K, CC, AM, EPI_S, IN_S);
    hipLaunchKernelGGL((k_conv_rows2<WGM, WGN, TM, TN, K, CC, AM, EPI_S, IN_S>), grid, dim3(256), lds, s, pp, X, Xact,
                       W, bias, res, Y, Yact);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

template <int K, int CC, int AM, int EPI_S, int IN_S = 1>
int launch_tile(int tile, const Row2P& p, const float* X, const float* Xact, const float* W,
                const float* bias, const float* res, float* Y, float* Yact, dim3 grid, hipStream_t s) {
    switch (tile) {
        case MSR2_128x128: return launch_inst<2, 2, 2, 2, K, CC, AM, EPI_S, IN_S>(p, X, Xact, W, bias, res, Y, Yact, grid, s);
        case MSR2_64x128: return launch_inst<2, 2, 1, 2, K, CC, AM, EPI_S, IN_S>(p, X, Xact, W, bias, res, Y, Yact, grid, s);
        case MSR2_64x64: return launch_inst<2, 2, 1, 1, K, CC, AM, EPI_S, IN_S>(p, X, Xact, W, bias, res, Y, Yact, grid, s);
        case MSR2_32x256: return launch_inst<1, 4, 1, 2, K, CC, AM, EPI_S, IN_S>(p, X, Xact, W, bias, res, Y, Yact, grid, s);
        default: return MS_ERR_UNSUPPORTED;
    }
}

}  // namespace

bool msr2_supported(int tile, int K, int CC, int act_mode, int epi_s, const Row2P& p, int in_s) {
    const char* e = getenv("MSYNTH_ROWS2");            // tuning / test switch (0 disables)
    if (e && atoi(e) == 0) return false;
    if (tile < 0 || tile > MSR2_32x256) return false;
    if (act_mode < 0 || act_mode > 3) return false;
    if (in_s == 0) {      // short-row mode: R whole rows of any length per tile (the 1024 -> 1024 k5 conv at L = 17 / 9)
        const int bnS = tile == MSR2_32x256 ? 256 : (tile == MSR2_64x64 ? 64 : 128);
        // (backward data only: 92 vs 110 us at L = 17, 56 vs 76 us at L = 9; the forward measured 5-8 % slower
        //  than the first-generation kernel, as at L = 32)
        return K == 5 && CC == 16 && epi_s == 0 && act_mode == 1 && p.tiles_per_row == 1 &&
               p.R * p.L <= bnS && (tile == MSR2_64x128 || tile == MSR2_64x64 || tile == MSR2_128x128) && p.M % 4 == 0;
    }
    if (p.L % 4) return false;
    if (in_s != 1) {      // transposed-conv backward data: phase-split input rows, pre-packed weights
        if (!(in_s == 2 || in_s == 8) || (act_mode != 2 && act_mode != 0) || K != 2 || CC != 8 || epi_s != 0) return false;
        const int bnI = tile == MSR2_32x256 ? 256 : (tile == MSR2_64x64 ? 64 : 128);
        return p.R * (CC / in_s) * ((p.SS * in_s + 6) / 4) <= 256 * msr2_nxq(CC, bnI);
    }
    if (act_mode == 2 || (act_mode == 3 && K != 2)) return false;
    // transposed-conv forward with the two live taps per phase (act_mode 3: LeakyReLU in front, on load):
    // 128x128 tile or 64x128 as 1x4 waves, whole 64-row groups
    if (K == 2) return CC == 8 && (act_mode == 0 || act_mode == 3) && (epi_s == 2 || epi_s == 8) &&
                       (tile == MSR2_128x128 || tile == MSR2_64x128) && p.M % 64 == 0 &&
                       CC * (p.R * ((p.SS + 6) / 4)) <= 256 * msr2_nxq(CC, 128);
    const bool k3 = K == 3 && (CC == 8 || CC == 16) && (epi_s == 0 || ((epi_s == 2 || epi_s == 8) && CC == 8 && act_mode == 0));
    // (k5 forward: the first-generation kernel measured 8 % faster, 194 vs 212 us at B*L = 2048)
    const bool k5 = K == 5 && CC == 16 && epi_s == 0 && act_mode == 1;
    // pointwise convs (shortcuts): measured 44 vs 60 us on the 128-channel layer, but no better than the
    // first generation on the smaller tiles (4 chunks of 32 channels: the prologue dominates)
    const bool k1 = K == 1 && CC == 32 && epi_s == 0 && act_mode == 0 && tile == MSR2_128x128;
    if (!k3 && !k5 && !k1) return false;
    int bn = tile == MSR2_32x256 ? 256 : (tile == MSR2_64x64 ? 64 : 128);
    const int nvt = p.R * ((p.SS + 6) / 4);
    return CC * nvt <= 256 * msr2_nxq(CC, bn);
}

int msr2_launch(int tile, int K, int CC, int act_mode, int epi_s, const Row2P& p, const float* X,
                const float* Xact, const float* W, const float* bias, const float* res, float* Y,
                float* Yact, unsigned gx, unsigned gy, unsigned gz, hipStream_t s, int in_s) {
    const dim3 grid(gx, gy, gz);
    if (in_s == 0) return launch_tile<5, 16, 1, 0, 0>(tile, p, X, Xact, W, bias, res, Y, Yact, grid, s);
    if (in_s == 8 && act_mode == 2) return launch_tile<2, 8, 2, 0, 8>(tile, p, X, Xact, W, bias, res, Y, Yact, grid, s);
    if (in_s == 2 && act_mode == 2) return launch_tile<2, 8, 2, 0, 2>(tile, p, X, Xact, W, bias, res, Y, Yact, grid, s);
    if (in_s == 8) return launch_tile<2, 8, 0, 0, 8>(tile, p, X, Xact, W, bias, res, Y, Yact, grid, s);
    if (in_s == 2) return launch_tile<2, 8, 0, 0, 2>(tile, p, X, Xact, W, bias, res, Y, Yact, grid, s);
#define MSR2_HALF(A)                                                                                              \
    if (K == 2 && CC == 8 && act_mode == A && tile == MSR2_128x128) {                                            \
        if (epi_s == 8) return launch_inst<2, 2, 2, 2, 2, 8, A, 8>(p, X, Xact, W, bias, res, Y, Yact, grid, s);  \
        if (epi_s == 2) return launch_inst<2, 2, 2, 2, 2, 8, A, 2>(p, X, Xact, W, bias, res, Y, Yact, grid, s);  \
    }                                                                                                            \
    if (K == 2 && CC == 8 && act_mode == A && tile == MSR2_64x128) {   /* 64 x 128 as 1 x 4 waves of 64 x 32 */  \
        if (epi_s == 8) return launch_inst<1, 4, 2, 1, 2, 8, A, 8>(p, X, Xact, W, bias, res, Y, Yact, grid, s);  \
        if (epi_s == 2) return launch_inst<1, 4, 2, 1, 2, 8, A, 2>(p, X, Xact, W, bias, res, Y, Yact, grid, s);  \
    }
    MSR2_HALF(0)
    MSR2_HALF(3)
#undef MSR2_HALF
#define MSR2_GO(KK, C, A, E) return launch_tile<KK, C, A, E>(tile, p, X, Xact, W, bias, res, Y, Yact, grid, s)
    if (K == 3 && epi_s == 0) {
        if (CC == 8) { if (act_mode) MSR2_GO(3, 8, 1, 0); else MSR2_GO(3, 8, 0, 0); }
        if (CC == 16) { if (act_mode) MSR2_GO(3, 16, 1, 0); else MSR2_GO(3, 16, 0, 0); }
    }
    if (K == 3 && CC == 8 && act_mode == 0) {
        if (epi_s == 2) MSR2_GO(3, 8, 0, 2);
        if (epi_s == 8) MSR2_GO(3, 8, 0, 8);
    }
    if (K == 1 && CC == 32 && epi_s == 0 && act_mode == 0) MSR2_GO(1, 32, 0, 0);
    if (K == 5 && CC == 16 && epi_s == 0) { if (act_mode) MSR2_GO(5, 16, 1, 0); else MSR2_GO(5, 16, 0, 0); }
#undef MSR2_GO
    return MS_ERR_UNSUPPORTED;
}
