// Data movement of the stage-1 generator's ConvTranspose2d-as-lines form (util/modules.py:HipConvTranspose2d; reference
// featuregenerator/upscale.py:85-99).  Activations travel as lines (B, H, C, W); output row sH*q + phase of a 2-D transposed
// conv (kernel rows 4 / stride 2 or 3 / stride 1, padding 1) is a 1-D transposed conv over the channels of the nt input rows
// q + dy[phase][j] that reach it.  Four pure HBM-stream kernels replace the pad / cat / contiguous / slice-add chains:
//   stack       out[phase][(b, q)][j][c][w] = x[b][q + dy[phase][j]][c][w]   (0 outside the image)      ONE pass, all phases
//   fold        gx[b][i][c][w] = sum_{phase, j : 0 <= i - dy < H} gstack[phase][(b, i - dy)][j][c][w]  (backward of stack;
//               fixed summation order: phase-major, then tap)
//   interleave  out[(b, q)][phase][n] = y[phase][(b, q)][n]      (the phases' output rows into image order)
//   split       the inverse (backward of interleave)
// 16-byte vectors throughout (C*W % 4 == 0), grid-stride, one line block (C*W floats) is contiguous on both sides.
#include "ms_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LinesP {
    int B, H, nph, nt;
    int dy[MS_LINES_MAX_PHASES][MS_LINES_MAX_TAPS];
    long long cw4;          // float4 vectors per line block (C * W / 4)
};

__global__ __launch_bounds__(256) void k_lines_stack(LinesP p, const f32x4* __restrict__ x, f32x4* __restrict__ out, long long total) {
    const long long rows = (long long)p.B * p.H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long blk = i / p.cw4, off = i - blk * p.cw4;
        const int j = (int)(blk % p.nt);
        const long long rr = blk / p.nt;
        const long long r = rr % rows;
        const int ph = (int)(rr / rows);
        const int b = (int)(r / p.H), q = (int)(r - (long long)b * p.H);
        const int src = q + p.dy[ph][j];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (src >= 0 && src < p.H) v = x[((long long)b * p.H + src) * p.cw4 + off];
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_lines_fold(LinesP p, const f32x4* __restrict__ gs, f32x4* __restrict__ gx, long long total) {
    const long long rows = (long long)p.B * p.H;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / p.cw4, off = i - r * p.cw4;
        const int b = (int)(r / p.H), li = (int)(r - (long long)b * p.H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ph = 0; ph < p.nph; ++ph)
            for (int j = 0; j < p.nt; ++j) {
                const int q = li - p.dy[ph][j];
                if (q >= 0 && q < p.H) acc += gs[(((long long)ph * rows + (long long)b * p.H + q) * p.nt + j) * p.cw4 + off];
            }
        gx[i] = acc;
    }
}

// forward: src (nph, rows, n4) -> dst (rows, nph, n4);  inverse: src (rows, nph, n4) -> dst (nph, rows, n4)
__global__ __launch_bounds__(256) void k_lines_interleave(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long long rows, int nph,
                                                         long long n4, int inverse, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long blk = i / n4, off = i - blk * n4;         // i indexes the (nph, rows, n4) side
        const long long r = blk % rows;
        const int ph = (int)(blk / rows);
        const long long k = (r * nph + ph) * n4 + off;            // the (rows, nph, n4) side
        if (!inverse) dst[k] = src[i];
        else dst[i] = src[k];
    }
}

bool lines_geometry(const ms_lines_desc* d, LinesP* p) {
    if (!d || d->B <= 0 || d->H <= 0 || d->C <= 0 || d->W <= 0) return false;
    if (d->phases < 1 || d->phases > MS_LINES_MAX_PHASES || d->taps < 1 || d->taps > MS_LINES_MAX_TAPS) return false;
    if (((long long)d->C * d->W) % 4) return false;
    p->B = d->B; p->H = d->H; p->nph = d->phases; p->nt = d->taps;
    p->cw4 = (long long)d->C * d->W / 4;
    for (int ph = 0; ph < MS_LINES_MAX_PHASES; ++ph)
        for (int j = 0; j < MS_LINES_MAX_TAPS; ++j) {
            const int dy = (ph < d->phases && j < d->taps) ? d->dy[ph * MS_LINES_MAX_TAPS + j] : 0;
            if (dy < -d->H || dy > d->H) return false;
            p->dy[ph][j] = dy;
        }
    return true;
}

unsigned stream_grid(long long total) {
    long long nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;            // 64 workgroups per CU: plenty in flight for an HBM stream
    return (unsigned)(nb < 1 ? 1 : nb);
}

bool aligned16(const void* a, const void* b) { return !((((uintptr_t)a) | ((uintptr_t)b)) & 15); }

}  // namespace

extern "C" {

int ms_lines_stack(const ms_lines_desc* d, const float* x, float* out, ms_stream_t stream) {
    LinesP p;
    if (!x || !out || !aligned16(x, out)) return MS_ERR_INVALID_ARG;
    if (!lines_geometry(d, &p)) return d ? MS_ERR_UNSUPPORTED : MS_ERR_INVALID_ARG;
    const long long total = (long long)p.nph * p.B * p.H * p.nt * p.cw4;
    hipLaunchKernelGGL(k_lines_stack, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, p, (const f32x4*)x, (f32x4*)out, total);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_lines_fold(const ms_lines_desc* d, const float* gstack, float* gx, ms_stream_t stream) {
    LinesP p;
    if (!gstack || !gx || !aligned16(gstack, gx)) return MS_ERR_INVALID_ARG;
    if (!lines_geometry(d, &p)) return d ? MS_ERR_UNSUPPORTED : MS_ERR_INVALID_ARG;
    const long long total = (long long)p.B * p.H * p.cw4;
    hipLaunchKernelGGL(k_lines_fold, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, p, (const f32x4*)gstack, (f32x4*)gx, total);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int ms_lines_interleave(const float* src, float* dst, int64_t rows, int32_t phases, int64_t n, int32_t inverse, ms_stream_t stream) {
    if (!src || !dst || !aligned16(src, dst) || rows <= 0 || phases < 1 || n <= 0 || (n % 4)) return MS_ERR_INVALID_ARG;
    const long long total = (long long)phases * rows * (n / 4);
    hipLaunchKernelGGL(k_lines_interleave, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst,
                       (long long)rows, (int)phases, (long long)(n / 4), inverse ? 1 : 0, total);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

}  // extern "C"
