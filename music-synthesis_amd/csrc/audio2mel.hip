// Audio2Mel (reference feature/feature.py:39-59): hann-windowed STFT frame -> magnitude ->
// mel filterbank -> log10(clamp(., 1e-5)).  One workgroup per (batch, frame): the frame is read
// once from HBM (coalesced, contiguous audio samples), windowed into LDS in bit-reversed order,
// transformed by an in-LDS radix-2 FFT, and the mel projection is taken from the LDS magnitudes.
#include "ms_common.h"

namespace {

__global__ __launch_bounds__(256) void k_audio2mel(const float* __restrict__ audio, int N,
                                                  const float* __restrict__ window, int n_fft,
                                                  int log2n, int hop, int frames,
                                                  const float* __restrict__ basis, int n_mel,
                                                  float* __restrict__ out) {
    extern __shared__ float smem[];
    float* re = smem;
    float* im = smem + n_fft;
    const int fr = blockIdx.x, b = blockIdx.y;
    const float* a = audio + (size_t)b * N;
    for (int i = threadIdx.x; i < n_fft; i += 256) {
        const int s = fr * hop + i;
        const float v = (s < N ? a[s] : 0.f) * window[i];  // right zero-padding, feature.py:44-45
        const int r = (int)(__brev((unsigned)i) >> (32 - log2n));
        re[r] = v;
        im[r] = 0.f;
    }
    __syncthreads();
    for (int st = 1; st <= log2n; ++st) {
        const int m = 1 << st, half = m >> 1;
        for (int j = threadIdx.x; j < (n_fft >> 1); j += 256) {
            const int grp = j / half, pos = j - grp * half;
            const int i0 = grp * m + pos, i1 = i0 + half;
            float sn, cs;
            sincospif(2.0f * (float)pos / (float)m, &sn, &cs);  // w = exp(-2 pi i pos / m)
            const float xr = re[i1], xi = im[i1];
            const float tr = xr * cs + xi * sn;
            const float ti = xi * cs - xr * sn;
            const float ur = re[i0], ui = im[i0];
            re[i0] = ur + tr; im[i0] = ui + ti;
            re[i1] = ur - tr; im[i1] = ui - ti;
        }
        __syncthreads();
    }
    const int nb = (n_fft >> 1) + 1;
    // magnitudes into re[0..nb) (bins only read their own slot, so in place is safe)
    for (int j = threadIdx.x; j < nb; j += 256) {
        const float r = re[j], q = im[j];
        re[j] = sqrtf(r * r + q * q);
    }
    __syncthreads();
    for (int mi = threadIdx.x; mi < n_mel; mi += 256) {
        const float* br = basis + (size_t)mi * nb;
        float acc = 0.f;
        for (int j = 0; j < nb; ++j) acc = fmaf(br[j], re[j], acc);
        out[((size_t)b * n_mel + mi) * frames + fr] = log10f(fmaxf(acc, 1e-5f));
    }
}

}  // namespace

extern "C" {

int ms_audio2mel_frames(int32_t N, int32_t n_fft, int32_t hop) {
    if (N <= 0 || n_fft <= 0 || hop <= 0) return 0;
    const int total = N + (n_fft - hop) / 2;
    return total < n_fft ? 0 : (total - n_fft) / hop + 1;
}

int ms_audio2mel_fwd(const float* audio, int32_t B, int32_t N, const float* window, int32_t n_fft,
                     int32_t hop, const float* mel_basis, int32_t n_mel, float* out,
                     ms_stream_t stream) {
    if (!audio || !window || !mel_basis || !out || B <= 0 || N <= 0 || n_mel <= 0 || hop <= 0)
        return MS_ERR_INVALID_ARG;
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    if ((1 << log2n) != n_fft || n_fft < 64 || n_fft > 4096) return MS_ERR_UNSUPPORTED;
    const int frames = ms_audio2mel_frames(N, n_fft, hop);
    if (frames <= 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_audio2mel, dim3(frames, B), dim3(256), (size_t)2 * n_fft * sizeof(float),
                       (hipStream_t)stream, audio, N, window, n_fft, log2n, hop, frames, mel_basis,
                       n_mel, out);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// audio() front-end of the dataset pass (feature/feature.py:64-71): band-limited sinc resampling
// (librosa.resample's default 'kaiser_best' = resampy's interpolated Kaiser-windowed sinc table) and peak
// normalisation x / max|x| * 0.95.  One thread per output sample walks the two wings of the filter exactly as
// resampy's resample_f does (table lookup + linear interpolation between table entries); the half-window table
// and its first differences come from the host (numpy, like resampy itself builds them).
namespace {

__global__ __launch_bounds__(256) void k_resample_sinc(const float* __restrict__ x, int rows, int n_in,
                                                      float* __restrict__ y, int n_out, double ratio,
                                                      const float* __restrict__ win,
                                                      const float* __restrict__ delta, int nwin, int num_table) {
    const double scale = ratio < 1.0 ? ratio : 1.0;
    const int index_step = (int)(scale * num_table);
    const long long total = (long long)rows * n_out;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / n_out), t = (int)(i - (long long)r * n_out);
        const float* xr = x + (size_t)r * n_in;
        const double time_register = (double)t / ratio;
        const int n = (int)time_register;
        float acc = 0.f;
        // left wing
        double frac = scale * (time_register - n);
        double index_frac = frac * num_table;
        int offset = (int)index_frac;
        float eta = (float)(index_frac - offset);
        int i_max = (nwin - offset) / index_step;
        if (i_max > n + 1) i_max = n + 1;
        for (int k = 0; k < i_max; ++k) {
            const int w = offset + k * index_step;
            acc += (win[w] + eta * delta[w]) * xr[n - k];
        }
        // right wing
        frac = scale - frac;
        index_frac = frac * num_table;
        offset = (int)index_frac;
        eta = (float)(index_frac - offset);
        int k_max = (nwin - offset) / index_step;
        if (k_max > n_in - n - 1) k_max = n_in - n - 1;
        for (int k = 0; k < k_max; ++k) {
            const int w = offset + k * index_step;
            acc += (win[w] + eta * delta[w]) * xr[n + k + 1];
        }
        y[i] = acc;
    }
}

__global__ __launch_bounds__(256) void k_row_absmax(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float red[4];
    const float* xr = x + (size_t)blockIdx.x * n;
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(xr[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void k_row_scale(float* __restrict__ x, int n, const float* __restrict__ amax,
                                                  float scale) {
    const float m = amax[blockIdx.y];
    const float f = m > 1.17549435e-38f ? scale / m : scale;      // librosa.util.normalize: rows below tiny stay as they are
    float* xr = x + (size_t)blockIdx.y * n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) xr[i] *= f;
}

}  // namespace

extern "C" int ms_resample_sinc_fwd(const float* x, int32_t rows, int32_t n_in, float* y, int32_t n_out, double ratio,
                                    const float* interp_win, const float* interp_delta, int32_t nwin,
                                    int32_t num_table, ms_stream_t stream) {
    if (!x || !y || !interp_win || !interp_delta || rows <= 0 || n_in <= 0 || n_out <= 0 || ratio <= 0.0 || nwin <= 0 ||
        num_table <= 0)
        return MS_ERR_INVALID_ARG;
    const double scale = ratio < 1.0 ? ratio : 1.0;
    if ((int)(scale * num_table) < 1) return MS_ERR_UNSUPPORTED;
    long long total = (long long)rows * n_out;
    unsigned nb = (unsigned)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
    hipLaunchKernelGGL(k_resample_sinc, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, rows, n_in, y, n_out, ratio,
                       interp_win, interp_delta, nwin, num_table);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

extern "C" int ms_peak_normalize(float* x, int32_t rows, int32_t n, float scale, float* workspace /* rows floats */,
                                 ms_stream_t stream) {
    if (!x || !workspace || rows <= 0 || n <= 0) return MS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_row_absmax, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, n, workspace);
    MS_CHECK_LAUNCH();
    unsigned nbx = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipLaunchKernelGGL(k_row_scale, dim3(nbx, rows), dim3(256), 0, (hipStream_t)stream, x, n, workspace, scale);
    MS_CHECK_LAUNCH();
    return MS_OK;
}
