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
