/*
 * msynth_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the arithmetic on the reference's stage-2 mel->waveform
 * GAN hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker.  The product
 * path (music-synthesis_amd/) never links, imports or calls it.
 *
 * Every routine restates what stock PyTorch computes for the nn.Module /
 * functional call the reference makes (the reference has no native code; its
 * hot path is nn.Conv1d / nn.ConvTranspose1d / F.leaky_relu / F.avg_pool1d /
 * torch.stft calls).  Citations are file:line under /root/reference/.
 *
 * Layout: fp32, contiguous (B, C, L) exactly as PyTorch.  Sums are carried in
 * double and rounded once per output element, so the oracle sits closer to the
 * exact result than either fp32 implementation it is compared with.
 *
 * Parity pin: tests/test_oracle_golden.py checks these routines against
 * fixtures in tests/golden/ that tools/make_golden.py produced by running the
 * imported, unmodified reference modules on CPU in the build container.
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define ORC_ACT_NONE 0
#define ORC_ACT_LRELU 1
#define ORC_ACT_TANH 2
#define ORC_PAD_ZERO 0
#define ORC_PAD_REFLECT 1

static inline float orc_act(double v, int act, float slope) {
    if (act == ORC_ACT_LRELU) return (float)(v > 0.0 ? v : v * (double)slope);
    if (act == ORC_ACT_TANH) return (float)tanh(v);
    return (float)v;
}

/* index into the un-padded input for padded position t (t in [-pad, Lin+pad)) */
static inline int orc_src_index(int t, int Lin, int pad_mode) {
    if (t >= 0 && t < Lin) return t;
    if (pad_mode == ORC_PAD_REFLECT) { /* nn.ReflectionPad1d: generator/full.py:23 */
        if (t < 0) t = -t;
        if (t >= Lin) t = 2 * (Lin - 1) - t;
        return (t >= 0 && t < Lin) ? t : -1;
    }
    return -1;
}

int orc_conv1d_out_len(int Lin, int K, int stride, int pad, int dil) {
    return (Lin + 2 * pad - dil * (K - 1) - 1) / stride + 1;
}

/*
 * nn.Conv1d forward (+ optional fused activation and residual add).
 *   y[b,co,t] = res[b,co,t] + act(bias[co] + sum_{ci in group, k} w[co,ci,k] * xpad[b, g*Cg+ci, t*stride + k*dil - pad])
 * Restates: generator/full.py:24,43 ; util/modules.py:358-365,384-388 ;
 * discriminator/full.py:14-22,36-39.  w is (Cout, Cin/groups, K).
 */
void orc_conv1d_fwd(const float* x, const float* w, const float* bias, const float* res, float* y,
                    int B, int Cin, int Lin, int Cout, int K, int stride, int pad, int dil,
                    int groups, int pad_mode, int act, float slope) {
    const int Lout = orc_conv1d_out_len(Lin, K, stride, pad, dil);
    const int Cg = Cin / groups, Og = Cout / groups;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co) {
            const int g = co / Og;
            for (int t = 0; t < Lout; ++t) {
                double acc = bias ? (double)bias[co] : 0.0;
                for (int ci = 0; ci < Cg; ++ci) {
                    const float* xr = x + ((size_t)b * Cin + (size_t)g * Cg + ci) * Lin;
                    const float* wr = w + ((size_t)co * Cg + ci) * K;
                    for (int k = 0; k < K; ++k) {
                        const int s = orc_src_index(t * stride + k * dil - pad, Lin, pad_mode);
                        if (s >= 0) acc += (double)wr[k] * (double)xr[s];
                    }
                }
                const size_t o = ((size_t)b * Cout + co) * Lout + t;
                float v = orc_act(acc, act, slope);
                if (res) v = res[o] + v;
                y[o] = v;
            }
        }
}

/* gradient through the fused activation, from the saved post-activation output */
void orc_act_bwd(const float* y_act, const float* gy, float* gpre, size_t n, int act, float slope) {
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        float g = gy[i];
        if (act == ORC_ACT_LRELU) g = y_act[i] > 0.f ? g : g * slope;
        else if (act == ORC_ACT_TANH) g = (float)((double)g * (1.0 - (double)y_act[i] * (double)y_act[i]));
        gpre[i] = g;
    }
}

/* d(loss)/dx of the conv above given d(loss)/d(pre-activation output) */
void orc_conv1d_bwd_data(const float* gy, const float* w, float* gx,
                         int B, int Cin, int Lin, int Cout, int K, int stride, int pad, int dil,
                         int groups, int pad_mode) {
    const int Lout = orc_conv1d_out_len(Lin, K, stride, pad, dil);
    const int Cg = Cin / groups, Og = Cout / groups;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < Cin; ++c) {
            const int g = c / Cg, ci = c % Cg;
            double* acc = (double*)calloc((size_t)Lin, sizeof(double));
            for (int oc = 0; oc < Og; ++oc) {
                const int co = g * Og + oc;
                const float* gr = gy + ((size_t)b * Cout + co) * Lout;
                const float* wr = w + ((size_t)co * Cg + ci) * K;
                for (int t = 0; t < Lout; ++t)
                    for (int k = 0; k < K; ++k) {
                        const int s = orc_src_index(t * stride + k * dil - pad, Lin, pad_mode);
                        if (s >= 0) acc[s] += (double)wr[k] * (double)gr[t];
                    }
            }
            float* out = gx + ((size_t)b * Cin + c) * Lin;
            for (int s = 0; s < Lin; ++s) out[s] = (float)acc[s];
            free(acc);
        }
}

/* d(loss)/dw and d(loss)/dbias (overwrite, not accumulate) */
void orc_conv1d_bwd_weight(const float* x, const float* gy, float* gw, float* gb,
                           int B, int Cin, int Lin, int Cout, int K, int stride, int pad, int dil,
                           int groups, int pad_mode) {
    const int Lout = orc_conv1d_out_len(Lin, K, stride, pad, dil);
    const int Cg = Cin / groups, Og = Cout / groups;
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; ++co) {
        const int g = co / Og;
        for (int ci = 0; ci < Cg; ++ci)
            for (int k = 0; k < K; ++k) {
                double acc = 0.0;
                for (int b = 0; b < B; ++b) {
                    const float* xr = x + ((size_t)b * Cin + (size_t)g * Cg + ci) * Lin;
                    const float* gr = gy + ((size_t)b * Cout + co) * Lout;
                    for (int t = 0; t < Lout; ++t) {
                        const int s = orc_src_index(t * stride + k * dil - pad, Lin, pad_mode);
                        if (s >= 0) acc += (double)gr[t] * (double)xr[s];
                    }
                }
                gw[((size_t)co * Cg + ci) * K + k] = (float)acc;
            }
        if (gb) {
            double acc = 0.0;
            for (int b = 0; b < B; ++b) {
                const float* gr = gy + ((size_t)b * Cout + co) * Lout;
                for (int t = 0; t < Lout; ++t) acc += (double)gr[t];
            }
            gb[co] = (float)acc;
        }
    }
}

int orc_conv_transpose1d_out_len(int Lin, int K, int stride, int pad) {
    return (Lin - 1) * stride - 2 * pad + K;
}

/*
 * nn.ConvTranspose1d forward (+ fused activation); w is (Cin, Cout, K).
 *   y[b,co,o] = act(bias[co] + sum_{ci,k : o + pad - k = i*stride} w[ci,co,k] * x[b,ci,i])
 * Restates generator/full.py:27,31,35,39.
 */
void orc_conv_transpose1d_fwd(const float* x, const float* w, const float* bias, float* y,
                              int B, int Cin, int Lin, int Cout, int K, int stride, int pad,
                              int act, float slope) {
    const int Lout = orc_conv_transpose1d_out_len(Lin, K, stride, pad);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int o = 0; o < Lout; ++o) {
                double acc = bias ? (double)bias[co] : 0.0;
                for (int k = 0; k < K; ++k) {
                    const int num = o + pad - k;
                    if (num < 0 || num % stride) continue;
                    const int i = num / stride;
                    if (i >= Lin) continue;
                    for (int ci = 0; ci < Cin; ++ci)
                        acc += (double)w[((size_t)ci * Cout + co) * K + k] *
                               (double)x[((size_t)b * Cin + ci) * Lin + i];
                }
                y[((size_t)b * Cout + co) * Lout + o] = orc_act(acc, act, slope);
            }
}

void orc_conv_transpose1d_bwd_data(const float* gy, const float* w, float* gx,
                                   int B, int Cin, int Lin, int Cout, int K, int stride, int pad) {
    const int Lout = orc_conv_transpose1d_out_len(Lin, K, stride, pad);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int ci = 0; ci < Cin; ++ci)
            for (int i = 0; i < Lin; ++i) {
                double acc = 0.0;
                for (int k = 0; k < K; ++k) {
                    const int o = i * stride - pad + k;
                    if (o < 0 || o >= Lout) continue;
                    for (int co = 0; co < Cout; ++co)
                        acc += (double)w[((size_t)ci * Cout + co) * K + k] *
                               (double)gy[((size_t)b * Cout + co) * Lout + o];
                }
                gx[((size_t)b * Cin + ci) * Lin + i] = (float)acc;
            }
}

void orc_conv_transpose1d_bwd_weight(const float* x, const float* gy, float* gw, float* gb,
                                     int B, int Cin, int Lin, int Cout, int K, int stride, int pad) {
    const int Lout = orc_conv_transpose1d_out_len(Lin, K, stride, pad);
#pragma omp parallel for collapse(2) schedule(static)
    for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co)
            for (int k = 0; k < K; ++k) {
                double acc = 0.0;
                for (int b = 0; b < B; ++b)
                    for (int i = 0; i < Lin; ++i) {
                        const int o = i * stride - pad + k;
                        if (o < 0 || o >= Lout) continue;
                        acc += (double)x[((size_t)b * Cin + ci) * Lin + i] *
                               (double)gy[((size_t)b * Cout + co) * Lout + o];
                    }
                gw[((size_t)ci * Cout + co) * K + k] = (float)acc;
            }
    if (gb) {
#pragma omp parallel for schedule(static)
        for (int co = 0; co < Cout; ++co) {
            double acc = 0.0;
            for (int b = 0; b < B; ++b)
                for (int o = 0; o < Lout; ++o) acc += (double)gy[((size_t)b * Cout + co) * Lout + o];
            gb[co] = (float)acc;
        }
    }
}

int orc_avg_pool1d_out_len(int Lin, int k, int s, int p) { return (Lin + 2 * p - k) / s + 1; }

/*
 * F.avg_pool1d(x, kernel_size=4, stride=2, padding=2), count_include_pad=True
 * (zeros are counted: divisor is always k).  discriminator/melgan.py:22.
 */
void orc_avg_pool1d_fwd(const float* x, float* y, int BC, int Lin, int k, int s, int p) {
    const int Lout = orc_avg_pool1d_out_len(Lin, k, s, p);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < BC; ++r)
        for (int o = 0; o < Lout; ++o) {
            double acc = 0.0;
            for (int j = 0; j < k; ++j) {
                const int i = o * s - p + j;
                if (i >= 0 && i < Lin) acc += (double)x[(size_t)r * Lin + i];
            }
            y[(size_t)r * Lout + o] = (float)(acc / (double)k);
        }
}

void orc_avg_pool1d_bwd(const float* gy, float* gx, int BC, int Lin, int k, int s, int p) {
    const int Lout = orc_avg_pool1d_out_len(Lin, k, s, p);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < BC; ++r)
        for (int i = 0; i < Lin; ++i) {
            double acc = 0.0;
            for (int o = 0; o < Lout; ++o) {
                const int j = i - (o * s - p);
                if (j >= 0 && j < k) acc += (double)gy[(size_t)r * Lout + o];
            }
            gx[(size_t)r * Lin + i] = (float)(acc / (double)k);
        }
}

/*
 * hinge_discriminator_loss: mean(relu(1-r) + relu(1+f))   loss/loss.py:17-18
 * Returns the loss; if gr/gf non-NULL writes d loss / d r, d loss / d f scaled by `gscale`.
 */
double orc_hinge_d(const float* r, const float* f, size_t n, float* gr, float* gf, double gscale) {
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const double a = 1.0 - (double)r[i], c = 1.0 + (double)f[i];
        acc += (a > 0 ? a : 0) + (c > 0 ? c : 0);
        if (gr) gr[i] = (float)(a > 0 ? -gscale / (double)n : 0.0);
        if (gf) gf[i] = (float)(c > 0 ? gscale / (double)n : 0.0);
    }
    return acc / (double)n;
}

/* hinge_generator_loss: mean(-j)   loss/loss.py:9-10 */
double orc_hinge_g(const float* f, size_t n, float* gf, double gscale) {
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) {
        acc -= (double)f[i];
        if (gf) gf[i] = (float)(-gscale / (double)n);
    }
    return acc / (double)n;
}

/* F.l1_loss(r, f) = mean(|r - f|); grad wrt f = -sign(r-f) * gscale / n   loss/loss.py:62 */
double orc_l1_mean(const float* r, const float* f, size_t n, float* gf, double gscale) {
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const double d = (double)r[i] - (double)f[i];
        acc += fabs(d);
        if (gf) gf[i] = (float)((d > 0 ? -1.0 : (d < 0 ? 1.0 : 0.0)) * gscale / (double)n);
    }
    return acc / (double)n;
}

/* least-squares variants, loss/loss.py:5-6,13-14 */
double orc_ls_g(const float* j, size_t n) {
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) { const double d = (double)j[i] - 1.0; acc += d * d; }
    return 0.5 * acc / (double)n;
}
double orc_ls_d(const float* r, const float* f, size_t n) {
    double a = 0.0, c = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const double d = (double)r[i] - 1.0; a += d * d; c += (double)f[i] * (double)f[i];
    }
    return 0.5 * (a / (double)n + c / (double)n);
}

/*
 * torch.optim.Adam single step (no weight decay, no amsgrad), as configured at
 * experiment/experiment.py:111-117.  `step` is the 1-based step count.
 */
void orc_adam_step(float* p, const float* g, float* m, float* v, size_t n,
                   double lr, double b1, double b2, double eps, int step) {
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    const double step_size = lr / bc1, bc2s = sqrt(bc2);
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)n; ++i) {
        const double gi = g[i];
        const double mi = b1 * (double)m[i] + (1.0 - b1) * gi;
        const double vi = b2 * (double)v[i] + (1.0 - b2) * gi * gi;
        m[i] = (float)mi; v[i] = (float)vi;
        const double denom = sqrt(vi) / bc2s + eps;
        p[i] = (float)((double)p[i] - step_size * (mi / denom));
    }
}

/* ---- Audio2Mel (feature/feature.py:11-59) ---- */

static double orc_hz_to_mel(double f) { /* Slaney scale, librosa.filters.mel htk=False */
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = 1000.0 / (200.0 / 3.0);
    const double logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
static double orc_mel_to_hz(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = 1000.0 / (200.0 / 3.0);
    const double logstep = log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

/*
 * librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with htk=False, norm='slaney'
 * (third-party, not under /root/reference; called at feature/feature.py:27-29;
 * version unpinned in requirements.txt:3 -- restated from the published
 * algorithm; see DESIGN.md "parity unpinned" note).  out is (n_mels, 1+n_fft/2).
 */
void orc_mel_basis(double sr, int n_fft, int n_mels, double fmin, double fmax, float* out) {
    const int nb = 1 + n_fft / 2;
    if (fmax <= 0) fmax = sr / 2.0;
    double* edges = (double*)malloc(sizeof(double) * (size_t)(n_mels + 2));
    const double m0 = orc_hz_to_mel(fmin), m1 = orc_hz_to_mel(fmax);
    for (int i = 0; i < n_mels + 2; ++i)
        edges[i] = orc_mel_to_hz(m0 + (m1 - m0) * (double)i / (double)(n_mels + 1));
    for (int i = 0; i < n_mels; ++i) {
        const double lo = edges[i], ce = edges[i + 1], hi = edges[i + 2];
        const double enorm = 2.0 / (hi - lo);
        for (int j = 0; j < nb; ++j) {
            const double f = (sr / 2.0) * (double)j / (double)(nb - 1);
            const double lower = (f - lo) / (ce - lo), upper = (hi - f) / (hi - ce);
            double wgt = lower < upper ? lower : upper;
            if (wgt < 0) wgt = 0;
            out[(size_t)i * nb + j] = (float)(wgt * enorm);
        }
    }
    free(edges);
}

/* torch.hann_window(n) (periodic): 0.5 - 0.5 cos(2 pi i / n)   feature/feature.py:26 */
void orc_hann_periodic(int n, float* out) {
    for (int i = 0; i < n; ++i) out[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)n));
}

int orc_audio2mel_frames(int N, int n_fft, int hop) {
    const int p = (n_fft - hop) / 2;
    const int total = N + p;
    return total < n_fft ? 0 : (total - n_fft) / hop + 1;
}

/*
 * Audio2Mel.forward: right-pad (n_fft-hop)//2 zeros; STFT (center=False) with
 * `window`; magnitude; mel_basis @ magnitude; log10(clamp(., 1e-5)).
 * audio is (B, N); out is (B, n_mel, frames).  feature/feature.py:44-58.
 * Direct DFT in double (the oracle favours obviousness over speed).
 */
void orc_audio2mel(const float* audio, int B, int N, const float* window, int n_fft, int hop,
                   const float* mel_basis, int n_mel, float* out) {
    const int frames = orc_audio2mel_frames(N, n_fft, hop);
    const int nb = 1 + n_fft / 2;
    double* ctab = (double*)malloc(sizeof(double) * (size_t)n_fft);
    double* stab = (double*)malloc(sizeof(double) * (size_t)n_fft);
    for (int i = 0; i < n_fft; ++i) {
        ctab[i] = cos(2.0 * M_PI * (double)i / (double)n_fft);
        stab[i] = sin(2.0 * M_PI * (double)i / (double)n_fft);
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int fr = 0; fr < frames; ++fr) {
            double* seg = (double*)malloc(sizeof(double) * (size_t)n_fft);
            double* mag = (double*)malloc(sizeof(double) * (size_t)nb);
            for (int i = 0; i < n_fft; ++i) {
                const int s = fr * hop + i;
                seg[i] = (s < N ? (double)audio[(size_t)b * N + s] : 0.0) * (double)window[i];
            }
            for (int j = 0; j < nb; ++j) {
                double re = 0.0, im = 0.0;
                for (int i = 0; i < n_fft; ++i) {
                    const int idx = (int)(((long long)i * j) % n_fft);
                    re += seg[i] * ctab[idx];
                    im -= seg[i] * stab[idx];
                }
                mag[j] = sqrt(re * re + im * im);
            }
            for (int m = 0; m < n_mel; ++m) {
                double acc = 0.0;
                for (int j = 0; j < nb; ++j) acc += (double)mel_basis[(size_t)m * nb + j] * mag[j];
                if (acc < 1e-5) acc = 1e-5;
                out[((size_t)b * n_mel + m) * frames + fr] = (float)log10(acc);
            }
            free(seg); free(mag);
        }
    free(ctab); free(stab);
}
