// extern "C" entry points for the convolution family: argument validation, geometry, and the
// dispatch between the direct (vector FMA) kernels and the f32-MFMA implicit-GEMM kernels.
#include "ms_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include "conv_mfma.h"
#include "gconv_mfma.h"
#include "conv_thin.h"

// Profile session of the calling thread (ms_profile_kernels / ms_profile_take): nothing is recorded outside one.
static thread_local int g_prof_on = 0, g_prof_launches = 0, g_prof_products = 0;
static thread_local double g_prof_us = 0.0;
static thread_local char g_prof_kernel[MS_PROFILE_NAME_MAX] = "";

void ms_note_kernel(int products, const char* fmt, ...) {
    if (!g_prof_on) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_prof_kernel, sizeof(g_prof_kernel), fmt, ap);
    va_end(ap);
    g_prof_products = products;
}

bool ms_prof_on() { return g_prof_on != 0; }

void ms_prof_add(hipEvent_t e0, hipEvent_t e1) {
    float ms = 0.f;
    if (e0 && e1 && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
        g_prof_us += 1e3 * (double)ms;
        ++g_prof_launches;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
}

namespace {

bool make_conv(const ms_conv1d_desc* d, ConvP* p) {
    if (!d) return false;
    if (d->B <= 0 || d->Cin <= 0 || d->Lin <= 0 || d->Cout <= 0 || d->K <= 0 || d->stride <= 0 ||
        d->pad < 0 || d->dil <= 0 || d->groups <= 0)
        return false;
    if (d->Cin % d->groups || d->Cout % d->groups) return false;
    if (d->pad_mode != MS_PAD_ZERO && d->pad_mode != MS_PAD_REFLECT) return false;
    if (d->pad_mode == MS_PAD_REFLECT && d->pad >= d->Lin) return false;  // nn.ReflectionPad1d rule
    if (d->act < MS_ACT_NONE || d->act > MS_ACT_TANH) return false;
    const int eff = d->Lin + 2 * d->pad - d->dil * (d->K - 1) - 1;
    if (eff < 0) return false;
    p->B = d->B; p->Cin = d->Cin; p->Lin = d->Lin; p->Cout = d->Cout; p->K = d->K;
    p->stride = d->stride; p->pad = d->pad; p->dil = d->dil; p->groups = d->groups;
    p->Cg = d->Cin / d->groups; p->Og = d->Cout / d->groups;
    p->Lout = eff / d->stride + 1;
    p->pad_mode = d->pad_mode; p->act = d->act; p->slope = d->slope;
    if (d->in_act != MS_ACT_NONE && d->in_act != MS_ACT_LRELU) return false;
    p->in_act = d->in_act;
    return true;
}

// conv whose backward-data IS the transposed conv: channels swap roles (w layout (Cin_T, Cout_T, K)
// equals the (Cout_conv, Cin_conv, K) layout of that conv).
bool make_convt(const ms_convt1d_desc* d, ConvP* p) {
    if (!d) return false;
    if (d->B <= 0 || d->Cin <= 0 || d->Lin <= 0 || d->Cout <= 0 || d->K <= 0 || d->stride <= 0 ||
        d->pad < 0)
        return false;
    if (d->act < MS_ACT_NONE || d->act > MS_ACT_TANH) return false;
    const int Lout = (d->Lin - 1) * d->stride - 2 * d->pad + d->K;
    if (Lout <= 0) return false;
    p->B = d->B; p->Cin = d->Cout; p->Lin = Lout; p->Cout = d->Cin; p->K = d->K;
    p->stride = d->stride; p->pad = d->pad; p->dil = 1; p->groups = 1;
    p->Cg = d->Cout; p->Og = d->Cin;
    p->Lout = d->Lin;
    p->pad_mode = MS_PAD_ZERO; p->act = d->act; p->slope = d->slope;
    if (d->in_act != MS_ACT_NONE && d->in_act != MS_ACT_LRELU) return false;
    p->in_act = d->in_act;   // activation in front of the TRANSPOSED conv (on its input x)
    // consistency: the conv's own output length for Lin=Lout_T must give back Lin_T
    const int chk = (Lout + 2 * d->pad - (d->K - 1) - 1) / d->stride + 1;
    return chk == d->Lin;
}

}  // namespace

extern "C" {

int ms_version(void) { return MSYNTH_VERSION; }

void ms_profile_kernels(int on) {
    g_prof_on = on ? 1 : 0;
    g_prof_us = 0.0;
    g_prof_launches = 0;
    g_prof_products = 0;
    g_prof_kernel[0] = 0;
}

int ms_profile_take(ms_profile_record* out) {
    if (!out) return MS_ERR_INVALID_ARG;
    out->kernels = g_prof_launches;
    out->products = g_prof_products;
    out->device_us = g_prof_us;
    memcpy(out->kernel, g_prof_kernel, sizeof(out->kernel));
    g_prof_us = 0.0;
    g_prof_launches = 0;
    g_prof_products = 0;
    g_prof_kernel[0] = 0;
    return MS_OK;
}

// Debug aid (tests/conftest.py): native frames on a fatal signal.  A hipGraphLaunch / kernel-launch fault inside the runtime
// leaves Python's faulthandler with Python frames only (profiles/r03_forked_replay_segfault.txt); this handler writes the
// native backtrace of the faulting thread to stderr (async-signal-safe calls only), then re-raises with the default action.
static struct sigaction g_prev_action[32];

static void ms_crash_handler(int sig) {
    static const char head[] = "\n[msynth] fatal signal: native backtrace of the faulting thread\n";
    (void)!write(2, head, sizeof(head) - 1);
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    // hand on to whoever was installed before (Python's faulthandler prints the interpreter frames and re-raises)
    if (sig > 0 && sig < 32) sigaction(sig, &g_prev_action[sig], nullptr);
    else signal(sig, SIG_DFL);
    raise(sig);
}

int ms_debug_install_crash_handler(void) {
    void* warm[4];
    (void)backtrace(warm, 4);              // (loads libgcc's unwinder now, not inside the handler)
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = ms_crash_handler;
    sa.sa_flags = SA_NODEFER | SA_RESETHAND;
    sigemptyset(&sa.sa_mask);
    int rc = 0;
    for (int sig : {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL}) rc |= sigaction(sig, &sa, &g_prev_action[sig]);
    return rc == 0 ? MS_OK : MS_ERR_INVALID_ARG;
}

const char* ms_status_string(int status) {
    switch (status) {
        case MS_OK: return "ok";
        case MS_ERR_INVALID_ARG: return "invalid argument (null pointer, bad size or inconsistent shape)";
        case MS_ERR_UNSUPPORTED: return "configuration not supported by the MI355X kernels";
        case MS_ERR_WORKSPACE: return "workspace missing or too small";
        case MS_ERR_LAUNCH: return "HIP kernel launch failed";
        case MS_ERR_COMM: return "RCCL unavailable or an RCCL call failed (ms_comm_last_error)";
        default: return "unknown status";
    }
}

// ---- rows whose length is not a multiple of 4 on the 16-byte kernels
// The D 1024 -> 1024 k5 conv at the pooled scales runs on rows of 17 / 9 samples: no row is 16-byte aligned, so the
// split-bf16 row kernel takes its dword loader (4x the load instructions, four-wave form only): 58-74 TFLOP/s against
// 160-170 for the same layer at L = 32.  For such layers the operands are copied into rows padded with zeros to
// L' = 4 ceil(L / 4) (a few MB: microseconds), the conv runs on the padded problem through the aligned paired kernel,
// and the valid columns are copied back.  Zero columns behind a row are what the conv's zero padding reads anyway, so
// the valid outputs are unchanged; the padded outputs are discarded.  Measured (tools/scratch/microbench_pad4.py, B = 64):
// L = 17 forward 197 -> 150 us, backward 232 -> 164 us; L = 33: 267 -> 212 / 379 -> 241 us.  With fewer than ~1000
// columns (B = 32 at L = 17, any batch at L = 9) the extra columns cost more than the loader saves: not padded.
namespace {

__global__ __launch_bounds__(256) void k_pad_rows(const float* __restrict__ src, float* __restrict__ dst,
                                                 size_t n_dst, int L, int Lp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_dst) return;
    const size_t row = i / (unsigned)Lp;
    const int t = (int)(i - row * (unsigned)Lp);
    dst[i] = t < L ? src[row * (unsigned)L + t] : 0.f;
}

__global__ __launch_bounds__(256) void k_unpad_rows(const float* __restrict__ src, const float* __restrict__ add,
                                                   float* __restrict__ dst, size_t n_dst, int L, int Lp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_dst) return;
    const size_t row = i / (unsigned)L;
    const int t = (int)(i - row * (unsigned)L);
    const float v = src[row * (unsigned)Lp + t];
    dst[i] = add ? v + add[i] : v;
}

bool pad4_applicable(const ConvP& p) {
    return p.groups == 1 && p.stride == 1 && p.Lout == p.Lin && (p.Lin & 3) != 0 && p.Lin >= 5 && p.Lin <= 256 &&
           (long long)p.B * p.Lin >= 1000 &&
           p.K == 5 && p.Cin >= 256 && p.Cout >= 256 && p.Cin % 16 == 0 && p.Cout % 16 == 0 &&
           p.pad_mode == MS_PAD_ZERO && !p.in_act;
}

ConvP pad4_conv(const ConvP& p) {
    ConvP q = p;
    q.Lin = q.Lout = (p.Lin + 3) & ~3;
    return q;
}

size_t pad4_bytes(const ConvP& q, int channels) { return (size_t)q.B * channels * q.Lin * sizeof(float); }

int pad_rows(const float* src, float* dst, size_t rows, int L, int Lp, hipStream_t s) {
    const size_t n = rows * (size_t)Lp;
    hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n, L, Lp);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

int unpad_rows(const float* src, const float* add, float* dst, size_t rows, int L, int Lp, hipStream_t s) {
    const size_t n = rows * (size_t)L;
    hipLaunchKernelGGL(k_unpad_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, add, dst, n, L, Lp);
    MS_CHECK_LAUNCH();
    return MS_OK;
}

}  // namespace

int ms_conv1d_out_len(const ms_conv1d_desc* d) {
    ConvP p;
    return make_conv(d, &p) ? p.Lout : MS_ERR_INVALID_ARG;
}

// a single tensor as a table of one part (the 4 x 4 group kernels of gconv4.hip take 1 .. 3 parts)
static ms_conv1d_parts one_part(const ConvP& p) {
    ms_conv1d_parts q{};
    q.count = 1;
    q.B[0] = p.B; q.Lin[0] = p.Lin;
    return q;
}

int ms_conv1d_fwd(const ms_conv1d_desc* d, const float* x, const float* w, const float* bias,
                  const float* residual, float* y, float* y_act, void* workspace,
                  size_t workspace_bytes, ms_stream_t stream) {
    ConvP p;
    if (!make_conv(d, &p) || !x || !w || !y) return MS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    // activation in front of the conv: applied to x on load (operand modifier "LeakyReLU of the value")
    const float* xa = p.in_act ? x : nullptr;
    const int xk = p.in_act ? MS_MOD_LRELU_FWD : 0;
    if (mst_fwd_short_applicable(p) && !y_act && !residual)   // judge conv: a 12-MFLOP reduction, not a GEMM
        return mst_conv1d_fwd(p, x, w, bias, residual, y, s);
    if (!y_act && !residual && mss_conv_applicable(p))        // a few dozen columns in the whole batch: a weight stream
        return mss_conv_fwd(p, x, w, bias, y, s);
    if (msm_fwd_applicable(p) && pad4_applicable(p) && !residual && !y_act) {
        const ConvP q = pad4_conv(p);
        const size_t xb = pad4_bytes(q, p.Cin), yb = pad4_bytes(q, p.Cout);
        if (workspace && workspace_bytes >= xb + yb + msm_fwd_ws(q) && (((uintptr_t)workspace) & 15) == 0) {
            float* xp = (float*)workspace;
            float* yp = (float*)((char*)workspace + xb);
            int rc = pad_rows(x, xp, (size_t)p.B * p.Cin, p.Lin, q.Lin, s);
            if (rc != MS_OK) return rc;
            rc = msm_conv1d_fwd(q, xp, nullptr, 0, w, bias, nullptr, yp, nullptr, (char*)workspace + xb + yb,
                                workspace_bytes - xb - yb, s);
            if (rc != MS_OK) return rc;
            return unpad_rows(yp, nullptr, y, (size_t)p.B * p.Cout, p.Lin, q.Lin, s);
        }
    }
    if (msm_fwd_applicable(p))
        return msm_conv1d_fwd(p, x, xa, xk, w, bias, residual, y, y_act, workspace, workspace_bytes, s);
    if (!residual && !y_act && !p.in_act) {
        ms_conv1d_parts q = one_part(p);
        q.x[0] = x; q.y[0] = y;
        if (msg4_parts_applicable(p, &q)) return msg4_parts_fwd(p, &q, w, bias, s);
    }
    if (msg3_fwd_applicable(p) && !residual && !y_act && !p.in_act) return msg3_conv1d_fwd(p, x, w, bias, y, s);
    if (msg_fwd_applicable(p) && !residual && !y_act && !p.in_act) return msg_conv1d_fwd(p, x, w, bias, y, s);
    if (mst_fwd_applicable(p) && !y_act) return mst_conv1d_fwd(p, x, w, bias, residual, y, s);
    return msk_conv1d_fwd_direct(p, x, xa, xk, w, bias, residual, y, y_act, s);
}

int ms_conv1d_bwd_data(const ms_conv1d_desc* d, const float* gy, const float* y_act,
                       const float* w, const float* gx_add, float* gx, void* workspace,
                       size_t workspace_bytes, ms_stream_t stream) {
    ConvP p;
    if (!make_conv(d, &p) || !gy || !w || !gx) return MS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    // reflection padding: the zero-padded backward gives the gradient of the in-range taps; the
    // taps that read mirrored samples are folded back onto their sources by a small edge kernel
    const bool reflect = p.pad_mode == MS_PAD_REFLECT;
    if (reflect && (p.stride != 1 || p.groups != 1)) return MS_ERR_UNSUPPORTED;
    p.pad_mode = MS_PAD_ZERO;
    int rc;
    bool padded = false;
    if (msm_bwd_data_applicable(p) && pad4_applicable(p)) {
        const ConvP q = pad4_conv(p);
        const size_t gb = pad4_bytes(q, p.Cout), xb = pad4_bytes(q, p.Cin);
        const size_t need = gb * (y_act ? 2 : 1) + xb + msm_bwd_data_ws(q);
        if (workspace && workspace_bytes >= need && (((uintptr_t)workspace) & 15) == 0) {
            char* wsp = (char*)workspace;
            float* gp = (float*)wsp; wsp += gb;
            float* ap = nullptr;
            if (y_act) { ap = (float*)wsp; wsp += gb; }
            float* xp = (float*)wsp; wsp += xb;
            rc = pad_rows(gy, gp, (size_t)p.B * p.Cout, p.Lin, q.Lin, s);
            if (rc == MS_OK && y_act) rc = pad_rows(y_act, ap, (size_t)p.B * p.Cout, p.Lin, q.Lin, s);
            if (rc == MS_OK)
                rc = msm_conv1d_bwd_data(q, gp, ap, w, nullptr, xp, wsp, workspace_bytes - (size_t)(wsp - (char*)workspace), s);
            if (rc == MS_OK) rc = unpad_rows(xp, gx_add, gx, (size_t)p.B * p.Cin, p.Lin, q.Lin, s);
            padded = true;
        }
    }
    if (padded) {
    } else if (msm_bwd_data_applicable(p))
        rc = msm_conv1d_bwd_data(p, gy, y_act, w, gx_add, gx, workspace, workspace_bytes, s);
    else if (ms_conv1d_parts q4 = one_part(p); msg4_parts_applicable(p, &q4)) {
        q4.gy[0] = gy; q4.y_act[0] = y_act; q4.gx_add[0] = gx_add; q4.gx[0] = gx;
        rc = msg4_parts_bwd_data(p, &q4, w, s);
    } else if (msg3_bwd_data_applicable(p))
        rc = msg3_conv1d_bwd_data(p, gy, y_act, w, gx_add, gx, s);
    else if (msg_bwd_data_applicable(p))
        rc = msg_conv1d_bwd_data(p, gy, y_act, w, gx_add, gx, s);
    else if (mst_bwd_data_applicable(p) && !reflect)
        rc = mst_conv1d_bwd_data(p, gy, y_act, w, gx_add, gx, s);
    else
        rc = msk_conv1d_bwd_data_direct(p, gy, y_act, w, nullptr, MS_ACT_NONE, gx_add, gx, s);
    if (rc != MS_OK || !reflect) return rc;
    return msk_reflect_fold_bwd(p, gy, y_act, w, gx, s);
}

int ms_conv1d_bwd_weight(const ms_conv1d_desc* d, const float* x, const float* gy,
                         const float* y_act, float* gw, float* gb, float beta, void* workspace,
                         size_t workspace_bytes, ms_stream_t stream) {
    ConvP p;
    if (!make_conv(d, &p) || !x || !gy || !gw) return MS_ERR_INVALID_ARG;
    if (beta != 0.f && beta != 1.f) return MS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const float* xa = p.in_act ? x : nullptr;
    const int xk = p.in_act ? MS_MOD_LRELU_FWD : 0;
    if (mst_bwd_weight_applicable(p))   // one-channel side: HBM-bound stream kernels
        return mst_conv1d_bwd_weight(p, x, gy, y_act, gw, gb, beta, workspace, workspace_bytes, s);
    if (msws_applicable(p) && workspace && workspace_bytes >= msws_ws(p)) {     // reflection-padded conv on short rows
        const int rc = msws_bwd_weight(p, x, gy, y_act, gw, gb, beta, workspace, workspace_bytes, s);
        if (rc != MS_ERR_UNSUPPORTED) return rc;
    }
    if (msw32_applicable(p)) {          // 32 -> 32 k3 atoms: HBM-bound, per-wave units
        const int rc = msw32_bwd_weight(p, x, gy, y_act, gw, gb, beta, workspace, workspace_bytes, s);
        if (rc != MS_ERR_UNSUPPORTED) return rc;
    }
    if (msw5_applicable(p) && workspace && workspace_bytes >= msw5_ws(p))    // 1024 -> 1024 k5 on short rows
        return msw5_bwd_weight(p, x, gy, y_act, gw, gb, beta, workspace, workspace_bytes, s);
    if (msw_bwd_weight_applicable(p)) { // dense stride-1 convs: row-tile MFMA form
        const int rc = msw_conv1d_bwd_weight(p, x, gy, y_act, gw, gb, beta, workspace, workspace_bytes, s);
        if (rc != MS_ERR_UNSUPPORTED) return rc;     // (unaligned operands: the im2col form below)
    }
    if (msm_bwd_weight_applicable(p))
        return msm_conv1d_bwd_weight(p, x, xa, xk, gy, y_act, p.act, gw, gb, beta, workspace,
                                     workspace_bytes, s);
    if (!p.in_act) {
        ms_conv1d_parts q = one_part(p);
        q.x[0] = x; q.gy[0] = gy; q.y_act[0] = y_act;
        if (msg4_parts_applicable(p, &q)) return msg4_parts_bwd_weight(p, &q, gw, gb, beta, s);
    }
    if (msg3_bwd_weight_applicable(p) && !p.in_act)
        return msg3_conv1d_bwd_weight(p, x, gy, y_act, gw, gb, beta, workspace, workspace_bytes, s);
    if (msg_bwd_weight_applicable(p) && !p.in_act)
        return msg_conv1d_bwd_weight(p, x, gy, y_act, gw, gb, beta, workspace, workspace_bytes, s);
    return msk_conv1d_bwd_weight_direct(p, x, xa, xk, gy, y_act, p.act, gw, gb, beta, workspace,
                                        workspace_bytes, s);
}

// ---- one layer over several inputs (ms_conv1d_parts): a parts kernel where one applies, else part by part
namespace {

bool parts_ok(const ms_conv1d_desc* d, const ms_conv1d_parts* parts, ConvP* c) {
    if (!d || !parts || parts->count < 1 || parts->count > MS_CONV_PARTS_MAX) return false;
    ms_conv1d_desc d0 = *d;
    for (int i = 0; i < parts->count; ++i) {
        d0.B = parts->B[i]; d0.Lin = parts->Lin[i];
        ConvP p;
        if (!make_conv(&d0, &p)) return false;
        if (i == 0) *c = p;
    }
    return true;
}

ms_conv1d_desc part_desc(const ms_conv1d_desc* d, const ms_conv1d_parts* parts, int i) {
    ms_conv1d_desc di = *d;
    di.B = parts->B[i]; di.Lin = parts->Lin[i];
    return di;
}

// 1: a parts kernel takes the call; 0: part by part
int parts_kernel(const ConvP& c, const ms_conv1d_parts* parts, int which, bool with_image) {
    if (parts->count < 2 || c.in_act) return 0;
    if (msd_parts_applicable(c, parts, which) || msg4_parts_applicable(c, parts)) return 1;
    if (which == 0) return (with_image && ms5_parts_applicable(c, parts, false)) || msg3_parts_fwd_applicable(c, parts);
    if (which == 1) return (with_image && ms5_parts_applicable(c, parts, true)) || msg3_parts_bwd_data_applicable(c, parts);
    return msw5_parts_applicable(c, parts) || msg3_parts_bwd_weight_applicable(c, parts);
}

}  // namespace

int ms_conv1d_parts_launches(const ms_conv1d_desc* d, const ms_conv1d_parts* parts, int which, int with_image) {
    ConvP c;
    if (!parts_ok(d, parts, &c) || which < 0 || which > 2) return MS_ERR_INVALID_ARG;
    return parts_kernel(c, parts, which, with_image != 0) ? 1 : parts->count;
}

size_t ms_conv1d_parts_workspace_bytes(const ms_conv1d_desc* d, const ms_conv1d_parts* parts, int which, int with_image) {
    ConvP c;
    if (!parts_ok(d, parts, &c) || which < 0 || which > 2) return 0;
    if (parts_kernel(c, parts, which, with_image != 0)) {
        if (msg4_parts_applicable(c, parts)) return 0;
        if (which != 2) {
            if (msd_parts_applicable(c, parts, which)) return 0;
            return (with_image && ms5_parts_applicable(c, parts, which == 1)) ? ms5_parts_ws(c, parts, which == 1) : 0;
        }
        if (msd_parts_applicable(c, parts, which)) return msd_parts_bwd_weight_ws(c, parts);
        return msw5_parts_applicable(c, parts) ? msw5_parts_ws(c, parts) : msg3_parts_bwd_weight_ws(c, parts);
    }
    size_t n = 0;
    for (int i = 0; i < parts->count; ++i) {
        const ms_conv1d_desc di = part_desc(d, parts, i);
        size_t m = ms_conv1d_workspace_bytes(&di, which);
        if (with_image && which < 2 && ms_conv1d_img_bytes(&di)) m = ms_conv1d_img_workspace_bytes(&di, which);
        if (m > n) n = m;
    }
    return n;
}

int ms_conv1d_parts_fwd(const ms_conv1d_desc* d, const ms_conv1d_parts* parts, const float* w, const float* bias,
                        const void* image, void* workspace, size_t workspace_bytes, ms_stream_t stream) {
    ConvP c;
    if (!parts_ok(d, parts, &c) || (!w && !image)) return MS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (parts->count >= 2 && !c.in_act) {
        if (w && msd_parts_applicable(c, parts, 0)) {
            const int rc = msd_parts_fwd(c, parts, w, bias, s);
            if (rc != MS_ERR_UNSUPPORTED) return rc;
        }
        if (w && msg4_parts_applicable(c, parts)) return msg4_parts_fwd(c, parts, w, bias, s);
        if (image && ms5_parts_applicable(c, parts, false))
            return ms5_parts_fwd(c, parts, image, bias, workspace, workspace_bytes, s);
        if (w && msg3_parts_fwd_applicable(c, parts)) return msg3_parts_fwd(c, parts, w, bias, s);
    }
    for (int i = 0; i < parts->count; ++i) {
        const ms_conv1d_desc di = part_desc(d, parts, i);
        int rc;
        if (image && ms_conv1d_img_bytes(&di))
            rc = ms_conv1d_img_fwd(&di, parts->x[i], image, bias, parts->y[i], workspace, workspace_bytes, stream);
        else if (w)
            rc = ms_conv1d_fwd(&di, parts->x[i], w, bias, nullptr, parts->y[i], nullptr, workspace, workspace_bytes, stream);
        else
            rc = MS_ERR_UNSUPPORTED;
        if (rc != MS_OK) return rc;
    }
    return MS_OK;
}

int ms_conv1d_parts_bwd_data(const ms_conv1d_desc* d, const ms_conv1d_parts* parts, const float* w, const void* image_bwd,
                             void* workspace, size_t workspace_bytes, ms_stream_t stream) {
    ConvP c;
    if (!parts_ok(d, parts, &c) || (!w && !image_bwd)) return MS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (parts->count >= 2 && !c.in_act) {
        if (w && msd_parts_applicable(c, parts, 1)) {
            const int rc = msd_parts_bwd_data(c, parts, w, s);
            if (rc != MS_ERR_UNSUPPORTED) return rc;
        }
        if (w && msg4_parts_applicable(c, parts)) return msg4_parts_bwd_data(c, parts, w, s);
        if (image_bwd && ms5_parts_applicable(c, parts, true))
            return ms5_parts_bwd_data(c, parts, image_bwd, workspace, workspace_bytes, s);
        if (w && msg3_parts_bwd_data_applicable(c, parts)) return msg3_parts_bwd_data(c, parts, w, s);
    }
    for (int i = 0; i < parts->count; ++i) {
        const ms_conv1d_desc di = part_desc(d, parts, i);
        const float* ya = d->act == MS_ACT_NONE ? nullptr : parts->y_act[i];
        int rc;
        if (image_bwd && ms_conv1d_img_bytes(&di))
            rc = ms_conv1d_img_bwd_data(&di, parts->gy[i], ya, image_bwd, parts->gx_add[i], parts->gx[i], workspace,
                                        workspace_bytes, stream);
        else if (w)
            rc = ms_conv1d_bwd_data(&di, parts->gy[i], ya, w, parts->gx_add[i], parts->gx[i], workspace, workspace_bytes, stream);
        else
            rc = MS_ERR_UNSUPPORTED;
        if (rc != MS_OK) return rc;
    }
    return MS_OK;
}

int ms_conv1d_parts_bwd_weight(const ms_conv1d_desc* d, const ms_conv1d_parts* parts, float* gw, float* gb, float beta,
                               void* workspace, size_t workspace_bytes, ms_stream_t stream) {
    ConvP c;
    if (!parts_ok(d, parts, &c) || !gw || (beta != 0.f && beta != 1.f)) return MS_ERR_INVALID_ARG;
    if (parts->count >= 2 && !c.in_act && msd_parts_applicable(c, parts, 2)) {
        const int rc = msd_parts_bwd_weight(c, parts, gw, gb, beta, workspace, workspace_bytes, (hipStream_t)stream);
        if (rc != MS_ERR_UNSUPPORTED) return rc;
    }
    if (parts->count >= 2 && !c.in_act && msg4_parts_applicable(c, parts))
        return msg4_parts_bwd_weight(c, parts, gw, gb, beta, (hipStream_t)stream);
    if (parts->count >= 2 && !c.in_act && msw5_parts_applicable(c, parts))
        return msw5_parts_bwd_weight(c, parts, gw, gb, beta, workspace, workspace_bytes, (hipStream_t)stream);
    if (parts->count >= 2 && !c.in_act && msg3_parts_bwd_weight_applicable(c, parts))
        return msg3_parts_bwd_weight(c, parts, gw, gb, beta, workspace, workspace_bytes, (hipStream_t)stream);
    for (int i = 0; i < parts->count; ++i) {             // the parts' gradients accumulate in order on the one stream
        const ms_conv1d_desc di = part_desc(d, parts, i);
        const float* ya = d->act == MS_ACT_NONE ? nullptr : parts->y_act[i];
        const int rc = ms_conv1d_bwd_weight(&di, parts->x[i], parts->gy[i], ya, gw, gb, i == 0 ? beta : 1.f, workspace,
                                            workspace_bytes, stream);
        if (rc != MS_OK) return rc;
    }
    return MS_OK;
}

static int multi_convs(const ms_wgrad_multi_desc* d, ConvP* cs) {
    if (!d || d->count <= 0 || d->count > MS_WGRAD_MULTI_MAX) return 0;
    for (int i = 0; i < d->count; ++i)
        if (!make_conv(&d->conv[i], &cs[i])) return 0;
    return d->count;
}

size_t ms_conv1d_bwd_weight_multi_workspace_bytes(const ms_wgrad_multi_desc* d) {
    ConvP cs[MS_WGRAD_MULTI_MAX];
    const int n = multi_convs(d, cs);
    size_t need = n ? msw_multi_ws(cs, n) : 0;
    if (n && msw32_multi_ws(cs, n) > need) need = msw32_multi_ws(cs, n);
    for (int i = 0; i < n; ++i) {
        const size_t one = ms_conv1d_workspace_bytes(&d->conv[i], 2);
        if (one > need) need = one;
    }
    return need;
}

int ms_conv1d_bwd_weight_multi(const ms_wgrad_multi_desc* d, void* workspace, size_t workspace_bytes,
                               ms_stream_t stream) {
    ConvP cs[MS_WGRAD_MULTI_MAX];
    const int n = multi_convs(d, cs);
    if (!n) return MS_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i)
        if (!d->x[i] || !d->gy[i] || !d->gw[i] || (d->beta[i] != 0.f && d->beta[i] != 1.f)) return MS_ERR_INVALID_ARG;
    // sign words in place of the activations: all entries or none, and only the two batched kernels read them
    int nsig = 0;
    for (int i = 0; i < n; ++i) nsig += d->y_signs[i] ? 1 : 0;
    if (nsig != 0 && nsig != n) return MS_ERR_INVALID_ARG;
    const float* ya[MS_WGRAD_MULTI_MAX];
    for (int i = 0; i < n; ++i) ya[i] = nsig ? reinterpret_cast<const float*>(d->y_signs[i]) : d->y_act[i];
    int rc = msw_conv1d_bwd_weight_multi(cs, n, d->x, d->gy, ya, d->gw, d->gb, d->beta, d->xmax, d->gmax, nsig ? 1 : 0, workspace,
                                         workspace_bytes, (hipStream_t)stream);
    if (rc != MS_ERR_UNSUPPORTED) return rc;
    rc = msw32_bwd_weight_multi(cs, n, d->x, d->gy, ya, d->gw, d->gb, d->beta, nsig ? 1 : 0, workspace, workspace_bytes,
                                (hipStream_t)stream);
    if (rc != MS_ERR_UNSUPPORTED) return rc;
    if (nsig) return MS_ERR_UNSUPPORTED;
    for (int i = 0; i < n; ++i) {       // geometry differs / unaligned operands: entry by entry
        const int r1 = ms_conv1d_bwd_weight(&d->conv[i], d->x[i], d->gy[i], d->y_act[i], d->gw[i], d->gb[i],
                                            d->beta[i], workspace, workspace_bytes, stream);
        if (r1 != MS_OK) return r1;
    }
    return MS_OK;
}

int ms_residual_stack_signs_supported(const ms_stack_desc* d) {
    const char* sw = getenv("MSYNTH_ATOM_SIGNS");                // tuning / test switch (0: the fp32 activations are saved)
    if (sw && atoi(sw) == 0) return 0;
    if (!d || d->count < 1 || d->count > MS_STACK_MAX || 2 * d->count > MS_WGRAD_MULTI_MAX) return 0;
    ConvP cs[MS_WGRAD_MULTI_MAX];
    int n = 0;
    for (int i = 0; i < d->count; ++i) {
        ms_atom_desc a = {d->B, d->C, d->L, d->dil[i], d->slope};
        if (!ms_residual_atom_sign_words(&a) || !ms_residual_atom_bwd_supported(&a)) return 0;
        // the two weight gradients of the atom: the dilation-1 conv (input t, gradient g masked by u) and the dilated one
        ms_conv1d_desc c1 = {d->B, d->C, d->L, d->C, 3, 1, 1, 1, 1, MS_PAD_ZERO, MS_ACT_LRELU, d->slope, MS_ACT_NONE};
        ms_conv1d_desc c0 = {d->B, d->C, d->L, d->C, 3, 1, d->dil[i], d->dil[i], 1, MS_PAD_ZERO, MS_ACT_LRELU, d->slope, MS_ACT_NONE};
        if (!make_conv(&c1, &cs[n++]) || !make_conv(&c0, &cs[n++])) return 0;
    }
    return (msw_multi_takes_signs(cs, n) || msw32_multi_takes_signs(cs, n)) ? 1 : 0;
}

size_t ms_conv1d_workspace_bytes(const ms_conv1d_desc* d, int which) {
    ConvP p;
    if (!make_conv(d, &p)) return 0;
    if (which == 0) {
        if (!msm_fwd_applicable(p)) return 0;
        size_t n = msm_fwd_ws(p);
        if (pad4_applicable(p)) {         // padded copies of x and y + the padded problem's own workspace
            const ConvP q = pad4_conv(p);
            const size_t m = pad4_bytes(q, p.Cin) + pad4_bytes(q, p.Cout) + msm_fwd_ws(q);
            if (m > n) n = m;
        }
        return n;
    }
    if (which == 1) {
        // (reflection padding runs the zero-padded backward + the edge fold, see ms_conv1d_bwd_data)
        if (p.pad_mode == MS_PAD_REFLECT && p.stride == 1 && p.groups == 1) p.pad_mode = MS_PAD_ZERO;
        if (!msm_bwd_data_applicable(p)) return 0;
        size_t n = msm_bwd_data_ws(p);
        if (pad4_applicable(p)) {         // padded copies of gy, y_act and gx + the padded problem's own workspace
            const ConvP q = pad4_conv(p);
            const size_t m = 2 * pad4_bytes(q, p.Cout) + pad4_bytes(q, p.Cin) + msm_bwd_data_ws(q);
            if (m > n) n = m;
        }
        return n;
    }
    if (which == 2) {
        if (mst_bwd_weight_applicable(p)) return mst_bwd_weight_ws(p);
        if (msws_applicable(p)) {
            const size_t a = msws_ws(p);
            const size_t rest = msm_bwd_weight_applicable(p) ? msm_bwd_weight_ws(p) : msk_conv1d_bwd_weight_ws(p);
            return a > rest ? a : rest;
        }
        if (msw32_applicable(p)) {
            const size_t a32 = msw32_ws(p);
            const size_t rest = msm_bwd_weight_applicable(p) ? msm_bwd_weight_ws(p) : msk_conv1d_bwd_weight_ws(p);
            return a32 > rest ? a32 : rest;
        }
        if (msw_bwd_weight_applicable(p)) {
            size_t a = msw_bwd_weight_ws(p);
            if (msw5_applicable(p) && msw5_ws(p) > a) a = msw5_ws(p);
            const size_t rest = msm_bwd_weight_applicable(p) ? msm_bwd_weight_ws(p)
                                : (msg_bwd_weight_applicable(p) ? msg_bwd_weight_ws(p) : msk_conv1d_bwd_weight_ws(p));
            return a > rest ? a : rest;
        }
        return msm_bwd_weight_applicable(p) ? msm_bwd_weight_ws(p)
               : (msg_bwd_weight_applicable(p) ? msg_bwd_weight_ws(p) : msk_conv1d_bwd_weight_ws(p));
    }
    return 0;
}

const char* ms_conv1d_kernel_name(const ms_conv1d_desc* d, int which) {
    ConvP p;
    if (!make_conv(d, &p)) return "";
    if (which == 0 && mst_fwd_short_applicable(p)) return mst_fwd_name(p);
    if (which == 0 && mss_conv_applicable(p)) return mss_conv_name(p);
    // (rows padded to a multiple of 4, see pad4_applicable: the kernel of the padded problem)
    if (which == 0 && msm_fwd_applicable(p) && pad4_applicable(p)) return msm_fwd_name(pad4_conv(p));
    if (which == 1 && msm_bwd_data_applicable(p) && pad4_applicable(p)) return msm_bwd_data_name(pad4_conv(p));
    if (!p.in_act && !msm_fwd_applicable(p)) {
        const ms_conv1d_parts q = one_part(p);
        if (msg4_parts_applicable(p, &q)) return which == 0 ? "k_g4_fwd" : (which == 1 ? "k_g4_bwd_data" : "k_g4_wgrad");
    }
    if (which == 0)
        return msm_fwd_applicable(p) ? msm_fwd_name(p)
               : (msg3_fwd_applicable(p) ? msg3_fwd_name(p)
                  : msg_fwd_applicable(p) ? msg_fwd_name(p)
                  : (mst_fwd_applicable(p) ? mst_fwd_name(p) : msk_conv1d_fwd_direct_name(p)));
    if (which == 1 && p.pad_mode == MS_PAD_REFLECT && p.stride == 1 && p.groups == 1) {
        ConvP z = p;
        z.pad_mode = MS_PAD_ZERO;       // zero-padded backward + edge fold
        if (msm_bwd_data_applicable(z)) return msm_bwd_data_name(z);
    }
    if (which == 1)
        return msm_bwd_data_applicable(p) ? msm_bwd_data_name(p)
               : (msg3_bwd_data_applicable(p) ? msg3_bwd_data_name(p)
                  : msg_bwd_data_applicable(p) ? msg_bwd_data_name(p)
                  : (mst_bwd_data_applicable(p) && p.pad_mode == MS_PAD_ZERO ? mst_bwd_data_name(p)
                                                                              : msk_conv1d_bwd_data_direct_name(p)));
    if (which == 2 && mst_bwd_weight_applicable(p)) return mst_bwd_weight_name(p);
    if (which == 2 && msws_applicable(p)) return "k_wgrad_short";
    if (which == 2 && msw32_applicable(p)) return p.act == MS_ACT_LRELU ? "k_wgrad32<1>" : "k_wgrad32<0>";
    if (which == 2 && msw5_applicable(p)) return msw5_name(p);
    if (which == 2 && msw_bwd_weight_applicable(p)) return msw_bwd_weight_name(p);
    if (which == 2)
        return msm_bwd_weight_applicable(p) ? msm_bwd_weight_name(p)
               : (msg3_bwd_weight_applicable(p) ? msg3_bwd_weight_name(p)
                  : msg_bwd_weight_applicable(p) ? msg_bwd_weight_name(p) : msk_conv1d_bwd_weight_direct_name(p));
    return "";
}

const char* ms_convt1d_kernel_name(const ms_convt1d_desc* d, int which) {
    ConvP p;
    if (!make_convt(d, &p)) return "";
    if (which == 0 && mst_convt1_applicable(p)) return mst_convt1_fwd_name();
    if (which == 0 && mss_convt_applicable(d)) return mss_convt_name(d);
    if (which == 2 && mst_convt1_applicable(p)) return mst_convt1_wgrad_name();
    if (which == 0) return msm_convt_fwd_applicable(p) ? msm_convt_fwd_name(p) : msk_conv1d_bwd_data_direct_name(p);
    if (which == 1) {
        if (msm_convt_bwd_applicable(p)) return msm_convt_bwd_data_name(p);
        ConvP q = p;
        q.act = MS_ACT_NONE;
        return msm_fwd_applicable(q) ? msm_fwd_name(q) : msk_conv1d_fwd_direct_name(q);
    }
    if (which == 2) {
        if (mswt8_applicable(p)) return mswt8_name(p);
        if (mswt2s_applicable(p)) return mswt2s_name(p);
        if (msm_convt_bwd_applicable(p)) return msm_convt_bwd_weight_name(p);
        return msm_bwd_weight_applicable(p) ? msm_bwd_weight_name(p) : msk_conv1d_bwd_weight_direct_name(p);
    }
    return "";
}

int ms_convt1d_out_len(const ms_convt1d_desc* d) {
    ConvP p;
    return make_convt(d, &p) ? p.Lin : MS_ERR_INVALID_ARG;
}

// y = act(bias + conv_transpose(x, w)) == backward-data of the mirrored conv, with epilogue
int ms_convt1d_fwd(const ms_convt1d_desc* d, const float* x, const float* w, const float* bias,
                   float* y, void* workspace, size_t workspace_bytes, ms_stream_t stream) {
    ConvP p;
    if (!make_convt(d, &p) || !x || !w || !y) return MS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (mst_convt1_applicable(p)) return mst_convt1_fwd(p, x, w, bias, y, s);      // one output channel: a stream
    if (mss_convt_applicable(d)) {                                                 // inference batch: a weight stream
        const int rc = mss_convt_fwd(d, x, w, bias, y, s);
        if (rc != MS_ERR_UNSUPPORTED) return rc;
    }
    if (msm_convt_fwd_applicable(p))
        return msm_convt1d_fwd(p, x, w, bias, y, workspace, workspace_bytes, s);
    ConvP q = p;     // direct path: the loader modifier kind rides in q.act, the epilogue gets p.act
    q.act = p.in_act ? MS_MOD_LRELU_FWD : MS_ACT_NONE;
    return msk_conv1d_bwd_data_direct(q, x, p.in_act ? x : nullptr, w, bias, p.act, nullptr, y, s);
}

// gx = conv(gy * act'(y_act), w) with the mirrored conv geometry (no bias / activation)
int ms_convt1d_bwd_data(const ms_convt1d_desc* d, const float* gy, const float* y_act,
                        const float* w, float* gx, void* workspace, size_t workspace_bytes,
                        ms_stream_t stream) {
    ConvP p;
    if (!make_convt(d, &p) || !gy || !w || !gx) return MS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (msm_convt_bwd_applicable(p))
        return msm_convt1d_bwd_data(p, gy, y_act, w, gx, workspace, workspace_bytes, s);
    ConvP q = p;
    q.act = MS_ACT_NONE;
    if (msm_fwd_applicable(q))
        return msm_conv1d_fwd(q, gy, y_act, p.act, w, nullptr, nullptr, gx, nullptr, workspace,
                              workspace_bytes, s);
    return msk_conv1d_fwd_direct(q, gy, y_act, p.act, w, nullptr, nullptr, gx, nullptr, s);
}

// gw[ci_T, co_T, k] = sum x[b,ci_T,i] * gp[b,co_T,i*stride - pad + k]: the mirrored conv's weight
// grad with its "input" = gp (activation modifier on that side) and its "output grad" = x.
int ms_convt1d_bwd_weight(const ms_convt1d_desc* d, const float* x, const float* gy,
                          const float* y_act, float* gw, float* gb, float beta, void* workspace,
                          size_t workspace_bytes, ms_stream_t stream) {
    ConvP p;
    if (!make_convt(d, &p) || !x || !gy || !gw) return MS_ERR_INVALID_ARG;
    if (beta != 0.f && beta != 1.f) return MS_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    int rc = MS_ERR_UNSUPPORTED;
    if (mst_convt1_applicable(p) && workspace && workspace_bytes >= mst_convt1_wgrad_ws(p) + msk_channel_sum_ws(p.Cin) + 32)
        rc = mst_convt1_bwd_weight(p, x, gy, y_act, gw, beta, workspace, workspace_bytes, s);
    else if (mswt8_applicable(p) && workspace && workspace_bytes >= mswt8_ws(p) + msk_channel_sum_ws(p.Cin) + 32)
        rc = mswt8_bwd_weight(p, x, gy, y_act, gw, beta, workspace, workspace_bytes, s);
    else if (mswt2s_applicable(p) && workspace && workspace_bytes >= mswt2s_ws(p) + msk_channel_sum_ws(p.Cin) + 32)
        rc = mswt2s_bwd_weight(p, x, gy, y_act, gw, beta, workspace, workspace_bytes, s);
    if (rc != MS_ERR_UNSUPPORTED) {
    } else if (msm_convt_bwd_applicable(p))
        rc = msm_convt1d_bwd_weight(p, x, gy, y_act, gw, beta, workspace, workspace_bytes, s);
    else if (msm_bwd_weight_applicable(p))
        rc = msm_conv1d_bwd_weight(p, gy, y_act, p.act, x, p.in_act ? x : nullptr,
                                   p.in_act ? MS_MOD_LRELU_FWD : 0, gw, nullptr, beta, workspace,
                                   workspace_bytes, s);
    else
        rc = msk_conv1d_bwd_weight_direct(p, gy, y_act, p.act, x, p.in_act ? x : nullptr,
                                          p.in_act ? MS_MOD_LRELU_FWD : 0, gw, nullptr, beta,
                                          workspace, workspace_bytes, s);
    if (rc != MS_OK) return rc;
    if (gb) {   // bias grad: the slice partials live at the tail of the workspace
        const size_t tail = msk_channel_sum_ws(p.Cin);
        if (!workspace || workspace_bytes < tail) return MS_ERR_WORKSPACE;
        char* wtail = (char*)workspace + (workspace_bytes - tail);
        wtail -= ((uintptr_t)wtail) & 15;
        if (wtail < (char*)workspace) return MS_ERR_WORKSPACE;
        return msk_channel_sum(gy, y_act, p.act, p.slope, p.B, p.Cin, p.Lin, gb, beta, wtail, tail, s);
    }
    return MS_OK;
}

size_t ms_convt1d_workspace_bytes(const ms_convt1d_desc* d, int which) {
    ConvP p;
    if (!make_convt(d, &p)) return 0;
    if (which == 0) return msm_convt_fwd_applicable(p) ? msm_convt_fwd_ws(p) : 0;
    if (which == 1) {
        if (msm_convt_bwd_applicable(p)) return msm_convt_bwd_data_ws(p);
        ConvP q = p;
        q.act = MS_ACT_NONE;
        return msm_fwd_applicable(q) ? msm_fwd_ws(q) : 0;
    }
    if (which == 2) {
        const size_t tail = msk_channel_sum_ws(p.Cin) + 32;   // bias-grad slice partials
        size_t t8 = mswt8_applicable(p) ? mswt8_ws(p) : 0;
        if (mst_convt1_applicable(p) && mst_convt1_wgrad_ws(p) > t8) t8 = mst_convt1_wgrad_ws(p);
        if (mswt2s_applicable(p) && mswt2s_ws(p) > t8) t8 = mswt2s_ws(p);
        size_t n = msm_convt_bwd_applicable(p) ? msm_convt_bwd_weight_ws(p)
                   : (msm_bwd_weight_applicable(p) ? msm_bwd_weight_ws(p) : msk_conv1d_bwd_weight_ws(p));
        if (t8 > n) n = t8;
        return n + tail;
    }
    return 0;
}

}  // extern "C"
