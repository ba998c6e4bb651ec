// Shared device/host helpers for libmsynth_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "msynth.h"

#define MS_WAVE 64

#define MS_CHECK_LAUNCH()                                   \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return MS_ERR_LAUNCH; \
    } while (0)

// Profiling aid: while the calling thread is in a profile session (ms_profile_kernels) the launchers of the templated dense
// families note the instantiation they dispatched, in the spelling rocprofv3 prints ("k_conv_rows3p<2, 2, 2, 3, 0, 0, false>"),
// and the arithmetic it runs: `products` = matrix-pipe products per fp32 multiply -- 6 (exact three-piece bf16 split), 3
// (block-scaled two-piece fp16 split), 0 (fp32-input MFMA or vector FMA).  ms_profile_take() hands both to the caller, so a
// profiler line is matched to a layer and priced on the right pipe without re-deriving the dispatch.  No-op outside a session.
void ms_note_kernel(int products, const char* fmt, ...);

// Profiling aid (ms_profile_kernels / ms_profile_take, api.hip): while a thread is in profile mode every kernel launch of the
// library carries a start / stop event pair (hipExtLaunchKernelGGL: the events take the dispatch's own begin / end timestamps,
// the durations rocprofv3 reports), is waited for, and its device time is added to the thread's counter.  Outside profile mode
// a launch is the plain <<< >>> it always was (capturable, no events).
#include <hip/hip_ext.h>
bool ms_prof_on();
void ms_prof_add(hipEvent_t e0, hipEvent_t e1);
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                       \
    do {                                                                                                   \
        if (ms_prof_on()) {                                                                                \
            hipEvent_t ms_e0_ = nullptr, ms_e1_ = nullptr;                                                 \
            (void)hipEventCreate(&ms_e0_);                                                                 \
            (void)hipEventCreate(&ms_e1_);                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, ms_e0_, ms_e1_, 0, __VA_ARGS__);     \
            ms_prof_add(ms_e0_, ms_e1_);                                                                   \
        } else {                                                                                           \
            (kernel)<<<(grid), (block), (shmem), (stream)>>>(__VA_ARGS__);                                 \
        }                                                                                                  \
    } while (0)


// Launch parameters that are set / queried once are cached PER DEVICE (a process may drive several GPUs: the > 64 KiB
// dynamic-LDS opt-in is a per-device function attribute, CU counts differ) and without locks: the first callers on a device
// may race, they all compute and store the same values.  ms_first_on_device(mask): true until the calling site has
// finished its one-time work on the current device (call ms_done_on_device(mask) then).
static inline int ms_current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev & 63;
}
static inline bool ms_first_on_device(const unsigned long long& mask) {
    return !((__atomic_load_n(&mask, __ATOMIC_ACQUIRE) >> ms_current_device()) & 1ull);
}
static inline void ms_done_on_device(unsigned long long& mask) {
    __atomic_fetch_or(&mask, 1ull << ms_current_device(), __ATOMIC_RELEASE);
}

struct ConvP {  // kernel-side copy of ms_conv1d_desc (+ derived sizes)
    int B, Cin, Lin, Cout, Lout, K, stride, pad, dil, groups, Cg, Og, pad_mode, act;
    float slope;
    int in_act;   // activation applied to the conv input on load (NONE / LRELU)
};

// operand-modifier kind: "apply LeakyReLU to the loaded value itself" (pre-activation convs);
// the kinds 0..2 (ms_act) mean "multiply by the derivative of that activation at ya"
#define MS_MOD_LRELU_FWD 3

static inline int ms_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int ms_floor_div(int a, int b) {  // b > 0
    int q = a / b;
    if ((a % b) && (a < 0)) --q;
    return q;
}

__device__ __forceinline__ float ms_apply_act(float v, int act, float slope) {
    if (act == MS_ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == MS_ACT_TANH) return tanhf(v);
    return v;
}

// d act(pre)/d pre expressed through the saved post-activation value ya
__device__ __forceinline__ float ms_act_grad(float g, float ya, int act, float slope) {
    if (act == MS_MOD_LRELU_FWD) return g > 0.f ? g : g * slope;
    if (act == MS_ACT_LRELU) return ya > 0.f ? g : g * slope;
    if (act == MS_ACT_TANH) return g * (1.f - ya * ya);
    return g;
}

// position in the unpadded row for padded coordinate t; -1 = zero padding
__device__ __forceinline__ int ms_src_index(int t, int L, int pad_mode) {
    if (t >= 0 && t < L) return t;
    if (pad_mode == MS_PAD_REFLECT) {
        if (t < 0) t = -t;
        if (t >= L) t = 2 * (L - 1) - t;
        return (t >= 0 && t < L) ? t : -1;
    }
    return -1;
}

__device__ __forceinline__ float ms_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum; result valid in thread 0.  `red` is >= (blockDim.x/64) floats of LDS.
__device__ __forceinline__ float ms_block_sum(float v, float* red) {
    v = ms_wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += red[i];
    }
    return r;
}

// internal launchers (defined in the .hip files, called from ms_api.hip)
int msk_conv1d_fwd_direct(const ConvP& p, const float* x, const float* x_act, int x_act_kind,
                          const float* w, const float* bias, const float* residual, float* y,
                          float* y_act, hipStream_t s);
int msk_conv1d_bwd_data_direct(const ConvP& p, const float* gy, const float* y_act, const float* w,
                               const float* bias, int out_act, const float* gx_add, float* gx,
                               hipStream_t s);
size_t msk_conv1d_bwd_weight_ws(const ConvP& p);
int msk_conv1d_bwd_weight_direct(const ConvP& p, const float* x, const float* x_act,
                                 int x_act_kind, const float* gy, const float* y_act,
                                 int y_act_kind, float* gw, float* gb, float beta, void* ws,
                                 size_t ws_bytes, hipStream_t s);
int msk_reflect_fold_bwd(const ConvP& p, const float* gy, const float* y_act, const float* w,
                         float* gx, hipStream_t s);
const char* msk_conv1d_fwd_direct_name(const ConvP& p);
const char* msk_conv1d_bwd_data_direct_name(const ConvP& p);
const char* msk_conv1d_bwd_weight_direct_name(const ConvP& p);
// out = beta*out + sum over nsplit partial slabs (weights, then nbias bias entries per slab)
int msk_reduce_partials(const float* partial, size_t partial_stride, int nsplit, size_t wsize,
                        int nbias, float* gw, float* gb, float beta, hipStream_t s);
size_t msk_channel_sum_ws(int C);
int msk_channel_sum(const float* g, const float* y_act, int act, float slope, int B, int C, int L,
                    float* out, float beta, void* ws, size_t ws_bytes, hipStream_t s);
