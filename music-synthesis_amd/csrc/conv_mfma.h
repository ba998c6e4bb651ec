// f32-MFMA implicit-GEMM convolution kernels (conv_mfma.hip): applicability predicates,
// workspace queries and launchers used by api.hip.
#pragma once
#include "ms_common.h"

bool msm_fwd_applicable(const ConvP& p);
bool msm_bwd_data_applicable(const ConvP& p);
bool msm_bwd_weight_applicable(const ConvP& p);
bool msm_convt_fwd_applicable(const ConvP& p);  // p = mirrored conv of the transposed conv
size_t msm_fwd_ws(const ConvP& p);
size_t msm_bwd_data_ws(const ConvP& p);
size_t msm_bwd_weight_ws(const ConvP& p);
size_t msm_convt_fwd_ws(const ConvP& p);

const char* msm_fwd_name(const ConvP& p);
const char* msm_bwd_data_name(const ConvP& p);
const char* msm_bwd_weight_name(const ConvP& p);
const char* msm_convt_fwd_name(const ConvP& p);

int msm_conv1d_fwd(const ConvP& p, const float* x, const float* x_act, int x_act_kind,
                   const float* w, const float* bias, const float* residual, float* y,
                   float* y_act, void* ws, size_t ws_bytes, hipStream_t s);
int msm_conv1d_bwd_data(const ConvP& p, const float* gy, const float* y_act, const float* w,
                        const float* gx_add, float* gx, void* ws, size_t ws_bytes, hipStream_t s);
int msm_conv1d_bwd_weight(const ConvP& p, const float* x, const float* x_act, int x_act_kind,
                          const float* gy, const float* y_act, int y_act_kind, float* gw,
                          float* gb, float beta, void* ws, size_t ws_bytes, hipStream_t s);
int msm_convt1d_fwd(const ConvP& p, const float* x, const float* w, const float* bias, float* y,
                    void* ws, size_t ws_bytes, hipStream_t s);

// ConvTranspose1d backward passes in phase-split form (p = mirrored conv of the transposed conv)
bool msm_convt_bwd_applicable(const ConvP& p);
size_t msm_convt_bwd_data_ws(const ConvP& p);
size_t msm_convt_bwd_weight_ws(const ConvP& p);
const char* msm_convt_bwd_data_name(const ConvP& p);
const char* msm_convt_bwd_weight_name(const ConvP& p);
int msm_convt1d_bwd_data(const ConvP& p, const float* gy, const float* y_act, const float* w,
                         float* gx, void* ws, size_t ws_bytes, hipStream_t s);
int msm_convt1d_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act,
                           float* gw, float beta, void* ws, size_t ws_bytes, hipStream_t s);

// deterministic split-K reduce shared by the weight-gradient kernels (conv_mfma.hip)
int msm_wgrad_reduce(const float* partial, size_t stride_floats, int nsplit, size_t wsize, int nbias,
                     float* gw, float* gb, float beta, hipStream_t s);

// split-bf16 weight gradient of the many-channel k5 conv on short rows (wgrad_k5.hip); MSYNTH_WGRAD5=0 disables it
bool msw5_applicable(const ConvP& p);
size_t msw5_ws(const ConvP& p);
const char* msw5_name(const ConvP& p);
int msw5_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act, float* gw, float* gb,
                    float beta, void* ws, size_t ws_bytes, hipStream_t s);

// split-bf16 weight gradient of the stride-8 / kernel-16 transposed conv (wgrad_convt.hip); MSYNTH_WGRADT8=0 disables
bool mswt8_applicable(const ConvP& p);
size_t mswt8_ws(const ConvP& p);
const char* mswt8_name(const ConvP& p);
int mswt8_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act, float* gw, float beta,
                     void* ws, size_t ws_bytes, hipStream_t s);

// split-bf16 weight gradient of the stride-2 / kernel-4 transposed conv on rows of 4 .. 32 positions (wgrad_convt2s.hip);
// MSYNTH_WGRADT2S=0 disables
bool mswt2s_applicable(const ConvP& p);
size_t mswt2s_ws(const ConvP& p);
const char* mswt2s_name(const ConvP& p);
int mswt2s_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act, float* gw, float beta,
                      void* ws, size_t ws_bytes, hipStream_t s);

// row-tile weight gradient (wgrad_rows.hip)
bool msw_bwd_weight_applicable(const ConvP& p);
size_t msw_bwd_weight_ws(const ConvP& p);
size_t msw32_multi_ws(const ConvP* cs, int n);
// signs: y_act[] are SIGN WORDS of the activations (atom_fused.hip, MASK), not the fp32 tensors
int msw32_bwd_weight_multi(const ConvP* cs, int n, const float* const* x, const float* const* gy,
                           const float* const* y_act, float* const* gw, float* const* gb, const float* beta, int signs,
                           void* ws, size_t ws_bytes, hipStream_t s);
bool msw32_multi_takes_signs(const ConvP* cs, int n);
bool msw_multi_takes_signs(const ConvP* cs, int n);
// n weight gradients of identical geometry in one launch (0 / UNSUPPORTED: the caller loops over single calls)
size_t msw_multi_ws(const ConvP* cs, int n);
int msw_conv1d_bwd_weight_multi(const ConvP* cs, int n, const float* const* x, const float* const* gy,
                                const float* const* y_act, float* const* gw, float* const* gb,
                                const float* beta, const float* const* xmax, const float* const* gmax, int signs, void* ws,
                                size_t ws_bytes, hipStream_t s);
const char* msw_bwd_weight_name(const ConvP& p);
int msw_conv1d_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act,
                          float* gw, float* gb, float beta, void* ws, size_t ws_bytes,
                          hipStream_t s);

// transposed-conv weight gradient, phase-split row-tile form (wgrad_rows.hip): dWq[Cin_T][(co, r), d]
size_t msw_convt_ws(const ConvP& p);
int msw_convt_dwq(const ConvP& p, const float* x, const float* gy, const float* y_act, float* dwq,
                  void* ws, size_t ws_bytes, hipStream_t s);

// 32 -> 32 k3 weight gradient, per-wave units (wgrad_rows.hip)
bool msw32_applicable(const ConvP& p);
size_t msw32_ws(const ConvP& p);
int msw32_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act, float* gw,
                     float* gb, float beta, void* ws, size_t ws_bytes, hipStream_t s);

// reflection-padded dense conv on short rows (the generator's first layer): weight gradient on fp32 MFMA (wgrad_short.hip)
bool msws_applicable(const ConvP& c);
size_t msws_ws(const ConvP& c);
int msws_bwd_weight(const ConvP& c, const float* x, const float* gy, const float* y_act, float* gw, float* gb, float beta,
                    void* ws, size_t ws_bytes, hipStream_t s);
