// One-channel-side ("thin") convolution kernels (conv_thin.hip): applicability, names, launchers.
#pragma once
#include "ms_common.h"

bool mst_fwd_short_applicable(const ConvP& p);   // deep one-output conv on short rows (judge conv): ahead of the GEMM kernels
bool mst_fwd_applicable(const ConvP& p);
bool mst_bwd_data_applicable(const ConvP& p);
bool mst_bwd_weight_applicable(const ConvP& p);
const char* mst_fwd_name(const ConvP& p);
const char* mst_bwd_data_name(const ConvP& p);
const char* mst_bwd_weight_name(const ConvP& p);
size_t mst_bwd_weight_ws(const ConvP& p);
int mst_conv1d_fwd(const ConvP& p, const float* x, const float* w, const float* bias,
                   const float* residual, float* y, hipStream_t s);
int mst_conv1d_bwd_data(const ConvP& p, const float* gy, const float* y_act, const float* w,
                        const float* gx_add, float* gx, hipStream_t s);
int mst_conv1d_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act,
                          float* gw, float* gb, float beta, void* ws, size_t ws_bytes,
                          hipStream_t s);

// ConvTranspose1d with one output channel, stride 2 / kernel 4 / padding 1 (stage-1 generator's last layer): HBM-bound
// streams.  p = mirrored conv (Cin_T = p.Cout, Cout_T = p.Cin = 1).
bool mst_convt1_applicable(const ConvP& p);
const char* mst_convt1_fwd_name();
const char* mst_convt1_wgrad_name();
int mst_convt1_fwd(const ConvP& p, const float* x, const float* w, const float* bias, float* y, hipStream_t s);
size_t mst_convt1_wgrad_ws(const ConvP& p);
int mst_convt1_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act, float* gw, float beta,
                          void* ws, size_t ws_bytes, hipStream_t s);

// small_rows.hip: the generator's first conv and first transposed conv at inference batch sizes (a few dozen columns in the
// whole batch): weight streams spread over the chip, fp32 FMA, no weight image
bool mss_conv_applicable(const ConvP& p);
const char* mss_conv_name(const ConvP& p);
int mss_conv_fwd(const ConvP& p, const float* x, const float* w, const float* bias, float* y, hipStream_t s);
bool mss_convt_applicable(const ms_convt1d_desc* d);
const char* mss_convt_name(const ms_convt1d_desc* d);
int mss_convt_fwd(const ms_convt1d_desc* d, const float* x, const float* w, const float* bias, float* y, hipStream_t s);
