// Grouped k41/stride-4 conv kernels on the fp32 matrix cores (gconv_mfma.hip).
#pragma once
#include "ms_common.h"

bool msg_fwd_applicable(const ConvP& p);
const char* msg_fwd_name(const ConvP& p);
int msg_conv1d_fwd(const ConvP& p, const float* x, const float* w, const float* bias, float* y,
                   hipStream_t s);

bool msg_bwd_data_applicable(const ConvP& p);
const char* msg_bwd_data_name(const ConvP& p);
int msg_conv1d_bwd_data(const ConvP& p, const float* gy, const float* y_act, const float* w,
                        const float* gx_add, float* gx, hipStream_t s);

bool msg_bwd_weight_applicable(const ConvP& p);
size_t msg_bwd_weight_ws(const ConvP& p);
const char* msg_bwd_weight_name(const ConvP& p);
int msg_conv1d_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act,
                          float* gw, float* gb, float beta, void* ws, size_t ws_bytes, hipStream_t s);
int msg_wgrad_gridx(const ConvP& p);       // workgroup columns of the weight-gradient kernels (4 slabs each)
int msg_reduce_slabs(const ConvP& p, const float* partial, size_t stride, int nslabs, float* gw, float* gb,
                     float beta, hipStream_t s);

// split-bf16 generation (gconv_split.hip); MSYNTH_GCONV3=0 falls back to the fp32-MFMA kernels above
bool msg3_fwd_applicable(const ConvP& p);
const char* msg3_fwd_name(const ConvP& p);
int msg3_conv1d_fwd(const ConvP& p, const float* x, const float* w, const float* bias, float* y,
                    hipStream_t s);
const char* msg3_bwd_weight_name(const ConvP& p);
bool msg3_bwd_weight_applicable(const ConvP& p);
int msg3_conv1d_bwd_weight(const ConvP& p, const float* x, const float* gy, const float* y_act,
                           float* gw, float* gb, float beta, void* ws, size_t ws_bytes, hipStream_t s);
bool msg3_bwd_data_applicable(const ConvP& p);
const char* msg3_bwd_data_name(const ConvP& p);
int msg3_conv1d_bwd_data(const ConvP& p, const float* gy, const float* y_act, const float* w,
                         const float* gx_add, float* gx, hipStream_t s);

// parts launches (gconv_split.hip): one layer over the discriminator's scales, include/msynth.h ms_conv1d_parts
bool msg3_parts_fwd_applicable(const ConvP& c, const ms_conv1d_parts* parts);
int msg3_parts_fwd(const ConvP& c, const ms_conv1d_parts* parts, const float* w, const float* bias, hipStream_t s);
bool msg3_parts_bwd_data_applicable(const ConvP& c, const ms_conv1d_parts* parts);
int msg3_parts_bwd_data(const ConvP& c, const ms_conv1d_parts* parts, const float* w, hipStream_t s);
bool msg3_parts_bwd_weight_applicable(const ConvP& c, const ms_conv1d_parts* parts);
size_t msg3_parts_bwd_weight_ws(const ConvP& c, const ms_conv1d_parts* parts);
int msg3_parts_bwd_weight(const ConvP& c, const ms_conv1d_parts* parts, float* gw, float* gb, float beta, void* ws,
                          size_t ws_bytes, hipStream_t s);
// the k5 layer over the scales (conv5_img.hip)
bool ms5_parts_applicable(const ConvP& c, const ms_conv1d_parts* parts, bool backward);
size_t ms5_parts_ws(const ConvP& c, const ms_conv1d_parts* parts, bool backward);
int ms5_parts_fwd(const ConvP& c, const ms_conv1d_parts* parts, const void* image, const float* bias, void* ws, size_t ws_bytes,
                  hipStream_t s);
int ms5_parts_bwd_data(const ConvP& c, const ms_conv1d_parts* parts, const void* image_bwd, void* ws, size_t ws_bytes,
                       hipStream_t s);
// the k5 layer's weight gradient over the scales (wgrad_k5.hip)
bool msw5_parts_applicable(const ConvP& c, const ms_conv1d_parts* parts);
size_t msw5_parts_ws(const ConvP& c, const ms_conv1d_parts* parts);
int msw5_parts_bwd_weight(const ConvP& c, const ms_conv1d_parts* parts, float* gw, float* gb, float beta, void* ws,
                          size_t ws_bytes, hipStream_t s);
// the discriminator's first conv (1 -> 16, k15) and judge conv (1024 -> 1, k3) over the scales (disc_parts.hip)
bool msd_parts_applicable(const ConvP& c, const ms_conv1d_parts* parts, int which);
size_t msd_parts_bwd_weight_ws(const ConvP& c, const ms_conv1d_parts* parts);
int msd_parts_fwd(const ConvP& c, const ms_conv1d_parts* parts, const float* w, const float* bias, hipStream_t s);
int msd_parts_bwd_data(const ConvP& c, const ms_conv1d_parts* parts, const float* w, hipStream_t s);
int msd_parts_bwd_weight(const ConvP& c, const ms_conv1d_parts* parts, float* gw, float* gb, float beta, void* ws,
                         size_t ws_bytes, hipStream_t s);
// the 256-group layer (4 x 4 channels per group) on the vector pipe, fp32 FMA (gconv4.hip); count = 1 .. 3 parts
bool msg4_parts_applicable(const ConvP& c, const ms_conv1d_parts* parts);
int msg4_parts_fwd(const ConvP& c, const ms_conv1d_parts* parts, const float* w, const float* bias, hipStream_t s);
int msg4_parts_bwd_data(const ConvP& c, const ms_conv1d_parts* parts, const float* w, hipStream_t s);
int msg4_parts_bwd_weight(const ConvP& c, const ms_conv1d_parts* parts, float* gw, float* gb, float beta, hipStream_t s);
