// f32-MFMA implicit-GEMM convolution kernels for gfx950 (placeholder: predicates return false
// until the kernels land, so every layer runs on the direct kernels).
#include "conv_mfma.h"

bool msm_fwd_applicable(const ConvP&) { return false; }
bool msm_bwd_data_applicable(const ConvP&) { return false; }
bool msm_bwd_weight_applicable(const ConvP&) { return false; }
bool msm_convt_fwd_applicable(const ConvP&) { return false; }
size_t msm_fwd_ws(const ConvP&) { return 0; }
size_t msm_bwd_data_ws(const ConvP&) { return 0; }
size_t msm_bwd_weight_ws(const ConvP&) { return 0; }
size_t msm_convt_fwd_ws(const ConvP&) { return 0; }
int msm_conv1d_fwd(const ConvP&, const float*, const float*, int, const float*, const float*,
                   const float*, float*, float*, void*, size_t, hipStream_t) { return MS_ERR_UNSUPPORTED; }
int msm_conv1d_bwd_data(const ConvP&, const float*, const float*, const float*, const float*,
                        float*, void*, size_t, hipStream_t) { return MS_ERR_UNSUPPORTED; }
int msm_conv1d_bwd_weight(const ConvP&, const float*, const float*, int, const float*,
                          const float*, int, float*, float*, float, void*, size_t, hipStream_t) { return MS_ERR_UNSUPPORTED; }
int msm_convt1d_fwd(const ConvP&, const float*, const float*, const float*, float*, void*, size_t,
                    hipStream_t) { return MS_ERR_UNSUPPORTED; }
const char* msm_fwd_name(const ConvP&) { return ""; }
const char* msm_bwd_data_name(const ConvP&) { return ""; }
const char* msm_bwd_weight_name(const ConvP&) { return ""; }
const char* msm_convt_fwd_name(const ConvP&) { return ""; }
