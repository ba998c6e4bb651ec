// Pipelined row-tile forward / backward-data convolution (conv_rows2.hip), called from the row-tile
// dispatch in conv_mfma.hip when its requirements hold.
#pragma once
#include "ms_common.h"
#include <stdlib.h>

struct Row2P {
    int B, CK, L, M, dil, off0, act, KG;
    int Lt, R, SS, RSZ, tiles_per_row, PX, CKs, scratch_off;
    long long zstride;
    float slope;
};

enum { MSR2_128x128 = 0, MSR2_64x128 = 1, MSR2_64x64 = 2, MSR2_32x256 = 3 };

// activation pieces (16-byte loads per thread and chunk) a tile of BN columns and CC channels needs
constexpr int msr2_nxq(int CC, int BN) { return (CC * (BN / 4 + 12) + 255) / 256; }

// act_mode 0: no activation operand; 1: LeakyReLU derivative from Xact, W in the forward layout;
// 2: the derivative with pre-packed W and in_s > 1 (transposed-conv backward data)
bool msr2_supported(int tile, int K, int CC, int act_mode, int epi_s, const Row2P& p, int in_s = 1);
int msr2_launch(int tile, int K, int CC, int act_mode, int epi_s, const Row2P& p, const float* X,
                const float* Xact, const float* W, const float* bias, const float* res, float* Y,
                float* Yact, unsigned gx, unsigned gy, unsigned gz, hipStream_t s, int in_s = 1);

// third generation (conv_rows3.hip): the same row tiles on the bf16 matrix pipe with every fp32 operand split
// exactly into three bf16 pieces (six partial products, fp32 accumulate): stride-1 plain rows, K in {3, 5},
// 16-channel chunks, act_mode 0 / 1, plain epilogue.  MSYNTH_ROWS3=0 disables it.
bool msr3_supported(int tile, int K, int act_mode, int epi_s, const Row2P& p, int in_s = 1);
int msr3_launch(int tile, int K, int act_mode, const Row2P& p, const float* X, const float* Xact, const float* W,
                const float* bias, const float* res, float* Y, float* Yact, unsigned gx, unsigned gy, unsigned gz,
                hipStream_t s);
// paired eight-wave form: bm (64 or 128) rows x two adjacent 128-column tiles per workgroup, the two wave groups
// alternating between the matrix pipe and the staging work (MSYNTH_ROWS3P=0 disables it)
bool msr3p_supported(int bm, int K, int act_mode, int epi_s, const Row2P& p, int in_s = 1);
int msr3p_launch(int bm, int K, int act_mode, const Row2P& p, const float* X, const float* Xact, const float* W,
                 const float* bias, const float* res, float* Y, float* Yact, unsigned gz, hipStream_t s);

// transposed-conv forward on the paired split-bf16 kernel (two live taps per phase; see k_conv_rows3p, HS form)
bool msr3p_convt_supported(int bm, int S, const Row2P& p);
int msr3p_convt_launch(int bm, int S, bool in_act, const Row2P& p, const float* X, const float* W, const float* bias,
                       float* Y, unsigned gz, hipStream_t s);
