// conv3x3_wgrad_t.h -- the transposed-image 3x3 filter-gradient kernel (conv3x3_wgrad_t.hip) as dpp_conv3x3_wgrad (conv3x3.hip) calls it.
#pragma once
#include "dpp_common.h"

// does the kernel take this layer (16 or 32 channels in and out, maps at least 12 wide)?
bool dpp_conv3x3_wgrad_t_ok(int N, int H, int W, int Ci, int Co, const dpp_act* act);
// partial [nblk][C][9][C]; grid (nblk, 9 / taps_pb) as wgrad_geometry chose it; store: DPP_ST_A (X) / DPP_ST_B (dY) hold bf16 elements;
// precision 1: bf16 MFMA operands (act(X) rounded RNE after the prologue, dY rounded RNE -- exact when it is bf16-stored)
int dpp_conv3x3_wgrad_t_launch(const float* X, int N, int H, int W, int C, const dpp_act* act, const float* dY, float* partial, int nblk,
                               int taps_pb, int store, int precision, hipStream_t stream);
