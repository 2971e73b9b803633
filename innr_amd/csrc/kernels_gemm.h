// kernels_gemm.h -- f32 MFMA GEMM engine (placeholder until the kernel lands in this round).
#pragma once
#include "common.h"
