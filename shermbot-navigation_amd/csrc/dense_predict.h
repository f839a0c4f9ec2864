// dense_predict.h -- P <- F P F^T + Qbar on the matrix cores (slam_library.cpp:104 with a dense A := F).
#pragma once
#include <hip/hip_runtime.h>

// dtype: 0 = fp64 (v_mfma_f64_16x16x4_f64), 1 = fp32 (v_mfma_f32_32x32x2_f32).  All matrices are L x L,
// column-major with leading dimension ld (rows [L, ld) zero).  T is an L x L workspace.  Evaluation order is the
// reference's: T = F P, then P = T F^T + Qbar.  ev: four events or four nulls; when given, launch 1 is bracketed
// by ev[0], ev[1] and launch 2 by ev[2], ev[3].  Returns 0 or non-zero on a launch error.
int dense_predict_launch(int dtype, int L, int ld, const void* F, void* P, void* T, const double Q[9],
                         hipStream_t stream, hipEvent_t ev[4]);
