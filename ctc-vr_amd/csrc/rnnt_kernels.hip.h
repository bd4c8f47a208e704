// HIP kernels of librnnt_hip.so (gfx950), one translation unit: include order matters (helpers first).
#pragma once
#include "rnnt_common.hip.h"
#include "rnnt_gemm.hip.h"
#include "rnnt_gemm_bf.hip.h"
#include "rnnt_joint.hip.h"
#include "rnnt_encoder.hip.h"
#include "rnnt_encoder_lm.hip.h"
#include "rnnt_decode.hip.h"
#include "rnnt_frontend.hip.h"
#include "rnnt_beam.hip.h"
#include "rnnt_misc.hip.h"
#include "rnnt_fused.hip.h"
#include "rnnt_gemm_as.hip.h"
