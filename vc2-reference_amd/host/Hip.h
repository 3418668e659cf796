// Hip.h -- process-wide libvc2hip contexts and the error-code -> exception mapping.
#ifndef VC2HOST_HIP_H
#define VC2HOST_HIP_H

#include "vc2hip.h"

vc2hip_ctx *hipContext(int device = 0);    // created on first use; throws std::runtime_error without a GPU
void hipCheck(vc2hip_ctx *ctx, int rc);    // throws the reference's exception type with its what() string
#endif
