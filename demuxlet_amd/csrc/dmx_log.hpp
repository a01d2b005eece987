// Natural logarithm for the likelihood kernels.
//
// The reference calls libm's log() (glibc: < 1 ulp, not correctly rounded), so no device implementation can promise
// the same bits; what STRICT mode promises is the reference's operation ORDER around it.  dmx_log() must therefore be
// (a) accurate to < 1 ulp so per-term differences stay at the 1e-16 level and (b) cheap, because it is >45 % of the
// FP64 work of every kernel.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ double dmx_log(double x) { return log(x); }
