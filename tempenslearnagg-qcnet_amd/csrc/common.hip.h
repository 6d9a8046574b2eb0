// Shared helpers for the gfx950 kernels of libqcnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qcnet_hip.h"

#define QCN_WAVE 64

#define QCN_CHECK_LAUNCH()                                  \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return QCN_ERR_LAUNCH_FAILED; \
    } while (0)

// Bijective XCD-aware block remap (cdna_hip_programming.md T1): blocks b and b+8 share an XCD, so give
// each XCD label a contiguous chunk of the logical block range; neighbouring destinations (which gather
// the same source rows) then hit the same 4 MiB L2.
__device__ __forceinline__ unsigned qcn_xcd_remap(unsigned bid, unsigned nblocks)
{
    const unsigned nx = 8;
    if (nblocks < 2 * nx) return bid;
    const unsigned q = nblocks / nx, r = nblocks % nx;
    const unsigned xcd = bid % nx, idx = bid / nx;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

static inline hipStream_t qcn_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
