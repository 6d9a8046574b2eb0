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

// The shipped library reads NO environment variable and holds no tunable state: launch parameters are fixed or arguments.
// The A/B knobs of the kernel experiments (QCN_MODES_WG_PER_CU, QCN_MODES_LEAN, QCN_ATTN_R, QCN_PAIR_MIN_ROWS,
// QCN_CHAIN_SKEW, QCN_CHAIN_WAVES) exist only in diagnostic builds (-DQCN_DEBUG_KNOBS, scripts/dbg/build_variant.sh).
#ifdef QCN_DEBUG_KNOBS
#include <stdlib.h>
static inline const char *qcn_knob(const char *name) { return getenv(name); }
#else
static inline const char *qcn_knob(const char *) { return nullptr; }
#endif

// Compute units of the current device (grids of the persistent kernels are sized by it); read once per device.
// Plain int cache: concurrent first calls write the same value.
static inline int qcn_num_cus()
{
    static int cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cache[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev] = n;
    }
    return cache[dev];
}
