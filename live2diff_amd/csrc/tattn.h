// Shared between tattn.hip (register-resident / chunked / warm-up kernels) and tattn_ring.hip (LDS-DMA ring kernel).
#pragma once
#include "common.h"

struct TAttnArgs {
    const h16 *qkv;
    h16 *cache;
    const h16 *q_pe, *k_pe, *v_pe;
    const long long *pe_idx, *update_idx;
    const h16 *bias;
    h16 *out;
    int N, T, C, L, H, variant;
#ifdef L2D_PROBES
    unsigned long long *probe;   // analysis builds: s_memtime stamps of two stages per block (tools/tattn_probe.py)
#endif
};

// true when the ring kernel covers the shape: C in {320, 640, 1280}, L in {12, 16}, T % 8 == 0, and the caller gave
// a 16-byte zero page (the DMA source of masked slots)
bool l2d_tattn_ring_ok(const TAttnArgs &a, const void *zero_page);
int l2d_launch_tattn_ring(const TAttnArgs &a, const void *zero_page, hipStream_t s);
