// Shared helpers for the gfx950 kernels of libemoportraits_hip.so (device code is CDNA4-only: wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/emo_hip.h"

#define EMO_WAVE 64

static inline int emo_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? EMO_OK : (int)e;
}

static inline bool emo_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

static inline int emo_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// compute units of the CURRENT device (the persistent grids of the split convolution are sized by it), cached per device id.
// Rounded down to a multiple of 8, and at least 8: those kernels hand out their work items in eight contiguous ranges, one per
// XCD (block b walks the range of XCD b % 8: conv_igemm_bf16x3.h), so a grid of min(items, this) blocks covers every item only
// when it holds all eight residues equally often.  MI355X has 256 / 128 / 64 / 32 CUs per partition: unchanged there.
static inline int emo_cu_count() {
  static int cached[64] = {0};     // (benign race: every writer stores the same value)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n < 8 ? 8 : (n & ~7);
  }
  return cached[dev];
}
