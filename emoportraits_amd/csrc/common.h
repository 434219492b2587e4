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
