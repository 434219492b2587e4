// ABI identity of libemoportraits_hip.so (see include/emo_hip.h).
#include "common.h"

extern "C" int emo_abi_version(void) { return EMO_ABI_VERSION; }

extern "C" const char* emo_build_info(void) {
  return "libemoportraits_hip gfx950 (CDNA4, wave64) fp32; built " __DATE__ " " __TIME__
         " with hipcc " __clang_version__;
}
