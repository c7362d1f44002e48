#include "ipdm_common.h"
extern "C" int ipdm_abi_version(void) { return IPDM_ABI_VERSION; }
extern "C" const char* ipdm_build_arch(void) { return "gfx950"; }
