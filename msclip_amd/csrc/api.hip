// Library introspection entry points (no GPU work).
#include "../../include/msclip_hip.h"

extern "C" int msclip_abi_version(void) { return 1; }
extern "C" const char* msclip_build_arch(void) { return "gfx950"; }
