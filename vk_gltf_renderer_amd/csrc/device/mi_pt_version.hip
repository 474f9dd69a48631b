// Build identity of libmi_pt.so, in a translation unit of its own: the id changes with every edit of any device source, and this
// is the only file that has to be recompiled for it (csrc/Makefile builds one object per source).
#include "mi_pt.h"

#ifndef MI_PT_SRC_ID
#define MI_PT_SRC_ID "unknown"
#endif
#ifndef MI_PT_GIT_ID
#define MI_PT_GIT_ID "nogit"
#endif

extern "C" const char* mi_pt_version(void)
{
  // src = sha1 of the device sources + public headers this binary was compiled from (csrc/Makefile), git = HEAD at build time
  return "mi_pt 0.3 (gfx950 wavefront path tracer) src=" MI_PT_SRC_ID " git=" MI_PT_GIT_ID;
}

extern "C" int mi_pt_abi_version(void) { return MI_PT_ABI_VERSION; }
