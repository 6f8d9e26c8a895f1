// Identity of the sources this library was built from (build.py: source_id(), sha256 over csrc/*.hip, csrc/*.h,
// include/sgdm_hip.h and the compile flags).  Recompiled on every change of any of them -- it is a one-line unit.
#include "../../include/sgdm_hip.h"

#ifndef SGDM_BUILD_ID
#define SGDM_BUILD_ID "unidentified"
#endif

extern "C" const char* sgd_build_id(void) { return SGDM_BUILD_ID; }
