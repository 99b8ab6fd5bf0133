// pire_hip_build_info(): which compiler this library's kernels were audited with (tools/audit/build_audit.py).
// The product build (make, the default target) compiles this file AFTER the ISA audits of every kernel that keeps text
// on its way in registers have passed and with their summary in build/build_info.h; the link needs this object, so a
// library whose kernels fail an audit is not linked.  Every other build of the same sources (sanitizers, tuning,
// experiments) says that it is not that.
#include "../../include/pire_hip.h"

#ifdef PIRE_HIP_AUDITED
#include "build/build_info.h"
#else
#define PIRE_HIP_BUILD_INFO "libpire_hip: NOT the audited product build (a sanitizer / tuning / experiment library of the same sources)"
#endif

extern "C" const char* pire_hip_build_info(void) { return PIRE_HIP_BUILD_INFO; }
