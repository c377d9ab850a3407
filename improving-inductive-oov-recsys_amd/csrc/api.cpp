// Library-level entry points of libmi_oov.so: version, error strings.
#include "common.hpp"

namespace mi_oov {
thread_local int g_last_hip_error = 0;
}

extern "C" int mi_oov_version(void) { return MI_OOV_VERSION; }

extern "C" int mi_oov_last_hip_error(void) { return mi_oov::g_last_hip_error; }

extern "C" const char* mi_oov_strerror(int code) {
  switch (code) {
    case MI_OOV_OK: return "ok";
    case MI_OOV_ERR_NULL: return "required pointer is NULL";
    case MI_OOV_ERR_SHAPE: return "invalid or unsupported dimension";
    case MI_OOV_ERR_KIND: return "unknown kind";
    case MI_OOV_ERR_LAUNCH: return "HIP launch failed (see mi_oov_last_hip_error)";
    case MI_OOV_ERR_ALIGN: return "pointer is not aligned as documented";
    case MI_OOV_ERR_WORKSPACE: return "workspace too small";
    case MI_OOV_ERR_ALIAS: return "an output buffer is also an input";
    default: return "unknown error code";
  }
}
