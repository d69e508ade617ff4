#include "common.h"

extern "C" int32_t u3d_version(void) { return 1; }

extern "C" const char* u3d_strerror(int32_t code) {
  switch (code) {
    case U3D_OK: return "ok";
    case U3D_ERR_ARG: return "bad argument";
    case U3D_ERR_UNSUPPORTED: return "unsupported shape/dtype";
    case U3D_ERR_LAUNCH: return "kernel launch failed";
    case U3D_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}
