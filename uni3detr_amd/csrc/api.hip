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

extern "C" int32_t u3d_event_create(void** event) {
  U3D_REQUIRE(event, U3D_ERR_ARG);
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return U3D_ERR_LAUNCH;
  *event = (void*)e;
  return U3D_OK;
}
extern "C" int32_t u3d_event_record(void* event, int32_t external, u3d_stream s) {
  U3D_REQUIRE(event, U3D_ERR_ARG);
  if (external) {
    // hipEventRecordWithFlags(.., hipEventRecordExternal) is refused during capture on ROCm 7.2: add the event-record node by hand
    // behind the stream's current capture dependencies and make it the new dependency
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    if (hipStreamGetCaptureInfo_v2(s, &st, &id, &graph, &deps, &ndeps) != hipSuccess) return U3D_ERR_LAUNCH;
    if (st == hipStreamCaptureStatusActive) {
      hipGraphNode_t node;
      if (hipGraphAddEventRecordNode(&node, graph, deps, ndeps, (hipEvent_t)event) != hipSuccess) return U3D_ERR_LAUNCH;
      if (hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies) != hipSuccess) return U3D_ERR_LAUNCH;
      return U3D_OK;
    }
  }
  return hipEventRecord((hipEvent_t)event, s) == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
}
extern "C" int32_t u3d_event_elapsed_ms(void* start, void* stop, float* ms) {
  U3D_REQUIRE(start && stop && ms, U3D_ERR_ARG);
  return hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
}
extern "C" int32_t u3d_event_destroy(void* event) {
  U3D_REQUIRE(event, U3D_ERR_ARG);
  return hipEventDestroy((hipEvent_t)event) == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
}
