// Library-level entry points: version + thread-local error string.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace tgmx {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace tgmx

extern "C" int tgmx_version(void) { return TGMX_ABI_VERSION; }

extern "C" size_t tgmx_abi_sizeof(int32_t which) {
  switch (which) {
    case 0: return sizeof(tgmx_adj_t);
    case 1: return sizeof(tgmx_recency_step_t);
    case 2: return sizeof(tgmx_tgat_layer_t);
    case 3: return sizeof(tgmx_tgat_model_t);
    case 4: return sizeof(tgmx_tgat_hop_t);
    case 5: return sizeof(tgmx_tgat_layout_t);
    case 6: return sizeof(tgmx_pipeline_t);
    case 7: return sizeof(tgmx_pipeline_out_t);
    case 8: return sizeof(tgmx_dropout_t);
    case 9: return sizeof(tgmx_tgn_memory_fwd_t);
    case 10: return sizeof(tgmx_tconv_fwd_t);
    case 11: return sizeof(tgmx_pipeline_post_t);
    case 12: return sizeof(tgmx_tgn_step_t);
    default: return 0;
  }
}
extern "C" const char* tgmx_last_error(void) { return tgmx::g_err; }

extern "C" int tgmx_event_synchronize(tgmx_event_t ev) {
  TGMX_REQUIRE(ev, "event_synchronize: null event");
  if (hipEventSynchronize((hipEvent_t)ev) != hipSuccess) {
    tgmx::set_error("event_synchronize: hipEventSynchronize failed");
    return TGMX_E_LAUNCH;
  }
  return TGMX_OK;
}

extern "C" int tgmx_event_create(tgmx_event_t* ev) {
  TGMX_REQUIRE(ev, "event_create: null pointer");
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) {
    tgmx::set_error("event_create: hipEventCreate failed");
    return TGMX_E_LAUNCH;
  }
  *ev = (tgmx_event_t)e;
  return TGMX_OK;
}

extern "C" int tgmx_event_create_sync(tgmx_event_t* ev) {
  TGMX_REQUIRE(ev, "event_create_sync: null pointer");
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
    tgmx::set_error("event_create_sync: hipEventCreateWithFlags failed");
    return TGMX_E_LAUNCH;
  }
  *ev = (tgmx_event_t)e;
  return TGMX_OK;
}

extern "C" int tgmx_event_record(tgmx_event_t ev, tgmx_stream_t stream) {
  TGMX_REQUIRE(ev, "event_record: null event");
  if (hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) != hipSuccess) {
    tgmx::set_error("event_record: hipEventRecord failed");
    return TGMX_E_LAUNCH;
  }
  return TGMX_OK;
}

extern "C" int tgmx_stream_wait_event(tgmx_stream_t stream, tgmx_event_t ev) {
  TGMX_REQUIRE(ev, "stream_wait_event: null event");
  if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0) != hipSuccess) {
    tgmx::set_error("stream_wait_event: hipStreamWaitEvent failed");
    return TGMX_E_LAUNCH;
  }
  return TGMX_OK;
}

extern "C" int tgmx_stream_handoff(tgmx_stream_t from, tgmx_stream_t to, tgmx_event_t ev) {
  if (int rc = tgmx_event_record(ev, from)) return rc;
  return tgmx_stream_wait_event(to, ev);
}

extern "C" int tgmx_event_destroy(tgmx_event_t ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
  return TGMX_OK;
}

extern "C" int tgmx_event_elapsed_ms(tgmx_event_t start, tgmx_event_t stop, float* ms) {
  TGMX_REQUIRE(start && stop && ms, "event_elapsed_ms: null pointer");
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess || hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) {
    tgmx::set_error("event_elapsed_ms: events not recorded");
    return TGMX_E_LAUNCH;
  }
  return TGMX_OK;
}
