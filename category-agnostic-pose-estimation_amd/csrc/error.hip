// error.hip -- thread-local error string + ABI version.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

int cape_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

extern "C" const char* cape_last_error(void) { return g_err; }
extern "C" int cape_abi_version(void) { return 11; }

__global__ void rng_advance_kernel(uint64_t* st) { st[1] += 1; }
extern "C" int cape_rng_advance(uint64_t* rng_state, cape_stream_t stream) {
  CAPE_REQUIRE(rng_state != nullptr, "cape_rng_advance: null state");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, as_stream(stream), rng_state);
  CAPE_LAUNCH_CHECK("cape_rng_advance");
  return 0;
}

// ---- two-stream fork / join without a host round trip through a framework: `side` continues after everything enqueued on
// `main` so far (fork) / `main` after everything on `side` (join).  Events come from a small pool (an event may be re-recorded
// while an earlier wait on it is still pending: the wait captured the record it followed).  Works inside a stream capture:
// the record / wait pair becomes a graph edge.
static hipEvent_t g_ev[64];
static int g_ev_next = 0, g_ev_init = 0;
static int order_after(hipStream_t first, hipStream_t then, const char* what) {
  if (!g_ev_init) {
    for (int i = 0; i < 64; ++i) {
      hipError_t e = hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming);
      if (e != hipSuccess) return cape_set_error("%s: hipEventCreate: %s", what, hipGetErrorString(e));
    }
    g_ev_init = 1;
  }
  hipEvent_t ev = g_ev[g_ev_next];
  g_ev_next = (g_ev_next + 1) & 63;
  hipError_t e = hipEventRecord(ev, first);
  if (e == hipSuccess) e = hipStreamWaitEvent(then, ev, 0);
  if (e != hipSuccess) return cape_set_error("%s: %s", what, hipGetErrorString(e));
  return 0;
}
extern "C" int cape_stream_fork(cape_stream_t main_stream, cape_stream_t side_stream) {
  return order_after(as_stream(main_stream), as_stream(side_stream), "cape_stream_fork");
}
extern "C" int cape_stream_join(cape_stream_t main_stream, cape_stream_t side_stream) {
  return order_after(as_stream(side_stream), as_stream(main_stream), "cape_stream_join");
}
