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
extern "C" int cape_abi_version(void) { return 5; }

__global__ void rng_advance_kernel(uint64_t* st) { st[1] += 1; }
extern "C" int cape_rng_advance(uint64_t* rng_state, cape_stream_t stream) {
  CAPE_REQUIRE(rng_state != nullptr, "cape_rng_advance: null state");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, as_stream(stream), rng_state);
  CAPE_LAUNCH_CHECK("cape_rng_advance");
  return 0;
}
