// Library identity and error plumbing of libupflow_hip.so.
#include "common.hpp"

namespace upf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return (int)e > 0 ? (int)e : 1;
  }
  return UPF_OK;
}

}  // namespace upf

extern "C" const char* upf_version(void) { return "upflow_hip 0.1.0 gfx950"; }
extern "C" const char* upf_last_error(void) { return upf::g_err; }
