// pending_api.hip - entry points declared in include/dmpc.h whose kernels are not written yet.
// They fail loudly (DMPC_E_UNSUPPORTED); nothing falls back to a CPU path.
#include "../../include/dmpc.h"

extern "C" {

int dmpc_pnqp(int, int, const float *, const float *, const float *, const float *, const float *, int,
              float *, float *, int32_t *, float *, int32_t *, int32_t *, dmpc_stream_t) {
  return DMPC_E_UNSUPPORTED;
}
size_t dmpc_mpc_step_workspace_bytes(int, int, int, int) { return 0; }
int dmpc_mpc_step_forward(int, int, int, int, const float *, const float *, const float *, const float *,
                          const float *, const float *, const float *, const float *, const float *,
                          const float *, const float *, const float *, int, float, int, int, float *, float *,
                          float *, float *, float *, float *, float *, float *, int32_t *, int32_t *, void *,
                          size_t, int32_t *, dmpc_stream_t) {
  return DMPC_E_UNSUPPORTED;
}
int dmpc_mpc_step_backward(int, int, int, int, const float *, const float *, const float *, const float *,
                           const float *, const float *, const float *, const float *, const float *, float *,
                           float *, float *, float *, float *, void *, size_t, int32_t *, dmpc_stream_t) {
  return DMPC_E_UNSUPPORTED;
}
}
