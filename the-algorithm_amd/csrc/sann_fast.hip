// sann_fast.hip -- LDS fast path of the (query, partition) unit.  (placeholder: not enabled yet)
#include <hip/hip_runtime.h>
#include "sann_kernels.h"
namespace sann {
hipError_t launch_unit_fast(const IndexView &, const BatchView &, const FastParams &, int, hipStream_t) {
  return hipErrorInvalidValue;
}
}  // namespace sann
