// Wavefront (queue-based) pipeline.  Placeholder until the queue kernels land: requesting
// it reports "unsupported" instead of silently running something else.
#include "pt_kernels.hpp"
#include "pt_host.hpp"

namespace rtx {

int wavefront_render(WavefrontPool&, const DScene<float>&, const DCamera<float>&, const DRenderParams&, float*, DCounters*, int, bool,
                     hipStream_t, uint32_t&, std::string& err) {
  err = "not built into this library";
  return RTX_ERR_UNSUPPORTED;
}

}  // namespace rtx
