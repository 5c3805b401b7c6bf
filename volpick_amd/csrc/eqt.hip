#include "net.h"
namespace vp {
int plan_eqt(Net& net, const ParamView& pv) {
  (void)net; (void)pv;
  set_error("EQTransformer plan not built yet");
  return VP_ERR_UNSUPPORTED;
}
}  // namespace vp
