// rpf_impl_d27_mid.hip -- one translation unit of the fused per-pixel kernels: layout d27 (4 random parameters, 18 features,
// __half planes), size-class part 2 (see rpf_filter_impl.inc).  Compiled with -ffp-contract=off like every kernel TU.
#include "rpf_device_common.h"

namespace rpf {
namespace d27 {
namespace {
#define RPF_IMPL_NR 4
#define RPF_IMPL_NF 18
#define RPF_IMPL_PLANE_T __half
#define RPF_IMPL_PART 2
} // namespace
#include "rpf_filter_impl.inc"
} // namespace d27
} // namespace rpf
