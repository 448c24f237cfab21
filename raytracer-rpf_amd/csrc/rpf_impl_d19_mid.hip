// rpf_impl_d19_mid.hip -- one translation unit of the fused per-pixel kernels: layout d19 (2 random parameters, 12 features,
// float planes), size-class part 2 (see rpf_filter_impl.inc).  Compiled with -ffp-contract=off like every kernel TU.
#include "rpf_device_common.h"

namespace rpf {
namespace d19 {
namespace {
#define RPF_IMPL_NR 2
#define RPF_IMPL_NF 12
#define RPF_IMPL_PLANE_T float
#define RPF_IMPL_PART 2
} // namespace
#include "rpf_filter_impl.inc"
} // namespace d19
} // namespace rpf
