// alac_encode_v1_d24.hip — the 24-bit instantiations of the tap-parallel encode pipeline (one translation unit per bit
// depth: the build compiles them side by side).
#include "alac_encode_v1_impl.hpp"

namespace alacdev {
template void launch_v1_typed<24, 1>(const V1Args &, uint32_t, uint32_t, hipStream_t, hipEvent_t *, const PackArgs &, const V1Streams &, const AlacOptions &);
template void launch_v1_typed<24, 2>(const V1Args &, uint32_t, uint32_t, hipStream_t, hipEvent_t *, const PackArgs &, const V1Streams &, const AlacOptions &);
}  // namespace alacdev
