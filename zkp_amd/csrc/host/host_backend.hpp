// Host execution of the engine entry points the host-transcript route of the toolbox uses (host_backend.cpp): the device headers
// fe25519.h / ge25519.h compiled for the host.  Signatures = zkp_mi355x.h (1), (2), (3) without the context.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../../include/zkp_mi355x.h"

namespace zkp {
namespace hostbk {
int msm_many(uint32_t n_msm, const uint32_t* off, const uint8_t* scalars, const uint32_t* pidx, const uint8_t* points, uint32_t n_points, int flags,
             uint8_t* out, uint8_t* status);
int msm_optional(uint64_t n, const uint8_t* scalars, const uint8_t* points, uint8_t out_point[32], int* status);
int decode_check(uint64_t n, const uint8_t* points, uint8_t* status);
}  // namespace hostbk
}  // namespace zkp
