/* gt4hip_nway_body.h -- the N-way tile kernel, its partition kernels and the host orchestration of one call, compiled
 * TWICE by gt4hip_nway.hip: GT4_KM = 8 (up to eight lists per launch: 64-record slots, samples every 128 records -- the
 * geometry of rounds 3 and 4, unchanged) and GT4_KM = 32 (nine to thirty-two lists in ONE pass, round 5: 32-record
 * half-slots, samples every 64 records -- glistmaker's collation width, reference src/glistmaker.c:787-835).
 * No include guard on purpose.  See gt4hip_nway.hip for what the code restates. */
namespace gt4 {

namespace {

namespace GT4_KM_NS {

#include "gt4hip_nway_part.h"
#include "gt4hip_nway_tile.h"
#include "gt4hip_nway_host.h"

}  // namespace GT4_KM_NS

}  // namespace

}  // namespace gt4
