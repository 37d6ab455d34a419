// amp_tile.hpp -- tile kernel (variant 2). Placeholder until the fused kernel lands.
#pragma once
#include "amp_read.hpp"
namespace amp {
static inline int tile_launch(const KParams &, const amp_dev_reads &, uint64_t, const DevOut &, uint32_t *, uint32_t *,
                              amp_ins_event *, unsigned long long *, long long, int, hipStream_t) {
    return (int)hipErrorNotSupported;
}
}  // namespace amp
