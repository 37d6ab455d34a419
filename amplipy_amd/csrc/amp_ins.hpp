// amp_ins.hpp -- internal interface of amp_ins.hip (on-device aggregation of insertion events)
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/amplihip.h"

namespace amp {

// bytes of device scratch ins_aggregate needs for n_slots event-list slots
size_t ins_scratch_bytes(int64_t n_slots);

// Sorts and run-length encodes the events held in the 8 shard regions of `ev` (region s: shard_n[s] slots of `cap`; slots
// reserved and not used carry ref_pos = -1).  d_runs: room for one record per slot.  Returns 0 or a hipError_t / -1;
// *n_events = real events, *n_runs = records written (device memory d_runs[0 .. *n_runs)).  Synchronises the stream.
int ins_aggregate(hipStream_t s, const amp_dev_reads &rd, uint64_t read_base, const amp_ins_event *ev, long long cap, const unsigned long long *shard_n,
                  void *scratch, amp_ins_run *d_runs, int64_t *n_events, int64_t *n_runs);

}  // namespace amp
