// host_comm.hpp — fallback transport for the all-gathers of a sharded context (cfg.shard) when RCCL
// cannot be used: ranks that are processes of ONE node exchange through a POSIX shared-memory segment
// (device -> segment -> device, two barriers per all-gather on counters inside the segment).  It exists
// so that the multi-process path of `pj-learn -gpus N` can run, and be tested, where librccl is absent
// or where several ranks have to share one GPU (RCCL refuses two ranks on one device); it is correct
// and deterministic but host-staged, i.e. slow — the production transport is rccl_comm.hpp.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace dlco {

class HostComm {
public:
    // Every rank opens (the first one creates) the segment `name`; `slot_bytes` = largest per-rank
    // message.  The segment must be fresh (all zeros); it is unlinked once every rank has attached.
    HostComm(const std::string &name, int rank, int world, size_t slot_bytes);
    ~HostComm();
    // in-place all-gather of the device buffer buf viewed as [world][bytes_per_rank], ordered after the
    // work queued on s and complete (on the device) when it returns
    void allgather_inplace(void *buf, size_t bytes_per_rank, hipStream_t s);
    // in-place sum all-reduce of `count` floats of the device buffer: slot-sized pieces go device -> segment, every rank
    // adds the pieces of all ranks in rank order on the host (the same bits everywhere) and uploads the sum
    void allreduce_sum_f32(float *buf, size_t count, hipStream_t s);

private:
    void barrier();
    struct Header;
    Header *hdr_ = nullptr;
    char *data_ = nullptr;
    size_t map_bytes_ = 0, slot_bytes_ = 0;
    int rank_ = 0, world_ = 1;
    unsigned gen_ = 0;
    float *sum_ = nullptr;               // pinned staging of one reduced piece
};

}  // namespace dlco
