// host_comm.cpp — see host_comm.hpp.
#include "host_comm.hpp"

#include "dlco_internal.hpp"

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>

namespace dlco {

struct HostComm::Header {
    std::atomic<unsigned> attached;      // ranks that have mapped the segment
    std::atomic<unsigned> arrived;       // barrier: ranks that reached the current generation
    std::atomic<unsigned> generation;    // barrier generation
    std::atomic<unsigned> failed;        // a rank that gives up sets this so that the others stop waiting
    char pad[64 - 4 * sizeof(std::atomic<unsigned>)];
};

namespace {
constexpr double kTimeoutSeconds = 300.0;
}

HostComm::HostComm(const std::string &name, int rank, int world, size_t slot_bytes)
    : slot_bytes_((slot_bytes + 63) & ~(size_t)63), rank_(rank), world_(world)
{
    static_assert(sizeof(Header) == 64, "header is one cache line");
    map_bytes_ = sizeof(Header) + (size_t)world * slot_bytes_;
    const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) throw Error(-6, "host transport: shm_open(" + name + ") failed: " + std::strerror(errno));
    if (ftruncate(fd, (off_t)map_bytes_) != 0) {               // every rank asks for the same size: idempotent
        close(fd);
        throw Error(-6, "host transport: ftruncate failed: " + std::string(std::strerror(errno)));
    }
    void *p = mmap(nullptr, map_bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) throw Error(-6, "host transport: mmap failed: " + std::string(std::strerror(errno)));
    hdr_ = static_cast<Header *>(p);
    data_ = static_cast<char *>(p) + sizeof(Header);
    if (hdr_->attached.fetch_add(1) + 1 == (unsigned)world) shm_unlink(name.c_str());   // last one in: the name can go
    barrier();                                                  // nobody proceeds before everyone is attached
}

HostComm::~HostComm()
{
    if (hdr_) {
        hdr_->failed.store(1);                                  // whoever still waits for this rank stops waiting
        munmap(hdr_, map_bytes_);
    }
    if (sum_) (void)hipHostFree(sum_);
}

void HostComm::barrier()
{
    const unsigned my_gen = gen_++;
    if (hdr_->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (unsigned)world_) {
        hdr_->arrived.store(0, std::memory_order_relaxed);
        hdr_->generation.store(my_gen + 1, std::memory_order_release);
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (hdr_->generation.load(std::memory_order_acquire) == my_gen) {
        if (++spins % 64 == 0) {
            sched_yield();
            if (hdr_->failed.load(std::memory_order_acquire)) {
                if (hdr_->generation.load(std::memory_order_acquire) != my_gen) break;   // it left after this barrier opened
                throw Error(-6, "host transport: another rank left the run");
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutSeconds) {
                hdr_->failed.store(1);
                throw Error(-6, "host transport: timed out waiting for the other ranks");
            }
        }
    }
}

void HostComm::allgather_inplace(void *buf, size_t bytes_per_rank, hipStream_t s)
{
    DLCO_CHECK(bytes_per_rank <= slot_bytes_, -6, "host transport: message larger than the segment's slots");
    char *dev = static_cast<char *>(buf);
    DLCO_HIP(hipMemcpyAsync(data_ + (size_t)rank_ * slot_bytes_, dev + (size_t)rank_ * bytes_per_rank, bytes_per_rank,
                            hipMemcpyDeviceToHost, s));
    DLCO_HIP(hipStreamSynchronize(s));
    barrier();                                                  // every slot is filled
    for (int g = 0; g < world_; g++) {
        if (g == rank_) continue;
        DLCO_HIP(hipMemcpyAsync(dev + (size_t)g * bytes_per_rank, data_ + (size_t)g * slot_bytes_, bytes_per_rank,
                                hipMemcpyHostToDevice, s));
    }
    DLCO_HIP(hipStreamSynchronize(s));
    barrier();                                                  // every slot has been read: it may be refilled
}

void HostComm::allreduce_sum_f32(float *buf, size_t count, hipStream_t s)
{
    const size_t piece = slot_bytes_ / sizeof(float);
    DLCO_CHECK(piece > 0, -6, "host transport: empty slots");
    if (!sum_) DLCO_HIP(hipHostMalloc((void **)&sum_, slot_bytes_));
    for (size_t o = 0; o < count; o += piece) {
        const size_t n = std::min(piece, count - o);
        DLCO_HIP(hipMemcpyAsync(data_ + (size_t)rank_ * slot_bytes_, buf + o, n * sizeof(float), hipMemcpyDeviceToHost, s));
        DLCO_HIP(hipStreamSynchronize(s));
        barrier();                                              // every rank's piece is in its slot
        const float *first = reinterpret_cast<const float *>(data_);
        for (size_t i = 0; i < n; i++) sum_[i] = first[i];
        for (int g = 1; g < world_; g++) {
            const float *src = reinterpret_cast<const float *>(data_ + (size_t)g * slot_bytes_);
            for (size_t i = 0; i < n; i++) sum_[i] += src[i];
        }
        DLCO_HIP(hipMemcpyAsync(buf + o, sum_, n * sizeof(float), hipMemcpyHostToDevice, s));
        DLCO_HIP(hipStreamSynchronize(s));
        barrier();                                              // every slot has been read: it may be refilled
    }
}

}  // namespace dlco
