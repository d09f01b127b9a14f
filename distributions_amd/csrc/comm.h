// The communicators behind dist_comm_* (include/distributions_hip.h).
//
//   RCCL      bound at run time (the librccl already in the process wins):
//             one process per GPU, the all-reduce on the engine's stream.
//   host      ranks that SHARE a GPU (RCCL refuses two ranks on one device):
//             a POSIX shared-memory segment, the all-reduce staged through
//             the host -- D2H into the rank's slot, barrier, every rank sums
//             the slots in rank order, barrier, H2D.  It drains the stream
//             and is meant for tests and single-GPU rehearsals of the
//             multi-rank protocol, not for speed.  Unlike RCCL it CHECKS what
//             it is asked: every rank posts (count, type, op, serial) and a
//             mismatch, or a peer that does not arrive within
//             DIST_COMM_TIMEOUT_S seconds (default 120), fails the call on
//             every rank instead of hanging or exchanging garbage.
#pragma once

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>   // declarations only: RCCL is bound with dlopen

#include "common.h"

namespace dist {

enum CommType { COMM_I32 = 0, COMM_F64 = 1 };
enum CommOp { COMM_SUM = 0, COMM_MIN = 1 };

// ---- RCCL, bound at run time ----------------------------------------------
struct Rccl {
    void * handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
};
inline Rccl & rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy that is already in the process wins (PyTorch ships its own)
        const char * names[] = {"librccl.so.1", "librccl.so",
                                "/opt/rocm/lib/librccl.so.1"};
        for (const char * name : names) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (r.handle) break;
        }
        for (const char * name : names) {
            if (r.handle) break;
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.handle) return;
#define DIST_SYM(field, symbol)                                              \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, symbol))
        DIST_SYM(get_unique_id, "ncclGetUniqueId");
        DIST_SYM(comm_init_rank, "ncclCommInitRank");
        DIST_SYM(comm_destroy, "ncclCommDestroy");
        DIST_SYM(all_reduce, "ncclAllReduce");
        DIST_SYM(error_string, "ncclGetErrorString");
#undef DIST_SYM
        r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy
            && r.all_reduce && r.error_string;
    });
    return r;
}
#define RCCL_CHECK(expr)                                                     \
    do {                                                                     \
        ncclResult_t rc_ = (expr);                                           \
        if (rc_ != ncclSuccess)                                              \
            throw ::dist::Error(std::string("RCCL error: ")                  \
                                + ::dist::rccl().error_string(rc_)           \
                                + " at " #expr);                             \
    } while (0)

// ---- the host transport -----------------------------------------------------
constexpr char kHostIdMagic[8] = {'D', 'I', 'S', 'T', 'H', 'O', 'S', 'T'};
constexpr int kHostMaxWorld = 64;

struct HostSegment {
    std::atomic<uint32_t> ready;        // rank 0 has initialised the header
    uint32_t world;
    uint64_t slot_bytes;
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> failed;       // a rank gave up: all later calls fail
    struct Desc {
        uint64_t count, serial;
        uint32_t type, op;
    } desc[kHostMaxWorld];
    // the ranks' slots follow, 4096-byte aligned
};

class HostComm {
public:
    HostComm(const uint8_t id[128], int rank_, int world_)
        : rank(rank_), world(world_) {
        DIST_REQUIRE(world <= kHostMaxWorld, "host transport: at most 64 ranks");
        name = "/distcomm-";
        for (int i = 8; i < 8 + 24 && id[i]; ++i) name += (char)id[i];
        const char * t = getenv("DIST_COMM_TIMEOUT_S");
        timeout_s = t ? atof(t) : 120.0;
        if (timeout_s <= 0.0) timeout_s = 120.0;
        const char * sb = getenv("DIST_COMM_SLOT_BYTES");
        const size_t slot = sb ? (size_t)atoll(sb) : ((size_t)4 << 20);
        slot_bytes = (std::max<size_t>(slot, 4096) + 4095) & ~(size_t)4095;
        bytes = header_bytes() + (size_t)world * slot_bytes;
        const double deadline = now() + std::max(timeout_s, 300.0);
        int fd = -1;
        if (rank == 0) {
            fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            DIST_REQUIRE(fd >= 0, "host transport: cannot create " + name);
            if (ftruncate(fd, (off_t)bytes) != 0) {
                close(fd);
                shm_unlink(name.c_str());
                throw Error("ERROR host transport: cannot size " + name);
            }
        } else {
            // (the segment appears when rank 0 has made it, at full size)
            for (;;) {
                fd = shm_open(name.c_str(), O_RDWR, 0600);
                if (fd >= 0) {
                    struct stat st;
                    if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes)
                        break;
                    close(fd);
                    fd = -1;
                }
                DIST_REQUIRE(now() < deadline,
                             "host transport: rank 0 never created " + name);
                nap();
            }
        }
        void * m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED,
                        fd, 0);
        close(fd);
        if (m == MAP_FAILED) {
            if (rank == 0) shm_unlink(name.c_str());
            throw Error("ERROR host transport: cannot map " + name);
        }
        seg = static_cast<HostSegment *>(m);
        if (rank == 0) {
            seg->world = (uint32_t)world;
            seg->slot_bytes = slot_bytes;
            seg->arrived.store(0);
            seg->generation.store(0);
            seg->failed.store(0);
            seg->ready.store(1, std::memory_order_release);
        } else {
            while (!seg->ready.load(std::memory_order_acquire)) {
                if (now() >= deadline) {
                    munmap(seg, bytes);
                    throw Error("ERROR host transport: rank 0 never finished "
                                + name);
                }
                nap();
            }
            if (seg->world != (uint32_t)world || seg->slot_bytes != slot_bytes) {
                seg->failed.store(1);
                munmap(seg, bytes);
                throw Error("ERROR host transport: ranks disagree on the "
                            "world size or the slot size");
            }
        }
        // every rank has the segment mapped: its name can go
        try {
            barrier(std::max(timeout_s, 300.0));
        } catch (...) {
            if (rank == 0) shm_unlink(name.c_str());
            munmap(seg, bytes);
            throw;
        }
        if (rank == 0) shm_unlink(name.c_str());
    }
    ~HostComm() {
        if (seg) munmap(seg, bytes);
    }
    HostComm(const HostComm &) = delete;
    HostComm & operator=(const HostComm &) = delete;

    void all_reduce(void * dev, size_t count, CommType type, CommOp op,
                    hipStream_t s) {
        DIST_REQUIRE(!seg->failed.load(),
                     "host transport: the communicator failed earlier "
                     "(a rank diverged or did not arrive)");
        HIP_CHECK(hipStreamSynchronize(s));
        HostSegment::Desc mine{(uint64_t)count, serial, (uint32_t)type,
                               (uint32_t)op};
        seg->desc[rank] = mine;
        barrier(timeout_s);
        for (int r = 0; r < world; ++r) {
            const HostSegment::Desc d = seg->desc[r];
            if (d.count != mine.count || d.type != mine.type || d.op != mine.op
                || d.serial != mine.serial) {
                seg->failed.store(1);
                throw Error(
                    "ERROR ranks diverged: rank " + std::to_string(rank)
                    + " issues collective #" + std::to_string(mine.serial)
                    + " of " + std::to_string(mine.count) + " words (type "
                    + std::to_string(mine.type) + ", op "
                    + std::to_string(mine.op) + "), rank " + std::to_string(r)
                    + " #" + std::to_string(d.serial) + " of "
                    + std::to_string(d.count) + " words (type "
                    + std::to_string(d.type) + ", op " + std::to_string(d.op)
                    + ")");
            }
        }
        serial += 1;
        const size_t elem = type == COMM_F64 ? 8 : 4;
        const size_t per = slot_bytes / elem;
        acc.resize(slot_bytes);
        // (at least one round, even of nothing: no rank may post its next
        // collective while a peer still reads this one's)
        for (size_t off = 0; off < count || off == 0; off += per) {
            const size_t n = std::min(per, count - off);
            char * at = static_cast<char *>(dev) + off * elem;
            HIP_CHECK(hipMemcpy(slot(rank), at, n * elem,
                                hipMemcpyDeviceToHost));
            barrier(timeout_s);
            if (type == COMM_F64) reduce<double>(n, op);
            else reduce<int32_t>(n, op);
            barrier(timeout_s);   // (all have read: the slots may be rewritten)
            HIP_CHECK(hipMemcpy(at, acc.data(), n * elem,
                                hipMemcpyHostToDevice));
        }
    }
    uint64_t collectives() const { return serial; }

private:
    int rank, world;
    std::string name;
    double timeout_s = 120.0;
    size_t slot_bytes = 0, bytes = 0;
    HostSegment * seg = nullptr;
    uint64_t serial = 0;
    std::vector<char> acc;

    static size_t header_bytes() {
        return (sizeof(HostSegment) + 4095) & ~(size_t)4095;
    }
    char * slot(int r) const {
        return reinterpret_cast<char *>(seg) + header_bytes()
               + (size_t)r * slot_bytes;
    }
    static double now() {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
    static void nap() {
        struct timespec ts = {0, 50000};   // 50 us
        nanosleep(&ts, nullptr);
    }
    // summed in rank order by every rank: the same bits everywhere
    template <class T>
    void reduce(size_t n, CommOp op) {
        T * out = reinterpret_cast<T *>(acc.data());
        const T * first = reinterpret_cast<const T *>(slot(0));
        for (size_t i = 0; i < n; ++i) out[i] = first[i];
        for (int r = 1; r < world; ++r) {
            const T * in = reinterpret_cast<const T *>(slot(r));
            if (op == COMM_MIN) {
                for (size_t i = 0; i < n; ++i)
                    out[i] = in[i] < out[i] ? in[i] : out[i];
            } else {
                for (size_t i = 0; i < n; ++i) out[i] += in[i];
            }
        }
    }
    void barrier(double limit_s) {
        const uint32_t gen = seg->generation.load(std::memory_order_acquire);
        if (seg->arrived.fetch_add(1, std::memory_order_acq_rel) + 1
            == (uint32_t)world) {
            seg->arrived.store(0, std::memory_order_relaxed);
            seg->generation.store(gen + 1, std::memory_order_release);
            return;
        }
        const double deadline = now() + limit_s;
        unsigned spins = 0;
        while (seg->generation.load(std::memory_order_acquire) == gen) {
            if (seg->failed.load(std::memory_order_relaxed))
                throw Error("ERROR ranks diverged: a peer gave up on the "
                            "collective (see its error)");
            if (++spins < 2000) {
                sched_yield();
            } else {
                nap();
                if (now() >= deadline) {
                    seg->failed.store(1);
                    throw Error(
                        "ERROR ranks diverged: rank " + std::to_string(rank)
                        + " waited " + std::to_string((int)limit_s)
                        + " s at collective #" + std::to_string(serial)
                        + " for a peer that never arrived");
                }
            }
        }
    }
};

}  // namespace dist

struct dist_comm {
    ncclComm_t comm = nullptr;              // RCCL ...
    std::unique_ptr<dist::HostComm> host;   // ... or the host transport
    int rank = 0, world = 1;
    void all_reduce(void * dev, size_t count, dist::CommType type,
                    dist::CommOp op, hipStream_t s) {
        if (host) {
            host->all_reduce(dev, count, type, op, s);
            return;
        }
        RCCL_CHECK(dist::rccl().all_reduce(
            dev, dev, count, type == dist::COMM_F64 ? ncclDouble : ncclInt32,
            op == dist::COMM_MIN ? ncclMin : ncclSum, comm, s));
    }
    bool valid() const { return comm != nullptr || host != nullptr; }
};
