// TEST DOUBLE, not a product path.  A stand-in for librccl.so that lets TWO (or more) processes sharing ONE GPU run the
// engine's native data-parallel driver (isk_dp_run): real RCCL rejects two ranks on one device ("Duplicate GPU
// detected"), and the build loop has single-GPU boxes only.  It implements exactly the entry points the engine
// binds (engine_dp_native.hip rccl_load: six, plus the optional group pair and ncclCommInitAll) with host staging through a POSIX shared-memory segment: every collective drains the
// stream, copies the operand to the rank's slot, meets the other ranks at a barrier, reduces / concatenates in RANK ORDER
// (so every rank computes the same bits) and copies the result back.  Slow and synchronous by design -- what it checks
// is the engine's collective SEQUENCE, counts, offsets and shard arithmetic at world > 1, and that no rank leaves the
// others waiting (the barrier gives up after ISK_MOCK_RCCL_TIMEOUT_S (60) seconds and the collective returns an error instead of hanging).
// Selected with ISK_RCCL_LIB=<path> (tests/test_gpu_dp_native2.py).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
constexpr size_t SLOT_BYTES = 16u << 20;
int mock_timeout_s() {
    static const int t = [] { const char* v = getenv("ISK_MOCK_RCCL_TIMEOUT_S"); return v && atoi(v) > 0 ? atoi(v) : 60; }();
    return t;
}

struct Header {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<long long> ncalls;      // collectives completed (rank 0 counts): the test reads it back
    std::atomic<int> broken;            // sticky: a barrier timed out -- every later collective on this communicator fails
    std::atomic<int> ready;             // rank 0 has created and zeroed the segment (the other ranks wait for it at init)
};

struct MockComm {
    int rank, world;
    char name[64];
    uint8_t* base;
    size_t bytes;
    Header* hdr() const { return reinterpret_cast<Header*>(base); }
    uint8_t* slot(int r) const { return base + 4096 + (size_t)r * SLOT_BYTES; }
};

bool barrier(MockComm* c) {
    Header* h = c->hdr();
    if (h->broken.load(std::memory_order_acquire)) return false;
    const int gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
        h->arrived.store(0, std::memory_order_relaxed);
        h->generation.fetch_add(1, std::memory_order_release);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->generation.load(std::memory_order_acquire) == gen) {
        std::this_thread::yield();
        if (h->broken.load(std::memory_order_acquire)) return false;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(mock_timeout_s())) {
            // the arrival count is now meaningless (a later barrier could release with a rank missing): the communicator is
            // dead for everybody, as a real one is after its watchdog fires
            h->broken.store(1, std::memory_order_release);
            return false;
        }
    }
    return true;
}

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclFloat: return 4;
        case ncclInt64: return 8;
        case ncclDouble: return 8;
        default: return 0;
    }
}

template <typename T>
void reduce_into(T* acc, const T* x, size_t n, ncclRedOp_t op) {
    if (op == ncclSum) for (size_t i = 0; i < n; ++i) acc[i] += x[i];
    else for (size_t i = 0; i < n; ++i) acc[i] = x[i] > acc[i] ? x[i] : acc[i];
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "/iskmock_%d_%lld", (int)getpid(),
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

// rank 0 creates the segment afresh (a name left behind by a crashed run is unlinked first, O_EXCL), sizes it -- new pages are
// zero -- and raises `ready`; the other ranks open it without O_CREAT and wait for that flag
static MockComm* attach(const char* name, int world, int rank, bool create) {
    MockComm* c = new MockComm();
    c->rank = rank; c->world = world;
    std::snprintf(c->name, sizeof(c->name), "%s", name);
    c->bytes = 4096 + (size_t)world * SLOT_BYTES;
    int fd = -1;
    if (create) {
        shm_unlink(c->name);
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { if (fd >= 0) close(fd); delete c; return nullptr; }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        struct stat sb;
        while ((fd = shm_open(c->name, O_RDWR, 0600)) < 0 || fstat(fd, &sb) != 0 || (size_t)sb.st_size < c->bytes) {
            if (fd >= 0) { close(fd); fd = -1; }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(mock_timeout_s())) { delete c; return nullptr; }
            std::this_thread::yield();
        }
    }
    c->base = (uint8_t*)mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->base == MAP_FAILED) { delete c; return nullptr; }
    if (create) c->hdr()->ready.store(1, std::memory_order_release);
    else {
        const auto t0 = std::chrono::steady_clock::now();
        while (!c->hdr()->ready.load(std::memory_order_acquire)) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(mock_timeout_s())) { munmap(c->base, c->bytes); delete c; return nullptr; }
            std::this_thread::yield();
        }
    }
    return c;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
    if (!comm || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
    MockComm* c = attach(id.internal, world, rank, rank == 0);
    if (!c) return ncclSystemError;
    if (!barrier(c)) { munmap(c->base, c->bytes); delete c; return ncclInternalError; }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

// single-process form (isk_dp_create_all): every rank's communicator at once, the engines then call their collectives
// from one host thread each.  The device list is not looked at (the point of the mock: several ranks on ONE GPU).
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    (void)devlist;
    if (!comms || ndev < 1) return ncclInvalidArgument;
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    for (int r = 0; r < ndev; ++r) {
        MockComm* c = attach(id.internal, ndev, r, r == 0);
        if (!c) return ncclSystemError;
        comms[r] = reinterpret_cast<ncclComm_t>(c);
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    if (!c) return ncclSuccess;
    if (const char* out = getenv("ISK_MOCK_RCCL_COUNT_FILE")) {
        if (c->rank == 0) {
            if (FILE* f = std::fopen(out, "w")) { std::fprintf(f, "%lld\n", c->hdr()->ncalls.load()); std::fclose(f); }
        }
    }
    munmap(c->base, c->bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    const size_t ts = type_size(dt);
    if (!c || !ts || (op != ncclSum && op != ncclMax) || count * ts > SLOT_BYTES) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->slot(c->rank), send, count * ts, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclInternalError;
    std::vector<uint8_t> acc(c->slot(0), c->slot(0) + count * ts);
    for (int r = 1; r < c->world; ++r) {
        if (dt == ncclFloat) reduce_into((float*)acc.data(), (const float*)c->slot(r), count, op);
        else if (dt == ncclDouble) reduce_into((double*)acc.data(), (const double*)c->slot(r), count, op);
        else reduce_into((long long*)acc.data(), (const long long*)c->slot(r), count, op);
    }
    if (!barrier(c)) return ncclInternalError;           // nobody overwrites a slot before every rank has read it
    if (c->rank == 0) c->hdr()->ncalls.fetch_add(1);
    if (hipMemcpy(recv, acc.data(), count * ts, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    const size_t ts = type_size(dt);
    if (!c || !ts || sendcount * ts > SLOT_BYTES) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->slot(c->rank), send, sendcount * ts, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclInternalError;
    for (int r = 0; r < c->world; ++r)
        if (hipMemcpy((uint8_t*)recv + (size_t)r * sendcount * ts, c->slot(r), sendcount * ts, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclInternalError;
    if (c->rank == 0) c->hdr()->ncalls.fetch_add(1);
    return ncclSuccess;
}

// groups: the mock's collectives are synchronous, a group is just its members in order
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "success";
        case ncclInternalError: return "mock rccl: a rank did not reach the collective (barrier timeout)";
        case ncclInvalidArgument: return "mock rccl: invalid argument";
        case ncclSystemError: return "mock rccl: shared memory";
        default: return "mock rccl: hip error";
    }
}

}  // extern "C"
