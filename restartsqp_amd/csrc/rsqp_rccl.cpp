// rsqp_rccl.cpp -- the collectives of the path as native calls behind the C ABI: a C++ RestartSQP host that
// shards a batch of independent QPs over the GPUs of a node (one process per GPU) needs no Python and no communicator code
// of its own. The reference has no counterpart (single-threaded, no MPI / NCCL: SURVEY 5); the exchange steps are the two
// BASELINE.json names: a broadcast of shared problem data from rank 0 and the gather of the fixed-stride result records.
//
// RCCL is bound at FIRST USE (dlopen of librccl.so.1, signatures taken from <rccl/rccl.h>): librsqp_hip.so keeps loading
// on a host without RCCL, and a single-GPU run never touches it. Every function returns RSQP_ERR_DEVICE with the RCCL error
// text in rsqp_last_error() when the library or a call fails -- nothing falls back to a host path.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// a build host without the RCCL headers (single-GPU installs): the handful of types and prototypes this file binds by name,
// as <rccl/rccl.h> declares them -- the library still loads, and the entries fail with RSQP_ERR_DEVICE where librccl is missing
extern "C" {
typedef struct ncclComm *ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclChar = 0, ncclDouble = 8 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *);
ncclResult_t ncclCommInitRank(ncclComm_t *, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclCommUserRank(const ncclComm_t, int *);
ncclResult_t ncclCommCount(const ncclComm_t, int *);
ncclResult_t ncclAllGather(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
ncclResult_t ncclBroadcast(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
const char *ncclGetErrorString(ncclResult_t);
}
#endif

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/rsqp_hip.h"

int rsqp_fail_msg(int code, const char *msg);                  // rsqp_api.hip: sets rsqp_last_error()
hipStream_t rsqp_batch_stream_internal(rsqp_batch *b);         // rsqp_api.hip
int rsqp_batch_device_internal(const rsqp_batch *b);
int rsqp_batch_nq_internal(const rsqp_batch *b);

namespace {
struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};
Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    // RSQP_RCCL_LIBRARY names the one file to load instead of the usual candidates (a non-standard install; the CPU test of the
    // "library missing" path points it at a file that does not exist)
    const char *forced = getenv("RSQP_RCCL_LIBRARY");
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why;
    for (const char *n : names) {
        if (forced) n = forced;
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.h) break;
        const char *e = dlerror();       // (ONE call: it returns the message and clears it -- ADVICE r4)
        why = e ? e : "?";
        if (forced) break;
    }
    if (!g_rccl.h) { g_rccl.err = std::string("librccl.so not loadable: ") + why; return; }
#define RSQP_SYM(f)                                                                        \
    g_rccl.f = reinterpret_cast<decltype(g_rccl.f)>(dlsym(g_rccl.h, "nccl" #f));            \
    if (!g_rccl.f) { g_rccl.err = "librccl.so lacks nccl" #f; return; }
    RSQP_SYM(GetUniqueId) RSQP_SYM(CommInitRank) RSQP_SYM(CommDestroy) RSQP_SYM(CommUserRank) RSQP_SYM(CommCount)
    RSQP_SYM(AllGather) RSQP_SYM(Broadcast) RSQP_SYM(GetErrorString)
#undef RSQP_SYM
}
bool have_rccl() {
    std::call_once(g_once, load_rccl);
    return g_rccl.err.empty();
}
int rccl_fail(const char *where, ncclResult_t r) {
    std::string m = std::string(where) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return rsqp_fail_msg(RSQP_ERR_DEVICE, m.c_str());
}
#define NCCLCHK(call, where)                                  \
    do {                                                      \
        ncclResult_t r_ = (call);                             \
        if (r_ != ncclSuccess) return rccl_fail(where, r_);   \
    } while (0)
}  // namespace

extern "C" int rsqp_rccl_unique_id(char id[RSQP_RCCL_UNIQUE_ID_BYTES]) {
    static_assert(RSQP_RCCL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    if (!id) return rsqp_fail_msg(RSQP_ERR_ARG, "rsqp_rccl_unique_id");
    if (!have_rccl()) return rsqp_fail_msg(RSQP_ERR_DEVICE, g_rccl.err.c_str());
    ncclUniqueId u;
    NCCLCHK(g_rccl.GetUniqueId(&u), "ncclGetUniqueId");
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return RSQP_OK;
}

extern "C" int rsqp_rccl_comm_create(const char id[RSQP_RCCL_UNIQUE_ID_BYTES], int rank, int world, int device, void **comm) {
    if (!id || !comm || world <= 0 || rank < 0 || rank >= world) return rsqp_fail_msg(RSQP_ERR_ARG, "rsqp_rccl_comm_create");
    if (!have_rccl()) return rsqp_fail_msg(RSQP_ERR_DEVICE, g_rccl.err.c_str());
    if (hipSetDevice(device) != hipSuccess) return rsqp_fail_msg(RSQP_ERR_DEVICE, "rsqp_rccl_comm_create: hipSetDevice");
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    NCCLCHK(g_rccl.CommInitRank(&c, world, u, rank), "ncclCommInitRank");
    *comm = c;
    return RSQP_OK;
}

extern "C" int rsqp_rccl_comm_destroy(void *comm) {
    if (!comm) return RSQP_OK;
    if (!have_rccl()) return rsqp_fail_msg(RSQP_ERR_DEVICE, g_rccl.err.c_str());
    NCCLCHK(g_rccl.CommDestroy(static_cast<ncclComm_t>(comm)), "ncclCommDestroy");
    return RSQP_OK;
}

extern "C" int rsqp_rccl_broadcast_dev(void *comm, void *buf_dev, long long bytes, int root, void *hip_stream) {
    if (!comm || !buf_dev || bytes < 0) return rsqp_fail_msg(RSQP_ERR_ARG, "rsqp_rccl_broadcast_dev");
    if (!have_rccl()) return rsqp_fail_msg(RSQP_ERR_DEVICE, g_rccl.err.c_str());
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    NCCLCHK(g_rccl.Broadcast(buf_dev, buf_dev, (size_t)bytes, ncclChar, root, static_cast<ncclComm_t>(comm), st), "ncclBroadcast");
    if (hipStreamSynchronize(st) != hipSuccess) return rsqp_fail_msg(RSQP_ERR_DEVICE, "rsqp_rccl_broadcast_dev: stream");
    return RSQP_OK;
}

extern "C" int rsqp_batch_allgather_records(rsqp_batch *b, void *comm, int count_per_rank, double *all_dev) {
    if (!b || !comm || !all_dev || count_per_rank < rsqp_batch_nq_internal(b))
        return rsqp_fail_msg(RSQP_ERR_ARG, "rsqp_batch_allgather_records: count_per_rank must be >= the members of every rank");
    if (!have_rccl()) return rsqp_fail_msg(RSQP_ERR_DEVICE, g_rccl.err.c_str());
    if (hipSetDevice(rsqp_batch_device_internal(b)) != hipSuccess) return rsqp_fail_msg(RSQP_ERR_DEVICE, "hipSetDevice");
    ncclComm_t c = static_cast<ncclComm_t>(comm);
    int rank = 0, world = 0;
    NCCLCHK(g_rccl.CommUserRank(c, &rank), "ncclCommUserRank");
    NCCLCHK(g_rccl.CommCount(c, &world), "ncclCommCount");
    const long long stride = rsqp_batch_record_stride(b), nq = rsqp_batch_nq_internal(b);
    const size_t per_rank = (size_t)count_per_rank * (size_t)stride;
    hipStream_t st = rsqp_batch_stream_internal(b);
    // in place: this rank's records are packed straight into its slot of the gathered array (RCCL's in-place all-gather:
    // sendbuff == recvbuff + rank * sendcount), the padding records of a rank with fewer members are zeros (Exitflag 0)
    double *mine = all_dev + (size_t)rank * per_rank;
    if (nq < count_per_rank &&
        hipMemsetAsync(mine + (size_t)nq * stride, 0, sizeof(double) * (size_t)(count_per_rank - nq) * stride, st) != hipSuccess)
        return rsqp_fail_msg(RSQP_ERR_DEVICE, "rsqp_batch_allgather_records: memset");
    int rc = rsqp_batch_pack_records_dev(b, mine);
    if (rc != RSQP_OK) return rc;
    NCCLCHK(g_rccl.AllGather(mine, all_dev, per_rank, ncclDouble, c, st), "ncclAllGather");
    if (hipStreamSynchronize(st) != hipSuccess) return rsqp_fail_msg(RSQP_ERR_DEVICE, "rsqp_batch_allgather_records: stream");
    (void)world;
    return RSQP_OK;
}
