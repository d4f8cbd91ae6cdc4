// sharding.cpp -- the exchange step of the vertex-sharded path (no reference analogue): an all-gather of equal-sized
// per-rank blocks laid out back to back in the handle's exchange buffer.  Two transports:
//   * RCCL inside the library (tnqs_set_sharding_rccl): librccl.so is loaded at run time, every rank joins one communicator through
//     an ncclUniqueId the host distributes by whatever means it has (the Julia shim: MPI.jl / a shared file; the Python host:
//     torch.distributed's store), and the in-place ncclAllGather is ENQUEUED ON THE HANDLE'S STREAM -- no stream synchronisation, no
//     host callback, no Python in the loop;
//   * a host callback (tnqs_set_sharding), kept for the gloo-based tests in which several ranks share one GPU (RCCL refuses that).
#include "engine_internal.hpp"
#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace tnqs {

// the few RCCL entry points used, resolved with dlsym (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE; ncclChar = 0)
struct NcclId { char internal[128]; };
typedef int (*fn_get_unique_id)(NcclId*);
typedef int (*fn_comm_init_rank)(void**, int, NcclId, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);
struct RcclApi {
    void* lib = nullptr; std::string path;
    fn_get_unique_id get_unique_id = nullptr; fn_comm_init_rank comm_init_rank = nullptr; fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr; fn_error_string error_string = nullptr;
};
static RcclApi& rccl() {
    static RcclApi api; static std::once_flag once;
    std::call_once(once, [] {
        // a copy that is already in the process (PyTorch-ROCm ships its own librccl.so) is reused, so that one process never runs two; a copy
        // loaded here stays out of the global symbol scope (RTLD_LOCAL): with RTLD_GLOBAL a second copy loaded later (import torch) bound
        // its own globals to this one's and the process died with a double free at exit
        const char* env = std::getenv("TNQS_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) { if (!n || !*n) continue; void* h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL); if (h) { api.lib = h; api.path = n; break; } }
        if (!api.lib) for (const char* n : names) { if (!n || !*n) continue; void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) { api.lib = h; api.path = n; break; } }
        if (!api.lib) return;
        api.get_unique_id = (fn_get_unique_id)dlsym(api.lib, "ncclGetUniqueId");
        api.comm_init_rank = (fn_comm_init_rank)dlsym(api.lib, "ncclCommInitRank");
        api.comm_destroy = (fn_comm_destroy)dlsym(api.lib, "ncclCommDestroy");
        api.all_gather = (fn_all_gather)dlsym(api.lib, "ncclAllGather");
        api.error_string = (fn_error_string)dlsym(api.lib, "ncclGetErrorString");
    });
    if (!api.lib) throw Err(TNQS_ERR_COMM, "RCCL: librccl.so could not be loaded (set TNQS_RCCL_LIB to its path)");
    if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_gather) throw Err(TNQS_ERR_COMM, "RCCL: " + api.path + " lacks a required entry point");
    return api;
}
static void ncclchk(int rc, const char* what) {
    if (rc == 0) return;
    RcclApi& a = rccl();
    throw Err(TNQS_ERR_COMM, std::string("RCCL: ") + what + " failed: " + (a.error_string ? a.error_string(rc) : std::to_string(rc).c_str()));
}

RcclComm::~RcclComm() {
    if (comm) { (void)hipSetDevice(device); (void)hipDeviceSynchronize(); (void)rccl().comm_destroy(comm); }
    if (exch_owned) (void)hipFree(exch_owned);
}

void rccl_unique_id(void* out128) {
    NcclId id; std::memset(&id, 0, sizeof id);
    ncclchk(rccl().get_unique_id(&id), "ncclGetUniqueId");
    std::memcpy(out128, &id, sizeof id);
}

void set_sharding_rccl(State* s, int rank, int nranks, const int32_t* owner, const void* unique_id128, int64_t exch_bytes) {
    if (nranks < 1 || rank < 0 || rank >= nranks) throw Err(TNQS_ERR_INVALID, "set_sharding_rccl: bad rank");
    if (!unique_id128 || exch_bytes <= 0) throw Err(TNQS_ERR_INVALID, "set_sharding_rccl: unique id and a positive exchange size are required");
    if (nranks > 1 && !owner) throw Err(TNQS_ERR_INVALID, "set_sharding_rccl: vertex_owner is required for nranks > 1");
    if (owner) for (int v = 0; v < s->g->nv; ++v) if (owner[v] < 0 || owner[v] >= nranks) throw Err(TNQS_ERR_INVALID, "set_sharding_rccl: owner out of range");
    hipchk(hipSetDevice(s->device), "hipSetDevice");
    materialize_pending_all(s);      // deferred one-site gates are applied while every tensor is still here (see tnqs_set_sharding)
    // everything that can fail locally happens BEFORE the collective communicator set-up: a rank that threw after ncclCommInitRank would
    // leave its peers joined to a communicator that is about to be destroyed
    RcclApi& api = rccl();
    auto c = std::make_shared<RcclComm>();
    c->device = s->device;
    hipchk(hipMalloc(&c->exch_owned, (size_t)exch_bytes), "hipMalloc (exchange buffer)");
    std::vector<int> owner_copy; if (owner) owner_copy.assign(owner, owner + s->g->nv);
    NcclId id; std::memcpy(&id, unique_id128, sizeof id);
    ncclchk(api.comm_init_rank(&c->comm, nranks, id, rank), "ncclCommInitRank");
    s->comm = c;
    s->rank = rank; s->nranks = nranks; s->ag_fn = nullptr; s->ag_ctx = nullptr; s->exch = c->exch_owned; s->exch_bytes = (size_t)exch_bytes;
    s->owner.swap(owner_copy);
    if (s->owner.empty()) s->owner.assign(s->g->nv, 0);
    s->force_exchange = nranks == 1 && envflag("TNQS_FORCE_EXCHANGE");      // (State::sharded)
    if (nranks > 1) for (int v = 0; v < s->g->nv; ++v) if (!s->owns(v)) { s->site[v] = nullptr; s->sscale[v] = nullptr; }
}

// local preflight of the RCCL transport, no collective involved: the library loads and exports what is needed (throws otherwise)
void rccl_preflight() { (void)rccl(); }

// in-place all-gather on the handle's stream: rank r's block sits at exch + r * bytes_per_rank
void rccl_allgather(State* s, size_t bytes_per_rank) {
    RcclComm& c = *s->comm;
    char* base = reinterpret_cast<char*>(s->exch);
    ncclchk(rccl().all_gather(base + (size_t)s->rank * bytes_per_rank, base, bytes_per_rank, /*ncclChar*/ 0, c.comm, s->stream), "ncclAllGather");
    c.n_exchanges += 1; c.bytes_exchanged += (long long)(bytes_per_rank * (size_t)s->nranks);
}

// one-rank self test (a single GPU cannot host two RCCL ranks): unique id, communicator, in-place all-gather on a private stream, teardown
void rccl_selftest(int device, int64_t bytes) {
    if (bytes <= 0 || bytes % 4) throw Err(TNQS_ERR_INVALID, "rccl_selftest: bytes must be a positive multiple of 4");
    hipchk(hipSetDevice(device), "hipSetDevice");
    NcclId id; std::memset(&id, 0, sizeof id);
    ncclchk(rccl().get_unique_id(&id), "ncclGetUniqueId");
    void* comm = nullptr;
    ncclchk(rccl().comm_init_rank(&comm, 1, id, 0), "ncclCommInitRank");
    std::vector<uint32_t> host((size_t)bytes / 4), back((size_t)bytes / 4, 0);
    for (size_t i = 0; i < host.size(); ++i) host[i] = (uint32_t)(i * 2654435761u);
    void* buf = nullptr; hipStream_t st = nullptr;
    hipchk(hipMalloc(&buf, (size_t)bytes), "hipMalloc"); hipchk(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    hipchk(hipMemcpyAsync(buf, host.data(), (size_t)bytes, hipMemcpyHostToDevice, st), "H2D");
    int rc = rccl().all_gather(buf, buf, (size_t)bytes, 0, comm, st);
    hipError_t e1 = hipMemcpyAsync(back.data(), buf, (size_t)bytes, hipMemcpyDeviceToHost, st);
    hipError_t e2 = hipStreamSynchronize(st);
    (void)rccl().comm_destroy(comm); (void)hipFree(buf); (void)hipStreamDestroy(st);
    ncclchk(rc, "ncclAllGather"); hipchk(e1, "D2H"); hipchk(e2, "hipStreamSynchronize");
    if (back != host) throw Err(TNQS_ERR_COMM, "rccl_selftest: data mismatch after the all-gather");
}


// all-gather of equal-sized per-rank blocks laid out back to back in the host-provided exchange buffer
void check_exchange(const State* s, size_t bytes_per_rank) {
    if (!s->sharded()) return;
    if (bytes_per_rank * (size_t)s->nranks > s->exch_bytes)
        throw Err(TNQS_ERR_COMM, "exchange buffer too small for this batch: " + std::to_string(bytes_per_rank * (size_t)s->nranks) + " bytes needed, " +
                                     std::to_string(s->exch_bytes) + " available (raise the buffer size passed to tnqs_set_sharding)");
}
void exchange(State* s, size_t bytes_per_rank) {
    if (!s->sharded()) return;
    check_exchange(s, bytes_per_rank);
    if (s->comm) { rccl_allgather(s, bytes_per_rank); return; }       // RCCL: enqueued on the handle's stream, nothing to wait for here
    if (!s->ag_fn) throw Err(TNQS_ERR_COMM, "sharded handle without a transport (tnqs_set_sharding_rccl or tnqs_set_sharding)");
    HIPCHK(hipStreamSynchronize(s->stream)); drained(s);
    int rc = s->ag_fn(s->ag_ctx, s->exch, (int64_t)bytes_per_rank, s->nranks);
    if (rc != 0) throw Err(TNQS_ERR_COMM, "all-gather callback failed");
}


}  // namespace tnqs
