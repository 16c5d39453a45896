// collective.cpp -- the one exchange step of the path: the sum of the dense per-taxid tallies across GPUs
// (the serial merge of src/read_label.cpp:1760-1800), as an RCCL all-reduce over xGMI on the tally buffers where they
// lie in HBM.  Two shapes:
//   * one process drives several GPUs (read_label -t N): lmat_counts_allreduce(ctxs, n) -> ncclCommInitAll over the
//     contexts' devices + one grouped ncclAllReduce per array and context;
//   * one process per GPU (bench.py, any launcher): lmat_comm_unique_id on one rank, the 128 bytes handed to the
//     others by whatever the launcher offers, lmat_comm_init on every rank, lmat_comm_allreduce_counts.
// librccl (573 MB) is opened on first use, not at load time: a one-GPU run never maps it.  Inside a process that
// already holds an RCCL (PyTorch's) the same soname resolves to that copy, so there is one RCCL per process.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>
#include "lmat_internal.hpp"

using namespace lmat;

namespace {
struct Rccl {
    void* h = nullptr;
    std::string why;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.h) { r.why = std::string("cannot open librccl: ") + (dlerror() ? dlerror() : "?"); return; }
#define LMAT_SYM(f) r.f = (decltype(r.f))dlsym(r.h, "nccl" #f); if (!r.f) { r.why = "librccl lacks nccl" #f; return; }
        LMAT_SYM(GetUniqueId) LMAT_SYM(CommInitRank) LMAT_SYM(CommInitAll) LMAT_SYM(CommDestroy) LMAT_SYM(AllReduce)
        LMAT_SYM(GroupStart) LMAT_SYM(GroupEnd) LMAT_SYM(GetErrorString)
#undef LMAT_SYM
        r.ok = true;
    });
    return r;
}
int nccl_err(lmat_ctx* c, const char* what, ncclResult_t e) {
    return set_err(c, LMAT_E_DEVICE, std::string(what) + ": " + rccl().GetErrorString(e));
}
// Queued launches leave their last kernels -- which add to the tallies -- on the context's side streams (LMAT_PIPELINE): the
// merge is ordered behind them here, whatever the caller did or did not wait for.
void order_behind_launches(lmat_ctx* c) {
    if (c->set_in_flight && c->ev_done) hipStreamWaitEvent(c->stream, c->ev_done, 0);
    if (c->parked.in_flight && c->parked.done) hipStreamWaitEvent(c->stream, c->parked.done, 0);
}
// the three arrays of a tally buffer, queued on the context's stream (inside a group)
ncclResult_t queue_tallies(Rccl& R, lmat_ctx* c, ncclComm_t comm) {
    const uint32_t ids = c->dev.n_ids;
    uint64_t* cnt = (uint64_t*)c->d_counts;
    double* sc = (double*)(cnt + ids);
    uint64_t* nm = (uint64_t*)(sc + ids);
    ncclResult_t e = R.AllReduce(cnt, cnt, ids, ncclUint64, ncclSum, comm, c->stream);
    if (e == ncclSuccess) e = R.AllReduce(sc, sc, ids, ncclDouble, ncclSum, comm, c->stream);
    if (e == ncclSuccess) e = R.AllReduce(nm, nm, 3, ncclUint64, ncclSum, comm, c->stream);
    return e;
}
// contexts that share a device (two contexts on one GPU: tests, LMAT_DEVICES=0,0) cannot be ranks of one communicator:
// their arrays are copied out, summed on the host and copied back
int host_sum(lmat_ctx** ctxs, int n) {
    const uint64_t bytes = ctxs[0]->counts_bytes;
    const uint32_t ids = ctxs[0]->dev.n_ids;
    std::vector<unsigned char> sum(bytes, 0), buf(bytes);
    uint64_t* sc = (uint64_t*)sum.data();
    double* ss = (double*)(sc + ids);
    uint64_t* sn = (uint64_t*)(ss + ids);
    for (int i = 0; i < n; ++i) {
        lmat_ctx* c = ctxs[i];
        hipSetDevice(c->device);
        order_behind_launches(c);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(buf.data(), c->d_counts, bytes, hipMemcpyDeviceToHost) != hipSuccess)
            return set_err(c, LMAT_E_DEVICE, "copy of the tallies to the host failed");
        const uint64_t* bc = (const uint64_t*)buf.data();
        const double* bs = (const double*)(bc + ids);
        const uint64_t* bn = (const uint64_t*)(bs + ids);
        for (uint32_t j = 0; j < ids; ++j) { sc[j] += bc[j]; ss[j] += bs[j]; }
        for (int j = 0; j < 3; ++j) sn[j] += bn[j];
    }
    for (int i = 0; i < n; ++i) {
        hipSetDevice(ctxs[i]->device);
        if (hipMemcpy(ctxs[i]->d_counts, sum.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
            return set_err(ctxs[i], LMAT_E_DEVICE, "copy of the merged tallies to the device failed");
    }
    return LMAT_OK;
}
}  // namespace

namespace lmat {
void comm_free(lmat_ctx* c) {
    if (!c->comm_ranks && !c->comm_local) return;  // (never opens the library just to find nothing to free)
    Rccl& R = rccl();
    if (c->comm_ranks) { if (R.ok) R.CommDestroy((ncclComm_t)c->comm_ranks); c->comm_ranks = nullptr; }
    if (c->comm_local) { if (R.ok) R.CommDestroy((ncclComm_t)c->comm_local); c->comm_local = nullptr; c->comm_local_key.clear(); }
}
}  // namespace lmat

extern "C" {

int lmat_comm_available(void) { return rccl().ok ? 1 : 0; }

int lmat_comm_unique_id(uint8_t* id) {
    if (!id) return LMAT_E_ARG;
    Rccl& R = rccl();
    if (!R.ok) return LMAT_E_DEVICE;
    static_assert(sizeof(ncclUniqueId) == LMAT_COMM_ID_BYTES, "id size");
    ncclUniqueId u;
    if (R.GetUniqueId(&u) != ncclSuccess) return LMAT_E_DEVICE;
    memcpy(id, &u, sizeof u);
    return LMAT_OK;
}

int lmat_comm_init(lmat_ctx* c, const uint8_t* id, int n_ranks, int rank) {
    if (!c || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return LMAT_E_ARG;
    Rccl& R = rccl();
    if (!R.ok) return set_err(c, LMAT_E_DEVICE, R.why);
    hipSetDevice(c->device);
    if (c->comm_ranks) { R.CommDestroy((ncclComm_t)c->comm_ranks); c->comm_ranks = nullptr; }
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    const ncclResult_t e = R.CommInitRank(&comm, n_ranks, u, rank);
    if (e != ncclSuccess) return nccl_err(c, "ncclCommInitRank", e);
    c->comm_ranks = comm;
    c->comm_n = n_ranks;
    c->comm_rank = rank;
    return LMAT_OK;
}

int lmat_comm_allreduce_counts(lmat_ctx* c) {
    if (!c || !c->d_counts) return LMAT_E_ARG;
    if (!c->comm_ranks) return set_err(c, LMAT_E_ARG, "lmat_comm_init first");
    Rccl& R = rccl();
    hipSetDevice(c->device);
    order_behind_launches(c);
    ncclResult_t e = R.GroupStart();
    if (e == ncclSuccess) e = queue_tallies(R, c, (ncclComm_t)c->comm_ranks);
    const ncclResult_t e2 = R.GroupEnd();
    if (e != ncclSuccess || e2 != ncclSuccess) return nccl_err(c, "ncclAllReduce (tallies)", e != ncclSuccess ? e : e2);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return set_err(c, LMAT_E_DEVICE, "all-reduce of the tallies failed");
    return LMAT_OK;
}

int lmat_comm_size(const lmat_ctx* c) { return c && c->comm_ranks ? c->comm_n : 0; }

void lmat_comm_destroy(lmat_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    comm_free(c);
}

int lmat_counts_allreduce(lmat_ctx** ctxs, int n) {
    if (!ctxs || n < 1) return LMAT_E_ARG;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i] || !ctxs[i]->d_counts || ctxs[i]->counts_bytes != ctxs[0]->counts_bytes)
            return ctxs[i] ? set_err(ctxs[i], LMAT_E_ARG, "contexts must hold the same taxonomy") : LMAT_E_ARG;
    std::set<int> devs;
    for (int i = 0; i < n; ++i) devs.insert(ctxs[i]->device);
    if (n == 1) return LMAT_OK;
    if ((int)devs.size() != n) return host_sum(ctxs, n);  // a device twice: no communicator can hold both
    Rccl& R = rccl();
    if (!R.ok) return set_err(ctxs[0], LMAT_E_DEVICE, R.why);
    // one communicator per context, made together and kept for the next merge of the same set of contexts
    std::string key;
    for (int i = 0; i < n; ++i) key += std::to_string(ctxs[i]->device) + (i + 1 < n ? "," : "");
    bool have = true;
    for (int i = 0; i < n; ++i) have = have && ctxs[i]->comm_local && ctxs[i]->comm_local_key == key;
    if (!have) {
        std::vector<int> devlist(n);
        std::vector<ncclComm_t> comms(n, nullptr);
        for (int i = 0; i < n; ++i) {
            devlist[i] = ctxs[i]->device;
            if (ctxs[i]->comm_local) { R.CommDestroy((ncclComm_t)ctxs[i]->comm_local); ctxs[i]->comm_local = nullptr; }
        }
        const ncclResult_t e = R.CommInitAll(comms.data(), n, devlist.data());
        if (e != ncclSuccess) return nccl_err(ctxs[0], "ncclCommInitAll", e);
        for (int i = 0; i < n; ++i) { ctxs[i]->comm_local = comms[i]; ctxs[i]->comm_local_key = key; }
    }
    ncclResult_t e = R.GroupStart();
    for (int i = 0; i < n && e == ncclSuccess; ++i) {
        hipSetDevice(ctxs[i]->device);
        order_behind_launches(ctxs[i]);
        e = queue_tallies(R, ctxs[i], (ncclComm_t)ctxs[i]->comm_local);
    }
    const ncclResult_t e2 = R.GroupEnd();
    if (e != ncclSuccess || e2 != ncclSuccess) return nccl_err(ctxs[0], "ncclAllReduce (tallies)", e != ncclSuccess ? e : e2);
    for (int i = 0; i < n; ++i) {
        hipSetDevice(ctxs[i]->device);
        if (hipStreamSynchronize(ctxs[i]->stream) != hipSuccess) return set_err(ctxs[i], LMAT_E_DEVICE, "all-reduce of the tallies failed");
    }
    return LMAT_OK;
}

}  // extern "C"
