// pna_comm.cpp -- the one exchange step of the multi-GPU path behind the C ABI (BASELINE.json north_star: "RCCL over xGMI only for
// the final ordered gather of compressed chunks into the serial PNA stream").  One process per GPU; rank r has compressed the contiguous index range
// r of the entries (the fan-out of cli/src/command/core.rs:496-537 with GPUs in place of worker threads) into an archive PART in its HBM; the parts in
// rank order are the archive (the ordered drain of drain_entry_results, core.rs:471-493).  pna_gpu_gather_ordered: ncclAllGather of the parts' sizes,
// then grouped ncclSend / ncclRecv -- every sending rank has its own xGMI link to the root, so the seven transfers of an 8-GPU node run side by side.
// RCCL is taken with dlopen("librccl.so.1") at the first call: libpna_gpu.so itself links only the HIP runtime, and a process that never gathers never
// loads RCCL (nor a second copy next to the one PyTorch brings along).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/pna_gpu.h"

namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl *rccl() {
    static Rccl r; static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (r.h) break; }
        if (!r.h) return;
#define SYM(f) r.f = (decltype(r.f))dlsym(r.h, "nccl" #f)
        SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(AllGather); SYM(Send); SYM(Recv); SYM(GroupStart); SYM(GroupEnd); SYM(GetErrorString);
#undef SYM
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) { dlclose(r.h); r.h = nullptr; }
    });
    return r.h ? &r : nullptr;
}
}

struct pna_gpu_comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
    uint64_t *d_sizes = nullptr;         // nranks + 1 words in HBM: the all-gathered sizes, then this rank's own
    std::string err;
};

extern "C" const char *pna_gpu_comm_last_error(const pna_gpu_comm *m) { return m ? m->err.c_str() : "null communicator"; }

extern "C" int pna_gpu_comm_unique_id(void *id128) {
    Rccl *R = rccl();
    if (!R || !id128) return R ? PNA_E_INVAL : PNA_E_UNSUPPORTED;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return PNA_E_HIP;
    static_assert(sizeof(id) == PNA_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return PNA_OK;
}

extern "C" int pna_gpu_comm_init(int device_id, const void *id128, int nranks, int rank, pna_gpu_comm **out) {
    if (!out) return PNA_E_INVAL;
    *out = nullptr;
    Rccl *R = rccl();
    if (!R) return PNA_E_UNSUPPORTED;                            // no RCCL on this host
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return PNA_E_INVAL;
    if (hipSetDevice(device_id) != hipSuccess) return PNA_E_NODEVICE;
    pna_gpu_comm *m = new pna_gpu_comm();
    m->nranks = nranks; m->rank = rank; m->device = device_id;
    ncclUniqueId id; memcpy(&id, id128, sizeof(id));
    if (R->CommInitRank(&m->comm, nranks, id, rank) != ncclSuccess || hipMalloc((void **)&m->d_sizes, (size_t)(nranks + 1) * 8) != hipSuccess) {
        if (m->comm) (void)R->CommDestroy(m->comm);
        delete m; return PNA_E_HIP;
    }
    *out = m;
    return PNA_OK;
}

extern "C" void pna_gpu_comm_destroy(pna_gpu_comm *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->d_sizes) (void)hipFree(m->d_sizes);
    Rccl *R = rccl();
    if (R && m->comm) (void)R->CommDestroy(m->comm);
    delete m;
}

// Where rank r's part starts in the gathered stream: the exclusive prefix sums of the sizes (offs[nranks] = the total).  Pure host arithmetic,
// exported so that the CPU tests pin it; the gather below and any other transport (the direct-D2H comparison path of bench.py) use the same offsets.
extern "C" int pna_gather_offsets(const uint64_t *sizes, int nranks, uint64_t *offs) {
    if (!sizes || !offs || nranks < 1) return PNA_E_INVAL;
    uint64_t pos = 0;
    for (int r = 0; r < nranks; r++) { offs[r] = pos; if (pos + sizes[r] < pos) return PNA_E_INVAL; pos += sizes[r]; }
    offs[nranks] = pos;
    return PNA_OK;
}

extern "C" int pna_gpu_gather_ordered(pna_gpu_comm *m, const void *d_local, uint64_t local_len, int root, void *d_out, uint64_t out_cap,
                                      uint64_t *sizes, uint64_t *total, void *hip_stream) {
    Rccl *R = rccl();
    if (!m || !R) return PNA_E_INVAL;
    auto bad = [&](int code, const char *what, ncclResult_t nr = ncclSuccess) { m->err = what; if (nr != ncclSuccess && R->GetErrorString) { m->err += ": "; m->err += R->GetErrorString(nr); } return code; };
    if (root < 0 || root >= m->nranks || (local_len && !d_local)) return bad(PNA_E_INVAL, "bad argument");
    if (hipSetDevice(m->device) != hipSuccess) return bad(PNA_E_HIP, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)hip_stream;
    // (1) everybody learns every part's size: 8 bytes per rank
    uint64_t *d_mine = m->d_sizes + m->nranks;
    if (hipMemcpyAsync(d_mine, &local_len, 8, hipMemcpyHostToDevice, st) != hipSuccess) return bad(PNA_E_HIP, "size upload failed");
    ncclResult_t nr = R->AllGather(d_mine, m->d_sizes, 1, ncclUint64, m->comm, st);
    if (nr != ncclSuccess) return bad(PNA_E_HIP, "ncclAllGather", nr);
    std::vector<uint64_t> sz((size_t)m->nranks), offs((size_t)m->nranks + 1);
    if (hipMemcpyAsync(sz.data(), m->d_sizes, (size_t)m->nranks * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return bad(PNA_E_HIP, "size download failed");
    if (pna_gather_offsets(sz.data(), m->nranks, offs.data()) != PNA_OK) return bad(PNA_E_INVAL, "sizes overflow");
    if (sizes) memcpy(sizes, sz.data(), (size_t)m->nranks * 8);
    if (total) *total = offs[m->nranks];
    // (2) the parts travel to the root, each over its own link; the root's own part is a device copy
    if (m->rank == root) {
        if (offs[m->nranks] > out_cap || (offs[m->nranks] && !d_out)) return bad(PNA_E_DSTSIZE, "gather destination too small");   // (the senders' data then stays unsent: the caller aborts the job)
        if (local_len && hipMemcpyAsync((uint8_t *)d_out + offs[root], d_local, local_len, hipMemcpyDeviceToDevice, st) != hipSuccess) return bad(PNA_E_HIP, "local copy failed");
        if ((nr = R->GroupStart()) != ncclSuccess) return bad(PNA_E_HIP, "ncclGroupStart", nr);
        for (int r = 0; r < m->nranks; r++)
            if (r != root && sz[r] && (nr = R->Recv((uint8_t *)d_out + offs[r], sz[r], ncclUint8, r, m->comm, st)) != ncclSuccess) { (void)R->GroupEnd(); return bad(PNA_E_HIP, "ncclRecv", nr); }
        if ((nr = R->GroupEnd()) != ncclSuccess) return bad(PNA_E_HIP, "ncclGroupEnd", nr);
    } else if (local_len) {
        if ((nr = R->Send(d_local, local_len, ncclUint8, root, m->comm, st)) != ncclSuccess) return bad(PNA_E_HIP, "ncclSend", nr);
    }
    if (hipStreamSynchronize(st) != hipSuccess) return bad(PNA_E_HIP, "gather failed");
    return PNA_OK;
}
