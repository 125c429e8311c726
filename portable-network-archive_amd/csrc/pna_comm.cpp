// pna_comm.cpp -- the one exchange step of the multi-GPU path behind the C ABI (BASELINE.json north_star: "RCCL over xGMI only for
// the final ordered gather of compressed chunks into the serial PNA stream").  One process per GPU; rank r has compressed the contiguous index range
// r of the entries (the fan-out of cli/src/command/core.rs:496-537 with GPUs in place of worker threads) into an archive PART in its HBM; the parts in
// rank order are the archive (the ordered drain of drain_entry_results, core.rs:471-493).  pna_gpu_gather_ordered: ncclAllGather of the parts' sizes
// and the root's capacity (one verdict for all ranks), then grouped ncclSend / ncclRecv -- every sending rank has its own xGMI link to the root, so the seven transfers of an 8-GPU node run side by side.
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
    uint64_t *d_sizes = nullptr;         // 2 (nranks + 1) words in HBM: the all-gathered (size, capacity) pairs, then this rank's own
    uint64_t *h_pairs = nullptr;         // page-locked: this rank's pair, then the gathered pairs
    hipStream_t st = nullptr;            // the communicator's stream: every RCCL call of this communicator is issued here
    hipEvent_t ev = nullptr;             // orders a gather behind the caller's stream
    static constexpr int RING = 8;
    hipEvent_t done[RING] = {};          // done[t % RING]: recorded behind the t-th gather posted (tickets count from 1)
    uint64_t posted = 0;
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

extern "C" void pna_gpu_comm_destroy(pna_gpu_comm *m);
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
    if (R->CommInitRank(&m->comm, nranks, id, rank) != ncclSuccess || hipMalloc((void **)&m->d_sizes, (size_t)(nranks + 1) * 16) != hipSuccess ||
        hipHostMalloc((void **)&m->h_pairs, (size_t)(nranks + 1) * 16, hipHostMallocDefault) != hipSuccess ||
        hipStreamCreateWithFlags(&m->st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&m->ev, hipEventDisableTiming) != hipSuccess) {
        pna_gpu_comm_destroy(m); return PNA_E_HIP;
    }
    for (int i = 0; i < pna_gpu_comm::RING; i++)
        if (hipEventCreateWithFlags(&m->done[i], hipEventDisableTiming) != hipSuccess) { pna_gpu_comm_destroy(m); return PNA_E_HIP; }
    *out = m;
    return PNA_OK;
}

extern "C" void pna_gpu_comm_destroy(pna_gpu_comm *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->st) { (void)hipStreamSynchronize(m->st); }
    Rccl *R = rccl();
    if (R && m->comm) (void)R->CommDestroy(m->comm);
    if (m->d_sizes) (void)hipFree(m->d_sizes);
    if (m->h_pairs) (void)hipHostFree(m->h_pairs);
    if (m->ev) (void)hipEventDestroy(m->ev);
    for (int i = 0; i < pna_gpu_comm::RING; i++) if (m->done[i]) (void)hipEventDestroy(m->done[i]);
    if (m->st) (void)hipStreamDestroy(m->st);
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

// What every rank decides from the all-gathered (size, capacity) pairs -- the SAME verdict on every rank, so nobody is left inside a send whose receive is
// never posted: PNA_OK and the offsets, PNA_E_INVAL when the sizes overflow 64 bits, PNA_E_DSTSIZE when the parts do not fit the root's destination.
// pairs[2 r] = rank r's part length, pairs[2 r + 1] = the capacity rank r offers (only the root's counts).  Host arithmetic, exported for the CPU tests.
extern "C" int pna_gather_verdict(const uint64_t *pairs, int nranks, int root, uint64_t *sizes, uint64_t *offs) {
    if (!pairs || nranks < 1 || root < 0 || root >= nranks) return PNA_E_INVAL;
    std::vector<uint64_t> sz((size_t)nranks), of((size_t)nranks + 1);
    for (int r = 0; r < nranks; r++) sz[(size_t)r] = pairs[2 * r];
    if (pna_gather_offsets(sz.data(), nranks, of.data()) != PNA_OK) return PNA_E_INVAL;
    if (sizes) memcpy(sizes, sz.data(), (size_t)nranks * 8);
    if (offs) memcpy(offs, of.data(), ((size_t)nranks + 1) * 8);
    return of[(size_t)nranks] > pairs[2 * root + 1] ? PNA_E_DSTSIZE : PNA_OK;
}

// The gather, posted: the size exchange (16 bytes per rank) runs on the communicator's OWN stream -- behind the transfers of earlier gathers, so a start waits for
// those as well -- and is what this call waits for; the parts'
// transfers are queued on that stream behind an event recorded on `hip_stream` (the stream whose work produced d_local) and the call returns.
// pna_gpu_gather_wait() blocks until they are done; until then d_local and d_out belong to the gather.  So a host that compresses piece k + 1 into a
// second buffer after this call overlaps it with piece k's transfer -- one RCCL stream per communicator, no RCCL call ever issued on the caller's stream.
extern "C" int pna_gpu_gather_ordered_start(pna_gpu_comm *m, const void *d_local, uint64_t local_len, int root, void *d_out, uint64_t out_cap,
                                            uint64_t *sizes, uint64_t *total, void *hip_stream) {
    Rccl *R = rccl();
    if (!m || !R) return PNA_E_INVAL;
    auto bad = [&](int code, const char *what, ncclResult_t nr = ncclSuccess) { m->err = what; if (nr != ncclSuccess && R->GetErrorString) { m->err += ": "; m->err += R->GetErrorString(nr); } return code; };
    if (root < 0 || root >= m->nranks || (local_len && !d_local)) return bad(PNA_E_INVAL, "bad argument");
    if (hipSetDevice(m->device) != hipSuccess) return bad(PNA_E_HIP, "hipSetDevice failed");
    hipStream_t cs = m->st;
    // (1) everybody learns every part's size and the root's capacity: two words per rank.  (A root without a destination offers capacity 0.)
    m->h_pairs[0] = local_len;
    m->h_pairs[1] = (m->rank == root && d_out) ? out_cap : 0;
    uint64_t *d_mine = m->d_sizes + 2 * (size_t)m->nranks;
    if (hipMemcpyAsync(d_mine, m->h_pairs, 16, hipMemcpyHostToDevice, cs) != hipSuccess) return bad(PNA_E_HIP, "size upload failed");
    ncclResult_t nr = R->AllGather(d_mine, m->d_sizes, 2, ncclUint64, m->comm, cs);
    if (nr != ncclSuccess) return bad(PNA_E_HIP, "ncclAllGather", nr);
    uint64_t *pairs = m->h_pairs + 2;
    if (hipMemcpyAsync(pairs, m->d_sizes, (size_t)m->nranks * 16, hipMemcpyDeviceToHost, cs) != hipSuccess || hipStreamSynchronize(cs) != hipSuccess)
        return bad(PNA_E_HIP, "size download failed");
    std::vector<uint64_t> sz((size_t)m->nranks), offs((size_t)m->nranks + 1);
    const int verdict = pna_gather_verdict(pairs, m->nranks, root, sz.data(), offs.data());
    if (verdict == PNA_E_INVAL) return bad(PNA_E_INVAL, "sizes overflow");
    if (sizes) memcpy(sizes, sz.data(), (size_t)m->nranks * 8);
    if (total) *total = offs[(size_t)m->nranks];
    // every rank has the same pairs, hence the same verdict: on an overflow NOBODY sends and every rank returns the error
    if (verdict == PNA_E_DSTSIZE) return bad(PNA_E_DSTSIZE, "gather destination too small (no part was sent; every rank returns this)");
    // (2) the parts travel to the root, each over its own link, once the caller's stream has produced them; the root's own part is a device copy
    if (hipEventRecord(m->ev, (hipStream_t)hip_stream) != hipSuccess || hipStreamWaitEvent(cs, m->ev, 0) != hipSuccess) return bad(PNA_E_HIP, "stream order failed");
    if (m->rank == root) {
        if (local_len && hipMemcpyAsync((uint8_t *)d_out + offs[(size_t)root], d_local, local_len, hipMemcpyDeviceToDevice, cs) != hipSuccess) return bad(PNA_E_HIP, "local copy failed");
        if ((nr = R->GroupStart()) != ncclSuccess) return bad(PNA_E_HIP, "ncclGroupStart", nr);
        for (int r = 0; r < m->nranks; r++)
            if (r != root && sz[(size_t)r] && (nr = R->Recv((uint8_t *)d_out + offs[(size_t)r], sz[(size_t)r], ncclUint8, r, m->comm, cs)) != ncclSuccess) { (void)R->GroupEnd(); return bad(PNA_E_HIP, "ncclRecv", nr); }
        if ((nr = R->GroupEnd()) != ncclSuccess) return bad(PNA_E_HIP, "ncclGroupEnd", nr);
    } else if (local_len) {
        if ((nr = R->Send(d_local, local_len, ncclUint8, root, m->comm, cs)) != ncclSuccess) return bad(PNA_E_HIP, "ncclSend", nr);
    }
    m->posted++;
    if (hipEventRecord(m->done[m->posted % pna_gpu_comm::RING], cs) != hipSuccess) return bad(PNA_E_HIP, "event record failed");
    return PNA_OK;
}

// the number of gathers this communicator has posted = the ticket of the latest one
extern "C" uint64_t pna_gpu_gather_ticket(const pna_gpu_comm *m) { return m ? m->posted : 0; }

// Blocks until the gathers up to `ticket` are done (0 or a ticket out of the ring's reach: all of them).  A host that double-buffers waits for the
// gather that used a buffer two pieces ago while the latest one is still travelling.
extern "C" int pna_gpu_gather_wait_for(pna_gpu_comm *m, uint64_t ticket) {
    if (!m) return PNA_E_INVAL;
    if (hipSetDevice(m->device) != hipSuccess) { m->err = "hipSetDevice failed"; return PNA_E_HIP; }
    hipError_t e;
    if (ticket == 0 || ticket > m->posted || m->posted - ticket >= (uint64_t)pna_gpu_comm::RING) e = hipStreamSynchronize(m->st);
    else e = hipEventSynchronize(m->done[ticket % pna_gpu_comm::RING]);
    if (e != hipSuccess) { m->err = "gather failed"; return PNA_E_HIP; }
    return PNA_OK;
}
extern "C" int pna_gpu_gather_wait(pna_gpu_comm *m) { return pna_gpu_gather_wait_for(m, 0); }

// the synchronous form: start + wait
extern "C" int pna_gpu_gather_ordered(pna_gpu_comm *m, const void *d_local, uint64_t local_len, int root, void *d_out, uint64_t out_cap,
                                      uint64_t *sizes, uint64_t *total, void *hip_stream) {
    const int rc = pna_gpu_gather_ordered_start(m, d_local, local_len, root, d_out, out_cap, sizes, total, hip_stream);
    return rc != PNA_OK ? rc : pna_gpu_gather_wait(m);
}
