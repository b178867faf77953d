// alac_comm.cpp — stream re-assembly of a sharded encode across the GPUs of one node, on RCCL (BASELINE
// north_star: "frames are sharded across the 8 GPUs of one node with RCCL all-gather over xGMI to reassemble the stream").
//
// One process per GPU has encoded a contiguous range of the independent segments (alac_hip_shard_range).  Packets are byte
// aligned (codec/ALACEncoder.cu:1039 in the reference), so the stream is the concatenation of the shards in rank order and
// re-assembly is pure byte placement at the exclusive prefix sums of the shard lengths (alac_hip_shard_offsets).  The
// exchange is an all-gather with PER-RANK counts, which RCCL does not have as one call:
//   begin   ncclAllGather of {shard bytes, shard capacity, output capacity} (24 B per rank) and, optionally, of the
//           per-packet sizes (the CAF 'pakt' table: equal counts per rank); the table is copied to pinned host memory
//           behind it and an event recorded — no host wait
//   finish  the host waits for THAT event only (the table of a pass enqueued a step ago has long arrived), forms the
//           offsets, and posts ONE group: ncclRecv of every peer's shard straight at its final offset in the caller's
//           stream buffer + ncclSend of the own shard to every peer (ncclGroupStart ... ncclGroupEnd), own shard = a local
//           device copy.  No padding to the longest shard, no staging buffer, no second copy.  xGMI is point to point:
//           a rank's 7 receives arrive over 7 different links at once.
// Every precondition that makes finish refuse (a shard longer than its buffer, a stream longer than ANY rank's output
// buffer) is evaluated from the gathered table, which is identical on every rank: either all ranks post the group or none
// does — a rank that bails out alone would hang its peers inside RCCL.
//
// librccl is loaded with dlopen on first use (libalac_hip.so has no link-time dependency on it: a single-GPU user never
// needs it); inside a PyTorch process "librccl.so.1" resolves to the copy torch has already mapped.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "alac_hip.h"

namespace {

// the slice of rccl.h this file uses (declared here so that the build does not need the header's include path and the
// dlopen'ed entry points are typed)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;  // ncclSuccess = 0
enum { kNcclUint8 = 1, kNcclUint32 = 3, kNcclUint64 = 5 };  // ncclDataType_t values of rccl.h
static_assert(sizeof(ncclUniqueId) == ALAC_HIP_COMM_ID_BYTES, "NCCL_UNIQUE_ID_BYTES");

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl &rccl()
{
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {getenv("ALAC_HIP_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            R.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (R.lib) break;
        }
        if (!R.lib) {
            R.why = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "");
            return;
        }
        bool ok = true;
        auto sym = [&](const char *s) {
            void *p = dlsym(R.lib, s);
            if (!p) {
                ok = false;
                R.why = std::string("librccl lacks ") + s;
            }
            return p;
        };
        R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
        R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
        R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
        R.Send = (decltype(R.Send))sym("ncclSend");
        R.Recv = (decltype(R.Recv))sym("ncclRecv");
        R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
        R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
        R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
        if (!ok) {
            dlclose(R.lib);
            R.lib = nullptr;
        }
    });
    return R;
}

constexpr uint32_t kSlots = ALAC_HIP_COMM_SLOTS;
constexpr uint32_t kCols = 3;  // {shard bytes, shard capacity, output capacity} per rank

}  // namespace

struct alac_hip_comm {
    int device = 0;
    uint32_t rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    std::string err;
    // per slot: this rank's row and the gathered table on the device, the table again in pinned host memory, an event
    uint64_t *dMine = nullptr;   // [kSlots][kCols]
    uint64_t *dTable = nullptr;  // [kSlots][world][kCols]
    uint64_t *hMine = nullptr;   // pinned [kSlots][kCols] (source of the capacities)
    uint64_t *hTable = nullptr;  // pinned [kSlots][world][kCols]
    hipEvent_t ready[kSlots] = {};
    bool pending[kSlots] = {};
};

namespace {

int32_t cfail(alac_hip_comm *c, int32_t code, const std::string &what)
{
    if (c) c->err = what;
    return code;
}

int32_t nfail(alac_hip_comm *c, const char *what, ncclResult_t r)
{
    const Rccl &R = rccl();
    return cfail(c, ALAC_HIP_ParamError, std::string(what) + ": " + (R.GetErrorString ? R.GetErrorString(r) : "rccl error"));
}

}  // namespace

extern "C" {

int32_t alac_hip_comm_unique_id(uint8_t *h_id)
{
    Rccl &R = rccl();
    if (!h_id || !R.lib) return ALAC_HIP_ParamError;
    ncclUniqueId id;
    if (R.GetUniqueId(&id) != 0) return ALAC_HIP_ParamError;
    memcpy(h_id, id.internal, sizeof(id.internal));
    return ALAC_HIP_noErr;
}

int32_t alac_hip_comm_create(alac_hip_comm **out, int32_t device, const uint8_t *h_id, uint32_t rank, uint32_t world)
{
    if (!out) return ALAC_HIP_ParamError;
    *out = nullptr;
    if (!h_id || world == 0 || rank >= world) return ALAC_HIP_ParamError;
    Rccl &R = rccl();
    if (!R.lib) {
        fprintf(stderr, "alac_hip_comm_create: %s\n", R.why.c_str());
        return ALAC_HIP_UnimplementedError;
    }
    if (hipSetDevice(device) != hipSuccess) return ALAC_HIP_ParamError;
    alac_hip_comm *c = new (std::nothrow) alac_hip_comm;
    if (!c) return ALAC_HIP_MemFullError;
    c->device = device;
    c->rank = rank;
    c->world = world;
    bool ok = hipMalloc((void **)&c->dMine, kSlots * kCols * 8) == hipSuccess &&
              hipMalloc((void **)&c->dTable, (size_t)kSlots * world * kCols * 8) == hipSuccess &&
              hipHostMalloc((void **)&c->hMine, kSlots * kCols * 8, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void **)&c->hTable, (size_t)kSlots * world * kCols * 8, hipHostMallocDefault) == hipSuccess;
    for (uint32_t s = 0; ok && s < kSlots; s++) ok = hipEventCreateWithFlags(&c->ready[s], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        alac_hip_comm_destroy(c);
        return ALAC_HIP_MemFullError;
    }
    ncclUniqueId id;
    memcpy(id.internal, h_id, sizeof(id.internal));
    const ncclResult_t r = R.CommInitRank(&c->comm, (int)world, id, (int)rank);
    if (r != 0) {
        fprintf(stderr, "alac_hip_comm_create: ncclCommInitRank: %s\n", R.GetErrorString(r));
        c->comm = nullptr;
        alac_hip_comm_destroy(c);
        return ALAC_HIP_ParamError;
    }
    *out = c;
    return ALAC_HIP_noErr;
}

void alac_hip_comm_destroy(alac_hip_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    for (uint32_t s = 0; s < kSlots; s++)
        if (c->ready[s]) (void)hipEventDestroy(c->ready[s]);
    if (c->dMine) (void)hipFree(c->dMine);
    if (c->dTable) (void)hipFree(c->dTable);
    if (c->hMine) (void)hipHostFree(c->hMine);
    if (c->hTable) (void)hipHostFree(c->hTable);
    delete c;
}

uint32_t alac_hip_comm_rank(const alac_hip_comm *c) { return c ? c->rank : 0; }
uint32_t alac_hip_comm_world(const alac_hip_comm *c) { return c ? c->world : 0; }
const char *alac_hip_comm_last_error(const alac_hip_comm *c) { return c ? c->err.c_str() : ""; }

int32_t alac_hip_reassemble_begin(alac_hip_comm *c, uint32_t slot, const uint64_t *d_shard_bytes, uint64_t shard_capacity,
                                  uint64_t out_capacity, const uint32_t *d_packet_bytes, uint32_t num_packets,
                                  uint32_t *d_all_packet_bytes, void *stream)
{
    if (!c || slot >= kSlots || !d_shard_bytes) return cfail(c, ALAC_HIP_ParamError, "reassemble_begin: bad argument");
    if ((d_packet_bytes == nullptr) != (d_all_packet_bytes == nullptr))
        return cfail(c, ALAC_HIP_ParamError, "reassemble_begin: d_packet_bytes and d_all_packet_bytes go together");
    if (c->pending[slot]) return cfail(c, ALAC_HIP_ParamError, "reassemble_begin: slot still pending (finish it first)");
    Rccl &R = rccl();
    hipStream_t st = (hipStream_t)stream;
    if (hipSetDevice(c->device) != hipSuccess) return cfail(c, ALAC_HIP_ParamError, "hipSetDevice");
    uint64_t *dMine = c->dMine + slot * kCols, *hMine = c->hMine + slot * kCols;
    uint64_t *dTable = c->dTable + (size_t)slot * c->world * kCols, *hTable = c->hTable + (size_t)slot * c->world * kCols;
    hMine[1] = shard_capacity;
    hMine[2] = out_capacity;
    // the shard's length is on the device (the last entry of alac_hip_encode's offsets): it never visits the host here
    if (hipMemcpyAsync(dMine, d_shard_bytes, 8, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(dMine + 1, hMine + 1, 16, hipMemcpyHostToDevice, st) != hipSuccess)
        return cfail(c, ALAC_HIP_ParamError, "reassemble_begin: staging this rank's row");
    ncclResult_t r = R.AllGather(dMine, dTable, kCols, kNcclUint64, c->comm, st);
    if (r != 0) return nfail(c, "ncclAllGather (shard table)", r);
    if (d_packet_bytes && num_packets) {
        r = R.AllGather(d_packet_bytes, d_all_packet_bytes, num_packets, kNcclUint32, c->comm, st);
        if (r != 0) return nfail(c, "ncclAllGather (packet sizes)", r);
    }
    if (hipMemcpyAsync(hTable, dTable, (size_t)c->world * kCols * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipEventRecord(c->ready[slot], st) != hipSuccess)
        return cfail(c, ALAC_HIP_ParamError, "reassemble_begin: table read-back");
    c->pending[slot] = true;
    return ALAC_HIP_noErr;
}

int32_t alac_hip_reassemble_finish(alac_hip_comm *c, uint32_t slot, const uint8_t *d_shard, uint8_t *d_stream_out,
                                   uint64_t *h_offsets, void *stream)
{
    if (!c || slot >= kSlots || !d_shard || !d_stream_out) return cfail(c, ALAC_HIP_ParamError, "reassemble_finish: bad argument");
    if (!c->pending[slot]) return cfail(c, ALAC_HIP_ParamError, "reassemble_finish: no begin on this slot");
    Rccl &R = rccl();
    hipStream_t st = (hipStream_t)stream;
    if (hipSetDevice(c->device) != hipSuccess) return cfail(c, ALAC_HIP_ParamError, "hipSetDevice");
    c->pending[slot] = false;
    if (hipEventSynchronize(c->ready[slot]) != hipSuccess) return cfail(c, ALAC_HIP_ParamError, "reassemble_finish: table event");
    const uint64_t *T = c->hTable + (size_t)slot * c->world * kCols;
    std::vector<uint64_t> lens(c->world), offs(c->world + 1);
    uint64_t minOut = UINT64_MAX;
    for (uint32_t r = 0; r < c->world; r++) {
        lens[r] = T[r * kCols];
        if (lens[r] > T[r * kCols + 1]) {
            char b[160];
            snprintf(b, sizeof b, "rank %u's shard buffer (%llu B) is shorter than its declared length (%llu B)", r,
                     (unsigned long long)T[r * kCols + 1], (unsigned long long)lens[r]);
            return cfail(c, ALAC_HIP_ParamError, b);  // the same verdict on every rank: nobody posts the group
        }
        minOut = T[r * kCols + 2] < minOut ? T[r * kCols + 2] : minOut;
    }
    if (alac_hip_shard_offsets(lens.data(), c->world, offs.data()) != 0)
        return cfail(c, ALAC_HIP_ParamError, "reassemble_finish: shard lengths overflow");
    if (offs[c->world] > minOut) {
        char b[160];
        snprintf(b, sizeof b, "the re-assembled stream (%llu B) does not fit the smallest output buffer of the job (%llu B)",
                 (unsigned long long)offs[c->world], (unsigned long long)minOut);
        return cfail(c, ALAC_HIP_ParamError, b);
    }
    if (h_offsets) memcpy(h_offsets, offs.data(), (c->world + 1) * 8);
    const uint32_t me = c->rank;
    if (lens[me] && hipMemcpyAsync(d_stream_out + offs[me], d_shard, lens[me], hipMemcpyDeviceToDevice, st) != hipSuccess)
        return cfail(c, ALAC_HIP_ParamError, "reassemble_finish: placing the own shard");
    if (c->world > 1) {
        ncclResult_t r = R.GroupStart();
        if (r != 0) return nfail(c, "ncclGroupStart", r);
        ncclResult_t bad = 0;
        const char *where = "";
        for (uint32_t p = 0; p < c->world && bad == 0; p++) {
            if (p == me) continue;
            if (lens[p]) {
                bad = R.Recv(d_stream_out + offs[p], lens[p], kNcclUint8, (int)p, c->comm, st);
                where = "ncclRecv";
            }
            if (bad == 0 && lens[me]) {
                bad = R.Send(d_shard, lens[me], kNcclUint8, (int)p, c->comm, st);
                where = "ncclSend";
            }
        }
        r = R.GroupEnd();  // always closed, also after a failed post
        if (bad != 0) return nfail(c, where, bad);
        if (r != 0) return nfail(c, "ncclGroupEnd", r);
    }
    return ALAC_HIP_noErr;
}

int32_t alac_hip_reassemble(alac_hip_comm *c, const uint8_t *d_shard, const uint64_t *d_shard_bytes, uint64_t shard_capacity,
                            const uint32_t *d_packet_bytes, uint32_t num_packets, uint32_t *d_all_packet_bytes,
                            uint8_t *d_stream_out, uint64_t out_capacity, uint64_t *h_offsets, void *stream)
{
    const int32_t rc = alac_hip_reassemble_begin(c, 0, d_shard_bytes, shard_capacity, out_capacity, d_packet_bytes, num_packets,
                                                 d_all_packet_bytes, stream);
    if (rc != 0) return rc;
    return alac_hip_reassemble_finish(c, 0, d_shard, d_stream_out, h_offsets, stream);
}

}  // extern "C"
