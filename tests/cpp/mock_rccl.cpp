// mock_rccl.cpp — TEST INFRASTRUCTURE ONLY (tests/test_gpu_comm.py): the nine librccl entry points alac_comm.cpp resolves
// with dlsym, for ranks that are THREADS of one process sharing one GPU.  RCCL itself refuses two ranks on one device, and
// this builder's box has one GPU, so the product's N > 1 exchange (all-gather of the shard table, ONE group of ncclSend /
// ncclRecv at prefix-sum offsets) could otherwise only run at world 1.  The library loads this file instead of librccl when
// ALAC_HIP_RCCL_LIB names it.
//
// Semantics kept: ncclCommInitRank is collective over the ranks of a unique id; ncclAllGather places rank r's `count` elements
// at r * count; a send to peer p matches the receive posted by p from this rank, both inside ncclGroupStart / ncclGroupEnd,
// and their element counts must be equal (a mismatch is ncclInvalidArgument here; in RCCL it is a hang or corruption).
// Semantics NOT kept: asynchrony — every operation synchronises its stream, meets the other ranks at a barrier and copies
// with hipMemcpy; nothing about bandwidth or overlap can be learnt from it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <condition_variable>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace {

struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    hipStream_t stream;
};

struct World {
    int n = 0, joined = 0;
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    std::vector<const void *> gatherPtr;
    std::vector<std::vector<Op> > posted;  // [rank] ops of the group being closed
    int bad = 0;
    void barrier()
    {
        std::unique_lock<std::mutex> l(m);
        const uint64_t g = generation;
        if (++waiting == n) {
            waiting = 0;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(l, [&] { return generation != g; });
        }
    }
};

struct Comm {
    World *w;
    int rank;
};

std::mutex gMutex;
std::map<std::string, World *> gWorlds;
thread_local std::vector<Op> tGroup;
thread_local int tDepth = 0;
thread_local void *tComm = nullptr;  // the communicator the open group's operations named (one per thread in alac_comm.cpp)

size_t dtype_bytes(int t) { return t == 0 || t == 1 ? 1 : (t == 2 || t == 3 || t == 7) ? 4 : 8; }

int run_group(Comm *c, std::vector<Op> &ops)
{
    World *w = c->w;
    for (const Op &o : ops) (void)hipStreamSynchronize(o.stream);  // what the sends read is complete
    {
        std::lock_guard<std::mutex> l(w->m);
        w->posted[c->rank] = ops;
    }
    w->barrier();
    int bad = 0;
    for (const Op &o : ops) {
        if (o.send) continue;
        const Op *match = nullptr;
        for (const Op &s : w->posted[o.peer])
            if (s.send && s.peer == c->rank) match = &s;
        if (!match || match->bytes != o.bytes) {
            bad = 1;
            continue;
        }
        if (o.bytes && hipMemcpy(o.buf, match->buf, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) bad = 1;
    }
    // every send must have been matched by a receive of the same size
    for (const Op &o : ops) {
        if (!o.send) continue;
        bool found = false;
        for (const Op &r : w->posted[o.peer])
            if (!r.send && r.peer == c->rank && r.bytes == o.bytes) found = true;
        if (!found) bad = 1;
    }
    if (bad) {
        std::lock_guard<std::mutex> l(w->m);
        w->bad = 1;
    }
    w->barrier();
    return w->bad ? 4 /* ncclInvalidArgument */ : 0;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(void *id)
{
    std::random_device rd;
    uint8_t *p = (uint8_t *)id;
    for (int i = 0; i < 128; i++) p[i] = (uint8_t)rd();
    return 0;
}

struct IdArg {
    char b[128];
};

int ncclCommInitRank(void **comm, int nranks, IdArg id, int rank)
{
    World *w;
    {
        std::lock_guard<std::mutex> l(gMutex);
        const std::string key(id.b, 128);
        auto it = gWorlds.find(key);
        if (it == gWorlds.end()) {
            w = new World;
            w->n = nranks;
            w->gatherPtr.assign(nranks, nullptr);
            w->posted.assign(nranks, std::vector<Op>());
            gWorlds[key] = w;
        } else {
            w = it->second;
        }
    }
    if (w->n != nranks || rank < 0 || rank >= nranks) return 4;
    w->barrier();  // collective
    Comm *c = new Comm;
    c->w = w;
    c->rank = rank;
    *comm = c;
    return 0;
}

int ncclCommDestroy(void *comm)
{
    delete (Comm *)comm;
    return 0;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream)
{
    Comm *c = (Comm *)comm;
    World *w = c->w;
    const size_t bytes = count * dtype_bytes(dtype);
    (void)hipStreamSynchronize(stream);
    {
        std::lock_guard<std::mutex> l(w->m);
        w->gatherPtr[c->rank] = send;
    }
    w->barrier();
    int bad = 0;
    for (int r = 0; r < w->n; r++)
        if (bytes && hipMemcpy((uint8_t *)recv + (size_t)r * bytes, w->gatherPtr[r], bytes, hipMemcpyDeviceToDevice) != hipSuccess) bad = 1;
    w->barrier();
    return bad ? 1 : 0;
}

int ncclGroupStart()
{
    tDepth++;
    return 0;
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    if (tDepth == 0) return 4;  // the product posts every send / receive inside ONE group
    tComm = comm;
    tGroup.push_back(Op{true, (void *)buf, count * dtype_bytes(dtype), peer, stream});
    return 0;
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    if (tDepth == 0) return 4;
    tComm = comm;
    tGroup.push_back(Op{false, buf, count * dtype_bytes(dtype), peer, stream});
    return 0;
}

int ncclGroupEnd()
{
    if (--tDepth > 0) return 0;
    std::vector<Op> ops;
    ops.swap(tGroup);
    void *comm = tComm;
    tComm = nullptr;
    if (ops.empty()) return 0;  // (every shard empty: no rank posted anything)
    return run_group((Comm *)comm, ops);
}

const char *ncclGetErrorString(int r) { return r == 0 ? "no error" : (r == 4 ? "mock: invalid argument / unmatched send-receive" : "mock: failure"); }

}  // extern "C"
