// api_node.hip -- one frame over the GPUs of a node (north_star subsystem 4): mipt_node_* and
// mipt_render_node of include/mipt.h.
//
// The reference renders a frame with one kernel launch and reads it back (gpu/src/cs488.h:3209-3216); here the frame
// is cut into tiles, rank r (one host thread + one HIP stream + one mipt_ctx per GPU, the scene arena replicated)
// renders the tiles with (tx + 3*ty) % N == r, and ONE collective per frame moves the owned tiles to rank 0:
//     render -> k_pack_tiles (owned tiles -> dense send buffer) -> ncclGather to rank 0 -> k_unpack_tiles -> readback
// RCCL over xGMI: rank 0 receives from its N-1 peers over N-1 distinct links (3.1 MB per rank for a 1080p frame on 8
// GPUs); there is no all-reduce (a ring all-reduce of whole frames would move N times the bytes for nothing).  The
// per-pixel counter RNG is keyed on the global pixel index, so the gathered frame is bit-identical to the one-GPU
// frame.  RCCL is loaded with dlopen when the first node is made (a process that never makes one never loads it, and
// a process that already holds a copy -- torch ships one -- shares it).  Ranks that share a GPU (device ids repeated
// in mipt_node_create: rehearsal on a one-GPU machine) cannot form an RCCL communicator; their gather is N
// device-to-device copies ordered by events, everything else is the same code.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../device/kernels.hpp"
#include "../pt/host.hpp"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional: ends a collective that a failed rank never joined
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;  // an RCCL extension: may be absent
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
std::mutex g_rcclMutex;
Rccl g_rccl;

// nullptr (with g_rccl.error set) when the library or one of its symbols is missing
const Rccl* load_rccl() {
    std::lock_guard<std::mutex> lock(g_rcclMutex);
    if (g_rccl.handle) return &g_rccl;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) { g_rccl.error = std::string("RCCL not loadable: ") + dlerror(); return nullptr; }
    Rccl r;
    r.handle = h;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
    r.Gather = (decltype(r.Gather))dlsym(h, "ncclGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    r.Send = (decltype(r.Send))dlsym(h, "ncclSend");
    r.Recv = (decltype(r.Recv))dlsym(h, "ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
    if (getenv("MIPT_NODE_NO_NCCLGATHER")) r.Gather = nullptr;  // tests: the grouped send / receive path also for one GPU
    const bool p2p = r.Send && r.Recv && r.GroupStart && r.GroupEnd;
    if (!r.CommInitAll || !r.CommDestroy || !r.GetErrorString || (!r.Gather && !p2p)) {
        g_rccl.error = "RCCL lacks ncclCommInitAll / ncclCommDestroy / ncclGetErrorString, or both ncclGather and ncclSend / ncclRecv";
        dlclose(h);
        return nullptr;
    }
    g_rccl = r;
    return &g_rccl;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Rank {
    int device = 0;
    hipStream_t stream = nullptr;
    mipt_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t xfer = nullptr;   // the gather runs here, beside the next frame's kernels on `stream`
    hipEvent_t packed = nullptr;  // same-GPU gather: this rank's tiles have arrived in rank 0's receive buffer
    hipEvent_t evPacked = nullptr;    // `stream`: this frame's tiles are in its send slot (and the frame itself is complete)
    hipEvent_t evSent[2] = {nullptr, nullptr};  // `xfer`: the transfer out of send slot k has finished
    hipEvent_t evDone = nullptr;      // `xfer`: everything of this frame is done (rank 0: the other ranks' tiles are in its frame)
    unsigned long long frameNo = 0;   // frames rendered with this layout: send slot = frameNo & 1
    // frame state, (re)allocated when the resolution or the tiling changes
    void* frameMem = nullptr;     // accum | rgb | send x 2 | (rank 0) recv | tile lists
    float *accum = nullptr, *rgb = nullptr, *send = nullptr, *recv = nullptr;
    int* tiles = nullptr;         // this rank's owned tile ids; rank 0: every rank's list, rank r's at tileOfs[r]
    unsigned nTiles = 0;
    unsigned long long layoutSeen = 0;  // the node's layoutGen this rank's frame memory was laid out for
    int rc = MIPT_OK;
    std::string error;
};

enum Job { JOB_NONE, JOB_RENDER, JOB_QUIT };

}  // namespace

struct mipt_node {
    int n = 0;
    bool useRccl = false;
    const Rccl* rccl = nullptr;
    std::vector<Rank> ranks;
    std::vector<std::thread> threads;
    // frame layout (valid while key matches)
    std::vector<int> key;
    int tileW = 0, tileH = 0, tilesX = 0, tilesY = 0;
    std::vector<std::vector<int>> owned;   // per rank
    std::vector<size_t> tileOfs;           // start of rank r's list inside rank 0's tile array
    size_t maxTiles = 0, slotFloats = 0;   // floats per rank in the gather (maxTiles x tile pixels x 3)
    unsigned long long layoutGen = 0;      // bumped whenever the above changes
    std::atomic<int> failed{0};            // a rank failed before the gather: nobody enters it
    std::atomic<int> failedGather{0};      // a rank failed to enqueue its part of the gather: nobody waits for it
    bool exact = false;                    // RCCL: grouped send / receive of exactly the owned tiles instead of ncclGather
                                           // over maxTiles-sized slots (rank 0 keeps its own tiles where they are)
    // job hand-over
    std::mutex m;
    std::condition_variable cvGo, cvDone;
    unsigned long long generation = 0;
    int job = JOB_NONE, pending = 0, arrived = 0;
    unsigned long long arriveGen = 0;
    mipt_camera cam{};
    mipt_render_params rp{};
    float* rgbHost = nullptr;
    bool clear = false;
    bool async = true;   // a call without a host buffer returns with its work enqueued (MIPT_NODE_ASYNC=0: always wait)
    bool dead = false;   // a failed gather aborted the communicator
    double lastMs = 0.0;
};

namespace {

int fail(int code, const std::string& msg) {
    pt::set_error(msg);
    return code;
}

#define NODE_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            r.error = std::string(#expr) + " failed: " + hipGetErrorString(e_);             \
            return MIPT_ERR_DEVICE;                                                         \
        }                                                                                   \
    } while (0)

// all ranks meet here (host side) before rank 0 waits for their copies
void rank_barrier(mipt_node* nd) {
    std::unique_lock<std::mutex> lk(nd->m);
    const unsigned long long gen = nd->arriveGen;
    if (++nd->arrived == nd->n) {
        nd->arrived = 0;
        nd->arriveGen++;
        nd->cvDone.notify_all();
    } else {
        nd->cvDone.wait(lk, [&] { return nd->arriveGen != gen; });
    }
}

int ensure_frame(mipt_node* nd, int ri) {
    Rank& r = nd->ranks[(size_t)ri];
    const size_t frameBytes = align256((size_t)3 * nd->cam.width * nd->cam.height * sizeof(float));
    const size_t sendBytes = align256(nd->slotFloats * sizeof(float));
    const size_t recvBytes = ri == 0 ? align256(nd->slotFloats * sizeof(float) * (size_t)nd->n) : 0;
    size_t nList = nd->owned[(size_t)ri].size();
    if (ri == 0) { nList = 0; for (const auto& o : nd->owned) nList += o.size(); }
    const size_t listBytes = align256(nList * sizeof(int) + 4);
    NODE_HIP(hipStreamSynchronize(r.stream));
    NODE_HIP(hipStreamSynchronize(r.xfer));
    if (r.frameMem) { NODE_HIP(hipFree(r.frameMem)); r.frameMem = nullptr; }
    NODE_HIP(hipMalloc(&r.frameMem, 2 * frameBytes + 2 * sendBytes + recvBytes + listBytes));
    unsigned char* b = (unsigned char*)r.frameMem;
    r.accum = (float*)b; r.rgb = (float*)(b + frameBytes); r.send = (float*)(b + 2 * frameBytes);  // slot k at send + k * slotStride
    r.recv = ri == 0 ? (float*)(b + 2 * frameBytes + 2 * sendBytes) : nullptr;
    r.tiles = (int*)(b + 2 * frameBytes + 2 * sendBytes + recvBytes);
    r.frameNo = 0;
    std::vector<int> list;
    if (ri == 0) for (const auto& o : nd->owned) list.insert(list.end(), o.begin(), o.end());
    else list = nd->owned[(size_t)ri];
    if (!list.empty()) NODE_HIP(hipMemcpy(r.tiles, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice));
    NODE_HIP(hipMemsetAsync(r.frameMem, 0, 2 * frameBytes + 2 * sendBytes + recvBytes, r.stream));
    NODE_HIP(hipStreamSynchronize(r.stream));  // (the receive buffer is written from other streams: zeroed before anyone can)
    r.nTiles = (unsigned)nd->owned[(size_t)ri].size();
    return MIPT_OK;
}

// floats of rank k's owned tiles
size_t owned_floats(const mipt_node* nd, int k) { return nd->owned[(size_t)k].size() * (size_t)nd->tileW * nd->tileH * 3; }
float* send_slot(const mipt_node* nd, const Rank& r) { return r.send + (r.frameNo & 1ull) * (align256(nd->slotFloats * sizeof(float)) / sizeof(float)); }

// phase A: everything up to and including this rank's packed tiles, enqueued on its render stream
int render_rank_enqueue(mipt_node* nd, int ri) {
    Rank& r = nd->ranks[(size_t)ri];
    NODE_HIP(hipSetDevice(r.device));
    if (r.layoutSeen != nd->layoutGen || !r.frameMem) {
        const int rc = ensure_frame(nd, ri);
        if (rc != MIPT_OK) return rc;
        r.layoutSeen = nd->layoutGen;
    } else if (nd->clear) {
        // a fresh frame: the previous one's transfer (rank 0: the unpack into this very buffer) must be over first
        NODE_HIP(hipStreamWaitEvent(r.stream, r.evDone, 0));
        const size_t frameBytes = align256((size_t)3 * nd->cam.width * nd->cam.height * sizeof(float));
        NODE_HIP(hipMemsetAsync(r.frameMem, 0, 2 * frameBytes, r.stream));  // AccumulationBuffer.clear(), sampleCount = 0
    }
    mipt_render_params rp = nd->rp;
    rp.part_index = ri; rp.part_count = nd->n; rp.tile_w = nd->tileW; rp.tile_h = nd->tileH;
    const int rc = mipt_render_device(r.ctx, &nd->cam, &rp, r.accum, r.rgb);
    if (rc != MIPT_OK) { r.error = mipt_last_error(); return rc; }
    const bool sends = nd->useRccl ? (!nd->exact || ri != 0 || nd->n == 1) : (nd->n > 1 && ri != 0);  // rank 0's own tiles are already in its frame
    if (sends) {
        // the send slots are used in turn: the transfer that read this one two frames ago must be over
        NODE_HIP(hipStreamWaitEvent(r.stream, r.evSent[r.frameNo & 1ull], 0));
        ptd::launch_pack_tiles(r.stream, r.rgb, ri == 0 ? r.tiles + nd->tileOfs[0] : r.tiles, r.nTiles, nd->tileW, nd->tileH,
                               nd->tilesX, nd->cam.width, nd->cam.height, send_slot(nd, r));
    }
    NODE_HIP(hipGetLastError());
    NODE_HIP(hipEventRecord(r.evPacked, r.stream));
    return MIPT_OK;
}
// phase B: the gather to rank 0 and rank 0's unpack, enqueued on the TRANSFER stream behind evPacked, so that the next
// frame's kernels (render stream) run beside them.  Entered only when EVERY rank came through phase A (a collective that
// one rank never joins would leave the others' streams waiting for ever).
int render_rank_gather_enqueue(mipt_node* nd, int ri) {
    Rank& r = nd->ranks[(size_t)ri];
    NODE_HIP(hipStreamWaitEvent(r.xfer, r.evPacked, 0));
    if (nd->n > 1 || nd->useRccl) {
        float* const slot = send_slot(nd, r);
        if (nd->useRccl) {
            ncclResult_t e;
            if (!nd->exact) {
                e = nd->rccl->Gather(slot, r.recv, nd->slotFloats, ncclFloat, 0, r.comm, r.xfer);
            } else {
                // one group of point-to-point operations with exactly the owned bytes: every rank but 0 sends, rank 0
                // receives N - 1 (a one-GPU node sends to itself, so that the same calls run on every N)
                e = nd->rccl->GroupStart();
                if (e == ncclSuccess && (ri != 0 || nd->n == 1)) e = nd->rccl->Send(slot, owned_floats(nd, ri), ncclFloat, 0, r.comm, r.xfer);
                for (int k = nd->n == 1 ? 0 : 1; ri == 0 && k < nd->n && e == ncclSuccess; k++)
                    e = nd->rccl->Recv(r.recv + (size_t)k * nd->slotFloats, owned_floats(nd, k), ncclFloat, k, r.comm, r.xfer);
                const ncclResult_t e2 = nd->rccl->GroupEnd();
                if (e == ncclSuccess) e = e2;
            }
            if (e != ncclSuccess) { r.error = std::string("RCCL gather failed: ") + nd->rccl->GetErrorString(e); return MIPT_ERR_DEVICE; }
        } else if (ri != 0) {  // ranks share a GPU: a device-to-device copy into rank 0's slot once rank 0 is done with the last frame's
            NODE_HIP(hipStreamWaitEvent(r.xfer, nd->ranks[0].evDone, 0));
            NODE_HIP(hipMemcpyAsync(nd->ranks[0].recv + (size_t)ri * nd->slotFloats, slot, owned_floats(nd, ri) * sizeof(float),
                                    hipMemcpyDeviceToDevice, r.xfer));
            NODE_HIP(hipEventRecord(r.packed, r.xfer));
        }
    }
    NODE_HIP(hipEventRecord(r.evSent[r.frameNo & 1ull], r.xfer));
    return MIPT_OK;
}
// phase C (after a host barrier: every rank's `packed` event is recorded): rank 0 unpacks; everyone marks the frame done
int render_rank_finish(mipt_node* nd, int ri, bool waitForIt) {
    Rank& r = nd->ranks[(size_t)ri];
    if (ri == 0 && nd->n > 1) {
        if (!nd->useRccl)
            for (int k = 1; k < nd->n; k++) NODE_HIP(hipStreamWaitEvent(r.xfer, nd->ranks[(size_t)k].packed, 0));
        for (int k = 1; k < nd->n; k++)
            ptd::launch_unpack_tiles(r.xfer, r.recv + (size_t)k * nd->slotFloats, r.tiles + nd->tileOfs[(size_t)k],
                                     (unsigned)nd->owned[(size_t)k].size(), nd->tileW, nd->tileH, nd->tilesX, nd->cam.width,
                                     nd->cam.height, r.rgb);
        NODE_HIP(hipGetLastError());
    }
    if (ri == 0 && nd->rgbHost)
        NODE_HIP(hipMemcpyAsync(nd->rgbHost, r.rgb, (size_t)3 * nd->cam.width * nd->cam.height * sizeof(float), hipMemcpyDeviceToHost, r.xfer));
    NODE_HIP(hipEventRecord(r.evDone, r.xfer));
    r.frameNo++;
    if (waitForIt) {
        NODE_HIP(hipStreamSynchronize(r.xfer));
        NODE_HIP(hipStreamSynchronize(r.stream));
    }
    return MIPT_OK;
}
int render_rank(mipt_node* nd, int ri) {
    Rank& r = nd->ranks[(size_t)ri];
    int rc = render_rank_enqueue(nd, ri);
    if (rc != MIPT_OK) nd->failed.store(1);
    rank_barrier(nd);  // every rank has enqueued its frame (or failed)
    if (nd->failed.load()) {
        (void)hipStreamSynchronize(r.stream);
        if (rc == MIPT_OK) { rc = MIPT_ERR_DEVICE; r.error = "another rank failed before the gather"; }
        return rc;
    }
    rc = render_rank_gather_enqueue(nd, ri);
    if (rc != MIPT_OK) nd->failedGather.store(1);
    rank_barrier(nd);  // every rank has enqueued its part of the gather (or failed to); same-GPU ranks have recorded their events
    if (nd->failedGather.load()) {
        // the collective is short of a rank: nobody waits for it.  Where RCCL offers it the communicator is aborted, which
        // ends the operations already enqueued; the node is unusable afterwards and says so.
        if (nd->useRccl && nd->rccl->CommAbort && r.comm) { (void)nd->rccl->CommAbort(r.comm); r.comm = nullptr; }
        if (rc == MIPT_OK) { rc = MIPT_ERR_DEVICE; r.error = "another rank failed to enqueue its part of the gather"; }
        return rc;
    }
    // a caller that passed a host buffer gets the finished frame; one that did not (progressive accumulation, the frame
    // fetched by a later call) gets control back with the work enqueued
    rc = render_rank_finish(nd, ri, nd->rgbHost != nullptr || !nd->async);
    if (!nd->useRccl && nd->n > 1) rank_barrier(nd);  // rank 0's evDone is recorded before anyone enqueues a copy behind it
    return rc;
}

void worker(mipt_node* nd, int ri) {
    unsigned long long seen = 0;
    for (;;) {
        int job;
        {
            std::unique_lock<std::mutex> lk(nd->m);
            nd->cvGo.wait(lk, [&] { return nd->generation != seen; });
            seen = nd->generation;
            job = nd->job;
        }
        if (job == JOB_QUIT) return;
        Rank& r = nd->ranks[(size_t)ri];
        r.error.clear();
        const int rc = render_rank(nd, ri);
        {
            std::lock_guard<std::mutex> lk(nd->m);
            r.rc = rc;
            if (--nd->pending == 0) nd->cvDone.notify_all();
        }
    }
}

void run_job(mipt_node* nd, int job) {
    std::unique_lock<std::mutex> lk(nd->m);
    nd->job = job;
    nd->pending = nd->n;
    nd->generation++;
    nd->cvGo.notify_all();
    if (job != JOB_QUIT) nd->cvDone.wait(lk, [&] { return nd->pending == 0; });
}

}  // namespace

extern "C" {

int mipt_node_create(const mipt_scene* scene, int ngpus, const int* devices, mipt_node** out) {
    if (!scene || !out || ngpus <= 0 || ngpus > 64) return fail(MIPT_ERR_INVALID, "mipt_node_create: bad argument (1..64 GPUs)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MIPT_ERR_DEVICE, "no HIP device available");
    struct KeepDevice {  // the caller's current device is the caller's
        int dev = -1;
        KeepDevice() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
        ~KeepDevice() { if (dev >= 0) (void)hipSetDevice(dev); }
    } keep;
    std::vector<int> devs((size_t)ngpus);
    for (int i = 0; i < ngpus; i++) devs[(size_t)i] = devices ? devices[i] : i;
    bool distinct = true;
    for (int i = 0; i < ngpus; i++) {
        if (devs[(size_t)i] < 0 || devs[(size_t)i] >= ndev)
            return fail(MIPT_ERR_DEVICE, "mipt_node_create: device " + std::to_string(devs[(size_t)i]) + " of " + std::to_string(ndev) + " does not exist");
        for (int k = 0; k < i; k++) distinct = distinct && devs[(size_t)k] != devs[(size_t)i];
    }
    mipt_node* nd = nullptr;
    try {
        nd = new mipt_node();
        nd->n = ngpus;
        nd->ranks.resize((size_t)ngpus);
        for (int i = 0; i < ngpus; i++) {
            Rank& r = nd->ranks[(size_t)i];
            r.device = devs[(size_t)i];
            if (hipSetDevice(r.device) != hipSuccess || hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&r.xfer, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&r.packed, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&r.evPacked, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&r.evSent[0], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&r.evSent[1], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&r.evDone, hipEventDisableTiming) != hipSuccess) {
                mipt_node_destroy(nd);
                return fail(MIPT_ERR_DEVICE, "mipt_node_create: stream / event creation failed on device " + std::to_string(r.device));
            }
            const int rc = mipt_context_create(scene, r.device, r.stream, &r.ctx);
            if (rc != MIPT_OK) { mipt_node_destroy(nd); return rc; }
        }
        if (distinct) {  // one communicator over the node's GPUs (also for N = 1: the same code runs on every N)
            nd->rccl = load_rccl();
            if (!nd->rccl) { const std::string e = g_rccl.error; mipt_node_destroy(nd); return fail(MIPT_ERR_DEVICE, e); }
            std::vector<ncclComm_t> comms((size_t)ngpus, nullptr);
            const ncclResult_t e = nd->rccl->CommInitAll(comms.data(), ngpus, devs.data());
            if (e != ncclSuccess) {
                const std::string msg = std::string("ncclCommInitAll failed: ") + nd->rccl->GetErrorString(e);
                mipt_node_destroy(nd);
                return fail(MIPT_ERR_DEVICE, msg);
            }
            for (int i = 0; i < ngpus; i++) nd->ranks[(size_t)i].comm = comms[(size_t)i];
            nd->useRccl = true;
            // exactly the owned bytes, point to point, wherever RCCL has send / receive (MIPT_NODE_GATHER=gather keeps the
            // one-call collective over padded slots; without ncclGather there is no choice)
            const bool p2p = nd->rccl->Send && nd->rccl->Recv && nd->rccl->GroupStart && nd->rccl->GroupEnd;
            const char* gv = getenv("MIPT_NODE_GATHER");
            nd->exact = p2p && !(gv && std::string(gv) == "gather" && nd->rccl->Gather);
            if (!nd->rccl->Gather) nd->exact = true;
        }
        if (const char* av = getenv("MIPT_NODE_ASYNC")) nd->async = std::string(av) != "0";
        for (int i = 0; i < ngpus; i++) nd->threads.emplace_back(worker, nd, i);
    } catch (const std::exception& e) {
        if (nd) mipt_node_destroy(nd);
        return fail(MIPT_ERR_NOMEM, std::string("mipt_node_create: ") + e.what());
    }
    *out = nd;
    return MIPT_OK;
}

void mipt_node_destroy(mipt_node* nd) {
    if (!nd) return;
    int callerDev = -1;
    if (hipGetDevice(&callerDev) != hipSuccess) callerDev = -1;
    if (!nd->threads.empty()) {
        run_job(nd, JOB_QUIT);
        for (auto& t : nd->threads) t.join();
    }
    for (Rank& r : nd->ranks) {
        (void)hipSetDevice(r.device);
        if (r.stream) (void)hipStreamSynchronize(r.stream);
        if (r.xfer) (void)hipStreamSynchronize(r.xfer);
        if (r.comm && nd->rccl) (void)nd->rccl->CommDestroy(r.comm);
        if (r.ctx) mipt_context_destroy(r.ctx);
        if (r.frameMem) (void)hipFree(r.frameMem);
        for (hipEvent_t e : {r.packed, r.evPacked, r.evSent[0], r.evSent[1], r.evDone})
            if (e) (void)hipEventDestroy(e);
        if (r.xfer) (void)hipStreamDestroy(r.xfer);
        if (r.stream) (void)hipStreamDestroy(r.stream);
    }
    delete nd;
    if (callerDev >= 0) (void)hipSetDevice(callerDev);
}

int mipt_node_num_gpus(const mipt_node* nd) { return nd ? nd->n : 0; }
int mipt_node_uses_rccl(const mipt_node* nd) { return (nd && nd->useRccl) ? 1 : 0; }

int mipt_node_render(mipt_node* nd, const mipt_camera* cam, const mipt_render_params* rp, float* rgb_host) {
    if (!nd || !cam || !rp) return fail(MIPT_ERR_INVALID, "mipt_node_render: null argument");
    if (cam->width <= 0 || cam->height <= 0 || cam->width > 32768 || cam->height > 32768)
        return fail(MIPT_ERR_INVALID, "camera: bad resolution");
    const int tileW = rp->tile_w > 0 ? rp->tile_w : 64, tileH = rp->tile_h > 0 ? rp->tile_h : 8;
    if ((tileW & 7) || (tileH & 7)) return fail(MIPT_ERR_INVALID, "tile_w and tile_h must be multiples of 8");
    try {
        const std::vector<int> key = {cam->width, cam->height, tileW, tileH};
        const bool relayout = key != nd->key;
        if (relayout) {
            nd->tileW = tileW; nd->tileH = tileH;
            nd->tilesX = (cam->width + tileW - 1) / tileW; nd->tilesY = (cam->height + tileH - 1) / tileH;
            nd->owned.assign((size_t)nd->n, {});
            for (int ty = 0; ty < nd->tilesY; ty++)
                for (int tx = 0; tx < nd->tilesX; tx++) nd->owned[(size_t)((tx + 3 * ty) % nd->n)].push_back(ty * nd->tilesX + tx);  // as mipt_render_device
            nd->tileOfs.assign((size_t)nd->n, 0);
            nd->maxTiles = 0;
            size_t o = 0;
            for (int r = 0; r < nd->n; r++) { nd->tileOfs[(size_t)r] = o; o += nd->owned[(size_t)r].size(); nd->maxTiles = std::max(nd->maxTiles, nd->owned[(size_t)r].size()); }
            nd->slotFloats = nd->maxTiles * (size_t)tileW * tileH * 3;
            nd->key = key;
            nd->layoutGen++;  // the workers lay their frame memory out again
        }
        nd->cam = *cam; nd->rp = *rp; nd->rgbHost = rgb_host;
        if (nd->dead) return fail(MIPT_ERR_DEVICE, "mipt_node_render: an earlier gather failed and the communicator was aborted; make a new node");
        nd->failed.store(0);
        nd->failedGather.store(0);
        nd->clear = !(rp->mode == MIPT_MODE_PATH && rp->sample_offset > 0);  // a progressive call continues its accumulation
        const auto t0 = std::chrono::steady_clock::now();
        run_job(nd, JOB_RENDER);
        nd->lastMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    } catch (const std::exception& e) {
        return fail(MIPT_ERR_NOMEM, std::string("mipt_node_render: ") + e.what());
    }
    for (int i = 0; i < nd->n; i++)
        if (nd->ranks[(size_t)i].rc != MIPT_OK) {
            const int rc = nd->ranks[(size_t)i].rc;
            nd->key.clear();  // whatever state the ranks are in, the next call lays the frame out again
            if (nd->failedGather.load() && nd->useRccl) nd->dead = true;
            return fail(rc, "rank " + std::to_string(i) + " (device " + std::to_string(nd->ranks[(size_t)i].device) + "): " + nd->ranks[(size_t)i].error);
        }
    return MIPT_OK;
}

int mipt_node_synchronize(mipt_node* nd) {
    if (!nd) return fail(MIPT_ERR_INVALID, "mipt_node_synchronize: null node");
    int callerDev = -1;
    if (hipGetDevice(&callerDev) != hipSuccess) callerDev = -1;
    int rc = MIPT_OK;
    for (Rank& r : nd->ranks) {
        if (hipSetDevice(r.device) != hipSuccess || hipStreamSynchronize(r.stream) != hipSuccess || hipStreamSynchronize(r.xfer) != hipSuccess)
            rc = fail(MIPT_ERR_DEVICE, "mipt_node_synchronize: a stream of device " + std::to_string(r.device) + " failed");
    }
    if (callerDev >= 0) (void)hipSetDevice(callerDev);
    return rc;
}

int mipt_node_get_stats(mipt_node* nd, mipt_stats* out, double* last_frame_ms) {
    if (!nd || !out) return fail(MIPT_ERR_INVALID, "mipt_node_get_stats: null argument");
    int callerDev = -1;
    if (hipGetDevice(&callerDev) != hipSuccess) callerDev = -1;
    mipt_stats sum{};
    for (Rank& r : nd->ranks) {
        mipt_stats s{};
        (void)hipSetDevice(r.device);
        (void)hipStreamSynchronize(r.xfer);  // (a call without a host buffer leaves its transfer enqueued)
        const int rc = mipt_context_get_stats(r.ctx, &s);
        if (rc != MIPT_OK) return rc;
        sum.rays_closest += s.rays_closest; sum.rays_occluded += s.rays_occluded;
        sum.launches_closest += s.launches_closest; sum.launches_occluded += s.launches_occluded; sum.launches_other += s.launches_other;
        sum.ms_closest += s.ms_closest; sum.ms_occluded += s.ms_occluded; sum.ms_other += s.ms_other;
        sum.node_visits_closest += s.node_visits_closest; sum.tri_tests_closest += s.tri_tests_closest;
        sum.node_visits_occluded += s.node_visits_occluded; sum.tri_tests_occluded += s.tri_tests_occluded;
        sum.launches_shade += s.launches_shade; sum.ms_shade += s.ms_shade;
    }
    if (callerDev >= 0) (void)hipSetDevice(callerDev);
    *out = sum;
    if (last_frame_ms) *last_frame_ms = nd->lastMs;
    return MIPT_OK;
}

int mipt_render_node(const mipt_scene* scene, const mipt_camera* cam, const mipt_render_params* params, int ngpus,
                     float* rgb_out_rank0) {
    if (!rgb_out_rank0) return fail(MIPT_ERR_INVALID, "mipt_render_node: null output");
    mipt_node* nd = nullptr;
    int rc = mipt_node_create(scene, ngpus, nullptr, &nd);
    if (rc != MIPT_OK) return rc;
    rc = mipt_node_render(nd, cam, params, rgb_out_rank0);
    const std::string err = rc != MIPT_OK ? mipt_last_error() : "";
    mipt_node_destroy(nd);
    if (rc != MIPT_OK) pt::set_error(err);
    return rc;
}

}  // extern "C"
