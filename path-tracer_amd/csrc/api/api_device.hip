// api_device.hip -- the device half of the C-ABI (include/mipt.h): scene
// validation and copy, the one-arena upload, and the entry points that drive
// the HIP kernels.  There is no CPU path here: without a usable GPU every
// compute entry point fails with MIPT_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <thread>
#include <atomic>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

#include "../device/kernels.hpp"
#include "../device/shading.hpp"
#include "../pt/host.hpp"

using namespace ptd;

namespace {

unsigned pcg_hash_host(unsigned v) {
    const unsigned state = v * 747796405u + 2891336453u;
    const unsigned word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            pt::set_error(std::string(#expr) + " failed: " + hipGetErrorString(e_));           \
            return MIPT_ERR_DEVICE;                                                            \
        }                                                                                      \
    } while (0)

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// ---- scratch of the entry point that allocates nothing (mipt_trace_device_noalloc) ----
// What mipt_trace_device takes from hipMalloc'd memory (the launch cursors and the list of rays k_trace_wide hands
// back, mipt_ctx::slow) lives here instead, in the library's own image: every GPU the library runs on holds its own
// copy at its own address until the process ends.  Layout: words 0 / 1 = count and second-launch cursor of the list,
// word 32 = the launch cursor (a 128-B line of its own), bytes 256.. = the list, one int per ray.
constexpr size_t kNoAllocMaxRays = (size_t)1 << 20;
constexpr size_t kNoAllocHeaderBytes = 256;
static __device__ __align__(256) unsigned char scratch[kNoAllocHeaderBytes + sizeof(int) * kNoAllocMaxRays];
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratchMutex;
void* g_scratchPtr[kMaxDevices] = {};  // device address of `scratch` per GPU; the array's NAME is only a host-side stand-in
// once per device, with that device current (mipt_context_create)
int resolve_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) { pt::set_error("device index beyond the library's table"); return MIPT_ERR_LIMIT; }
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (g_scratchPtr[device]) return MIPT_OK;
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) { pt::set_error("hipGetSymbolAddress(scratch) returned null"); return MIPT_ERR_DEVICE; }
    HIP_TRY(hipMemset((unsigned char*)p + kNoAllocHeaderBytes, 0xff, sizeof(int) * kNoAllocMaxRays));  // the list rests at -1 (ensure_slow)
    g_scratchPtr[device] = p;
    return MIPT_OK;
}
}  // namespace

// host copy of a validated scene
struct mipt_scene {
    std::vector<float4> nodes, leafTris, triAttr, materials, lights;
    std::vector<unsigned char> texels;
    std::vector<float> env;
    int nNodes = 0, nLeafTris = 0, nTris = 0, nMaterials = 0, envW = 0, envH = 0, nLights = 0;
    bool planesTame = true;  // see DevScene::planesTame
    int nTopNodes = 0;       // see DevScene::nTopNodes
    // 4-wide tree over the same leaves for the order-free traversal (device_scene.hpp "wide")
    std::vector<float4> wnodes, leafBox, leafHead;
    std::vector<int> nodeRight;
    std::vector<int> roots;     // device numbering: the root of every object's tree (one entry for a scene of one object)
    std::vector<int> objTriBase, objMatBase;  // mipt_scene_create_multi: first global triangle / material id of every object
    std::vector<int4> btLinks;  // StacklessBVH2 links in device numbering
    bool hasSpecular = false;   // some material is metal or glass: only then does shade() recurse (C:2455-2490)
    bool btOk = false;
    bool slowOk = false;  // binary tree of depth <= 96: k_trace_slow may run
    int nWide = 0;
    float wideInvScale = 1.0f;  // 1 / S: the wide nodes' fp16 planes are the scene's planes times S, a power of two
    bool wideOk = false;
};

// float -> fp16 bits rounded towards -inf (up == false) or +inf (up == true); finite input.  Returns 0x7c00 / 0xfc00
// (infinity) when the value lies beyond the fp16 range on the side the rounding goes to.
static unsigned short half_towards(float f, bool up) {
    unsigned b;
    memcpy(&b, &f, 4);
    const unsigned sign = b >> 31, mag = b & 0x7fffffffu;
    const bool away = up != (sign != 0);  // the magnitude is rounded away from zero
    const int e = (int)(mag >> 23) - 127;
    const unsigned m = mag & 0x7fffffu;
    unsigned h;
    if (mag == 0) h = 0;
    else if (e < -24) h = away ? 1u : 0u;  // below the smallest fp16 denormal, 2^-24
    else if (e < -14) {                    // fp16 denormal: units of 2^-24
        const unsigned full = m | 0x800000u;
        const int shift = -(e + 1);        // e = -15 -> 14 ... e = -24 -> 23
        const unsigned q = full >> shift, rem = full & ((1u << shift) - 1u);
        h = q + ((rem && away) ? 1u : 0u);
    } else if (e > 15) h = away ? 0x7c00u : 0x7bffu;
    else {
        h = ((unsigned)(e + 15) << 10) | (m >> 13);
        if ((m & 0x1fffu) && away) h += 1u;  // a carry runs into the exponent as it should
    }
    return (unsigned short)((sign << 15) | h);
}
static float half_value(unsigned short h) {
    const unsigned sign = h >> 15, e = (h >> 10) & 31u, m = h & 1023u;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m | 1024u), (int)e - 25);
    return sign ? -v : v;
}

// Is the threaded tree a proper binary tree (every interior node's idHit = its left child, that child's idMiss = the
// right child, both further on in preorder, every node but the root with exactly one parent)?  The bit-trail links
// and k_trace_slow's right-child table need exactly this; the box planes play no part in it.
// Several objects per scene (Scene::objects, cs488.h:2006; mipt_scene_create_multi) arrive as a FOREST in one node array:
// the trees one after the other, every link by which the walk leaves tree k (0 in the object's own numbering) pointing at
// the root of tree k + 1.  The literal walk then visits the objects in order, exactly as Scene::intersect's loop does
// (cs488.h:2075-2083: strict '<' between objects = strict '<' in visit order).  The roots are the chain 0 -> id_miss[0] ->
// ...; a single tree is a forest of one.
static bool forest_roots(const mipt_scene_desc* d, std::vector<int>& roots) {
    roots.clear();
    const int n = d->n_nodes;
    for (int r = 0;;) {
        roots.push_back(r);
        const int nx = d->id_miss[r];
        if (nx == 0) return true;
        if (!(nx > r && nx < n)) return false;
        r = nx;
    }
}
static bool is_binary_tree(const mipt_scene_desc* d, const std::vector<int>& roots) {
    const int n = d->n_nodes;
    std::vector<char> seen((size_t)n, 0);
    for (const int r : roots) seen[(size_t)r] = 1;
    for (int i = 0; i < n; i++) {
        if (d->leaf_count[i] > 0) continue;
        const int l = d->id_hit[i];
        if (!(l > i && l < n)) return false;
        const int r = d->id_miss[l];
        if (!(r > l && r < n)) return false;
        if (seen[l] || seen[r]) return false;
        if (d->id_miss[r] != d->id_miss[i]) return false;  // the walk leaves the right subtree where it leaves the node's (cs488.h:1655-1658)
        seen[l] = seen[r] = 1;
    }
    for (int i = 0; i < n; i++)
        if (!seen[i]) return false;
    return true;
}

// The binary tree build_wide collapses: either the reference's own (copied from the descriptor) or a hierarchy of
// our own over the reference's LEAVES (rebuild_top).  refLeaf[i] >= 0 marks a leaf and names the reference node whose
// exact box, leaf_first and leaf_count it stands for.
struct BinTree {
    std::vector<float> mn, mx;  // 3 per node
    std::vector<int> left, right, refLeaf;
    int add(const float* lo, const float* hi, int ref) {
        for (int a = 0; a < 3; a++) { mn.push_back(lo[a]); mx.push_back(hi[a]); }
        left.push_back(-1); right.push_back(-1); refLeaf.push_back(ref);
        return (int)refLeaf.size() - 1;
    }
    int size() const { return (int)refLeaf.size(); }
};

// A hierarchy of our own over the reference's leaves.  The order-free traversal needs the reference's LEAVES (their
// exact boxes, their triangles in leafTris order) and above them any hierarchy whose boxes contain what lies below
// (kernels_trace_wide.hip: a leaf whose exact box passes the exact predicate is reached through every ancestor whose
// stored box contains it, because the conservative slot test is monotone in the planes).  The reference's upper tree
// is one such hierarchy, but not a good one: its SAH sweep looks at the longest axis only (cs488.h:1463) and its
// binary shape was not made to be collapsed four wide.  Here: a top-down SAH build over the leaf boxes, exact sweep
// along all three axes at every node, weights = triangles per leaf.  Same leaves, same admission test, same
// tie-break, hence the same hits bit for bit -- with fewer node visits per ray.
static void rebuild_top(const mipt_scene_desc* d, BinTree& t) {
    struct Leaf { float lo[3], hi[3], c[3]; int ref, w; };
    std::vector<Leaf> leaves;
    for (int i = 0; i < d->n_nodes; i++) {
        if (d->leaf_count[i] <= 0) continue;
        Leaf l;
        for (int a = 0; a < 3; a++) {
            l.lo[a] = d->node_min[3 * i + a]; l.hi[a] = d->node_max[3 * i + a];
            l.c[a] = 0.5f * l.lo[a] + 0.5f * l.hi[a];
        }
        l.ref = i; l.w = d->leaf_count[i];
        leaves.push_back(l);
    }
    const int nl = (int)leaves.size();
    std::vector<int> idx((size_t)nl);
    for (int i = 0; i < nl; i++) idx[(size_t)i] = i;
    auto area = [](const float* lo, const float* hi) -> double {
        const double ex = (double)hi[0] - lo[0], ey = (double)hi[1] - lo[1], ez = (double)hi[2] - lo[2];
        return ex * ey + ey * ez + ez * ex;
    };
    auto grow = [](float* lo, float* hi, const Leaf& l) {
        for (int a = 0; a < 3; a++) { if (l.lo[a] < lo[a]) lo[a] = l.lo[a]; if (l.hi[a] > hi[a]) hi[a] = l.hi[a]; }
    };
    struct Job { int begin, end, node; };
    struct Scratch { std::vector<double> rightArea; std::vector<long long> rightW; std::vector<int> best, tmp; bool wide = false; };
    auto boxOf = [&](int begin, int end, float* lo, float* hi) {
        for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
        for (int k = begin; k < end; k++) grow(lo, hi, leaves[(size_t)idx[(size_t)k]]);
    };
    // one node: the cheapest split over the three axes, its two children added to `tree`, the interior ones pushed on `jobs`
    auto splitJob = [&](const Job j, BinTree& tree, std::vector<Job>& jobs, Scratch& sc) {
        const int cnt = j.end - j.begin;
        double bestCost = INFINITY;
        int bestSplit = -1, bestAxis = -1;
        if (cnt >= 16384 && sc.wide) {  // a big range near the root: its three axes on three threads, the same choice
            struct AxisBest { double cost = INFINITY; int split = -1; std::vector<int> order; };
            AxisBest ab[3];
            auto sweep = [&](int axis) {
                AxisBest& r = ab[axis];
                r.order.assign(idx.begin() + j.begin, idx.begin() + j.end);
                std::sort(r.order.begin(), r.order.end(), [&](int x, int y) {
                    const float cx = leaves[(size_t)x].c[axis], cy = leaves[(size_t)y].c[axis];
                    return cx < cy || (cx == cy && x < y);
                });
                std::vector<double> ra((size_t)cnt + 1, 0.0);
                std::vector<long long> rw((size_t)cnt + 1, 0);
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (int k = cnt - 1; k > 0; k--) {
                    grow(lo, hi, leaves[(size_t)r.order[(size_t)k]]);
                    ra[(size_t)k] = area(lo, hi);
                    rw[(size_t)k] = rw[(size_t)k + 1] + leaves[(size_t)r.order[(size_t)k]].w;
                }
                for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
                long long wl = 0;
                for (int k = 1; k < cnt; k++) {
                    grow(lo, hi, leaves[(size_t)r.order[(size_t)k - 1]]);
                    wl += leaves[(size_t)r.order[(size_t)k - 1]].w;
                    const double cost = area(lo, hi) * (double)wl + ra[(size_t)k] * (double)rw[(size_t)k];
                    if (cost < r.cost) { r.cost = cost; r.split = k; }
                }
            };
            std::thread ty(sweep, 1), tz(sweep, 2);
            sweep(0);
            ty.join();
            tz.join();
            for (int axis = 0; axis < 3; axis++)  // strictly cheaper wins, in axis order: what the loop below does
                if (ab[axis].cost < bestCost) { bestCost = ab[axis].cost; bestSplit = ab[axis].split; bestAxis = axis; }
            if (bestAxis >= 0) sc.best.swap(ab[bestAxis].order);
        } else
        for (int axis = 0; axis < 3; axis++) {
            sc.tmp.assign(idx.begin() + j.begin, idx.begin() + j.end);
            std::sort(sc.tmp.begin(), sc.tmp.end(), [&](int x, int y) {
                const float cx = leaves[(size_t)x].c[axis], cy = leaves[(size_t)y].c[axis];
                return cx < cy || (cx == cy && x < y);
            });
            sc.rightArea.assign((size_t)cnt + 1, 0.0);
            sc.rightW.assign((size_t)cnt + 1, 0);
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int k = cnt - 1; k > 0; k--) {
                grow(lo, hi, leaves[(size_t)sc.tmp[(size_t)k]]);
                sc.rightArea[(size_t)k] = area(lo, hi);
                sc.rightW[(size_t)k] = sc.rightW[(size_t)k + 1] + leaves[(size_t)sc.tmp[(size_t)k]].w;
            }
            for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
            long long wl = 0;
            for (int k = 1; k < cnt; k++) {  // split before position k: [0, k) | [k, cnt)
                grow(lo, hi, leaves[(size_t)sc.tmp[(size_t)k - 1]]);
                wl += leaves[(size_t)sc.tmp[(size_t)k - 1]].w;
                const double cost = area(lo, hi) * (double)wl + sc.rightArea[(size_t)k] * (double)sc.rightW[(size_t)k];
                if (cost < bestCost) { bestCost = cost; bestSplit = k; bestAxis = axis; }
            }
            if (bestAxis == axis) sc.best.swap(sc.tmp);  // keep the order of the axis that holds the best split so far
        }
        if (bestSplit < 0) { bestSplit = cnt / 2; sc.best.assign(idx.begin() + j.begin, idx.begin() + j.end); }  // NaN-free input never gets here
        std::copy(sc.best.begin(), sc.best.end(), idx.begin() + j.begin);
        const int mid = j.begin + bestSplit;
        int child[2];
        const int range[2][2] = {{j.begin, mid}, {mid, j.end}};
        for (int side = 0; side < 2; side++) {
            float lo[3], hi[3];
            boxOf(range[side][0], range[side][1], lo, hi);
            const bool leaf = range[side][1] - range[side][0] == 1;
            child[side] = tree.add(lo, hi, leaf ? leaves[(size_t)idx[(size_t)range[side][0]]].ref : -1);
            if (!leaf) jobs.push_back({range[side][0], range[side][1], child[side]});
        }
        tree.left[(size_t)j.node] = child[0];
        tree.right[(size_t)j.node] = child[1];
    };
    std::vector<Job> jobs;
    Scratch sc0;
    sc0.wide = true;  // the sequential top of the tree may put a node's three axes on three threads
    {
        float lo[3], hi[3];
        boxOf(0, nl, lo, hi);
        if (nl == 1) { t.add(lo, hi, leaves[0].ref); return; }
        jobs.push_back({0, nl, t.add(lo, hi, -1)});
    }
    // Round 3: the top of the tree one node at a time, largest range first, until there are 64 ranges; those are
    // independent (disjoint ranges of idx), so each is built into a tree of its own by whichever thread is free and the
    // pieces are appended in range order: the result does not depend on the threads.
    if (nl >= 4096) {
        while (jobs.size() < 64) {
            size_t big = 0;
            for (size_t k = 1; k < jobs.size(); k++)
                if (jobs[k].end - jobs[k].begin > jobs[big].end - jobs[big].begin) big = k;
            const Job j = jobs[big];
            jobs.erase(jobs.begin() + (long)big);
            splitJob(j, t, jobs, sc0);
        }
        std::sort(jobs.begin(), jobs.end(), [](const Job& a, const Job& b) { return a.begin < b.begin; });
        std::vector<BinTree> piece(jobs.size());
        unsigned nThreads = std::thread::hardware_concurrency();
        nThreads = nThreads < 1 ? 1 : (nThreads > 16 ? 16 : nThreads);
        if (const char* tv = getenv("MIPT_BUILD_THREADS")) nThreads = (unsigned)std::max(1, atoi(tv));
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            Scratch sc;
            std::vector<Job> mine;
            for (size_t k = next.fetch_add(1); k < jobs.size(); k = next.fetch_add(1)) {
                BinTree& lt = piece[k];
                float lo[3], hi[3];
                boxOf(jobs[k].begin, jobs[k].end, lo, hi);
                mine.assign(1, Job{jobs[k].begin, jobs[k].end, lt.add(lo, hi, -1)});  // local node 0 stands for jobs[k].node
                while (!mine.empty()) {
                    const Job j = mine.back();
                    mine.pop_back();
                    splitJob(j, lt, mine, sc);
                }
            }
        };
        std::vector<std::thread> pool;
        for (unsigned k = 1; k < nThreads; k++) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
        for (size_t k = 0; k < jobs.size(); k++) {
            const BinTree& lt = piece[k];
            const int base = t.size() - 1;  // local node i >= 1 becomes base + i
            for (int i = 1; i < lt.size(); i++) t.add(&lt.mn[3 * (size_t)i], &lt.mx[3 * (size_t)i], lt.refLeaf[(size_t)i]);
            auto g = [&](int i) { return i == 0 ? jobs[k].node : base + i; };
            for (int i = 0; i < lt.size(); i++)
                if (lt.refLeaf[(size_t)i] < 0) { t.left[(size_t)g(i)] = g(lt.left[(size_t)i]); t.right[(size_t)g(i)] = g(lt.right[(size_t)i]); }
        }
        return;
    }
    while (!jobs.empty()) {
        const Job j = jobs.back();
        jobs.pop_back();
        splitJob(j, t, jobs, sc0);
    }
}

// Insertion-based optimisation of a binary hierarchy (after Bittner, Hapala, Havran 2013): take a subtree out (its
// parent goes with it, the sibling moves up), find by branch and bound the place where putting it back adds the least
// surface area to the tree (the new parent's box plus what the ancestors grow by), put it there.  Putting it back where
// it was is one of the candidates, so the sum of interior box areas -- the expected number of node visits -- never
// grows.  Nodes are tried largest box first, in passes, until a pass gains under half a per cent.  Afterwards the
// nodes are renumbered parents first (the collapse sweeps rely on children having larger ids).
// Round 3: the passes run per SUBTREE, the subtrees in parallel.  The tree is cut below its top (the largest subtree is
// opened until there are 64 of them, or fewer for small trees); a subtree keeps its leaves, so the box of its root
// -- whichever node ends up there -- stays what it was and the nodes above the cut are untouched.  What is lost are moves
// ACROSS the cut (after an all-axes SAH build there are few: measured below); what is gained is the host's cores, the
// pass over the 92,501 nodes of the bench scene being the bulk of mipt_scene_create.  Each subtree is worked on by one
// thread with its own passes and its own stopping rule, and no thread touches another's nodes: the result does not
// depend on the number of threads or on their timing.
static void optimise_top(BinTree& t) {
    const int n = t.size();
    if (n < 7 || n > 2000000) return;  // nothing to move / more than a million leaves: the passes would take minutes, the SAH build alone has to do
    std::vector<int> parent((size_t)n, -1);
    for (int i = 0; i < n; i++)
        if (t.refLeaf[(size_t)i] < 0) { parent[(size_t)t.left[(size_t)i]] = i; parent[(size_t)t.right[(size_t)i]] = i; }
    auto areaOf = [&](int v) -> double {
        const double ex = (double)t.mx[3 * (size_t)v] - t.mn[3 * (size_t)v], ey = (double)t.mx[3 * (size_t)v + 1] - t.mn[3 * (size_t)v + 1],
                     ez = (double)t.mx[3 * (size_t)v + 2] - t.mn[3 * (size_t)v + 2];
        return ex * ey + ey * ez + ez * ex;
    };
    auto unionArea = [&](int a, int b) -> double {
        double e[3];
        for (int k = 0; k < 3; k++)
            e[k] = (double)std::max(t.mx[3 * (size_t)a + k], t.mx[3 * (size_t)b + k]) - std::min(t.mn[3 * (size_t)a + k], t.mn[3 * (size_t)b + k]);
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    };
    auto refit = [&](int v) {  // v and its ancestors up to the root of its subtree: box = union of the children's
        for (; v >= 0; v = parent[(size_t)v]) {
            const int l = t.left[(size_t)v], r = t.right[(size_t)v];
            for (int k = 0; k < 3; k++) {
                t.mn[3 * (size_t)v + k] = std::min(t.mn[3 * (size_t)l + k], t.mn[3 * (size_t)r + k]);
                t.mx[3 * (size_t)v + k] = std::max(t.mx[3 * (size_t)l + k], t.mx[3 * (size_t)r + k]);
            }
        }
    };
    // the cut: open the subtree with the most nodes until there are enough of them (sizes by a sweep over the children-
    // after-parents numbering both inputs have)
    std::vector<int> size((size_t)n, 1);
    for (int i = n - 1; i >= 0; i--)
        if (t.refLeaf[(size_t)i] < 0) size[(size_t)i] = 1 + size[(size_t)t.left[(size_t)i]] + size[(size_t)t.right[(size_t)i]];
    const bool stats = getenv("MIPT_WIDE_STATS") != nullptr;
    double areaBefore = 0.0;
    if (stats) for (int i = 0; i < n; i++) if (t.refLeaf[(size_t)i] < 0) areaBefore += areaOf(i);
    int root = 0;
    // two rounds: 64 subtrees (cheap: small search spaces), then 16 with at most two passes each, which lets nodes cross
    // the first round's cuts
    const int nPhases = n < 8192 ? 1 : 2;
    for (int phase = 0; phase < nPhases; phase++) {
    const int maxPasses = phase == 0 ? 8 : 2;
    if (phase > 0) {  // sizes of the tree as it stands (ids are no longer parents-first: walk it)
        std::vector<int> order, st{root};
        while (!st.empty()) {
            const int v = st.back();
            st.pop_back();
            order.push_back(v);
            if (t.refLeaf[(size_t)v] < 0) { st.push_back(t.left[(size_t)v]); st.push_back(t.right[(size_t)v]); }
        }
        for (size_t k = order.size(); k-- > 0;) {
            const int v = order[k];
            size[(size_t)v] = t.refLeaf[(size_t)v] < 0 ? 1 + size[(size_t)t.left[(size_t)v]] + size[(size_t)t.right[(size_t)v]] : 1;
        }
    }
    std::vector<int> roots{root};
    const size_t wanted = n < 8192 ? 1 : (phase == 0 ? 64 : 16);
    while (roots.size() < wanted) {
        size_t big = 0;
        for (size_t k = 1; k < roots.size(); k++)
            if (size[(size_t)roots[k]] > size[(size_t)roots[big]]) big = k;
        const int v = roots[big];
        if (t.refLeaf[(size_t)v] >= 0 || size[(size_t)v] < 64) break;
        roots[big] = t.left[(size_t)v];
        roots.push_back(t.right[(size_t)v]);
    }
    std::sort(roots.begin(), roots.end(), [&](int a, int b) { return size[(size_t)a] > size[(size_t)b] || (size[(size_t)a] == size[(size_t)b] && a < b); });

    // one subtree: detached from its parent while it is worked on, hung back (under whichever node is its root then) after
    auto optimiseSubtree = [&](const int R) {
        const int up = parent[(size_t)R];
        const bool wasLeft = up >= 0 && t.left[(size_t)up] == R;
        parent[(size_t)R] = -1;
        int root = R;
        std::vector<int> nodes;  // the subtree's nodes: the set stays, the shape changes
        {
            std::vector<int> st{R};
            while (!st.empty()) {
                const int v = st.back();
                st.pop_back();
                nodes.push_back(v);
                if (t.refLeaf[(size_t)v] < 0) { st.push_back(t.right[(size_t)v]); st.push_back(t.left[(size_t)v]); }
            }
        }
        if (nodes.size() >= 7) {
            auto total = [&]() { double c = 0.0; for (const int i : nodes) if (t.refLeaf[(size_t)i] < 0) c += areaOf(i); return c; };
            std::vector<std::pair<double, int>> keyed(nodes.size());
            std::priority_queue<std::pair<double, int>> pq;  // (-induced cost, node)
            double before = total();
            for (int pass = 0; pass < maxPasses; pass++) {
                for (size_t i = 0; i < nodes.size(); i++) keyed[i] = {areaOf(nodes[i]), nodes[i]};
                std::sort(keyed.begin(), keyed.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) {
                    return a.first > b.first || (a.first == b.first && a.second < b.second);
                });
                for (const auto& kn : keyed) {
                    const int N = kn.second;
                    const int P = parent[(size_t)N];
                    if (P < 0) continue;                         // the root stays
                    const int G = parent[(size_t)P];
                    if (G < 0 && t.refLeaf[(size_t)(t.left[(size_t)P] == N ? t.right[(size_t)P] : t.left[(size_t)P])] >= 0) continue;  // two nodes left: nothing to choose
                    const int S = t.left[(size_t)P] == N ? t.right[(size_t)P] : t.left[(size_t)P];
                    // take N (and P) out
                    parent[(size_t)S] = G;
                    if (G < 0) root = S;
                    else {
                        if (t.left[(size_t)G] == P) t.left[(size_t)G] = S; else t.right[(size_t)G] = S;
                        refit(G);
                    }
                    // where does N go back?
                    const double aN = areaOf(N);
                    double best = INFINITY;
                    int bestX = S;
                    while (!pq.empty()) pq.pop();
                    pq.push({-0.0, root});
                    while (!pq.empty()) {
                        const double ci = -pq.top().first;
                        const int X = pq.top().second;
                        pq.pop();
                        if (ci + aN >= best) break;
                        const double ct = ci + unionArea(X, N);
                        if (ct < best) { best = ct; bestX = X; }
                        if (t.refLeaf[(size_t)X] < 0) {
                            const double cc = ct - areaOf(X);
                            if (cc + aN < best) { pq.push({-cc, t.left[(size_t)X]}); pq.push({-cc, t.right[(size_t)X]}); }
                        }
                    }
                    // P becomes the parent of bestX and N, in bestX's place
                    const int X = bestX, XP = parent[(size_t)X];
                    t.left[(size_t)P] = X; t.right[(size_t)P] = N;
                    parent[(size_t)X] = P; parent[(size_t)N] = P; parent[(size_t)P] = XP;
                    if (XP < 0) root = P;
                    else if (t.left[(size_t)XP] == X) t.left[(size_t)XP] = P; else t.right[(size_t)XP] = P;
                    refit(P);
                }
                const double after = total();
                const bool done = !(after < before * 0.995);
                before = after;
                if (done) break;
            }
        }
        parent[(size_t)root] = up;
        if (up >= 0) { if (wasLeft) t.left[(size_t)up] = root; else t.right[(size_t)up] = root; }
        return root;
    };
    if (roots.size() == 1) {
        root = optimiseSubtree(root);
    } else {
        unsigned nThreads = std::thread::hardware_concurrency();
        nThreads = nThreads < 1 ? 1 : (nThreads > 16 ? 16 : nThreads);
        if (const char* tv = getenv("MIPT_BUILD_THREADS")) nThreads = (unsigned)std::max(1, atoi(tv));
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t k = next.fetch_add(1); k < roots.size(); k = next.fetch_add(1)) (void)optimiseSubtree(roots[k]);
        };
        std::vector<std::thread> pool;
        for (unsigned k = 1; k < nThreads; k++) pool.emplace_back(worker);
        worker();
        for (auto& th : pool) th.join();
    }
    }  // phases
    if (stats) {
        double areaAfter = 0.0;
        for (int i = 0; i < n; i++) if (t.refLeaf[(size_t)i] < 0) areaAfter += areaOf(i);
        fprintf(stderr, "optimise_top: sum of interior areas %.6g -> %.6g\n", areaBefore, areaAfter);
    }
    // renumber: parents before children, root = 0
    BinTree o;
    std::vector<int> stack{root}, newId((size_t)n, -1), oldOf;
    while (!stack.empty()) {
        const int v = stack.back();
        stack.pop_back();
        newId[(size_t)v] = o.add(&t.mn[3 * (size_t)v], &t.mx[3 * (size_t)v], t.refLeaf[(size_t)v]);
        oldOf.push_back(v);
        if (t.refLeaf[(size_t)v] < 0) { stack.push_back(t.right[(size_t)v]); stack.push_back(t.left[(size_t)v]); }
    }
    for (int i = 0; i < o.size(); i++) {
        const int v = oldOf[(size_t)i];
        if (t.refLeaf[(size_t)v] < 0) { o.left[(size_t)i] = newId[(size_t)t.left[(size_t)v]]; o.right[(size_t)i] = newId[(size_t)t.right[(size_t)v]]; }
    }
    t = o;
}

// Builds the 4-wide node array from a binary tree over the reference's leaves -- the reference's own upper tree, or,
// with `rebuilt`, the hierarchy of rebuild_top: a wide node stands for a binary interior node, its (up
// to 4) slots are that node's grandchildren, a child that is a leaf taking one slot itself (or, by default, the
// result of opening the largest box until four slots are taken).  Wide nodes are numbered in the order their
// parents are opened, largest box first, so the interior children of a node are consecutive (childBase + ordinal)
// and the nodes met most often come first (they are the ones k_trace_wide keeps in LDS); the leaves get ids of our own, again consecutive per wide
// node (leafBase + ordinal).  Returns false, leaving the scene on the literal walk, when the input is not the
// properly nested binary tree the order-free traversal relies on (see kernels_trace.hip) or exceeds its limits.
static bool build_wide(const mipt_scene_desc* d, mipt_scene* s, const int collapse, const int top) {
    constexpr int K = ptd::kWideK;  // slots per wide node
    const bool greedy = collapse == 1, rebuilt = top > 0;  // top: 0 the reference's upper tree, 1 rebuild_top, 2 + optimise_top
    s->wnodes.clear(); s->leafBox.clear(); s->leafHead.clear();
    if (!s->planesTame) return false;
    BinTree bt;
    {
        // the premise about the REFERENCE's tree, whichever hierarchy is walked: a proper binary tree, boxes nested
        const int n = d->n_nodes;
        std::vector<char> seen((size_t)n, 0);
        std::vector<int> roots;
        if (!forest_roots(d, roots)) return false;
        if (roots.size() > 1 && !rebuilt) return false;  // several objects: only a hierarchy of our own joins them under one root
        for (const int r : roots) seen[(size_t)r] = 1;
        for (int i = 0; i < n; i++) bt.add(&d->node_min[3 * i], &d->node_max[3 * i], d->leaf_count[i] > 0 ? i : -1);
        for (int i = 0; i < n; i++) {
            if (d->leaf_count[i] > 0) continue;
            const int l = d->id_hit[i];
            if (!(l > i && l < n)) return false;
            const int r = d->id_miss[l];
            if (!(r > l && r < n)) return false;
            if (seen[l] || seen[r]) return false;  // every node has exactly one parent
            if (d->id_miss[r] != d->id_miss[i]) return false;  // ... and the links are the threading of that tree (cs488.h:1655-1658)
            seen[l] = seen[r] = 1;
            bt.left[(size_t)i] = l; bt.right[(size_t)i] = r;
            for (const int c : {l, r})
                for (int a = 0; a < 3; a++)  // exact nesting: the child's box lies inside the parent's
                    if (!(d->node_min[3 * i + a] <= d->node_min[3 * c + a] && d->node_max[3 * c + a] <= d->node_max[3 * i + a])) return false;
        }
        for (int i = 0; i < n; i++)
            if (!seen[i]) return false;
        if (rebuilt) {
            bt = BinTree();
            const auto t0 = std::chrono::steady_clock::now();
            rebuild_top(d, bt);
            const auto t1 = std::chrono::steady_clock::now();
            if (top > 1) optimise_top(bt);
            if (getenv("MIPT_WIDE_STATS"))
                fprintf(stderr, "build_wide: SAH over the leaves %.3f s, insertion optimisation %.3f s\n", std::chrono::duration<double>(t1 - t0).count(),
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
        }
    }
    const int n = bt.size();
    auto isLeaf = [&](int i) { return bt.refLeaf[(size_t)i] >= 0; };
    const std::vector<int>& left = bt.left;
    const std::vector<int>& right = bt.right;
    const float* node_min = bt.mn.data();
    const float* node_max = bt.mx.data();
    struct W { int bnode, depth, childBase; };
    std::vector<W> wide;  // in id order
    std::vector<int> slotsOf;  // 4 per wide node: binary node of the slot or -1
    wide.push_back({0, 0, 0});
    int nLeaves = 0;
    auto boxArea = [&](int b) -> double {
        const double ex = (double)node_max[3 * b] - node_min[3 * b], ey = (double)node_max[3 * b + 1] - node_min[3 * b + 1],
                     ez = (double)node_max[3 * b + 2] - node_min[3 * b + 2];
        return ex * ey + ey * ez + ez * ex;
    };
    // Nodes are opened largest box first and the children of a node get the next free ids, so that the ids below
    // kWideTopNodes (the ones k_trace_wide keeps in LDS) are the nodes rays meet most often -- on a balanced tree this
    // is breadth-first order -- while the children of a node stay consecutive.
    std::vector<int> triCount((size_t)n, 0);  // triangles below each binary node (in both trees children have larger ids than their parent)
    for (int i = n - 1; i >= 0; i--) triCount[i] = isLeaf(i) ? d->leaf_count[bt.refLeaf[(size_t)i]] : triCount[left[i]] + triCount[right[i]];
    // collapse == 2: the collapse that minimises the expected number of node visits, sum of area(wide node) -- every
    // visit costs the same whatever it finds, and the leaves' own terms are fixed.  T[v][k] = least cost of standing
    // for the subtree of v with at most k slots of a wide node: a leaf costs nothing; an interior v either becomes a
    // wide node itself (one slot; area(v) + the best split of ITS four slots over its children) or, given k >= 2 slots,
    // is opened in place and shares them out between its children.  Children have larger ids: one backward sweep.
    std::vector<double> T;
    std::vector<unsigned char> how;  // per (v, k): 0 = v is a slot itself, i > 0 = opened, i slots to the left child; [v][0]: the split as a wide node
    if (collapse == 2) {
        T.assign((size_t)(K + 1) * n, 0.0);
        how.assign((size_t)(K + 1) * n, 0);
        for (int v = n - 1; v >= 0; v--) {
            if (isLeaf(v)) continue;
            const int l = left[v], r = right[v];
            double best = INFINITY;
            int bi = 1;
            for (int i = 1; i < K; i++) {
                const double c = T[(size_t)(K + 1) * l + i] + T[(size_t)(K + 1) * r + K - i];
                if (c < best) { best = c; bi = i; }
            }
            const double self = boxArea(v) + best;
            how[(size_t)(K + 1) * v] = (unsigned char)bi;
            T[(size_t)(K + 1) * v + 1] = self;
            for (int k = 2; k <= K; k++) {
                double bk = self;
                int hk = 0;
                for (int i = 1; i < k; i++) {
                    const double c = T[(size_t)(K + 1) * l + i] + T[(size_t)(K + 1) * r + k - i];
                    if (c < bk) { bk = c; hk = i; }
                }
                T[(size_t)(K + 1) * v + k] = bk;
                how[(size_t)(K + 1) * v + k] = (unsigned char)hk;
            }
        }
    }
    const char* const slotOrderEnv = getenv("MIPT_WIDE_SLOT_ORDER");  // read once, not once per wide node
    const std::string slotOrder = slotOrderEnv ? slotOrderEnv : (d->n_lights > 0 ? "lightnear" : "big");
    std::priority_queue<std::pair<double, int>> open;
    open.push({boxArea(0), 0});
    while (!open.empty()) {
        const size_t w = (size_t)open.top().second;
        open.pop();
        const int b = wide[w].bnode;
        int sl[K], ns = 0;
        for (int k = 0; k < K; k++) sl[k] = -1;
        if (isLeaf(b)) {
            sl[ns++] = b;  // a tree that is one leaf: the root wide node has that leaf as its only slot
        } else if (collapse == 2) {
            struct Item { int v, k; };
            Item st[2 * K];
            int top = 0;
            const int i0 = how[(size_t)(K + 1) * b];
            st[top++] = {right[b], K - i0};
            st[top++] = {left[b], i0};
            while (top > 0) {
                const Item it = st[--top];
                const int h = (isLeaf(it.v) || it.k == 1) ? 0 : how[(size_t)(K + 1) * it.v + it.k];
                if (h == 0) { sl[ns++] = it.v; continue; }
                st[top++] = {right[it.v], it.k - h};
                st[top++] = {left[it.v], h};
            }
        } else if (greedy) {
            // open the slot with the largest box (the one most rays pierce) until the four slots are taken
            sl[ns++] = left[b]; sl[ns++] = right[b];
            while (ns < K) {
                int pick = -1;
                double best = -1.0;
                for (int k = 0; k < ns; k++) {
                    if (isLeaf(sl[k])) continue;
                    const double ex = (double)node_max[3 * sl[k]] - node_min[3 * sl[k]];
                    const double ey = (double)node_max[3 * sl[k] + 1] - node_min[3 * sl[k] + 1];
                    const double ez = (double)node_max[3 * sl[k] + 2] - node_min[3 * sl[k] + 2];
                    const double area = ex * ey + ey * ez + ez * ex;
                    if (area > best) { best = area; pick = k; }
                }
                if (pick < 0) break;
                const int c = sl[pick];
                sl[pick] = left[c];
                sl[ns++] = right[c];
            }
        } else {
            for (const int c : {left[b], right[b]}) {
                if (isLeaf(c)) sl[ns++] = c;
                else { sl[ns++] = left[c]; sl[ns++] = right[c]; }
            }
        }
        {
            // Slot order.  A ray takes the lowest hit slot first; for closest-hit rays the order changes nothing (they
            // visit every hit slot), an any-hit ray ends at its first accepted triangle.  Round 1: the largest box first
            // (55.9 -> 42.7 slot tests per shadow ray of the bench on four-slot nodes).  Round 2: shadow rays all END AT
            // A LIGHT, so whatever stands close to a light is crossed by rays from everywhere: the slot whose box centre
            // is nearest to a light goes first ("lightnear": 59.5 -> 53.3 slot tests and 2.66 -> 2.34 triangle tests per
            // shadow ray of the bench on eight-slot nodes, 1.31 -> 1.13 ms per launch; 67.2 -> 55.5 on the 257 k-triangle
            // atrium; bunny 66.9 -> 65.4, testObj 37.6 -> 36.8: never worse than "big").  Ordering by the solid angle a
            // box subtends at the light (58.1), by distance descending (69.9), by triangle count (66.4 / 56.4 ascending)
            // or leaving the tree's order (56.9) all lose to it.  A scene without lights keeps "big".
            // MIPT_WIDE_SLOT_ORDER = big | small | count | fewest | tree | treerev | lightnear | lightfar | lightangle.
            const std::string& mode = slotOrder;
            if (mode == "big") std::stable_sort(sl, sl + ns, [&](int a, int b) { return boxArea(a) > boxArea(b); });
            else if (mode == "small") std::stable_sort(sl, sl + ns, [&](int a, int b) { return boxArea(a) < boxArea(b); });
            else if (mode == "count") std::stable_sort(sl, sl + ns, [&](int a, int b) { return triCount[a] > triCount[b]; });
            else if (mode == "treerev") std::reverse(sl, sl + ns);
            else if (mode == "lightangle" && d->n_lights > 0) {
                // shadow rays all end at a light: the share of them that pierce a box is about the solid angle the box
                // subtends there -- its area over the squared distance (summed over the lights), largest first
                auto angle = [&](int b) {
                    double q = 0.0;
                    for (int l = 0; l < d->n_lights; l++) {
                        double r2 = 0.0;
                        for (int a = 0; a < 3; a++) {
                            const double lo = node_min[3 * b + a], hi = node_max[3 * b + a], p = d->lights[l].position[a];
                            const double c = p < lo ? lo - p : (p > hi ? p - hi : 0.0);  // distance to the box, 0 inside
                            r2 += c * c;
                        }
                        const double ex = (double)node_max[3 * b] - node_min[3 * b], ey = (double)node_max[3 * b + 1] - node_min[3 * b + 1],
                                     ez = (double)node_max[3 * b + 2] - node_min[3 * b + 2];
                        const double diag2 = ex * ex + ey * ey + ez * ez;
                        q += boxArea(b) / (r2 + 0.25 * diag2 + 1e-30);
                    }
                    return q;
                };
                std::stable_sort(sl, sl + ns, [&](int a, int b) { return angle(a) > angle(b); });
            }
            else if ((mode == "lightfar" || mode == "lightnear") && d->n_lights > 0) {
                auto dist2 = [&](int b) {  // squared distance from the box centre to the nearest light
                    double best = INFINITY;
                    for (int l = 0; l < d->n_lights; l++) {
                        double q = 0.0;
                        for (int a = 0; a < 3; a++) {
                            const double c = 0.5 * ((double)node_min[3 * b + a] + node_max[3 * b + a]) - d->lights[l].position[a];
                            q += c * c;
                        }
                        if (q < best) best = q;
                    }
                    return best;
                };
                const bool far = mode == "lightfar";
                std::stable_sort(sl, sl + ns, [&](int a, int b) { return far ? dist2(a) > dist2(b) : dist2(a) < dist2(b); });
            }
            else if (mode == "fewest") std::stable_sort(sl, sl + ns, [&](int a, int b) { return triCount[a] < triCount[b]; });
        }
        if (K == 8) std::stable_partition(sl, sl + ns, [&](int a) { return !isLeaf(a); });  // eight-slot records: interior slots first (device_scene.hpp)
        if (slotsOf.size() < (size_t)K * wide.size()) slotsOf.resize((size_t)K * wide.size(), -1);
        wide[w].childBase = (int)wide.size();
        for (int k = 0; k < K; k++) {
            slotsOf[(size_t)K * w + k] = sl[k];
            if (sl[k] >= 0 && !isLeaf(sl[k])) {
                open.push({boxArea(sl[k]), (int)wide.size()});
                wide.push_back({sl[k], wide[w].depth + 1, 0});
            }
        }
        if (wide[w].depth >= ptd::kWideMaxDepth) return false;
        if (wide.size() >= ((size_t)1 << 22)) return false;
    }
    const int nw = (int)wide.size();
    slotsOf.resize((size_t)K * nw, -1);
    s->wnodes.assign((size_t)ptd::kWideNodeStrideF4 * nw, make_float4(0, 0, 0, 0));
    // fp16 holds magnitudes up to 65504 and is coarse below 6e-5: the planes are stored times a power of two S that
    // brings the largest magnitude of the tree into [16384, 32768) -- exact in fp32, undone exactly in the kernel (it
    // multiplies the ray's reciprocals by 1 / S), so every tame scene keeps fp16's full relative precision
    float planeScale = 1.0f;
    {
        float big = 0.0f;
        for (int i = 0; i < n; i++)
            for (int a = 0; a < 3; a++) big = std::max(big, std::max(std::fabs(node_min[3 * i + a]), std::fabs(node_max[3 * i + a])));
        if (big > 0.0f && std::isfinite(big)) {
            int e = 0;
            (void)std::frexp(big, &e);  // big = m 2^e, m in [0.5, 1)
            planeScale = std::ldexp(1.0f, 15 - e);
        }
    }
#if !PT_WIDE_HALF
    planeScale = 1.0f;  // fp32 planes are stored as they are
#endif
    s->wideInvScale = 1.0f / planeScale;
    // second pass: records (children of wide node w start at its childBase, in the order the first pass appended them)
    int nextChild = 1;
    for (int w = 0; w < nw; w++) {
        const int* sl = &slotsOf[(size_t)K * w];
        int interiorMask = 0, leafMask = 0, childBase = wide[w].childBase, leafBase = nLeaves;
        float lo[3][K], hi[3][K];
        for (int k = 0; k < K; k++) {
            const int src = sl[k] >= 0 ? sl[k] : sl[0];  // an empty slot repeats slot 0's (tame) box; the masks leave it out
            for (int a = 0; a < 3; a++) { lo[a][k] = node_min[3 * src + a] * planeScale; hi[a][k] = node_max[3 * src + a] * planeScale; }
            if (sl[k] < 0) continue;
            if (isLeaf(sl[k])) {
                leafMask |= 1 << k;
                const int b = bt.refLeaf[(size_t)sl[k]];  // the reference's leaf: its exact box decides admission
                s->leafBox.push_back(make_float4(d->node_min[3 * b], d->node_max[3 * b], d->node_min[3 * b + 1], d->node_max[3 * b + 1]));
                s->leafBox.push_back(make_float4(d->node_min[3 * b + 2], d->node_max[3 * b + 2],
                                                 __builtin_bit_cast(float, d->leaf_first[b]), 0.0f));
                for (int r = 0; r < 3; r++) s->leafHead.push_back(s->leafTris[(size_t)3 * d->leaf_first[b] + r]);  // its first triangle
                s->leafHead.back().w = __builtin_bit_cast(float, d->leaf_first[b]);  // ... tagged with its leafTris index
                for (int r = 3; r < ptd::kLeafHeadStrideF4; r++) s->leafHead.push_back(make_float4(0, 0, 0, 0));  // padding to the record stride
                nLeaves++;
            } else {
                interiorMask |= 1 << k;
                nextChild++;
            }
        }
        float4* rec = &s->wnodes[(size_t)ptd::kWideNodeStrideF4 * w];
#if PT_WIDE_HALF
        for (int a = 0; a < 3; a++) {  // row a: {lo of slots 0..K-1, hi of slots 0..K-1} as fp16, each rounded outwards
            unsigned short h[2 * K];
            for (int k = 0; k < K; k++) {
                h[k] = half_towards(lo[a][k], false);
                h[K + k] = half_towards(hi[a][k], true);
                // the slot test must never be tighter than the box: checked, not assumed (and no infinities)
                if (!(half_value(h[k]) <= lo[a][k]) || !(half_value(h[K + k]) >= hi[a][k]) || (h[k] & 0x7fffu) >= 0x7c00u ||
                    (h[K + k] & 0x7fffu) >= 0x7c00u)
                    return false;  // a plane beyond the fp16 range: this scene takes the literal walk
                // An EMPTY slot of an eight-slot node holds the box no ray can hit: lo = +inf, hi = -inf on every axis.  Whatever
                // the ray (its reciprocals are finite and nonzero, its constants finite: fast_div_ok), the near quotient
                // comes out +inf and the far one -inf, so the slot test rejects it without a mask of the used slots
                // (kernels_trace_wide.hip keeps only the count of interior slots).  Four-slot nodes keep slot 0's box and their masks.
                if (K == 8 && sl[k] < 0) { h[k] = 0x7c00u; h[K + k] = 0xfc00u; }
            }
            memcpy((char*)rec + 4 * K * a, h, 4 * K);
        }
        if (K == 8) {  // interior slots first: counts instead of masks
            const int nInt = __builtin_popcount((unsigned)interiorMask), nLf = __builtin_popcount((unsigned)leafMask);
            if (interiorMask != (1 << nInt) - 1 || leafMask != (((1 << nLf) - 1) << nInt) || nLeaves >= (1 << 24)) return false;
            rec[6] = make_float4(__builtin_bit_cast(float, childBase), __builtin_bit_cast(float, leafBase),
                                 __builtin_bit_cast(float, nInt | (nLf << 8)), 0.0f);
        } else {
            rec[3] = make_float4(__builtin_bit_cast(float, childBase), __builtin_bit_cast(float, leafBase),
                                 __builtin_bit_cast(float, interiorMask | (leafMask << 4)), 0.0f);
        }
#else
        for (int a = 0; a < 3; a++) {
            rec[2 * a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
            rec[2 * a + 1] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
        }
        rec[6] = make_float4(__builtin_bit_cast(float, childBase), __builtin_bit_cast(float, leafBase),
                             __builtin_bit_cast(float, interiorMask | (leafMask << 4)), 0.0f);
#endif
    }
    if (nextChild != nw) return false;
    s->nWide = nw;
    if (getenv("MIPT_WIDE_STATS")) {  // expected visits of a random ray that hits the root: sum of box areas over the root's
        double visits = 0.0, leafOpens = 0.0;
        int maxDepth = 0;
        for (int w = 0; w < nw; w++) {
            visits += boxArea(wide[(size_t)w].bnode);
            if (wide[(size_t)w].depth > maxDepth) maxDepth = wide[(size_t)w].depth;
            for (int k = 0; k < K; k++)
                if (slotsOf[(size_t)K * w + k] >= 0 && isLeaf(slotsOf[(size_t)K * w + k])) leafOpens += boxArea(slotsOf[(size_t)K * w + k]);
        }
        fprintf(stderr, "%d-wide tree (%s): %d nodes, %d leaves, depth %d, expected visits %.3f, expected leaf opens %.3f\n",
                K, top == 2 ? "optimised top" : top == 1 ? "rebuilt top" : "reference top", nw, nLeaves, maxDepth, visits / boxArea(0), leafOpens / boxArea(0));
    }
    return true;
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return MIPT_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        HIP_TRY(hipMalloc(&p, need));
        bytes = need;
        return MIPT_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct TimedLaunch {
    hipEvent_t a, b;
    int cls;
};

struct mipt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    void* arena = nullptr;
    size_t arenaBytes = 0;
    DevScene ds{};
    // scratch of the host-pointer entry points
    DevBuf sRays, sHits, sOut, sAux, sAux2;
    DevBuf slow;           // rays the wide kernel handed back (trace_closest)
    DevBuf slowOcc;        // ... and those of trace_occluded, which may run beside a closest-hit launch on the side stream
    int slowParity = 0, slowOccParity = 0;  // which of a list's two headers the next launch uses (ensure_slow)
    // Path mode overlaps occluded(it) with closest(it+1): once shade(it) has run the two are independent, and on a
    // second stream the blocks of the later kernel move onto CUs as the earlier one's drain, which fills the
    // 0.1-0.2 ms tail of every traversal launch (MIPT_OVERLAP=0 in the environment turns it off for A/B runs).
    hipStream_t side = nullptr;
    bool overlap = true;
    bool fuse = true;         // ... or, better, in ONE launch (k_trace_wide_both; MIPT_FUSE=0 goes back to two streams)
    bool shadowFirst = false; // order of the two queues inside the fused launch (MIPT_FUSE_ORDER=shadow)
    std::vector<hipEvent_t> evShaded, evOccluded;  // per bounce iteration, created on first use (no timing)
    unsigned char* scratch = nullptr;  // this GPU's copy of the file-scope `scratch` array (mipt_trace_device_noalloc)
    bool useWide = true;   // MIPT_TRAVERSAL=walk in the environment forces the literal walk (A/B runs)
    bool useBitTrail = false;  // mipt_context_set_traversal(MIPT_TRAVERSAL_BITTRAIL)
    bool hasSpecular = false;  // copied from the scene
    // integrator state
    DevBuf pathMem, tileMem, frameMem;
    PathBuffers pb{};
    std::vector<int> tileKey;
    int nOwnedTiles = 0;
    size_t pbPixels = 0;
    bool countersClean = false;  // pb.counters are all zero (k_fold_counters leaves them so at the end of a frame)
    // stats
    bool profiling = false, counting = false;
    unsigned long long* dTotals = nullptr;  // [0] ext rays [1] shadow rays [2..3] closest visits/tests [4..5] occluded
    mipt_stats acc{};
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> freeEvents;
};

namespace {

int fail(int code, const std::string& msg) {
    pt::set_error(msg);
    return code;
}

// nothing may throw across the C boundary: every extern "C" body that allocates host memory runs inside this
template <class F>
int guarded(const char* what, F&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(MIPT_ERR_NOMEM, std::string(what) + ": out of host memory");
    } catch (const std::exception& e) {
        return fail(MIPT_ERR_INVALID, std::string(what) + ": " + e.what());
    } catch (...) {
        return fail(MIPT_ERR_INVALID, std::string(what) + ": unknown exception");
    }
}

struct Timer {
    mipt_ctx* c;
    hipEvent_t a = nullptr, b = nullptr;
    int cls;
    hipStream_t st;
    Timer(mipt_ctx* ctx, int klass, hipStream_t stream = nullptr) : c(ctx), cls(klass), st(stream ? stream : ctx->stream) {
        if (!c->profiling) return;
        a = take();
        b = take();
        if (a && b) (void)hipEventRecord(a, st);
    }
    hipEvent_t take() {
        if (!c->freeEvents.empty()) {
            hipEvent_t e = c->freeEvents.back();
            c->freeEvents.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    ~Timer() {
        if (!c->profiling || !a || !b) return;
        (void)hipEventRecord(b, st);
        c->timed.push_back(TimedLaunch{a, b, cls});
    }
};

void count_launch(mipt_ctx* c, int cls) {
    if (cls == 0) c->acc.launches_closest++;
    else if (cls == 1) c->acc.launches_occluded++;
    else {
        c->acc.launches_other++;
        if (cls == 3) c->acc.launches_shade++;
    }
}

// cursor == nullptr: use (and zero) the context's own cursor word.
// Scenes that qualify (DevScene::nWide) go through the wide kernel first; it hands back the rays it may not take
// (direction components that are zero, denormal, non-finite ...) in c->slow, and the literal walk traces those.
// c->slow: [0] count, [1] cursor of the second launch, [2..] queue slots.
// c->slow / c->slowOcc: TWO headers of four words used in turn ([0] count, [1] cursor of the list-mode launch, [2] blocks of
// the wide launch that are done, [3] -), then the queue slots from word kSlowHeader on.  A wide launch zeroes the OTHER
// header for the launch after it, so nothing is memset between launches.  The list RESTS at -1 everywhere: whoever
// takes an entry puts -1 back (k_trace_wide's draining wavefronts tell "not written yet / taken" from "to do" by that),
// so a fresh buffer is filled once.
constexpr size_t kSlowHeader = 8;
struct SlowView {
    unsigned* head;  // this launch's header
    unsigned* next;  // the one it zeroes
    int* list;
};
// use: a launch follows (the headers change places); false: only make sure the buffer is large enough
int ensure_slow(mipt_ctx* c, DevBuf& buf, int& parity, size_t capacity, hipStream_t st, SlowView& v, bool use = true) {
    const void* before = buf.p;
    int rc = buf.ensure((capacity + kSlowHeader) * sizeof(int));
    if (rc != MIPT_OK) return rc;
    if (buf.p != before) {
        if (hipMemsetAsync(buf.p, 0xff, buf.bytes, st) != hipSuccess || hipMemsetAsync(buf.p, 0, kSlowHeader * sizeof(int), st) != hipSuccess)
            return fail(MIPT_ERR_DEVICE, "hipMemsetAsync(hand-back list) failed");
        parity = 0;
    }
    unsigned* base = (unsigned*)buf.p;
    v.head = base + 4 * parity;
    v.next = base + 4 * (parity ^ 1);
    v.list = (int*)(base + kSlowHeader);
    if (use) parity ^= 1;
    return MIPT_OK;
}
// k_trace_wide empties SHORT hand-back lists itself (kernels_trace_wide.hip, "Draining the hand-back list"); what it leaves -- a
// long list, a tree too deep for its exact walk -- is looked after by list-mode k_trace, which returns at once when nothing
// is left.  (A launch that emptied lists of any length through a claim cursor, and no follow-up, was tried: 1,500 entries
// took 23 ms.)  The short-list kernel k_trace_slow runs only when the in-kernel drain is off.
void trace_closest(mipt_ctx* c, const float4* rays, const int* index, QueueView q, unsigned* cursor, float4* hits) {
    if (!cursor) {
        cursor = (unsigned*)(c->dTotals + 6);
        (void)hipMemsetAsync(cursor, 0, 4, c->stream);
    }
    Timer t(c, 0);
    count_launch(c, 0);
    unsigned long long* cnt = c->counting ? c->dTotals + 2 : nullptr;
    if (c->useBitTrail) {
        launch_trace_bittrail_closest(c->stream, c->ds, rays, index, q, hits, cnt);
    } else if (SlowView v{}; c->ds.nWide > 0 && c->useWide && ensure_slow(c, c->slow, c->slowParity, q.space, c->stream, v) == MIPT_OK) {
        launch_trace_wide_closest(c->stream, c->ds, rays, index, q, cursor, hits, cnt, v.head, v.list, v.next);
        if (!c->ds.drain) launch_trace_slow_closest(c->stream, c->ds, rays, index, hits, cnt, v.list, v.head);  // short list: a wavefront per ray
        launch_trace_closest(c->stream, c->ds, rays, index, QueueView{nullptr, 0u, q.space}, v.head + 1, hits, cnt, v.list, v.head);  // long list, deep tree: a lane per ray
    } else {
        launch_trace_closest(c->stream, c->ds, rays, index, q, cursor, hits, cnt);
    }
}
// st: the context's stream, or its side stream (path mode's overlap); a DevBuf that grows frees its old memory, so
// growing is only safe while nothing is in flight on either stream -- ensure_path_buffers sizes slowOcc up front
void trace_occluded(mipt_ctx* c, const float4* rays, const int* target, QueueView q, unsigned* cursor, unsigned char* occ,
                    hipStream_t st = nullptr) {
    if (!st) st = c->stream;
    if (!cursor) {
        cursor = (unsigned*)(c->dTotals + 7);
        (void)hipMemsetAsync(cursor, 0, 4, st);
    }
    Timer t(c, 1, st);
    count_launch(c, 1);
    unsigned long long* cnt = c->counting ? c->dTotals + 4 : nullptr;
    if (c->useBitTrail) {
        launch_trace_bittrail_occluded(st, c->ds, rays, target, q, occ, cnt);
    } else if (SlowView v{}; c->ds.nWide > 0 && c->useWide && ensure_slow(c, c->slowOcc, c->slowOccParity, q.space, st, v) == MIPT_OK) {
        launch_trace_wide_occluded(st, c->ds, rays, target, q, cursor, occ, cnt, v.head, v.list, v.next);
        if (!c->ds.drain) launch_trace_slow_occluded(st, c->ds, rays, target, occ, cnt, v.list, v.head);
        launch_trace_occluded(st, c->ds, rays, target, QueueView{nullptr, 0u, q.space}, v.head + 1, occ, cnt, v.list, v.head);
    } else {
        launch_trace_occluded(st, c->ds, rays, target, q, cursor, occ, cnt);
    }
}
// path mode, one launch for both traversals of a bounce (kernels_trace_wide.hip: k_trace_wide_both): the extension rays
// (queue qC, hits) and the shadow rays (queue qO, occ); the rays either body hands back go through the same follow-up
// launches as after the separate launches.  false: this context / scene cannot (the caller launches them separately)
bool trace_both(mipt_ctx* c, const float4* raysC, const int* indexC, QueueView qC, unsigned* cursorC, float4* hits,
                const float4* raysO, const int* targetO, QueueView qO, unsigned* cursorO, unsigned char* occ) {
#if PT_WIDE_K == 8
    if (c->useBitTrail || !(c->ds.nWide > 0 && c->useWide)) return false;
    SlowView vc{}, vo{};
    if (ensure_slow(c, c->slow, c->slowParity, qC.space, c->stream, vc) != MIPT_OK ||
        ensure_slow(c, c->slowOcc, c->slowOccParity, qO.space, c->stream, vo) != MIPT_OK) return false;
    count_launch(c, 0);
    count_launch(c, 1);
    unsigned long long* cntC = c->counting ? c->dTotals + 2 : nullptr;
    unsigned long long* cntO = c->counting ? c->dTotals + 4 : nullptr;
    launch_trace_wide_both(c->stream, c->ds, raysC, indexC, qC, cursorC, hits, cntC, vc.head, vc.list, raysO, targetO, qO, cursorO, occ,
                           cntO, vo.head, vo.list, c->shadowFirst ? 1 : 0, vc.next, vo.next);
    if (!c->ds.drain) launch_trace_slow_closest(c->stream, c->ds, raysC, indexC, hits, cntC, vc.list, vc.head);
    launch_trace_closest(c->stream, c->ds, raysC, indexC, QueueView{nullptr, 0u, qC.space}, vc.head + 1, hits, cntC, vc.list, vc.head);
    if (!c->ds.drain) launch_trace_slow_occluded(c->stream, c->ds, raysO, targetO, occ, cntO, vo.list, vo.head);
    launch_trace_occluded(c->stream, c->ds, raysO, targetO, QueueView{nullptr, 0u, qO.space}, vo.head + 1, occ, cntO, vo.list, vo.head);
    return true;
#else
    return false;
#endif
}
QueueView dense(size_t n) { return QueueView{nullptr, 0u, (unsigned)n}; }

int make_camera(const mipt_camera* cam, CameraDev& out) {
    if (!cam || cam->width <= 0 || cam->height <= 0 || cam->width > 32768 || cam->height > 32768)
        return fail(MIPT_ERR_INVALID, "camera: bad resolution");
    out.width = cam->width;
    out.height = cam->height;
    out.eye = pt::vec3{cam->eye[0], cam->eye[1], cam->eye[2]};
    out.up = pt::normalize(pt::vec3{cam->up[0], cam->up[1], cam->up[2]});                                // C:83
    out.viewDir = pt::normalize(pt::vec3{cam->lookat[0], cam->lookat[1], cam->lookat[2]} - out.eye);      // C:2609
    out.aspect = (float)(cam->width / (float)cam->height);                                               // C:65
    out.filmSize = cam->film_size;                                                                       // C:69
    const float DegToRad = kPI / 180.0f;                                                                 // C:51
    out.distanceToFilm = cam->film_size / (2.0f * tanf(cam->fov_deg * DegToRad * 0.5f));                 // C:70
    return MIPT_OK;
}

// ---- the wide hierarchy on disk (mipt_scene_create_cached) ----
// build_wide is a function of the tree, the leaf order, the triangle positions (leafHead), the lights (slot order) and the
// build's switches; the file holds its output under a 64-bit FNV-1a key over exactly those, and is used only when the key,
// the layout constants and its own checksum all agree.  Written to a temporary name and renamed, so that ranks racing
// for one path read either nothing or a whole file.
constexpr unsigned kWideCacheVersion = 4;  // 4: empty slots of an eight-slot node hold the box no ray hits (lo = +inf, hi = -inf)
struct Fnv {
    unsigned long long h = 1469598103934665603ull;
    void add(const void* p, size_t n) {
        const unsigned char* b = (const unsigned char*)p;
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    }
    template <class T> void pod(const T& v) { add(&v, sizeof v); }
};
unsigned long long wide_key(const mipt_scene_desc* d, int mode, int top) {
    Fnv f;
    f.pod(kWideCacheVersion); f.pod(ptd::kWideK); f.pod(ptd::kWideNodeStrideF4); f.pod(ptd::kLeafHeadStrideF4); f.pod(ptd::kWideMaxDepth);
    f.pod(mode); f.pod(top);
    const char* so = getenv("MIPT_WIDE_SLOT_ORDER");
    if (so) f.add(so, strlen(so));
    f.pod(d->n_nodes); f.pod(d->n_tri_index); f.pod(d->n_tris); f.pod(d->n_lights);
    f.add(d->node_min, sizeof(float) * 3 * (size_t)d->n_nodes); f.add(d->node_max, sizeof(float) * 3 * (size_t)d->n_nodes);
    f.add(d->id_hit, sizeof(int) * (size_t)d->n_nodes); f.add(d->id_miss, sizeof(int) * (size_t)d->n_nodes);
    f.add(d->leaf_first, sizeof(int) * (size_t)d->n_nodes); f.add(d->leaf_count, sizeof(int) * (size_t)d->n_nodes);
    f.add(d->tri_index, sizeof(int) * (size_t)d->n_tri_index);
    f.add(d->pos, sizeof(float) * 9 * (size_t)d->n_tris);
    for (int i = 0; i < d->n_lights; i++) f.add(d->lights[i].position, sizeof d->lights[i].position);
    return f.h;
}
struct WideCacheHeader {
    char magic[8];
    unsigned version, wideOk;
    unsigned long long key;
    int nWide;
    float wideInvScale;
    unsigned long long nWnodes, nLeafBox, nLeafHead, payloadHash;
};
bool load_wide_cache(const char* path, unsigned long long key, mipt_scene* s) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return false;
    WideCacheHeader h{};
    bool ok = fread(&h, sizeof h, 1, fp) == 1 && memcmp(h.magic, "MIPTWIDE", 8) == 0 && h.version == kWideCacheVersion && h.key == key &&
              h.nWnodes < (1ull << 28) && h.nLeafBox < (1ull << 28) && h.nLeafHead < (1ull << 28);
    std::vector<float4> wn, lb, lh;
    if (ok) {
        wn.resize((size_t)h.nWnodes); lb.resize((size_t)h.nLeafBox); lh.resize((size_t)h.nLeafHead);
        ok = (wn.empty() || fread(wn.data(), sizeof(float4), wn.size(), fp) == wn.size()) &&
             (lb.empty() || fread(lb.data(), sizeof(float4), lb.size(), fp) == lb.size()) &&
             (lh.empty() || fread(lh.data(), sizeof(float4), lh.size(), fp) == lh.size()) && fgetc(fp) == EOF;
    }
    fclose(fp);
    if (!ok) return false;
    Fnv f;
    f.add(wn.data(), wn.size() * sizeof(float4)); f.add(lb.data(), lb.size() * sizeof(float4)); f.add(lh.data(), lh.size() * sizeof(float4));
    if (f.h != h.payloadHash) return false;
    if (h.wideOk && (h.nWide <= 0 || wn.size() != (size_t)ptd::kWideNodeStrideF4 * (size_t)h.nWide)) return false;
    s->wnodes.swap(wn); s->leafBox.swap(lb); s->leafHead.swap(lh);
    s->nWide = h.nWide; s->wideInvScale = h.wideInvScale; s->wideOk = h.wideOk != 0;
    return true;
}
void save_wide_cache(const char* path, unsigned long long key, const mipt_scene* s) {
    WideCacheHeader h{};
    memcpy(h.magic, "MIPTWIDE", 8);
    h.version = kWideCacheVersion; h.wideOk = s->wideOk ? 1u : 0u; h.key = key; h.nWide = s->nWide; h.wideInvScale = s->wideInvScale;
    h.nWnodes = s->wnodes.size(); h.nLeafBox = s->leafBox.size(); h.nLeafHead = s->leafHead.size();
    Fnv f;
    f.add(s->wnodes.data(), s->wnodes.size() * sizeof(float4)); f.add(s->leafBox.data(), s->leafBox.size() * sizeof(float4));
    f.add(s->leafHead.data(), s->leafHead.size() * sizeof(float4));
    h.payloadHash = f.h;
    const std::string tmp = std::string(path) + ".tmp" + std::to_string((unsigned long long)(uintptr_t)s);
    FILE* fp = fopen(tmp.c_str(), "wb");
    if (!fp) return;  // a cache that cannot be written is not an error of scene creation
    const bool ok = fwrite(&h, sizeof h, 1, fp) == 1 &&
                    (s->wnodes.empty() || fwrite(s->wnodes.data(), sizeof(float4), s->wnodes.size(), fp) == s->wnodes.size()) &&
                    (s->leafBox.empty() || fwrite(s->leafBox.data(), sizeof(float4), s->leafBox.size(), fp) == s->leafBox.size()) &&
                    (s->leafHead.empty() || fwrite(s->leafHead.data(), sizeof(float4), s->leafHead.size(), fp) == s->leafHead.size());
    if (fclose(fp) != 0 || !ok || rename(tmp.c_str(), path) != 0) (void)remove(tmp.c_str());
}

int scene_create_impl(const mipt_scene_desc* d, const char* wide_cache_path, mipt_scene** out);

}  // namespace

extern "C" {

int mipt_scene_create(const mipt_scene_desc* d, mipt_scene** out) { return scene_create_impl(d, nullptr, out); }
int mipt_scene_create_cached(const mipt_scene_desc* d, const char* wide_cache_path, mipt_scene** out) {
    return scene_create_impl(d, (wide_cache_path && *wide_cache_path) ? wide_cache_path : nullptr, out);
}

// Scene::addObject (cs488.h:2020) + preCalc's one BVH per object (cs488.h:2030-2034).  The objects are laid end to end in
// ONE description -- nodes, leaf ranges, triangles, materials and textures renumbered by their offsets -- and every link
// by which the walk leaves object k's tree (0 there) is pointed at the root of object k + 1: the literal walk then visits
// the objects in the order Scene::intersect's loop does (cs488.h:2075), its strict '<' in visit order is that loop's strict
// '<' between objects, and every other traversal sees one scene whose leafTris order is (object, the object's own order),
// which is exactly the reference's tie-break.  Environment and lights are the first object's.
int mipt_scene_create_multi(const mipt_scene_desc* objs, int32_t n_objects, const char* wide_cache_path, mipt_scene** out) {
    if (!objs || !out || n_objects <= 0) return fail(MIPT_ERR_INVALID, "mipt_scene_create_multi: null argument or no objects");
    const char* cache = (wide_cache_path && *wide_cache_path) ? wide_cache_path : nullptr;
    return guarded("mipt_scene_create_multi", [&]() -> int {
        std::vector<float> nodeMin, nodeMax, pos, nrm, uv;
        std::vector<int32_t> idHit, idMiss, leafFirst, leafCount, triIndex, matId;
        std::vector<mipt_material> mats;
        std::vector<mipt_texture> texs;
        std::vector<int> triBase, matBase;
        long long nNodes = 0, nTriIndex = 0, nTris = 0;
        for (int k = 0; k < n_objects; k++) {
            const mipt_scene_desc& d = objs[k];
            if (d.n_nodes <= 0 || d.n_tris <= 0 || d.n_tri_index <= 0 || !d.node_min || !d.node_max || !d.id_hit || !d.id_miss ||
                !d.leaf_first || !d.leaf_count || !d.tri_index || !d.pos || !d.nrm || !d.uv || !d.mat_id || d.n_materials <= 0 ||
                !d.materials || d.n_textures < 0 || (d.n_textures && !d.textures))
                return fail(MIPT_ERR_INVALID, "mipt_scene_create_multi: an object lacks geometry or materials");
            nNodes += d.n_nodes; nTriIndex += d.n_tri_index; nTris += d.n_tris;
        }
        if (nNodes > 0x7fffffffll || nTriIndex > 0x7fffffffll || nTris > 0x7fffffffll)
            return fail(MIPT_ERR_LIMIT, "mipt_scene_create_multi: the objects together exceed 2^31 nodes or triangles");
        int nodeOfs = 0, triIdxOfs = 0, triOfs = 0;
        for (int k = 0; k < n_objects; k++) {
            const mipt_scene_desc& d = objs[k];
            const int nextRoot = k + 1 < n_objects ? nodeOfs + d.n_nodes : 0;  // where the walk goes when it leaves this tree
            const int mOfs = (int)mats.size(), tOfs = (int)texs.size();
            triBase.push_back(triOfs); matBase.push_back(mOfs);
            nodeMin.insert(nodeMin.end(), d.node_min, d.node_min + (size_t)3 * d.n_nodes);
            nodeMax.insert(nodeMax.end(), d.node_max, d.node_max + (size_t)3 * d.n_nodes);
            for (int i = 0; i < d.n_nodes; i++) {
                const int h = d.id_hit[i], m = d.id_miss[i];
                // links are checked again on the merged arrays (forward or 0); here only: stay inside the own tree
                if (h < 0 || h >= d.n_nodes || m < 0 || m >= d.n_nodes) return fail(MIPT_ERR_INVALID, "mipt_scene_create_multi: a link leaves its object's tree");
                idHit.push_back(h == 0 ? nextRoot : h + nodeOfs);
                idMiss.push_back(m == 0 ? nextRoot : m + nodeOfs);
                leafFirst.push_back(d.leaf_count[i] > 0 ? d.leaf_first[i] + triIdxOfs : d.leaf_first[i]);
                leafCount.push_back(d.leaf_count[i]);
            }
            for (int j = 0; j < d.n_tri_index; j++) {
                if (d.tri_index[j] < 0 || d.tri_index[j] >= d.n_tris) return fail(MIPT_ERR_INVALID, "bad tri_index entry");
                triIndex.push_back(d.tri_index[j] + triOfs);
            }
            pos.insert(pos.end(), d.pos, d.pos + (size_t)9 * d.n_tris);
            nrm.insert(nrm.end(), d.nrm, d.nrm + (size_t)9 * d.n_tris);
            uv.insert(uv.end(), d.uv, d.uv + (size_t)6 * d.n_tris);
            for (int t = 0; t < d.n_tris; t++) {
                if (d.mat_id[t] < 0 || d.mat_id[t] >= d.n_materials) return fail(MIPT_ERR_INVALID, "mat_id out of range");
                matId.push_back(d.mat_id[t] + mOfs);
            }
            for (int i = 0; i < d.n_materials; i++) {
                mipt_material m = d.materials[i];
                if (m.tex_id >= d.n_textures) return fail(MIPT_ERR_INVALID, "material tex_id out of range");
                if (m.tex_id >= 0) m.tex_id += tOfs;
                mats.push_back(m);
            }
            texs.insert(texs.end(), d.textures, d.textures + d.n_textures);
            nodeOfs += d.n_nodes; triIdxOfs += d.n_tri_index; triOfs += d.n_tris;
        }
        mipt_scene_desc m = objs[0];  // environment and lights
        m.n_nodes = nodeOfs; m.node_min = nodeMin.data(); m.node_max = nodeMax.data();
        m.id_hit = idHit.data(); m.id_miss = idMiss.data(); m.leaf_first = leafFirst.data(); m.leaf_count = leafCount.data();
        m.n_tri_index = triIdxOfs; m.tri_index = triIndex.data();
        m.n_tris = triOfs; m.pos = pos.data(); m.nrm = nrm.data(); m.uv = uv.data(); m.mat_id = matId.data();
        m.n_materials = (int32_t)mats.size(); m.materials = mats.data();
        m.n_textures = (int32_t)texs.size(); m.textures = texs.empty() ? nullptr : texs.data();
        mipt_scene* sc = nullptr;
        const int rc = scene_create_impl(&m, cache, &sc);
        if (rc != MIPT_OK) return rc;
        sc->objTriBase = triBase; sc->objMatBase = matBase;
        *out = sc;
        return MIPT_OK;
    });
}
int32_t mipt_scene_object_count(const mipt_scene* s) { return s ? (int32_t)s->roots.size() : 0; }
int32_t mipt_scene_object_tri_base(const mipt_scene* s, int32_t object) {
    if (!s || object < 0) return -1;
    if (s->objTriBase.empty()) return object == 0 ? 0 : -1;
    return object < (int32_t)s->objTriBase.size() ? s->objTriBase[(size_t)object] : -1;
}
int32_t mipt_scene_object_material_base(const mipt_scene* s, int32_t object) {
    if (!s || object < 0) return -1;
    if (s->objMatBase.empty()) return object == 0 ? 0 : -1;
    return object < (int32_t)s->objMatBase.size() ? s->objMatBase[(size_t)object] : -1;
}

}  // extern "C"

namespace {

int scene_create_impl(const mipt_scene_desc* d, const char* wide_cache_path, mipt_scene** out) {
    if (!d || !out) return fail(MIPT_ERR_INVALID, "mipt_scene_create: null argument");
    if (d->n_nodes <= 0 || d->n_tris <= 0 || d->n_tri_index <= 0 || !d->node_min || !d->node_max || !d->id_hit ||
        !d->id_miss || !d->leaf_first || !d->leaf_count || !d->tri_index || !d->pos || !d->nrm || !d->uv || !d->mat_id)
        return fail(MIPT_ERR_INVALID, "mipt_scene_create: missing geometry arrays");
    if (d->n_materials <= 0 || !d->materials) return fail(MIPT_ERR_INVALID, "mipt_scene_create: no materials");
    if (d->n_lights < 0 || d->n_lights > kMaxLights || (d->n_lights && !d->lights))
        return fail(MIPT_ERR_LIMIT, "mipt_scene_create: at most 16 point lights");
    if (d->n_textures < 0 || (d->n_textures && !d->textures)) return fail(MIPT_ERR_INVALID, "bad textures");
    if ((d->env_width > 0) != (d->env_height > 0) || (d->env_width > 0 && !d->env_rgb))
        return fail(MIPT_ERR_INVALID, "bad environment image");
    return guarded("mipt_scene_create", [&]() -> int {
        std::unique_ptr<mipt_scene> owner(new mipt_scene());
        mipt_scene* s = owner.get();
        const int n = d->n_nodes;
        // Link validation.  Preorder threading means every link points forward
        // (or to 0 = done); enforcing that here guarantees that every walk on
        // the GPU terminates, whatever the caller passed in.
        s->nodes.resize((size_t)2 * n);
        // Node numbering on the device: the kTopNodes nodes most likely to be visited come first, so that the
        // traversal kernel can keep them in LDS (kernels_trace.hip).  Chosen greedily from the root by box surface
        // area (the SAH's hit probability), children entering when their parent is taken, which keeps the set a
        // connected top of the tree; then numbered in the original (preorder) order.  Only the numbering changes:
        // every walk follows the same links, so visit order and results are untouched.
        std::vector<int> perm((size_t)n);
        {
            auto area = [&](int i) -> float {
                const float ex = d->node_max[3 * i] - d->node_min[3 * i], ey = d->node_max[3 * i + 1] - d->node_min[3 * i + 1],
                            ez = d->node_max[3 * i + 2] - d->node_min[3 * i + 2];
                const float a = 2.0f * (ex * ey + ey * ez + ez * ex);
                return a == a ? a : 0.0f;
            };
            std::vector<char> top((size_t)n, 0);
            std::priority_queue<std::pair<float, int>> pq;
            {
                std::vector<int> seeds;
                if (!forest_roots(d, seeds)) seeds.assign(1, 0);
                for (const int r : seeds) pq.push({area(r), r});  // several objects: every tree's root is met by every ray
            }
            int nTop = 0;
            while (!pq.empty() && nTop < ptd::kTopNodes) {
                const int i = pq.top().second;
                pq.pop();
                if (top[i]) continue;
                top[i] = 1;
                nTop++;
                if (d->leaf_count[i] > 0) continue;
                const int l = d->id_hit[i];  // left child; the right one is where the left subtree's walk exits
                if (l > i && l < n) {
                    if (!top[l]) pq.push({area(l), l});
                    const int r = d->id_miss[l];
                    if (r > l && r < n && !top[r]) pq.push({area(r), r});
                }
            }
            s->nTopNodes = nTop;
            int next = 0;
            for (int i = 0; i < n; i++) if (top[i]) perm[i] = next++;
            for (int i = 0; i < n; i++) if (!top[i]) perm[i] = next++;
        }
        s->leafTris.resize((size_t)3 * d->n_tri_index);
        std::vector<char> covered((size_t)d->n_tri_index, 0);
        for (int i = 0; i < n; i++) {
            const int miss = d->id_miss[i], hit = d->id_hit[i];
            const int cnt = d->leaf_count[i], first = d->leaf_first[i];
            if (!(miss == 0 || (miss > i && miss < n))) { return fail(MIPT_ERR_INVALID, "id_miss must point forward or be 0"); }
            int link;
            if (cnt > 0) {
                if (hit != miss) { return fail(MIPT_ERR_INVALID, "a leaf's id_hit must equal its id_miss"); }
                if (first < 0 || (long long)first + cnt > d->n_tri_index) { return fail(MIPT_ERR_INVALID, "leaf range out of tri_index"); }
                link = ~first;
                for (int k = 0; k < cnt; k++) {
                    const int j = first + k;
                    const int t = d->tri_index[j];
                    if (t < 0 || t >= d->n_tris || covered[j]) { return fail(MIPT_ERR_INVALID, "bad tri_index entry"); }
                    covered[j] = 1;
                    const float* p = d->pos + (size_t)9 * t;
                    const pt::vec3 p0{p[0], p[1], p[2]}, p1{p[3], p[4], p[5]}, p2{p[6], p[7], p[8]};
                    const pt::vec3 a = p0 - p1, b = p0 - p2;  // C:782-783
                    s->leafTris[(size_t)3 * j] = make_float4(p0.x, p0.y, p0.z, __builtin_bit_cast(float, t));
                    s->leafTris[(size_t)3 * j + 1] = make_float4(a.x, a.y, a.z, __builtin_bit_cast(float, (int)(k == cnt - 1)));
                    s->leafTris[(size_t)3 * j + 2] = make_float4(b.x, b.y, b.z, 0.0f);
                }
            } else {
                if (!(hit > i && hit < n)) { return fail(MIPT_ERR_INVALID, "an interior node's id_hit must point forward"); }
                link = perm[hit];
            }
            for (int a = 0; a < 3; a++) {
                if (!(d->node_min[3 * i + a] <= d->node_max[3 * i + a])) s->planesTame = false;  // slab_hit_fast needs lo <= hi
                for (const float v : {d->node_min[3 * i + a], d->node_max[3 * i + a]}) {
                    const float m = std::fabs(v);
                    if (!(v == 0.0f || (m >= std::ldexp(1.0f, ptd::kTameLo) && m <= std::ldexp(1.0f, ptd::kTameHi)))) s->planesTame = false;  // also catches NaN / inf
                }
            }
            const size_t slot = (size_t)perm[i];
            s->nodes[2 * slot] = make_float4(d->node_min[3 * i], d->node_max[3 * i], d->node_min[3 * i + 1], d->node_max[3 * i + 1]);
            s->nodes[2 * slot + 1] = make_float4(d->node_min[3 * i + 2], d->node_max[3 * i + 2],
                                                 __builtin_bit_cast(float, perm[miss]), __builtin_bit_cast(float, link));
        }
        for (char cv : covered)
            if (!cv) { return fail(MIPT_ERR_INVALID, "tri_index entry not owned by any leaf"); }
        // slots by largest box first; a tree that this makes deeper than the kernel's stacks falls back to plain grandchildren
        const char* collapse = getenv("MIPT_WIDE_COLLAPSE");  // "pairs": grandchildren only (for comparisons)
        const char* topEnv = getenv("MIPT_WIDE_TOP");  // reference | rebuilt | optimised (default): the hierarchy above the leaves
        int mode = 2, top = 2;  // MIPT_WIDE_COLLAPSE = pairs | greedy | optimal (default)
        if (collapse && !strcmp(collapse, "pairs")) mode = 0;
        else if (collapse && !strcmp(collapse, "greedy")) mode = 1;
        if (topEnv && !strcmp(topEnv, "reference")) top = 0;
        else if (topEnv && !strcmp(topEnv, "rebuilt")) top = 1;
        // the best tree first; the only way one fails is by being deeper than the kernel's stacks, then the next best is
        // tried, down to the plain grandchildren of the reference's tree
        s->wideOk = false;
        const unsigned long long key = wide_cache_path ? wide_key(d, mode, top) : 0ull;
        if (!(wide_cache_path && load_wide_cache(wide_cache_path, key, s))) {
            for (int m = mode; m >= 0 && !s->wideOk; m--)
                for (int t = (m > 0 ? top : 0); t >= 0 && !s->wideOk; t--) s->wideOk = build_wide(d, s, m, t);
            if (!s->wideOk) { s->wnodes.clear(); s->leafBox.clear(); s->leafHead.clear(); s->nWide = 0; }
            if (wide_cache_path) save_wide_cache(wide_cache_path, key, s);
        }
        if (!forest_roots(d, s->roots)) s->roots.assign(1, 0);
        if (is_binary_tree(d, s->roots)) {  // right child = where the left subtree's walk exits; independent of the wide tree and of the planes
            s->nodeRight.assign((size_t)n, 0);
            s->btLinks.assign((size_t)n, make_int4(0, 0, 0, 0));  // root: idParent = idSibling = 0, C:1716-1717
            std::vector<int> depthOf((size_t)n, 0);
            int maxDepth = 0;
            for (int i = 0; i < n; i++) {  // preorder: a parent comes before its children
                if (d->leaf_count[i] != 0) continue;
                const int l = d->id_hit[i], r = d->id_miss[l];
                s->nodeRight[(size_t)perm[i]] = perm[r];
                s->btLinks[(size_t)perm[i]].x = perm[r];
                s->btLinks[(size_t)perm[l]].y = perm[i]; s->btLinks[(size_t)perm[l]].z = perm[r];  // C:1720-1723
                s->btLinks[(size_t)perm[r]].y = perm[i]; s->btLinks[(size_t)perm[r]].z = perm[l];
                depthOf[l] = depthOf[r] = depthOf[i] + 1;
                if (depthOf[l] > maxDepth) maxDepth = depthOf[l];
            }
            s->btOk = maxDepth <= 31;  // the reference's 32-bit trail holds one bit per level
            s->slowOk = maxDepth <= 96 && s->roots.size() <= 64;  // k_trace_slow: depth < kSlowReserve - 128 (since round 2 the wide tree no longer bounds the given tree's depth); its walk starts with every root on the stack
        }
        for (int& r : s->roots) r = perm[(size_t)r];  // device numbering
        // textures -> one byte pool
        std::vector<int> texOfs((size_t)d->n_textures, 0);
        for (int i = 0; i < d->n_textures; i++) {
            const mipt_texture& t = d->textures[i];
            if (!t.rgb || t.width <= 0 || t.height <= 0) { return fail(MIPT_ERR_INVALID, "empty texture"); }
            texOfs[i] = (int)s->texels.size();
            s->texels.insert(s->texels.end(), t.rgb, t.rgb + (size_t)3 * t.width * t.height);
        }
        s->materials.resize((size_t)4 * d->n_materials);
        for (int i = 0; i < d->n_materials; i++) {
            const mipt_material& m = d->materials[i];
            if (m.type == MIPT_MAT_METAL || m.type == MIPT_MAT_GLASS) s->hasSpecular = true;
            if (m.tex_id >= d->n_textures) { return fail(MIPT_ERR_INVALID, "material tex_id out of range"); }
            const int ofs = m.tex_id >= 0 ? texOfs[m.tex_id] : -1;
            const int tw = m.tex_id >= 0 ? d->textures[m.tex_id].width : 0;
            const int th = m.tex_id >= 0 ? d->textures[m.tex_id].height : 0;
            s->materials[(size_t)4 * i] = make_float4(__builtin_bit_cast(float, (int)m.type), m.eta, m.Ns, __builtin_bit_cast(float, ofs));
            s->materials[(size_t)4 * i + 1] = make_float4(m.Kd[0], m.Kd[1], m.Kd[2], __builtin_bit_cast(float, tw));
            s->materials[(size_t)4 * i + 2] = make_float4(m.Ks[0], m.Ks[1], m.Ks[2], __builtin_bit_cast(float, th));
            s->materials[(size_t)4 * i + 3] = make_float4(m.Ka[0], m.Ka[1], m.Ka[2], 0.0f);
        }
        s->triAttr.assign((size_t)ptd::kTriAttrF4 * d->n_tris, make_float4(0, 0, 0, 0));
        for (int t = 0; t < d->n_tris; t++) {
            if (d->mat_id[t] < 0 || d->mat_id[t] >= d->n_materials) { return fail(MIPT_ERR_INVALID, "mat_id out of range"); }
            float f[28];
            memcpy(f, d->pos + (size_t)9 * t, 36);
            memcpy(f + 9, d->nrm + (size_t)9 * t, 36);
            const int mid = d->mat_id[t];
            memcpy(f + 18, &mid, 4);  // the first 80 B hold what every shaded hit reads, the texture coordinates follow
            f[19] = 0.0f;
            memcpy(f + 20, d->uv + (size_t)6 * t, 24);
            f[26] = f[27] = 0.0f;
            memcpy(&s->triAttr[(size_t)ptd::kTriAttrF4 * t], f, sizeof f);
        }
        if (d->env_width > 0) s->env.assign(d->env_rgb, d->env_rgb + (size_t)3 * d->env_width * d->env_height);
        s->lights.resize((size_t)2 * d->n_lights);
        for (int i = 0; i < d->n_lights; i++) {
            const mipt_light& l = d->lights[i];
            s->lights[(size_t)2 * i] = make_float4(l.position[0], l.position[1], l.position[2], 0.0f);
            s->lights[(size_t)2 * i + 1] = make_float4(l.wattage[0], l.wattage[1], l.wattage[2], 0.0f);
        }
        s->nNodes = n; s->nLeafTris = d->n_tri_index; s->nTris = d->n_tris; s->nMaterials = d->n_materials;
        s->envW = d->env_width; s->envH = d->env_height; s->nLights = d->n_lights;
        *out = owner.release();
        return MIPT_OK;
    });
}

}  // namespace

extern "C" {

void mipt_scene_destroy(mipt_scene* s) { delete s; }
int mipt_scene_exact_reciprocal(const mipt_scene* s) { return (s && s->planesTame) ? 1 : 0; }
int mipt_scene_wide_nodes(const mipt_scene* s) { return (s && s->wideOk) ? s->nWide : 0; }
uint64_t mipt_scene_wide_hash(const mipt_scene* s) {
    if (!s) return 0;
    Fnv f;
    f.pod(s->nWide); f.pod(s->wideInvScale);
    f.add(s->wnodes.data(), s->wnodes.size() * sizeof(float4)); f.add(s->leafBox.data(), s->leafBox.size() * sizeof(float4));
    f.add(s->leafHead.data(), s->leafHead.size() * sizeof(float4));
    return f.h;
}

// What the order-free traversal rests on, checked on the finished arrays (host only, no GPU): the wide tree reaches
// every leaf of the scene exactly once, every slot's STORED box (fp16, rounded outwards) contains the stored boxes of
// the node below it resp. the exact box of its leaf, and every leaf record points at a distinct first triangle.
int mipt_scene_selfcheck_wide(const mipt_scene* s, int* leaves_out, int* depth_out) {
    return guarded("mipt_scene_selfcheck_wide", [&]() -> int {
    if (!s) return fail(MIPT_ERR_INVALID, "mipt_scene_selfcheck_wide: null argument");
    if (leaves_out) *leaves_out = 0;
    if (depth_out) *depth_out = 0;
    if (!s->wideOk) return MIPT_OK;  // nothing to check: every ray takes the literal walk
#if !PT_WIDE_HALF
    return fail(MIPT_ERR_INVALID, "mipt_scene_selfcheck_wide: built without the fp16 node layout");
#else
    const int nw = s->nWide, nLeaves = (int)(s->leafBox.size() / 2);
    std::vector<char> nodeSeen((size_t)nw, 0), leafSeen((size_t)nLeaves, 0);
    struct Box { float lo[3], hi[3]; };
    constexpr int K = ptd::kWideK;
    auto slotBox = [&](int w, int k) {
        Box b;
        for (int a = 0; a < 3; a++) {
            unsigned short h[2 * K];
            memcpy(h, (const char*)&s->wnodes[(size_t)ptd::kWideNodeStrideF4 * w] + 4 * K * a, 4 * K);
            b.lo[a] = half_value(h[k]) * s->wideInvScale; b.hi[a] = half_value(h[K + k]) * s->wideInvScale;
        }
        return b;
    };
    auto masksOf = [&](int w, int& childBase, int& leafBase, int& imask, int& lmask) {
        const float4 meta = s->wnodes[(size_t)ptd::kWideNodeStrideF4 * w + (K == 8 ? 6 : 3)];
        childBase = __builtin_bit_cast(int, meta.x); leafBase = __builtin_bit_cast(int, meta.y);
        const int m = __builtin_bit_cast(int, meta.z);
        if (K == 8) { const int ni = m & 255, nl = m >> 8; imask = (1 << ni) - 1; lmask = ((1 << nl) - 1) << ni; }
        else { imask = m & 15; lmask = (m >> 4) & 15; }
    };
    auto inside = [](const Box& in, const Box& out) {
        for (int a = 0; a < 3; a++)
            if (!(out.lo[a] <= in.lo[a] && in.hi[a] <= out.hi[a])) return false;
        return true;
    };
    struct Item { int w, depth; };
    std::vector<Item> stack{{0, 0}};
    int maxDepth = 0, reached = 0;
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        if (it.w < 0 || it.w >= nw || nodeSeen[(size_t)it.w]) return fail(MIPT_ERR_INVALID, "wide tree: a node is out of range or reached twice");
        nodeSeen[(size_t)it.w] = 1;
        if (it.depth > maxDepth) maxDepth = it.depth;
        int childBase, leafBase, imask, lmask;
        masksOf(it.w, childBase, leafBase, imask, lmask);
        if ((imask & lmask) != 0 || (imask | lmask) == 0 || ((imask | lmask) >> K) != 0) return fail(MIPT_ERR_INVALID, "wide tree: bad slot masks");
        int ni = 0, nl = 0;
        for (int k = 0; k < K; k++) {
            const Box sb = slotBox(it.w, k);
            if (imask & (1 << k)) {
                const int c = childBase + ni++;
                if (c <= it.w || c >= nw) return fail(MIPT_ERR_INVALID, "wide tree: a child id does not point forward");
                int cb, clb, cim, clm;
                masksOf(c, cb, clb, cim, clm);
                for (int j = 0; j < K; j++)
                    if ((cim | clm) & (1 << j))
                        if (!inside(slotBox(c, j), sb)) return fail(MIPT_ERR_INVALID, "wide tree: a slot's box does not contain a box of the node below it");
                stack.push_back({c, it.depth + 1});
            } else if (lmask & (1 << k)) {
                const int l = leafBase + nl++;
                if (l < 0 || l >= nLeaves || leafSeen[(size_t)l]) return fail(MIPT_ERR_INVALID, "wide tree: a leaf is out of range or reached twice");
                leafSeen[(size_t)l] = 1;
                reached++;
                const float4 b0 = s->leafBox[(size_t)2 * l], b1 = s->leafBox[(size_t)2 * l + 1];
                const Box lb{{b0.x, b0.z, b1.x}, {b0.y, b0.w, b1.y}};
                if (!inside(lb, sb)) return fail(MIPT_ERR_INVALID, "wide tree: a slot's box does not contain its leaf's exact box");
            }
        }
    }
    if (reached != nLeaves) return fail(MIPT_ERR_INVALID, "wide tree: not every leaf is reached");
    for (int w = 0; w < nw; w++)
        if (!nodeSeen[(size_t)w]) return fail(MIPT_ERR_INVALID, "wide tree: an unreachable node");
    std::vector<int> firsts;
    for (int l = 0; l < nLeaves; l++) firsts.push_back(__builtin_bit_cast(int, s->leafBox[(size_t)2 * l + 1].z));
    std::sort(firsts.begin(), firsts.end());
    for (size_t i = 1; i < firsts.size(); i++)
        if (firsts[i] == firsts[i - 1]) return fail(MIPT_ERR_INVALID, "wide tree: two leaves share their first triangle");
    if (leaves_out) *leaves_out = nLeaves;
    if (depth_out) *depth_out = maxDepth;
    return MIPT_OK;
#endif
    });
}

int mipt_context_create(const mipt_scene* s, int device, void* stream, mipt_ctx** out) {
    return guarded("mipt_context_create", [&]() -> int {
    if (!s || !out) return fail(MIPT_ERR_INVALID, "mipt_context_create: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MIPT_ERR_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(MIPT_ERR_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    {
        const int rcs = resolve_scratch(device);  // the no-alloc entry point's buffer on this GPU
        if (rcs != MIPT_OK) return rcs;
    }
    struct CtxDeleter { void operator()(mipt_ctx* p) const { mipt_context_destroy(p); } };
    std::unique_ptr<mipt_ctx, CtxDeleter> owner(new mipt_ctx());  // destroyed on every failure path, exceptions included
    mipt_ctx* c = owner.get();
    c->device = device;
    c->scratch = (unsigned char*)g_scratchPtr[device];
    if (const char* tv = getenv("MIPT_TRAVERSAL")) {  // A/B runs without touching the caller: walk | bittrail
        c->useWide = std::string(tv) != "walk";
        c->useBitTrail = std::string(tv) == "bittrail" && s->btOk;
    }
    if (const char* ov = getenv("MIPT_OVERLAP")) c->overlap = std::string(ov) != "0";
    if (const char* fv = getenv("MIPT_FUSE")) c->fuse = std::string(fv) != "0";
    if (const char* fo = getenv("MIPT_FUSE_ORDER")) c->shadowFirst = std::string(fo) == "shadow";
    if (stream) c->stream = (hipStream_t)stream;
    else {
        if (hipStreamCreate(&c->stream) != hipSuccess) return fail(MIPT_ERR_DEVICE, "hipStreamCreate failed");
        c->ownStream = true;
    }
    // one arena, 256-B aligned sections
    const size_t szNodes = align256(s->nodes.size() * sizeof(float4));
    const size_t szLeaf = align256(s->leafTris.size() * sizeof(float4));
    const size_t szAttr = align256(s->triAttr.size() * sizeof(float4));
    const size_t szMat = align256(s->materials.size() * sizeof(float4));
    const size_t szTex = align256(s->texels.size() + 4);
    const size_t szEnv = align256(s->env.size() * sizeof(float) + 4);
    const size_t szLights = align256(s->lights.size() * sizeof(float4) + 16);
    const size_t szWide = align256(s->wnodes.size() * sizeof(float4) + 16);
    const size_t szLeafBox = align256(s->leafBox.size() * sizeof(float4) + 16);
    const size_t szLeafHead = align256(s->leafHead.size() * sizeof(float4) + 16);
    const size_t szRight = align256(s->nodeRight.size() * sizeof(int) + 16);
    const size_t szBt = align256(s->btLinks.size() * sizeof(int4) + 16);
    const size_t szRoots = align256(s->roots.size() * sizeof(int) + 16);
    const size_t total = szNodes + szLeaf + szAttr + szMat + szTex + szEnv + szLights + szWide + szLeafBox + szLeafHead + szRight + szBt + szRoots;
    std::vector<unsigned char> stage(total, 0);
    size_t o = 0;
    const size_t oNodes = o; memcpy(&stage[o], s->nodes.data(), s->nodes.size() * sizeof(float4)); o += szNodes;
    const size_t oLeaf = o; memcpy(&stage[o], s->leafTris.data(), s->leafTris.size() * sizeof(float4)); o += szLeaf;
    const size_t oAttr = o; memcpy(&stage[o], s->triAttr.data(), s->triAttr.size() * sizeof(float4)); o += szAttr;
    const size_t oMat = o; memcpy(&stage[o], s->materials.data(), s->materials.size() * sizeof(float4)); o += szMat;
    const size_t oTex = o; if (!s->texels.empty()) memcpy(&stage[o], s->texels.data(), s->texels.size()); o += szTex;
    const size_t oEnv = o; if (!s->env.empty()) memcpy(&stage[o], s->env.data(), s->env.size() * sizeof(float)); o += szEnv;
    const size_t oLights = o; if (!s->lights.empty()) memcpy(&stage[o], s->lights.data(), s->lights.size() * sizeof(float4)); o += szLights;
    const size_t oWide = o; if (!s->wnodes.empty()) memcpy(&stage[o], s->wnodes.data(), s->wnodes.size() * sizeof(float4)); o += szWide;
    const size_t oLeafBox = o; if (!s->leafBox.empty()) memcpy(&stage[o], s->leafBox.data(), s->leafBox.size() * sizeof(float4)); o += szLeafBox;
    const size_t oLeafHead = o; if (!s->leafHead.empty()) memcpy(&stage[o], s->leafHead.data(), s->leafHead.size() * sizeof(float4)); o += szLeafHead;
    const size_t oRight = o; if (!s->nodeRight.empty()) memcpy(&stage[o], s->nodeRight.data(), s->nodeRight.size() * sizeof(int)); o += szRight;
    const size_t oBt = o; if (!s->btLinks.empty()) memcpy(&stage[o], s->btLinks.data(), s->btLinks.size() * sizeof(int4)); o += szBt;
    const size_t oRoots = o; if (!s->roots.empty()) memcpy(&stage[o], s->roots.data(), s->roots.size() * sizeof(int));
    if (hipMalloc(&c->arena, total) != hipSuccess) return fail(MIPT_ERR_DEVICE, "hipMalloc(arena) failed");
    c->arenaBytes = total;
    if (hipMemcpy(c->arena, stage.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        return fail(MIPT_ERR_DEVICE, "arena upload failed");
    }
    unsigned char* base = (unsigned char*)c->arena;
    c->ds.nodes = (const float4*)(base + oNodes);
    c->ds.leafTris = (const float4*)(base + oLeaf);
    c->ds.triAttr = (const float4*)(base + oAttr);
    c->ds.materials = (const float4*)(base + oMat);
    c->ds.texels = base + oTex;
    c->ds.env = (const float*)(base + oEnv);
    c->ds.lights = (const float4*)(base + oLights);
    c->ds.nNodes = s->nNodes; c->ds.nLeafTris = s->nLeafTris; c->ds.nTris = s->nTris; c->ds.nMaterials = s->nMaterials;
    c->ds.envW = s->envW; c->ds.envH = s->envH; c->ds.nLights = s->nLights;
    c->ds.planesTame = s->planesTame ? 1 : 0;
    c->ds.drain = PT_WIDE_DRAIN ? 1 : 0;
    if (const char* dv = getenv("MIPT_DRAIN")) c->ds.drain = std::string(dv) != "0" ? 1 : 0;  // A/B runs
    if (const char* hb = getenv("MIPT_HAND_BACK_ALL")) c->ds.handBackAll = std::string(hb) == "1" ? 1 : 0;  // tests: every ray through k_trace_slow / list-mode k_trace
    c->ds.nTopNodes = s->nTopNodes;
    c->ds.wnodes = (const float4*)(base + oWide);
    c->ds.leafBox = (const float4*)(base + oLeafBox);
    c->ds.leafHead = (const float4*)(base + oLeafHead);
    c->ds.nodeRight = (const int*)(base + oRight);
    c->ds.btLinks = (const int4*)(base + oBt);
    c->ds.roots = (const int*)(base + oRoots);
    c->ds.nRoots = (int)s->roots.size();
    c->ds.btOk = s->btOk ? 1 : 0;
    c->ds.slowOk = s->slowOk ? 1 : 0;
    c->hasSpecular = s->hasSpecular;
    // k_trace_wide addresses leafHead and leafTris with 32-bit offsets from leafTris
    c->ds.nWide = (s->wideOk && oLeafHead + szLeafHead - oLeaf < ((size_t)1 << 32)) ? s->nWide : 0;
    c->ds.nWideTop = s->nWide < ptd::kWideTopNodes ? s->nWide : ptd::kWideTopNodes;
    c->ds.wideInvScale = s->wideInvScale;
    if (hipMalloc((void**)&c->dTotals, 8 * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(c->dTotals, 0, 8 * sizeof(unsigned long long)) != hipSuccess) {
        return fail(MIPT_ERR_DEVICE, "hipMalloc(stats) failed");
    }
    *out = owner.release();
    return MIPT_OK;
    });
}

void mipt_context_destroy(mipt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    for (auto e : c->evShaded) (void)hipEventDestroy(e);
    for (auto e : c->evOccluded) (void)hipEventDestroy(e);
    for (auto& t : c->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto e : c->freeEvents) (void)hipEventDestroy(e);
    c->slow.release(); c->slowOcc.release(); c->sRays.release(); c->sHits.release(); c->sOut.release(); c->sAux.release(); c->sAux2.release();
    c->pathMem.release(); c->tileMem.release(); c->frameMem.release();
    if (c->dTotals) (void)hipFree(c->dTotals);
    if (c->arena) (void)hipFree(c->arena);
    if (c->ownStream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

size_t mipt_context_arena_bytes(const mipt_ctx* c) { return c ? c->arenaBytes : 0; }

int mipt_buffer_alloc(mipt_ctx* c, size_t bytes, void** dev_ptr) {
    if (!c || !dev_ptr || bytes == 0) return fail(MIPT_ERR_INVALID, "mipt_buffer_alloc: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(dev_ptr, bytes));
    return MIPT_OK;
}
int mipt_buffer_free(mipt_ctx* c, void* dev_ptr) {
    if (!c) return fail(MIPT_ERR_INVALID, "null context");
    if (!dev_ptr) return MIPT_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(dev_ptr));
    return MIPT_OK;
}
int mipt_buffer_zero(mipt_ctx* c, void* dev_ptr, size_t bytes) {
    if (!c || (bytes && !dev_ptr)) return fail(MIPT_ERR_INVALID, "mipt_buffer_zero: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (bytes) HIP_TRY(hipMemsetAsync(dev_ptr, 0, bytes, c->stream));
    return MIPT_OK;
}
int mipt_pack_rgb8_device(mipt_ctx* c, const float* rgb_dev, size_t n_values, uint8_t* out_dev) {
    if (!c || (n_values && (!rgb_dev || !out_dev))) return fail(MIPT_ERR_INVALID, "mipt_pack_rgb8_device: bad argument");
    if (((uintptr_t)rgb_dev & 15u) || ((uintptr_t)out_dev & 3u))
        return fail(MIPT_ERR_INVALID, "mipt_pack_rgb8_device: rgb_dev must be 16-byte aligned, out_dev 4-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    if (n_values) ptd::launch_pack_rgb8(c->stream, rgb_dev, n_values, out_dev);
    HIP_TRY(hipGetLastError());
    return MIPT_OK;
}
int mipt_buffer_upload(mipt_ctx* c, void* dev_ptr, const void* host_ptr, size_t bytes) {
    if (!c || (bytes && (!dev_ptr || !host_ptr))) return fail(MIPT_ERR_INVALID, "mipt_buffer_upload: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (bytes) HIP_TRY(hipMemcpyAsync(dev_ptr, host_ptr, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MIPT_OK;
}
int mipt_buffer_download(mipt_ctx* c, void* host_ptr, const void* dev_ptr, size_t bytes) {
    if (!c || (bytes && (!dev_ptr || !host_ptr))) return fail(MIPT_ERR_INVALID, "mipt_buffer_download: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (bytes) HIP_TRY(hipMemcpyAsync(host_ptr, dev_ptr, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MIPT_OK;
}

int mipt_context_synchronize(mipt_ctx* c) {
    if (!c) return fail(MIPT_ERR_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->side) HIP_TRY(hipStreamSynchronize(c->side));
    return MIPT_OK;
}
int mipt_context_set_overlap(mipt_ctx* c, int enabled) {
    if (!c) return fail(MIPT_ERR_INVALID, "null context");
    c->overlap = enabled != 0;
    return MIPT_OK;
}
int mipt_context_set_profiling(mipt_ctx* c, int enabled) {
    if (!c) return fail(MIPT_ERR_INVALID, "null context");
    c->profiling = enabled != 0;
    return MIPT_OK;
}
int mipt_context_set_counting(mipt_ctx* c, int enabled) {
    if (!c) return fail(MIPT_ERR_INVALID, "null context");
    c->counting = enabled != 0;
    return MIPT_OK;
}

static int drain_timers(mipt_ctx* c) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->side) HIP_TRY(hipStreamSynchronize(c->side));
    for (auto& t : c->timed) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            if (t.cls == 0) c->acc.ms_closest += ms;
            else if (t.cls == 1) c->acc.ms_occluded += ms;
            else {
                c->acc.ms_other += ms;
                if (t.cls == 3) c->acc.ms_shade += ms;
            }
        }
        c->freeEvents.push_back(t.a);
        c->freeEvents.push_back(t.b);
    }
    c->timed.clear();
    return MIPT_OK;
}

int mipt_context_set_traversal(mipt_ctx* c, int traversal) {
    if (!c) return fail(MIPT_ERR_INVALID, "null context");
    if (traversal == MIPT_TRAVERSAL_DEFAULT) { c->useBitTrail = false; c->useWide = true; return MIPT_OK; }
    if (traversal == MIPT_TRAVERSAL_THREADED) { c->useBitTrail = false; c->useWide = false; return MIPT_OK; }
    if (traversal == MIPT_TRAVERSAL_BITTRAIL) {
        if (!c->ds.btOk)
            return fail(MIPT_ERR_LIMIT, "bit-trail traversal needs a proper binary tree (two children per interior node, one parent each) of depth <= 31, the width of the reference's 32-bit trail");
        c->useBitTrail = true;
        return MIPT_OK;
    }
    return fail(MIPT_ERR_INVALID, "unknown traversal");
}
int mipt_context_get_stats(mipt_ctx* c, mipt_stats* out) {
    return guarded("mipt_context_get_stats", [&]() -> int {
    if (!c || !out) return fail(MIPT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = drain_timers(c);
    if (rc != MIPT_OK) return rc;
    unsigned long long h[8];
    HIP_TRY(hipMemcpy(h, c->dTotals, sizeof h, hipMemcpyDeviceToHost));
    *out = c->acc;
    out->rays_closest += h[0];
    out->rays_occluded += h[1];
    out->node_visits_closest = h[2];
    out->tri_tests_closest = h[3];
    out->node_visits_occluded = h[4];
    out->tri_tests_occluded = h[5];
    return MIPT_OK;
    });
}
int mipt_context_reset_stats(mipt_ctx* c) {
    if (!c) return fail(MIPT_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    int rc = drain_timers(c);
    if (rc != MIPT_OK) return rc;
    memset(&c->acc, 0, sizeof c->acc);
    HIP_TRY(hipMemset(c->dTotals, 0, 8 * sizeof(unsigned long long)));
    return MIPT_OK;
}

int mipt_selftest_fast_division(mipt_ctx* c, uint64_t n_pairs, uint32_t seed, uint64_t* mismatches) {
    if (!c || !mismatches) return fail(MIPT_ERR_INVALID, "mipt_selftest_fast_division: null argument");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long* dBad = c->dTotals + 6;
    HIP_TRY(hipMemsetAsync(dBad, 0, 8, c->stream));
    const unsigned blocks = 4096, perThread = (unsigned)((n_pairs + (uint64_t)blocks * kBlock - 1) / ((uint64_t)blocks * kBlock));
    launch_divcheck(c->stream, seed, blocks, perThread, dBad);
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, dBad, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    *mismatches = h;
    return MIPT_OK;
}

int mipt_selftest_vec_ops(mipt_ctx* c, int op, const float* a, const float* b, const float* cc, const float* d, size_t n, float* out) {
    return guarded("mipt_selftest_vec_ops", [&]() -> int {
    static const int outW[8] = {1, 3, 3, 1, 3, 1, 2, 3};
    static const int nIn[8] = {2, 2, 1, 3, 4, 1, 1, 2};
    if (op < 0 || op > 7) return fail(MIPT_ERR_INVALID, "mipt_selftest_vec_ops: unknown op");
    if (n && (!a || !out || (nIn[op] > 1 && !b) || (nIn[op] > 2 && !cc) || (nIn[op] > 3 && !d)))
        return fail(MIPT_ERR_INVALID, "mipt_selftest_vec_ops: null operand");
    if (n > 0x7fffffffu / 3) return fail(MIPT_ERR_LIMIT, "mipt_selftest_vec_ops: too many elements");
    if (n == 0) return MIPT_OK;
    if (!c) {  // the host build of pt/vec.hpp
        vec_ops_host(op, a, nIn[op] > 1 ? b : nullptr, nIn[op] > 2 ? cc : nullptr, nIn[op] > 3 ? d : nullptr, n, out);
        return MIPT_OK;
    }
    HIP_TRY(hipSetDevice(c->device));
    const size_t inBytes = n * 3 * sizeof(float), outBytes = n * (size_t)outW[op] * sizeof(float);
    int rc;
    if ((rc = c->sAux.ensure(4 * align256(inBytes))) != MIPT_OK) return rc;
    if ((rc = c->sAux2.ensure(outBytes)) != MIPT_OK) return rc;
    const float* src[4] = {a, b, cc, d};
    float* dv[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < nIn[op]; k++) {
        dv[k] = (float*)((unsigned char*)c->sAux.p + (size_t)k * align256(inBytes));
        HIP_TRY(hipMemcpyAsync(dv[k], src[k], inBytes, hipMemcpyHostToDevice, c->stream));
    }
    launch_vec_ops(c->stream, op, dv[0], dv[1], dv[2], dv[3], n, (float*)c->sAux2.p);
    HIP_TRY(hipMemcpyAsync(out, c->sAux2.p, outBytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return MIPT_OK;
    });
}

int mipt_trace_device(mipt_ctx* c, const void* rays_dev, size_t n, void* hits_dev) {
    if (!c || (n && (!rays_dev || !hits_dev))) return fail(MIPT_ERR_INVALID, "mipt_trace_device: null argument");
    if (n > 0x7fffffffu) return fail(MIPT_ERR_LIMIT, "mipt_trace_device: at most 2^31-1 rays per call");
    if (n == 0) return MIPT_OK;
    HIP_TRY(hipSetDevice(c->device));
    trace_closest(c, (const float4*)rays_dev, nullptr, dense(n), nullptr, (float4*)hits_dev);
    c->acc.rays_closest += n;
    HIP_TRY(hipGetLastError());
    return MIPT_OK;
}

// mipt_trace_device for callers that cannot allocate: same kernels, same results, but the launch cursors and the list
// of handed-back rays come from this GPU's copy of `scratch` (c->scratch, resolved with hipGetSymbolAddress when the
// context was made) instead of hipMalloc'd memory.  No hipMalloc / hipFree / event creation here.  Calls on one GPU
// share that GPU's array: two that overlap in time (two contexts or streams on the same device) race.
int mipt_trace_device_noalloc(mipt_ctx* c, const void* rays_dev, size_t n, void* hits_dev) {
    if (!c || (n && (!rays_dev || !hits_dev))) return fail(MIPT_ERR_INVALID, "mipt_trace_device_noalloc: null argument");
    if (n > kNoAllocMaxRays) return fail(MIPT_ERR_LIMIT, "mipt_trace_device_noalloc: at most 1,048,576 rays per call (the size of the library's static scratch)");
    if (n == 0) return MIPT_OK;
    if (!c->scratch) return fail(MIPT_ERR_DEVICE, "mipt_trace_device_noalloc: scratch not resolved on this device");
    HIP_TRY(hipSetDevice(c->device));
    unsigned* head = (unsigned*)c->scratch;                 // [0] list count, [1] cursor of the list-mode launch, [2] blocks done, [3] complement of the least count a draining wavefront saw
    unsigned* cursor = head + 32;                           // cursor of the main launch
    int* list = (int*)(c->scratch + kNoAllocHeaderBytes);   // queue slots of the handed-back rays, at most n
    HIP_TRY(hipMemsetAsync(head, 0, kNoAllocHeaderBytes, c->stream));  // the array keeps the last call's contents
    const float4* rays = (const float4*)rays_dev;
    float4* hits = (float4*)hits_dev;
    const QueueView q = dense(n);
    unsigned long long* cnt = c->counting ? c->dTotals + 2 : nullptr;
    count_launch(c, 0);
    if (c->useBitTrail) {
        launch_trace_bittrail_closest(c->stream, c->ds, rays, nullptr, q, hits, cnt);
    } else if (c->ds.nWide > 0 && c->useWide) {
        launch_trace_wide_closest(c->stream, c->ds, rays, nullptr, q, cursor, hits, cnt, head, list);
        if (!c->ds.drain) launch_trace_slow_closest(c->stream, c->ds, rays, nullptr, hits, cnt, list, head);
        launch_trace_closest(c->stream, c->ds, rays, nullptr, QueueView{nullptr, 0u, q.space}, head + 1, hits, cnt, list, head);
    } else {
        launch_trace_closest(c->stream, c->ds, rays, nullptr, q, cursor, hits, cnt);
    }
    c->acc.rays_closest += n;
    HIP_TRY(hipGetLastError());
    return MIPT_OK;
}

int mipt_trace(mipt_ctx* c, const float* ray_o, const float* ray_d, size_t n, float tmin, float tmax, int32_t* tri_id,
               float* t, float* bary, float* hitrec) {
    return guarded("mipt_trace", [&]() -> int {
    if (!c || (n && (!ray_o || !ray_d))) return fail(MIPT_ERR_INVALID, "mipt_trace: null argument");
    if (n > 0x7fffffffu) return fail(MIPT_ERR_LIMIT, "mipt_trace: at most 2^31-1 rays per call");
    if (n == 0) return MIPT_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<float4> rays((size_t)2 * n);
    for (size_t i = 0; i < n; i++) {
        rays[2 * i] = make_float4(ray_o[3 * i], ray_o[3 * i + 1], ray_o[3 * i + 2], tmin);
        rays[2 * i + 1] = make_float4(ray_d[3 * i], ray_d[3 * i + 1], ray_d[3 * i + 2], tmax);
    }
    int rc;
    if ((rc = c->sRays.ensure(rays.size() * sizeof(float4))) != MIPT_OK) return rc;
    if ((rc = c->sHits.ensure(n * sizeof(float4))) != MIPT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c->sRays.p, rays.data(), rays.size() * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    trace_closest(c, (const float4*)c->sRays.p, nullptr, dense(n), nullptr, (float4*)c->sHits.p);
    c->acc.rays_closest += n;
    std::vector<float4> hits(n);
    HIP_TRY(hipMemcpyAsync(hits.data(), c->sHits.p, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (hitrec) {
        if ((rc = c->sOut.ensure(n * 10 * sizeof(float))) != MIPT_OK) return rc;
        {
            Timer tm(c, 2);
            count_launch(c, 2);
            launch_resolve_hits(c->stream, c->ds, (const float4*)c->sRays.p, (const float4*)c->sHits.p, n, (float*)c->sOut.p);
        }
        HIP_TRY(hipMemcpyAsync(hitrec, c->sOut.p, n * 10 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    for (size_t i = 0; i < n; i++) {
        int id;
        memcpy(&id, &hits[i].x, 4);
        if (tri_id) tri_id[i] = id;
        if (t) t[i] = hits[i].y;
        if (bary) { bary[2 * i] = hits[i].z; bary[2 * i + 1] = hits[i].w; }
    }
    return MIPT_OK;
    });
}

int mipt_occluded(mipt_ctx* c, const float* ray_o, const float* ray_d, const float* tmin, const float* tmax, size_t n,
                  uint8_t* occluded) {
    return guarded("mipt_occluded", [&]() -> int {
    if (!c || (n && (!ray_o || !ray_d || !tmin || !tmax || !occluded))) return fail(MIPT_ERR_INVALID, "mipt_occluded: null argument");
    if (n > 0x7fffffffu) return fail(MIPT_ERR_LIMIT, "mipt_occluded: at most 2^31-1 rays per call");
    if (n == 0) return MIPT_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<float4> rays((size_t)2 * n);
    for (size_t i = 0; i < n; i++) {
        rays[2 * i] = make_float4(ray_o[3 * i], ray_o[3 * i + 1], ray_o[3 * i + 2], tmin[i]);
        rays[2 * i + 1] = make_float4(ray_d[3 * i], ray_d[3 * i + 1], ray_d[3 * i + 2], tmax[i]);
    }
    int rc;
    if ((rc = c->sRays.ensure(rays.size() * sizeof(float4))) != MIPT_OK) return rc;
    if ((rc = c->sAux.ensure(n)) != MIPT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c->sRays.p, rays.data(), rays.size() * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->sAux.p, 0, n, c->stream));
    trace_occluded(c, (const float4*)c->sRays.p, nullptr, dense(n), nullptr, (unsigned char*)c->sAux.p);
    c->acc.rays_occluded += n;
    HIP_TRY(hipMemcpyAsync(occluded, c->sAux.p, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return MIPT_OK;
    });
}

int mipt_eye_rays(mipt_ctx* c, const mipt_camera* cam, size_t first, size_t count, float* ray_o, float* ray_d) {
    return guarded("mipt_eye_rays", [&]() -> int {
    if (!c || !cam || (count && (!ray_o || !ray_d))) return fail(MIPT_ERR_INVALID, "mipt_eye_rays: null argument");
    CameraDev cd;
    int rc = make_camera(cam, cd);
    if (rc != MIPT_OK) return rc;
    if (first + count > (size_t)cam->width * cam->height) return fail(MIPT_ERR_INVALID, "mipt_eye_rays: pixel range outside the image");
    if (count > 0x7fffffffu) return fail(MIPT_ERR_LIMIT, "mipt_eye_rays: too many rays");
    if (count == 0) return MIPT_OK;
    HIP_TRY(hipSetDevice(c->device));
    if ((rc = c->sRays.ensure(count * 2 * sizeof(float4))) != MIPT_OK) return rc;
    {
        Timer tm(c, 2);
        count_launch(c, 2);
        launch_eye_rays(c->stream, cd, first, count, (float4*)c->sRays.p);
    }
    std::vector<float4> rays(count * 2);
    HIP_TRY(hipMemcpyAsync(rays.data(), c->sRays.p, rays.size() * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    for (size_t i = 0; i < count; i++) {
        ray_o[3 * i] = rays[2 * i].x; ray_o[3 * i + 1] = rays[2 * i].y; ray_o[3 * i + 2] = rays[2 * i].z;
        ray_d[3 * i] = rays[2 * i + 1].x; ray_d[3 * i + 1] = rays[2 * i + 1].y; ray_d[3 * i + 2] = rays[2 * i + 1].z;
    }
    return MIPT_OK;
    });
}

int mipt_background(mipt_ctx* c, const float* dir, size_t n, float* rgb) {
    if (!c || (n && (!dir || !rgb))) return fail(MIPT_ERR_INVALID, "mipt_background: null argument");
    if (n > 0x7fffffffu / 3) return fail(MIPT_ERR_LIMIT, "mipt_background: too many directions");
    if (n == 0) return MIPT_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->sAux.ensure(n * 3 * sizeof(float))) != MIPT_OK) return rc;
    if ((rc = c->sAux2.ensure(n * 3 * sizeof(float))) != MIPT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(c->sAux.p, dir, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    {
        Timer tm(c, 2);
        count_launch(c, 2);
        launch_background(c->stream, c->ds, (const float*)c->sAux.p, n, (float*)c->sAux2.p);
    }
    HIP_TRY(hipMemcpyAsync(rgb, c->sAux2.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return MIPT_OK;
}

// the side stream and one event pair per bounce iteration (created once; events carry no timing)
static int ensure_side(mipt_ctx* c, int nIter) {
    if (!c->side) {
        // MIPT_SIDE_PRIORITY=high|low (A/B runs): the shadow-ray launches ahead of / behind the extension-ray launch they
        // run beside, when the hardware has to choose between their blocks
        int lo = 0, hi = 0;
        const char* pr = getenv("MIPT_SIDE_PRIORITY");
        if (pr && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
            HIP_TRY(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, std::string(pr) == "high" ? hi : lo));
        else
            HIP_TRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    }
    while ((int)c->evShaded.size() < nIter) {
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        c->evShaded.push_back(a);
        HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        c->evOccluded.push_back(b);
    }
    return MIPT_OK;
}

// carve the integrator's buffers out of one allocation
// cap = path slots (pixel slots x samples in flight); nPix = pixel slots (the metal chain of direct mode is per pixel)
static int ensure_path_buffers(mipt_ctx* c, size_t cap, size_t nPix) {
    const int nl = c->ds.nLights > 0 ? c->ds.nLights : 1;
    if (c->pb.capacity >= cap && c->pb.nLights == nl && c->pbPixels >= nPix) return MIPT_OK;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align256(bytes); return at; };
    const size_t f4 = sizeof(float4);
    const size_t regionWaves = ((cap + 63) / 64 + kShards - 1) / kShards;
    const size_t regionSize = regionWaves * 64, space = regionSize * kShards;  // >= cap
    const size_t oRays = take(2 * cap * f4), oHit = take(cap * f4), oMult = take(cap * f4), oCnst = take(cap * f4);
    const size_t oPend = take(cap * f4), oHm = take(cap * f4), oVP = take(cap * f4), oVN = take(cap * f4), oVT = take(cap * f4);
    const size_t oContrib = take(cap * nl * f4), oOcc = take(cap * nl), oKs = take(nPix * 6 * f4), oPix = take(nPix * 4);  // (pixel index: one per pixel slot, shared by the samples in flight)
    const size_t oRes = take(cap * f4);
    const size_t oQ0 = take(space * 4), oQ1 = take(space * 4), oQA = take(space * 4);
    const size_t oSh = take(2 * space * nl * f4), oShT = take(space * nl * 4), oCnt = take(kCounterWords * 4);
    if (c->side) HIP_TRY(hipStreamSynchronize(c->side));  // nothing may be in flight while buffers are replaced
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc = c->pathMem.ensure(o);
    if (rc != MIPT_OK) { c->pb.capacity = 0; c->pbPixels = 0; return rc; }
    SlowView unused{};
    if ((rc = ensure_slow(c, c->slow, c->slowParity, space, c->stream, unused, false)) != MIPT_OK ||
        (rc = ensure_slow(c, c->slowOcc, c->slowOccParity, space * nl, c->stream, unused, false)) != MIPT_OK) {
        c->pb.capacity = 0; c->pbPixels = 0; return rc;
    }
    c->pbPixels = nPix;
    c->countersClean = false;
    unsigned char* b = (unsigned char*)c->pathMem.p;
    PathBuffers& pb = c->pb;
    pb.capacity = cap; pb.nLights = nl; pb.regionSize = (unsigned)regionSize; pb.space = (unsigned)space;
    pb.rays = (float4*)(b + oRays); pb.hit = (float4*)(b + oHit); pb.mult = (float4*)(b + oMult); pb.cnst = (float4*)(b + oCnst);
    pb.pendL = (float4*)(b + oPend); pb.hitMul = (float4*)(b + oHm); pb.vP = (float4*)(b + oVP); pb.vN = (float4*)(b + oVN);
    pb.vTex = (float4*)(b + oVT); pb.contrib = (float4*)(b + oContrib); pb.occ = b + oOcc; pb.ksStack = (float4*)(b + oKs);
    pb.res = (float4*)(b + oRes);
    pb.pixel = (int*)(b + oPix); pb.qExt[0] = (int*)(b + oQ0); pb.qExt[1] = (int*)(b + oQ1); pb.qAct = (int*)(b + oQA);
    pb.shRays = (float4*)(b + oSh); pb.shTarget = (int*)(b + oShT); pb.counters = (unsigned*)(b + oCnt);
    pb.visitCounters = c->dTotals + 2;
    return MIPT_OK;
}

int mipt_render_device(mipt_ctx* c, const mipt_camera* cam, const mipt_render_params* rp, float* accum_dev, float* rgb_dev) {
    return guarded("mipt_render_device", [&]() -> int {
    if (!c || !cam || !rp || !rgb_dev) return fail(MIPT_ERR_INVALID, "mipt_render_device: null argument");
    if (rp->mode < MIPT_MODE_PRIMARY || rp->mode > MIPT_MODE_PATH) return fail(MIPT_ERR_INVALID, "unknown render mode");
    if (rp->mode == MIPT_MODE_PATH && (!accum_dev || rp->spp <= 0 || rp->sample_offset < 0))
        return fail(MIPT_ERR_INVALID, "path mode needs accum_dev, spp > 0, sample_offset >= 0");
    const int partCount = rp->part_count > 0 ? rp->part_count : 1;
    if (rp->part_index < 0 || rp->part_index >= partCount) return fail(MIPT_ERR_INVALID, "part_index out of range");
    const int tileW = rp->tile_w > 0 ? rp->tile_w : 64, tileH = rp->tile_h > 0 ? rp->tile_h : 8;
    if ((tileW & 7) || (tileH & 7)) return fail(MIPT_ERR_INVALID, "tile_w and tile_h must be multiples of 8");
    CameraDev cd;
    int rc = make_camera(cam, cd);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));

    // owned tiles: diagonal interleave over the parts
    const int tilesX = (cam->width + tileW - 1) / tileW, tilesY = (cam->height + tileH - 1) / tileH;
    const std::vector<int> key = {cam->width, cam->height, tileW, tileH, rp->part_index, partCount};
    if (key != c->tileKey) {
        std::vector<int> tiles;
        for (int ty = 0; ty < tilesY; ty++)
            for (int tx = 0; tx < tilesX; tx++)
                if ((tx + 3 * ty) % partCount == rp->part_index) tiles.push_back(ty * tilesX + tx);
        // a frame still in flight on c->stream (which may be a non-blocking stream the null stream does not order
        // against) reads the old list: let it finish before the list is replaced (rare: only when the key changes)
        HIP_TRY(hipStreamSynchronize(c->stream));
        if ((rc = c->tileMem.ensure(tiles.size() * sizeof(int) + 4)) != MIPT_OK) return rc;
        if (!tiles.empty()) HIP_TRY(hipMemcpy(c->tileMem.p, tiles.data(), tiles.size() * sizeof(int), hipMemcpyHostToDevice));
        c->tileKey = key;
        c->nOwnedTiles = (int)tiles.size();
    }
    if (c->nOwnedTiles == 0) return MIPT_OK;
    const size_t nPix = (size_t)c->nOwnedTiles * tileW * tileH;  // pixel slots of this partition
    if (nPix > ((size_t)1 << 25)) return fail(MIPT_ERR_LIMIT, "more than 2^25 pixels in one partition");
    // path mode traces several samples of every pixel together: bigger queues keep the persistent traversal
    // wavefronts fed, and more samples of a pixel side by side put whole wavefronts on one pixel (HBM is large: ~257 B per path
    // slot, at most 2^27 slots = 34 GB; up to r3v23 2^25)
    int inFlight = 1;
    if (rp->mode == MIPT_MODE_PATH) {
        size_t slotsLog2 = 27;  // 2^27 path slots = 34 GB of path state of the GPU's 288 GB (MIPT_PATH_SLOTS_LOG2: 20 .. 28)
        if (const char* e = getenv("MIPT_PATH_SLOTS_LOG2")) { const long v = atol(e); if (v >= 20 && v <= 28) slotsLog2 = (size_t)v; }
        const size_t fit = ((size_t)1 << slotsLog2) / nPix;
        inFlight = (int)(fit < 1 ? 1 : (fit > (size_t)rp->spp ? (size_t)rp->spp : fit));
    }
    const size_t cap = nPix * (size_t)inFlight;
    if ((rc = ensure_path_buffers(c, cap, nPix)) != MIPT_OK) return rc;
    const PathBuffers& pb = c->pb;
    const int nl = pb.nLights;

    FrameParams fp{};
    fp.mode = rp->mode;
    fp.width = cam->width; fp.height = cam->height;
    fp.tileW = tileW; fp.tileH = tileH; fp.tilesX = tilesX; fp.nOwnedTiles = c->nOwnedTiles;
    fp.nTilePaths = (unsigned)nPix;
    fp.tileList = (const int*)c->tileMem.p;
    fp.accum = accum_dev; fp.rgb = rgb_dev;
    fp.seedHash = pcg_hash_host(rp->seed);
    const bool unbounded = rp->mode == MIPT_MODE_PATH && rp->max_bounces < 0;
    int lastIter;
    if (rp->mode == MIPT_MODE_PATH) {
        fp.maxBounces = (unbounded || rp->max_bounces > kMaxIter - 2) ? kMaxIter - 2 : rp->max_bounces;
        lastIter = fp.maxBounces;
    } else {
        fp.maxBounces = -1;
        // shade(): level > 5 ends every chain; without a metal or glass material no chain ever starts, and the
        // six further rounds would only launch kernels over empty queues
        lastIter = (rp->mode == MIPT_MODE_DIRECT && c->hasSpecular) ? 6 : 0;
    }
    fp.lastIter = lastIter;
    const int nSamples = rp->mode == MIPT_MODE_PATH ? rp->spp : 1;
    for (int base = 0; base < nSamples; base += inFlight) {
        fp.sample = (unsigned)(rp->sample_offset + base);
        fp.samplesInFlight = (nSamples - base) < inFlight ? (nSamples - base) : inFlight;
        // the queue counters start at zero: k_fold_counters of the frame before left them so; a memset after a new
        // allocation or a frame that did not reach its end
        if (!c->countersClean) HIP_TRY(hipMemsetAsync(pb.counters, 0, kCounterWords * 4, c->stream));
        c->countersClean = false;
        {
            Timer tm(c, 2);
            count_launch(c, 2);
            launch_generate(c->stream, cd, fp, pb);
        }
        int itDone = 0;
        const bool pathMode = rp->mode == MIPT_MODE_PATH;
        // path mode: occluded(it) runs on the side stream beside closest(it+1) (both depend only on shade(it));
        // shade(it+1) waits for both.  Everything the side stream does is waited for on c->stream before this call
        // returns, so the caller's ordering on c->stream holds.
        // ... or, when the scene qualifies for the wide traversal, both in ONE launch (trace_both): occluded(it) is then
        // deferred to the launch of closest(it+1), and the last one runs alone after the loop
        const bool fused = pathMode && c->overlap && c->fuse && PT_WIDE_K == 8 && !c->useBitTrail && c->ds.nWide > 0 && c->useWide;
        const bool overlap = pathMode && c->overlap && !fused && !c->useBitTrail && ensure_side(c, lastIter + 1) == MIPT_OK;
        bool shadowsOwed = false;  // fused: the shadow rays of bounce it-1 are still to be traced
        auto shadowQueue = [&](int it) {
            return QueueView{pb.counters + counter_index(it, 2, 0), pb.regionSize * (unsigned)nl, pb.space * (unsigned)nl};
        };
        for (int it = 0; it <= lastIter; it++) {
            const QueueView qExt{pb.counters + counter_index(it, 0, 0), pb.regionSize, pb.space};
            if (!(shadowsOwed && trace_both(c, pb.rays, pb.qExt[it & 1], qExt, pb.counters + counter_index(it, 3, 0), pb.hit, pb.shRays,
                                            pb.shTarget, shadowQueue(it - 1), pb.counters + counter_index(it - 1, 4, 0), pb.occ))) {
                if (shadowsOwed)
                    trace_occluded(c, pb.shRays, pb.shTarget, shadowQueue(it - 1), pb.counters + counter_index(it - 1, 4, 0), pb.occ, c->stream);
                trace_closest(c, pb.rays, pb.qExt[it & 1], qExt, pb.counters + counter_index(it, 3, 0), pb.hit);
            }
            shadowsOwed = false;
            if (overlap && it > 0) HIP_TRY(hipStreamWaitEvent(c->stream, c->evOccluded[(size_t)it - 1], 0));  // shade(it) reads occ of bounce it-1
            {
                Timer tm(c, pathMode ? 3 : 2);
                count_launch(c, pathMode ? 3 : 2);
                if (pathMode) launch_shade_path(c->stream, c->ds, fp, pb, it, false);
                else launch_shade_a(c->stream, c->ds, fp, pb, it);
            }
            if (fused) {
                shadowsOwed = true;
            } else if (rp->mode != MIPT_MODE_PRIMARY) {
                hipStream_t occStream = c->stream;
                if (overlap) {
                    HIP_TRY(hipEventRecord(c->evShaded[(size_t)it], c->stream));
                    HIP_TRY(hipStreamWaitEvent(c->side, c->evShaded[(size_t)it], 0));
                    occStream = c->side;
                }
                trace_occluded(c, pb.shRays, pb.shTarget,
                               QueueView{pb.counters + counter_index(it, 2, 0), pb.regionSize * (unsigned)nl,
                                         pb.space * (unsigned)nl},
                               pb.counters + counter_index(it, 4, 0), pb.occ, occStream);
                if (overlap) HIP_TRY(hipEventRecord(c->evOccluded[(size_t)it], c->side));
                if (!pathMode) {
                    Timer tm(c, 2);
                    count_launch(c, 2);
                    launch_shade_b(c->stream, c->ds, fp, pb, it);
                }
            }
            itDone = it + 1;
            if (unbounded && it >= 8 && (it & 3) == 0 && it < lastIter) {
                // the reference has no bounce cap: stop when no path is left
                std::vector<unsigned> cnt((size_t)kShards * kCntStride);
                HIP_TRY(hipMemcpyAsync(cnt.data(), pb.counters + counter_index(it + 1, 0, 0), cnt.size() * 4,
                                       hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                unsigned long long next = 0;
                for (int sh = 0; sh < kShards; sh++) next += cnt[(size_t)sh * kCntStride];
                if (next == 0) break;
            }
        }
        if (shadowsOwed)  // fused: the shadow rays of the last bounce, alone
            trace_occluded(c, pb.shRays, pb.shTarget, shadowQueue(itDone - 1), pb.counters + counter_index(itDone - 1, 4, 0), pb.occ, c->stream);
        if (overlap && itDone > 0) HIP_TRY(hipStreamWaitEvent(c->stream, c->evOccluded[(size_t)itDone - 1], 0));
        if (rp->mode == MIPT_MODE_PATH) {
            Timer tm(c, 2);
            // paths that ended at a Lambertian vertex of the last bounce wait in the "fin" queue for their shadow rays -- except
            // with one light, where the cap's paths are settled by k_resolve_samples (RES_PENDING) and only Russian roulette
            // (bounces after the fifth, C:2311) fills that queue: with a cap of at most five bounces it is empty
            const bool finEmpty = nl == 1 && !unbounded && fp.maxBounces <= 5;
            if (!finEmpty) {
                count_launch(c, 2);
                launch_shade_path(c->stream, c->ds, fp, pb, itDone, true);
            }
            count_launch(c, 2);
            launch_resolve_samples(c->stream, fp, pb);
        }
        launch_fold_counters(c->stream, pb.counters, itDone, c->dTotals);
        c->countersClean = true;
        c->acc.launches_other++;
    }
    HIP_TRY(hipGetLastError());
    return MIPT_OK;
    });
}

int mipt_render(mipt_ctx* c, const mipt_camera* cam, const mipt_render_params* rp, float* rgb_host) {
    if (!c || !cam || !rp || !rgb_host) return fail(MIPT_ERR_INVALID, "mipt_render: null argument");
    if (cam->width <= 0 || cam->height <= 0) return fail(MIPT_ERR_INVALID, "camera: bad resolution");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)3 * cam->width * cam->height * sizeof(float);
    int rc = c->frameMem.ensure(2 * align256(bytes));
    if (rc != MIPT_OK) return rc;
    float* accum = (float*)c->frameMem.p;
    float* rgb = (float*)((unsigned char*)c->frameMem.p + align256(bytes));
    HIP_TRY(hipMemsetAsync(c->frameMem.p, 0, 2 * align256(bytes), c->stream));
    rc = mipt_render_device(c, cam, rp, accum, rgb);
    if (rc != MIPT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(rgb_host, rgb, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipGetLastError());
    return MIPT_OK;
}

}  // extern "C"
