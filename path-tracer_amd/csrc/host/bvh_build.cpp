// bvh_build.cpp -- SAH-BVH build on the host + the stackless link sets.
//
// Produces the SAME tree as the reference (C: = reference cpu/src/cs488.h):
// node numbering, boxes and leaf triangle order decide traversal order and
// therefore hit-ID tie-breaks, so these rules are kept exactly:
//   * TriangleMesh::preCalc C:817-831: center = (p0 + p1 + p2) * (1/3);
//   * split axis = AABB::getLargestAxis of the node's box, C:544-552, C:1463;
//   * order along the axis = the reference's Hoare quicksort with the middle
//     element's key as pivot, C:1368-1399 (not a stable sort: ties end up
//     where THAT partition scheme leaves them);
//   * SAH sweep over the sorted order, cost = (0.1 + aL/aP*nL) + (0.1 + aR/aP*nR)
//     in fp32 with that association, first minimum wins, C:1406-1409, C:1488-1502;
//   * leaf when min_cost > n or n <= 1, C:1504; a child inherits the sweep's
//     box, C:1550-1551; preorder ids (left subtree numbered before right);
//   * links: assign_hitmiss C:1649-1659, assign_extra C:1718-1728.
// Structure differs from the reference: one index array partitioned in place,
// an explicit work stack instead of recursion, one reusable suffix-box buffer
// instead of two fresh arrays per node.
#include <algorithm>
#include <cfloat>
#include <cstring>

#include "../pt/host.hpp"

namespace pt {

namespace {

struct Box {
    vec3 lo, hi, size;
    void reset() {
        lo = vec3{FLT_MAX, FLT_MAX, FLT_MAX};
        hi = vec3{-FLT_MAX, -FLT_MAX, -FLT_MAX};
        size = vec3{0.0f, 0.0f, 0.0f};
    }
    // AABB::fit, C:554-564
    void fit(const float* p) {
        if (lo.x > p[0]) lo.x = p[0];
        if (lo.y > p[1]) lo.y = p[1];
        if (lo.z > p[2]) lo.z = p[2];
        if (hi.x < p[0]) hi.x = p[0];
        if (hi.y < p[1]) hi.y = p[1];
        if (hi.z < p[2]) hi.z = p[2];
        size = hi - lo;
    }
    void fitTriangle(const float* tri9) {
        fit(tri9);
        fit(tri9 + 3);
        fit(tri9 + 6);
    }
    float area() const { return (2.0f * (size.x * size.y + size.y * size.z + size.z * size.x)); }
    int largestAxis() const {
        if ((size.x > size.y) && (size.x > size.z)) return 0;
        if (size.y > size.z) return 1;
        return 2;
    }
};

struct Builder {
    const float* pos;
    std::vector<float> center;  // 3 per triangle
    std::vector<int32_t> index; // permuted in place
    std::vector<Box> suffix;    // scratch for the right-to-left sweep
    StacklessBVH* out;

    float key(int tri, int axis) const { return center[(size_t)3 * tri + axis]; }

    // the reference's partition scheme on index[li..ri], with an explicit stack
    void sortAxis(int axis, int li0, int ri0) {
        std::vector<std::pair<int, int>> todo;
        todo.emplace_back(li0, ri0);
        while (!todo.empty()) {
            const int li = todo.back().first, ri = todo.back().second;
            todo.pop_back();
            int i = li, j = ri;
            const float pivot = key(index[(li + ri) / 2], axis);
            while (true) {
                while (key(index[i], axis) < pivot) ++i;
                while (key(index[j], axis) > pivot) --j;
                if (i >= j) break;
                const int32_t tmp = index[i];
                index[i] = index[j];
                index[j] = tmp;
                ++i;
                --j;
            }
            if (li < (i - 1)) todo.emplace_back(li, i - 1);
            if ((j + 1) < ri) todo.emplace_back(j + 1, ri);
        }
    }
};

struct Work {
    int first, count;  // range of `index`
    Box box;
    int parent;        // node to patch, -1 for the root
    bool isRight;
    int depth;
};

}  // namespace

int StacklessBVH::build(const TriangleMesh* mesh, bool allAxes) {
    const int n = mesh->numTriangles();
    if (n <= 0) {
        set_error("BVH build: mesh has no triangles");
        return MIPT_ERR_INVALID;
    }
    Builder b;
    b.pos = mesh->positions.data();
    b.out = this;
    b.center.resize((size_t)3 * n);
    b.index.resize((size_t)n);
    b.suffix.resize((size_t)n);
    Box root;
    root.reset();
    for (int i = 0; i < n; i++) {
        const float* p = b.pos + (size_t)9 * i;
        const vec3 c = ((vec3{p[0], p[1], p[2]} + vec3{p[3], p[4], p[5]}) + vec3{p[6], p[7], p[8]}) * (1.0f / 3.0f);
        b.center[3 * i] = c.x;
        b.center[3 * i + 1] = c.y;
        b.center[3 * i + 2] = c.z;
        b.index[i] = i;
        root.fitTriangle(p);
    }

    const size_t cap = (size_t)2 * n;
    nodeNum = leafNum = depth = maxLeaf = 0;
    nodeMin.assign(3 * cap, 0.0f);
    nodeMax.assign(3 * cap, 0.0f);
    leafFirst.assign(cap, -1);
    leafCount.assign(cap, 0);
    idLeft.assign(cap, -1);
    idRight.assign(cap, -1);
    triIndex.clear();
    triIndex.reserve((size_t)n);

    std::vector<Work> stack;
    stack.push_back(Work{0, n, root, -1, false, 1});
    while (!stack.empty()) {
        const Work w = stack.back();
        stack.pop_back();
        const int id = nodeNum++;
        if (w.parent >= 0) (w.isRight ? idRight : idLeft)[w.parent] = id;
        if (w.depth > depth) depth = w.depth;
        nodeMin[3 * id] = w.box.lo.x; nodeMin[3 * id + 1] = w.box.lo.y; nodeMin[3 * id + 2] = w.box.lo.z;
        nodeMax[3 * id] = w.box.hi.x; nodeMax[3 * id + 1] = w.box.hi.y; nodeMax[3 * id + 2] = w.box.hi.z;

        const int cnt = w.count;
        int32_t* idx = b.index.data() + w.first;
        float minCost = FLT_MAX;
        int bestSplit = -1;  // left gets idx[0..bestSplit]
        Box bestL, bestR;
        const float parentArea = w.box.area();
        // the sweep along the current order of idx: the reference's cost, first minimum wins
        auto sweep = [&]() -> bool {
            bool improved = false;
            Box acc;  // suffix[k] = box of idx[k..cnt)
            acc.reset();
            for (int k = cnt - 1; k >= 1; --k) {
                acc.fitTriangle(b.pos + (size_t)9 * idx[k]);
                b.suffix[k] = acc;
            }
            Box left;
            left.reset();
            for (int k = 0; k < cnt - 1; ++k) {
                left.fitTriangle(b.pos + (size_t)9 * idx[k]);
                const Box& right = b.suffix[k + 1];
                const float costL = 0.1f + left.area() / parentArea * (k + 1) * 1.0f;
                const float costR = 0.1f + right.area() / parentArea * (cnt - k - 1) * 1.0f;
                const float cost = costL + costR;
                if (cost < minCost) {
                    bestL = left;
                    bestR = right;
                    bestSplit = k;
                    minCost = cost;
                    improved = true;
                }
            }
            return improved;
        };
        if (!allAxes) {
            b.sortAxis(w.box.largestAxis(), w.first, w.first + cnt - 1);
            sweep();
        } else {
            // quality mode (NOT the reference's tree): the same sweep along each of the three axes, the cheapest
            // split of all wins; triangles ordered by (key, index), so the tree does not depend on a sort's whims
            int bestAxis = -1;
            for (int axis = 0; axis < 3; axis++) {
                std::sort(idx, idx + cnt, [&](int32_t x, int32_t y) {
                    const float kx = b.key(x, axis), ky = b.key(y, axis);
                    return kx < ky || (kx == ky && x < y);
                });
                if (sweep()) bestAxis = axis;
            }
            if (bestAxis >= 0 && bestAxis != 2)
                std::sort(idx, idx + cnt, [&](int32_t x, int32_t y) {
                    const float kx = b.key(x, bestAxis), ky = b.key(y, bestAxis);
                    return kx < ky || (kx == ky && x < y);
                });
        }
        const bool leaf = (minCost > (float)cnt * 1.0f) || cnt <= 1;
        if (leaf) {
            leafFirst[id] = (int32_t)triIndex.size();
            leafCount[id] = cnt;
            for (int k = 0; k < cnt; k++) triIndex.push_back(idx[k]);
            if (cnt > maxLeaf) maxLeaf = cnt;
            leafNum++;
        } else {
            const int nl = bestSplit + 1;
            // right first so that the left subtree is numbered first (preorder)
            stack.push_back(Work{w.first + nl, cnt - nl, bestR, id, true, w.depth + 1});
            stack.push_back(Work{w.first, nl, bestL, id, false, w.depth + 1});
        }
    }

    nodeMin.resize((size_t)3 * nodeNum);
    nodeMax.resize((size_t)3 * nodeNum);
    leafFirst.resize(nodeNum);
    leafCount.resize(nodeNum);
    idLeft.resize(nodeNum);
    idRight.resize(nodeNum);

    // threaded links: a node's miss target is where the walk resumes when its
    // box is missed; a leaf's hit target is the same; an interior's is its left child.
    idHit.assign(nodeNum, 0);
    idMiss.assign(nodeNum, 0);
    idParent.assign(nodeNum, 0);
    idSibling.assign(nodeNum, 0);
    // preorder ids => parents come before children: one forward pass suffices
    for (int i = 0; i < nodeNum; i++) {
        if (leafCount[i] > 0) {
            idHit[i] = idMiss[i];
        } else {
            const int l = idLeft[i], r = idRight[i];
            idHit[i] = l;
            idMiss[l] = r;
            idMiss[r] = idMiss[i];
            idParent[l] = idParent[r] = i;
            idSibling[l] = r;
            idSibling[r] = l;
        }
    }
    return MIPT_OK;
}

}  // namespace pt
