/*
 * pt_oracle.c -- CPU restatement of the reference's ray/scene hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing the product ships links, loads or calls
 * this file: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, as the checker / the reported CPU baseline.
 *
 * Every function cites the reference lines it follows
 * (C: = /root/reference/cpu/src/cs488.h, L: = /root/reference/cpu/src/linalg.h).
 * Arithmetic is fp32 with the reference's exact operation order; build with
 * -O2 -ffp-contract=off (no FMA contraction, IEEE divide and sqrt).
 *
 * Parity pinning: cs488.h cannot be built in this image (it needs GLEW/GLFW
 * headers that are absent, and stand-in headers are not allowed), and the
 * reference holds no tests or golden vectors.  PINNED by a build of the
 * reference's own unmodified source (oracle/Makefile `ref`, oracle/_ref/,
 * tests/test_ref_pins.py, vectors in tests/golden/ref_linalg.npz): the linalg
 * restatement below (dot, cross, normalize, determinant, mul, division), bit
 * for bit.  PARITY UNPINNED by the rules: everything restated from cs488.h
 * (BVH build, traversals, triangle test, eyeRay, getBackground, shade,
 * shade_pathtrace, PCG32) -- it reproduces the outputs of the unmodified
 * reference that SURVEY.md 8(c) recorded (BVH node/leaf/depth counts, hit
 * counts, sum of t and the FNV-1a hash of hit-t bits for three scenes, 1-spp
 * path-traced mean radiance of four scenes: tests/test_oracle_pins.py), but
 * those were recorded from a build against stand-in GL headers, which does
 * not count as a pin.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* linalg restatement (L:367, L:401, L:500-504, L:536, L:567)          */
/* ------------------------------------------------------------------ */
typedef struct { float x, y, z; } f3;
typedef struct { float x, y; } f2;

static inline f3 v3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 add3(f3 a, f3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 cmul3(f3 a, f3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 cdiv3(f3 a, f3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline f3 scale3(f3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }   /* vec * scalar */
static inline f3 sscale3(float s, f3 a) { return v3(s * a.x, s * a.y, s * a.z); }  /* scalar * vec */
static inline f3 divs3(f3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline f3 neg3(f3 a) { return v3(-a.x, -a.y, -a.z); }
/* L:501 dot = sum(a*b) = fold(add, 0, a*b) = ((0 + x) + y) + z  (L:367, L:401) */
static inline float dot3(f3 a, f3 b) { return ((0.0f + a.x * b.x) + a.y * b.y) + a.z * b.z; }
static inline float dot2(f2 a, f2 b) { return (0.0f + a.x * b.x) + a.y * b.y; }
/* L:500 */
static inline f3 cross3(f3 a, f3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* L:502-504: normalize = a / sqrt(dot(a,a)), three divides */
static inline f3 normalize3(f3 a) { return divs3(a, sqrtf(dot3(a, a))); }
static inline f2 normalize2(f2 a) { float l = sqrtf(dot2(a, a)); f2 r = { a.x / l, a.y / l }; return r; }
/* L:567 determinant of the matrix whose COLUMNS are a, b, c */
static inline float det3(f3 a, f3 b, f3 c) {
    return a.x * (b.y * c.z - c.y * b.z) + a.y * (b.z * c.x - c.z * b.x) + a.z * (b.x * c.y - c.x * b.y);
}
/* L:536 mul(mat(c0,c1,c2), v) = c0*v.x + c1*v.y + c2*v.z */
static inline f3 mulcols3(f3 c0, f3 c1, f3 c2, f3 v) {
    return add3(add3(scale3(c0, v.x), scale3(c1, v.y)), scale3(c2, v.z));
}

/* ------------------------------------------------------------------ */
/* scene data model (C:388-433, C:510-517, C:522-542, C:615-622)       */
/* ------------------------------------------------------------------ */
enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_GLASS = 2, MAT_LIGHT = 3 };

typedef struct {
    int type;
    float eta;
    f3 Ka, Kd, Ks;
    float Ns;
    int isTextured;
    const unsigned char *texture;  /* 8-bit RGB */
    int textureWidth, textureHeight;
} Material;

typedef struct { f3 minp, maxp, size; } AABB;

typedef struct {
    f3 positions[3];
    f3 normals[3];
    f2 texcoords[3];
    int idMaterial;
    AABB bbox;
    f3 center;
} Triangle;

struct orc_scene;
typedef struct {
    float t;
    f3 P, N;
    f2 T;
    int material;  /* index into the hit OBJECT's materials instead of the reference's pointer (C:772) */
    const struct orc_scene *obj;  /* the object that was hit (what the reference's material pointer implies) */
    int triId;     /* not in the reference's HitInfo: recorded where the hit is accepted; index within the object */
    float beta, gamma;
} HitInfo;

typedef struct { f3 o, d; } Ray;
typedef struct { f3 position, wattage; } PointLightSource;

typedef struct {
    int isLeaf;
    int idLeft, idRight, idParent, idSibling;
    int triListNum;
    int triListOfs;  /* offset into the shared triList pool (reference: one new[] per leaf) */
    AABB bbox;
} BVHNode;

typedef struct { int idHit, idMiss; } CollNode;

typedef struct {
    /* camera: C:65-70, C:81-85 with width/height made parameters */
    int width, height;
    f3 eye, up, viewDir;
    float aspect, filmSize, distanceToFilm;
} Camera;

typedef struct orc_scene {
    int ntris;
    Triangle *tris;
    int nmat;
    Material *mats;
    unsigned char **texcopies;
    AABB bbox;
    /* BVH */
    BVHNode *node;
    int nodeNum, leafNum;
    int *triList;
    int triListUsed;
    CollNode *refNode;
    /* environment + lights */
    int envW, envH;
    f3 *env;
    int nlights;
    PointLightSource *lights;
    /* counters (instrumentation, not in the reference) */
    long long cntBox, cntTri;
    int maxDepth, maxLeaf;
    /* Scene::objects (C:2006, addObject C:2020): this struct is object 0 AND the scene (environment, lights); further
     * objects hang off `next`, each with its own triangles, materials and BVH (C:2030-2034).  triBase / matBase number
     * triangles and materials through all objects, for the ids the exported functions report. */
    struct orc_scene *next;
    int triBase, matBase;
} orc_scene;

/* ------------------------------------------------------------------ */
/* AABB (C:532-608)                                                    */
/* ------------------------------------------------------------------ */
static void aabb_reset(AABB *b) {
    b->minp = v3(FLT_MAX, FLT_MAX, FLT_MAX);
    b->maxp = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    b->size = v3(0.0f, 0.0f, 0.0f);
}
/* C:554-564 */
static void aabb_fit(AABB *b, f3 v) {
    if (b->minp.x > v.x) b->minp.x = v.x;
    if (b->minp.y > v.y) b->minp.y = v.y;
    if (b->minp.z > v.z) b->minp.z = v.z;
    if (b->maxp.x < v.x) b->maxp.x = v.x;
    if (b->maxp.y < v.y) b->maxp.y = v.y;
    if (b->maxp.z < v.z) b->maxp.z = v.z;
    b->size = sub3(b->maxp, b->minp);
}
/* C:544-552 */
static int aabb_largest_axis(const AABB *b) {
    if ((b->size.x > b->size.y) && (b->size.x > b->size.z)) return 0;
    else if (b->size.y > b->size.z) return 1;
    else return 2;
}
/* C:566-568 */
static float aabb_area(const AABB *b) {
    return (2.0f * (b->size.x * b->size.y + b->size.y * b->size.z + b->size.z * b->size.x));
}
/* C:571-608: slab test, writes the ENTRY distance t1 */
static int aabb_intersect(const AABB *b, float *tEntry, const Ray *ray) {
    float tx1 = (b->minp.x - ray->o.x) / ray->d.x;
    float ty1 = (b->minp.y - ray->o.y) / ray->d.y;
    float tz1 = (b->minp.z - ray->o.z) / ray->d.z;
    float tx2 = (b->maxp.x - ray->o.x) / ray->d.x;
    float ty2 = (b->maxp.y - ray->o.y) / ray->d.y;
    float tz2 = (b->maxp.z - ray->o.z) / ray->d.z;
    if (tx1 > tx2) { const float temp = tx1; tx1 = tx2; tx2 = temp; }
    if (ty1 > ty2) { const float temp = ty1; ty1 = ty2; ty2 = temp; }
    if (tz1 > tz2) { const float temp = tz1; tz1 = tz2; tz2 = temp; }
    float t1 = tx1; if (t1 < ty1) t1 = ty1; if (t1 < tz1) t1 = tz1;
    float t2 = tx2; if (t2 > ty2) t2 = ty2; if (t2 > tz2) t2 = tz2;
    if (t1 > t2) return 0;
    if ((t1 < 0.0) && (t2 < 0.0)) return 0;
    *tEntry = t1;
    return 1;
}

/* ------------------------------------------------------------------ */
/* ray-triangle test (C:777-813): Cramer's rule on four determinants   */
/* ------------------------------------------------------------------ */
static int raytraceTriangle(orc_scene *s, HitInfo *result, const Ray *ray, int triId, float tMin, float tMax) {
    const Triangle *tri = &s->tris[triId];
    s->cntTri++;
    f3 a = sub3(tri->positions[0], tri->positions[1]);
    f3 b = sub3(tri->positions[0], tri->positions[2]);
    f3 c = ray->d;
    f3 d = sub3(tri->positions[0], ray->o);

    float D = det3(a, b, c);
    float A = det3(d, b, c);
    float B = det3(a, d, c);
    float C = det3(a, b, d);

    float beta = A / D;
    float gamma = B / D;
    float alpha = 1 - beta - gamma;

    if (alpha > 1 || alpha < 0 || beta < 0 || gamma < 0) return 0;

    f3 phi = v3(alpha, beta, gamma);
    result->t = C / D;
    result->material = tri->idMaterial;
    result->obj = s;
    result->P = mulcols3(tri->positions[0], tri->positions[1], tri->positions[2], phi);
    result->N = mulcols3(tri->normals[0], tri->normals[1], tri->normals[2], phi);
    result->T.x = (tri->texcoords[0].x * phi.x + tri->texcoords[1].x * phi.y) + tri->texcoords[2].x * phi.z;
    result->T.y = (tri->texcoords[0].y * phi.x + tri->texcoords[1].y * phi.y) + tri->texcoords[2].y * phi.z;
    result->triId = triId;
    result->beta = beta;
    result->gamma = gamma;

    if (dot3(result->N, ray->d) > 0) result->N = neg3(result->N);
    return result->t > tMin && result->t < tMax && result->t > 0.00001;
}

/* ------------------------------------------------------------------ */
/* TriangleMesh::preCalc (C:817-831)                                   */
/* ------------------------------------------------------------------ */
static void mesh_preCalc(orc_scene *s) {
    aabb_reset(&s->bbox);
    for (int i = 0; i < s->ntris; i++) {
        Triangle *t = &s->tris[i];
        aabb_reset(&t->bbox);
        aabb_fit(&t->bbox, t->positions[0]);
        aabb_fit(&t->bbox, t->positions[1]);
        aabb_fit(&t->bbox, t->positions[2]);
        t->center = scale3(add3(add3(t->positions[0], t->positions[1]), t->positions[2]), (1.0f / 3.0f));
        aabb_fit(&s->bbox, t->positions[0]);
        aabb_fit(&s->bbox, t->positions[1]);
        aabb_fit(&s->bbox, t->positions[2]);
    }
}

/* ------------------------------------------------------------------ */
/* SAH-BVH build (C:1368-1595)                                         */
/* ------------------------------------------------------------------ */
static inline float center_axis(const orc_scene *s, int tri, int axis) {
    const f3 *c = &s->tris[tri].center;
    return axis == 0 ? c->x : (axis == 1 ? c->y : c->z);
}
/* C:1368-1399: Hoare quicksort, pivot = middle element's key */
static void sortAxis(const orc_scene *s, int *obj_index, int axis, int li, int ri) {
    int i = li, j = ri;
    float pivot = center_axis(s, obj_index[(li + ri) / 2], axis);
    while (1) {
        while (center_axis(s, obj_index[i], axis) < pivot) ++i;
        while (center_axis(s, obj_index[j], axis) > pivot) --j;
        if (i >= j) break;
        int temp = obj_index[i];
        obj_index[i] = obj_index[j];
        obj_index[j] = temp;
        ++i;
        --j;
    }
    if (li < (i - 1)) sortAxis(s, obj_index, axis, li, i - 1);
    if ((j + 1) < ri) sortAxis(s, obj_index, axis, j + 1, ri);
}
/* C:1402-1409 */
static inline float SAHnosplitcost(int obj_num) { const float C0 = 1; return obj_num * C0; }
static inline float SAHcost(const AABB *child, const AABB *parent, int obj_num) {
    const float Cb = .1f, C0 = 1;
    return Cb + aabb_area(child) / aabb_area(parent) * obj_num * C0;
}
/* C:1412-1558 (SAHBVH branch) */
static int splitBVH(orc_scene *s, int *obj_index, int obj_num, const AABB *bbox, int depth) {
    AABB bestbboxL, bestbboxR;
    int nosplit = 0;
    int bestAxis, bestIndex = 0;
    aabb_reset(&bestbboxL);
    aabb_reset(&bestbboxR);
    if (depth > s->maxDepth) s->maxDepth = depth;

    AABB bboxL, bboxR;
    AABB *bboxLlist = (AABB *)malloc(sizeof(AABB) * (size_t)(obj_num > 1 ? obj_num - 1 : 1));
    AABB *bboxRlist = (AABB *)malloc(sizeof(AABB) * (size_t)(obj_num > 1 ? obj_num - 1 : 1));
    int *sorted_obj_index = (int *)malloc(sizeof(int) * (size_t)obj_num);

    bestAxis = aabb_largest_axis(bbox);
    sortAxis(s, obj_index, bestAxis, 0, obj_num - 1);
    for (int i = 0; i < obj_num; ++i) sorted_obj_index[i] = obj_index[i];

    aabb_reset(&bboxL);
    aabb_reset(&bboxR);
    for (int i = 0; i < obj_num - 1; ++i) {
        const Triangle *tri = &s->tris[obj_index[i]];
        aabb_fit(&bboxL, tri->positions[0]);
        aabb_fit(&bboxL, tri->positions[1]);
        aabb_fit(&bboxL, tri->positions[2]);
        bboxLlist[i] = bboxL;
        const Triangle *tri2 = &s->tris[obj_index[obj_num - 1 - i]];
        aabb_fit(&bboxR, tri2->positions[0]);
        aabb_fit(&bboxR, tri2->positions[1]);
        aabb_fit(&bboxR, tri2->positions[2]);
        bboxRlist[i] = bboxR;
    }

    float min_cost = FLT_MAX;
    for (int i = 0; i < obj_num - 1; ++i) {
        const AABB *bbox1 = &bboxLlist[i];
        const AABB *bbox2 = &bboxRlist[obj_num - 2 - i];
        float cost1 = SAHcost(bbox1, bbox, i + 1);
        float cost2 = SAHcost(bbox2, bbox, obj_num - i - 1);
        float cost = cost1 + cost2;
        if (cost < min_cost) {
            bestbboxL = bboxLlist[i];
            bestbboxR = bboxRlist[obj_num - i - 2];
            bestIndex = i;
            min_cost = cost;
        }
    }
    if (min_cost > SAHnosplitcost(obj_num) || obj_num <= 1) nosplit = 1;
    free(bboxLlist);
    free(bboxRlist);

    if (nosplit) {
        free(sorted_obj_index);
        s->nodeNum++;
        BVHNode *n = &s->node[s->nodeNum - 1];
        n->bbox = *bbox;
        n->isLeaf = 1;
        n->triListNum = obj_num;
        n->triListOfs = s->triListUsed;
        n->idLeft = n->idRight = -1;
        for (int i = 0; i < obj_num; i++) s->triList[s->triListUsed + i] = obj_index[i];
        s->triListUsed += obj_num;
        if (obj_num > s->maxLeaf) s->maxLeaf = obj_num;
        s->leafNum++;
        return s->nodeNum - 1;
    } else {
        int obj_numL = bestIndex + 1;
        int obj_numR = obj_num - (bestIndex + 1);
        int *obj_indexL = (int *)malloc(sizeof(int) * (size_t)obj_numL);
        int *obj_indexR = (int *)malloc(sizeof(int) * (size_t)obj_numR);
        for (int i = 0; i <= bestIndex; ++i) obj_indexL[i] = sorted_obj_index[i];
        for (int i = bestIndex + 1; i < obj_num; ++i) obj_indexR[i - (bestIndex + 1)] = sorted_obj_index[i];
        free(sorted_obj_index);

        s->nodeNum++;
        int temp_id = s->nodeNum - 1;
        s->node[temp_id].bbox = *bbox;
        s->node[temp_id].isLeaf = 0;
        s->node[temp_id].triListNum = 0;
        s->node[temp_id].triListOfs = -1;
        int l = splitBVH(s, obj_indexL, obj_numL, &bestbboxL, depth + 1);
        s->node[temp_id].idLeft = l;
        int r = splitBVH(s, obj_indexR, obj_numR, &bestbboxR, depth + 1);
        s->node[temp_id].idRight = r;
        free(obj_indexL);
        free(obj_indexR);
        return temp_id;
    }
}
/* C:1649-1659 StacklessBVH::build: idHit/idMiss links */
static void assign_hitmiss(orc_scene *s, int missidx, int idx) {
    if (s->node[idx].isLeaf) s->refNode[idx].idHit = missidx;
    else {
        s->refNode[idx].idHit = s->node[idx].idLeft;
        assign_hitmiss(s, s->node[idx].idRight, s->node[idx].idLeft);
        assign_hitmiss(s, missidx, s->node[idx].idRight);
    }
    s->refNode[idx].idMiss = missidx;
}
/* C:1718-1728 StacklessBVH2::build: parent/sibling links */
static void assign_extra(orc_scene *s, int idx) {
    if (!s->node[idx].isLeaf) {
        s->node[s->node[idx].idLeft].idSibling = s->node[idx].idRight;
        s->node[s->node[idx].idRight].idSibling = s->node[idx].idLeft;
        s->node[s->node[idx].idLeft].idParent = idx;
        s->node[s->node[idx].idRight].idParent = idx;
        assign_extra(s, s->node[idx].idLeft);
        assign_extra(s, s->node[idx].idRight);
    }
}
/* C:1562-1595 BVH::build, then both stackless link sets (C:1645-1660, C:1713-1730) */
void orc_build(orc_scene *s) {
    mesh_preCalc(s);
    const int obj_num = s->ntris;
    int *obj_index = (int *)malloc(sizeof(int) * (size_t)(obj_num > 0 ? obj_num : 1));
    for (int i = 0; i < obj_num; ++i) obj_index[i] = i;
    free(s->node); free(s->triList); free(s->refNode);
    s->nodeNum = 0;
    s->node = (BVHNode *)calloc((size_t)(obj_num * 2 + 1), sizeof(BVHNode));
    s->triList = (int *)malloc(sizeof(int) * (size_t)(obj_num > 0 ? obj_num : 1));
    s->triListUsed = 0;
    s->leafNum = 0;
    s->maxDepth = 0;
    s->maxLeaf = 0;
    AABB bbox;
    aabb_reset(&bbox);
    for (int i = 0; i < obj_num; i++) {
        const Triangle *tri = &s->tris[obj_index[i]];
        aabb_fit(&bbox, tri->positions[0]);
        aabb_fit(&bbox, tri->positions[1]);
        aabb_fit(&bbox, tri->positions[2]);
    }
    splitBVH(s, obj_index, obj_num, &bbox, 1);
    free(obj_index);
    s->refNode = (CollNode *)calloc((size_t)s->nodeNum, sizeof(CollNode));
    assign_hitmiss(s, 0, 0);
    s->node[0].idSibling = 0;
    s->node[0].idParent = 0;
    assign_extra(s, 0);
}

/* ------------------------------------------------------------------ */
/* traversals                                                          */
/* ------------------------------------------------------------------ */
/* C:1661-1683 StacklessBVH::intersect (threaded idHit/idMiss) */
static int threaded_intersect(orc_scene *s, HitInfo *minHit, const Ray *ray, float tMin, float tMax) {
    HitInfo tempMinHit;
    int node_idx = 0;
    int hit = 0;
    minHit->t = FLT_MAX;
    minHit->triId = -1;
    do {
        s->cntBox++;
        if (aabb_intersect(&s->node[node_idx].bbox, &tempMinHit.t, ray)) {
            if (s->node[node_idx].isLeaf) {
                for (int i = 0; i < s->node[node_idx].triListNum; i++)
                    if (raytraceTriangle(s, &tempMinHit, ray, s->triList[s->node[node_idx].triListOfs + i], tMin, tMax)) {
                        hit = 1;
                        if (tempMinHit.t < minHit->t) *minHit = tempMinHit;
                    }
            }
            node_idx = s->refNode[node_idx].idHit;
        } else {
            node_idx = s->refNode[node_idx].idMiss;
        }
    } while (node_idx != 0);
    return hit;
}
/* C:1684-1705 StacklessBVH::isHit (any-hit) */
static int threaded_isHit(orc_scene *s, const Ray *ray, float tMin, float tMax) {
    HitInfo tempMinHit;
    int node_idx = 0;
    do {
        s->cntBox++;
        if (aabb_intersect(&s->node[node_idx].bbox, &tempMinHit.t, ray)) {
            if (s->node[node_idx].isLeaf) {
                for (int i = 0; i < s->node[node_idx].triListNum; i++)
                    if (raytraceTriangle(s, &tempMinHit, ray, s->triList[s->node[node_idx].triListOfs + i], tMin, tMax))
                        return 1;
            }
            node_idx = s->refNode[node_idx].idHit;
        } else {
            node_idx = s->refNode[node_idx].idMiss;
        }
    } while (node_idx != 0);
    return 0;
}
/* C:1732-1803 StacklessBVH2::intersect (parent/sibling + 32-bit trail) */
static int bittrail_intersect(orc_scene *s, HitInfo *minHit, const Ray *ray, float tMin, float tMax) {
    HitInfo tempMinHit;
    int node_idx = 0;
    int bittrail = 0;
    int hit = 0;
    minHit->t = FLT_MAX;
    minHit->triId = -1;
    float tL = 0, tR = 0;
    const BVHNode *node = s->node;
    s->cntBox++;
    if (!aabb_intersect(&node[node_idx].bbox, &tempMinHit.t, ray)) return 0;
    while (1) {
        while (!node[node_idx].isLeaf) {
            s->cntBox += 2;
            int hit1 = aabb_intersect(&node[node[node_idx].idLeft].bbox, &tL, ray);
            int hit2 = aabb_intersect(&node[node[node_idx].idRight].bbox, &tR, ray);
            hit1 = hit1 && (tL < minHit->t);
            hit2 = hit2 && (tR < minHit->t);
            if (hit1 || hit2) bittrail = (int)((unsigned)bittrail << 1);
            if (hit1 && hit2) {
                bittrail ^= 1;
                if (tL < tR) node_idx = node[node_idx].idLeft;
                else node_idx = node[node_idx].idRight;
            } else if (hit1) node_idx = node[node_idx].idLeft;
            else if (hit2) node_idx = node[node_idx].idRight;
            else break;
        }
        if (node[node_idx].isLeaf) {
            for (int i = 0; i < node[node_idx].triListNum; i++)
                if (raytraceTriangle(s, &tempMinHit, ray, s->triList[node[node_idx].triListOfs + i], tMin, tMax)) {
                    hit = 1;
                    if (tempMinHit.t < minHit->t) *minHit = tempMinHit;
                }
        }
        while (1) {
            if (bittrail == 0) return hit;
            while ((bittrail & 1) == 0) {
                bittrail >>= 1;
                node_idx = node[node_idx].idParent;
            }
            if (bittrail & 0x1) {
                node_idx = node[node_idx].idSibling;
                bittrail ^= 0x1;
                s->cntBox++;
                aabb_intersect(&node[node_idx].bbox, &tempMinHit.t, ray);
                if (tempMinHit.t < minHit->t) break;
            }
        }
    }
    return hit;
}
/* C:1345-1357 + C:1599-1634 recursive BVH */
static int rec_traverse(orc_scene *s, HitInfo *minHit, const Ray *ray, int node_id, float tMin, float tMax) {
    int hit = 0;
    HitInfo tempMinHit;
    float tL = 0, tR = 0;
    const BVHNode *node = s->node;
    if (node[node_id].isLeaf) {
        for (int i = 0; i < node[node_id].triListNum; ++i) {
            if (raytraceTriangle(s, &tempMinHit, ray, s->triList[node[node_id].triListOfs + i], tMin, tMax)) {
                hit = 1;
                if (tempMinHit.t < minHit->t) *minHit = tempMinHit;
            }
        }
    } else {
        s->cntBox += 2;
        int hit1 = aabb_intersect(&node[node[node_id].idLeft].bbox, &tL, ray);
        int hit2 = aabb_intersect(&node[node[node_id].idRight].bbox, &tR, ray);
        hit1 = hit1 && (tL < minHit->t);
        hit2 = hit2 && (tR < minHit->t);
        if (hit1 && hit2) {
            if (tL < tR) {
                hit = rec_traverse(s, minHit, ray, node[node_id].idLeft, tMin, tMax);
                hit = rec_traverse(s, minHit, ray, node[node_id].idRight, tMin, tMax);
            } else {
                hit = rec_traverse(s, minHit, ray, node[node_id].idRight, tMin, tMax);
                hit = rec_traverse(s, minHit, ray, node[node_id].idLeft, tMin, tMax);
            }
        } else if (hit1) {
            hit = rec_traverse(s, minHit, ray, node[node_id].idLeft, tMin, tMax);
        } else if (hit2) {
            hit = rec_traverse(s, minHit, ray, node[node_id].idRight, tMin, tMax);
        }
    }
    return hit;
}
static int recursive_intersect(orc_scene *s, HitInfo *result, const Ray *ray, float tMin, float tMax) {
    int hit = 0;
    float t0;
    result->t = FLT_MAX;
    result->triId = -1;
    s->cntBox++;
    if (aabb_intersect(&s->node[0].bbox, &t0, ray)) hit = rec_traverse(s, result, ray, 0, tMin, tMax);
    if (result->t != FLT_MAX) hit = 1;
    return hit;
}
/* C:931-947 */
static int bruteforce_intersect(orc_scene *s, HitInfo *result, const Ray *ray, float tMin, float tMax) {
    int hit = 0;
    HitInfo tempMinHit;
    result->t = FLT_MAX;
    result->triId = -1;
    for (int i = 0; i < s->ntris; ++i) {
        if (raytraceTriangle(s, &tempMinHit, ray, i, tMin, tMax)) {
            if (tempMinHit.t < result->t) {
                hit = 1;
                *result = tempMinHit;
            }
        }
    }
    return hit;
}

enum { ORC_THREADED = 0, ORC_BITTRAIL = 1, ORC_RECURSIVE = 2, ORC_BRUTEFORCE = 3 };

static int bvh_intersect(orc_scene *s, int variant, HitInfo *h, const Ray *ray, float tMin, float tMax) {
    switch (variant) {
    case ORC_THREADED: return threaded_intersect(s, h, ray, tMin, tMax);
    case ORC_BITTRAIL: return bittrail_intersect(s, h, ray, tMin, tMax);
    case ORC_RECURSIVE: return recursive_intersect(s, h, ray, tMin, tMax);
    default: return bruteforce_intersect(s, h, ray, tMin, tMax);
    }
}
/* C:2070-2085 Scene::intersect: every object's BVH in turn, strict '<' between objects (the earlier object keeps a tie) */
static int scene_intersect(orc_scene *s, int variant, HitInfo *minHit, const Ray *ray, float tMin, float tMax) {
    int hit = 0;
    HitInfo tempMinHit;
    minHit->t = FLT_MAX;
    minHit->triId = -1;
    minHit->obj = s;
    for (orc_scene *o = s; o; o = o->next) {
        if (bvh_intersect(o, variant, &tempMinHit, ray, tMin, tMax)) {
            if (tempMinHit.t < minHit->t) {
                hit = 1;
                *minHit = tempMinHit;
            }
        }
    }
    return hit;
}
/* C:2056-2068 Scene::isHit: "does a full intersect of some object return true", objects in order, stops at the first */
static int scene_isHit(orc_scene *s, int variant, const Ray *ray, float tMin, float tMax) {
    HitInfo tempMinHit;
    for (orc_scene *o = s; o; o = o->next)
        if (bvh_intersect(o, variant, &tempMinHit, ray, tMin, tMax)) return 1;
    return 0;
}

/* ------------------------------------------------------------------ */
/* camera (C:2183-2197), background (C:2038-2053), texture (C:445-457) */
/* ------------------------------------------------------------------ */
static Ray eyeRay(const Camera *cam, int x, int y) {
    const f3 wDir = normalize3(neg3(cam->viewDir));
    const f3 uDir = normalize3(cross3(cam->up, wDir));
    const f3 vDir = cross3(wDir, uDir);
    const float imPlaneUPos = (x + 0.5f) / (float)cam->width - 0.5f;
    const float imPlaneVPos = (y + 0.5f) / (float)cam->height - 0.5f;
    f3 pixelPos = add3(cam->eye, sscale3((float)(cam->aspect * cam->filmSize * imPlaneUPos), uDir));
    pixelPos = add3(pixelPos, sscale3((float)(cam->filmSize * imPlaneVPos), vDir));
    pixelPos = sub3(pixelPos, sscale3(cam->distanceToFilm, wDir));
    Ray r;
    r.o = cam->eye;
    r.d = normalize3(sub3(pixelPos, cam->eye));
    return r;
}
/* int(float) with the out-of-range/NaN case made explicit: x86 cvttss2si
 * returns INT_MIN ("integer indefinite"); the build's GPU code uses the same
 * guard so both sides agree. */
static inline int float_to_int(float v) {
    if (!(v > -2147483648.0f && v < 2147483648.0f)) return (int)0x80000000u;
    return (int)v;
}
static f3 getBackground(const orc_scene *s, f3 dir) {
    float r = acosf(dir.z) / 6.283184f;
    f2 dxy = { dir.x, dir.y };
    f2 xy = normalize2(dxy);
    f2 tex = { r * xy.x + .5f, -r * xy.y + .5f };
    int x = float_to_int(tex.x * s->envW) % s->envW;
    int y = float_to_int(tex.y * s->envH) % s->envH;
    if (x < 0) x += s->envW;
    if (y < 0) y += s->envH;
    int pix = (x + y * s->envW);
    return s->env[pix];
}
static f3 fetchTexture(const Material *m, f2 tex) {
    int x = float_to_int(tex.x * m->textureWidth) % m->textureWidth;
    int y = float_to_int(tex.y * m->textureHeight) % m->textureHeight;
    if (x < 0) x += m->textureWidth;
    if (y < 0) y += m->textureHeight;
    int pix = (x + y * m->textureWidth) * 3;
    const unsigned char r = m->texture[pix + 0];
    const unsigned char g = m->texture[pix + 1];
    const unsigned char b = m->texture[pix + 2];
    return divs3(v3(r, g, b), 255.0f);
}

/* ------------------------------------------------------------------ */
/* RNG                                                                 */
/* ------------------------------------------------------------------ */
static const float PI = 3.14159265358979f; /* C:50 */

/* C:134-147 PCG32 (global sequential stream) */
static uint64_t mcg_state = 0xcafef00dd15ea5e5u;
static uint32_t pcg32_fast(void) {
    uint64_t x = mcg_state;
    const unsigned count = (unsigned)(x >> 61);
    mcg_state = x * 6364136223846793005u;
    x ^= x >> 22;
    return (uint32_t)(x >> (22 + count));
}
/* counter RNG of the build (not in the reference): one 32-bit draw per
 * (seed, pixel, sample, dimension), dimension = index of the draw along the
 * path.  Three nested rounds of the PCG output hash. */
static inline uint32_t pcg_hash(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
uint32_t orc_counter_u32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
    return pcg_hash(pixel + pcg_hash(sample + pcg_hash(dim + pcg_hash(seed))));
}
typedef struct {
    int mode; /* 0 = PCG32 global, 1 = counter */
    uint32_t seed, pixel, sample, dim;
} Rng;
/* C:144-146: float(double(u32) / 2^32), can round to 1.0f */
static float rng_rand(Rng *r) {
    uint32_t u;
    if (r->mode == 0) u = pcg32_fast();
    else u = orc_counter_u32(r->seed, r->pixel, r->sample, r->dim++);
    return (float)((double)u / 4294967296.0);
}

/* ------------------------------------------------------------------ */
/* Material (C:459-503)                                                */
/* ------------------------------------------------------------------ */
static f3 mat_BRDF(const Material *m) {
    f3 brdfValue = v3(0.0f, 0.0f, 0.0f);
    if (m->type == MAT_LAMBERTIAN) brdfValue = divs3(m->Kd, PI);
    else if (m->type == MAT_METAL) brdfValue = m->Ks;
    return brdfValue;
}
static float mat_PDF(const Material *m) {
    float pdfValue = 0.0f;
    if (m->type == MAT_LAMBERTIAN) pdfValue = 1 / (PI);
    else if (m->type == MAT_METAL) pdfValue = 1;
    else if (m->type == MAT_GLASS) pdfValue = 1;
    return pdfValue;
}
static f3 mat_sampler(const Material *m, f3 wGiven, f3 norm, float *pdfValue, Rng *rng) {
    f3 smp = v3(0.0f, 0.0f, 0.0f);
    if (m->type == MAT_LAMBERTIAN) {
        float r1 = 2 * PI * rng_rand(rng), r2 = rng_rand(rng), r2s = sqrtf(r2);
        f3 w = norm;
        f3 u = normalize3(cross3(fabsf(w.x) > .1 ? v3(0, 1, 0) : v3(1, 0, 0), w));
        f3 v = cross3(w, u);
        f3 s1 = scale3(scale3(u, cosf(r1)), r2s);
        f3 s2 = scale3(scale3(v, sinf(r1)), r2s);
        f3 s3 = scale3(w, sqrtf(1 - r2));
        smp = normalize3(add3(add3(s1, s2), s3));
    } else if (m->type == MAT_METAL) {
        f3 nw = neg3(wGiven);
        smp = normalize3(sub3(nw, scale3(norm, 2 * dot3(nw, norm))));
    }
    *pdfValue = mat_PDF(m);
    return smp;
}

/* ------------------------------------------------------------------ */
/* integrators                                                         */
/* ------------------------------------------------------------------ */
typedef struct {
    orc_scene *s;
    int variant;       /* traversal used by Scene::intersect / isHit */
    int max_bounces;   /* extension of the build: <0 = unbounded as in the reference */
    long long rays;    /* closest-hit + any-hit traversal calls */
} Ctx;

static int ctx_intersect(Ctx *c, HitInfo *h, f3 o, f3 d, float tMin, float tMax) {
    Ray r; r.o = o; r.d = d;
    c->rays++;
    return scene_intersect(c->s, c->variant, h, &r, tMin, tMax);
}
static int ctx_isHit(Ctx *c, f3 o, f3 d, float tMin, float tMax) {
    Ray r; r.o = o; r.d = d;
    c->rays++;
    return scene_isHit(c->s, c->variant, &r, tMin, tMax);
}
static inline float fmax_std(float a, float b) { return (a < b) ? b : a; } /* std::max */

/* C:2267-2397 shade_pathtrace.  The bounce cap is the build's extension
 * (SURVEY 7 "hard parts"): after max_bounces extension rays a Lambertian
 * vertex returns constTerm + multTerm*L (as on a miss, before any RNG draw)
 * and a metal/glass vertex returns constTerm. */
static f3 shade_pathtrace(Ctx *c, const HitInfo *hit, f3 viewDir, int level, float eta, Rng *rng) {
    orc_scene *s = c->s;
    f3 currDir = viewDir;
    f3 multTerm = v3(1, 1, 1);
    f3 constTerm = v3(0.0f, 0.0f, 0.0f);
    int currlevel = level;
    HitInfo recHit = *hit;
    float currEta = eta;

    while (1) {
        const Material *m = &recHit.obj->mats[recHit.material];
        const int capped = (c->max_bounces >= 0) && (currlevel - level >= c->max_bounces);
        if (m->type == MAT_LAMBERTIAN) {
            f3 L = v3(0.0f, 0.0f, 0.0f);
            f3 brdf, irradiance;
            f3 texture = v3(1, 1, 1);
            if (m->isTextured) texture = cmul3(texture, fetchTexture(m, recHit.T));
            for (int i = 0; i < s->nlights; i++) {
                f3 l = sub3(s->lights[i].position, recHit.P);
                const float falloff = dot3(l, l);
                l = divs3(l, sqrtf(falloff));
                brdf = mat_BRDF(m);
                irradiance = sscale3((float)(fmax_std(0.0f, dot3(recHit.N, l)) / (4.0 * PI * falloff)), s->lights[i].wattage);
                if (dot3(recHit.N, l) > 0 && ctx_isHit(c, recHit.P, normalize3(l), 0, sqrtf(falloff))) continue;
                L = add3(L, cmul3(cmul3(irradiance, texture), brdf));
            }
            if (capped) return add3(constTerm, cmul3(multTerm, L));
            if (currlevel > 5) {
                if (rng_rand(rng) < .25f) return add3(constTerm, cmul3(multTerm, L));
            }
            float pdf;
            f3 sample = mat_sampler(m, currDir, recHit.N, &pdf, rng);
            f3 sample_brdf = mat_BRDF(m);
            HitInfo tempHit;
            int hasHit = ctx_intersect(c, &tempHit, recHit.P, sample, 0.001, FLT_MAX);
            if (hasHit) {
                constTerm = add3(constTerm, cmul3(L, multTerm));
                multTerm = cmul3(multTerm, divs3(cmul3(texture, sample_brdf), pdf));
                currDir = sample;
                currlevel += 1;
                recHit = tempHit;
            } else {
                return add3(constTerm, cmul3(multTerm, L));
            }
        } else if (m->type == MAT_METAL) {
            if (capped) return constTerm;
            float pdf;
            f3 sample = mat_sampler(m, currDir, recHit.N, &pdf, rng);
            f3 sample_brdf = mat_BRDF(m);
            HitInfo tempHit;
            int hasHit = ctx_intersect(c, &tempHit, recHit.P, sample, 0.001, FLT_MAX);
            if (hasHit) {
                multTerm = cmul3(multTerm, sample_brdf);
                currDir = sample;
                currlevel += 1;
                recHit = tempHit;
            } else
                return add3(constTerm, cmul3(multTerm, getBackground(s, sample)));
        } else if (m->type == MAT_GLASS) {
            if (capped) return constTerm;
            float eta2 = m->eta == currEta ? 1 : m->eta;
            f3 w = neg3(currDir);
            f3 n = recHit.N;
            float inside = 1 - pow(currEta / eta2, 2) * (1 - pow(dot3(w, n), 2));
            HitInfo tempHit;
            if (inside >= 0) {
                f3 refracted = neg3(normalize3(sub3(sscale3(currEta / eta2, sub3(w, sscale3(dot3(w, n), n))),
                                                    sscale3(sqrtf(inside), n))));
                int hasHit = ctx_intersect(c, &tempHit, recHit.P, neg3(refracted), 0.01, FLT_MAX);
                if (!hasHit) return add3(constTerm, cmul3(multTerm, getBackground(s, neg3(refracted))));
                currDir = refracted;
                currlevel += 1;
                recHit = tempHit;
                currEta = eta2;
            } else {
                f3 reflected = normalize3(sub3(w, sscale3(2 * dot3(w, n), n)));
                int hasHit = ctx_intersect(c, &tempHit, recHit.P, reflected, 0.0f, FLT_MAX);
                if (!hasHit) return add3(constTerm, cmul3(multTerm, getBackground(s, reflected)));
                currDir = reflected;
                currlevel += 1;
                recHit = tempHit;
            }
        } else {
            return v3(0, 0.0f, 0);
        }
    }
}

/* C:2402-2496 shade (direct lighting + recursive mirror/glass).  The
 * hemisphere-direction loop C:2437-2451 is dead (total_angles_h/v = 0). */
static f3 shade(Ctx *c, const HitInfo *hit, f3 viewDir, int level, float eta) {
    orc_scene *s = c->s;
    const Material *m = &hit->obj->mats[hit->material];
    if (m->type == MAT_LAMBERTIAN) {
        f3 L = v3(0.0f, 0.0f, 0.0f);
        f3 brdf, irradiance;
        for (int i = 0; i < s->nlights; i++) {
            f3 l = sub3(s->lights[i].position, hit->P);
            const float falloff = dot3(l, l);
            l = divs3(l, sqrtf(falloff));
            irradiance = sscale3((float)(fmax_std(0.0f, dot3(hit->N, l)) / (4.0 * PI * falloff)), s->lights[i].wattage);
            brdf = mat_BRDF(m);
            if (m->isTextured) brdf = cmul3(brdf, fetchTexture(m, hit->T));
            if (dot3(hit->N, l) > 0 && ctx_isHit(c, hit->P, normalize3(l), 0, sqrtf(falloff))) {
                irradiance = cmul3(v3(0.1f, 0.1f, 0.1f), irradiance);
            }
            L = add3(L, cmul3(irradiance, brdf));
        }
        return L;
    } else if (m->type == MAT_METAL) {
        if (level > 5) return v3((float).1, (float).1, (float).1);
        HitInfo tempHit;
        f3 nv = neg3(viewDir);
        f3 reflected = normalize3(sub3(nv, sscale3(2 * dot3(nv, hit->N), hit->N)));
        int hasHit = ctx_intersect(c, &tempHit, hit->P, reflected, 0.001, FLT_MAX);
        if (!hasHit) return getBackground(s, neg3(reflected));
        return cmul3(m->Ks, shade(c, &tempHit, reflected, level + 1, eta));
    } else if (m->type == MAT_GLASS) {
        if (level > 5) return v3(0, 0, 0);
        float eta2 = m->eta == eta ? 1 : m->eta;
        f3 w = neg3(viewDir);
        f3 n = hit->N;
        float inside = 1 - pow(eta / eta2, 2) * (1 - pow(dot3(w, n), 2));
        HitInfo tempHit;
        if (inside >= 0) {
            f3 refracted = normalize3(sub3(sscale3(eta / eta2, sub3(w, sscale3(dot3(w, n), n))), sscale3(sqrtf(inside), n)));
            int hasHit = ctx_intersect(c, &tempHit, hit->P, refracted, 0.0f, FLT_MAX);
            if (!hasHit) return getBackground(s, neg3(refracted));
            return shade(c, &tempHit, neg3(refracted), level + 1, eta2);
        } else {
            f3 reflected = normalize3(sub3(w, sscale3(2 * dot3(w, n), n)));
            int hasHit = ctx_intersect(c, &tempHit, hit->P, reflected, 0.0f, FLT_MAX);
            if (!hasHit) return getBackground(s, neg3(reflected));
            return shade(c, &tempHit, reflected, level + 1, eta);
        }
    } else {
        return v3(100.0f, 0.0f, 100.0f);
    }
}

/* ------------------------------------------------------------------ */
/* exported C interface (ctypes)                                       */
/* ------------------------------------------------------------------ */
orc_scene *orc_scene_create(int ntris, const float *pos, const float *nrm, const float *uv, const int *mat_id) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->ntris = ntris;
    s->tris = (Triangle *)calloc((size_t)(ntris > 0 ? ntris : 1), sizeof(Triangle));
    for (int i = 0; i < ntris; i++) {
        for (int k = 0; k < 3; k++) {
            s->tris[i].positions[k] = v3(pos[9 * i + 3 * k], pos[9 * i + 3 * k + 1], pos[9 * i + 3 * k + 2]);
            s->tris[i].normals[k] = v3(nrm[9 * i + 3 * k], nrm[9 * i + 3 * k + 1], nrm[9 * i + 3 * k + 2]);
            s->tris[i].texcoords[k].x = uv[6 * i + 2 * k];
            s->tris[i].texcoords[k].y = uv[6 * i + 2 * k + 1];
        }
        s->tris[i].idMaterial = mat_id ? mat_id[i] : 0;
    }
    return s;
}
/* materials: per material 12 floats {type, eta, Ka3, Kd3, Ks3, Ns} and a
 * texture slot (NULL = untextured) */
void orc_scene_set_materials(orc_scene *s, int nmat, const float *params, const unsigned char **tex,
                             const int *texw, const int *texh) {
    s->nmat = nmat;
    s->mats = (Material *)calloc((size_t)nmat, sizeof(Material));
    s->texcopies = (unsigned char **)calloc((size_t)nmat, sizeof(unsigned char *));
    for (int i = 0; i < nmat; i++) {
        const float *p = params + 12 * i;
        Material *m = &s->mats[i];
        m->type = (int)p[0];
        m->eta = p[1];
        m->Ka = v3(p[2], p[3], p[4]);
        m->Kd = v3(p[5], p[6], p[7]);
        m->Ks = v3(p[8], p[9], p[10]);
        m->Ns = p[11];
        if (tex && tex[i]) {
            size_t n = (size_t)texw[i] * (size_t)texh[i] * 3;
            s->texcopies[i] = (unsigned char *)malloc(n);
            memcpy(s->texcopies[i], tex[i], n);
            m->isTextured = 1;
            m->texture = s->texcopies[i];
            m->textureWidth = texw[i];
            m->textureHeight = texh[i];
        }
    }
}
void orc_scene_set_env(orc_scene *s, int w, int h, const float *rgb) {
    free(s->env);
    s->envW = w; s->envH = h;
    s->env = (f3 *)malloc(sizeof(f3) * (size_t)w * (size_t)h);
    memcpy(s->env, rgb, sizeof(f3) * (size_t)w * (size_t)h);
}
void orc_scene_set_lights(orc_scene *s, int n, const float *pos_wattage) {
    free(s->lights);
    s->nlights = n;
    s->lights = (PointLightSource *)calloc((size_t)(n > 0 ? n : 1), sizeof(PointLightSource));
    for (int i = 0; i < n; i++) {
        s->lights[i].position = v3(pos_wattage[6 * i], pos_wattage[6 * i + 1], pos_wattage[6 * i + 2]);
        s->lights[i].wattage = v3(pos_wattage[6 * i + 3], pos_wattage[6 * i + 4], pos_wattage[6 * i + 5]);
    }
}
/* Scene::addObject (C:2020): obj (built by the caller with orc_build, as preCalc builds one BVH per object, C:2030-2034)
 * becomes the scene's last object.  The scene does not own it. */
void orc_scene_add_object(orc_scene *s, orc_scene *obj) {
    orc_scene *last = s;
    while (last->next) last = last->next;
    obj->triBase = last->triBase + last->ntris;
    obj->matBase = last->matBase + last->nmat;
    obj->next = NULL;
    last->next = obj;
}
void orc_scene_destroy(orc_scene *s) {
    if (!s) return;
    for (int i = 0; i < s->nmat; i++) free(s->texcopies[i]);
    free(s->texcopies); free(s->mats); free(s->tris); free(s->node); free(s->triList);
    free(s->refNode); free(s->env); free(s->lights); free(s);
}
/* stats[0..4] = nodeNum, leafNum, maxDepth, maxLeaf, triListUsed */
void orc_bvh_stats(const orc_scene *s, int *stats) {
    stats[0] = s->nodeNum; stats[1] = s->leafNum; stats[2] = s->maxDepth; stats[3] = s->maxLeaf; stats[4] = s->triListUsed;
}
/* flat copies of the tree, in the layout the build's host library exports */
void orc_bvh_export(const orc_scene *s, float *node_min, float *node_max, int *id_hit, int *id_miss,
                    int *leaf_first, int *leaf_count, int *id_left, int *id_right, int *id_parent,
                    int *id_sibling, int *tri_index) {
    for (int i = 0; i < s->nodeNum; i++) {
        const BVHNode *n = &s->node[i];
        node_min[3 * i] = n->bbox.minp.x; node_min[3 * i + 1] = n->bbox.minp.y; node_min[3 * i + 2] = n->bbox.minp.z;
        node_max[3 * i] = n->bbox.maxp.x; node_max[3 * i + 1] = n->bbox.maxp.y; node_max[3 * i + 2] = n->bbox.maxp.z;
        id_hit[i] = s->refNode[i].idHit;
        id_miss[i] = s->refNode[i].idMiss;
        leaf_first[i] = n->isLeaf ? n->triListOfs : -1;
        leaf_count[i] = n->isLeaf ? n->triListNum : 0;
        id_left[i] = n->idLeft; id_right[i] = n->idRight;
        id_parent[i] = n->idParent; id_sibling[i] = n->idSibling;
    }
    memcpy(tri_index, s->triList, sizeof(int) * (size_t)s->triListUsed);
}
void orc_make_camera(Camera *cam, int width, int height, const float *eye, const float *lookat, const float *up,
                     float fov_deg, float film) {
    cam->width = width; cam->height = height;
    cam->eye = v3(eye[0], eye[1], eye[2]);
    cam->up = normalize3(v3(up[0], up[1], up[2]));                                     /* C:83 */
    cam->viewDir = normalize3(sub3(v3(lookat[0], lookat[1], lookat[2]), cam->eye));    /* C:2609 */
    cam->aspect = (float)(width / (float)height);                                      /* C:65 */
    cam->filmSize = film;                                                              /* C:69 */
    const float DegToRad = PI / 180.0f;                                                /* C:51 */
    cam->distanceToFilm = film / (2.0f * tanf(fov_deg * DegToRad * 0.5f));             /* C:70 */
}
/* eye rays for pixels [first, first+count) in row-major (j,i) order */
void orc_eye_rays(const Camera *cam, long long first, long long count, float *o, float *d) {
    for (long long p = 0; p < count; p++) {
        long long pix = first + p;
        Ray r = eyeRay(cam, (int)(pix % cam->width), (int)(pix / cam->width));
        o[3 * p] = r.o.x; o[3 * p + 1] = r.o.y; o[3 * p + 2] = r.o.z;
        d[3 * p] = r.d.x; d[3 * p + 1] = r.d.y; d[3 * p + 2] = r.d.z;
    }
}
/* closest-hit records for caller-supplied rays.  Outputs may be NULL.
 * hitrec: per ray 10 floats {t, P3, N3, T2, (float)material}; tri_id = -1 on miss.
 * counts[0..1] += node visits / box tests, triangle tests. */
void orc_intersect(orc_scene *s, int variant, long long n, const float *o, const float *d, float tMin, float tMax,
                   int *tri_id, float *t, float *bary, float *hitrec, long long *counts) {
    for (orc_scene *ob = s; ob; ob = ob->next) ob->cntBox = ob->cntTri = 0;
    for (long long i = 0; i < n; i++) {
        Ray r;
        r.o = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
        r.d = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        HitInfo h;
        memset(&h, 0, sizeof h);
        int hit = scene_intersect(s, variant, &h, &r, tMin, tMax);
        if (tri_id) tri_id[i] = hit ? h.obj->triBase + h.triId : -1;
        if (t) t[i] = hit ? h.t : FLT_MAX;
        if (bary) { bary[2 * i] = hit ? h.beta : 0.0f; bary[2 * i + 1] = hit ? h.gamma : 0.0f; }
        if (hitrec) {
            float *q = hitrec + 10 * i;
            if (hit) {
                q[0] = h.t; q[1] = h.P.x; q[2] = h.P.y; q[3] = h.P.z; q[4] = h.N.x; q[5] = h.N.y; q[6] = h.N.z;
                q[7] = h.T.x; q[8] = h.T.y; q[9] = (float)(h.obj->matBase + h.material);
            } else {
                q[0] = FLT_MAX;
                for (int k = 1; k < 10; k++) q[k] = 0.0f;
                q[9] = -1.0f;
            }
        }
    }
    if (counts)
        for (orc_scene *ob = s; ob; ob = ob->next) { counts[0] += ob->cntBox; counts[1] += ob->cntTri; }
}
/* any-hit (C:1684-1705 via the threaded walk, or "full intersect returns true" C:2056-2068) */
void orc_occluded(orc_scene *s, int variant, long long n, const float *o, const float *d, const float *tmin,
                  const float *tmax, unsigned char *occluded) {
    for (long long i = 0; i < n; i++) {
        Ray r;
        r.o = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
        r.d = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        if (variant == ORC_THREADED) {
            occluded[i] = 0;
            for (orc_scene *ob = s; ob && !occluded[i]; ob = ob->next) occluded[i] = (unsigned char)threaded_isHit(ob, &r, tmin[i], tmax[i]);
        } else {
            occluded[i] = (unsigned char)scene_isHit(s, variant, &r, tmin[i], tmax[i]);
        }
    }
}
void orc_background(const orc_scene *s, long long n, const float *dir, float *rgb) {
    for (long long i = 0; i < n; i++) {
        f3 c = getBackground(s, v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
}
void orc_pcg32_reset(void) { mcg_state = 0xcafef00dd15ea5e5u; }
float orc_pcg32_rand(void) { Rng r; r.mode = 0; return rng_rand(&r); }

/* Scene::Raytrace (C:2200-2229) over rows [row0,row1): rgb is the whole
 * W*H*3 framebuffer, row-major, origin bottom-left (C:199-202).
 * mode: 0 = primary only (t as grey, debugging), 1 = shade. Returns rays traced. */
long long orc_raytrace(orc_scene *s, const Camera *cam, int variant, int row0, int row1, float *rgb) {
    Ctx c; c.s = s; c.variant = variant; c.max_bounces = -1; c.rays = 0;
    for (int j = row0; j < row1; ++j) {
        for (int i = 0; i < cam->width; ++i) {
            const Ray ray = eyeRay(cam, i, j);
            HitInfo hitInfo;
            f3 px;
            if (ctx_intersect(&c, &hitInfo, ray.o, ray.d, 0.0f, FLT_MAX)) px = shade(&c, &hitInfo, neg3(ray.d), 0, 1);
            else px = getBackground(s, neg3(ray.d));
            float *q = rgb + 3 * ((size_t)i + (size_t)j * cam->width);
            q[0] = px.x; q[1] = px.y; q[2] = px.z;
        }
    }
    return c.rays;
}
/* Scene::Pathtrace (C:2230-2261) for samples [sample0, sample0+spp) over rows
 * [row0,row1).  accum is the AccumulationBuffer (caller-zeroed before sample
 * 0), rgb the FrameBuffer: hit pixels = accum/sampleCount, miss pixels =
 * getBackground(-d) unaccumulated.  rng_mode 0 = the reference's global PCG32
 * stream in serial pixel order; 1 = counter RNG keyed on the global pixel
 * index (what the GPU path uses). Returns rays traced. */
long long orc_pathtrace(orc_scene *s, const Camera *cam, int variant, int row0, int row1, int sample0, int spp,
                        int max_bounces, int rng_mode, unsigned seed, float *accum, float *rgb) {
    Ctx c; c.s = s; c.variant = variant; c.max_bounces = max_bounces; c.rays = 0;
    for (int sm = 0; sm < spp; sm++) {
        unsigned sampleCount = (unsigned)(sample0 + sm + 1);
        for (int j = row0; j < row1; ++j) {
            for (int i = 0; i < cam->width; ++i) {
                const Ray ray = eyeRay(cam, i, j);
                HitInfo hitInfo;
                size_t p = (size_t)i + (size_t)j * cam->width;
                float *a = accum + 3 * p, *q = rgb + 3 * p;
                if (ctx_intersect(&c, &hitInfo, ray.o, ray.d, 0.0f, FLT_MAX)) {
                    Rng rng; rng.mode = rng_mode; rng.seed = seed; rng.pixel = (uint32_t)p;
                    rng.sample = (uint32_t)(sample0 + sm); rng.dim = 0;
                    f3 result = shade_pathtrace(&c, &hitInfo, neg3(ray.d), 0, 1, &rng);
                    a[0] += result.x; a[1] += result.y; a[2] += result.z;
                    q[0] = a[0] / sampleCount; q[1] = a[1] / sampleCount; q[2] = a[2] / sampleCount;
                } else {
                    f3 bg = getBackground(s, neg3(ray.d));
                    q[0] = bg.x; q[1] = bg.y; q[2] = bg.z;
                }
            }
        }
    }
    return c.rays;
}
int orc_sizeof_camera(void) { return (int)sizeof(Camera); }

/* The linalg restatement at the top of this file, op by op over plain arrays, so that tests can compare it with the
 * reference's own linalg.h (oracle/_ref/libref_linalg.so, built from the unmodified header; tests/test_ref_pins.py).
 * op: 0 dot(a,b) -> 1 | 1 cross(a,b) -> 3 | 2 normalize(a) -> 3 | 3 det of columns (a,b,c) -> 1 | 4 mul(mat(a,b,c), d) -> 3
 *     5 length2(a) -> 1 | 6 normalize(float2(a.x, a.y)) -> 2 | 7 a / b componentwise -> 3 */
int orc_vec_ops(int op, const float *a, const float *b, const float *c, const float *d, long long n, float *out) {
    for (long long i = 0; i < n; i++) {
        const f3 A = v3(a[3 * i], a[3 * i + 1], a[3 * i + 2]);
        const f3 B = b ? v3(b[3 * i], b[3 * i + 1], b[3 * i + 2]) : v3(0, 0, 0);
        const f3 Cc = c ? v3(c[3 * i], c[3 * i + 1], c[3 * i + 2]) : v3(0, 0, 0);
        const f3 D = d ? v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]) : v3(0, 0, 0);
        f3 r;
        switch (op) {
        case 0: out[i] = dot3(A, B); break;
        case 1: r = cross3(A, B); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break;
        case 2: r = normalize3(A); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break;
        case 3: out[i] = det3(A, B, Cc); break;
        case 4: r = mulcols3(A, B, Cc, D); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break;
        case 5: out[i] = dot3(A, A); break;
        case 6: { f2 q = { A.x, A.y }; q = normalize2(q); out[2 * i] = q.x; out[2 * i + 1] = q.y; break; }
        case 7: r = cdiv3(A, B); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break;
        default: return -1;
        }
    }
    return 0;
}
