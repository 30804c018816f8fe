/*
 * mipt.h -- C-ABI of the MI355X-native ray/scene path (libmipt.so).
 *
 * The reference (flamebrawler/Path-tracer) has no plugin / FFI interface: the
 * path is reached through global state in one header.  The boundary below is
 * therefore shaped after the reference's own call sites; each entry point
 * names the reference interface it replaces
 *   C: = reference cpu/src/cs488.h     G: = reference gpu/src/cs488.h
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative mipt_status; mipt_last_error() gives the message of
 * the calling thread's last failure.  Nothing aborts or throws across the
 * boundary.  The compute entry points need a gfx950 GPU and fail with
 * MIPT_ERR_DEVICE when none is usable -- there is no CPU fallback.
 *
 * Images are float RGB, row-major, origin bottom-left (Image::pixel, C:199-202).
 */
#ifndef MIPT_H
#define MIPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mipt_status {
    MIPT_OK = 0,
    MIPT_ERR_INVALID = -1, /* bad argument / malformed scene */
    MIPT_ERR_IO = -2,      /* file missing or unparsable */
    MIPT_ERR_DEVICE = -3,  /* no usable GPU, HIP call failed */
    MIPT_ERR_NOMEM = -4,
    MIPT_ERR_LIMIT = -5    /* size beyond a documented maximum */
} mipt_status;

/* enumMaterialType, C:407-412 */
enum { MIPT_MAT_LAMBERTIAN = 0, MIPT_MAT_METAL = 1, MIPT_MAT_GLASS = 2, MIPT_MAT_LIGHT = 3 };

/* Material, C:413-433 (name/glossiness dropped; texture by index) */
typedef struct mipt_material {
    int32_t type;
    float eta;
    float Ka[3], Kd[3], Ks[3];
    float Ns;
    int32_t tex_id; /* -1 = untextured */
} mipt_material;

/* Material::texture, 8-bit RGB, row 0 = top row of the file (C:975-982) */
typedef struct mipt_texture {
    const uint8_t* rgb;
    int32_t width, height;
} mipt_texture;

/* PointLightSource, C:388-391 */
typedef struct mipt_light {
    float position[3];
    float wattage[3];
} mipt_light;

/*
 * Flat scene description: what TriangleMesh + StacklessBVH + Scene hold after
 * Scene::preCalc (C:2027-2036), as caller-owned SoA arrays.  Replaces the
 * per-pointer deep copy create(&deviceScene, globalScene) (G:3181,
 * G:2291-2365, G:2442-2460, G:2667-2693).
 *
 * Node i: box node_min/node_max[3i..3i+2]; threaded links id_hit/id_miss
 * (CollNode, C:1637-1639, C:1649-1659; 0 = "done"); leaf_count[i] = 0 for an
 * interior node, else the leaf owns tri_index[leaf_first[i] ..
 * leaf_first[i]+leaf_count[i]) in BVHNode::triList order (C:1522-1524).
 * Triangle t: pos/nrm[9t..9t+8] (3 vertices x xyz), uv[6t..6t+5], mat_id[t].
 * One description is ONE object (mesh + its BVH); the reference's Scene loops over `objects` (C:2060, C:2075, addObject
 * C:2020): a scene of several objects is made from several descriptions by mipt_scene_create_multi below.
 * env_rgb: Image::pixels of Scene::background after Image::load's vertical
 * flip (C:214-219); may be NULL (env_width = 0): then a lookup returns black
 * (the reference divides by zero, C:2043).
 */
typedef struct mipt_scene_desc {
    int32_t n_nodes;
    const float* node_min;
    const float* node_max;
    const int32_t* id_hit;
    const int32_t* id_miss;
    const int32_t* leaf_first;
    const int32_t* leaf_count;
    int32_t n_tri_index;
    const int32_t* tri_index;
    int32_t n_tris;
    const float* pos;
    const float* nrm;
    const float* uv;
    const int32_t* mat_id;
    int32_t n_materials;
    const mipt_material* materials;
    int32_t n_textures;
    const mipt_texture* textures;
    int32_t env_width, env_height;
    const float* env_rgb;
    int32_t n_lights;
    const mipt_light* lights;
} mipt_scene_desc;

/* camera globals, C:65-70 + C:81-85, with the resolution made a parameter */
typedef struct mipt_camera {
    int32_t width, height;
    float eye[3];
    float lookat[3];
    float up[3];
    float fov_deg;   /* globalFOV = 45 */
    float film_size; /* globalFilmSize = 0.032 */
} mipt_camera;

enum {
    MIPT_MODE_PRIMARY = 0, /* closest-hit only: rgb = (t, tri_id, material) for debugging */
    MIPT_MODE_DIRECT = 1,  /* Scene::Raytrace + shade, C:2200-2229, C:2402-2496 */
    MIPT_MODE_PATH = 2     /* Scene::Pathtrace + shade_pathtrace, C:2230-2261, C:2267-2397 */
};

typedef struct mipt_render_params {
    int32_t mode;
    int32_t spp;           /* samples taken by this call (path mode); direct mode ignores it */
    int32_t sample_offset; /* index of the first sample: progressive rendering continues here */
    int32_t max_bounces;   /* extension rays per path; <0 = no cap as in the reference (C:2339-2389), which here means
                            * at most 64 bounce iterations: a path still alive then (closed metal / glass chains;
                            * a Lambertian path survives 58 rounds of roulette with probability 0.75^58 = 6e-8) is
                            * ended as at the cap, a bias below one part in 10^7 on shipped scenes */
    uint32_t seed;
    /* pixel partition for multi-GPU: the image is cut into tiles of
     * tile_w x tile_h pixels; this call renders tile (tx, ty) iff
     * (tx + 3*ty) % part_count == part_index (diagonal interleave, so sky and
     * geometry spread over the parts). part_count <= 1 renders all. */
    int32_t part_index, part_count;
    int32_t tile_w, tile_h; /* multiples of 8; 0 = default 64 x 8 */
} mipt_render_params;

/* per-kernel-class statistics of a context, since the last reset */
typedef struct mipt_stats {
    uint64_t rays_closest;   /* rays given to the closest-hit kernel */
    uint64_t rays_occluded;  /* rays given to the any-hit kernel */
    uint64_t launches_closest, launches_occluded, launches_other;
    double ms_closest, ms_occluded, ms_other; /* HIP-event time, only with profiling on */
    /* only with counting on: box tests (node visits) and triangle tests, per traversal kernel */
    uint64_t node_visits_closest, tri_tests_closest;
    uint64_t node_visits_occluded, tri_tests_occluded;
    /* the path-mode shading kernel (k_shade_path), a subset of launches_other / ms_other */
    uint64_t launches_shade;
    double ms_shade;
} mipt_stats;

typedef struct mipt_mesh mipt_mesh;
typedef struct mipt_image mipt_image;
typedef struct mipt_bvh mipt_bvh;
typedef struct mipt_scene mipt_scene;
typedef struct mipt_ctx mipt_ctx;

const char* mipt_last_error(void);
const char* mipt_version(void);

/* ---- host side: loaders, SAH-BVH build (CPU; stay on the host per north_star) ---- */

/* TriangleMesh::load, C:835-923 (ParseOBJ C:1035-1312, LoadMTL C:990-1033,
 * material typing C:849-868).  Unlike the reference, a missing file is an
 * error code, not a crash. */
int mipt_mesh_load_obj(const char* path, mipt_mesh** out);
/* TriangleMesh::createSingleTriangle, C:949-970 */
int mipt_mesh_create_single_triangle(mipt_mesh** out);
/* a mesh from caller arrays (copied): pos/nrm 9 floats, uv 6 floats per triangle */
int mipt_mesh_create(int32_t n_tris, const float* pos, const float* nrm, const float* uv, const int32_t* mat_id,
                     int32_t n_materials, const mipt_material* materials, int32_t n_textures,
                     const mipt_texture* textures, mipt_mesh** out);
int32_t mipt_mesh_num_triangles(const mipt_mesh*);
int32_t mipt_mesh_num_materials(const mipt_mesh*);
int32_t mipt_mesh_num_textures(const mipt_mesh*);
/* pointers into the mesh, valid until mipt_mesh_destroy */
const float* mipt_mesh_positions(const mipt_mesh*);
const float* mipt_mesh_normals(const mipt_mesh*);
const float* mipt_mesh_texcoords(const mipt_mesh*);
const int32_t* mipt_mesh_material_ids(const mipt_mesh*);
const mipt_material* mipt_mesh_materials(const mipt_mesh*);
const mipt_texture* mipt_mesh_textures(const mipt_mesh*);
void mipt_mesh_destroy(mipt_mesh*);

/* Image::load, C:204-222 (.hdr Radiance RGBE or .png; rows flipped so j = 0 is the bottom row) */
int mipt_image_load(const char* path, mipt_image** out);
int mipt_image_create(int32_t width, int32_t height, const float* rgb, mipt_image** out);
int32_t mipt_image_width(const mipt_image*);
int32_t mipt_image_height(const mipt_image*);
const float* mipt_image_pixels(const mipt_image*);
/* Image::save, C:223-236: clamp to [0,1], gamma 1.0, 8-bit PNG, top row first */
int mipt_image_save_png(const char* path, int32_t width, int32_t height, const float* rgb);
/* float output for RMSE checks (not in the reference): Radiance .hdr and .pfm */
int mipt_image_save_hdr(const char* path, int32_t width, int32_t height, const float* rgb);
int mipt_image_save_pfm(const char* path, int32_t width, int32_t height, const float* rgb);
/* One channel of Image::save (C:223-236): (unsigned char)(255.0f * gammaCorrection(toneMapping(v))), gamma 1.0,
 * for n_values floats in the order given (no flip); mipt_image_save_png uses the same conversion. */
int mipt_image_pack_rgb8(const float* rgb, size_t n_values, uint8_t* out);
/* RMSE over all channels and the two means, accumulated in double (convergence curves of config 5,
 * image parity tolerances); any output may be NULL.  No counterpart in the reference. */
int mipt_image_compare(const float* a, const float* b, size_t n_values, double* rmse, double* mean_a, double* mean_b);
void mipt_image_destroy(mipt_image*);

/* TriangleMesh::preCalc C:817-831 + BVH::build C:1562-1595 (SAH split
 * C:1412-1558) + StacklessBVH::build links C:1645-1660 + StacklessBVH2::build
 * links C:1713-1730.  Node numbering and leaf order equal the reference's. */
int mipt_bvh_build(const mipt_mesh* mesh, mipt_bvh** out);
/* SURVEY 8(f).1, NOT the reference's tree: the same SAH sweep and cost (C:1406-1409, C:1488-1504) along each of the
 * three axes instead of the largest one only (C:1463), the cheapest split of all wins.  Fewer boxes per ray, but
 * another node numbering and other leaf boxes: hit distances equal the reference's except where a triangle's own
 * leaf box decides (floating-point edge cases), ids may differ on exact ties.  Everything downstream (links, scene,
 * traversals) takes it like the reference's tree. */
int mipt_bvh_build_all_axes(const mipt_mesh* mesh, mipt_bvh** out);
int32_t mipt_bvh_num_nodes(const mipt_bvh*);
int32_t mipt_bvh_num_leaves(const mipt_bvh*);
int32_t mipt_bvh_depth(const mipt_bvh*);
int32_t mipt_bvh_max_leaf_size(const mipt_bvh*);
const float* mipt_bvh_node_min(const mipt_bvh*);
const float* mipt_bvh_node_max(const mipt_bvh*);
const int32_t* mipt_bvh_id_hit(const mipt_bvh*);
const int32_t* mipt_bvh_id_miss(const mipt_bvh*);
const int32_t* mipt_bvh_leaf_first(const mipt_bvh*);
const int32_t* mipt_bvh_leaf_count(const mipt_bvh*);
const int32_t* mipt_bvh_id_left(const mipt_bvh*);
const int32_t* mipt_bvh_id_right(const mipt_bvh*);
const int32_t* mipt_bvh_id_parent(const mipt_bvh*);
const int32_t* mipt_bvh_id_sibling(const mipt_bvh*);
const int32_t* mipt_bvh_tri_index(const mipt_bvh*);
void mipt_bvh_destroy(mipt_bvh*);

/* Flat binary cache of a loaded mesh and its built BVH.  The reference has no counterpart: it
 * re-parses the .obj (TriangleMesh::load C:835-923) and re-runs the SAH build (BVH::build C:1562-1595) at every
 * start.  save writes one file (magic, endianness probe, FNV-1a checksum, counted arrays); load returns a mesh and
 * a bvh equal, array for array, to the ones that were saved, or MIPT_ERR_IO if the file is missing, truncated,
 * corrupted or from an incompatible machine.  Array CONTENT is validated where it is used (mipt_scene_create). */
int mipt_scene_cache_save(const char* path, const mipt_mesh* mesh, const mipt_bvh* bvh);
int mipt_scene_cache_load(const char* path, mipt_mesh** mesh_out, mipt_bvh** bvh_out);
/* The same with a source tag: free text the writer chooses to say what the cache was built from (mipt_render stores
 * the fingerprints of the .obj / .mtl files and the BVH mode); a load that passes expect_tag (not NULL) returns
 * MIPT_ERR_IO for a cache written with another tag, so a stale cache, or one of another scene or tree, is rebuilt
 * instead of rendered.  mipt_scene_cache_save = tag "", mipt_scene_cache_load = any tag. */
int mipt_scene_cache_save_tagged(const char* path, const mipt_mesh* mesh, const mipt_bvh* bvh, const char* tag);
int mipt_scene_cache_load_tagged(const char* path, const char* expect_tag, mipt_mesh** mesh_out, mipt_bvh** bvh_out);
/* "name:size:fnv1a64" of a file's content ("name:absent" if unreadable), NUL-terminated into out */
int mipt_file_fingerprint(const char* path, char* out, size_t out_size);

/* Scene::addObject / addLight / background + preCalc, C:2020-2036: fills a
 * mipt_scene_desc that points INTO mesh, bvh and env (keep them alive while
 * the desc is used).  env may be NULL. */
int mipt_scene_desc_fill(const mipt_mesh* mesh, const mipt_bvh* bvh, const mipt_image* env, const mipt_light* lights,
                         int32_t n_lights, mipt_scene_desc* out);

/* ---- device side ---- */

/* Validates and COPIES the description (caller may free its arrays at once). */
int mipt_scene_create(const mipt_scene_desc* desc, mipt_scene** out);
/* mipt_scene_create with the library's own hierarchy above the leaves kept in a file (no counterpart in the reference,
 * whose start pays BVH::build every time, cs488.h:1562-1595, and whose CUDA program copies the tree node by node,
 * gpu/src/cs488.h:2442-2460): when wide_cache_path holds the hierarchy built from exactly these nodes, leaf order, triangle
 * positions, lights and build switches (a 64-bit key over them, layout constants and a checksum are checked), it is
 * loaded -- the same arrays bit for bit -- else it is built (on the host's cores) and written there.  An unreadable,
 * foreign or damaged file is ignored and overwritten; a file that cannot be written is not an error.  Every rank of a
 * multi-GPU run and every start of mipt_render --cache uses this.  NULL or "": as mipt_scene_create. */
int mipt_scene_create_cached(const mipt_scene_desc* desc, const char* wide_cache_path, mipt_scene** out);
/* Scene::addObject (C:2020) n_objects times + Scene::preCalc's one BVH per object (C:2027-2035): a scene of several
 * objects, each with its own mesh, materials, textures and BVH (objects[k] as mipt_scene_desc_fill made it; environment and
 * lights are objects[0]'s).  Hits are those of Scene::intersect's loop over the objects (C:2070-2085): the smallest t, an
 * exact tie going to the EARLIER object and, within one object, to the triangle its own walk meets first; Scene::isHit
 * (C:2056-2068) is "some object is hit".  Triangle ids (mipt_trace) and material ids (hitrec) count through the objects in
 * order: object k's triangle i is mipt_scene_object_tri_base(scene, k) + i.  wide_cache_path: as mipt_scene_create_cached,
 * or NULL.  The bit-trail traversal (MIPT_TRAVERSAL_BITTRAIL) walks the objects one after the other as the reference
 * does; with more than 64 objects the few rays the default traversal hands back take the literal walk, a lane per ray. */
int mipt_scene_create_multi(const mipt_scene_desc* objects, int32_t n_objects, const char* wide_cache_path, mipt_scene** out);
/* objects of a scene (1 for mipt_scene_create) and the first scene-wide triangle / material id of object k (-1: no such object) */
int32_t mipt_scene_object_count(const mipt_scene*);
int32_t mipt_scene_object_tri_base(const mipt_scene*, int32_t object);
int32_t mipt_scene_object_material_base(const mipt_scene*, int32_t object);
void mipt_scene_destroy(mipt_scene*);
/* 1 when every box plane of the scene is 0 or of magnitude in [2^-61, 2^40] (and min <= max), so the traversal
 * kernel may replace the six IEEE divisions of AABB::intersect (C:574-580) by its exact per-ray reciprocal
 * sequence; 0: every ray of this scene takes the literal IEEE-division slab test.  Results are bit-identical
 * either way; only the speed differs.  No counterpart in the reference. */
int mipt_scene_exact_reciprocal(const mipt_scene*);
/* Number of nodes of the wide tree (eight slots per node) the library built over the scene's LEAVES for its
 * order-free traversal kernel, or 0 when the scene does not qualify (box planes outside the range above, boxes not
 * nested child-inside-parent, not a binary tree) and every ray takes the literal threaded walk of
 * StacklessBVH::intersect (C:1661-1683).  The depth of the given tree does not matter.  Results are bit-identical
 * either way.  No counterpart in the reference. */
int mipt_scene_wide_nodes(const mipt_scene*);
/* 64-bit FNV-1a over the arrays of that tree as the GPU will read them (nodes, exact leaf boxes, leaf heads, scale): two
 * scenes with the same value walk the same hierarchy -- what a test of mipt_scene_create_cached, or of a build on
 * another number of threads, compares.  No counterpart in the reference. */
uint64_t mipt_scene_wide_hash(const mipt_scene*);
/* Host-side check (no GPU) of what the order-free traversal rests on: the 4-wide tree -- a hierarchy of the library's
 * own over the reference's LEAVES, see DESIGN.md 4 -- reaches every leaf exactly once, and every slot's stored box
 * contains everything stored below it, down to the leaf's exact box.  MIPT_OK (with the number of leaves and the
 * depth of the wide tree, 0 and 0 for a scene on the literal walk) or MIPT_ERR_INVALID with the reason in
 * mipt_last_error().  No counterpart in the reference. */
int mipt_scene_selfcheck_wide(const mipt_scene*, int* leaves_out, int* depth_out);

/* One context per GPU: uploads the scene as ONE hipMalloc'd arena of flat
 * SoA sections (replaces create(&deviceScene, globalScene), G:3181).  A
 * context is not thread-safe; different contexts may be driven from
 * different threads or processes.  stream: a hipStream_t, or NULL for a
 * stream the context creates. */
int mipt_context_create(const mipt_scene* scene, int device, void* stream, mipt_ctx** out);
void mipt_context_destroy(mipt_ctx*);
size_t mipt_context_arena_bytes(const mipt_ctx*);

/* Scene::intersect, C:2070-2085, over a batch of caller-supplied rays with
 * the threaded traversal StacklessBVH::intersect, C:1661-1683, and
 * raytraceTriangle, C:777-813.  HOST pointers: ray_o/ray_d 3 floats per ray.
 * Outputs (any may be NULL): tri_id (-1 = miss), t (FLT_MAX on miss),
 * bary = (beta, gamma), hitrec = HitInfo as 10 floats {t, P3, N3, T2, material}
 * (C:510-517; material = index, -1 on miss). */
int mipt_trace(mipt_ctx* ctx, const float* ray_o, const float* ray_d, size_t n, float tmin, float tmax,
               int32_t* tri_id, float* t, float* bary, float* hitrec);
/* the same with DEVICE pointers on the context's stream; no synchronisation.
 * rays: n x {o.xyz, tmin, d.xyz, tmax} (32 B); hits: n x {tri_id, t, beta, gamma} (16 B). */
int mipt_trace_device(mipt_ctx* ctx, const void* rays_dev, size_t n, void* hits_dev);
/* mipt_trace_device for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  Same kernels and bit-identical results, but the launch
 * cursors and the list of rays the wide traversal hands back live in a file-scope __device__ array of the library
 * (4 MiB + 256 B per GPU, held until the process ends) instead of memory the context allocates: this call performs
 * no hipMalloc / hipFree / event creation.  At most MIPT_NOALLOC_MAX_RAYS rays per call (MIPT_ERR_LIMIT beyond).
 * Every call on a GPU shares that GPU's array: calls that overlap in time on one device (two contexts, two streams)
 * race; calls on one stream are ordered.  HIP-event profiling (mipt_context_set_profiling) does not cover it. */
#define MIPT_NOALLOC_MAX_RAYS 1048576
int mipt_trace_device_noalloc(mipt_ctx* ctx, const void* rays_dev, size_t n, void* hits_dev);
/* StacklessBVH::isHit, C:1684-1705 (any-hit), as Scene::isHit uses it for
 * shadow rays, C:2056-2068.  HOST pointers; per-ray tmin/tmax. */
int mipt_occluded(mipt_ctx* ctx, const float* ray_o, const float* ray_d, const float* tmin, const float* tmax,
                  size_t n, uint8_t* occluded);
/* Scene::eyeRay, C:2183-2197, for pixels [first, first+count) in row-major
 * (j, i) order; HOST outputs. Computed on the GPU by the generate kernel. */
int mipt_eye_rays(mipt_ctx* ctx, const mipt_camera* cam, size_t first, size_t count, float* ray_o, float* ray_d);
/* Scene::getBackground, C:2038-2053; HOST pointers. */
int mipt_background(mipt_ctx* ctx, const float* dir, size_t n, float* rgb);

/* Scene::Raytrace / Scene::Pathtrace (C:2200-2261) for this call's pixel
 * partition.  accum_dev / rgb_dev are DEVICE buffers of width*height*3
 * floats (AccumulationBuffer / FrameBuffer, C:240-243); pixels outside the
 * partition are not touched.  accum_dev must be zero before sample 0 and is
 * carried by the caller between calls (progressive rendering); it may be
 * NULL in primary/direct mode.  Asynchronous on the context's stream. */
int mipt_render_device(mipt_ctx* ctx, const mipt_camera* cam, const mipt_render_params* params, float* accum_dev,
                       float* rgb_dev);
/* convenience: whole image to a HOST buffer (allocates, renders, copies back, synchronises) */
int mipt_render(mipt_ctx* ctx, const mipt_camera* cam, const mipt_render_params* params, float* rgb_host);

/* ---- one frame over the GPUs of a node (north_star subsystem 4) ----
 * Replaces the reference's per-frame launch + readback (G:3209-3216) when more than one GPU renders the frame.  A
 * node holds one context, one HIP stream and one host thread per GPU, the scene arena replicated.  A frame is cut
 * into tiles (mipt_render_params.tile_w/h), GPU r renders the tiles with (tx + 3*ty) % ngpus == r, and ONE RCCL
 * collective per call (ncclGather over xGMI) moves the owned tiles to GPU 0, which writes the frame.  The result is
 * bit-identical to the one-GPU frame (the per-pixel RNG is keyed on the global pixel index).
 * devices: ngpus device ids, or NULL for 0 .. ngpus-1.  Repeating an id makes ranks share a GPU (rehearsal on a
 * machine with fewer GPUs): those ranks cannot form an RCCL communicator, so the gather is device-to-device copies;
 * mipt_node_uses_rccl() tells which transport a node got.  RCCL is loaded (dlopen) when the first node is made.
 * A node is driven from one host thread at a time. */
typedef struct mipt_node mipt_node;
int mipt_node_create(const mipt_scene* scene, int ngpus, const int* devices, mipt_node** out);
void mipt_node_destroy(mipt_node*);
int mipt_node_num_gpus(const mipt_node*);
int mipt_node_uses_rccl(const mipt_node*);
/* Scene::Raytrace / Scene::Pathtrace (C:2200-2261) of the whole frame; params->part_index / part_count are ignored.
 * Path mode is progressive across calls as in mipt_render_device: a call with sample_offset > 0 continues the
 * accumulation of the previous calls (same camera and resolution), sample_offset == 0 starts a new frame.
 * rgb_host: width*height*3 floats on the HOST: the call returns when the frame is complete and copied.  NULL: the frame
 * stays on GPU 0 and the call returns with its work ENQUEUED (round 3): the gather of this call runs on a transfer stream
 * per GPU beside the kernels of the next call (two send slots used in turn), which is what a progressive sequence --
 * many calls, the frame fetched by the last one -- wants; mipt_node_synchronize waits for everything enqueued.
 * The gather moves exactly the owned tiles, point to point (grouped ncclSend / ncclRecv; rank 0's own tiles stay where
 * they are), or, with MIPT_NODE_GATHER=gather, one ncclGather over equal slots. */
int mipt_node_render(mipt_node*, const mipt_camera* cam, const mipt_render_params* params, float* rgb_host);
/* waits until every GPU of the node has finished what mipt_node_render enqueued */
int mipt_node_synchronize(mipt_node*);
/* sum of the per-GPU statistics (mipt_context_get_stats); last_frame_ms (may be NULL): host wall time of the last
 * mipt_node_render call */
int mipt_node_get_stats(mipt_node*, mipt_stats* out, double* last_frame_ms);
/* convenience: create a node over GPUs 0 .. ngpus-1, render one frame to the host, destroy the node */
int mipt_render_node(const mipt_scene* scene, const mipt_camera* cam, const mipt_render_params* params, int ngpus,
                     float* rgb_out_rank0);

/* Device buffers for callers without a GPU allocator of their own (the
 * frame readback of the reference, G:3213-3216, becomes mipt_buffer_download).
 * All on the context's device; upload/download/zero are ordered on its stream
 * and download synchronises. */
/* mipt_image_pack_rgb8 on the device, asynchronous on the context's stream: the frame leaves the GPU (readback,
 * or the tile gather to rank 0) as 3 bytes per pixel instead of 12.  Same bytes as the host conversion.
 * rgb_dev 16-byte aligned, out_dev 4-byte aligned (any hipMalloc / torch allocation is). */
int mipt_pack_rgb8_device(mipt_ctx* ctx, const float* rgb_dev, size_t n_values, uint8_t* out_dev);
int mipt_buffer_alloc(mipt_ctx* ctx, size_t bytes, void** dev_ptr);
int mipt_buffer_free(mipt_ctx* ctx, void* dev_ptr);
int mipt_buffer_zero(mipt_ctx* ctx, void* dev_ptr, size_t bytes);
int mipt_buffer_upload(mipt_ctx* ctx, void* dev_ptr, const void* host_ptr, size_t bytes);
int mipt_buffer_download(mipt_ctx* ctx, void* host_ptr, const void* dev_ptr, size_t bytes);

/* Self-test (no counterpart in the reference): the traversal kernel replaces the six IEEE divisions of
 * AABB::intersect (C:574-580) by a per-ray reciprocal plus residual correction that must round exactly like
 * `/`; this runs that sequence against `/` on n_pairs operand pairs on the GPU and reports the mismatches (0). */
int mipt_selftest_fast_division(mipt_ctx* ctx, uint64_t n_pairs, uint32_t seed, uint64_t* mismatches);

/* Self-test (no counterpart in the reference): the library's own vector primitives (csrc/pt/vec.hpp, shared by host and
 * device code) over arrays of float3, so that tests can compare them bit for bit with the reference's linalg.h
 * (linalg.h:367, :401, :500-504, :536, :567 -- SURVEY 8(a) row 3; tests/test_ref_pins.py holds vectors produced by the
 * unmodified header).  ctx == NULL: the host build; otherwise the device build, on that context's GPU.
 * op: 0 dot(a,b) -> n | 1 cross(a,b) -> 3n | 2 normalize(a) -> 3n | 3 determinant of the matrix with COLUMNS a,b,c -> n |
 *     4 mul(mat(a,b,c), d) -> 3n | 5 length2(a) -> n | 6 normalize(float2(a.x, a.y)) -> 2n | 7 a / b componentwise -> 3n.
 * a..d: n float3 each (those the op does not use may be NULL); out: host memory. */
int mipt_selftest_vec_ops(mipt_ctx* ctx, int op, const float* a, const float* b, const float* c, const float* d, size_t n,
                          float* out);

/* Which of the reference's traversals answers mipt_trace / mipt_occluded / the render loops of this context.
 * All three return the same bits on every ray tested; they differ in the work they do.
 *   DEFAULT   StacklessBVH::intersect / isHit (C:1661-1705) semantics through the order-free 4-wide traversal
 *             (k_trace_wide), with the literal walk for the rays and scenes that do not qualify;
 *   THREADED  the literal idHit / idMiss walk for every ray (k_trace) -- what MIPT_TRAVERSAL=walk selects;
 *   BITTRAIL  StacklessBVH2::intersect (C:1732-1803), the traversal Scene instantiates (C:2008), literal
 *             (k_trace_bittrail): near-first, culls on box.t < minHit.t, 32-bit trail; its box / triangle test
 *             counts are the reference's.  MIPT_ERR_LIMIT for trees deeper than 31 levels (the trail's width). */
enum { MIPT_TRAVERSAL_DEFAULT = 0, MIPT_TRAVERSAL_THREADED = 1, MIPT_TRAVERSAL_BITTRAIL = 2 };
int mipt_context_set_traversal(mipt_ctx* ctx, int traversal);

int mipt_context_synchronize(mipt_ctx* ctx);
/* Path mode runs the shadow rays of bounce k (any-hit) on a side stream of the context, beside the extension rays of
 * bounce k+1 (closest hit): the two are independent once the vertices of bounce k are shaded, and the later kernel's
 * blocks fill the CUs the earlier one's tail leaves idle.  On by default; 0 serialises them on the context's stream
 * (per-kernel HIP-event and rocprof durations are only meaningful then).  Frames are bit-identical either way.
 * MIPT_OVERLAP=0 in the environment sets the default off.  No counterpart in the reference. */
int mipt_context_set_overlap(mipt_ctx* ctx, int enabled);
/* profiling: HIP events around every kernel launch on the context's stream */
int mipt_context_set_profiling(mipt_ctx* ctx, int enabled);
/* counting: the traversal kernels also count node visits and triangle tests */
int mipt_context_set_counting(mipt_ctx* ctx, int enabled);
int mipt_context_get_stats(mipt_ctx* ctx, mipt_stats* out);
int mipt_context_reset_stats(mipt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MIPT_H */
