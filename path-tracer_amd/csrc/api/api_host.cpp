// api_host.cpp -- the host half of the C-ABI declared in include/mipt.h:
// loaders, images, SAH-BVH build, scene description.  No GPU involved.
#include <cstring>
#include <memory>
#include <new>

#include "../pt/host.hpp"

namespace pt {
static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
const char* get_error() { return g_error.c_str(); }
}  // namespace pt

struct mipt_mesh {
    pt::TriangleMesh mesh;
};
struct mipt_image {
    pt::Image image;
};
struct mipt_bvh {
    pt::StacklessBVH bvh;
};

#define MIPT_GUARD_BEGIN try {
#define MIPT_GUARD_END                                   \
    }                                                    \
    catch (const std::bad_alloc&) {                      \
        pt::set_error("out of host memory");             \
        return MIPT_ERR_NOMEM;                           \
    }                                                    \
    catch (const std::exception& e) {                    \
        pt::set_error(std::string("internal: ") + e.what()); \
        return MIPT_ERR_INVALID;                         \
    }

extern "C" {

const char* mipt_last_error(void) { return pt::get_error(); }
const char* mipt_version(void) { return "mipt 0.1 (gfx950)"; }

int mipt_mesh_load_obj(const char* path, mipt_mesh** out) {
    if (!path || !out) { pt::set_error("mipt_mesh_load_obj: null argument"); return MIPT_ERR_INVALID; }
    MIPT_GUARD_BEGIN
    mipt_mesh* m = new mipt_mesh();
    int rc = m->mesh.load(path);
    if (rc != MIPT_OK) { delete m; return rc; }
    *out = m;
    return MIPT_OK;
    MIPT_GUARD_END
}

int mipt_mesh_create_single_triangle(mipt_mesh** out) {
    if (!out) { pt::set_error("null argument"); return MIPT_ERR_INVALID; }
    MIPT_GUARD_BEGIN
    mipt_mesh* m = new mipt_mesh();
    m->mesh.createSingleTriangle();
    *out = m;
    return MIPT_OK;
    MIPT_GUARD_END
}

int mipt_mesh_create(int32_t n_tris, const float* pos, const float* nrm, const float* uv, const int32_t* mat_id,
                     int32_t n_materials, const mipt_material* materials, int32_t n_textures,
                     const mipt_texture* textures, mipt_mesh** out) {
    if (!out || n_tris <= 0 || !pos || !nrm) { pt::set_error("mipt_mesh_create: bad arguments"); return MIPT_ERR_INVALID; }
    if (n_materials < 0 || n_textures < 0 || (n_materials && !materials) || (n_textures && !textures)) {
        pt::set_error("mipt_mesh_create: bad material/texture arguments");
        return MIPT_ERR_INVALID;
    }
    MIPT_GUARD_BEGIN
    mipt_mesh* m = new mipt_mesh();
    pt::TriangleMesh& tm = m->mesh;
    tm.positions.assign(pos, pos + (size_t)9 * n_tris);
    tm.normals.assign(nrm, nrm + (size_t)9 * n_tris);
    if (uv) tm.texcoords.assign(uv, uv + (size_t)6 * n_tris);
    else tm.texcoords.assign((size_t)6 * n_tris, 0.0f);
    if (mat_id) tm.idMaterial.assign(mat_id, mat_id + n_tris);
    else tm.idMaterial.assign((size_t)n_tris, 0);
    for (int i = 0; i < n_textures; i++) {
        pt::Texture t;
        if (!textures[i].rgb || textures[i].width <= 0 || textures[i].height <= 0) {
            delete m;
            pt::set_error("mipt_mesh_create: empty texture");
            return MIPT_ERR_INVALID;
        }
        t.width = textures[i].width;
        t.height = textures[i].height;
        t.rgb.assign(textures[i].rgb, textures[i].rgb + (size_t)3 * t.width * t.height);
        tm.textures.push_back(std::move(t));
    }
    if (n_materials == 0) tm.materials.assign(1, pt::Material());
    for (int i = 0; i < n_materials; i++) {
        pt::Material mm;
        const mipt_material& c = materials[i];
        mm.type = c.type;
        mm.eta = c.eta;
        mm.Ka = pt::vec3{c.Ka[0], c.Ka[1], c.Ka[2]};
        mm.Kd = pt::vec3{c.Kd[0], c.Kd[1], c.Kd[2]};
        mm.Ks = pt::vec3{c.Ks[0], c.Ks[1], c.Ks[2]};
        mm.Ns = c.Ns;
        if (c.tex_id >= n_textures) {
            delete m;
            pt::set_error("mipt_mesh_create: tex_id out of range");
            return MIPT_ERR_INVALID;
        }
        mm.isTextured = c.tex_id >= 0;
        mm.texId = c.tex_id;
        tm.materials.push_back(mm);
    }
    for (int i = 0; i < n_tris; i++) {
        if (tm.idMaterial[i] < 0 || tm.idMaterial[i] >= (int)tm.materials.size()) {
            delete m;
            pt::set_error("mipt_mesh_create: mat_id out of range");
            return MIPT_ERR_INVALID;
        }
    }
    tm.refreshCViews();
    *out = m;
    return MIPT_OK;
    MIPT_GUARD_END
}

int32_t mipt_mesh_num_triangles(const mipt_mesh* m) { return m ? m->mesh.numTriangles() : 0; }
int32_t mipt_mesh_num_materials(const mipt_mesh* m) { return m ? (int32_t)m->mesh.materials.size() : 0; }
int32_t mipt_mesh_num_textures(const mipt_mesh* m) { return m ? (int32_t)m->mesh.textures.size() : 0; }
const float* mipt_mesh_positions(const mipt_mesh* m) { return m ? m->mesh.positions.data() : nullptr; }
const float* mipt_mesh_normals(const mipt_mesh* m) { return m ? m->mesh.normals.data() : nullptr; }
const float* mipt_mesh_texcoords(const mipt_mesh* m) { return m ? m->mesh.texcoords.data() : nullptr; }
const int32_t* mipt_mesh_material_ids(const mipt_mesh* m) { return m ? m->mesh.idMaterial.data() : nullptr; }
const mipt_material* mipt_mesh_materials(const mipt_mesh* m) { return m ? m->mesh.cMaterials.data() : nullptr; }
const mipt_texture* mipt_mesh_textures(const mipt_mesh* m) { return m ? m->mesh.cTextures.data() : nullptr; }
void mipt_mesh_destroy(mipt_mesh* m) { delete m; }

int mipt_image_load(const char* path, mipt_image** out) {
    if (!path || !out) { pt::set_error("mipt_image_load: null argument"); return MIPT_ERR_INVALID; }
    MIPT_GUARD_BEGIN
    mipt_image* im = new mipt_image();
    int rc = im->image.load(path);
    if (rc != MIPT_OK) { delete im; return rc; }
    *out = im;
    return MIPT_OK;
    MIPT_GUARD_END
}
int mipt_image_create(int32_t width, int32_t height, const float* rgb, mipt_image** out) {
    if (!out || width <= 0 || height <= 0 || !rgb) { pt::set_error("mipt_image_create: bad arguments"); return MIPT_ERR_INVALID; }
    MIPT_GUARD_BEGIN
    mipt_image* im = new mipt_image();
    im->image.width = width;
    im->image.height = height;
    im->image.pixels.assign(rgb, rgb + (size_t)3 * width * height);
    *out = im;
    return MIPT_OK;
    MIPT_GUARD_END
}
int32_t mipt_image_width(const mipt_image* im) { return im ? im->image.width : 0; }
int32_t mipt_image_height(const mipt_image* im) { return im ? im->image.height : 0; }
const float* mipt_image_pixels(const mipt_image* im) { return im ? im->image.pixels.data() : nullptr; }
int mipt_image_save_png(const char* path, int32_t w, int32_t h, const float* rgb) {
    if (!path) { pt::set_error("null path"); return MIPT_ERR_INVALID; }
    MIPT_GUARD_BEGIN
    return pt::savePNG(path, w, h, rgb);
    MIPT_GUARD_END
}
int mipt_image_save_hdr(const char* path, int32_t w, int32_t h, const float* rgb) {
    if (!path) { pt::set_error("null path"); return MIPT_ERR_INVALID; }
    MIPT_GUARD_BEGIN
    return pt::saveHDR(path, w, h, rgb);
    MIPT_GUARD_END
}
int mipt_image_save_pfm(const char* path, int32_t w, int32_t h, const float* rgb) {
    if (!path) { pt::set_error("null path"); return MIPT_ERR_INVALID; }
    MIPT_GUARD_BEGIN
    return pt::savePFM(path, w, h, rgb);
    MIPT_GUARD_END
}
int mipt_image_pack_rgb8(const float* rgb, size_t n_values, uint8_t* out) {
    if (n_values && (!rgb || !out)) { pt::set_error("mipt_image_pack_rgb8: null argument"); return MIPT_ERR_INVALID; }
    pt::packRGB8(rgb, n_values, out);
    return MIPT_OK;
}
int mipt_image_compare(const float* a, const float* b, size_t n_values, double* rmse, double* mean_a, double* mean_b) {
    if (n_values && (!a || !b)) { pt::set_error("mipt_image_compare: null argument"); return MIPT_ERR_INVALID; }
    pt::compareImages(a, b, n_values, rmse, mean_a, mean_b);
    return MIPT_OK;
}
void mipt_image_destroy(mipt_image* im) { delete im; }

int mipt_bvh_build(const mipt_mesh* mesh, mipt_bvh** out) {
    if (!mesh || !out) { pt::set_error("mipt_bvh_build: null argument"); return MIPT_ERR_INVALID; }
    MIPT_GUARD_BEGIN
    mipt_bvh* b = new mipt_bvh();
    int rc = b->bvh.build(&mesh->mesh);
    if (rc != MIPT_OK) { delete b; return rc; }
    *out = b;
    return MIPT_OK;
    MIPT_GUARD_END
}
int mipt_bvh_build_all_axes(const mipt_mesh* mesh, mipt_bvh** out) {
    if (!mesh || !out) { pt::set_error("mipt_bvh_build_all_axes: null argument"); return MIPT_ERR_INVALID; }
    MIPT_GUARD_BEGIN
    mipt_bvh* b = new mipt_bvh();
    int rc = b->bvh.build(&mesh->mesh, true);
    if (rc != MIPT_OK) { delete b; return rc; }
    *out = b;
    return MIPT_OK;
    MIPT_GUARD_END
}
int32_t mipt_bvh_num_nodes(const mipt_bvh* b) { return b ? b->bvh.nodeNum : 0; }
int32_t mipt_bvh_num_leaves(const mipt_bvh* b) { return b ? b->bvh.leafNum : 0; }
int32_t mipt_bvh_depth(const mipt_bvh* b) { return b ? b->bvh.depth : 0; }
int32_t mipt_bvh_max_leaf_size(const mipt_bvh* b) { return b ? b->bvh.maxLeaf : 0; }
const float* mipt_bvh_node_min(const mipt_bvh* b) { return b ? b->bvh.nodeMin.data() : nullptr; }
const float* mipt_bvh_node_max(const mipt_bvh* b) { return b ? b->bvh.nodeMax.data() : nullptr; }
const int32_t* mipt_bvh_id_hit(const mipt_bvh* b) { return b ? b->bvh.idHit.data() : nullptr; }
const int32_t* mipt_bvh_id_miss(const mipt_bvh* b) { return b ? b->bvh.idMiss.data() : nullptr; }
const int32_t* mipt_bvh_leaf_first(const mipt_bvh* b) { return b ? b->bvh.leafFirst.data() : nullptr; }
const int32_t* mipt_bvh_leaf_count(const mipt_bvh* b) { return b ? b->bvh.leafCount.data() : nullptr; }
const int32_t* mipt_bvh_id_left(const mipt_bvh* b) { return b ? b->bvh.idLeft.data() : nullptr; }
const int32_t* mipt_bvh_id_right(const mipt_bvh* b) { return b ? b->bvh.idRight.data() : nullptr; }
const int32_t* mipt_bvh_id_parent(const mipt_bvh* b) { return b ? b->bvh.idParent.data() : nullptr; }
const int32_t* mipt_bvh_id_sibling(const mipt_bvh* b) { return b ? b->bvh.idSibling.data() : nullptr; }
const int32_t* mipt_bvh_tri_index(const mipt_bvh* b) { return b ? b->bvh.triIndex.data() : nullptr; }
void mipt_bvh_destroy(mipt_bvh* b) { delete b; }

int mipt_scene_cache_save_tagged(const char* path, const mipt_mesh* mesh, const mipt_bvh* bvh, const char* tag) {
    if (!path || !mesh || !bvh) { pt::set_error("mipt_scene_cache_save: null argument"); return MIPT_ERR_INVALID; }
    MIPT_GUARD_BEGIN
    return pt::saveSceneCache(path, mesh->mesh, bvh->bvh, tag ? tag : "");
    MIPT_GUARD_END
}
int mipt_scene_cache_save(const char* path, const mipt_mesh* mesh, const mipt_bvh* bvh) {
    return mipt_scene_cache_save_tagged(path, mesh, bvh, "");
}
int mipt_scene_cache_load_tagged(const char* path, const char* expect_tag, mipt_mesh** mesh_out, mipt_bvh** bvh_out) {
    if (!path || !mesh_out || !bvh_out) { pt::set_error("mipt_scene_cache_load: null argument"); return MIPT_ERR_INVALID; }
    MIPT_GUARD_BEGIN
    std::unique_ptr<mipt_mesh> m(new mipt_mesh());
    std::unique_ptr<mipt_bvh> b(new mipt_bvh());
    const std::string want = expect_tag ? expect_tag : "";
    const int rc = pt::loadSceneCache(path, m->mesh, b->bvh, expect_tag ? &want : nullptr);
    if (rc != MIPT_OK) return rc;
    *mesh_out = m.release();
    *bvh_out = b.release();
    return MIPT_OK;
    MIPT_GUARD_END
}
int mipt_scene_cache_load(const char* path, mipt_mesh** mesh_out, mipt_bvh** bvh_out) {
    return mipt_scene_cache_load_tagged(path, nullptr, mesh_out, bvh_out);
}
/* "name:size:fnv1a64" of a file's content into out (NUL-terminated, truncated to out_size) */
int mipt_file_fingerprint(const char* path, char* out, size_t out_size) {
    if (!path || !out || out_size == 0) { pt::set_error("mipt_file_fingerprint: null argument"); return MIPT_ERR_INVALID; }
    MIPT_GUARD_BEGIN
    const std::string fp = pt::fileFingerprint(path);
    snprintf(out, out_size, "%s", fp.c_str());
    return MIPT_OK;
    MIPT_GUARD_END
}

int mipt_scene_desc_fill(const mipt_mesh* mesh, const mipt_bvh* bvh, const mipt_image* env, const mipt_light* lights,
                         int32_t n_lights, mipt_scene_desc* out) {
    if (!mesh || !bvh || !out || n_lights < 0 || (n_lights && !lights)) {
        pt::set_error("mipt_scene_desc_fill: bad arguments");
        return MIPT_ERR_INVALID;
    }
    memset(out, 0, sizeof(*out));
    const pt::StacklessBVH& b = bvh->bvh;
    const pt::TriangleMesh& m = mesh->mesh;
    out->n_nodes = b.nodeNum;
    out->node_min = b.nodeMin.data();
    out->node_max = b.nodeMax.data();
    out->id_hit = b.idHit.data();
    out->id_miss = b.idMiss.data();
    out->leaf_first = b.leafFirst.data();
    out->leaf_count = b.leafCount.data();
    out->n_tri_index = (int32_t)b.triIndex.size();
    out->tri_index = b.triIndex.data();
    out->n_tris = m.numTriangles();
    out->pos = m.positions.data();
    out->nrm = m.normals.data();
    out->uv = m.texcoords.data();
    out->mat_id = m.idMaterial.data();
    out->n_materials = (int32_t)m.cMaterials.size();
    out->materials = m.cMaterials.data();
    out->n_textures = (int32_t)m.cTextures.size();
    out->textures = m.cTextures.data();
    if (env) {
        out->env_width = env->image.width;
        out->env_height = env->image.height;
        out->env_rgb = env->image.pixels.data();
    }
    out->n_lights = n_lights;
    out->lights = lights;
    return MIPT_OK;
}

}  // extern "C"
