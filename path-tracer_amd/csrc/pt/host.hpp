// host.hpp -- host-side mirror of the reference's scene interface for the
// ray/scene path: TriangleMesh (loader), Image (env map / output), BVH +
// StacklessBVH (SAH build + threaded links).  Same names and argument meaning
// as the reference's classes (C: = reference cpu/src/cs488.h); storage is
// flat SoA so it can be handed to the device arena without conversion.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/mipt.h"
#include "vec.hpp"

namespace pt {

void set_error(const std::string& msg);   // thread-local message behind mipt_last_error()
const char* get_error();

// Material, C:413-433 (the BRDF/PDF/sampler members live in the device integrator)
struct Material {
    std::string name;
    int type = MIPT_MAT_LAMBERTIAN;
    float eta = 1.0f;
    vec3 Ka{0.0f, 0.0f, 0.0f};
    vec3 Kd{0.9f, 0.9f, 0.9f};
    vec3 Ks{0.0f, 0.0f, 0.0f};
    float Ns = 0.0f;
    bool isTextured = false;
    int texId = -1;
};

struct Texture {
    std::vector<uint8_t> rgb;
    int width = 0, height = 0;
};

// TriangleMesh, C:629-633, as SoA: 9 floats of positions / normals and 6 of
// texcoords per triangle, in file order.
class TriangleMesh {
public:
    std::vector<float> positions, normals, texcoords;
    std::vector<int32_t> idMaterial;
    std::vector<Material> materials;
    std::vector<Texture> textures;
    // flattened copies behind the C accessors
    std::vector<mipt_material> cMaterials;
    std::vector<mipt_texture> cTextures;

    int numTriangles() const { return (int)idMaterial.size(); }
    // C:835-923; returns a mipt_status
    int load(const char* filename);
    // C:949-970
    void createSingleTriangle();
    void refreshCViews();
};

// Image, C:153-237
class Image {
public:
    std::vector<float> pixels;  // rgb, row-major, row 0 = bottom
    int width = 0, height = 0;
    int load(const char* fileName);  // C:204-222
};
int savePNG(const char* fileName, int width, int height, const float* rgb);  // Image::save C:223-236
int saveHDR(const char* fileName, int width, int height, const float* rgb);
int savePFM(const char* fileName, int width, int height, const float* rgb);
uint8_t packChannel8(float v);                                        // one channel of Image::save, C:228
void packRGB8(const float* rgb, size_t nValues, uint8_t* out);
void compareImages(const float* a, const float* b, size_t nValues, double* rmse, double* meanA, double* meanB);
// decoders shared by Image::load and the texture loader
int decodePNG(const char* fileName, std::vector<uint8_t>& rgb, int& w, int& h);     // as stbi_load(..., 3)
int decodeHDR(const char* fileName, std::vector<float>& rgb, int& w, int& h);       // as stbi_loadf(..., 3)

// BVH + StacklessBVH + StacklessBVH2 link sets, C:1318-1327, C:1637-1660, C:1713-1730
class StacklessBVH {
public:
    int nodeNum = 0, leafNum = 0, depth = 0, maxLeaf = 0;
    std::vector<float> nodeMin, nodeMax;  // 3 per node
    std::vector<int32_t> idHit, idMiss;   // CollNode
    std::vector<int32_t> leafFirst, leafCount;
    std::vector<int32_t> idLeft, idRight, idParent, idSibling;
    std::vector<int32_t> triIndex;        // all leaves' triList, concatenated in node order
    // preCalc (C:817-831) + BVH::build (C:1562-1595) + link assignment
    // allAxes: sweep all three axes per node (a better tree, but not the reference's: new node numbering)
    int build(const TriangleMesh* mesh, bool allAxes = false);
};

// flat binary cache of a loaded mesh + built BVH (host/scene_cache.cpp)
// tag: what the cache was built from (free text); a load with expectTag != nullptr refuses a cache with another tag
int saveSceneCache(const char* path, const TriangleMesh& mesh, const StacklessBVH& bvh, const std::string& tag = std::string());
int loadSceneCache(const char* path, TriangleMesh& mesh, StacklessBVH& bvh, const std::string* expectTag = nullptr);
std::string fileFingerprint(const char* path);  // "name:size:fnv1a64" of a file's content

}  // namespace pt
