// scene_cache.cpp -- flat binary cache of a loaded mesh and its built BVH (parsing bunny.obj and
// the O(n log n) SAH sweep dominate start-up; the cache makes benchmark runs start at once).  One file:
//   "MIPTSC02", u32 endianness probe, u32 record sizes, u64 FNV-1a of everything after the header, then the
//   source tag (a string the writer chose to identify what the cache was built from: the CLI stores the .obj /
//   .mtl sizes and content hash and the BVH mode; a reader that passes a tag gets MIPT_ERR_IO for a cache with
//   another one) and counted sections of plain arrays in the order written below.  Host byte order, no pointers.
// Loading checks the magic, the probe, the record sizes, the checksum and every count against the file size;
// the arrays' CONTENT is validated where it is used (mipt_scene_create checks every link and index).
#include <cstdio>
#include <cstring>

#include "../pt/host.hpp"

namespace pt {
namespace {

constexpr char kMagic[8] = {'M', 'I', 'P', 'T', 'S', 'C', '0', '2'};  // 02: the payload starts with the source tag
constexpr uint32_t kProbe = 0x01020304u;

struct Writer {
    std::vector<uint8_t> buf;
    void raw(const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; buf.insert(buf.end(), b, b + n); }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    template <class T> void vec(const std::vector<T>& v) { u64(v.size()); if (!v.empty()) raw(v.data(), v.size() * sizeof(T)); }
    void str(const std::string& s) { u64(s.size()); raw(s.data(), s.size()); }
};

struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    bool raw(void* out, size_t n) {
        if (!ok || (size_t)(end - p) < n) { ok = false; return false; }
        memcpy(out, p, n);
        p += n;
        return true;
    }
    uint32_t u32() { uint32_t v = 0; raw(&v, 4); return v; }
    uint64_t u64() { uint64_t v = 0; raw(&v, 8); return v; }
    template <class T> bool vec(std::vector<T>& v) {
        const uint64_t n = u64();
        if (!ok || n > (uint64_t)(end - p) / sizeof(T)) { ok = false; return false; }
        v.resize((size_t)n);
        return n == 0 || raw(v.data(), (size_t)n * sizeof(T));
    }
    bool str(std::string& s) {
        const uint64_t n = u64();
        if (!ok || n > (uint64_t)(end - p)) { ok = false; return false; }
        s.assign((const char*)p, (size_t)n);
        p += n;
        return true;
    }
};

uint64_t fnv1a64(const uint8_t* p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

}  // namespace

int saveSceneCache(const char* path, const TriangleMesh& m, const StacklessBVH& b, const std::string& tag) {
    Writer w;
    w.str(tag);
    w.vec(m.positions); w.vec(m.normals); w.vec(m.texcoords); w.vec(m.idMaterial);
    w.u64(m.materials.size());
    for (const Material& mt : m.materials) {
        w.str(mt.name);
        w.u32((uint32_t)mt.type);
        w.raw(&mt.eta, 4); w.raw(&mt.Ka, 12); w.raw(&mt.Kd, 12); w.raw(&mt.Ks, 12); w.raw(&mt.Ns, 4);
        w.u32(mt.isTextured ? 1u : 0u);
        w.u32((uint32_t)mt.texId);
    }
    w.u64(m.textures.size());
    for (const Texture& t : m.textures) { w.u32((uint32_t)t.width); w.u32((uint32_t)t.height); w.vec(t.rgb); }
    w.u32((uint32_t)b.nodeNum); w.u32((uint32_t)b.leafNum); w.u32((uint32_t)b.depth); w.u32((uint32_t)b.maxLeaf);
    w.vec(b.nodeMin); w.vec(b.nodeMax); w.vec(b.idHit); w.vec(b.idMiss); w.vec(b.leafFirst); w.vec(b.leafCount);
    w.vec(b.idLeft); w.vec(b.idRight); w.vec(b.idParent); w.vec(b.idSibling); w.vec(b.triIndex);

    FILE* f = fopen(path, "wb");
    if (!f) { set_error(std::string("cannot write ") + path); return MIPT_ERR_IO; }
    const uint32_t sizes[2] = {(uint32_t)sizeof(float), (uint32_t)sizeof(int32_t)};
    const uint64_t sum = fnv1a64(w.buf.data(), w.buf.size());
    bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(&kProbe, 4, 1, f) == 1 && fwrite(sizes, 4, 2, f) == 2 &&
              fwrite(&sum, 8, 1, f) == 1 && fwrite(w.buf.data(), 1, w.buf.size(), f) == w.buf.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok) { set_error(std::string("short write to ") + path); return MIPT_ERR_IO; }
    return MIPT_OK;
}

int loadSceneCache(const char* path, TriangleMesh& m, StacklessBVH& b, const std::string* expectTag) {
    FILE* f = fopen(path, "rb");
    if (!f) { set_error(std::string("cannot open ") + path); return MIPT_ERR_IO; }
    std::vector<uint8_t> all;
    uint8_t chunk[1 << 16];
    size_t got;
    while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) all.insert(all.end(), chunk, chunk + got);
    fclose(f);
    const size_t header = 8 + 4 + 8 + 8;
    uint32_t probe = 0, sizes[2] = {0, 0};
    uint64_t sum = 0;
    if (all.size() < header || memcmp(all.data(), kMagic, 8) != 0) { set_error(std::string(path) + ": not a mipt scene cache"); return MIPT_ERR_IO; }
    memcpy(&probe, &all[8], 4); memcpy(sizes, &all[12], 8); memcpy(&sum, &all[20], 8);
    if (probe != kProbe || sizes[0] != sizeof(float) || sizes[1] != sizeof(int32_t)) { set_error(std::string(path) + ": written on an incompatible machine"); return MIPT_ERR_IO; }
    if (fnv1a64(all.data() + header, all.size() - header) != sum) { set_error(std::string(path) + ": checksum mismatch"); return MIPT_ERR_IO; }
    Reader r{all.data() + header, all.data() + all.size()};
    std::string tag;
    if (!r.str(tag)) { set_error(std::string(path) + ": truncated"); return MIPT_ERR_IO; }
    if (expectTag && tag != *expectTag) { set_error(std::string(path) + ": built from another source (cache tag '" + tag + "', wanted '" + *expectTag + "')"); return MIPT_ERR_IO; }
    r.vec(m.positions); r.vec(m.normals); r.vec(m.texcoords); r.vec(m.idMaterial);
    const uint64_t nMat = r.u64();
    if (!r.ok || nMat > all.size()) { set_error(std::string(path) + ": truncated"); return MIPT_ERR_IO; }
    m.materials.assign((size_t)nMat, Material());
    for (Material& mt : m.materials) {
        r.str(mt.name);
        mt.type = (int)r.u32();
        r.raw(&mt.eta, 4); r.raw(&mt.Ka, 12); r.raw(&mt.Kd, 12); r.raw(&mt.Ks, 12); r.raw(&mt.Ns, 4);
        mt.isTextured = r.u32() != 0;
        mt.texId = (int)r.u32();
    }
    const uint64_t nTex = r.u64();
    if (!r.ok || nTex > all.size()) { set_error(std::string(path) + ": truncated"); return MIPT_ERR_IO; }
    m.textures.assign((size_t)nTex, Texture());
    for (Texture& t : m.textures) { t.width = (int)r.u32(); t.height = (int)r.u32(); r.vec(t.rgb); }
    b.nodeNum = (int)r.u32(); b.leafNum = (int)r.u32(); b.depth = (int)r.u32(); b.maxLeaf = (int)r.u32();
    r.vec(b.nodeMin); r.vec(b.nodeMax); r.vec(b.idHit); r.vec(b.idMiss); r.vec(b.leafFirst); r.vec(b.leafCount);
    r.vec(b.idLeft); r.vec(b.idRight); r.vec(b.idParent); r.vec(b.idSibling); r.vec(b.triIndex);
    if (!r.ok || r.p != r.end) { set_error(std::string(path) + ": truncated or trailing bytes"); return MIPT_ERR_IO; }
    // shape checks: everything the accessors and mipt_scene_desc_fill rely on
    const size_t nt = m.idMaterial.size(), nn = (size_t)(b.nodeNum < 0 ? 0 : b.nodeNum);
    bool shape = nt > 0 && m.positions.size() == 9 * nt && m.normals.size() == 9 * nt && m.texcoords.size() == 6 * nt &&
                 !m.materials.empty() && nn > 0 && b.nodeMin.size() == 3 * nn && b.nodeMax.size() == 3 * nn &&
                 b.idHit.size() == nn && b.idMiss.size() == nn && b.leafFirst.size() == nn && b.leafCount.size() == nn &&
                 b.idLeft.size() == nn && b.idRight.size() == nn && b.idParent.size() == nn && b.idSibling.size() == nn;
    for (size_t i = 0; shape && i < nt; i++) shape = m.idMaterial[i] >= 0 && (size_t)m.idMaterial[i] < m.materials.size();
    for (const Material& mt : m.materials) shape = shape && mt.texId < (int)m.textures.size() && (!mt.isTextured || mt.texId >= 0);
    for (const Texture& t : m.textures) shape = shape && t.width > 0 && t.height > 0 && t.rgb.size() == (size_t)3 * t.width * t.height;
    if (!shape) { set_error(std::string(path) + ": inconsistent array sizes"); return MIPT_ERR_IO; }
    m.refreshCViews();
    return MIPT_OK;
}

// size and FNV-1a of a file's content as "name:size:hash"; "name:absent" when it cannot be read
std::string fileFingerprint(const char* path) {
    const char* base = strrchr(path, '/');
    std::string out = base ? base + 1 : path;
    FILE* f = fopen(path, "rb");
    if (!f) return out + ":absent";
    uint64_t h = 1469598103934665603ull, n = 0;
    uint8_t chunk[1 << 16];
    size_t got;
    while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) {
        for (size_t i = 0; i < got; i++) { h ^= chunk[i]; h *= 1099511628211ull; }
        n += got;
    }
    fclose(f);
    char buf[64];
    snprintf(buf, sizeof buf, ":%llu:%016llx", (unsigned long long)n, (unsigned long long)h);
    return out + buf;
}

}  // namespace pt
