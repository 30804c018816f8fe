// obj_loader.cpp -- TriangleMesh::load for .obj/.mtl scenes.
//
// Restates the behaviour of the reference loader (C: = reference
// cpu/src/cs488.h): TriangleMesh::load C:835-923, LoadMTL C:990-1033,
// ParseOBJ C:1035-1312, createSingleTriangle C:949-970.  Behaviour kept on
// purpose, because it decides triangle order, normals and materials:
//   * records are read in pieces of at most 79 characters (fgets(line, 80));
//   * only the first three vertices of a face are used;
//   * if ANY face vertex lacks a normal (texcoord) index, the whole mesh gets
//     face normals (zero texcoords);
//   * a material exists once its "Ns" record is read; "map_Kd" attaches to
//     the material pushed last; material names keep their line terminator and
//     "usemtl" must match it exactly;
//   * typing: Ns == 100 -> metal, name "glass*" -> glass (eta 1.5), name
//     "light*" -> light; without any matched usemtl the mesh gets one default
//     material and the typing pass is skipped.
// Deviation: unreadable files and out-of-range indices give an error code
// where the reference dereferences a null FILE* / reads out of bounds.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../pt/host.hpp"

namespace pt {

namespace {

// one fgets(line, 80, fp) worth of text
struct PieceReader {
    FILE* fp;
    char line[81];
    bool next() { return fgets(line, 80, fp) != nullptr; }
};

std::string baseDir(const std::string& filepath) {
    size_t p = filepath.find_last_of("/\\");
    std::string d = (p != std::string::npos) ? filepath.substr(0, p) : std::string("");
    return d + "/";
}

bool startsWith(const char* s, const char* prefix) { return strncmp(s, prefix, strlen(prefix)) == 0; }

// text after the first `skip` characters (std::string::erase(0, skip))
std::string tail(const char* line, size_t skip) {
    size_t n = strlen(line);
    return n > skip ? std::string(line + skip) : std::string();
}

void dropLast(std::string& s) {
    if (!s.empty()) s.erase(s.size() - 1, 1);
}

// "v/t/n" -> indices, 0 where absent (atoi of " ")
void splitFaceWord(char* word, int& v, int& t, int& n) {
    const char* tp = " ";
    const char* np = " ";
    bool haveT = false;
    for (char* p = word; *p != '\0'; ++p) {
        if (*p == '/') {
            if (!haveT) {
                tp = p + 1;
                haveT = true;
            } else {
                np = p + 1;
            }
            *p = '\0';
        }
    }
    v = atoi(word);
    t = atoi(tp);
    n = atoi(np);
}

int loadMTL(const std::string& fileName, const std::string& dir, TriangleMesh& mesh) {
    FILE* fp = fopen(fileName.c_str(), "r");
    if (!fp) {
        set_error("cannot open material library \"" + fileName + "\"");
        return MIPT_ERR_IO;
    }
    PieceReader rd{fp, {0}};
    Material mtl;  // carries Ka/Kd/Ks/Ns over to the next material when a record is omitted
    int rc = MIPT_OK;
    while (rd.next()) {
        const char* line = rd.line;
        float r = 0, g = 0, b = 0, s = 0;
        const int count = (int)mesh.materials.size();
        if (startsWith(line, "newmtl")) {
            mtl.name = tail(line, 7);
            mtl.isTextured = false;
            mtl.texId = -1;
        } else if (startsWith(line, "Ka")) {
            sscanf(tail(line, 3).c_str(), "%f %f %f\n", &r, &g, &b);
            mtl.Ka = vec3{r, g, b};
        } else if (startsWith(line, "Kd")) {
            sscanf(tail(line, 3).c_str(), "%f %f %f\n", &r, &g, &b);
            mtl.Kd = vec3{r, g, b};
        } else if (startsWith(line, "Ks")) {
            sscanf(tail(line, 3).c_str(), "%f %f %f\n", &r, &g, &b);
            mtl.Ks = vec3{r, g, b};
        } else if (startsWith(line, "Ns")) {
            sscanf(tail(line, 3).c_str(), "%f\n", &s);
            mtl.Ns = s;
            mesh.materials.push_back(mtl);
        } else if (startsWith(line, "map_Kd")) {
            std::string texName = tail(line, 7);
            dropLast(texName);
            if (count == 0) {
                set_error("map_Kd before any material in \"" + fileName + "\"");
                rc = MIPT_ERR_INVALID;
                break;
            }
            Texture tex;
            rc = decodePNG((dir + texName).c_str(), tex.rgb, tex.width, tex.height);
            if (rc != MIPT_OK) break;
            mesh.materials[count - 1].isTextured = true;
            mesh.materials[count - 1].texId = (int)mesh.textures.size();
            mesh.textures.push_back(std::move(tex));
        }
    }
    fclose(fp);
    return rc;
}

}  // namespace

void TriangleMesh::refreshCViews() {
    cMaterials.resize(materials.size());
    for (size_t i = 0; i < materials.size(); ++i) {
        const Material& m = materials[i];
        mipt_material& c = cMaterials[i];
        c.type = m.type;
        c.eta = m.eta;
        c.Ka[0] = m.Ka.x; c.Ka[1] = m.Ka.y; c.Ka[2] = m.Ka.z;
        c.Kd[0] = m.Kd.x; c.Kd[1] = m.Kd.y; c.Kd[2] = m.Kd.z;
        c.Ks[0] = m.Ks.x; c.Ks[1] = m.Ks.y; c.Ks[2] = m.Ks.z;
        c.Ns = m.Ns;
        c.tex_id = m.isTextured ? m.texId : -1;
    }
    cTextures.resize(textures.size());
    for (size_t i = 0; i < textures.size(); ++i) {
        cTextures[i].rgb = textures[i].rgb.data();
        cTextures[i].width = textures[i].width;
        cTextures[i].height = textures[i].height;
    }
}

void TriangleMesh::createSingleTriangle() {
    positions = {-0.5f, -0.5f, 0.0f, 0.5f, -0.5f, 0.0f, 0.0f, 0.5f, 0.0f};
    const vec3 p0{positions[0], positions[1], positions[2]};
    const vec3 p1{positions[3], positions[4], positions[5]};
    const vec3 p2{positions[6], positions[7], positions[8]};
    const vec3 n = normalize(cross(p1 - p0, p2 - p0));
    normals = {n.x, n.y, n.z, n.x, n.y, n.z, n.x, n.y, n.z};
    texcoords = {0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 0.0f};
    idMaterial = {0};
    materials.assign(1, Material());
    textures.clear();
    refreshCViews();
}

int TriangleMesh::load(const char* filename) {
    positions.clear(); normals.clear(); texcoords.clear(); idMaterial.clear();
    materials.clear(); textures.clear();

    FILE* fp = fopen(filename, "r");
    if (!fp) {
        set_error(std::string("cannot open \"") + filename + "\" for reading");
        return MIPT_ERR_IO;
    }
    const std::string dir = baseDir(filename);
    PieceReader rd{fp, {0}};

    // pass 1: counts, and the material library
    int nv = 0, nn = 0, nf = 0, nt = 0;
    while (rd.next()) {
        const char* line = rd.line;
        if (startsWith(line, "mtllib")) {
            std::string lib = tail(line, 7);
            dropLast(lib);
            int rc = loadMTL(dir + lib, dir, *this);
            if (rc != MIPT_OK) {
                fclose(fp);
                return rc;
            }
        }
        if (line[0] == 'v') {
            if (line[1] == 'n') nn++;
            else if (line[1] == 't') nt++;
            else nv++;
        } else if (line[0] == 'f') {
            nf++;
        }
    }
    fseek(fp, 0, SEEK_SET);

    std::vector<float> v((size_t)3 * nv), n((size_t)3 * nn), t((size_t)2 * nt);
    std::vector<int> vInd((size_t)3 * nf, -1), nInd((size_t)3 * nf, -1), tInd((size_t)3 * nf, -1), mInd((size_t)nf, 0);
    size_t nvertices = 0, nnormals = 0, ntexcoords = 0, nindices = 0;
    int ntriangles = 0;
    bool noNormals = false, noTexCoords = false, noMaterials = true;
    int cmaterial = 0;

    // pass 2
    while (rd.next()) {
        char* line = rd.line;
        if (line[0] == 'v') {
            if (line[1] == 'n') {
                float x = 0, y = 0, z = 0;
                sscanf(&line[2], "%f %f %f\n", &x, &y, &z);
                float l = sqrtf(x * x + y * y + z * z);
                x = x / l; y = y / l; z = z / l;
                if (nnormals + 3 <= n.size()) { n[nnormals] = x; n[nnormals + 1] = y; n[nnormals + 2] = z; }
                nnormals += 3;
            } else if (line[1] == 't') {
                float a = 0, b = 0;
                sscanf(&line[2], "%f %f\n", &a, &b);
                if (ntexcoords + 2 <= t.size()) { t[ntexcoords] = a; t[ntexcoords + 1] = b; }
                ntexcoords += 2;
            } else {
                float x = 0, y = 0, z = 0;
                sscanf(&line[1], "%f %f %f\n", &x, &y, &z);
                if (nvertices + 3 <= v.size()) { v[nvertices] = x; v[nvertices + 1] = y; v[nvertices + 2] = z; }
                nvertices += 3;
            }
        }
        if (startsWith(line, "usemtl")) {
            const std::string name = tail(line, 7);
            for (size_t i = 0; i < materials.size(); i++) {
                if (name == materials[i].name) {
                    cmaterial = (int)i;
                    noMaterials = false;
                    break;
                }
            }
        } else if (line[0] == 'f') {
            char s[3][80];
            s[0][0] = s[1][0] = s[2][0] = '\0';
            sscanf(&line[1], "%79s %79s %79s\n", s[0], s[1], s[2]);
            if (ntriangles >= nf) break;
            mInd[ntriangles] = cmaterial;
            for (int k = 0; k < 3; k++) {
                int vI, tI, nI;
                splitFaceWord(s[k], vI, tI, nI);
                vInd[nindices] = vI - 1;
                if (nI) nInd[nindices] = nI - 1; else noNormals = true;
                if (tI) tInd[nindices] = tI - 1; else noTexCoords = true;
                nindices++;
            }
            ntriangles++;
        }
    }
    fclose(fp);

    if (ntriangles == 0) {
        set_error(std::string("\"") + filename + "\" holds no faces");
        return MIPT_ERR_INVALID;
    }
    // bounds (the reference reads out of range here)
    for (size_t i = 0; i < nindices; i++) {
        if (vInd[i] < 0 || vInd[i] >= nv) { set_error("vertex index out of range"); return MIPT_ERR_INVALID; }
        if (!noNormals && (nInd[i] < 0 || nInd[i] >= nn)) { set_error("normal index out of range"); return MIPT_ERR_INVALID; }
        if (!noTexCoords && (tInd[i] < 0 || tInd[i] >= nt)) { set_error("texcoord index out of range"); return MIPT_ERR_INVALID; }
    }

    // TriangleMesh::load proper, C:849-913
    if (!noMaterials) {
        for (Material& m : materials) {
            m.type = MIPT_MAT_LAMBERTIAN;
            if (m.Ns == 100.0f) m.type = MIPT_MAT_METAL;
            if (m.name.compare(0, 5, "glass", 0, 5) == 0) {
                m.type = MIPT_MAT_GLASS;
                m.eta = 1.5f;
            }
            if (m.name.compare(0, 5, "light", 0, 5) == 0) m.type = MIPT_MAT_LIGHT;
        }
    } else {
        materials.resize(1);
    }

    positions.resize((size_t)9 * ntriangles);
    normals.resize((size_t)9 * ntriangles);
    texcoords.assign((size_t)6 * ntriangles, 0.0f);
    idMaterial.assign((size_t)ntriangles, 0);
    for (int i = 0; i < ntriangles; i++) {
        for (int k = 0; k < 3; k++) {
            const int vi = vInd[3 * i + k];
            positions[9 * i + 3 * k + 0] = v[3 * vi + 0];
            positions[9 * i + 3 * k + 1] = v[3 * vi + 1];
            positions[9 * i + 3 * k + 2] = v[3 * vi + 2];
        }
        if (!noNormals) {
            for (int k = 0; k < 3; k++) {
                const int ni = nInd[3 * i + k];
                normals[9 * i + 3 * k + 0] = n[3 * ni + 0];
                normals[9 * i + 3 * k + 1] = n[3 * ni + 1];
                normals[9 * i + 3 * k + 2] = n[3 * ni + 2];
            }
        } else {
            const float* p = &positions[9 * i];
            const vec3 p0{p[0], p[1], p[2]}, p1{p[3], p[4], p[5]}, p2{p[6], p[7], p[8]};
            const vec3 fn = normalize(cross(p1 - p0, p2 - p0));
            for (int k = 0; k < 3; k++) {
                normals[9 * i + 3 * k + 0] = fn.x;
                normals[9 * i + 3 * k + 1] = fn.y;
                normals[9 * i + 3 * k + 2] = fn.z;
            }
        }
        if (!noMaterials) {
            const int m = mInd[i];
            if (!noTexCoords && materials[m].isTextured) {
                for (int k = 0; k < 3; k++) {
                    const int ti = tInd[3 * i + k];
                    texcoords[6 * i + 2 * k + 0] = t[2 * ti + 0];
                    texcoords[6 * i + 2 * k + 1] = t[2 * ti + 1];
                }
            }
            idMaterial[i] = m;
        }
    }
    refreshCViews();
    return MIPT_OK;
}

}  // namespace pt
