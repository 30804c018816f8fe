/* stb_ref.c -- TEST INFRASTRUCTURE ONLY (oracle/_ref).  Compiles the reference's own, UNMODIFIED stb_image.h (found by
 * -I /root/reference/cpu/src; nothing of it is copied here) and exports the two decode calls the reference makes:
 * stbi_loadf(file, &w, &h, &comp, 3) in Image::load (cs488.h:204-222) and stbi_load(file, &w, &h, &comp, 3) in
 * TriangleMesh::loadTexture (cs488.h:975-988), so that path-tracer_amd/csrc/host/image_io.cpp (an own PNG / RGBE
 * decoder) can be compared with them value for value.  The vertical flip of Image::load lives in cs488.h, which
 * cannot be compiled here; the test applies it (rows reversed) and says so.
 * Built by oracle/Makefile into oracle/_ref/libref_stb.so; outputs only there. */
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"

float* ref_stbi_loadf_rgb(const char* path, int* w, int* h, int* comp) { return stbi_loadf(path, w, h, comp, 3); }
unsigned char* ref_stbi_load_rgb(const char* path, int* w, int* h, int* comp) { return stbi_load(path, w, h, comp, 3); }
void ref_stbi_free(void* p) { stbi_image_free(p); }
