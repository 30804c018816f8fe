// shading.hpp -- device-side pieces of the integrators: hit reconstruction,
// camera, environment lookup, texture fetch, counter RNG, material sampling.
// fp32 with the reference's operation order; the few double-precision
// sub-expressions of the reference (4.0*PI*falloff, pow(x,2)) are evaluated
// in double here too.  Transcendentals (acosf, sinf, cosf) are the device
// library's, so shading agrees with the CPU oracle to a stated tolerance, not
// bitwise.  Reference: cs488.h (C:).
#pragma once
#include "traverse.hpp"

namespace ptd {

using pt::vec2;

constexpr float kPI = 3.14159265358979f;  // C:50

struct CameraDev {  // C:65-70, C:81-85; distanceToFilm is computed on the host (tanf)
    int width, height;
    vec3 eye, up, viewDir;
    float aspect, filmSize, distanceToFilm;
};

struct HitInfo {  // C:510-517
    float t;
    vec3 P, N;
    vec2 T;
    int material;
};

// second half of raytraceTriangle, C:800-809.  The record's first five float4 hold everything but the texture
// coordinates (device_scene.hpp): an untextured vertex gathers five 16-B pieces of its triangle's line, not seven; T is
// filled in by hit_texcoord for the vertices whose material has a texture (and for mipt_trace's HitInfo output).
__device__ __forceinline__ HitInfo reconstruct_hit(const DevScene& s, int tri, float t, float beta, float gamma,
                                                   const vec3 rayD) {
    const float4* a = s.triAttr + (size_t)kTriAttrF4 * tri;
    const float4 q0 = a[0], q1 = a[1], q2 = a[2], q3 = a[3], q4 = a[4];
    const vec3 p0{q0.x, q0.y, q0.z}, p1{q0.w, q1.x, q1.y}, p2{q1.z, q1.w, q2.x};
    const vec3 n0{q2.y, q2.z, q2.w}, n1{q3.x, q3.y, q3.z}, n2{q3.w, q4.x, q4.y};
    const float alpha = 1.0f - beta - gamma;
    const vec3 phi{alpha, beta, gamma};
    HitInfo h;
    h.t = t;
    h.material = f2i(q4.z);
    h.P = pt::mulcols(p0, p1, p2, phi);
    h.N = pt::mulcols(n0, n1, n2, phi);
    h.T = vec2{0.0f, 0.0f};
    if (pt::dot(h.N, rayD) > 0.0f) h.N = -h.N;
    return h;
}
// C:806: the interpolated texture coordinate of the same hit
__device__ __forceinline__ vec2 hit_texcoord(const DevScene& s, int tri, float beta, float gamma) {
    const float4* a = s.triAttr + (size_t)kTriAttrF4 * tri;
    const float4 q5 = a[5], q6 = a[6];
    const vec2 t0{q5.x, q5.y}, t1{q5.z, q5.w}, t2{q6.x, q6.y};
    const float alpha = 1.0f - beta - gamma;
    return pt::mulcols(t0, t1, t2, vec3{alpha, beta, gamma});
}

// Scene::eyeRay, C:2183-2197
__device__ __forceinline__ void eye_ray(const CameraDev& cam, int x, int y, vec3& o, vec3& d) {
    const vec3 wDir = pt::normalize(-cam.viewDir);
    const vec3 uDir = pt::normalize(pt::cross(cam.up, wDir));
    const vec3 vDir = pt::cross(wDir, uDir);
    const float imPlaneUPos = ((float)x + 0.5f) / (float)cam.width - 0.5f;
    const float imPlaneVPos = ((float)y + 0.5f) / (float)cam.height - 0.5f;
    const vec3 pixelPos = ((cam.eye + (cam.aspect * cam.filmSize * imPlaneUPos) * uDir) +
                           (cam.filmSize * imPlaneVPos) * vDir) - cam.distanceToFilm * wDir;
    o = cam.eye;
    d = pt::normalize(pixelPos - cam.eye);
}

// int(float) as x86 cvttss2si does it (the oracle's rule): NaN / out of range -> INT_MIN
__device__ __forceinline__ int float_to_int(float v) {
    if (!(v > -2147483648.0f && v < 2147483648.0f)) return (int)0x80000000u;
    return (int)v;
}

// Scene::getBackground, C:2038-2053; black when no environment is loaded
__device__ __forceinline__ vec3 get_background(const DevScene& s, const vec3 dir) {
    if (s.envW <= 0) return vec3{0.0f, 0.0f, 0.0f};
    const float r = acosf(dir.z) / 6.283184f;
    const vec2 xy = pt::normalize(vec2{dir.x, dir.y});
    const vec2 tex{r * xy.x + .5f, -r * xy.y + .5f};
    int x = float_to_int(tex.x * (float)s.envW) % s.envW;
    int y = float_to_int(tex.y * (float)s.envH) % s.envH;
    if (x < 0) x += s.envW;
    if (y < 0) y += s.envH;
    const float* px = s.env + (size_t)3 * ((size_t)x + (size_t)y * s.envW);
    return vec3{px[0], px[1], px[2]};
}

struct MaterialDev {
    int type;
    float eta;
    vec3 Kd, Ks;
    int texOfs, texW, texH;  // texOfs < 0: untextured
};

__device__ __forceinline__ MaterialDev load_material(const DevScene& s, int m) {
    const float4* q = s.materials + (size_t)4 * m;
    const float4 a = q[0], b = q[1], c = q[2];
    MaterialDev r;
    r.type = f2i(a.x);
    r.eta = a.y;
    r.texOfs = f2i(a.w);
    r.Kd = vec3{b.x, b.y, b.z};
    r.texW = f2i(b.w);
    r.Ks = vec3{c.x, c.y, c.z};
    r.texH = f2i(c.w);
    return r;
}

// Material::fetchTexture, C:445-457
__device__ __forceinline__ vec3 fetch_texture(const DevScene& s, const MaterialDev& m, const vec2 tex) {
    int x = float_to_int(tex.x * (float)m.texW) % m.texW;
    int y = float_to_int(tex.y * (float)m.texH) % m.texH;
    if (x < 0) x += m.texW;
    if (y < 0) y += m.texH;
    const unsigned char* px = s.texels + (size_t)m.texOfs + (size_t)3 * ((size_t)x + (size_t)y * m.texW);
    return vec3{(float)px[0], (float)px[1], (float)px[2]} / 255.0f;
}

// counter RNG: one draw per (seed, pixel, sample, dimension)
__device__ __forceinline__ unsigned pcg_hash(unsigned v) {
    const unsigned state = v * 747796405u + 2891336453u;
    const unsigned word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
__device__ __forceinline__ float counter_rand(unsigned seedHash, unsigned pixel, unsigned sample, unsigned dim) {
    const unsigned u = pcg_hash(pixel + pcg_hash(sample + pcg_hash(dim + seedHash)));
    // PCG32::rand, C:144-146: float(double(u) / 2^32); (float)u rounds the same way and the scale is exact
    return (float)u * 2.3283064365386963e-10f;
}

// point-light irradiance factor, C:2301 / C:2422: float(max(0, N.l) / (4.0 * PI * falloff))
__device__ __forceinline__ float light_factor(float ndotl, float falloff) {
    const float m = (0.0f < ndotl) ? ndotl : 0.0f;  // std::max(0.0f, x)
    return (float)((double)m / (4.0 * (double)kPI * (double)falloff));
}

// Material::sampler for MAT_LAMBERTIAN, C:490-494
__device__ __forceinline__ vec3 sample_lambert(const vec3 norm, float u1, float u2) {
    const float r1 = 2.0f * kPI * u1, r2 = u2, r2s = sqrtf(r2);
    const vec3 w = norm;
    const vec3 u = pt::normalize(pt::cross((double)fabsf(w.x) > .1 ? vec3{0, 1, 0} : vec3{1, 0, 0}, w));
    const vec3 v = pt::cross(w, u);
    return pt::normalize((u * cosf(r1) * r2s + v * sinf(r1) * r2s) + w * sqrtf(1.0f - r2));
}
// mirror direction, C:496 and C:2380 / C:2459 / C:2485: normalize(w - 2*dot(w,n)*n)
__device__ __forceinline__ vec3 mirror(const vec3 w, const vec3 n) { return pt::normalize(w - (2.0f * pt::dot(w, n)) * n); }

// Snell term, C:2362 / C:2471: float(1 - pow(eta/eta2, 2) * (1 - pow(dot(w,n), 2))) in double
__device__ __forceinline__ float snell_inside(float etaRatio, float wdotn) {
    const double q = (double)etaRatio, c = (double)wdotn;
    return (float)(1.0 - (q * q) * (1.0 - c * c));
}
// C:2367 / C:2475: normalize(etaRatio * (w - dot(w,n)*n) - sqrtf(inside)*n)
__device__ __forceinline__ vec3 refract_dir(const vec3 w, const vec3 n, float etaRatio, float inside) {
    return pt::normalize(etaRatio * (w - pt::dot(w, n) * n) - sqrtf(inside) * n);
}

}  // namespace ptd
