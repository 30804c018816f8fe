// linalg_ref.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref).  Compiles the reference's own, UNMODIFIED linalg.h
// (found by -I /root/reference/cpu/src; nothing of it is copied here) and exports its vector primitives over
// plain float arrays, so that the oracle's restatement (pt_oracle.c: dot3, cross3, normalize3, det3, mulcols3) and
// the product's pt/vec.hpp can be compared with the reference bit for bit.
// The calls are exactly the ones cs488.h makes: dot / cross / normalize / length2 on float3 (cs488.h:2183-2197,
// :777-813), determinant(float3x3(c0, c1, c2)) (cs488.h:786-789) and mul(float3x3(c0, c1, c2), v) (cs488.h:800-803).
// Built by oracle/Makefile into oracle/_ref/libref_linalg.so; outputs only there.
#include "linalg.h"

using namespace linalg::aliases;

extern "C" {

// op: 0 dot(a,b) -> 1 | 1 cross(a,b) -> 3 | 2 normalize(a) -> 3 | 3 determinant(float3x3(a,b,c)) -> 1
//     4 mul(float3x3(a,b,c), d) -> 3 | 5 length2(a) -> 1 | 6 normalize(float2(a.x, a.y)) -> 2
//     7 a / b (float3 / float3: three IEEE divides, the slab test's quotients) -> 3
int ref_vec_ops(int op, const float* a, const float* b, const float* c, const float* d, long long n, float* out) {
    for (long long i = 0; i < n; i++) {
        const float3 A(a[3 * i], a[3 * i + 1], a[3 * i + 2]);
        const float3 B = b ? float3(b[3 * i], b[3 * i + 1], b[3 * i + 2]) : float3(0, 0, 0);
        const float3 Cc = c ? float3(c[3 * i], c[3 * i + 1], c[3 * i + 2]) : float3(0, 0, 0);
        const float3 D = d ? float3(d[3 * i], d[3 * i + 1], d[3 * i + 2]) : float3(0, 0, 0);
        switch (op) {
        case 0: out[i] = dot(A, B); break;
        case 1: { const float3 r = cross(A, B); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
        case 2: { const float3 r = normalize(A); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
        case 3: out[i] = determinant(float3x3(A, B, Cc)); break;
        case 4: { const float3 r = mul(float3x3(A, B, Cc), D); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
        case 5: out[i] = length2(A); break;
        case 6: { const float2 r = normalize(float2(A.x, A.y)); out[2 * i] = r.x; out[2 * i + 1] = r.y; break; }
        case 7: { const float3 r = A / B; out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
        default: return -1;
        }
    }
    return 0;
}

}  // extern "C"
