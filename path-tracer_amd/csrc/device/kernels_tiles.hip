// kernels_tiles.hip -- the two small kernels either side of the multi-GPU gather (api_node.hip): a rank copies the
// image tiles it owns out of its frame buffer into one dense send buffer ("pack"), rank 0 copies a rank's received
// tiles into the frame ("unpack").  Tile k of a list covers pixels [tx*tileW, +tileW) x [ty*tileH, +tileH) of a
// row-major W x H float3 image; tiles on the right / top edge are clipped, their out-of-image slots carry zeros and
// are skipped on unpack.  One thread per pixel slot, grid-strided; pure copies (HBM-bound, 24 B per pixel).
#include "kernels.hpp"

namespace ptd {

__global__ __launch_bounds__(kBlock) void k_pack_tiles(const float* __restrict__ rgb, const int* __restrict__ tiles, unsigned nTiles,
                                                       int tileW, int tileH, int tilesX, int width, int height,
                                                       float* __restrict__ out) {
    const unsigned tilePix = (unsigned)(tileW * tileH);
    const size_t n = (size_t)nTiles * tilePix;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const unsigned k = (unsigned)(i / tilePix), r = (unsigned)(i % tilePix);
        const int t = tiles[k];
        const int x = (t % tilesX) * tileW + (int)(r % (unsigned)tileW), y = (t / tilesX) * tileH + (int)(r / (unsigned)tileW);
        float a = 0.0f, b = 0.0f, c = 0.0f;
        if (x < width && y < height) {
            const float* p = rgb + 3 * ((size_t)y * width + x);
            a = p[0]; b = p[1]; c = p[2];
        }
        float* q = out + 3 * i;
        q[0] = a; q[1] = b; q[2] = c;
    }
}

__global__ __launch_bounds__(kBlock) void k_unpack_tiles(const float* __restrict__ in, const int* __restrict__ tiles, unsigned nTiles,
                                                         int tileW, int tileH, int tilesX, int width, int height,
                                                         float* __restrict__ rgb) {
    const unsigned tilePix = (unsigned)(tileW * tileH);
    const size_t n = (size_t)nTiles * tilePix;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const unsigned k = (unsigned)(i / tilePix), r = (unsigned)(i % tilePix);
        const int t = tiles[k];
        const int x = (t % tilesX) * tileW + (int)(r % (unsigned)tileW), y = (t / tilesX) * tileH + (int)(r / (unsigned)tileW);
        if (x < width && y < height) {
            const float* q = in + 3 * i;
            float* p = rgb + 3 * ((size_t)y * width + x);
            p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
        }
    }
}

static inline unsigned tiles_grid(size_t n) {
    size_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > 16384) g = 16384;
    return (unsigned)g;
}
void launch_pack_tiles(hipStream_t st, const float* rgb, const int* tiles, unsigned nTiles, int tileW, int tileH, int tilesX,
                       int width, int height, float* out) {
    if (!nTiles) return;
    hipLaunchKernelGGL(k_pack_tiles, dim3(tiles_grid((size_t)nTiles * tileW * tileH)), dim3(kBlock), 0, st, rgb, tiles, nTiles,
                       tileW, tileH, tilesX, width, height, out);
}
void launch_unpack_tiles(hipStream_t st, const float* in, const int* tiles, unsigned nTiles, int tileW, int tileH, int tilesX,
                         int width, int height, float* rgb) {
    if (!nTiles) return;
    hipLaunchKernelGGL(k_unpack_tiles, dim3(tiles_grid((size_t)nTiles * tileW * tileH)), dim3(kBlock), 0, st, in, tiles, nTiles,
                       tileW, tileH, tilesX, width, height, rgb);
}

}  // namespace ptd
