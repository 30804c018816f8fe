// image_io.cpp -- Image::load / Image::save for the environment map, the
// material textures and the headless image output.
//
// The reference goes through stb_image / stb_image_write (vendored in the
// reference tree, not reproduced here).  What is kept, because pixels depend
// on it (C: = reference cpu/src/cs488.h):
//   * Image::load C:204-222: stbi_loadf(..., 3) then a vertical flip, so row
//     j = 0 is the BOTTOM row of the file;
//   * Radiance .hdr decode as stb_image 2.12 does it: rgb = mantissa byte *
//     ldexp(1, e - 136), (0,0,0) when e == 0; new-style RLE and flat scanlines;
//   * an 8-bit file loaded as float goes through pow(v/255, 2.2) (stb's
//     ldr->hdr default gamma);
//   * textures: stbi_load(..., 3): 8-bit RGB, top row first, palette expanded,
//     grey replicated, alpha dropped, 16-bit reduced to the high byte;
//   * Image::save C:223-236: clamp to [0,1], gamma 1.0, truncate to 8 bits,
//     top row first.
// PNG inflate/deflate is zlib's.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "../pt/host.hpp"

namespace pt {

namespace {

bool readFile(const char* path, std::vector<uint8_t>& data) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (n < 0) {
        fclose(fp);
        return false;
    }
    data.resize((size_t)n);
    size_t got = n ? fread(data.data(), 1, (size_t)n, fp) : 0;
    fclose(fp);
    return got == (size_t)n;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
    int p = a + b - c;
    int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

bool endsWith(const std::string& s, const char* suffix) {
    size_t n = strlen(suffix);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; i++)
        if (tolower((unsigned char)s[s.size() - n + i]) != suffix[i]) return false;
    return true;
}

}  // namespace

int decodePNG(const char* fileName, std::vector<uint8_t>& rgb, int& w, int& h) {
    std::vector<uint8_t> file;
    if (!readFile(fileName, file)) {
        set_error(std::string("unable to load texture: ") + fileName);
        return MIPT_ERR_IO;
    }
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) {
        set_error(std::string("not a PNG file: ") + fileName);
        return MIPT_ERR_IO;
    }
    uint32_t width = 0, height = 0;
    int depth = 0, color = 0, interlace = 0;
    std::vector<uint8_t> idat, palette;
    size_t p = 8;
    bool gotEnd = false;
    while (p + 12 <= file.size() && !gotEnd) {
        const uint32_t len = be32(&file[p]);
        const uint8_t* type = &file[p + 4];
        const uint8_t* body = &file[p + 8];
        if (p + 12 + (size_t)len > file.size()) break;
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            width = be32(body);
            height = be32(body + 4);
            depth = body[8];
            color = body[9];
            interlace = body[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            gotEnd = true;
        }
        p += 12 + (size_t)len;
    }
    if (width == 0 || height == 0 || width > 32768 || height > 32768 || interlace != 0) {
        set_error(std::string("unsupported PNG (size or interlacing): ") + fileName);
        return MIPT_ERR_IO;
    }
    int channels;
    switch (color) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: set_error("bad PNG colour type"); return MIPT_ERR_IO;
    }
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) {
        set_error("bad PNG bit depth");
        return MIPT_ERR_IO;
    }
    const size_t bitsPerPixel = (size_t)channels * depth;
    const size_t stride = ((size_t)width * bitsPerPixel + 7) / 8;
    const size_t bpp = (bitsPerPixel + 7) / 8;  // filter distance in bytes
    std::vector<uint8_t> raw((stride + 1) * (size_t)height);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) {
        set_error(std::string("PNG inflate failed: ") + fileName);
        return MIPT_ERR_IO;
    }
    // unfilter in place
    std::vector<uint8_t> img(stride * (size_t)height);
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t f = raw[(stride + 1) * y];
        const uint8_t* src = &raw[(stride + 1) * y + 1];
        uint8_t* cur = &img[stride * y];
        const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= bpp ? cur[x - bpp] : 0;
            const int b = up ? up[x] : 0;
            const int c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = src[x];
            switch (f) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: set_error("bad PNG filter"); return MIPT_ERR_IO;
            }
            cur[x] = (uint8_t)v;
        }
    }
    w = (int)width;
    h = (int)height;
    rgb.assign((size_t)3 * width * height, 0);
    static const int scale[9] = {0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01};
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t* row = &img[stride * y];
        for (uint32_t x = 0; x < width; x++) {
            uint8_t s[4] = {0, 0, 0, 0};
            for (int c = 0; c < channels; c++) {
                if (depth == 8) s[c] = row[(size_t)x * channels + c];
                else if (depth == 16) s[c] = row[((size_t)x * channels + c) * 2];  // high byte
                else {
                    const size_t bit = (size_t)x * depth;
                    const int v = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                    s[c] = (uint8_t)(color == 3 ? v : v * scale[depth]);
                }
            }
            uint8_t* o = &rgb[3 * ((size_t)y * width + x)];
            if (color == 3) {
                const size_t pi = (size_t)s[0] * 3;
                if (pi + 2 < palette.size()) { o[0] = palette[pi]; o[1] = palette[pi + 1]; o[2] = palette[pi + 2]; }
            } else if (channels <= 2) {
                o[0] = o[1] = o[2] = s[0];
            } else {
                o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
            }
        }
    }
    return MIPT_OK;
}

int decodeHDR(const char* fileName, std::vector<float>& rgb, int& w, int& h) {
    std::vector<uint8_t> f;
    if (!readFile(fileName, f)) {
        set_error(std::string("unable to load: ") + fileName);
        return MIPT_ERR_IO;
    }
    size_t p = 0;
    auto getLine = [&](std::string& s) {
        s.clear();
        while (p < f.size() && f[p] != '\n') s.push_back((char)f[p++]);
        if (p < f.size()) p++;
    };
    std::string line;
    getLine(line);
    if (line != "#?RADIANCE" && line != "#?RGBE") {
        set_error(std::string("not a Radiance HDR file: ") + fileName);
        return MIPT_ERR_IO;
    }
    bool fmt = false;
    while (true) {
        getLine(line);
        if (line.empty()) break;
        if (line == "FORMAT=32-bit_rle_rgbe") fmt = true;
        if (p >= f.size()) break;
    }
    if (!fmt) {
        set_error("unsupported HDR format");
        return MIPT_ERR_IO;
    }
    getLine(line);
    int hh = 0, ww = 0;
    if (sscanf(line.c_str(), "-Y %d +X %d", &hh, &ww) != 2 || hh <= 0 || ww <= 0) {
        set_error("unsupported HDR data layout");
        return MIPT_ERR_IO;
    }
    // the header's dimensions against what the file can hold, before anything is allocated: an RLE scanline takes at
    // least 4 bytes of header plus 2 bytes per run of up to 127 pixels and channel
    {
        const size_t left = f.size() > p ? f.size() - p : 0;
        const size_t minRow = ww >= 8 && ww < 32768 ? 4 + 8 * (((size_t)ww + 126) / 127) : (size_t)4 * ww;
        if (ww > 32768 || hh > 32768 || (size_t)hh * (minRow < (size_t)4 * ww ? minRow : (size_t)4 * ww) > left) {
            set_error("HDR header announces more pixels than the file holds");
            return MIPT_ERR_IO;
        }
    }
    w = ww;
    h = hh;
    rgb.assign((size_t)3 * ww * hh, 0.0f);
    auto convert = [](float* out, const uint8_t* in) {
        if (in[3] != 0) {
            const float f1 = (float)ldexp(1.0f, in[3] - (int)(128 + 8));
            out[0] = in[0] * f1;
            out[1] = in[1] * f1;
            out[2] = in[2] * f1;
        } else {
            out[0] = out[1] = out[2] = 0.0f;
        }
    };
    std::vector<uint8_t> scan((size_t)ww * 4);
    bool flat = (ww < 8 || ww >= 32768);
    for (int j = 0; j < hh; j++) {
        if (!flat) {
            if (p + 4 > f.size()) { set_error("HDR file truncated"); return MIPT_ERR_IO; }
            const uint8_t c1 = f[p], c2 = f[p + 1], len = f[p + 2];
            if (c1 != 2 || c2 != 2 || (len & 0x80)) {
                // not run-length encoded: the whole file is flat data (only legal on the first scanline)
                if (j != 0) { set_error("corrupt HDR scanline"); return MIPT_ERR_IO; }
                flat = true;
            } else {
                const int width = (len << 8) | f[p + 3];
                if (width != ww) { set_error("invalid decoded scanline length"); return MIPT_ERR_IO; }
                p += 4;
                for (int k = 0; k < 4; k++) {
                    int i = 0;
                    while (i < ww) {
                        if (p >= f.size()) { set_error("HDR file truncated"); return MIPT_ERR_IO; }
                        uint8_t count = f[p++];
                        if (count > 128) {
                            count -= 128;
                            if (p >= f.size() || i + count > ww) { set_error("corrupt HDR run"); return MIPT_ERR_IO; }
                            const uint8_t v = f[p++];
                            for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = v;
                        } else {
                            if (p + count > f.size() || i + count > ww || count == 0) { set_error("corrupt HDR run"); return MIPT_ERR_IO; }
                            for (int z = 0; z < count; z++) scan[(size_t)(i++) * 4 + k] = f[p++];
                        }
                    }
                }
                for (int i = 0; i < ww; i++) convert(&rgb[3 * ((size_t)j * ww + i)], &scan[(size_t)i * 4]);
                continue;
            }
        }
        // flat scanline
        if (p + (size_t)ww * 4 > f.size()) { set_error("HDR file truncated"); return MIPT_ERR_IO; }
        for (int i = 0; i < ww; i++) convert(&rgb[3 * ((size_t)j * ww + i)], &f[p + (size_t)i * 4]);
        p += (size_t)ww * 4;
    }
    return MIPT_OK;
}

int Image::load(const char* fileName) {
    std::vector<float> top;  // top row first
    int w = 0, h = 0;
    const std::string name(fileName);
    if (endsWith(name, ".hdr") || endsWith(name, ".pic")) {
        int rc = decodeHDR(fileName, top, w, h);
        if (rc != MIPT_OK) return rc;
    } else {
        std::vector<uint8_t> ldr;
        int rc = decodePNG(fileName, ldr, w, h);
        if (rc != MIPT_OK) return rc;
        top.resize(ldr.size());
        for (size_t i = 0; i < ldr.size(); i++) top[i] = (float)(powf(ldr[i] / 255.0f, 2.2f) * 1.0f);
    }
    width = w;
    height = h;
    pixels.resize((size_t)3 * w * h);
    size_t k = 0;
    for (int j = h - 1; j >= 0; j--) {
        memcpy(&pixels[(size_t)3 * j * w], &top[k], sizeof(float) * 3 * (size_t)w);
        k += (size_t)3 * w;
    }
    return MIPT_OK;
}

namespace {

void putBE32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}
void putChunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& body) {
    putBE32(out, (uint32_t)body.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    putBE32(out, (uint32_t)crc32(0L, &out[start], (uInt)(out.size() - start)));
}

}  // namespace

// one channel of Image::save, C:228: (unsigned char)(255.0f * gammaCorrection(toneMapping(v)))
uint8_t packChannel8(float r) {
    const float tm = std::fmax(std::fmin(1.0f, r), 0.0f);  // toneMapping C:159-162; a NaN comes out as 1, as std::min(1.0f, NaN) does
    const float g = powf(tm, 1.0f / 1.0f);                  // gammaCorrection C:164-168
    return (unsigned char)(255.0f * g);
}
void packRGB8(const float* rgb, size_t nValues, uint8_t* out) {
    for (size_t i = 0; i < nValues; i++) out[i] = packChannel8(rgb[i]);
}
// root-mean-square difference over all channels and the two means, in double
void compareImages(const float* a, const float* b, size_t nValues, double* rmse, double* meanA, double* meanB) {
    double se = 0.0, sa = 0.0, sb = 0.0;
    for (size_t i = 0; i < nValues; i++) {
        const double d = (double)a[i] - (double)b[i];
        se += d * d; sa += a[i]; sb += b[i];
    }
    const double n = nValues ? (double)nValues : 1.0;
    if (rmse) *rmse = std::sqrt(se / n);
    if (meanA) *meanA = sa / n;
    if (meanB) *meanB = sb / n;
}

int savePNG(const char* fileName, int width, int height, const float* rgb) {
    if (width <= 0 || height <= 0 || !rgb) {
        set_error("savePNG: bad arguments");
        return MIPT_ERR_INVALID;
    }
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)3 * width + 1) * height);
    for (int j = height - 1; j >= 0; j--) {
        raw.push_back(0);
        for (int i = 0; i < width; i++) {
            for (int c = 0; c < 3; c++) {
                raw.push_back(packChannel8(rgb[3 * ((size_t)j * width + i) + c]));
            }
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) {
        set_error("savePNG: deflate failed");
        return MIPT_ERR_IO;
    }
    z.resize(zlen);
    std::vector<uint8_t> out = {137, 80, 78, 71, 13, 10, 26, 10};
    std::vector<uint8_t> ihdr;
    putBE32(ihdr, (uint32_t)width);
    putBE32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    putChunk(out, "IHDR", ihdr);
    putChunk(out, "IDAT", z);
    putChunk(out, "IEND", {});
    FILE* fp = fopen(fileName, "wb");
    if (!fp) {
        set_error(std::string("cannot write ") + fileName);
        return MIPT_ERR_IO;
    }
    fwrite(out.data(), 1, out.size(), fp);
    fclose(fp);
    return MIPT_OK;
}

int saveHDR(const char* fileName, int width, int height, const float* rgb) {
    if (width <= 0 || height <= 0 || !rgb) {
        set_error("saveHDR: bad arguments");
        return MIPT_ERR_INVALID;
    }
    FILE* fp = fopen(fileName, "wb");
    if (!fp) {
        set_error(std::string("cannot write ") + fileName);
        return MIPT_ERR_IO;
    }
    fprintf(fp, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", height, width);
    std::vector<uint8_t> row((size_t)4 * width);
    for (int j = height - 1; j >= 0; j--) {  // top row first
        for (int i = 0; i < width; i++) {
            const float* px = &rgb[3 * ((size_t)j * width + i)];
            float m = px[0] > px[1] ? px[0] : px[1];
            if (px[2] > m) m = px[2];
            uint8_t* o = &row[(size_t)4 * i];
            if (!(m >= 1e-32f)) {
                o[0] = o[1] = o[2] = o[3] = 0;
            } else {
                int e;
                const float norm = frexpf(m, &e) * 256.0f / m;
                o[0] = (uint8_t)(px[0] > 0 ? px[0] * norm : 0);
                o[1] = (uint8_t)(px[1] > 0 ? px[1] * norm : 0);
                o[2] = (uint8_t)(px[2] > 0 ? px[2] * norm : 0);
                o[3] = (uint8_t)(e + 128);
            }
        }
        if (width < 8 || width >= 32768) {
            fwrite(row.data(), 1, row.size(), fp);  // flat scanline
        } else {
            // new-style scanline: per channel, literal chunks of <= 128 bytes
            const uint8_t head[4] = {2, 2, (uint8_t)(width >> 8), (uint8_t)(width & 255)};
            fwrite(head, 1, 4, fp);
            std::vector<uint8_t> chan;
            for (int k = 0; k < 4; k++) {
                chan.clear();
                for (int i = 0; i < width;) {
                    const int n = (width - i) < 128 ? (width - i) : 128;
                    chan.push_back((uint8_t)n);
                    for (int z = 0; z < n; z++) chan.push_back(row[(size_t)4 * (i + z) + k]);
                    i += n;
                }
                fwrite(chan.data(), 1, chan.size(), fp);
            }
        }
    }
    fclose(fp);
    return MIPT_OK;
}

int savePFM(const char* fileName, int width, int height, const float* rgb) {
    if (width <= 0 || height <= 0 || !rgb) {
        set_error("savePFM: bad arguments");
        return MIPT_ERR_INVALID;
    }
    FILE* fp = fopen(fileName, "wb");
    if (!fp) {
        set_error(std::string("cannot write ") + fileName);
        return MIPT_ERR_IO;
    }
    fprintf(fp, "PF\n%d %d\n-1.0\n", width, height);  // little-endian, bottom row first
    fwrite(rgb, sizeof(float), (size_t)3 * width * height, fp);
    fclose(fp);
    return MIPT_OK;
}

}  // namespace pt
