#include <new>
// png_reader.cpp — minimal PNG (and binary PGM/PPM) reader for the dataset camera: zlib inflate + the five PNG
// scan-line filters; colour types 0/2/3/4/6, bit depths 8 and 16 (1/2/4 for grey and palette), no interlace.
// Output follows what the reference obtains through libgd's truecolor accessors (src/VideoLib/datasetcam.cpp:
// 152-160): r,g,b bytes, grey replicated, alpha ignored.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rebvo/datasetcam.h"

namespace rebvo {
bool decode_jpeg(const std::vector<unsigned char> &d, std::vector<RGB24Pixel> &out, unsigned &w, unsigned &h, std::string &err, bool *mono);   // jpeg_reader.cpp
namespace {

bool read_file(const std::string &file, std::vector<unsigned char> &buf, std::string &err) {
    FILE *f = fopen(file.c_str(), "rb");
    if (!f) { err = "cannot open " + file; return false; }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? n : 0);
    const size_t got = n > 0 ? fread(buf.data(), 1, n, f) : 0;
    fclose(f);
    if ((long)got != n) { err = "short read on " + file; return false; }
    return true;
}
inline unsigned be32(const unsigned char *p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Header-supplied dimensions size every buffer below: a corrupt file must not wrap size_t or ask for gigabytes (the camera
// sizes this library handles are at most 1024 wide; 16384 x 16384 leaves any real dataset plenty of room).
inline bool sane_size(unsigned w, unsigned h) { return w >= 1 && h >= 1 && w <= 16384 && h <= 16384; }

bool decode_png(const std::vector<unsigned char> &d, std::vector<RGB24Pixel> &out, unsigned &w, unsigned &h, std::string &err, bool *mono) {
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (d.size() < 33 || memcmp(d.data(), sig, 8) != 0) { err = "not a PNG"; return false; }
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
    bool have_hdr = false;
    while (pos + 12 <= d.size()) {
        const unsigned len = be32(&d[pos]);
        const char *type = (const char *)&d[pos + 4];
        if (pos + 12 + (size_t)len > d.size()) { err = "truncated PNG chunk"; return false; }
        const unsigned char *body = &d[pos + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) { err = "bad IHDR"; return false; }
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            have_hdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_hdr || w == 0 || h == 0) { err = "PNG without a valid IHDR"; return false; }
    if (!sane_size(w, h)) { err = "PNG header declares an implausible image size"; return false; }   // before any allocation sized by it
    if (interlace) { err = "interlaced PNG is not supported"; return false; }
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: err = "unknown PNG colour type"; return false;
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) {
        err = "unsupported PNG bit depth"; return false;
    }
    const size_t bpp_bits = (size_t)channels * depth;
    const size_t stride = ((size_t)w * bpp_bits + 7) / 8, fb = bpp_bits >= 8 ? bpp_bits / 8 : 1;   // filter byte distance
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf rawlen = raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), idat.size()) != Z_OK || rawlen != raw.size()) { err = "PNG inflate failed"; return false; }
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    out.resize((size_t)w * h);
    for (unsigned y = 0; y < h; y++) {
        const unsigned char *line = &raw[(stride + 1) * y];
        const int ft = line[0];
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= fb ? cur[i - fb] : 0, b = prev[i], c = i >= fb ? prev[i - fb] : 0;
            int v = line[1 + i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: err = "bad PNG filter"; return false;
            }
            cur[i] = (unsigned char)v;
        }
        for (unsigned x = 0; x < w; x++) {
            unsigned char r, g, bl;
            if (depth >= 8) {
                const size_t step = depth / 8;
                const unsigned char *p = &cur[(size_t)x * channels * step];
                if (ctype == 0 || ctype == 4) { r = g = bl = p[0]; }
                else if (ctype == 3) {
                    const unsigned idx = p[0];
                    if ((idx + 1) * 3 > plte.size()) { r = g = bl = 0; } else { r = plte[idx * 3]; g = plte[idx * 3 + 1]; bl = plte[idx * 3 + 2]; }
                } else { r = p[0]; g = p[step]; bl = p[2 * step]; }
            } else {
                const size_t bit = (size_t)x * depth;
                const unsigned v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
                if (ctype == 3) {
                    if ((v + 1) * 3 > plte.size()) { r = g = bl = 0; } else { r = plte[v * 3]; g = plte[v * 3 + 1]; bl = plte[v * 3 + 2]; }
                } else {
                    r = g = bl = (unsigned char)(v * 255 / ((1u << depth) - 1));
                }
            }
            RGB24Pixel &o = out[(size_t)y * w + x];
            o.pix.r = r; o.pix.g = g; o.pix.b = bl;
        }
        prev.swap(cur);
    }
    if (mono) *mono = (ctype == 0 || ctype == 4);   // grey / grey + alpha: r = g = b for every pixel
    return true;
}

bool decode_pnm(const std::vector<unsigned char> &d, std::vector<RGB24Pixel> &out, unsigned &w, unsigned &h, std::string &err, bool *mono) {
    size_t pos = 2;
    auto next_int = [&](unsigned &v) {
        while (pos < d.size()) {
            if (d[pos] == '#') { while (pos < d.size() && d[pos] != '\n') pos++; }
            else if (isspace(d[pos])) pos++;
            else break;
        }
        if (pos >= d.size() || !isdigit(d[pos])) return false;
        v = 0;
        while (pos < d.size() && isdigit(d[pos])) v = v * 10 + (d[pos++] - '0');
        return true;
    };
    unsigned maxv = 0;
    if (!next_int(w) || !next_int(h) || !next_int(maxv) || maxv == 0 || maxv > 255) { err = "bad PNM header"; return false; }
    if (!sane_size(w, h)) { err = "PNM header declares an implausible image size"; return false; }
    pos++;   // single whitespace after maxval
    const int ch = d[1] == '5' ? 1 : 3;
    if (pos + (size_t)w * h * ch > d.size()) { err = "truncated PNM"; return false; }
    out.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const unsigned char *p = &d[pos + i * ch];
        out[i].pix.r = p[0]; out[i].pix.g = p[ch == 1 ? 0 : 1]; out[i].pix.b = p[ch == 1 ? 0 : 2];
    }
    if (mono) *mono = ch == 1;
    return true;
}

}  // namespace

bool LoadImageRGB24(const std::string &file, std::vector<RGB24Pixel> &out, unsigned &w, unsigned &h, std::string &err, bool *mono) {
    if (mono) *mono = false;
    std::vector<unsigned char> d;
    if (!read_file(file, d, err)) return false;
    try {   // an allocation failure is a camera error (DataSetCam ends the sequence), not std::terminate in the track thread
        if (d.size() > 8 && d[0] == 137 && d[1] == 'P') return decode_png(d, out, w, h, err, mono);
        if (d.size() > 2 && d[0] == 'P' && (d[1] == '5' || d[1] == '6')) return decode_pnm(d, out, w, h, err, mono);
        if (d.size() > 2 && d[0] == 0xFF && d[1] == 0xD8) {   // datasetcam.cpp:128-131 (gdImageCreateFromJpeg): baseline JPEG, jpeg_reader.cpp
            if (decode_jpeg(d, out, w, h, err, mono)) return true;
            err = file + ": " + err;
            return false;
        }
    } catch (const std::bad_alloc &) {
        err = "out of memory while decoding " + file;
        return false;
    }

    err = "unsupported image format (PNG, baseline JPEG, PGM, PPM are read): " + file;
    return false;
}

}  // namespace rebvo
