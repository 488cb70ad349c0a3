// jpeg_reader.cpp — baseline and progressive (Huffman, 8-bit) JPEG decoder for the dataset camera.
//
// The reference reads JPEG frames through libgd (src/VideoLib/datasetcam.cpp:128-131: gdImageCreateFromJpeg), i.e. through
// libjpeg with its default settings: the slow-but-accurate integer IDCT (jidctint.c), "fancy" triangle-filter upsampling of
// subsampled chroma (jdsample.c: h2v1_fancy_upsample / h2v2_fancy_upsample) and the 16.16 fixed-point YCbCr -> RGB tables
// (jdcolor.c).  A frame has to reach the tracker with the reference's pixel values, so this decoder restates exactly those
// three published integer algorithms (Independent JPEG Group / libjpeg-turbo, whose outputs agree bit for bit); the test
// (tests/test_dataset_cpu.py) compares it with PIL's libjpeg-turbo on grey, 4:4:4, 4:2:2 and 4:2:0 images, odd sizes, restart
// intervals and optimised Huffman tables (the vertical-only 4:4:0 form follows libjpeg-turbo's h1v2_fancy_upsample; no encoder
// at hand writes it, so that one is untested).  Progressive files (SOF2; round 5) are decoded scan by scan into coefficient planes —
// spectral selection and successive approximation as ITU T.81 G.1.2 / libjpeg's jdphuff.c describe them (DC first / refinement, AC
// first with end-of-band runs, AC refinement with its correction bits) — and go through the same IDCT, upsampling and colour
// conversion once the last scan is in: a complete file decodes to libjpeg's pixels (its inter-block smoothing only applies to files
// that end early).  Not decoded: arithmetic / lossless / 12-bit / CMYK files (a clear message; tools/jpeg_to_png.py converts
// anything PIL reads).
#include <cstdint>
#include <cstring>
#include <new>

#include "rebvo/datasetcam.h"

namespace rebvo {
namespace {

const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {   // canonical Huffman table, decoded length by length (ITU T.81 F.2.2.3)
    bool set = false;
    int mincode[17], maxcode[18], valptr[17];
    unsigned char vals[256];
};

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int wblocks = 0, hblocks = 0;          // allocated size in blocks (whole MCUs)
    int cw = 0, ch = 0;                    // true component size in samples: ceil(image * samp / max samp)
    int pred = 0;
    bool scanned = false;                  // a scan has carried this component's data
    std::vector<unsigned char> plane;      // [hblocks * 8][wblocks * 8]
    std::vector<int> coef;                 // progressive: [hblocks][wblocks][64] quantised coefficients, natural order
    uint16_t q[64];                        // progressive: the quantiser as it stood at the component's first scan (libjpeg latches it there)
    bool q_latched = false;
};

struct Bits {
    const unsigned char *p, *end;
    uint32_t acc = 0;
    int n = 0;
    bool marker_hit = false;
    int fill() {   // one more byte of entropy-coded data (FF 00 -> FF; any other marker ends the segment: zeros from then on)
        unsigned c = 0;
        if (!marker_hit && p < end) {
            c = *p++;
            if (c == 0xFF) {
                if (p < end && *p == 0x00) p++;
                else { marker_hit = true; p--; c = 0; }
            }
        } else marker_hit = true;
        acc = (acc << 8) | c;
        n += 8;
        return 0;
    }
    inline int bit() {
        if (n == 0) fill();
        n--;
        return (acc >> n) & 1;
    }
    inline int receive(int s) {
        int v = 0;
        for (int i = 0; i < s; i++) v = (v << 1) | bit();
        return v;
    }
    void reset() { acc = 0; n = 0; marker_hit = false; }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

bool build_huff(Huff &h, const unsigned char *counts, const unsigned char *vals, int nvals) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        h.valptr[l] = k;
        h.mincode[l] = code;
        code += counts[l - 1];
        k += counts[l - 1];
        h.maxcode[l] = counts[l - 1] ? code - 1 : -1;
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    if (k > 256 || k != nvals) return false;
    memcpy(h.vals, vals, k);
    h.set = true;
    return true;
}
inline int decode_sym(Bits &b, const Huff &h) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | b.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
}

inline unsigned char clamp8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// jpeg_idct_islow (jidctint.c): 8x8 inverse DCT on dequantised coefficients, 13-bit constants, two passes
void idct_islow(const int *coef /* natural order, dequantised */, unsigned char *out, int stride) {
    constexpr int CB = 13, P1 = 2;
    constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
                  F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
                  F_2_562915447 = 20995, F_3_072711026 = 25172;
    auto descale = [](int64_t x, int n) { return (int)((x + ((int64_t)1 << (n - 1))) >> n); };
    int ws[64];
    for (int c = 0; c < 8; c++) {
        const int *in = coef + c;
        int *w = ws + c;
        if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0 && in[48] == 0 && in[56] == 0) {
            const int dc = (int)((int64_t)in[0] * (1 << P1));   // (|in[0]| <= 2^24: the decoder rejects larger coefficients)
            for (int r = 0; r < 8; r++) w[8 * r] = dc;
            continue;
        }
        int64_t z2 = in[16], z3 = in[48];
        int64_t z1 = (z2 + z3) * F_0_541196100;
        int64_t tmp2 = z1 + z3 * (-F_1_847759065);
        int64_t tmp3 = z1 + z2 * F_0_765366865;
        z2 = in[0]; z3 = in[32];
        int64_t tmp0 = (z2 + z3) * (1 << CB);
        int64_t tmp1 = (z2 - z3) * (1 << CB);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        w[0] = descale(tmp10 + tmp3, CB - P1); w[56] = descale(tmp10 - tmp3, CB - P1);
        w[8] = descale(tmp11 + tmp2, CB - P1); w[48] = descale(tmp11 - tmp2, CB - P1);
        w[16] = descale(tmp12 + tmp1, CB - P1); w[40] = descale(tmp12 - tmp1, CB - P1);
        w[24] = descale(tmp13 + tmp0, CB - P1); w[32] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; r++) {
        const int *w = ws + 8 * r;
        unsigned char *o = out + (size_t)r * stride;
        int64_t z2 = w[2], z3 = w[6];
        int64_t z1 = (z2 + z3) * F_0_541196100;
        int64_t tmp2 = z1 + z3 * (-F_1_847759065);
        int64_t tmp3 = z1 + z2 * F_0_765366865;
        int64_t tmp0 = ((int64_t)w[0] + w[4]) * (1 << CB);
        int64_t tmp1 = ((int64_t)w[0] - w[4]) * (1 << CB);
        const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int64_t z4 = tmp1 + tmp3;
        const int64_t z5 = (z3 + z4) * F_1_175875602;
        tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
        z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        constexpr int SH = CB + P1 + 3;
        o[0] = clamp8(descale(tmp10 + tmp3, SH) + 128); o[7] = clamp8(descale(tmp10 - tmp3, SH) + 128);
        o[1] = clamp8(descale(tmp11 + tmp2, SH) + 128); o[6] = clamp8(descale(tmp11 - tmp2, SH) + 128);
        o[2] = clamp8(descale(tmp12 + tmp1, SH) + 128); o[5] = clamp8(descale(tmp12 - tmp1, SH) + 128);
        o[3] = clamp8(descale(tmp13 + tmp0, SH) + 128); o[4] = clamp8(descale(tmp13 - tmp0, SH) + 128);
    }
}

// one output row of 2:1 horizontal fancy upsampling (h2v1_fancy_upsample): in[cw] -> out[2 cw]
void up_h2_row(const unsigned char *in, int cw, unsigned char *out) {
    if (cw == 1) { out[0] = out[1] = in[0]; return; }
    int v = in[0];
    *out++ = (unsigned char)v;
    *out++ = (unsigned char)((v * 3 + in[1] + 2) >> 2);
    for (int c = 1; c < cw - 1; c++) {
        v = in[c] * 3;
        *out++ = (unsigned char)((v + in[c - 1] + 1) >> 2);
        *out++ = (unsigned char)((v + in[c + 1] + 2) >> 2);
    }
    v = in[cw - 1];
    *out++ = (unsigned char)((v * 3 + in[cw - 2] + 1) >> 2);
    *out++ = (unsigned char)v;
}
// one output row of 2:1 x 2:1 fancy upsampling (h2v2_fancy_upsample): the nearer input row `in0` weighs 3, the further `in1` 1
void up_h2v2_row(const unsigned char *in0, const unsigned char *in1, int cw, unsigned char *out) {
    if (cw == 1) {
        const int s = in0[0] * 3 + in1[0];
        out[0] = (unsigned char)((s * 4 + 8) >> 4);
        out[1] = (unsigned char)((s * 4 + 7) >> 4);
        return;
    }
    int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
    *out++ = (unsigned char)((thiscol * 4 + 8) >> 4);
    *out++ = (unsigned char)((thiscol * 3 + nextcol + 7) >> 4);
    lastcol = thiscol; thiscol = nextcol;
    for (int c = 2; c < cw; c++) {
        nextcol = in0[c] * 3 + in1[c];
        *out++ = (unsigned char)((thiscol * 3 + lastcol + 8) >> 4);
        *out++ = (unsigned char)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol; thiscol = nextcol;
    }
    *out++ = (unsigned char)((thiscol * 3 + lastcol + 8) >> 4);
    *out++ = (unsigned char)((thiscol * 4 + 7) >> 4);
}
// 1:1 x 2:1 (vertical only) fancy upsampling (h1v2_fancy_upsample, libjpeg-turbo): (3 near + far + bias) >> 2, bias 1 for the upper
// output row of a pair and 2 for the lower
void up_v2_row(const unsigned char *in0, const unsigned char *in1, int cw, int bias, unsigned char *out) {
    for (int c = 0; c < cw; c++) out[c] = (unsigned char)((in0[c] * 3 + in1[c] + bias) >> 2);
}

}  // namespace

bool decode_jpeg(const std::vector<unsigned char> &d, std::vector<RGB24Pixel> &out, unsigned &w, unsigned &h, std::string &err, bool *mono) {
    if (mono) *mono = false;
    if (d.size() < 4 || d[0] != 0xFF || d[1] != 0xD8) { err = "not a JPEG"; return false; }
    uint16_t qt[4][64];
    bool qt_set[4] = {false, false, false, false};
    Huff hdc[4], hac[4];
    std::vector<Comp> comps;
    int restart = 0, adobe_transform = -1;
    bool have_sof = false, decoded = false, progressive = false;
    size_t pos = 2;
    auto be16 = [&](size_t p) { return (unsigned)d[p] << 8 | d[p + 1]; };
    int hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    while (pos + 4 <= d.size() && !decoded) {
        if (d[pos] != 0xFF) { err = "JPEG: marker expected"; return false; }
        while (pos < d.size() && d[pos] == 0xFF) pos++;   // fill bytes
        if (pos >= d.size()) break;
        const unsigned m = d[pos++];
        if (m == 0xD9) break;                             // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > d.size()) { err = "JPEG: truncated"; return false; }
        const unsigned len = be16(pos);
        if (len < 2 || pos + len > d.size()) { err = "JPEG: bad segment length"; return false; }
        const unsigned char *s = &d[pos + 2];
        const unsigned n = len - 2;
        switch (m) {
        case 0xDB: {   // DQT
            unsigned o = 0;
            while (o < n) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                o++;
                if (tq > 3 || o + (pq ? 128u : 64u) > n) { err = "JPEG: bad DQT"; return false; }
                for (int i = 0; i < 64; i++) {
                    qt[tq][kZigzag[i]] = pq ? (uint16_t)(s[o] << 8 | s[o + 1]) : s[o];
                    o += pq ? 2 : 1;
                }
                qt_set[tq] = true;
            }
            break;
        }
        case 0xC4: {   // DHT
            unsigned o = 0;
            while (o + 17 <= n) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                int total = 0;
                for (int i = 0; i < 16; i++) total += s[o + 1 + i];
                if (tc > 1 || th > 3 || o + 17 + total > n || !build_huff(tc ? hac[th] : hdc[th], s + o + 1, s + o + 17, total)) { err = "JPEG: bad DHT"; return false; }
                o += 17 + total;
            }
            break;
        }
        case 0xC0: case 0xC1: case 0xC2: {   // SOF0 / SOF1: sequential, SOF2: progressive; Huffman
            progressive = m == 0xC2;
            if (have_sof) { err = "JPEG: a second frame header"; return false; }   // (libjpeg: JERR_SOF_DUPLICATE)
            if (n < 6 || s[0] != 8) { err = "JPEG: only 8-bit samples are decoded"; return false; }
            h = be16(pos + 3); w = be16(pos + 5);
            const int nf = s[5];
            if ((nf != 1 && nf != 3) || n < 6u + 3u * nf || w < 1 || h < 1 || w > 16384 || h > 16384) { err = "JPEG: unsupported frame (1 or 3 components, at most 16384 x 16384)"; return false; }
            comps.resize(nf);
            for (int i = 0; i < nf; i++) {
                comps[i].id = s[6 + 3 * i]; comps[i].h = s[7 + 3 * i] >> 4; comps[i].v = s[7 + 3 * i] & 15; comps[i].tq = s[8 + 3 * i] & 3;
                if (comps[i].h < 1 || comps[i].h > 2 || comps[i].v < 1 || comps[i].v > 2) { err = "JPEG: sampling factors above 2 are not decoded"; return false; }
                hmax = comps[i].h > hmax ? comps[i].h : hmax; vmax = comps[i].v > vmax ? comps[i].v : vmax;
            }
            mcux = ((int)w + 8 * hmax - 1) / (8 * hmax); mcuy = ((int)h + 8 * vmax - 1) / (8 * vmax);
            for (Comp &c : comps) {
                c.wblocks = mcux * c.h; c.hblocks = mcuy * c.v;
                c.cw = ((int)w * c.h + hmax - 1) / hmax; c.ch = ((int)h * c.v + vmax - 1) / vmax;
                c.plane.assign((size_t)c.wblocks * 8 * c.hblocks * 8, 0);
                if (progressive) c.coef.assign((size_t)c.wblocks * c.hblocks * 64, 0);
            }
            have_sof = true;
            break;
        }
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            err = "JPEG: lossless / hierarchical / arithmetic-coded files are not decoded (Huffman baseline and progressive only; tools/jpeg_to_png.py converts them)";
            return false;
        case 0xDD: if (n >= 2) restart = be16(pos + 2); break;
        case 0xEE: if (n >= 12 && !memcmp(s, "Adobe", 5)) adobe_transform = s[11]; break;
        case 0xDA: {   // SOS + entropy-coded data
            if (!have_sof) { err = "JPEG: scan before frame header"; return false; }
            if (n < 1) { err = "JPEG: bad SOS"; return false; }   // (the component count is the segment's first byte)
            const int ns = s[0];
            if (ns < 1 || ns > (int)comps.size() || n < 1u + 2u * ns + 3u) { err = "JPEG: bad SOS"; return false; }
            std::vector<Comp *> sc;
            for (int i = 0; i < ns; i++) {
                Comp *c = nullptr;
                for (Comp &k : comps) if (k.id == s[1 + 2 * i]) c = &k;
                if (!c) { err = "JPEG: scan names an unknown component"; return false; }
                c->td = s[2 + 2 * i] >> 4; c->ta = s[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3 || !qt_set[c->tq] || (!progressive && (!hdc[c->td].set || !hac[c->ta].set))) { err = "JPEG: scan uses an undefined table"; return false; }
                if (progressive && !c->q_latched) { memcpy(c->q, qt[c->tq], sizeof c->q); c->q_latched = true; }
                c->pred = 0;
                c->scanned = true;
                sc.push_back(c);
            }
            Bits b;
            b.p = &d[pos + len];
            b.end = d.data() + d.size();
            // MCU grid of this scan: interleaved (all components: mcux x mcuy MCUs of h x v blocks each) or one component alone
            // (its own blocks in raster order, only those that hold image samples: ceil(cw / 8) x ceil(ch / 8))
            const bool inter = ns > 1;
            const int gx = inter ? mcux : (sc[0]->cw + 7) / 8, gy = inter ? mcuy : (sc[0]->ch + 7) / 8;
            if (progressive) {
                // ---- one progressive scan: spectral band [Ss, Se], successive approximation Ah -> Al (T.81 G.1.2, jdphuff.c) ----
                const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
                if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || (Ah != 0 && Ah != Al + 1)) { err = "JPEG: bad progressive scan parameters"; return false; }
                for (Comp *c : sc)
                    if (Ss == 0 ? (Ah == 0 && !hdc[c->td].set) : !hac[c->ta].set) { err = "JPEG: scan uses an undefined table"; return false; }
                int until = restart, eobrun = 0;
                const int p1 = 1 << Al, m1 = -(1 << Al);
                for (int my = 0; my < gy; my++)
                    for (int mx = 0; mx < gx; mx++) {
                        if (restart && until == 0) {
                            b.reset();
                            while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) b.p++;
                            if (b.p + 1 < b.end) b.p += 2;
                            for (Comp *c : sc) c->pred = 0;
                            eobrun = 0;
                            until = restart;
                        }
                        until--;
                        for (Comp *c : sc) {
                            const int bh = inter ? c->h : 1, bv = inter ? c->v : 1;
                            for (int by = 0; by < bv; by++)
                                for (int bx = 0; bx < bh; bx++) {
                                    const int col = (inter ? mx * c->h + bx : mx), row = (inter ? my * c->v + by : my);
                                    int *cf = &c->coef[((size_t)row * c->wblocks + col) * 64];
                                    if (Ss == 0) {
                                        if (Ah == 0) {   // DC, first pass
                                            const int t = decode_sym(b, hdc[c->td]);
                                            if (t < 0 || t > 11) { err = "JPEG: corrupt DC code"; return false; }
                                            c->pred += t ? extend(b.receive(t), t) : 0;
                                            if (c->pred < -32768 || c->pred > 32767) { err = "JPEG: DC prediction out of range"; return false; }
                                            cf[0] = c->pred * (1 << Al);
                                        } else if (b.bit()) {   // DC refinement: one more bit
                                            cf[0] |= p1;
                                        }
                                    } else if (Ah == 0) {   // AC, first pass
                                        if (eobrun > 0) { eobrun--; continue; }
                                        for (int k = Ss; k <= Se; k++) {
                                            const int rs = decode_sym(b, hac[c->ta]);
                                            if (rs < 0) { err = "JPEG: corrupt AC code"; return false; }
                                            const int r = rs >> 4, sz = rs & 15;
                                            if (sz) {
                                                k += r;
                                                if (k > Se) { err = "JPEG: corrupt block"; return false; }
                                                cf[kZigzag[k]] = extend(b.receive(sz), sz) * (1 << Al);
                                            } else if (r == 15) {
                                                k += 15;
                                            } else {   // EOBr: this block and (2^r + bits - 1) more end here
                                                eobrun = (1 << r) - 1;
                                                if (r) eobrun += b.receive(r);
                                                break;
                                            }
                                        }
                                    } else {   // AC refinement (jdphuff.c: decode_mcu_AC_refine)
                                        auto correct = [&](int &v) { if (b.bit() && (v & p1) == 0) v += v >= 0 ? p1 : m1; };
                                        int k = Ss;
                                        if (eobrun == 0) {
                                            for (; k <= Se; k++) {
                                                const int rs = decode_sym(b, hac[c->ta]);
                                                if (rs < 0) { err = "JPEG: corrupt AC code"; return false; }
                                                int r = rs >> 4, val = 0;
                                                if (rs & 15) {
                                                    if ((rs & 15) != 1) { err = "JPEG: corrupt refinement scan"; return false; }
                                                    val = b.bit() ? p1 : m1;
                                                } else if (r != 15) {
                                                    eobrun = 1 << r;
                                                    if (r) eobrun += b.receive(r);
                                                    break;   // the rest of the band is handled below
                                                }
                                                // over the coefficients that are already non-zero (a correction bit each) and r that are still zero
                                                do {
                                                    int &v = cf[kZigzag[k]];
                                                    if (v != 0) correct(v);
                                                    else if (--r < 0) break;
                                                    k++;
                                                } while (k <= Se);
                                                if (val) {
                                                    if (k > Se) { err = "JPEG: corrupt block"; return false; }
                                                    cf[kZigzag[k]] = val;
                                                }
                                            }
                                        }
                                        if (eobrun > 0) {   // inside an end-of-band run: only correction bits for what is non-zero already
                                            for (; k <= Se; k++) {
                                                int &v = cf[kZigzag[k]];
                                                if (v != 0) correct(v);
                                            }
                                            eobrun--;
                                        }
                                    }
                                }
                        }
                    }
                size_t q2 = b.p - d.data();
                while (q2 + 1 < d.size() && !(d[q2] == 0xFF && d[q2 + 1] != 0x00 && !(d[q2 + 1] >= 0xD0 && d[q2 + 1] <= 0xD7))) q2++;
                pos = q2;
                continue;
            }
            int coef[64];
            int until_restart = restart;
            for (int my = 0; my < gy; my++)
                for (int mx = 0; mx < gx; mx++) {
                    if (restart && until_restart == 0) {   // RSTn: byte-align, skip the marker, reset the predictions
                        b.reset();
                        while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) b.p++;
                        if (b.p + 1 < b.end) b.p += 2;
                        for (Comp *c : sc) c->pred = 0;
                        until_restart = restart;
                    }
                    until_restart--;
                    for (Comp *c : sc) {
                        const int bh = inter ? c->h : 1, bv = inter ? c->v : 1;
                        for (int by = 0; by < bv; by++)
                            for (int bx = 0; bx < bh; bx++) {
                                memset(coef, 0, sizeof coef);
                                const int t = decode_sym(b, hdc[c->td]);
                                if (t < 0 || t > 11) { err = "JPEG: corrupt DC code"; return false; }
                                c->pred += t ? extend(b.receive(t), t) : 0;
                                // the DC value of 8-bit data lies in [-2^11, 2^11): a prediction that runs away is a corrupt stream, and
                                // bounding it (and with it every product with a 16-bit quantiser) keeps the integer IDCT inside int
                                if (c->pred < -32768 || c->pred > 32767) { err = "JPEG: DC prediction out of range"; return false; }
                                const uint16_t *q = qt[c->tq];
                                // (dequantised coefficients of 8-bit data stay below 2^15; anything beyond 2^24 is a corrupt stream, and
                                // refusing it keeps every product of the integer IDCT inside its type)
                                auto dequant = [&](int v, int qv, int &dst) { const int64_t x = (int64_t)v * qv; dst = (int)x; return x >= -(1 << 24) && x <= (1 << 24); };
                                if (!dequant(c->pred, q[0], coef[0])) { err = "JPEG: coefficient out of range"; return false; }
                                for (int k = 1; k < 64;) {
                                    const int rs = decode_sym(b, hac[c->ta]);
                                    if (rs < 0) { err = "JPEG: corrupt AC code"; return false; }
                                    const int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) {
                                        if (r != 15) break;
                                        k += 16;
                                        continue;
                                    }
                                    k += r;
                                    if (k > 63) { err = "JPEG: corrupt block"; return false; }
                                    if (!dequant(extend(b.receive(sz), sz), q[kZigzag[k]], coef[kZigzag[k]])) { err = "JPEG: coefficient out of range"; return false; }
                                    k++;
                                }
                                const int col = (inter ? mx * c->h + bx : mx), row = (inter ? my * c->v + by : my);
                                idct_islow(coef, &c->plane[((size_t)row * 8) * (c->wblocks * 8) + (size_t)col * 8], c->wblocks * 8);
                            }
                    }
                }
            // the next marker (the bit reader stopped in front of it, or inside the last data bytes)
            size_t q = b.p - d.data();
            while (q + 1 < d.size() && !(d[q] == 0xFF && d[q + 1] != 0x00 && !(d[q + 1] >= 0xD0 && d[q + 1] <= 0xD7))) q++;
            pos = q;
            bool all = true;   // a file may spread its components over several scans: done when EOI comes
            (void)all;
            continue;
        }
        default: break;   // APPn, COM, DNL, ...: skipped
        }
        pos += len;
    }
    if (!have_sof) { err = "JPEG: no frame header"; return false; }
    for (const Comp &c : comps)   // a truncated file (no scan, or a component no scan covered) is a camera error, not a black frame
        if (!c.scanned) { err = "JPEG: no image data"; return false; }
    if (progressive)   // every scan is in: the coefficient planes through the quantiser and the IDCT, block by block
        for (Comp &c : comps)
            for (int row = 0; row < c.hblocks; row++)
                for (int col = 0; col < c.wblocks; col++) {
                    const int *cf = &c.coef[((size_t)row * c.wblocks + col) * 64];
                    int dq[64];
                    for (int k = 0; k < 64; k++) {
                        const int64_t x = (int64_t)cf[k] * c.q[k];
                        if (x < -(1 << 24) || x > (1 << 24)) { err = "JPEG: coefficient out of range"; return false; }
                        dq[k] = (int)x;
                    }
                    idct_islow(dq, &c.plane[((size_t)row * 8) * (c.wblocks * 8) + (size_t)col * 8], c.wblocks * 8);
                }
    // ---- upsampling (jdsample.c) and colour conversion (jdcolor.c) ----
    const int W = (int)w, H = (int)h;
    out.assign((size_t)W * H, RGB24Pixel{0, 0, 0});
    std::vector<std::vector<unsigned char>> full(comps.size());
    for (size_t ci = 0; ci < comps.size(); ci++) {
        const Comp &c = comps[ci];
        const int stride = c.wblocks * 8;
        const int hx = hmax / c.h, vx = vmax / c.v;
        std::vector<unsigned char> &f = full[ci];
        const int fw = c.cw * hx;                      // >= W
        f.assign((size_t)fw * (c.ch * vx), 0);
        auto rowp = [&](int r) { r = r < 0 ? 0 : (r >= c.ch ? c.ch - 1 : r); return &c.plane[(size_t)r * stride]; };   // the image's edge rows repeat (jdmainct.c context rows)
        for (int r = 0; r < c.ch; r++) {
            if (hx == 1 && vx == 1) memcpy(&f[(size_t)r * fw], rowp(r), c.cw);
            else if (hx == 2 && vx == 1) up_h2_row(rowp(r), c.cw, &f[(size_t)r * fw]);
            else if (hx == 2 && vx == 2) {
                up_h2v2_row(rowp(r), rowp(r - 1), c.cw, &f[(size_t)(2 * r) * fw]);
                up_h2v2_row(rowp(r), rowp(r + 1), c.cw, &f[(size_t)(2 * r + 1) * fw]);
            } else {   // hx == 1, vx == 2
                up_v2_row(rowp(r), rowp(r - 1), c.cw, 1, &f[(size_t)(2 * r) * fw]);
                up_v2_row(rowp(r), rowp(r + 1), c.cw, 2, &f[(size_t)(2 * r + 1) * fw]);
            }
        }
    }
    if (comps.size() == 1) {
        const int fw = comps[0].cw;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const unsigned char g = full[0][(size_t)y * fw + x];
                out[(size_t)y * W + x] = RGB24Pixel{g, g, g};
            }
        if (mono) *mono = true;
        return true;
    }
    const bool rgb_direct = adobe_transform == 0 || (comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
    const int fw0 = comps[0].cw * (hmax / comps[0].h), fw1 = comps[1].cw * (hmax / comps[1].h), fw2 = comps[2].cw * (hmax / comps[2].h);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int Y = full[0][(size_t)y * fw0 + x], cb = full[1][(size_t)y * fw1 + x], cr = full[2][(size_t)y * fw2 + x];
            RGB24Pixel px;
            if (rgb_direct) {
                px = RGB24Pixel{(unsigned char)Y, (unsigned char)cb, (unsigned char)cr};
            } else {   // ycc_rgb_convert: 16.16 fixed point, ONE_HALF rounding, arithmetic right shifts
                const int64_t half = 1 << 15;
                const int r = Y + (int)((91881 * (int64_t)(cr - 128) + half) >> 16);
                const int g = Y + (int)((-22554 * (int64_t)(cb - 128) + half + -46802 * (int64_t)(cr - 128)) >> 16);
                const int bl = Y + (int)((116130 * (int64_t)(cb - 128) + half) >> 16);
                px = RGB24Pixel{clamp8(r), clamp8(g), clamp8(bl)};
            }
            out[(size_t)y * W + x] = px;
        }
    return true;
}

}  // namespace rebvo
