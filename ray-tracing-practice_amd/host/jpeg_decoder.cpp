// jpeg_decoder.cpp — baseline (sequential, Huffman, 8-bit) JPEG decoder for the host texture loader.
//
// PROVENANCE.  This file is a restatement of the JPEG path of stb_image v2.30 (Sean Barrett et al.; public domain /
// MIT, dual-licensed — https://github.com/nothings/stb), the decoder the reference vendors as include/stb_image.h and
// calls as stbi_loadf(path, &w, &h, &c, 4) (src/main.cu:52-60).  It is NOT an independent design: a texture decoded by
// a different JPEG decoder (other IDCT rounding, other chroma upsampling) gives different texels and therefore a
// different image than the reference's, so the arithmetic follows stb_image's step for step and several tables and
// idioms are stb_image's own: the de-zigzag table with its 15 padding entries, the maxcode/delta canonical-Huffman
// layout with a 9-bit fast table, the bit-buffer refill ("grow") and the rotate-based extend-receive, the 12-bit
// fixed-point "islow"-style IDCT applied to coefficients dequantised at entropy-decode time (its constants and
// temporaries), the triangle ("fancy") h2/v2/hv2 chroma upsamplers with their div4/div16 rounding, and the 20-bit
// fixed-point YCbCr→RGB with the `& 0xffff0000` step.  SURVEY.md lists stb itself as third-party and out of scope;
// what is restated here is only the baseline path the reference's floor.jpg needs, reorganised around std::vector
// and a Decoder struct.  tests/test_oracle_pins.py checks it texel for texel against the reference's own vendored
// stb_image.h on the reference's floor.jpg (build container only).
//
// Supported: SOF0/SOF1 (baseline / extended sequential, 8-bit), 1 or 3 components, sampling
// factors 1 or 2, restart intervals.  Progressive, arithmetic-coded, 12-bit and CMYK files are
// rejected (load fails like a failed stbi_loadf: the material stays untextured).
// Unlike stb_image, images larger than kMaxTextureDim per side or kMaxTexturePixels in all are rejected before
// anything is allocated (a 65535 x 65535 header would otherwise ask for 12.9 GB of RGB plus 68 GB of float RGBA).
#include "jpeg_decoder.h"

#include <cstdint>
#include <cstring>
#include <fstream>
#include <vector>

namespace rtp {
namespace {

constexpr int kMaxTextureDim = 16384;
constexpr int64_t kMaxTexturePixels = int64_t(1) << 26;      // 64 Mpixel = 1 GiB of float RGBA on the device

const uint8_t kDezigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                    6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                    39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman {
    uint8_t size[257];
    uint16_t code[256];
    uint8_t values[256];
    unsigned maxcode[18];
    int delta[17];
    bool build(const int *count) {
        int k = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < count[i]; ++j) {
                if (k >= 256) return false;
                size[k++] = static_cast<uint8_t>(i + 1);
            }
        size[k] = 0;
        unsigned c = 0;
        k = 0;
        for (int j = 1; j <= 16; ++j) {
            delta[j] = k - static_cast<int>(c);
            if (size[k] == j) {
                while (size[k] == j) code[k++] = static_cast<uint16_t>(c++);
                if (c - 1 >= (1u << j)) return false;
            }
            maxcode[j] = c << (16 - j);
            c <<= 1;
        }
        maxcode[17] = 0xffffffff;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dc_pred = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;
    std::vector<uint8_t> data;
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t buffer = 0;
    int bits = 0;
    int marker = -1;      // marker hit while filling (0xD0..0xD7 restart, others end the scan)
    bool nomore = false;
    void reset() { buffer = 0; bits = 0; marker = -1; nomore = false; }
    void grow() {
        do {
            unsigned b = nomore ? 0 : (p < end ? *p++ : 0);
            if (b == 0xff) {
                unsigned c = p < end ? *p++ : 0;
                while (c == 0xff) c = p < end ? *p++ : 0;
                if (c != 0) {
                    marker = static_cast<int>(c);
                    nomore = true;
                    return;
                }
            }
            buffer |= b << (24 - bits);
            bits += 8;
        } while (bits <= 24);
    }
    int decode(const Huffman &h) {
        if (bits < 16) grow();
        const unsigned temp = buffer >> 16;
        int k = 1;
        while (k <= 16 && temp >= h.maxcode[k]) ++k;
        if (k == 17) { bits -= 16; return -1; }
        if (k > bits) return -1;
        const int c = static_cast<int>((buffer >> (32 - k)) & ((1u << k) - 1)) + h.delta[k];
        if (c < 0 || c >= 256) return -1;
        bits -= k;
        buffer <<= k;
        return h.values[c];
    }
    // receive n bits and sign-extend JPEG style
    int extend_receive(int n) {
        if (n == 0) return 0;
        if (bits < n) grow();
        const int sgn = static_cast<int32_t>(buffer) >> 31;     // 0 or -1 from the top bit
        const unsigned k = (buffer << n) | (buffer >> (32 - n));   // rotate left by n
        const unsigned mask = (1u << n) - 1;
        buffer = k & ~mask;
        bits -= n;
        const int bias = -(1 << n) + 1;       // (-1 << n) + 1
        return static_cast<int>(k & mask) + (bias & ~sgn);
    }
};

inline uint8_t clamp8(int x) {
    if (static_cast<unsigned>(x) > 255) return x < 0 ? 0 : 255;
    return static_cast<uint8_t>(x);
}

#define F2F(x) (static_cast<int>(((x) * 4096 + 0.5)))
#define FSH(x) ((x) * 4096)
#define IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                          \
    int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                              \
    p2 = s2; p3 = s6;                                                                    \
    p1 = (p2 + p3) * F2F(0.5411961f);                                                    \
    t2 = p1 + p3 * F2F(-1.847759065f);                                                   \
    t3 = p1 + p2 * F2F(0.765366865f);                                                    \
    p2 = s0; p3 = s4;                                                                    \
    t0 = FSH(p2 + p3); t1 = FSH(p2 - p3);                                                \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                              \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                  \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                              \
    p5 = (p3 + p4) * F2F(1.175875602f);                                                  \
    t0 = t0 * F2F(0.298631336f); t1 = t1 * F2F(2.053119869f);                            \
    t2 = t2 * F2F(3.072711026f); t3 = t3 * F2F(1.501321110f);                            \
    p1 = p5 + p1 * F2F(-0.899976223f); p2 = p5 + p2 * F2F(-2.562915447f);                \
    p3 = p3 * F2F(-1.961570560f); p4 = p4 * F2F(-0.390180644f);                          \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

void idct_block(uint8_t *out, int out_stride, const short data[64]) {
    int val[64];
    for (int i = 0; i < 8; ++i) {
        const short *d = data + i;
        int *v = val + i;
        if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
            const int dcterm = d[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
        } else {
            IDCT_1D(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    for (int i = 0; i < 8; ++i) {
        const int *v = val + i * 8;
        uint8_t *o = out + i * out_stride;
        IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
        o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
        o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
        o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
    }
}

struct Decoder {
    const uint8_t *base = nullptr, *p = nullptr, *end = nullptr;
    Huffman huff_dc[4], huff_ac[4];
    bool have_dc[4] = {}, have_ac[4] = {};
    uint16_t dequant[4][64];
    int img_x = 0, img_y = 0, img_n = 0;
    Component comp[4];
    int h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0, mcu_w = 0, mcu_h = 0;
    int restart_interval = 0;
    int scan_n = 0, order[4];
    BitReader br;

    int get8() { return p < end ? *p++ : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }

    bool decode_block(short data[64], const Huffman &hdc, const Huffman &hac, int b, const uint16_t *dq) {
        if (br.bits < 16) br.grow();
        const int t = br.decode(hdc);
        if (t < 0 || t > 15) return false;
        std::memset(data, 0, 64 * sizeof(short));
        const int diff = t ? br.extend_receive(t) : 0;
        const int dc = comp[b].dc_pred + diff;
        comp[b].dc_pred = dc;
        data[0] = static_cast<short>(dc * dq[0]);
        int k = 1;
        do {
            const int rs = br.decode(hac);
            if (rs < 0) return false;
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break;   // end of block
                k += 16;
            } else {
                k += r;
                const unsigned zig = kDezigzag[k++];
                data[zig] = static_cast<short>(br.extend_receive(s) * dq[zig]);
            }
        } while (k < 64);
        return true;
    }

    bool parse_entropy() {
        br.p = p;
        br.end = end;
        br.reset();
        for (int i = 0; i < 4; ++i) comp[i].dc_pred = 0;
        int todo = restart_interval ? restart_interval : 0x7fffffff;
        short data[64];
        auto restart = [&]() {
            if (br.bits < 24) br.grow();
            if (br.marker >= 0xd0 && br.marker <= 0xd7) {
                br.reset();
                for (int i = 0; i < 4; ++i) comp[i].dc_pred = 0;
                todo = restart_interval ? restart_interval : 0x7fffffff;
                return true;
            }
            return false;       // some other marker: the scan is over
        };
        if (scan_n == 1) {
            const int n = order[0];
            const int w = (comp[n].x + 7) >> 3, h = (comp[n].y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    if (!decode_block(data, huff_dc[comp[n].hd], huff_ac[comp[n].ha], n, dequant[comp[n].tq])) return false;
                    idct_block(comp[n].data.data() + comp[n].w2 * j * 8 + i * 8, comp[n].w2, data);
                    if (--todo <= 0 && !restart()) { p = br.p; return true; }
                }
        } else {
            for (int j = 0; j < mcu_y; ++j)
                for (int i = 0; i < mcu_x; ++i) {
                    for (int k = 0; k < scan_n; ++k) {
                        const int n = order[k];
                        for (int y = 0; y < comp[n].v; ++y)
                            for (int x = 0; x < comp[n].h; ++x) {
                                const int x2 = (i * comp[n].h + x) * 8, y2 = (j * comp[n].v + y) * 8;
                                if (!decode_block(data, huff_dc[comp[n].hd], huff_ac[comp[n].ha], n, dequant[comp[n].tq])) return false;
                                idct_block(comp[n].data.data() + comp[n].w2 * y2 + x2, comp[n].w2, data);
                            }
                    }
                    if (--todo <= 0 && !restart()) { p = br.p; return true; }
                }
        }
        p = br.p;
        return true;
    }

    bool process_marker(int m) {
        switch (m) {
            case 0xDD:
                if (get16() != 4) return false;
                restart_interval = get16();
                return true;
            case 0xDB: {
                int L = get16() - 2;
                while (L > 0) {
                    const int q = get8(), prec = q >> 4, t = q & 15;
                    if ((prec != 0 && prec != 1) || t > 3) return false;
                    for (int i = 0; i < 64; ++i) dequant[t][kDezigzag[i]] = static_cast<uint16_t>(prec ? get16() : get8());
                    L -= prec ? 129 : 65;
                }
                return L == 0;
            }
            case 0xC4: {
                int L = get16() - 2;
                while (L > 0) {
                    int sizes[16], n = 0;
                    const int q = get8(), tc = q >> 4, th = q & 15;
                    if (tc > 1 || th > 3) return false;
                    for (int i = 0; i < 16; ++i) { sizes[i] = get8(); n += sizes[i]; }
                    if (n > 256) return false;
                    L -= 17;
                    Huffman &h = tc == 0 ? huff_dc[th] : huff_ac[th];
                    if (!h.build(sizes)) return false;
                    for (int i = 0; i < n; ++i) h.values[i] = static_cast<uint8_t>(get8());
                    (tc == 0 ? have_dc : have_ac)[th] = true;
                    L -= n;
                }
                return L == 0;
            }
            default:
                if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
                    const int L = get16();
                    if (L < 2) return false;
                    p += L - 2;
                    return p <= end;
                }
                return false;
        }
    }

    bool process_frame_header() {
        const int Lf = get16();
        if (Lf < 11) return false;
        if (get8() != 8) return false;       // 8-bit only
        img_y = get16();
        img_x = get16();
        if (img_x <= 0 || img_y <= 0) return false;
        if (img_x > kMaxTextureDim || img_y > kMaxTextureDim || static_cast<int64_t>(img_x) * img_y > kMaxTexturePixels) return false;
        img_n = get8();
        if (img_n != 3 && img_n != 1) return false;
        if (Lf != 8 + 3 * img_n) return false;
        for (int i = 0; i < img_n; ++i) {
            comp[i].id = get8();
            const int q = get8();
            comp[i].h = q >> 4;
            comp[i].v = q & 15;
            if (comp[i].h < 1 || comp[i].h > 2 || comp[i].v < 1 || comp[i].v > 2) return false;
            comp[i].tq = get8();
            if (comp[i].tq > 3) return false;
            if (comp[i].h > h_max) h_max = comp[i].h;
            if (comp[i].v > v_max) v_max = comp[i].v;
        }
        mcu_w = h_max * 8;
        mcu_h = v_max * 8;
        mcu_x = (img_x + mcu_w - 1) / mcu_w;
        mcu_y = (img_y + mcu_h - 1) / mcu_h;
        for (int i = 0; i < img_n; ++i) {
            comp[i].x = (img_x * comp[i].h + h_max - 1) / h_max;
            comp[i].y = (img_y * comp[i].v + v_max - 1) / v_max;
            comp[i].w2 = mcu_x * comp[i].h * 8;
            comp[i].h2 = mcu_y * comp[i].v * 8;
            comp[i].data.assign(static_cast<size_t>(comp[i].w2) * comp[i].h2, 0);
        }
        return true;
    }

    bool process_scan_header() {
        const int Ls = get16();
        scan_n = get8();
        if (scan_n < 1 || scan_n > img_n || Ls != 6 + 2 * scan_n) return false;
        for (int i = 0; i < scan_n; ++i) {
            const int id = get8(), q = get8();
            int which = 0;
            for (; which < img_n; ++which)
                if (comp[which].id == id) break;
            if (which == img_n) return false;
            comp[which].hd = q >> 4;
            comp[which].ha = q & 15;
            if (comp[which].hd > 3 || comp[which].ha > 3 || !have_dc[comp[which].hd] || !have_ac[comp[which].ha]) return false;
            order[i] = which;
        }
        const int ss = get8();
        get8();     // spectral end: 63 for sequential files
        const int aa = get8();
        return ss == 0 && aa == 0;
    }

    int next_marker() {
        if (br.marker >= 0) { const int m = br.marker; br.marker = -1; return m; }
        int x = get8();
        if (x != 0xff) return -1;
        while (x == 0xff) x = get8();
        return x;
    }

    bool decode(const std::vector<uint8_t> &file) {
        base = p = file.data();
        end = p + file.size();
        if (get8() != 0xff || get8() != 0xd8) return false;
        int m = next_marker();
        while (!(m == 0xC0 || m == 0xC1)) {
            if (m == 0xC2 || m < 0) return false;            // progressive or garbage
            if (!process_marker(m)) return false;
            m = next_marker();
            while (m < 0) {
                if (p >= end) return false;
                m = next_marker();
            }
        }
        if (!process_frame_header()) return false;
        m = next_marker();
        while (m != 0xD9) {
            if (m == 0xDA) {
                if (!process_scan_header()) return false;
                if (!parse_entropy()) return false;
                if (br.marker < 0) {
                    // look for the next marker after the entropy-coded data
                    while (p < end) {
                        if (*p++ == 0xff) {
                            while (p < end && *p == 0xff) ++p;
                            if (p < end && *p != 0) { br.marker = *p++; break; }
                        }
                    }
                }
            } else if (m < 0) {
                if (p >= end) break;
            } else if (!process_marker(m)) {
                return false;
            }
            m = next_marker();
            if (m < 0 && p >= end) break;
        }
        return true;
    }
};

inline uint8_t div4(int x) { return static_cast<uint8_t>(x >> 2); }
inline uint8_t div16(int x) { return static_cast<uint8_t>(x >> 4); }

// chroma upsamplers (near = row being centred on, far = the neighbouring row, w = low-res width)
uint8_t *resample_1(uint8_t *, uint8_t *near, uint8_t *, int, int) { return near; }
uint8_t *resample_v2(uint8_t *out, uint8_t *near, uint8_t *far, int w, int) {
    for (int i = 0; i < w; ++i) out[i] = div4(3 * near[i] + far[i] + 2);
    return out;
}
uint8_t *resample_h2(uint8_t *out, uint8_t *in, uint8_t *, int w, int) {
    if (w == 1) { out[0] = out[1] = in[0]; return out; }
    out[0] = in[0];
    out[1] = div4(in[0] * 3 + in[1] + 2);
    int i;
    for (i = 1; i < w - 1; ++i) {
        const int n = 3 * in[i] + 2;
        out[i * 2 + 0] = div4(n + in[i - 1]);
        out[i * 2 + 1] = div4(n + in[i + 1]);
    }
    out[i * 2 + 0] = div4(in[w - 2] * 3 + in[w - 1] + 2);
    out[i * 2 + 1] = in[w - 1];
    return out;
}
uint8_t *resample_hv2(uint8_t *out, uint8_t *near, uint8_t *far, int w, int) {
    if (w == 1) { out[0] = out[1] = div4(3 * near[0] + far[0] + 2); return out; }
    int t1 = 3 * near[0] + far[0];
    out[0] = div4(t1 + 2);
    for (int i = 1; i < w; ++i) {
        const int t0 = t1;
        t1 = 3 * near[i] + far[i];
        out[i * 2 - 1] = div16(3 * t0 + t1 + 8);
        out[i * 2] = div16(3 * t1 + t0 + 8);
    }
    out[w * 2 - 1] = div4(t1 + 2);
    return out;
}

#define FLOAT2FIXED(x) ((static_cast<int>((x) * 4096.0f + 0.5f)) << 8)

}  // namespace

bool decode_jpeg_rgb8(const std::string &path, int &width, int &height, std::vector<uint8_t> &rgb) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (file.size() < 4) return false;
    Decoder d;
    if (!d.decode(file)) return false;
    width = d.img_x;
    height = d.img_y;
    const int n = d.img_n;
    rgb.assign(static_cast<size_t>(width) * height * 3, 0);

    typedef uint8_t *(*Resampler)(uint8_t *, uint8_t *, uint8_t *, int, int);
    struct Resample {
        Resampler fn;
        uint8_t *line0, *line1;
        int hs, vs, w_lores, ystep, ypos;
        std::vector<uint8_t> linebuf;
    } rs[3];
    for (int k = 0; k < n; ++k) {
        Resample &r = rs[k];
        r.hs = d.h_max / d.comp[k].h;
        r.vs = d.v_max / d.comp[k].v;
        r.ystep = r.vs >> 1;
        r.w_lores = (width + r.hs - 1) / r.hs;
        r.ypos = 0;
        r.line0 = r.line1 = d.comp[k].data.data();
        r.linebuf.assign(static_cast<size_t>(width) + 3, 0);
        if (r.hs == 1 && r.vs == 1) r.fn = resample_1;
        else if (r.hs == 1 && r.vs == 2) r.fn = resample_v2;
        else if (r.hs == 2 && r.vs == 1) r.fn = resample_h2;
        else r.fn = resample_hv2;
    }
    for (int j = 0; j < height; ++j) {
        uint8_t *coutput[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < n; ++k) {
            Resample &r = rs[k];
            const bool y_bot = r.ystep >= (r.vs >> 1);
            coutput[k] = r.fn(r.linebuf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < d.comp[k].y) r.line1 += d.comp[k].w2;
            }
        }
        uint8_t *out = &rgb[static_cast<size_t>(j) * width * 3];
        if (n == 3) {
            const uint8_t *y = coutput[0], *pcb = coutput[1], *pcr = coutput[2];
            for (int i = 0; i < width; ++i) {
                const int y_fixed = (y[i] << 20) + (1 << 19);
                const int cr = pcr[i] - 128, cb = pcb[i] - 128;
                int r = y_fixed + cr * FLOAT2FIXED(1.40200f);
                int g = y_fixed + (cr * -FLOAT2FIXED(0.71414f)) + ((cb * -FLOAT2FIXED(0.34414f)) & 0xffff0000);
                int b = y_fixed + cb * FLOAT2FIXED(1.77200f);
                r >>= 20; g >>= 20; b >>= 20;
                out[3 * i] = clamp8(r);
                out[3 * i + 1] = clamp8(g);
                out[3 * i + 2] = clamp8(b);
            }
        } else {
            for (int i = 0; i < width; ++i) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = coutput[0][i];
        }
    }
    return true;
}

}  // namespace rtp
