#include "texture_io.h"

#include "jpeg_decoder.h"

#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>

namespace rtp {

void ldr_to_linear_rgba(const unsigned char *px, int width, int height, int channels, TextureImage &out) {
    out.width = width;
    out.height = height;
    out.rgba.resize(static_cast<size_t>(width) * height * 4);
    // 256-entry table of pow(c/255.0f, 2.2f) evaluated like stb does: float divide, double pow.
    float lut[256];
    for (int c = 0; c < 256; ++c) lut[c] = static_cast<float>(pow(c / 255.0f, 2.2f) * 1.0f);
    for (size_t i = 0; i < static_cast<size_t>(width) * height; ++i) {
        for (int k = 0; k < 3; ++k) out.rgba[i * 4 + k] = lut[px[i * channels + (channels >= 3 ? k : 0)]];
        out.rgba[i * 4 + 3] = channels == 4 ? px[i * 4 + 3] / 255.0f : 1.0f;
    }
}

namespace {
bool read_token(std::istream &in, std::string &tok) {
    tok.clear();
    int c;
    while ((c = in.get()) != EOF) {
        if (c == '#') { while ((c = in.get()) != EOF && c != '\n') {} continue; }
        if (!isspace(c)) { tok.push_back(static_cast<char>(c)); break; }
    }
    while ((c = in.peek()) != EOF && !isspace(c)) tok.push_back(static_cast<char>(in.get()));
    return !tok.empty();
}
}  // namespace

bool load_texture(const std::string &path, TextureImage &out) {
    {   // JPEG (SOI marker FF D8): the reference's floor textures are JPEGs
        std::ifstream probe(path, std::ios::binary);
        unsigned char magic[2] = {0, 0};
        probe.read(reinterpret_cast<char *>(magic), 2);
        if (probe.gcount() == 2 && magic[0] == 0xFF && magic[1] == 0xD8) {
            int w = 0, h = 0;
            std::vector<uint8_t> rgb;
            if (!decode_jpeg_rgb8(path, w, h, rgb)) {
                std::cerr << "Failed to load texture: " << path << std::endl;
                return false;
            }
            ldr_to_linear_rgba(rgb.data(), w, h, 3, out);
            return true;
        }
    }
    std::ifstream in(path, std::ios::binary);
    std::string magic, tw, th, tmax;
    if (!in || !read_token(in, magic) || (magic != "P6" && magic != "PF") || !read_token(in, tw) ||
        !read_token(in, th) || !read_token(in, tmax)) {
        std::cerr << "Failed to load texture: " << path << std::endl;
        return false;
    }
    in.get();  // single whitespace after the header
    const int w = atoi(tw.c_str()), h = atoi(th.c_str());
    // same caps as the JPEG path: a hostile header must not size a multi-gigabyte allocation
    if (w <= 0 || h <= 0 || w > 16384 || h > 16384 || static_cast<long long>(w) * h > (1ll << 26)) {
        std::cerr << "Failed to load texture: " << path << std::endl;
        return false;
    }
    const size_t n = static_cast<size_t>(w) * h;
    if (magic == "P6") {
        std::vector<unsigned char> buf(n * 3);
        in.read(reinterpret_cast<char *>(buf.data()), static_cast<std::streamsize>(buf.size()));
        if (in.gcount() != static_cast<std::streamsize>(buf.size()) || atoi(tmax.c_str()) != 255) {
            std::cerr << "Failed to load texture: " << path << std::endl;
            return false;
        }
        ldr_to_linear_rgba(buf.data(), w, h, 3, out);
        return true;
    }
    // PF: little-endian float RGB when the scale token is negative; rows bottom to top.
    std::vector<float> buf(n * 3);
    in.read(reinterpret_cast<char *>(buf.data()), static_cast<std::streamsize>(buf.size() * 4));
    if (in.gcount() != static_cast<std::streamsize>(buf.size() * 4) || atof(tmax.c_str()) >= 0) {
        std::cerr << "Failed to load texture: " << path << std::endl;
        return false;
    }
    out.width = w;
    out.height = h;
    out.rgba.resize(n * 4);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float *s = &buf[(static_cast<size_t>(h - 1 - y) * w + x) * 3];
            float *d = &out.rgba[(static_cast<size_t>(y) * w + x) * 4];
            d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = 1.0f;
        }
    return true;
}

void make_checker_texture(int size, TextureImage &out) {
    std::vector<unsigned char> px(static_cast<size_t>(size) * size * 3);
    for (int y = 0; y < size; ++y)
        for (int x = 0; x < size; ++x) {
            const bool dark = (((x * 16) / size) + ((y * 16) / size)) & 1;
            unsigned char *p = &px[(static_cast<size_t>(y) * size + x) * 3];
            p[0] = static_cast<unsigned char>(dark ? 60 : 200 + (x * 55) / size);
            p[1] = static_cast<unsigned char>(dark ? 70 : 190);
            p[2] = static_cast<unsigned char>(dark ? 90 + (y * 100) / size : 170);
        }
    ldr_to_linear_rgba(px.data(), size, size, 3, out);
}

}  // namespace rtp
