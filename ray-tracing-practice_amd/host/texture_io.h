// texture_io.h — host-side texture loading into the CpuTexture layout of the reference
// (include/materials.h:14-18): float RGBA, row-major, top row first.
//
// The reference decodes with the vendored stb_image (stbi_loadf(...,4), src/main.cu:52-60), which
// converts 8-bit channels to linear floats as pow(c/255, 2.2) and alpha as a/255
// (stb_image.h:1858-1884).  stb_image is third-party code that is not part of this repository;
// this loader reads baseline JPEG (own decoder, jpeg_decoder.cpp, checked byte for byte against
// the reference's decoder on the reference's floor.jpg), binary PPM (P6, 8-bit) and PFM (PF) files
// and applies the same 8-bit→float rule.  Anything else fails exactly like a failed stbi_loadf:
// message on stderr, untextured material (src/main.cu:55-58).
#pragma once
#include <string>
#include <vector>

namespace rtp {

struct TextureImage {
    std::vector<float> rgba;
    int width = 0, height = 0;
};

bool load_texture(const std::string &path, TextureImage &out);

// 8-bit sRGB-ish → linear float RGBA, the stbi_loadf rule (gamma 2.2, scale 1).
void ldr_to_linear_rgba(const unsigned char *rgb, int width, int height, int channels, TextureImage &out);

// Deterministic procedural texture for tests/benchmarks (two-tone checker with a gradient).
void make_checker_texture(int size, TextureImage &out);

}  // namespace rtp
