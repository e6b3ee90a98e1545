// png_writer.h — minimal RGB8 PNG encoder (stored deflate blocks, no external library).
#pragma once
#include <cstdint>
#include <string>
namespace rtp {
bool write_png_rgb8(const std::string &path, int width, int height, const uint8_t *rgb);
}
