// jpeg_decoder.h — baseline JPEG → RGB8 for the host texture loader (see jpeg_decoder.cpp).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
namespace rtp {
// Returns false for anything that is not a baseline / extended-sequential 8-bit JPEG with 1 or 3
// components and sampling factors of 1 or 2.  rgb: height rows of width RGB triples, top row first.
bool decode_jpeg_rgb8(const std::string &path, int &width, int &height, std::vector<uint8_t> &rgb);
}
