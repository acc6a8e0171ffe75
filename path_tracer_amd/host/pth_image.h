// pth_image.h -- decoded 8-bit images for the texture path (graphics.rs:383-516) and the PNG
// writer behind save_render_texture_to_file (renderer.rs:805-815).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace pth {

struct Image {
    uint32_t width = 0, height = 0;
    int channels = 0;                 // 3 = R8G8B8, 4 = R8G8B8A8 (gltf::image::Format)
    std::vector<uint8_t> pixels;      // row 0 first
};

// PNG (8-bit RGB / RGBA / palette, non-interlaced) or baseline JPEG (YCbCr / RGB).
bool decode_image_file(const std::string &path, Image &img, std::string &err);
bool encode_png_rgb8(const std::string &path, uint32_t w, uint32_t h, const uint8_t *rgb, std::string &err);

}  // namespace pth
