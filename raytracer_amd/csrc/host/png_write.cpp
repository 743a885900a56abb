// Minimal RGB8 PNG encoder (zlib deflate + CRC32), standing in for the png crate use at main.rs:192-200:
// colour type RGB, bit depth 8, row 0 = top of the image, no interlace, filter 0 on every row.
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt {

namespace {
void put_u32(std::vector<unsigned char>& v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)), v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)), v.push_back((unsigned char)x);
}
void chunk(std::vector<unsigned char>& out, const char type[4], const unsigned char* data, size_t len) {
    put_u32(out, (uint32_t)len);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), data, data + len);
    uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(len + 4));
    put_u32(out, crc);
}
}  // namespace

void write_png_rgb8(const std::string& path, const uint8_t* rgb, uint32_t width, uint32_t height) {
    if (width == 0 || height == 0) throw std::runtime_error("could not write image: zero-sized image");
    size_t stride = (size_t)width * 3;
    std::vector<unsigned char> raw((stride + 1) * height);
    for (uint32_t y = 0; y < height; ++y) {
        raw[(stride + 1) * y] = 0;  // filter type None
        std::memcpy(&raw[(stride + 1) * y + 1], rgb + stride * y, stride);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK)
        throw std::runtime_error("could not write image: deflate failed");

    std::vector<unsigned char> out;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.insert(out.end(), sig, sig + 8);
    std::vector<unsigned char> ihdr;
    put_u32(ihdr, width), put_u32(ihdr, height);
    ihdr.push_back(8), ihdr.push_back(2), ihdr.push_back(0), ihdr.push_back(0), ihdr.push_back(0);
    chunk(out, "IHDR", ihdr.data(), ihdr.size());
    chunk(out, "IDAT", z.data(), zlen);
    chunk(out, "IEND", nullptr, 0);

    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("could not create '" + path + "': " + std::strerror(errno));
    size_t w = std::fwrite(out.data(), 1, out.size(), f);
    if (std::fclose(f) != 0 || w != out.size()) throw std::runtime_error("could not write image");
}

}  // namespace rt
