// mrt_image_io.cpp — writers for the two lossless formats the reference's CLI is used with
// (`img.save(&filename)`, src/cli.rs:168,174: the README renders are .png, its example command line writes .ppm).
// PNG is written with stored (uncompressed) deflate blocks: valid for every decoder, no zlib dependency.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include <errno.h>

#include "../../include/mrt.h"

int mrt_internal_fail(int code, const char *msg);     // mrt_api.cpp: sets mrt_last_error / mrt_last_status
void mrt_internal_ok();

namespace {

int failf(int code, const std::string &msg) { return mrt_internal_fail(code, msg.c_str()); }

struct CrcTable {
    uint32_t t[256];
    CrcTable() { for (uint32_t i = 0; i < 256; ++i) { uint32_t k = i; for (int j = 0; j < 8; ++j) k = (k & 1) ? 0xedb88320u ^ (k >> 1) : k >> 1; t[i] = k; } }
};
uint32_t crc32(const uint8_t *p, size_t n, uint32_t c = 0xffffffffu)
{
    static const CrcTable table;          // thread-safe one-time initialisation
    for (size_t i = 0; i < n; ++i) c = table.t[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c;
}
void be32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &data)
{
    be32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    be32(out, crc32(out.data() + at, out.size() - at) ^ 0xffffffffu);
}

}  // namespace

extern "C" int mrt_save_image(const char *path, const uint8_t *rgb8, uint32_t w, uint32_t h)
{
    if (!path || !rgb8 || !w || !h) return failf(MRT_ERR_ARG, "mrt_save_image: null path / pixels or empty image");
    // one IDAT chunk with a 32-bit length (and the 32-bit sizes of PPM readers): refuse what cannot be represented
    if ((unsigned long long)h * ((unsigned long long)w * 3u + 1u) > 0x7ff00000ull)
        return failf(MRT_ERR_LIMIT, "mrt_save_image: " + std::to_string(w) + "x" + std::to_string(h) + " exceeds the 2 GiB limit of this writer");
    const char *dot = strrchr(path, '.');
    const std::string ext = dot ? dot + 1 : "";
    std::vector<uint8_t> out;
    if (ext == "ppm" || ext == "PPM") {
        char hdr[64];
        const int n = snprintf(hdr, sizeof hdr, "P6\n%u %u\n255\n", w, h);
        out.insert(out.end(), hdr, hdr + n);
        out.insert(out.end(), rgb8, rgb8 + (size_t)w * h * 3);
    } else if (ext == "png" || ext == "PNG") {
        static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
        out.insert(out.end(), sig, sig + 8);
        std::vector<uint8_t> ihdr;
        be32(ihdr, w); be32(ihdr, h);
        ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB
        chunk(out, "IHDR", ihdr);
        std::vector<uint8_t> raw;                               // filter byte 0 + row
        raw.reserve((size_t)h * (w * 3 + 1));
        for (uint32_t y = 0; y < h; ++y) { raw.push_back(0); raw.insert(raw.end(), rgb8 + (size_t)y * w * 3, rgb8 + (size_t)(y + 1) * w * 3); }
        std::vector<uint8_t> z;
        z.push_back(0x78); z.push_back(0x01);                   // zlib header, no compression
        uint32_t a = 1, b = 0;                                  // adler32
        for (size_t pos = 0; pos < raw.size();) {
            const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
            z.push_back(pos + n == raw.size() ? 1 : 0);         // BFINAL, BTYPE = 00 (stored)
            z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
            z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
            for (size_t i = 0; i < n; ++i) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
            pos += n;
        }
        be32(z, (b << 16) | a);
        chunk(out, "IDAT", z);
        chunk(out, "IEND", {});
    } else {
        return failf(MRT_ERR_ARG, std::string("mrt_save_image: unsupported extension in '") + path + "' (.ppm and .png are written)");
    }
    FILE *f = fopen(path, "wb");
    if (!f) return failf(MRT_ERR_STATE, std::string("mrt_save_image: cannot open '") + path + "': " + strerror(errno));
    const size_t wr = fwrite(out.data(), 1, out.size(), f);
    const int werr = errno;
    if (fclose(f) != 0 || wr != out.size()) return failf(MRT_ERR_STATE, std::string("mrt_save_image: short write to '") + path + "': " + strerror(werr ? werr : errno));
    mrt_internal_ok();
    return MRT_OK;
}
