// host_ppm.cpp — [host] Canvas::write_to_file_simple (ch1/src/canvas.rs:86-109): ASCII PPM "P3",
// one text line per canvas row, components separated by single spaces, no 70-column
// wrapping, each component quantised by Color::scale (ch1/src/color.rs:100-114).
// And a PNG writer for Canvas::write_to_file (canvas.rs:80-84: `to_imgbuf().save(path)`, the `image` crate picks the
// codec by extension): PNG is lossless, so a decoder gives back exactly to_imgbuf's pixels whichever encoder wrote the
// file. This one has no compressor: the zlib stream inside IDAT consists of stored blocks (RFC 1950 / 1951 §3.2.4).
#include "rtc.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

namespace {

// Color::scale(component, 255): `(component * 255.0) as i32` is Rust's truncating,
// saturating float->int cast (NaN -> 0), then clamp to [0, scale].
int scale255(double component) {
    const double v = component * 255.0;
    long long q;
    if (v != v) q = 0;
    else if (v >= 2147483647.0) q = 2147483647LL;
    else if (v <= -2147483648.0) q = -2147483648LL;
    else q = static_cast<long long>(v); // truncates toward zero
    if (q < 0) return 0;
    if (q > 255) return 255;
    return static_cast<int>(q);
}

void append_uint(std::string &s, unsigned v) {
    char tmp[16];
    int n = 0;
    do { tmp[n++] = static_cast<char>('0' + v % 10); v /= 10; } while (v);
    while (n) s.push_back(tmp[--n]);
}

// `component(i)` = the i-th 8-bit component of the row-major canvas (3 per pixel)
template <class F> std::string encode_with(uint32_t width, uint32_t height, F component) {
    std::string s;
    s.reserve(static_cast<size_t>(width) * height * 12 + 32);
    s += "P3\n";
    append_uint(s, width);
    s.push_back(' ');
    append_uint(s, height);
    s += "\n255\n";
    for (uint32_t row = 0; row < height; ++row) {
        const size_t line = static_cast<size_t>(row) * width * 3;
        for (uint32_t col = 0; col < width; ++col) {
            if (col) s.push_back(' ');
            for (int ch = 0; ch < 3; ++ch) {
                if (ch) s.push_back(' ');
                append_uint(s, component(line + col * 3 + ch));
            }
        }
        s.push_back('\n');
    }
    return s;
}

std::string encode(const double *rgb, uint32_t width, uint32_t height) {
    return encode_with(width, height, [rgb](size_t i) { return static_cast<unsigned>(scale255(rgb[i])); });
}
// the same file from components that are ALREADY Color::scale'd (the device's 8-bit frame, rtc_render_rgb8)
std::string encode8(const uint8_t *rgb8, uint32_t width, uint32_t height) {
    return encode_with(width, height, [rgb8](size_t i) { return static_cast<unsigned>(rgb8[i]); });
}

rtc_status write_file(const char *path, const std::string &s) {
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return RTC_ERR_IO; // reference: panic!("Could not open output file ...") canvas.rs:87-91
    const size_t w = std::fwrite(s.data(), 1, s.size(), f);
    const int c = std::fclose(f);
    return (w == s.size() && c == 0) ? RTC_OK : RTC_ERR_IO;
}

size_t copy_out(const std::string &s, char *buf, size_t cap) {
    if (buf && cap) {
        const size_t n = s.size() < cap ? s.size() : cap;
        s.copy(buf, n);
        if (n < cap) buf[n] = 0;
    }
    return s.size();
}

// ---- PNG (ISO/IEC 15948): signature, IHDR, IDAT..., IEND; 8 bits per sample, colour type 2 (RGB) or 6 (RGBA), no filter
struct Crc32 {
    uint32_t table[256];
    Crc32() {
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
    }
    uint32_t update(uint32_t crc, const unsigned char *p, size_t n) const {
        for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
        return crc;
    }
};

void put_be32(unsigned char *p, uint32_t v) { p[0] = static_cast<unsigned char>(v >> 24); p[1] = static_cast<unsigned char>(v >> 16); p[2] = static_cast<unsigned char>(v >> 8); p[3] = static_cast<unsigned char>(v); }

// Emits the file through `sink(bytes, n)` (false = stop). One IDAT chunk per stored block of at most 65535 bytes.
template <class Sink> bool encode_png(const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels, Sink sink) {
    static const Crc32 crc;
    auto chunk = [&](const char type[4], const unsigned char *data, uint32_t n) {
        unsigned char head[8], tail[4];
        put_be32(head, n);
        std::memcpy(head + 4, type, 4);
        uint32_t c = crc.update(0xffffffffu, head + 4, 4);
        c = crc.update(c, data, n);
        put_be32(tail, c ^ 0xffffffffu);
        return sink(head, 8) && (n == 0 || sink(data, n)) && sink(tail, 4);
    };
    static const unsigned char signature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (!sink(signature, 8)) return false;
    unsigned char ihdr[13];
    put_be32(ihdr, width);
    put_be32(ihdr + 4, height);
    ihdr[8] = 8;                          // bit depth
    ihdr[9] = channels == 4 ? 6 : 2;      // RGBA : RGB
    ihdr[10] = ihdr[11] = ihdr[12] = 0;   // deflate, adaptive filtering (every row uses filter 0), no interlace
    if (!chunk("IHDR", ihdr, 13)) return false;
    // the raw stream: per row one filter byte (0) and the row; cut into stored blocks
    const size_t row = static_cast<size_t>(width) * channels, raw = (row + 1) * height;
    std::string block;
    block.reserve(65535 + 16);
    uint32_t a = 1, b = 0; // Adler-32 of the raw stream
    size_t done = 0;
    bool first = true;
    auto adler = [&](const unsigned char *p, size_t n) {
        while (n) {
            const size_t k = n < 5552 ? n : 5552;
            for (size_t i = 0; i < k; ++i) { a += p[i]; b += a; }
            a %= 65521u; b %= 65521u;
            p += k; n -= k;
        }
    };
    while (done < raw || first) {
        const size_t n = (raw - done) < 65535 ? (raw - done) : 65535;
        block.clear();
        if (first) { block.push_back(0x78); block.push_back(0x01); } // zlib header: deflate, 32 K window, no preset dictionary
        first = false;
        const bool last = done + n == raw;
        block.push_back(last ? 1 : 0);      // BFINAL, BTYPE = 00 (stored)
        block.push_back(static_cast<char>(n & 0xff));
        block.push_back(static_cast<char>(n >> 8));
        block.push_back(static_cast<char>(~n & 0xff));
        block.push_back(static_cast<char>((~n >> 8) & 0xff));
        const size_t payload_at = block.size();
        for (size_t k = 0; k < n;) { // bytes [done, done + n) of the raw stream
            const size_t pos = done + k, y = pos / (row + 1), x = pos % (row + 1);
            if (x == 0) { block.push_back(0); ++k; continue; }
            const size_t take = (row + 1 - x) < (n - k) ? (row + 1 - x) : (n - k);
            block.append(reinterpret_cast<const char *>(pixels + y * row + (x - 1)), take);
            k += take;
        }
        adler(reinterpret_cast<const unsigned char *>(block.data()) + payload_at, n);
        done += n;
        if (last) {
            unsigned char sum[4];
            put_be32(sum, (b << 16) | a);
            block.append(reinterpret_cast<const char *>(sum), 4);
        }
        if (!chunk("IDAT", reinterpret_cast<const unsigned char *>(block.data()), static_cast<uint32_t>(block.size()))) return false;
    }
    return chunk("IEND", nullptr, 0);
}

} // namespace

extern "C" {

size_t rtc_canvas_format_png8(const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels, uint8_t *buf, size_t cap) {
    if (!pixels || (channels != 3u && channels != 4u) || width == 0 || height == 0) return 0;
    size_t total = 0;
    encode_png(pixels, width, height, channels, [&](const unsigned char *p, size_t n) {
        if (buf && total < cap) std::memcpy(buf + total, p, (cap - total) < n ? (cap - total) : n);
        total += n;
        return true;
    });
    return total;
}

rtc_status rtc_canvas_write_png8(const char *path, const uint8_t *pixels, uint32_t width, uint32_t height, uint32_t channels) {
    if (!path || !pixels || (channels != 3u && channels != 4u) || width == 0 || height == 0) return RTC_ERR_ARG;
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return RTC_ERR_IO;
    const bool ok = encode_png(pixels, width, height, channels, [f](const unsigned char *p, size_t n) { return std::fwrite(p, 1, n, f) == n; });
    const int c = std::fclose(f);
    return (ok && c == 0) ? RTC_OK : RTC_ERR_IO;
}

size_t rtc_canvas_format_ppm(const double *rgb, uint32_t width, uint32_t height, char *buf, size_t cap) {
    if (!rgb) return 0;
    return copy_out(encode(rgb, width, height), buf, cap);
}

size_t rtc_canvas_format_ppm_rgb8(const uint8_t *rgb8, uint32_t width, uint32_t height, char *buf, size_t cap) {
    if (!rgb8) return 0;
    return copy_out(encode8(rgb8, width, height), buf, cap);
}

void rtc_color_scale255(const double *components, size_t n, uint8_t *out) {
    if (!components || !out) return;
    for (size_t i = 0; i < n; ++i) out[i] = static_cast<uint8_t>(scale255(components[i]));
}

void rtc_canvas_to_rgba8(const double *rgb, uint32_t width, uint32_t height, float gamma, uint8_t *out) {
    if (!rgb || !out) return;
    // Color::{red,green,blue}_scaled_gamma color.rs:55-65: scale(c.powf(gamma.recip().into()), 255);
    // the reciprocal is taken in f32 and widened, as in the reference
    const double e = static_cast<double>(1.0f / gamma);
    const size_t n = static_cast<size_t>(width) * height;
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) out[i * 4 + k] = static_cast<uint8_t>(scale255(std::pow(rgb[i * 3 + k], e)));
        out[i * 4 + 3] = 255; // std::u8::MAX canvas.rs:74
    }
}

rtc_status rtc_canvas_write_ppm(const char *path, const double *rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb) return RTC_ERR_ARG;
    return write_file(path, encode(rgb, width, height));
}

rtc_status rtc_canvas_write_ppm_rgb8(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height) {
    if (!path || !rgb8) return RTC_ERR_ARG;
    return write_file(path, encode8(rgb8, width, height));
}

} // extern "C"
