// host_ppm.cpp — [host] Canvas::write_to_file_simple (ch1/src/canvas.rs:86-109): ASCII PPM "P3",
// one text line per canvas row, components separated by single spaces, no 70-column
// wrapping, each component quantised by Color::scale (ch1/src/color.rs:100-114).
#include "rtc.h"

#include <cmath>
#include <cstdio>
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

} // namespace

extern "C" {

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
