// Host-only robustness check of the scene loader, the matrix helpers and the PPM / RGBA8 writers,
// meant to be built with -fsanitize=address,undefined (tests/test_host_cpu.py does that): feeds the
// loader the shipped scene, truncations of it at every byte, and a set of malformed documents. No GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "rtc.h"

static int load(const std::string &text) {
    rtc_shape *shapes = nullptr;
    uint32_t n = 0;
    rtc_light light;
    rtc_camera cam;
    char err[256] = {0};
    const int st = rtc_scene_load_yaml(text.c_str(), &shapes, &n, &light, &cam, err, sizeof err);
    if (st == RTC_OK && n > 0 && shapes == nullptr) std::abort();
    rtc_free(shapes);
    return st;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1]);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string doc = ss.str();
    if (load(doc) != RTC_OK) { std::puts("shipped scene does not load"); return 1; }
    int ok = 0, bad = 0;
    for (size_t cut = 0; cut < doc.size(); cut += 1) (load(doc.substr(0, cut)) == RTC_OK ? ok : bad)++;
    const char *junk[] = {"", "\n", "- add: sphere", "- add: camera\n  width: -3\n  height: 1e99\n  field-of-view: x",
                          "- add: sphere\n  transform:\n    - [ scale, 0, 0, 0 ]\n", "- define: a\n  extend: a\n  value:\n    color: [1,1,1]\n",
                          "- add: cube\n  material: nosuch\n", "- add: plane\n  transform: [[[[[[[[[[\n", "- add: sphere\n  material: { color: [1, 2 }\n",
                          "- add: light\n  at: [1,2]\n  intensity: [1,1,1,1,1]\n", "\t\t- add: sphere\n", "- add: sphere\n  material:\n    pattern:\n      type: nosuch\n",
                          "- define: t\n  value:\n    - [ translate, 1, 2, 3 ]\n- add: sphere\n  transform:\n    - t\n    - t\n    - [ rotate-x, nan ]\n"};
    for (const char *j : junk) (load(j) == RTC_OK ? ok : bad)++;
    std::string big = "- add: camera\n  width: 10\n  height: 5\n  field-of-view: 1.0\n  from: [0, 0, -5]\n  to: [0, 0, 0]\n  up: [0, 1, 0]\n"
                      "- add: light\n  at: [1, 2, 3]\n  intensity: [1, 1, 1]\n";
    for (int i = 0; i < 3000; ++i) big += "- add: sphere\n  transform:\n    - [ translate, " + std::to_string(i) + ", 0, 0 ]\n";
    if (load(big) != RTC_OK) { std::puts("large document rejected"); return 1; }
    // matrix helpers and writers on edge values
    double m[16], inv[16], id[16];
    rtc_matrix_identity(id);
    rtc_matrix_scaling(id, 0., 1., 1., m);
    if (rtc_matrix_inverse(m, inv) != RTC_ERR_SINGULAR) return 1;
    std::vector<double> rgb(7 * 5 * 3, 0.5);
    rgb[0] = 1e308; rgb[1] = -1e308; rgb[2] = 0.0 / 1.0;
    std::vector<uint8_t> rgba(7 * 5 * 4);
    rtc_canvas_to_rgba8(rgb.data(), 7, 5, 2.2f, rgba.data());
    const size_t need = rtc_canvas_format_ppm(rgb.data(), 7, 5, nullptr, 0);
    std::vector<char> buf(need + 1);
    if (rtc_canvas_format_ppm(rgb.data(), 7, 5, buf.data(), buf.size()) != need) return 1;
    std::printf("loader: %d documents accepted, %d rejected, no crash\n", ok, bad);
    return 0;
}
