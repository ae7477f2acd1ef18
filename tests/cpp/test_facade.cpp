// test_facade.cpp — the reference's own tests for the render path, re-expressed against the C++
// mirror of its API (raytracer-challenge_amd/host/ch1.hpp). Every render below runs on the MI355X
// through the C-ABI. Run by tests/test_gpu_facade.py (marked gpu); exits non-zero on failure.
//
//   camera.rs:216-231  test_render1      camera.rs:250-255 test_async1
//   shape.rs:1073-1112 test_color_at1-3  shape.rs:1249-1266 test_reflect4
//   shape.rs:1397-1420 test_refract_5    shape.rs:1452-1476 test_schlick_4
//   benches/render.rs:10-79 the Criterion scene (smoke: renders, canvas is not black)
#include <cmath>
#include <cstdio>
#include <string>

#include "ch1.hpp"

using namespace ch1;

static int failures = 0;
static bool floats_equal(double a, double b) { return std::fabs(a - b) < 0.0001; } // lib.rs:13-15
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static void test_render1() {
    World world = World::default_();
    Camera camera = Camera::new_with_transform(11, 11, M_PI / 2.,
        Matrix::make_view_transform(Point::new_(0., 0., -5.), Point::new_(0., 0., 0.), Vector::new_(0., 1., 0.)));
    Canvas c = camera.render(world);
    Color test_color = c.get_pixel(5, 5);
    EXPECT(floats_equal(test_color.red, 0.38066));
    EXPECT(floats_equal(test_color.green, 0.47583));
    EXPECT(floats_equal(test_color.blue, 0.2855));
    // Camera::render leaves the last row/column black; render_async does not (camera.rs:120-121,149)
    Color edge = c.get_pixel(10, 5);
    EXPECT(edge.red == 0. && edge.green == 0. && edge.blue == 0.);
    Canvas a = camera.render_async(world);
    EXPECT(a.get_pixel(5, 5).red == test_color.red);
}

static void test_async1() {
    Camera c = Camera::new_(20, 10, 1.5);
    World w = World::default_();
    Canvas cv = c.render_async(w);
    EXPECT(cv.width == 20 && cv.height == 10);
}

static void test_color_at() {
    World w = World::default_();
    Color c = w.color_at(Point::new_(0., 0., -5.), Vector::new_(0., 1., 0.), 1);
    EXPECT(floats_equal(c.red, 0.) && floats_equal(c.green, 0.) && floats_equal(c.blue, 0.));
    c = w.color_at(Point::new_(0., 0., -5.), Vector::new_(0., 0., 1.), 1);
    EXPECT(floats_equal(c.red, 0.38066) && floats_equal(c.green, 0.47583) && floats_equal(c.blue, 0.2855));
    w.get_shape_mut(0).get_material_mut().ambient = 1.0;
    w.get_shape_mut(1).get_material_mut().ambient = 1.0;
    const Color inner = w.get_shape(1).get_material().color;
    c = w.color_at(Point::new_(0., 0., 0.75), Vector::new_(0., 0., -1.), 1);
    EXPECT(floats_equal(c.red, inner.red) && floats_equal(c.green, inner.green) && floats_equal(c.blue, inner.blue));
}

static World floor_world(double kr, double tr, bool ball) {
    World world = World::default_();
    Material m = Material::default_();
    m.reflectiveness = kr;
    m.transparency = tr;
    if (tr != 0.) m.refractive_index = 1.5;
    world.add_shape(Plane::new_with_transform_and_material(Matrix::identity().translation(0., -1., 0.), m));
    if (ball) {
        Material mball = Material::solid_with_defaults(Color::new_(1.0, 0., 0.));
        mball.ambient = 0.5;
        world.add_shape(Sphere::new_with_transform_and_material(Matrix::identity().translation(0., -3.5, -0.5), mball));
    }
    return world;
}

static void test_reflect_refract_schlick() {
    const double n = std::sqrt(2.0) / 2.0;
    // the reference calls shade_hit on the floor intersection; the floor is the ray's first hit,
    // so color_at(ray, remaining) returns the same value
    Color c = floor_world(0.5, 0., false).color_at(Point::new_(0., 0., -3.), Vector::new_(0., -n, n), 1);
    EXPECT(floats_equal(c.red, 0.87677) && floats_equal(c.green, 0.92436) && floats_equal(c.blue, 0.82918));
    c = floor_world(0., 0.5, true).color_at(Point::new_(0., 0., -3.), Vector::new_(0., -n, n), 5);
    EXPECT(floats_equal(c.red, 0.93642) && floats_equal(c.green, 0.68642) && floats_equal(c.blue, 0.68642));
    c = floor_world(0.5, 0.5, true).color_at(Point::new_(0., 0., -3.), Vector::new_(0., -n, n), 5);
    EXPECT(floats_equal(c.red, 0.93391) && floats_equal(c.green, 0.69643) && floats_equal(c.blue, 0.69243));
}

static void criterion_scene(const std::string &ppm_path) { // benches/render.rs:10-79
    World world = World::new_(Light::default_());
    Material m1 = Material::DEFAULT(); m1.color = Color::RED(); m1.diffuse = 0.1; m1.transparency = 1.0; m1.refractive_index = 1.15; m1.specular = 0.1; m1.ambient = 0.1;
    world.add_shape(Sphere::new_with_transform_and_material(Matrix::identity().translation(-0.5, 1., 0.5), m1));
    Material m2 = Material::DEFAULT(); m2.color = Color::GREEN(); m2.diffuse = 0.1; m2.transparency = 1.0; m2.ambient = 0.1; m2.refractive_index = 1.5; m2.specular = 0.1;
    world.add_shape(Sphere::new_with_transform_and_material(Matrix::identity().scaling(0.5, 0.5, 0.5).translation(1., 0.7, -3.5), m2));
    Material m3 = Material::DEFAULT(); m3.color = Color::BLUE(); m3.diffuse = 0.7; m3.specular = 0.3;
    world.add_shape(Sphere::new_with_transform_and_material(
        Matrix::identity().scaling(0.8, 0.8, 0.8).translation(-2.5, 0.53, -0.75).rotation_x(M_PI / 4.0), m3));
    Material mp = Material::pattern_with_defaults(CheckerPattern(Color::WHITE(), Color::BLACK()));
    mp.diffuse = 0.2; mp.ambient = 0.6; mp.specular = 0.3;
    world.add_shape(Plane::new_with_transform_and_material(Matrix::identity().translation(0., -3., 0.), mp));
    Camera camera = Camera::new_with_transform(400, 300, M_PI / 3.0,
        Matrix::make_view_transform(Point::new_(0., 7., 0.), Point::new_(0., 0., 0.), Vector::new_(1., 0., 0.)));
    Canvas canvas = camera.render_async(world);
    double sum = 0.;
    for (double v : canvas.pixels) sum += v;
    EXPECT(sum > 1000.);
    if (!ppm_path.empty()) {
        canvas.write_to_file_simple(ppm_path);
        // the quantised Canvas (only the device's 8-bit frame crossed PCIe) writes the same file, byte for byte
        Canvas q = camera.render_async_rgb8(world);
        EXPECT(q.is_quantised() && q.pixels.empty() && q.rgb8.size() == 400u * 300u * 3u);
        q.write_to_file_simple(ppm_path + ".rgb8");
        bool threw = false;
        try { (void)q.get_pixel(0, 0); } catch (const std::logic_error &) { threw = true; }
        EXPECT(threw);
    }
}

static void test_panics() {
    bool threw = false;
    try { Sphere::new_with_transform(Matrix::identity().scaling(0., 1., 1.)); } catch (const Panic &p) { threw = p.status == RTC_ERR_SINGULAR; }
    EXPECT(threw); // "Matrix is not invertable" transform.rs:177
    threw = false;
    try {
        World w = World::new_(Light::default_());
        Material m = Material::default_(); m.has_color = false;
        w.add_shape(Sphere::new_with_transform_and_material(Matrix::identity(), m));
        Camera::new_(4, 4, 1.0).render(w);
    } catch (const Panic &p) { threw = p.status == RTC_ERR_NO_COLOR; }
    EXPECT(threw); // material.rs:331
}

// The façade keeps ONE flattened World resident on the device between calls (ch1::World::Resident): an unchanged
// World must not be re-uploaded, an edited one must be, and antialiasing_samples = 0 takes the 4-sub-sample branch
// (camera.rs:96-99: only the value 1 is the one-ray branch).
static void test_resident_world_and_samples() {
    World world = World::default_();
    Camera cam = Camera::new_with_transform(24, 16, M_PI / 2.0,
        Matrix::make_view_transform(Point::new_(0., 0., -5.), Point::new_(0., 0., 0.), Vector::new_(0., 1., 0.)));
    const Canvas a = cam.render_async(world);
    rtc_world *first = World::Resident::instance().get({}, rtc_light{}); // forces a different (empty) World in ...
    (void)first;
    const Canvas b = cam.render_async(world);                              // ... so this call uploads again
    EXPECT(a.pixels == b.pixels);
    const Canvas c = cam.render_async(world);                              // unchanged: served from the resident copy
    EXPECT(a.pixels == c.pixels);
    world.get_shape_mut(0).material.ambient = 0.9;                         // an edit must be seen
    const Canvas d = cam.render_async(world);
    EXPECT(!(a.pixels == d.pixels));
    Camera aa = cam;
    aa.set_samples(0);
    const Canvas e = aa.render_async(world);
    aa.set_samples(4);
    const Canvas f = aa.render_async(world);
    EXPECT(e.pixels == f.pixels && !(e.pixels == d.pixels));
}

// lua.rs:50-91 render_lua on a script of this repository (raytracer-challenge_amd/data/orbit_animation.lua): frames arrive in
// the script's order, AddFrame frames numbered, the still last; a frame equals Camera::render_async_rgb8 of the same job.
static void test_render_lua(const std::string &script) {
    if (script.empty()) return;
    std::vector<Canvas> frames;
    std::vector<std::string> names;
    std::vector<int> numbers;
    const std::string printed = render_lua(script, [&](const Canvas &c, const std::string &file, int frame) {
        frames.push_back(c);
        names.push_back(file);
        numbers.push_back(frame);
    });
    EXPECT(frames.size() == 13 && names[0] == "orbit.gif" && names[12] == "orbit_top.ppm");
    for (int k = 0; k < 12 && k < static_cast<int>(numbers.size()); ++k) EXPECT(numbers[k] == k);
    EXPECT(numbers.size() == 13 && numbers[12] == -1);
    EXPECT(printed.find("frames: 12") != std::string::npos);
    EXPECT(frames.size() == 13 && frames[0].is_quantised() && frames[0].width == 320 && frames[0].height == 200);
    EXPECT(frames.size() == 13 && frames[0].rgb8 != frames[1].rgb8);
    if (frames.size() == 13) { // Canvas::write_to_file: PNG of the quantised frame; any other extension panics
        frames[12].write_to_file(script + ".top.png");
        bool refused = false;
        try { frames[12].write_to_file(script + ".top.jpg"); } catch (const Panic &) { refused = true; }
        EXPECT(refused);
    }
    bool threw = false;
    try { render_lua(script, [&](const Canvas &, const std::string &, int) { throw std::runtime_error("sink"); }); }
    catch (const std::runtime_error &e) { threw = std::string(e.what()) == "sink"; }
    EXPECT(threw);
    threw = false;
    try { render_lua(script + ".missing", [&](const Canvas &, const std::string &, int) {}); }
    catch (const Panic &) { threw = true; }
    EXPECT(threw);
}

int main(int argc, char **argv) {
    try {
        test_resident_world_and_samples();
        test_render1();
        test_async1();
        test_color_at();
        test_reflect_refract_schlick();
        criterion_scene(argc > 1 ? argv[1] : "");
        test_panics();
        test_render_lua(argc > 2 ? argv[2] : "");
    } catch (const std::exception &e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "%d FAILED\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
