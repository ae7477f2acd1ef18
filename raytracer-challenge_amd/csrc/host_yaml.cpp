// host_yaml.cpp — [host] scene loader for the `jamis.yml` vocabulary (ch1/jamis.yml:1-183).
//
// The reference ships jamis.yml as a data file but has no YAML loader (its only loader is
// Lua, ch1/src/lua.rs), so this is written new. It builds the same World /
// Camera the reference's constructors would:
//   - transform lists apply in listed order by LEFT-multiplication, exactly like the fluent
//     builders `identity().scaling(..).translation(..)` (transform.rs:53-69);
//   - materials start from Material::default() (white .1/.9/.9/200 0/0/1.0) as lua.rs:187 does,
//     and an unknown material key is an error as in lua.rs:216;
//   - shapes go through {Sphere,Plane,Cube}::new_with_transform_and_material (inverse +
//     transpose at construction, shape.rs:308-317) and get world ids like World::add_shape
//     (shape.rs:661-667); only the first light is used (lua.rs:148-150);
//   - camera = Camera::new_with_transform(width, height, field-of-view,
//     make_view_transform(from, to, up)) (camera.rs:33, transform.rs:204).
//
// The parser accepts the YAML subset the vocabulary needs: block maps and sequences by
// indentation, "- key: value" items, flow sequences "[a, b]" and flow maps "{k: v}", comments,
// plain scalars.
#include "rtc.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

struct Node {
    enum Kind { Scalar, Seq, Map } kind = Scalar;
    std::string scalar;
    std::vector<std::shared_ptr<Node>> seq;
    std::vector<std::pair<std::string, std::shared_ptr<Node>>> map;
    int line = 0;

    const Node *get(const std::string &key) const {
        for (const auto &kv : map)
            if (kv.first == key) return kv.second.get();
        return nullptr;
    }
};
using NodeP = std::shared_ptr<Node>;

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

[[noreturn]] void fail(int line, const std::string &msg) {
    throw ParseError("line " + std::to_string(line) + ": " + msg);
}

struct Line {
    int indent;
    std::string text; // without indentation, comments and trailing blanks
    int number;
};

std::string strip(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
    return s.substr(a, b - a);
}

std::vector<Line> split_lines(const char *text) {
    std::vector<Line> out;
    int number = 0;
    const char *p = text;
    while (*p) {
        const char *e = std::strchr(p, '\n');
        std::string raw = e ? std::string(p, e) : std::string(p);
        p = e ? e + 1 : p + raw.size();
        ++number;
        // drop comments (a '#' at line start or preceded by whitespace; no quoted strings here)
        for (size_t i = 0; i < raw.size(); ++i)
            if (raw[i] == '#' && (i == 0 || raw[i - 1] == ' ' || raw[i - 1] == '\t')) { raw.resize(i); break; }
        size_t ind = 0;
        while (ind < raw.size() && raw[ind] == ' ') ++ind;
        std::string body = strip(raw);
        if (body.empty() || body == "---") continue;
        if (ind < raw.size() && raw[ind] == '\t') fail(number, "tab indentation is not supported");
        out.push_back(Line{static_cast<int>(ind), body, number});
    }
    return out;
}

// ---- flow values: scalars and [ ... ] -------------------------------------------------
NodeP parse_flow(const std::string &s, size_t &i, int line) {
    while (i < s.size() && s[i] == ' ') ++i;
    auto n = std::make_shared<Node>();
    n->line = line;
    if (i < s.size() && s[i] == '[') {
        n->kind = Node::Seq;
        ++i;
        for (;;) {
            while (i < s.size() && s[i] == ' ') ++i;
            if (i >= s.size()) fail(line, "unterminated '['");
            if (s[i] == ']') { ++i; break; }
            n->seq.push_back(parse_flow(s, i, line));
            while (i < s.size() && s[i] == ' ') ++i;
            if (i < s.size() && s[i] == ',') { ++i; continue; }
            if (i < s.size() && s[i] == ']') { ++i; break; }
            fail(line, "expected ',' or ']' in flow sequence");
        }
        return n;
    }
    if (i < s.size() && s[i] == '{') { // flow map {key: value, ...}
        n->kind = Node::Map;
        ++i;
        for (;;) {
            while (i < s.size() && s[i] == ' ') ++i;
            if (i >= s.size()) fail(line, "unterminated '{'");
            if (s[i] == '}') { ++i; break; }
            const size_t c = s.find(':', i);
            if (c == std::string::npos) fail(line, "expected 'key: value' in flow map");
            const std::string key = strip(s.substr(i, c - i));
            if (key.empty() || key.find_first_of(",{}[]") != std::string::npos) fail(line, "bad key in flow map");
            i = c + 1;
            n->map.emplace_back(key, parse_flow(s, i, line));
            while (i < s.size() && s[i] == ' ') ++i;
            if (i < s.size() && s[i] == ',') { ++i; continue; }
            if (i < s.size() && s[i] == '}') { ++i; break; }
            fail(line, "expected ',' or '}' in flow map");
        }
        return n;
    }
    size_t start = i;
    while (i < s.size() && s[i] != ',' && s[i] != ']' && s[i] != '}') ++i;
    n->kind = Node::Scalar;
    n->scalar = strip(s.substr(start, i - start));
    if (n->scalar.size() >= 2 && ((n->scalar.front() == '"' && n->scalar.back() == '"') ||
                                  (n->scalar.front() == '\'' && n->scalar.back() == '\'')))
        n->scalar = n->scalar.substr(1, n->scalar.size() - 2);
    return n;
}

NodeP parse_inline_value(const std::string &s, int line) {
    size_t i = 0;
    NodeP n = parse_flow(s, i, line);
    while (i < s.size() && s[i] == ' ') ++i;
    if (i != s.size() && n->kind != Node::Scalar) fail(line, "trailing characters after flow value");
    if (n->kind == Node::Scalar) n->scalar = strip(s);
    return n;
}

// ---- block structure --------------------------------------------------------------------
struct Parser {
    std::vector<Line> lines;
    size_t pos = 0;

    static bool split_key(const std::string &t, std::string &key, std::string &rest) {
        if (t.empty() || t[0] == '[' || t[0] == '-') return false;
        size_t c = t.find(':');
        if (c == std::string::npos) return false;
        if (c + 1 < t.size() && t[c + 1] != ' ') return false;
        key = strip(t.substr(0, c));
        rest = strip(t.substr(c + 1));
        return !key.empty();
    }

    NodeP parse_block(int indent) {
        if (pos >= lines.size()) fail(lines.empty() ? 0 : lines.back().number, "unexpected end of document");
        const Line &l = lines[pos];
        if (l.text[0] == '-' && (l.text.size() == 1 || l.text[1] == ' ')) return parse_seq(indent);
        return parse_map(indent);
    }

    NodeP parse_seq(int indent) {
        auto n = std::make_shared<Node>();
        n->kind = Node::Seq;
        n->line = lines[pos].number;
        while (pos < lines.size() && lines[pos].indent == indent && lines[pos].text[0] == '-' &&
               (lines[pos].text.size() == 1 || lines[pos].text[1] == ' ')) {
            Line l = lines[pos];
            std::string rest = strip(l.text.substr(1));
            if (rest.empty()) {
                ++pos;
                if (pos >= lines.size() || lines[pos].indent <= indent) fail(l.number, "empty sequence item");
                n->seq.push_back(parse_block(lines[pos].indent));
                continue;
            }
            std::string key, val;
            if (split_key(rest, key, val)) {
                // "- key: value": a map whose first key sits on the dash line; re-enter it as a
                // map at the column of the key.
                int key_col = indent + static_cast<int>(l.text.size() - strip(l.text.substr(1)).size());
                lines[pos].indent = key_col;
                lines[pos].text = rest;
                n->seq.push_back(parse_map(key_col));
            } else {
                n->seq.push_back(parse_inline_value(rest, l.number));
                ++pos;
            }
        }
        if (pos < lines.size() && lines[pos].indent > indent) fail(lines[pos].number, "bad indentation in sequence");
        return n;
    }

    NodeP parse_map(int indent) {
        auto n = std::make_shared<Node>();
        n->kind = Node::Map;
        n->line = lines[pos].number;
        while (pos < lines.size() && lines[pos].indent == indent) {
            const Line l = lines[pos];
            std::string key, val;
            if (!split_key(l.text, key, val)) {
                if (l.text[0] == '-') break; // next item of an enclosing sequence at this column
                fail(l.number, "expected 'key: value'");
            }
            ++pos;
            NodeP child;
            if (!val.empty()) {
                child = parse_inline_value(val, l.number);
            } else if (pos < lines.size() && (lines[pos].indent > indent ||
                       (lines[pos].indent == indent && lines[pos].text[0] == '-' && indent > 0))) {
                child = parse_block(lines[pos].indent);
            } else {
                child = std::make_shared<Node>(); // empty scalar
                child->line = l.number;
            }
            n->map.emplace_back(key, child);
        }
        if (pos < lines.size() && lines[pos].indent > indent) fail(lines[pos].number, "bad indentation in map");
        return n;
    }
};

// ---- interpretation ---------------------------------------------------------------------
double as_number(const Node *n, const char *what) {
    if (!n || n->kind != Node::Scalar || n->scalar.empty()) fail(n ? n->line : 0, std::string("expected a number for ") + what);
    char *end = nullptr;
    const double v = std::strtod(n->scalar.c_str(), &end);
    if (end == n->scalar.c_str() || *end != 0) fail(n->line, std::string("invalid number for ") + what + ": " + n->scalar);
    return v;
}

void as_triple(const Node *n, const char *what, double out[3]) {
    if (!n || n->kind != Node::Seq || n->seq.size() != 3) fail(n ? n->line : 0, std::string("expected [x, y, z] for ") + what);
    for (int i = 0; i < 3; ++i) out[i] = as_number(n->seq[i].get(), what);
}

struct Defs {
    std::map<std::string, NodeP> values;
};

// Resolve `define`d names and `extend:` chains into a flat key list: base keys first, then the
// map's own keys in listed order (later entries override earlier ones when applied).
void flatten_material(const Defs &defs, const Node *n, std::vector<std::pair<std::string, const Node *>> &out, int depth = 0) {
    if (!n) return;
    if (depth > 16) fail(n->line, "definition recursion too deep");
    if (n->kind == Node::Scalar) {
        auto it = defs.values.find(n->scalar);
        if (it == defs.values.end()) fail(n->line, "unknown material name: " + n->scalar);
        flatten_material(defs, it->second.get(), out, depth + 1);
        return;
    }
    if (n->kind != Node::Map) fail(n->line, "material must be a name or a map");
    if (const Node *ext = n->get("extend")) flatten_material(defs, ext, out, depth + 1);
    for (const auto &kv : n->map)
        if (kv.first != "extend") out.emplace_back(kv.first, kv.second.get());
}

void build_transform(const Defs &defs, const Node *list, double m[16], int depth = 0) {
    if (!list) return;
    if (depth > 16) fail(list->line, "definition recursion too deep");
    if (list->kind != Node::Seq) fail(list->line, "transform must be a list");
    for (const auto &itp : list->seq) {
        const Node *it = itp.get();
        if (it->kind == Node::Scalar) { // reference to a defined transform list
            auto d = defs.values.find(it->scalar);
            if (d == defs.values.end()) fail(it->line, "unknown transform name: " + it->scalar);
            build_transform(defs, d->second.get(), m, depth + 1);
            continue;
        }
        if (it->kind != Node::Seq || it->seq.empty() || it->seq[0]->kind != Node::Scalar) fail(it->line, "bad transform item");
        const std::string &op = it->seq[0]->scalar;
        auto arg = [&](size_t i) { return as_number(i < it->seq.size() ? it->seq[i].get() : nullptr, op.c_str()); };
        auto need = [&](size_t n) { if (it->seq.size() != n + 1) fail(it->line, op + " takes " + std::to_string(n) + " numbers"); };
        if (op == "scale") { need(3); rtc_matrix_scaling(m, arg(1), arg(2), arg(3), m); }
        else if (op == "translate") { need(3); rtc_matrix_translation(m, arg(1), arg(2), arg(3), m); }
        else if (op == "rotate-x") { need(1); rtc_matrix_rotation_x(m, arg(1), m); }
        else if (op == "rotate-y") { need(1); rtc_matrix_rotation_y(m, arg(1), m); }
        else if (op == "rotate-z") { need(1); rtc_matrix_rotation_z(m, arg(1), m); }
        else if (op == "shear") { need(6); rtc_matrix_shearing(m, arg(1), arg(2), arg(3), arg(4), arg(5), arg(6), m); }
        else fail(it->line, "unknown transform: " + op);
    }
}

void apply_pattern(const Defs &defs, const Node *p, rtc_material &mat) {
    if (!p || p->kind != Node::Map) fail(p ? p->line : 0, "pattern must be a map");
    const Node *type = p->get("type");
    if (!type || type->kind != Node::Scalar) fail(p->line, "pattern needs a type");
    uint32_t kind;
    if (type->scalar == "stripes") kind = RTC_PATTERN_STRIPE;
    else if (type->scalar == "checkers") kind = RTC_PATTERN_CHECKER;
    else if (type->scalar == "gradient") kind = RTC_PATTERN_GRADIENT;
    else if (type->scalar == "ring" || type->scalar == "rings") kind = RTC_PATTERN_RING;
    else if (type->scalar == "grid") kind = RTC_PATTERN_GRID;
    else if (type->scalar == "test") kind = RTC_PATTERN_TEST;
    else fail(type->line, "unknown pattern type: " + type->scalar);
    double a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
    const Node *colors = p->get("colors");
    if (kind != RTC_PATTERN_TEST) {
        if (!colors || colors->kind != Node::Seq || colors->seq.size() != 2) fail(p->line, "pattern needs colors: [a, b]");
        as_triple(colors->seq[0].get(), "pattern colour", a);
        as_triple(colors->seq[1].get(), "pattern colour", b);
    }
    double xf[16];
    rtc_matrix_identity(xf);
    build_transform(defs, p->get("transform"), xf);
    for (const auto &kv : p->map)
        if (kv.first != "type" && kv.first != "colors" && kv.first != "transform") fail(kv.second->line, "Invalid pattern property: " + kv.first);
    const rtc_status st = rtc_material_set_pattern(&mat, kind, a, b, xf);
    if (st != RTC_OK) fail(p->line, "pattern transform: " + std::string(rtc_strerror(st)));
}

void build_material(const Defs &defs, const Node *n, rtc_material &mat) {
    rtc_material_default(&mat); // lua.rs:187
    std::vector<std::pair<std::string, const Node *>> pairs;
    flatten_material(defs, n, pairs);
    for (const auto &kv : pairs) {
        const std::string &key = kv.first;
        const Node *v = kv.second;
        if (key == "color") { as_triple(v, "color", mat.color); mat.has_color = 1; }
        else if (key == "ambient") mat.ambient = as_number(v, "ambient");
        else if (key == "diffuse") mat.diffuse = as_number(v, "diffuse");
        else if (key == "specular") mat.specular = as_number(v, "specular");
        else if (key == "shininess") mat.shininess = as_number(v, "shininess");
        else if (key == "reflective") mat.reflective = as_number(v, "reflective");
        else if (key == "transparency") mat.transparency = as_number(v, "transparency");
        else if (key == "refractive-index") mat.refractive_index = as_number(v, "refractive-index");
        else if (key == "pattern") apply_pattern(defs, v, mat);
        else fail(v->line, "Invalid material property: " + key); // lua.rs:216
    }
}

struct Scene {
    std::vector<rtc_shape> shapes;
    rtc_light light;
    bool have_light = false;
    rtc_camera camera;
    bool have_camera = false;
};

void interpret(const Node &root, Scene &sc) {
    if (root.kind != Node::Seq) fail(root.line, "scene must be a sequence of entries");
    Defs defs;
    for (const auto &ep : root.seq) {
        const Node &e = *ep;
        if (e.kind != Node::Map) fail(e.line, "scene entry must be a map");
        if (const Node *d = e.get("define")) {
            const Node *val = e.get("value");
            if (!val || d->kind != Node::Scalar) fail(e.line, "define needs a name and a value");
            NodeP stored;
            for (const auto &kv : e.map) if (kv.first == "value") stored = kv.second;
            if (const Node *ext = e.get("extend")) { // Jamis's format: extend sits beside value
                auto merged = std::make_shared<Node>(*stored);
                if (merged->kind != Node::Map) fail(e.line, "extend needs a map value");
                auto extn = std::make_shared<Node>(*ext);
                merged->map.insert(merged->map.begin(), {"extend", extn});
                stored = merged;
            }
            defs.values[d->scalar] = stored;
            continue;
        }
        const Node *add = e.get("add");
        if (!add || add->kind != Node::Scalar) fail(e.line, "entry needs 'add' or 'define'");
        const std::string &what = add->scalar;
        if (what == "camera") {
            double from[3], to[3], up[3], view[16];
            as_triple(e.get("from"), "from", from);
            as_triple(e.get("to"), "to", to);
            as_triple(e.get("up"), "up", up);
            const double w = as_number(e.get("width"), "width"), h = as_number(e.get("height"), "height");
            if (w < 1 || h < 1 || w > 2147483647. || h > 2147483647. || w != static_cast<double>(static_cast<uint32_t>(w)) ||
                h != static_cast<double>(static_cast<uint32_t>(h)))
                fail(e.line, "camera width/height must be positive integers"); // lua.rs:159-170
            rtc_view_transform(from, to, up, view);
            const rtc_status st = rtc_camera_init(static_cast<uint32_t>(w), static_cast<uint32_t>(h),
                                                  as_number(e.get("field-of-view"), "field-of-view"), view, &sc.camera);
            if (st != RTC_OK) fail(e.line, std::string("camera: ") + rtc_strerror(st));
            if (const Node *s = e.get("samples")) {
                const double sv = as_number(s, "samples");
                if (sv < 0 || sv > 255 || sv != static_cast<double>(static_cast<uint32_t>(sv))) fail(s->line, "samples out of bounds"); // lua.rs:172-183
                sc.camera.samples = static_cast<uint32_t>(sv);
            }
            sc.have_camera = true;
        } else if (what == "light") {
            rtc_light l;
            as_triple(e.get("at"), "at", l.position);
            as_triple(e.get("intensity"), "intensity", l.intensity);
            if (!sc.have_light) { sc.light = l; sc.have_light = true; } // lights[1] only, lua.rs:148-150
        } else if (what == "sphere" || what == "plane" || what == "cube") {
            const uint32_t kind = what == "sphere" ? RTC_SPHERE : what == "plane" ? RTC_PLANE : RTC_CUBE;
            double xf[16];
            rtc_matrix_identity(xf);
            build_transform(defs, e.get("transform"), xf);
            rtc_material mat;
            build_material(defs, e.get("material"), mat);
            rtc_shape s;
            const rtc_status st = rtc_shape_init(kind, xf, &mat, &s);
            if (st != RTC_OK) fail(e.line, what + ": " + rtc_strerror(st));
            s.world_id = static_cast<uint32_t>(sc.shapes.size()) + 1; // World::add_shape shape.rs:661-667
            sc.shapes.push_back(s);
        } else {
            fail(add->line, "Invalid shape type: " + what); // lua.rs:322-326
        }
    }
    if (!sc.have_light) fail(root.line, "scene has no light");
    if (!sc.have_camera) fail(root.line, "scene has no camera");
}

void set_err(char *errbuf, size_t len, const std::string &msg) {
    if (errbuf && len) {
        std::snprintf(errbuf, len, "%s", msg.c_str());
    }
}

} // namespace

extern "C" {

rtc_status rtc_scene_load_yaml(const char *text, rtc_shape **shapes_out, uint32_t *n_out,
                               rtc_light *light_out, rtc_camera *camera_out, char *errbuf,
                               size_t errbuf_len) {
    if (!text || !shapes_out || !n_out || !light_out || !camera_out) return RTC_ERR_ARG;
    *shapes_out = nullptr;
    *n_out = 0;
    try {
        Parser p;
        p.lines = split_lines(text);
        if (p.lines.empty()) fail(0, "empty document");
        NodeP root = p.parse_block(p.lines[0].indent);
        if (p.pos != p.lines.size()) fail(p.lines[p.pos].number, "unexpected content");
        Scene sc;
        interpret(*root, sc);
        const size_t bytes = sizeof(rtc_shape) * (sc.shapes.empty() ? 1 : sc.shapes.size());
        rtc_shape *arr = static_cast<rtc_shape *>(std::malloc(bytes));
        if (!arr) return RTC_ERR_NOMEM;
        if (!sc.shapes.empty()) std::memcpy(arr, sc.shapes.data(), sizeof(rtc_shape) * sc.shapes.size());
        *shapes_out = arr;
        *n_out = static_cast<uint32_t>(sc.shapes.size());
        *light_out = sc.light;
        *camera_out = sc.camera;
        return RTC_OK;
    } catch (const ParseError &e) {
        set_err(errbuf, errbuf_len, e.what());
        return RTC_ERR_PARSE;
    } catch (const std::bad_alloc &) {
        return RTC_ERR_NOMEM;
    } catch (...) {
        set_err(errbuf, errbuf_len, "internal error");
        return RTC_ERR_PARSE;
    }
}

rtc_status rtc_scene_load_yaml_file(const char *path, rtc_shape **shapes_out, uint32_t *n_out,
                                    rtc_light *light_out, rtc_camera *camera_out, char *errbuf,
                                    size_t errbuf_len) {
    if (!path) return RTC_ERR_ARG;
    std::FILE *f = std::fopen(path, "rb");
    if (!f) { set_err(errbuf, errbuf_len, std::string("cannot open ") + path); return RTC_ERR_IO; }
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    std::fclose(f);
    return rtc_scene_load_yaml(text.c_str(), shapes_out, n_out, light_out, camera_out, errbuf, errbuf_len);
}

} // extern "C"
