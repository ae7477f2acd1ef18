// host_lua.cpp — [host] scene loader for the TABLE-LITERAL subset of the reference's Lua front-end.
//
// The reference describes scenes as Lua tables and hands them to `Render(world, camera, outfile)` (ch1/src/lua.rs:57-79);
// `world_from_table` / `camera_from_table` / `material_from_table` / `transform_from_table` / `pattern_from_table`
// (lua.rs:109-330) turn the tables into World and Camera. Real Lua programs (ex2.lua, functions.lua: functions, loops,
// math.random) need an interpreter, which this image does not have and this library does not contain. But a scene file like
// ch1/jamis.lua is nothing but global assignments of table constructors and one Render call: this loader evaluates exactly
// that subset —
//     chunk  := { ['local'] Name '=' exp  |  Name '(' [exp {',' exp}] ')' } ;
//     exp    := constant arithmetic (+ - * / ^ %, unary -, parentheses) over numbers, strings, nil / true / false,
//               table constructors { k = v, [exp] = v, v, ... }, and Name{.Name} lookups of earlier globals (math.pi, math.huge)
// — and then applies the *_from_table functions' rules to the tables, statement for statement:
//   * transform_from_table (lua.rs:257-291): rotate_x, rotate_y, rotate_z, scale (uniform), position — in THAT order,
//     each LEFT-multiplied (transform.rs:53-105), whatever order the keys are written in;
//   * material_from_table (lua.rs:186-239): starts from Material::default(); inside `material` the keys ambient, diffuse,
//     specular, shininess, reflectiveness, transparency, refractive_index, color, pattern — anything else is an error
//     ("Invalid material property"); then a shape-level `color` (or else `pattern`) overrides, silently ignored when malformed;
//   * pattern_from_table (lua.rs:109-143): "checks" / "stripes" with color_a, color_b; "grid" = white on black; the pattern
//     table's own rotate_* / scale / position are its transform;
//   * lights: only lights[1] (lua.rs:148-150); camera: screenwidth, screenheight (Lua INTEGERS, lua.rs:158-170), position,
//     lookat, up, fov, optional samples (integer 0..255, lua.rs:172-183);
//   * shapes: "sphere" | "plane" | "cube" through *::new_with_transform_and_material, world ids as World::add_shape.
// Anything outside the subset is RTC_ERR_PARSE with a message that says an interpreter is needed.
// PARITY UNPINNED: the reference holds no test of its Lua path; this follows lua.rs by source reading.
#include "rtc.h"

#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

struct LuaError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] void fail(int line, const std::string &msg) { throw LuaError("line " + std::to_string(line) + ": " + msg); }
[[noreturn]] void need_vm(int line, const std::string &what) {
    fail(line, what + ": this loader evaluates table literals, constant arithmetic and Render(world, camera, file) only — the script needs a Lua interpreter");
}

struct Table;
struct Value {
    enum Kind { Nil, Bool, Int, Num, Str, Tab } kind = Nil;
    bool b = false;
    long long i = 0;
    double n = 0.;
    std::string s;
    std::shared_ptr<Table> t;
    bool is_number() const { return kind == Int || kind == Num; }
    double number() const { return kind == Int ? static_cast<double>(i) : n; }
};
struct Table {
    std::vector<std::pair<std::string, Value>> fields; // string keys, in writing order (a later duplicate wins, as in Lua)
    std::map<long long, Value> array;                   // integer keys (positional entries: 1, 2, ...)
    const Value *get(const std::string &k) const {
        const Value *r = nullptr;
        for (const auto &kv : fields)
            if (kv.first == k) r = &kv.second;
        return (r && r->kind != Value::Nil) ? r : nullptr;
    }
    const Value *at(long long k) const {
        auto it = array.find(k);
        return (it != array.end() && it->second.kind != Value::Nil) ? &it->second : nullptr;
    }
};

// ---- lexer ---------------------------------------------------------------------------------------------------------
struct Tok {
    enum Kind { End, Name, Number, String, Sym } kind = End;
    std::string text;
    bool is_int = false;
    long long i = 0;
    double n = 0.;
    int line = 1;
};

struct Lexer {
    const char *p;
    int line = 1;
    explicit Lexer(const char *text) : p(text) {}

    void skip() {
        for (;;) {
            while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') { if (*p == '\n') ++line; ++p; }
            if (p[0] == '-' && p[1] == '-') {
                p += 2;
                if (p[0] == '[' && (p[1] == '[' || p[1] == '=')) { // long comment --[[ ... ]] / --[=[ ... ]=]
                    const char *q = p + 1;
                    int eq = 0;
                    while (*q == '=') { ++eq; ++q; }
                    if (*q == '[') {
                        p = q + 1;
                        for (;;) {
                            if (!*p) fail(line, "unterminated long comment");
                            if (*p == '\n') ++line;
                            if (*p == ']') {
                                const char *r = p + 1;
                                int e2 = 0;
                                while (*r == '=') { ++e2; ++r; }
                                if (e2 == eq && *r == ']') { p = r + 1; break; }
                            }
                            ++p;
                        }
                        continue;
                    }
                }
                while (*p && *p != '\n') ++p;
                continue;
            }
            break;
        }
    }

    Tok next() {
        skip();
        Tok t;
        t.line = line;
        if (!*p) return t;
        const unsigned char c = static_cast<unsigned char>(*p);
        if (std::isalpha(c) || c == '_') {
            const char *s = p;
            while (std::isalnum(static_cast<unsigned char>(*p)) || *p == '_') ++p;
            t.kind = Tok::Name;
            t.text.assign(s, p);
            return t;
        }
        if (std::isdigit(c) || (c == '.' && std::isdigit(static_cast<unsigned char>(p[1])))) {
            const char *s = p;
            bool is_int = true;
            if (p[0] == '0' && (p[1] == 'x' || p[1] == 'X')) {
                char *e = nullptr;
                t.i = std::strtoll(p, &e, 16);
                if (e == p + 2) fail(line, "malformed number");
                p = e;
            } else {
                while (std::isdigit(static_cast<unsigned char>(*p))) ++p;
                if (*p == '.') { is_int = false; ++p; while (std::isdigit(static_cast<unsigned char>(*p))) ++p; }
                if (*p == 'e' || *p == 'E') {
                    is_int = false;
                    ++p;
                    if (*p == '+' || *p == '-') ++p;
                    if (!std::isdigit(static_cast<unsigned char>(*p))) fail(line, "malformed number");
                    while (std::isdigit(static_cast<unsigned char>(*p))) ++p;
                }
                const std::string txt(s, p);
                if (is_int) {
                    errno = 0;
                    t.i = std::strtoll(txt.c_str(), nullptr, 10);
                    if (errno == ERANGE) is_int = false; // Lua: an integer literal that overflows becomes a float
                }
                t.n = std::strtod(txt.c_str(), nullptr); // correctly rounded, like Lua's own strtod
            }
            if (std::isalpha(static_cast<unsigned char>(*p)) || *p == '_') fail(line, "malformed number");
            t.kind = Tok::Number;
            t.is_int = is_int;
            if (is_int) t.n = static_cast<double>(t.i);
            return t;
        }
        if (c == '"' || c == '\'') {
            const char q = *p++;
            t.kind = Tok::String;
            for (;;) {
                if (!*p || *p == '\n') fail(line, "unterminated string");
                if (*p == q) { ++p; break; }
                if (*p == '\\') {
                    ++p;
                    switch (*p) {
                    case 'n': t.text.push_back('\n'); break;
                    case 't': t.text.push_back('\t'); break;
                    case '\\': t.text.push_back('\\'); break;
                    case '"': t.text.push_back('"'); break;
                    case '\'': t.text.push_back('\''); break;
                    default: fail(line, "unsupported escape in string");
                    }
                    ++p;
                    continue;
                }
                t.text.push_back(*p++);
            }
            return t;
        }
        if (c == '[' && (p[1] == '[' || p[1] == '=')) need_vm(line, "long bracket string");
        t.kind = Tok::Sym;
        static const char *two[] = {"==", "~=", "<=", ">=", "//", "..", "::", "<<", ">>"};
        for (const char *s : two)
            if (p[0] == s[0] && p[1] == s[1]) { t.text.assign(s); p += 2; return t; }
        t.text.assign(1, *p++);
        return t;
    }
};

// ---- parser / evaluator of the subset ----------------------------------------------------------------------------
struct RenderCall {
    Value world, camera;
    std::string outfile;
    int line = 0;
};

struct Interp {
    Lexer lx;
    Tok cur;
    std::map<std::string, Value> globals;
    std::vector<RenderCall> renders;

    explicit Interp(const char *text) : lx(text) {
        cur = lx.next();
        auto math = std::make_shared<Table>();
        Value pi; pi.kind = Value::Num; pi.n = 3.141592653589793; // Lua's math.pi (= M_PI)
        Value huge; huge.kind = Value::Num; huge.n = HUGE_VAL;
        math->fields.emplace_back("pi", pi);
        math->fields.emplace_back("huge", huge);
        Value m; m.kind = Value::Tab; m.t = math;
        globals["math"] = m;
    }
    void advance() { cur = lx.next(); }
    bool sym(const char *s) const { return cur.kind == Tok::Sym && cur.text == s; }
    bool name(const char *s) const { return cur.kind == Tok::Name && cur.text == s; }
    void expect(const char *s) {
        if (!sym(s)) fail(cur.line, std::string("expected '") + s + "'" + (cur.kind == Tok::End ? " before the end of the script" : " near '" + cur.text + "'"));
        advance();
    }

    static bool keyword(const std::string &s) {
        static const char *kw[] = {"and", "break", "do", "else", "elseif", "end", "for", "function", "goto", "if", "in", "not", "or", "repeat",
                                   "return", "then", "until", "while"};
        for (const char *k : kw)
            if (s == k) return true;
        return false;
    }

    static Value num(double v) { Value r; r.kind = Value::Num; r.n = v; return r; }
    static Value integer(long long v) { Value r; r.kind = Value::Int; r.i = v; return r; }

    Value arith(const std::string &op, const Value &a, const Value &b, int line) {
        if (!a.is_number() || !b.is_number()) fail(line, "attempt to perform arithmetic on a non-number value");
        if (a.kind == Value::Int && b.kind == Value::Int && (op == "+" || op == "-" || op == "*")) { // Lua 5.3 integer arithmetic wraps
            const unsigned long long x = static_cast<unsigned long long>(a.i), y = static_cast<unsigned long long>(b.i);
            return integer(static_cast<long long>(op == "+" ? x + y : op == "-" ? x - y : x * y));
        }
        const double x = a.number(), y = b.number();
        if (op == "+") return num(x + y);
        if (op == "-") return num(x - y);
        if (op == "*") return num(x * y);
        if (op == "/") return num(x / y);
        if (op == "^") return num(std::pow(x, y));
        if (op == "%") {
            if (a.kind == Value::Int && b.kind == Value::Int) {
                if (b.i == 0) fail(line, "attempt to perform 'n%%0'");
                long long r = a.i % b.i;
                if (r != 0 && ((r ^ b.i) < 0)) r += b.i;
                return integer(r);
            }
            double r = std::fmod(x, y);
            if (r != 0. && ((r < 0.) != (y < 0.))) r += y;
            return num(r);
        }
        need_vm(line, "operator '" + op + "'");
    }

    // precedence climbing: 1: + -   2: * / %   3: unary -   4: ^ (right associative)
    Value expr(int min_prec = 1) {
        Value lhs = unary();
        for (;;) {
            if (cur.kind != Tok::Sym) break;
            const std::string op = cur.text;
            int prec;
            if (op == "+" || op == "-") prec = 1;
            else if (op == "*" || op == "/" || op == "%") prec = 2;
            else if (op == "==" || op == "~=" || op == "<" || op == ">" || op == "<=" || op == ">=" || op == ".." || op == "//" || op == "<<" || op == ">>" ||
                     op == "&" || op == "|" || op == "~" || op == "#")
                need_vm(cur.line, "operator '" + op + "'");
            else break;
            if (prec < min_prec) break;
            const int line = cur.line;
            advance();
            const Value rhs = expr(prec + 1);
            lhs = arith(op, lhs, rhs, line);
        }
        return lhs;
    }
    Value unary() {
        if (sym("-")) {
            const int line = cur.line;
            advance();
            const Value v = unary(); // (-x^y = -(x^y): power() binds tighter, handled below)
            if (!v.is_number()) fail(line, "attempt to negate a non-number value");
            return v.kind == Value::Int ? integer(static_cast<long long>(0ull - static_cast<unsigned long long>(v.i))) : num(-v.n);
        }
        if (name("not") || sym("#") || sym("~")) need_vm(cur.line, "operator '" + cur.text + "'");
        return power();
    }
    Value power() {
        Value base = primary();
        if (sym("^")) {
            const int line = cur.line;
            advance();
            const Value e = unary(); // right associative, binds tighter than unary minus on its left
            return arith("^", base, e, line);
        }
        return base;
    }
    Value primary() {
        const int line = cur.line;
        if (cur.kind == Tok::Number) {
            Value v = cur.is_int ? integer(cur.i) : num(cur.n);
            advance();
            return v;
        }
        if (cur.kind == Tok::String) {
            Value v; v.kind = Value::Str; v.s = cur.text;
            advance();
            return v;
        }
        if (sym("{")) return table();
        if (sym("(")) {
            advance();
            Value v = expr();
            expect(")");
            return v;
        }
        if (cur.kind == Tok::Name) {
            if (cur.text == "nil") { advance(); return Value{}; }
            if (cur.text == "true" || cur.text == "false") { Value v; v.kind = Value::Bool; v.b = cur.text == "true"; advance(); return v; }
            if (cur.text == "function") need_vm(line, "function definition");
            if (keyword(cur.text)) need_vm(line, "'" + cur.text + "'");
            std::string path = cur.text;
            auto g = globals.find(cur.text);
            Value v = g == globals.end() ? Value{} : g->second; // an undefined global is nil, as in Lua
            advance();
            while (sym(".")) {
                advance();
                if (cur.kind != Tok::Name) fail(cur.line, "expected a field name after '.'");
                if (v.kind != Value::Tab) fail(cur.line, "attempt to index a " + std::string(v.kind == Value::Nil ? "nil" : "non-table") + " value (" + path + ")");
                const Value *f = v.t->get(cur.text);
                path += "." + cur.text;
                v = f ? *f : Value{};
                advance();
            }
            if (sym("(") || sym(":") || cur.kind == Tok::String || sym("{")) need_vm(line, "call of '" + path + "'");
            if (sym("[")) need_vm(line, "indexing with '[]'");
            return v;
        }
        fail(line, cur.kind == Tok::End ? "unexpected end of the script" : "unexpected '" + cur.text + "'");
    }
    Value table() {
        expect("{");
        auto t = std::make_shared<Table>();
        long long next_index = 1;
        while (!sym("}")) {
            if (cur.kind == Tok::End) fail(cur.line, "unterminated table constructor");
            if (sym("[")) { // [exp] = exp
                advance();
                const Value k = expr();
                expect("]");
                expect("=");
                const Value v = expr();
                if (k.kind == Value::Str) t->fields.emplace_back(k.s, v);
                else if (k.kind == Value::Int) t->array[k.i] = v;
                else if (k.kind == Value::Num && k.n == std::floor(k.n) && std::fabs(k.n) < 9e15) t->array[static_cast<long long>(k.n)] = v;
                else fail(cur.line, "unsupported table key");
            } else if (cur.kind == Tok::Name && !keyword(cur.text) && cur.text != "nil" && cur.text != "true" && cur.text != "false") {
                // Name '=' exp, or a positional expression that starts with a name: one token of look-ahead
                const Lexer save_lx = lx;
                const Tok save_cur = cur;
                const std::string key = cur.text;
                advance();
                if (sym("=")) {
                    advance();
                    t->fields.emplace_back(key, expr());
                } else {
                    lx = save_lx;
                    cur = save_cur;
                    t->array[next_index++] = expr();
                }
            } else {
                t->array[next_index++] = expr();
            }
            if (sym(",") || sym(";")) { advance(); continue; }
            if (!sym("}")) fail(cur.line, "expected ',' or '}' in table constructor near '" + cur.text + "'");
        }
        advance();
        Value v; v.kind = Value::Tab; v.t = t;
        return v;
    }

    void run() {
        while (cur.kind != Tok::End) {
            if (sym(";")) { advance(); continue; }
            const int line = cur.line;
            if (cur.kind != Tok::Name) fail(line, "unexpected '" + cur.text + "'");
            if (cur.text == "local") {
                advance();
                if (cur.kind != Tok::Name || keyword(cur.text)) need_vm(line, "'local " + cur.text + "'");
            }
            if (cur.text == "function" || keyword(cur.text)) need_vm(line, "'" + cur.text + "'");
            const std::string target = cur.text;
            advance();
            if (sym("=")) {
                advance();
                globals[target] = expr(); // (a `local` at chunk level is visible to the rest of the chunk: same thing here)
                continue;
            }
            if (sym("(")) {
                advance();
                std::vector<Value> args;
                if (!sym(")")) {
                    args.push_back(expr());
                    while (sym(",")) { advance(); args.push_back(expr()); }
                }
                expect(")");
                if (target == "Render") { // lua.rs:57-72: (worldtable, cameratable, outfile)
                    if (args.size() < 3 || args[0].kind != Value::Tab || args[1].kind != Value::Tab || args[2].kind != Value::Str)
                        fail(line, "Render expects (world table, camera table, output file name)");
                    renders.push_back(RenderCall{args[0], args[1], args[2].s, line});
                    continue;
                }
                if (target == "print") continue; // harmless
                need_vm(line, "call of '" + target + "'");
            }
            need_vm(line, "statement starting with '" + target + "'");
        }
    }
};

// ---- the *_from_table functions of lua.rs ------------------------------------------------------------------------
double float_value(const Value *v, const char *what, int line) { // lua.rs:152-158
    if (!v || !v->is_number()) fail(line, std::string("Invalid number: ") + what);
    return v->number();
}
const Table &table_of(const Value *v, const char *what, int line) {
    if (!v || v->kind != Value::Tab) fail(line, std::string(what) + " must be a table");
    return *v->t;
}
void xyz(const Table &t, const char *a, const char *b, const char *c, double out[3], const char *what, int line) { // lua.rs:91-107
    out[0] = float_value(t.get(a), what, line);
    out[1] = float_value(t.get(b), what, line);
    out[2] = float_value(t.get(c), what, line);
}

void transform_from_table(const Table &t, double m[16], int line) { // lua.rs:257-291: this order, whatever the writing order
    rtc_matrix_identity(m);
    if (const Value *v = t.get("rotate_x")) rtc_matrix_rotation_x(m, float_value(v, "rotate_x", line), m);
    if (const Value *v = t.get("rotate_y")) rtc_matrix_rotation_y(m, float_value(v, "rotate_y", line), m);
    if (const Value *v = t.get("rotate_z")) rtc_matrix_rotation_z(m, float_value(v, "rotate_z", line), m);
    if (const Value *v = t.get("scale")) {
        const double s = float_value(v, "scale", line);
        rtc_matrix_scaling(m, s, s, s, m);
    }
    if (const Value *v = t.get("position")) {
        double p[3];
        xyz(table_of(v, "position", line), "x", "y", "z", p, "position", line);
        rtc_matrix_translation(m, p[0], p[1], p[2], m);
    }
}

void pattern_from_table(const Table &t, rtc_material &mat, int line) { // lua.rs:109-143
    double xf[16];
    transform_from_table(t, xf, line);
    const Value *type = t.get("type");
    if (!type || type->kind != Value::Str) fail(line, "pattern needs a type");
    uint32_t kind;
    double a[3] = {1., 1., 1.}, b[3] = {0., 0., 0.}; // "grid": GridPattern::new(Color::WHITE, Color::BLACK)
    if (type->s == "checks" || type->s == "stripes") {
        kind = type->s == "checks" ? RTC_PATTERN_CHECKER : RTC_PATTERN_STRIPE;
        xyz(table_of(t.get("color_a"), "color_a", line), "r", "g", "b", a, "color_a", line);
        xyz(table_of(t.get("color_b"), "color_b", line), "r", "g", "b", b, "color_b", line);
    } else if (type->s == "grid") {
        kind = RTC_PATTERN_GRID;
    } else {
        fail(line, "invalid pattern type: " + type->s);
    }
    const rtc_status st = rtc_material_set_pattern(&mat, kind, a, b, xf);
    if (st != RTC_OK) fail(line, std::string("pattern transform: ") + rtc_strerror(st)); // Matrix::inverse panics, transform.rs:177
}

void material_from_table(const Table &shape, rtc_material &mat, int line) { // lua.rs:186-239
    rtc_material_default(&mat);
    if (const Value *mv = shape.get("material")) {
        const Table &mt = table_of(mv, "material", line);
        if (!mt.array.empty()) fail(line, "Invalid material property: " + std::to_string(mt.array.begin()->first));
        for (const auto &kv : mt.fields) {
            const std::string &key = kv.first;
            const Value *v = &kv.second;
            if (v->kind == Value::Nil) continue; // (a nil field does not exist)
            if (key == "ambient") mat.ambient = float_value(v, "ambient", line);
            else if (key == "diffuse") mat.diffuse = float_value(v, "diffuse", line);
            else if (key == "specular") mat.specular = float_value(v, "specular", line);
            else if (key == "shininess") mat.shininess = float_value(v, "shininess", line);
            else if (key == "reflectiveness") mat.reflective = float_value(v, "reflectiveness", line);
            else if (key == "transparency") mat.transparency = float_value(v, "transparency", line);
            else if (key == "refractive_index") mat.refractive_index = float_value(v, "refractive_index", line);
            else if (key == "color") {
                if (v->kind != Value::Tab) fail(line, "invalid color");
                xyz(*v->t, "r", "g", "b", mat.color, "color", line);
                mat.has_color = 1;
            } else if (key == "pattern") {
                if (v->kind != Value::Tab) fail(line, "invali pattern");
                pattern_from_table(*v->t, mat, line);
            } else {
                fail(line, "Invalid material property: " + key);
            }
        }
    }
    // shape-level overrides: `color` if it is a table (a malformed one is ignored, and then `pattern` is not even looked at), else `pattern`
    const Value *c = shape.get("color");
    if (c && c->kind == Value::Tab) {
        const Value *r = c->t->get("r"), *g = c->t->get("g"), *b = c->t->get("b");
        if (r && g && b && r->is_number() && g->is_number() && b->is_number()) {
            mat.color[0] = r->number(); mat.color[1] = g->number(); mat.color[2] = b->number();
            mat.has_color = 1;
        }
    } else if (const Value *p = shape.get("pattern")) {
        if (p->kind == Value::Tab) {
            rtc_material tmp = mat;
            try {
                pattern_from_table(*p->t, tmp, line);
                mat = tmp;
            } catch (const LuaError &) { // `if let Ok(pattern) = pattern_from_table(..)`: errors are dropped
            }
        }
    }
}

struct Scene {
    std::vector<rtc_shape> shapes;
    rtc_light light;
    rtc_camera camera;
};

long long integer_value(const Value *v, const char *what, long long lo, long long hi, int line) { // u32_value / u8_value lua.rs:160-183
    if (!v || v->kind != Value::Int) fail(line, std::string("Invalid number: ") + what + " must be a Lua integer");
    if (v->i < lo || v->i > hi) fail(line, std::string("Number out of bounds: ") + what);
    return v->i;
}

void camera_from_table(const Table &t, rtc_camera &cam, int line) { // lua.rs:241-255
    double position[3], lookat[3], up[3], view[16];
    xyz(table_of(t.get("position"), "camera position", line), "x", "y", "z", position, "position", line);
    xyz(table_of(t.get("lookat"), "camera lookat", line), "x", "y", "z", lookat, "lookat", line);
    xyz(table_of(t.get("up"), "camera up", line), "x", "y", "z", up, "up", line);
    const long long w = integer_value(t.get("screenwidth"), "screenwidth", 0, 2147483647LL, line);
    const long long h = integer_value(t.get("screenheight"), "screenheight", 0, 2147483647LL, line);
    const double fov = float_value(t.get("fov"), "fov", line);
    rtc_view_transform(position, lookat, up, view);
    const rtc_status st = rtc_camera_init(static_cast<uint32_t>(w), static_cast<uint32_t>(h), fov, view, &cam);
    if (st != RTC_OK) fail(line, std::string("camera: ") + rtc_strerror(st));
    if (const Value *s = t.get("samples")) cam.samples = static_cast<uint32_t>(integer_value(s, "samples", 0, 255, line));
}

void world_from_table(const Table &t, Scene &sc, int line) { // lua.rs:293-330
    const Table &lights = table_of(t.get("lights"), "world.lights", line);
    const Table &l1 = table_of(lights.at(1), "world.lights[1]", line); // lights_from_table: only the first
    xyz(table_of(l1.get("color"), "light color", line), "r", "g", "b", sc.light.intensity, "light color", line);
    xyz(table_of(l1.get("position"), "light position", line), "x", "y", "z", sc.light.position, "light position", line);
    const Table &shapes = table_of(t.get("shapes"), "world.shapes", line);
    for (long long k = 1;; ++k) { // sequence_values: 1, 2, ... until the first nil
        const Value *sv = shapes.at(k);
        if (!sv) break;
        const Table &st = table_of(sv, "a shape", line);
        const Value *type = st.get("type");
        if (!type || type->kind != Value::Str) fail(line, "shape " + std::to_string(k) + " needs a type");
        uint32_t kind;
        if (type->s == "sphere") kind = RTC_SPHERE;
        else if (type->s == "plane") kind = RTC_PLANE;
        else if (type->s == "cube") kind = RTC_CUBE;
        else fail(line, "Invalid shape type: " + type->s);
        rtc_material mat;
        material_from_table(st, mat, line);
        double xf[16];
        transform_from_table(st, xf, line);
        rtc_shape s;
        const rtc_status rs = rtc_shape_init(kind, xf, &mat, &s);
        if (rs != RTC_OK) fail(line, type->s + " " + std::to_string(k) + ": " + rtc_strerror(rs));
        s.world_id = static_cast<uint32_t>(sc.shapes.size()) + 1; // World::add_shape shape.rs:661-667
        sc.shapes.push_back(s);
    }
}

void set_err(char *errbuf, size_t len, const std::string &msg) {
    if (errbuf && len) std::snprintf(errbuf, len, "%s", msg.c_str());
}

} // namespace

extern "C" {

rtc_status rtc_scene_load_lua(const char *text, uint32_t render_index, rtc_shape **shapes_out, uint32_t *n_out, rtc_light *light_out,
                              rtc_camera *camera_out, char *outfile, size_t outfile_len, uint32_t *renders_out, char *errbuf,
                              size_t errbuf_len) {
    if (!text || !shapes_out || !n_out || !light_out || !camera_out) return RTC_ERR_ARG;
    *shapes_out = nullptr;
    *n_out = 0;
    if (renders_out) *renders_out = 0;
    if (outfile && outfile_len) outfile[0] = 0;
    try {
        Interp in(text);
        in.run();
        if (renders_out) *renders_out = static_cast<uint32_t>(in.renders.size());
        RenderCall rc;
        if (in.renders.empty()) { // no Render call: the globals `world` and `camera`, if the script defines them
            auto w = in.globals.find("world"), c = in.globals.find("camera");
            if (render_index != 0 || w == in.globals.end() || c == in.globals.end() || w->second.kind != Value::Tab || c->second.kind != Value::Tab)
                fail(in.cur.line, "the script calls Render(world, camera, file) " + std::to_string(in.renders.size()) + " time(s) and defines no global world / camera tables");
            rc.world = w->second;
            rc.camera = c->second;
        } else {
            if (render_index >= in.renders.size()) fail(in.cur.line, "the script calls Render " + std::to_string(in.renders.size()) + " time(s)");
            rc = in.renders[render_index];
        }
        Scene sc;
        world_from_table(*rc.world.t, sc, rc.line);     // the reference converts the world first, then the camera (lua.rs:60-63)
        camera_from_table(*rc.camera.t, sc.camera, rc.line);
        if (outfile && outfile_len) std::snprintf(outfile, outfile_len, "%s", rc.outfile.c_str());
        const size_t bytes = sizeof(rtc_shape) * (sc.shapes.empty() ? 1 : sc.shapes.size());
        rtc_shape *arr = static_cast<rtc_shape *>(std::malloc(bytes));
        if (!arr) return RTC_ERR_NOMEM;
        if (!sc.shapes.empty()) std::memcpy(arr, sc.shapes.data(), sizeof(rtc_shape) * sc.shapes.size());
        *shapes_out = arr;
        *n_out = static_cast<uint32_t>(sc.shapes.size());
        *light_out = sc.light;
        *camera_out = sc.camera;
        return RTC_OK;
    } catch (const LuaError &e) {
        set_err(errbuf, errbuf_len, e.what());
        return RTC_ERR_PARSE;
    } catch (const std::bad_alloc &) {
        return RTC_ERR_NOMEM;
    } catch (...) {
        set_err(errbuf, errbuf_len, "internal error");
        return RTC_ERR_PARSE;
    }
}

rtc_status rtc_scene_load_lua_file(const char *path, uint32_t render_index, rtc_shape **shapes_out, uint32_t *n_out, rtc_light *light_out,
                                   rtc_camera *camera_out, char *outfile, size_t outfile_len, uint32_t *renders_out, char *errbuf,
                                   size_t errbuf_len) {
    if (!path) return RTC_ERR_ARG;
    std::FILE *f = std::fopen(path, "rb");
    if (!f) { set_err(errbuf, errbuf_len, std::string("cannot open ") + path); return RTC_ERR_IO; }
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    std::fclose(f);
    return rtc_scene_load_lua(text.c_str(), render_index, shapes_out, n_out, light_out, camera_out, outfile, outfile_len, renders_out, errbuf, errbuf_len);
}

} // extern "C"
