// host_lua.cpp — [host] the reference's Lua front-end (ch1/src/lua.rs) without a Lua library: a small tree-walking
// interpreter for the part of Lua 5.3 its scene scripts are written in, and lua.rs's *_from_table rules on top of it.
//
// The reference embeds Lua 5.3 (rlua 0.17, Cargo.toml:19) and gives a script three things (lua.rs:50-91):
//   Render(world, camera, outfile)            world_from_table + camera_from_table, Camera::render_async, write the file
//   StartAnimation(outfile) -> encoder        a GIF encoder object with
//   encoder:AddFrame(world, camera)             the same conversion + render_async, one frame appended   (lua.rs:34-41)
//   encoder:Finish()                            (does nothing, lua.rs:43-45)
// Its scripts (ch1/jamis.lua, ex1.lua, ex2.lua + functions.lua) are table constructors, a few functions, numeric for
// loops, field assignments, math.sin / math.cos / math.random, table.insert, string.format, print and require.
// This image has no Lua, so the library carries its own interpreter of that language level:
//   statements   local, assignment (multiple), calls, do, while, repeat, if / elseif / else, numeric and generic for,
//                function / local function (methods with ':'), return, break
//   expressions  nil true false numbers (integer / float subtypes as in 5.3) strings tables functions (closures, varargs);
//                + - * / // % ^ .. # == ~= < <= > >= and or not; indexing, calls, method calls
//   library      print (collected, not written to stdout), type, tostring, tonumber, ipairs, pairs, next, select, assert,
//                error, pcall, require (files beside the script), math.*, string.format / len / sub / rep / upper /
//                lower, table.insert / remove / concat / unpack
//   not there    metatables, coroutines, goto, bitwise operators, integer-for overflow corner cases, io / os
// Every Render and AddFrame call is converted AT THE CALL (the tables are mutable: ex2.lua moves camera.position between
// frames) and recorded as a job; the caller renders the jobs (rtc_lua_run → rtc_lua_program_job). A script cannot loop
// for ever: it runs under a step budget.
//
// math.random follows Lua 5.3's lmathlib.c on POSIX: l_rand() = random(), L_RANDMAX = 2^31-1, randomseed(n) =
// srandom((unsigned)n) followed by one discarded draw, random() = r / 2^31, random(m, n) = m + floor(r / 2^31 * (n-m+1)).
// glibc's random() (the TYPE_3 additive-feedback generator, x[i] = x[i-3] + x[i-31]) is restated here so that a script's
// random world does not depend on the C library it runs on; tests/test_host_cpu.py checks the restatement against this
// machine's srandom()/random().
//
// The *_from_table rules, statement for statement:
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
// A Lua error — syntax, runtime, or a table lua.rs would reject — is RTC_ERR_PARSE with the message (the reference
// unwrap()s: it panics).
// PARITY UNPINNED: the reference holds no test of its Lua path; this follows lua.rs and the Lua 5.3
// manual by reading.
#include "rtc.h"

#include <cctype>
#include <cerrno>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct LuaError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] void fail(int line, const std::string &msg) { throw LuaError("line " + std::to_string(line) + ": " + msg); }

struct Table;
struct Function;
struct Value {
    enum Kind { Nil, Bool, Int, Num, Str, Tab, Fun } kind = Nil;
    bool b = false;
    long long i = 0;
    double n = 0.;
    std::string s;
    std::shared_ptr<Table> t;
    std::shared_ptr<Function> f;
    bool is_number() const { return kind == Int || kind == Num; }
    double number() const { return kind == Int ? static_cast<double>(i) : n; }
    bool truthy() const { return !(kind == Nil || (kind == Bool && !b)); }
    static Value boolean(bool v) { Value r; r.kind = Bool; r.b = v; return r; }
    static Value integer(long long v) { Value r; r.kind = Int; r.i = v; return r; }
    static Value num(double v) { Value r; r.kind = Num; r.n = v; return r; }
    static Value str(std::string v) { Value r; r.kind = Str; r.s = std::move(v); return r; }
    static Value table(std::shared_ptr<Table> v) { Value r; r.kind = Tab; r.t = std::move(v); return r; }
};
typedef std::vector<Value> Values;

const char *type_name(const Value &v) {
    switch (v.kind) {
    case Value::Nil: return "nil";
    case Value::Bool: return "boolean";
    case Value::Int: case Value::Num: return "number";
    case Value::Str: return "string";
    case Value::Tab: return "table";
    default: return "function";
    }
}

// Keys: strings and integers (a float key with an integer value is that integer, as in Lua). String keys keep their
// order of first assignment, which is the order pairs() and material_from_table's loop see.
struct Table {
    std::vector<std::pair<std::string, Value>> fields;
    std::unordered_map<std::string, size_t> index;
    std::map<long long, Value> array;
    // memory budget of the script that made the table (Interp::mem): 4 units per live table + 1 per entry, returned when it dies
    std::shared_ptr<long long> mem;
    long long units = 0;
    void charge(long long n) { units += n; if (mem) *mem += n; }
    Table() = default;
    Table(const Table &) = delete;
    Table &operator=(const Table &) = delete;
    ~Table() { if (mem) *mem -= units; }
    const Value *get(const std::string &k) const {
        auto it = index.find(k);
        if (it == index.end()) return nullptr;
        const Value &v = fields[it->second].second;
        return v.kind != Value::Nil ? &v : nullptr;
    }
    void set(const std::string &k, const Value &v) {
        auto it = index.find(k);
        if (it != index.end()) { fields[it->second].second = v; return; }
        if (v.kind == Value::Nil) return;
        index.emplace(k, fields.size());
        fields.emplace_back(k, v);
        charge(1);
    }
    const Value *at(long long k) const {
        auto it = array.find(k);
        return it != array.end() ? &it->second : nullptr;
    }
    void seti(long long k, const Value &v) {
        if (v.kind == Value::Nil) { charge(-static_cast<long long>(array.erase(k))); return; }
        auto it = array.find(k);
        if (it != array.end()) { it->second = v; return; }
        array.emplace(k, v);
        charge(1);
    }
    long long length() const { // a border: t[n] ~= nil and t[n+1] == nil, counted from 1
        if (array.empty()) return 0;
        if (array.begin()->first == 1 && array.rbegin()->first == static_cast<long long>(array.size())) return array.rbegin()->first; // a plain sequence
        long long n = 0;
        for (auto it = array.find(1); it != array.end() && it->first == n + 1; ++it) ++n;
        return n;
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

bool scan_number(const char *&p, Tok &t) { // Lua numeral at p (decimal or hex integer); false = malformed
    const char *s = p;
    bool is_int = true;
    if (p[0] == '0' && (p[1] == 'x' || p[1] == 'X')) {
        char *e = nullptr;
        t.i = static_cast<long long>(std::strtoull(p, &e, 16)); // hex integers wrap modulo 2^64
        if (e == p + 2) return false;
        p = e;
    } else {
        while (std::isdigit(static_cast<unsigned char>(*p))) ++p;
        if (*p == '.') { is_int = false; ++p; while (std::isdigit(static_cast<unsigned char>(*p))) ++p; }
        if (p == s || (p == s + 1 && *s == '.')) return false;
        if (*p == 'e' || *p == 'E') {
            is_int = false;
            ++p;
            if (*p == '+' || *p == '-') ++p;
            if (!std::isdigit(static_cast<unsigned char>(*p))) return false;
            while (std::isdigit(static_cast<unsigned char>(*p))) ++p;
        }
        const std::string txt(s, p);
        if (is_int) {
            errno = 0;
            t.i = std::strtoll(txt.c_str(), nullptr, 10);
            if (errno == ERANGE) is_int = false; // an integer literal that overflows becomes a float
        }
        t.n = std::strtod(txt.c_str(), nullptr); // correctly rounded, like Lua's own strtod
    }
    t.kind = Tok::Number;
    t.is_int = is_int;
    if (is_int) t.n = static_cast<double>(t.i);
    return true;
}

struct Lexer {
    const char *p;
    int line = 1;
    explicit Lexer(const char *text) : p(text) {}

    // at p: '[' '='* '[' ... ; reads the long bracket's body (comment or string)
    bool long_bracket(std::string *out) {
        const char *q = p + 1;
        int eq = 0;
        while (*q == '=') { ++eq; ++q; }
        if (*q != '[') return false;
        p = q + 1;
        if (*p == '\r') ++p;
        if (*p == '\n') { ++line; ++p; } // a newline right after the opening bracket is skipped
        for (;;) {
            if (!*p) fail(line, "unfinished long string / comment");
            if (*p == ']') {
                const char *r = p + 1;
                int e2 = 0;
                while (*r == '=') { ++e2; ++r; }
                if (e2 == eq && *r == ']') { p = r + 1; return true; }
            }
            if (*p == '\n') ++line;
            if (out) out->push_back(*p);
            ++p;
        }
    }

    void skip() {
        for (;;) {
            while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\f' || *p == '\v') { if (*p == '\n') ++line; ++p; }
            if (p[0] == '-' && p[1] == '-') {
                p += 2;
                if (p[0] == '[' && long_bracket(nullptr)) continue;
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
            const char *s0 = p;
            const bool ok = scan_number(p, t);
            if (!ok || std::isalnum(static_cast<unsigned char>(*p)) || *p == '_' || *p == '.') {
                while (std::isalnum(static_cast<unsigned char>(*p)) || *p == '_' || *p == '.') ++p;
                fail(line, "malformed number near '" + std::string(s0, p) + "'");
            }
            t.text.assign(s0, p);
            return t;
        }
        if (c == '"' || c == '\'') {
            const char q = *p++;
            t.kind = Tok::String;
            for (;;) {
                if (!*p || *p == '\n') fail(line, "unfinished string");
                if (*p == q) { ++p; break; }
                if (*p == '\\') {
                    ++p;
                    switch (*p) {
                    case 'n': t.text.push_back('\n'); break;
                    case 't': t.text.push_back('\t'); break;
                    case 'r': t.text.push_back('\r'); break;
                    case 'a': t.text.push_back('\a'); break;
                    case 'b': t.text.push_back('\b'); break;
                    case 'f': t.text.push_back('\f'); break;
                    case 'v': t.text.push_back('\v'); break;
                    case '\\': t.text.push_back('\\'); break;
                    case '"': t.text.push_back('"'); break;
                    case '\'': t.text.push_back('\''); break;
                    case '\n': t.text.push_back('\n'); ++line; break;
                    default:
                        if (std::isdigit(static_cast<unsigned char>(*p))) { // \ddd
                            int v = 0, k = 0;
                            while (k < 3 && std::isdigit(static_cast<unsigned char>(*p))) { v = v * 10 + (*p - '0'); ++p; ++k; }
                            if (v > 255) fail(line, "decimal escape too large");
                            t.text.push_back(static_cast<char>(v));
                            continue;
                        }
                        fail(line, "invalid escape sequence in string");
                    }
                    ++p;
                    continue;
                }
                t.text.push_back(*p++);
            }
            return t;
        }
        if (c == '[' && (p[1] == '[' || p[1] == '=')) {
            const char *save = p;
            std::string body;
            if (long_bracket(&body)) { t.kind = Tok::String; t.text = body; return t; }
            p = save;
        }
        t.kind = Tok::Sym;
        if (p[0] == '.' && p[1] == '.' && p[2] == '.') { t.text = "..."; p += 3; return t; }
        static const char *two[] = {"==", "~=", "<=", ">=", "//", "..", "::", "<<", ">>"};
        for (const char *s : two)
            if (p[0] == s[0] && p[1] == s[1]) { t.text.assign(s); p += 2; return t; }
        t.text.assign(1, *p++);
        return t;
    }
};

// ---- syntax tree ---------------------------------------------------------------------------------------------------
struct Expr;
struct Stmt;
typedef std::unique_ptr<Expr> ExprP;
typedef std::vector<std::unique_ptr<Stmt>> Block;

struct FuncBody {
    std::vector<std::string> params;
    bool vararg = false;
    Block body;
    std::string name;
    int line = 0;
};

enum Op { OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_IDIV, OP_MOD, OP_POW, OP_CONCAT, OP_EQ, OP_NE, OP_LT, OP_LE, OP_GT, OP_GE, OP_AND, OP_OR,
          OP_NEG, OP_NOT, OP_LEN };

struct TableItem {
    enum Kind { Positional, Named, Keyed } kind = Positional;
    std::string name;
    ExprP key, value;
};

struct Expr {
    enum Kind { Const, Vararg, Func, Name, Index, Call, Method, Binary, Unary, TableCons, Paren } kind = Const;
    int line = 0;
    Value value;                     // Const
    std::string name;                // Name; Method: the method's name
    Op op = OP_ADD;                  // Binary / Unary
    ExprP a, b;                      // Index: a[b]; Call: a(args); Method: a:name(args); Binary: a op b; Unary / Paren: a
    std::vector<ExprP> args;
    std::vector<TableItem> items;    // TableCons
    std::shared_ptr<FuncBody> func;  // Func
};

struct Stmt {
    enum Kind { Local, Assign, CallStmt, Do, While, Repeat, If, NumFor, GenFor, LocalFunc, Return, Break } kind = Do;
    int line = 0;
    std::vector<std::string> names;  // Local / GenFor: variables; NumFor: names[0]; LocalFunc: names[0]
    std::vector<ExprP> targets;      // Assign
    std::vector<ExprP> exprs;        // Local / Assign: right-hand sides; Return: values; GenFor: explist; NumFor: start, stop[, step];
                                     // If: the conditions; While / Repeat: exprs[0]; CallStmt: exprs[0]
    std::vector<Block> blocks;       // Do / While / Repeat / NumFor / GenFor: blocks[0]; If: one per condition (+ the else block)
    bool has_else = false;
    std::shared_ptr<FuncBody> func;  // LocalFunc
};

struct Parser {
    Lexer lx;
    Tok cur, ahead;
    bool has_ahead = false;
    int depth = 0;

    explicit Parser(const char *text) : lx(text) { cur = lx.next(); }
    void advance() {
        if (has_ahead) { cur = ahead; has_ahead = false; }
        else cur = lx.next();
    }
    const Tok &peek() {
        if (!has_ahead) { ahead = lx.next(); has_ahead = true; }
        return ahead;
    }
    bool sym(const char *s) const { return cur.kind == Tok::Sym && cur.text == s; }
    bool kw(const char *s) const { return cur.kind == Tok::Name && cur.text == s; }
    std::string near() const { return cur.kind == Tok::End ? "<eof>" : "'" + cur.text + "'"; }
    void expect_sym(const char *s) {
        if (!sym(s)) fail(cur.line, std::string("'") + s + "' expected near " + near());
        advance();
    }
    void expect_kw(const char *s, const char *opener, int open_line) {
        if (!kw(s)) fail(cur.line, std::string("'") + s + "' expected (to close '" + opener + "' at line " + std::to_string(open_line) + ") near " + near());
        advance();
    }
    static bool keyword(const std::string &s) {
        static const char *kws[] = {"and", "break", "do", "else", "elseif", "end", "false", "for", "function", "goto", "if", "in", "local", "nil",
                                    "not", "or", "repeat", "return", "then", "true", "until", "while"};
        for (const char *k : kws)
            if (s == k) return true;
        return false;
    }
    std::string expect_name() {
        if (cur.kind != Tok::Name || keyword(cur.text)) fail(cur.line, "<name> expected near " + near());
        std::string s = cur.text;
        advance();
        return s;
    }
    struct Depth {
        Parser &p;
        explicit Depth(Parser &q) : p(q) { if (++p.depth > 180) fail(p.cur.line, "chunk has too many syntax levels"); }
        ~Depth() { --p.depth; }
    };

    bool block_end() const { return cur.kind == Tok::End || kw("end") || kw("else") || kw("elseif") || kw("until"); }

    Block block() {
        Depth d(*this);
        Block b;
        while (!block_end()) {
            if (sym(";")) { advance(); continue; }
            if (kw("return")) {
                auto s = std::make_unique<Stmt>();
                s->kind = Stmt::Return;
                s->line = cur.line;
                advance();
                if (!block_end() && !sym(";")) s->exprs = exprlist();
                if (sym(";")) advance();
                b.push_back(std::move(s));
                if (!block_end()) fail(cur.line, "'end' expected after 'return' near " + near());
                break;
            }
            b.push_back(statement());
        }
        return b;
    }

    std::vector<ExprP> exprlist() {
        std::vector<ExprP> v;
        v.push_back(expr());
        while (sym(",")) { advance(); v.push_back(expr()); }
        return v;
    }

    std::shared_ptr<FuncBody> funcbody(const std::string &name, bool method, int line) {
        auto f = std::make_shared<FuncBody>();
        f->name = name;
        f->line = line;
        if (method) f->params.push_back("self");
        expect_sym("(");
        if (!sym(")")) {
            for (;;) {
                if (sym("...")) { advance(); f->vararg = true; break; }
                f->params.push_back(expect_name());
                if (!sym(",")) break;
                advance();
            }
        }
        expect_sym(")");
        f->body = block();
        expect_kw("end", "function", line);
        return f;
    }

    std::unique_ptr<Stmt> statement() {
        auto s = std::make_unique<Stmt>();
        s->line = cur.line;
        const int line = cur.line;
        if (kw("if")) {
            s->kind = Stmt::If;
            advance();
            s->exprs.push_back(expr());
            expect_kw("then", "if", line);
            s->blocks.push_back(block());
            while (kw("elseif")) {
                advance();
                s->exprs.push_back(expr());
                expect_kw("then", "elseif", line);
                s->blocks.push_back(block());
            }
            if (kw("else")) { advance(); s->blocks.push_back(block()); s->has_else = true; }
            expect_kw("end", "if", line);
            return s;
        }
        if (kw("while")) {
            s->kind = Stmt::While;
            advance();
            s->exprs.push_back(expr());
            expect_kw("do", "while", line);
            s->blocks.push_back(block());
            expect_kw("end", "while", line);
            return s;
        }
        if (kw("do")) {
            s->kind = Stmt::Do;
            advance();
            s->blocks.push_back(block());
            expect_kw("end", "do", line);
            return s;
        }
        if (kw("for")) {
            advance();
            s->names.push_back(expect_name());
            if (sym("=")) {
                s->kind = Stmt::NumFor;
                advance();
                s->exprs.push_back(expr());
                expect_sym(",");
                s->exprs.push_back(expr());
                if (sym(",")) { advance(); s->exprs.push_back(expr()); }
            } else {
                s->kind = Stmt::GenFor;
                while (sym(",")) { advance(); s->names.push_back(expect_name()); }
                if (!kw("in")) fail(cur.line, "'=' or 'in' expected near " + near());
                advance();
                s->exprs = exprlist();
            }
            expect_kw("do", "for", line);
            s->blocks.push_back(block());
            expect_kw("end", "for", line);
            return s;
        }
        if (kw("repeat")) {
            s->kind = Stmt::Repeat;
            advance();
            s->blocks.push_back(block());
            expect_kw("until", "repeat", line);
            s->exprs.push_back(expr());
            return s;
        }
        if (kw("function")) { // function a.b.c:m(...) ... end  ==  a.b.c.m = function (self, ...) ... end
            advance();
            auto target = std::make_unique<Expr>();
            target->kind = Expr::Name;
            target->line = line;
            target->name = expect_name();
            std::string full = target->name;
            bool method = false;
            while (sym(".") || sym(":")) {
                const bool colon = sym(":");
                advance();
                auto key = std::make_unique<Expr>();
                key->kind = Expr::Const;
                key->line = cur.line;
                key->value = Value::str(expect_name());
                full += (colon ? ":" : ".") + key->value.s;
                auto idx = std::make_unique<Expr>();
                idx->kind = Expr::Index;
                idx->line = line;
                idx->a = std::move(target);
                idx->b = std::move(key);
                target = std::move(idx);
                if (colon) { method = true; break; }
            }
            auto fe = std::make_unique<Expr>();
            fe->kind = Expr::Func;
            fe->line = line;
            fe->func = funcbody(full, method, line);
            s->kind = Stmt::Assign;
            s->targets.push_back(std::move(target));
            s->exprs.push_back(std::move(fe));
            return s;
        }
        if (kw("local")) {
            advance();
            if (kw("function")) {
                advance();
                s->kind = Stmt::LocalFunc;
                s->names.push_back(expect_name());
                s->func = funcbody(s->names[0], false, line);
                return s;
            }
            s->kind = Stmt::Local;
            s->names.push_back(expect_name());
            while (sym(",")) { advance(); s->names.push_back(expect_name()); }
            if (sym("=")) { advance(); s->exprs = exprlist(); }
            return s;
        }
        if (kw("break")) { s->kind = Stmt::Break; advance(); return s; }
        if (kw("goto") || sym("::")) fail(line, "goto and labels are not supported by this interpreter");
        // exprstat: a call, or an assignment
        ExprP e = suffixed();
        if (sym("=") || sym(",")) {
            s->kind = Stmt::Assign;
            s->targets.push_back(std::move(e));
            while (sym(",")) { advance(); s->targets.push_back(suffixed()); }
            expect_sym("=");
            s->exprs = exprlist();
            for (const auto &t : s->targets)
                if (t->kind != Expr::Name && t->kind != Expr::Index) fail(line, "syntax error: cannot assign to this expression");
            return s;
        }
        if (e->kind != Expr::Call && e->kind != Expr::Method) fail(line, "syntax error near " + near());
        s->kind = Stmt::CallStmt;
        s->exprs.push_back(std::move(e));
        return s;
    }

    // ---- expressions: or < and < comparison < .. < + - < * / // % < unary < ^
    static int binary_prec(const Tok &t, Op &op, bool &right) {
        right = false;
        if (t.kind == Tok::Name) {
            if (t.text == "or") { op = OP_OR; return 1; }
            if (t.text == "and") { op = OP_AND; return 2; }
            return 0;
        }
        if (t.kind != Tok::Sym) return 0;
        const std::string &s = t.text;
        if (s == "<") { op = OP_LT; return 3; }
        if (s == ">") { op = OP_GT; return 3; }
        if (s == "<=") { op = OP_LE; return 3; }
        if (s == ">=") { op = OP_GE; return 3; }
        if (s == "~=") { op = OP_NE; return 3; }
        if (s == "==") { op = OP_EQ; return 3; }
        if (s == "..") { op = OP_CONCAT; right = true; return 9; }
        if (s == "+") { op = OP_ADD; return 10; }
        if (s == "-") { op = OP_SUB; return 10; }
        if (s == "*") { op = OP_MUL; return 11; }
        if (s == "/") { op = OP_DIV; return 11; }
        if (s == "//") { op = OP_IDIV; return 11; }
        if (s == "%") { op = OP_MOD; return 11; }
        if (s == "^") { op = OP_POW; right = true; return 14; }
        return 0;
    }
    static const int UNARY_PREC = 12;

    ExprP expr(int limit = 0) {
        Depth d(*this);
        ExprP lhs;
        if (kw("not") || sym("-") || sym("#")) {
            auto u = std::make_unique<Expr>();
            u->kind = Expr::Unary;
            u->line = cur.line;
            u->op = kw("not") ? OP_NOT : sym("-") ? OP_NEG : OP_LEN;
            advance();
            u->a = expr(UNARY_PREC);
            lhs = std::move(u);
        } else if (sym("~") || sym("&") || sym("|") || sym("<<") || sym(">>")) {
            fail(cur.line, "bitwise operators are not supported by this interpreter");
        } else {
            lhs = simple();
        }
        for (;;) {
            if (sym("&") || sym("|") || sym("~") || sym("<<") || sym(">>")) fail(cur.line, "bitwise operators are not supported by this interpreter");
            Op op;
            bool right;
            const int prec = binary_prec(cur, op, right);
            if (prec == 0 || prec <= limit) break;
            auto bin = std::make_unique<Expr>();
            bin->kind = Expr::Binary;
            bin->line = cur.line;
            bin->op = op;
            advance();
            bin->a = std::move(lhs);
            bin->b = expr(right ? prec - 1 : prec);
            lhs = std::move(bin);
        }
        return lhs;
    }

    ExprP constant(Value v, int line) {
        auto e = std::make_unique<Expr>();
        e->kind = Expr::Const;
        e->line = line;
        e->value = std::move(v);
        return e;
    }

    ExprP simple() {
        const int line = cur.line;
        if (cur.kind == Tok::Number) {
            ExprP e = constant(cur.is_int ? Value::integer(cur.i) : Value::num(cur.n), line);
            advance();
            return e;
        }
        if (cur.kind == Tok::String) {
            ExprP e = constant(Value::str(cur.text), line);
            advance();
            return e;
        }
        if (kw("nil")) { advance(); return constant(Value{}, line); }
        if (kw("true")) { advance(); return constant(Value::boolean(true), line); }
        if (kw("false")) { advance(); return constant(Value::boolean(false), line); }
        if (sym("...")) {
            advance();
            auto e = std::make_unique<Expr>();
            e->kind = Expr::Vararg;
            e->line = line;
            return e;
        }
        if (sym("{")) return tablecons();
        if (kw("function")) {
            advance();
            auto e = std::make_unique<Expr>();
            e->kind = Expr::Func;
            e->line = line;
            e->func = funcbody("anonymous function", false, line);
            return e;
        }
        return suffixed();
    }

    ExprP primary() {
        const int line = cur.line;
        if (sym("(")) {
            advance();
            auto e = std::make_unique<Expr>();
            e->kind = Expr::Paren; // (f()) is exactly one value
            e->line = line;
            e->a = expr();
            expect_sym(")");
            return e;
        }
        if (cur.kind == Tok::Name && !keyword(cur.text)) {
            auto e = std::make_unique<Expr>();
            e->kind = Expr::Name;
            e->line = line;
            e->name = cur.text;
            advance();
            return e;
        }
        fail(line, "unexpected symbol near " + near());
    }

    std::vector<ExprP> callargs() {
        std::vector<ExprP> args;
        if (cur.kind == Tok::String) {
            args.push_back(constant(Value::str(cur.text), cur.line));
            advance();
        } else if (sym("{")) {
            args.push_back(tablecons());
        } else {
            expect_sym("(");
            if (!sym(")")) args = exprlist();
            expect_sym(")");
        }
        return args;
    }

    ExprP suffixed() {
        ExprP e = primary();
        for (;;) {
            const int line = cur.line;
            if (sym(".")) {
                advance();
                auto idx = std::make_unique<Expr>();
                idx->kind = Expr::Index;
                idx->line = line;
                idx->a = std::move(e);
                idx->b = constant(Value::str(expect_name()), line);
                e = std::move(idx);
            } else if (sym("[")) {
                advance();
                auto idx = std::make_unique<Expr>();
                idx->kind = Expr::Index;
                idx->line = line;
                idx->a = std::move(e);
                idx->b = expr();
                expect_sym("]");
                e = std::move(idx);
            } else if (sym(":")) {
                advance();
                auto m = std::make_unique<Expr>();
                m->kind = Expr::Method;
                m->line = line;
                m->name = expect_name();
                m->a = std::move(e);
                m->args = callargs();
                e = std::move(m);
            } else if (sym("(") || sym("{") || cur.kind == Tok::String) {
                auto c = std::make_unique<Expr>();
                c->kind = Expr::Call;
                c->line = line;
                c->a = std::move(e);
                c->args = callargs();
                e = std::move(c);
            } else {
                return e;
            }
        }
    }

    ExprP tablecons() {
        auto e = std::make_unique<Expr>();
        e->kind = Expr::TableCons;
        e->line = cur.line;
        const int open_line = cur.line;
        expect_sym("{");
        while (!sym("}")) {
            if (cur.kind == Tok::End) fail(cur.line, "'}' expected (to close '{' at line " + std::to_string(open_line) + ") near <eof>");
            TableItem it;
            if (sym("[")) {
                advance();
                it.kind = TableItem::Keyed;
                it.key = expr();
                expect_sym("]");
                expect_sym("=");
                it.value = expr();
            } else if (cur.kind == Tok::Name && !keyword(cur.text) && peek().kind == Tok::Sym && peek().text == "=") {
                it.kind = TableItem::Named;
                it.name = cur.text;
                advance();
                advance();
                it.value = expr();
            } else {
                it.value = expr();
            }
            e->items.push_back(std::move(it));
            if (sym(",") || sym(";")) { advance(); continue; }
            if (!sym("}")) fail(cur.line, "'}' expected (to close '{' at line " + std::to_string(open_line) + ") near " + near());
        }
        advance();
        return e;
    }
};

// ---- runtime -------------------------------------------------------------------------------------------------------
struct Env {
    std::vector<std::pair<std::string, std::shared_ptr<Value>>> vars; // innermost last; a closure keeps the cells alive
    std::shared_ptr<Env> parent;
    std::shared_ptr<Values> varargs; // of the enclosing vararg function
    std::shared_ptr<Value> find(const std::string &name) const {
        for (const Env *e = this; e; e = e->parent.get())
            for (size_t k = e->vars.size(); k-- > 0;)
                if (e->vars[k].first == name) return e->vars[k].second;
        return nullptr;
    }
    void declare(const std::string &name, const Value &v) { vars.emplace_back(name, std::make_shared<Value>(v)); }
};

struct Interp;
struct Function {
    std::string name;
    std::shared_ptr<FuncBody> body; // a Lua function ...
    std::shared_ptr<Env> env;
    std::function<void(Interp &, Values &args, Values &rets, int line)> native; // ... or a built-in
};

struct SceneData {
    std::vector<rtc_shape> shapes;
    rtc_light light;
};
struct Job {
    std::shared_ptr<SceneData> scene; // shared with the previous job when the converted world is identical
    rtc_camera camera;
    std::string outfile;
    uint32_t kind = RTC_LUA_JOB_RENDER, animation = 0, frame = 0;
    bool same_world = false;
    int line = 0;
};

// glibc random(): TYPE_3, degree 31, separation 3 (see the header comment)
struct PosixRandom {
    int fptr = 0, rptr = 0;
    uint32_t state[31];
    PosixRandom() { seed(1u); }
    void seed(uint32_t s) {
        if (s == 0u) s = 1u;
        int32_t word = static_cast<int32_t>(s);
        state[0] = static_cast<uint32_t>(word);
        for (int i = 1; i < 31; ++i) { // x = 16807 * x mod (2^31 - 1), Schrage's form
            const long hi = word / 127773, lo = word % 127773;
            long w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            word = static_cast<int32_t>(w);
            state[i] = static_cast<uint32_t>(word);
        }
        fptr = 3;
        rptr = 0;
        for (int i = 0; i < 310; ++i) (void)next();
    }
    uint32_t next() {
        state[fptr] += state[rptr];
        const uint32_t out = state[fptr] >> 1;
        if (++fptr >= 31) { fptr = 0; ++rptr; }
        else if (++rptr >= 31) rptr = 0;
        return out;
    }
};

void world_from_table(const Table &t, SceneData &sc, int line);
void camera_from_table(const Table &t, rtc_camera &cam, int line);

enum Flow { FLOW_NORMAL, FLOW_BREAK, FLOW_RETURN };

struct Interp {
    std::shared_ptr<Table> globals = std::make_shared<Table>();
    std::vector<Job> jobs;
    std::string output;             // what the script print()ed
    std::string base_dir;           // where require() looks; empty = require is an error
    bool have_base_dir = false;
    std::map<std::string, Value> loaded;
    std::vector<std::shared_ptr<FuncBody>> chunks; // keeps required chunks' trees alive
    uint32_t animations = 0;
    std::vector<uint32_t> frames_of; // per animation
    unsigned long long steps = 0, step_limit = 100000000ull;
    size_t job_limit = 1000000, shape_bytes = 0, shape_bytes_limit = size_t(2) << 30;
    int call_depth = 0;
    PosixRandom rng;
    Values ret;                      // values of the `return` in flight
    // Closures and tables may form reference cycles (a local function in the scope it captures, _G._G): everything made
    // while the script runs is registered here and emptied when the program is freed.
    std::vector<std::weak_ptr<Table>> all_tables;
    std::vector<std::weak_ptr<Function>> all_functions;
    // Live tables and entries of the script, in units of roughly 150 bytes: a loop that builds tables for ever stops here
    // long before the step budget would stop it (100 M steps could hold 15 GB).
    std::shared_ptr<long long> mem = std::make_shared<long long>(0);
    long long mem_limit = 8000000; // about 1 GB; a world of 100 000 shapes is about 1.5 M units
    void check_mem(int line) {
        if (*mem > mem_limit) fail(line, "the script exceeded its memory budget (" + std::to_string(mem_limit) + " live tables / entries)");
    }
    std::shared_ptr<Table> new_table() {
        auto t = std::make_shared<Table>();
        t->mem = mem;
        t->charge(4);
        if (all_tables.size() >= 4096 && all_tables.size() == all_tables.capacity()) { // drop the dead ones before the registry grows
            size_t keep = 0;
            for (auto &w : all_tables)
                if (!w.expired()) all_tables[keep++] = std::move(w);
            all_tables.resize(keep);
            if (all_tables.capacity() < 2 * keep + 4096) all_tables.reserve(2 * keep + 4096);
        }
        all_tables.push_back(t);
        return t;
    }
    Value new_function() {
        Value v;
        v.kind = Value::Fun;
        v.f = std::make_shared<Function>();
        if (all_functions.size() >= 4096 && all_functions.size() == all_functions.capacity()) {
            size_t keep = 0;
            for (auto &w : all_functions)
                if (!w.expired()) all_functions[keep++] = std::move(w);
            all_functions.resize(keep);
            if (all_functions.capacity() < 2 * keep + 4096) all_functions.reserve(2 * keep + 4096);
        }
        all_functions.push_back(v.f);
        return v;
    }
    Interp() { all_tables.push_back(globals); }
    Interp(const Interp &) = delete;
    Interp &operator=(const Interp &) = delete;
    ~Interp() {
        ret.clear();
        loaded.clear();
        for (auto &w : all_functions)
            if (auto f = w.lock()) { f->env.reset(); f->native = nullptr; }
        for (auto &w : all_tables)
            if (auto t = w.lock()) { t->fields.clear(); t->index.clear(); t->array.clear(); }
    }

    void tick(int line) {
        if (++steps > step_limit) fail(line, "the script exceeded its step budget (" + std::to_string(step_limit) + " statements / iterations)");
    }

    // ---- conversions
    static std::string tostring(const Value &v) {
        char buf[64];
        switch (v.kind) {
        case Value::Nil: return "nil";
        case Value::Bool: return v.b ? "true" : "false";
        case Value::Int: std::snprintf(buf, sizeof buf, "%lld", v.i); return buf;
        case Value::Num: { // lua_Number2str "%.14g", then ".0" when it looks like an integer
            if (std::isinf(v.n)) return v.n > 0 ? "inf" : "-inf";
            if (std::isnan(v.n)) return std::signbit(v.n) ? "-nan" : "nan";
            std::snprintf(buf, sizeof buf, "%.14g", v.n);
            std::string s = buf;
            if (s.find_first_not_of("-0123456789") == std::string::npos) s += ".0";
            return s;
        }
        case Value::Str: return v.s;
        case Value::Tab: std::snprintf(buf, sizeof buf, "table: %p", static_cast<const void *>(v.t.get())); return buf;
        default: std::snprintf(buf, sizeof buf, "%s: %p", v.f && v.f->native ? "builtin" : "function", static_cast<const void *>(v.f.get())); return buf;
        }
    }
    static bool str_to_number(const std::string &s, Value &out) { // l_str2int / l_str2d: surrounding blanks allowed
        const char *p = s.c_str();
        while (std::isspace(static_cast<unsigned char>(*p))) ++p;
        bool neg = false;
        if (*p == '-') { neg = true; ++p; }
        else if (*p == '+') ++p;
        if (!(std::isdigit(static_cast<unsigned char>(*p)) || (*p == '.' && std::isdigit(static_cast<unsigned char>(p[1]))))) return false;
        Tok t;
        if (!scan_number(p, t)) return false;
        while (std::isspace(static_cast<unsigned char>(*p))) ++p;
        if (*p) return false;
        if (t.is_int) out = Value::integer(neg ? static_cast<long long>(0ull - static_cast<unsigned long long>(t.i)) : t.i);
        else out = Value::num(neg ? -t.n : t.n);
        return true;
    }
    static bool to_number(const Value &v, Value &out) { // arithmetic coerces strings
        if (v.is_number()) { out = v; return true; }
        if (v.kind == Value::Str) return str_to_number(v.s, out);
        return false;
    }
    static bool float_to_integer(double d, long long &out) { // exact integral value in range
        if (!(d >= -9223372036854775808.0 && d < 9223372036854775808.0) || d != std::floor(d)) return false;
        out = static_cast<long long>(d);
        return true;
    }
    static bool to_integer(const Value &v, long long &out) {
        if (v.kind == Value::Int) { out = v.i; return true; }
        if (v.kind == Value::Num) return float_to_integer(v.n, out);
        Value n;
        if (v.kind == Value::Str && str_to_number(v.s, n)) return to_integer(n, out);
        return false;
    }

    // ---- operators
    Value arith(Op op, const Value &av, const Value &bv, int line) {
        Value a, b;
        if (!to_number(av, a) || !to_number(bv, b))
            fail(line, std::string("attempt to perform arithmetic on a ") + type_name(to_number(av, a) ? bv : av) + " value");
        if (a.kind == Value::Int && b.kind == Value::Int) {
            const unsigned long long x = static_cast<unsigned long long>(a.i), y = static_cast<unsigned long long>(b.i);
            switch (op) { // integer arithmetic wraps (Lua 5.3 §3.4.1)
            case OP_ADD: return Value::integer(static_cast<long long>(x + y));
            case OP_SUB: return Value::integer(static_cast<long long>(x - y));
            case OP_MUL: return Value::integer(static_cast<long long>(x * y));
            case OP_IDIV: {
                if (b.i == 0) fail(line, "attempt to perform 'n//0'");
                if (b.i == -1) return Value::integer(static_cast<long long>(0ull - x));
                long long q = a.i / b.i;
                if ((a.i % b.i != 0) && ((a.i < 0) != (b.i < 0))) --q;
                return Value::integer(q);
            }
            case OP_MOD: {
                if (b.i == 0) fail(line, "attempt to perform 'n%%0'");
                if (b.i == -1) return Value::integer(0);
                long long r = a.i % b.i;
                if (r != 0 && ((r ^ b.i) < 0)) r += b.i;
                return Value::integer(r);
            }
            default: break;
            }
        }
        const double x = a.number(), y = b.number();
        switch (op) {
        case OP_ADD: return Value::num(x + y);
        case OP_SUB: return Value::num(x - y);
        case OP_MUL: return Value::num(x * y);
        case OP_DIV: return Value::num(x / y);
        case OP_POW: return Value::num(std::pow(x, y));
        case OP_IDIV: return Value::num(std::floor(x / y));
        case OP_MOD: {
            double r = std::fmod(x, y);
            if (r != 0. && ((r < 0.) != (y < 0.))) r += y;
            return Value::num(r);
        }
        default: fail(line, "internal: not an arithmetic operator");
        }
    }
    static bool raw_equal(const Value &a, const Value &b) {
        if (a.is_number() && b.is_number()) {
            if (a.kind == Value::Int && b.kind == Value::Int) return a.i == b.i;
            return a.number() == b.number();
        }
        if (a.kind != b.kind) return false;
        switch (a.kind) {
        case Value::Nil: return true;
        case Value::Bool: return a.b == b.b;
        case Value::Str: return a.s == b.s;
        case Value::Tab: return a.t == b.t;
        default: return a.f == b.f;
        }
    }
    bool less(const Value &a, const Value &b, bool or_equal, int line) {
        if (a.is_number() && b.is_number()) {
            if (a.kind == Value::Int && b.kind == Value::Int) return or_equal ? a.i <= b.i : a.i < b.i;
            return or_equal ? a.number() <= b.number() : a.number() < b.number();
        }
        if (a.kind == Value::Str && b.kind == Value::Str) return or_equal ? a.s <= b.s : a.s < b.s;
        if (a.kind == b.kind || (a.is_number() && b.is_number())) fail(line, std::string("attempt to compare two ") + type_name(a) + " values");
        fail(line, std::string("attempt to compare ") + type_name(a) + " with " + type_name(b));
    }
    std::string concat_piece(const Value &v, int line) {
        if (v.kind == Value::Str) return v.s;
        if (v.is_number()) return tostring(v);
        fail(line, std::string("attempt to concatenate a ") + type_name(v) + " value");
    }

    // ---- tables
    static bool normalise_key(const Value &k, bool &is_int, long long &ik) {
        if (k.kind == Value::Int) { is_int = true; ik = k.i; return true; }
        if (k.kind == Value::Num) {
            if (float_to_integer(k.n, ik)) { is_int = true; return true; }
            return false;
        }
        is_int = false;
        return k.kind == Value::Str;
    }
    Value index(const Value &obj, const Value &key, int line, const std::string &what) {
        if (obj.kind == Value::Str) { // methods of strings: ("x"):rep(3), s:format(...)
            const Value *lib = globals->get("string");
            if (lib && lib->kind == Value::Tab && key.kind == Value::Str) {
                const Value *m = lib->t->get(key.s);
                return m ? *m : Value{};
            }
            return Value{};
        }
        if (obj.kind != Value::Tab) fail(line, std::string("attempt to index a ") + type_name(obj) + " value" + (what.empty() ? "" : " (" + what + ")"));
        bool is_int;
        long long ik = 0;
        if (!normalise_key(key, is_int, ik)) {
            if (key.kind == Value::Num || key.kind == Value::Nil) return Value{}; // t[1.5], t[nil] read as nil
            fail(line, std::string("a ") + type_name(key) + " table key is not supported by this interpreter");
        }
        const Value *v = is_int ? obj.t->at(ik) : obj.t->get(key.s);
        return v ? *v : Value{};
    }
    void setindex(const Value &obj, const Value &key, const Value &v, int line, const std::string &what) {
        if (obj.kind != Value::Tab) fail(line, std::string("attempt to index a ") + type_name(obj) + " value" + (what.empty() ? "" : " (" + what + ")"));
        bool is_int;
        long long ik = 0;
        if (key.kind == Value::Nil) fail(line, "table index is nil");
        if (key.kind == Value::Num && std::isnan(key.n)) fail(line, "table index is NaN");
        if (!normalise_key(key, is_int, ik)) fail(line, std::string("a ") + (key.kind == Value::Num ? "fractional number" : type_name(key)) + " table key is not supported by this interpreter");
        if (is_int) obj.t->seti(ik, v);
        else obj.t->set(key.s, v);
        check_mem(line);
    }

    // ---- calls
    static std::string describe(const Expr &e) {
        if (e.kind == Expr::Name) return "global '" + e.name + "'"; // (or a local: good enough for a message)
        if (e.kind == Expr::Index && e.b && e.b->kind == Expr::Const && e.b->value.kind == Value::Str) return "field '" + e.b->value.s + "'";
        if (e.kind == Expr::Method) return "method '" + e.name + "'";
        return "";
    }
    void call(const Value &fn, Values &args, Values &rets, int line, const std::string &what) {
        rets.clear();
        if (fn.kind != Value::Fun) fail(line, std::string("attempt to call a ") + type_name(fn) + " value" + (what.empty() ? "" : " (" + what + ")"));
        tick(line);
        if (fn.f->native) { fn.f->native(*this, args, rets, line); return; }
        if (++call_depth > 200) { --call_depth; fail(line, "stack overflow (function calls nested deeper than 200)"); }
        struct Pop { int &d; ~Pop() { --d; } } pop{call_depth};
        CDepth guard(c_depth, line), guard2(c_depth, line), guard3(c_depth, line); // (a call's own frames)
        auto env = std::make_shared<Env>();
        env->parent = fn.f->env;
        const FuncBody &body = *fn.f->body;
        for (size_t k = 0; k < body.params.size(); ++k) env->declare(body.params[k], k < args.size() ? args[k] : Value{});
        if (body.vararg) {
            env->varargs = std::make_shared<Values>();
            for (size_t k = body.params.size(); k < args.size(); ++k) env->varargs->push_back(args[k]);
        }
        if (exec_block(body.body, env) == FLOW_RETURN) rets = std::move(ret);
        ret.clear();
    }

    // ---- expressions
    void eval_multi(const Expr &e, const std::shared_ptr<Env> &env, Values &out) { // appends every value of e
        switch (e.kind) {
        case Expr::Call: {
            const Value fn = eval(*e.a, env);
            Values args, rets;
            eval_list(e.args, env, args);
            call(fn, args, rets, e.line, describe(*e.a));
            for (auto &v : rets) out.push_back(std::move(v));
            return;
        }
        case Expr::Method: {
            const Value obj = eval(*e.a, env);
            const Value fn = index(obj, Value::str(e.name), e.line, describe(*e.a));
            Values args, rets;
            args.push_back(obj);
            eval_list(e.args, env, args);
            call(fn, args, rets, e.line, "method '" + e.name + "'");
            for (auto &v : rets) out.push_back(std::move(v));
            return;
        }
        case Expr::Vararg: {
            const Env *s = env.get();
            while (s && !s->varargs) s = s->parent.get();
            if (!s) fail(e.line, "cannot use '...' outside a vararg function");
            for (const auto &v : *s->varargs) out.push_back(v);
            return;
        }
        default: out.push_back(eval(e, env));
        }
    }
    void eval_list(const std::vector<ExprP> &list, const std::shared_ptr<Env> &env, Values &out) { // the last one expands
        for (size_t k = 0; k < list.size(); ++k) {
            if (k + 1 == list.size()) eval_multi(*list[k], env, out);
            else out.push_back(eval(*list[k], env));
        }
    }
    // C stack: a nested call costs about 4 KB of it, a nested expression level about 0.3 KB; both count against one bound
    // (200 calls of ordinary expressions, or fewer calls of deeply parenthesised ones: well under 2 MB either way).
    int c_depth = 0;
    struct CDepth {
        int &d;
        CDepth(int &depth, int line) : d(depth) { if (++d > 2400) { --d; fail(line, "stack overflow (expressions and calls nested too deeply)"); } }
        ~CDepth() { --d; }
    };
    Value eval(const Expr &e, const std::shared_ptr<Env> &env) {
        if (e.kind == Expr::Const) return e.value;
        CDepth guard(c_depth, e.line);
        switch (e.kind) {
        case Expr::Const: return e.value;
        case Expr::Paren: return eval(*e.a, env);
        case Expr::Name: {
            if (auto cell = env->find(e.name)) return *cell;
            const Value *g = globals->get(e.name);
            return g ? *g : Value{};
        }
        case Expr::Index: return index(eval(*e.a, env), eval(*e.b, env), e.line, describe(*e.a));
        case Expr::Call: case Expr::Method: case Expr::Vararg: {
            Values v;
            eval_multi(e, env, v);
            return v.empty() ? Value{} : v[0];
        }
        case Expr::Func: {
            Value v = new_function();
            v.f->name = e.func->name;
            v.f->body = e.func;
            v.f->env = env;
            return v;
        }
        case Expr::TableCons: {
            auto t = new_table();
            check_mem(e.line);
            long long next_index = 1;
            for (size_t k = 0; k < e.items.size(); ++k) {
                const TableItem &it = e.items[k];
                if (it.kind == TableItem::Named) {
                    t->set(it.name, eval(*it.value, env));
                } else if (it.kind == TableItem::Keyed) {
                    const Value key = eval(*it.key, env);
                    setindex(Value::table(t), key, eval(*it.value, env), e.line, "");
                } else if (k + 1 == e.items.size()) { // the last positional item expands
                    Values vs;
                    eval_multi(*it.value, env, vs);
                    for (auto &v : vs) t->seti(next_index++, v);
                } else {
                    t->seti(next_index++, eval(*it.value, env));
                }
            }
            return Value::table(t);
        }
        case Expr::Unary: {
            const Value v = eval(*e.a, env);
            if (e.op == OP_NOT) return Value::boolean(!v.truthy());
            if (e.op == OP_LEN) {
                if (v.kind == Value::Str) return Value::integer(static_cast<long long>(v.s.size()));
                if (v.kind == Value::Tab) return Value::integer(v.t->length());
                fail(e.line, std::string("attempt to get length of a ") + type_name(v) + " value");
            }
            Value n;
            if (!to_number(v, n)) fail(e.line, std::string("attempt to perform arithmetic on a ") + type_name(v) + " value");
            return n.kind == Value::Int ? Value::integer(static_cast<long long>(0ull - static_cast<unsigned long long>(n.i))) : Value::num(-n.n);
        }
        case Expr::Binary: {
            if (e.op == OP_AND) { Value a = eval(*e.a, env); return a.truthy() ? eval(*e.b, env) : a; }
            if (e.op == OP_OR) { Value a = eval(*e.a, env); return a.truthy() ? a : eval(*e.b, env); }
            const Value a = eval(*e.a, env), b = eval(*e.b, env);
            switch (e.op) {
            case OP_EQ: return Value::boolean(raw_equal(a, b));
            case OP_NE: return Value::boolean(!raw_equal(a, b));
            case OP_LT: return Value::boolean(less(a, b, false, e.line));
            case OP_LE: return Value::boolean(less(a, b, true, e.line));
            case OP_GT: return Value::boolean(less(b, a, false, e.line));
            case OP_GE: return Value::boolean(less(b, a, true, e.line));
            case OP_CONCAT: {
                std::string r = concat_piece(a, e.line);
                r += concat_piece(b, e.line);
                if (r.size() > (size_t(1) << 26)) fail(e.line, "resulting string too large");
                return Value::str(std::move(r));
            }
            default: return arith(e.op, a, b, e.line);
            }
        }
        default: fail(e.line, "internal: unknown expression");
        }
    }

    // ---- statements
    void assign(const Expr &target, const Value &v, const std::shared_ptr<Env> &env) {
        if (target.kind == Expr::Name) {
            if (auto cell = env->find(target.name)) *cell = v;
            else globals->set(target.name, v);
            return;
        }
        setindex(eval(*target.a, env), eval(*target.b, env), v, target.line, describe(*target.a));
    }

    Flow exec_block(const Block &b, const std::shared_ptr<Env> &env) {
        for (const auto &s : b) {
            const Flow f = exec(*s, env);
            if (f != FLOW_NORMAL) return f;
        }
        return FLOW_NORMAL;
    }
    std::shared_ptr<Env> scope(const std::shared_ptr<Env> &env) {
        auto e = std::make_shared<Env>();
        e->parent = env;
        return e;
    }

    Flow exec(const Stmt &s, const std::shared_ptr<Env> &env) {
        tick(s.line);
        switch (s.kind) {
        case Stmt::Local: {
            Values vs;
            eval_list(s.exprs, env, vs);
            for (size_t k = 0; k < s.names.size(); ++k) env->declare(s.names[k], k < vs.size() ? vs[k] : Value{});
            return FLOW_NORMAL;
        }
        case Stmt::LocalFunc: {
            env->declare(s.names[0], Value{}); // visible inside its own body (recursion)
            Value v = new_function();
            v.f->name = s.func->name;
            v.f->body = s.func;
            v.f->env = env;
            *env->vars.back().second = v;
            return FLOW_NORMAL;
        }
        case Stmt::Assign: {
            if (s.targets.size() == 1 && s.exprs.size() == 1) {
                assign(*s.targets[0], eval(*s.exprs[0], env), env);
                return FLOW_NORMAL;
            }
            // "In a multiple assignment, Lua first evaluates all values and only then executes the assignments" (manual 3.3.3):
            // in `i, a[i] = i + 1, 20` the i of a[i] is the old one — table and key of every indexed target are taken first.
            std::vector<std::pair<Value, Value>> where(s.targets.size());
            for (size_t k = 0; k < s.targets.size(); ++k)
                if (s.targets[k]->kind == Expr::Index) where[k] = {eval(*s.targets[k]->a, env), eval(*s.targets[k]->b, env)};
            Values vs;
            eval_list(s.exprs, env, vs);
            for (size_t k = 0; k < s.targets.size(); ++k) {
                const Value v = k < vs.size() ? vs[k] : Value{};
                if (s.targets[k]->kind == Expr::Index) setindex(where[k].first, where[k].second, v, s.targets[k]->line, describe(*s.targets[k]->a));
                else assign(*s.targets[k], v, env);
            }
            return FLOW_NORMAL;
        }
        case Stmt::CallStmt: {
            Values discard;
            eval_multi(*s.exprs[0], env, discard);
            return FLOW_NORMAL;
        }
        case Stmt::Do: return exec_block(s.blocks[0], scope(env));
        case Stmt::While:
            while (eval(*s.exprs[0], env).truthy()) {
                tick(s.line);
                const Flow f = exec_block(s.blocks[0], scope(env));
                if (f == FLOW_BREAK) break;
                if (f == FLOW_RETURN) return f;
            }
            return FLOW_NORMAL;
        case Stmt::Repeat:
            for (;;) {
                tick(s.line);
                auto inner = scope(env); // the condition sees the body's locals
                const Flow f = exec_block(s.blocks[0], inner);
                if (f == FLOW_BREAK) break;
                if (f == FLOW_RETURN) return f;
                if (eval(*s.exprs[0], inner).truthy()) break;
            }
            return FLOW_NORMAL;
        case Stmt::If:
            for (size_t k = 0; k < s.exprs.size(); ++k)
                if (eval(*s.exprs[k], env).truthy()) return exec_block(s.blocks[k], scope(env));
            if (s.has_else) return exec_block(s.blocks.back(), scope(env));
            return FLOW_NORMAL;
        case Stmt::NumFor: {
            Value v0, v1, v2 = Value::integer(1);
            if (!to_number(eval(*s.exprs[0], env), v0)) fail(s.line, "'for' initial value must be a number");
            if (!to_number(eval(*s.exprs[1], env), v1)) fail(s.line, "'for' limit must be a number");
            if (s.exprs.size() > 2 && !to_number(eval(*s.exprs[2], env), v2)) fail(s.line, "'for' step must be a number");
            if (v0.kind == Value::Int && v2.kind == Value::Int) { // integer loop; a float limit is clipped (forlimit)
                long long limit;
                if (v1.kind == Value::Int) limit = v1.i;
                else if (std::isnan(v1.n)) return FLOW_NORMAL;
                else if (v1.n >= 9223372036854775808.0) limit = INT64_MAX;
                else if (v1.n < -9223372036854775808.0) limit = INT64_MIN;
                else limit = static_cast<long long>(v2.i > 0 ? std::floor(v1.n) : std::ceil(v1.n));
                const long long step = v2.i;
                if (step == 0) fail(s.line, "'for' step is zero");
                for (long long i = v0.i; step > 0 ? i <= limit : i >= limit;) {
                    tick(s.line);
                    auto inner = scope(env);
                    inner->declare(s.names[0], Value::integer(i));
                    const Flow f = exec_block(s.blocks[0], inner);
                    if (f == FLOW_BREAK) break;
                    if (f == FLOW_RETURN) return f;
                    const unsigned long long left = step > 0 ? static_cast<unsigned long long>(limit) - static_cast<unsigned long long>(i)
                                                             : static_cast<unsigned long long>(i) - static_cast<unsigned long long>(limit);
                    const unsigned long long ustep = step > 0 ? static_cast<unsigned long long>(step) : 0ull - static_cast<unsigned long long>(step);
                    if (left < ustep) break; // the next value would pass the limit (or wrap)
                    i = static_cast<long long>(static_cast<unsigned long long>(i) + static_cast<unsigned long long>(step));
                }
                return FLOW_NORMAL;
            }
            const double start = v0.number(), limit = v1.number(), step = v2.number();
            if (step == 0.) fail(s.line, "'for' step is zero");
            for (double x = start; step > 0. ? x <= limit : x >= limit; x += step) {
                tick(s.line);
                auto inner = scope(env);
                inner->declare(s.names[0], Value::num(x));
                const Flow f = exec_block(s.blocks[0], inner);
                if (f == FLOW_BREAK) break;
                if (f == FLOW_RETURN) return f;
            }
            return FLOW_NORMAL;
        }
        case Stmt::GenFor: {
            Values init;
            eval_list(s.exprs, env, init);
            init.resize(3);
            const Value iter = init[0], state = init[1];
            Value control = init[2];
            for (;;) {
                tick(s.line);
                Values args{state, control}, rets;
                call(iter, args, rets, s.line, "for iterator");
                if (rets.empty() || rets[0].kind == Value::Nil) break;
                control = rets[0];
                auto inner = scope(env);
                for (size_t k = 0; k < s.names.size(); ++k) inner->declare(s.names[k], k < rets.size() ? rets[k] : Value{});
                const Flow f = exec_block(s.blocks[0], inner);
                if (f == FLOW_BREAK) break;
                if (f == FLOW_RETURN) return f;
            }
            return FLOW_NORMAL;
        }
        case Stmt::Return: {
            Values vs;
            eval_list(s.exprs, env, vs);
            ret = std::move(vs);
            return FLOW_RETURN;
        }
        case Stmt::Break: return FLOW_BREAK;
        }
        return FLOW_NORMAL;
    }

    // ---- chunks
    Values run_chunk(const char *text, const std::string &chunk_name) {
        std::shared_ptr<FuncBody> body = std::make_shared<FuncBody>();
        body->name = chunk_name;
        body->vararg = true;
        try {
            Parser p(text);
            body->body = p.block();
            if (p.cur.kind != Tok::End) fail(p.cur.line, "'<eof>' expected near " + p.near());
        } catch (const LuaError &e) {
            throw LuaError(chunk_name + ": " + e.what());
        }
        chunks.push_back(body);
        auto env = std::make_shared<Env>();
        env->varargs = std::make_shared<Values>();
        Values out;
        try {
            if (exec_block(body->body, env) == FLOW_RETURN) out = std::move(ret);
            ret.clear();
        } catch (const LuaError &e) {
            throw LuaError(chunk_name + ": " + e.what());
        }
        return out;
    }

    // ---- the library
    void def(const std::shared_ptr<Table> &t, const char *name, std::function<void(Interp &, Values &, Values &, int)> fn) {
        Value v = new_function();
        v.f->name = name;
        v.f->native = std::move(fn);
        t->set(name, v);
    }
    static const Value &arg(const Values &a, size_t k) {
        static const Value nil;
        return k < a.size() ? a[k] : nil;
    }
    static double check_number(const Values &a, size_t k, const char *fn, int line) {
        Value n;
        if (!to_number(arg(a, k), n)) fail(line, "bad argument #" + std::to_string(k + 1) + " to '" + fn + "' (number expected, got " + (k < a.size() ? type_name(a[k]) : "no value") + ")");
        return n.number();
    }
    static long long check_integer(const Values &a, size_t k, const char *fn, int line) {
        long long v;
        if (!to_integer(arg(a, k), v)) {
            Value n;
            if (to_number(arg(a, k), n)) fail(line, "bad argument #" + std::to_string(k + 1) + " to '" + fn + "' (number has no integer representation)");
            fail(line, "bad argument #" + std::to_string(k + 1) + " to '" + fn + "' (number expected, got " + (k < a.size() ? type_name(a[k]) : "no value") + ")");
        }
        return v;
    }
    static const std::shared_ptr<Table> &check_table(const Values &a, size_t k, const char *fn, int line) {
        if (arg(a, k).kind != Value::Tab) fail(line, "bad argument #" + std::to_string(k + 1) + " to '" + fn + "' (table expected, got " + (k < a.size() ? type_name(a[k]) : "no value") + ")");
        return a[k].t;
    }
    static std::string check_string(const Values &a, size_t k, const char *fn, int line) {
        const Value &v = arg(a, k);
        if (v.kind == Value::Str) return v.s;
        if (v.is_number()) return tostring(v);
        fail(line, "bad argument #" + std::to_string(k + 1) + " to '" + fn + "' (string expected, got " + (k < a.size() ? type_name(a[k]) : "no value") + ")");
    }

    std::string format(const Values &a, int line) { // string.format: C conversions, %s through tostring
        const std::string fmt = check_string(a, 0, "format", line);
        std::string out;
        size_t argi = 1;
        for (size_t k = 0; k < fmt.size(); ++k) {
            if (fmt[k] != '%') { out.push_back(fmt[k]); continue; }
            if (++k >= fmt.size()) fail(line, "invalid conversion '%' to 'format'");
            if (fmt[k] == '%') { out.push_back('%'); continue; }
            std::string spec = "%";
            while (k < fmt.size() && std::strchr("-+ #0", fmt[k])) spec.push_back(fmt[k++]);
            size_t digits = 0;
            while (k < fmt.size() && std::isdigit(static_cast<unsigned char>(fmt[k])) && digits++ < 2) spec.push_back(fmt[k++]);
            if (k < fmt.size() && fmt[k] == '.') {
                spec.push_back(fmt[k++]);
                digits = 0;
                while (k < fmt.size() && std::isdigit(static_cast<unsigned char>(fmt[k])) && digits++ < 2) spec.push_back(fmt[k++]);
            }
            if (k >= fmt.size() || spec.size() > 12) fail(line, "invalid conversion '" + spec + "' to 'format'");
            const char c = fmt[k];
            char buf[512];
            if (argi >= a.size()) fail(line, "bad argument #" + std::to_string(argi + 1) + " to 'format' (no value)");
            switch (c) {
            case 'd': case 'i': {
                spec += "lld";
                std::snprintf(buf, sizeof buf, spec.c_str(), check_integer(a, argi, "format", line));
                out += buf;
                break;
            }
            case 'u': case 'o': case 'x': case 'X': {
                spec += std::string("ll") + c;
                std::snprintf(buf, sizeof buf, spec.c_str(), static_cast<unsigned long long>(check_integer(a, argi, "format", line)));
                out += buf;
                break;
            }
            case 'c': out.push_back(static_cast<char>(check_integer(a, argi, "format", line))); break;
            case 'e': case 'E': case 'f': case 'F': case 'g': case 'G': case 'a': case 'A': {
                spec.push_back(c);
                std::snprintf(buf, sizeof buf, spec.c_str(), check_number(a, argi, "format", line));
                out += buf;
                break;
            }
            case 's': {
                const std::string s = tostring(a[argi]);
                if (spec == "%") { out += s; break; }
                spec.push_back('s');
                const int need = std::snprintf(nullptr, 0, spec.c_str(), s.c_str());
                std::string tmp(static_cast<size_t>(need) + 1, '\0');
                std::snprintf(&tmp[0], tmp.size(), spec.c_str(), s.c_str());
                tmp.resize(static_cast<size_t>(need));
                out += tmp;
                break;
            }
            case 'q': {
                out.push_back('"');
                for (char ch : check_string(a, argi, "format", line)) {
                    if (ch == '"' || ch == '\\') { out.push_back('\\'); out.push_back(ch); }
                    else if (ch == '\n') out += "\\n";
                    else out.push_back(ch);
                }
                out.push_back('"');
                break;
            }
            default: fail(line, std::string("invalid conversion '") + spec + c + "' to 'format'");
            }
            ++argi;
        }
        return out;
    }

    bool next_entry(const Table &t, const Value &key, Value &k_out, Value &v_out, int line) { // integer keys ascending, then strings
        size_t field_from = 0;
        if (key.kind == Value::Nil) {
            if (!t.array.empty()) { k_out = Value::integer(t.array.begin()->first); v_out = t.array.begin()->second; return true; }
        } else {
            bool is_int;
            long long ik = 0;
            if (!normalise_key(key, is_int, ik)) fail(line, "invalid key to 'next'");
            if (is_int) {
                auto it = t.array.upper_bound(ik);
                if (it != t.array.end()) { k_out = Value::integer(it->first); v_out = it->second; return true; }
            } else {
                auto it = t.index.find(key.s);
                if (it == t.index.end()) fail(line, "invalid key to 'next'");
                field_from = it->second + 1;
            }
        }
        for (size_t k = field_from; k < t.fields.size(); ++k)
            if (t.fields[k].second.kind != Value::Nil) { k_out = Value::str(t.fields[k].first); v_out = t.fields[k].second; return true; }
        return false;
    }

    void record_job(uint32_t kind, const Value &world, const Value &camera, const std::string &outfile, uint32_t animation, int line) {
        if (world.kind != Value::Tab || camera.kind != Value::Tab)
            fail(line, std::string(kind == RTC_LUA_JOB_RENDER ? "Render" : "AddFrame") + " expects (world table, camera table" + (kind == RTC_LUA_JOB_RENDER ? ", output file name)" : ")"));
        if (jobs.size() >= job_limit) fail(line, "the script renders more than " + std::to_string(job_limit) + " frames");
        Job j;
        j.kind = kind;
        j.outfile = outfile;
        j.animation = animation;
        j.line = line;
        auto sc = std::make_shared<SceneData>();
        world_from_table(*world.t, *sc, line);   // the reference converts the world first, then the camera (lua.rs:36-37,60-63)
        camera_from_table(*camera.t, j.camera, line);
        if (!jobs.empty()) { // an animation usually renders one world from many cameras: keep one copy
            const SceneData &prev = *jobs.back().scene;
            if (prev.shapes.size() == sc->shapes.size() && std::memcmp(&prev.light, &sc->light, sizeof(rtc_light)) == 0 &&
                (sc->shapes.empty() || std::memcmp(prev.shapes.data(), sc->shapes.data(), sizeof(rtc_shape) * sc->shapes.size()) == 0)) {
                sc = jobs.back().scene;
                j.same_world = true;
            }
        }
        if (!j.same_world) {
            shape_bytes += sizeof(rtc_shape) * sc->shapes.size();
            if (shape_bytes > shape_bytes_limit) fail(line, "the script's worlds exceed " + std::to_string(shape_bytes_limit >> 20) + " MiB of shape records");
        }
        j.scene = sc;
        if (kind == RTC_LUA_JOB_ADD_FRAME) j.frame = frames_of[animation]++;
        jobs.push_back(std::move(j));
    }

    void open_libraries() {
        const std::shared_ptr<Table> &G = globals;
        G->set("_G", Value::table(G));
        G->set("_VERSION", Value::str("Lua 5.3"));
        def(G, "print", [](Interp &in, Values &a, Values &, int) {
            for (size_t k = 0; k < a.size(); ++k) { if (k) in.output.push_back('\t'); in.output += tostring(a[k]); }
            in.output.push_back('\n');
            if (in.output.size() > (size_t(16) << 20)) in.output.erase(0, in.output.size() - (size_t(8) << 20)); // keep the tail
        });
        def(G, "type", [](Interp &, Values &a, Values &r, int line) {
            if (a.empty()) fail(line, "bad argument #1 to 'type' (value expected)");
            r.push_back(Value::str(type_name(a[0])));
        });
        def(G, "tostring", [](Interp &, Values &a, Values &r, int) { r.push_back(Value::str(tostring(arg(a, 0)))); });
        def(G, "tonumber", [](Interp &, Values &a, Values &r, int line) {
            if (a.size() > 1 && a[1].kind != Value::Nil) fail(line, "tonumber with a base is not supported by this interpreter");
            Value n;
            if (arg(a, 0).is_number()) r.push_back(a[0]);
            else if (arg(a, 0).kind == Value::Str && str_to_number(a[0].s, n)) r.push_back(n);
            else r.push_back(Value{});
        });
        def(G, "next", [](Interp &in, Values &a, Values &r, int line) {
            Value k, v;
            if (in.next_entry(*check_table(a, 0, "next", line), arg(a, 1), k, v, line)) { r.push_back(k); r.push_back(v); }
            else r.push_back(Value{});
        });
        def(G, "pairs", [](Interp &in, Values &a, Values &r, int line) {
            check_table(a, 0, "pairs", line);
            r.push_back(*in.globals->get("next"));
            r.push_back(a[0]);
            r.push_back(Value{});
        });
        {
            Value iter = new_function();
            iter.f->name = "ipairs iterator";
            iter.f->native = [](Interp &, Values &a, Values &r, int line) {
                const long long k = check_integer(a, 1, "ipairs", line) + 1;
                const Value *v = check_table(a, 0, "ipairs", line)->at(k);
                if (!v) { r.push_back(Value{}); return; }
                r.push_back(Value::integer(k));
                r.push_back(*v);
            };
            def(G, "ipairs", [iter](Interp &, Values &a, Values &r, int line) {
                check_table(a, 0, "ipairs", line);
                r.push_back(iter);
                r.push_back(a[0]);
                r.push_back(Value::integer(0));
            });
        }
        def(G, "select", [](Interp &, Values &a, Values &r, int line) {
            if (arg(a, 0).kind == Value::Str && a[0].s == "#") { r.push_back(Value::integer(static_cast<long long>(a.size()) - 1)); return; }
            long long n = check_integer(a, 0, "select", line);
            const long long count = static_cast<long long>(a.size()) - 1;
            if (n < 0) n = count + n + 1;
            if (n < 1) fail(line, "bad argument #1 to 'select' (index out of range)");
            for (long long k = n; k <= count; ++k) r.push_back(a[static_cast<size_t>(k)]);
        });
        def(G, "assert", [](Interp &, Values &a, Values &r, int line) {
            if (a.empty()) fail(line, "bad argument #1 to 'assert' (value expected)");
            if (!a[0].truthy()) fail(line, a.size() > 1 ? tostring(a[1]) : "assertion failed!");
            r = a;
        });
        def(G, "error", [](Interp &, Values &a, Values &, int line) { fail(line, a.empty() ? "nil" : tostring(a[0])); });
        def(G, "pcall", [](Interp &in, Values &a, Values &r, int line) {
            if (a.empty()) fail(line, "bad argument #1 to 'pcall' (value expected)");
            const Value fn = a[0];
            Values args(a.begin() + 1, a.end()), rets;
            const int depth = in.call_depth;
            try {
                in.call(fn, args, rets, line, "");
                r.push_back(Value::boolean(true));
                for (auto &v : rets) r.push_back(std::move(v));
            } catch (const LuaError &e) {
                if (in.steps > in.step_limit) throw; // the budget is not catchable
                in.call_depth = depth; // (c_depth unwinds with its guards)
                in.ret.clear();
                r.clear();
                r.push_back(Value::boolean(false));
                r.push_back(Value::str(e.what()));
            }
        });
        def(G, "rawequal", [](Interp &, Values &a, Values &r, int) { r.push_back(Value::boolean(raw_equal(arg(a, 0), arg(a, 1)))); });
        def(G, "rawlen", [](Interp &, Values &a, Values &r, int line) {
            if (arg(a, 0).kind == Value::Str) r.push_back(Value::integer(static_cast<long long>(a[0].s.size())));
            else r.push_back(Value::integer(check_table(a, 0, "rawlen", line)->length()));
        });
        def(G, "rawget", [](Interp &in, Values &a, Values &r, int line) { check_table(a, 0, "rawget", line); r.push_back(in.index(a[0], arg(a, 1), line, "")); });
        def(G, "rawset", [](Interp &in, Values &a, Values &r, int line) { check_table(a, 0, "rawset", line); in.setindex(a[0], arg(a, 1), arg(a, 2), line, ""); r.push_back(a[0]); });
        def(G, "setmetatable", [](Interp &, Values &, Values &, int line) { fail(line, "metatables are not supported by this interpreter"); });
        def(G, "getmetatable", [](Interp &, Values &, Values &r, int) { r.push_back(Value{}); });
        def(G, "require", [](Interp &in, Values &a, Values &r, int line) {
            const std::string name = check_string(a, 0, "require", line);
            auto it = in.loaded.find(name);
            if (it != in.loaded.end()) { r.push_back(it->second); return; }
            if (!in.have_base_dir) fail(line, "module '" + name + "' not found: require needs the script's directory (load the script from a file, or pass base_dir)");
            std::string rel = name;
            for (char &ch : rel) {
                if (ch == '.') ch = '/';
                else if (!(std::isalnum(static_cast<unsigned char>(ch)) || ch == '_' || ch == '-' || ch == '/')) fail(line, "module '" + name + "' not found: unsupported module name");
            }
            if (rel.find("//") != std::string::npos || rel.empty() || rel[0] == '/') fail(line, "module '" + name + "' not found: unsupported module name");
            const std::string path = (in.base_dir.empty() ? std::string(".") : in.base_dir) + "/" + rel + ".lua";
            std::FILE *f = std::fopen(path.c_str(), "rb");
            if (!f) fail(line, "module '" + name + "' not found:\n\tno file '" + path + "'");
            std::string text;
            char buf[4096];
            size_t n;
            while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) {
                text.append(buf, n);
                if (text.size() > (size_t(16) << 20)) { std::fclose(f); fail(line, "module '" + name + "' is larger than 16 MiB"); }
            }
            std::fclose(f);
            if (text.find('\0') != std::string::npos) fail(line, "module '" + name + "' is not a text file");
            in.loaded[name] = Value::boolean(true); // a module that requires itself does not recurse
            if (++in.call_depth > 200) { --in.call_depth; fail(line, "stack overflow (require nested too deeply)"); }
            Values out;
            try { out = in.run_chunk(text.c_str(), rel + ".lua"); } catch (...) { --in.call_depth; throw; }
            --in.call_depth;
            if (!out.empty() && out[0].kind != Value::Nil) in.loaded[name] = out[0];
            r.push_back(in.loaded[name]);
        });

        // math
        auto M = new_table();
        G->set("math", Value::table(M));
        M->set("pi", Value::num(3.141592653589793));
        M->set("huge", Value::num(HUGE_VAL));
        M->set("maxinteger", Value::integer(INT64_MAX));
        M->set("mininteger", Value::integer(INT64_MIN));
#define RTC_LUA_MATH1(NAME, EXPR) def(M, NAME, [](Interp &, Values &a, Values &r, int line) { const double x = check_number(a, 0, NAME, line); r.push_back(Value::num(EXPR)); })
        RTC_LUA_MATH1("sin", std::sin(x));
        RTC_LUA_MATH1("cos", std::cos(x));
        RTC_LUA_MATH1("tan", std::tan(x));
        RTC_LUA_MATH1("asin", std::asin(x));
        RTC_LUA_MATH1("acos", std::acos(x));
        RTC_LUA_MATH1("sqrt", std::sqrt(x));
        RTC_LUA_MATH1("exp", std::exp(x));
        RTC_LUA_MATH1("rad", x * (3.141592653589793 / 180.0));
        RTC_LUA_MATH1("deg", x * (180.0 / 3.141592653589793));
#undef RTC_LUA_MATH1
        def(M, "atan", [](Interp &, Values &a, Values &r, int line) {
            r.push_back(Value::num(std::atan2(check_number(a, 0, "atan", line), a.size() > 1 ? check_number(a, 1, "atan", line) : 1.0)));
        });
        def(M, "log", [](Interp &, Values &a, Values &r, int line) {
            const double x = check_number(a, 0, "log", line);
            if (a.size() < 2 || a[1].kind == Value::Nil) { r.push_back(Value::num(std::log(x))); return; }
            const double b = check_number(a, 1, "log", line);
            r.push_back(Value::num(b == 2.0 ? std::log2(x) : b == 10.0 ? std::log10(x) : std::log(x) / std::log(b)));
        });
        def(M, "pow", [](Interp &, Values &a, Values &r, int line) { r.push_back(Value::num(std::pow(check_number(a, 0, "pow", line), check_number(a, 1, "pow", line)))); });
        def(M, "fmod", [](Interp &, Values &a, Values &r, int line) { r.push_back(Value::num(std::fmod(check_number(a, 0, "fmod", line), check_number(a, 1, "fmod", line)))); });
        def(M, "abs", [](Interp &, Values &a, Values &r, int line) {
            if (arg(a, 0).kind == Value::Int) { r.push_back(Value::integer(a[0].i < 0 ? static_cast<long long>(0ull - static_cast<unsigned long long>(a[0].i)) : a[0].i)); return; }
            r.push_back(Value::num(std::fabs(check_number(a, 0, "abs", line))));
        });
        def(M, "floor", [](Interp &, Values &a, Values &r, int line) {
            if (arg(a, 0).kind == Value::Int) { r.push_back(a[0]); return; }
            const double f = std::floor(check_number(a, 0, "floor", line));
            long long i;
            r.push_back(float_to_integer(f, i) ? Value::integer(i) : Value::num(f));
        });
        def(M, "ceil", [](Interp &, Values &a, Values &r, int line) {
            if (arg(a, 0).kind == Value::Int) { r.push_back(a[0]); return; }
            const double f = std::ceil(check_number(a, 0, "ceil", line));
            long long i;
            r.push_back(float_to_integer(f, i) ? Value::integer(i) : Value::num(f));
        });
        def(M, "tointeger", [](Interp &, Values &a, Values &r, int) {
            long long i;
            if (arg(a, 0).is_number() && to_integer(a[0], i)) r.push_back(Value::integer(i));
            else r.push_back(Value{});
        });
        def(M, "type", [](Interp &, Values &a, Values &r, int line) {
            if (a.empty()) fail(line, "bad argument #1 to 'type' (value expected)");
            if (a[0].kind == Value::Int) r.push_back(Value::str("integer"));
            else if (a[0].kind == Value::Num) r.push_back(Value::str("float"));
            else r.push_back(Value{});
        });
        def(M, "max", [](Interp &in, Values &a, Values &r, int line) {
            check_number(a, 0, "max", line);
            size_t best = 0;
            for (size_t k = 1; k < a.size(); ++k) { check_number(a, k, "max", line); if (in.less(a[best], a[k], false, line)) best = k; }
            Value n; to_number(a[best], n); r.push_back(n);
        });
        def(M, "min", [](Interp &in, Values &a, Values &r, int line) {
            check_number(a, 0, "min", line);
            size_t best = 0;
            for (size_t k = 1; k < a.size(); ++k) { check_number(a, k, "min", line); if (in.less(a[k], a[best], false, line)) best = k; }
            Value n; to_number(a[best], n); r.push_back(n);
        });
        def(M, "randomseed", [](Interp &in, Values &a, Values &, int line) { // lmathlib.c math_randomseed
            const double n = check_number(a, 0, "randomseed", line);
            long long i;
            if (arg(a, 0).kind == Value::Int) i = a[0].i;
            else if (!float_to_integer(std::floor(n), i)) i = 0; // (lua_Integer)n of an out-of-range float: unspecified in C
            in.rng.seed(static_cast<uint32_t>(static_cast<unsigned long long>(i)));
            (void)in.rng.next(); // "discards first value to avoid undesirable correlations"
        });
        def(M, "random", [](Interp &in, Values &a, Values &r, int line) { // lmathlib.c math_random
            const double u = static_cast<double>(in.rng.next()) * (1.0 / (2147483647.0 + 1.0));
            long long low, up;
            if (a.empty()) { r.push_back(Value::num(u)); return; }
            if (a.size() == 1) { low = 1; up = check_integer(a, 0, "random", line); }
            else if (a.size() == 2) { low = check_integer(a, 0, "random", line); up = check_integer(a, 1, "random", line); }
            else fail(line, "wrong number of arguments to 'random'");
            if (low > up) fail(line, "bad argument #" + std::to_string(a.size()) + " to 'random' (interval is empty)");
            if (!(low >= 0 || up <= INT64_MAX + low)) fail(line, "bad argument #" + std::to_string(a.size()) + " to 'random' (interval too large)");
            const double scaled = u * (static_cast<double>(up - low) + 1.0);
            r.push_back(Value::integer(static_cast<long long>(scaled) + low));
        });

        // string
        auto S = new_table();
        G->set("string", Value::table(S));
        def(S, "format", [](Interp &in, Values &a, Values &r, int line) { r.push_back(Value::str(in.format(a, line))); });
        def(S, "len", [](Interp &, Values &a, Values &r, int line) { r.push_back(Value::integer(static_cast<long long>(check_string(a, 0, "len", line).size()))); });
        def(S, "sub", [](Interp &, Values &a, Values &r, int line) {
            const std::string s = check_string(a, 0, "sub", line);
            const long long len = static_cast<long long>(s.size());
            long long i = a.size() > 1 ? check_integer(a, 1, "sub", line) : 1, j = (a.size() > 2 && a[2].kind != Value::Nil) ? check_integer(a, 2, "sub", line) : -1;
            if (i < 0) i = std::max<long long>(len + i + 1, 1); else if (i == 0) i = 1;
            if (j < 0) j = len + j + 1; else if (j > len) j = len;
            r.push_back(Value::str(i > j ? std::string() : s.substr(static_cast<size_t>(i - 1), static_cast<size_t>(j - i + 1))));
        });
        def(S, "rep", [](Interp &, Values &a, Values &r, int line) {
            const std::string s = check_string(a, 0, "rep", line), sep = a.size() > 2 ? check_string(a, 2, "rep", line) : std::string();
            const long long n = check_integer(a, 1, "rep", line);
            if (n > 0 && (s.size() + sep.size()) * static_cast<unsigned long long>(n) > (1ull << 26)) fail(line, "resulting string too large");
            std::string out;
            for (long long k = 0; k < n; ++k) { if (k) out += sep; out += s; }
            r.push_back(Value::str(out));
        });
        def(S, "upper", [](Interp &, Values &a, Values &r, int line) { std::string s = check_string(a, 0, "upper", line); for (char &c : s) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c))); r.push_back(Value::str(s)); });
        def(S, "lower", [](Interp &, Values &a, Values &r, int line) { std::string s = check_string(a, 0, "lower", line); for (char &c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c))); r.push_back(Value::str(s)); });
        def(S, "byte", [](Interp &, Values &a, Values &r, int line) {
            const std::string s = check_string(a, 0, "byte", line);
            long long i = a.size() > 1 ? check_integer(a, 1, "byte", line) : 1;
            if (i < 0) i = static_cast<long long>(s.size()) + i + 1;
            if (i >= 1 && i <= static_cast<long long>(s.size())) r.push_back(Value::integer(static_cast<unsigned char>(s[static_cast<size_t>(i - 1)])));
        });
        def(S, "char", [](Interp &, Values &a, Values &r, int line) {
            std::string s;
            for (size_t k = 0; k < a.size(); ++k) {
                const long long c = check_integer(a, k, "char", line);
                if (c < 0 || c > 255) fail(line, "bad argument #" + std::to_string(k + 1) + " to 'char' (value out of range)");
                s.push_back(static_cast<char>(c));
            }
            r.push_back(Value::str(s));
        });

        // table
        auto T = new_table();
        G->set("table", Value::table(T));
        def(T, "insert", [](Interp &, Values &a, Values &, int line) { // ltablib.c tinsert
            const auto &t = check_table(a, 0, "insert", line);
            const long long e = t->length() + 1;
            if (a.size() == 2) { t->seti(e, a[1]); return; }
            if (a.size() != 3) fail(line, "wrong number of arguments to 'insert'");
            const long long pos = check_integer(a, 1, "insert", line);
            if (pos < 1 || pos > e) fail(line, "bad argument #2 to 'insert' (position out of bounds)");
            for (long long k = e; k > pos; --k) { const Value *v = t->at(k - 1); t->seti(k, v ? *v : Value{}); }
            t->seti(pos, a[2]);
        });
        def(T, "remove", [](Interp &, Values &a, Values &r, int line) { // ltablib.c tremove
            const auto &t = check_table(a, 0, "remove", line);
            const long long size = t->length();
            long long pos = a.size() > 1 ? check_integer(a, 1, "remove", line) : size;
            if (a.size() > 1 && size + 1 != pos && (pos < 1 || pos > size + 1)) fail(line, "bad argument #2 to 'remove' (position out of bounds)");
            const Value *v = t->at(pos);
            r.push_back(v ? *v : Value{});
            for (; pos < size; ++pos) { const Value *nx = t->at(pos + 1); t->seti(pos, nx ? *nx : Value{}); }
            if (pos <= size) t->seti(pos, Value{});
        });
        def(T, "concat", [](Interp &in, Values &a, Values &r, int line) {
            const auto &t = check_table(a, 0, "concat", line);
            const std::string sep = (a.size() > 1 && a[1].kind != Value::Nil) ? check_string(a, 1, "concat", line) : std::string();
            const long long i = a.size() > 2 ? check_integer(a, 2, "concat", line) : 1, j = a.size() > 3 ? check_integer(a, 3, "concat", line) : t->length();
            std::string out;
            for (long long k = i; k <= j; ++k) {
                const Value *v = t->at(k);
                if (!v || !(v->kind == Value::Str || v->is_number())) fail(line, "invalid value (at index " + std::to_string(k) + ") in table for 'concat'");
                if (k > i) out += sep;
                out += in.concat_piece(*v, line);
                if (out.size() > (size_t(1) << 26)) fail(line, "resulting string too large");
            }
            r.push_back(Value::str(out));
        });
        def(T, "unpack", [](Interp &, Values &a, Values &r, int line) {
            const auto &t = check_table(a, 0, "unpack", line);
            const long long i = (a.size() > 1 && a[1].kind != Value::Nil) ? check_integer(a, 1, "unpack", line) : 1,
                            j = (a.size() > 2 && a[2].kind != Value::Nil) ? check_integer(a, 2, "unpack", line) : t->length();
            if (j - i >= 1000000) fail(line, "too many results to unpack");
            for (long long k = i; k <= j; ++k) { const Value *v = t->at(k); r.push_back(v ? *v : Value{}); }
        });
        G->set("unpack", *T->get("unpack"));

        // the reference's own three entry points (lua.rs:50-91)
        def(G, "Render", [](Interp &in, Values &a, Values &r, int line) {
            if (arg(a, 2).kind != Value::Str && !arg(a, 2).is_number()) fail(line, "Render expects (world table, camera table, output file name)");
            in.record_job(RTC_LUA_JOB_RENDER, arg(a, 0), arg(a, 1), tostring(a[2]), 0, line);
            r.push_back(Value::str("Inside")); // lua.rs:69
        });
        def(G, "StartAnimation", [](Interp &in, Values &a, Values &r, int line) {
            if (arg(a, 0).kind != Value::Str && !arg(a, 0).is_number()) fail(line, "StartAnimation expects an output file name");
            const uint32_t id = in.animations++;
            in.frames_of.push_back(0);
            const std::string outfile = tostring(a[0]);
            auto enc = in.new_table(); // the GifEncoder userdata: an object with two methods
            in.def(enc, "AddFrame", [id, outfile](Interp &in2, Values &b, Values &, int l) { in2.record_job(RTC_LUA_JOB_ADD_FRAME, arg(b, 1), arg(b, 2), outfile, id, l); });
            in.def(enc, "Finish", [](Interp &, Values &, Values &, int) {});
            r.push_back(Value::table(enc));
        });
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

void world_from_table(const Table &t, SceneData &sc, int line) { // lua.rs:293-330
    const Table &lights = table_of(t.get("lights"), "world.lights", line);
    const Table &l1 = table_of(lights.at(1), "world.lights[1]", line); // lights_from_table: only the first
    std::memset(&sc.light, 0, sizeof sc.light);
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
        std::memset(&s, 0, sizeof s); // (padding too: worlds are compared byte for byte)
        const rtc_status rs = rtc_shape_init(kind, xf, &mat, &s);
        if (rs != RTC_OK) fail(line, type->s + " " + std::to_string(k) + ": " + rtc_strerror(rs));
        s.world_id = static_cast<uint32_t>(sc.shapes.size()) + 1; // World::add_shape shape.rs:661-667
        sc.shapes.push_back(s);
    }
}

void set_err(char *errbuf, size_t len, const std::string &msg) {
    if (errbuf && len) std::snprintf(errbuf, len, "%s", msg.c_str());
}

bool read_file(const char *path, std::string &text, std::string &why) {
    std::FILE *f = std::fopen(path, "rb");
    if (!f) { why = std::string("cannot open ") + path; return false; }
    char buf[4096];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) {
        text.append(buf, n);
        if (text.size() > (size_t(16) << 20)) { std::fclose(f); why = std::string(path) + " is larger than 16 MiB"; return false; }
    }
    std::fclose(f);
    return true;
}

} // namespace

struct rtc_lua_program {
    Interp in;
};

namespace {

rtc_status run_program(const char *text, const char *base_dir, const char *chunk_name, uint64_t step_limit, rtc_lua_program **out, char *errbuf,
                       size_t errbuf_len) {
    if (!text || !out) return RTC_ERR_ARG;
    *out = nullptr;
    std::unique_ptr<rtc_lua_program> prog;
    try {
        prog.reset(new rtc_lua_program());
        Interp &in = prog->in;
        if (base_dir) { in.base_dir = base_dir; in.have_base_dir = true; }
        if (step_limit) in.step_limit = step_limit;
        in.open_libraries();
        in.run_chunk(text, chunk_name);
        *out = prog.release();
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

std::string dir_of(const char *path) {
    const std::string p = path;
    const size_t k = p.find_last_of('/');
    return k == std::string::npos ? std::string(".") : (k == 0 ? std::string("/") : p.substr(0, k));
}

} // namespace

extern "C" {

rtc_status rtc_lua_run(const char *text, const char *base_dir, uint64_t step_limit, rtc_lua_program **out, char *errbuf, size_t errbuf_len) {
    return run_program(text, base_dir, "script", step_limit, out, errbuf, errbuf_len);
}

rtc_status rtc_lua_run_file(const char *path, uint64_t step_limit, rtc_lua_program **out, char *errbuf, size_t errbuf_len) {
    if (!path || !out) return RTC_ERR_ARG;
    *out = nullptr;
    std::string text, why;
    if (!read_file(path, text, why)) { set_err(errbuf, errbuf_len, why); return RTC_ERR_IO; }
    if (text.find('\0') != std::string::npos) { set_err(errbuf, errbuf_len, std::string(path) + " is not a text file"); return RTC_ERR_PARSE; }
    const std::string p = path;
    const size_t k = p.find_last_of('/');
    return run_program(text.c_str(), dir_of(path).c_str(), (k == std::string::npos ? p : p.substr(k + 1)).c_str(), step_limit, out, errbuf, errbuf_len);
}

uint32_t rtc_lua_program_jobs(const rtc_lua_program *prog) { return prog ? static_cast<uint32_t>(prog->in.jobs.size()) : 0u; }

rtc_status rtc_lua_program_job(const rtc_lua_program *prog, uint32_t index, rtc_lua_job *job) {
    if (!prog || !job || index >= prog->in.jobs.size()) return RTC_ERR_ARG;
    const Job &j = prog->in.jobs[index];
    job->shapes = j.scene->shapes.empty() ? nullptr : j.scene->shapes.data();
    job->n_shapes = static_cast<uint32_t>(j.scene->shapes.size());
    job->light = j.scene->light;
    job->camera = j.camera;
    job->outfile = j.outfile.c_str();
    job->kind = j.kind;
    job->animation = j.animation;
    job->frame = j.frame;
    job->same_world_as_previous = j.same_world ? 1u : 0u;
    job->line = static_cast<uint32_t>(j.line);
    return RTC_OK;
}

const char *rtc_lua_program_output(const rtc_lua_program *prog) { return prog ? prog->in.output.c_str() : ""; }

void rtc_lua_program_free(rtc_lua_program *prog) { delete prog; }

// The single-scene form: run the script, hand out one of its jobs as malloc'ed arrays.
static rtc_status load_one(rtc_lua_program *prog, uint32_t render_index, rtc_shape **shapes_out, uint32_t *n_out, rtc_light *light_out,
                           rtc_camera *camera_out, char *outfile, size_t outfile_len, uint32_t *renders_out, char *errbuf, size_t errbuf_len) {
    std::unique_ptr<rtc_lua_program> own(prog);
    Interp &in = prog->in;
    try {
        if (renders_out) *renders_out = static_cast<uint32_t>(in.jobs.size());
        if (in.jobs.empty()) { // no Render / AddFrame call: the globals `world` and `camera`, if the script defines them
            const Value *w = in.globals->get("world"), *c = in.globals->get("camera");
            if (render_index != 0 || !w || !c || w->kind != Value::Tab || c->kind != Value::Tab)
                fail(0, "the script calls Render / AddFrame 0 time(s) and defines no global world / camera tables");
            in.record_job(RTC_LUA_JOB_RENDER, *w, *c, "", 0, 0);
        } else if (render_index >= in.jobs.size()) {
            fail(0, "the script calls Render / AddFrame " + std::to_string(in.jobs.size()) + " time(s)");
        }
        const Job &j = in.jobs[render_index];
        if (outfile && outfile_len) std::snprintf(outfile, outfile_len, "%s", j.outfile.c_str());
        const size_t count = j.scene->shapes.size();
        rtc_shape *arr = static_cast<rtc_shape *>(std::malloc(sizeof(rtc_shape) * (count ? count : 1)));
        if (!arr) return RTC_ERR_NOMEM;
        if (count) std::memcpy(arr, j.scene->shapes.data(), sizeof(rtc_shape) * count);
        *shapes_out = arr;
        *n_out = static_cast<uint32_t>(count);
        *light_out = j.scene->light;
        *camera_out = j.camera;
        return RTC_OK;
    } catch (const LuaError &e) {
        set_err(errbuf, errbuf_len, e.what());
        return RTC_ERR_PARSE;
    } catch (const std::bad_alloc &) {
        return RTC_ERR_NOMEM;
    }
}

rtc_status rtc_scene_load_lua(const char *text, uint32_t render_index, rtc_shape **shapes_out, uint32_t *n_out, rtc_light *light_out,
                              rtc_camera *camera_out, char *outfile, size_t outfile_len, uint32_t *renders_out, char *errbuf,
                              size_t errbuf_len) {
    if (!text || !shapes_out || !n_out || !light_out || !camera_out) return RTC_ERR_ARG;
    *shapes_out = nullptr;
    *n_out = 0;
    if (renders_out) *renders_out = 0;
    if (outfile && outfile_len) outfile[0] = 0;
    rtc_lua_program *prog = nullptr;
    const rtc_status st = rtc_lua_run(text, nullptr, 0, &prog, errbuf, errbuf_len);
    if (st != RTC_OK) return st;
    return load_one(prog, render_index, shapes_out, n_out, light_out, camera_out, outfile, outfile_len, renders_out, errbuf, errbuf_len);
}

rtc_status rtc_scene_load_lua_file(const char *path, uint32_t render_index, rtc_shape **shapes_out, uint32_t *n_out, rtc_light *light_out,
                                   rtc_camera *camera_out, char *outfile, size_t outfile_len, uint32_t *renders_out, char *errbuf,
                                   size_t errbuf_len) {
    if (!path || !shapes_out || !n_out || !light_out || !camera_out) return RTC_ERR_ARG;
    *shapes_out = nullptr;
    *n_out = 0;
    if (renders_out) *renders_out = 0;
    if (outfile && outfile_len) outfile[0] = 0;
    rtc_lua_program *prog = nullptr;
    const rtc_status st = rtc_lua_run_file(path, 0, &prog, errbuf, errbuf_len);
    if (st != RTC_OK) return st;
    return load_one(prog, render_index, shapes_out, n_out, light_out, camera_out, outfile, outfile_len, renders_out, errbuf, errbuf_len);
}

} // extern "C"
