// YAML subset + meval-compatible expressions + typed reads; mirrors the
// behaviour of src/setup.rs (and of the yaml-rust 0.4.3 / meval 0.2.0 crates it
// sits on, as far as opal's input files exercise them).
#include "config.hpp"

#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace opal {

// src/constants.rs
static const double K_C = 2.997925e8;
static const double K_EPS0 = 8.854188e-12;
static const double K_E = 1.602177e-19;
static const double K_ME = 9.109383e-31;
static const double K_MP = 1.672622e-27;

// =============================================================== expressions
struct Expr {
    enum Kind { Num, Var, Neg, Pos, Fact, Add, Sub, Mul, Div, Rem, Pow, Call } kind;
    double value = 0.0;
    std::string name;
    std::vector<ExprPtr> args;
};

namespace {

struct Parser {
    const std::string &s;
    size_t i = 0;
    std::string err;
    explicit Parser(const std::string &t) : s(t) {}

    void ws()
    {
        while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
    }
    bool peek(char c)
    {
        ws();
        return i < s.size() && s[i] == c;
    }
    ExprPtr mk(Expr::Kind k, ExprPtr a = nullptr, ExprPtr b = nullptr)
    {
        auto e = std::make_shared<Expr>();
        e->kind = k;
        if (a) e->args.push_back(a);
        if (b) e->args.push_back(b);
        return e;
    }
    // meval precedences: + - (1, left) < * / % (2, left) < unary + - (3) < ^ (4, right) < ! (5)
    ExprPtr expr()
    {
        ExprPtr l = term();
        while (l) {
            if (peek('+')) { i++; ExprPtr r = term(); if (!r) return nullptr; l = mk(Expr::Add, l, r); }
            else if (peek('-')) { i++; ExprPtr r = term(); if (!r) return nullptr; l = mk(Expr::Sub, l, r); }
            else break;
        }
        return l;
    }
    ExprPtr term()
    {
        ExprPtr l = unary();
        while (l) {
            if (peek('*')) { i++; ExprPtr r = unary(); if (!r) return nullptr; l = mk(Expr::Mul, l, r); }
            else if (peek('/')) { i++; ExprPtr r = unary(); if (!r) return nullptr; l = mk(Expr::Div, l, r); }
            else if (peek('%')) { i++; ExprPtr r = unary(); if (!r) return nullptr; l = mk(Expr::Rem, l, r); }
            else break;
        }
        return l;
    }
    ExprPtr unary()
    {
        if (peek('-')) { i++; ExprPtr a = unary(); return a ? mk(Expr::Neg, a) : nullptr; }
        if (peek('+')) { i++; ExprPtr a = unary(); return a ? mk(Expr::Pos, a) : nullptr; }
        return power();
    }
    ExprPtr power()
    {
        ExprPtr base = postfix();
        if (!base) return nullptr;
        if (peek('^')) {
            i++;
            ExprPtr ex = unary();  // right associative; a unary sign may follow '^'
            if (!ex) return nullptr;
            return mk(Expr::Pow, base, ex);
        }
        return base;
    }
    ExprPtr postfix()
    {
        ExprPtr a = atom();
        while (a && peek('!')) { i++; a = mk(Expr::Fact, a); }
        return a;
    }
    ExprPtr atom()
    {
        ws();
        if (i >= s.size()) { err = "unexpected end of expression"; return nullptr; }
        char c = s[i];
        if (c == '(') {
            i++;
            ExprPtr e = expr();
            if (!e) return nullptr;
            if (!peek(')')) { err = "missing ')'"; return nullptr; }
            i++;
            return e;
        }
        if (std::isdigit((unsigned char)c) || c == '.') {
            size_t j = i;
            while (j < s.size() && std::isdigit((unsigned char)s[j])) j++;
            if (j < s.size() && s[j] == '.') { j++; while (j < s.size() && std::isdigit((unsigned char)s[j])) j++; }
            if (j < s.size() && (s[j] == 'e' || s[j] == 'E')) {
                size_t k = j + 1;
                if (k < s.size() && (s[k] == '+' || s[k] == '-')) k++;
                if (k < s.size() && std::isdigit((unsigned char)s[k])) {
                    while (k < s.size() && std::isdigit((unsigned char)s[k])) k++;
                    j = k;
                }
            }
            std::string tok = s.substr(i, j - i);
            if (tok == ".") { err = "bad number"; return nullptr; }
            auto e = mk(Expr::Num);
            e->value = std::strtod(tok.c_str(), nullptr);
            i = j;
            return e;
        }
        if (std::isalpha((unsigned char)c) || c == '_') {
            size_t j = i;
            while (j < s.size() && (std::isalnum((unsigned char)s[j]) || s[j] == '_')) j++;
            std::string name = s.substr(i, j - i);
            i = j;
            if (peek('(')) {
                i++;
                auto e = mk(Expr::Call);
                e->name = name;
                if (peek(')')) { i++; return e; }
                for (;;) {
                    ExprPtr a = expr();
                    if (!a) return nullptr;
                    e->args.push_back(a);
                    if (peek(',')) { i++; continue; }
                    if (peek(')')) { i++; break; }
                    err = "expected ',' or ')' in call to " + name;
                    return nullptr;
                }
                return e;
            }
            auto e = mk(Expr::Var);
            e->name = name;
            return e;
        }
        err = std::string("unexpected character '") + c + "'";
        return nullptr;
    }
};

}  // namespace

ExprPtr parse_expr(const std::string &text, std::string &err)
{
    Parser p(text);
    ExprPtr e = p.expr();
    if (e) {
        p.ws();
        if (p.i != text.size()) {
            err = "trailing characters in expression";
            return nullptr;
        }
    } else {
        err = p.err.empty() ? "syntax error" : p.err;
    }
    return e;
}

bool eval_expr(const Expr &e, const Context &ctx, const std::vector<std::string> &names, const double *values,
               double &out, std::string &err)
{
    auto sub = [&](size_t k, double &v) { return eval_expr(*e.args[k], ctx, names, values, v, err); };
    double a, b;
    switch (e.kind) {
    case Expr::Num: out = e.value; return true;
    case Expr::Var: {
        for (size_t k = 0; k < names.size(); k++)
            if (names[k] == e.name) { out = values[k]; return true; }
        auto it = ctx.vars.find(e.name);
        if (it == ctx.vars.end()) { err = "unknown variable " + e.name; return false; }
        out = it->second;
        return true;
    }
    case Expr::Neg: if (!sub(0, a)) return false; out = -a; return true;
    case Expr::Pos: if (!sub(0, a)) return false; out = a; return true;
    case Expr::Fact: if (!sub(0, a)) return false; out = std::tgamma(a + 1.0); return true;
    case Expr::Add: if (!sub(0, a) || !sub(1, b)) return false; out = a + b; return true;
    case Expr::Sub: if (!sub(0, a) || !sub(1, b)) return false; out = a - b; return true;
    case Expr::Mul: if (!sub(0, a) || !sub(1, b)) return false; out = a * b; return true;
    case Expr::Div: if (!sub(0, a) || !sub(1, b)) return false; out = a / b; return true;
    case Expr::Rem: if (!sub(0, a) || !sub(1, b)) return false; out = std::fmod(a, b); return true;
    case Expr::Pow: if (!sub(0, a) || !sub(1, b)) return false; out = std::pow(a, b); return true;
    case Expr::Call: {
        auto it = ctx.funcs.find(e.name);
        if (it == ctx.funcs.end()) { err = "unknown function " + e.name; return false; }
        int arity = it->second.first;
        if ((arity >= 0 && (int)e.args.size() != arity) || (arity < 0 && e.args.empty())) {
            err = "wrong number of arguments to " + e.name;
            return false;
        }
        std::vector<double> av(e.args.size());
        for (size_t k = 0; k < e.args.size(); k++)
            if (!sub(k, av[k])) return false;
        out = it->second.second(av);
        return true;
    }
    }
    return false;
}

Context Context::builtins()
{
    Context c;
    c.vars["pi"] = 3.14159265358979323846264338327950288;
    c.vars["e"] = 2.71828182845904523536028747135266250;
    auto f1 = [&c](const char *n, double (*fn)(double)) {
        c.funcs[n] = {1, [fn](const std::vector<double> &a) { return fn(a[0]); }};
    };
    f1("sqrt", std::sqrt); f1("exp", std::exp); f1("ln", std::log); f1("abs", std::fabs);
    f1("sin", std::sin); f1("cos", std::cos); f1("tan", std::tan);
    f1("asin", std::asin); f1("acos", std::acos); f1("atan", std::atan);
    f1("sinh", std::sinh); f1("cosh", std::cosh); f1("tanh", std::tanh);
    f1("asinh", std::asinh); f1("acosh", std::acosh); f1("atanh", std::atanh);
    f1("floor", std::floor); f1("ceil", std::ceil); f1("round", std::round);
    c.funcs["signum"] = {1, [](const std::vector<double> &a) { return std::isnan(a[0]) ? a[0] : std::copysign(1.0, a[0]); }};
    c.funcs["atan2"] = {2, [](const std::vector<double> &a) { return std::atan2(a[0], a[1]); }};
    c.funcs["max"] = {-1, [](const std::vector<double> &a) { double m = a[0]; for (double v : a) m = std::fmax(m, v); return m; }};
    c.funcs["min"] = {-1, [](const std::vector<double> &a) { double m = a[0]; for (double v : a) m = std::fmin(m, v); return m; }};
    return c;
}

// A bound expression flattened to a stack program: names are resolved once (arguments to slots,
// context variables to constants), so evaluating it per particle costs no lookups.
struct Program {
    enum Op { PushConst, PushArg, Neg, Fact, Add, Sub, Mul, Div, Rem, Pow, CallN };
    struct Ins {
        Op op;
        double value;
        int arg;
        int nargs;
        const NativeFn *fn;
    };
    std::vector<Ins> code;
    std::vector<NativeFn> fns;
    int max_stack = 0;
    double run(const double *args) const
    {
        double st[64];
        int sp = 0;
        for (const Ins &in : code) {
            switch (in.op) {
            case PushConst: st[sp++] = in.value; break;
            case PushArg: st[sp++] = args[in.arg]; break;
            case Neg: st[sp - 1] = -st[sp - 1]; break;
            case Fact: st[sp - 1] = std::tgamma(st[sp - 1] + 1.0); break;
            case Add: sp--; st[sp - 1] = st[sp - 1] + st[sp]; break;
            case Sub: sp--; st[sp - 1] = st[sp - 1] - st[sp]; break;
            case Mul: sp--; st[sp - 1] = st[sp - 1] * st[sp]; break;
            case Div: sp--; st[sp - 1] = st[sp - 1] / st[sp]; break;
            case Rem: sp--; st[sp - 1] = std::fmod(st[sp - 1], st[sp]); break;
            case Pow: sp--; st[sp - 1] = std::pow(st[sp - 1], st[sp]); break;
            case CallN: {
                static thread_local std::vector<double> av;
                av.assign(st + sp - in.nargs, st + sp);
                sp -= in.nargs;
                st[sp++] = fns[(size_t)in.arg](av);
                break;
            }
            }
        }
        return st[0];
    }
};

static bool compile_rec(const Expr &e, const Context &ctx, const std::vector<std::string> &names, Program &p, int depth,
                        std::string &err)
{
    if (depth > 48) { err = "expression too deep"; return false; }
    auto emit = [&](Program::Op op, double v = 0.0, int arg = 0, int nargs = 0) {
        p.code.push_back({op, v, arg, nargs, nullptr});
    };
    auto sub = [&](size_t k, int d) { return compile_rec(*e.args[k], ctx, names, p, d, err); };
    switch (e.kind) {
    case Expr::Num: emit(Program::PushConst, e.value); return true;
    case Expr::Var: {
        for (size_t k = 0; k < names.size(); k++)
            if (names[k] == e.name) { emit(Program::PushArg, 0.0, (int)k); return true; }
        auto it = ctx.vars.find(e.name);
        if (it == ctx.vars.end()) { err = "unknown variable " + e.name; return false; }
        emit(Program::PushConst, it->second);
        return true;
    }
    case Expr::Neg: if (!sub(0, depth)) return false; emit(Program::Neg); return true;
    case Expr::Pos: return sub(0, depth);
    case Expr::Fact: if (!sub(0, depth)) return false; emit(Program::Fact); return true;
    case Expr::Add: case Expr::Sub: case Expr::Mul: case Expr::Div: case Expr::Rem: case Expr::Pow: {
        if (!sub(0, depth) || !sub(1, depth + 1)) return false;
        Program::Op op = e.kind == Expr::Add ? Program::Add : e.kind == Expr::Sub ? Program::Sub
                         : e.kind == Expr::Mul ? Program::Mul : e.kind == Expr::Div ? Program::Div
                         : e.kind == Expr::Rem ? Program::Rem : Program::Pow;
        emit(op);
        return true;
    }
    case Expr::Call: {
        auto it = ctx.funcs.find(e.name);
        if (it == ctx.funcs.end()) { err = "unknown function " + e.name; return false; }
        int arity = it->second.first;
        if ((arity >= 0 && (int)e.args.size() != arity) || (arity < 0 && e.args.empty())) {
            err = "wrong number of arguments to " + e.name;
            return false;
        }
        for (size_t k = 0; k < e.args.size(); k++)
            if (!sub(k, depth + (int)k)) return false;
        p.fns.push_back(it->second.second);
        emit(Program::CallN, 0.0, (int)p.fns.size() - 1, (int)e.args.size());
        return true;
    }
    }
    return false;
}

static bool compile(const Expr &e, const Context &ctx, const std::vector<std::string> &names, Program &p, std::string &err)
{
    return compile_rec(e, ctx, names, p, 0, err);
}

// ======================================================================= YAML
namespace {

bool rust_parse_i64(const std::string &v, long long &out)
{
    if (v.empty()) return false;
    size_t i = 0;
    if (v[0] == '+' || v[0] == '-') i = 1;
    if (i >= v.size()) return false;
    for (size_t k = i; k < v.size(); k++)
        if (!std::isdigit((unsigned char)v[k])) return false;
    errno = 0;
    char *end = nullptr;
    out = std::strtoll(v.c_str(), &end, 10);
    return errno == 0 && end && *end == 0;
}

// Rust's <f64 as FromStr> grammar: [sign] (inf | infinity | nan | digits [. digits] [exp] | . digits [exp])
bool rust_parse_f64(const std::string &v, double &out)
{
    size_t i = 0, n = v.size();
    if (n == 0) return false;
    if (v[i] == '+' || v[i] == '-') i++;
    std::string rest = v.substr(i);
    std::string lower;
    for (char c : rest) lower.push_back((char)std::tolower((unsigned char)c));
    if (lower == "inf" || lower == "infinity") { out = (v[0] == '-') ? -INFINITY : INFINITY; return true; }
    if (lower == "nan") { out = NAN; return true; }
    size_t j = i, digits = 0;
    while (j < n && std::isdigit((unsigned char)v[j])) { j++; digits++; }
    if (j < n && v[j] == '.') { j++; while (j < n && std::isdigit((unsigned char)v[j])) { j++; digits++; } }
    if (digits == 0) return false;
    if (j < n && (v[j] == 'e' || v[j] == 'E')) {
        j++;
        if (j < n && (v[j] == '+' || v[j] == '-')) j++;
        size_t ed = 0;
        while (j < n && std::isdigit((unsigned char)v[j])) { j++; ed++; }
        if (ed == 0) return false;
    }
    if (j != n) return false;
    out = std::strtod(v.c_str(), nullptr);
    return true;
}

// yaml-rust 0.4.3 Yaml::from_str + parse_f64
YamlNode resolve_plain(const std::string &v)
{
    YamlNode n;
    n.text = v;
    long long iv;
    double dv;
    if (v.rfind("0x", 0) == 0) {
        char *end = nullptr;
        long long x = std::strtoll(v.c_str() + 2, &end, 16);
        if (end && *end == 0 && v.size() > 2) { n.kind = YamlNode::Integer; n.ival = x; return n; }
    }
    if (v.rfind("0o", 0) == 0) {
        char *end = nullptr;
        long long x = std::strtoll(v.c_str() + 2, &end, 8);
        if (end && *end == 0 && v.size() > 2) { n.kind = YamlNode::Integer; n.ival = x; return n; }
    }
    if (!v.empty() && v[0] == '+' && rust_parse_i64(v.substr(1), iv)) { n.kind = YamlNode::Integer; n.ival = iv; return n; }
    if (v == "~" || v == "null" || v.empty()) { n.kind = YamlNode::Null; return n; }
    if (v == "true") { n.kind = YamlNode::Boolean; n.bval = true; return n; }
    if (v == "false") { n.kind = YamlNode::Boolean; n.bval = false; return n; }
    if (rust_parse_i64(v, iv)) { n.kind = YamlNode::Integer; n.ival = iv; return n; }
    if (v == ".inf" || v == ".Inf" || v == ".INF" || v == "+.inf" || v == "+.Inf" || v == "+.INF" ||
        v == "-.inf" || v == "-.Inf" || v == "-.INF" || v == ".nan" || v == "NaN" || v == ".NAN" ||
        rust_parse_f64(v, dv)) {
        n.kind = YamlNode::Real;
        return n;
    }
    n.kind = YamlNode::String;
    return n;
}

struct Line {
    int indent;
    std::string text;  // without indentation, comments stripped, right-trimmed
    std::string raw;   // original line (for block scalars)
};

std::string rtrim(const std::string &s)
{
    size_t e = s.size();
    while (e > 0 && std::isspace((unsigned char)s[e - 1])) e--;
    return s.substr(0, e);
}
std::string trim(const std::string &s)
{
    size_t b = 0;
    while (b < s.size() && std::isspace((unsigned char)s[b])) b++;
    return rtrim(s.substr(b));
}

std::string strip_comment(const std::string &s)
{
    bool sq = false, dq = false;
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[i];
        if (c == '\'' && !dq) sq = !sq;
        else if (c == '"' && !sq) dq = !dq;
        else if (c == '#' && !sq && !dq && (i == 0 || std::isspace((unsigned char)s[i - 1]))) return s.substr(0, i);
    }
    return s;
}

struct YamlParser {
    std::vector<Line> lines;
    std::string err;

    YamlNode scalar(const std::string &v)
    {
        if (v.size() >= 2 && ((v.front() == '"' && v.back() == '"') || (v.front() == '\'' && v.back() == '\''))) {
            YamlNode n;
            n.kind = YamlNode::String;
            n.text = v.substr(1, v.size() - 2);
            return n;
        }
        return resolve_plain(v);
    }

    YamlNode flow_seq(const std::string &v)
    {
        YamlNode n;
        n.kind = YamlNode::Array;
        std::string inner = trim(v.substr(1, v.size() - 2));
        // split on commas outside brackets/parentheses/quotes
        int depth = 0;
        bool sq = false, dq = false;
        std::string cur;
        auto push = [&]() {
            std::string t = trim(cur);
            if (!t.empty()) n.items.push_back(scalar(t));
            cur.clear();
        };
        for (char c : inner) {
            if (c == '\'' && !dq) sq = !sq;
            if (c == '"' && !sq) dq = !dq;
            if (!sq && !dq) {
                if (c == '[' || c == '(' || c == '{') depth++;
                if (c == ']' || c == ')' || c == '}') depth--;
                if (c == ',' && depth == 0) { push(); continue; }
            }
            cur.push_back(c);
        }
        push();
        return n;
    }

    // block scalar starting after line `idx` (the key line has indent `key_indent`)
    YamlNode block_scalar(size_t &idx, int key_indent, char style, char chomp)
    {
        std::vector<std::string> body;
        int bi = -1;
        size_t k = idx;
        for (; k < raw_lines.size(); k++) {
            const std::string &r = raw_lines[k];
            std::string t = trim(r);
            int ind = 0;
            while (ind < (int)r.size() && r[ind] == ' ') ind++;
            if (t.empty()) { body.push_back(""); continue; }
            if (ind <= key_indent) break;
            if (bi < 0) bi = ind;
            body.push_back(rtrim(r.substr(std::min<size_t>((size_t)bi, r.size()))));
        }
        idx = k;
        while (!body.empty() && body.back().empty()) body.pop_back();
        std::string out;
        if (style == '|') {
            for (auto &l : body) { out += l; out += "\n"; }
        } else {
            for (size_t q = 0; q < body.size(); q++) {
                if (body[q].empty()) { out += "\n"; continue; }
                if (!out.empty() && out.back() != '\n') out += " ";
                out += body[q];
            }
            if (!body.empty()) out += "\n";
        }
        if (chomp == '-')
            while (!out.empty() && out.back() == '\n') out.pop_back();
        YamlNode n;
        n.kind = YamlNode::String;
        n.text = out;
        return n;
    }

    std::vector<std::string> raw_lines;
    size_t pos = 0;  // index into raw_lines

    bool next_content(size_t from, size_t &idx, int &indent, std::string &text)
    {
        for (size_t k = from; k < raw_lines.size(); k++) {
            std::string s = rtrim(strip_comment(raw_lines[k]));
            std::string t = trim(s);
            if (t.empty() || t == "---" || t == "...") continue;
            int ind = 0;
            while (ind < (int)s.size() && s[ind] == ' ') ind++;
            idx = k;
            indent = ind;
            text = t;
            return true;
        }
        return false;
    }

    // parses the block whose first content line is at `pos` with indentation `indent`
    bool block(int indent, YamlNode &out)
    {
        size_t idx;
        int ind;
        std::string text;
        if (!next_content(pos, idx, ind, text) || ind != indent) { out.kind = YamlNode::Null; return true; }
        if (text.rfind("- ", 0) == 0 || text == "-") {
            out.kind = YamlNode::Array;
            while (next_content(pos, idx, ind, text) && ind == indent && (text.rfind("- ", 0) == 0 || text == "-")) {
                pos = idx + 1;
                std::string v = trim(text.substr(1));
                if (!v.empty() && v.front() == '[' && v.back() == ']') out.items.push_back(flow_seq(v));
                else out.items.push_back(scalar(v));
            }
            return true;
        }
        out.kind = YamlNode::Hash;
        while (next_content(pos, idx, ind, text) && ind == indent) {
            size_t colon = std::string::npos;
            for (size_t k = 0; k < text.size(); k++)
                if (text[k] == ':' && (k + 1 == text.size() || text[k + 1] == ' ')) { colon = k; break; }
            if (colon == std::string::npos) { err = "expected 'key: value' at line " + std::to_string(idx + 1); return false; }
            std::string key = trim(text.substr(0, colon));
            if (key.size() >= 2 && (key.front() == '"' || key.front() == '\'')) key = key.substr(1, key.size() - 2);
            std::string val = trim(text.substr(colon + 1));
            pos = idx + 1;
            YamlNode child;
            if (val.empty()) {
                size_t nidx;
                int nind;
                std::string ntext;
                if (next_content(pos, nidx, nind, ntext) && nind > indent) {
                    if (!block(nind, child)) return false;
                } else {
                    child.kind = YamlNode::Null;
                }
            } else if (val[0] == '>' || val[0] == '|') {
                char chomp = val.size() > 1 ? val[1] : ' ';
                size_t k = pos;
                child = block_scalar(k, indent, val[0], chomp);
                pos = k;
            } else if (val.front() == '[' && val.back() == ']') {
                child = flow_seq(val);
            } else {
                child = scalar(val);
            }
            out.entries.emplace_back(key, child);
        }
        if (next_content(pos, idx, ind, text) && ind > indent) {
            err = "unexpected indentation at line " + std::to_string(idx + 1);
            return false;
        }
        return true;
    }
};

const YamlNode BAD_NODE;

}  // namespace

const YamlNode &YamlNode::operator[](const std::string &key) const
{
    if (kind != Hash) return BAD_NODE;
    for (auto &e : entries)
        if (e.first == key) return e.second;
    return BAD_NODE;
}

bool parse_yaml(const std::string &text, YamlNode &root, std::string &err)
{
    YamlParser p;
    std::istringstream is(text);
    std::string line;
    while (std::getline(is, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        for (char &c : line)
            if (c == '\t') c = ' ';
        p.raw_lines.push_back(line);
    }
    size_t idx;
    int ind;
    std::string t;
    if (!p.next_content(0, idx, ind, t)) { err = "empty document"; return false; }
    p.pos = idx;
    if (!p.block(ind, root)) { err = p.err; return false; }
    if (p.next_content(p.pos, idx, ind, t)) { err = "could not parse line " + std::to_string(idx + 1); return false; }
    return true;
}

// ===================================================================== Config
std::string ConfigError::message() const
{
    const char *help = "Usage: mpirun -n np ./opal input-file";
    switch (kind) {
    case ConfigErrorKind::MissingFile: return std::string("Unable to open configuration file.\n") + help;
    case ConfigErrorKind::MissingSection: return "Could not find section \"" + section + "\".\n" + help;
    case ConfigErrorKind::MissingField: return "Could not find field \"" + field + "\" in section \"" + section + "\".\n" + help;
    case ConfigErrorKind::ConversionFailure:
        return "Could not convert field \"" + field + "\" in section \"" + section + "\" to target type.\n" + help;
    }
    return help;
}

bool Config::from_file(const std::string &path, Config &out, ConfigError &err)
{
    std::ifstream f(path);
    if (!f) { err = {ConfigErrorKind::MissingFile, "", ""}; return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    return from_string(ss.str(), out, err);
}

bool Config::from_string(const std::string &text, Config &out, ConfigError &err)
{
    std::string perr;
    YamlNode root;
    if (!parse_yaml(text, root, perr)) { err = {ConfigErrorKind::MissingFile, "", ""}; return false; }
    out.input_ = root;
    out.ctx_ = Context::builtins();
    return true;
}

// setup.rs:110-179
void Config::with_context(const std::string &section)
{
    auto &v = ctx_.vars;
    v["m"] = K_ME; v["me"] = K_ME; v["mp"] = K_MP; v["c"] = K_C;
    v["e"] = K_E; v["eV"] = K_E; v["keV"] = 1.0e3 * K_E; v["MeV"] = 1.0e6 * K_E;
    v["femto"] = 1.0e-15; v["pico"] = 1.0e-12; v["nano"] = 1.0e-9; v["micro"] = 1.0e-6; v["milli"] = 1.0e-3;
    ctx_.funcs["step"] = {3, [](const std::vector<double> &a) { return (a[0] >= a[1] && a[0] < a[2]) ? 1.0 : 0.0; }};
    ctx_.funcs["gauss"] = {3, [](const std::vector<double> &a) {
        double d = a[0] - a[1];
        return std::exp(-(d * d) / (2.0 * (a[2] * a[2])));
    }};
    ctx_.funcs["critical"] = {1, [](const std::vector<double> &a) { return K_EPS0 * K_ME * (a[0] * a[0]) / (K_E * K_E); }};
    ctx_.funcs["gauss_pulse_re"] = {4, [](const std::vector<double> &a) {
        double t = a[0], x = a[1], omega = a[2], sigma = a[3];
        double phi = omega * (t - x / K_C);
        double carrier = std::sin(phi) + phi * std::cos(phi) / (sigma * sigma);
        double envelope = std::exp(-(phi * phi) / (2.0 * (sigma * sigma)));
        return carrier * envelope;
    }};
    ctx_.funcs["gauss_pulse_im"] = {4, [](const std::vector<double> &a) {
        double t = a[0], x = a[1], omega = a[2], sigma = a[3];
        double phi = omega * (t - x / K_C);
        double carrier = std::cos(phi) - phi * std::sin(phi) / (sigma * sigma);
        double envelope = std::exp(-(phi * phi) / (2.0 * (sigma * sigma)));
        return carrier * envelope;
    }};
    const YamlNode &sec = input_[section];
    if (sec.is_bad() || sec.kind != YamlNode::Hash) return;
    Context tmp = ctx_;  // "a constant cannot depend on other constants yet" (:160)
    for (auto &kv : sec.entries) {
        const YamlNode &b = kv.second;
        if (b.kind == YamlNode::Real) {
            double num;
            if (rust_parse_f64(b.text, num)) ctx_.vars[kv.first] = num;
        } else if (b.kind == YamlNode::String) {
            std::string perr;
            ExprPtr e = parse_expr(b.text, perr);
            double num;
            if (e && eval_expr(*e, tmp, {}, nullptr, num, perr)) ctx_.vars[kv.first] = num;
        }  // Integer constants are silently ignored, as in the reference (:165-175)
    }
}

bool Config::contains(const std::string &section) const { return !input_[section].is_bad(); }

bool Config::locate(const std::string &section, const std::string &field, const YamlNode *&node, ConfigError &err) const
{
    if (input_[section].is_bad()) { err = {ConfigErrorKind::MissingSection, section, field}; return false; }
    const YamlNode &n = input_[section][field];
    if (n.is_bad()) { err = {ConfigErrorKind::MissingField, section, field}; return false; }
    node = &n;
    return true;
}

// setup.rs:287-304
bool Config::read_f64(const std::string &section, const std::string &field, double &out, ConfigError &err) const
{
    const YamlNode *n;
    if (!locate(section, field, n, err)) return false;
    err = {ConfigErrorKind::ConversionFailure, section, field};
    if (n->kind == YamlNode::Real) {
        if (n->text == ".inf" || n->text == ".Inf" || n->text == ".INF") return false;  // Rust's parse rejects these
        return rust_parse_f64(n->text, out);
    }
    if (n->kind == YamlNode::Integer) { out = (double)n->ival; return true; }
    if (n->kind == YamlNode::String) {
        std::string perr;
        ExprPtr e = parse_expr(n->text, perr);
        return e && eval_expr(*e, ctx_, {}, nullptr, out, perr);
    }
    return false;
}

// setup.rs:306-322
bool Config::read_usize(const std::string &section, const std::string &field, size_t &out, ConfigError &err) const
{
    const YamlNode *n;
    if (!locate(section, field, n, err)) return false;
    err = {ConfigErrorKind::ConversionFailure, section, field};
    if (n->kind != YamlNode::Integer || n->ival < 0) return false;
    out = (size_t)n->ival;
    return true;
}

// setup.rs:324-332
bool Config::read_bool(const std::string &section, const std::string &field, bool &out, ConfigError &err) const
{
    const YamlNode *n;
    if (!locate(section, field, n, err)) return false;
    err = {ConfigErrorKind::ConversionFailure, section, field};
    if (n->kind != YamlNode::Boolean) return false;
    out = n->bval;
    return true;
}

// setup.rs:362-370
bool Config::read_string(const std::string &section, const std::string &field, std::string &out, ConfigError &err) const
{
    const YamlNode *n;
    if (!locate(section, field, n, err)) return false;
    err = {ConfigErrorKind::ConversionFailure, section, field};
    if (n->kind != YamlNode::String) return false;
    out = n->text;
    return true;
}

// setup.rs:334-360
bool Config::read_strings(const std::string &section, const std::string &field, std::vector<std::string> &out,
                          ConfigError &err) const
{
    const YamlNode *n;
    if (!locate(section, field, n, err)) return false;
    err = {ConfigErrorKind::ConversionFailure, section, field};
    out.clear();
    if (n->kind == YamlNode::String) { out.push_back(n->text); return true; }
    if (n->kind == YamlNode::Array) {
        for (auto &it : n->items)
            if (it.kind == YamlNode::String) out.push_back(it.text);
        return !out.empty();
    }
    return false;
}

// setup.rs:207-285
bool Config::func(const std::string &section, const std::string &field, const std::vector<std::string> &args,
                  std::function<double(const double *)> &out, ConfigError &err) const
{
    const YamlNode *n;
    if (!locate(section, field, n, err)) return false;
    err = {ConfigErrorKind::ConversionFailure, section, field};
    if (n->kind != YamlNode::String && n->kind != YamlNode::Real) return false;
    std::string perr;
    ExprPtr e = parse_expr(n->text, perr);
    if (!e) return false;
    // bind_with_context resolves every name once; unknown names fail here, not at call time
    auto prog = std::make_shared<Program>();
    if (!compile(*e, ctx_, args, *prog, perr)) return false;
    out = [prog](const double *v) { return prog->run(v); };
    return true;
}

bool Config::eval(const std::string &expr, double &out, std::string &err) const
{
    ExprPtr e = parse_expr(expr, err);
    return e && eval_expr(*e, ctx_, {}, nullptr, out, err);
}

}  // namespace opal
