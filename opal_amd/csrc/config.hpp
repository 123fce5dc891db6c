// Input configuration: the YAML subset opal's input files use plus a meval-
// compatible expression evaluator; mirrors src/setup.rs of the reference.
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace opal {

// ---------------------------------------------------------------- expressions
struct Expr;
using ExprPtr = std::shared_ptr<Expr>;
using NativeFn = std::function<double(const std::vector<double> &)>;

struct Context {
    std::map<std::string, double> vars;
    std::map<std::string, std::pair<int, NativeFn>> funcs;  // arity (-1 = variadic, >= 1 args)
    static Context builtins();                              // meval's default context
};

// parse_expr returns null and sets `err` on a syntax error
ExprPtr parse_expr(const std::string &text, std::string &err);
// evaluates with `bound` (positional arg names/values) shadowing the context; false on unknown name etc.
bool eval_expr(const Expr &e, const Context &ctx, const std::vector<std::string> &arg_names,
               const double *arg_values, double &out, std::string &err);

// ----------------------------------------------------------------------- YAML
struct YamlNode {
    enum Kind { Bad, Null, Boolean, Integer, Real, String, Array, Hash } kind = Bad;
    std::string text;  // scalar text (Real keeps its text, as yaml-rust does)
    long long ival = 0;
    bool bval = false;
    std::vector<YamlNode> items;                              // Array
    std::vector<std::pair<std::string, YamlNode>> entries;    // Hash, in file order
    const YamlNode &operator[](const std::string &key) const;
    bool is_bad() const { return kind == Bad; }
};
bool parse_yaml(const std::string &text, YamlNode &root, std::string &err);

// --------------------------------------------------------------------- Config
enum class ConfigErrorKind { MissingFile, MissingSection, MissingField, ConversionFailure };
struct ConfigError {
    ConfigErrorKind kind;
    std::string section, field;
    std::string message() const;  // setup.rs:50-61
};

class Config {
public:
    static bool from_file(const std::string &path, Config &out, ConfigError &err);
    static bool from_string(const std::string &text, Config &out, ConfigError &err);
    void with_context(const std::string &section);  // setup.rs:110-179
    bool contains(const std::string &section) const;

    bool read_f64(const std::string &section, const std::string &field, double &out, ConfigError &err) const;
    bool read_usize(const std::string &section, const std::string &field, size_t &out, ConfigError &err) const;
    bool read_bool(const std::string &section, const std::string &field, bool &out, ConfigError &err) const;
    bool read_string(const std::string &section, const std::string &field, std::string &out, ConfigError &err) const;
    bool read_strings(const std::string &section, const std::string &field, std::vector<std::string> &out,
                      ConfigError &err) const;
    // Config::func / func2 / func3 (setup.rs:207-285)
    bool func(const std::string &section, const std::string &field, const std::vector<std::string> &args,
              std::function<double(const double *)> &out, ConfigError &err) const;
    bool eval(const std::string &expr, double &out, std::string &err) const;

    const Context &context() const { return ctx_; }

private:
    bool locate(const std::string &section, const std::string &field, const YamlNode *&node, ConfigError &err) const;
    YamlNode input_;
    Context ctx_;
};

}  // namespace opal
