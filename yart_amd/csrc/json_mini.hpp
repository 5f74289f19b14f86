// json_mini.hpp — small JSON document reader for the glTF importer (gltf_reader.hpp). Host only.
// RFC 8259 values into a tree; numbers are kept as double (glTF readers narrow to float at the
// point of use, as fastgltf does for the reference, src/gltf/gltf.cpp:319-341).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace yart_hip {
namespace json {

struct Value {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;

  bool isNull() const { return kind == Null; }
  bool isObject() const { return kind == Object; }
  bool isArray() const { return kind == Array; }
  bool isNumber() const { return kind == Number; }
  bool isString() const { return kind == String; }
  // member lookup; a shared null value when absent (so chains of get() never throw)
  const Value& get(const char* key) const {
    static const Value none;
    if (kind != Object) return none;
    for (const auto& kv : obj) if (kv.first == key) return kv.second;
    return none;
  }
  bool has(const char* key) const { return !get(key).isNull(); }
  size_t size() const { return kind == Array ? arr.size() : kind == Object ? obj.size() : 0; }
  const Value& at(size_t i) const {
    if (kind != Array || i >= arr.size()) throw std::runtime_error("json: array index out of range");
    return arr[i];
  }
  double number(double dflt) const { return kind == Number ? num : dflt; }
  float numberF(float dflt) const { return kind == Number ? float(num) : dflt; }
  int64_t integer(int64_t dflt) const { return kind == Number ? int64_t(num) : dflt; }
  bool boolean(bool dflt) const { return kind == Bool ? b : dflt; }
  const std::string& string() const {
    static const std::string empty;
    return kind == String ? str : empty;
  }
};

class Parser {
 public:
  Parser(const char* p, size_t n) : p_(p), end_(p + n) {}
  Value parseDocument() {
    if (end_ - p_ >= 3 && uint8_t(p_[0]) == 0xEF && uint8_t(p_[1]) == 0xBB && uint8_t(p_[2]) == 0xBF) p_ += 3;
    Value v = parseValue(0);
    skipWs();
    if (p_ != end_) fail("trailing characters after the document");
    return v;
  }

 private:
  const char* p_;
  const char* end_;
  [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("json: ") + what); }
  void skipWs() { while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) p_++; }
  bool literal(const char* s) {
    size_t n = std::strlen(s);
    if (size_t(end_ - p_) >= n && std::memcmp(p_, s, n) == 0) { p_ += n; return true; }
    return false;
  }
  static void appendUtf8(std::string& out, uint32_t cp) {
    if (cp < 0x80) out.push_back(char(cp));
    else if (cp < 0x800) { out.push_back(char(0xC0 | (cp >> 6))); out.push_back(char(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
      out.push_back(char(0xE0 | (cp >> 12))); out.push_back(char(0x80 | ((cp >> 6) & 0x3F)));
      out.push_back(char(0x80 | (cp & 0x3F)));
    } else {
      out.push_back(char(0xF0 | (cp >> 18))); out.push_back(char(0x80 | ((cp >> 12) & 0x3F)));
      out.push_back(char(0x80 | ((cp >> 6) & 0x3F))); out.push_back(char(0x80 | (cp & 0x3F)));
    }
  }
  uint32_t hex4() {
    if (end_ - p_ < 4) fail("truncated \\u escape");
    uint32_t v = 0;
    for (int i = 0; i < 4; i++) {
      char c = *p_++;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= uint32_t(c - '0');
      else if (c >= 'a' && c <= 'f') v |= uint32_t(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= uint32_t(c - 'A' + 10);
      else fail("bad \\u escape");
    }
    return v;
  }
  std::string parseString() {
    if (p_ >= end_ || *p_ != '"') fail("expected a string");
    p_++;
    std::string out;
    for (;;) {
      if (p_ >= end_) fail("unterminated string");
      char c = *p_++;
      if (c == '"') break;
      if (c != '\\') { out.push_back(c); continue; }
      if (p_ >= end_) fail("unterminated escape");
      char e = *p_++;
      switch (e) {
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case '/': out.push_back('/'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u': {
          uint32_t cp = hex4();
          if (cp >= 0xD800 && cp < 0xDC00 && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
            p_ += 2;
            uint32_t lo = hex4();
            if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          appendUtf8(out, cp);
          break;
        }
        default: fail("bad escape");
      }
    }
    return out;
  }
  Value parseValue(int depth) {
    if (depth > 256) fail("nesting too deep");
    skipWs();
    if (p_ >= end_) fail("unexpected end of input");
    Value v;
    char c = *p_;
    if (c == '{') {
      p_++;
      v.kind = Value::Object;
      skipWs();
      if (p_ < end_ && *p_ == '}') { p_++; return v; }
      for (;;) {
        skipWs();
        std::string key = parseString();
        skipWs();
        if (p_ >= end_ || *p_ != ':') fail("expected ':'");
        p_++;
        v.obj.emplace_back(std::move(key), parseValue(depth + 1));
        skipWs();
        if (p_ < end_ && *p_ == ',') { p_++; continue; }
        if (p_ < end_ && *p_ == '}') { p_++; break; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      p_++;
      v.kind = Value::Array;
      skipWs();
      if (p_ < end_ && *p_ == ']') { p_++; return v; }
      for (;;) {
        v.arr.push_back(parseValue(depth + 1));
        skipWs();
        if (p_ < end_ && *p_ == ',') { p_++; continue; }
        if (p_ < end_ && *p_ == ']') { p_++; break; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"') {
      v.kind = Value::String;
      v.str = parseString();
    } else if (literal("true")) { v.kind = Value::Bool; v.b = true; }
    else if (literal("false")) { v.kind = Value::Bool; v.b = false; }
    else if (literal("null")) { v.kind = Value::Null; }
    else if (c == '-' || (c >= '0' && c <= '9')) {
      const char* s = p_;
      if (*p_ == '-') p_++;
      while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '+' || *p_ == '-')) p_++;
      std::string tok(s, size_t(p_ - s));
      char* e = nullptr;
      v.kind = Value::Number;
      v.num = std::strtod(tok.c_str(), &e);
      if (e == tok.c_str() || *e != '\0') fail("bad number");
    } else fail("unexpected character");
    return v;
  }
};

inline Value parse(const char* data, size_t len) { return Parser(data, len).parseDocument(); }

}  // namespace json
}  // namespace yart_hip
