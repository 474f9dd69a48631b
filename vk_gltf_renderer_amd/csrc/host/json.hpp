// Minimal JSON DOM (RFC 8259) for the glTF front end. The reference parses glTF through tinygltf + nlohmann::json
// (third-party, not vendored in /root/reference); this is a from-scratch reader sized for glTF documents.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mijson {

struct Value
{
  enum Type { Null, Bool, Number, String, Array, Object };
  Type                         type = Null;
  bool                         b    = false;
  double                       num  = 0.0;
  std::string                  str;
  std::vector<Value>           arr;
  std::vector<std::pair<std::string, Value>> obj;  // keeps document order

  bool isNull() const { return type == Null; }
  bool isObject() const { return type == Object; }
  bool isArray() const { return type == Array; }
  bool isNumber() const { return type == Number; }
  bool isString() const { return type == String; }

  const Value* find(const std::string& key) const
  {
    if(type != Object)
      return nullptr;
    for(const auto& kv : obj)
      if(kv.first == key)
        return &kv.second;
    return nullptr;
  }
  bool has(const std::string& key) const { return find(key) != nullptr; }
  const Value& operator[](const std::string& key) const
  {
    static const Value nullValue;
    const Value*       v = find(key);
    return v ? *v : nullValue;
  }
  const Value& operator[](size_t i) const
  {
    static const Value nullValue;
    return (type == Array && i < arr.size()) ? arr[i] : nullValue;
  }
  size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }

  double      number(double def = 0.0) const { return type == Number ? num : def; }
  // (out-of-range / non-finite numbers fall back to the default: converting them would be undefined behaviour)
  int         integer(int def = 0) const { return (type == Number && num > -2147483648.0 && num < 2147483647.0) ? int(std::llround(num)) : def; }
  bool        boolean(bool def = false) const { return type == Bool ? b : def; }
  std::string string(const std::string& def = "") const { return type == String ? str : def; }
};

class Parser
{
public:
  explicit Parser(const char* begin, const char* end)
      : p(begin)
      , e(end)
  {
  }
  Value parse()
  {
    Value v = value();
    ws();
    if(p != e)
      fail("trailing characters");
    return v;
  }

private:
  const char* p;
  const char* e;
  [[noreturn]] void fail(const char* msg) { throw std::runtime_error(std::string("JSON: ") + msg); }
  void ws()
  {
    while(p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r'))
      ++p;
  }
  Value value()
  {
    ws();
    if(p >= e)
      fail("unexpected end");
    switch(*p)
    {
      case '{':
        return object();
      case '[':
        return array();
      case '"': {
        Value v;
        v.type = Value::String;
        v.str  = string();
        return v;
      }
      case 't':
        return literal("true", true);
      case 'f':
        return literal("false", false);
      case 'n': {
        expect("null");
        return Value();
      }
      default:
        return number();
    }
  }
  void expect(const char* lit)
  {
    size_t n = strlen(lit);
    if(size_t(e - p) < n || strncmp(p, lit, n) != 0)
      fail("bad literal");
    p += n;
  }
  Value literal(const char* lit, bool b)
  {
    expect(lit);
    Value v;
    v.type = Value::Bool;
    v.b    = b;
    return v;
  }
  Value number()
  {
    char*  endp = nullptr;
    double d    = strtod(p, &endp);
    if(endp == p)
      fail("bad number");
    p = endp;
    Value v;
    v.type = Value::Number;
    v.num  = d;
    return v;
  }
  static void appendUtf8(std::string& s, uint32_t cp)
  {
    if(cp < 0x80)
      s += char(cp);
    else if(cp < 0x800)
    {
      s += char(0xC0 | (cp >> 6));
      s += char(0x80 | (cp & 0x3F));
    }
    else if(cp < 0x10000)
    {
      s += char(0xE0 | (cp >> 12));
      s += char(0x80 | ((cp >> 6) & 0x3F));
      s += char(0x80 | (cp & 0x3F));
    }
    else
    {
      s += char(0xF0 | (cp >> 18));
      s += char(0x80 | ((cp >> 12) & 0x3F));
      s += char(0x80 | ((cp >> 6) & 0x3F));
      s += char(0x80 | (cp & 0x3F));
    }
  }
  uint32_t hex4()
  {
    if(e - p < 4)
      fail("bad \\u escape");
    uint32_t v = 0;
    for(int i = 0; i < 4; ++i)
    {
      char c = *p++;
      v <<= 4;
      if(c >= '0' && c <= '9')
        v |= uint32_t(c - '0');
      else if(c >= 'a' && c <= 'f')
        v |= uint32_t(c - 'a' + 10);
      else if(c >= 'A' && c <= 'F')
        v |= uint32_t(c - 'A' + 10);
      else
        fail("bad hex digit");
    }
    return v;
  }
  std::string string()
  {
    ++p;  // opening quote
    std::string s;
    while(true)
    {
      if(p >= e)
        fail("unterminated string");
      char c = *p++;
      if(c == '"')
        break;
      if(c == '\\')
      {
        if(p >= e)
          fail("bad escape");
        char esc = *p++;
        switch(esc)
        {
          case '"': s += '"'; break;
          case '\\': s += '\\'; break;
          case '/': s += '/'; break;
          case 'b': s += '\b'; break;
          case 'f': s += '\f'; break;
          case 'n': s += '\n'; break;
          case 'r': s += '\r'; break;
          case 't': s += '\t'; break;
          case 'u': {
            uint32_t cp = hex4();
            if(cp >= 0xD800 && cp <= 0xDBFF && e - p >= 6 && p[0] == '\\' && p[1] == 'u')
            {
              p += 2;
              uint32_t lo = hex4();
              cp          = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
            }
            appendUtf8(s, cp);
            break;
          }
          default: fail("unknown escape");
        }
      }
      else
        s += c;
    }
    return s;
  }
  Value array()
  {
    ++p;
    Value v;
    v.type = Value::Array;
    ws();
    if(p < e && *p == ']')
    {
      ++p;
      return v;
    }
    while(true)
    {
      v.arr.push_back(value());
      ws();
      if(p >= e)
        fail("unterminated array");
      if(*p == ',')
      {
        ++p;
        continue;
      }
      if(*p == ']')
      {
        ++p;
        break;
      }
      fail("expected , or ]");
    }
    return v;
  }
  Value object()
  {
    ++p;
    Value v;
    v.type = Value::Object;
    ws();
    if(p < e && *p == '}')
    {
      ++p;
      return v;
    }
    while(true)
    {
      ws();
      if(p >= e || *p != '"')
        fail("expected key");
      std::string k = string();
      ws();
      if(p >= e || *p != ':')
        fail("expected :");
      ++p;
      v.obj.emplace_back(std::move(k), value());
      ws();
      if(p >= e)
        fail("unterminated object");
      if(*p == ',')
      {
        ++p;
        continue;
      }
      if(*p == '}')
      {
        ++p;
        break;
      }
      fail("expected , or }");
    }
    return v;
  }
};

inline Value parse(const std::string& text)
{
  return Parser(text.data(), text.data() + text.size()).parse();
}

}  // namespace mijson
