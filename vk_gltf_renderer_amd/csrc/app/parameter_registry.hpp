// Minimal command-line registry in the spirit of nvutils::ParameterRegistry / ParameterParser (external to the reference
// tree; call sites src/main.cpp:83-130): `--name value...` options bound to variables, unknown options are an error.
#pragma once
#include <cstdlib>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

class ParameterRegistry
{
public:
  struct Param
  {
    std::string                                        help;
    int                                                arity;
    std::function<void(const std::vector<std::string>&)> set;
  };
  void add(const std::string& name, const std::string& help, int* v)
  {
    m_params[name] = {help, 1, [v](const std::vector<std::string>& a) { *v = std::atoi(a[0].c_str()); }};
  }
  void add(const std::string& name, const std::string& help, float* v)
  {
    m_params[name] = {help, 1, [v](const std::vector<std::string>& a) { *v = float(std::atof(a[0].c_str())); }};
  }
  void add(const std::string& name, const std::string& help, bool* v, bool flagOnly = false)
  {
    if(flagOnly)
      m_params[name] = {help, 0, [v](const std::vector<std::string>&) { *v = true; }};
    else
      m_params[name] = {help, 1, [v](const std::vector<std::string>& a) { *v = std::atoi(a[0].c_str()) != 0 || a[0] == "true"; }};
  }
  void add(const std::string& name, const std::string& help, std::string* v)
  {
    m_params[name] = {help, 1, [v](const std::vector<std::string>& a) { *v = a[0]; }};
  }
  void addVec2(const std::string& name, const std::string& help, int* v)
  {
    m_params[name] = {help, 2, [v](const std::vector<std::string>& a) {
                        v[0] = std::atoi(a[0].c_str());
                        v[1] = std::atoi(a[1].c_str());
                      }};
  }
  void addCallback(const std::string& name, const std::string& help, int arity, std::function<void(const std::vector<std::string>&)> f)
  {
    m_params[name] = {help, arity, std::move(f)};
  }
  // Parses argv; positional arguments are returned.
  std::vector<std::string> parse(int argc, char** argv) { return parseTokens(std::vector<std::string>(argv + (argc > 0 ? 1 : 0), argv + argc)); }
  // The same for a token list (a SEQUENCE block of a benchmark script).
  std::vector<std::string> parseTokens(const std::vector<std::string>& tokens)
  {
    std::vector<std::string> positional;
    for(size_t i = 0; i < tokens.size(); ++i)
    {
      const std::string& a = tokens[i];
      if(a.rfind("--", 0) != 0)
      {
        positional.push_back(a);
        continue;
      }
      auto it = m_params.find(a.substr(2));
      if(it == m_params.end())
        throw std::runtime_error("unknown option " + a);
      std::vector<std::string> args;
      for(int k = 0; k < it->second.arity; ++k)
      {
        if(i + 1 >= tokens.size())
          throw std::runtime_error("option " + a + " needs " + std::to_string(it->second.arity) + " value(s)");
        args.push_back(tokens[++i]);
      }
      it->second.set(args);
    }
    return positional;
  }
  std::string usage() const
  {
    std::string s;
    for(const auto& kv : m_params)
      s += "  --" + kv.first + (kv.second.arity ? " <" + std::to_string(kv.second.arity) + ">" : "") + "   " + kv.second.help + "\n";
    return s;
  }

private:
  std::map<std::string, Param> m_params;
};
