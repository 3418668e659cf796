// Args.h -- minimal command-line parser with the flag conventions of the reference tools
// (short "-x 1920" and long "--width 1920" value flags, switches, two positional file names).
#ifndef VC2HOST_ARGS_H
#define VC2HOST_ARGS_H
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

struct ArgSpec { char s; const char *l; bool takes_value; const char *help; };

class Args {
 public:
  Args(const std::vector<ArgSpec> &specs, int argc, char **argv) {
    for (int i = 1; i < argc; ++i) {
      const std::string a = argv[i];
      if (a == "-" || a.empty() || a[0] != '-') { positional.push_back(a); continue; }
      const ArgSpec *sp = nullptr;
      std::string inline_val;
      bool has_inline = false;
      if (a.size() > 2 && a[1] == '-') {
        std::string name = a.substr(2);
        const std::size_t eq = name.find('=');
        if (eq != std::string::npos) { inline_val = name.substr(eq + 1); name = name.substr(0, eq); has_inline = true; }
        for (const ArgSpec &s : specs) if (name == s.l) sp = &s;
      } else if (a.size() == 2) {
        for (const ArgSpec &s : specs) if (a[1] == s.s) sp = &s;
      }
      if (!sp) throw std::invalid_argument("Couldn't find match for argument for arg " + a);
      if (sp->takes_value) {
        if (!has_inline) {
          if (i + 1 >= argc) throw std::invalid_argument(std::string("Missing a value for this argument! for arg --") + sp->l);
          inline_val = argv[++i];
        }
        values[sp->l] = inline_val;
      } else {
        values[sp->l] = "1";
      }
    }
  }
  bool isSet(const char *l) const { return values.count(l) != 0; }
  std::string get(const char *l, const std::string &def = "") const { auto it = values.find(l); return it == values.end() ? def : it->second; }
  int getInt(const char *l, int def) const {
    if (!isSet(l)) return def;
    try { std::size_t p; const int v = std::stoi(get(l), &p); if (p != get(l).size()) throw 0; return v; }
    catch (...) { throw std::invalid_argument(std::string("Couldn't read argument value from string '") + get(l) + "' for arg --" + l); }
  }
  void require(const char *l) const { if (!isSet(l)) throw std::invalid_argument(std::string("Required argument missing: ") + l); }
  std::vector<std::string> positional;

 private:
  std::map<std::string, std::string> values;
};
#endif
