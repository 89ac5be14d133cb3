#include "tfile.h"
#include <cctype>
#include <fstream>
#include <regex>
#include <sstream>

namespace optamd {
namespace {

std::string stripComments(const std::string& s) {   // Lua: "--" to end of line, "--[[ ... ]]" blocks
    std::string o; o.reserve(s.size());
    size_t i = 0;
    bool inStr = false;
    while (i < s.size()) {
        if (!inStr && s.compare(i, 4, "--[[") == 0) {
            size_t e = s.find("]]", i + 4);
            i = (e == std::string::npos) ? s.size() : e + 2;
            continue;
        }
        if (!inStr && s.compare(i, 2, "--") == 0) {
            while (i < s.size() && s[i] != '\n') ++i;
            continue;
        }
        if (s[i] == '"') inStr = !inStr;
        o.push_back(s[i++]);
    }
    return o;
}

std::vector<std::string> splitNames(const std::string& s) {
    std::vector<std::string> out; std::string cur;
    for (char c : s) {
        if (std::isalnum((unsigned char)c) || c == '_') cur.push_back(c);
        else if (!cur.empty()) { out.push_back(cur); cur.clear(); }
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

}  // namespace

bool readTFile(const std::string& path, TFile& out, std::string& err) {
    std::ifstream f(path);
    if (!f.good()) { err = "cannot open problem specification '" + path + "'"; return false; }
    std::stringstream ss; ss << f.rdbuf();
    std::string text = stripComments(ss.str());
    if (text.size() >= 3 && (unsigned char)text[0] == 0xEF && (unsigned char)text[1] == 0xBB && (unsigned char)text[2] == 0xBF) text = text.substr(3);
    out.path = path;
    size_t slash = path.find_last_of("/\\");
    std::string base = (slash == std::string::npos) ? path : path.substr(slash + 1);
    size_t dot = base.find_last_of('.');
    out.stem = (dot == std::string::npos) ? base : base.substr(0, dot);

    unsigned long h = 1469598103934665603ul;
    for (char c : text) if (!std::isspace((unsigned char)c)) { h ^= (unsigned char)c; h *= 1099511628211ul; }
    out.bodyHash = h;

    using std::regex; using std::sregex_iterator;
    {   // Dim("W",0) / opt.Dim("N",0)
        regex re(R"re((?:opt\.)?Dim\s*\(\s*"(\w+)"\s*,\s*(\d+)\s*\))re");
        for (sregex_iterator it(text.begin(), text.end(), re), e; it != e; ++it) {
            TDecl d; d.kind = TDecl::kDim; d.name = (*it)[1]; d.index = std::stoi((*it)[2]); out.decls.push_back(d);
        }
    }
    {   // Unknown("Offset",opt_float2,{W,H},0) / Array(...) / Image(...) ; the type may be omitted (o.t:946-949)
        regex re(R"re(\b(Unknown|Array|Image)\s*\(\s*"(\w+)"\s*,\s*(?:(\w+)\s*,\s*)?\{([^}]*)\}\s*,\s*(\d+)\s*\))re");
        for (sregex_iterator it(text.begin(), text.end(), re), e; it != e; ++it) {
            TDecl d; d.kind = ((*it)[1] == "Unknown") ? TDecl::kUnknown : TDecl::kArray;
            d.name = (*it)[2]; d.type = (*it)[3]; d.dims = splitNames((*it)[4]); d.index = std::stoi((*it)[5]);
            out.decls.push_back(d);
        }
    }
    {   // Param("w_fitSqrt", float, 5)  -- literal index only
        regex re(R"re(\bParam\s*\(\s*"(\w+)"\s*,\s*(\w+)\s*,\s*(\d+)\s*\))re");
        for (sregex_iterator it(text.begin(), text.end(), re), e; it != e; ++it) {
            TDecl d; d.kind = TDecl::kParam; d.name = (*it)[1]; d.type = (*it)[2]; d.index = std::stoi((*it)[3]); out.decls.push_back(d);
        }
    }
    {   // Graph("G", 6, "v0", {N}, 7, "v1", {N}, 8)
        regex re(R"re(\bGraph\s*\(\s*"(\w+)"\s*,\s*(\d+)\s*((?:,\s*"\w+"\s*,\s*\{[^}]*\}\s*,\s*\d+\s*)+)\))re");
        regex slot(R"re("(\w+)"\s*,\s*\{([^}]*)\}\s*,\s*(\d+))re");
        for (sregex_iterator it(text.begin(), text.end(), re), e; it != e; ++it) {
            TDecl d; d.kind = TDecl::kGraph; d.name = (*it)[1]; d.index = std::stoi((*it)[2]);
            std::string rest = (*it)[3];
            for (sregex_iterator jt(rest.begin(), rest.end(), slot), je; jt != je; ++jt)
                d.slots.push_back({(*jt)[1], splitNames((*jt)[2]), std::stoi((*jt)[3])});
            out.decls.push_back(d);
        }
    }
    {
        regex re(R"re(\bUsePreconditioner\s*\(\s*(true|false)\s*\))re");
        std::smatch m;
        if (std::regex_search(text, m, re)) { out.hasUsePreconditioner = true; out.usePreconditioner = (m[1] == "true"); }
    }
    out.hasExclude = std::regex_search(text, regex(R"re(\bExclude\s*\()re"));
    {
        regex re(R"re(\bEnergy\s*\()re");
        out.energyCalls = (int)std::distance(sregex_iterator(text.begin(), text.end(), re), sregex_iterator());
    }
    return true;
}

}  // namespace optamd
