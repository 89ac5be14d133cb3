// Binding reader for Opt ".t" energy files.
//
// A .t file is a Lua program (reference API/src/o.t:840-853 runs it in a sandbox built by lib.t); this
// backend does not evaluate it.  It reads the DECLARATION calls that define the C-API contract -- which
// slot of `problemparams` holds what, with which type, over which index space (reference o.t:320-324
// Dim, :946-958 Image/Unknown, :1055-1058 Param, :1043-1053 Graph, lib.t:33 UsePreconditioner) -- and
// checks them against the binding layout of the registered kernel set for that energy.  Declarations
// produced by Lua control flow (e.g. shape_from_shading.t's `for i=1,9 do Param("L_"..i, ...)`) are not
// literal and are taken from the registry.
#pragma once
#include <string>
#include <vector>

namespace optamd {

struct TDecl {
    enum Kind { kDim, kUnknown, kArray, kParam, kGraph } kind;
    std::string name;
    std::string type;               // as written; "" if omitted (-> opt_float)
    int index = -1;                 // binding / dimension index
    std::vector<std::string> dims;  // index-space names
    struct GraphSlot { std::string name; std::vector<std::string> dims; int index; };
    std::vector<GraphSlot> slots;   // kGraph only
};

struct TFile {
    std::string path, stem;
    std::vector<TDecl> decls;
    bool hasUsePreconditioner = false, usePreconditioner = false;
    bool hasExclude = false;
    int energyCalls = 0;
    unsigned long bodyHash = 0;     // FNV-1a of the comment-/whitespace-stripped text
};

// Returns false (with `err` set) if the file cannot be read.
bool readTFile(const std::string& path, TFile& out, std::string& err);

}  // namespace optamd
