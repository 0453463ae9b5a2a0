// tools/suffix_array.cpp -- the `suffix-array <file>` driver of the reference
// (/root/reference/src/main.rs:8-15: read a file, SuffixTable::new, print "Suffixes: N"),
// over the MI355X engine's C++ host mirror (include/suffix_table.hpp -> libsuffix_hip.so),
// extended into the large-file driver SURVEY.md 8(f) asks for:
//
//   suffix-array FILE [--lcp] [--dump PREFIX] [--load PREFIX] [--query Q]... [--time]
//
//   --dump PREFIX   write PREFIX.sa (and PREFIX.lcp with --lcp) as raw little-endian u32
//                   arrays -- the on-disk form SuffixTable::from_parts (:111-119) reloads
//   --load PREFIX   skip construction: from_parts(text, PREFIX.sa)
//   --query Q       positions(Q) (:223-259): prints count and the first few positions
//   --time          wall-clock milliseconds of construction / LCP (host pointers, i.e.
//                   including the PCIe copies: the device-resident rate is bench.py's)
//
// Exit status: 0 ok, 1 usage / IO error, 2 engine error (message on stderr; the
// reference panics in those places, :380 / :117).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "suffix_table.hpp"

static bool read_file(const std::string& path, std::string* out)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const std::streamsize n = f.tellg();
    f.seekg(0);
    out->resize((size_t)n);
    return n == 0 || (bool)f.read(&(*out)[0], n);
}
static bool write_u32(const std::string& path, const std::vector<uint32_t>& v)
{
    std::ofstream f(path, std::ios::binary);
    return f && (v.empty() || f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * 4)));
}
static bool read_u32(const std::string& path, std::vector<uint32_t>* v)
{
    std::string raw;
    if (!read_file(path, &raw) || raw.size() % 4) return false;
    v->resize(raw.size() / 4);
    if (!raw.empty()) memcpy(v->data(), raw.data(), raw.size());
    return true;
}
static double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char** argv)
{
    std::string file, dump, load;
    std::vector<std::string> queries;
    bool want_lcp = false, timing = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](const char* opt) -> const char* {
            if (i + 1 >= argc) { fprintf(stderr, "%s needs an argument\n", opt); exit(1); }
            return argv[++i];
        };
        if (a == "--lcp") want_lcp = true;
        else if (a == "--time") timing = true;
        else if (a == "--dump") dump = need("--dump");
        else if (a == "--load") load = need("--load");
        else if (a == "--query") queries.push_back(need("--query"));
        else if (!a.empty() && a[0] == '-') { fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
        else file = a;
    }
    if (file.empty()) {
        fprintf(stderr, "usage: suffix-array FILE [--lcp] [--dump PREFIX] [--load PREFIX] [--query Q]... [--time]\n");
        return 1;
    }
    std::string text;
    if (!read_file(file, &text)) { fprintf(stderr, "cannot read %s\n", file.c_str()); return 1; }
    try {
        auto t0 = std::chrono::steady_clock::now();
        suffix::SuffixTable st = [&] {
            if (load.empty()) return suffix::SuffixTable::new_(std::move(text));
            std::vector<uint32_t> sa;
            if (!read_u32(load + ".sa", &sa)) { fprintf(stderr, "cannot read %s.sa\n", load.c_str()); exit(1); }
            return suffix::SuffixTable::from_parts(std::move(text), std::move(sa));     // :111-119
        }();
        const double t_sa = ms_since(t0);
        std::cout << "Suffixes: " << st.len() << "\n";                                      // src/main.rs:14
        if (timing) std::cout << (load.empty() ? "construction" : "load") << " ms: " << t_sa << "\n";
        std::vector<uint32_t> lcp;
        if (want_lcp) {
            t0 = std::chrono::steady_clock::now();
            lcp = st.lcp_lens();
            if (timing) std::cout << "lcp ms: " << ms_since(t0) << "\n";
            uint64_t sum = 0;
            uint32_t mx = 0;
            for (uint32_t v : lcp) { sum += v; if (v > mx) mx = v; }
            std::cout << "LCP: max " << mx << " mean " << (lcp.empty() ? 0.0 : (double)sum / (double)lcp.size()) << "\n";
        }
        if (!dump.empty()) {
            if (!write_u32(dump + ".sa", st.table()) || (want_lcp && !write_u32(dump + ".lcp", lcp))) {
                fprintf(stderr, "cannot write %s.*\n", dump.c_str());
                return 1;
            }
        }
        if (!queries.empty()) {
            std::vector<std::string_view> qs(queries.begin(), queries.end());
            auto se = st.positions_batch(qs);
            for (size_t k = 0; k < qs.size(); k++) {
                const uint32_t s = se[k].first, e = se[k].second;
                std::cout << "positions(\"" << queries[k] << "\"): " << (e - s);
                for (uint32_t r = s; r < e && r < s + 8; r++) std::cout << (r == s ? " [" : ", ") << st.table()[r];
                if (e > s) std::cout << (e - s > 8 ? ", ...]" : "]");
                std::cout << "\n";
            }
        }
    } catch (const std::exception& ex) {
        fprintf(stderr, "suffix-array: %s\n", ex.what());
        return 2;
    }
    return 0;
}
