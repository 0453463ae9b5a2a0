// C++ host-mirror test, mirroring /root/reference/tests/tests.rs through
// include/suffix_table.hpp.  Compiled on CPU by `pytest -m "not gpu"` (build +
// link check only); executed on the MI355X by the gpu-marked test.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "suffix_table.hpp"

using suffix::SuffixTable;

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

static std::vector<uint32_t> naive(const std::string& t)        // naive_table, src/table.rs:367-376
{
    std::vector<uint32_t> sa(t.size());
    for (size_t i = 0; i < t.size(); i++) sa[i] = (uint32_t)i;
    std::sort(sa.begin(), sa.end(), [&](uint32_t a, uint32_t b) { return t.compare(a, std::string::npos, t, b, std::string::npos) < 0; });
    return sa;
}
static std::vector<uint32_t> pos(const SuffixTable& st, const char* q)
{
    auto r = st.positions(q);
    return std::vector<uint32_t>(r.first, r.second);
}

int main()
{
    // tests.rs:22-70
    for (const char* s : {"apple", "banana", "mississippi", "tgtgtgtgcaccg", "", "a", "ab", "aa", "\xe2\x98\x83" "abc" "\xe2\x98\x83"}) {
        SuffixTable st = SuffixTable::new_(s);
        EXPECT(st.table() == naive(s));
        EXPECT(st.len() == std::string(s).size());
    }
    {   // nul_is_ok
        std::string z(1, '\0');
        EXPECT(SuffixTable::new_(z).table() == naive(z));
    }
    // tests.rs:100-168
    EXPECT(pos(SuffixTable::new_(""), "a").empty() && !SuffixTable::new_("").contains("a"));
    EXPECT(pos(SuffixTable::new_("a"), "").empty() && !SuffixTable::new_("a").contains(""));
    EXPECT((pos(SuffixTable::new_("a"), "a") == std::vector<uint32_t>{0}));
    EXPECT((pos(SuffixTable::new_("aa"), "a") == std::vector<uint32_t>{1, 0}));
    EXPECT((pos(SuffixTable::new_("zzzzzaazzzzz"), "a") == std::vector<uint32_t>{5, 6}));
    EXPECT((pos(SuffixTable::new_("zzzzabczzzzzabczzzzzz"), "abc") == std::vector<uint32_t>{4, 12}));
    EXPECT(pos(SuffixTable::new_("az"), "mnomnomnomnomnomnomno").empty());
    // tests.rs:202-213 and the doc-tests
    SuffixTable q = SuffixTable::new_("The quick brown fox was very quick.");
    EXPECT((pos(q, "quick") == std::vector<uint32_t>{4, 29}));
    EXPECT(q.contains("quick") && !q.contains("faux"));
    auto ap = q.any_position("quick");
    EXPECT(ap && (*ap == 4 || *ap == 29));
    EXPECT((pos(SuffixTable::new_("\xe2\x98\x83" "abc" "\xe2\x98\x83"), "\xe2\x98\x83") == std::vector<uint32_t>{6, 0}));
    // tests.rs:18-20: sais(text) == naive(text), through the mirror's own new_naive (:93-100)
    for (const char* lit : {"banana", "mississippi", "aaaaaa", "abracadabra", "\xe2\x98\x83" "abc" "\xe2\x98\x83"})
        EXPECT(SuffixTable::new_(lit) == SuffixTable::new_naive(lit));
    // tests.rs:170-179 parts()
    SuffixTable a = SuffixTable::new_("po\xc3\xabzie");
    SuffixTable b = a;
    auto parts = std::move(b).into_parts();
    SuffixTable c = SuffixTable::from_parts(parts.first, parts.second);
    EXPECT(a == c);
    // lcp_lens, SURVEY.md 8c literal
    EXPECT((SuffixTable::new_("banana").lcp_lens() == std::vector<uint32_t>{0, 1, 3, 0, 0, 2}));
    // positions(&self) is lock-free and Sync in the reference: many threads, one table, first query races for
    // the lazily created device index (GPU only: the emulator runs one workgroup at a time on one OS thread)
    if (std::getenv("SFX_CPP_THREADS")) {
        std::string big;
        for (int i = 0; i < 20000; i++) big += "the quick brown fox " + std::to_string(i * 7919 % 1000) + " ";
        SuffixTable shared = SuffixTable::new_(big);
        const std::vector<uint32_t> want = pos(shared, "fox 7 ");
        SuffixTable fresh = SuffixTable::from_parts(shared.text(), shared.table());     // index not created yet
        std::vector<std::thread> th;
        std::vector<int> ok(16, 0);
        for (int t = 0; t < 16; t++)
            th.emplace_back([&, t] {
                bool good = true;
                for (int r = 0; r < 20; r++) good = good && pos(fresh, "fox 7 ") == want && fresh.contains("quick");
                ok[t] = good;
            });
        for (auto& x : th) x.join();
        for (int t = 0; t < 16; t++) EXPECT(ok[t]);
        // concurrent SuffixTable::new from several threads (per-thread streams in the host entry points)
        std::vector<std::thread> nb;
        std::vector<int> ok2(8, 0);
        for (int t = 0; t < 8; t++)
            nb.emplace_back([&, t] { ok2[t] = SuffixTable::new_(big).table() == shared.table(); });
        for (auto& x : nb) x.join();
        for (int t = 0; t < 8; t++) EXPECT(ok2[t]);
    }
    std::printf(failures ? "FAILED %d\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
