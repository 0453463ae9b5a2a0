// include/suffix_table.hpp -- C++ host-side mirror of the reference's public
// type `SuffixTable` (/root/reference/src/table.rs:54-294) on top of the C ABI
// (suffix_hip.h).  Header-only; link with -lsuffix_hip.  The reference is
// compiled code (Rust) and this image has no Rust toolchain, so this is the
// compiled-language host side; rust/suffix-hip/src/lib.rs shows the Rust binding.
//
// Same names, argument meaning and error behaviour as the Rust API:
//   new_ / new_naive (doc-hidden upstream, :93-100: the definition, sorted on the host -- the reference's own test oracle,
//   tests/tests.rs:18-20; never a fallback of new_) / from_parts / into_parts / lcp_lens / table / text / len / is_empty /
//   suffix / suffix_bytes / contains / positions / any_position,
// plus the additive positions_batch / contains_batch.  Errors that are panics
// in the reference (assert! :380, assert_eq! :117) are std::runtime_error /
// std::length_error here.  Text is indexed by BYTES (:379).
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "suffix_hip.h"

namespace suffix {

class SuffixTable {
public:
    // SuffixTable::new (:78-85): builds the table on the GPU.
    static SuffixTable new_(std::string text)
    {
        SuffixTable st;
        st.text_ = std::move(text);
        st.table_.assign(st.text_.size(), 0u);                       // vec![0u32; n] (:381)
        check(sfx_build_sa_u32(bytes(st.text_), st.text_.size(), st.table_.data()), "SuffixTable::new");
        return st;
    }
    // SuffixTable::new_naive (:93-100, #[doc(hidden)]) -> naive_table (:367-376): the definition -- all byte suffixes sorted by
    // comparison on the host, O(n^2 log n).  Upstream keeps it as the oracle of its own tests (tests/tests.rs:18-20); it is the same
    // here: what a caller compares new_() against, never something new_() falls back to.
    static SuffixTable new_naive(std::string text)
    {
        SuffixTable st;
        st.text_ = std::move(text);
        if (st.text_.size() > 0xFFFFFFFFull) throw std::length_error("SuffixTable::new_naive: text longer than u32::MAX");
        st.table_.resize(st.text_.size());
        for (size_t i = 0; i < st.table_.size(); i++) st.table_[i] = (uint32_t)i;
        const std::string_view t(st.text_);
        std::sort(st.table_.begin(), st.table_.end(), [t](uint32_t a, uint32_t b) { return t.substr(a) < t.substr(b); });
        return st;
    }
    // SuffixTable::from_parts (:111-119): unchecked except for the lengths.
    static SuffixTable from_parts(std::string text, std::vector<uint32_t> table)
    {
        if (text.size() != table.size()) throw std::length_error("text.len() != table.len()");   // :117
        SuffixTable st;
        st.text_ = std::move(text);
        st.table_ = std::move(table);
        return st;
    }
    std::pair<std::string, std::vector<uint32_t>> into_parts() && { return {std::move(text_), std::move(table_)}; }

    SuffixTable(SuffixTable&& o) noexcept { *this = std::move(o); }
    SuffixTable& operator=(SuffixTable&& o) noexcept
    {
        text_ = std::move(o.text_);
        table_ = std::move(o.table_);
        lazy_ = std::move(o.lazy_);
        o.lazy_ = std::make_unique<LazyIndex>();
        return *this;
    }
    SuffixTable(const SuffixTable& o) : text_(o.text_), table_(o.table_) {}      // derive(Clone)
    ~SuffixTable() = default;
    bool operator==(const SuffixTable& o) const { return text_ == o.text_ && table_ == o.table_; }   // :54

    // lcp_lens (:130-138)
    std::vector<uint32_t> lcp_lens() const
    {
        std::vector<uint32_t> lcp(table_.size(), 0u);
        check(sfx_build_lcp_u32(bytes(text_), text_.size(), table_.data(), lcp.data()), "lcp_lens");
        return lcp;
    }
    const std::vector<uint32_t>& table() const { return table_; }     // :142
    const std::string& text() const { return text_; }                 // :148
    size_t len() const { return table_.size(); }                      // :156
    bool is_empty() const { return table_.empty(); }                  // :162
    std::string_view suffix(size_t i) const { return std::string_view(text_).substr(table_.at(i)); }   // :168
    std::string_view suffix_bytes(size_t i) const { return suffix(i); }                                  // :174

    // positions (:223-259): the occurrences of `query`, in suffix-array order,
    // as a view into table().
    std::pair<const uint32_t*, const uint32_t*> positions(std::string_view query) const
    {
        if (text_.empty() || query.empty()) return {table_.data(), table_.data()};   // :228-229
        auto se = positions_batch({query});
        return {table_.data() + se[0].first, table_.data() + se[0].second};
    }
    // any_position (:279-293); which occurrence is arbitrary by contract (:261-262)
    std::optional<uint32_t> any_position(std::string_view query) const
    {
        if (query.empty() || text_.empty()) return std::nullopt;
        uint64_t off[2] = {0, query.size()};
        uint8_t found = 0;
        uint32_t pos = 0;
        check(sfx_contains_batch(index(), reinterpret_cast<const uint8_t*>(query.data()), off, 1, &found, &pos),
              "any_position");
        if (!found) return std::nullopt;
        return pos;
    }
    bool contains(std::string_view query) const { return any_position(query).has_value(); }   // :197-199

    // additive: many queries in one launch; (start, end) index pairs into table()
    std::vector<std::pair<uint32_t, uint32_t>> positions_batch(const std::vector<std::string_view>& qs) const
    {
        std::vector<uint64_t> off(qs.size() + 1, 0);
        std::string blob;
        for (size_t k = 0; k < qs.size(); k++) { blob.append(qs[k]); off[k + 1] = blob.size(); }
        std::vector<uint32_t> s(qs.size()), e(qs.size());
        if (!qs.empty())
            check(sfx_positions_batch(index(), reinterpret_cast<const uint8_t*>(blob.data()), off.data(), qs.size(),
                                      s.data(), e.data()), "positions_batch");
        std::vector<std::pair<uint32_t, uint32_t>> out(qs.size());
        for (size_t k = 0; k < qs.size(); k++) out[k] = {s[k], e[k]};
        return out;
    }

private:
    SuffixTable() = default;
    static const uint8_t* bytes(const std::string& s) { return reinterpret_cast<const uint8_t*>(s.data()); }
    static void check(int status, const char* what)
    {
        if (status == SFX_OK) return;
        std::string msg = std::string(what) + ": " + sfx_strerror(status) + " " + sfx_last_hip_error();
        if (status == SFX_ERR_TOO_LARGE) throw std::length_error(msg);               // assert! at :380
        throw std::runtime_error(msg);
    }
    // The device-resident index is made on the first query.  positions(&self) is lock-free and callable
    // from many threads in the reference, so the lazy creation must be race-free: std::call_once; a
    // creation that throws leaves the flag unset and the next caller tries again.  Queries on the finished
    // index only read it (the C ABI's *_batch calls are safe to run concurrently on one sfx_index).
    struct LazyIndex {
        std::once_flag once;
        sfx_index* ix = nullptr;
        ~LazyIndex() { if (ix) sfx_index_destroy(ix); }
    };
    sfx_index* index() const
    {
        LazyIndex& l = *lazy_;
        std::call_once(l.once, [&] {
            check(sfx_index_create(bytes(text_), text_.size(), table_.data(), &l.ix), "sfx_index_create");
        });
        return l.ix;
    }
    std::string text_;
    std::vector<uint32_t> table_;
    mutable std::unique_ptr<LazyIndex> lazy_ = std::make_unique<LazyIndex>();
};

}  // namespace suffix
