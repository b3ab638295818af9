// suffix_array_amd.hpp -- C++ host-side mirror of the reference's construction interface, over
// the C ABI of suffix_array_amd.h.  Same names, argument meaning and error behaviour as the
// Rust crate for the path in scope:
//   saca(), MAX_LENGTH                      reference src/saca.rs:6-15
//   SuffixArray::new_/set/len/is_empty/into_parts/from_parts/unchecked_from_parts
//                                           reference src/sa.rs:23-70, check_integrity src/sa.rs:72-84
//   SuffixArray::enable_buckets / buckets   reference src/sa.rs:89-119 (the table, built on the GPU from the text alone)
// Rust panics (assert!, engine failure) are std::logic_error / std::runtime_error here.
#pragma once
#include "suffix_array_amd.h"

#include <cstdint>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace suffix_array {

constexpr std::size_t MAX_LENGTH = SA_AMD_MAX_LENGTH;            // src/saca.rs:6

// pub fn saca(s: &[u8], sa: &mut [u32])                          // src/saca.rs:9-15
inline void saca(const std::uint8_t *s, std::size_t n, std::uint32_t *sa, std::size_t sa_len)
{
    if (n > MAX_LENGTH) throw std::logic_error("assertion failed: s.len() <= MAX_LENGTH");      // :10
    if (n + 1 != sa_len) throw std::logic_error("assertion failed: s.len() + 1 == sa.len()");   // :11
    const std::int32_t rc = sa_amd_saca_u8(s, sa, static_cast<std::int32_t>(n));                 // :13-14
    if (rc != SA_AMD_OK) throw std::runtime_error(std::string("suffix_array_amd: ") + sa_amd_strerror(rc));
}

class SuffixArray {
public:
    // SuffixArray::new                                            // src/sa.rs:23-27
    static SuffixArray new_(const std::uint8_t *s, std::size_t n)
    {
        SuffixArray r(s, n, std::vector<std::uint32_t>(n + 1, 0));
        saca(s, n, r.sa_.data(), r.sa_.size());
        return r;
    }
    // SuffixArray::set: re-runs construction into the resized buffer; like the reference it does
    // not replace the stored text                                 // src/sa.rs:30-33
    void set(const std::uint8_t *s, std::size_t n)
    {
        sa_.resize(n + 1, 0);
        saca(s, n, sa_.data(), sa_.size());
    }
    void fit() { sa_.shrink_to_fit(); }                           // src/sa.rs:36-38
    std::size_t len() const { return n_; }                        // src/sa.rs:41-43
    bool is_empty() const { return n_ == 0; }                     // src/sa.rs:46-48
    std::pair<const std::uint8_t *, std::vector<std::uint32_t>> into_parts() &&   // src/sa.rs:51-53
    {
        return { s_, std::move(sa_) };
    }
    const std::vector<std::uint32_t> &sa() const { return sa_; }
    // from_parts: compose and check the integrity                 // src/sa.rs:57-64
    static std::optional<SuffixArray> from_parts(const std::uint8_t *s, std::size_t n, std::vector<std::uint32_t> sa)
    {
        SuffixArray r(s, n, std::move(sa));
        if (r.check_integrity()) return r;
        return std::nullopt;
    }
    static SuffixArray unchecked_from_parts(const std::uint8_t *s, std::size_t n, std::vector<std::uint32_t> sa)
    {
        return SuffixArray(s, n, std::move(sa));                  // src/sa.rs:68-70
    }
    // enable_buckets: 256 * 257 + 1 right bucket edges in the layout of src/sa.rs:94, from the bigram counts of the text
    // (src/sa.rs:96-116); a second call is a no-op like the reference's (src/sa.rs:90-92)
    void enable_buckets()
    {
        if (!bkt_.empty()) return;
        std::vector<std::uint32_t> b(SA_AMD_BUCKET_TABLE_LEN);
        const std::int32_t rc = sa_amd_bucket_table(s_, static_cast<std::int32_t>(n_), nullptr, b.data());
        if (rc != SA_AMD_OK) throw std::runtime_error(std::string("suffix_array_amd: ") + sa_amd_strerror(rc));
        bkt_ = std::move(b);
    }
    const std::vector<std::uint32_t> &buckets() const { return bkt_; }     // empty: not enabled (the reference's bkt: None)

private:
    SuffixArray(const std::uint8_t *s, std::size_t n, std::vector<std::uint32_t> sa) : s_(s), n_(n), sa_(std::move(sa)) {}
    bool check_integrity() const                                   // src/sa.rs:72-84 (literal form)
    {
        if (n_ + 1 != sa_.size()) return false;
        for (std::size_t i = 1; i < sa_.size(); ++i) {
            const std::size_t a = sa_[i - 1], b = sa_[i];
            if (a > n_ || b > n_) throw std::out_of_range("suffix offset out of range");   // slice index panics
            const std::size_t la = n_ - a, lb = n_ - b, l = la < lb ? la : lb;
            int c = l ? std::memcmp(s_ + a, s_ + b, l) : 0;
            if (c == 0) c = (la > lb) - (la < lb);
            if (c >= 0) return false;
        }
        return true;
    }
    const std::uint8_t *s_;
    std::size_t n_;
    std::vector<std::uint32_t> sa_;
    std::vector<std::uint32_t> bkt_;
};

}  // namespace suffix_array
