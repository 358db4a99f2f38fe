#include "overlap.hpp"

#include <algorithm>
#include <cstdlib>

#include "fatal.hpp"
#include "nw_path.hpp"
#include "sequence.hpp"

namespace racon {

namespace {

// Walks a CIGAR string: fn(count, op) for every operation.
template <class F>
void for_each_cigar_op(const std::string& cigar, F fn) {
    uint64_t count = 0; bool have = false;
    for (char c : cigar) {
        if (c >= '0' && c <= '9') { count = count * 10 + static_cast<uint64_t>(c - '0'); have = true; }
        else { fn(static_cast<uint32_t>(have ? count : 0), c); count = 0; have = false; }
    }
}

bool consumes_both(char op) { return op == 'M' || op == '=' || op == 'X'; }

}  // namespace

void Overlap::set_extent(uint32_t q_span, uint32_t t_span) {
    length_ = std::max(q_span, t_span);
    error_ = 1 - std::min(q_span, t_span) / static_cast<double>(length_);
}

Overlap::Overlap(const io::MhapRecord& r)
        : q_id_(r.a_id - 1), q_begin_(r.a_begin), q_end_(r.a_end), q_length_(r.a_len),
          t_id_(r.b_id - 1), t_begin_(r.b_begin), t_end_(r.b_end), t_length_(r.b_len), strand_(r.a_rc ^ r.b_rc) {
    set_extent(q_end_ - q_begin_, t_end_ - t_begin_);
}

Overlap::Overlap(const io::PafRecord& r)
        : q_name_(r.q_name, r.q_name_len), q_begin_(r.q_begin), q_end_(r.q_end), q_length_(r.q_len),
          t_name_(r.t_name, r.t_name_len), t_begin_(r.t_begin), t_end_(r.t_end), t_length_(r.t_len),
          strand_(r.orientation == '-') {
    set_extent(q_end_ - q_begin_, t_end_ - t_begin_);
}

Overlap::Overlap(const io::SamRecord& r)
        : q_name_(r.q_name, r.q_name_len), t_name_(r.t_name, r.t_name_len), t_begin_(r.t_begin - 1),
          strand_(r.flag & 0x10), cigar_(r.cigar, r.cigar_len), is_valid_(!(r.flag & 0x4)) {
    if (cigar_.size() < 2 && is_valid_) fatal("[Racon::Overlap::Overlap] error: missing alignment from SAM object!");
    if (cigar_.size() < 2) return;
    // query coordinates come from the clips and the aligned operations
    uint32_t q_aligned = 0, q_clipped = 0, t_aligned = 0;
    bool first = true;
    for_each_cigar_op(cigar_, [&](uint32_t n, char op) {
        if (first && (op == 'S' || op == 'H')) q_begin_ = n;
        first = false;
        if (consumes_both(op)) { q_aligned += n; t_aligned += n; }
        else if (op == 'I') q_aligned += n;
        else if (op == 'D' || op == 'N') t_aligned += n;
        else if (op == 'S' || op == 'H') q_clipped += n;
    });
    q_end_ = q_begin_ + q_aligned;
    q_length_ = q_clipped + q_aligned;
    if (strand_) { const uint32_t b = q_begin_; q_begin_ = q_length_ - q_end_; q_end_ = q_length_ - b; }
    t_end_ = t_begin_ + t_aligned;
    set_extent(q_aligned, t_aligned);
}

void Overlap::transmute(const std::vector<std::unique_ptr<Sequence>>& sequences,
                        const std::unordered_map<std::string, uint64_t>& name_to_id,
                        const std::unordered_map<uint64_t, uint64_t>& id_to_id) {
    if (!is_valid_ || is_transmuted_) return;
    auto resolve = [&](std::string& name, uint64_t& id, const char* tag, uint64_t low_bit) -> bool {
        if (!name.empty()) {
            auto it = name_to_id.find(name + tag);
            if (it == name_to_id.end()) return false;
            id = it->second;
            std::string().swap(name);
        } else {
            auto it = id_to_id.find(id << 1 | low_bit);
            if (it == id_to_id.end()) return false;
            id = it->second;
        }
        return true;
    };
    if (!resolve(q_name_, q_id_, "q", 0)) { is_valid_ = false; return; }
    if (q_length_ != sequences[q_id_]->data().size())
        fatal("[racon::Overlap::transmute] error: unequal lengths in sequence and overlap file for sequence " +
              sequences[q_id_]->name() + "!");
    if (!resolve(t_name_, t_id_, "t", 1)) { is_valid_ = false; return; }
    if (t_length_ != 0 && t_length_ != sequences[t_id_]->data().size())
        fatal("[racon::Overlap::transmute] error: unequal lengths in target and overlap file for target " +
              sequences[t_id_]->name() + "!");
    t_length_ = static_cast<uint32_t>(sequences[t_id_]->data().size());   // SAM records carry no target length
    is_transmuted_ = true;
}

void Overlap::find_breaking_points(const std::vector<std::unique_ptr<Sequence>>& sequences, uint32_t window_length, bool keep_cigar, bool cigar_only,
                                   bool no_align) {
    if (!is_transmuted_) fatal("[racon::Overlap::find_breaking_points] error: overlap is not transmuted!");
    if (!breaking_points_.empty()) return;
    if (cigar_.empty() && !(cigar_only && no_align)) {
        const char* q = !strand_ ? &(sequences[q_id_]->data()[q_begin_])
                                 : &(sequences[q_id_]->reverse_complement()[q_length_ - q_end_]);
        const char* t = &(sequences[t_id_]->data()[t_begin_]);
        cigar_ = nwpath::align_cigar(q, q_end_ - q_begin_, t, t_end_ - t_begin_);
    }
    if (cigar_only) { breaking_points_.emplace_back(0, 0); breaking_points_.emplace_back(0, 0); return; }   // (non-empty: "done")
    breaking_points_from_cigar(window_length);
    if (!keep_cigar) std::string().swap(cigar_);
}

void Overlap::breaking_points_from_cigar(uint32_t window_length) {
    // last target position of every window the overlap touches
    std::vector<int64_t> ends;
    for (uint64_t i = 0; i < t_end_; i += window_length) if (i > t_begin_) ends.push_back(static_cast<int64_t>(i) - 1);
    ends.push_back(static_cast<int64_t>(t_end_) - 1);

    size_t w = 0;
    bool open = false;                                   // a match has been seen in the current window
    std::pair<uint32_t, uint32_t> first{0, 0}, last{0, 0};
    int64_t q = static_cast<int64_t>(strand_ ? (q_length_ - q_end_) : q_begin_) - 1;
    int64_t t = static_cast<int64_t>(t_begin_) - 1;
    auto close_window_if_at_end = [&]() {
        if (w < ends.size() && t == ends[w]) {
            if (open) { breaking_points_.push_back(first); breaking_points_.push_back(last); }
            open = false; ++w;
        }
    };
    for_each_cigar_op(cigar_, [&](uint32_t n, char op) {
        if (consumes_both(op)) {
            for (uint32_t k = 0; k < n; ++k) {
                ++q; ++t;
                if (!open) { open = true; first = {static_cast<uint32_t>(t), static_cast<uint32_t>(q)}; }
                last = {static_cast<uint32_t>(t + 1), static_cast<uint32_t>(q + 1)};
                close_window_if_at_end();
            }
        } else if (op == 'I') {
            q += n;
        } else if (op == 'D' || op == 'N') {
            for (uint32_t k = 0; k < n; ++k) { ++t; close_window_if_at_end(); }
        }
    });
}

}  // namespace racon
