// overlap.hpp — one read-to-target overlap, mirroring racon::Overlap
// (reference src/overlap.hpp:32-125, src/overlap.cpp): three record forms
// (MHAP / PAF / SAM), name -> index resolution, global pre-alignment of the two
// segments when the record has no CIGAR (host CPU, see nw_path.hpp) and the
// CIGAR -> per-window (target, query) breaking points.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "parsers.hpp"

namespace racon {

class Sequence;

class Overlap {
public:
    explicit Overlap(const io::MhapRecord& r);   // reference src/overlap.cpp:15-27
    explicit Overlap(const io::PafRecord& r);    // reference src/overlap.cpp:29-42
    explicit Overlap(const io::SamRecord& r);    // reference src/overlap.cpp:44-108
    Overlap(const Overlap&) = delete;
    Overlap& operator=(const Overlap&) = delete;

    uint32_t q_id() const { return static_cast<uint32_t>(q_id_); }
    uint32_t t_id() const { return static_cast<uint32_t>(t_id_); }
    uint32_t strand() const { return strand_; }
    bool is_valid() const { return is_valid_; }
    uint32_t length() const { return length_; }
    double error() const { return error_; }
    const std::string& cigar() const { return cigar_; }
    const std::vector<std::pair<uint32_t, uint32_t>>& breaking_points() const { return breaking_points_; }

    // reference src/overlap.cpp:129-177
    void transmute(const std::vector<std::unique_ptr<Sequence>>& sequences,
                   const std::unordered_map<std::string, uint64_t>& name_to_id,
                   const std::unordered_map<uint64_t, uint64_t>& id_to_id);
    // reference src/overlap.cpp:179-203
    void find_breaking_points(const std::vector<std::unique_ptr<Sequence>>& sequences, uint32_t window_length, bool keep_cigar = false,
                              bool cigar_only = false,    // cigar_only: align if needed, leave the CIGAR walk to the device
                              bool no_align = false);     // no_align (with cigar_only): an overlap without a CIGAR is aligned on the device too
    // first query position on the overlap's strand / target extent: what breaking_points_from_cigar starts from
    uint32_t q_start_on_strand() const { return strand_ ? (q_length_ - q_end_) : q_begin_; }
    uint32_t q_begin() const { return q_begin_; }          // the aligned segment on the forward read
    uint32_t q_end() const { return q_end_; }
    uint32_t t_begin() const { return t_begin_; }
    uint32_t t_end() const { return t_end_; }

private:
    void set_extent(uint32_t q_span, uint32_t t_span);
    void breaking_points_from_cigar(uint32_t window_length);   // reference src/overlap.cpp:226-292

    std::string q_name_; uint64_t q_id_ = 0; uint32_t q_begin_ = 0, q_end_ = 0, q_length_ = 0;
    std::string t_name_; uint64_t t_id_ = 0; uint32_t t_begin_ = 0, t_end_ = 0, t_length_ = 0;
    uint32_t strand_ = 0, length_ = 0;
    double error_ = 0;
    std::string cigar_;
    bool is_valid_ = true, is_transmuted_ = false;
    std::vector<std::pair<uint32_t, uint32_t>> breaking_points_;
};

}  // namespace racon
