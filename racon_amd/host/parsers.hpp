// parsers.hpp — gz-aware record readers for the five formats racon accepts
// (reference src/polisher.cpp:85-135 picks a bioparser 3.0.15 parser by file
// suffix; bioparser itself is an un-vendored dependency, CMakeLists.txt:26-31).
// Each reader hands raw fields to a callback; Sequence / Overlap construction
// stays with the caller, exactly the fields the reference's private ctors take
// (src/sequence.hpp:64-69, src/overlap.hpp:84-99).
#pragma once
#include <cstdint>
#include <functional>
#include <string>

namespace racon {
namespace io {

struct SeqRecord { const char* name; uint32_t name_len; const char* data; uint32_t data_len; const char* qual; uint32_t qual_len; };
struct PafRecord { const char* q_name; uint32_t q_name_len; uint32_t q_len, q_begin, q_end; char orientation;
                   const char* t_name; uint32_t t_name_len; uint32_t t_len, t_begin, t_end, matches, length, mapq; };
struct MhapRecord { uint64_t a_id, b_id; double error; uint32_t minmers, a_rc, a_begin, a_end, a_len, b_rc, b_begin, b_end, b_len; };
struct SamRecord { const char* q_name; uint32_t q_name_len; uint32_t flag; const char* t_name; uint32_t t_name_len; uint32_t t_begin, mapq;
                   const char* cigar; uint32_t cigar_len; const char* seq; uint32_t seq_len; const char* qual; uint32_t qual_len; };

// All readers throw std::runtime_error on unreadable files / malformed records.
void read_fasta(const std::string& path, const std::function<void(const SeqRecord&)>& cb);
void read_fastq(const std::string& path, const std::function<void(const SeqRecord&)>& cb);
void read_paf(const std::string& path, const std::function<void(const PafRecord&)>& cb);
void read_mhap(const std::string& path, const std::function<void(const MhapRecord&)>& cb);
void read_sam(const std::string& path, const std::function<void(const SamRecord&)>& cb);

bool has_suffix(const std::string& s, const std::string& suffix);
bool is_fasta_path(const std::string& p);   // .fasta .fna .fa (+ .gz)
bool is_fastq_path(const std::string& p);   // .fastq .fq (+ .gz)

}  // namespace io
}  // namespace racon
