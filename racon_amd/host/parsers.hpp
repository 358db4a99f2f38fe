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
#include <utility>
#include <vector>

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

// ---- parallel ingest (the reference parses on the calling thread, src/polisher.cpp:200-349) ----
// One thread inflates the file (an UNCOMPRESSED file is memory-mapped instead and framed in place: no copy through zlib's buffer, no
// copy into the batch -- the framing thread then runs at memchr speed and the parse workers are the limit) and frames whole records
// into batches of a few MiB of text; `threads` workers take the
// batches (in any order) and call `work`: field parsing and object construction (upper-casing, quality checks, CIGAR
// scans) leave the inflating thread.  Batch::number counts batches in file order and Batch::index0 is the ordinal of its
// first record, so the caller can restore file order.  threads <= 1: everything on the calling thread, in order.
// Exceptions thrown by `work` or by the reader are rethrown on the calling thread.
enum class Format { kFasta, kFastq, kPaf, kMhap, kSam };
struct Batch {
    std::string text;                                // the batch's bytes (inflated input), or
    const char* ptr = nullptr;                       // ... a view into the memory-mapped file (uncompressed input): valid inside `work`
    const char* data() const { return ptr ? ptr : text.data(); }
    std::vector<std::pair<size_t, size_t>> recs;     // (offset, length) of every record in text, terminators stripped
    uint64_t number = 0, index0 = 0;
};
void read_batches(const std::string& path, Format format, uint32_t threads, const std::function<void(Batch&)>& work);
// One framed record -> fields.  `data` / `qual` receive the unwrapped lines of multi-line FASTA / FASTQ records and must
// outlive `r`.  Throw std::runtime_error on malformed records (`path` only names the file in the message).
void parse_seq(Format format, const char* s, size_t n, const std::string& path, std::string& data, std::string& qual, SeqRecord& r);
void parse_paf(const char* s, size_t n, const std::string& path, PafRecord& r);
void parse_mhap(const char* s, size_t n, const std::string& path, MhapRecord& r);
bool parse_sam(const char* s, size_t n, const std::string& path, SamRecord& r);     // false: header line
Format format_of(const std::string& path);          // by suffix; the caller has validated it

bool has_suffix(const std::string& s, const std::string& suffix);
bool is_fasta_path(const std::string& p);   // .fasta .fna .fa (+ .gz)
bool is_fastq_path(const std::string& p);   // .fastq .fq (+ .gz)

}  // namespace io
}  // namespace racon
