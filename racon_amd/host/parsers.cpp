#include "parsers.hpp"

#include <zlib.h>

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace racon {
namespace io {

namespace {

// Line reader over gzread (plain files pass through zlib untouched).
class GzLines {
public:
    explicit GzLines(const std::string& path) : buf_(1 << 20) {
        f_ = gzopen(path.c_str(), "rb");
        if (!f_) throw std::runtime_error("[racon::io] error: unable to open file " + path + "!");
        gzbuffer(f_, 1 << 18);
    }
    ~GzLines() { if (f_) gzclose(f_); }
    // next line without its terminator and without trailing whitespace; false at EOF
    bool next(std::string& line) {
        line.clear();
        bool any = false;
        for (;;) {
            if (pos_ == len_) {
                const int n = gzread(f_, buf_.data(), static_cast<unsigned>(buf_.size()));
                if (n < 0) throw std::runtime_error("[racon::io] error: corrupted compressed stream!");
                if (n == 0) break;
                pos_ = 0; len_ = static_cast<size_t>(n);
            }
            any = true;
            const char* s = buf_.data() + pos_;
            const char* nl = static_cast<const char*>(memchr(s, '\n', len_ - pos_));
            if (nl) { line.append(s, nl - s); pos_ += (nl - s) + 1; break; }
            line.append(s, len_ - pos_); pos_ = len_;
        }
        while (!line.empty() && isspace(static_cast<unsigned char>(line.back()))) line.pop_back();
        return any;
    }
private:
    gzFile f_ = nullptr;
    std::vector<char> buf_;
    size_t pos_ = 0, len_ = 0;
};

// record name = header text up to the first whitespace (bioparser's Shorten)
uint32_t short_name(const std::string& s, size_t from) {
    size_t i = from;
    while (i < s.size() && !isspace(static_cast<unsigned char>(s[i]))) ++i;
    return static_cast<uint32_t>(i - from);
}

void split(const std::string& line, char sep, std::vector<std::pair<const char*, uint32_t>>& out, size_t max_fields) {
    out.clear();
    size_t a = 0;
    while (out.size() + 1 < max_fields) {
        const size_t b = line.find(sep, a);
        if (b == std::string::npos) break;
        out.emplace_back(line.data() + a, static_cast<uint32_t>(b - a));
        a = b + 1;
    }
    size_t b = line.find(sep, a);
    if (b == std::string::npos) b = line.size();
    out.emplace_back(line.data() + a, static_cast<uint32_t>(b - a));
}

uint32_t to_u32(const std::pair<const char*, uint32_t>& f) { return static_cast<uint32_t>(strtoull(std::string(f.first, f.second).c_str(), nullptr, 10)); }

}  // namespace

bool has_suffix(const std::string& s, const std::string& suffix) {
    return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}
bool is_fasta_path(const std::string& p) {
    for (const char* e : {".fasta", ".fasta.gz", ".fna", ".fna.gz", ".fa", ".fa.gz"}) if (has_suffix(p, e)) return true;
    return false;
}
bool is_fastq_path(const std::string& p) {
    for (const char* e : {".fastq", ".fastq.gz", ".fq", ".fq.gz"}) if (has_suffix(p, e)) return true;
    return false;
}

void read_fasta(const std::string& path, const std::function<void(const SeqRecord&)>& cb) {
    GzLines in(path);
    std::string line, header, data;
    bool open = false;
    auto flush = [&]() {
        if (!open) return;
        const uint32_t nl = short_name(header, 1);
        if (nl == 0 || data.empty()) throw std::runtime_error("[racon::io] error: invalid FASTA record in " + path + "!");
        cb(SeqRecord{header.data() + 1, nl, data.data(), static_cast<uint32_t>(data.size()), nullptr, 0});
    };
    while (in.next(line)) {
        if (!line.empty() && line[0] == '>') { flush(); header = line; data.clear(); open = true; }
        else if (open) data += line;
        else if (!line.empty()) throw std::runtime_error("[racon::io] error: invalid FASTA file " + path + "!");
    }
    flush();
}

void read_fastq(const std::string& path, const std::function<void(const SeqRecord&)>& cb) {
    // multi-line FASTQ: bases run until the '+' line, qualities until they are as long as the bases
    GzLines in(path);
    std::string line, header, data, qual;
    while (in.next(line)) {
        if (line.empty()) continue;
        if (line[0] != '@') throw std::runtime_error("[racon::io] error: invalid FASTQ file " + path + "!");
        header = line; data.clear(); qual.clear();
        bool plus = false;
        while (in.next(line)) { if (!line.empty() && line[0] == '+') { plus = true; break; } data += line; }
        while (plus && qual.size() < data.size() && in.next(line)) qual += line;
        const uint32_t nl = short_name(header, 1);
        if (!plus || nl == 0 || data.empty() || qual.size() != data.size())
            throw std::runtime_error("[racon::io] error: invalid FASTQ record in " + path + "!");
        cb(SeqRecord{header.data() + 1, nl, data.data(), static_cast<uint32_t>(data.size()), qual.data(), static_cast<uint32_t>(qual.size())});
    }
}

void read_paf(const std::string& path, const std::function<void(const PafRecord&)>& cb) {
    GzLines in(path);
    std::string line;
    std::vector<std::pair<const char*, uint32_t>> f;
    while (in.next(line)) {
        if (line.empty()) continue;
        split(line, '\t', f, 13);
        if (f.size() < 12) throw std::runtime_error("[racon::io] error: invalid PAF record in " + path + "!");
        PafRecord r{f[0].first, short_name(std::string(f[0].first, f[0].second), 0), to_u32(f[1]), to_u32(f[2]), to_u32(f[3]),
                    f[4].second ? f[4].first[0] : '+', f[5].first, short_name(std::string(f[5].first, f[5].second), 0),
                    to_u32(f[6]), to_u32(f[7]), to_u32(f[8]), to_u32(f[9]), to_u32(f[10]), to_u32(f[11])};
        cb(r);
    }
}

void read_mhap(const std::string& path, const std::function<void(const MhapRecord&)>& cb) {
    GzLines in(path);
    std::string line;
    std::vector<std::pair<const char*, uint32_t>> f;
    while (in.next(line)) {
        if (line.empty()) continue;
        split(line, ' ', f, 13);
        if (f.size() < 12) throw std::runtime_error("[racon::io] error: invalid MHAP record in " + path + "!");
        MhapRecord r{strtoull(std::string(f[0].first, f[0].second).c_str(), nullptr, 10),
                     strtoull(std::string(f[1].first, f[1].second).c_str(), nullptr, 10),
                     atof(std::string(f[2].first, f[2].second).c_str()), to_u32(f[3]), to_u32(f[4]), to_u32(f[5]), to_u32(f[6]),
                     to_u32(f[7]), to_u32(f[8]), to_u32(f[9]), to_u32(f[10]), to_u32(f[11])};
        cb(r);
    }
}

void read_sam(const std::string& path, const std::function<void(const SamRecord&)>& cb) {
    GzLines in(path);
    std::string line;
    std::vector<std::pair<const char*, uint32_t>> f;
    while (in.next(line)) {
        if (line.empty() || line[0] == '@') continue;
        split(line, '\t', f, 12);
        if (f.size() < 11) throw std::runtime_error("[racon::io] error: invalid SAM record in " + path + "!");
        SamRecord r{f[0].first, f[0].second, to_u32(f[1]), f[2].first, f[2].second, to_u32(f[3]), to_u32(f[4]),
                    f[5].first, f[5].second, f[9].first, f[9].second, f[10].first, f[10].second};
        cb(r);
    }
}

}  // namespace io
}  // namespace racon
