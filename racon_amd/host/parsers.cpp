#include "parsers.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cctype>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

namespace racon {
namespace io {

namespace {

inline bool is_space(char c) { return isspace(static_cast<unsigned char>(c)) != 0; }

// record name = header text up to the first whitespace (bioparser's Shorten)
uint32_t short_name(const char* s, size_t n) {
    size_t i = 0;
    while (i < n && !is_space(s[i])) ++i;
    return static_cast<uint32_t>(i);
}

typedef std::pair<const char*, uint32_t> Field;

void split(const char* s, size_t n, char sep, std::vector<Field>& out, size_t max_fields) {
    out.clear();
    size_t a = 0;
    while (out.size() + 1 < max_fields) {
        const char* p = static_cast<const char*>(memchr(s + a, sep, n - a));
        if (!p) break;
        out.emplace_back(s + a, static_cast<uint32_t>(p - (s + a)));
        a = static_cast<size_t>(p - s) + 1;
    }
    const char* p = static_cast<const char*>(memchr(s + a, sep, n - a));
    const size_t b = p ? static_cast<size_t>(p - s) : n;
    out.emplace_back(s + a, static_cast<uint32_t>(b - a));
}

uint64_t to_u64(const Field& f) {
    uint64_t v = 0;
    uint32_t i = 0;
    while (i < f.second && is_space(f.first[i])) ++i;
    for (; i < f.second && f.first[i] >= '0' && f.first[i] <= '9'; ++i) v = v * 10 + static_cast<uint64_t>(f.first[i] - '0');
    return v;
}
uint32_t to_u32(const Field& f) { return static_cast<uint32_t>(to_u64(f)); }

// strips trailing whitespace (the serial reader did it per line)
size_t rstrip(const char* s, size_t n) { while (n > 0 && is_space(s[n - 1])) --n; return n; }

// ---- framing: how many leading bytes of buf[0, n) are whole records; their spans go to recs -------------------------
// Line formats: a record is a line.  FASTA: a record runs to the next '>' at the start of a line.  FASTQ (multi-line
// tolerant, as the reference's parser): '@' header line, bases until the '+' line, qualities until they are as long as the bases.
size_t frame_lines(const char* buf, size_t n, bool eof, std::vector<std::pair<size_t, size_t>>& recs) {
    size_t pos = 0;
    while (pos < n) {
        const char* nl = static_cast<const char*>(memchr(buf + pos, '\n', n - pos));
        if (!nl && !eof) break;
        const size_t end = nl ? static_cast<size_t>(nl - buf) : n;
        const size_t len = rstrip(buf + pos, end - pos);
        if (len) recs.emplace_back(pos, len);
        pos = nl ? end + 1 : n;
    }
    return pos;
}

size_t frame_fasta(const char* buf, size_t n, bool eof, std::vector<std::pair<size_t, size_t>>& recs, const std::string& path) {
    size_t pos = 0;
    // leading blank lines
    while (pos < n && is_space(buf[pos])) ++pos;
    if (pos < n && buf[pos] != '>') throw std::runtime_error("[racon::io] error: invalid FASTA file " + path + "!");
    while (pos < n) {
        // next record start: "\n>" after pos
        size_t q = pos + 1;
        const char* next = nullptr;
        while (q < n) {
            const char* nl = static_cast<const char*>(memchr(buf + q, '\n', n - q));
            if (!nl) break;
            q = static_cast<size_t>(nl - buf) + 1;
            if (q < n && buf[q] == '>') { next = buf + q; break; }
            if (q >= n) break;
        }
        if (next) { recs.emplace_back(pos, rstrip(buf + pos, static_cast<size_t>(next - buf) - pos)); pos = static_cast<size_t>(next - buf); }
        else if (eof) { const size_t len = rstrip(buf + pos, n - pos); if (len) recs.emplace_back(pos, len); pos = n; }
        else break;
    }
    return pos;
}

size_t frame_fastq(const char* buf, size_t n, bool eof, std::vector<std::pair<size_t, size_t>>& recs, const std::string& path) {
    size_t pos = 0;
    for (;;) {
        while (pos < n && is_space(buf[pos])) ++pos;
        if (pos >= n) return n;
        if (buf[pos] != '@') throw std::runtime_error("[racon::io] error: invalid FASTQ file " + path + "!");
        size_t q = pos;
        auto next_line = [&](size_t& a, size_t& b) -> bool {      // [a, b) = next line without '\n'; false: not complete
            if (q >= n) return false;
            const char* nl = static_cast<const char*>(memchr(buf + q, '\n', n - q));
            if (!nl && !eof) return false;
            a = q; b = nl ? static_cast<size_t>(nl - buf) : n;
            q = nl ? b + 1 : n;
            return true;
        };
        size_t a, b;
        if (!next_line(a, b)) return pos;                          // header
        size_t bases = 0; bool plus = false;
        while (next_line(a, b)) {
            const size_t len = rstrip(buf + a, b - a);
            if (len && buf[a] == '+') { plus = true; break; }
            bases += len;
        }
        if (!plus) { if (eof) throw std::runtime_error("[racon::io] error: invalid FASTQ record in " + path + "!"); return pos; }
        size_t quals = 0; size_t end = q;
        while (quals < bases) {
            if (!next_line(a, b)) { if (eof) throw std::runtime_error("[racon::io] error: invalid FASTQ record in " + path + "!"); return pos; }
            quals += rstrip(buf + a, b - a); end = q;
        }
        if (bases == 0 && !eof && q >= n) return pos;              // nothing after '+' yet
        recs.emplace_back(pos, rstrip(buf + pos, end - pos));
        pos = end;
    }
}

size_t frame(Format f, const char* buf, size_t n, bool eof, std::vector<std::pair<size_t, size_t>>& recs, const std::string& path) {
    switch (f) {
        case Format::kFasta: return frame_fasta(buf, n, eof, recs, path);
        case Format::kFastq: return frame_fastq(buf, n, eof, recs, path);
        default: return frame_lines(buf, n, eof, recs);
    }
}

constexpr size_t kBlock = 4u << 20;       // inflate granularity = target batch size

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
Format format_of(const std::string& p) {
    if (is_fasta_path(p)) return Format::kFasta;
    if (is_fastq_path(p)) return Format::kFastq;
    if (has_suffix(p, ".mhap") || has_suffix(p, ".mhap.gz")) return Format::kMhap;
    if (has_suffix(p, ".paf") || has_suffix(p, ".paf.gz")) return Format::kPaf;
    return Format::kSam;
}

namespace {
// An uncompressed regular file, mapped read-only (nullptr: gzip data, a pipe, an empty or unmappable file -- the zlib path takes those).
struct Mapped {
    const char* base = nullptr; size_t size = 0;
    explicit Mapped(const std::string& path) {
        if (getenv("RACON_HIP_NO_MMAP")) return;
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        unsigned char magic[2] = {0, 0};
        if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 2 && pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
            void* p = mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
            if (p != MAP_FAILED) { base = static_cast<const char*>(p); size = static_cast<size_t>(st.st_size); madvise(p, size, MADV_SEQUENTIAL); }
        }
        close(fd);
    }
    ~Mapped() { if (base) munmap(const_cast<char*>(base), size); }
    Mapped(const Mapped&) = delete;
};
}  // namespace

void read_batches(const std::string& path, Format format, uint32_t threads, const std::function<void(Batch&)>& work) {
    const Mapped mapped(path);
    gzFile f = mapped.base ? nullptr : gzopen(path.c_str(), "rb");
    if (!mapped.base && !f) throw std::runtime_error("[racon::io] error: unable to open file " + path + "!");
    if (f) gzbuffer(f, 1 << 18);
    struct Closer { gzFile f; ~Closer() { if (f) gzclose(f); } } closer{f};

    // bounded queue between the inflating thread (this one) and the workers
    std::mutex m;
    std::condition_variable cv_full, cv_empty;
    std::deque<Batch> queue;
    bool done = false;
    std::exception_ptr error;
    const size_t cap = std::max<size_t>(2, 2 * threads);
    std::vector<std::thread> pool;
    if (threads > 1) {
        for (uint32_t t = 0; t < threads; ++t)
            pool.emplace_back([&] {
                for (;;) {
                    Batch b;
                    {
                        std::unique_lock<std::mutex> lock(m);
                        cv_empty.wait(lock, [&] { return !queue.empty() || done; });
                        if (queue.empty()) return;
                        b = std::move(queue.front()); queue.pop_front();
                    }
                    cv_full.notify_one();
                    try { work(b); }
                    catch (...) {
                        std::lock_guard<std::mutex> lock(m);
                        if (!error) error = std::current_exception();
                    }
                }
            });
    }
    auto finish = [&]() {
        { std::lock_guard<std::mutex> lock(m); done = true; }
        cv_empty.notify_all();
        for (auto& t : pool) t.join();
    };
    try {
        std::string carry;                    // bytes after the last whole record of the previous block
        uint64_t number = 0, index = 0;
        bool eof = false;
        auto hand_over = [&](Batch& b) -> bool {          // false: a worker failed, stop reading
            b.number = number++; b.index0 = index; index += b.recs.size();
            if (threads > 1) {
                std::unique_lock<std::mutex> lock(m);
                cv_full.wait(lock, [&] { return queue.size() < cap || error; });
                if (error) return false;
                queue.push_back(std::move(b));
                lock.unlock();
                cv_empty.notify_one();
            } else {
                work(b);
            }
            return true;
        };
        if (mapped.base) {
            // framed in place: a window of a block's size from the first unframed byte, doubled while it holds no whole record
            size_t pos = 0, want = kBlock;
            while (pos < mapped.size) {
                const size_t n = std::min(want, mapped.size - pos);
                Batch b;
                b.ptr = mapped.base + pos;
                const size_t used = frame(format, b.ptr, n, pos + n == mapped.size, b.recs, path);
                if (b.recs.empty()) {
                    if (pos + n == mapped.size) break;                      // (trailing blank lines)
                    if (used == 0) { want *= 2; continue; }
                    pos += used; continue;
                }
                pos += used; want = kBlock;
                if (!hand_over(b)) break;
            }
            eof = true;
        }
        while (!eof) {
            Batch b;
            b.text.swap(carry);
            const size_t had = b.text.size();
            // A record longer than a block (a chromosome-scale FASTA entry) is carried over and framed again from its start
            // with every block: reading as much as is already carried doubles the block each time, so a record of N bytes
            // is scanned and copied O(N) in total, not O(N^2 / block).
            const size_t want = std::min<size_t>(std::max(kBlock, had), size_t(1) << 30);
            b.text.resize(had + want);
            size_t got = 0;
            while (got < want) {
                const int n = gzread(f, &b.text[had + got], static_cast<unsigned>(want - got));
                if (n < 0) throw std::runtime_error("[racon::io] error: corrupted compressed stream!");
                if (n == 0) { eof = true; break; }
                got += static_cast<size_t>(n);
            }
            b.text.resize(had + got);
            const size_t used = frame(format, b.text.data(), b.text.size(), eof, b.recs, path);
            carry.assign(b.text, used, std::string::npos);
            b.text.resize(used);
            if (b.recs.empty()) continue;
            if (!hand_over(b)) break;
        }
    } catch (...) {
        finish();
        throw;
    }
    finish();
    if (error) std::rethrow_exception(error);
}

void parse_seq(Format format, const char* s, size_t n, const std::string& path, std::string& data, std::string& qual, SeqRecord& r) {
    data.clear(); qual.clear();
    // header line
    const char* nl = static_cast<const char*>(memchr(s, '\n', n));
    const size_t hend = nl ? static_cast<size_t>(nl - s) : n;
    const size_t hlen = rstrip(s, hend);
    const uint32_t name_len = hlen > 1 ? short_name(s + 1, hlen - 1) : 0;
    size_t pos = nl ? hend + 1 : n;
    bool plus = format == Format::kFasta;
    if (format == Format::kFasta) {
        while (pos < n) {
            const char* e = static_cast<const char*>(memchr(s + pos, '\n', n - pos));
            const size_t end = e ? static_cast<size_t>(e - s) : n;
            data.append(s + pos, rstrip(s + pos, end - pos));
            pos = e ? end + 1 : n;
        }
        if (name_len == 0 || data.empty()) throw std::runtime_error("[racon::io] error: invalid FASTA record in " + path + "!");
        r = SeqRecord{s + 1, name_len, data.data(), static_cast<uint32_t>(data.size()), nullptr, 0};
        return;
    }
    {
        // the usual record -- header, ONE line of bases, '+', ONE line of qualities -- is handed on in place: no copy into data / qual
        const char* e1 = static_cast<const char*>(memchr(s + pos, '\n', n - pos));
        if (e1) {
            const size_t b0 = pos, b1 = static_cast<size_t>(e1 - s), blen = rstrip(s + b0, b1 - b0);
            const size_t p0 = b1 + 1;
            const char* e2 = p0 < n ? static_cast<const char*>(memchr(s + p0, '\n', n - p0)) : nullptr;
            if (e2 && s[p0] == '+' && blen && s[b0] != '+') {
                const size_t q0 = static_cast<size_t>(e2 - s) + 1;
                if (q0 < n) {
                    const char* e3 = static_cast<const char*>(memchr(s + q0, '\n', n - q0));
                    const size_t q1 = e3 ? static_cast<size_t>(e3 - s) : n, qlen = rstrip(s + q0, q1 - q0);
                    if (qlen == blen && name_len && rstrip(s + q0, n - q0) == qlen) {          // (nothing but white space behind the quality line)
                        r = SeqRecord{s + 1, name_len, s + b0, static_cast<uint32_t>(blen), s + q0, static_cast<uint32_t>(qlen)};
                        return;
                    }
                }
            }
        }
    }
    while (pos < n) {
        const char* e = static_cast<const char*>(memchr(s + pos, '\n', n - pos));
        const size_t end = e ? static_cast<size_t>(e - s) : n;
        const size_t len = rstrip(s + pos, end - pos);
        const bool is_plus = !plus && len && s[pos] == '+';
        if (is_plus) plus = true;
        else if (!plus) data.append(s + pos, len);
        else if (qual.size() < data.size()) qual.append(s + pos, len);
        pos = e ? end + 1 : n;
    }
    if (!plus || name_len == 0 || data.empty() || qual.size() != data.size())
        throw std::runtime_error("[racon::io] error: invalid FASTQ record in " + path + "!");
    r = SeqRecord{s + 1, name_len, data.data(), static_cast<uint32_t>(data.size()), qual.data(), static_cast<uint32_t>(qual.size())};
}

void parse_paf(const char* s, size_t n, const std::string& path, PafRecord& r) {
    std::vector<Field> f;
    split(s, n, '\t', f, 13);
    if (f.size() < 12) throw std::runtime_error("[racon::io] error: invalid PAF record in " + path + "!");
    r = PafRecord{f[0].first, short_name(f[0].first, f[0].second), to_u32(f[1]), to_u32(f[2]), to_u32(f[3]),
                  f[4].second ? f[4].first[0] : '+', f[5].first, short_name(f[5].first, f[5].second),
                  to_u32(f[6]), to_u32(f[7]), to_u32(f[8]), to_u32(f[9]), to_u32(f[10]), to_u32(f[11])};
}

void parse_mhap(const char* s, size_t n, const std::string& path, MhapRecord& r) {
    std::vector<Field> f;
    split(s, n, ' ', f, 13);
    if (f.size() < 12) throw std::runtime_error("[racon::io] error: invalid MHAP record in " + path + "!");
    r = MhapRecord{to_u64(f[0]), to_u64(f[1]), atof(std::string(f[2].first, f[2].second).c_str()), to_u32(f[3]), to_u32(f[4]),
                   to_u32(f[5]), to_u32(f[6]), to_u32(f[7]), to_u32(f[8]), to_u32(f[9]), to_u32(f[10]), to_u32(f[11])};
}

bool parse_sam(const char* s, size_t n, const std::string& path, SamRecord& r) {
    if (n == 0 || s[0] == '@') return false;
    std::vector<Field> f;
    split(s, n, '\t', f, 12);
    if (f.size() < 11) throw std::runtime_error("[racon::io] error: invalid SAM record in " + path + "!");
    r = SamRecord{f[0].first, f[0].second, to_u32(f[1]), f[2].first, f[2].second, to_u32(f[3]), to_u32(f[4]),
                  f[5].first, f[5].second, f[9].first, f[9].second, f[10].first, f[10].second};
    return true;
}

// ---- the serial readers: the same framing and parsing on the calling thread, records in file order ---------------------
void read_fasta(const std::string& path, const std::function<void(const SeqRecord&)>& cb) {
    std::string data, qual;
    read_batches(path, Format::kFasta, 1, [&](Batch& b) {
        for (const auto& rc : b.recs) { SeqRecord r; parse_seq(Format::kFasta, b.data() + rc.first, rc.second, path, data, qual, r); cb(r); }
    });
}
void read_fastq(const std::string& path, const std::function<void(const SeqRecord&)>& cb) {
    std::string data, qual;
    read_batches(path, Format::kFastq, 1, [&](Batch& b) {
        for (const auto& rc : b.recs) { SeqRecord r; parse_seq(Format::kFastq, b.data() + rc.first, rc.second, path, data, qual, r); cb(r); }
    });
}
void read_paf(const std::string& path, const std::function<void(const PafRecord&)>& cb) {
    read_batches(path, Format::kPaf, 1, [&](Batch& b) {
        for (const auto& rc : b.recs) { PafRecord r; parse_paf(b.data() + rc.first, rc.second, path, r); cb(r); }
    });
}
void read_mhap(const std::string& path, const std::function<void(const MhapRecord&)>& cb) {
    read_batches(path, Format::kMhap, 1, [&](Batch& b) {
        for (const auto& rc : b.recs) { MhapRecord r; parse_mhap(b.data() + rc.first, rc.second, path, r); cb(r); }
    });
}
void read_sam(const std::string& path, const std::function<void(const SamRecord&)>& cb) {
    read_batches(path, Format::kSam, 1, [&](Batch& b) {
        for (const auto& rc : b.recs) { SamRecord r; if (parse_sam(b.data() + rc.first, rc.second, path, r)) cb(r); }
    });
}

}  // namespace io
}  // namespace racon
