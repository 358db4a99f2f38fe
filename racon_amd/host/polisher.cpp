#include "polisher.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <exception>
#include <cstring>
#include <mutex>
#include <iostream>
#include <thread>
#include <unordered_map>

#include <sys/mman.h>
#include <sys/stat.h>

#include "fatal.hpp"
#include "hip_engine.hpp"
#include "host_util.hpp"
#include "nw_path.hpp"
#include "overlap.hpp"
#include "parsers.hpp"
#include "sequence.hpp"

namespace racon {

// ---------------------------------------------------------------- fatal / logger
namespace { bool g_fatal_throws = false; thread_local bool t_fatal_throws = false; }
void set_fatal_throws(bool on) { g_fatal_throws = on; }
void fatal(const std::string& message) {
    if (g_fatal_throws || t_fatal_throws) throw FatalError(message);
    fprintf(stderr, "%s\n", message.c_str());
    exit(1);
}
FatalThrowsScope::FatalThrowsScope() : previous(t_fatal_throws) { t_fatal_throws = true; }
FatalThrowsScope::~FatalThrowsScope() { t_fatal_throws = previous; }
void fatal_from(const std::exception_ptr& error) {
    try { std::rethrow_exception(error); }
    catch (const std::exception& e) { fatal(e.what()); }
    catch (...) { fatal("[racon::] error: unknown exception in a worker thread!"); }
}

void Logger::log() {
    const auto now = std::chrono::steady_clock::now();
    if (time_point_ != std::chrono::time_point<std::chrono::steady_clock>())
        time_ += std::chrono::duration_cast<std::chrono::duration<double>>(now - time_point_).count();
    time_point_ = now;
}
void Logger::log(const std::string& msg) const { std::cerr << msg << " " << std::fixed << seconds_since(time_point_) << " s" << std::endl; }
void Logger::bar(const std::string& msg) {
    ++bar_;
    std::cerr << msg << " [" << std::string(bar_, '=') << (bar_ == 20 ? "" : ">" + std::string(19 - bar_, ' ')) << "] "
              << std::fixed << seconds_since(time_point_) << " s";
    bar_ %= 20;
    std::cerr << (bar_ == 0 ? "\n" : "\r") << std::flush;
}
void Logger::total(const std::string& msg) const { std::cerr << msg << " " << std::fixed << time_ + seconds_since(time_point_) << " s" << std::endl; }

// ---------------------------------------------------------------- factory
std::unique_ptr<Polisher> createPolisher(const std::string& sequences_path, const std::string& overlaps_path,
    const std::string& target_path, PolisherType type, uint32_t window_length, double quality_threshold,
    double error_threshold, bool trim, int8_t match, int8_t mismatch, int8_t gap, uint32_t num_threads,
    uint32_t cudapoa_batches, bool cuda_banded_alignment, uint32_t cudaaligner_batches, uint32_t cudaaligner_band_width) {
    // -b / --cudaaligner-band-width: the reference's banded approximations; the DP here is exact at any width
    (void)cuda_banded_alignment; (void)cudaaligner_band_width;
    if (type != PolisherType::kC && type != PolisherType::kF) fatal("[racon::createPolisher] error: invalid polisher type!");
    if (window_length == 0) fatal("[racon::createPolisher] error: invalid window length!");
    const std::string seq_ext = "(valid extensions: .fasta, .fasta.gz, .fna, .fna.gz, .fa, .fa.gz, .fastq, .fastq.gz, .fq, .fq.gz)!";
    if (!io::is_fasta_path(sequences_path) && !io::is_fastq_path(sequences_path))
        fatal("[racon::createPolisher] error: file " + sequences_path + " has unsupported format extension " + seq_ext);
    bool ovl_ok = false;
    for (const char* e : {".mhap", ".mhap.gz", ".paf", ".paf.gz", ".sam", ".sam.gz"}) ovl_ok |= io::has_suffix(overlaps_path, e);
    if (!ovl_ok)
        fatal("[racon::createPolisher] error: file " + overlaps_path + " has unsupported format extension "
              "(valid extensions: .mhap, .mhap.gz, .paf, .paf.gz, .sam, .sam.gz)!");
    if (!io::is_fasta_path(target_path) && !io::is_fastq_path(target_path))
        fatal("[racon::createPolisher] error: file " + target_path + " has unsupported format extension " + seq_ext);
    std::unique_ptr<Polisher> polisher(new Polisher(sequences_path, overlaps_path, target_path, type, window_length,
        quality_threshold, error_threshold, trim, match, mismatch, gap, num_threads, cudapoa_batches));
    // --cudaaligner-batches n > 0: overlap alignment on the device, as in the reference (src/main.cpp:125-127 ->
    // src/polisher.cpp:137-147 -> CUDAPolisher::find_overlap_breaking_points, src/cuda/cudapolisher.cpp:74-214): overlaps
    // without a CIGAR are aligned in HBM by the byte-exact pair aligner and the windows are cut there
    // (rcn_engine_build_windows_from_pairs); a file whose overlaps carry CIGARs has nothing to align and takes the
    // device's CIGAR walk.  Same FASTA either way (tests/test_cli_e2e.py).
    if (cudaaligner_batches > 0) polisher->device_windows(true, false, true);
    return polisher;
}

Polisher::Polisher(const std::string& sequences_path, const std::string& overlaps_path, const std::string& target_path,
                   PolisherType type, uint32_t window_length, double quality_threshold, double error_threshold, bool trim,
                   int8_t match, int8_t mismatch, int8_t gap, uint32_t num_threads, uint32_t hip_batches)
        : sequences_path_(sequences_path), overlaps_path_(overlaps_path), target_path_(target_path), type_(type),
          quality_threshold_(quality_threshold), error_threshold_(error_threshold), trim_(trim), match_(match),
          mismatch_(mismatch), gap_(gap), num_threads_(std::max<uint32_t>(1, num_threads)),
          hip_batches_(std::max<uint32_t>(1, hip_batches)), dummy_quality_(window_length, '!'),
          window_length_(window_length), logger_(new Logger()) {}

Polisher::~Polisher() {
    if (device_warmup_.joinable()) device_warmup_.join();
    if (cleanup_.joinable()) cleanup_.join();
    if (cleanup2_.joinable()) cleanup2_.join();
    logger_->total("[racon::Polisher::] total =");
}

// ---------------------------------------------------------------- initialize
namespace {
// Parallel ingest (reference src/polisher.cpp:200-349 parses on the calling thread): the inflating thread frames records,
// `threads` workers build the objects (Sequence: upper-casing, quality scan; Overlap: field parsing, CIGAR extents); the
// per-batch vectors are put back into file order afterwards.
template <class T, class Make>
void load_records(const std::string& path, uint32_t threads, std::vector<std::unique_ptr<T>>& dst, Make make) {
    const io::Format format = io::format_of(path);
    std::vector<std::vector<std::unique_ptr<T>>> parts;
    std::mutex m;
    try {
        io::read_batches(path, format, threads, [&](io::Batch& b) {
            FatalThrowsScope scope;             // (a parse worker: reported by the thread that called load_records)
            std::vector<std::unique_ptr<T>> local;
            local.reserve(b.recs.size());
            std::string data, qual;
            for (const auto& rc : b.recs) make(format, b.data() + rc.first, rc.second, data, qual, local);
            std::lock_guard<std::mutex> lock(m);
            if (parts.size() <= b.number) parts.resize(b.number + 1);
            parts[b.number] = std::move(local);
        });
    } catch (const std::exception& e) { fatal(e.what()); }      // (a worker's FatalError included: print + exit or rethrow, per mode)
    size_t n = 0;
    for (const auto& p : parts) n += p.size();
    dst.reserve(dst.size() + n);
    for (auto& p : parts) for (auto& x : p) dst.emplace_back(std::move(x));
}

void load_sequences(const std::string& path, std::vector<std::unique_ptr<Sequence>>& dst, uint32_t threads) {
    load_records<Sequence>(path, threads, dst, [&](io::Format f, const char* s, size_t n, std::string& data, std::string& qual,
                                                   std::vector<std::unique_ptr<Sequence>>& out) {
        io::SeqRecord r;
        io::parse_seq(f, s, n, path, data, qual, r);
        out.emplace_back(r.qual ? new Sequence(r.name, r.name_len, r.data, r.data_len, r.qual, r.qual_len)
                                : new Sequence(r.name, r.name_len, r.data, r.data_len));
    });
}

void load_overlaps(const std::string& path, std::vector<std::unique_ptr<Overlap>>& dst, uint32_t threads) {
    load_records<Overlap>(path, threads, dst, [&](io::Format f, const char* s, size_t n, std::string&, std::string&,
                                                  std::vector<std::unique_ptr<Overlap>>& out) {
        if (f == io::Format::kMhap) { io::MhapRecord r; io::parse_mhap(s, n, path, r); out.emplace_back(new Overlap(r)); }
        else if (f == io::Format::kPaf) { io::PafRecord r; io::parse_paf(s, n, path, r); out.emplace_back(new Overlap(r)); }
        else { io::SamRecord r; if (io::parse_sam(s, n, path, r)) out.emplace_back(new Overlap(r)); }
    });
}
}  // namespace

void Polisher::initialize() {
    if (!windows_.empty()) {
        fprintf(stderr, "[racon::Polisher::initialize] warning: object already initialized!\n");
        return;
    }
    logger_->log();
    // RACON_HIP_TIMING: initialize() by step, milliseconds since it began (the Logger's lines give the stages)
    const bool timing_init = getenv("RACON_HIP_TIMING") != nullptr;
    const auto init_begin = std::chrono::steady_clock::now();
    auto step = [&](const char* what) { if (timing_init) fprintf(stderr, "[racon::Polisher::initialize] timing: %s at %.1f ms\n", what, 1e3 * seconds_since(init_begin)); };
    rank_.clear(); chunks_.clear(); planned_refs_.clear();       // (plans belong to one set of windows)
    if (!device_warmup_.joinable() && engines_.empty() && getenv("RACON_HIP_NO_WARMUP") == nullptr)
        device_warmup_ = std::thread([this] {
            FatalThrowsScope scope;
            try { create_engines(); } catch (const std::exception& e) { engines_error_ = e.what(); engines_.clear(); }
        });

    // The three input files are read concurrently (one inflating thread each, num_threads_ parse workers shared out):
    // their contents only meet below, in file order, when names are resolved.  RACON_HIP_SERIAL_INGEST=1: one file after
    // the other on the calling thread (the reference's order of events, src/polisher.cpp:200-349).
    const bool serial_ingest = getenv("RACON_HIP_SERIAL_INGEST") != nullptr || num_threads_ <= 1;
    const uint32_t parse_threads = serial_ingest ? 1 : std::max<uint32_t>(2, num_threads_ / 2);
    std::vector<std::unique_ptr<Sequence>> reads;
    std::vector<std::unique_ptr<Overlap>> overlaps;
    std::exception_ptr reads_error, overlaps_error;
    std::thread reads_thread, overlaps_thread;
    // Fragment correction is usually run as `racon -f reads overlaps reads`: targets and reads are ONE file.  Every read is then a
    // duplicate of the target of its ordinal ("a read that is also a target is stored once", below) and parsing the file a second
    // time only makes objects to throw away -- for cfg5, 2 GB of FASTQ: the read loop below walks the targets instead.
    const bool reads_are_targets = [&] {
        struct stat a, b;
        return !getenv("RACON_HIP_NO_SAME_FILE") && stat(sequences_path_.c_str(), &a) == 0 && stat(target_path_.c_str(), &b) == 0 && a.st_dev == b.st_dev && a.st_ino == b.st_ino;
    }();
    if (!serial_ingest) {
        if (!reads_are_targets)
        reads_thread = std::thread([&] { FatalThrowsScope scope; try { load_sequences(sequences_path_, reads, parse_threads); } catch (...) { reads_error = std::current_exception(); } });
        overlaps_thread = std::thread([&] { FatalThrowsScope scope; try { load_overlaps(overlaps_path_, overlaps, parse_threads); } catch (...) { overlaps_error = std::current_exception(); } });
    }
    struct Joiner { std::thread& a; std::thread& b; ~Joiner() { if (a.joinable()) a.join(); if (b.joinable()) b.join(); } } joiner{reads_thread, overlaps_thread};

    // ---- targets (reference src/polisher.cpp:200-221)
    load_sequences(target_path_, sequences_, parse_threads);
    const uint64_t targets_size = sequences_.size();
    if (targets_size == 0) fatal("[racon::Polisher::initialize] error: empty target sequences set!");
    std::unordered_map<std::string, uint64_t> name_to_id;     // "<name>t" / "<name>q" -> index in sequences_
    std::unordered_map<uint64_t, uint64_t> id_to_id;          // (ordinal << 1 | is_target) -> index (MHAP numeric ids)
    for (uint64_t i = 0; i < targets_size; ++i) { name_to_id[sequences_[i]->name() + "t"] = i; id_to_id[i << 1 | 1] = i; }
    logger_->log("[racon::Polisher::initialize] loaded target sequences");
    logger_->log();

    // ---- reads; a read that is also a target is stored once (reference src/polisher.cpp:223-278)
    uint64_t sequences_size = 0, total_sequences_length = 0;
    {
        if (reads_are_targets) {}
        else if (serial_ingest) load_sequences(sequences_path_, reads, 1);
        else { reads_thread.join(); if (reads_error) fatal_from(reads_error); }
        const uint64_t n_reads = reads_are_targets ? targets_size : reads.size();
        for (uint64_t k = 0; k < n_reads; ++k) {
            // (one file: read k is the record target k was made from)
            const Sequence& read = reads_are_targets ? *sequences_[k] : *reads[k];
            total_sequences_length += read.data().size();
            const auto it = name_to_id.find(read.name() + "t");
            uint64_t index;
            if (it != name_to_id.end()) {
                const auto& twin = sequences_[it->second];
                if (read.data().size() != twin->data().size() || read.quality().size() != twin->quality().size())
                    fatal("[racon::Polisher::initialize] error: duplicate sequence " + read.name() + " with unequal data");
                index = it->second;
            } else {
                index = sequences_.size();
                sequences_.emplace_back(std::move(reads[k]));
            }
            name_to_id[sequences_[index]->name() + "q"] = index;
            id_to_id[sequences_size << 1 | 0] = index;
            ++sequences_size;
        }
        std::vector<std::unique_ptr<Sequence>>().swap(reads);
    }
    if (sequences_size == 0) fatal("[racon::Polisher::initialize] error: empty sequences set!");
    const WindowType window_type = static_cast<double>(total_sequences_length) / sequences_size <= 1000 ? WindowType::kNGS : WindowType::kTGS;
    logger_->log("[racon::Polisher::initialize] loaded sequences");
    logger_->log();

    // ---- overlaps: resolve ids, then filter each run of consecutive overlaps of one query
    //      (reference src/polisher.cpp:283-358)
    if (serial_ingest) load_overlaps(overlaps_path_, overlaps, 1);
    else { overlaps_thread.join(); if (overlaps_error) fatal_from(overlaps_error); }
    auto filter_group = [&](uint64_t begin, uint64_t end) {
        for (uint64_t i = begin; i < end; ++i) {
            if (!overlaps[i]) continue;
            if (overlaps[i]->error() > error_threshold_ || overlaps[i]->q_id() == overlaps[i]->t_id()) { overlaps[i].reset(); continue; }
            if (type_ != PolisherType::kC) continue;
            // contig mode keeps one overlap per read: the longest, decided by pairwise duels in file order
            for (uint64_t j = i + 1; j < end; ++j) {
                if (!overlaps[j]) continue;
                if (overlaps[i]->length() >= overlaps[j]->length()) overlaps[j].reset();
                else { overlaps[i].reset(); break; }
            }
        }
    };
    {
        uint64_t group = 0;
        for (uint64_t i = 0; i < overlaps.size(); ++i) {
            overlaps[i]->transmute(sequences_, name_to_id, id_to_id);
            if (!overlaps[i]->is_valid()) { overlaps[i].reset(); continue; }
            while (!overlaps[group]) ++group;
            if (overlaps[group]->q_id() != overlaps[i]->q_id()) { filter_group(group, i); group = i; }
        }
        filter_group(group, overlaps.size());
        overlaps.erase(std::remove(overlaps.begin(), overlaps.end(), nullptr), overlaps.end());
    }
    std::vector<bool> has_name(sequences_.size(), false), has_data(sequences_.size(), false), has_reverse_data(sequences_.size(), false);
    for (uint64_t i = 0; i < targets_size; ++i) has_name[i] = has_data[i] = true;
    for (const auto& o : overlaps) { if (o->strand()) has_reverse_data[o->q_id()] = true; else has_data[o->q_id()] = true; }
    std::unordered_map<std::string, uint64_t>().swap(name_to_id);
    std::unordered_map<uint64_t, uint64_t>().swap(id_to_id);
    if (overlaps.empty()) fatal("[racon::Polisher::initialize] error: empty overlap set!");
    logger_->log("[racon::Polisher::initialize] loaded overlaps");
    logger_->log();

    step("overlaps resolved and filtered");
    {
        // Where the windows are built.  RACON_HIP_DEVICE_WINDOWS = 0 (host: Window::add_layer, packed per chunk inside polish()),
        // 1 / 2 / 3 (in HBM: from host breaking points / + the CIGAR walk / + the pairwise alignment), or auto -- what the
        // `racon_hip` binary defaults to (set_default_device_mode): in HBM with the CIGAR walk whenever reads, windows and scratch
        // fit the devices with room to spare, on the host (chunks streamed through the engines, any size) otherwise.
        const char* dv_env = getenv("RACON_HIP_DEVICE_WINDOWS");
        const std::string dv = dv_env ? dv_env : default_device_mode_;
        if (!dv.empty() && dv[0] >= '1' && dv[0] <= '3') device_windows(true, dv[0] == '2', dv[0] == '3');
        else if (dv == "auto" && !keep_layout_) {
            uint64_t read_bases = 0, layer_bases = 0, cigar_bytes = 0;
            for (const auto& s_ : sequences_) read_bases += std::max(s_->data().size(), s_->reverse_complement().size());
            for (const auto& o : overlaps) { layer_bases += o->q_end() - o->q_begin(); cigar_bytes += o->cigar().size(); }
            for (uint64_t i = 0; i < targets_size; ++i) layer_bases += sequences_[i]->data().size();
            // (the warm-up thread is allocating on the devices: what is free is only known once it is done)
            if (device_warmup_.joinable()) device_warmup_.join();
            const int32_t devices = HipEngine::DeviceCount();
            // overlaps without a CIGAR get theirs from the aligner (host or device) after this decision: about one op character per
            // two or three bases of a noisy read -- bounded by half the layer bases
            if (cigar_bytes == 0) cigar_bytes = layer_bases / 2;
            // per device: every read (bases + qualities), its share of the packed windows (x 2 for the sort / gather buffers next to
            // them), CIGAR text, and a scratch arena; against HALF of what is free on the device with the least
            const double need = 2.0 * read_bases + (4.0 * layer_bases + 1.0 * cigar_bytes) / std::max(1, devices) + 24e9;
            uint64_t least_free = ~uint64_t(0);
            for (int32_t d = 0; d < devices; ++d) least_free = std::min(least_free, HipEngine::FreeMemory(d));
            const double have = devices > 0 ? 0.5 * static_cast<double>(least_free) : 0.0;
            if (devices > 0 && need < have) device_windows(true, true, device_align_);
            else if (device_windows_ && need >= have)
                fprintf(stderr, "[racon::Polisher::initialize] warning: the device-side construction needs about %.0f GB per device\n", need / 1e9);
        }
    }
    if (device_align_) {
        // all or nothing per overlap file: SAM records carry CIGARs (nothing to align), PAF / MHAP records do not
        bool any_cigar = false;
        for (const auto& o : overlaps) any_cigar = any_cigar || !o->cigar().empty();
        if (any_cigar) device_align_ = false;
    }
    if (device_windows_) {
        // The device cuts its layers out of the FORWARD strand of every read (it complements on the fly) and the flattened layout below
        // holds nothing else: a reverse complement is only made for the reads the HOST aligner will look at (overlaps on the reverse
        // strand that came without a CIGAR and are not left to the device aligner) -- for cfg3 that is none, and half of the reads
        // (0.75 GB of bases and as many qualities) are not copied backwards for nothing.
        std::vector<bool> host_rc(sequences_.size(), false);
        if (!device_align_) for (const auto& o : overlaps) if (o->strand() && o->cigar().empty()) host_rc[o->q_id()] = true;
        parallel_for(sequences_.size(), num_threads_, [&](uint64_t j) { sequences_[j]->transmute(has_name[j], has_data[j] || has_reverse_data[j], host_rc[j]); });
    } else
    parallel_for(sequences_.size(), num_threads_, [&](uint64_t j) { sequences_[j]->transmute(has_name[j], has_data[j], has_reverse_data[j]); });
    step("sequences transmuted");

    find_overlap_breaking_points(overlaps);
    logger_->log();
    step("breaking points");

    if (keep_layout_) {
        layout_ = Layout();
        layout_.n_targets = targets_size;
        layout_.window_type = window_type == WindowType::kTGS ? 1 : 0;
        // The forward strand of every sequence, flat (transmute() may have kept only the reverse complement: complementing again
        // gives it back, the table of Sequence::create_reverse_complement is an involution).  Offsets first, then the bytes by all
        // host threads into buffers sized once: grown by insert() on one thread this was most of the second that "transformed data
        // into windows (on the device)" took at 50 000 windows -- the device's own share of it is 80 ms.
        const uint64_t n_seq = sequences_.size();
        layout_.seq_off.assign(n_seq + 1, 0);
        layout_.seq_has_qual.assign(n_seq, 0);
        for (uint64_t i = 0; i < n_seq; ++i) {
            const auto& sq = sequences_[i];
            const bool rev_only = sq->data().empty() && !sq->reverse_complement().empty();
            layout_.seq_off[i + 1] = layout_.seq_off[i] + (rev_only ? sq->reverse_complement().size() : sq->data().size());
            layout_.seq_has_qual[i] = (rev_only ? !sq->reverse_quality().empty() : !sq->quality().empty()) ? 1 : 0;
        }
        layout_.bases.resize(layout_.seq_off[n_seq]);
        layout_.quals.resize(layout_.seq_off[n_seq]);
        // (gigabytes about to be touched for the first time: in 2 MB pages where the kernel hands them out on request --
        //  transparent_hugepage = madvise -- the fill below is a copy, not 750 000 page faults per 3 GB)
        auto huge = [](void* p, size_t n) {
            const uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 4095) & ~uintptr_t(4095), z = (reinterpret_cast<uintptr_t>(p) + n) & ~uintptr_t(4095);
            if (z > a + (8u << 20)) (void)madvise(reinterpret_cast<void*>(a), z - a, MADV_HUGEPAGE);
        };
        huge(layout_.bases.data(), layout_.bases.size()); huge(layout_.quals.data(), layout_.quals.size());
        parallel_for(n_seq, num_threads_, [&](uint64_t i) {
            static const struct Comp { uint8_t t[256]; Comp() { for (int k = 0; k < 256; ++k) t[k] = static_cast<uint8_t>(k); t['A'] = 'T'; t['T'] = 'A'; t['C'] = 'G'; t['G'] = 'C'; } } comp;
            const auto& sq = sequences_[i];
            uint8_t* db = layout_.bases.data() + layout_.seq_off[i];
            uint8_t* dq = layout_.quals.data() + layout_.seq_off[i];
            const uint64_t n = layout_.seq_off[i + 1] - layout_.seq_off[i];
            if (sq->data().empty() && !sq->reverse_complement().empty()) {
                const std::string& rc = sq->reverse_complement();
                for (uint64_t k = 0; k < n; ++k) db[k] = comp.t[static_cast<uint8_t>(rc[n - 1 - k])];
                const std::string& rq = sq->reverse_quality();
                if (!rq.empty()) for (uint64_t k = 0; k < n; ++k) dq[k] = static_cast<uint8_t>(rq[n - 1 - k]);
                else std::memset(dq, '!', n);
            } else {
                std::memcpy(db, sq->data().data(), n);
                if (!sq->quality().empty()) std::memcpy(dq, sq->quality().data(), n); else std::memset(dq, '!', n);
            }
        });
        const uint64_t n_ovl_all = overlaps.size();
        layout_.cigar_off.assign(n_ovl_all + 1, 0);
        for (uint64_t k = 0; k < n_ovl_all; ++k) layout_.cigar_off[k + 1] = layout_.cigar_off[k] + overlaps[k]->cigar().size();
        layout_.cigar.resize(layout_.cigar_off[n_ovl_all]);
        huge(layout_.cigar.data(), layout_.cigar.size());
        parallel_for(n_ovl_all, num_threads_, [&](uint64_t k) {
            const std::string& cg = overlaps[k]->cigar();
            if (!cg.empty()) std::memcpy(layout_.cigar.data() + layout_.cigar_off[k], cg.data(), cg.size());
        });
        for (auto* v : {&layout_.q_id, &layout_.t_id, &layout_.q_start, &layout_.t_begin, &layout_.t_end, &layout_.q_begin, &layout_.q_end}) v->reserve(n_ovl_all);
        layout_.strand.reserve(n_ovl_all); layout_.bp_off.reserve(n_ovl_all + 1);
        for (const auto& o : overlaps) {
            layout_.q_id.push_back(static_cast<uint32_t>(o->q_id()));
            layout_.t_id.push_back(static_cast<uint32_t>(o->t_id()));
            layout_.strand.push_back(o->strand() ? 1 : 0);
            if (!device_cigars_) for (const auto& bp : o->breaking_points()) { layout_.bp_t.push_back(bp.first); layout_.bp_q.push_back(bp.second); }
            layout_.bp_off.push_back(layout_.bp_t.size());
            layout_.q_start.push_back(o->q_start_on_strand()); layout_.t_begin.push_back(o->t_begin()); layout_.t_end.push_back(o->t_end());
            layout_.q_begin.push_back(o->q_begin()); layout_.q_end.push_back(o->q_end());
        }
    }

    if (keep_layout_) step("layout flattened");
    // ---- windows over every target (reference src/polisher.cpp:388-403)
    std::vector<uint64_t> first_window(targets_size + 1, 0);
    for (uint64_t i = 0; i < targets_size; ++i) {
        const std::string& data = sequences_[i]->data();
        const std::string& quality = sequences_[i]->quality();
        uint32_t k = 0;
        for (uint32_t j = 0; j < data.size(); j += window_length_, ++k) {
            const uint32_t length = std::min(j + window_length_, static_cast<uint32_t>(data.size())) - j;
            windows_.emplace_back(createWindow(i, k, window_type, &data[j], length,
                                               quality.empty() ? &dummy_quality_[0] : &quality[j], length));
        }
        first_window[i + 1] = first_window[i] + k;
    }

    // ---- layers (reference src/polisher.cpp:405-461): serial, in overlap order
    targets_coverages_.assign(targets_size, 0);
    step("windows created");
    if (device_windows_) {                      // the layers are cut on the device, from layout_ (polish())
        for (auto& o : overlaps) { ++targets_coverages_[o->t_id()]; o.reset(); }
        // the reads now live in layout_ only (the targets stay: windows_ point into their backbones)
        for (uint64_t i = targets_size; i < sequences_.size(); ++i) sequences_[i]->release_data();
        step("overlaps and reads released");
        build_device_windows();
        step("device windows built");
        logger_->log(device_built_ ? "[racon::Polisher::initialize] transformed data into windows (on the device)"
                                   : "[racon::Polisher::initialize] transformed data into windows");
        return;
    }
    for (auto& o : overlaps) {
        ++targets_coverages_[o->t_id()];
        const auto& sequence = sequences_[o->q_id()];
        const auto& bp = o->breaking_points();
        const bool rev = o->strand() != 0;
        const std::string& bases = rev ? sequence->reverse_complement() : sequence->data();
        const std::string& quality = rev ? sequence->reverse_quality() : sequence->quality();
        const bool read_has_quality = !sequence->quality().empty() || !sequence->reverse_quality().empty();
        for (uint32_t j = 0; j + 1 < bp.size(); j += 2) {
            const uint32_t q0 = bp[j].second, q1 = bp[j + 1].second;
            if (q1 - q0 < 0.02 * window_length_) continue;
            if (read_has_quality) {
                double average_quality = 0;
                for (uint32_t k = q0; k < q1; ++k) average_quality += static_cast<uint32_t>(quality[k]) - 33;
                average_quality /= q1 - q0;
                if (average_quality < quality_threshold_) continue;
            }
            const uint32_t window_rank = bp[j].first / window_length_;
            const uint32_t window_start = window_rank * window_length_;
            const char* q = quality.empty() ? nullptr : &quality[q0];
            windows_[first_window[o->t_id()] + window_rank]->add_layer(&bases[q0], q1 - q0, q, q ? q1 - q0 : 0,
                bp[j].first - window_start, bp[j + 1].first - window_start - 1);
        }
        o.reset();
    }
    logger_->log("[racon::Polisher::initialize] transformed data into windows");
    reserve_for_windows();
}

// ---------------------------------------------------------------- engines
// One engine = one device + its streams (a CUDABatchProcessor, reference src/cuda/cudabatch.hpp:27-122).  Two per batch
// object and device: the kernel of one chunk fills the compute units that the tail of the other engine's chunk leaves
// idle, while a host thread packs the next one.  Created with a first reservation for `-w` sized windows at ONT-like
// depth; reserve_for_windows() corrects it once the windows exist.
namespace { constexpr uint64_t kMaxChunkWindows = 8192, kMinChunkWindows = 2048, kMaxChunkBases = 512ull << 20, kMinChunkBases = 64ull << 20; }

void Polisher::create_engines() {
    const int32_t real_devices = HipEngine::DeviceCount();      // (0 without the library or a device: polish() reports that)
    n_devices_ = real_devices;
    if (real_devices <= 0) return;
    // RACON_HIP_FAKE_DEVICES=n: drive n logical devices (engine k belongs to logical device k mod n, which is physical device
    // (k mod n) mod the real count): the multi-device code paths on a one-GPU box (tests)
    if (const char* fd = getenv("RACON_HIP_FAKE_DEVICES")) n_devices_ = std::max(1, atoi(fd));
    uint32_t engines_per_device = 2 * hip_batches_;
    if (const char* ed = getenv("RACON_HIP_ENGINES_PER_DEVICE")) engines_per_device = static_cast<uint32_t>(std::max(1, atoi(ed)));   // experiments
    const uint32_t sharing = engines_per_device * static_cast<uint32_t>((n_devices_ + real_devices - 1) / real_devices);
    std::vector<std::shared_ptr<HipEngine>> engines;
    for (uint32_t k = 0; k < static_cast<uint32_t>(n_devices_) * engines_per_device; ++k) {
        const int32_t device = static_cast<int32_t>(k % static_cast<uint32_t>(n_devices_)) % real_devices;
        // engines of one device split its free HBM (each would otherwise budget 80 % of it for its own scratch)
        const uint64_t arena = static_cast<uint64_t>(HipEngine::FreeMemory(device) * 0.8 / sharing);
        engines.emplace_back(HipEngine::Create(device, match_, mismatch_, gap_, arena));
    }
    // first use of the code object, the streams' queues and the copy engines; the arenas and the pinned staging are sized
    // once the windows exist (reserve_for_windows): device memory that is allocated, freed and allocated again is what
    // costs (the driver clears it), so nothing is allocated on a guess
    for (auto& e : engines) e->reserve(0, 0, 0, window_length_, 0);
    engines_.swap(engines);
}

// The work list of polish(): chunks of the window index space in DEEPEST-FIRST order (see polish()).
void Polisher::plan_chunks() {
    const uint64_t nw = windows_.size();
    const uint64_t n_engines = std::max<size_t>(1, engines_.size());
    if (nw > 0xffffffffull) fatal("[racon::Polisher::polish] error: more than 2^32 windows!");
    rank_.resize(nw);
    chunks_.clear();
    std::vector<uint64_t> cost(nw), bases(nw);
    for (uint64_t i = 0; i < nw; ++i) {
        uint64_t b = 0;
        for (const auto& sq : windows_[i]->sequences_) b += sq.second;
        bases[i] = b;
        cost[i] = windows_[i]->sequences_.size() < 3 ? 0 : b * windows_[i]->sequences_.size();
        rank_[i] = static_cast<uint32_t>(i);
    }
    std::stable_sort(rank_.begin(), rank_.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    // three chunks per two engines (two engines: thirds).  Sweep on one GPU's share of cfg3, 12 500 windows, polish() of the binary
    // (profiles/r03/j_chunksweep.txt): chunks of 1600 / 2100 / 3200 / 4200 / 6300 windows 111 / 110 / 98 / 92 / 99 ms with two
    // engines, 108 / 104 / 98 / 95 / 95 with three, 116 / 113 / 96 / 95 / 94 with four: fewer, larger chunks -- every chunk's launch
    // has a tail of its own -- and the number of engines beyond two does not matter
    uint64_t target = std::max(kMinChunkWindows, std::min(kMaxChunkWindows, (2 * nw + 3 * n_engines - 1) / (3 * n_engines)));
    // (at least 64 MB of bases too -- what 2048 windows of 500 bases at 30x hold: short-read windows are a third of that,
    //  and a chunk's fixed costs -- packing before its first launch, its launch tail -- do not shrink with them:
    //  5000 windows of 150-base reads went as 2048 + 2048 + 904, the last chunk alone on the device for 6 of 23 ms.
    //  But never so much that an engine is left without a chunk: the floor is capped at an even share of the job.)
    uint64_t total_bases = 0;
    for (uint64_t i = 0; i < nw; ++i) total_bases += bases[i];
    uint64_t floor_bases = std::min<uint64_t>(kMinChunkBases, total_bases / n_engines);
    // RACON_HIP_CHUNK_WINDOWS=n (tests, sweeps): chunks of exactly n windows, no floor
    if (const char* cw = getenv("RACON_HIP_CHUNK_WINDOWS")) { if (atoi(cw) > 0) { target = static_cast<uint64_t>(atoi(cw)); floor_bases = 0; } }
    for (uint64_t a = 0; a < nw;) {
        uint64_t b = a, sum = 0;
        while (b < nw && (b - a < target || sum < floor_bases) && b - a < 4 * kMaxChunkWindows && sum < kMaxChunkBases) sum += bases[rank_[b++]];
        chunks_.emplace_back(a, b); a = b;
    }
}

void Polisher::reserve_for_windows() {
    if (device_warmup_.joinable()) device_warmup_.join();
    if (engines_.empty() || windows_.empty() || device_windows_) return;
    const bool timing = getenv("RACON_HIP_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    plan_chunks();
    // engine k takes chunk k first (the shared cursor of polish() hands them out in this order): each gets exactly what that
    // chunk needs -- later chunks are shallower and fit in the same buffers -- and an engine without a chunk gets nothing
    // (one thread per engine: pinned staging costs 0.23 ms per MB and an arena can wait a second for the driver's page
    //  clearing -- sixteen engines on an eight-GPU node must not pay that one after the other)
    planned_refs_.assign(std::min(engines_.size(), chunks_.size()), WindowRefs());
    std::vector<std::string> errors(planned_refs_.size());
    auto reserve_one = [&](size_t k) {
        FatalThrowsScope scope;
        try {
            WindowRefs& refs = planned_refs_[k];
            for (uint64_t i = chunks_[k].first; i < chunks_[k].second; ++i) refs.add(*windows_[rank_[i]]);
            engines_[k]->reserve(refs, chunks_.size() > 1);
        } catch (const std::exception& e) { errors[k] = e.what(); }
    };
    if (planned_refs_.size() == 1) reserve_one(0);
    else {
        std::vector<std::thread> pool;
        for (size_t k = 0; k < planned_refs_.size(); ++k) pool.emplace_back(reserve_one, k);
        for (auto& t : pool) t.join();
    }
    // a reservation that failed (a device short of memory at this moment) is not an error yet: polish() sizes its calls as it
    // goes and halves a chunk the device has no room for
    for (const auto& e : errors) if (!e.empty()) {
        fprintf(stderr, "[racon::Polisher::initialize] warning: could not reserve device buffers ahead of polish() (%s)\n", e.c_str());
        planned_refs_.clear();
        break;
    }
    if (timing) fprintf(stderr, "[racon::Polisher::initialize] timing: %zu chunk(s) planned, engines reserved in %.1f ms\n", chunks_.size(), 1e3 * seconds_since(t0));
}

void Polisher::find_overlap_breaking_points(std::vector<std::unique_ptr<Overlap>>& overlaps) {
    parallel_for(overlaps.size(), num_threads_, [&](uint64_t j) { overlaps[j]->find_breaking_points(sequences_, window_length_, keep_layout_, device_cigars_, device_align_); });
    logger_->log(device_align_ ? "[racon::Polisher::initialize] left the overlaps to the device aligner" : "[racon::Polisher::initialize] aligned overlaps");
}

// ---------------------------------------------------------------- windows built on the device
// End of initialize() with device-side construction: where the reference cuts its windows (src/polisher.cpp:388-461) -- and, on
// its GPU path, aligns its overlaps (CUDAPolisher::find_overlap_breaking_points, called from initialize()) -- the engines do
// the same in HBM and keep the windows resident; polish() is then the consensus alone.  Possible when every shard has an engine
// of its own (shards <= devices); RACON_HIP_BUILD_IN_POLISH=1 keeps everything in polish() (the round-4 behaviour, experiments).
void Polisher::build_device_windows() {
    device_built_ = false;
    device_plan_ = DevicePlan();
    if (getenv("RACON_HIP_BUILD_IN_POLISH")) return;
    if (device_warmup_.joinable()) device_warmup_.join();
    if (!engines_error_.empty() || engines_.empty() || windows_.empty()) return;        // polish() reports what is wrong
    const int32_t n_devices = n_devices_ > 0 ? n_devices_ : HipEngine::DeviceCount();
    const uint32_t n_shards = device_shards();
    if (n_devices <= 0 || n_shards > static_cast<uint32_t>(n_devices) || static_cast<size_t>(n_devices) > engines_.size()) return;
    FatalThrowsScope scope;
    try {
        device_job(1, nullptr, nullptr, nullptr);
        device_built_ = true;
    } catch (const std::exception& e) {
        // No room at this moment, or an input the device aligner cannot take: polish() builds shard by shard, with its fallbacks --
        // and, when it was memory, in MORE and smaller shards than devices, one after the other (the host never ran add_layer for
        // this job and the reads are gone: cutting the job finer is the way down, not the host-built path)
        fprintf(stderr, "[racon::Polisher::initialize] warning: windows not built ahead of polish() (%s)\n", e.what());
        device_plan_ = DevicePlan();
        bool memory = false;
        for (const auto& eng : engines_) memory = memory || eng->last_rc() == RCN_E_NOMEM || eng->last_rc() == RCN_E_CAPACITY;
        if (memory) device_min_shards_ = std::max<uint32_t>(2 * n_shards, 2);
    }
}

// ---------------------------------------------------------------- polish
void Polisher::pack_windows(PackedBatch* out) const {
    out->clear();
    for (const auto& w : windows_) out->add(*w);
}

void Polisher::assemble(const std::function<const std::string&(uint64_t)>& consensus, const std::function<bool(uint64_t)>& polished,
                        std::vector<std::unique_ptr<Sequence>>& dst, bool drop_unpolished_sequences) {
    assemble_views([&](uint64_t i) { const std::string& c = consensus(i); return ConsensusView(c.data(), c.size()); }, polished, dst, drop_unpolished_sequences);
}

void Polisher::assemble_views(const std::function<ConsensusView(uint64_t)>& consensus, const std::function<bool(uint64_t)>& polished,
                              std::vector<std::unique_ptr<Sequence>>& dst, bool drop_unpolished_sequences) {
    std::string polished_data;
    uint32_t num_polished_windows = 0;
    for (uint64_t i = 0; i < windows_.size(); ++i) {
        if (windows_[i]->rank() == 0) {
            // (one allocation per target, moved into the Sequence: a megabase grown by doubling and then copied is its pages touched twice)
            uint64_t total = 0;
            for (uint64_t k = i; k < windows_.size() && (k == i || windows_[k]->rank() != 0); ++k) total += consensus(k).second;
            polished_data.reserve(total);
        }
        num_polished_windows += polished(i) ? 1 : 0;
        { const ConsensusView c = consensus(i); polished_data.append(c.first, c.second); }
        if (i == windows_.size() - 1 || windows_[i + 1]->rank() == 0) {       // last window of this target
            const double polished_ratio = num_polished_windows / static_cast<double>(windows_[i]->rank() + 1);
            if (!drop_unpolished_sequences || polished_ratio > 0) {
                std::string tags = type_ == PolisherType::kF ? "r" : "";
                tags += " LN:i:" + std::to_string(polished_data.size());
                tags += " RC:i:" + std::to_string(targets_coverages_[windows_[i]->id()]);
                tags += " XC:f:" + std::to_string(polished_ratio);
                dst.emplace_back(createSequence(sequences_[windows_[i]->id()]->name() + tags, std::move(polished_data)));
            }
            num_polished_windows = 0;
            polished_data = std::string();
        }
    }
    // The reference frees every window and every sequence here, inside the interval its Logger brackets
    // (src/polisher.cpp:532,545-546).  For one GPU's share of cfg3 that is ~400 MB in ~40 000 heap blocks -- 30 ms of
    // free() in a 110 ms polish() -- and nobody waits for it: a helper thread does it while the caller goes on with the
    // polished sequences (joined by the destructor / the next call).
    if (cleanup_.joinable()) cleanup_.join();
    auto* old_windows = new std::vector<std::shared_ptr<Window>>(std::move(windows_));
    auto* old_sequences = new std::vector<std::unique_ptr<Sequence>>(std::move(sequences_));
    // (the work list and the pointer tables planned for these windows go the same way: a short-read job's tables are tens of MB)
    auto* old_refs = new std::vector<WindowRefs>(std::move(planned_refs_));
    auto* old_rank = new std::vector<uint32_t>(std::move(rank_));
    windows_.clear(); sequences_.clear(); planned_refs_.clear(); rank_.clear(); chunks_.clear();
    cleanup_ = std::thread([old_windows, old_sequences, old_refs, old_rank] { delete old_windows; delete old_sequences; delete old_refs; delete old_rank; });
}

void Polisher::polish(std::vector<std::unique_ptr<Sequence>>& dst, bool drop_unpolished_sequences) {
    logger_->log();
    const auto polish_begin = std::chrono::steady_clock::now();
    struct Stamp { const std::chrono::steady_clock::time_point& t0; double& out; ~Stamp() { out = seconds_since(t0); } } stamp{polish_begin, polish_seconds_};
    if (device_warmup_.joinable()) device_warmup_.join();
    if (!engines_error_.empty()) fatal(engines_error_);
    if (engines_.empty()) {                         // the warm-up was switched off, or initialize() was not called
        std::exception_ptr error;
        { FatalThrowsScope scope; try { create_engines(); } catch (...) { error = std::current_exception(); } }
        if (error) fatal_from(error);
    }
    const int32_t n_devices = n_devices_ > 0 ? n_devices_ : HipEngine::DeviceCount();
    if (n_devices <= 0 || engines_.empty())
        fatal("[racon::Polisher::polish] error: no MI355X device / libracon_hip.so available (the consensus stage has no CPU fallback)!");

    const uint64_t nw = windows_.size();
    const uint32_t n_engines = static_cast<uint32_t>(engines_.size());
    std::vector<std::string> cons(nw);
    std::vector<uint8_t> pol(nw, 0), chim(nw, 0);
    if (device_windows_) {
        // windows built in HBM (device_job): by initialize() already where every shard has an engine of its own -- polish() is then
        // the consensus of resident windows --, otherwise built and polished here, shard after shard
        const bool timing_d = getenv("RACON_HIP_TIMING") != nullptr;
        if (timing_d) fprintf(stderr, "[racon::Polisher::polish] timing: engines ready, result arrays made at %.2f ms\n", 1e3 * seconds_since(polish_begin));
        // (resident windows, one shard per engine: no string per window, the targets are assembled from the engines' result blocks)
        if (device_built_) { cons_views_.assign(nw, ConsensusView(nullptr, 0)); device_job(2, &cons, &pol, &chim); }
        else {
            // shard after shard inside polish(); a shard the device has no room for cuts the job finer (twice, four times ... the shards)
            for (;;) {
                std::string error;
                { FatalThrowsScope scope; try { device_job(0, &cons, &pol, &chim); } catch (const std::exception& e) { error = e.what(); } }
                if (error.empty()) break;
                bool memory = false;
                for (const auto& eng : engines_) memory = memory || eng->last_rc() == RCN_E_NOMEM || eng->last_rc() == RCN_E_CAPACITY;
                const uint32_t now = device_shards();
                if (!memory || now >= 1024 || now >= nw) fatal(error);
                device_min_shards_ = 2 * now;
                fprintf(stderr, "[racon::Polisher::polish] warning: no room on the device for a shard of the job (%s): cutting it into %u shards\n", error.c_str(), device_min_shards_);
            }
        }
        for (uint64_t i = 0; i < nw; ++i)
            if (chim[i]) fprintf(stderr, "[racon::Window::generate_consensus] warning: contig %lu might be chimeric in window %u!\n",
                                 static_cast<unsigned long>(windows_[i]->id()), windows_[i]->rank());
        if (timing_d) fprintf(stderr, "[racon::Polisher::polish] timing: windows polished at %.2f ms\n", 1e3 * seconds_since(polish_begin));
        // (a window with a view has no string and the other way round: a lane with several shards on its engine keeps strings)
        assemble_views([&](uint64_t i) { return (i < cons_views_.size() && cons_views_[i].first) ? cons_views_[i] : ConsensusView(cons[i].data(), cons[i].size()); },
                       [&](uint64_t i) { return pol[i] != 0; }, dst, drop_unpolished_sequences);
        cons_views_.clear();
        if (timing_d) fprintf(stderr, "[racon::Polisher::polish] timing: assembled at %.2f ms\n", 1e3 * seconds_since(polish_begin));
        logger_->log("[racon::Polisher::polish] generated consensus");
        return;
    }
    // ---- host-built windows: chunks of the window index space in DEEPEST-FIRST order, pulled from a shared cursor ----
    // (reference src/cuda/cudapolisher.cpp:254-276 hands out ranges under a mutex the same way.)  Results go back by
    // window index, so the order windows are polished in is free -- and a queue of unequal jobs ends soonest when the long
    // ones start first: the windows are ranked by the engine's own cost proxy (sequences x bases), chunk k is ranks
    // [k C, (k+1) C).  The deepest windows are under way in the first launch, the last chunk holds the shallowest ones and
    // its tail is short.  A job that fits one chunk (cfg2: 2000 windows) is one engine call with the windows resident
    // all at once; larger jobs alternate between the engines of a device, each chunk flagged as part of a queue.
    if (chunks_.empty() || rank_.size() != nw) plan_chunks();
    const std::vector<uint32_t>& rank = rank_;
    const std::vector<std::pair<uint64_t, uint64_t>>& chunks = chunks_;
    const bool timing = getenv("RACON_HIP_TIMING") != nullptr;
    const bool queued = chunks.size() > 1;
    const uint32_t n_workers = static_cast<uint32_t>(std::min<size_t>(n_engines, std::max<size_t>(1, chunks.size())));
    std::atomic<size_t> cursor{n_workers};          // engine k starts with chunk k (what reserve_for_windows sized it for)
    std::vector<std::exception_ptr> errors(n_engines);
    std::vector<uint32_t> taken(n_engines, 0);      // chunks each engine took (rcnh_polisher_polish_plan: the tests assert the plan)
    auto worker = [&](uint32_t k) {
        FatalThrowsScope scope;
        try {
            auto& engine = engines_[k];
            WindowRefs refs;
            std::vector<std::string> c; std::vector<uint8_t> p, h;
            // A chunk the device has no room for (RCN_E_NOMEM / RCN_E_CAPACITY: other engines, reads and overlaps share its
            // memory) is polished in halves, on the GPU: the reference completes its run when a batch fails
            // (src/cuda/cudapolisher.cpp:357-373 redoes those windows); there is no CPU path to fall back to here.
            std::function<void(uint64_t, uint64_t)> polish_range = [&](uint64_t a, uint64_t b) {
                refs.clear();
                for (uint64_t i = a; i < b; ++i) refs.add(*windows_[rank[i]]);
                engine->set_verify_ids([&rank, a](uint32_t j) { return static_cast<uint64_t>(rank[a + j]); });
                try {
                    engine->consensus(refs, queued, trim_, &c, &p, &h);
                } catch (const FatalError&) {
                    if ((engine->last_rc() != RCN_E_NOMEM && engine->last_rc() != RCN_E_CAPACITY) || b - a < 2) throw;
                    fprintf(stderr, "[racon::Polisher::polish] warning: no room on the device for %lu windows at once, polishing them in halves\n", static_cast<unsigned long>(b - a));
                    const uint64_t mid = a + (b - a) / 2;
                    polish_range(a, mid); polish_range(mid, b);
                    return;
                }
                for (uint64_t i = a, j = 0; i < b; ++i, ++j) { const uint32_t w = rank[i]; cons[w].swap(c[j]); pol[w] = p[j]; chim[w] = h[j]; }
            };
            for (size_t ci = k; ci < chunks.size(); ci = cursor.fetch_add(1)) {
                const double t_a = seconds_since(polish_begin);
                const bool planned = ci < planned_refs_.size() && planned_refs_[ci].n_windows() == chunks[ci].second - chunks[ci].first;
                const double t_b = seconds_since(polish_begin);
                bool done = false;
                if (planned) {
                    try {
                        engine->set_verify_ids([&rank, a0 = chunks[ci].first](uint32_t j) { return static_cast<uint64_t>(rank[a0 + j]); });
                        engine->consensus(planned_refs_[ci], queued, trim_, &c, &p, &h);
                        for (uint64_t i = chunks[ci].first, j = 0; i < chunks[ci].second; ++i, ++j) { const uint32_t w = rank[i]; cons[w].swap(c[j]); pol[w] = p[j]; chim[w] = h[j]; }
                        done = true;
                    } catch (const FatalError&) {
                        if (engine->last_rc() != RCN_E_NOMEM && engine->last_rc() != RCN_E_CAPACITY) throw;
                    }
                }
                if (!done) polish_range(chunks[ci].first, chunks[ci].second);
                ++taken[k];
                const double t_c = seconds_since(polish_begin);
                if (timing) fprintf(stderr, "[racon::Polisher::polish] timing: engine %u chunk %zu (%lu windows): start %.1f ms, refs %.1f, engine done %.1f (kernel %.1f), stored %.1f\n",
                                    k, ci, static_cast<unsigned long>(chunks[ci].second - chunks[ci].first), 1e3 * t_a, 1e3 * t_b, 1e3 * t_c, engine->last_kernel_ms(), 1e3 * seconds_since(polish_begin));
            }
        } catch (...) { errors[k] = std::current_exception(); cursor.store(chunks.size()); }
    };
    {
        if (n_workers <= 1) worker(0);
        else {
            std::vector<std::thread> pool;
            for (uint32_t k = 0; k < n_workers; ++k) pool.emplace_back(worker, k);
            for (auto& t : pool) t.join();
        }
    }
    for (const auto& e : errors) if (e) fatal_from(e);
    if (timing) fprintf(stderr, "[racon::Polisher::polish] timing: engines done at %.2f ms\n", 1e3 * seconds_since(polish_begin));
    polish_chunks_ = static_cast<uint32_t>(chunks.size());
    polish_engines_used_ = 0;
    for (uint32_t n : taken) polish_engines_used_ += n > 0 ? 1 : 0;
    for (uint64_t i = 0; i < nw; ++i)
        if (chim[i]) fprintf(stderr, "[racon::Window::generate_consensus] warning: contig %lu might be chimeric in window %u!\n",
                             static_cast<unsigned long>(windows_[i]->id()), windows_[i]->rank());
    assemble([&](uint64_t i) -> const std::string& { return cons[i]; }, [&](uint64_t i) { return pol[i] != 0; },
             dst, drop_unpolished_sequences);
    if (timing) fprintf(stderr, "[racon::Polisher::polish] timing: assembled at %.2f ms\n", 1e3 * seconds_since(polish_begin));
    {
        // (2000 ... 100 000 result strings: freed next to the windows, not inside the interval)
        auto* old_cons = new std::vector<std::string>(std::move(cons));
        if (cleanup2_.joinable()) cleanup2_.join();
        cleanup2_ = std::thread([old_cons] { delete old_cons; });
    }
    logger_->log("[racon::Polisher::polish] generated consensus");
}

}  // namespace racon
