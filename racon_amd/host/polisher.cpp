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

#include "fatal.hpp"
#include "hip_engine.hpp"
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

namespace {
double seconds_since(const std::chrono::time_point<std::chrono::steady_clock>& t) {
    return std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - t).count();
}

// fn(i) for i in [0, n) on `threads` host threads (dynamic distribution)
template <class F>
void parallel_for(uint64_t n, uint32_t threads, F fn) {
    threads = std::max<uint32_t>(1, std::min<uint64_t>(threads, n));
    if (threads == 1) { for (uint64_t i = 0; i < n; ++i) fn(i); return; }
    // an exception inside a worker (fatal() in library mode, nw_path's runtime_error, bad_alloc) must not escape its thread
    // (std::terminate): the first one is kept and rethrown on the caller's thread after the join
    std::atomic<uint64_t> next{0};
    std::exception_ptr first_error;
    std::mutex error_mutex;
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < threads; ++t)
        pool.emplace_back([&] {
            FatalThrowsScope scope;
            try {
                for (uint64_t i; (i = next.fetch_add(1)) < n;) fn(i);
            } catch (...) {
                std::lock_guard<std::mutex> lock(error_mutex);
                if (!first_error) first_error = std::current_exception();
                next.store(n);                                     // the other workers stop at their next item
            }
        });
    for (auto& t : pool) t.join();
    if (first_error) fatal_from(first_error);
}
}  // namespace

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
            for (const auto& rc : b.recs) make(format, b.text.data() + rc.first, rc.second, data, qual, local);
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
    if (!serial_ingest) {
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
        if (serial_ingest) load_sequences(sequences_path_, reads, 1);
        else { reads_thread.join(); if (reads_error) fatal_from(reads_error); }
        for (auto& read : reads) {
            total_sequences_length += read->data().size();
            const auto it = name_to_id.find(read->name() + "t");
            uint64_t index;
            if (it != name_to_id.end()) {
                const auto& twin = sequences_[it->second];
                if (read->data().size() != twin->data().size() || read->quality().size() != twin->quality().size())
                    fatal("[racon::Polisher::initialize] error: duplicate sequence " + read->name() + " with unequal data");
                index = it->second;
            } else {
                index = sequences_.size();
                sequences_.emplace_back(std::move(read));
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

    parallel_for(sequences_.size(), num_threads_, [&](uint64_t j) { sequences_[j]->transmute(has_name[j], has_data[j], has_reverse_data[j]); });

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
            const int32_t devices = HipEngine::DeviceCount();
            // per device: every read (bases + qualities), its share of the packed windows (x 2 for the sort / gather buffers next to
            // them), CIGAR text, and a scratch arena; against HALF of what is free on device 0
            const double need = 2.0 * read_bases + (4.0 * layer_bases + 1.0 * cigar_bytes) / std::max(1, devices) + 24e9;
            const double have = devices > 0 ? 0.5 * static_cast<double>(HipEngine::FreeMemory(0)) : 0.0;
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
    find_overlap_breaking_points(overlaps);
    logger_->log();

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
    if (device_windows_) {                      // the layers are cut on the device, from layout_ (polish())
        for (auto& o : overlaps) { ++targets_coverages_[o->t_id()]; o.reset(); }
        // the reads now live in layout_ only (the targets stay: windows_ point into their backbones)
        for (uint64_t i = targets_size; i < sequences_.size(); ++i) sequences_[i]->release_data();
        build_device_windows();
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
    device_cut_.clear();
    if (getenv("RACON_HIP_BUILD_IN_POLISH")) return;
    if (device_warmup_.joinable()) device_warmup_.join();
    if (!engines_error_.empty() || engines_.empty() || windows_.empty()) return;        // polish() reports what is wrong
    const int32_t n_devices = n_devices_ > 0 ? n_devices_ : HipEngine::DeviceCount();
    uint32_t n_shards = static_cast<uint32_t>(std::max(1, n_devices));
    if (const char* sh = getenv("RACON_HIP_DEVICE_SHARDS")) n_shards = std::max(1, atoi(sh));
    if (n_devices <= 0 || n_shards > static_cast<uint32_t>(n_devices) || static_cast<size_t>(n_devices) > engines_.size()) return;
    FatalThrowsScope scope;
    try {
        device_job(1, nullptr, nullptr, nullptr);
        device_built_ = true;
    } catch (const std::exception& e) {
        // (no room at this moment, an input the device aligner cannot take: polish() builds shard by shard, with its fallbacks)
        fprintf(stderr, "[racon::Polisher::initialize] warning: windows not built ahead of polish() (%s)\n", e.what());
        device_cut_.clear();
    }
}

// One engine per shard builds its windows in HBM from the reads and the overlaps' breaking points / CIGARs / segment pairs
// (rcn_engine_build_windows*: reference src/polisher.cpp:388-461, src/overlap.cpp:176-292 on the device) and polishes them there.
//   phase 1  build only: the end of initialize(), where the reference builds its windows (and, with --cudaaligner-batches, aligns
//            its overlaps: CUDAPolisher::find_overlap_breaking_points is called from initialize()); the windows stay resident and
//            the engine reserves what its run will need (rcn_engine_reserve_run)
//   phase 2  run only: polish() of windows built by phase 1
//   phase 0  both, shard after shard on its device's engine: polish() when the shards outnumber the engines that could keep them
//            resident (RACON_HIP_DEVICE_SHARDS > devices: a job cut into pieces to fit one device), or when phase 1 found no room
void Polisher::device_job(int phase, std::vector<std::string>* cons_out, std::vector<uint8_t>* pol_out, std::vector<uint8_t>* chim_out) {
    const int32_t n_devices = n_devices_ > 0 ? n_devices_ : HipEngine::DeviceCount();
    const uint64_t nw = windows_.size();
    static std::vector<std::string> no_cons; static std::vector<uint8_t> no_flags;
    std::vector<std::string>& cons = cons_out ? *cons_out : no_cons;
    std::vector<uint8_t>& pol = pol_out ? *pol_out : no_flags;
    std::vector<uint8_t>& chim = chim_out ? *chim_out : no_flags;
    rcn_read_set r{}; rcn_overlap_set o{};
    r.n_seqs = layout_.seq_off.size() - 1; r.n_targets = layout_.n_targets; r.seq_off = layout_.seq_off.data();
    r.bases = layout_.bases.data(); r.quals = layout_.quals.data(); r.seq_has_qual = layout_.seq_has_qual.data();
    o.n_overlaps = layout_.q_id.size(); o.q_id = layout_.q_id.data(); o.t_id = layout_.t_id.data(); o.strand = layout_.strand.data();
    o.bp_off = layout_.bp_off.data(); o.bp_t = layout_.bp_t.data(); o.bp_q = layout_.bp_q.data();
    // Shards: the window index space is cut into one contiguous range per device, balanced by the bases of the
    // overlaps that fall into it (the reference's multi-device code hands window ranges to per-device batches the
    // same way, src/cuda/cudapolisher.cpp:228-240).  A shard's engine gets every read (they are what overlaps
    // point into) and the overlaps that touch its range -- an overlap across a boundary goes to both sides, the
    // windows outside a shard's range come out as bare backbones there and are dropped.  RACON_HIP_DEVICE_SHARDS
    // forces a shard count (tests: several shards on one device).
    uint32_t n_shards = static_cast<uint32_t>(n_devices);
    if (const char* sh = getenv("RACON_HIP_DEVICE_SHARDS")) n_shards = std::max(1, atoi(sh));
    n_shards = static_cast<uint32_t>(std::min<uint64_t>(n_shards, std::max<uint64_t>(1, nw)));
    std::vector<uint64_t> first_window(layout_.n_targets + 1, 0);
    for (uint64_t t = 0; t < layout_.n_targets; ++t) {
        const uint64_t len = layout_.seq_off[t + 1] - layout_.seq_off[t];
        first_window[t + 1] = first_window[t] + (len + window_length_ - 1) / window_length_;
    }
    const uint64_t n_ovl = o.n_overlaps;
    std::vector<uint64_t> w_lo, w_hi;                        // windows [w_lo, w_hi] an overlap touches
    std::vector<double> win_cost;
    const bool planned = phase == 2 && device_cut_.size() == static_cast<size_t>(n_shards) + 1;      // (phase 1 left its cut behind)
    if (!planned) { w_lo.resize(n_ovl); w_hi.resize(n_ovl); win_cost.assign(nw + 1, 0.0); }
    for (uint64_t k = 0; k < n_ovl && !planned; ++k) {
        const uint64_t tb = layout_.t_begin[k], te = std::max<uint64_t>(layout_.t_end[k], tb + 1);
        w_lo[k] = first_window[o.t_id[k]] + tb / window_length_;
        w_hi[k] = std::min<uint64_t>(first_window[o.t_id[k]] + (te - 1) / window_length_, nw - 1);
        for (uint64_t w = w_lo[k]; w <= w_hi[k]; ++w) win_cost[w] += 1.0;
    }
    std::vector<uint64_t> cut(n_shards + 1, nw);
    cut[0] = 0;
    if (planned) cut = device_cut_;
    else {
        double total = 0; for (uint64_t w = 0; w < nw; ++w) total += win_cost[w] + 0.05;
        double acc = 0; uint32_t sidx = 1;
        for (uint64_t w = 0; w < nw && sidx < n_shards; ++w) {
            acc += win_cost[w] + 0.05;
            if (acc >= total * sidx / n_shards) cut[sidx++] = w + 1;
        }
    }
    if (phase == 1) device_cut_ = cut;
    const auto job_begin = std::chrono::steady_clock::now();
    std::mutex peak_mutex; uint64_t peak_used = 0;
    std::vector<std::string> shard_errors(n_shards);
    auto run_shard = [&](uint32_t sidx) {
        try {
            const uint64_t wa = cut[sidx], wb = cut[sidx + 1];
            if (wa >= wb) return;
            const double t_shard = seconds_since(job_begin);
            if (phase == 2) {                                      // built by initialize(): the consensus of the resident windows
                auto engine = engines_[static_cast<size_t>(sidx % static_cast<uint32_t>(n_devices))];
                engine->set_fetch_range(wa, wb);
                struct FetchAll { std::shared_ptr<HipEngine> e; ~FetchAll() { e->set_fetch_range(0, ~uint64_t(0)); } } fetch_all{engine};
                std::vector<std::string> c; std::vector<uint8_t> pl, ch;
                engine->run(trim_, &c, &pl, &ch);
                if (c.size() != nw) throw std::runtime_error("[racon::Polisher::polish] error: window count mismatch between host and device!");
                for (uint64_t w = wa; w < wb; ++w) { cons[w].swap(c[w]); pol[w] = pl[w]; chim[w] = ch[w]; }
                return;
            }
            std::vector<uint64_t> sel;
            for (uint64_t k = 0; k < n_ovl; ++k) if (w_hi[k] >= wa && w_lo[k] < wb) sel.push_back(k);
            const bool all = sel.size() == n_ovl;
            // the selected overlaps' slices of the layout arrays
            std::vector<uint32_t> q_id, t_id, bp_t, bp_q, q_start, t_begin, t_end, q_begin, q_end;
            std::vector<uint8_t> strand, cigar;
            std::vector<uint64_t> bp_off{0}, cigar_off{0};
            if (!all) {
                for (uint64_t k : sel) {
                    q_id.push_back(o.q_id[k]); t_id.push_back(o.t_id[k]); strand.push_back(o.strand[k]);
                    q_start.push_back(layout_.q_start[k]); t_begin.push_back(layout_.t_begin[k]); t_end.push_back(layout_.t_end[k]);
                    q_begin.push_back(layout_.q_begin[k]); q_end.push_back(layout_.q_end[k]);
                    bp_t.insert(bp_t.end(), layout_.bp_t.begin() + layout_.bp_off[k], layout_.bp_t.begin() + layout_.bp_off[k + 1]);
                    bp_q.insert(bp_q.end(), layout_.bp_q.begin() + layout_.bp_off[k], layout_.bp_q.begin() + layout_.bp_off[k + 1]);
                    bp_off.push_back(bp_t.size());
                    cigar.insert(cigar.end(), layout_.cigar.begin() + layout_.cigar_off[k], layout_.cigar.begin() + layout_.cigar_off[k + 1]);
                    cigar_off.push_back(cigar.size());
                }
            }
            rcn_overlap_set so = o;
            const uint32_t* p_q_start = layout_.q_start.data(); const uint32_t* p_t_begin = layout_.t_begin.data(); const uint32_t* p_t_end = layout_.t_end.data();
            const uint32_t* p_q_begin = layout_.q_begin.data(); const uint32_t* p_q_end = layout_.q_end.data();
            const uint64_t* p_cigar_off = layout_.cigar_off.data(); const uint8_t* p_cigar = layout_.cigar.data();
            static const uint8_t kNoByte = 0; static const uint32_t kNoWord = 0;
            if (!all) {
                so.n_overlaps = sel.size(); so.q_id = q_id.empty() ? &kNoWord : q_id.data(); so.t_id = t_id.empty() ? &kNoWord : t_id.data();
                so.strand = strand.empty() ? &kNoByte : strand.data(); so.bp_off = bp_off.data();
                so.bp_t = bp_t.empty() ? &kNoWord : bp_t.data(); so.bp_q = bp_q.empty() ? &kNoWord : bp_q.data();
                p_q_start = q_start.data(); p_t_begin = t_begin.data(); p_t_end = t_end.data(); p_q_begin = q_begin.data(); p_q_end = q_end.data();
                p_cigar_off = cigar_off.data(); p_cigar = cigar.empty() ? &kNoByte : cigar.data();
            }
            // The reads this shard's overlaps point into, and nothing else: every shard needs every target (their
            // windows outside the range come out as bare backbones), but of the reads only its own -- one eighth of
            // cfg3's 1.5 G bases per device instead of all of them on each (the reference's multi-device path keeps the
            // reads on the host and packs per batch, src/cuda/cudapolisher.cpp:254-276).
            rcn_read_set sr = r;
            std::vector<uint64_t> r_seq_off; std::vector<uint8_t> r_bases, r_quals, r_hq;
            if (!all && r.n_seqs > r.n_targets) {
                constexpr uint32_t kUnused = 0xffffffffu;
                std::vector<uint32_t> remap(r.n_seqs, kUnused), old_of;
                for (uint64_t t = 0; t < r.n_targets; ++t) { remap[t] = static_cast<uint32_t>(t); old_of.push_back(static_cast<uint32_t>(t)); }
                for (uint32_t& q : q_id) {
                    if (remap[q] == kUnused) { remap[q] = static_cast<uint32_t>(old_of.size()); old_of.push_back(q); }
                    q = remap[q];
                }
                uint64_t total = 0;
                for (uint32_t old : old_of) total += r.seq_off[old + 1] - r.seq_off[old];
                r_seq_off.assign(1, 0); r_seq_off.reserve(old_of.size() + 1);
                r_bases.resize(total + 1); r_quals.resize(total + 1); r_hq.reserve(old_of.size());
                for (uint32_t old : old_of) {
                    const uint64_t a = r.seq_off[old], len = r.seq_off[old + 1] - a, d = r_seq_off.back();
                    std::copy(r.bases + a, r.bases + a + len, r_bases.begin() + d);
                    std::copy(r.quals + a, r.quals + a + len, r_quals.begin() + d);
                    r_hq.push_back(r.seq_has_qual[old]);
                    r_seq_off.push_back(d + len);
                }
                sr.n_seqs = old_of.size(); sr.seq_off = r_seq_off.data(); sr.bases = r_bases.data(); sr.quals = r_quals.data(); sr.seq_has_qual = r_hq.data();
            }
            // (the shards of one device run one after the other on its lane thread: they share the device's first engine)
            const int32_t device = static_cast<int32_t>(sidx % static_cast<uint32_t>(n_devices));
            auto engine = engines_[static_cast<size_t>(device)];
            engine->set_fetch_range(wa, wb);                       // the strings of its own windows only
            struct FetchAll { std::shared_ptr<HipEngine> e; ~FetchAll() { e->set_fetch_range(0, ~uint64_t(0)); } } fetch_all{engine};   // (also when a call below throws)
            std::vector<std::string> c; std::vector<uint8_t> pl, ch;
            auto take = [&]() {
                if (c.size() != nw) throw std::runtime_error("[racon::Polisher::polish] error: window count mismatch between host and device!");
                for (uint64_t w = wa; w < wb; ++w) { cons[w].swap(c[w]); pol[w] = pl[w]; chim[w] = ch[w]; }
            };
            bool aligned_on_device = false;
            std::vector<uint8_t> host_cigar; std::vector<uint64_t> host_cigar_off; std::vector<uint32_t> host_q_start;
            if (device_align_) {
                rcn_pair_set ps{};
                ps.n_pairs = so.n_overlaps; ps.q_id = so.q_id; ps.t_id = so.t_id; ps.strand = so.strand;
                ps.q_begin = p_q_begin; ps.q_end = p_q_end; ps.t_begin = p_t_begin; ps.t_end = p_t_end;
                try {
                    if (getenv("RACON_HIP_FORCE_ALIGN_FALLBACK")) throw FatalError("forced");      // (tests)
                    engine->build(sr, ps, window_length_, quality_threshold_, layout_.window_type);
                    aligned_on_device = true;
                } catch (const FatalError&) {
                    // The device aligner holds one op byte per row + column of every overlap and a per-wave scratch sized
                    // for the longest read: an input it has no room for (RCN_E_CAPACITY / RCN_E_NOMEM; also a read beyond
                    // its 3 Mbp limit) is aligned HERE instead, by the host's edlib-equivalent (reference
                    // src/overlap.cpp:205-224) -- same paths, hence the same windows -- and goes on through the CIGAR path.
                    if (!getenv("RACON_HIP_FORCE_ALIGN_FALLBACK") && engine->last_rc() != RCN_E_CAPACITY && engine->last_rc() != RCN_E_NOMEM) throw;
                    static const struct Comp { char t[256]; Comp() { for (int i = 0; i < 256; ++i) t[i] = static_cast<char>(i); t['A'] = 'T'; t['T'] = 'A'; t['C'] = 'G'; t['G'] = 'C'; } } comp;
                    const uint64_t n = so.n_overlaps;
                    std::vector<std::string> cg(n);
                    host_q_start.resize(n);
                    parallel_for(n, num_threads_, [&](uint64_t k) {
                        const uint64_t qa = sr.seq_off[so.q_id[k]], ql = sr.seq_off[so.q_id[k] + 1] - qa, ta = sr.seq_off[so.t_id[k]];
                        std::string q(reinterpret_cast<const char*>(sr.bases + qa + p_q_begin[k]), p_q_end[k] - p_q_begin[k]);
                        if (so.strand[k]) { std::reverse(q.begin(), q.end()); for (char& ch_ : q) ch_ = comp.t[static_cast<unsigned char>(ch_)]; }
                        cg[k] = nwpath::align_cigar(q.data(), static_cast<uint32_t>(q.size()), reinterpret_cast<const char*>(sr.bases + ta + p_t_begin[k]), p_t_end[k] - p_t_begin[k]);
                        host_q_start[k] = so.strand[k] ? static_cast<uint32_t>(ql - p_q_end[k]) : p_q_begin[k];       // reference src/overlap.cpp:241-242
                    });
                    host_cigar_off.assign(1, 0);
                    for (const auto& s_ : cg) { host_cigar.insert(host_cigar.end(), s_.begin(), s_.end()); host_cigar_off.push_back(host_cigar.size()); }
                }
            }
            if (device_align_ && !aligned_on_device) {
                rcn_cigar_set a{};
                static const uint8_t kNoCigar = 0;
                a.n_overlaps = so.n_overlaps; a.q_id = so.q_id; a.t_id = so.t_id; a.strand = so.strand;
                a.q_start = host_q_start.empty() ? &kNoWord : host_q_start.data(); a.t_begin = p_t_begin; a.t_end = p_t_end;
                a.cigar_off = host_cigar_off.data(); a.cigar = host_cigar.empty() ? &kNoCigar : host_cigar.data();
                engine->build(sr, a, window_length_, quality_threshold_, layout_.window_type);
            } else if (device_align_) {
            } else if (device_cigars_) {
                rcn_cigar_set a{};
                a.n_overlaps = so.n_overlaps; a.q_id = so.q_id; a.t_id = so.t_id; a.strand = so.strand;
                a.q_start = p_q_start; a.t_begin = p_t_begin; a.t_end = p_t_end; a.cigar_off = p_cigar_off; a.cigar = p_cigar;
                engine->build(sr, a, window_length_, quality_threshold_, layout_.window_type);
            } else {
                engine->build(sr, so, window_length_, quality_threshold_, layout_.window_type);
            }
            const bool timing = getenv("RACON_HIP_TIMING") != nullptr;
            const double t_built = seconds_since(job_begin);
            if (phase == 1) {                                      // the windows stay resident for polish()
                engine->reserve_run();
                if (timing) fprintf(stderr, "[racon::Polisher::initialize] timing: shard %u (windows %lu..%lu, %lu overlaps) built on device %d in %.1f ms, run reserved after %.1f ms, %.2f GB of HBM in use\n",
                                    sidx, static_cast<unsigned long>(wa), static_cast<unsigned long>(wb), static_cast<unsigned long>(so.n_overlaps), device,
                                    1e3 * (t_built - t_shard), 1e3 * (seconds_since(job_begin) - t_shard), HipEngine::UsedMemory(device % std::max(1, HipEngine::DeviceCount())) / 1e9);
                return;
            }
            const uint64_t used_built = timing ? HipEngine::UsedMemory(device % std::max(1, HipEngine::DeviceCount())) : 0;
            engine->run(trim_, &c, &pl, &ch);
            take();
            if (timing) {
                const uint64_t used = std::max(used_built, HipEngine::UsedMemory(device % std::max(1, HipEngine::DeviceCount())));
                fprintf(stderr, "[racon::Polisher::polish] timing: shard %u (windows %lu..%lu, %lu overlaps) on device %d: built in %.1f ms, consensus + results in %.1f ms (kernel %.1f), %.2f GB of HBM in use\n",
                        sidx, static_cast<unsigned long>(wa), static_cast<unsigned long>(wb), static_cast<unsigned long>(so.n_overlaps), device,
                        1e3 * (t_built - t_shard), 1e3 * (seconds_since(job_begin) - t_built), engine->last_kernel_ms(), used / 1e9);
                std::lock_guard<std::mutex> lock(peak_mutex); peak_used = std::max(peak_used, used);
            }
        } catch (const std::exception& ex) { shard_errors[sidx] = ex.what(); }
    };
    {
        // one thread per device; the shards of one device run one after the other on it
        std::vector<std::thread> pool;
        const uint32_t lanes = std::min<uint32_t>(n_shards, static_cast<uint32_t>(n_devices));
        for (uint32_t l = 0; l < lanes; ++l) pool.emplace_back([&, l]() { FatalThrowsScope scope; for (uint32_t sidx = l; sidx < n_shards; sidx += lanes) run_shard(sidx); });
        for (auto& th : pool) th.join();
    }
    for (const auto& e : shard_errors) if (!e.empty()) fatal(e);
    if (peak_used) fprintf(stderr, "[racon::Polisher::polish] timing: %u shard(s), peak HBM in use %.2f GB\n", n_shards, peak_used / 1e9);
}

// ---------------------------------------------------------------- polish
void Polisher::pack_windows(PackedBatch* out) const {
    out->clear();
    for (const auto& w : windows_) out->add(*w);
}

void Polisher::assemble(const std::function<const std::string&(uint64_t)>& consensus, const std::function<bool(uint64_t)>& polished,
                        std::vector<std::unique_ptr<Sequence>>& dst, bool drop_unpolished_sequences) {
    std::string polished_data;
    uint32_t num_polished_windows = 0;
    for (uint64_t i = 0; i < windows_.size(); ++i) {
        if (windows_[i]->rank() == 0) {
            // (one allocation per target, moved into the Sequence: a megabase grown by doubling and then copied is its pages touched twice)
            uint64_t total = 0;
            for (uint64_t k = i; k < windows_.size() && (k == i || windows_[k]->rank() != 0); ++k) total += consensus(k).size();
            polished_data.reserve(total);
        }
        num_polished_windows += polished(i) ? 1 : 0;
        polished_data += consensus(i);
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
        device_job(device_built_ ? 2 : 0, &cons, &pol, &chim);
        for (uint64_t i = 0; i < nw; ++i)
            if (chim[i]) fprintf(stderr, "[racon::Window::generate_consensus] warning: contig %lu might be chimeric in window %u!\n",
                                 static_cast<unsigned long>(windows_[i]->id()), windows_[i]->rank());
        assemble([&](uint64_t i) -> const std::string& { return cons[i]; }, [&](uint64_t i) { return pol[i] != 0; },
                 dst, drop_unpolished_sequences);
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
