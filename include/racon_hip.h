/*
 * racon_hip.h — C ABI of the MI355X-native window-consensus engine.
 *
 * This is the drop-in boundary for ONE path of lbcb-sci/racon: the per-window
 * POA consensus, i.e. what `racon::Window::generate_consensus`
 * (reference src/window.cpp:65-149) computes through spoa, batched the way the
 * reference's own accelerator seam batches it (`CUDABatchProcessor`,
 * reference src/cuda/cudabatch.hpp:27-122 and src/cuda/cudabatch.cpp:77-270).
 *
 * Plain pointers and sizes only; no C++ / torch types cross this boundary.
 * All functions return 0 on success, >0 for "soft" conditions documented per
 * function, <0 for errors (see rcn_strerror).
 */
#ifndef RACON_HIP_H_
#define RACON_HIP_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------
 * Packed window batch: the information a racon::Window holds
 * (reference src/window.hpp:64-73: id_, rank_, type_, sequences_, qualities_,
 * positions_) for many windows, flattened.  Sequence 0 of every window is the
 * backbone (positions (0,0), reference src/window.cpp:29-37); the others are
 * the layers in add_layer() arrival order (reference src/window.cpp:42-63).
 * The engine does the begin-ordered `std::sort` of window.cpp:79-86 itself.
 * ------------------------------------------------------------------------- */
typedef struct rcn_batch {
    uint32_t n_windows;
    uint32_t n_seqs;               /* == win_seq_off[n_windows]                        */
    const uint32_t* win_seq_off;   /* [n_windows+1] first sequence of each window      */
    const uint8_t*  win_type;      /* [n_windows]  0 = kNGS, 1 = kTGS (window.hpp:21)   */
    const uint64_t* seq_off;       /* [n_seqs+1]   byte offset of each sequence         */
    const uint8_t*  seq_has_qual;  /* [n_seqs]     0 => quality pointer was nullptr     */
    const uint32_t* seq_begin;     /* [n_seqs]     positions_.first  (backbone: 0)      */
    const uint32_t* seq_end;       /* [n_seqs]     positions_.second (backbone: 0)      */
    const uint8_t*  bases;         /* [seq_off[n_seqs]] upper-case ASCII as in Sequence */
    const uint8_t*  quals;         /* [seq_off[n_seqs]] phred+33; ignored if !has_qual  */
} rcn_batch;

/* Result of one batch.  Pointers are owned by the producer (engine handle or
 * oracle result object) and stay valid until the next run/reset/destroy. */
typedef struct rcn_result {
    uint32_t n_windows;
    const uint64_t* cons_off;      /* [n_windows+1] */
    const uint8_t*  cons;          /* consensus bytes (after TGS trimming)             */
    const uint8_t*  polished;      /* [n_windows] return value of generate_consensus   */
    const uint8_t*  chimeric;      /* [n_windows] 1 => the "might be chimeric" warning
                                      of window.cpp:139-142 applies                    */
} rcn_result;

/* Engine configuration; mirrors the arguments of createCUDABatch
 * (reference src/cuda/cudabatch.hpp:25) and spoa::AlignmentEngine::Create
 * (reference src/polisher.cpp:180-182). */
typedef struct rcn_engine_config {
    int32_t device;        /* HIP device ordinal                                        */
    int8_t  match;         /* -m */
    int8_t  mismatch;      /* -x */
    int8_t  gap;           /* -g (linear)                                               */
    uint8_t trim;          /* Polisher trim_ flag (main.cpp:87-89: --no-trimming -> 0)  */
    uint64_t arena_bytes;  /* device scratch budget; 0 = pick from free memory          */
    uint32_t max_slots;    /* resident windows (wavefronts) in flight; 0 = auto         */
    uint32_t flags;        /* RCN_FLAG_*                                                */
} rcn_engine_config;

#define RCN_FLAG_NONE        0u
#define RCN_FLAG_PROFILE     1u   /* record per-launch HIP event timings               */

typedef struct rcn_engine rcn_engine;

/* Per-run timing / work counters (for bench.py's roofline line). */
typedef struct rcn_run_stats {
    double   kernel_ms;        /* HIP-event time of the consensus kernel launches       */
    double   h2d_ms, d2h_ms;   /* staging copies (0 when inputs are device resident)    */
    uint32_t n_launches;
    uint32_t n_retried;        /* windows re-run with worst-case capacities             */
    uint64_t dp_cells;         /* sum over alignments of (V'+1)(l+1), counted on device */
    uint64_t dp_pred_cells;    /* sum of cells * (#in-edge rows read), on device        */
    uint64_t bytes_in, bytes_out;
    uint64_t dp_bytes;         /* algorithmic DP bytes (SURVEY 8(d) yardstick), on device     */
    uint64_t phase_clocks[8];  /* summed wave clocks: subgraph, row-desc, DP, traceback,
                                  add-alignment, toposort, consensus, queue/other           */
    uint64_t n_sink_ties;      /* alignments that needed spoa's exact rank order (sink tie)    */
    /* exact banded DP: dp_cells / dp_pred_cells / dp_bytes count the cells actually evaluated; the figures of the
     * full matrices (what an unbanded pass over the same alignments evaluates) are reported alongside */
    uint64_t dp_cells_full, dp_bytes_full;
    uint64_t n_banded;         /* alignments done by the banded pass (certificate held)               */
    uint64_t n_band_redone;    /* alignments redone on full rows because the certificate failed       */
    uint64_t band_redo_why[8]; /* redo reasons, counted: source row off the left edge, predecessor older than the LDS ring,
                                  >2 window shifts inside the ring, >8 in-edges, (shift span), no end cell, alive last window
                                  cell, alive dropped cell                                                                   */
    uint32_t wg_per_cu;        /* work-groups of the consensus kernel per CU in the last launch: 8, or 6 when the batch is
                                  resident all at once and lasts as long as its deepest window (engine.hip: wg_per_cu)     */
    uint32_t split_deep;       /* split launch (engine.hip: split_plan): windows of the deep launch, 0 = one uniform launch */
    uint32_t split_cus;        /* CUs the deep launch ran on (the other launch: all the others)                            */
    uint32_t split_deep_per_cu;/* work-groups per CU of the deep launch                                                    */
    double   launch_ms[2];     /* HIP-event durations of the (up to two, concurrent) launches of the first pass; kernel_ms is
                                  the interval they cover together                                                         */
    uint64_t n_code_wave;      /* banded alignments whose move codes were assembled by waves 1-3 of the work-group next to the DP
                                  wave (windows with a CU and its LDS to themselves: the deep launch; poa_band.hpp)          */
    uint64_t n_small;          /* windows polished by the small-window kernel (one wave per window, graph in LDS: poa_small.hpp) */
    uint64_t n_small_bailed;   /* windows that kernel sent back (outside its shape) and poa_window_kernel2 polished instead       */
    uint64_t small_bail_why[9];/* ... by reason: graph capacity, ninth in-edge, predecessor > 16 rows back, aligned ring (or a symbol
                                  besides A/C/G/T), int16 range, layer > 255 bases, sink tie beyond the id rule, consensus scratch,
                                  internal inconsistency (must be zero)                                                           */
    uint64_t small_work[6];    /* work of that kernel: alignments, DP rows, Subgraph sweep chunks, traceback boxes, boxes whose gather
                                  pipeline had to start over (the walk left the diagonal), Subgraph masks that were a rank interval */
    /* three-tier split launch: a middle tier between the deep launch and the rest (launch_ms[0] = deep, launch_ms[1] = rest) */
    double   launch_ms_mid;    /* HIP-event duration of the middle tier's launch (0: no middle tier)                              */
    uint32_t split_mid;        /* windows of the middle tier                                                                       */
    uint32_t split_mid_cus;    /* CUs it ran on                                                                                    */
    uint32_t split_mid_per_cu; /* its work-groups per CU                                                                           */
    uint32_t reserved_;
} rcn_run_stats;

/* --- engine lifetime (replaces createCUDABatch, cudabatch.cpp:24-75) ------- */
int  rcn_engine_create(const rcn_engine_config* cfg, rcn_engine** out);
void rcn_engine_destroy(rcn_engine* e);

/* --- whole-batch form: upload (H2D) / run / fetch ------------------------- */
/* rcn_engine_upload packs and copies a batch into HBM (may be called once and
 * followed by many rcn_engine_run calls: inputs stay resident).               */
int  rcn_engine_upload(rcn_engine* e, const rcn_batch* b);
/* Runs the consensus kernel(s) over the resident batch and brings results back. */
int  rcn_engine_run(rcn_engine* e);
/* upload + run of one batch with the copy hidden behind the kernel: what one task of Polisher::polish does per batch
 * (reference src/cuda/cudapolisher.cpp:254-333: fill a batch, generateConsensus, read back).  The windows are packed
 * deepest first into pinned staging, copied in pieces on a copy stream, and each piece is polished by its own launch as
 * soon as it has arrived.  Results and statistics as after rcn_engine_upload + rcn_engine_run; the batch stays
 * resident (rcn_engine_run may follow).                                                                               */
int  rcn_engine_polish(rcn_engine* e, const rcn_batch* b);

/* The same batch as BORROWED POINTERS, the form a racon::Window holds its sequences in (reference src/window.hpp:71-73:
 * `(const char*, uint32_t)` pairs into Polisher::sequences_; what CUDABatchProcessor::addWindow reads,
 * src/cuda/cudabatch.cpp:80-122).  rcn_engine_polish_refs packs straight from these pointers into the engine's pinned
 * staging (host threads, deepest window first) -- one host copy of the bases instead of two -- and otherwise does what
 * rcn_engine_polish does.  Nothing is retained after the call returns.                                               */
typedef struct rcn_window_refs {
    uint32_t n_windows;
    uint32_t n_seqs;                 /* == win_seq_off[n_windows]                                   */
    const uint32_t* win_seq_off;     /* [n_windows+1]                                               */
    const uint8_t*  win_type;        /* [n_windows]  0 = kNGS, 1 = kTGS                              */
    const uint8_t* const* seq;       /* [n_seqs] Window::sequences_[i].first                         */
    const uint8_t* const* qual;      /* [n_seqs] Window::qualities_[i].first; NULL = no quality      */
    const uint32_t* seq_len;         /* [n_seqs] Window::sequences_[i].second                        */
    const uint32_t* seq_begin;       /* [n_seqs] positions_.first  (backbone: 0)                     */
    const uint32_t* seq_end;         /* [n_seqs] positions_.second (backbone: 0)                     */
    uint32_t flags;                  /* RCN_REFS_*                                                   */
} rcn_window_refs;
#define RCN_REFS_QUEUED  1u          /* this batch is one of several that the caller keeps in flight on the device (two
                                        engines alternating): run it at full residency (eight work-groups per CU) even
                                        when its own deepest window would otherwise rule the launch                  */
int  rcn_engine_polish_refs(rcn_engine* e, const rcn_window_refs* w);

/* Allocation ahead of the first batch (the role of spoa::AlignmentEngine::Prealloc(window_length, 5), reference
 * src/polisher.cpp:180-182, and of createCUDABatch's up-front device allocation, src/cuda/cudabatch.cpp:24-75):
 * device input arrays, the per-slot scratch arena, pinned staging for inputs and results, and the first use of the
 * code object and the copy engines, so that the first rcn_engine_polish* call pays none of it.  Every figure is a
 * hint: a batch that needs more grows the buffers as before.                                                      */
typedef struct rcn_reserve_hint {
    uint32_t n_windows;              /* windows per batch                                            */
    uint32_t n_seqs;                 /* sequences (backbones + layers) per batch                     */
    uint64_t n_bases;                /* bases per batch                                              */
    uint32_t window_length;          /* backbone length (-w)                                         */
    uint32_t max_layer_length;       /* longest layer; 0 = 1.3 x window_length                       */
    uint64_t max_window_bases;       /* bases of the deepest window, backbone included (it sizes a resident window's
                                        scratch slot); 0 = n_bases / n_windows                        */
} rcn_reserve_hint;
int  rcn_engine_reserve(rcn_engine* e, const rcn_reserve_hint* hint);
/* The same for ONE known batch: exactly what rcn_engine_polish_refs(w) will allocate (inputs, pinned staging, results,
 * the scratch of its launches as it will plan them), without packing, copying or running anything.                */
int  rcn_engine_reserve_refs(rcn_engine* e, const rcn_window_refs* w);
/* ... and for the batch that is RESIDENT (rcn_engine_upload / rcn_engine_build_windows*): exactly what rcn_engine_run will
 * allocate -- the scratch of its launches as it will plan them, the pinned result block -- without running anything.  The
 * host layer builds its windows in HBM inside Polisher::initialize (where the reference builds them: src/polisher.cpp:388-461)
 * and calls this there, so that the interval around Polisher::polish holds the consensus only.                               */
int  rcn_engine_reserve_run(rcn_engine* e);

int  rcn_engine_result(rcn_engine* e, rcn_result* out);
int  rcn_engine_stats(rcn_engine* e, rcn_run_stats* out);
/* --- self-check (no counterpart in the reference: its CPU path has no shortcuts to check) ---------------------------------
 * The consensus kernels take proved shortcuts where spoa does plain work: an exact 256-column band with a certificate, one-byte
 * move codes instead of the score matrix, a rule instead of spoa's DFS order at tied sinks and in the consensus, a rank-interval
 * Subgraph, a one-wave kernel for small windows.  rcn_engine_verify re-polishes a deterministic sample of the windows of the LAST
 * run ON THE GPU with every one of them switched off -- full rows, score-matrix traceback, spoa's DFS order wherever an order is
 * asked for, poa_window_kernel2 (and the int32 kernel for what outgrows it) -- and compares consensus bytes and flags with what
 * the run returned.  No oracle, no CPU: the device inputs of the run are still resident.  `fraction` in (0, 1]: the share of the
 * windows sampled (by a hash of the window index: the same windows every time), at least one.  Returns RCN_OK when the check RAN;
 * the caller looks at n_differ (the host layer treats > 0 as fatal and names first_window).  Product switch of the host layer:
 * RACON_HIP_VERIFY=<fraction>. */
typedef struct rcn_verify_report {
    uint32_t n_checked;        /* windows re-polished                                                             */
    uint32_t n_differ;         /* ... whose bytes or flags differ from the run's                                  */
    uint32_t first_window;     /* the first of them (index within the run's batch), 0xffffffff = none             */
    uint32_t n_int32;          /* sampled windows the exact pass of poa_window_kernel2 handed on to the int32 kernel */
    double   ms;               /* wall time of the check                                                          */
} rcn_verify_report;
int  rcn_engine_verify(rcn_engine* e, double fraction, rcn_verify_report* out);

/* Changes the trim flag (Window::generate_consensus takes it per call, reference
 * src/window.hpp:47-48); takes effect at the next rcn_engine_run.                 */
int  rcn_engine_set_trim(rcn_engine* e, int trim);

/* --- incremental form (replaces CUDABatchProcessor::addWindow / hasWindows /
 *     generateConsensus / reset, cudabatch.hpp:39-64) ----------------------- */
typedef struct rcn_window_desc {
    uint8_t  type;                 /* 0 kNGS / 1 kTGS                                   */
    uint32_t n_seqs;               /* backbone + layers                                 */
    const char* const* seq;        /* [n_seqs] borrowed pointers, copied at add time    */
    const uint32_t* seq_len;       /* [n_seqs]                                          */
    const char* const* qual;       /* [n_seqs] NULL => no quality                       */
    const uint32_t* begin;         /* [n_seqs]                                          */
    const uint32_t* end;           /* [n_seqs]                                          */
} rcn_window_desc;
/* 0 = added; 1 = batch full (window NOT consumed), like addWindow()==false.    */
int  rcn_engine_add_window(rcn_engine* e, const rcn_window_desc* w);
int  rcn_engine_has_windows(rcn_engine* e);
/* upload+run over the windows added so far */
int  rcn_engine_generate_consensus(rcn_engine* e);
int  rcn_engine_reset(rcn_engine* e);
/* A handle that goes on to ANOTHER job (the host layer keeps engines alive across Polishers): forgets what earlier batches taught it --
 * the first-pass capacity level, the vote against the small-window kernel -- so that the next job starts as on a new engine.  Buffers stay. */
int  rcn_engine_forget(rcn_engine* e);

/* --- device-side window construction -----------------
 * Replaces the two serial host loops at the end of Polisher::initialize: targets
 * cut into backbone windows (reference src/polisher.cpp:388-403, createWindow
 * src/window.cpp:15-40) and every overlap cut into layers at its breaking points
 * (reference src/polisher.cpp:405-461: length filter :415, mean-quality filter
 * :419-433, window rank / begin / end :436-457; Window::add_layer
 * src/window.cpp:42-63; reverse strand = Sequence::create_reverse_complement,
 * src/sequence.cpp:49-84).  Reads, qualities and breaking points go to HBM once;
 * filter, per-window counting, stable ordering (layers keep overlap order, as
 * the serial loop adds them) and the gather (with reverse complement) run there,
 * and the packed batch stays resident for rcn_engine_run -- no host packing, no
 * second copy of the bases over PCIe.                                            */
typedef struct rcn_read_set {
    uint64_t n_seqs;               /* Polisher::sequences_ (src/polisher.hpp:86): targets first   */
    uint64_t n_targets;            /* sequences [0, n_targets) are the targets                    */
    const uint64_t* seq_off;       /* [n_seqs+1] byte offsets into bases / quals                   */
    const uint8_t*  bases;         /* forward strand, upper case (Sequence ctor, sequence.cpp:19-31) */
    const uint8_t*  quals;         /* phred+33 at the same offsets; ignored where !seq_has_qual     */
    const uint8_t*  seq_has_qual;  /* [n_seqs] 0 = FASTA record or all-'!' quality (sequence.cpp:33-47) */
} rcn_read_set;

typedef struct rcn_overlap_set {
    uint64_t n_overlaps;           /* valid overlaps, in the order Polisher::initialize keeps them */
    const uint32_t* q_id;          /* [n_overlaps] index into the read set                         */
    const uint32_t* t_id;          /* [n_overlaps] target index (< n_targets)                      */
    const uint8_t*  strand;        /* [n_overlaps] 1 = query reverse-complemented                  */
    const uint64_t* bp_off;        /* [n_overlaps+1] offsets, in points, into bp_t / bp_q (even counts) */
    const uint32_t* bp_t;          /* Overlap::breaking_points_[k].first  (target position)        */
    const uint32_t* bp_q;          /* Overlap::breaking_points_[k].second (query position on the overlap's strand) */
} rcn_overlap_set;

/* The alignments instead of the breaking points: Overlap::find_breaking_points' CIGAR walk
 * (reference src/overlap.cpp:226-292) then also runs on the device, one thread per overlap, and its output feeds the
 * window construction without leaving HBM.  q_id / t_id / strand as in rcn_overlap_set.                        */
typedef struct rcn_cigar_set {
    uint64_t n_overlaps;
    const uint32_t* q_id; const uint32_t* t_id; const uint8_t* strand;     /* [n_overlaps]                        */
    const uint32_t* q_start;       /* first query position on the overlap's strand: strand ? q_length - q_end : q_begin
                                      (reference src/overlap.cpp:241-242)                                          */
    const uint32_t* t_begin;       /* [n_overlaps] Overlap::t_begin_                                               */
    const uint32_t* t_end;         /* [n_overlaps] Overlap::t_end_                                                 */
    const uint64_t* cigar_off;     /* [n_overlaps+1] byte offsets into cigar                                        */
    const uint8_t*  cigar;         /* CIGAR text (M = X match/mismatch, I, D N, S H P ignored), as in Overlap::cigar_ */
} rcn_cigar_set;

typedef struct rcn_build_stats {
    double   h2d_ms;               /* reads + overlaps to HBM                                      */
    double   kernel_ms;            /* filter + sort + scans + gather, HIP events                   */
    double   gather_ms;            /* the gather kernel alone (the HBM-bound part)                 */
    uint64_t n_pairs, n_layers;    /* breaking-point pairs seen / layers kept                      */
    uint64_t gather_bytes;         /* bytes read + written by the gather (bases and qualities)     */
} rcn_build_stats;

/* Builds the resident batch (every window of every target, in target order).  window_type: 0 kNGS,
 * 1 kTGS (src/polisher.cpp:277-278).  RCN_E_LAYER when a layer violates the add_layer contract
 * (src/window.cpp:49-58), where the reference exits with its fatal error; RCN_E_ARG for malformed arguments. */
int  rcn_engine_build_windows(rcn_engine* e, const rcn_read_set* reads, const rcn_overlap_set* overlaps,
                              uint32_t window_length, double quality_threshold, uint8_t window_type);
/* The same from alignments: breaking points (reference src/overlap.cpp:226-292) + window construction, all in HBM.
 * rcn_build_stats.n_pairs then counts the slots (one per window an overlap touches), kept or not.                */
int  rcn_engine_build_windows_from_cigars(rcn_engine* e, const rcn_read_set* reads, const rcn_cigar_set* alignments,
                                          uint32_t window_length, double quality_threshold, uint8_t window_type);
int  rcn_engine_build_stats(rcn_engine* e, rcn_build_stats* out);

/* --- exact pairwise alignment on the device ------------
 * Replaces the host call of Overlap::find_breaking_points for overlaps that come without a CIGAR (PAF / MHAP):
 *   edlibAlign(q, ql, t, tl, edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_PATH, NULL, 0)) + edlibAlignmentToCigar
 * (reference src/overlap.cpp:205-224; the reference's own batched device aligner is src/cuda/cudaaligner.cpp:51-102).
 * Global unit-cost alignment of the query segment (reverse-complemented when strand = 1, src/overlap.cpp:193-195)
 * against the target segment, and THE SAME co-optimal path edlib 1.2.7 returns (plain traceback
 * preferring insertion, deletion, then (mis)match below 1 MiB of traceback state, Hirschberg on the target axis with the
 * smallest optimal query split above) -- byte-identical CIGARs, hence byte-identical breaking points and windows.      */
typedef struct rcn_pair_set {
    uint64_t n_pairs;
    const uint32_t* q_id;          /* [n_pairs] index into the read set                                            */
    const uint32_t* t_id;          /* [n_pairs] target index (< n_targets)                                         */
    const uint8_t*  strand;        /* [n_pairs] 1 = the query segment is reverse-complemented                      */
    const uint32_t* q_begin;       /* [n_pairs] Overlap::q_begin_ / q_end_: the segment on the FORWARD read        */
    const uint32_t* q_end;
    const uint32_t* t_begin;       /* [n_pairs] Overlap::t_begin_ / t_end_                                         */
    const uint32_t* t_end;
} rcn_pair_set;

typedef struct rcn_align_stats {
    double   h2d_ms;               /* reads + pair table to HBM (0 when the reads were resident)                   */
    double   kernel_ms;            /* the alignment kernel, HIP events                                            */
    uint64_t n_pairs;
    uint64_t cells;                /* sum of rows x columns: the cells of the full matrices                        */
    uint64_t ops_bytes;            /* path positions written (sum of rows + columns)                               */
    uint32_t slots;                /* resident wavefronts (one overlap each)                                       */
} rcn_align_stats;

/* Aligns every pair; the paths stay in HBM (one op byte per path position).  `reads` as in rcn_engine_build_windows. */
int  rcn_engine_align_pairs(rcn_engine* e, const rcn_read_set* reads, const rcn_pair_set* pairs);
/* The alignments of the last rcn_engine_align_pairs as CIGAR strings (run-length encoded on the host after one D2H copy)
 * and edit distances.  cigar_off: [n_pairs + 1]; `cigar` may be NULL or `cap` too small: *need receives the total bytes. */
int  rcn_engine_alignment_cigars(rcn_engine* e, uint64_t* cigar_off, char* cigar, uint64_t cap, uint64_t* need, int32_t* distance);
int  rcn_engine_align_stats(rcn_engine* e, rcn_align_stats* out);
/* Alignment, breaking points and window construction in one go: reads and the pair table go to HBM once, nothing of the
 * alignments comes back to the host (reference src/overlap.cpp:176-292 + src/polisher.cpp:388-461).                  */
int  rcn_engine_build_windows_from_pairs(rcn_engine* e, const rcn_read_set* reads, const rcn_pair_set* pairs,
                                         uint32_t window_length, double quality_threshold, uint8_t window_type);

/* Dimensions and a copy (D2H) of the resident batch, uploaded or built; any output pointer may be NULL. */
typedef struct rcn_batch_dims { uint32_t n_windows, n_seqs; uint64_t n_bases; } rcn_batch_dims;
int  rcn_engine_batch_dims(rcn_engine* e, rcn_batch_dims* out);
int  rcn_engine_export_batch(rcn_engine* e, uint32_t* win_seq_off, uint8_t* win_type, uint64_t* seq_off,
                             uint8_t* seq_has_qual, uint32_t* seq_begin, uint32_t* seq_end, uint8_t* bases, uint8_t* quals);

/* --- misc ------------------------------------------------------------------ */
int  rcn_device_count(void);
/* free / total HBM of a device (several engines on one device share it: give each its part as arena_bytes) */
int  rcn_device_free_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes);
const char* rcn_strerror(int code);
const char* rcn_version(void);

#define RCN_OK              0
#define RCN_BATCH_FULL      1
#define RCN_E_NO_DEVICE   (-1)
#define RCN_E_HIP         (-2)
#define RCN_E_ARG         (-3)
#define RCN_E_NOMEM       (-4)
#define RCN_E_STATE       (-5)
#define RCN_E_CAPACITY    (-6)
#define RCN_E_LAYER       (-7)   /* a layer violates the Window::add_layer contract (src/window.cpp:49-58): the reference's fatal error */

#ifdef __cplusplus
}
#endif
#endif /* RACON_HIP_H_ */
