/*
 * racon_host.h — C ABI of the host layer around the MI355X consensus engine
 * (racon_amd/host, libracon_host.so): racon's Polisher surface
 * (reference src/polisher.hpp:42-57: createPolisher / initialize / polish) for
 * callers that cannot use the C++ classes directly (the Python bindings and the
 * parity harness).  CPU-only code; the consensus stage itself is include/racon_hip.h.
 *
 * The two halves of Polisher::polish are also exposed separately —
 *   rcnh_polisher_windows : every racon::Window flattened into an rcn_batch
 *   rcnh_polisher_assemble: per-target stitching + tags from per-window results
 * — so that any consensus backend (the HIP engine in production, the CPU oracle
 * in tests/) can be run on exactly the same window bytes.
 *
 * Errors: functions return 0 on success, <0 on failure; rcnh_last_error() gives
 * the message the reference would have printed before exit(1).
 */
#ifndef RACON_HOST_H_
#define RACON_HOST_H_

#include <stdint.h>
#include "racon_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rcnh_polisher rcnh_polisher;

typedef struct rcnh_params {
    uint32_t type;               /* 0 = kC contig polishing, 1 = kF fragment correction (polisher.hpp:28-31) */
    uint32_t window_length;      /* -w */
    double   quality_threshold;  /* -q */
    double   error_threshold;    /* -e */
    uint8_t  trim;               /* !--no-trimming */
    int8_t   match, mismatch, gap;
    uint32_t num_threads;        /* -t */
    uint32_t hip_batches;        /* -c: engines per device (0 = 1) */
} rcnh_params;

/* createPolisher (reference src/polisher.cpp:57-163) */
int  rcnh_polisher_create(const char* sequences_path, const char* overlaps_path, const char* target_path,
                          const rcnh_params* params, rcnh_polisher** out);
/* Polisher::initialize (reference src/polisher.cpp:190-464) */
int  rcnh_polisher_initialize(rcnh_polisher* p);
/* All windows as one packed batch; pointers stay valid until assemble/destroy. */
int  rcnh_polisher_windows(rcnh_polisher* p, rcn_batch* out);
/* The flattened input of those windows -- sequences (forward strand) and kept overlaps with their breaking points,
 * i.e. what reference src/polisher.cpp:388-461 loops over -- for rcn_engine_build_windows (include/racon_hip.h).
 * rcnh_polisher_keep_layout(p, 1) must precede rcnh_polisher_initialize; pointers stay valid until destroy. */
int  rcnh_polisher_keep_layout(rcnh_polisher* p, int on);
int  rcnh_polisher_layout(rcnh_polisher* p, rcn_read_set* reads, rcn_overlap_set* overlaps, uint8_t* window_type,
                          uint32_t* window_length, double* quality_threshold);
/* ... and the alignments those breaking points were derived from (the CIGAR of the overlap file or of the host's
 * pairwise alignment), for rcn_engine_build_windows_from_cigars. */
int  rcnh_polisher_alignments(rcnh_polisher* p, rcn_cigar_set* alignments);
/* ... and, for overlaps that came without an alignment (PAF / MHAP), the segment pairs the pre-alignment works on
 * (reference src/overlap.cpp:176-224): the input of rcn_engine_align_pairs / rcn_engine_build_windows_from_pairs. */
int  rcnh_polisher_pairs(rcnh_polisher* p, rcn_pair_set* pairs);
/* Stitch per-window results (same order as the batch) into FASTA text
 * ">name tags\nsequence\n..." exactly as reference src/main.cpp:159-161 prints it. */
int  rcnh_polisher_assemble(rcnh_polisher* p, const rcn_result* results, int drop_unpolished_sequences,
                            const char** fasta, uint64_t* fasta_length);
/* Polisher::polish on the MI355X (loads libracon_hip.so; fails without a device) + FASTA text. */
int  rcnh_polisher_polish(rcnh_polisher* p, int drop_unpolished_sequences, const char** fasta, uint64_t* fasta_length);
/* Seconds of the last rcnh_polisher_polish: the interval the reference's Logger brackets around Polisher::polish
 * (reference src/polisher.cpp:493 -> :539-543, "[racon::Polisher::polish] generated consensus") -- the interval
 * "polished windows / second" is defined on.  Window count of the job alongside.               */
double   rcnh_polisher_polish_seconds(rcnh_polisher* p);
/* How the last rcnh_polisher_polish cut its job (host-built windows): chunks of the deepest-first work list and the engines
 * that took at least one (reference: the ranges CUDAPolisher hands its per-device batches, src/cuda/cudapolisher.cpp:254-276). */
int      rcnh_polisher_polish_plan(rcnh_polisher* p, uint32_t* chunks, uint32_t* engines_used);
uint64_t rcnh_polisher_num_windows(rcnh_polisher* p);     /* valid between initialize and polish/assemble */
/* The device-built path's PLAN for a job cut into n_shards window ranges (racon_amd/host/device_job.cpp; the reference hands window
 * ranges to per-device batches, src/cuda/cudapolisher.cpp:228-240) -- pure host code, no device needed; rcnh_polisher_keep_layout +
 * initialize first.  cut[n + 1]: shard s owns windows [cut[s], cut[s + 1]); target_lo / target_hi[n]: the targets those windows lie in;
 * n_overlaps[n]: overlaps of each shard (one across a boundary counts on both sides).  Arrays of the caller (n = the return value <=
 * n_shards must fit: size them for n_shards), any may be NULL.  Returns the number of shards planned, < 0 on error. */
int  rcnh_polisher_device_plan(rcnh_polisher* p, uint32_t n_shards, uint64_t* cut, uint64_t* target_lo, uint64_t* target_hi, uint64_t* n_overlaps);
/* ... and ONE shard's input as its engine gets it (rcn_engine_build_windows*): the shard's targets first, then the reads its overlaps
 * point into, re-numbered; its overlaps' breaking points / alignments / segment pairs.  The engine numbers its windows from the
 * shard's first target: local window l is the job's window window_base + l.  Pointers stay valid until the next call / destroy. */
typedef struct rcnh_shard_dims {
    uint64_t window_first, window_last;   /* the shard's windows [first, last) in the job's numbering */
    uint64_t window_base, n_windows_local;
} rcnh_shard_dims;
int  rcnh_polisher_shard_input(rcnh_polisher* p, uint32_t n_shards, uint32_t shard, rcnh_shard_dims* dims, rcn_read_set* reads,
                               rcn_overlap_set* overlaps, rcn_cigar_set* alignments, rcn_pair_set* pairs);
void rcnh_polisher_destroy(rcnh_polisher* p);

/* Pairwise global alignment used for overlaps without CIGAR (reference src/overlap.cpp:205-224).
 * Writes a NUL-terminated CIGAR into a malloc'ed buffer (*cigar, release with rcnh_free). */
int  rcnh_align_cigar(const char* query, uint32_t query_length, const char* target, uint32_t target_length, char** cigar);
/* Global edit distance (the test helper of reference test/racon_test.cpp:14-23). */
uint64_t rcnh_edit_distance(const char* query, uint64_t query_length, const char* target, uint64_t target_length);
void rcnh_free(void* p);

const char* rcnh_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RACON_HOST_H_ */
