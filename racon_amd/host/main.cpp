// main.cpp — `racon_hip`: racon's command line (reference src/main.cpp) in front of the
// MI355X consensus engine.  Same positional arguments, same options and defaults,
// same FASTA on stdout.  -c/--cudapoa-batches [n] keeps its spelling and now means
// "HIP engines (batches in flight) per device"; --cudaaligner-batches n > 0 moves the
// overlap alignment to the device as it does in the reference (src/main.cpp:125-127 ->
// src/polisher.cpp:137-147); -b and --cudaaligner-band-width are accepted and ignored
// (the device DP is exact, there is no band to set).  The consensus stage always runs
// on the GPU.
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "polisher.hpp"
#include "sequence.hpp"

namespace {
const char* kVersion = "1.5.0-mi355x";

void help() {
    printf(
        "usage: racon_hip [options ...] <sequences> <overlaps> <target sequences>\n"
        "\n"
        "    #default output is stdout\n"
        "    <sequences>\n"
        "        input file in FASTA/FASTQ format (can be compressed with gzip)\n"
        "        containing sequences used for correction\n"
        "    <overlaps>\n"
        "        input file in MHAP/PAF/SAM format (can be compressed with gzip)\n"
        "        containing overlaps between sequences and target sequences\n"
        "    <target sequences>\n"
        "        input file in FASTA/FASTQ format (can be compressed with gzip)\n"
        "        containing sequences which will be corrected\n"
        "\n"
        "    options:\n"
        "        -u, --include-unpolished\n"
        "            output unpolished target sequences\n"
        "        -f, --fragment-correction\n"
        "            perform fragment correction instead of contig polishing\n"
        "            (overlaps file should contain dual/self overlaps!)\n"
        "        -w, --window-length <int>\n"
        "            default: 500\n"
        "            size of window on which POA is performed\n"
        "        -q, --quality-threshold <float>\n"
        "            default: 10.0\n"
        "            threshold for average base quality of windows used in POA\n"
        "        -e, --error-threshold <float>\n"
        "            default: 0.3\n"
        "            maximum allowed error rate used for filtering overlaps\n"
        "        --no-trimming\n"
        "            disables consensus trimming at window ends\n"
        "        -m, --match <int>\n"
        "            default: 3\n"
        "            score for matching bases\n"
        "        -x, --mismatch <int>\n"
        "            default: -5\n"
        "            score for mismatching bases\n"
        "        -g, --gap <int>\n"
        "            default: -4\n"
        "            gap penalty (must be negative)\n"
        "        -t, --threads <int>\n"
        "            default: 1\n"
        "            number of host threads (parsing, overlap pre-alignment)\n"
        "        -c, --cudapoa-batches <int>\n"
        "            default: 1\n"
        "            number of MI355X consensus engines per device\n"
        "        --cudaaligner-batches <int>\n"
        "            default: 0\n"
        "            > 0: overlaps without a CIGAR are aligned on the MI355X (exact, same\n"
        "            paths as the host pre-alignment) and the windows are built there\n"
        "        -b, --cuda-banded-alignment / --cudaaligner-band-width <int>\n"
        "            accepted and ignored (the device DP is exact, unbanded)\n"
        "        --version\n"
        "            prints the version number\n"
        "        -h, --help\n"
        "            prints the usage\n");
}
}  // namespace

int main(int argc, char** argv) {
    static const struct option options[] = {
        {"include-unpolished", no_argument, 0, 'u'}, {"fragment-correction", no_argument, 0, 'f'},
        {"window-length", required_argument, 0, 'w'}, {"quality-threshold", required_argument, 0, 'q'},
        {"error-threshold", required_argument, 0, 'e'}, {"no-trimming", no_argument, 0, 'T'},
        {"match", required_argument, 0, 'm'}, {"mismatch", required_argument, 0, 'x'}, {"gap", required_argument, 0, 'g'},
        {"threads", required_argument, 0, 't'}, {"version", no_argument, 0, 'v'}, {"help", no_argument, 0, 'h'},
        {"cudapoa-batches", optional_argument, 0, 'c'}, {"cuda-banded-alignment", no_argument, 0, 'b'},
        {"cudaaligner-batches", required_argument, 0, 10000}, {"cudaaligner-band-width", required_argument, 0, 10001},
        {0, 0, 0, 0}};
    uint32_t window_length = 500, type = 0, num_threads = 1, hip_batches = 1, hipaligner_batches = 0, hipaligner_band_width = 0;
    bool hip_banded_alignment = false;
    double quality_threshold = 10.0, error_threshold = 0.3;
    bool trim = true, drop_unpolished_sequences = true;
    int8_t match = 3, mismatch = -5, gap = -4;

    int argument;
    while ((argument = getopt_long(argc, argv, "ufw:q:e:m:x:g:t:hbc::", options, nullptr)) != -1) {
        switch (argument) {
            case 'u': drop_unpolished_sequences = false; break;
            case 'f': type = 1; break;
            case 'w': window_length = atoi(optarg); break;
            case 'q': quality_threshold = atof(optarg); break;
            case 'e': error_threshold = atof(optarg); break;
            case 'T': trim = false; break;
            case 'm': match = atoi(optarg); break;
            case 'x': mismatch = atoi(optarg); break;
            case 'g': gap = atoi(optarg); break;
            case 't': num_threads = atoi(optarg); break;
            case 'v': printf("%s\n", kVersion); return 0;
            case 'h': help(); return 0;
            case 'c':
                hip_batches = 1;
                if (optarg == nullptr && argv[optind] != nullptr && argv[optind][0] != '-') hip_batches = atoi(argv[optind++]);
                if (optarg != nullptr) hip_batches = atoi(optarg);
                break;
            case 'b': hip_banded_alignment = true; break;
            case 10000: hipaligner_batches = atoi(optarg); break;
            case 10001: hipaligner_band_width = atoi(optarg); break;
            default: return 1;
        }
    }
    std::vector<std::string> input_paths;
    for (int i = optind; i < argc; ++i) input_paths.emplace_back(argv[i]);
    if (input_paths.size() < 3) {
        fprintf(stderr, "[racon::] error: missing input file(s)!\n");
        help();
        return 1;
    }
    auto polisher = racon::createPolisher(input_paths[0], input_paths[1], input_paths[2],
        type == 0 ? racon::PolisherType::kC : racon::PolisherType::kF, window_length, quality_threshold, error_threshold,
        trim, match, mismatch, gap, num_threads, hip_batches, hip_banded_alignment, hipaligner_batches, hipaligner_band_width);
    // windows are built in HBM at the end of initialize() when everything fits the device(s) (RACON_HIP_DEVICE_WINDOWS=0: on the host)
    polisher->set_default_device_mode("auto");
    polisher->initialize();
    std::vector<std::unique_ptr<racon::Sequence>> polished_sequences;
    polisher->polish(polished_sequences, drop_unpolished_sequences);
    for (const auto& it : polished_sequences) fprintf(stdout, ">%s\n%s\n", it->name().c_str(), it->data().c_str());
    return 0;
}
