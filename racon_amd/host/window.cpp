#include "window.hpp"

#include <cstdio>

#include "fatal.hpp"
#include "hip_engine.hpp"

namespace racon {

std::shared_ptr<Window> createWindow(uint64_t id, uint32_t rank, WindowType type, const char* backbone,
                                     uint32_t backbone_length, const char* quality, uint32_t quality_length) {
    if (backbone_length == 0 || backbone_length != quality_length)
        fatal("[racon::createWindow] error: empty backbone sequence/unequal quality length!");
    return std::shared_ptr<Window>(new Window(id, rank, type, backbone, backbone_length, quality, quality_length));
}

Window::Window(uint64_t id, uint32_t rank, WindowType type, const char* backbone, uint32_t backbone_length,
               const char* quality, uint32_t quality_length)
        : id_(id), rank_(rank), type_(type) {
    sequences_.emplace_back(backbone, backbone_length);
    qualities_.emplace_back(quality, quality_length);
    positions_.emplace_back(0, 0);
}

void Window::add_layer(const char* sequence, uint32_t sequence_length, const char* quality, uint32_t quality_length,
                       uint32_t begin, uint32_t end) {
    if (sequence_length == 0 || begin == end) return;
    if (quality != nullptr && sequence_length != quality_length)
        fatal("[racon::Window::add_layer] error: unequal quality size!");
    const uint32_t backbone_length = sequences_.front().second;
    if (begin >= end || begin > backbone_length || end > backbone_length)
        fatal("[racon::Window::add_layer] error: layer begin and end positions are invalid!");
    sequences_.emplace_back(sequence, sequence_length);
    qualities_.emplace_back(quality, quality_length);
    positions_.emplace_back(begin, end);
}

bool Window::generate_consensus(std::shared_ptr<HipEngine> engine, bool trim) {
    if (!engine) fatal("[racon::Window::generate_consensus] error: no HIP engine (there is no CPU fallback)!");
    PackedBatch batch;
    batch.add(*this);
    std::vector<std::string> cons; std::vector<uint8_t> polished, chimeric;
    engine->consensus(batch, trim, &cons, &polished, &chimeric);
    consensus_.swap(cons[0]);
    if (chimeric[0])
        fprintf(stderr, "[racon::Window::generate_consensus] warning: contig %lu might be chimeric in window %u!\n",
                static_cast<unsigned long>(id_), rank_);
    return polished[0] != 0;
}

}  // namespace racon
