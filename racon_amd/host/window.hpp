// window.hpp — racon::Window (reference src/window.hpp:19-76, src/window.cpp):
// a backbone slice plus the read fragments ("layers") that overlap it, all as
// BORROWED (pointer, length) pairs, and the consensus string once polished.
// The consensus itself is computed on the MI355X: generate_consensus() takes the
// HIP engine where the reference takes a spoa::AlignmentEngine.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace racon {

enum class WindowType { kNGS, kTGS };      // reference src/window.hpp:21-24

class HipEngine;
class Window;
struct PackedBatch;
struct WindowRefs;

std::shared_ptr<Window> createWindow(uint64_t id, uint32_t rank, WindowType type, const char* backbone,
                                     uint32_t backbone_length, const char* quality, uint32_t quality_length);

class Window {
public:
    uint64_t id() const { return id_; }
    uint32_t rank() const { return rank_; }
    WindowType type() const { return type_; }
    const std::string& consensus() const { return consensus_; }
    size_t num_sequences() const { return sequences_.size(); }

    // One-window form of the hot path (reference src/window.cpp:65-149), run on the GPU.
    bool generate_consensus(std::shared_ptr<HipEngine> engine, bool trim);

    // reference src/window.cpp:42-63
    void add_layer(const char* sequence, uint32_t sequence_length, const char* quality, uint32_t quality_length,
                   uint32_t begin, uint32_t end);

    friend std::shared_ptr<Window> createWindow(uint64_t, uint32_t, WindowType, const char*, uint32_t, const char*, uint32_t);
    friend struct PackedBatch;      // reads the borrowed pointers (as CUDABatchProcessor does, window.hpp:58-60)
    friend struct WindowRefs;       // ... and hands them to the engine as they are
    friend class Polisher;          // writes consensus_ back (as cudabatch.cpp:221,230,253 does)

private:
    Window(uint64_t id, uint32_t rank, WindowType type, const char* backbone, uint32_t backbone_length,
           const char* quality, uint32_t quality_length);
    Window(const Window&) = delete;
    Window& operator=(const Window&) = delete;

    uint64_t id_; uint32_t rank_; WindowType type_;
    std::string consensus_;
    std::vector<std::pair<const char*, uint32_t>> sequences_;
    std::vector<std::pair<const char*, uint32_t>> qualities_;
    std::vector<std::pair<uint32_t, uint32_t>> positions_;
};

}  // namespace racon
