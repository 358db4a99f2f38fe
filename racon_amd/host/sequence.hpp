// sequence.hpp — host data model of one read / target, mirroring racon::Sequence
// (reference src/sequence.hpp:26-76, src/sequence.cpp:19-105): upper-cased bases,
// quality kept only when it carries information, lazily built reverse complement.
#pragma once
#include <cstdint>
#include <memory>
#include <string>

namespace racon {

class Sequence;
std::unique_ptr<Sequence> createSequence(const std::string& name, const std::string& data);
std::unique_ptr<Sequence> createSequence(const std::string& name, std::string&& data);     // (takes the buffer over)

class Sequence {
public:
    // FASTA record (reference src/sequence.cpp:19-28)
    Sequence(const char* name, uint32_t name_length, const char* data, uint32_t data_length);
    // FASTQ record (reference src/sequence.cpp:30-42): an all-'!' quality string means "no quality"
    Sequence(const char* name, uint32_t name_length, const char* data, uint32_t data_length,
             const char* quality, uint32_t quality_length);
    Sequence(const std::string& name, const std::string& data);
    Sequence(const std::string& name, std::string&& data);
    Sequence(const Sequence&) = delete;
    Sequence& operator=(const Sequence&) = delete;

    const std::string& name() const { return name_; }
    const std::string& data() const { return data_; }
    const std::string& reverse_complement() const { return reverse_complement_; }
    const std::string& quality() const { return quality_; }
    const std::string& reverse_quality() const { return reverse_quality_; }

    void create_reverse_complement();                                   // src/sequence.cpp:49-84
    void transmute(bool has_name, bool has_data, bool has_reverse_data); // src/sequence.cpp:86-103
    // (not in the reference) drops bases and qualities of both strands: the device-windows path keeps the reads in its
    // flattened layout only (Polisher::initialize)
    void release_data() { std::string().swap(data_); std::string().swap(quality_); std::string().swap(reverse_complement_); std::string().swap(reverse_quality_); }

private:
    std::string name_, data_, reverse_complement_, quality_, reverse_quality_;
};

}  // namespace racon
