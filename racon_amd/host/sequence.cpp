#include "sequence.hpp"

#include <cctype>

namespace racon {

std::unique_ptr<Sequence> createSequence(const std::string& name, const std::string& data) {
    return std::unique_ptr<Sequence>(new Sequence(name, data));
}
std::unique_ptr<Sequence> createSequence(const std::string& name, std::string&& data) {
    return std::unique_ptr<Sequence>(new Sequence(name, std::move(data)));
}

Sequence::Sequence(const char* name, uint32_t name_length, const char* data, uint32_t data_length)
        : name_(name, name_length), data_(data, data_length) {
    for (char& c : data_) c = static_cast<char>(toupper(static_cast<unsigned char>(c)));
}

Sequence::Sequence(const char* name, uint32_t name_length, const char* data, uint32_t data_length,
                   const char* quality, uint32_t quality_length)
        : Sequence(name, name_length, data, data_length) {
    uint32_t informative = 0;                      // uint32 sum of (q - '!'), as the reference does
    for (uint32_t i = 0; i < quality_length; ++i) informative += quality[i] - '!';
    if (informative > 0) quality_.assign(quality, quality_length);
}

Sequence::Sequence(const std::string& name, const std::string& data) : name_(name), data_(data) {}
Sequence::Sequence(const std::string& name, std::string&& data) : name_(name), data_(std::move(data)) {}

void Sequence::create_reverse_complement() {
    if (!reverse_complement_.empty()) return;
    static const struct Table {
        char t[256];
        Table() { for (int i = 0; i < 256; ++i) t[i] = static_cast<char>(i); t['A'] = 'T'; t['T'] = 'A'; t['C'] = 'G'; t['G'] = 'C'; }
    } comp;
    reverse_complement_.assign(data_.rbegin(), data_.rend());
    for (char& c : reverse_complement_) c = comp.t[static_cast<unsigned char>(c)];
    reverse_quality_.assign(quality_.rbegin(), quality_.rend());
}

void Sequence::transmute(bool has_name, bool has_data, bool has_reverse_data) {
    if (!has_name) std::string().swap(name_);
    if (has_reverse_data) create_reverse_complement();
    if (!has_data) { std::string().swap(data_); std::string().swap(quality_); }
}

}  // namespace racon
