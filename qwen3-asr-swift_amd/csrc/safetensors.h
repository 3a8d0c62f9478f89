// safetensors.h -- sharded safetensors directory reader shared by the Qwen3-ASR and the Omnilingual loaders.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace qasr {

struct SafeEntry {
    std::string dtype;                 // "F32" | "F16" | "BF16" | "U32" | ...
    std::vector<int64_t> shape;
    const uint8_t* data;               // inside a mapping owned by the SafeTensorsDir
    size_t bytes;
    size_t numel() const { size_t n = 1; for (auto d : shape) n *= (size_t)d; return n; }
};

// element i of a float tensor (F32 / F16 / BF16) as f32
float safe_elem_f32(const SafeEntry& e, size_t i);

class SafeTensorsDir {
public:
    explicit SafeTensorsDir(const std::string& dir);     // maps every *.safetensors of the directory, validates the headers
    ~SafeTensorsDir();
    std::map<std::string, SafeEntry> entries;
private:
    struct Mapping;
    std::vector<std::unique_ptr<Mapping>> maps_;
};

}  // namespace qasr
