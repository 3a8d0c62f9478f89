// safetensors.cpp -- checkpoint directory loader (HF safetensors shards, reference key names).
// Reference: Sources/Qwen3ASR/WeightLoading.swift:17-126,235-323, Sources/MLXCommon/WeightLoading.swift:48-165.
//   audio_tower.*  matrices (conv / linear weights; f32 / f16 / bf16 on disk) -> bf16 in HBM: they are MFMA operands, the one
//                  place an f16 / f32 checkpoint loses bits (the stated bf16-operand deviation);
//                  vectors (biases, LayerNorm gain / shift) -> f32, exact for every checkpoint dtype: the reference runs
//                  the encoder in f32 on the widened tensors (tests/test_oracle_loader_precision.py prices both)
//   model.*        either float Linear weights (FloatTextDecoder) or MLX affine-quantised triplets
//                  {weight: uint32 [out, in*bits/32], scales, biases: [out, in/group]}, uploaded AS THEY ARE: the packed
//                  words stay packed in HBM (csrc/dec_quant.h); scales / biases keep bf16, f16 / f32 ones are widened to
//                  f32 (exact) so no checkpoint value is rounded.
#include "engine.h"
#include "json.h"
#include "safetensors.h"
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <cstring>

namespace qasr {

struct SafeTensorsDir::Mapping {
    void* p = MAP_FAILED;
    size_t n = 0;
    int fd = -1;
    explicit Mapping(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error("cannot stat " + path); }
        n = (size_t)st.st_size;
        p = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) { ::close(fd); throw std::runtime_error("cannot mmap " + path); }
    }
    ~Mapping() {
        if (p != MAP_FAILED) munmap(p, n);
        if (fd >= 0) ::close(fd);
    }
};

namespace {
float half_to_float(uint16_t h) {
    uint32_t sign = (h & 0x8000u) << 16, exp = (h >> 10) & 0x1F, man = h & 0x3FF, out;
    if (exp == 0) {
        if (man == 0) out = sign;
        else {
            int e = -1;
            do { ++e; man <<= 1; } while (!(man & 0x400));
            out = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FF) << 13);
        }
    } else if (exp == 31) out = sign | 0x7F800000u | (man << 13);
    else out = sign | ((exp + 112) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &out, 4);
    return f;
}
}  // namespace

float safe_elem_f32(const SafeEntry& e, size_t i) {
    if (e.dtype == "F32") { float f; std::memcpy(&f, e.data + 4 * i, 4); return f; }
    uint16_t h;
    std::memcpy(&h, e.data + 2 * i, 2);
    if (e.dtype == "BF16") return bf16_to_f32(h);
    if (e.dtype == "F16") return half_to_float(h);
    throw std::runtime_error("unsupported float dtype " + e.dtype);
}

SafeTensorsDir::~SafeTensorsDir() = default;

SafeTensorsDir::SafeTensorsDir(const std::string& dir) {
    std::vector<std::string> files;
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* de = readdir(d)) {
            std::string f = de->d_name;
            if (f.size() > 12 && f.compare(f.size() - 12, 12, ".safetensors") == 0) files.push_back(dir + "/" + f);
        }
        closedir(d);
    } else throw std::runtime_error("cannot open model directory " + dir);
    if (files.empty()) throw std::runtime_error("no .safetensors files in " + dir);
    std::sort(files.begin(), files.end());
    for (auto& f : files) {
        maps_.push_back(std::make_unique<Mapping>(f));
        const uint8_t* base = (const uint8_t*)maps_.back()->p;
        const size_t n = maps_.back()->n;
        if (n < 8) throw std::runtime_error("truncated safetensors file " + f);
        uint64_t hlen;
        std::memcpy(&hlen, base, 8);
        if (hlen > n - 8) throw std::runtime_error("bad safetensors header length in " + f);
        Json h = JsonParser((const char*)base + 8, (size_t)hlen).parse();
        const uint8_t* data = base + 8 + hlen;
        const size_t data_n = n - 8 - (size_t)hlen;
        for (auto& kv : h.obj) {
            if (kv.first == "__metadata__") continue;
            const Json *dt = kv.second.get("dtype"), *sh = kv.second.get("shape"), *off = kv.second.get("data_offsets");
            if (!dt || !sh || !off || off->arr.size() != 2) throw std::runtime_error("bad tensor entry " + kv.first);
            SafeEntry e;
            e.dtype = dt->str;
            for (auto& d : sh->arr) {
                if (d.type != Json::Num || d.num < 0 || d.num > 9.0e15 || d.num != (double)(int64_t)d.num)
                    throw std::runtime_error("tensor " + kv.first + ": bad shape entry");
                e.shape.push_back((int64_t)d.num);
            }
            if (e.shape.size() > 8) throw std::runtime_error("tensor " + kv.first + ": too many dimensions");
            {   // the element count must not overflow: every later size is a product of these
                unsigned __int128 prod = 1;
                for (auto d : e.shape) { prod *= (unsigned __int128)d; if (prod > ((unsigned __int128)1 << 48)) throw std::runtime_error("tensor " + kv.first + ": shape too large"); }
            }
            if (off->arr[0].type != Json::Num || off->arr[1].type != Json::Num || off->arr[0].num < 0 || off->arr[1].num < 0)
                throw std::runtime_error("tensor " + kv.first + ": bad data_offsets");
            size_t b = (size_t)off->arr[0].num, en = (size_t)off->arr[1].num;
            if (b > en || en > data_n) throw std::runtime_error("tensor " + kv.first + " out of file bounds");
            e.data = data + b;
            e.bytes = en - b;
            entries[kv.first] = e;
        }
    }
}

void Engine::load_directory(const std::string& dir) {
    SafeTensorsDir st(dir);
    std::map<std::string, SafeEntry>& entries = st.entries;
    typedef SafeEntry Entry;
    auto elem_f32 = [](const Entry& e, size_t i) { return safe_elem_f32(e, i); };
    std::vector<bf16_t> tmp;
    auto upload_float = [&](const std::string& name, const Entry& e) {
        size_t numel = 1;
        for (auto d : e.shape) numel *= (size_t)d;
        size_t el = e.dtype == "F32" ? 4 : 2;
        if (numel * el != e.bytes) throw std::runtime_error("tensor " + name + ": byte size does not match shape");
        if (e.dtype == "BF16") { set_tensor(name, e.data, QASR_DTYPE_BF16, e.shape.data(), (int)e.shape.size()); return; }
        if (e.shape.size() == 1 && name.compare(0, 12, "audio_tower.") == 0) {     // encoder vector: keep every bit
            std::vector<float> wide(numel);
            for (size_t i = 0; i < numel; ++i) wide[i] = elem_f32(e, i);
            set_tensor(name, wide.data(), QASR_DTYPE_F32, e.shape.data(), 1);
            return;
        }
        tmp.resize(numel);
        for (size_t i = 0; i < numel; ++i) tmp[i] = f32_to_bf16_host(elem_f32(e, i));
        set_tensor(name, tmp.data(), QASR_DTYPE_BF16, e.shape.data(), (int)e.shape.size());
    };
    // Forced-aligner checkpoints (WeightLoader.loadForcedAlignerWeights, WeightLoading.swift:135-232) prefix every key
    // with "thinker.", keep the Conv2d weights in PyTorch layout [out, in, kH, kW] (transposed to the engine's
    // [out, kH, kW, in] here, :184-187) and carry the un-quantised classify head as `lm_head.{weight,bias}`.
    const bool aligner = cfg_.classify_num > 0;
    if (aligner) {
        std::map<std::string, Entry> stripped;
        for (auto& kv : entries)
            stripped[kv.first.compare(0, 8, "thinker.") == 0 ? kv.first.substr(8) : kv.first] = kv.second;
        entries.swap(stripped);
    }
    for (auto& kv : entries) {
        const std::string& name = kv.first;
        const bool audio = name.compare(0, 12, "audio_tower.") == 0, text = name.compare(0, 6, "model.") == 0;
        const bool head = aligner && name.compare(0, 8, "lm_head.") == 0;
        if (!audio && !text && !head) continue;
        if (aligner && audio && name.find(".conv2d") != std::string::npos && name.size() > 7 &&
            name.compare(name.size() - 7, 7, ".weight") == 0) {
            const Entry& e = kv.second;
            if (e.shape.size() != 4 || e.shape[2] != 3 || e.shape[3] != 3) throw std::runtime_error(name + ": expected a PyTorch [out, in, 3, 3] conv weight");
            const int64_t O = e.shape[0], I = e.shape[1];
            const size_t el = e.dtype == "F32" ? 4 : 2;
            if ((size_t)(O * I * 9) * el != e.bytes) throw std::runtime_error("tensor " + name + ": byte size does not match shape");
            tmp.resize((size_t)(O * I * 9));
            for (int64_t o = 0; o < O; ++o)
                for (int64_t i = 0; i < I; ++i)
                    for (int64_t k = 0; k < 9; ++k)
                        tmp[(size_t)((o * 9 + k) * I + i)] = f32_to_bf16_host(elem_f32(e, (size_t)((o * I + i) * 9 + k)));
            int64_t shp[4] = {O, 3, 3, I};
            set_tensor(name, tmp.data(), QASR_DTYPE_BF16, shp, 4);
            continue;
        }
        auto ends = [&](const char* s) { size_t l = strlen(s); return name.size() > l && name.compare(name.size() - l, l, s) == 0; };
        if (ends(".scales") || ends(".biases")) continue;            // consumed with their .weight
        const Entry& e = kv.second;
        if (e.dtype == "U32" && ends(".weight")) {
            const std::string stem = name.substr(0, name.size() - 7);
            auto si = entries.find(stem + ".scales"), bi = entries.find(stem + ".biases");
            if (si == entries.end() || bi == entries.end()) throw std::runtime_error("quantised " + name + " lacks scales/biases");
            const int bits = cfg_.bits == 8 ? 8 : 4, per = 32 / bits, group = cfg_.group_size;
            if (group <= 0 || e.shape.size() != 2 || e.shape[0] <= 0 || e.shape[1] <= 0 || e.shape[0] > (1 << 24) || e.shape[1] > (1 << 24))
                throw std::runtime_error("quantised " + name + ": expected a 2-D uint32 tensor");
            const int64_t out = e.shape[0], in = e.shape[1] * per;
            if ((size_t)(out * e.shape[1] * 4) != e.bytes || in % group != 0)
                throw std::runtime_error("quantised " + name + ": byte size / group mismatch (bits?)");
            // scales and biases: 2-D [out, in / group], a float dtype, byte size == numel * element size (untrusted metadata)
            std::vector<float> wide;
            auto upload_sb = [&](const std::string& nm, const Entry& t) {
                if (t.shape != std::vector<int64_t>{out, in / group}) throw std::runtime_error(nm + ": expected [out, in / group]");
                const size_t numel = (size_t)out * (size_t)(in / group);
                const size_t el = t.dtype == "F32" ? 4 : (t.dtype == "BF16" || t.dtype == "F16") ? 2 : 0;
                if (el == 0) throw std::runtime_error(nm + ": scales / biases must be F32, F16 or BF16");
                if (numel * el != t.bytes) throw std::runtime_error(nm + ": byte size does not match shape");
                if (t.dtype == "BF16") { set_tensor(nm, t.data, QASR_DTYPE_BF16, t.shape.data(), 2); return; }
                wide.resize(numel);
                for (size_t i = 0; i < numel; ++i) wide[i] = elem_f32(t, i);          // f16 -> f32 is exact
                set_tensor(nm, wide.data(), QASR_DTYPE_F32, t.shape.data(), 2);
            };
            if (si->second.dtype != bi->second.dtype) throw std::runtime_error(stem + ": scales and biases differ in dtype");
            upload_sb(stem + ".scales", si->second);
            upload_sb(stem + ".biases", bi->second);
            set_tensor(name, e.data, QASR_DTYPE_U32, e.shape.data(), 2);
        } else {
            upload_float(name, e);
        }
    }
    load_vocab_files(dir);
}

}  // namespace qasr
