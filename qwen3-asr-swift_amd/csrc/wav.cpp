// wav.cpp -- PCM16 WAV reader (harness input).  Restates AudioFileLoader.loadWAV
// (Sources/AudioCommon/AudioFileLoader.swift:70-157) including its bounds checks, which the reference pins
// with Tests/Qwen3ASRTests/SecurityHardeningTests.swift:83-196: header > 44 bytes, "RIFF"/"WAVE" tags, PCM
// (format 1), channels > 0, 16 bits, chunk walk from offset 36 with overflow-safe advance, data chunk fully
// inside the file, first channel only, sample = int16 / 32768.
#include "qasr.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

extern "C" int qasr_load_wav(const char* path, float** samples, size_t* n_samples, int* sample_rate) {
    if (!path || !samples || !n_samples || !sample_rate) return QASR_ERR_INVALID;
    *samples = nullptr;
    *n_samples = 0;
    try {
    FILE* f = std::fopen(path, "rb");
    if (!f) return QASR_ERR_IO;
    std::vector<uint8_t> d;
    uint8_t buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + got);
    std::fclose(f);
    const size_t size = d.size();
    if (size <= 44) return QASR_ERR_IO;                                   // :74-76
    if (std::memcmp(d.data(), "RIFF", 4) != 0 || std::memcmp(d.data() + 8, "WAVE", 4) != 0) return QASR_ERR_IO;
    const uint16_t fmt = rd16(&d[20]), channels = rd16(&d[22]), bits = rd16(&d[34]);
    const uint32_t rate = rd32(&d[24]);
    if (fmt != 1 || channels == 0 || bits != 16) return QASR_ERR_IO;      // :96-106
    size_t off = 36;
    bool found = false;
    uint32_t chunk = 0;
    while (off + 8 < size) {                                              // :111 (dataOffset < count - 8)
        const uint32_t csz = rd32(&d[off + 4]);
        if (std::memcmp(&d[off], "data", 4) == 0) { off += 8; chunk = csz; found = true; break; }
        const uint64_t next = (uint64_t)off + 8 + csz;                    // :121-125 overflow-safe advance
        if (next > size) return QASR_ERR_IO;
        off = (size_t)next;
    }
    if (!found) return QASR_ERR_IO;
    if (off > size || (uint64_t)off + chunk > size) return QASR_ERR_IO;   // :133-135
    const size_t frame = 2u * channels, count = chunk / frame;
    float* out = (float*)std::malloc((count ? count : 1) * sizeof(float));
    if (!out) return QASR_ERR_INVALID;
    for (size_t i = 0; i < count; ++i) out[i] = (float)(int16_t)rd16(&d[off + i * frame]) / 32768.0f;
    *samples = out;
    *n_samples = count;
    *sample_rate = (int)rate;
    return QASR_OK;
    } catch (...) { return QASR_ERR_IO; }          // std::bad_alloc on a huge file: nothing crosses the C ABI
}

extern "C" void qasr_free(void* p) { std::free(p); }
