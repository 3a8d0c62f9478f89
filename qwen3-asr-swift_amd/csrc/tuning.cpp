// tuning.cpp -- knob table (tuning.h): name lookup, environment seeding.
#include "tuning.h"
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <string>

namespace qasr {

namespace {
struct Entry { const char* key; int Tuning::*field; };
const Entry kEntries[] = {
    {"gemv_splitb", &Tuning::gemv_splitb}, {"gemv_w1024", &Tuning::gemv_w1024}, {"gemv_wide", &Tuning::gemv_wide}, {"gemv_partial", &Tuning::gemv_partial},
    {"da_waves", &Tuning::da_waves}, {"da_spec", &Tuning::da_spec},
    {"pa_form", &Tuning::pa_form}, {"pa_mt", &Tuning::pa_mt}, {"pa_order", &Tuning::pa_order}, {"qknr_wide", &Tuning::qknr_wide},
    {"enc_attn", &Tuning::enc_attn}, {"mha_form", &Tuning::mha_form}, {"gemm_p8", &Tuning::gemm_p8}, {"gemm_nbuf", &Tuning::gemm_nbuf},
    {"lmh_q_ring", &Tuning::lmh_q_ring}, {"lmh_grid", &Tuning::lmh_grid}, {"lmh_diag", &Tuning::lmh_diag},
    {"decode_split", &Tuning::decode_split}, {"decode_gran", &Tuning::decode_gran}, {"graph_steps", &Tuning::graph_steps}, {"use_graph", &Tuning::use_graph}, {"device_sampler", &Tuning::device_sampler},
    {"da_stamps", &Tuning::da_stamps}, {"gemv_stamps", &Tuning::gemv_stamps}, {"stamps_insitu", &Tuning::stamps_insitu},
};
}  // namespace

Tuning& tuning() {
    static Tuning t = [] {
        Tuning v;
        for (const Entry& e : kEntries) {
            std::string name = "QASR_";
            for (const char* p = e.key; *p; ++p) name += (char)std::toupper((unsigned char)*p);
            if (const char* s = std::getenv(name.c_str())) v.*(e.field) = std::atoi(s);
        }
        return v;
    }();
    return t;
}

bool tuning_set(const char* key, int value) {
    for (const Entry& e : kEntries)
        if (std::strcmp(e.key, key) == 0) {
            Tuning& t = tuning();
            if (t.*(e.field) != value) { t.*(e.field) = value; ++t.epoch; }
            return true;
        }
    return false;
}

bool tuning_get(const char* key, int* value) {
    for (const Entry& e : kEntries)
        if (std::strcmp(e.key, key) == 0) { *value = tuning().*(e.field); return true; }
    return false;
}

}  // namespace qasr
