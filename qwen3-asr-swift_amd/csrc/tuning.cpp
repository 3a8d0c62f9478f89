// tuning.cpp -- knob table (tuning.h): name lookup, environment seeding.
#include "tuning.h"
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace qasr {

namespace {
// allowed: the enumerated values of a knob (terminated by -1), or {lo, hi, -2} for an inclusive range.  Anything else is refused:
// a value outside the table never reaches a kernel launch (decode_gran = 0 would divide by zero, lmh_grid <= 0 sizes empty buffers).
struct Entry { const char* key; int Tuning::*field; int allowed[6]; bool diag_only; };
const Entry kEntries[] = {
    {"gemv_splitb", &Tuning::gemv_splitb, {0, 1, 2, -1}, false}, {"gemv_xbar", &Tuning::gemv_xbar, {0, 1, 2, 3, 4, -1}, false}, {"gemv_w1024", &Tuning::gemv_w1024, {4, 8, -1}, false},
    {"gemv_nt", &Tuning::gemv_nt, {0, 1, -1}, false}, {"lmh_nt", &Tuning::lmh_nt, {0, 1, -1}, false},
    {"gemv_wide", &Tuning::gemv_wide, {0, 1, -1}, false}, {"gemv_earlyw", &Tuning::gemv_earlyw, {0, 1, 2, 3, -1}, false}, {"gemv_partial", &Tuning::gemv_partial, {0, 1, -1}, false},
    {"chain", &Tuning::chain, {0, 1, 2, 3, -1}, false}, {"chain_fault", &Tuning::chain_fault, {0, 1, -1}, false}, {"chain_proto", &Tuning::chain_proto, {0, 1, -1}, false}, {"chain_pf", &Tuning::chain_pf, {0, 1, -1}, false}, {"qa", &Tuning::qa, {0, 1, -1}, false},
    {"qa_early", &Tuning::qa_early, {0, 5, -2}, false}, {"qa_gate", &Tuning::qa_gate, {0, 1, 2, -1}, false}, {"qa_gran", &Tuning::qa_gran, {0, 1, -1}, false}, {"qa_split", &Tuning::qa_split, {0, 1, 2, -1}, false}, {"qa_xbar", &Tuning::qa_xbar, {0, 1, 2, -1}, false},
    {"da_unr", &Tuning::da_unr, {1, 2, -1}, false},
    {"da_waves", &Tuning::da_waves, {8, 16, -1}, false}, {"da_spec", &Tuning::da_spec, {0, 1, 2, 3, -1}, false}, {"da_earlyq", &Tuning::da_earlyq, {0, 1, -1}, false},
    {"pa_form", &Tuning::pa_form, {1, 2, -1}, false}, {"pa_mt", &Tuning::pa_mt, {1, 2, -1}, false}, {"pa_order", &Tuning::pa_order, {0, 1, -1}, false}, {"pa_vfrag", &Tuning::pa_vfrag, {0, 1, -1}, false},
    {"pp_fuse_qk", &Tuning::pp_fuse_qk, {0, 1, -1}, false}, {"qknr_wide", &Tuning::qknr_wide, {0, 1, -1}, false},
    {"conv_ktile", &Tuning::conv_ktile, {0, 1, -1}, false}, {"enc_attn", &Tuning::enc_attn, {0, 1, -1}, false}, {"mha_form", &Tuning::mha_form, {0, 1, 2, -1}, false},
    {"gemm_p8", &Tuning::gemm_p8, {0, 1, 2, -1}, false}, {"gemm_nbuf", &Tuning::gemm_nbuf, {0, 1, 2, -1}, false}, {"gemm_tm", &Tuning::gemm_tm, {1, 4, 8, 16, -1}, false},
    {"lmh_q_ring", &Tuning::lmh_q_ring, {0, 1, -1}, false}, {"lmh_grid", &Tuning::lmh_grid, {1, 4096, -2}, false}, {"lmh_order", &Tuning::lmh_order, {0, 1, -1}, false},
    {"lmh_diag", &Tuning::lmh_diag, {0, 1, -1}, true},
    {"decode_split", &Tuning::decode_split, {1, 4, -2}, false}, {"decode_gran", &Tuning::decode_gran, {16, 32, 48, 64, -1}, false},
    {"graph_steps", &Tuning::graph_steps, {1, 2, 4, 8, -1}, false}, {"use_graph", &Tuning::use_graph, {0, 1, -1}, false},
    {"device_sampler", &Tuning::device_sampler, {0, 1, -1}, false},
    {"da_stamps", &Tuning::da_stamps, {0, 1, -1}, true}, {"gemv_stamps", &Tuning::gemv_stamps, {0, 1, -1}, true},
    {"stamps_insitu", &Tuning::stamps_insitu, {0, 1, -1}, true}, {"pa_stamps", &Tuning::pa_stamps, {0, 1, -1}, true},
};

bool value_allowed(const Entry& e, int v) {
#ifndef QASR_DIAG_STAMPS
    if (e.diag_only && v != 0) return false;        // timing-only / wrong-result diagnostics exist in `make DIAG=1` builds only
#endif
    if (e.allowed[2] == -2) return v >= e.allowed[0] && v <= e.allowed[1];
    for (int i = 0; i < 6 && e.allowed[i] != -1; ++i)
        if (e.allowed[i] == v) return true;
    return false;
}
}  // namespace

static thread_local bool t_shared = false;
void tuning_thread_shared(bool shared) { t_shared = shared; }
bool tuning_thread_is_shared() { return t_shared; }

Tuning& tuning() {
    static Tuning t = [] {
        Tuning v;
        for (const Entry& e : kEntries) {
            std::string name = "QASR_";
            for (const char* p = e.key; *p; ++p) name += (char)std::toupper((unsigned char)*p);
            if (const char* s = std::getenv(name.c_str())) {
                const int x = std::atoi(s);
                if (value_allowed(e, x)) v.*(e.field) = x;
                else std::fprintf(stderr, "[qasr] %s=%s is not an allowed value, keeping %d\n", name.c_str(), s, v.*(e.field));
            }
        }
        return v;
    }();
    return t;
}

bool tuning_set(const char* key, int value) {
    for (const Entry& e : kEntries)
        if (std::strcmp(e.key, key) == 0) {
            if (!value_allowed(e, value)) return false;
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
