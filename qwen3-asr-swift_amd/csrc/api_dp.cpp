// api_dp.cpp -- utterance-batch data parallelism behind the C ABI (include/qasr.h: qasr_dp_*).
//
// The reference transcribes a list of files one after the other on one device (Sources/AudioCLILib/TranscribeBatchCommand.swift:82-93).
// Clips are independent (per-clip mel maximum, per-clip attention windows, per-clip KV cache: Qwen3ASR.swift:131-164), so the list shards
// over GPUs with no data-path exchange: one process, one engine (= one HIP device + one stream, weights replicated) and one host thread
// per GPU; clip block [lo, hi) of GPU i is the contiguous partition bench.py and qasr/dist.py use (the first B % n devices take one extra
// clip).  The only cross-device step is the gather of the decoded token streams: every engine copies its [rows, max_new_tokens + 1] int32
// block device -> host straight into the caller's [B, max_new_tokens + 1] buffer (57 KB per GPU at 32 clips: a host-memory caller has no
// use for a device-side all-gather; the multi-process form of the same partition, with the RCCL all_gather of that block, is bench.py).
#include "engine.h"
#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

// One whole batch in flight on one engine (qasr_dp_submit / qasr_dp_collect): the engine's host thread, the result block it fills and
// the status it ends with.  The caller's PCM buffers are read by that thread until the ticket is collected.
struct Lane {
    std::thread worker;
    int64_t ticket = -1;                  // -1: idle
    size_t B = 0;
    int rc = QASR_OK;
    std::vector<int32_t> tokens, lens;
};

struct qasr_dp {
    std::vector<qasr_engine*> engines;
    std::vector<float> last_ms;           // wall time of each engine's share of the last call
    std::vector<Lane> lanes;              // one per engine
    int64_t next_ticket = 0;
    std::string last_error;
};

static thread_local std::string g_dp_create_error;

static void shard_bounds(size_t n_clips, size_t world, size_t rank, size_t* lo, size_t* hi) {      // qasr/dist.py: shard_bounds
    const size_t base = n_clips / world, extra = n_clips % world;
    *lo = rank * base + std::min(rank, extra);
    *hi = *lo + base + (rank < extra ? 1 : 0);
}

extern "C" {

int qasr_dp_create(const char* model_dir, const qasr_config* cfg, const int32_t* devices, int32_t n_devices, qasr_dp** out) {
    if (!cfg || !out || n_devices <= 0 || n_devices > 64 || !devices) return QASR_ERR_INVALID;
    *out = nullptr;
    auto dp = std::make_unique<qasr_dp>();
    // engines are created one after the other (checkpoint parsing is host work; a parallel load would only contend for the same files)
    for (int i = 0; i < n_devices; ++i) {
        qasr_config c = *cfg;
        c.device = devices[i];
        qasr_engine* e = nullptr;
        const int rc = qasr_create(model_dir, &c, &e);
        if (rc != QASR_OK) {
            g_dp_create_error = std::string("device ") + std::to_string(devices[i]) + ": " + qasr_last_error(nullptr);
            for (qasr_engine* p : dp->engines) qasr_destroy(p);
            return rc;
        }
        dp->engines.push_back(e);
    }
    // engines that share a GPU run concurrently (qasr_dp_submit lanes): no launch of theirs may need its whole grid resident
    for (int i = 0; i < n_devices; ++i)
        if (std::count(devices, devices + n_devices, devices[i]) > 1) qasr_set_shared_device(dp->engines[(size_t)i], 1);
    dp->last_ms.assign((size_t)n_devices, 0.f);
    dp->lanes.resize((size_t)n_devices);
    *out = dp.release();
    return QASR_OK;
}

void qasr_dp_destroy(qasr_dp* dp) {
    if (!dp) return;
    for (Lane& l : dp->lanes)
        if (l.worker.joinable()) l.worker.join();          // a batch still in flight finishes before its engine goes away
    for (qasr_engine* e : dp->engines) qasr_destroy(e);
    delete dp;
}

int qasr_dp_n_devices(const qasr_dp* dp) { return dp ? (int)dp->engines.size() : 0; }
qasr_engine* qasr_dp_engine(qasr_dp* dp, int32_t i) { return (dp && i >= 0 && (size_t)i < dp->engines.size()) ? dp->engines[(size_t)i] : nullptr; }
const char* qasr_dp_last_error(const qasr_dp* dp) { return dp ? dp->last_error.c_str() : g_dp_create_error.c_str(); }

static int for_all(qasr_dp* dp, const char* what, int (*fn)(qasr_engine*, void*), void* arg) {
    for (size_t i = 0; i < dp->engines.size(); ++i) {
        const int rc = fn(dp->engines[i], arg);
        if (rc != QASR_OK) {
            dp->last_error = std::string(what) + " on engine " + std::to_string(i) + ": " + qasr_last_error(dp->engines[i]);
            return rc;
        }
    }
    return QASR_OK;
}

struct SetTensorArg { const char* name; const void* host; int dtype; const int64_t* shape; int ndim; };
int qasr_dp_set_tensor(qasr_dp* dp, const char* name, const void* host_data, int dtype, const int64_t* shape, int ndim) {
    if (!dp) return QASR_ERR_INVALID;
    SetTensorArg a{name, host_data, dtype, shape, ndim};
    return for_all(dp, "set_tensor", [](qasr_engine* e, void* p) {
        auto* a = static_cast<SetTensorArg*>(p);
        return qasr_set_tensor(e, a->name, a->host, a->dtype, a->shape, a->ndim);
    }, &a);
}
int qasr_dp_finalize(qasr_dp* dp) {
    if (!dp) return QASR_ERR_INVALID;
    return for_all(dp, "finalize", [](qasr_engine* e, void*) { return qasr_finalize(e); }, nullptr);
}

int qasr_dp_transcribe_batch(qasr_dp* dp, const float* const* pcm, const size_t* n, size_t B, int sample_rate, const qasr_options* opt,
                             int32_t* tokens, int32_t* lens) {
    if (!dp || dp->engines.empty() || (B && (!pcm || !n || !tokens || !lens))) return QASR_ERR_INVALID;
    if (B == 0) return QASR_OK;
    const size_t G = dp->engines.size();
    for (size_t g = 0; g < G; ++g)
        if (dp->lanes[g].ticket >= 0) {
            dp->last_error = "engine " + std::to_string(g) + " holds ticket " + std::to_string(dp->lanes[g].ticket) + ": collect it first";
            return QASR_ERR_INVALID;
        }
    std::vector<int> rc(G, QASR_OK);
    std::vector<std::thread> pool;
    auto work = [&](size_t g) {
        size_t lo, hi;
        shard_bounds(B, G, g, &lo, &hi);
        qasr_engine* e = dp->engines[g];
        const qasr_config& c = e->impl->config();
        const size_t stride = (size_t)c.max_new_tokens + 1, cap = (size_t)c.max_batch;
        const auto t0 = std::chrono::steady_clock::now();
        // a block larger than the engine's capacity goes through it in slices; each slice lands in the caller's rows directly
        for (size_t b0 = lo; b0 < hi && rc[g] == QASR_OK; b0 += cap) {
            const size_t nb = std::min(cap, hi - b0);
            rc[g] = qasr_transcribe_batch(e, pcm + b0, n + b0, nb, sample_rate, opt, tokens + b0 * stride, lens + b0);
        }
        dp->last_ms[g] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    try {
        for (size_t g = 1; g < G; ++g) pool.emplace_back(work, g);
    } catch (const std::exception& ex) {
        for (auto& t : pool) t.join();
        dp->last_error = std::string("cannot start worker threads: ") + ex.what();
        return QASR_ERR_INVALID;
    }
    work(0);                                   // the calling thread drives engine 0
    for (auto& t : pool) t.join();
    for (size_t g = 0; g < G; ++g)
        if (rc[g] != QASR_OK) {
            dp->last_error = std::string("engine ") + std::to_string(g) + " (device " + std::to_string(dp->engines[g]->impl->config().device) +
                             "): " + qasr_last_error(dp->engines[g]);
            return rc[g];
        }
    return QASR_OK;
}

// ---- whole batches in flight, one per engine ------------------------------------------------------------------------------------
// The decode stage of a pass is bound by the latency of its 142 dependent launches per token, not by bytes or flops: a
// second pass on the same GPU, on its own streams, runs in the gaps.  Engines listed on the same device are such lanes; batch k goes to
// engine k % n whole (one pass = one batch, as in the reference's loop, TranscribeBatchCommand.swift:82-93), so n passes are in flight.
int qasr_dp_submit(qasr_dp* dp, const float* const* pcm, const size_t* n, size_t B, int sample_rate, const qasr_options* opt,
                   int64_t* ticket) {
    if (!dp || dp->engines.empty() || !ticket || (B && (!pcm || !n))) return QASR_ERR_INVALID;
    const size_t g = (size_t)(dp->next_ticket % (int64_t)dp->engines.size());
    Lane& l = dp->lanes[g];
    if (l.ticket >= 0) {
        dp->last_error = "engine " + std::to_string(g) + " still holds ticket " + std::to_string(l.ticket) + ": collect it before the next submit";
        return QASR_ERR_INVALID;
    }
    qasr_engine* e = dp->engines[g];
    const qasr_config& c = e->impl->config();
    const size_t stride = (size_t)c.max_new_tokens + 1, cap = (size_t)c.max_batch;
    try {
        l.tokens.assign(B * stride, -1);
        l.lens.assign(B, 0);
        l.B = B;
        l.rc = QASR_OK;
        const qasr_options o = opt ? *opt : qasr_options{};
        const bool has_opt = opt != nullptr;
        l.worker = std::thread([dp, g, e, pcm, n, B, sample_rate, o, has_opt, stride, cap]() {
            Lane& l = dp->lanes[g];
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t b0 = 0; b0 < B && l.rc == QASR_OK; b0 += cap) {
                const size_t nb = std::min(cap, B - b0);
                l.rc = qasr_transcribe_batch(e, pcm + b0, n + b0, nb, sample_rate, has_opt ? &o : nullptr, l.tokens.data() + b0 * stride,
                                             l.lens.data() + b0);
            }
            dp->last_ms[g] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        });
    } catch (const std::exception& ex) {
        dp->last_error = std::string("cannot start the engine's host thread: ") + ex.what();
        return QASR_ERR_INVALID;
    }
    l.ticket = dp->next_ticket++;
    *ticket = l.ticket;
    return QASR_OK;
}

int qasr_dp_collect(qasr_dp* dp, int64_t ticket, int32_t* tokens, int32_t* lens) {
    if (!dp || dp->engines.empty() || ticket < 0) return QASR_ERR_INVALID;
    const size_t g = (size_t)(ticket % (int64_t)dp->engines.size());
    Lane& l = dp->lanes[g];
    if (l.ticket != ticket) {
        dp->last_error = "ticket " + std::to_string(ticket) + " is not in flight (already collected, or never issued)";
        return QASR_ERR_INVALID;
    }
    if (l.B && (!tokens || !lens)) {                            // the batch stays in flight: call again with buffers
        dp->last_error = "qasr_dp_collect: ticket " + std::to_string(ticket) + " needs token and length buffers (the batch stays in flight)";
        return QASR_ERR_INVALID;
    }
    l.worker.join();
    l.ticket = -1;
    if (l.rc != QASR_OK) {
        dp->last_error = "engine " + std::to_string(g) + " (device " + std::to_string(dp->engines[g]->impl->config().device) + "): " +
                         qasr_last_error(dp->engines[g]);
        return l.rc;
    }
    if (l.B) {
        std::memcpy(tokens, l.tokens.data(), l.tokens.size() * sizeof(int32_t));
        std::memcpy(lens, l.lens.data(), l.lens.size() * sizeof(int32_t));
    }
    return QASR_OK;
}

int qasr_dp_timings(const qasr_dp* dp, float* ms, int32_t cap) {
    if (!dp || !ms || cap < (int32_t)dp->engines.size()) return QASR_ERR_INVALID;
    for (size_t i = 0; i < dp->engines.size(); ++i) ms[i] = dp->last_ms[i];
    return QASR_OK;
}

}  // extern "C"
