// dec_qa.hip -- one decoder layer's q|k|v projection and decode attention as ONE launch (declarations: dec_chain.h).
//
// Why: the decode attention (dec_attention.hip) spends 10 of its 15 us waiting for its K / V bytes (61 MB per launch at 32 x 30 s), and those
// bytes depend on nothing the layer computes -- only the token's own q / k / v rows do.  As a separate launch the stream cannot start before
// the q|k|v projection's launch has ended (5 us + a kernel boundary).  Here every workgroup first requests the projection's operands, then the
// whole first round of its attention unit's K / V chunks into registers, runs the projection while they fly, hands the 32 x 4096 projected
// rows over inside the launch (dec_chain_dev.h: write-through stores, one arrival counter per kv head, sc1 loads) and sweeps when both are in.
//
// Grid: 256 workgroups x 512 threads, one per CU.  Workgroup g projects weight-row tile g (16 of the 4096 q|k|v columns, all batch rows) and then
// owns attention unit (batch row g / 8, kv head g % 8).  The arithmetic is decode_gemv2_kernel's and decode_attention_mfma_kernel's, chunk for
// chunk and wave for wave: same bits as the two launches (tests/test_gpu_chain.py).
//
// Request schedule (second form; the first one requested every chunk before the projection and measured the projection's rows staged only at
// 9.3 us, profiles/r04_stamps_chain.txt: a wave blocks at ISSUE while its CU's request queue drains at the CU's HBM share, so 32 chunk requests
// per wave in front of the projection put the whole stream in front of it):
//   entry              context length, norm weights, activation rows, weight tile, then ONLY the K half of each wave's first chunk (8 KB per wave:
//                      the stream starts, the queue never fills)
//   projection summed  waves 1..7: the rest of their two chunks (V of the first, K and V of the second) -- nothing of theirs is on the critical
//                      path any more; wave 0: epilogue stores, drain, signal, poll
//   hand-off over      wave 0 reads the token's own q / k / v rows, norms + ropes them ONCE for the workgroup (shared LDS image; the stand-alone
//                      kernel does it per wave to save a barrier, same values), appends k / v, then requests the rest of its chunks
// vmcnt is one in-order counter for loads and stores (gfx9): a wave that drains its stores, or waits for a poll or for the own rows, waits for
// every older load of its own -- which is why wave 0 holds nothing but its first K half (landed long before its drain) until the rows are in.
#include "dec_chain_dev.h"
#include "dec_rope.h"
#include "dec_quant_dev.h"
#include <mutex>

namespace qasr {
namespace {

using namespace chain_dev;

constexpr int QA_HD = 128, QA_HEADS = 16, QA_KVH = 8, QA_NQKV = 4096, QA_UNR = 2;

template <int HD>
__device__ __forceinline__ long qa_vfrag_index(int key, int d) {       // = vfrag_index of dec_attention.hip
    constexpr int DT = HD / 16;
    const int kb = key >> 5, r = key & 31, half = r >> 4, g = (r & 15) >> 2, j = r & 3;
    return (((long)kb * DT + (d >> 4)) * 64 + (d & 15) + 16 * g) * 8 + half * 4 + j;
}

// The projection on an MLX affine-quantised q|k|v matrix (WQ = 4 or 8 bits, group 64, bf16 scales): decode_gemvq_kernel's arithmetic (dec_quant.hip)
// for ONE 16-row weight tile and NB batch tiles -- rows normalised on their way into LDS together with the f32 sums of each 64-column group of
// the STAGED values, k-blocks interleaved over the 8 waves, per group two MFMAs from a zero accumulator (A = activations, B = the integer
// fragment) then tot += scale * acc + bias' * xsum on the vector unit, cross-wave sums in the order wave 0 + 1 + ... + 7.  Same bits as that
// kernel (tests/test_gpu_chain.py).  acc[p] is valid on wave 0: batch row 16 p + (lane >> 2), weight rows 4 (lane & 3) .. + 3 of the tile.
template <int BITS, int NB, class Hook>
__device__ __forceinline__ void qa_project_q(const uint4 (&wqr)[BITS == 4 ? 1 : 2], const float (&qsc)[BITS == 4 ? 1 : 2][BITS == 4 ? 2 : 1],
                                             const float (&qbi)[BITS == 4 ? 1 : 2][BITS == 4 ? 2 : 1], const uint4 (&xr)[NB][4],
                                             const char* s_normw, float eps, char* s_x, float* s_red, f32x4 (&acc)[1][NB], Hook after_stage) {
    constexpr int K = CH_H, XSTRIDE = 2 * K + 16, KBW = BITS == 4 ? 1 : 2, GPB = BITS == 4 ? 2 : 1, TPR = 32, GH = K / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fc = lane >> 4;
    const int srow = tid >> 5, scol = tid & 31;
    float* s_xs = reinterpret_cast<float*>(s_x + (size_t)NB * 16 * XSTRIDE);            // [NB][GH][16]
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        char* xrow = s_x + (size_t)(p * 16 + srow) * XSTRIDE + scol * 16;
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&xr[p][i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(e[j]); ss = fmaf(f, f, ss); }
        }
        ss = lane_sum<TPR>(ss);
        const float inv = rsqrtf(ss / (float)K + eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 nw = *reinterpret_cast<const uint4*>(s_normw + (scol + i * TPR) * 16);
            uint4 o = xr[p][i];
            o = make_uint4(rmsnorm_pair_bf16(o.x, nw.x, inv), rmsnorm_pair_bf16(o.y, nw.y, inv),
                           rmsnorm_pair_bf16(o.z, nw.z, inv), rmsnorm_pair_bf16(o.w, nw.w, inv));
            *reinterpret_cast<uint4*>(xrow + i * TPR * 16) = o;
            const bf16_t* oe = reinterpret_cast<const bf16_t*>(&o);
            float s8 = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s8 += bf16_to_f32(oe[j]);
            s8 = lane_sum8(s8);
            if ((scol & 7) == 0) s_xs[(p * GH + ((scol + TPR * i) >> 3)) * 16 + srow] = s8;
        }
    }
    after_stage();
    __syncthreads();
    f32x4 tot[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) tot[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KBW; ++i) {
        const int blk = wave + CWAVES * i;
#pragma unroll
        for (int h = 0; h < GPB; ++h) {
            const int g = blk * GPB + h;
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const f32x4 xs = *reinterpret_cast<const f32x4*>(s_xs + (p * GH + g) * 16 + fc * 4);
                const char* xb = s_x + (size_t)(p * 16 + fr) * XSTRIDE;
                const uint4 x0 = *reinterpret_cast<const uint4*>(xb + ((g * 2 + 0) * 32 + fc * 8) * 2);
                const uint4 x1 = *reinterpret_cast<const uint4*>(xb + ((g * 2 + 1) * 32 + fc * 8) * 2);
                f32x4 a2 = f32x4{0.f, 0.f, 0.f, 0.f};
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, x0), frag_of<BITS>(wqr[i], 2 * h + 0), a2, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, x1), frag_of<BITS>(wqr[i], 2 * h + 1), a2, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) tot[p][j] += qsc[i][h] * a2[j] + qbi[i][h] * xs[j];
            }
        }
    }
#pragma unroll
    for (int p = 0; p < NB; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) s_red[((size_t)(wave * NB + p) * 16 + fc * 4 + j) * 16 + fr] = tot[p][j];
    __syncthreads();
    if (wave == 0) {
        const int erow = lane >> 2, eq = lane & 3;
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            acc[0][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int wv = 0; wv < CWAVES; ++wv)
                acc[0][p] += *reinterpret_cast<const f32x4*>(s_red + ((size_t)(wv * NB + p) * 16 + erow) * 16 + eq * 4);
        }
    }
}

// ST: diagnostic instantiation (qasr_kernel_probe 7), stamps of the 100 MHz clock in a.dbg[wg * 32 + i]: thread 0: 0 entry, 1 rows staged,
// 2 projection summed, 3 signalled, 4 wait over, 7 output stored, 8 query prepared; thread 64 (wave 1): 9 its own K / V chunks in, 5 the barrier
// behind the preparation passed (= every wave's requests have landed: __syncthreads drains them), 6 sweep done
// EARLY: which waves request the K half of their first chunk before the projection: 0 none | 1 all | 2 waves 4..7 (A/B, knob qa_early).
// GATE: 1 = waves 1..7 hold their remaining requests until wave 0 has stored, drained and signalled the projection (so that the hand-off's
//       write-through stores are not queued behind the stream) | 0 = they request as soon as the sums are in (knob qa_gate).
// GRAN: 1 = the hand-off as data-tagged granules (DecQaArgs::gran): wave 0 stores {two values, tag} words straight from the sums -- no drain, no
//       counter -- and the consumer's wave 0 re-reads ITS 256 granules until every tag is this launch's: the poll and the rows' fetch are one
//       round trip | 0 = write-through rows + one arrival counter per kv head + sc1 row loads (the first form; knob qa_gran).
// WQ: 0 = bf16 weights (fragment-major image) | 4, 8 = MLX affine-quantised q|k|v matrix of that many bits (qa_project_q; granule hand-off only)
// SPLIT (up to 16 batch rows, where the 8 x B attention units leave most CUs without one and a unit's K / V rows -- 240 KB at 30 s -- come through ONE
//       CU's request path): a unit is spread over SPLIT = 2, 4 or 8 workgroups.  Workgroup (unit, s) sweeps the chunks that waves w = s (mod SPLIT) swept
//       before -- wave w keeps its chunk set and its running (max, sum, output), so the eight per-wave partials are the SAME numbers -- and the workgroups
//       s != 0 hand theirs to workgroup (unit, 0) as tagged granules (DecQaArgs::part); there the waves without chunks poll one partner each into the
//       LDS slots the merge reads: same merge, same order, same bits.
template <int NB, bool ST, int EARLY, int GATE, int GRAN, int WQ = 0, int SPLIT = 1>
__global__ __launch_bounds__(CT, 2) void decode_qa_kernel(DecQaArgs a) {
    static_assert(WQ == 0 || (GRAN == 1 && !ST), "quantised projection: granule hand-off, no stamps");
    static_assert(SPLIT == 1 || ((SPLIT == 2 || SPLIT == 4 || SPLIT == 8) && GRAN == 1 && NB == 1 && !ST), "context split: granules, one batch tile");
    constexpr int HD = QA_HD, REP = 2, KS = HD / 32, DT = HD / 16, HALF = HD / 2, WAVES = CWAVES, UNR = QA_UNR;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4, fc = g;
    const int srow = tid >> 5, scol = tid & 31;
    const int wg = blockIdx.x;
    const int B = a.B;
    const int unit = wg / SPLIT, sp = wg % SPLIT;
    const int b = unit >> 3, kvh = unit & 7;
    const bool has_att = b < B;
    const bool has_work = SPLIT == 1 || (wave % SPLIT) == sp;          // this wave sweeps chunks in this workgroup
    const int bq = has_att ? b : 0;
    char* s_x = dsm + L_X;
    float* s_red = reinterpret_cast<float*>(dsm + L_RED);
    int* s_flag = reinterpret_cast<int*>(dsm + L_FLAG);
    const KVLayout cache = a.cache;
    unsigned long long* st = ST ? a.dbg + (long)wg * 32 : nullptr;
#define QA_STAMP(i, t) do { if constexpr (ST) { if (tid == (t)) st[i] = wall_clock64(); } } while (0)
    QA_STAMP(0, 0);

    // ---- requests, oldest first: context length, norm weights, activation rows, weight tile, then the K / V chunks ------------------
    int pos = a.ctx_len[bq];
    unsigned tag = 0;
    if constexpr (GRAN) tag = (a.ctr[CHAIN_SEQ_WORD] << 8) | (a.epoch + 1u);      // written by an earlier launch (finalize / reset): plain load
    const uint4 nw = reinterpret_cast<const uint4*>(a.ln1)[tid & 127];
    uint4 xr[NB][4];
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        const int row = p * 16 + srow;
        const bf16_t* xp = a.x + (long)(row < B ? row : 0) * CH_H + scol * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[p][i] = *reinterpret_cast<const uint4*>(xp + i * 32 * 8);
    }
    uint4 wC[1][4];
    constexpr int QKBW = WQ == 8 ? 2 : 1, QGPB = WQ == 8 ? 1 : 2, QBLK = WQ == 8 ? 64 : 128;      // k-blocks per wave, groups per block, columns per block
    uint4 wqr[QKBW];
    float qsc[QKBW][QGPB], qbi[QKBW][QGPB];
    auto request_w = [&]() {
        if constexpr (WQ == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wC[0][i] = *reinterpret_cast<const uint4*>(a.wqkv_p + ((long)wg * (CH_H / 32) + wave + CWAVES * i) * 512 + lane * 8);
        } else {
            // tile wg of the packed image: [tile][K / BLK blocks][64 lanes][16 B], scales / biases [tile][16 rows][K / 64 groups][2] (bf16)
            const uint32_t* qp = a.wq_qp + ((long)wg * (CH_H / QBLK) * 64 + lane) * 4;
#pragma unroll
            for (int i = 0; i < QKBW; ++i) wqr[i] = *reinterpret_cast<const uint4*>(qp + (long)(wave + CWAVES * i) * 256);
#pragma unroll
            for (int i = 0; i < QKBW; ++i) {
                sb_load<false, QGPB>(a.wq_sb, (((long)wg * 16 + fr) * (CH_H / 64) + (wave + CWAVES * i) * QGPB) * 2, qsc[i], qbi[i]);
#pragma unroll
                for (int h = 0; h < QGPB; ++h) qbi[i][h] = eff_bias<WQ == 8 ? 8 : 4>(qsc[i][h], qbi[i][h]);
            }
        }
    };
    if constexpr (EARLY != 4) {
        // every wave's row requests enter the CU's request queue before any wave's weight requests (a bare barrier: nothing is waited for) -- the rows of
        // a late wave otherwise queue behind the weight misses of the early ones: -0.7 % decode at 32 rows, nothing at 16 (knob qa_xbar)
        if (a.xbar) __builtin_amdgcn_s_barrier();
        request_w();
    }

    const bf16_t* kb = cache.k + cache.off(bq, kvh, 0) + g * 8;
    const bf16_t* vfb = cache.vf + cache.off(bq, kvh, 0) + lane * 8;
    const bf16_t* kdummy = cache.k + cache.off(bq, kvh, 0);
    const int max_chunk = cache.max_ctx / 32 - 1;
    pos = has_att ? __builtin_amdgcn_readfirstlane(pos) : 1;         // wave-uniform by construction: scalar loop bounds below
    const int nchunks = (pos + 31) >> 5;
    uint4 kreg[UNR][2 * KS], vreg[UNR][DT];
    int ch_early[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        int ch = wave + u * WAVES;
        ch = ch < nchunks - 1 ? ch : nchunks - 1;                     // chunks past the context re-read the last one (cache hits), skipped in the sweep
        ch_early[u] = ch < max_chunk ? ch : max_chunk;
    }
    const bool live = has_att && has_work;                            // waves without chunks request one cached line instead
    // unconditional request code (addresses selected, not branches), so that hipcc's counted waits for the projection's operands stay exact
    auto request_k = [&](int u) {
        const bf16_t* kr = kb + ((long)ch_early[u] * 32 + fr) * HD;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                kreg[u][h * KS + ks] = *reinterpret_cast<const uint4*>(live ? kr + (long)h * 16 * HD + ks * 32 : kdummy);
    };
    auto request_v = [&](int u) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            vreg[u][dt] = *reinterpret_cast<const uint4*>(live ? vfb + ((long)ch_early[u] * DT + dt) * 512 : kdummy);
    };
    // (wave 0 may take part: its 8 requests have landed long before its drain)
    // EARLY 3: the same request from every wave once its activation rows have left for LDS (chain_mma's after-stage hook); 4: before the
    // weight tile instead of behind it
    const bool early_wave = EARLY == 1 || EARLY == 3 || EARLY == 4 || (EARLY == 2 && wave >= 4);
    if constexpr (EARLY == 1 || EARLY == 2 || EARLY == 4) { if (early_wave) request_k(0); }
    if constexpr (EARLY == 4) request_w();
    if (tid < 128) reinterpret_cast<uint4*>(dsm + L_NORM)[tid] = nw;
    __syncthreads();

    // ---- q|k|v projection of tile wg for every batch row ------------------------------------------------------------------------------
#pragma unroll
    for (int p = 0; p < NB; ++p)
        if (p * 16 + srow >= B) {
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[p][i] = make_uint4(0, 0, 0, 0);
        }
    {
        f32x4 acc[1][NB];
        auto hook = [&]() { if constexpr (EARLY == 3) request_k(0); };
        if constexpr (WQ == 0) chain_mma<1, NB, 4, true, ST>(wC, xr, dsm + L_NORM, a.eps, s_x, s_red, acc, st + 1, hook);
        else qa_project_q<WQ, NB>(wqr, qsc, qbi, xr, dsm + L_NORM, a.eps, s_x, s_red, acc, hook);
        // a lane of wave 0 now holds four consecutive values of one batch row of the tile: row orow, values 4 ocol .. + 3
        const int orow = WQ == 0 ? fr : lane >> 2, ocol = WQ == 0 ? fc : lane & 3;
        auto rest = [&]() { if (!early_wave) request_k(0); request_v(0); request_k(1); request_v(1); };
        if constexpr (!GATE) { if (wave != 0) rest(); }
        if (wave == 0) {
            if constexpr (GRAN) {
                // two granules per lane and row tile: values (4 fc, 4 fc + 1) and (4 fc + 2, 4 fc + 3) of tile wg, each ONE aligned 8-byte store
                if (!(a.fault && wg == 5)) {
#pragma unroll
                    for (int p = 0; p < NB; ++p) {
                        const int row = p * 16 + orow;
                        if (row < B) {
                            const uint2 v = pack_bf16x4(make_float4(acc[0][p][0], acc[0][p][1], acc[0][p][2], acc[0][p][3]));
                            unsigned long long* gp = a.gran + (long)row * QA_GRAN_ROW + wg * 8 + ocol * 2;
                            __hip_atomic_store(gp, ((unsigned long long)tag << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(gp + 1, ((unsigned long long)tag << 32) | v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < NB; ++p) {
                    const int row = p * 16 + fr;
                    if (row < B)
                        st8_sc1(a.qkv + (long)row * QA_NQKV + wg * 16 + fc * 4,
                                pack_bf16x4(make_float4(acc[0][p][0], acc[0][p][1], acc[0][p][2], acc[0][p][3])));
                }
                // tile -> kv head it feeds: q columns (tiles 0..127, 16 per kv head), then k (8 per head), then v
                const int grp = wg < 128 ? wg >> 4 : (wg < 192 ? (wg - 128) >> 3 : (wg - 192) >> 3);
                if (!(a.fault && wg == 5)) seam_signal(a.ctr, 3, grp);
            }
            QA_STAMP(3, 0);
        }
        if constexpr (GATE) {
            __syncthreads();
            if (wave != 0) rest();
        }
    }
    if (!has_att) return;
    if constexpr (!GRAN) {
        if (!seam_wait_n(a.ctr + (3 * CHAIN_SHARDS + kvh) * CHAIN_SHARD_WORDS, 1, (a.epoch + 1) * 32, a.err, s_flag)) return;
        QA_STAMP(4, 0);
    }

    // ---- attention unit (b, kvh): decode_attention_mfma_kernel's body with the first round of K / V already requested -----------------
    auto issue = [&](int chunk0, int limit) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            int ch = chunk0 + u * WAVES;
            if (ch >= limit) continue;
            ch = ch < max_chunk ? ch : max_chunk;
            const bf16_t* kr = kb + ((long)ch * 32 + fr) * HD;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    kreg[u][h * KS + ks] = *reinterpret_cast<const uint4*>(kr + (long)h * 16 * HD + ks * 32);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                vreg[u][dt] = *reinterpret_cast<const uint4*>(vfb + ((long)ch * DT + dt) * 512);
        }
    };
    bf16_t* s_q = reinterpret_cast<bf16_t*>(s_x);                                       // [REP][HD]  (one image for the workgroup)
    float* s_o = reinterpret_cast<float*>(s_x + WAVES * REP * HD * 2);                  // [WAVES][REP][HD]
    float* s_m = s_o + WAVES * REP * HD;                                                // [WAVES][REP]
    float* s_l = s_m + WAVES * REP;
    float* s_new = s_l + WAVES * REP;                                                   // [REP]
    float* s_vn = s_new + REP;                                                          // [HD]
    if (wave == 0) {
        // the token's own rows: handed-off bytes -> sc1 loads, one dword (two elements) per lane, then a lane permute puts element `lane`
        // and element `lane + 64` on every lane like the two-byte loads of the stand-alone kernel
        unsigned wq[REP], wk, wv;
        if constexpr (GRAN) {
            // the sweep: this unit's 4 x 64 granules (lane <-> values 2 lane, 2 lane + 1 of a 128-value row) until every tag is this launch's;
            // the error word rides along, so that a step which has lost an arrival fails every later wait at its first pass
            const unsigned long long* gr = a.gran + (long)b * QA_GRAN_ROW + lane;
            const unsigned long long t0 = wall_clock64();
            bool ok;
            for (;;) {
                unsigned long long x[REP + 2];
#pragma unroll
                for (int r = 0; r < REP; ++r) x[r] = __hip_atomic_load(gr + (kvh * REP + r) * (HD / 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x[REP] = __hip_atomic_load(gr + (QA_HEADS + kvh) * (HD / 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x[REP + 1] = __hip_atomic_load(gr + (QA_HEADS + QA_KVH + kvh) * (HD / 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned e = __hip_atomic_load(reinterpret_cast<const unsigned*>(a.err), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool mine = true;
#pragma unroll
                for (int r = 0; r < REP + 2; ++r) mine &= (unsigned)(x[r] >> 32) == tag;
#pragma unroll
                for (int r = 0; r < REP; ++r) wq[r] = (unsigned)x[r];
                wk = (unsigned)x[REP];
                wv = (unsigned)x[REP + 1];
                const bool dead = (e & CHAIN_ERR_TIMEOUT) != 0;
                ok = !dead && __builtin_amdgcn_ballot_w64(!mine) == 0;
                if (ok || dead || wall_clock64() - t0 > CH_SPIN_TICKS) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) {
                *s_flag = ok ? 1 : 0;
                if (!ok) atomicOr(a.err, CHAIN_ERR_TIMEOUT);
            }
            QA_STAMP(4, 0);
        } else {
            const bf16_t* row = a.qkv + (long)b * QA_NQKV;
#pragma unroll
            for (int r = 0; r < REP; ++r) wq[r] = ld4_sc1(row + (long)(kvh * REP + r) * HD + 2 * lane);
            wk = ld4_sc1(row + (long)(QA_HEADS + kvh) * HD + 2 * lane);
            wv = ld4_sc1(row + (long)(QA_HEADS + QA_KVH + kvh) * HD + 2 * lane);
        }
        const float rc = a.rope_cos[(long)b * HALF + lane], rs = a.rope_sin[(long)b * HALF + lane];
        const bf16_t w1r = a.qn_w[lane], w2r = a.qn_w[lane + HALF], kw1r = a.kn_w[lane], kw2r = a.kn_w[lane + HALF];
        // wave 0's remaining chunk requests go out BEHIND its own-row requests (which therefore come back first: loads return in order) and
        // before anything uses them -- the query preparation below then runs while these chunks fly, instead of in front of their request
        // (unconditional, on the clamped chunk indices: a request under a branch would make hipcc wait for ALL of them at the rows' first use)
        __builtin_amdgcn_sched_barrier(0);
        {
            constexpr bool W0_EARLY = EARLY == 1 || EARLY == 3 || EARLY == 4;      // wave 0's first K half came in with everybody's
            if constexpr (!W0_EARLY) {
                const bf16_t* kr = kb + ((long)ch_early[0] * 32 + fr) * HD;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) kreg[0][h * KS + ks] = *reinterpret_cast<const uint4*>(live ? kr + (long)h * 16 * HD + ks * 32 : kdummy);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vreg[0][dt] = *reinterpret_cast<const uint4*>(live ? vfb + ((long)ch_early[0] * DT + dt) * 512 : kdummy);
            const bf16_t* kr = kb + ((long)ch_early[1] * 32 + fr) * HD;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kreg[1][h * KS + ks] = *reinterpret_cast<const uint4*>(live ? kr + (long)h * 16 * HD + ks * 32 : kdummy);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vreg[1][dt] = *reinterpret_cast<const uint4*>(live ? vfb + ((long)ch_early[1] * DT + dt) * 512 : kdummy);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float w1 = bf16_to_f32(w1r), w2 = bf16_to_f32(w2r), kw1 = bf16_to_f32(kw1r), kw2 = bf16_to_f32(kw2r);
        auto pick = [&](unsigned w, int src_lane) {
            const unsigned v = (unsigned)__shfl((int)w, src_lane, 64);
            return (bf16_t)((lane & 1) ? v >> 16 : v & 0xffffu);
        };
        float x1[REP], x2[REP];
#pragma unroll
        for (int r = 0; r < REP; ++r) { x1[r] = bf16_to_f32(pick(wq[r], lane >> 1)); x2[r] = bf16_to_f32(pick(wq[r], 32 + (lane >> 1))); }
        const float kx1 = bf16_to_f32(pick(wk, lane >> 1)), kx2 = bf16_to_f32(pick(wk, 32 + (lane >> 1)));
        const bf16_t vown[2] = {pick(wv, lane >> 1), pick(wv, 32 + (lane >> 1))};
        float qa[REP][2];
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const float inv = rsqrtf(lane_sum<64>(x1[r] * x1[r] + x2[r] * x2[r]) / (float)HD + a.eps);
            norm_rope_pair(x1[r], x2[r], w1, w2, inv, rc, rs, qa[r][0], qa[r][1]);
            s_q[r * HD + lane] = f32_to_bf16(qa[r][0]);
            s_q[r * HD + lane + HALF] = f32_to_bf16(qa[r][1]);
        }
        if (SPLIT == 1 || sp == 0) {       // the token's own k / v: appended once per unit, its score and value enter the merge there
            const float inv = rsqrtf(lane_sum<64>(kx1 * kx1 + kx2 * kx2) / (float)HD + a.eps);
            float k1, k2;
            norm_rope_pair(kx1, kx2, kw1, kw2, inv, rc, rs, k1, k2);
            bf16_t* dk = cache.k + cache.off(b, kvh, pos);
            dk[lane] = f32_to_bf16(k1);
            dk[lane + HALF] = f32_to_bf16(k2);
#pragma unroll
            for (int r = 0; r < REP; ++r) {
                const float d = lane_sum<64>(qa[r][0] * k1 + qa[r][1] * k2);
                if (lane == 0) s_new[r] = d * a.scale;
            }
            bf16_t* dvf = cache.vf + cache.off(b, kvh, 0);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = lane + 64 * ii;
                s_vn[i] = bf16_to_f32(vown[ii]);
                dvf[qa_vfrag_index<HD>(pos, i)] = vown[ii];
            }
        }
        QA_STAMP(8, 0);                                                // wave 0: query prepared, k / v appended
    }
    if constexpr (ST) { if (wave == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st[9] = wall_clock64(); } }     // wave 1: its two chunks are in (all lanes write the same word)
    __syncthreads();
    if constexpr (GRAN) { if (*s_flag == 0) return; }                 // the sweep gave up (uniform: read behind the barrier)
    if constexpr (ST) { if (wave == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    QA_STAMP(5, 64);
    mfma_bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (fr < REP) u = *reinterpret_cast<const uint4*>(&s_q[fr * HD + ks * 32 + g * 8]);
        qf[ks] = __builtin_bit_cast(mfma_bf16x8, u);
    }
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.0f;
    const float scale = a.scale;
    for (int c0 = has_work ? wave : nchunks; c0 < nchunks; c0 += WAVES * UNR) {
        if (c0 != wave) issue(c0, nchunks);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int chunk = c0 + u * WAVES;
            if (chunk < nchunks) {                                       // wave-uniform
                f32x4 sc[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, kreg[u][h * KS + ks]), qf[ks], acc, 0, 0, 0);
                    sc[h] = acc;
                }
                const int key0 = chunk * 32 + g * 4;
                float mx = -INFINITY;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = key0 + h * 16 + j < pos ? sc[h][j] * scale : -INFINITY;
                        sc[h][j] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = rows4_max(mx);
                const float m_new = fmaxf(m_run, mx);
                const float alpha = __expf(m_run - m_new);
                float rsum = 0.0f;
                unsigned pk[4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const unsigned pw = pack_bf16x2(__expf(sc[h][j] - m_new), __expf(sc[h][j + 1] - m_new));
                        rsum += bf16_lo(pw) + bf16_hi(pw);
                        pk[h * 2 + j / 2] = pw;
                    }
                l_run = l_run * alpha + rsum;
                m_run = m_new;
                const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(alpha), 0));
                const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(alpha), 1));
                uint4 vv[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) vv[dt] = vreg[u][dt];
                if (chunk * 32 + 32 > pos) {
                    unsigned msk[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int k_lo = key0 + (w >> 1) * 16 + (w & 1) * 2;
                        msk[w] = (k_lo < pos ? 0x0000ffffu : 0u) | (k_lo + 1 < pos ? 0xffff0000u : 0u);
                    }
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) { vv[dt].x &= msk[0]; vv[dt].y &= msk[1]; vv[dt].z &= msk[2]; vv[dt].w &= msk[3]; }
                }
                const mfma_bf16x8 pa = __builtin_bit_cast(mfma_bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    o[dt][0] *= a0;
                    o[dt][1] *= a1;
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, __builtin_bit_cast(mfma_bf16x8, vv[dt]), o[dt], 0, 0, 0);
                }
            }
        }
    }
    QA_STAMP(6, 64);
    l_run = rows4_sum(l_run);
    if constexpr (SPLIT > 1) {
        // partial of wave w of unit u: QA_PART_STRIDE granules -- [r * HD + d] the outputs, [REP * HD + r] the running maxima, [REP * HD + REP + r] the sums
        unsigned long long* pw = a.part + ((long)unit * WAVES + wave) * QA_PART_STRIDE;
        const unsigned long long tg = (unsigned long long)tag << 32;
        if (sp != 0) {
            if (has_work) {
                if (lane < REP) {
                    __hip_atomic_store(pw + REP * HD + lane, tg | __float_as_uint(m_run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(pw + REP * HD + REP + lane, tg | __float_as_uint(l_run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (g == 0) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        __hip_atomic_store(pw + 0 * HD + dt * 16 + fr, tg | __float_as_uint(o[dt][0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(pw + 1 * HD + dt * 16 + fr, tg | __float_as_uint(o[dt][1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            return;                         // (every wave of this workgroup: sp is uniform)
        }
        if (!has_work) {
            // workgroup (unit, 0), a wave without chunks: fetch the partial that wave `wave` of workgroup (unit, wave % SPLIT) sends, into the slots the
            // merge reads.  Bounded like every in-launch wait; on a give-up the slot gets an empty partial and the step ends in QASR_ERR_HIP.
            constexpr int NG = REP * HD + 2 * REP, NL = (NG + 63) / 64;
            const unsigned long long t0 = wall_clock64();
            unsigned long long x[NL];
            bool ok;
            for (;;) {
                bool mine = true;
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const int idx = lane + 64 * i;
                    x[i] = idx < NG ? __hip_atomic_load(pw + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tg;
                    mine &= (unsigned)(x[i] >> 32) == tag;
                }
                const unsigned e = __hip_atomic_load(reinterpret_cast<const unsigned*>(a.err), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool dead = (e & CHAIN_ERR_TIMEOUT) != 0;
                ok = !dead && __builtin_amdgcn_ballot_w64(!mine) == 0;
                if (ok || dead || wall_clock64() - t0 > CH_SPIN_TICKS) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok && lane == 0) atomicOr(a.err, CHAIN_ERR_TIMEOUT);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int idx = lane + 64 * i;
                const float v = __uint_as_float((unsigned)x[i]);
                if (idx < REP * HD) s_o[wave * REP * HD + idx] = ok ? v : 0.0f;
                else if (idx < REP * HD + REP) s_m[wave * REP + idx - REP * HD] = ok ? v : -INFINITY;
                else if (idx < NG) s_l[wave * REP + idx - REP * HD - REP] = ok ? v : 0.0f;
            }
        }
    }
    if (has_work) {
        if (lane < REP) { s_m[wave * REP + lane] = m_run; s_l[wave * REP + lane] = l_run; }
        if (g == 0) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                s_o[(wave * REP + 0) * HD + dt * 16 + fr] = o[dt][0];
                s_o[(wave * REP + 1) * HD + dt * 16 + fr] = o[dt][1];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < REP * HD; i += WAVES * 64) {
        const int r = i / HD, d = i - r * HD;
        float mm = s_new[r];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) mm = fmaxf(mm, s_m[w * REP + r]);
        const float pn = __expf(s_new[r] - mm);
        float num = pn * s_vn[d], den = pn;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const float mw = s_m[w * REP + r];
            if (mw != -INFINITY) {
                const float al = __expf(mw - mm);
                num += s_o[(w * REP + r) * HD + d] * al;
                den += s_l[w * REP + r] * al;
            }
        }
        a.out[(long)b * QA_HEADS * HD + (long)(kvh * REP + r) * HD + d] = f32_to_bf16(num / den);
    }
    QA_STAMP(7, 0);
#undef QA_STAMP
}

}  // namespace

bool decode_qa_supported(int H, int heads, int kv_heads, int hd, int B, int max_ctx) {
    hipDeviceProp_t p;
    int dev = 0;
    static std::mutex mu;
    static int cus[64] = {0};
    QASR_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(mu);
        if (dev >= 0 && dev < 64 && !cus[dev]) { QASR_HIP(hipGetDeviceProperties(&p, dev)); cus[dev] = p.multiProcessorCount; }
    }
    return H == CH_H && heads == QA_HEADS && kv_heads == QA_KVH && hd == QA_HD && B >= 1 && B <= 32 && max_ctx % 32 == 0 && dev < 64 &&
           cus[dev] >= 256;
}

void decode_qa_launch(const DecQaArgs& a0, hipStream_t s) {
    DecQaArgs a = a0;
    a.fault = tuning().chain_fault;
    a.xbar = tuning().qa_xbar == 2 ? (a.B > 16 ? 1 : 0) : tuning().qa_xbar;
    if (a.B < 1 || a.B > 32) throw std::invalid_argument("decode qa: 1..32 batch rows");
    if (!a.cache.vf || a.cache.max_ctx % 32) throw std::invalid_argument("decode qa: fragment-major V image / capacity");
    auto go = [&](auto k) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(k), L_TOTAL);
        hipLaunchKernelGGL(k, dim3(256), dim3(CT), L_TOTAL, s, a);
    };
    // 5 = by batch: 16 rows and more -> 3 (behind the staging), else 4 (in front of the weight tile): profiles/r04_ab_fused_layer.txt
    const bool gran = tuning().qa_gran != 0 && a.gran != nullptr;
    const int early = tuning().qa_early == 5 ? (a.B >= 16 ? 3 : 4) : tuning().qa_early;
    const int gate = tuning().qa_gate == 2 ? (gran ? 1 : 0) : tuning().qa_gate;
    // up to 8 rows: a unit's context over 8 / 4 workgroups (default schedule only: granules, gate, first K half by batch); measured -4.4 % decode at 1 row,
    // -3.5 % at 4, -1.4 % at 8, +0.2 % at 16 with two workgroups per unit (profiles/r04_ab_fused_layer.txt); qa_split 2 forces the split up to 16 rows
    int split = 1;
    if (tuning().qa_split && gran && a.part && a.B <= (tuning().qa_split == 2 ? 16 : 8) && gate == 1 && (early == 3 || early == 4))
        split = a.B <= 4 ? 8 : a.B <= 8 ? 4 : 2;
    if (split > 1) {
        if (a.wq_bits && ((a.wq_bits != 4 && a.wq_bits != 8) || !a.wq_qp || !a.wq_sb)) throw std::invalid_argument("decode qa: 4 / 8-bit image expected");
#define QA_GOS(W_, S_) do { if (early == 3) go(decode_qa_kernel<1, false, 3, 1, 1, W_, S_>); else go(decode_qa_kernel<1, false, 4, 1, 1, W_, S_>); } while (0)
#define QA_GOSW(S_) do { if (a.wq_bits == 4) QA_GOS(4, S_); else if (a.wq_bits == 8) QA_GOS(8, S_); else QA_GOS(0, S_); } while (0)
        if (split == 8) QA_GOSW(8); else if (split == 4) QA_GOSW(4); else QA_GOSW(2);
#undef QA_GOSW
#undef QA_GOS
        return;
    }
    if (a.wq_bits) {
        // quantised projection: the default schedule only (granules, gate, first K half behind the staging from 16 rows up, else in front of the weights)
        if (!gran || (a.wq_bits != 4 && a.wq_bits != 8) || !a.wq_qp || !a.wq_sb) throw std::invalid_argument("decode qa: quantised projection needs granules and a 4 / 8-bit image");
        const bool e3 = early == 3 || (early != 4 && a.B >= 16);
#define QA_GOQ(W_)                                                                                                              \
        do {                                                                                                                      \
            if (e3) { if (a.B <= 16) go(decode_qa_kernel<1, false, 3, 1, 1, W_>); else go(decode_qa_kernel<2, false, 3, 1, 1, W_>); }  \
            else { if (a.B <= 16) go(decode_qa_kernel<1, false, 4, 1, 1, W_>); else go(decode_qa_kernel<2, false, 4, 1, 1, W_>); }     \
        } while (0)
        if (a.wq_bits == 4) QA_GOQ(4); else QA_GOQ(8);
#undef QA_GOQ
        return;
    }
    if (!gran) a.gran = nullptr;
#define QA_GO2(E_, G_, R_)                                                                                                      \
    do {                                                                                                                          \
        if (a.dbg) { if (a.B <= 16) go(decode_qa_kernel<1, true, E_, G_, R_>); else go(decode_qa_kernel<2, true, E_, G_, R_>); }  \
        else { if (a.B <= 16) go(decode_qa_kernel<1, false, E_, G_, R_>); else go(decode_qa_kernel<2, false, E_, G_, R_>); }      \
    } while (0)
#define QA_GO(E_, G_) do { if (gran) QA_GO2(E_, G_, 1); else QA_GO2(E_, G_, 0); } while (0)
    if (early == 0) { if (gate) QA_GO(0, 1); else QA_GO(0, 0); }
    else if (early == 3) { if (gate) QA_GO(3, 1); else QA_GO(3, 0); }
    else if (early == 4) { if (gate) QA_GO(4, 1); else QA_GO(4, 0); }
    else if (early == 2) { if (gate) QA_GO(2, 1); else QA_GO(2, 0); }
    else { if (gate) QA_GO(1, 1); else QA_GO(1, 0); }
#undef QA_GO
#undef QA_GO2
}

}  // namespace qasr
