// dec_attention.hip -- decode-step attention on the matrix cores (the dominant kernel of the headline: bench.py stamps this file for the PMC traffic figure) (declarations: dec_kernels.h).
#include "dec_kernels.h"
#include "dec_rope.h"
#include <cstdlib>
#include <cstdio>

namespace qasr {

// ------------------------------------------------------------------------------------------------
// Decode attention on the matrix cores.  One workgroup per (kv head, batch row); a wave owns 32-key
// chunks of the context (chunk = wave, wave + WAVES, ...), all of whose loads are issued up front:
//   S^T = K Q^T   16x16x32 MFMA, A = 16 cached key rows straight from HBM (natural [key][hd] layout),
//                 B = the two query heads of this kv head in columns 0/1 (other columns zero);
//                 the accumulator puts keys (lane>>4)*4+j of query (lane&15) on a lane, which IS the
//                 A-operand layout of the next MFMA, so P never leaves registers;
//   O  += P V     16x16x32 MFMA over the chunk's 32 keys, B = V in the fragment-major cache image
//                 (KVLayout::vf), one 16-byte load per lane and d tile.
// Every wave norms + ropes the two query rows itself (no workgroup barrier before the sweep); the
// token's own key / value are computed by the last two waves, appended to the caches and folded in
// at the cross-wave merge.  Softmax statistics stay in f32; P is rounded to bf16 like the prompt pass.
// ------------------------------------------------------------------------------------------------
template <int HD>
__device__ __forceinline__ long vfrag_index(int key, int d) {
    // fragment-major V: [key/32][d/16][lane = (d%16) + 16*g][e = half*4 + j],  key%32 = half*16 + g*4 + j
    constexpr int DT = HD / 16;
    const int kb = key >> 5, r = key & 31, half = r >> 4, g = (r & 15) >> 2, j = r & 3;
    return (((long)kb * DT + (d >> 4)) * 64 + (d & 15) + 16 * g) * 8 + half * 4 + j;
}

// EARLYQ: the loads of the token's own q / k / v rows, the norm weights and the rope row are requested BEFORE the K / V stream.  Vector
// loads return in order, so in the other order the query math waits behind the whole K / V stream of its wave: in-kernel stamps at 1 clip
// put "query ready" 5.2 us after the first wave's start (ctx_len round trip -> K / V from HBM -> q rows), with the sweep itself 0.85 us.
// Same arithmetic either way (bit-identical results).
template <int HD, int WAVES, int UNR, int SPEC, bool EARLYQ>
__global__ __launch_bounds__(WAVES * 64) void decode_attention_mfma_kernel(
    const bf16_t* __restrict__ qkv, const int* __restrict__ ctx_len, int heads, int kv_heads,
    const bf16_t* __restrict__ qn_w, const bf16_t* __restrict__ kn_w, float eps, const float* __restrict__ rope_cos,
    const float* __restrict__ rope_sin, KVLayout cache, bf16_t* __restrict__ out, float scale,
    unsigned long long* __restrict__ dbg) {
    constexpr int REP = 2, KS = HD / 32, DT = HD / 16, HALF = HD / 2;
#if QASR_DIAG_STAMPS
#define QASR_STAMP(i) do { if (dbg && (threadIdx.x & 63) == 0) dbg[((blockIdx.y * gridDim.x + blockIdx.x) * WAVES + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define QASR_STAMP(i) do { } while (0)
#endif
    QASR_STAMP(0);
    __shared__ __attribute__((aligned(16))) bf16_t s_q[WAVES][REP][HD];      // wave-private query image
    __shared__ float s_m[WAVES][REP], s_l[WAVES][REP];
    __shared__ float s_o[WAVES][REP][HD];
    __shared__ float s_new[REP], s_vn[HD];
    const int kvh = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const bf16_t* kb = cache.k + cache.off(b, kvh, 0) + g * 8;
    const bf16_t* vfb = cache.vf + cache.off(b, kvh, 0) + lane * 8;
    const int max_chunk = cache.max_ctx / 32 - 1;
    uint4 kreg[UNR][2 * KS], vreg[UNR][DT];
    auto issue = [&](int chunk0, int limit, int u_lo = 0, int u_hi = UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (u < u_lo || u >= u_hi) continue;
            int ch = chunk0 + u * WAVES;
            if (ch >= limit) continue;                                   // wave-uniform: no bytes for chunks past the context
            ch = ch < max_chunk ? ch : max_chunk;                        // clamped to the allocation, masked later
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
    const int nh = heads + 2 * kv_heads;
    const bf16_t* row = qkv + (long)b * nh * HD;
    const bool act = lane < HALF;
    // raw loads only (no conversion, no select: either would need the data and put a wait in front of the K / V requests)
    const int lc = act ? lane : 0;           // inactive lanes (head_dim < 128) read element 0 and are zeroed after the requests
    float rc = 0.f, rs = 0.f;
    bf16_t wr1 = 0, wr2 = 0, xr1[REP], xr2[REP], kxr1 = 0, kxr2 = 0, kwr1 = 0, kwr2 = 0;
    bf16_t vown[(HD + 63) / 64];
    auto load_own = [&]() {                  // this token's q rows (+ k or v row on the two append waves), norm weights, rope row
        rc = rope_cos[(long)b * HALF + lc];
        rs = rope_sin[(long)b * HALF + lc];
        wr1 = qn_w[lc];
        wr2 = qn_w[lc + HALF];
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const bf16_t* src = row + (long)(kvh * REP + r) * HD;
            xr1[r] = src[lc];
            xr2[r] = src[lc + HALF];
        }
        // the k and v rows on EVERY wave (six 2-byte requests; only the two append waves use them): a wave-dependent branch here makes
        // hipcc end it with a full wait for the values it defines, i.e. a round trip in front of the K / V requests of those two waves
        {
            const bf16_t* src = row + (long)(heads + kvh) * HD;
            kxr1 = src[lc];
            kxr2 = src[lc + HALF];
            kwr1 = kn_w[lc];
            kwr2 = kn_w[lc + HALF];
        }
        {
            const bf16_t* src = row + (long)(heads + kv_heads + kvh) * HD;
#pragma unroll
            for (int i = 0; i < (HD + 63) / 64; ++i) vown[i] = src[lane + 64 * i < HD ? lane + 64 * i : 0];
        }
    };
    if (EARLYQ) {
        load_own();
        __builtin_amdgcn_sched_barrier(0);   // keep these requests ahead of the K / V stream (in-order return)
    }
    int pos;
    if (SPEC == 1) {
        // the wave's FIRST chunk before the position is known (rows past it are masked): with WAVES x 32 keys in the first round every
        // context of 256+ keys makes all of these requests useful; the later chunks wait for ctx_len, so no byte is fetched for chunks past
        // the context (requesting all of them blind fetched 1.26 x the algorithmic bytes at 32 x 30 s: profiles/r03_v2_pmc_traffic.json)
        issue(wave, 0x7fffffff, 0, 1);
        pos = ctx_len[b];
        issue(wave, (pos + 31) >> 5, 1, UNR);
    } else if (SPEC == 2) {
        issue(wave, 0x7fffffff);             // every chunk of the first round blind: latency-bound launches (few batch rows) only
        pos = ctx_len[b];
    } else {
        pos = ctx_len[b];
        issue(wave, (pos + 31) >> 5);
    }
    const int nchunks = (pos + 31) >> 5;
    if (EARLYQ) __builtin_amdgcn_sched_barrier(0);
    else load_own();
    if (!act) { rc = 0.f; rs = 0.f; }
    const float w1 = act ? bf16_to_f32(wr1) : 0.0f, w2 = act ? bf16_to_f32(wr2) : 0.0f;
    float x1[REP], x2[REP];
#pragma unroll
    for (int r = 0; r < REP; ++r) { x1[r] = act ? bf16_to_f32(xr1[r]) : 0.0f; x2[r] = act ? bf16_to_f32(xr2[r]) : 0.0f; }
    const float kx1 = act ? bf16_to_f32(kxr1) : 0.0f, kx2 = act ? bf16_to_f32(kxr2) : 0.0f;
    const float kw1 = act ? bf16_to_f32(kwr1) : 0.0f, kw2 = act ? bf16_to_f32(kwr2) : 0.0f;
    // ---- the two query heads: norm + rope on every wave, bf16 image in this wave's LDS slice ------------
    float qa[REP][2];
    {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const float inv = rsqrtf(lane_sum<64>(x1[r] * x1[r] + x2[r] * x2[r]) / (float)HD + eps);
            norm_rope_pair(x1[r], x2[r], w1, w2, inv, rc, rs, qa[r][0], qa[r][1]);
            if (act) {
                s_q[wave][r][lane] = f32_to_bf16(qa[r][0]);
                s_q[wave][r][lane + HALF] = f32_to_bf16(qa[r][1]);
            }
        }
    }
    // ---- the token's own key and value: cache append (waves WAVES-1 / WAVES-2) + merge terms -----------
    // The arithmetic runs on every wave and only the stores are wave-dependent: with the USES inside a wave-dependent branch hipcc sinks
    // the k / v row requests into it, i.e. behind the K / V stream (in-order return: the append waves would sit out the whole stream
    // before they could start).  It all happens while that stream is in flight.
    {
        const float inv = rsqrtf(lane_sum<64>(kx1 * kx1 + kx2 * kx2) / (float)HD + eps);
        float k1, k2;
        norm_rope_pair(kx1, kx2, kw1, kw2, inv, rc, rs, k1, k2);
        if (wave == WAVES - 1 && act) {
            bf16_t* dk = cache.k + cache.off(b, kvh, pos);
            dk[lane] = f32_to_bf16(k1);
            dk[lane + HALF] = f32_to_bf16(k2);
        }
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const float d = lane_sum<64>(act ? qa[r][0] * k1 + qa[r][1] * k2 : 0.0f);
            if (wave == WAVES - 1 && lane == 0) s_new[r] = d * scale;
        }
        bf16_t* dvf = cache.vf + cache.off(b, kvh, 0);
#pragma unroll
        for (int ii = 0; ii < (HD + 63) / 64; ++ii) {
            const int i = lane + 64 * ii;
            if (i < HD) {
                const bf16_t v = vown[ii];
                s_vn[i] = bf16_to_f32(v);                        // every wave writes the same value
                if (wave == WAVES - 2) dvf[vfrag_index<HD>(pos, i)] = v;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    mfma_bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (fr < REP) u = *reinterpret_cast<const uint4*>(&s_q[wave][fr][ks * 32 + g * 8]);
        qf[ks] = __builtin_bit_cast(mfma_bf16x8, u);
    }
    QASR_STAMP(1);
    // ---- sweep -------------------------------------------------------------------------------------
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.0f;
    for (int c0 = wave; c0 < nchunks; c0 += WAVES * UNR) {
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
                        const float v = key0 + h * 16 + j < pos ? sc[h][j] * scale : -INFINITY;   // select: stale rows may be NaN
                        sc[h][j] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = rows4_max(mx);
                const float m_new = fmaxf(m_run, mx);                    // finite: a chunk below nchunks has a valid key
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
                if (chunk * 32 + 32 > pos) {                             // partial chunk: stale V rows would give 0 * NaN
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
    QASR_STAMP(2);
    l_run = rows4_sum(l_run);
    if (lane < REP) { s_m[wave][lane] = m_run; s_l[wave][lane] = l_run; }
    if (g == 0) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            s_o[wave][0][dt * 16 + fr] = o[dt][0];
            s_o[wave][1][dt * 16 + fr] = o[dt][1];
        }
    }
    QASR_STAMP(3);
    QASR_STAMP(4);
    __syncthreads();
    QASR_STAMP(5);
    // merge waves + the token's own key/value: thread t < REP*HD owns one output element
    for (int i = tid; i < REP * HD; i += WAVES * 64) {
        const int r = i / HD, d = i - r * HD;
        float mm = s_new[r];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) mm = fmaxf(mm, s_m[w][r]);
        const float pn = __expf(s_new[r] - mm);
        float num = pn * s_vn[d], den = pn;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const float mw = s_m[w][r];
            if (mw != -INFINITY) {                                       // waves without a chunk left s_o unwritten
                const float a = __expf(mw - mm);
                num += s_o[w][r][d] * a;
                den += s_l[w][r] * a;
            }
        }
        out[(long)b * heads * HD + (long)(kvh * REP + r) * HD + d] = f32_to_bf16(num / den);
    }
    QASR_STAMP(6);
#undef QASR_STAMP
}

void decode_attention_launch(const bf16_t* qkv, const int* ctx_len, int B, int heads, int kv_heads, int hd,
                             const bf16_t* qn_w, const bf16_t* kn_w, float eps, const float* rope_cos,
                             const float* rope_sin, KVLayout cache, bf16_t* out, hipStream_t s, unsigned long long* dbg) {
    if (B <= 0) return;
    if (heads != 2 * kv_heads) throw std::invalid_argument("decode attention: built for 2 query heads per kv head");
    if (!cache.vf) throw std::invalid_argument("decode attention: the fragment-major V image is not allocated");
    if (cache.max_ctx % 32) throw std::invalid_argument("decode attention: cache capacity must be a multiple of 32 keys");
    const float scale = 1.0f / sqrtf((float)hd);
    dim3 grid(kv_heads, B);
    // A/B knobs: waves per workgroup (8 with two chunks in flight per wave | 16 with one) and speculative first loads
    const int nw = tuning().da_waves;
    // da_spec: 0 never | 1 always | 2 auto: only while the launch is latency-bound (few batch rows: the wasted bytes of chunks past the
    // context cost nothing there, and the K / V round trip no longer waits for the ctx_len round trip)
    // da_spec: 0 never | 1 the first chunk of every wave | 2 every chunk of the first round | 3 (default) 2 up to 8 batch rows (the launch is
    // latency-bound there and the bytes of chunks past the context cost nothing), 1 above
    const int spec_knob = tuning().da_spec;
    const int spec = spec_knob == 3 ? (B <= 8 ? 2 : 1) : spec_knob;
    const bool early = tuning().da_earlyq != 0;
#define QASR_DAM_GO(HD_, W_, U_, S_, E_)                                                                                         \
    hipLaunchKernelGGL((decode_attention_mfma_kernel<HD_, W_, U_, S_, E_>), grid, dim3(W_ * 64), 0, s, qkv, ctx_len, heads, kv_heads, \
                       qn_w, kn_w, eps, rope_cos, rope_sin, cache, out, scale, dbg)
    if (hd == 128) {
        if ((tuning().da_unr == 1 || tuning_thread_is_shared()) && nw == 8) {       // one chunk in flight per wave: ~130 registers, two such workgroups (or one + a GEMV's) share a CU
            if (spec) QASR_DAM_GO(128, 8, 1, 1, false); else QASR_DAM_GO(128, 8, 1, 0, false);
        } else
        if (nw == 16) { if (spec) QASR_DAM_GO(128, 16, 1, 2, true); else QASR_DAM_GO(128, 16, 1, 0, true); }
        else if (early) { if (spec == 2) QASR_DAM_GO(128, 8, 2, 2, true); else if (spec == 1) QASR_DAM_GO(128, 8, 2, 1, true); else QASR_DAM_GO(128, 8, 2, 0, true); }
        else { if (spec == 2) QASR_DAM_GO(128, 8, 2, 2, false); else if (spec == 1) QASR_DAM_GO(128, 8, 2, 1, false); else QASR_DAM_GO(128, 8, 2, 0, false); }
    } else if (hd == 32) {
        QASR_DAM_GO(32, 8, 1, 0, true);
    } else
        throw std::invalid_argument("decode attention: unsupported head_dim");
#undef QASR_DAM_GO
}

}  // namespace qasr
