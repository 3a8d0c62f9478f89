// dec_lmhead.hip -- persistent LM head with fused argmax partials, row argmax, greedy bookkeeping + next-token embedding gather (declarations: dec_kernels.h).
#include "dec_kernels.h"
#include "dec_epilogue.h"
#include "dec_quant.h"
#include <cstdlib>
#include <cstdio>

namespace qasr {

// ------------------------------------------------------------------------------------------------
// LM head (tied embedding, N = vocab): persistent form.  The batch rows are normalised (final RMSNorm)
// and staged into LDS ONCE per workgroup; every wave then walks its own 16-row weight tiles over the
// full K with a two-deep register pipeline (16 k-steps = 16 KB per buffer in flight per wave), so
// there is no cross-wave reduction and no barrier after the staging.  Each wave keeps a running
// (max, lowest index) per batch row over bf16-rounded logits; the workgroup publishes one partial.
// ------------------------------------------------------------------------------------------------
constexpr int LMH_WAVES = 8, LMH_CH = 16;     // waves per workgroup, k-steps per register buffer

struct LmHeadArgs {
    const bf16_t* W;        // fragment-major packed [N/16][K/32][64 lanes][8]
    const bf16_t* X;        // [B][K] un-normalised hidden rows
    const bf16_t* norm_w;   // [K]
    float eps;
    int B, N;
    float* logits;          // optional [B][N]
    float* part_val;        // [B][gridDim.x]
    int* part_idx;
    int diag;               // 1: diagnostic build of the loop without LDS reads / MFMA (wrong results, timing only)
    int wg_fastest;         // tile -> wave order (knob lmh_order): 1 workgroup fastest | 0 wave fastest (rounds 1-3)
};

typedef __attribute__((ext_vector_type(4))) unsigned lmh_u32x4;
// NTW: the weight stream (read once per step, 311 MB) by non-temporal loads -- a TEMPLATE parameter: as a run-time branch hipcc merged the
// two load blocks and dropped the hint (round 2); profiles/r04_ab_gemv_nt.txt
template <int K, int NB, bool NTW>
__global__ __launch_bounds__(LMH_WAVES * 64) void lm_head_kernel(LmHeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    constexpr int KCH = K / 8, XSTRIDE = 2 * K + 16, NC = K / (32 * LMH_CH);   // chunks per tile
    constexpr int TPR = 32, XI = KCH / TPR;                                      // 16 rows per staging pass
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fc = lane >> 4;
    char* s_x = dsm;
    // ---- weights of the first chunk go in flight before the activation staging -------------------------
    // wave index over the grid with the WORKGROUP fastest: the ntiles % total_waves waves that own one tile more are then spread over all
    // workgroups (5 or 6 per workgroup at vocab 151 936) instead of filling the first 163 of 256 -- which streamed 40 tiles each against 32
    const int total_waves = gridDim.x * LMH_WAVES, gw = a.wg_fastest ? wave * gridDim.x + blockIdx.x : blockIdx.x * LMH_WAVES + wave;
    const int ntiles = a.N / 16;
    const int my_tiles = gw < ntiles ? (ntiles - gw + total_waves - 1) / total_waves : 0;
    const int nitems = my_tiles * NC;
    uint4 wa[LMH_CH], wb[LMH_CH];
    auto issue = [&](uint4 (&w)[LMH_CH], int item) {
        const int tile = gw + (item / NC) * total_waves, ch = item % NC;
        const bf16_t* wp = a.W + ((long)tile * (K / 32) + ch * LMH_CH) * 512 + lane * 8;   // packed, see pack_mfma_a_kernel
#pragma unroll
        for (int i = 0; i < LMH_CH; ++i) {
            if constexpr (NTW) {
                const lmh_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const lmh_u32x4*>(wp + i * 512));
                w[i] = make_uint4(v.x, v.y, v.z, v.w);
            } else {
                w[i] = *reinterpret_cast<const uint4*>(wp + i * 512);
            }
        }
    };
    // ---- stage + RMSNorm the batch rows, 16 rows per pass (row on 32 adjacent lanes) --------------------
    // Request order: norm weights and the first pass's rows, THEN the first weight tiles -- loads come back in order, so rows asked for behind the
    // weight prefetch could not be normalised before 16 KB of weights per wave had arrived from HBM; and the norm weights as whole 16-byte loads up
    // front (inside the loop hipcc split each into four narrow loads, two of them behind a full wait).
    {
        const int srow = tid / TPR, scol = tid % TPR;
        constexpr int NPRE = NB <= 2 ? NB : 1;          // passes whose rows are requested in front of the weights (registers: 4 per chunk)
        uint4 nwr[XI], xr0[NPRE][XI];
#pragma unroll
        for (int i = 0; i < XI; ++i) nwr[i] = reinterpret_cast<const uint4*>(a.norm_w)[scol + i * TPR];
#pragma unroll
        for (int nb = 0; nb < NPRE; ++nb) {
            const bf16_t* xp0 = a.X + (long)(nb * 16 + srow < a.B ? nb * 16 + srow : 0) * K + scol * 8;
#pragma unroll
            for (int i = 0; i < XI; ++i) xr0[nb][i] = *reinterpret_cast<const uint4*>(xp0 + i * TPR * 8);
        }
        issue(wa, 0);       // unconditional (lm_head_supported: at least one tile per wave of the largest grid): under a branch hipcc's counted waits
                            // for the rows would have to assume the shorter path and wait for the weights as well
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int r = nb * 16 + srow;
            const bool live = r < a.B;
            const bf16_t* xp = a.X + (long)(live ? r : 0) * K + scol * 8;
            uint4 xr[XI];
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                if (nb < NPRE) xr[i] = xr0[nb < NPRE ? nb : 0][i];
                else xr[i] = *reinterpret_cast<const uint4*>(xp + i * TPR * 8);
            }
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const bf16_t* e = reinterpret_cast<const bf16_t*>(&xr[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = bf16_to_f32(e[j]); ss = fmaf(f, f, ss); }
            }
            if constexpr (TPR >= 8) ss = lane_sum<(TPR >= 8 ? TPR : 8)>(ss);
            else {
#pragma unroll
                for (int ofs = 1; ofs < TPR; ofs <<= 1) ss += __shfl_xor(ss, ofs, 64);
            }
            const float inv = rsqrtf(ss / (float)K + a.eps);
            char* xrow = s_x + (size_t)r * XSTRIDE + scol * 16;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const uint4 nw = nwr[i];
                const bf16_t* e = reinterpret_cast<const bf16_t*>(&xr[i]);
                const bf16_t* we = reinterpret_cast<const bf16_t*>(&nw);
                uint4 o;
                bf16_t* oe = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    oe[j] = live ? f32_to_bf16(bf16_to_f32(we[j]) * bf16_round(bf16_to_f32(e[j]) * inv)) : (bf16_t)0;
                *reinterpret_cast<uint4*>(xrow + i * TPR * 16) = o;
            }
        }
    }
    __syncthreads();
    // ---- stream the weight tiles ---------------------------------------------------------------------------
    f32x4 acc[NB];
    float best[NB];
    int bidx[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { acc[b] = f32x4{0.f, 0.f, 0.f, 0.f}; best[b] = -INFINITY; bidx[b] = 0x7fffffff; }
    auto consume = [&](const uint4 (&w)[LMH_CH], int item) {
        const int tile = gw + (item / NC) * total_waves, ch = item % NC;
        if (a.diag) {
#pragma unroll
            for (int i = 0; i < LMH_CH; ++i) acc[0][0] += __uint_as_float(w[i].x ^ w[i].y ^ w[i].z ^ w[i].w);
        } else
#pragma unroll
        for (int i = 0; i < LMH_CH; ++i) {
            const int kb = ((ch * LMH_CH + i) * 32 + fc * 8) * 2;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const uint4 xf = *reinterpret_cast<const uint4*>(s_x + (size_t)(b * 16 + fr) * XSTRIDE + kb);
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, w[i]),
                                                                 __builtin_bit_cast(mfma_bf16x8, xf), acc[b], 0, 0, 0);
            }
        }
        if (ch == NC - 1) {                                   // tile complete: acc[b][j] = logit[b*16+fr][tile*16+fc*4+j]
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int row = b * 16 + fr;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = tile * 16 + fc * 4 + j;
                    const float v = bf16_round(acc[b][j]);
                    if (a.logits && row < a.B) a.logits[(long)row * a.N + n] = v;
                    if (v > best[b] || (v == best[b] && n < bidx[b])) { best[b] = v; bidx[b] = n; }
                }
                acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    for (int item = 0; item < nitems; item += 2) {
        if (item + 1 < nitems) issue(wb, item + 1);
        consume(wa, item);
        if (item + 1 < nitems) {
            if (item + 2 < nitems) issue(wa, item + 2);
            consume(wb, item + 1);
        }
    }
    // ---- argmax partial of the workgroup ---------------------------------------------------------------------
    float* s_v = reinterpret_cast<float*>(dsm);               // the activation image is dead now
    int* s_i = reinterpret_cast<int*>(dsm + LMH_WAVES * NB * 16 * sizeof(float));
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int ofs = 16; ofs < 64; ofs <<= 1) {
            const float ov = __shfl_xor(best[b], ofs, 64);
            const int oi = __shfl_xor(bidx[b], ofs, 64);
            if (ov > best[b] || (ov == best[b] && oi < bidx[b])) { best[b] = ov; bidx[b] = oi; }
        }
        if (fc == 0) { s_v[(wave * NB + b) * 16 + fr] = best[b]; s_i[(wave * NB + b) * 16 + fr] = bidx[b]; }
    }
    __syncthreads();
    if (tid < NB * 16 && tid < a.B) {
        const int b = tid >> 4, r = tid & 15;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < LMH_WAVES; ++w) {
            const float ov = s_v[(w * NB + b) * 16 + r];
            const int oi = s_i[(w * NB + b) * 16 + r];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        a.part_val[(long)tid * gridDim.x + blockIdx.x] = bv;
        a.part_idx[(long)tid * gridDim.x + blockIdx.x] = bi;
    }
}

static int lmh_grid() {
    static const int g = tuning().lmh_grid;      // frozen at first use: sizes the argmax partial buffers
    return g;
}

template <int K, int NB>
static void lm_head_go(const LmHeadArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)NB * 16 * (2 * K + 16);
    auto go = [&](auto kern) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds);
        hipLaunchKernelGGL(kern, dim3(lmh_grid()), dim3(LMH_WAVES * 64), lds, s, a);
    };
    if (tuning().lmh_nt) go(lm_head_kernel<K, NB, true>);
    else go(lm_head_kernel<K, NB, false>);
}

bool lm_head_supported(int N, int K) { return (K == 1024 || K == 2048) && N % 16 == 0 && N / 16 >= 512 * LMH_WAVES; }
int lm_head_parts(int N, int K) { return lm_head_supported(N, K) ? lmh_grid() : decode_gemv_blocks(DEC_EPI_LOGITS, N); }

// final RMSNorm + tied LM head + per-workgroup argmax partials; returns the number of partials per row
int lm_head_launch(const bf16_t* W, const bf16_t* Wp, const bf16_t* X, const bf16_t* norm_w, float eps, int B, int N, int K,
                   float* logits, float* part_val, int* part_idx, bf16_t* norm_scratch, hipStream_t s) {
    if (B <= 0) return 0;
    const int nb = (B + 15) / 16;
    if (Wp && lm_head_supported(N, K) && nb <= (K == 1024 ? 4 : 2)) {
        const int diag = tuning().lmh_diag;
        LmHeadArgs a{Wp, X, norm_w, eps, B, N, logits, part_val, part_idx, diag, tuning().lmh_order};
        if (K == 1024) {
            switch (nb) {
                case 1: lm_head_go<1024, 1>(a, s); break;
                case 2: lm_head_go<1024, 2>(a, s); break;
                case 3: lm_head_go<1024, 3>(a, s); break;
                default: lm_head_go<1024, 4>(a, s); break;
            }
        } else {
            if (nb == 1) lm_head_go<2048, 1>(a, s); else lm_head_go<2048, 2>(a, s);
        }
        return lmh_grid();
    }
    if (Wp && lm_head_supported(N, K)) throw std::length_error("LM head: batch rows exceed the LDS image at this hidden size");
    DecGemvArgs g{};
    g.W = W; g.X = X; g.B = B; g.N = N; g.K = K; g.logits = logits; g.part_val = part_val; g.part_idx = part_idx;
    return decode_gemv_fused_launch(DEC_EPI_LOGITS, g, norm_w, eps, norm_scratch, s);
}

// ------------------------------------------------------------------------------------------------
// Row argmax over bf16 logits with MLX argMax's tie rule (lowest index): one workgroup per row.  Used by the forced
// aligner's timestamp head (ForcedAligner.swift:291-296).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_rows_kernel(const bf16_t* __restrict__ x, long ld, int n, int* __restrict__ out) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const bf16_t* row = x + (long)blockIdx.x * ld;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = bf16_to_f32(row[i]);
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }     // NaN never wins
    }
    s_v[threadIdx.x] = bv;
    s_i[threadIdx.x] = bi;
    __syncthreads();
    for (int ofs = 128; ofs > 0; ofs >>= 1) {
        if ((int)threadIdx.x < ofs) {
            const float ov = s_v[threadIdx.x + ofs];
            const int oi = s_i[threadIdx.x + ofs];
            if (oi != 0x7fffffff && (s_i[threadIdx.x] == 0x7fffffff || ov > s_v[threadIdx.x] || (ov == s_v[threadIdx.x] && oi < s_i[threadIdx.x]))) {
                s_v[threadIdx.x] = ov;
                s_i[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = s_i[0] == 0x7fffffff ? 0 : s_i[0];
}

void argmax_rows_launch(const bf16_t* x, long ld, int rows, int n, int* out, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, s, x, ld, n, out);
}

// ------------------------------------------------------------------------------------------------
// greedy bookkeeping + next-token embedding gather: one workgroup per batch row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void greedy_finalize_kernel(const float* __restrict__ part_val,
                                                              const int* __restrict__ part_idx, int n_parts,
                                                              GreedyState st, int advance_ctx,
                                                              const bf16_t* __restrict__ embed, bf16_t* __restrict__ x, int H,
                                                              RopeRows rr, QuantRaw qe) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b == 0) {
        for (int i = tid; i < st.clear_words; i += 256) st.clear[i] = 0u;
        if (tid == 0 && st.clear_words) st.clear[st.clear_words + 32] += 1u;      // the step sequence word (dec_chain.h CHAIN_SEQ_WORD)
    }
    // Request order (the launch was a chain of ~9 dependent round trips, one per step of the bookkeeping): everything that depends on nothing --
    // the row's state words, the argmax partials -- is requested up front; the rope row of the NEXT step's position (copied next to the batch row
    // so that the attention of that step need not chain a table lookup behind the ctx_len load) is one round trip behind ctx_len, both tables together.
    const int ctx_now = st.ctx_len[b];
    const int fin_now = st.finished[b], len_now = st.lens[b];
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    if (n_parts <= 256) {                              // one partial per thread: a plain load each, in flight with the words above
        if (tid < n_parts) {
            const float v = part_val[(long)b * n_parts + tid];
            const int n = part_idx[(long)b * n_parts + tid];
            if (v > best || (v == best && n < bidx)) { best = v; bidx = n; }
        }
    } else {
        for (int i = tid; i < n_parts; i += 256) {
            const float v = part_val[(long)b * n_parts + i];
            const int n = part_idx[(long)b * n_parts + i];
            if (v > best || (v == best && n < bidx)) { best = v; bidx = n; }
        }
    }
    const int next_pos = ctx_now + (advance_ctx ? 1 : 0);
    if (tid < rr.half) {
        const float c = rr.cos_table[(long)next_pos * rr.half + tid], sn = rr.sin_table[(long)next_pos * rr.half + tid];
        rr.cos_rows[(long)b * rr.half + tid] = c;
        rr.sin_rows[(long)b * rr.half + tid] = sn;
    }
    s_v[tid] = best;
    s_i[tid] = bidx;
    __syncthreads();
    for (int ofs = 128; ofs > 0; ofs >>= 1) {
        if (tid < ofs) {
            const float ov = s_v[tid + ofs];
            const int oi = s_i[tid + ofs];
            if (ov > s_v[tid] || (ov == s_v[tid] && oi < s_i[tid])) { s_v[tid] = ov; s_i[tid] = oi; }
        }
        __syncthreads();
    }
    const bool sane = (unsigned)s_i[0] < (unsigned)st.vocab && fabsf(s_v[0]) <= 3.0e38f;   // false for NaN / inf / no winner
    const int tok = (unsigned)s_i[0] < (unsigned)st.vocab ? s_i[0] : 0;
    if (tid == 0) {
        if (!sane && !fin_now) atomicOr(st.err, 1);
        if (advance_ctx) st.ctx_len[b] = ctx_now + 1;
        if (!fin_now) {
            const int n = len_now;
            st.tokens[(long)b * (st.max_new + 1) + n] = tok;
            st.lens[b] = n + 1;
            if ((tok == st.eos && !st.ignore_eos) || n + 1 >= st.max_tokens) {
                st.finished[b] = 1;
                atomicSub(st.n_active, 1);
            }
        }
    }
    uint4* dst = reinterpret_cast<uint4*>(x + (long)b * H);
    if (qe.wq) {                                     // quantised table: dequantized(row) (PreQuantizedEmbedding.swift:35-42)
        for (int i = tid; i < H / 8; i += 256) dst[i] = quant_dequant_chunk(qe, tok, i);
        return;
    }
    const uint4* src = reinterpret_cast<const uint4*>(embed + (long)tok * H);
    for (int i = tid; i < H / 8; i += 256) dst[i] = src[i];
}

void greedy_finalize_launch(const float* part_val, const int* part_idx, int n_parts, GreedyState st, int B,
                            int advance_ctx, const bf16_t* embed, bf16_t* x, int H, RopeRows rr, hipStream_t s,
                            const QuantRaw* qembed) {
    if (B <= 0) return;
    hipLaunchKernelGGL(greedy_finalize_kernel, dim3(B), dim3(256), 0, s, part_val, part_idx, n_parts, st, advance_ctx,
                       embed, x, H, rr, qembed ? *qembed : QuantRaw{});
}

}  // namespace qasr
