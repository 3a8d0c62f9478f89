// ctc_attention.hip -- full (unmasked) multi-head self-attention of the Omnilingual wav2vec2 encoder layers
// (Wav2Vec2EncoderLayer.swift:36-60 -> MLXCommon/SDPA.swift:18-37, mask nil).  Its own translation unit because it is built
// with -mllvm -amdgpu-mfma-vgpr-form (MFMA results stay in VGPRs: without it hipcc parks both the score and the output
// accumulators in AGPRs and moves 220 registers per key tile between the two files) and -fno-honor-nans (no canonicalising
// v_max in front of every fmaxf on an MFMA result; masked scores are -inf, never NaN: padding keys are zero-filled).
#include "ctc_kernels.h"
#include "tuning.h"
#include <stdexcept>

namespace qasr {

// ------------------------------------------------------------------------------------------------
// Full multi-head attention, flash style.  Workgroup = 64 query rows of one (clip, head): 4 waves x 16 rows.
// Per 64-key tile: K rows and a transposed V image go to LDS once for the four waves; S = Q K^T as 16x16x32 MFMAs with the
// query fragments held in registers; online softmax on the accumulator layout (a lane holds query rows 4 (lane >> 4) + j,
// key column lane & 15 of each 16-key tile: row statistics = four DPP / shuffle steps over the 16 lanes); P (bf16) through a
// wave-private LDS image to become the A operand of O += P V.
// ------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void mha_attention_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ cu, int D,
                                                            bf16_t* __restrict__ out, float scale) {
    constexpr int KT = 64, KS = HD / 32, DT = HD / 16, LDK = HD + 8, LDV = KT + 8, LDP = KT + 8;
    __shared__ __attribute__((aligned(16))) bf16_t s_k[KT][LDK];
    __shared__ __attribute__((aligned(16))) bf16_t s_vt[HD][LDV];
    __shared__ __attribute__((aligned(16))) bf16_t s_p[4][16][LDP];
    const int clip = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    const int r0 = cu[clip], L = cu[clip + 1] - r0;
    if (q0 >= L) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fc = lane >> 4;
    const long ld = 3L * D;
    const bf16_t* qb = qkv + (long)r0 * ld + h * HD;
    const bf16_t* kb = qb + D;
    const bf16_t* vb = qb + 2 * D;
    // query fragments of this wave's 16 rows (A operand: row fr, k = 8 fc + e of each 32-wide k-step)
    mfma_bf16x8 qf[KS];
    const int qrow = q0 + wave * 16 + fr;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (qrow < L) u = *reinterpret_cast<const uint4*>(qb + (long)qrow * ld + s * 32 + fc * 8);
        qf[s] = __builtin_bit_cast(mfma_bf16x8, u);
    }
    f32x4 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, l_run[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int CH = HD / 8;
    for (int k0 = 0; k0 < L; k0 += KT) {
        __syncthreads();                                    // previous tile fully consumed
        for (int i = tid; i < KT * CH; i += 256) {
            const int key = i / CH, ch = i - key * CH;
            uint4 uk = make_uint4(0, 0, 0, 0), uv = make_uint4(0, 0, 0, 0);
            if (k0 + key < L) {
                uk = *reinterpret_cast<const uint4*>(kb + (long)(k0 + key) * ld + ch * 8);
                uv = *reinterpret_cast<const uint4*>(vb + (long)(k0 + key) * ld + ch * 8);
            }
            *reinterpret_cast<uint4*>(&s_k[key][ch * 8]) = uk;
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&uv);
#pragma unroll
            for (int j = 0; j < 8; ++j) s_vt[ch * 8 + j][key] = e[j];
        }
        __syncthreads();
        // S tile: 4 key sub-tiles of 16
        f32x4 sc[KT / 16];
#pragma unroll
        for (int kt = 0; kt < KT / 16; ++kt) {
            sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const mfma_bf16x8 kf = *reinterpret_cast<const mfma_bf16x8*>(&s_k[kt * 16 + fr][s * 32 + fc * 8]);
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[s], kf, sc[kt], 0, 0, 0);
            }
        }
        // online softmax: lane holds rows fc*4 + j, key column fr of each sub-tile
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kt = 0; kt < KT / 16; ++kt) {
            const bool valid = k0 + kt * 16 + fr < L;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = valid ? sc[kt][j] * scale : -INFINITY;
                sc[kt][j] = v;
                mx[j] = fmaxf(mx[j], v);
            }
        }
        float alpha[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mx[j] = lane_max16(mx[j]);
            const float mn = fmaxf(m_run[j], mx[j]);          // finite: every tile below L holds a valid key
            alpha[j] = __expf(m_run[j] - mn);
            m_run[j] = mn;
        }
        float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT / 16; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16_t pb = f32_to_bf16(__expf(sc[kt][j] - m_run[j]));
                rs[j] += bf16_to_f32(pb);
                s_p[wave][fc * 4 + j][kt * 16 + fr] = pb;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rs[j] = lane_sum<16>(rs[j]);
            l_run[j] = l_run[j] * alpha[j] + rs[j];
        }
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[d][j] *= alpha[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                      // the wave's own P image is complete
        // O += P V: A = P rows (this wave's 16 queries), B = V^T image
#pragma unroll
        for (int ks = 0; ks < KT / 32; ++ks) {
            const mfma_bf16x8 pf = *reinterpret_cast<const mfma_bf16x8*>(&s_p[wave][fr][ks * 32 + fc * 8]);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const mfma_bf16x8 vf = *reinterpret_cast<const mfma_bf16x8*>(&s_vt[d * 16 + fr][ks * 32 + fc * 8]);
                o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[d], 0, 0, 0);
            }
        }
    }
    // o[d][j] = O[row fc*4 + j][d*16 + fr]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = q0 + wave * 16 + fc * 4 + j;
        if (row < L) {
            const float inv = 1.0f / l_run[j];
#pragma unroll
            for (int d = 0; d < DT; ++d) out[((long)r0 + row) * D + h * HD + d * 16 + fr] = f32_to_bf16(o[d][j] * inv);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// head_dim 64 (every published Omnilingual variant): the same online softmax on 32x32x16 MFMAs with the score tile
// transposed, so that nothing but K and V tiles ever goes through LDS.
//   workgroup = 128 queries of one (clip, head): 4 waves x 32 queries, the query fragments in registers as the B operand;
//   S^T = K Q^T  per 32-key block (A = K rows from an XOR-swizzled LDS image): a lane holds, for ITS query (lane & 31),
//                the scores of keys (i & 3) + 8 (i >> 2) + 4 (lane >> 5) in accumulator register i -- the row statistics are
//                in-lane maxima / sums plus one v_permlane32_swap between the two lane halves;
//   O^T += V^T P^T: registers 8 s .. 8 s + 7 of the exponentiated block, packed to bf16, ARE the B fragment of k-step s
//                (cdna_hip_programming.md section 3, "an accumulator tile as the next MFMA's operand"); the matching A fragment
//                = V^T rows in the same permuted key order comes from the row-major V tile by ds_read_b64_tr_b16;
//   K / V tiles of 64 keys are double-buffered in LDS (loads of tile t + 1 issued before the MFMAs of tile t, written
//   after them: one barrier per tile); O is rescaled only in tiles where some row maximum grew (exact, wave-uniform).
// Row sums add the unrounded f32 probabilities (the 16x16 kernel above adds the bf16-rounded ones): |delta| ~ 1e-4 relative.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short tr_b16x4;

__device__ __forceinline__ uint2 lds_read_tr16(const bf16_t* p) {
    const tr_b16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_b16x4*)(p));
    return __builtin_bit_cast(uint2, r);
}

__device__ __forceinline__ float half_swap_max(float x) {       // max over lanes l and l ^ 32
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void mha64_attention_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ cu, int D,
                                                                 int heads, int n_pairs, int nqb, bf16_t* __restrict__ out,
                                                                 float c_exp /* scale * log2(e) */) {
    constexpr int HD = 64, KT = 64, QW = 32, CPT = 512 / (NW * 64);     // 16-byte chunks per thread, tile and operand
    __shared__ __attribute__((aligned(16))) bf16_t s_k[2][KT * HD];     // row = key (128 B), 16-byte chunk ch at ch ^ ((key >> 1) & 7)
    __shared__ __attribute__((aligned(16))) bf16_t s_v[2][KT * HD];     // row = key, chunk ch at ch ^ (((key >> 1) & 1) << 2)
    // workgroup -> (clip, head, query block): consecutive ids go round the 8 XCDs, so id & 7 picks the XCD and the query
    // blocks of one (clip, head) -- which all sweep the same K / V rows -- stay on one XCD's L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / nqb) * 8 + xcd;
    if (pair >= n_pairs) return;
    const int clip = pair / heads, hq = pair - clip * heads, q0 = (slot % nqb) * (QW * NW);
    const int r0 = cu[clip], L = cu[clip + 1] - r0;
    if (q0 >= L) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const long ld = 3L * D;
    const bf16_t* qb = qkv + (long)r0 * ld + hq * HD;
    const bf16_t* kb = qb + D;
    const bf16_t* vb = qb + 2 * D;
    // staging map: 512 16-byte chunks per tile and operand
    const int skey = tid >> 3, sch = tid & 7;                            // keys skey + (NW * 8) i
    uint4 kst[CPT], vst[CPT];
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int key = k0 + skey + NW * 8 * i;
            kst[i] = make_uint4(0, 0, 0, 0);
            vst[i] = make_uint4(0, 0, 0, 0);
            if (key < L) {
                kst[i] = *reinterpret_cast<const uint4*>(kb + (long)key * ld + sch * 8);
                vst[i] = *reinterpret_cast<const uint4*>(vb + (long)key * ld + sch * 8);
            }
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int key = skey + NW * 8 * i;
            *reinterpret_cast<uint4*>(&s_k[buf][key * HD + ((sch ^ ((key >> 1) & 7)) << 3)]) = kst[i];
            *reinterpret_cast<uint4*>(&s_v[buf][key * HD + ((sch ^ (((key >> 1) & 1) << 2)) << 3)]) = vst[i];
        }
    };
    stage_load(0);
    // query fragments (B operand): column = query r, k = 16 s + 8 h + e
    mfma_bf16x8 qf[4];
    const int qrow = q0 + wave * QW + r;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (qrow < L) u = *reinterpret_cast<const uint4*>(qb + (long)qrow * ld + s * 16 + h * 8);
        qf[s] = __builtin_bit_cast(mfma_bf16x8, u);
    }
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                               // m in raw score units; l = this lane's share of the row sum
    stage_write(0);
    __syncthreads();
    // lane-constant LDS offsets (elements)
    const int g = lane >> 4, li = lane & 15;
    const int v_row = 4 * (g >> 1) + (li >> 2);                          // + 32 kb + 16 s2 (+ 8): key row this lane addresses
    const int v_ch = 2 * (g & 1) + ((li & 3) >> 1), v_in = (li & 1) * 4;  // + 4 dt: chunk, element inside the chunk
    const int ntiles = (L + KT - 1) / KT;
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0 = t * KT;
        if (t + 1 < ntiles) stage_load(k0 + KT);
        const bf16_t* sk = s_k[buf];
        const bf16_t* sv = s_v[buf];
        f32x16 sc[2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[kbk][i] = 0.f;
            const int key = kbk * 32 + r;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const mfma_bf16x8 kf =
                    *reinterpret_cast<const mfma_bf16x8*>(&sk[key * HD + (((2 * s + h) ^ ((key >> 1) & 7)) << 3)]);
                sc[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sc[kbk], 0, 0, 0);
            }
        }
        if (k0 + KT > L) {                                               // last tile: keys past the clip (wave-uniform branch)
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (k0 + kbk * 32 + (i & 3) + 8 * (i >> 2) + 4 * h >= L) sc[kbk][i] = -INFINITY;
        }
        float mx = fmaxf(sc[0][0], sc[1][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, fmaxf(sc[0][i], sc[1][i]));
        mx = half_swap_max(mx);                                          // finite: key k0 < L is in every tile
        if (__any(mx > m_run)) {
            const float mn = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - mn) * c_exp);
            m_run = mn;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
        }
        const float mc = m_run * c_exp;
        mfma_bf16x8 pf[2][2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 u;
                unsigned* w = reinterpret_cast<unsigned*>(&u);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p0 = __builtin_amdgcn_exp2f(fmaf(sc[kbk][8 * s2 + 2 * e], c_exp, -mc));
                    const float p1 = __builtin_amdgcn_exp2f(fmaf(sc[kbk][8 * s2 + 2 * e + 1], c_exp, -mc));
                    l_run += p0 + p1;
                    w[e] = pack_bf16x2(p0, p1);
                }
                pf[kbk][s2] = __builtin_bit_cast(mfma_bf16x8, u);
            }
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int ka = kbk * 32 + s2 * 16 + v_row, kc = ka + 8;
                    const uint2 lo = lds_read_tr16(&sv[ka * HD + (((4 * dt + v_ch) ^ (((ka >> 1) & 1) << 2)) << 3) + v_in]);
                    const uint2 hi = lds_read_tr16(&sv[kc * HD + (((4 * dt + v_ch) ^ (((kc >> 1) & 1) << 2)) << 3) + v_in]);
                    const mfma_bf16x8 vf = __builtin_bit_cast(mfma_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kbk][s2], o[dt], 0, 0, 0);
                }
        if (t + 1 < ntiles) stage_write(buf ^ 1);
        __syncthreads();
    }
    // o[dt][i] = O[query r][dt * 32 + (i & 3) + 8 (i >> 2) + 4 h]
    const float inv = 1.0f / half_swap_sum(l_run);
    if (qrow < L) {
        bf16_t* dst = out + ((long)r0 + qrow) * D + hq * HD + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 v = make_float4(o[dt][4 * q4] * inv, o[dt][4 * q4 + 1] * inv, o[dt][4 * q4 + 2] * inv, o[dt][4 * q4 + 3] * inv);
                *reinterpret_cast<uint2*>(dst + dt * 32 + q4 * 8) = pack_bf16x4(v);
            }
    }
}

void mha_attention_launch(const bf16_t* qkv, const int* cu, int n_clips, int max_len, int heads, int head_dim, bf16_t* out,
                          hipStream_t s) {
    if (n_clips <= 0 || max_len <= 0) return;
    const int D = heads * head_dim;
    const float scale = 1.0f / sqrtf((float)head_dim);
    const int form = tuning().mha_form;
    if (head_dim == 64 && form >= 1) {
        const int qpw = form == 2 ? 256 : 128, nqb = cdiv(max_len, qpw), n_pairs = heads * n_clips;
        const dim3 grid(8 * nqb * cdiv(n_pairs, 8));
        const float c_exp = scale * 1.4426950408889634f;
        if (form == 2) hipLaunchKernelGGL(mha64_attention_kernel<8>, grid, dim3(512), 0, s, qkv, cu, D, heads, n_pairs, nqb, out, c_exp);
        else hipLaunchKernelGGL(mha64_attention_kernel<4>, grid, dim3(256), 0, s, qkv, cu, D, heads, n_pairs, nqb, out, c_exp);
        return;
    }
    dim3 grid(cdiv(max_len, 64), heads, n_clips);
    if (head_dim == 64) hipLaunchKernelGGL(mha_attention_kernel<64>, grid, dim3(256), 0, s, qkv, cu, D, out, scale);
    else if (head_dim == 32) hipLaunchKernelGGL(mha_attention_kernel<32>, grid, dim3(256), 0, s, qkv, cu, D, out, scale);
    else throw std::invalid_argument("attention: head_dim must be 32 or 64");
}

}  // namespace qasr
