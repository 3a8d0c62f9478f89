// gemm.h -- bf16 MFMA GEMM  C[M,N] = A[M,K] . W[N,K]^T  (both operands K-contiguous, f32 accumulate)
// with pluggable A-operand gathers (dense rows / row-offset table / implicit 3x3-stride-2 conv) and
// fused epilogues.  gfx950 only: v_mfma_f32_16x16x32_bf16, 64-lane waves, XOR-swizzled LDS tiles.
//
// Tile: 128 x 128 x 64 per workgroup of 4 waves (2 x 2, each wave 64 x 64 = 4 x 4 MFMA tiles), operands by direct-to-LDS
// loads into an LDS image whose 16-byte chunk index is XORed with (row & 7): the MFMA fragment reads (ds_read_b128, 16 rows x
// one chunk column per 16 lanes) are bank-conflict free.  The epilogue goes through LDS so global stores are 16 bytes per
// lane along rows.  Launches of very many tiles take the 256 x 256 persistent form of gemm_p8.h instead.
#pragma once
#include "common.h"
#include "tuning.h"
#include <cstdlib>
#include <type_traits>

namespace qasr {

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 64, GEMM_THREADS = 256;

// ------------------------------------------------------------------------------------------------
// A-operand loaders.  row_init(m) is called once per staged row, load(row, k) once per K-step and
// must return 8 consecutive bf16 (k .. k+7) or zeros.
// ------------------------------------------------------------------------------------------------
struct ADense {
    const bf16_t* A;
    long lda;
    int M, K;
    struct Row { const bf16_t* p; };
    __device__ __forceinline__ Row row_init(int m) const { return {m < M ? A + (long)m * lda : nullptr}; }
    __device__ __forceinline__ uint4 load(const Row& r, int k) const {
        if (r.p && k < K) return *reinterpret_cast<const uint4*>(r.p + k);
        return make_uint4(0, 0, 0, 0);
    }
    __device__ __forceinline__ const bf16_t* addr(const Row& r, int k) const { return (r.p && k < K) ? r.p + k : nullptr; }
    // K-tile form of addr: ktile(k0, c8) once per staged K-tile (k0 wave-uniform, c8 this lane's chunk offset), addr_kt per row
    struct KT { int k; };
    __device__ __forceinline__ KT ktile(int k0, int c8) const { return {k0 + c8}; }
    __device__ __forceinline__ const bf16_t* addr_kt(const Row& r, const KT& t) const { return addr(r, t.k); }
};

// rows addressed through an element-offset table (packed valid tokens -> conv3 output rows)
struct ARowTable {
    const bf16_t* A;
    const long* row_off;     // [M] element offsets
    int M, K;
    struct Row { const bf16_t* p; };
    __device__ __forceinline__ Row row_init(int m) const { return {m < M ? A + row_off[m] : nullptr}; }
    __device__ __forceinline__ uint4 load(const Row& r, int k) const {
        if (r.p && k < K) return *reinterpret_cast<const uint4*>(r.p + k);
        return make_uint4(0, 0, 0, 0);
    }
    __device__ __forceinline__ const bf16_t* addr(const Row& r, int k) const { return (r.p && k < K) ? r.p + k : nullptr; }
    // K-tile form of addr: ktile(k0, c8) once per staged K-tile (k0 wave-uniform, c8 this lane's chunk offset), addr_kt per row
    struct KT { int k; };
    __device__ __forceinline__ KT ktile(int k0, int c8) const { return {k0 + c8}; }
    __device__ __forceinline__ const bf16_t* addr_kt(const Row& r, const KT& t) const { return addr(r, t.k); }
};

// Implicit GEMM for a 3x3 / stride 2 / pad 1 convolution over NHWC bf16 input [img][H][W][C].
// K index = (kh*3 + kw) * C + ci  (the reference's MLX weight layout [out][kh][kw][in],
// Sources/Qwen3ASR/WeightLoading.swift:284).  Output pixel order: hw_major ? m = (img*OH+oh)*OW+ow
// : m = (img*OW+ow)*OH+oh  (the latter makes conv3's rows directly the conv_out operand).
struct AConv3x3s2 {
    const bf16_t* in;
    int H, W, C, OH, OW, M, K;
    bool hw_major;
    struct Row { const bf16_t* base; unsigned mask; };   // base = pixel (2oh-1, 2ow-1); mask bit t = tap valid
    __device__ __forceinline__ Row row_init(int m) const {
        if (m >= M) return {nullptr, 0u};
        int img = m / (OH * OW), rem = m - img * (OH * OW);
        int oh, ow;
        if (hw_major) { oh = rem / OW; ow = rem - oh * OW; } else { ow = rem / OH; oh = rem - ow * OH; }
        int ih0 = 2 * oh - 1, iw0 = 2 * ow - 1;
        unsigned mask = 0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int ih = ih0 + kh, iw = iw0 + kw;
                if (ih >= 0 && ih < H && iw >= 0 && iw < W) mask |= 1u << (kh * 3 + kw);
            }
        return {in + (((long)img * H + ih0) * W + iw0) * C, mask};
    }
    __device__ __forceinline__ uint4 load(const Row& r, int k) const {
        if (k >= K) return make_uint4(0, 0, 0, 0);
        int tap = k / C, ci = k - tap * C;
        if (!((r.mask >> tap) & 1u)) return make_uint4(0, 0, 0, 0);
        int kh = tap / 3, kw = tap - kh * 3;
        return *reinterpret_cast<const uint4*>(r.base + ((long)kh * W + kw) * C + ci);
    }
    __device__ __forceinline__ const bf16_t* addr(const Row& r, int k) const {
        if (k >= K) return nullptr;
        // k / C without an integer division (C is a run-time value): exact for k < 2^20 with the rounded-up reciprocal
        int tap = (int)__umulhi((unsigned)k, (0xffffffffu / (unsigned)C) + 1u), ci = k - tap * C;
        if (!((r.mask >> tap) & 1u)) return nullptr;
        int kh = tap / 3, kw = tap - kh * 3;
        return r.base + ((long)kh * W + kw) * C + ci;
    }
    struct KT { int k; };
    __device__ __forceinline__ KT ktile(int k0, int c8) const { return {k0 + c8}; }
    __device__ __forceinline__ const bf16_t* addr_kt(const Row& r, const KT& t) const { return addr(r, t.k); }
};

// The same gather for C >= 64 (every real geometry: 480) with the K-tile's tap decomposed ONCE per staged K-tile on the scalar unit: a
// 64-wide K-tile starts inside tap0 and ends inside tap0 or tap0 + 1, so a lane only adds its chunk offset, tests one wrap and selects one of
// two tap offsets -- the per-chunk division, tap / 3 and multiplies of addr() sat on the critical path of the implicit-conv launches (a dense
// GEMM of the same shape ran 18 % faster, round 1; adding one more select to addr() cost the encoder 3 ms, profiles/r04_ab_conv_korder.txt).
// Same k order, same values: bit-identical to AConv3x3s2.
struct AConv3x3s2W : AConv3x3s2 {
    struct KT { int off; int tap; bool in_k; };
    __device__ __forceinline__ KT ktile(int k0, int c8) const {
        const unsigned k0u = (unsigned)__builtin_amdgcn_readfirstlane(k0);
        const int tap0 = (int)__umulhi(k0u, (0xffffffffu / (unsigned)C) + 1u), ci0 = (int)k0u - tap0 * C;      // scalar
        const int tap1 = tap0 + 1;
        const int off0 = (((tap0 * 11) >> 5) * W + (tap0 - 3 * ((tap0 * 11) >> 5))) * C;                        // (kh * W + kw) * C, tap < 10
        const int off1 = (((tap1 * 11) >> 5) * W + (tap1 - 3 * ((tap1 * 11) >> 5))) * C;
        const int c = ci0 + c8;
        const bool wrap = c >= C;
        return {(wrap ? off1 - C : off0) + c, wrap ? tap1 : tap0, (int)k0u + c8 < K};
    }
    __device__ __forceinline__ const bf16_t* addr_kt(const Row& r, const KT& t) const {
        return (t.in_k && ((r.mask >> t.tap) & 1u)) ? r.base + t.off : nullptr;
    }
};

// ------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

__device__ __forceinline__ float gemm_swiglu(float g_acc, float u_acc) {
    float g = bf16_round(g_acc), u = bf16_round(u_acc);
    // sigmoid through v_exp_f32 + v_rcp_f32 (1 ulp each): the library expf + IEEE division cost ~20 instructions per element
    // and the result is rounded to bf16 anyway; 1.1 G of these per prompt pass sit in the gate/up GEMM's epilogue
    float sg = bf16_round(g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g)));
    return bf16_round(sg * u);
}

__device__ __forceinline__ int gemm_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// MODE 0: epi(m, n, acc[n..n+3]).  MODE 1 (SwiGLU): weight rows come in blocks of 32 = 16 gate rows +
// the 16 matching up rows; epi(m, j, act[j..j+3]) receives bf16(bf16(silu(bf16 g)) * bf16 u) for the
// N/2 fused outputs (QuantizedTextDecoder.swift:132-137).
//
// Operands are staged with direct-to-LDS loads (global_load_lds_dwordx4): no VGPR round trip and no ds_write_b128 (LDS
// stores run at ~79 B/clk/CU on gfx950 and were the busiest pipe of the first, register-staged loop, removed in round 2).
// One wave instruction fills 1 KiB of LDS linearly = 8 tile rows; the XOR chunk swizzle moves to the per-lane SOURCE address
// (lane l of an 8-row group reads chunk (l%8) ^ (l/8)), the fragment reads keep the same swizzle.  Masked chunks (row tail,
// K tail, conv padding taps) read a 16-byte block of zeros.
//
// One body, two kernels:
//   NBUF = 2  gemm_nt_glds_kernel   64 KiB, tile k+1 in flight during the MFMAs of tile k, one barrier per K-step; 2 workgroups
//             per CU: the form for launches of few tiles
//   NBUF = 1  gemm_nt_glds1_kernel  ONE 32 KiB buffer, two barriers per K-step, no software double buffering: 4-5 workgroups
//             fit a CU and the overlap of loads with MFMAs comes from the other resident workgroups: the form for launches of
//             many tiles (and 256 x 256 tiles for launches of very many: gemm_p8.h)
// Every form sums an output in the same k order: bit-identical results (tests/test_gpu_gemm.py).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// epilogue traits: an epilogue with a nested `Pre` splits into prefetch(m, n) -> Pre (its memory reads) and apply(m, n, v, pre)
template <class E, class = void>
struct epi_has_pre { static constexpr bool value = false; };
template <class E>
struct epi_has_pre<E, std::void_t<typename E::Pre>> { static constexpr bool value = true; };
template <class E, bool HAS>
struct epi_pre_type { struct type {}; };
template <class E>
struct epi_pre_type<E, true> { using type = typename E::Pre; };

template <class ALoad, class Epi, int MODE, int NBUF>
__device__ __forceinline__ void gemm_nt_128_body(char* __restrict__ smem /* [NBUF][A|B][128 rows x 128 B] */, const ALoad& aload,
                                                 const bf16_t* __restrict__ Wt, long ldw, int M, int N, int K, const Epi& epi,
                                                 const bf16_t* __restrict__ zeros, int tm = 1) {
    constexpr int OPB = GEMM_BM * 128;                  // bytes of one operand image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive tile ids (sharing an A row panel) land on one XCD's L2
    const int nbx = (N + GEMM_BN - 1) / GEMM_BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // tm > 1: the tile list runs in blocks of tm row panels, column tile by column tile inside a block, panel fastest (like gemm_p8.h): the
    // ~128 tiles an XCD has in flight then span tm row panels x 128 / tm column tiles instead of 128 / nbx panels x all nbx columns -- fewer
    // distinct operand bytes behind them (the q|k|v GEMM of the prompt pass fetched 484 MB for 35 MB of operands, profiles/r04_v1_pmc_per_kernel.csv)
    int m0, n0;
    if (tm > 1) {
        const int nby = (M + GEMM_BM - 1) / GEMM_BM;
        const int blk = bid / (tm * nbx), i = bid - blk * (tm * nbx);
        const int rows_in_blk = nby - blk * tm < tm ? nby - blk * tm : tm;
        m0 = (blk * tm + i % rows_in_blk) * GEMM_BM;
        n0 = (i / rows_in_blk) * GEMM_BN;
    } else {
        m0 = (bid / nbx) * GEMM_BM;
        n0 = (bid % nbx) * GEMM_BN;
    }

    // staging: wave w, instruction i covers tile rows (w*4 + i)*8 .. +7; lane -> (row + lane/8, LDS slot lane%8)
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;              // source chunk = slot ^ (row & 7), rows are 8-aligned per group
    typename ALoad::Row arow[4];
    const bf16_t* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + srow;
        arow[i] = aload.row_init(m0 + r);
        const int n = n0 + r;
        wrow[i] = n < N ? Wt + (long)n * ldw : nullptr;
    }
    auto stage = [&](int buf, int kt) {
        const int k = kt * GEMM_BK + schunk * 8;
        const typename ALoad::KT akt = aload.ktile(kt * GEMM_BK, schunk * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* pa = aload.addr_kt(arow[i], akt);
            const bf16_t* pb = (wrow[i] && k < K) ? wrow[i] + k : nullptr;
            const int off = (wave * 4 + i) * 1024;      // wave-uniform LDS base of this instruction
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pa ? pa : zeros), (lds_ptr_t)&smem[(buf * 2 + 0) * OPB + off], 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pb ? pb : zeros), (lds_ptr_t)&smem[(buf * 2 + 1) * OPB + off], 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (K + GEMM_BK - 1) / GEMM_BK;
    const int fr = lane & 15, fc = lane >> 4;
    auto mfma_tile = [&](int cur) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            mfma_bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const mfma_bf16x8*>(&smem[(cur * 2 + 0) * OPB + gemm_lds_off(wm * 64 + i * 16 + fr, fc + 4 * s)]);
                b[i] = *reinterpret_cast<const mfma_bf16x8*>(&smem[(cur * 2 + 1) * OPB + gemm_lds_off(wn * 64 + i * 16 + fr, fc + 4 * s)]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (NBUF == 2) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // LDS-DMA completion is tracked by vmcnt only
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);       // in flight during this tile's MFMAs
            mfma_tile(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tile kt+1 has landed (this wave's part) ...
            __syncthreads();                                // ... and every wave's part after the barrier
        }
    } else {
        for (int kt = 0; kt < nkt; ++kt) {
            stage(0, kt);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            mfma_tile(0);
            __syncthreads();                                // every wave is done with the tile before it is overwritten
        }
    }

    // MODE 2 (head tiles: BN = 128 = one attention head of the q|k|v projection): a tile the epilogue claims (epi.head_tile(n0): q and k heads)
    // is rounded to bf16 into ONE 128 x 128 LDS image (32 KiB; 16-byte chunks XORed with (row >> 2) & 7 so that the four accumulator rows of
    // a store instruction hit different banks) and handed to epi.rows(), which sees whole rows of a head -- q/k RMSNorm + RoPE + cache write run
    // here instead of as a second pass over the projection's output (dec_kernels.h EpiQkHeads).  Other tiles (v heads) take the MODE 0 path.
    if constexpr (MODE == 2) {
        if (epi.head_tile(n0)) {
            unsigned short* T = reinterpret_cast<unsigned short*>(smem);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wm * 64 + i * 16 + fc * 4 + r, col = wn * 64 + j * 16 + fr;
                        T[row * 128 + ((((col >> 3) ^ ((row >> 2) & 7)) << 3) | (col & 7))] = f32_to_bf16(acc[i][j][r]);
                    }
            __syncthreads();
            epi.rows(m0, n0, M, T, tid);
            return;
        }
    }
    // epilogue through LDS so that global accesses are 16 bytes per lane along rows: the wave's 64 x 64 result in
    // HALVES passes of 64 x (64 / HALVES) floats (one pass with 64 KiB of LDS, two 32-column halves with 32 KiB)
    constexpr int HALVES = NBUF == 2 ? 1 : 2, CW = 64 / HALVES;
    float* ct = reinterpret_cast<float*>(smem) + wave * (64 * CW);
    // (Measured and dropped in round 3: all of a half's epilogue requests -- residual rows, bias -- issued before its LDS image is written
    // and applied from registers.  The per-row form below costs a request -> wait -> store round trip per row in the ISA, but the 4-5
    // co-resident workgroups of this form cover it; the batched form's 16-64 extra registers cost more: encoder +0.55 ms per pass.)
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
        if (h) __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4 / HALVES; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) ct[(i * 16 + fc * 4 + r) * CW + jj * 16 + fr] = acc[i][(4 / HALVES) * h + jj][r];
        __syncthreads();
        constexpr int LPR = CW / 4, RPI = 64 / LPR;         // lanes per row, rows per wave instruction
        if (MODE == 0 || MODE == 2) {
            const int er = lane / LPR, ec = (lane % LPR) * 4;
#pragma unroll 4
            for (int it = 0; it < 64 / RPI; ++it) {
                const int row = it * RPI + er;
                const int m = m0 + wm * 64 + row, n = n0 + wn * 64 + h * CW + ec;
                if (m < M && n < N) epi(m, n, lds_read_f4(&ct[row * CW + ec]));
            }
        } else {
            // 32-column blocks = 16 gate + 16 up: a lane takes 4 fused outputs of one block
            constexpr int BPR = CW / 32, LPB = 4, RPS = 64 / (BPR * LPB);
            const int er = lane / (BPR * LPB), blk = (lane / LPB) % BPR, e = (lane % LPB) * 4;
#pragma unroll 4
            for (int it = 0; it < 64 / RPS; ++it) {
                const int row = it * RPS + er;
                const int m = m0 + wm * 64 + row, n = n0 + wn * 64 + h * CW + blk * 32;
                if (m < M && n < N) {
                    const float4 g = lds_read_f4(&ct[row * CW + blk * 32 + e]);
                    const float4 u = lds_read_f4(&ct[row * CW + blk * 32 + 16 + e]);
                    float4 v;
                    v.x = gemm_swiglu(g.x, u.x); v.y = gemm_swiglu(g.y, u.y);
                    v.z = gemm_swiglu(g.z, u.z); v.w = gemm_swiglu(g.w, u.w);
                    epi(m, n / 2 + e, v);
                }
            }
        }
    }
}

template <class ALoad, class Epi, int MODE>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_nt_glds_kernel(ALoad aload, const bf16_t* __restrict__ Wt, long ldw, int M, int N,
                                                                     int K, Epi epi, const bf16_t* __restrict__ zeros, int tm) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * GEMM_BM * 128];
    gemm_nt_128_body<ALoad, Epi, MODE, 2>(smem, aload, Wt, ldw, M, N, K, epi, zeros, tm);
}

template <class ALoad, class Epi, int MODE>
__global__ __launch_bounds__(GEMM_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_nt_glds1_kernel(
    ALoad aload, const bf16_t* __restrict__ Wt, long ldw, int M, int N, int K, Epi epi, const bf16_t* __restrict__ zeros, int tm) {
    __shared__ __attribute__((aligned(1024))) char smem[1 * 2 * GEMM_BM * 128];
    gemm_nt_128_body<ALoad, Epi, MODE, 1>(smem, aload, Wt, ldw, M, N, K, epi, zeros, tm);
}

// Grouped form: gridDim.y independent GEMMs of one shape in ONE launch (the 16 groups of the wav2vec2 positional conv: each is
// only 375 tiles, a third of what the chip holds at once).  The operand / epilogue functors provide for_group(g) -> the functor
// of group g; W advances by w_group_stride elements per group.
template <class ALoad, class Epi, int MODE>
__global__ __launch_bounds__(GEMM_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_nt_glds1_groups_kernel(
    ALoad aload, const bf16_t* __restrict__ Wt, long ldw, long w_group_stride, int M, int N, int K, Epi epi, const bf16_t* __restrict__ zeros) {
    __shared__ __attribute__((aligned(1024))) char smem[1 * 2 * GEMM_BM * 128];
    const int g = blockIdx.y;
    gemm_nt_128_body<ALoad, Epi, MODE, 1>(smem, aload.for_group(g), Wt + (long)g * w_group_stride, ldw, M, N, K, epi.for_group(g), zeros);
}

// 256 bytes of zeros in HBM for masked direct-to-LDS chunks (one per translation unit)
inline const bf16_t* gemm_zero_block() {
    static std::mutex mu;
    static bf16_t* z[64] = {};                 // one block per device (an engine per GPU in one process: qasr_dp_*)
    int dev = 0;
    QASR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) throw std::invalid_argument("device ordinal outside [0, 64)");
    std::lock_guard<std::mutex> lock(mu);
    if (!z[dev]) {
        QASR_HIP(hipMalloc(&z[dev], 256));
        QASR_HIP(hipMemset(z[dev], 0, 256));
    }
    return z[dev];
}
// LDS buffers of the glds kernel for a launch of `grid` tiles.  1: 32 KiB + 124 VGPRs -> 4 workgroups per CU, the loads
// of one overlap the MFMAs of the others (encoder 18.1 -> 15.6 ms, prompt pass 20.8 -> 18.9 ms at 32 x 30 s against the
// double-buffered 64 KiB form at 2 per CU).  That only works when there ARE several workgroups per CU: small launches
// (1 clip: 32-192 tiles; the aligner's 200-tile encoder GEMMs) ran 10-25 % slower with it, so they keep the
// software-double-buffered form.  tuning knob gemm_nbuf = 1|2 forces one form (A/B, parity tests of both).
inline int gemm_nbuf(int grid) {
    const int v = tuning().gemm_nbuf;
    if (v == 1 || v == 2) return v;
    return grid >= 640 ? 1 : 2;             // 2.5 tiles per CU on the 256-CU part (686-tile launches measured faster with 1)
}
// the 256 x 256 form (gemm_p8.h, included at the end of this header)
template <class ALoad, class Epi, int MODE>
__global__ void gemm_nt_p8_kernel(ALoad aload, const bf16_t* __restrict__ Wt, long ldw, int M, int N, int K, Epi epi,
                                  const bf16_t* __restrict__ zeros);
inline bool gemm_use_p8(int M, int N);
inline int gemm_p8_grid(int M, int N);
// an A functor may opt out (static constexpr bool no_p8 = true): one whose per-row state does not fit the 256-register budget
template <class A, class = void>
struct gemm_p8_allowed { static constexpr bool value = true; };
template <class A>
struct gemm_p8_allowed<A, std::void_t<decltype(A::no_p8)>> { static constexpr bool value = !A::no_p8; };

// row panels per block of the 128 x 128 forms' tile order (gemm_nt_128_body): knob gemm_tm, 1 = row-panel-major (rounds 1-3); only where the
// launch has several blocks of that height and at least 32 column tiles (N >= 4096: the prompt pass's q|k|v projection, -7 %; at 24 column
// tiles -- the wav2vec2 q|k|v shape -- the blocked order measured +1.3 % on that model's transformer stage, profiles/r04_ab_gemm_order.txt)
inline int gemm_tile_rows(int M, int N) {
    const int tm = tuning().gemm_tm;
    const int nby = cdiv(M, GEMM_BM), nbx = cdiv(N, GEMM_BN);
    return (tm > 1 && nby >= 2 * tm && nbx >= 32) ? tm : 1;
}

template <class ALoad, class Epi>
// form: -1 = the engine's pick for this shape (tuning knobs gemm_p8 / gemm_nbuf); 0 = 128x128 double-buffered, 1 = 128x128 single LDS
// buffer, 2 = 256x256 ping-pong -- forced by qasr_gemm_probe without touching the process-wide knob table
inline void gemm_nt(const ALoad& a, const bf16_t* Wt, long ldw, int M, int N, int K, const Epi& epi, hipStream_t s, int form = -1) {
    if (M <= 0 || N <= 0) return;
    if constexpr (gemm_p8_allowed<ALoad>::value) if (form == 2 || (form < 0 && gemm_use_p8(M, N))) {
        hipLaunchKernelGGL((gemm_nt_p8_kernel<ALoad, Epi, 0>), dim3(gemm_p8_grid(M, N)), dim3(512), 0, s, a, Wt, ldw, M, N, K, epi,
                           gemm_zero_block());
        return;
    }
    int grid = cdiv(M, GEMM_BM) * cdiv(N, GEMM_BN);
    if (form == 1 || (form != 0 && gemm_nbuf(grid) == 1))
        hipLaunchKernelGGL((gemm_nt_glds1_kernel<ALoad, Epi, 0>), dim3(grid), dim3(GEMM_THREADS), 0, s, a, Wt, ldw, M, N, K, epi,
                           gemm_zero_block(), gemm_tile_rows(M, N));
    else
        hipLaunchKernelGGL((gemm_nt_glds_kernel<ALoad, Epi, 0>), dim3(grid), dim3(GEMM_THREADS), 0, s, a, Wt, ldw, M, N, K, epi,
                           gemm_zero_block(), gemm_tile_rows(M, N));
}

// MODE 2 launch (see gemm_nt_128_body): N is a multiple of 128 and column tile = head; 128 x 128 forms only
template <class ALoad, class Epi>
inline void gemm_nt_headtiles(const ALoad& a, const bf16_t* Wt, long ldw, int M, int N, int K, const Epi& epi, hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    if (N % GEMM_BN != 0) throw std::invalid_argument("head-tile gemm: N must be a multiple of 128");
    const int grid = cdiv(M, GEMM_BM) * cdiv(N, GEMM_BN);
    if (gemm_nbuf(grid) == 1)
        hipLaunchKernelGGL((gemm_nt_glds1_kernel<ALoad, Epi, 2>), dim3(grid), dim3(GEMM_THREADS), 0, s, a, Wt, ldw, M, N, K, epi,
                           gemm_zero_block(), gemm_tile_rows(M, N));
    else
        hipLaunchKernelGGL((gemm_nt_glds_kernel<ALoad, Epi, 2>), dim3(grid), dim3(GEMM_THREADS), 0, s, a, Wt, ldw, M, N, K, epi,
                           gemm_zero_block(), gemm_tile_rows(M, N));
}

template <class ALoad, class Epi>
inline void gemm_nt_groups(const ALoad& a, const bf16_t* Wt, long ldw, long w_group_stride, int groups, int M, int N, int K, const Epi& epi,
                           hipStream_t s) {
    if (M <= 0 || N <= 0 || groups <= 0) return;
    const int grid = cdiv(M, GEMM_BM) * cdiv(N, GEMM_BN);
    hipLaunchKernelGGL((gemm_nt_glds1_groups_kernel<ALoad, Epi, 0>), dim3(grid, groups), dim3(GEMM_THREADS), 0, s, a, Wt, ldw, w_group_stride,
                       M, N, K, epi, gemm_zero_block());
}

template <class ALoad, class Epi>
inline void gemm_nt_swiglu(const ALoad& a, const bf16_t* Wt, long ldw, int M, int N, int K, const Epi& epi, hipStream_t s) {
    if (M <= 0 || N <= 0) return;
    if (N % 32 != 0) throw std::invalid_argument("swiglu gemm: fused width must be a multiple of 32");
    if (gemm_use_p8(M, N)) {
        hipLaunchKernelGGL((gemm_nt_p8_kernel<ALoad, Epi, 1>), dim3(gemm_p8_grid(M, N)), dim3(512), 0, s, a, Wt, ldw, M, N, K, epi,
                           gemm_zero_block());
        return;
    }
    int grid = cdiv(M, GEMM_BM) * cdiv(N, GEMM_BN);
    if (gemm_nbuf(grid) == 1)
        hipLaunchKernelGGL((gemm_nt_glds1_kernel<ALoad, Epi, 1>), dim3(grid), dim3(GEMM_THREADS), 0, s, a, Wt, ldw, M, N, K, epi,
                           gemm_zero_block(), gemm_tile_rows(M, N));
    else
        hipLaunchKernelGGL((gemm_nt_glds_kernel<ALoad, Epi, 1>), dim3(grid), dim3(GEMM_THREADS), 0, s, a, Wt, ldw, M, N, K, epi,
                           gemm_zero_block(), gemm_tile_rows(M, N));
}

// ------------------------------------------------------------------------------------------------
// epilogues: operator()(m, n, float4 acc[n..n+3])   (N is always a multiple of 4)
// Optional split form for kernels that run one workgroup per CU and cannot hide an epilogue's global loads behind other
// workgroups (gemm_p8.h): a nested type Pre, `Pre prefetch(m, n)` = the loads only, `apply(m, n, acc, pre)` = the rest;
// operator() == apply(prefetch).  The caller issues the prefetch of the next output rows before it applies the current ones.
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint2 pack_bf16x4(float4 v) {
    uint2 o;
    o.x = pack_bf16x2(v.x, v.y);
    o.y = pack_bf16x2(v.z, v.w);
    return o;
}
__device__ __forceinline__ float4 unpack_bf16x4(uint2 u) {
    return make_float4(bf16_to_f32((bf16_t)(u.x & 0xffff)), bf16_to_f32((bf16_t)(u.x >> 16)),
                       bf16_to_f32((bf16_t)(u.y & 0xffff)), bf16_to_f32((bf16_t)(u.y >> 16)));
}
__device__ __forceinline__ float4 load_bf16x4(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf16_to_f32((bf16_t)(u.x & 0xffff)), bf16_to_f32((bf16_t)(u.x >> 16)),
                       bf16_to_f32((bf16_t)(u.y & 0xffff)), bf16_to_f32((bf16_t)(u.y >> 16)));
}

// out_bf16[m][n] = act(acc + bias[n]);  ACT: 0 none, 1 exact GELU
template <int ACT>
struct EpiBiasActBf16 {
    bf16_t* out; long ldo; const bf16_t* bias;   // bias may be null
    struct Pre { uint2 b; };
    __device__ __forceinline__ Pre prefetch(int m, int n) const {
        return {bias ? *reinterpret_cast<const uint2*>(bias + n) : make_uint2(0u, 0u)};
    }
    __device__ __forceinline__ void apply(int m, int n, float4 v, const Pre& p) const {
        const float4 b = unpack_bf16x4(p.b);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (ACT == 1) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        *reinterpret_cast<uint2*>(out + (long)m * ldo + n) = pack_bf16x4(v);
    }
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const { apply(m, n, v, prefetch(m, n)); }
};

// out_bf16[m][n] = bf16(acc): the Qwen3 text decoder's Linears carry no bias.  With no load in the epilogue there is nothing for it to
// wait on: behind EpiBiasActBf16's `bias ? load : 0` hipcc keeps an s_waitcnt vmcnt(0) per round, which in the 256^2 form also waits for
// the next tile's first K-tile (in flight under the epilogue by design) and for the round's own stores.
struct EpiStoreBf16 {
    bf16_t* out; long ldo;
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const {
        *reinterpret_cast<uint2*>(out + (long)m * ldo + n) = pack_bf16x4(v);
    }
};

// x_f32[m][n] += acc + bias[n]   (encoder residual stream, f32)
struct EpiResidF32 {
    float* x; long ldx; const bf16_t* bias;
    struct Pre { uint2 b; float4 r; };
    __device__ __forceinline__ Pre prefetch(int m, int n) const {
        return {*reinterpret_cast<const uint2*>(bias + n), *reinterpret_cast<const float4*>(x + (long)m * ldx + n)};
    }
    __device__ __forceinline__ void apply(int m, int n, float4 v, const Pre& p) const {
        const float4 b = unpack_bf16x4(p.b);
        float4 r = p.r;
        r.x += v.x + b.x; r.y += v.y + b.y; r.z += v.z + b.z; r.w += v.w + b.w;
        *reinterpret_cast<float4*>(x + (long)m * ldx + n) = r;
    }
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const { apply(m, n, v, prefetch(m, n)); }
};

}  // namespace qasr

#include "gemm_p8.h"
