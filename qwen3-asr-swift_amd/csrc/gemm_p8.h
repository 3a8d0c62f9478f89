// gemm_p8.h -- the 256 x 256 x 64 form of gemm_nt for launches of many tiles: 8 waves (2 x 4, each 128 x 64), one workgroup
// per CU, operands by direct-to-LDS loads into a 128 KiB ring that stay in flight across raw barriers.
//
// Schedule (MI355X guide, "The 256^2 8-phase template": this is a reconstruction from its description, the choreography
// below is this file's own and is what has to be checked when it is edited):
//   * A K-tile (64 wide) is staged as four 16 KiB units of 128 rows: A-lo / A-hi = the first / second 64 rows of each wave
//     row-group's 128, B-lo / B-hi = the first / second 32 columns of each wave column-group's 64.  Ring = 2 K-tiles x 4 units.
//   * A K-tile is consumed in four phases, one 64 x 32 quadrant of the wave's output each (16 MFMAs 16x16x32):
//       P0 reads A-lo (8 fragments), quadrant (lo, lo)                 P1 reads B-hi (4), quadrant (lo, hi)
//       P2 reads A-hi (8), quadrant (hi, hi)                           P3 reads B-lo of the NEXT K-tile (4), quadrant (hi, lo)
//     -- B-lo of a tile is read one phase before the tile starts (into the B-hi registers, handed over by 16 moves), which evens the LDS read bursts out to 8 / 4 / 8 / 4 per phase (12 / 4 / 8 / 0 before).
//     A unit's last LDS read is P0 (A-lo), P1 (B-hi), P2 (A-hi) of its tile or P3 of the tile before (B-lo).
//   * Every phase issues ONE unit (2 direct-to-LDS instructions per wave), always >= 2 phases after the last read of the
//     slot it overwrites and >= 5 phases before its first read:
//       P0(t): B-hi(t+1)   P1(t): A-hi(t+1)   P2(t): A-lo(t+2)   P3(t): B-lo(t+2)
//     and then waits vmcnt(6): everything but the three youngest units has landed, i.e. every unit the NEXT phase reads
//     (B-lo(t+1), issued in P3(t-1), is read in P3(t); three units in flight run as fast as four: profiles/r02_ab_p8_*).
//   * Phase = [LDS reads, unit issue, vmcnt(6)] s_barrier [lgkmcnt(0), 16 MFMAs] s_barrier.  The wave row-group 1 runs one
//     barrier behind group 0 (it takes one extra barrier before the loop, group 0 one after it): on every SIMD one wave is in
//     its MFMA segment while its partner reads LDS.  RAW: a unit is read in phase p + 1 after the vmcnt(6) of phase p of
//     BOTH groups (intervals 2p and 2p + 1) and the barrier that ends interval 2p + 1.  WAR: a unit issued in phase p (interval
//     2p at the earliest) overwrites data last read in phase <= p - 2, whose reads retired (lgkmcnt(0)) by interval 2p - 2.
//   * K-tiles past the end stage zeros (the operand functors return no address for k >= K): the wait counts stay uniform.
//   * Workgroups are persistent (one per CU): when a tile's K loop ends, the first K-tile of the workgroup's NEXT tile is issued
//     into the parity-0 half of the ring before the epilogue, which stages the accumulators through the parity-1 half
//     (wave-private 32 x 32 images, rows out as float4): the next tile's first-load latency hides under the epilogue.
#pragma once
#include "gemm.h"

namespace qasr {

constexpr int P8_BM = 256, P8_BN = 256, P8_THREADS = 512, P8_UNIT = 16384, P8_TM = 4;


template <class ALoad, class Epi, int MODE>
__global__ __launch_bounds__(P8_THREADS) void gemm_nt_p8_kernel(ALoad aload, const bf16_t* __restrict__ Wt, long ldw, int M, int N,
                                                                int K, Epi epi, const bf16_t* __restrict__ zeros) {
    __shared__ __attribute__((aligned(1024))) char smem[8 * P8_UNIT];       // [tile parity][A-lo, A-hi, B-lo, B-hi][128 rows x 128 B]
    constexpr int U_ALO = 0, U_AHI = 1, U_BLO = 2, U_BHI = 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // Persistent workgroups (one per CU).  Workgroup ids go round the 8 XCDs, so id & 7 names the XCD; XCD x owns the x-th
    // eighth of the tile list (order: tile_rows below) and its workgroups walk that range round-robin: the tiles in flight on
    // one XCD share a few A row panels and W column slices, each fetched into that L2 once.
    const int nbx = (N + P8_BN - 1) / P8_BN, nby = (M + P8_BM - 1) / P8_BM;
    const int ncls = gridDim.x < 8 ? gridDim.x : 8;         // a launch of fewer than 8 workgroups: one tile range each
    const int xcd = blockIdx.x % ncls, wgs_per_xcd = (gridDim.x - xcd + ncls - 1) / ncls;
    const long n_tiles = (long)nbx * nby;
    const int t_end = (int)(n_tiles * (xcd + 1) / ncls);
    int lt = (int)(n_tiles * xcd / ncls) + blockIdx.x / ncls;   // this workgroup's tile
    if (lt >= t_end) return;

    // staging: wave w, instruction i covers unit rows (2 w + i) * 8 .. + 7; lane -> (row + lane / 8, LDS slot lane % 8)
    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ srow;                   // source chunk = slot ^ (row & 7)
    typename ALoad::Row alo[2], ahi[2];
    const bf16_t *blo[2], *bhi[2];
    int m0, n0;
    auto tile_rows = [&](int t) {
        // tile list order: blocks of P8_TM row panels, column tile by column tile inside a block, panel fastest.  The 32
        // workgroups of an XCD then run 4 panels x 8 column tiles at a time: 12 distinct operand streams behind their 32 tiles
        // whatever N is (plain row-major order made it 1 panel x 32 column tiles = 33 streams at N = 8192, the 7B FFN-up shape).
        const int blk = t / (P8_TM * nbx), i = t - blk * (P8_TM * nbx);
        const int rows_in_blk = nby - blk * P8_TM < P8_TM ? nby - blk * P8_TM : P8_TM;
        m0 = (blk * P8_TM + i % rows_in_blk) * P8_BM;
        n0 = (i / rows_in_blk) * P8_BN;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int u = (wave * 2 + i) * 8 + srow;
            const int am = m0 + (u >> 6) * 128 + (u & 63);
            alo[i] = aload.row_init(am);
            ahi[i] = aload.row_init(am + 64);
            const int bn = n0 + (u >> 5) * 64 + (u & 31);
            blo[i] = bn < N ? Wt + (long)bn * ldw : nullptr;
            bhi[i] = bn + 32 < N ? Wt + (long)(bn + 32) * ldw : nullptr;
        }
    };
    auto stage_a = [&](int unit, int kt, const typename ALoad::Row* rows) {
        const typename ALoad::KT akt = aload.ktile(kt * GEMM_BK, schunk * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* p = aload.addr_kt(rows[i], akt);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(p ? p : zeros), (lds_ptr_t)&smem[((kt & 1) * 4 + unit) * P8_UNIT + (wave * 2 + i) * 1024], 16, 0, 0);
        }
    };
    auto stage_b = [&](int unit, int kt, const bf16_t* const* rows) {
        const int k = kt * GEMM_BK + schunk * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* p = (rows[i] && k < K) ? rows[i] + k : nullptr;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(p ? p : zeros), (lds_ptr_t)&smem[((kt & 1) * 4 + unit) * P8_UNIT + (wave * 2 + i) * 1024], 16, 0, 0);
        }
    };
    auto stage_first = [&]() {                              // K-tile 0 of a tile: the parity-0 half of the ring
        stage_a(U_ALO, 0, alo);
        stage_b(U_BLO, 0, blo);
        stage_b(U_BHI, 0, bhi);
        stage_a(U_AHI, 0, ahi);
    };

    const int nkt = (K + GEMM_BK - 1) / GEMM_BK;
    const int fr = lane & 15, fc = lane >> 4;
    // fragment byte offsets inside a unit (K-step s adds chunk 4 s: the XOR swizzle keeps bit 2 of the chunk, so + 64 B)
    int a_off[4], b_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[i] = gemm_lds_off(wr * 64 + i * 16 + fr, fc);
#pragma unroll
    for (int j = 0; j < 2; ++j) b_off[j] = gemm_lds_off(wc * 32 + j * 16 + fr, fc);

    tile_rows(lt);
    stage_first();
  for (;;) {
    const int cm0 = m0, cn0 = n0;                           // this tile's origin (tile_rows moves on before the epilogue)
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage_a(U_ALO, 1, alo);
    stage_b(U_BLO, 1, blo);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    mfma_bf16x8 af[4][2], bl[2][2], bh[2][2];
#define P8_READ_A(TILE, UNIT)                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int s = 0; s < 2; ++s)                               \
        af[i][s] = *reinterpret_cast<const mfma_bf16x8*>(&(TILE)[(UNIT) * P8_UNIT + (a_off[i] ^ (s << 6))]);
#define P8_READ_B(DST, TILE, UNIT)                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int s = 0; s < 2; ++s)                               \
        DST[j][s] = *reinterpret_cast<const mfma_bf16x8*>(&(TILE)[(UNIT) * P8_UNIT + (b_off[j] ^ (s << 6))]);
#define P8_SYNC_IN()                                                                                                         \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                                         \
    __builtin_amdgcn_s_barrier();                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                       \
    __builtin_amdgcn_s_setprio(1);
#define P8_MFMA(BF, MH, NH)                                                                                                  \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll")             \
        for (int j = 0; j < 2; ++j) acc[(MH) * 4 + i][(NH) * 2 + j] =                                                        \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][s], BF[j][s], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0);
#define P8_SYNC_OUT()                                                                                                        \
    __builtin_amdgcn_s_setprio(0);                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                                       \
    __builtin_amdgcn_s_barrier();
    // One K-tile = four phases.  bl holds B-lo of this tile on entry (read during the previous tile's last phase, or before the
    // loop); bh receives B-hi in P1 and, in P3 -- whose MFMAs use bl -- B-lo of the NEXT tile from the other half of the ring:
    // LDS reads per phase 8 / 4 / 8 / 4.  16 register moves per tile hand it over (two tile bodies with swapped register roles
    // spilled).
    {
        const char* tile0 = &smem[0];
        P8_READ_B(bl, tile0, U_BLO)                         // B-lo of K-tile 0 (landed: the wait + barrier above)
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();              // group 1 runs one barrier behind group 0 from here on
    for (int kt = 0; kt < nkt; ++kt) {
        const char* tile = &smem[(kt & 1) * 4 * P8_UNIT];
        const char* tnext = &smem[((kt + 1) & 1) * 4 * P8_UNIT];
        // P0
        P8_READ_A(tile, U_ALO)
        stage_b(U_BHI, kt + 1, bhi);
        P8_SYNC_IN()
        P8_MFMA(bl, 0, 0)
        P8_SYNC_OUT()
        // P1
        P8_READ_B(bh, tile, U_BHI)
        stage_a(U_AHI, kt + 1, ahi);
        P8_SYNC_IN()
        P8_MFMA(bh, 0, 1)
        P8_SYNC_OUT()
        // P2
        P8_READ_A(tile, U_AHI)
        stage_a(U_ALO, kt + 2, alo);
        P8_SYNC_IN()
        P8_MFMA(bh, 1, 1)
        P8_SYNC_OUT()
        // P3
        P8_READ_B(bh, tnext, U_BLO)
        stage_b(U_BLO, kt + 2, blo);
        P8_SYNC_IN()
        P8_MFMA(bl, 1, 0)
        P8_SYNC_OUT()
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) bl[j][s2] = bh[j][s2];
    }
#undef P8_READ_A
#undef P8_READ_B
#undef P8_SYNC_IN
#undef P8_MFMA
#undef P8_SYNC_OUT
    if (wr == 0) __builtin_amdgcn_s_barrier();              // the groups meet again
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the zero-filled units of K-tiles past the end
    __builtin_amdgcn_s_barrier();

    // next tile: its first K-tile goes to the parity-0 half of the ring now, in flight under this tile's epilogue
    lt += wgs_per_xcd;
    const bool more = lt < t_end;
    // epilogue: eight rounds (64-row half, 32-column half, 32-row half) through this wave's 32 x 36-float image in the
    // parity-1 half of the ring
    constexpr int LDC = 36;
    float* ct = reinterpret_cast<float*>(smem + 4 * P8_UNIT) + wave * (32 * LDC);
    constexpr bool PRE = MODE == 0 && epi_has_pre<Epi>::value;
    const int er0 = lane >> 3, ec0 = (lane & 7) * 4;        // MODE 0 read-back map: row it * 8 + er0, columns ec0 .. + 3
    auto round_base = [&](int rd, int& mb, int& nb) {       // round rd = (mh, nh, rh)
        mb = cm0 + wr * 128 + (rd >> 2) * 64 + (rd & 1) * 32;
        nb = cn0 + wc * 64 + ((rd >> 1) & 1) * 32;
    };
    // the epilogue's own global loads (bias, residual rows) of round rd + 1 are requested before round rd is applied
    auto prefetch_round = [&](int rd, auto& dst) {
        if constexpr (PRE) {
            int mb, nb;
            round_base(rd, mb, nb);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int m = mb + it * 8 + er0, n = nb + ec0;
                if (m < M && n < N) dst[it] = epi.prefetch(m, n);
            }
        }
    };
    struct NoPre {};
    using PreT = typename std::conditional<PRE, typename epi_pre_type<Epi, PRE>::type, NoPre>::type;
    PreT pre[4] = {}, nxt[4] = {};
    if (more) {
        tile_rows(lt);
        stage_first();
    }
    prefetch_round(0, pre);
#pragma unroll
    for (int rd = 0; rd < 8; ++rd) {
        const int mh = rd >> 2, nh = (rd >> 1) & 1, rh = rd & 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();                    // the previous round's reads are done
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ct[(i * 16 + fc * 4 + r) * LDC + jj * 16 + fr] = acc[mh * 4 + rh * 2 + i][nh * 2 + jj][r];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int mb, nb;
        round_base(rd, mb, nb);
        if (MODE == 0) {
            float4 v[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) v[it] = lds_read_f4(&ct[(it * 8 + er0) * LDC + ec0]);
            if (rd < 7) prefetch_round(rd + 1, nxt);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int m = mb + it * 8 + er0, n = nb + ec0;
                if (m < M && n < N) {
                    if constexpr (PRE) epi.apply(m, n, v[it], pre[it]);
                    else epi(m, n, v[it]);
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) pre[it] = nxt[it];
        } else {
            const int er = lane >> 2, e = (lane & 3) * 4;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 16 + er;
                if (mb + row < M && nb < N) {
                    const float4 g = lds_read_f4(&ct[row * LDC + e]);
                    const float4 u = lds_read_f4(&ct[row * LDC + 16 + e]);
                    float4 v;
                    v.x = gemm_swiglu(g.x, u.x); v.y = gemm_swiglu(g.y, u.y);
                    v.z = gemm_swiglu(g.z, u.z); v.w = gemm_swiglu(g.w, u.w);
                    epi(mb + row, nb / 2 + e, v);
                }
            }
        }
    }
    if (!more) break;
    __builtin_amdgcn_s_barrier();                           // every wave is done with its epilogue image: K-tile 1 may land there
  }
}

// The 256^2 form pays when the launch fills the chip several times over with little tail: >= 3 rounds of 256 tiles at >= 85 %
// of the last round used, or >= 8 rounds.  tuning knob gemm_p8: 0 never | 1 by this rule | 2 whenever the shape allows.
inline bool gemm_use_p8(int M, int N) {
    const int v = tuning().gemm_p8;
    if (v == 0) return false;
    const long tiles = (long)cdiv(M, P8_BM) * cdiv(N, P8_BN);
    if (v == 2) return true;
    const long rounds = (tiles + 255) / 256;
    return rounds >= 8 || (rounds >= 3 && tiles * 100 >= rounds * 256 * 85);
}

// persistent grid: one workgroup per CU (128 KiB of LDS each), fewer when the launch has fewer tiles
inline int gemm_p8_grid(int M, int N) {
    // CU count of the CALLING thread's device, cached per device under a lock: one process may drive an engine per GPU from several threads
    static std::mutex mu;
    static int cus_of[64] = {0};
    int dev = 0;
    QASR_HIP(hipGetDevice(&dev));
    int cus;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (dev < 0 || dev >= 64) throw std::invalid_argument("gemm_p8_grid: device ordinal out of range");
        if (!cus_of[dev]) {
            hipDeviceProp_t prop;
            QASR_HIP(hipGetDeviceProperties(&prop, dev));
            cus_of[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        }
        cus = cus_of[dev];
    }
    const long tiles = (long)cdiv(M, P8_BM) * cdiv(N, P8_BN);
    return (int)(tiles < cus ? tiles : cus);
}

}  // namespace qasr
