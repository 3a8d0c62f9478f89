// dec_chain.hip -- the decode layer's linears (o-proj -> gate|up -> down -> next layer's q|k|v) as ONE persistent launch (declarations: dec_chain.h).
//
// Why: a decode step at 32 rows is 142 dependent launches of 4..15 us; each of the four weight-streaming launches of a layer is the serial chain
//   [boundary 1.5 us] -> activation rows back (1.1 us) -> weight request -> last weight byte (2.7 us) -> MFMA / reduce / epilogue (1 us).
// Here every workgroup requests the weight tiles of ALL its phases when the launch starts (a layer's 31.46 MB = 123 KB per CU: they fit in the
// register file) and the phases hand their 64..192 KB of activations to each other inside the launch, so the weight latency of phases 2..4 is
// off the chain (MI355X_MICROARCH.md, prefetch-credit) and three kernel boundaries per layer are gone.
//
// Grid: 256 workgroups x 512 threads, ONE per CU (131 KB of LDS each forces that), all resident -- the host checks the CU count.
// Units (batch rows in tiles of 16; NB = tiles):
//   O     64 weight-row tiles x NB row groups      -> workgroups [0, 64 NB)              K = 2048, residual epilogue
//   GU    192 gate|up tile pairs, all rows         -> workgroups [0, 192)                K = 1024, RMSNorm prologue, SwiGLU epilogue
//   DOWN  64 tiles x NB row groups                 -> workgroups [128, 256) (NB = 1: [192, 256))   K = 3072, residual epilogue
//   QKV   256 tiles of the next layer, all rows    -> every workgroup                    K = 1024, RMSNorm prologue
// Arithmetic per 16-row group is the five-launch layer's (decode_gemv2_kernel, dec_gemv.hip): same k order over the 8 waves, same cross-wave
// order, same 32-thread norm sums -- the two paths give the same bits (tests/test_gpu_chain.py).
//
// Hand-off (cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md "Valid forms", first table row): a producer's wave 0 stores its
// outputs write-through (sc1), drains (s_waitcnt vmcnt(0)), then ONE lane adds to its shard of the seam's arrival counter (agent-scope atomic;
// 8 shards on lines of their own); a consumer's wave 0 polls every shard with sc1 loads, the workgroup's barrier follows, and EVERY load of
// handed-off bytes is an sc1 load into registers (buffer_load_dwordx4 sc1 / global_load_dwordx2 sc1): no fences.  Counters are zeroed by a
// memset node at the start of each decode step and count up through the step's 28 launches (epoch = layer index).  Every spin is bounded by
// a wall-clock budget; a wait that gives up sets CHAIN_ERR_TIMEOUT in *err and the whole workgroup leaves the kernel (the host reports it).
#include "dec_chain_dev.h"
#include <mutex>
#include <map>

namespace qasr {
namespace {

using namespace chain_dev;

__device__ __forceinline__ uint2 resid_add(uint2 r, const f32x4& acc) {
    float4 v = unpack_bf16x4(r);
    v.x += bf16_round(acc[0]); v.y += bf16_round(acc[1]); v.z += bf16_round(acc[2]); v.w += bf16_round(acc[3]);
    return pack_bf16x4(v);
}

// ST: diagnostic instantiation (qasr_kernel_probe 6): thread 0 of every workgroup stamps the 100 MHz clock into a.dbg[wg * 32 + i]:
//   0 entry | O: 1 staged 2 summed 3 signalled | GU: 4 wait over 5 rows in 6 staged 7 summed 8 signalled | DOWN: 9..13 | QKV: 14..18 (18 = stored)
// PF: weight requests staged -- a workgroup with an O unit asks for its later phases' tiles once the O sums are in, the others wait PF_TICKS
// before their first request (all at once, the 31 MB of requests delay the first phase's activation rows: staged at 4.7 us instead of 1.1,
// profiles/r04_stamps_chain.txt); !PF: everything at kernel entry.
template <int PH, int NB, bool PF, bool ST>
__global__ __launch_bounds__(CT, 2) void decode_chain_kernel(DecChainArgs a) {
    constexpr bool NTW = false;
    constexpr unsigned long long PF_TICKS = 250;                       // 2.5 us of the 100 MHz clock
    const unsigned long long t_entry = PF ? wall_clock64() : 0ull;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fc = lane >> 4;
    const int srow = tid >> 5, scol = tid & 31;
    const int wg = blockIdx.x;
    constexpr int DOWN0 = NB == 1 ? 192 : 128;
    constexpr bool P_GU = (PH & CHAIN_GU) != 0, P_DOWN = (PH & CHAIN_DOWN) != 0, P_QKV = (PH & CHAIN_QKV) != 0;
    const bool has_o = wg < 64 * NB;
    const bool has_gu = P_GU && wg < 192;
    const bool has_down = P_DOWN && wg >= DOWN0 && wg < DOWN0 + 64 * NB;
    char* s_x = dsm + L_X;
    float* s_red = reinterpret_cast<float*>(dsm + L_RED);
    int* s_flag = reinterpret_cast<int*>(dsm + L_FLAG);
    const int B = a.B;
    unsigned long long* st = ST ? a.dbg + (long)wg * 32 : nullptr;
#define CH_STAMP(i) do { if constexpr (ST) { if (tid == 0) st[i] = wall_clock64(); } } while (0)
#define CH_ROWS_IN(i) do { if constexpr (ST) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (tid == 0) st[i] = wall_clock64(); } } while (0)
    CH_STAMP(0);

    // ---- requests: norm weights (oldest: they come back first), the first phase's activation rows, then EVERY phase's weight tiles ----
    // one unconditional load per thread (threads 256.. read duplicates): a load under a divergent branch is waited for at the join
    const uint4 nw = reinterpret_cast<const uint4*>((tid & 128) && P_QKV ? a.ln1n : a.ln2)[tid & 127];
    uint4 wAD[1][12], wB[2][4], wC[1][4];
    if (has_o) {
        // ---- phase O on this workgroup: x[rows][16 cols] += attn[rows] . Wo[tile]^T ------------------------------------------------
        const int ot = wg & 63, r0 = (wg >> 6) * 16;
        uint4 xo[1][8];
        {
            const bf16_t* xp = a.attn + (long)(r0 + srow < B ? r0 + srow : 0) * CH_NQ + scol * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) xo[0][i] = *reinterpret_cast<const uint4*>(xp + i * 32 * 8);
        }
        bf16_t* xout = a.x + (long)(r0 + fr < B ? r0 + fr : 0) * CH_H + ot * 16 + fc * 4;
        uint2 rsd = make_uint2(0, 0);
        if (wave == 0) rsd = *reinterpret_cast<const uint2*>(xout);
        uint4 wO[1][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wO[0][i] = ld_weight<NTW>(a.wo_p + ((long)ot * (CH_NQ / 32) + wave + CWAVES * i) * 512 + lane * 8);
        auto later_tiles = [&]() {
            if constexpr (P_GU) {            // wg < 128 < 192: every O workgroup has a gate|up unit
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) wB[t][i] = ld_weight<NTW>(a.wgu_p + ((long)(2 * wg + t) * (CH_H / 32) + wave + CWAVES * i) * 512 + lane * 8);
            }
            if constexpr (P_QKV) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wC[0][i] = ld_weight<NTW>(a.wqkv_p + ((long)wg * (CH_H / 32) + wave + CWAVES * i) * 512 + lane * 8);
            }
        };
        if constexpr (!PF) later_tiles();
        if (r0 + srow >= B) {
#pragma unroll
            for (int i = 0; i < 8; ++i) xo[0][i] = make_uint4(0, 0, 0, 0);
        }
        f32x4 acc[1][1];
        chain_mma<1, 1, 8, false, ST>(wO, xo, nullptr, 0.f, s_x, s_red, acc, st + 1);
        if (wave == 0) {
            if (r0 + fr < B) st8_sc1(xout, resid_add(rsd, acc[0][0]));
            if (a.proto) seam_signal_r(a.ctr, 0); else seam_signal(a.ctr, 0, wg);
            CH_STAMP(3);
        }
        if constexpr (PF) later_tiles();
    } else {
        if constexpr (PF) { while (wall_clock64() - t_entry < PF_TICKS) __builtin_amdgcn_s_sleep(8); }
        if (has_down) {
            const int dt = (wg - DOWN0) & 63;
#pragma unroll
            for (int i = 0; i < 12; ++i) wAD[0][i] = ld_weight<NTW>(a.wdown_p + ((long)dt * (CH_I / 32) + wave + CWAVES * i) * 512 + lane * 8);
        }
        if (has_gu) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) wB[t][i] = ld_weight<NTW>(a.wgu_p + ((long)(2 * wg + t) * (CH_H / 32) + wave + CWAVES * i) * 512 + lane * 8);
        }
        if constexpr (P_QKV) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wC[0][i] = ld_weight<NTW>(a.wqkv_p + ((long)wg * (CH_H / 32) + wave + CWAVES * i) * 512 + lane * 8);
        }
    }
    if (tid < 256) reinterpret_cast<uint4*>(dsm + L_NORM)[tid] = nw;        // read only behind a seam_wait barrier

    // ---- phase GU: act[rows][16 cols] = swiglu(rmsnorm(x) . Wg^T, rmsnorm(x) . Wu^T) -------------------------------------------------
    if (has_gu) {
        if (!(a.proto ? seam_wait_r(a.ctr, 0, (a.epoch + 1) * (64 * NB), a.err, s_flag) : seam_wait(a.ctr, 0, (a.epoch + 1) * (8 * NB), a.err, s_flag))) return;
        CH_STAMP(4);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.x, 0, 32 * CH_H * 2, 0x00020000);
        uint4 xr[NB][4];
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int row = p * 16 + srow;
            const int off = ((row < B ? row : 0) * CH_H + scol * 8) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[p][i] = ld16_sc1(rs, off + i * 32 * 16);
        }
#pragma unroll
        for (int p = 0; p < NB; ++p)
            if (p * 16 + srow >= B) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[p][i] = make_uint4(0, 0, 0, 0);
            }
        CH_ROWS_IN(5);
        f32x4 acc[2][NB];
        chain_mma<2, NB, 4, true, ST>(wB, xr, dsm + L_NORM, a.eps, s_x, s_red, acc, st + 6);
        if (wave == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int row = b * 16 + fr;
                if (row < B) {
                    float4 v;
                    v.x = swiglu_bf16(acc[0][b][0], acc[1][b][0]);
                    v.y = swiglu_bf16(acc[0][b][1], acc[1][b][1]);
                    v.z = swiglu_bf16(acc[0][b][2], acc[1][b][2]);
                    v.w = swiglu_bf16(acc[0][b][3], acc[1][b][3]);
                    st8_sc1(a.act + (long)row * CH_I + wg * 16 + fc * 4, pack_bf16x4(v));
                }
            }
            if (a.proto) seam_signal_r(a.ctr, 1); else seam_signal(a.ctr, 1, wg);
            CH_STAMP(8);
        }
    }

    // ---- phase DOWN: x[rows][16 cols] += act[rows] . Wd[tile]^T --------------------------------------------------------------------
    if (has_down) {
        if (!(a.proto ? seam_wait_r(a.ctr, 1, (a.epoch + 1) * 192, a.err, s_flag) : seam_wait(a.ctr, 1, (a.epoch + 1) * 24, a.err, s_flag))) return;
        CH_STAMP(9);
        const int du = wg - DOWN0, dt = du & 63, r0 = (du >> 6) * 16;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.act, 0, 32 * CH_I * 2, 0x00020000);
        uint4 xr[1][12];
        {
            const int off = ((r0 + srow < B ? r0 + srow : 0) * CH_I + scol * 8) * 2;
#pragma unroll
            for (int i = 0; i < 12; ++i) xr[0][i] = ld16_sc1(rs, off + i * 32 * 16);
        }
        bf16_t* xout = a.x + (long)(r0 + fr < B ? r0 + fr : 0) * CH_H + dt * 16 + fc * 4;
        uint2 rsd = make_uint2(0, 0);
        if (wave == 0) rsd = ld8_sc1(xout);              // written by phase O of this launch (another workgroup)
        if (r0 + srow >= B) {
#pragma unroll
            for (int i = 0; i < 12; ++i) xr[0][i] = make_uint4(0, 0, 0, 0);
        }
        CH_ROWS_IN(10);
        f32x4 acc[1][1];
        chain_mma<1, 1, 12, false, ST>(wAD, xr, nullptr, 0.f, s_x, s_red, acc, st + 11);
        if (wave == 0) {
            if (r0 + fr < B) st8_sc1(xout, resid_add(rsd, acc[0][0]));
            if (a.proto) seam_signal_r(a.ctr, 2); else seam_signal(a.ctr, 2, du);
            CH_STAMP(13);
        }
    }

    // ---- phase QKV: the next layer's q|k|v[rows][16 cols] = rmsnorm(x) . Wqkv[tile]^T (read by the next launch: plain stores) --------
    if constexpr (P_QKV) {
        if (!(a.proto ? seam_wait_r(a.ctr, 2, (a.epoch + 1) * (64 * NB), a.err, s_flag) : seam_wait(a.ctr, 2, (a.epoch + 1) * (8 * NB), a.err, s_flag))) return;
        CH_STAMP(14);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.x, 0, 32 * CH_H * 2, 0x00020000);
        uint4 xr[NB][4];
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const int row = p * 16 + srow;
            const int off = ((row < B ? row : 0) * CH_H + scol * 8) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) xr[p][i] = ld16_sc1(rs, off + i * 32 * 16);
        }
#pragma unroll
        for (int p = 0; p < NB; ++p)
            if (p * 16 + srow >= B) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[p][i] = make_uint4(0, 0, 0, 0);
            }
        CH_ROWS_IN(15);
        f32x4 acc[1][NB];
        chain_mma<1, NB, 4, true, ST>(wC, xr, dsm + L_NORM + 2048, a.eps, s_x, s_red, acc, st + 16);
        if (wave == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int row = b * 16 + fr;
                if (row < B)
                    *reinterpret_cast<uint2*>(a.qkv + (long)row * CH_NQKV + wg * 16 + fc * 4) =
                        pack_bf16x4(make_float4(acc[0][b][0], acc[0][b][1], acc[0][b][2], acc[0][b][3]));
            }
            CH_STAMP(18);
        }
    }
#undef CH_STAMP
#undef CH_ROWS_IN
}

int device_cus() {
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    QASR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    hipDeviceProp_t p;
    QASR_HIP(hipGetDeviceProperties(&p, dev));
    cus[dev] = p.multiProcessorCount;
    return p.multiProcessorCount;
}

template <int PH, int NB, bool PF>
void chain_go(const DecChainArgs& a, hipStream_t s) {
    auto go = [&](auto kern) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(kern), L_TOTAL);
        hipLaunchKernelGGL(kern, dim3(CH_GRID), dim3(CT), L_TOTAL, s, a);
    };
    if (a.dbg) go(decode_chain_kernel<PH, NB, PF, true>);        // diagnostic launch
    else go(decode_chain_kernel<PH, NB, PF, false>);
}

template <int NB, bool PF>
void chain_ph(int phases, const DecChainArgs& a, hipStream_t s) {
    switch (phases) {
        case CHAIN_O | CHAIN_GU: chain_go<CHAIN_O | CHAIN_GU, NB, PF>(a, s); break;
        case CHAIN_O | CHAIN_GU | CHAIN_DOWN: chain_go<CHAIN_O | CHAIN_GU | CHAIN_DOWN, NB, PF>(a, s); break;
        case CHAIN_O | CHAIN_GU | CHAIN_DOWN | CHAIN_QKV: chain_go<CHAIN_O | CHAIN_GU | CHAIN_DOWN | CHAIN_QKV, NB, PF>(a, s); break;
        default: throw std::invalid_argument("decode chain: unsupported phase set");
    }
}

}  // namespace

bool decode_chain_supported(int H, int nq, int I, int nqkv, int B) {
    return H == CH_H && nq == CH_NQ && I == CH_I && nqkv == CH_NQKV && B >= 1 && B <= 32 && device_cus() >= CH_GRID;
}

void decode_chain_launch(int phases, const DecChainArgs& a, hipStream_t s) {
    if (a.B < 1 || a.B > 32) throw std::invalid_argument("decode chain: 1..32 batch rows");
    DecChainArgs b = a;
    b.proto = tuning().chain_proto;
    const bool pf = tuning().chain_pf != 0;
    if (a.B <= 16) { if (pf) chain_ph<1, true>(phases, b, s); else chain_ph<1, false>(phases, b, s); }
    else { if (pf) chain_ph<2, true>(phases, b, s); else chain_ph<2, false>(phases, b, s); }
}

// zero the arrival counters and bump the step sequence word (what greedy_finalize_kernel does in front of a greedy step)
__global__ __launch_bounds__(256) void chain_reset_kernel(unsigned* ctr) {
    for (int i = threadIdx.x; i < (int)(CHAIN_CTR_BYTES / sizeof(unsigned)); i += 256) ctr[i] = 0u;
    if (threadIdx.x == 0) ctr[CHAIN_SEQ_WORD] += 1u;
}
void decode_chain_reset(unsigned* ctr, hipStream_t s) { hipLaunchKernelGGL(chain_reset_kernel, dim3(1), dim3(256), 0, s, ctr); }

}  // namespace qasr
