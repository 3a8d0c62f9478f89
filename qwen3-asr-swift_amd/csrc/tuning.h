// tuning.h -- the one table of A/B and diagnostic knobs of libqasr.
//
// Every default is the measured winner; every other value is a kept alternative that the parity tests also pass
// (DESIGN.md section 5 lists the measurements).  Values come from, in order: qasr_set_tuning() (C ABI, tests and
// bench A/B legs), the environment variable QASR_<KEY IN UPPER CASE> read once at first use, the default below.
// A change bumps `epoch`, which invalidates captured decode graphs (engine.h: graph_key_).
#pragma once

namespace qasr {

struct Tuning {
    int gemv_xbar = 4;       // decode GEMVs (16-row forms without the early weight stream): weight requests behind a bare barrier instead of behind the rows' return: 0 | 1 residual | 2 norm | 3 all | 4 by batch
    int gemv_splitb = 2;     // decode GEMV batch row groups on gridDim.y: 0 none | 1 residual GEMVs | 2 all
    int gemv_w1024 = 8;      // waves per workgroup for K = 1024 (8 x 4 k-steps | 4 x 8)
    int gemv_partial = 1;    // fewer than 16 batch rows: norm GEMVs skip the normalisation of rows past the end (own instantiation) | 0 off
    int gemv_earlyw = 1;     // at most 8 batch rows: the weight stream requested without waiting for the activation rows | 0 after them (as at 16+ rows)
    int gemv_nt = 0;         // decode GEMV weight fragments: 1 non-temporal loads (template parameter) | 0 default cache policy
    int lmh_nt = 1;          // LM head weight stream: 1 non-temporal loads | 0 default
    int gemv_wide = 1;       // K = 6144 (1.7B down-projection): 1 two-phase LDS image, weights in registers (dec_gemv_wide.hip) | 0 generic kernel
    int chain = 0;           // decode layer's linears as one persistent launch with in-launch hand-offs (dec_chain.hip): 0 five launches per layer |
                             // 1 o-proj -> gate|up | 2 ... -> down | 3 ... -> the next layer's q|k|v (two launches per layer: attention + chain)
    int qa = 1;              // q|k|v projection + decode attention of a layer as one launch, K / V requested before the projection (dec_qa.hip): 1 | 0 two launches
    int qa_early = 5;        // dec_qa: when the K half of each wave's first chunk is requested: 0 with the rest (after the projection's sums) | 1 behind the
                             // weight tile | 2 like 1 on waves 4..7 only | 3 once the wave's activation rows are staged | 4 in front of the weight tile |
                             // 5 (default) 3 above 16 batch rows, 4 up to 16
    int qa_xbar = 2;         // dec_qa: a bare barrier between the row requests and the weight requests: 0 | 1 | 2 = above 16 rows
    int qa_split = 1;        // dec_qa: 1 up to 8 batch rows an attention unit's context goes over 8 / 4 workgroups (partials handed over as granules) | 2 also 9..16 rows over 2 | 0 one workgroup per unit
    int qa_gran = 1;         // dec_qa hand-off: 1 data-tagged 8-byte granules (the data is the flag) | 0 write-through rows + arrival counter + sc1 row loads
    int qa_gate = 2;         // dec_qa: 1 waves 1..7 hold their remaining K / V requests until wave 0 has sent the projection off | 0 as soon as the sums are in | 2 = 1 with granules, 0 with the counter form
    int chain_fault = 0;     // TEST ONLY: 1 = one workgroup of the fused q|k|v + attention launch never signals, so the bounded waits give up and the step
                             // ends with QASR_ERR_HIP (tests/test_gpu_chain.py::test_lost_arrival_ends_in_an_error_not_a_hang)
    int chain_proto = 0;     // chain arrival counters: 0 sharded (add to one of 8, poll all 8) | 1 replicated (add to all 8, poll one)
    int chain_pf = 0;        // chain weight requests: 0 every phase's tiles at kernel entry | 1 staged (first phase first)
    int da_unr = 2;          // stand-alone decode attention, 8 waves: chunks in flight per wave 2 | 1 (~130 registers: shares a CU; for engines that share a GPU)
    int da_waves = 8;        // decode attention waves per workgroup (8 with two chunks in flight | 16 with one)
    int da_spec = 3;         // K/V requests issued before ctx_len is known: 0 none | 1 each wave's first chunk (no byte past the context at 256+ keys) |
                             // 2 every chunk of the first round (1.26 x the algorithmic bytes at 32 x 30 s) | 3 = 2 up to 8 batch rows, 1 above
                             // (round 3: decode -2 % at 1 / 8 clips, -0.7 % at 32)
    int da_earlyq = 0;       // 1: the token's own q / k / v rows requested ahead of the K / V stream (loads return in order; +-0, measured) | 0 after it
    int pa_form = 2;         // prompt attention: 2 transposed-score form | 1 first form (a third, 32x32x16 form was measured and dropped:
                             // profiles/r02_ab_prompt_attention_form3.txt)
    int pa_vfrag = 1;        // prompt attention V operand: 1 the decode sweep's fragment-major image (one lane-linear 16-byte read per fragment) | 0 V^T rows
    int pa_order = 1;        // prompt attention workgroup order: 1 longest query tiles first, kv head = XCD | 0 query tile fastest
    int pa_mt = 1;           // row tiles per wave of the first form
    int pp_fuse_qk = 1;      // prompt pass: q/k RMSNorm + RoPE + cache write in the q|k|v projection's epilogue (head-tile GEMM, gemm.h MODE 2) | 0 separate launch
    int qknr_wide = 1;       // q/k norm + RoPE of the prompt pass: 16-byte accesses
    int conv_ktile = 1;      // implicit-GEMM convolutions (C >= 64): tap decomposition once per staged K-tile on the scalar unit (AConv3x3s2W) | 0 per chunk
    int enc_attn = 1;        // Qwen3 audio-encoder window attention at head_dim 64: 1 the wav2vec2 path's 32x32x16 kernel | 0 16-row kernel
    int mha_form = 1;        // Omnilingual attention at head_dim 64: 1 | 2 transposed scores on 32x32x16 MFMAs, 128 | 256 queries per workgroup; 0 16x16x32 form
    int gemm_p8 = 1;         // 256 x 256 ping-pong GEMM form: 0 never | 1 for launches of many tiles | 2 always
    int gemm_tm = 8;         // 128 x 128 GEMM forms: tile order in blocks of this many row panels (1 = row-panel-major) -- A/B in profiles/r04_ab_gemm_order.txt
    int gemm_nbuf = 0;       // GEMM LDS buffers: 0 auto (by tile count) | 1 | 2
    int lmh_q_ring = 1;      // quantised LM head weight stream: 1 wave-private LDS ring (direct-to-LDS) | 0 register ring of four blocks
    int lmh_order = 1;       // LM head tile -> wave order: 1 workgroup fastest (the waves that own one tile more are spread over all workgroups) | 0 wave fastest
    int lmh_grid = 256;      // persistent LM-head workgroups (read at qasr_finalize: sizes the argmax partials)
    int lmh_diag = 0;        // diagnostic: LM-head loop without LDS reads / MFMA (wrong results, timing only)
    int decode_split = 1;    // decode row groups on parallel graph branches
    int decode_gran = 16;    // rows per such group (multiple)
    int graph_steps = 8;     // decode steps captured per hipGraph launch: 1 | 2 | 4 | 8 (8 = the EOS poll interval: -0.4 ms of decode at 32 clips)
    int use_graph = 1;       // 0: issue every decode step eagerly (no hipGraph replay)
    int device_sampler = 1;  // non-default decoding options: 1 pickNextToken on the device (no host round trip per step) | 0 logits to the host,
                             // csrc/sampler.cpp picks (the reference's own structure)
    int da_stamps = 0, gemv_stamps = 0, stamps_insitu = 0, pa_stamps = 0;   // diagnostics of qasr_kernel_probe (make DIAG=1 builds)
    unsigned epoch = 0;
};

Tuning& tuning();                                   // process-wide; first call seeds it from the environment
// Set by an engine around the launches of a decode step on the CALLING thread: the engine shares its GPU with other engines (qasr_dp lanes),
// so kernels that leave room on a CU are preferred: the one-chunk attention (same chunk -> wave map, bit-identical; +2 % with three lanes, -0.3 % alone)
void tuning_thread_shared(bool shared);
bool tuning_thread_is_shared();
bool tuning_set(const char* key, int value);        // false: unknown key, or a value outside the knob's enumerated / ranged set
bool tuning_get(const char* key, int* value);

}  // namespace qasr
