// dec_chain.h -- the decode layer's weight-streaming linears as ONE persistent launch with in-launch hand-offs (dec_chain.hip).
//
// Reference loop served: generateGreedyAsyncEval, Sources/Qwen3ASR/Qwen3ASR.swift:317-390; the layer is
// QuantizedTextDecoderLayer / FloatTextDecoderLayer (QuantizedTextDecoder.swift:141-175, FloatTextDecoder.swift:120-160):
//     x += o_proj(attention);  x += down(silu(gate(norm(x))) * up(norm(x)));  next layer: q|k|v = qkv_proj(norm(x)).
#pragma once
#include "dec_kernels.h"

namespace qasr {

enum DecChainPhase { CHAIN_O = 1, CHAIN_GU = 2, CHAIN_DOWN = 4, CHAIN_QKV = 8 };

constexpr int CHAIN_SEAMS = 4;            // o -> gate|up, gate|up -> down, down -> next q|k|v; block 3: q|k|v -> attention, one counter per kv head (dec_qa.hip)
constexpr int CHAIN_SHARDS = 8;           // arrival counters are sharded over 8 lines (fan-in of up to 192 producers)
constexpr int CHAIN_SHARD_WORDS = 32;     // one 128-byte line per shard
constexpr size_t CHAIN_CTR_BYTES = (size_t)CHAIN_SEAMS * CHAIN_SHARDS * CHAIN_SHARD_WORDS * sizeof(unsigned);
// behind the counters, on a line of its own: the step sequence word (never zeroed; bumped once in front of every step's fused launches)
constexpr int CHAIN_SEQ_WORD = (int)(CHAIN_CTR_BYTES / sizeof(unsigned)) + CHAIN_SHARD_WORDS;
constexpr size_t CHAIN_STATE_BYTES = CHAIN_CTR_BYTES + 2 * CHAIN_SHARD_WORDS * sizeof(unsigned);
constexpr int CHAIN_ERR_TIMEOUT = 2;      // bit set in *err when a hand-off wait gave up (host: QASR_ERR_HIP)

struct DecChainArgs {
    const bf16_t* attn;     // [B][nq]  attention output of this layer (written by the previous launch)
    const bf16_t* wo_p;     // o_proj, fragment-major [H][nq]
    bf16_t* x;              // [B][H]   residual stream, updated in place by O and DOWN
    const bf16_t* ln2;      // [H]      post-attention RMSNorm weight
    const bf16_t* wgu_p;    // gate|up in 32-row blocks, fragment-major [2 I][H]
    bf16_t* act;            // [B][I]
    const bf16_t* wdown_p;  // down_proj, fragment-major [H][I]
    const bf16_t* ln1n;     // [H]      the NEXT layer's input RMSNorm weight (QKV phase)
    const bf16_t* wqkv_p;   // the NEXT layer's q|k|v, fragment-major [nqkv][H]
    bf16_t* qkv;            // [B][nqkv]
    int B;
    float eps;
    unsigned* ctr;          // CHAIN_CTR_BYTES, zeroed at the start of every decode step (decode_chain_reset)
    unsigned epoch;         // chain launches earlier in this step: the counters count up through a step
    int* err;               // device word; CHAIN_ERR_TIMEOUT is or-ed in when a wait gives up
    unsigned long long* dbg = nullptr;   // diagnostic: [256][32] phase stamps of this launch (qasr_kernel_probe 6), null in product launches
    int proto = 0;          // arrival counters: 0 sharded (a producer adds to one of 8, a consumer polls all 8) | 1 replicated (adds to all, polls one)
};

// q|k|v projection + decode attention of one layer as one launch (dec_qa.hip): the K / V stream is requested before the projection runs
struct DecQaArgs {
    const bf16_t* x;        // [B][H] layer input (written by the previous launch)
    const bf16_t* ln1;      // [H]
    const bf16_t* wqkv_p;   // fragment-major [nqkv][H]
    bf16_t* qkv;            // [B][nqkv] hand-off rows (write-through)
    const int* ctx_len;     // [B]
    const bf16_t *qn_w, *kn_w;
    const float *rope_cos, *rope_sin;   // per-row copies of the current position's table rows [B][hd / 2]
    KVLayout cache;
    bf16_t* out;            // [B][heads * hd]
    int B;
    float eps, scale;       // RMSNorm epsilon; 1 / sqrt(head_dim)
    unsigned* ctr;          // the chain's counter block (block 3 is used here)
    unsigned epoch;         // layer index
    int* err;
    unsigned long long* dbg = nullptr;   // diagnostic: [256][32] phase stamps (qasr_kernel_probe 7), null in product launches
    int fault = 0;          // test only (knob chain_fault): workgroup 5 never signals -> the bounded waits must end the step with CHAIN_ERR_TIMEOUT
    // hand-off form (knob qa_gran): non-null = the projected rows travel as 8-byte {two bf16 values, tag} granules and the data IS the flag
    // (cdna_hip_programming.md Guideline 16, R2): no drain, no counter, no second round trip for the rows.  [32 rows][QA_GRAN_ROW] granules;
    // tag = (ctr[CHAIN_SEQ_WORD] << 8) | (epoch + 1): the sequence word is bumped by whatever zeroes the counters in front of a step
    // (greedy_finalize_kernel, decode_chain_reset), so a tag never equals what an earlier step or layer left in the buffer
    unsigned long long* gran = nullptr;
    // MLX affine-quantised checkpoints: the q|k|v matrix as its packed decode-step image (dec_quant.h QuantImg; 4 or 8 bits, bf16 scales) instead
    // of wqkv_p; needs gran
    // up to 16 batch rows: an attention unit spread over 2 / 4 / 8 workgroups (knob qa_split); their per-wave partials travel here,
    // [unit][8 waves][QA_PART_STRIDE] granules, same tags
    unsigned long long* part = nullptr;
    int xbar = 0;
    const uint32_t* wq_qp = nullptr;
    const void* wq_sb = nullptr;
    int wq_bits = 0;
};
constexpr int QA_PART_STRIDE = 264;                                // 2 x 128 outputs + 2 maxima + 2 sums, padded to whole 64-byte blocks
constexpr size_t QA_PART_BYTES = (size_t)128 * 8 * QA_PART_STRIDE * 8;   // 16 rows x 8 kv heads = 128 units
constexpr int QA_GRAN_ROW = 2048;                                   // granules per batch row (4096 projected values)
constexpr size_t QA_GRAN_BYTES = (size_t)32 * QA_GRAN_ROW * 8;
bool decode_qa_supported(int H, int heads, int kv_heads, int hd, int B, int max_ctx);
void decode_qa_launch(const DecQaArgs& a, hipStream_t s);

// true when the geometry has a chain instantiation (0.6B decoder: hidden 1024, 16 x 128 query dims, inter 3072, q|k|v 4096 rows;
// 1..32 batch rows) and the device has a CU for every workgroup of the persistent grid
bool decode_chain_supported(int H, int nq, int I, int nqkv, int B);
// phases: CHAIN_O | CHAIN_GU [| CHAIN_DOWN [| CHAIN_QKV]]
void decode_chain_launch(int phases, const DecChainArgs& a, hipStream_t s);
void decode_chain_reset(unsigned* ctr, hipStream_t s);

}  // namespace qasr
