// dec_chain.h -- the decode layer's weight-streaming linears as ONE persistent launch with in-launch hand-offs (dec_chain.hip).
//
// Reference loop served: generateGreedyAsyncEval, Sources/Qwen3ASR/Qwen3ASR.swift:317-390; the layer is
// QuantizedTextDecoderLayer / FloatTextDecoderLayer (QuantizedTextDecoder.swift:141-175, FloatTextDecoder.swift:120-160):
//     x += o_proj(attention);  x += down(silu(gate(norm(x))) * up(norm(x)));  next layer: q|k|v = qkv_proj(norm(x)).
#pragma once
#include "dec_kernels.h"

namespace qasr {

enum DecChainPhase { CHAIN_O = 1, CHAIN_GU = 2, CHAIN_DOWN = 4, CHAIN_QKV = 8 };

constexpr int CHAIN_SEAMS = 3;            // o -> gate|up, gate|up -> down, down -> next q|k|v
constexpr int CHAIN_SHARDS = 8;           // arrival counters are sharded over 8 lines (fan-in of up to 192 producers)
constexpr int CHAIN_SHARD_WORDS = 32;     // one 128-byte line per shard
constexpr size_t CHAIN_CTR_BYTES = (size_t)CHAIN_SEAMS * CHAIN_SHARDS * CHAIN_SHARD_WORDS * sizeof(unsigned);
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
};

// true when the geometry has a chain instantiation (0.6B decoder: hidden 1024, 16 x 128 query dims, inter 3072, q|k|v 4096 rows;
// 1..32 batch rows) and the device has a CU for every workgroup of the persistent grid
bool decode_chain_supported(int H, int nq, int I, int nqkv, int B);
// phases: CHAIN_O | CHAIN_GU [| CHAIN_DOWN [| CHAIN_QKV]]
void decode_chain_launch(int phases, const DecChainArgs& a, hipStream_t s);
void decode_chain_reset(unsigned* ctr, hipStream_t s);

}  // namespace qasr
