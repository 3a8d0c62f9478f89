// dec_rope.h -- the q/k RMSNorm + RoPE arithmetic of one rotation pair, shared by the prompt pass (dec_prefill.hip) and the
// decode-step attention (dec_attention.hip): the two must round at the same places.
#pragma once
#include "common.h"

namespace qasr {

__device__ __forceinline__ void norm_rope_pair(float x1, float x2, float w1, float w2, float inv, float c, float sn,
                                               float& o1, float& o2) {
    // bf16(w * bf16(x * inv))  then  bf16(x1*cos - x2*sin), bf16(x1*sin + x2*cos)
    float y1 = bf16_round(w1 * bf16_round(x1 * inv));
    float y2 = bf16_round(w2 * bf16_round(x2 * inv));
    o1 = bf16_round(y1 * c - y2 * sn);
    o2 = bf16_round(y1 * sn + y2 * c);
}

}  // namespace qasr
