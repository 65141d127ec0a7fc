// End of the decode step (LLM.generate's loop body, /root/reference/models/base.py:628-635, sampling semantics of
// models/tensor_op.py:242-297 sample_token): top-p filter over the k sorted top-k logits, multinomial draw, and the
// step's device-side bookkeeping (RoPE position, generated-row slot, attended length, query-table index) in ONE launch.
// The PyTorch formulation is ~20 tiny launches (softmax, cumsum, cat, masked_fill, softmax, exponential_, div, argmax,
// gather, copy, and 2-3 per counter), ~60 us of a 4.9 ms step at bs = 1.
//   vals [bs][k] f32: top-k logits / temperature, sorted descending (torch.topk);  idx [bs][k] int64 their token ids
//   keep i  <=>  i == 0 or cumsum(softmax(vals))[i-1] <= top_p        (tensor_op.py:258-266 after the top-k filter)
//   draw     =  argmax_i p_i / e_i,  e_i ~ Exp(1)                     (== multinomial(p, 1))
// The uniforms come from a counter-based hash of (seed, position of the sequence, batch row, i): reproducible, no
// generator state, graph-capturable.  One wave per sequence; k <= 64.
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"

__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // lowbias32 finaliser
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(64) void skv_sample_advance_kernel(
    const float* __restrict__ vals, const int64_t* __restrict__ idx, int k, float top_p, unsigned long long seed,
    int64_t* __restrict__ token /*[bs]*/, int64_t* __restrict__ pos /*[bs]*/, int64_t* __restrict__ gen /*[1]*/,
    int64_t* __restrict__ row_idx /*[1]*/, int32_t* __restrict__ kv_len /*[1]*/, int64_t* __restrict__ step_idx /*[1]*/,
    long long base, long long slack, long long table_len) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const long long p0 = pos[b];
    const float v = lane < k ? vals[(size_t)b * k + lane] : -INFINITY;
    const float mx = wave_max_dpp(v);
    float e = lane < k ? __expf(v - mx) : 0.f;
    const float tot = wave_tree_sum(e);
    const float p = e / tot;
    // inclusive scan of p over the lanes (descending probabilities), exclusive = cumsum[i-1]
    float c = p;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float n = __shfl_up(c, o, 64);
        if (lane >= o) c += n;
    }
    const bool keep = lane < k && (top_p <= 0.f || lane == 0 || (c - p) <= top_p);
    const float pk = keep ? e : 0.f;                     // renormalisation is a common factor: irrelevant to the argmax
    // u in (0, 1]: 24 random bits
    uint32_t h = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x9e3779b9u * (uint32_t)p0));
    h = mix32(h ^ (0x85ebca6bu * (uint32_t)(b + 1)) ^ (0xc2b2ae35u * (uint32_t)(lane + 1)) ^ (uint32_t)(p0 >> 32));
    const float u = ((h >> 8) + 1) * (1.0f / 16777216.0f);
    const float ex = -__logf(u);                          // Exp(1), > 0 except u == 1 -> 0: guarded below
    float score = keep ? pk / fmaxf(ex, 1e-30f) : -1.f;
    // argmax with lowest-lane tie break
    float best = wave_max_dpp(score);
    const unsigned long long m = __ballot(score == best);
    const int win = __ffsll((long long)m) - 1;
    if (lane == win) token[b] = idx[(size_t)b * k + lane];
    if (lane == 0) {
        pos[b] = p0 + 1;
        if (b == 0) {                                     // gen counts generated tokens; past the slack the rows form a
            const long long g2 = gen[0] + 1;              // ring of the last `slack` tokens (GraphDecoder: benchmarks only,
            gen[0] = g2;                                  // the host refuses to step there otherwise)
            row_idx[0] = base + g2 % slack;
            kv_len[0] = (int32_t)(base + (g2 + 1 < slack ? g2 + 1 : slack));
            if (step_idx) step_idx[0] = (step_idx[0] + 1) % table_len;
        }
    }
}

extern "C" int skv_sample_advance(const float* vals, const int64_t* idx, int batch_size, int k, float top_p,
                                  unsigned long long seed, int64_t* token, int64_t* pos, int64_t* gen, int64_t* row_idx,
                                  int32_t* kv_len, int64_t* step_idx, long long base, long long slack,
                                  long long table_len, skv_stream_t stream) {
    if (!vals || !idx || !token || !pos || !gen || !row_idx || !kv_len || batch_size < 1) return SKV_ERR_ARG;
    if (k < 1 || k > 64 || slack < 1 || (step_idx && table_len < 1)) return SKV_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(skv_sample_advance_kernel, dim3(batch_size), dim3(64), 0, (hipStream_t)stream, vals, idx, k, top_p,
                       seed, token, pos, gen, row_idx, kv_len, step_idx, base, slack, table_len);
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}
