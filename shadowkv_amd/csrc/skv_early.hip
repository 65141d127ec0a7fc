// Speculative early V fetch (round 3): the state initialiser.  The device roles live in skv_early.h (they ride in the
// normalise and top-k launches of skv_select.hip), the consumer in skv_rebuild.hip.
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"
#include "skv_launch.h"

__global__ void skv_early_init_kernel(float* dthr, int n_dthr, int* ints, int n_ints, int* early_ids, int n_ids, short* early_of,
                                      long long n_of, int* map_ok, int n_map, int* near_cnt, int* near_ids, int* near_pub) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    for (long long k = i; k < n_dthr; k += stride) dthr[k] = INFINITY;
    for (long long k = i; k < n_ints; k += stride) ints[k] = 0;
    for (long long k = i; k < n_ids; k += stride) early_ids[k] = -1;  // every staging slot unused
    for (long long k = i; k < n_of; k += stride) early_of[k] = (short)-1;
    for (long long k = i; k < n_map; k += stride) map_ok[k] = 0;      // no slot -> id map yet: the list role gathers
    for (long long k = i; k < (long long)SKV_NEAR_LISTS * n_map; k += stride) near_cnt[k] = 0;    // (n_map = B) no near-miss lists, nothing staged ahead
    for (long long k = i; k < (long long)SKV_NEAR_LISTS * n_map * SKV_NEAR_MAX; k += stride) near_ids[k] = near_pub[k] = -1;
}

// slot -> chunk id map of one head (EarlyHooks::gap_slots): the landmark ids lm_idx[b][0 .. N) are chunk ids in ascending order
// with up to SKV_EARLY_GAPS chunks left out (the reference registers every chunk but the outliers, kv_cache.py:903-919), i.e.
// d_j = lm_idx[j] - j is a non-decreasing step function from >= 0 up to the number of gaps; gap_slots[i] = the first slot j with
// d_j > i.  Anything else (unsorted ids, more gaps): map_ok = 0 and the list role keeps its gather.
__global__ __launch_bounds__(1024) void skv_early_map_kernel(const int64_t* __restrict__ lm_idx, int N, int* __restrict__ gap_slots,
                                                             int* __restrict__ map_ok) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ int s_bad;
    if (tid == 0) s_bad = 0;
    for (int i = tid; i < SKV_EARLY_GAPS; i += 1024) gap_slots[(size_t)b * SKV_EARLY_GAPS + i] = 0x7fffffff;
    __syncthreads();
    for (int j = tid; j < N; j += 1024) {
        const long long d = lm_idx[(size_t)b * N + j] - j, dp = j > 0 ? lm_idx[(size_t)b * N + j - 1] - (j - 1) : 0;
        // both ends of the run [dp, d) are checked: an id below its slot index (duplicates, zeros) makes dp negative while d
        // may still be in range - the store loop must not start in front of this head's table
        if (d < dp || dp < 0 || d < 0 || d >= SKV_EARLY_GAPS) s_bad = 1;      // (the list role's binary search counts up to GAPS - 1)
        else
            for (long long i = dp; i < d; ++i) gap_slots[(size_t)b * SKV_EARLY_GAPS + i] = j;
    }
    __syncthreads();
    if (tid == 0) map_ok[b] = s_bad ? 0 : 1;
}

int skv_launch_early_map(const EarlyState& es, const int64_t* lm_idx, int B, int N, hipStream_t st) {
    hipLaunchKernelGGL(skv_early_map_kernel, dim3(B), dim3(1024), 0, st, lm_idx, N, es.gap_slots, es.map_ok);
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}

int skv_launch_early_init(const EarlyState& es, int B, int G, int n_landmarks, int n_chunks, int E, hipStream_t st) {
    // flag_cnt .. early_cnt are contiguous int regions (see skv_carve_early): zero from flag_cnt to the end of early_cnt
    const long long n_ints = ((unsigned char*)es.early_ids - (unsigned char*)es.flag_cnt) / 4;
    hipLaunchKernelGGL(skv_early_init_kernel, dim3(256), dim3(256), 0, st, es.dthr, B * G, es.flag_cnt, (int)n_ints, es.early_ids,
                       B * E, es.early_of, (long long)B * n_chunks, es.map_ok, B, es.near_cnt, es.near_ids, es.near_pub);
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}

