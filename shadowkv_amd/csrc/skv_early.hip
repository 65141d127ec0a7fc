// Speculative early V fetch (round 3): the state initialiser.  The device roles live in skv_early.h (they ride in the
// normalise and top-k launches of skv_select.hip), the consumer in skv_rebuild.hip.
#include "../../include/shadowkv_hip.h"
#include "skv_common.h"
#include "skv_launch.h"

__global__ void skv_early_init_kernel(float* dthr, int n_dthr, int* ints, int n_ints, short* early_of, long long n_of) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    for (long long k = i; k < n_dthr; k += stride) dthr[k] = INFINITY;
    for (long long k = i; k < n_ints; k += stride) ints[k] = 0;
    for (long long k = i; k < n_of; k += stride) early_of[k] = (short)-1;
}

int skv_launch_early_init(const EarlyState& es, int B, int G, int n_landmarks, int n_chunks, int E, hipStream_t st) {
    // flag_cnt .. early_ids are contiguous int regions (see skv_carve_early): zero from flag_cnt to the end of early_ids
    const long long n_ints = ((unsigned char*)es.early_of - (unsigned char*)es.flag_cnt) / 4;
    hipLaunchKernelGGL(skv_early_init_kernel, dim3(256), dim3(256), 0, st, es.dthr, B * G, es.flag_cnt, (int)n_ints,
                       es.early_of, (long long)B * n_chunks);
    return hipGetLastError() == hipSuccess ? SKV_OK : SKV_ERR_LAUNCH;
}

