// Host-side launch descriptors shared between the API layer (skv_api.hip) and the kernel files.
#pragma once

// attention role of the fused in-place fetch launch (skv_rebuild.hip)
struct AttnLaunch {
    const void* q;           // [bs][Hq][128] bf16
    void* ws;                // attention workspace, rec_splits records per query head
    const int* kv_len_dev;   // nullable
    int kv_len_host;
    int kv_rows;             // rows a head owns in the cache buffers: the device-side kv_len is clamped to it
    int G, splits, rec_splits;
    float scale;
    int resident_sets;       // slots of the sparse region (>= select_sets; the generated rows sit behind resident_sets * 8 rows)
};
