"""MI355X-native ShadowKV decode path (hand-written HIP kernels behind the reference's
`kv_cache` / `tensor_op` / `kernels.shadowkv` Python surface)."""
__version__ = "0.1.0"
