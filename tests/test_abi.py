"""CPU: the C-ABI library loads and exports every symbol include/shadowkv_hip.h declares
(no compute calls without a GPU); the product never imports the oracle."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "shadowkv_hip.h")).read()
    return sorted(set(re.findall(r"SKV_EXPORT\s+[\w\s\*]+?\b(skv_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from shadowkv_amd import _lib
    names = declared_symbols()
    assert len(names) >= 36
    l = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(l, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == names, "ctypes signature table out of sync with the header"
    assert _lib.lib().skv_abi_version() == 1
    assert _lib.lib().skv_select_workspace_bytes(8, 4, 15560) > 0
    assert _lib.lib().skv_attn_workspace_bytes(1, 32, 32) == 32 * 32 * 132 * 4


def test_header_compiles_as_plain_c():
    src = '#include "shadowkv_hip.h"\nint main(void){return skv_abi_version == 0;}\n'
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c", "-"],
                       input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()


def test_mirror_module_has_the_twelve_reference_names():
    from shadowkv_amd.kernels import shadowkv
    names = ["gather_copy", "gather_copy_d2d_with_offsets", "reorder_keys_and_compute_offsets",
             "gather_copy_with_offsets", "apply_rotary_pos_emb", "apply_rotary_pos_emb_new",
             "apply_rotary_pos_emb_new_v2", "apply_rotary_pos_emb_push_cache", "apply_rotary_pos_emb_push_cache_opt",
             "apply_rotary_pos_emb_push_cache_opt_glm", "batch_gather_gemm", "batch_gemm_softmax"]  # main.cu:42-81
    assert all(callable(getattr(shadowkv, n)) for n in names)


def test_product_never_touches_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "shadowkv_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "libshadowkv_oracle" not in text, f
                assert "/root/reference" not in text or f.endswith((".py", ".hip", ".h")), f


def test_round4_entries_refuse_bad_arguments_before_any_launch():
    """Argument checks of the round-4 entry points return their error codes on the host (no HIP call is made before them), so
    they can be exercised without a GPU: -1 = SKV_ERR_ARG, -2 = SKV_ERR_UNSUPPORTED."""
    from shadowkv_amd import _lib
    L = _lib.lib()
    p = 0x1000                                                  # a non-null pointer that is never dereferenced on the host
    # lm_head with range maxima: whole workgroups of 16 rows, the norm prologue's hidden size, a keys buffer
    assert L.skv_norm_gemv_rangemax_bf16(p, p, 0, p, 1e-5, 0, 0, p, 128250, 4096, p, 0) == -2      # N % 16 != 0
    assert L.skv_norm_gemv_rangemax_bf16(p, p, 0, p, 1e-5, 0, 0, p, 128256, 2048, p, 0) == -2      # K != 4096
    assert L.skv_norm_gemv_rangemax_bf16(p, p, 0, p, 1e-5, 0, 0, p, 128256, 4096, 0, 0) == -1      # no keys buffer
    # sampler through range maxima
    tail = (1, 50, 0.6, 0.9, 7, p, p, p, p, p, 0, 0, 96, 1, 0, 0, 0, 0)
    assert L.skv_sample_topk_advance_ranges(p, 128256, 128256, 0, 8016, *tail) == -1               # no keys
    assert L.skv_sample_topk_advance_ranges(p, 128264, 128264, p, 8024, *tail) == -2               # vocab % 16 != 0
    assert L.skv_sample_topk_advance_ranges(p, 524288, 524288, p, 32768, *tail) == -2              # more than 16,384 ranges
    assert L.skv_sample_topk_advance_ranges(p, 128256, 128256, p + 2, 8016, *tail) == -2           # keys not 16-B aligned
    assert L.skv_sample_topk_advance_ranges(p, 128256, 128256, p, 8015, *tail) == -2               # stride % 8 != 0
    # fused selection: shapes, state
    assert L.skv_select_fused_supported(4, 15560, 256) == 1 and L.skv_select_fused_supported(3, 15560, 256) == 0
    assert L.skv_select_state_stats_offset(8, 4) < L.skv_select_state_bytes(8, 4)
    assert L.skv_select_state_bytes(8, 4) - L.skv_select_state_stats_offset(8, 4) >= 8 * 2 * 4
    args = (p, p, p, p, p, p, p, p, p)
    assert L.skv_select_chunks_fused(*args, 8, 4, 15560, 256, 256, 0, 0.088, 0, 0, 0, 0, 0, 0, 0.0, 0) == -1   # no select state
    assert L.skv_select_chunks_fused(*args, 8, 3, 15560, 256, 256, 0, 0.088, p, 0, 0, 0, 0, 0, 0.0, 0) == -2   # G = 3
