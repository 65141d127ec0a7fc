"""Host-side decoder model for the ShadowKV decode path on MI355X.

Mirrors the call structure of the reference's `LLM` / `Llama` / `GLM`
(/root/reference/models/base.py:128-160 `inference`, :247-370 `layer_compute`;
/root/reference/models/llama.py:283-427; /root/reference/models/glm.py) for the DECODE branch:
pre_attention_compute -> apply_rotary_pos_emb -> kv_cache.update_kv_cache ->
kv_cache.get_retrieval_position_ids -> [copy_stream: get_value_cache] || get_key_cache -> attention ->
post_attention_compute.  Dense layers run on PyTorch-ROCm (`F.linear`, `F.rms_norm`); everything the
reference gets from flash-attn / vLLM / flashinfer comes from shadowkv_amd.tensor_op.

Weights: there are no checkpoints offline, so layers are created with random bf16 weights in the
named architecture's shapes (`random_init=True`); `load_state(...)` accepts real tensors with the same
fused layouts (wqkv = [q;k;v], gate_up = [gate;up]) as llama.py:111-128.
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import tensor_op
from ._lib import lib, check, ptr, current_stream_handle
from .kv_cache import KV_Cache, ShadowKVCache_CPU


@dataclass
class ModelConfig:
    name: str = "llama-3.1-8b"
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    vocab_size: int = 128256
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    qkv_bias: bool = False
    rope_style: str = "neox"          # "neox": half-split over 128 dims; "glm": interleaved pairs over the first 64 dims


LLAMA_3_1_8B = ModelConfig()
LLAMA_3_8B_1048K = ModelConfig(name="llama-3-8b-gradient-1048k", rope_theta=3580165449.0)
# 01-ai/Yi-9B-200K (test/e2e.py:76-94, index.html:240-258; served by models/llama.py): 48 layers, 32 query / 4 KV heads (G = 8
# with NeoX RoPE), hidden 4096, intermediate 11008, vocabulary 64000
YI_9B_200K = ModelConfig(name="yi-9b-200k", intermediate_size=11008, num_hidden_layers=48, num_key_value_heads=4,
                         vocab_size=64000, rms_norm_eps=1e-6, rope_theta=10000000.0)
GLM_4_9B_1M = ModelConfig(name="glm-4-9b-1m", intermediate_size=13696, num_hidden_layers=40, num_key_value_heads=4,
                          vocab_size=151552, rms_norm_eps=1.5625e-07, rope_theta=10000.0 * 1e4, qkv_bias=True,
                          rope_style="glm")


class DecoderLayer:
    """Fused weight layout of the reference's LlamaLayer (llama.py:59-152)."""

    def __init__(self, cfg, device, dtype, gen):
        h, i = cfg.hidden_size, cfg.intermediate_size
        d = h // cfg.num_attention_heads
        self.q_size, self.kv_size = h, cfg.num_key_value_heads * d

        def w(*shape):
            return (torch.randn(*shape, device=device, dtype=torch.float32, generator=gen) * 0.02).to(dtype)

        self.wqkv = w(self.q_size + 2 * self.kv_size, h)
        self.bqkv = w(self.q_size + 2 * self.kv_size) if cfg.qkv_bias else None
        self.wo = w(h, h)
        self.gate_up_proj = w(2 * i, h)
        self.down_proj = w(h, i)
        self.input_layernorm_weight = torch.ones(h, device=device, dtype=dtype)
        self.post_attention_layernorm_weight = torch.ones(h, device=device, dtype=dtype)
        self.input_layernorm_variance_epsilon = cfg.rms_norm_eps
        self.post_attention_layernorm_variance_epsilon = cfg.rms_norm_eps


def build_cos_sin_cache(cfg, max_pos, device, dtype):
    """[max_pos, 128] = cos[:64] | sin[:64] (llama.py:323-332) or, GLM, [max_pos, 64] = cos[:32] | sin[:32]
    (glm.py:261-273)."""
    rot = cfg.hidden_size // cfg.num_attention_heads if cfg.rope_style == "neox" else 64
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, rot, 2, dtype=torch.float32, device=device) / rot))
    freqs = torch.outer(torch.arange(max_pos, dtype=torch.float32, device=device), inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype).contiguous()


class DecoderLM:
    def __init__(self, cfg=LLAMA_3_1_8B, batch_size=1, max_length=64 * 1024, device="cuda:0", dtype=torch.bfloat16,
                 attn_mode="shadowkv_cpu", sparse_budget=2048, rank=160, chunk_size=8, random_init=True, seed=1234,
                 num_layers=None, chunk_layout="reference", v_offload=True, overlap_attention=False, max_new_tokens=1024,
                 resident_sets=None):
        if chunk_layout not in ("reference", "inplace"):
            raise ValueError("chunk_layout must be 'reference' (hits compacted to the front, the reference's slot order) "
                             "or 'inplace' (hits keep their slots, misses take the freed slots)")
        self.chunk_layout = chunk_layout     # used by forward_fused only; layer_compute follows the reference
        self.overlap_attention = overlap_attention   # in-place layout: attention over resident rows inside the fetch launch
        if attn_mode not in ("shadowkv_cpu", "full"):
            raise ValueError("attn_mode must be 'shadowkv_cpu' (ShadowKV offload path) or 'full' (full-attention baseline)")
        self.attn_mode = attn_mode
        self.cfg = cfg
        self.config = cfg
        self.batch_size, self.max_length = batch_size, max_length
        self.device, self.dtype = torch.device(device), dtype
        self.hidden_size = cfg.hidden_size
        self.num_heads = cfg.num_attention_heads
        self.num_key_value_heads = cfg.num_key_value_heads
        self.head_dim = cfg.hidden_size // cfg.num_attention_heads
        self.num_layers = num_layers or cfg.num_hidden_layers
        self.vocab_size = cfg.vocab_size
        gen = torch.Generator(device=self.device).manual_seed(seed)
        if not random_init:
            raise NotImplementedError("no checkpoints are available offline; use random_init=True or load_state()")
        self.embed_tokens = (torch.randn(cfg.vocab_size, cfg.hidden_size, device=self.device, dtype=torch.float32,
                                         generator=gen) * 0.02).to(dtype)
        self.lm_head = (torch.randn(cfg.vocab_size, cfg.hidden_size, device=self.device, dtype=torch.float32,
                                    generator=gen) * 0.02).to(dtype)
        self.norm_weight = torch.ones(cfg.hidden_size, device=self.device, dtype=dtype)
        self.norm_variance_epsilon = cfg.rms_norm_eps
        self.layers = [DecoderLayer(cfg, self.device, dtype, gen) for _ in range(self.num_layers)]
        for i, lay in enumerate(self.layers):
            lay.layer_idx = i            # (post_attention_compute has the reference's signature: the layer object only)
        # RoPE rows for the context plus every position decode can reach (the kernels index it with the device-side
        # position counter: it must never run past the table)
        self.max_new_tokens = int(max_new_tokens)
        self.cos_sin_cache = build_cos_sin_cache(cfg, max_length + self.max_new_tokens, self.device, dtype)

        class _CacheCfg:
            num_hidden_layers = self.num_layers
            num_attention_heads = cfg.num_attention_heads
            num_key_value_heads = cfg.num_key_value_heads
            hidden_size = cfg.hidden_size

        if attn_mode == "full":
            self.kv_cache = KV_Cache(_CacheCfg, batch_size=batch_size, max_length=max_length + 1024, device=device,
                                     dtype=dtype)
        else:
            self.kv_cache = ShadowKVCache_CPU(_CacheCfg, batch_size=batch_size, max_length=max_length, device=device,
                                              dtype=dtype, sparse_budget=sparse_budget, chunk_size=chunk_size,
                                              rank=rank, v_offload=v_offload, resident_sets=resident_sets)
        self.query_hook = None   # optional: q -> q used for selection/attention (bench: synthetic query walk)
        self._sampler_state = {}
        # lm_head -> sampler through range maxima (round 4): the lm_head launch of forward_fused leaves the largest of every 16
        # logits as a 16-bit key, the native sampler reads those instead of streaming the row through one CU (same token)
        self.sampler_ranges = True
        self._range_max = None          # int16 [1, vocab / 16 rounded up to 8]
        self._last_range_max = None     # the keys of the logits forward_fused returned last (None: not produced)

    def weight_bytes(self):
        n = self.lm_head.numel()   # one embedding row is read per token: not counted
        for l in self.layers:
            n += l.wqkv.numel() + l.wo.numel() + l.gate_up_proj.numel() + l.down_proj.numel()
        return n * 2

    # --------------------------------------------------------------- per-layer pieces (llama.py:283-427)
    def _decode_scratch(self, bs):
        """Result buffers of the one-token-per-sequence path of the reference-shaped methods (pre_attention_compute,
        apply_rotary_pos_emb, post_attention_compute), reused from layer to layer: every one of them is consumed on the
        stream before the same method of the next layer overwrites it (torch.empty per call was ~20 us of host time per
        layer on the eager call-order path).  Two sets alternate so that a layer's output never aliases its input."""
        sc = getattr(self, "_scratch", None)
        if sc is None or sc["bs"] != bs:
            dev, dt, h = self.device, self.dtype, self.hidden_size
            layer = self.layers[0]
            n_qkv = layer.q_size + 2 * layer.kv_size

            def one():
                return dict(qkv=torch.empty(bs, 1, n_qkv, device=dev, dtype=dt),
                            q=torch.empty(bs, self.num_heads, 1, self.head_dim, device=dev, dtype=dt),
                            k=torch.empty(bs, self.num_key_value_heads, 1, self.head_dim, device=dev, dtype=dt),
                            v=torch.empty(bs, self.num_key_value_heads, 1, self.head_dim, device=dev, dtype=dt),
                            attn=torch.empty(bs, 1, self.num_heads, self.head_dim, device=dev, dtype=dt),
                            o=torch.empty(bs, 1, h, device=dev, dtype=dt), h=torch.empty(bs, 1, h, device=dev, dtype=dt),
                            act=torch.empty(bs, 1, layer.gate_up_proj.shape[0] // 2, device=dev, dtype=dt),
                            down=torch.empty(bs, 1, h, device=dev, dtype=dt), out=torch.empty(bs, 1, h, device=dev, dtype=dt))
            sc = dict(bs=bs, sets=[one(), one()], i=0, row0=torch.zeros(1, dtype=torch.long, device=dev),
                      row=torch.zeros(1, dtype=torch.long, device=dev), row_host=None)
            self._scratch = sc
        return sc

    def pre_attention_compute(self, hidden_states, layer):
        """RMSNorm -> fused QKV projection -> split (llama.py:283-303).  One decode token per sequence takes the native
        kernels (norm in the GEMV's prologue at bs 1, the rows GEMM for 2..32 sequences); prefill-sized inputs F.linear.
        ALIASING CONTRACT of the one-token path (pre_attention_compute, apply_rotary_pos_emb, post_attention_compute,
        layer_compute): results are views of two per-model scratch sets that ALTERNATE per layer - this method flips to the
        other set - so a returned tensor is valid until the same method has run for the layer after the next one (the
        reference returns fresh tensors).  Host code that keeps a hidden state or q / k / v across more than one layer, or
        across steps (hidden-state capture), must clone() it."""
        if hidden_states.shape[1] == 1 and hidden_states.is_cuda:
            sc = self._decode_scratch(hidden_states.shape[0])
            sc["i"] ^= 1                                        # (a new layer: the other buffer set)
            _, qkv = tensor_op.norm_linear_decode(hidden_states, None, layer.input_layernorm_weight,
                                                  layer.input_layernorm_variance_epsilon, layer.wqkv, layer.bqkv,
                                                  out=sc["sets"][sc["i"]]["qkv"])
        else:
            hs = tensor_op.layer_norm(hidden_states, layer.input_layernorm_variance_epsilon, layer.input_layernorm_weight)
            qkv = F.linear(hs, layer.wqkv, layer.bqkv)
        q, k, v = qkv.split([layer.q_size, layer.kv_size, layer.kv_size], dim=-1)
        return q, k, v.view(v.shape[0], -1, self.num_key_value_heads, self.head_dim).transpose(1, 2)

    def apply_rotary_pos_emb(self, q, k, position_ids, layer_idx=None):
        """q [bs, s, Hq*D], k [bs, s, Hkv*D] -> [bs, H, s, D] rotated at position_ids [bs, s].
        layer_idx (optional, one decode token, ShadowKV cache): the native launch that rotates q and k ALSO pushes the
        rotated k and v into the layer's cache row - what update_kv_cache is about to do (kv_cache.py:1227-1271) - and the
        cache is told, so that update_kv_cache finds the rows in place and only does its bookkeeping; a synthetic-query
        hook (bench: QueryWalk) rides in the kernel's q-override slot instead of a separate launch."""
        bs, s = q.shape[0], q.shape[1]
        n_fused = (self.num_heads + 2 * self.num_key_value_heads) * self.head_dim
        self._hook_applied = False
        if (s == 1 and q.is_cuda and q.stride(-1) == 1 and q.stride(0) == n_fused and k.stride(0) == n_fused
                and q.untyped_storage().data_ptr() == k.untyped_storage().data_ptr()
                and k.storage_offset() - q.storage_offset() == self.num_heads * self.head_dim):
            # one decode token: q and k are the split views of the fused projection's output (pre_attention_compute) - ONE
            # native launch rotates both (the kernel of the fused step)
            qkv = q.as_strided((bs, 1, n_fused), (n_fused, n_fused, 1))
            sc = self._decode_scratch(bs)
            cur = sc["sets"][sc["i"]]
            hook = self.query_hook
            q_over = None
            if layer_idx is not None and hook is not None and hasattr(hook, "qb_layers"):
                q_over = hook.qb_layers[layer_idx]
                self._hook_applied = True
            c = self.kv_cache
            if layer_idx is not None and self.attn_mode != "full" and c.incoming_rows_writable(1):
                row = c.sparse_end + c.gen_offset
                if sc.get("row_host") != row:
                    sc["row"].fill_(row)
                    sc["row_host"] = row
                lv = c._layer(layer_idx)
                kbuf, vbuf = lv.kbuf, lv.vbuf
                qr = tensor_op.qkv_rope_update(qkv, self.cos_sin_cache, position_ids, sc["row"], kbuf, vbuf, self.num_heads,
                                               self.num_key_value_heads, q_override=q_over, q_out=cur["q"])
                c.note_rows_pushed(layer_idx, row, 1, v_src_ptr=k.data_ptr() + self.num_key_value_heads * self.head_dim * 2)
                return qr, kbuf[:, :, row:row + 1]
            qr = tensor_op.qkv_rope_update(qkv, self.cos_sin_cache, position_ids, sc["row0"], cur["k"], cur["v"],
                                           self.num_heads, self.num_key_value_heads, q_override=q_over, q_out=cur["q"])
            return qr, cur["k"]
        q = q.view(bs, s, self.num_heads, self.head_dim).transpose(1, 2)
        k = k.view(bs, s, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        if self.cfg.rope_style == "neox":
            pid_q = position_ids.unsqueeze(1).expand(-1, self.num_heads, -1).contiguous()
            pid_k = position_ids.unsqueeze(1).expand(-1, self.num_key_value_heads, -1).contiguous()
            return (tensor_op.apply_rotary_pos_emb_cuda(q.contiguous(), self.cos_sin_cache, pid_q),
                    tensor_op.apply_rotary_pos_emb_cuda(k.contiguous(), self.cos_sin_cache, pid_k))
        return self._rope_glm(q, position_ids), self._rope_glm(k, position_ids)

    def _rope_glm(self, x, position_ids):
        cs = self.cos_sin_cache[position_ids].unsqueeze(1)          # [bs, 1, s, 64]
        c, s = cs[..., :32], cs[..., 32:]
        xr, xp = x[..., :64], x[..., 64:]
        xe, xo = xr[..., 0::2], xr[..., 1::2]
        out = torch.stack((xe * c - xo * s, xo * c + xe * s), dim=-1).flatten(-2)
        return torch.cat((out, xp), dim=-1).contiguous()

    def post_attention_compute(self, attn_output, residual, layer):
        """o-projection + residual -> RMSNorm -> gate/up -> SiLU*mul -> down + residual (llama.py:405-427)."""
        if attn_output.shape[1] == 1 and attn_output.is_cuda:       # one decode token per sequence: native kernels
            sc = self._decode_scratch(attn_output.shape[0])
            cur = sc["sets"][sc["i"]]
            o = tensor_op.linear_decode(attn_output, layer.wo, out=cur["o"])
            # (one sequence on the ShadowKV cache with near_fetch: the launch's first workgroups stage this step's near misses)
            near = None
            if attn_output.shape[0] == 1 and self.attn_mode != "full" and getattr(layer, "layer_idx", None) is not None:
                near = self.kv_cache.near_pull_args(layer.layer_idx)
            residual, act = tensor_op.norm_linear_decode(o, residual, layer.post_attention_layernorm_weight,
                                                         layer.post_attention_layernorm_variance_epsilon,
                                                         layer.gate_up_proj, fuse_silu_mul=True, out=cur["act"], h_out=cur["h"],
                                                         near_pull=near)
            if residual.shape[0] == 1:      # one sequence: the residual rides in the GEMV's bias slot (bf16(W.act) + residual,
                #                             rounded like the separate add: same bits, one launch less)
                near1 = None if near is None else self.kv_cache.near_pull_args(layer.layer_idx, 1)
                return tensor_op.linear_decode(act, layer.down_proj, bias=residual, out=cur["out"], near_pull=near1)
            return torch.add(residual, tensor_op.linear_decode(act, layer.down_proj, out=cur["down"]), out=cur["out"])
        hs = residual + F.linear(attn_output, layer.wo)
        residual = hs
        hs = tensor_op.layer_norm(hs, layer.post_attention_layernorm_variance_epsilon,
                                  layer.post_attention_layernorm_weight)
        hs = F.linear(hs, layer.gate_up_proj)
        d = hs.shape[-1] // 2
        act = torch.empty(hs.shape[:-1] + (d,), dtype=hs.dtype, device=hs.device)
        tensor_op.silu_and_mul(act, hs)
        return residual + F.linear(act, layer.down_proj)

    # --------------------------------------------------------------- decode branch of layer_compute (base.py:315-341)
    @torch.inference_mode()
    def layer_compute(self, layer, layer_idx, hidden_states, position_ids):
        residual = hidden_states
        bsz, q_len, _ = hidden_states.shape
        q, k, v = self.pre_attention_compute(hidden_states, layer)
        q, k = self.apply_rotary_pos_emb(q, k, position_ids, layer_idx=layer_idx)
        if self.query_hook is not None and not self._hook_applied:
            q = self.query_hook(layer_idx, q)
        cache = self.kv_cache
        cache.update_kv_cache(k, v, layer_idx)
        chunk_ids = cache.get_retrieval_position_ids(layer_idx=layer_idx, query_states=q)
        # base.py:326-338: V fetch under copy_stream || K rebuild on the current stream, joined before the attention.
        # (wait_stream makes a new event per call; two events of the model are recorded again and again instead)
        curr = torch.cuda.current_stream()
        side = cache.copy_stream
        if getattr(self, "_fork_ev", None) is None:
            self._fork_ev, self._join_ev = torch.cuda.Event(), torch.cuda.Event()
        self._fork_ev.record(curr)
        with torch.cuda.stream(side):
            side.wait_event(self._fork_ev)
            v_view = cache.get_value_cache(layer_idx, chunk_ids)
            self._join_ev.record(side)
        k_view = cache.get_key_cache(layer_idx=layer_idx, position_ids=chunk_ids, rope_func=None,
                                     cos_sin_cache=self.cos_sin_cache)
        curr.wait_event(self._join_ev)
        sc = getattr(self, "_scratch", None)
        attn = tensor_op.sparse_attention_decode(q, k_view, v_view, out=None if sc is None or q_len != 1 or sc["bs"] != bsz
                                                 else sc["sets"][sc["i"]]["attn"])
        return self.post_attention_compute(attn.reshape(bsz, q_len, self.hidden_size), residual, layer)

    # --------------------------------------------------------------- fused decode step (MI355X launch sequence)
    @torch.inference_mode()
    def forward_fused(self, token, pos, row_idx, kv_len=0, kv_len_dev=None, q_table=None, as_float=True):
        """Same computation as inference() for q_len == 1, with the small ops fused and the step's
        scalars in device memory (graph-capturable): 10 launches per layer at bs == 1 (in-place layout) instead of ~40.
          token [bs,1] int64, pos [bs,1] int64 (RoPE position), row_idx [1] int64 (cache row of the new K/V),
          kv_len / kv_len_dev: rows attended (= row_idx + 1), q_table: optional [L, bs, Hq, 1, D] synthetic queries.
        Per layer: [add+RMSNorm+QKV GEMV+split/RoPE/cache-push] -> select (3) -> (reference layout: stage hits) ->
        [K rebuild || V fetch, one launch] -> attention (2) -> O GEMV -> [add+RMSNorm+gate/up GEMV+SiLU*mul] -> down GEMV."""
        c = self.kv_cache
        x = F.embedding(token, self.embed_tokens)
        residual = None
        bs = x.shape[0]
        full = self.attn_mode == "full"
        c.incoming_q_len = 1
        for l, layer in enumerate(self.layers):
            kbuf = c.k_cache[l] if full else c.k_cache_buffer[l]
            vbuf = c.v_cache[l] if full else c.v_cache_buffer[l]
            residual, q = tensor_op.norm_qkv_rope_update(
                x, residual, layer.input_layernorm_weight, layer.input_layernorm_variance_epsilon, layer.wqkv,
                layer.bqkv, self.cos_sin_cache, pos, row_idx, kbuf, vbuf, self.num_heads,
                self.num_key_value_heads, q_override=None if q_table is None else q_table[l])
            if not full and self.chunk_layout == "inplace" and self.overlap_attention and c.can_overlap_attention():
                attn = c.select_fetch_attend_inplace(l, q, self.cos_sin_cache, kv_len=kv_len, kv_len_dev=kv_len_dev)
            else:
                slot_args = {}
                if not full and self.chunk_layout == "inplace":
                    c.select_fetch_inplace(l, q, self.cos_sin_cache)
                    slot_args = c.attend_slot_args()
                elif not full:
                    c.fetch_kv_follows = True            # (lets the selection publish an early-fetch list: fetch_kv consumes it)
                    try:
                        ids = c.get_retrieval_position_ids(layer_idx=l, query_states=q)
                    finally:
                        c.fetch_kv_follows = False
                    c.fetch_kv(l, ids, self.cos_sin_cache)
                attn = tensor_op.sparse_attention_decode(q, kbuf, vbuf, kv_len=kv_len, kv_len_dev=kv_len_dev, **slot_args)
            o = tensor_op.linear_decode(attn.reshape(bs, 1, self.hidden_size), layer.wo)
            # residual add + RMSNorm ride in the gate/up GEMV's block-cooperative prologue (one launch when bs == 1
            # and hidden == 4096, bit-identical to add_rmsnorm + GEMV; otherwise norm_linear_decode splits it)
            residual, act = tensor_op.norm_linear_decode(o, residual, layer.post_attention_layernorm_weight,
                                                         layer.post_attention_layernorm_variance_epsilon,
                                                         layer.gate_up_proj, fuse_silu_mul=True,
                                                         near_pull=None if full or bs != 1 else c.near_pull_args(l))
            x = tensor_op.linear_decode(act, layer.down_proj, near_pull=None if full or bs != 1 else c.near_pull_args(l, 1))
        V = self.lm_head.shape[0]
        rm = None
        if self.sampler_ranges and not as_float and x.is_cuda and tensor_op.range_max_supported(x.numel(), x.shape[-1], V):
            if self._range_max is None:
                self._range_max = torch.zeros(1, (V // 16 + 7) // 8 * 8, dtype=torch.int16, device=x.device)
            rm = self._range_max
        self._last_range_max = rm
        _, logits = tensor_op.norm_linear_decode(x, residual, self.norm_weight, self.norm_variance_epsilon, self.lm_head,
                                                 range_max=rm)
        return logits.float() if as_float else logits        # (the native sampler reads the bf16 row as it is)

    def get_ctx(self, input_ids):
        past = self.kv_cache.get_kv_len()
        n = input_ids.size(1)
        return torch.arange(past, past + n, device=self.device, dtype=torch.long).unsqueeze(0).repeat(input_ids.size(0), 1)

    @torch.inference_mode()
    def inference(self, input_ids, position_ids, as_float=True):
        hs = F.embedding(input_ids, self.embed_tokens)
        for idx in range(self.num_layers):
            hs = self.layer_compute(self.layers[idx], idx, hs, position_ids)
        if hs.shape[1] == 1 and hs.is_cuda:
            _, logits = tensor_op.norm_linear_decode(hs, None, self.norm_weight, self.norm_variance_epsilon, self.lm_head)
            return logits.float() if as_float else logits    # (bf16: tensor_op.sample_token then samples in one launch)
        hs = tensor_op.layer_norm(hs, self.norm_variance_epsilon, self.norm_weight)
        return F.linear(hs, self.lm_head).float()

    @torch.inference_mode()
    def decode_step(self, next_token, temperature=0.6, top_p=0.9, top_k=50, q_table=None, fused=True):
        """One iteration of the reference's timed loop (base.py:628-635).  fused=False runs the reference's
        exact call order through the reference-shaped methods (inference / layer_compute)."""
        self._last_range_max = None
        if not fused:
            logits = self.inference(input_ids=next_token, position_ids=self.get_ctx(next_token), as_float=False)
        else:
            c = self.kv_cache
            full = self.attn_mode == "full"
            row = c.kv_offset if full else c.sparse_end + c.gen_offset
            rows = (c.k_cache if full else c.k_cache_buffer).shape[-2]
            if row >= rows:
                raise RuntimeError(f"no cache row left for the new token (row {row} of {rows}): the generated-row slack "
                                   "is exhausted; the reference silently drops such tokens (kv_cache.py:1255-1265)")
            if c.kv_offset >= self.cos_sin_cache.shape[0]:
                raise RuntimeError(f"position {c.kv_offset} is past the RoPE table ({self.cos_sin_cache.shape[0]} rows = "
                                   "max_length + max_new_tokens)")
            pos = self.get_ctx(next_token)
            row_idx = torch.tensor([row], device=self.device, dtype=torch.long)
            logits = self.forward_fused(next_token, pos, row_idx, kv_len=row + 1, q_table=q_table, as_float=False)
            c.note_kv_appended(1)
        return tensor_op.sample_token(logits[:, -1, :], temperature=temperature, top_p=top_p, top_k=top_k,
                                      state=self._sampler_state,   # (draw counters of THIS model)
                                      range_max=self._last_range_max)


class Llama(DecoderLM):
    pass


class GLM(DecoderLM):
    def __init__(self, cfg=GLM_4_9B_1M, **kw):
        super().__init__(cfg=cfg, **kw)


# ------------------------------------------------------------------------------------------------
# synthetic long-context state (no 122K-token prefill offline): SURVEY.md section 8d
# ------------------------------------------------------------------------------------------------
@torch.inference_mode()
def build_synthetic_context(model, context_len, seed=1234, q_scale=0.25):
    """Fills the model's ShadowKVCache_CPU as a `context_len`-token prefill would, from synthetic
    tensors: per layer U ~ N(0,1) [L, r], SV ~ N(0, 1/r) [kv, D, r] (so the pre-RoPE keys K = U.SV^T are
    exactly rank r, unit variance), V ~ N(0,1); keys are rotated and handed with V to the cache's own
    prefill_kv_cache (landmarks, outliers, local rows, host V table, initial selection).  Returns the
    last-token query used for the initial selection."""
    cache, cfg = model.kv_cache, model.cfg
    dev, dt = model.device, model.dtype
    kv, D, r, L = cfg.num_key_value_heads, model.head_dim, cache.rank, context_len
    bs = model.batch_size
    pos = torch.arange(L, device=dev).unsqueeze(0).expand(bs, -1)
    cache.U = torch.zeros(model.num_layers, bs, L, r, device=dev, dtype=dt)
    cache.SV = torch.zeros(model.num_layers, bs, kv, D, r, device=dev, dtype=dt)
    q_last = None
    for l in range(model.num_layers):
        g = torch.Generator(device=dev).manual_seed(seed + l)
        U = torch.randn(bs, L, r, device=dev, generator=g).to(dt)
        SV = (torch.randn(bs, kv, D, r, device=dev, generator=g) / math.sqrt(r)).to(dt)
        cache.U[l].copy_(U)
        cache.SV[l].copy_(SV)
        k_pre = torch.einsum("blr,bhdr->bhld", U.float(), SV.float()).to(dt)            # [bs, kv, L, D]
        if cfg.rope_style == "neox":
            pid = pos.unsqueeze(1).expand(-1, kv, -1).contiguous()
            k_roped = tensor_op.apply_rotary_pos_emb_cuda(k_pre.contiguous(), model.cos_sin_cache, pid)
        else:
            k_roped = model._rope_glm(k_pre, pos)
        v = torch.randn(bs, kv, L, D, device=dev, generator=g).to(dt)
        q_last = torch.randn(bs, cfg.num_attention_heads, 1, D, device=dev, generator=g)
        q_last = (q_last / q_last.norm(dim=-1, keepdim=True) * math.sqrt(D) * q_scale * 4).to(dt)
        cache.prefill_kv_cache(v, l, k_roped, q_last)
        del U, SV, k_pre, k_roped, v
    cache.H2D()
    torch.cuda.synchronize(dev)
    return q_last


@torch.inference_mode()
def build_synthetic_context_full(model, context_len, seed=1234):
    """Full-attention baseline: fills the KV cache with `context_len` synthetic tokens (K, V ~ N(0,1) bf16)."""
    c = model.kv_cache
    for l in range(model.num_layers):
        g = torch.Generator(device=model.device).manual_seed(seed + l)
        for t in (c.k_cache, c.v_cache):
            for b in range(t.shape[1]):          # one sequence at a time: the f32 temporary stays at 0.5 GB whatever the batch
                t[l][b, :, :context_len].copy_(torch.randn(t[l][b, :, :context_len].shape, device=model.device,
                                                           generator=g).to(model.dtype))
    c.kv_offset = context_len
    torch.cuda.synchronize(model.device)


class QueryWalk:
    """Synthetic per-layer query trajectory: q_t = normalize(q_{t-1} + step*N(0,1)) * |q|,
    so consecutive selections overlap like a real model's; `step` tunes the chunk hit rate.  Used as
    DecoderLM.query_hook: the walk REPLACES the values of the model's post-RoPE query while keeping its
    data dependency (q_model * 0 + q_walk), because random-init weights carry no attention locality."""

    def __init__(self, model, step=0.3, q_scale=0.25, seed=99):
        dev = model.device
        self.g = torch.Generator(device=dev).manual_seed(seed)
        D = model.head_dim
        self.norm = math.sqrt(D) * q_scale * 4
        self.step = step
        self.q = torch.randn(model.num_layers, model.batch_size, model.num_heads, 1, D, device=dev, generator=self.g)
        self.q = self.q / self.q.norm(dim=-1, keepdim=True) * self.norm
        self.qb = self.q.to(model.dtype)
        self.qb_layers = self.qb.unbind(0)
        self._zero = torch.zeros((), device=dev, dtype=model.dtype)

    def advance(self):
        n = torch.randn(self.q.shape, device=self.q.device, generator=self.g)
        self.q = self.q + self.step * self.norm / math.sqrt(self.q.shape[-1]) * n
        self.q = self.q / self.q.norm(dim=-1, keepdim=True) * self.norm
        self.qb = self.q.to(self.qb.dtype)
        self.qb_layers = self.qb.unbind(0)

    def __call__(self, layer_idx, q_model):
        return torch.addcmul(self.qb_layers[layer_idx], q_model, self._zero)


class GraphDecoder:
    """One decode step (embedding -> all layers -> sampling) captured once into a hipGraph and replayed.

    Same operations, same kernels, same order and the same two-stream overlap as DecoderLM.decode_step;
    only what changes from step to step moves from host integers into device memory so the captured
    launch sequence stays valid:
        token     int64 [bs,1]   input token (the previous step's sample is written back into it)
        pos       int64 [bs,1]   RoPE position of the new token            (= kv_cache.kv_offset)
        gen       int64 [1]      tokens generated so far                    (= kv_cache.gen_offset)
        row_idx   int64 [1]      buffer row the new K/V go to              (= sparse_end + gen)
        kv_len    int32 [1]      rows attended                              (= sparse_end + gen + 1)
        step      int64 [1]      index into the synthetic query table (bench only)
    step() raises once the generated-row slack (buf_len - sparse_end rows, 96 at 122K) or the RoPE table is used up -
    the reference silently drops such tokens (kv_cache.py:1255-1265).  `ring_slack=True` (benchmarks only, recorded in
    the bench line) keeps stepping instead: the generated rows become a ring of the last `slack` tokens (row = gen %
    slack, kv_len stays at its maximum), so every step does at least the work of the last in-range step.
    At bs = 1 a decode step is ~25 launches per layer; eager PyTorch is host-bound on that
    (MI355X_MICROARCH.md "graph-replay-floor"), the graph removes the per-launch host cost."""

    NATIVE_SAMPLER_MAX_VOCAB = 4 * 131072     # skv_sample_topk_advance: rows are searched in parts of <= 131,072 logits

    def __init__(self, model, temperature=0.6, top_p=0.9, top_k=50, walk_table=None, seed=1234, ring_slack=False,
                 allow_torch_topk_capture=False):
        self.m = model
        self.allow_torch_topk_capture = bool(allow_torch_topk_capture)
        self._capturing = False
        self.ring_slack = bool(ring_slack)
        self.seed = int(seed)
        self.temperature, self.top_p, self.top_k = temperature, top_p, top_k
        c = model.kv_cache
        dev = model.device
        self.full = model.attn_mode == "full"
        if self.full:                      # rows [ctx, ctx + slack) receive the generated tokens, then wrap
            self.base, self.slack, gen0 = c.kv_offset, c.k_cache.shape[-2] - c.kv_offset, 0
        else:
            self.base, self.slack, gen0 = c.sparse_end, c.k_cache_buffer.shape[-2] - c.sparse_end, c.gen_offset
        self.token = torch.zeros(model.batch_size, 1, dtype=torch.long, device=dev)
        self.pos = torch.full((model.batch_size, 1), c.kv_offset, dtype=torch.long, device=dev)
        self.gen = torch.full((1,), gen0, dtype=torch.long, device=dev)
        self.gen_host = int(gen0)
        self.row_idx = self.gen % self.slack + self.base
        self.kv_len = (self.gen + 1).clamp(max=self.slack).add(self.base).to(torch.int32)
        self.step_idx = torch.zeros(1, dtype=torch.long, device=dev)
        self.walk_table = walk_table                     # [T, L, bs, Hq, 1, D] or None
        self._zero = torch.zeros((), device=dev, dtype=model.dtype)
        self.graph = None
        self.pre_step = None          # optional callable run at the start of every (captured) step - benchmarks
        # chunk hits of all steps so far (summed inside the step's last kernel from the per-layer hit counts)
        self.hit_accum = torch.zeros(1, dtype=torch.long, device=dev)

    def _sample(self, logits):
        """tensor_op.sample_token's pipeline (top-k 50 -> top-p 0.9 -> multinomial) on the k sorted survivors of
        torch.topk instead of a sort of the whole vocabulary.  Differs from the reference on rows with ties at the k-th
        value: torch.topk keeps exactly k, the reference's filter (tensor_op.py:253-255) and the native sampler keep
        every logit tied with the k-th.  No host-side checks: graph-capturable (see _body for the bs > 1 guard)."""
        if self.temperature == 0.0:
            return logits.argmax(dim=-1, keepdim=True)
        k = min(self.top_k, logits.size(-1)) if self.top_k > 0 else logits.size(-1)
        vals, idx = self._topk(logits / self.temperature, k)                  # sorted, descending
        probs = F.softmax(vals, dim=-1)
        if self.top_p > 0.0:
            rm = torch.cumsum(probs, dim=-1) > self.top_p
            rm = torch.cat((torch.zeros_like(rm[..., :1]), rm[..., :-1]), dim=-1)
            probs = F.softmax(vals.masked_fill(rm, float("-inf")), dim=-1)
        pick = torch.argmax(probs / torch.empty_like(probs).exponential_(1.0), dim=-1, keepdim=True)  # = multinomial(1)
        return idx.gather(-1, pick)

    @staticmethod
    def _topk(x, k, parts=64):
        """torch.topk(x, k, dim=-1) for rows the native sampler does not take (vocabulary > 524,288, k > 64, f32 logits).  Large
        rows always go through two levels of SHORT-slice top-k (top-k of each of `parts` slices, then top-k of the
        parts * k survivors - the global top-k is a subset of them): PyTorch's multi-block top-k over long slices faulted
        under hipGraph replay on this ROCm build (round 1: bs > 1 over 128,256 logits), short slices take its single-block
        path.  The row is padded with -inf to a multiple of `parts`, so no shape falls back to a long-slice call."""
        bs, V = x.shape
        if V < 16384 or V // parts < k:          # short rows: single-block path as they are
            return torch.topk(x, k, dim=-1)
        w = (V + parts - 1) // parts
        if w * parts != V:
            x = torch.nn.functional.pad(x, (0, w * parts - V), value=float("-inf"))
        v1, i1 = torch.topk(x.view(bs, parts, w), k, dim=-1)                   # [bs, parts, k]
        i1 = i1 + (torch.arange(parts, device=x.device) * w).view(1, parts, 1)
        v2, i2 = torch.topk(v1.reshape(bs, parts * k), k, dim=-1)
        return v2, i1.reshape(bs, parts * k).gather(-1, i2)

    def _hit_args(self):
        c = self.m.kv_cache
        if self.full:
            return 0, 0, 0
        return ptr(c._cnts_layers), c._cnts_layers.numel(), ptr(self.hit_accum)

    def _body(self):
        m, c = self.m, self.m.kv_cache
        if self.pre_step is not None:
            self.pre_step()
        qstep = None
        if self.walk_table is not None:
            qstep = torch.index_select(self.walk_table, 0, self.step_idx)[0]
        logits = m.forward_fused(self.token, self.pos, self.row_idx, kv_len=0, kv_len_dev=self.kv_len, q_table=qstep,
                                 as_float=False)
        last = logits[:, -1, :]
        V = last.size(-1)
        k = min(self.top_k, V) if self.top_k > 0 else V
        tlen = self.walk_table.shape[0] if self.walk_table is not None else 1
        if (self.temperature > 0.0 and k <= 64 and last.is_cuda and last.dtype == torch.bfloat16 and V % 8 == 0
                and V <= self.NATIVE_SAMPLER_MAX_VOCAB and last.stride(-1) == 1 and last.stride(0) % 8 == 0
                and last.data_ptr() % 16 == 0):
            # ONE native launch from the bf16 logits: exact top-k, temperature, top-p, draw, every counter of the step
            tail = (last.shape[0], k, float(self.temperature), float(self.top_p), self.seed, ptr(self.token), ptr(self.pos),
                    ptr(self.gen), ptr(self.row_idx), ptr(self.kv_len), ptr(self.step_idx) if self.walk_table is not None else 0,
                    self.base, self.slack, tlen, *self._hit_args(), current_stream_handle())
            rm = m._last_range_max
            if rm is not None and V // 16 >= k:      # the lm_head left the range maxima of THIS row: the row is not streamed
                check(lib().skv_sample_topk_advance_ranges(ptr(last), last.stride(0), V, ptr(rm), rm.stride(0), *tail),
                      "sample_topk_advance_ranges")
            else:
                check(lib().skv_sample_topk_advance(ptr(last), last.stride(0), V, *tail), "sample_topk_advance")
            return
        # Not taken by any BASELINE configuration (GLM-4's 151,552 logits are native since round 3).  torch.topk's
        # multi-block path faulted under hipGraph replay in round 1 (profiles/r02_graph_fault_record.txt): a captured
        # step with bs > 1 must not reach it unless the caller asks for it.
        if self.temperature > 0.0 and self._capturing and last.shape[0] > 1 and not self.allow_torch_topk_capture:
            raise RuntimeError("the native sampler does not take this logit row (vocabulary %d, top_k %d, dtype %s) and the "
                               "torch.topk fallback must not be captured at bs > 1 (GPU fault under hipGraph replay in round 1);"
                               " GraphDecoder(allow_torch_topk_capture=True) overrides" % (V, k, last.dtype))
        last = last.float()
        if self.temperature > 0.0 and k <= 64 and last.is_cuda:
            # top-k by torch, then ONE native launch: top-p filter, draw, and every device-side counter of the step
            vals, idx = self._topk(last / self.temperature, k)
            check(lib().skv_sample_advance(ptr(vals), ptr(idx), vals.shape[0], k, float(self.top_p), self.seed,
                                           ptr(self.token), ptr(self.pos), ptr(self.gen), ptr(self.row_idx),
                                           ptr(self.kv_len), ptr(self.step_idx) if self.walk_table is not None else 0,
                                           self.base, self.slack, tlen, *self._hit_args(), current_stream_handle()),
                  "sample_advance")
            return
        self.token.copy_(self._sample(last))
        # advance the device-side counters (same arithmetic as skv_sample_advance)
        self.pos.add_(1)
        self.gen.add_(1)
        self.row_idx.copy_(self.gen % self.slack + self.base)
        self.kv_len.copy_((self.gen + 1).clamp(max=self.slack).add(self.base).to(torch.int32))
        if self.walk_table is not None:
            self.step_idx.copy_((self.step_idx + 1) % self.walk_table.shape[0])
        if not self.full:
            self.hit_accum.add_(c._cnts_layers.sum())

    def _host_advance(self):
        c = self.m.kv_cache
        c.kv_offset += 1
        self.gen_host += 1
        if not self.full:
            c.gen_offset = min(self.gen_host, self.slack)

    def _check_room(self):
        c = self.m.kv_cache
        if self.gen_host >= self.slack and not self.ring_slack:
            raise RuntimeError(f"generated-row slack exhausted after {self.gen_host} tokens ({self.slack} rows behind the "
                               "sparse region); GraphDecoder(ring_slack=True) is the benchmark-only way past it")
        if c.kv_offset >= self.m.cos_sin_cache.shape[0]:
            raise RuntimeError(f"position {c.kv_offset} is past the RoPE table ({self.m.cos_sin_cache.shape[0]} rows = "
                               "max_length + max_new_tokens)")

    @torch.inference_mode()
    def capture(self, warmup=2):
        s = torch.cuda.Stream(device=self.m.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):          # eager: sizes workspaces, sets kernel attributes, warms hipBLASLt
                self._check_room()
                self._body()
                self._host_advance()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize(self.m.device)
        self.graph = torch.cuda.CUDAGraph()
        self._capturing = True
        try:
            with torch.cuda.graph(self.graph, stream=s):
                self._body()
        except Exception:
            self.graph = None
            raise
        finally:
            self._capturing = False
        return warmup

    @torch.inference_mode()
    def step(self):
        self._check_room()
        if self.graph is None:
            self._body()
        else:
            self.graph.replay()
        self._host_advance()
        return self.token


def make_walk_table(model, steps, step=0.3, q_scale=0.25, seed=99):
    """[steps, L, bs, Hq, 1, D] bf16: the QueryWalk trajectory, precomputed so a captured graph can index it."""
    w = QueryWalk(model, step=step, q_scale=q_scale, seed=seed)
    out = []
    for _ in range(steps):
        w.advance()
        out.append(w.qb.clone())
    return torch.stack(out)
