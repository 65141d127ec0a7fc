import sys, math, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden"); sys.path.insert(0, ".")
import gen_inputs as G
from test_gpu_kv_cache import _build, DEV
from shadowkv_amd import tensor_op
case = "llama_cpu_b1024"
a, c, inp = _build(case); b, _, _ = _build(case)
cs = inp["cos_sin"].to(DEV)
for t in range(3):
    qd = inp["q_steps"][t].to(DEV)
    rows = a.sparse_end
    a.select_fetch_inplace(0, qd, cs)
    o_ref = tensor_op.sparse_attention_decode(qd, a.k_cache_buffer[0], a.v_cache_buffer[0], kv_len=rows)
    o_new = b.select_fetch_attend_inplace(0, qd, cs, kv_len=rows)
    torch.cuda.synchronize()
    d = (o_new.float() - o_ref.float()).abs().view(32, 128).max(dim=1).values
    print("step", t, "cnts", a.cnts.tolist(), "max diff per head", [round(float(x), 4) for x in d])
    print("  ref mean abs", float(o_ref.float().abs().mean()), "new mean abs", float(o_new.float().abs().mean()))
