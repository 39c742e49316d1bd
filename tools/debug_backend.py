import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pe_oracle as oracle
import vllm_paged_compression_amd.ops as ops
g = torch.Generator().manual_seed(1)
Hq, Hkv, D, bs, nb = 8, 2, 128, 16, 96
sc = D ** -0.5
x = 8
kc = ((torch.rand(nb, Hkv, D // x, bs, x, generator=g) * 2 - 1) * sc * 3).to(torch.bfloat16)
vc = ((torch.rand(nb, Hkv, D, bs, generator=g) * 2 - 1)).to(torch.bfloat16)
q = ((torch.rand(3, Hq, D, generator=g) * 2 - 1) * sc * 3).to(torch.bfloat16)
bt = torch.zeros(3, 13, dtype=torch.int32)
bt[0] = torch.randperm(nb, generator=g)[:13]
bt[1, :9] = torch.randperm(nb, generator=g)[:9]
bt[2, :7] = torch.randperm(nb, generator=g)[:7]
L = torch.tensor([208, 138, 98], dtype=torch.int32)
kv = torch.tensor([128, 138, 98], dtype=torch.int32)
rmv, out_t, _ = ops.evict_select_compact(bt.cuda(), L.cuda(), L.cuda(), "streamingLLM", 128, bs)
exp_t = oracle.compact_block_tables(bt, [1, -1, -1])
print("rmv", rmv.cpu().tolist(), "tables equal", torch.equal(out_t.cpu(), exp_t))
print(out_t.cpu()[1].tolist(), exp_t[1].tolist())
ref = oracle.paged_attention(q, kc, vc, exp_t, kv, sc)
out = torch.empty_like(q, device="cuda")
ops.paged_attention_v1(out, q.cuda(), kc.cuda(), vc.cuda(), Hkv, sc, out_t, kv.cuda(), bs, 208, None, "auto", 1.0, 1.0)
print("maxdiff per seq", (out.cpu().float() - ref.float()).abs().amax(dim=(1, 2)).tolist())
