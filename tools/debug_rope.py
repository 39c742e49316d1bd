import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import vllm_paged_compression_amd.ops as ops
from oracle import pe_oracle as oracle
from test_oracle_golden import rope_cache
g = torch.Generator().manual_seed(1)
T, H, D = 16, 8, 128
cs = rope_cache(2048, 128, torch.bfloat16)
pos = torch.randint(0, 2048, (T,), generator=g)
q = (torch.rand(T, H * D, generator=g) * 2 - 1).to(torch.bfloat16)
k = (torch.rand(T, H * D, generator=g) * 2 - 1).to(torch.bfloat16)
ref = oracle.rotary_embedding(pos, q, D, cs, True)
qd, kd = q.cuda(), k.cuda()
ops.rotary_embedding(pos.cuda(), qd, kd, D, cs.cuda(), True)
bad = (qd.cpu() != ref).nonzero()
print("mismatches", bad.shape[0], "of", q.numel())
for t, e in bad[:8].tolist():
    h, d = divmod(e, D)
    i = d % 64
    x, y = q[t, h * D + i].float().item(), q[t, h * D + 64 + i].float().item()
    c, s = cs[pos[t], i].float().item(), cs[pos[t], 64 + i].float().item()
    print(t, h, d, "x", x, "y", y, "c", c, "s", s, "gpu", qd[t, e].item(), "ref", ref[t, e].item(),
          "exact", x * c - y * s if d < 64 else y * c + x * s)
