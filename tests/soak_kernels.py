"""Randomised soak of the attention kernels against the oracle:  python tests/soak_kernels.py <first seed> <end seed>

Per seed: one decode-attention case (random head geometry, dtype, sequence lengths with the
block / tile boundaries over-represented, random block tables, optionally ALiBi, a random forced
split count), the same case through the fused decode step (KV write inside the launch; cache
bytes must equal the separate write), and one prefill case (random ragged prompts, kv lengths
shorter than the prompts for some requests, every tile shape the launcher can pick), and one
rotating decode step (rotation inside the launch against rotation in front of it).  Tolerances
are the ones of tests/test_gpu_kernels.py.  Not collected by pytest (long-running driver)."""
import os
import random
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from oracle import pe_oracle as oracle  # noqa: E402

oracle.build()
import vllm_paged_compression_amd.ops as ops  # noqa: E402

ATOL = {torch.float16: 1e-3, torch.bfloat16: 1e-3, torch.float32: 1e-4}
RTOL = {torch.float16: 1e-2, torch.bfloat16: 1.6e-2, torch.float32: 1e-4}
EDGE = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513]


def pick_len(rng, hi):
    if rng.random() < 0.5:
        return min(rng.choice(EDGE), hi)
    return rng.randint(1, hi)


def decode_case(seed):
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    D = rng.choice([64, 128, 128, 256, 80, 96, 32, 112, 120, 192])   # vllm/attention/ops/paged_attn.py:37
    Hkv = rng.choice([1, 2, 4, 8])
    Hq = Hkv * rng.choice([1, 2, 4, 8, 16])
    dtype = rng.choice([torch.bfloat16, torch.float16])
    bs = rng.choice([16, 16, 8, 32])      # the reference's page sizes (scripts/diff_page_sizes.sh:16)
    S = rng.randint(1, 12)
    hi = rng.choice([80, 600, 2100])
    lens = [pick_len(rng, hi) for _ in range(S)]
    nb = sum((n + bs - 1) // bs for n in lens) + 4
    sc = D ** -0.5
    x = 8
    # data scale of the reference's kernel tests, which their atol of 1e-3 is set for:
    # uniform(-head_size^-0.5, head_size^-0.5) (vllm/utils.py:627-651, tests/kernels/test_attention.py:158)
    kc = ((torch.rand(nb, Hkv, D // x, bs, x, generator=g) * 2 - 1) * sc).to(dtype)
    vc = ((torch.rand(nb, Hkv, D, bs, generator=g) * 2 - 1) * sc).to(dtype)
    q = ((torch.rand(S, Hq, D, generator=g) * 2 - 1) * sc).to(dtype)
    # disjoint tables (a fused step stores into the last block of every row)
    perm = torch.randperm(nb, generator=g).tolist()
    M = max((n + bs - 1) // bs for n in lens) + rng.randint(0, 2)
    bt = torch.zeros(S, M, dtype=torch.int32)
    for i, n in enumerate(lens):
        for j in range((n + bs - 1) // bs):
            bt[i, j] = perm.pop()
    lens_t = torch.tensor(lens, dtype=torch.int32)
    slopes = (torch.randn(Hq, generator=g) * 0.1) if rng.random() < 0.3 else None
    splits = rng.choice([0, 0, 1, 2, 3, 5])
    if splits:
        os.environ["PE_DECODE_SPLITS"] = str(splits)
    else:
        os.environ.pop("PE_DECODE_SPLITS", None)
    ops.reload_env()
    ref = oracle.paged_attention(q, kc, vc, bt, lens_t, sc, slopes)
    out = torch.empty_like(q, device="cuda")
    kcg, vcg = kc.cuda(), vc.cuda()
    ops.paged_attention_v1(out, q.cuda(), kcg, vcg, Hkv, sc, bt.cuda(), lens_t.cuda(), bs, max(lens),
                           None if slopes is None else slopes.cuda(), "auto", 1.0, 1.0)
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=ATOL[dtype], rtol=RTOL[dtype])
    # the same step with the last token's K/V arriving as rows: fused launch vs write + attention
    if (D % 32 == 0 or bs == 16) and Hq // Hkv <= 16:   # the one-launch (MFMA) path
        k_new = ((torch.rand(S, Hkv, D, generator=g) * 2 - 1) * sc).to(dtype).cuda()
        v_new = ((torch.rand(S, Hkv, D, generator=g) * 2 - 1) * sc).to(dtype).cuda()
        slots = torch.tensor([int(bt[i, (n - 1) // bs]) * bs + (n - 1) % bs for i, n in enumerate(lens)],
                             dtype=torch.int64).cuda()
        kc1, vc1 = kcg.clone(), vcg.clone()
        ops.reshape_and_cache(k_new, v_new, kc1, vc1, slots)
        o1 = torch.empty_like(out)
        ops.paged_attention_v1(o1, q.cuda(), kc1, vc1, Hkv, sc, bt.cuda(), lens_t.cuda(), bs, max(lens),
                               None if slopes is None else slopes.cuda(), "auto", 1.0, 1.0)
        kc2, vc2 = kcg.clone(), vcg.clone()
        o2 = torch.empty_like(out)
        ops.decode_step(o2, q.cuda(), k_new, v_new, kc2, vc2, slots, Hkv, sc, bt.cuda(), lens_t.cuda(), bs,
                        max(lens), None if slopes is None else slopes.cuda())
        assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2), "fused KV write differs"
        torch.testing.assert_close(o2.float(), o1.float(), atol=ATOL[dtype], rtol=RTOL[dtype])
    os.environ.pop("PE_DECODE_SPLITS", None)
    ops.reload_env()
    return f"decode Hq{Hq} Hkv{Hkv} D{D} bs{bs} {str(dtype)[6:]} S{S} max{max(lens)} splits{splits} alibi{slopes is not None}"


def prefill_case(seed):
    rng = random.Random(seed * 7 + 1)
    g = torch.Generator().manual_seed(seed * 7 + 1)
    D = rng.choice([64, 128, 128, 96, 32, 80, 112, 120, 192, 256])
    Hkv = rng.choice([1, 2, 8])
    Hq = Hkv * rng.choice([1, 4, 8])
    dtype = rng.choice([torch.bfloat16, torch.float16])
    n = rng.randint(1, 5)
    hi = rng.choice([70, 300, 700])
    q_lens = [pick_len(rng, hi) for _ in range(n)]
    kv_lens = [L if rng.random() < 0.6 else rng.randint(1, L) for L in q_lens]
    sc = D ** -0.5
    q = (torch.rand(sum(q_lens), Hq, D, generator=g) * 2 - 1).to(dtype)
    k = (torch.rand(sum(kv_lens), Hkv, D, generator=g) * 2 - 1).to(dtype)
    v = (torch.rand(sum(kv_lens), Hkv, D, generator=g) * 2 - 1).to(dtype)
    cq = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), dtype=torch.int32).cuda()
    ck = torch.tensor([0] + list(torch.tensor(kv_lens).cumsum(0)), dtype=torch.int32).cuda()
    ref = oracle.prefill_attention(q.float(), k.float(), v.float(), q_lens, kv_lens, sc)
    tol = {torch.bfloat16: (2e-2, 1.6e-2), torch.float16: (3e-3, 1e-2)}[dtype]
    outs = []
    # max_q_len is an upper bound: 513 / 2048 move the same data through the other tile shapes
    for m in sorted({max(q_lens), max(513, max(q_lens)), 2048}):
        out = ops.prefill_causal(q.cuda(), k.cuda(), v.cuda(), cq, ck, m, sc)
        torch.testing.assert_close(out.cpu().float(), ref, atol=tol[0], rtol=tol[1])
        outs.append(out)
    if D <= 128:
        assert all(torch.equal(outs[0], o) for o in outs[1:]), "tile shapes differ in bits"
    return f"prefill Hq{Hq} Hkv{Hkv} D{D} {str(dtype)[6:]} q{q_lens} kv{kv_lens}"


def rope_case(seed):
    """The rotating decode step (rotation + KV write + attention in one launch) against rotation in
    its own launch followed by the plain step (bit for bit), and the cache bytes against the
    oracle's rotation + write."""
    from test_oracle_golden import rope_cache
    rng = random.Random(seed * 13 + 5)
    g = torch.Generator().manual_seed(seed * 13 + 5)
    D = rng.choice([64, 128])
    Hkv = rng.choice([1, 2, 8])
    Hq = Hkv * rng.choice([1, 2, 4, 8])
    dtype = rng.choice([torch.bfloat16, torch.float16])
    bs = rng.choice([16, 8, 32])
    is_neox = rng.random() < 0.6
    rot = rng.choice([D, 64]) if is_neox else rng.choice([D, D // 2, 16, 48 if D == 64 else 80])
    S = rng.randint(1, 9)
    hi = rng.choice([80, 600, 1500])
    lens = [pick_len(rng, hi) for _ in range(S)]
    nb = sum((n + bs - 1) // bs for n in lens) + 4
    sc = D ** -0.5
    max_pos = 2048
    cs = rope_cache(max_pos, rot, dtype)
    pos = torch.randint(0, max_pos, (S,), generator=g)
    kc = ((torch.rand(nb, Hkv, D // 8, bs, 8, generator=g) * 2 - 1) * sc).to(dtype)
    vc = ((torch.rand(nb, Hkv, D, bs, generator=g) * 2 - 1) * sc).to(dtype)
    qkv = ((torch.rand(S, (Hq + 2 * Hkv) * D, generator=g) * 2 - 1) * 4 * sc).to(dtype)
    perm = torch.randperm(nb, generator=g).tolist()
    M = max((n + bs - 1) // bs for n in lens) + rng.randint(0, 2)
    bt = torch.zeros(S, M, dtype=torch.int32)
    for i, n in enumerate(lens):
        for j in range((n + bs - 1) // bs):
            bt[i, j] = perm.pop()
    slots = torch.tensor([int(bt[i, (n - 1) // bs]) * bs + (n - 1) % bs for i, n in enumerate(lens)],
                         dtype=torch.int64)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    splits = rng.choice([0, 0, 2, 3])
    if splits:
        os.environ["PE_DECODE_SPLITS"] = str(splits)
    else:
        os.environ.pop("PE_DECODE_SPLITS", None)
    ops.reload_env()
    assert ops.decode_step_rope_supported(Hq, Hkv, D, bs, dtype, rot, is_neox)

    def views(t):
        return (x.reshape(S, -1, D) for x in t.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
    qkv1 = qkv.cuda()
    q1, k1, v1 = views(qkv1)
    kc1, vc1 = kc.cuda(), vc.cuda()
    o1 = torch.empty(S, Hq, D, dtype=dtype, device="cuda")
    ops.decode_step_rope(o1, pos.cuda(), q1, k1, v1, kc1, vc1, slots.cuda(), Hkv, sc, bt.cuda(), lens_t.cuda(),
                         bs, max(lens), cs.cuda(), is_neox)
    assert torch.equal(qkv1.cpu(), qkv), "inputs were rotated in place"
    qkv2 = qkv.cuda()
    q2, k2, v2 = views(qkv2)
    kc2, vc2 = kc.cuda(), vc.cuda()
    ops.rotary_embedding(pos.cuda(), q2, k2, D, cs.cuda(), is_neox)
    o2 = torch.empty_like(o1)
    ops.decode_step(o2, q2, k2, v2, kc2, vc2, slots.cuda(), Hkv, sc, bt.cuda(), lens_t.cuda(), bs, max(lens))
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2), "cache bytes differ from the two-launch form"
    assert torch.equal(o1, o2), "output differs from the two-launch form"
    q, k, v = views(qkv)
    k_ref = oracle.rotary_embedding(pos, k.reshape(S, -1), D, cs, is_neox).reshape(S, Hkv, D)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    oracle.reshape_and_cache(k_ref, v.contiguous(), kc_ref, vc_ref, slots)
    assert torch.equal(kc1.cpu(), kc_ref) and torch.equal(vc1.cpu(), vc_ref), "cache bytes differ from the oracle"
    os.environ.pop("PE_DECODE_SPLITS", None)
    ops.reload_env()
    return f"rope Hq{Hq} Hkv{Hkv} D{D} bs{bs} rot{rot} neox{is_neox} {str(dtype)[6:]} S{S} max{max(lens)} splits{splits}"


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    t0 = time.time()
    fails = 0
    for seed in range(lo, hi):
        for fn in (decode_case, prefill_case, rope_case):
            try:
                desc = fn(seed)
            except Exception as e:     # noqa: BLE001
                fails += 1
                print(f"FAIL seed {seed} {fn.__name__}: {str(e)[:600]}", flush=True)
                if fails > 10:
                    sys.exit(1)
                continue
            if seed % 10 == 0:
                print(f"seed {seed} ok ({desc}), {time.time() - t0:.0f}s", flush=True)
    print("seeds", hi - lo, "fails", fails)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
