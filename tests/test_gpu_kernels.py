"""GPU parity tests: the HIP kernels (called through the C ABI via the torch-level ops)
against the CPU oracle on the same seeded inputs, and against the golden vectors made
from the reference's own code.  Bit-exact for byte/integer work; floating-point
tolerances are the ones the reference's kernel tests use."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, from_bits

pytestmark = pytest.mark.gpu

# tolerance of the reference's paged-attention test on ROCm
# (tests/kernels/test_attention.py:323-331 with tests/kernels/allclose_default.py:5-10)
ATOL = {torch.bfloat16: 1e-3, torch.float16: 1e-3, torch.float32: 1e-5}
RTOL = {torch.bfloat16: 1.6e-2, torch.float16: 1e-3, torch.float32: 1.3e-6}


@pytest.fixture(scope="module")
def ops():
    import vllm_paged_compression_amd.ops as ops
    import vllm_paged_compression_amd as pkg
    pkg.load_library()
    return ops


def make_cache(nb, H, D, bs, dtype, g, scale=None):
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    scale = D ** -0.5 if scale is None else scale
    kc = ((torch.rand(nb, H, D // x, bs, x, generator=g) * 2 - 1) * scale).to(dtype)
    vc = ((torch.rand(nb, H, D, bs, generator=g) * 2 - 1) * scale).to(dtype)
    return kc, vc


# ------------------------------------------------------------------ KV write
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("H,D,bs", [(8, 128, 16), (12, 64, 16), (2, 80, 16), (8, 128, 32), (4, 64, 8)])
@pytest.mark.parametrize("pattern", ["random", "prefill", "decode"])
def test_reshape_and_cache(ops, oracle, dtype, H, D, bs, pattern):
    g = torch.Generator().manual_seed(H * 7 + D + bs + len(pattern))
    nb = 64
    if pattern == "random":
        T = 57
        slots = torch.randperm(nb * bs, generator=g)[:T].to(torch.int64)
        slots[7] = -1
    elif pattern == "prefill":
        # consecutive slots of whole blocks + a partial last block, as _compute_slot_mapping builds
        blocks = torch.randperm(nb, generator=g)[:7].tolist()
        T = 6 * bs + 5
        slots = torch.tensor(oracle.prompt_slot_mapping(blocks, T, bs), dtype=torch.int64)
    else:
        T = 9
        slots = (torch.randperm(nb, generator=g)[:T] * bs + torch.randint(0, bs, (T,), generator=g)).to(torch.int64)
    k = (torch.rand(T, H, D, generator=g) - 0.5).to(dtype)
    v = (torch.rand(T, H, D, generator=g) - 0.5).to(dtype)
    kc, vc = make_cache(nb, H, D, bs, dtype, g)
    kc_d, vc_d = kc.cuda(), vc.cuda()
    oracle.reshape_and_cache(k, v, kc, vc, slots)
    ops.reshape_and_cache(k.cuda(), v.cuda(), kc_d, vc_d, slots.cuda(), "auto", 1.0, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(kc_d.cpu(), kc) and torch.equal(vc_d.cpu(), vc)


def test_reshape_and_cache_strided_and_empty(ops, oracle):
    g = torch.Generator().manual_seed(5)
    nb, H, D, bs, T = 16, 8, 128, 16, 33
    # key/value as slices of a fused qkv activation (token stride > H*D)
    qkv = (torch.rand(T, (32 + 2 * H) * D, generator=g) - 0.5).to(torch.bfloat16)
    k = qkv[:, 32 * D:(32 + H) * D].view(T, H, D)
    v = qkv[:, (32 + H) * D:].view(T, H, D)
    slots = torch.randperm(nb * bs, generator=g)[:T].to(torch.int64)
    kc, vc = make_cache(nb, H, D, bs, torch.bfloat16, g)
    kc_d, vc_d = kc.cuda(), vc.cuda()
    qkv_d = qkv.cuda()
    kd = qkv_d[:, 32 * D:(32 + H) * D].view(T, H, D)
    vd = qkv_d[:, (32 + H) * D:].view(T, H, D)
    oracle.reshape_and_cache(k, v, kc, vc, slots)
    ops.reshape_and_cache(kd, vd, kc_d, vc_d, slots.cuda())
    assert torch.equal(kc_d.cpu(), kc) and torch.equal(vc_d.cpu(), vc)
    # zero tokens: nothing happens
    ops.reshape_and_cache(kd[:0], vd[:0], kc_d, vc_d, slots[:0].cuda())
    assert torch.equal(kc_d.cpu(), kc)


def test_reshape_and_cache_gather_and_zero(ops, oracle):
    """Fused prompt-prune gather (streamingLLM keep index) and evicted-slot zeroing."""
    g = torch.Generator().manual_seed(6)
    nb, H, D, bs = 80, 8, 128, 16
    L, budget = 1030, 512
    k = (torch.rand(L, H, D, generator=g) - 0.5).to(torch.bfloat16)
    v = (torch.rand(L, H, D, generator=g) - 0.5).to(torch.bfloat16)
    keep = oracle.streaming_keep_indices(L, budget, bs)
    n_blocks, n_tok = oracle.num_required_blocks_after_prune(L, budget, bs)
    assert len(keep) == n_tok
    blocks = torch.randperm(nb, generator=g)[:n_blocks].tolist()
    slots = torch.tensor(oracle.prompt_slot_mapping(blocks, L, bs), dtype=torch.int64)
    assert slots.numel() == n_tok
    kc, vc = make_cache(nb, H, D, bs, torch.bfloat16, g)
    kc_d, vc_d = kc.cuda(), vc.cuda()
    pk, pv = oracle.prompt_prune(k, v, "streamingLLM", budget, bs)
    oracle.reshape_and_cache(pk, pv, kc, vc, slots)
    zero = torch.tensor([blocks[1] * bs + 3, -1, blocks[2] * bs + 15], dtype=torch.int64)
    zk = torch.zeros(3, H, D, dtype=torch.bfloat16)
    oracle.reshape_and_cache(zk, zk, kc, vc, zero)
    cu_q = torch.tensor([0, L], dtype=torch.int32).cuda()
    cu_kept = torch.tensor([0, n_tok], dtype=torch.int32).cuda()
    idx = ops.prompt_keep_index_streaming(cu_q, cu_kept, n_tok, budget, bs)
    assert idx.cpu().tolist() == keep
    ops.reshape_and_cache(k.cuda(), v.cuda(), kc_d, vc_d, slots.cuda(), src_index=idx,
                          zero_slots=zero.cuda())
    assert torch.equal(kc_d.cpu(), kc) and torch.equal(vc_d.cpu(), vc)
    # dense form of the same gather
    dk, dv = ops.prompt_prune_gather(k.cuda(), v.cuda(), idx)
    assert torch.equal(dk.cpu(), pk) and torch.equal(dv.cpu(), pv)


def test_keep_index_streaming_batch(ops, oracle):
    bs, budget = 16, 96
    lens = [40, 96, 97, 200, 333, 112]
    kept = [oracle.pruned_length(L, budget, bs) for L in lens]
    cu_q = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(np.cumsum(kept)), dtype=torch.int32)
    idx = ops.prompt_keep_index_streaming(cu_q.cuda(), cu_k.cuda(), int(cu_k[-1]), budget, bs).cpu().tolist()
    expect = []
    for r, L in enumerate(lens):
        expect += [int(cu_q[r]) + i for i in oracle.streaming_keep_indices(L, budget, bs)]
    assert idx == expect


def test_golden_ref_cpu_kernels(ops):
    """KV write exact, decode attention within tolerance of the reference's CPU kernels."""
    z = np.load(os.path.join(GOLDEN, "ref_cpu_kernels.npz"))
    n = len([k for k in z.files if k.endswith("_meta")])
    for c in range(n):
        Hq, Hkv, D, bs, nb, dt = [int(v) for v in z[f"c{c}_meta"]]
        x = 16 // (4 if dt == 0 else 2)
        k, v = from_bits(z[f"c{c}_k"], dt), from_bits(z[f"c{c}_v"], dt)
        kc = torch.zeros(nb, Hkv, D // x, bs, x, dtype=k.dtype, device="cuda")
        vc = torch.zeros(nb, Hkv, D, bs, dtype=k.dtype, device="cuda")
        ops.reshape_and_cache(k.cuda(), v.cuda(), kc, vc, torch.from_numpy(z[f"c{c}_slots"]).cuda())
        assert torch.equal(kc.cpu(), from_bits(z[f"c{c}_kc_out"], dt))
        assert torch.equal(vc.cpu(), from_bits(z[f"c{c}_vc_out"], dt))
        q = from_bits(z[f"c{c}_q"], dt).cuda()
        out = torch.empty_like(q)
        lens = torch.from_numpy(z[f"c{c}_lens"]).cuda()
        bt = torch.from_numpy(z[f"c{c}_bt"]).cuda()
        ops.paged_attention_v1(out, q, kc, vc, Hkv, D ** -0.5, bt, lens, bs, 90, None, "auto", 1.0, 1.0)
        ref = from_bits(z[f"c{c}_o"], dt)
        torch.testing.assert_close(out.cpu().float(), ref.float(), atol=1e-3, rtol=1.6e-2)


# ------------------------------------------------------------------ decode attention
DECODE_CASES = [
    # Hq, Hkv, D, bs, dtype, lens
    (32, 8, 128, 16, torch.bfloat16, [1, 15, 16, 17, 31, 32, 33, 300, 777, 784]),
    (32, 8, 128, 16, torch.float16, [768, 769, 783, 784, 5]),
    (12, 12, 64, 16, torch.float16, [1024, 1100, 1124, 7]),
    (8, 1, 128, 16, torch.bfloat16, [3072, 100, 3088]),
    (64, 8, 128, 16, torch.bfloat16, [200, 40]),
    (16, 1, 256, 16, torch.bfloat16, [130, 33]),
    (40, 40, 80, 16, torch.bfloat16, [77, 300]),      # head size 80: last k-step half empty
    (32, 8, 128, 32, torch.bfloat16, [100, 515]),     # 32-token blocks
    (8, 2, 64, 8, torch.float16, [9, 130]),           # 8-token blocks
    (32, 8, 128, 8, torch.bfloat16, [1, 7, 8, 9, 15, 16, 17, 31, 32, 33, 300, 777]),
    (32, 8, 64, 32, torch.float16, [1, 31, 32, 33, 63, 64, 65, 1000]),
    (8, 1, 128, 32, torch.bfloat16, [3072, 100, 3088]),
    # Llama-3.2-3B (24 query heads on 8 KV heads) and Llama-3.2-1B (head size 64), the other two
    # models of the reference's scripts (scripts/latency_test.sh:8), at their three page sizes
    (24, 8, 128, 16, torch.bfloat16, [777, 784, 15]), (24, 8, 128, 8, torch.bfloat16, [777, 9]),
    (24, 8, 128, 32, torch.bfloat16, [777, 33]), (32, 8, 64, 16, torch.bfloat16, [1040, 17]),
    (32, 8, 64, 8, torch.bfloat16, [1040, 17]),
    (16, 1, 256, 32, torch.bfloat16, [130, 33]),      # head size 256 with 32-token blocks
    (16, 2, 80, 32, torch.bfloat16, [130, 33]),       # generic path (head size 80 with 32-token blocks)
    (8, 2, 128, 16, torch.float32, [50, 333]),        # generic path (f32)
    (34, 2, 128, 16, torch.bfloat16, [70, 200]),      # 17 query heads per KV head -> generic
    # the remaining head sizes of the reference's sweep (tests/kernels/test_attention.py:30)
    (8, 2, 96, 16, torch.bfloat16, [45, 260]),
    (8, 8, 112, 16, torch.float16, [129, 16]),
    (6, 2, 120, 16, torch.bfloat16, [200, 31]),
    (8, 2, 192, 16, torch.bfloat16, [90, 513]),
    (8, 2, 192, 32, torch.float16, [65, 300]),        # head size 192 with 32-token blocks
    # 8- and 32-token blocks at the other head sizes made of whole k-steps
    (8, 8, 32, 8, torch.bfloat16, [1, 7, 8, 9, 33, 1500]), (8, 2, 32, 32, torch.float16, [31, 32, 33, 700]),
    (16, 2, 96, 8, torch.bfloat16, [1, 15, 16, 17, 2049]), (12, 4, 96, 32, torch.float16, [64, 65, 900]),
    (32, 8, 192, 8, torch.bfloat16, [9, 777, 2500]), (16, 2, 256, 8, torch.float16, [8, 9, 130, 1100]),
    (16, 2, 256, 32, torch.bfloat16, [1, 33, 2000]),
    # every head size of the reference's op (vllm/attention/ops/paged_attn.py:37) on the matrix-core
    # kernel, lengths on and around the 32-token tile and long enough for several KV splits
    (8, 8, 32, 16, torch.bfloat16, [1, 31, 32, 33, 1500]),
    (16, 4, 80, 16, torch.float16, [1, 16, 17, 33, 2100]),
    (64, 8, 80, 16, torch.bfloat16, [120, 4000]),      # test_attention.py:36 (64, 8) head geometry
    (16, 2, 96, 16, torch.bfloat16, [1, 32, 47, 2049, 784]),
    (16, 2, 112, 16, torch.bfloat16, [15, 16, 64, 1999]),
    (40, 40, 120, 16, torch.float16, [1, 33, 500]),
    (16, 2, 120, 16, torch.bfloat16, [31, 3000]),
    (32, 8, 192, 16, torch.bfloat16, [1, 17, 32, 777, 2500]),
    (16, 16, 192, 16, torch.float16, [40, 1030]),
]


@pytest.mark.parametrize("Hq,Hkv,D,bs,dtype,lens", DECODE_CASES)
@pytest.mark.parametrize("alibi", [False, True])
def test_paged_decode(ops, oracle, Hq, Hkv, D, bs, dtype, lens, alibi):
    g = torch.Generator().manual_seed(Hq * 131 + D + len(lens))
    nb = 300
    sc = D ** -0.5
    kc, vc = make_cache(nb, Hkv, D, bs, dtype, g)
    S = len(lens)
    q = ((torch.rand(S, Hq, D, generator=g) * 2 - 1) * sc).to(dtype)
    M = (max(lens) + bs - 1) // bs + 1
    bt = torch.randint(0, nb, (S, M), generator=g, dtype=torch.int32)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    slopes = torch.randn(Hq, generator=g) * 0.1 if alibi else None
    ref = oracle.paged_attention(q, kc, vc, bt, lens_t, sc, slopes)
    out = torch.empty_like(q, device="cuda")
    ops.paged_attention_v1(out, q.cuda(), kc.cuda(), vc.cuda(), Hkv, sc, bt.cuda(), lens_t.cuda(), bs,
                           max(lens), None if slopes is None else slopes.cuda(), "auto", 1.0, 1.0)
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=ATOL[dtype], rtol=RTOL[dtype])


@pytest.mark.parametrize("splits", [1, 2, 3, 7])
def test_paged_decode_split_kv(ops, oracle, splits, monkeypatch):
    """Forced split counts (the partial-merge path) give the same result."""
    monkeypatch.setenv("PE_DECODE_SPLITS", str(splits))
    g = torch.Generator().manual_seed(splits)
    Hq, Hkv, D, bs, nb = 32, 8, 128, 16, 512
    lens = [4000, 1, 2049, 33]
    sc = D ** -0.5
    kc, vc = make_cache(nb, Hkv, D, bs, torch.bfloat16, g)
    q = ((torch.rand(len(lens), Hq, D, generator=g) * 2 - 1) * sc).to(torch.bfloat16)
    M = 4000 // bs
    bt = torch.randint(0, nb, (len(lens), M), generator=g, dtype=torch.int32)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    ref = oracle.paged_attention(q, kc, vc, bt, lens_t, sc)
    out = torch.empty_like(q, device="cuda")
    tmp = torch.empty(1, device="cuda")
    ops.paged_attention_v2(out, tmp, tmp, tmp, q.cuda(), kc.cuda(), vc.cuda(), Hkv, sc, bt.cuda(),
                           lens_t.cuda(), bs, 4000, None, "auto", 1.0, 1.0)
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=1e-3, rtol=1.6e-2)


def test_paged_decode_forced_branches(ops, oracle):
    """Inputs that force the online-softmax rescale: one key far above the rest, placed late,
    and a sequence whose only large logits sit in the masked tail of the last block."""
    g = torch.Generator().manual_seed(11)
    Hq, Hkv, D, bs, nb = 32, 8, 128, 16, 64
    kc, vc = make_cache(nb, Hkv, D, bs, torch.bfloat16, g)
    q = ((torch.rand(2, Hq, D, generator=g) * 2 - 1) * 0.3).to(torch.bfloat16)
    lens = torch.tensor([700, 35], dtype=torch.int32)
    bt = torch.arange(0, 2 * 44, dtype=torch.int32).reshape(2, 44) % nb
    # spike: token 650 of sequence 0 aligned with q[0, head 5]
    blk, off = int(bt[0, 650 // bs]), 650 % bs
    kc[blk, 1, :, off, :] = (q[0, 5].float() * 40).to(torch.bfloat16).view(D // 8, 8)
    # garbage (huge) values beyond seq_len in the last block of sequence 1 must be ignored
    blk1 = int(bt[1, 2])
    kc[blk1, :, :, 4:, :] = 100.0
    vc[blk1, :, :, 4:] = float("nan")
    ref = oracle.paged_attention(q, kc, vc, bt, lens, D ** -0.5)
    out = torch.empty_like(q, device="cuda")
    ops.paged_attention_v1(out, q.cuda(), kc.cuda(), vc.cuda(), Hkv, D ** -0.5, bt.cuda(), lens.cuda(),
                           bs, 700, None, "auto", 1.0, 1.0)
    assert torch.isfinite(out).all()
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=2e-3, rtol=1.6e-2)


def test_paged_decode_errors(ops):
    q = torch.zeros(1, 8, 128, dtype=torch.bfloat16, device="cuda")
    kc = torch.zeros(4, 2, 16, 16, 8, dtype=torch.bfloat16, device="cuda")
    vc = torch.zeros(4, 2, 128, 16, dtype=torch.bfloat16, device="cuda")
    bt = torch.zeros(1, 2, dtype=torch.int32, device="cuda")
    lens = torch.ones(1, dtype=torch.int32, device="cuda")
    out = torch.empty_like(q)
    with pytest.raises(NotImplementedError):
        ops.paged_attention_v1(out, q, kc, vc, 2, 1.0, bt, lens, 16, 1, None, "fp8", 1.0, 1.0)
    # max_seq_len larger than the table addresses (the logical length under eviction) is fine
    ops.paged_attention_v1(out, q, kc, vc, 2, 1.0, bt, lens, 16, 100, None, "auto", 1.0, 1.0)
    with pytest.raises(RuntimeError):  # CPU tensors
        ops.paged_attention_v1(out.cpu(), q.cpu(), kc.cpu(), vc.cpu(), 2, 1.0, bt.cpu(), lens.cpu(), 16, 1,
                               None, "auto", 1.0, 1.0)
    with pytest.raises(ValueError):
        ops.paged_attention_v1(out, q, kc, vc, 2, 1.0, bt, lens, 16, 1, None, "auto", 1.0, 1.0, 0, 2, 4, 64, 0)


# ------------------------------------------------------------------ block copy / swap
def test_copy_and_swap_blocks(ops, oracle):
    g = torch.Generator().manual_seed(2)
    layers = 3
    kcs, vcs = zip(*[make_cache(10, 2, 64, 16, torch.bfloat16, g) for _ in range(layers)])
    kd, vd = [t.cuda() for t in kcs], [t.cuda() for t in vcs]
    mapping = torch.tensor([[0, 5], [0, 6], [3, 9]], dtype=torch.int64)
    for kc, vc in zip(kcs, vcs):
        oracle.copy_blocks(kc, vc, mapping)
    ops.copy_blocks(kd, vd, mapping.cuda())
    for i in range(layers):
        assert torch.equal(kd[i].cpu(), kcs[i]) and torch.equal(vd[i].cpu(), vcs[i])
    # swap GPU -> CPU -> GPU round trip
    src = kd[0]
    host = torch.zeros_like(kcs[0]).pin_memory()
    ops.swap_blocks(src, host, torch.tensor([[1, 2], [4, 0]], dtype=torch.int64))
    torch.cuda.synchronize()
    assert torch.equal(host[2], kcs[0][1]) and torch.equal(host[0], kcs[0][4])
    dst = torch.zeros_like(src)
    ops.swap_blocks(host, dst, torch.tensor([[2, 7]], dtype=torch.int64))
    torch.cuda.synchronize()
    assert torch.equal(dst[7].cpu(), kcs[0][1])
    with pytest.raises(RuntimeError):
        ops.swap_blocks(src, dst, torch.tensor([[0, 1]], dtype=torch.int64).cuda())


# ------------------------------------------------------------------ eviction scoring / decision
def test_block_score_golden(ops, oracle):
    z = np.load(os.path.join(GOLDEN, "block_scores.npz"))
    n = len([k for k in z.files if k.endswith("_meta")])
    for c in range(n):
        bs, H, D, dt = [int(v) for v in z[f"c{c}_meta"]]
        v = from_bits(z[f"c{c}_v"], dt)
        expect = from_bits(z[f"c{c}_score"], dt).float().item()
        got = ops.block_score_value_l2(v.cuda(), None, 1, ops.SCORE_TOKEN_MAJOR, H, D, bs).item()
        vc = v.permute(1, 2, 0).contiguous().cuda()
        got_c = ops.block_score_value_l2(vc, None, 1, ops.SCORE_CACHE, H, D, bs).item()
        if dt == 0:
            assert got == pytest.approx(expect, rel=2e-6) and got_c == pytest.approx(expect, rel=2e-6)
        else:
            assert got == expect and got_c == expect, (c, got, got_c, expect)


def test_block_score_cache_blocks(ops, oracle):
    g = torch.Generator().manual_seed(9)
    nb, H, D, bs = 40, 8, 128, 16
    _, vc = make_cache(nb, H, D, bs, torch.bfloat16, g)
    ids = torch.tensor([3, 17, 0, 39, 17], dtype=torch.int32)
    for layout in (1, 2):
        got = ops.block_score_value_l2(vc.cuda(), ids.cuda(), len(ids), layout, H, D, bs).cpu().tolist()
        expect = [oracle.block_score(vc[int(b)], bs, H, D, layout) for b in ids]
        assert got == expect


@pytest.mark.parametrize("method", ["streamingLLM", "streamingLLM-1"])
def test_evict_streaming(ops, oracle, method):
    bs, budget = 16, 96
    seq_lens = [50, 96, 97, 112, 113, 128, 200, 208]
    B = len(seq_lens)
    M = 16
    g = torch.Generator().manual_seed(1)
    tables = torch.randint(0, 1000, (B, M), generator=g, dtype=torch.int32)
    nblk = torch.tensor([min(M, -(-L // bs)) for L in seq_lens], dtype=torch.int32)
    rmv_e, zero_e = oracle.decode_streaming(seq_lens, tables.tolist(), budget, bs, method)
    rmv, out, zero = ops.evict_select_compact(tables.cuda(), torch.tensor(seq_lens, dtype=torch.int32).cuda(),
                                              nblk.cuda(), method, budget, bs)
    assert rmv.cpu().tolist() == rmv_e
    assert torch.equal(out.cpu(), oracle.compact_block_tables(tables, rmv_e))
    if method == "streamingLLM-1":
        assert [z for z in zero.cpu().tolist() if z >= 0] == zero_e
    # reference-compatible (quirk) compaction
    _, out_q1, _ = ops.evict_select_compact(tables.cuda(), torch.tensor(seq_lens, dtype=torch.int32).cuda(),
                                            nblk.cuda(), method, budget, bs, compat_q1=True)
    assert torch.equal(out_q1.cpu(), oracle.compact_block_tables(tables, rmv_e, compat_q1=True))


@pytest.mark.parametrize("method,topk", [("global", -1), ("local", 3)])
@pytest.mark.parametrize("drop", [True, False])
@pytest.mark.parametrize("layout", [1, 2])
def test_evict_topk_trace(ops, oracle, method, topk, drop, layout):
    """Multi-step decode trace: device score table + victim choice vs the restated
    _prune_decode_reqs_topk with value_l2 scores (random off)."""
    g = torch.Generator().manual_seed(21)
    bs, budget, H, D, nb = 16, 64, 8, 128, 128
    _, vc = make_cache(nb, H, D, bs, torch.bfloat16, g)
    vc_d = vc.cuda()
    B, W = 3, 32
    # per sequence: host model = list of physical blocks + list of scores
    tables = [torch.randperm(nb, generator=g)[:n].tolist() for n in (4, 5, 4)]
    seq_len = [64, 80, 63]
    scores = [[oracle.block_score(vc[b], bs, H, D, layout) for b in t] for t in tables]
    table_d = torch.zeros((B, W), dtype=torch.float32)
    counts = torch.zeros(B, dtype=torch.int32)
    for i, s in enumerate(scores):
        table_d[i, :len(s)] = torch.tensor(s)
        counts[i] = len(s)
    # sequences use score rows in a permuted order to exercise score_rows
    rows = torch.tensor([2, 0, 1], dtype=torch.int32)
    st = torch.zeros((B, W), dtype=torch.float32)
    sc = torch.zeros(B, dtype=torch.int32)
    for i in range(B):
        st[rows[i]] = table_d[i]
        sc[rows[i]] = counts[i]
    st_d, sc_d, rows_d = st.cuda(), sc.cuda(), rows.cuda()
    free = [b for b in range(nb) if all(b not in t for t in tables)]
    kv_len = list(seq_len)
    for step in range(70):
        for i in range(B):
            seq_len[i] += 1
            kv_len[i] += 1
            if kv_len[i] > len(tables[i]) * bs:  # last block full -> the scheduler appends one
                tables[i].append(free.pop(0))
        M = max(len(t) for t in tables)
        bt = torch.zeros((B, M), dtype=torch.int32)
        for i, t in enumerate(tables):
            bt[i, :len(t)] = torch.tensor(t)
        nblk = torch.tensor([len(t) for t in tables], dtype=torch.int32)
        rmv, out, _ = ops.evict_select_compact(
            bt.cuda(), torch.tensor(seq_len, dtype=torch.int32).cuda(), nblk.cuda(), method, budget, bs,
            topk, drop_victim_score=drop, score_layout=layout, score_table=st_d, score_counts=sc_d,
            score_rows=rows_d, value_cache=vc_d)
        rmv = rmv.cpu().tolist()
        expect = []
        for i in range(B):
            if oracle.decode_trigger(seq_len[i], budget, bs):
                new = oracle.block_score(vc[tables[i][-1]], bs, H, D, layout)
                expect.append(oracle.decode_topk_victim(scores[i], new, method, topk, drop))
            else:
                expect.append(-1)
        assert rmv == expect, (step, rmv, expect)
        assert torch.equal(out.cpu(), oracle.compact_block_tables(bt, expect))
        if any(e >= len(tables[i]) for i, e in enumerate(expect)):
            # without dropping the victim's score the list outgrows the table (the reference's
            # HEAD behaviour) and the index eventually points past it: the trace ends there
            assert not drop
            break
        for i in range(B):
            if expect[i] >= 0:
                free.append(tables[i].pop(expect[i]))
                kv_len[i] -= bs
    got_counts = sc_d.cpu()
    got_table = st_d.cpu()
    for i in range(B):
        n = int(got_counts[rows[i]])
        assert n == len(scores[i])
        assert got_table[rows[i], :n].tolist() == scores[i]


def test_evict_decisions_reference_golden(ops):
    """pe_evict_select_compact in its reference-compatible form (victim's score kept, in-loop
    table slice, the decode call site's flat view of the scored block) against what the
    reference's own decode-decision methods returned on multi-step traces
    (tests/golden/make_golden_decode.py): victims, the compacted table tensor, zeroed slots and
    the score rows after every step; HEAD's random victims go through victim_override."""
    import json
    import os
    from conftest import GOLDEN, from_bits
    meta = json.load(open(os.path.join(GOLDEN, "decode_decisions.json")))
    kv = from_bits(np.load(os.path.join(GOLDEN, "decode_decisions.npz"))["kv_cache"], 2)
    bs, H, D, nb = meta["block_size"], meta["num_kv_heads"], meta["head_size"], meta["num_blocks"]
    budget, topk = meta["budget"], meta["topk_blocks"]
    vc_d = kv[1].view(nb, H, D, bs).cuda()
    ran = 0
    for tr in meta["traces"]:
        method, formula, use_random = tr["method"], tr["formula"], tr["random"]
        if method == "inverse_key_l2" or (formula not in (None, "value_l2")):
            continue     # host-side draws / key-norm formulas: covered at the backend level
        B, W = len(tr["seqs"]), 32
        st_d = torch.zeros((B, W), dtype=torch.float32)
        sc_d = torch.zeros(B, dtype=torch.int32)
        for i, sc in enumerate(tr["init_scores"]):
            st_d[i, :len(sc)] = torch.tensor(sc)
            sc_d[i] = len(sc)
        st_d, sc_d = st_d.cuda(), sc_d.cuda()
        rows_d = torch.arange(B, dtype=torch.int32).cuda()
        for st in tr["steps"]:
            tables, M = st["tables"], st["width"]
            bt = torch.zeros((B, M), dtype=torch.int32)
            for i, t in enumerate(tables):
                bt[i, :len(t)] = torch.tensor(t, dtype=torch.int32)
            nblk = torch.tensor([len(t) for t in tables], dtype=torch.int32)
            expect = [-1] * B
            for i, (_, b) in st["rmv"].items():
                expect[int(i)] = b
            kw = {}
            if method in ("local", "global"):
                if use_random:
                    kw["victim_override"] = torch.tensor([max(e, 0) for e in expect], dtype=torch.int32).cuda()
                else:
                    kw.update(score_table=st_d, score_counts=sc_d, score_rows=rows_d, value_cache=vc_d,
                              score_layout=ops.SCORE_CACHE_FLAT)
            rmv, out, zero = ops.evict_select_compact(
                bt.cuda(), torch.tensor(st["seq_lens"], dtype=torch.int32).cuda(), nblk.cuda(), method,
                budget, bs, topk, drop_victim_score=False, compat_q1=True, **kw)
            assert rmv.cpu().tolist() == expect, (method, use_random, st["seq_lens"])
            if st["updated_tables"] is not None:
                assert out.cpu().tolist() == st["updated_tables"], (method, st["seq_lens"])
            if method == "streamingLLM-1":
                assert [z for z in zero.cpu().tolist() if z >= 0] == st["zero_slots"]
            if "scores_after" in st:
                cnt = sc_d.cpu().tolist()
                tab = st_d.cpu()
                for i, sc in enumerate(st["scores_after"]):
                    # the device table holds f32; the prompt scores of the trace are decimal literals
                    assert cnt[i] == len(sc) and tab[i, :len(sc)].tolist() == torch.tensor(sc).tolist(), \
                        (method, st["seq_lens"], i)
        ran += 1
    assert ran == 6
    ops.check_device_diagnostics()


def test_evict_victim_override(ops, oracle):
    """Host-chosen victims (the reference's random mode) are applied verbatim."""
    bs, budget = 16, 64
    tables = torch.arange(0, 3 * 8, dtype=torch.int32).reshape(3, 8)
    seq_lens = torch.tensor([128, 100, 112], dtype=torch.int32)
    nblk = torch.tensor([8, 7, 7], dtype=torch.int32)
    ov = torch.tensor([5, 3, 2], dtype=torch.int32)
    rmv, out, _ = ops.evict_select_compact(tables.cuda(), seq_lens.cuda(), nblk.cuda(), "global", budget, bs,
                                           victim_override=ov.cuda())
    assert rmv.cpu().tolist() == [5, -1, 2]
    assert torch.equal(out.cpu(), oracle.compact_block_tables(tables, [5, -1, 2]))


# ------------------------------------------------------------------ prefill attention
def _cu(lens):
    return torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)


PREFILL_CASES = [
    # Hq, Hkv, D, dtype, q_lens, kv_lens (kv < q: prompt pruned before attention)
    (32, 8, 128, torch.bfloat16, [1024], [768]),
    (32, 8, 128, torch.bfloat16, [1, 17, 130, 300], [1, 17, 130, 300]),
    (8, 2, 128, torch.float16, [200, 77, 513], [96, 77, 260]),
    (24, 8, 128, torch.bfloat16, [300, 130], [300, 96]),     # Llama-3.2-3B: 3 query heads per KV head
    (32, 8, 64, torch.bfloat16, [513, 64], [513, 64]),       # Llama-3.2-1B
    (12, 12, 64, torch.float16, [333, 64], [333, 64]),
    (8, 8, 64, torch.bfloat16, [150], [100]),
    (4, 2, 64, torch.float32, [50], [50]),               # generic path (f32)
    # the other head sizes of the reference's op (vllm/attention/ops/paged_attn.py:37) on the
    # matrix-core kernel: 32 in 64-element rows, 80 - 120 in 128-element rows, 192 and 256 in
    # 256-element rows (8-wave workgroups); prompts of one tile, several tiles, ragged, pruned
    (4, 4, 80, torch.bfloat16, [70, 33], [48, 33]),
    (4, 2, 96, torch.bfloat16, [90, 17], [60, 17]),
    (4, 4, 112, torch.float16, [65], [65]),
    (6, 2, 120, torch.bfloat16, [40, 130], [40, 100]),
    (4, 1, 192, torch.bfloat16, [150], [150]),
    (8, 2, 256, torch.float16, [70, 129], [70, 96]),
    (8, 2, 32, torch.bfloat16, [1, 64, 65, 300], [1, 64, 65, 300]),
    (8, 8, 80, torch.float16, [513, 128, 1], [513, 96, 1]),
    (16, 4, 96, torch.bfloat16, [700, 129], [512, 129]),
    (4, 1, 112, torch.bfloat16, [1024], [768]),
    (8, 4, 120, torch.float16, [257, 600], [257, 600]),
    (8, 2, 192, torch.bfloat16, [1, 127, 128, 129, 640], [1, 127, 128, 129, 400]),
    (16, 16, 192, torch.float16, [300], [300]),
    (8, 2, 256, torch.bfloat16, [1024, 63], [768, 63]),
    (32, 8, 256, torch.bfloat16, [385], [385]),
]


@pytest.mark.parametrize("Hq,Hkv,D,dtype,q_lens,kv_lens", PREFILL_CASES)
def test_prefill_causal(ops, oracle, Hq, Hkv, D, dtype, q_lens, kv_lens):
    g = torch.Generator().manual_seed(Hq + D + sum(q_lens))
    Tq, Tkv = sum(q_lens), sum(kv_lens)
    q = (torch.rand(Tq, Hq, D, generator=g) * 2 - 1).to(dtype)
    k = (torch.rand(Tkv, Hkv, D, generator=g) * 2 - 1).to(dtype)
    v = (torch.rand(Tkv, Hkv, D, generator=g) * 2 - 1).to(dtype)
    sc = D ** -0.5
    ref = oracle.prefill_attention(q.float(), k.float(), v.float(), q_lens, kv_lens, sc)
    out = ops.prefill_causal(q.cuda(), k.cuda(), v.cuda(), _cu(q_lens).cuda(), _cu(kv_lens).cuda(),
                             max(q_lens), sc)
    # SDPA on this path is a third-party kernel whose rounding no reference test pins:
    # tolerance vs an f32 evaluation of the same inputs (bf16: atol 2e-2, rtol 1.6e-2)
    tol = {torch.bfloat16: (2e-2, 1.6e-2), torch.float16: (3e-3, 1e-2), torch.float32: (1e-4, 1e-4)}[dtype]
    torch.testing.assert_close(out.cpu().float(), ref, atol=tol[0], rtol=tol[1])


@pytest.mark.parametrize("D", [32, 80, 96, 112, 120, 192, 256])
def test_prefill_matrix_core_kernel_against_the_general_one(ops, D, monkeypatch):
    """The two prefill kernels on the same ragged batch: the matrix-core kernel (rows padded to
    64 / 128 / 256 elements inside the kernel) and the general one (PE_PREFILL_FORCE_GENERIC=1),
    which walks exactly head_size elements."""
    g = torch.Generator().manual_seed(D)
    Hq, Hkv = 8, 2
    q_lens, kv_lens = [200, 1, 333, 64], [200, 1, 208, 64]
    q = (torch.rand(sum(q_lens), Hq, D, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    k = (torch.rand(sum(kv_lens), Hkv, D, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    v = (torch.rand(sum(kv_lens), Hkv, D, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    cq, ck = _cu(q_lens).cuda(), _cu(kv_lens).cuda()
    fast = ops.prefill_causal(q, k, v, cq, ck, max(q_lens), D ** -0.5)
    monkeypatch.setenv("PE_PREFILL_FORCE_GENERIC", "1")
    slow = ops.prefill_causal(q, k, v, cq, ck, max(q_lens), D ** -0.5)
    assert not torch.equal(fast, slow) or D == 0          # two kernels, two summation orders
    torch.testing.assert_close(fast.float(), slow.float(), atol=2e-2, rtol=1.6e-2)
    # strided views of a fused qkv buffer (what the model hands over) give the same bits
    qkv = torch.zeros(sum(q_lens), (Hq + 2 * Hkv) * D, dtype=torch.bfloat16, device="cuda")
    if q_lens == kv_lens:
        return
    monkeypatch.delenv("PE_PREFILL_FORCE_GENERIC")
    qv = qkv[:, :Hq * D].view(-1, Hq, D)
    qv.copy_(q)
    again = ops.prefill_causal(qv, k, v, cq, ck, max(q_lens), D ** -0.5)
    assert torch.equal(again, fast)


def test_prefill_reference_sdpa_forward_golden(ops):
    """pe_prefill_causal against outputs of the reference's own _run_sdpa_forward
    (torch_cuda_sdpa.py:1069-1127; run in f32 on CPU by tests/golden/make_golden_decode.py over the
    same bf16 inputs): grouped heads, per-request kv offsets by the pruned lengths, top-left
    aligned mask.  Tolerance of a bf16 kernel against an f32 result: atol 2e-2, rtol 1.6e-2 (the
    rounding of the CUDA SDPA kernel the reference would run is pinned by nothing)."""
    from test_oracle_golden import sdpa_cases
    n = 0
    for c, q, k, v, ref in sdpa_cases():
        out = ops.prefill_causal(q.cuda(), k.cuda(), v.cuda(), _cu(c["lens_q"]).cuda(),
                                 _cu(c["lens_kv"]).cuda(), max(c["lens_q"]), c["D"] ** -0.5)
        torch.testing.assert_close(out.cpu().float(), ref, atol=2e-2, rtol=1.6e-2)
        n += 1
    assert n == 4


def test_prefill_tile_shapes_same_bits(ops):
    """pe_prefill_causal picks its tile shape from the declared maximum prompt length and the
    number of tiles (16 rows x 8 waves up to 512 tokens, 32 x 8 for long prompts, 16 x 4 for
    launches of few tiles, 32 x 4 otherwise); max_q_len is an upper bound, so declaring a larger one
    moves the same data through each shape: the outputs must not differ in a single bit."""
    g = torch.Generator().manual_seed(12)
    Hq, Hkv, D = 32, 8, 128
    q_lens = [300, 512, 17, 129, 256, 1, 400, 333]
    kv_lens = [300, 384, 17, 129, 256, 1, 208, 333]
    q = (torch.rand(sum(q_lens), Hq, D, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    k = (torch.rand(sum(kv_lens), Hkv, D, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    v = (torch.rand(sum(kv_lens), Hkv, D, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    cq, ck = _cu(q_lens).cuda(), _cu(kv_lens).cuda()
    outs = [ops.prefill_causal(q, k, v, cq, ck, m, D ** -0.5) for m in (512, 1024, 2048)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # a launch with at most 256 tiles of 128 rows runs 64-row workgroups (16 rows x 4 waves): the
    # first two requests on their own
    n_q, n_kv = sum(q_lens[:2]), sum(kv_lens[:2])
    few = ops.prefill_causal(q[:n_q], k[:n_kv], v[:n_kv], cq[:3], ck[:3], 512, D ** -0.5)
    assert torch.equal(few, outs[0][:n_q])


def test_prefill_matches_torch_sdpa_top_left(ops):
    """The mask alignment is the one torch SDPA(is_causal=True) uses for L_q != L_kv."""
    g = torch.Generator().manual_seed(4)
    Hq, Hkv, D, Lq, Lkv = 8, 2, 128, 300, 200
    q = (torch.rand(Lq, Hq, D, generator=g) - 0.5).to(torch.bfloat16)
    k = (torch.rand(Lkv, Hkv, D, generator=g) - 0.5).to(torch.bfloat16)
    v = (torch.rand(Lkv, Hkv, D, generator=g) - 0.5).to(torch.bfloat16)
    kk = k.repeat_interleave(Hq // Hkv, dim=1).float().movedim(0, 1)[None]
    vv = v.repeat_interleave(Hq // Hkv, dim=1).float().movedim(0, 1)[None]
    ref = torch.nn.functional.scaled_dot_product_attention(
        q.float().movedim(0, 1)[None], kk, vv, is_causal=True, scale=D ** -0.5)[0].movedim(0, 1)
    out = ops.prefill_causal(q.cuda(), k.cuda(), v.cuda(), _cu([Lq]).cuda(), _cu([Lkv]).cuda(), Lq, D ** -0.5)
    torch.testing.assert_close(out.cpu().float(), ref, atol=2e-2, rtol=1.6e-2)


def test_prefill_spike_forces_rescale(ops, oracle):
    """A key that dominates late in the sequence forces the online-softmax rescale branch."""
    g = torch.Generator().manual_seed(8)
    Hq, Hkv, D, L = 4, 1, 128, 400
    q = (torch.rand(L, Hq, D, generator=g) - 0.5).to(torch.bfloat16)
    k = (torch.rand(L, Hkv, D, generator=g) - 0.5).to(torch.bfloat16)
    v = (torch.rand(L, Hkv, D, generator=g) - 0.5).to(torch.bfloat16)
    k[333, 0] = (q[350, 2].float() * 6).to(torch.bfloat16)
    ref = oracle.prefill_attention(q.float(), k.float(), v.float(), [L], [L], D ** -0.5)
    out = ops.prefill_causal(q.cuda(), k.cuda(), v.cuda(), _cu([L]).cuda(), _cu([L]).cuda(), L, D ** -0.5)
    torch.testing.assert_close(out.cpu().float(), ref, atol=2e-2, rtol=1.6e-2)


@pytest.mark.parametrize("dtype,D,step", [(torch.bfloat16, 128, 0.05), (torch.float16, 128, 0.05),
                                          (torch.bfloat16, 64, 0.03), (torch.bfloat16, 128, 0.2)])
def test_prefill_scores_creep_past_reference_point(ops, oracle, dtype, D, step):
    """Scores that grow by `step` per key: the row maximum drifts a few bits per 64-key tile, so
    the kernel's reference point for the exponentials lags behind it for a tile or two (weights
    above 1) before it moves - the normalised output must not notice.  step 0.2 moves it every
    tile."""
    g = torch.Generator().manual_seed(21)
    Hq, Hkv, L = 4, 2, 600
    u = torch.nn.functional.normalize(torch.randn(D, generator=g), dim=0)
    sc = D ** -0.5
    q = (u[None, None, :] * (1.0 / sc) ** 0.5).repeat(L, Hq, 1)
    q = (q + 0.05 * torch.randn(L, Hq, D, generator=g)).to(dtype)
    ramp = torch.arange(L, dtype=torch.float32)[:, None, None] * step
    k = (u[None, None, :] * ramp * (1.0 / sc) ** 0.5).repeat(1, Hkv, 1)
    k = (k + 0.05 * torch.randn(L, Hkv, D, generator=g)).to(dtype)
    v = (torch.rand(L, Hkv, D, generator=g) - 0.5).to(dtype)
    ref = oracle.prefill_attention(q.float(), k.float(), v.float(), [L], [L], sc)
    out = ops.prefill_causal(q.cuda(), k.cuda(), v.cuda(), _cu([L]).cuda(), _cu([L]).cuda(), L, sc)
    tol = (2e-2, 1.6e-2) if dtype == torch.bfloat16 else (3e-3, 1e-2)
    torch.testing.assert_close(out.cpu().float(), ref, atol=tol[0], rtol=tol[1])


# ------------------------------------------------------------------ token-level prompt prune
def test_prompt_token_select_golden(ops, oracle):
    """Against KVCachePruner.prune_prompt outputs (tie-free f32 inputs): the compat layout
    reproduces the reference bit for bit; the per-head layout is checked against the oracle."""
    z = np.load(os.path.join(GOLDEN, "prompt_prune_topk.npz"))
    n = len([k for k in z.files if k.endswith("_meta")])
    for c in range(n):
        L, budget, bs, H, D, m = [int(v) for v in z[f"c{c}_meta"]]
        method = ["global", "local", "inverse_key_l2"][m]
        k, v = from_bits(z[f"c{c}_k"], 0), from_bits(z[f"c{c}_v"], 0)
        n_kept = oracle.pruned_length(L, budget, bs)
        cu_q, cu_k = _cu([L]).cuda(), _cu([n_kept]).cuda()
        kd, vd = k.cuda(), v.cuda()
        idx = ops.prompt_token_select(kd, vd, cu_q, cu_k, n_kept, budget, bs, method, compat_q2=True)
        pk, pv = ops.prompt_prune_gather(kd, vd, idx, mode=2)
        assert torch.equal(pk.cpu(), from_bits(z[f"c{c}_pk"], 0)), c
        assert torch.equal(pv.cpu(), from_bits(z[f"c{c}_pv"], 0)), c
        idx = ops.prompt_token_select(kd, vd, cu_q, cu_k, n_kept, budget, bs, method)
        pk, pv = ops.prompt_prune_gather(kd, vd, idx)
        ek, ev = oracle.prompt_prune(k, v, method, budget, bs)
        assert torch.equal(pk.cpu(), ek) and torch.equal(pv.cpu(), ev), c


def test_prompt_token_select_full_size_golden(ops, oracle):
    """The reference's prune at the benchmark's prompt sizes (fixtures hold digests and kept
    bitmaps): exact for tie-free f32 inputs, same kept scores for 16-bit inputs."""
    from test_oracle_golden import _full_cases, _same_kept_scores, _sha
    for c, L, budget, bs, H, D, method, dtype, k, v, ref_kept, sha_k, sha_v in _full_cases():
        n_kept = oracle.pruned_length(L, budget, bs)
        cu_q, cu_k = _cu([L]).cuda(), _cu([n_kept]).cuda()
        kd, vd = k.cuda(), v.cuda()
        if dtype == torch.float32:
            idx = ops.prompt_token_select(kd, vd, cu_q, cu_k, n_kept, budget, bs, method, compat_q2=True)
            pk, pv = ops.prompt_prune_gather(kd, vd, idx, mode=2)
            assert _sha(pk.cpu()) == sha_k and _sha(pv.cpu()) == sha_v, c
        idx = ops.prompt_token_select(kd, vd, cu_q, cu_k, n_kept, budget, bs, method).cpu().long()
        end_first, end_mid, n_prune = oracle.prompt_slices(L, budget, bs)
        mid = idx[end_first:n_kept - (L - end_mid)] - end_first          # kept middle rows per head
        mine = torch.zeros(end_mid - end_first, H, dtype=torch.bool)
        mine.scatter_(0, mid, True)
        assert (mid[1:] > mid[:-1]).all()                                # order preserved per head
        scores = oracle.token_scores(k[end_first:end_mid], v[end_first:end_mid], method)
        if dtype == torch.float32:
            assert torch.equal(mine, ref_kept), c
        else:
            _same_kept_scores(scores, mine, ref_kept)


@pytest.mark.parametrize("method", ["global", "inverse_key_l2"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_prompt_token_select_batch_with_ties(ops, oracle, method, dtype):
    """Batch of prompts in a 16-bit type (many equal scores): ties drop the lowest token first,
    as the oracle's stable sort does; also checks the fused gather-into-cache form."""
    g = torch.Generator().manual_seed(12)
    H, D, bs, budget = 8, 128, 16, 96
    lens = [40, 300, 97, 1030]
    kept = [oracle.pruned_length(L, budget, bs) for L in lens]
    T = sum(lens)
    k = ((torch.rand(T, H, D, generator=g) * 2 - 1) * D ** -0.5).to(dtype)
    v = ((torch.rand(T, H, D, generator=g) * 2 - 1) * D ** -0.5).to(dtype)
    idx = ops.prompt_token_select(k.cuda(), v.cuda(), _cu(lens).cuda(), _cu(kept).cuda(), sum(kept),
                                  budget, bs, method)
    pk, pv = ops.prompt_prune_gather(k.cuda(), v.cuda(), idx)
    off = 0
    eks, evs = [], []
    for L in lens:
        ek, ev = oracle.prompt_prune(k[off:off + L], v[off:off + L], method, budget, bs)
        eks.append(ek)
        evs.append(ev)
        off += L
    ek, ev = torch.cat(eks), torch.cat(evs)
    assert torch.equal(pk.cpu(), ek) and torch.equal(pv.cpu(), ev)
    # fused: gather straight into the paged cache
    nb = 200
    kc, vc = make_cache(nb, H, D, bs, dtype, g)
    kc_d, vc_d = kc.cuda(), vc.cuda()
    slots = torch.randperm(nb * bs, generator=g)[:sum(kept)].to(torch.int64)
    oracle.reshape_and_cache(ek, ev, kc, vc, slots)
    ops.reshape_and_cache(k.cuda(), v.cuda(), kc_d, vc_d, slots.cuda(), src_index=idx)
    assert torch.equal(kc_d.cpu(), kc) and torch.equal(vc_d.cpu(), vc)


# ------------------------------------------------------------------ fused decode step
def disjoint_tables(lens, bs, nb, g):
    """Block tables whose rows share no block, as a block manager hands them out: the fused step
    stores a sequence's new token while other workgroups of the launch are reading, so a block
    that two rows used would be read by one row while the other writes into it."""
    need = [-(-L // bs) for L in lens]
    assert sum(need) <= nb
    perm = torch.randperm(nb, generator=g).to(torch.int32)
    bt = torch.zeros(len(lens), max(need) + 1, dtype=torch.int32)
    at = 0
    for i, n in enumerate(need):
        bt[i, :n] = perm[at:at + n]
        at += n
    return bt


@pytest.mark.parametrize("Hq,Hkv,D,dtype", [(32, 8, 128, torch.bfloat16), (12, 12, 64, torch.float16),
                                            (8, 1, 128, torch.bfloat16), (8, 2, 32, torch.bfloat16),
                                            (8, 2, 80, torch.bfloat16), (12, 4, 96, torch.float16),
                                            (8, 8, 112, torch.bfloat16), (6, 2, 120, torch.bfloat16),
                                            (8, 2, 192, torch.bfloat16), (16, 2, 256, torch.float16)])
@pytest.mark.parametrize("splits", [0, 3])
@pytest.mark.parametrize("bs", [16, 8, 32])
def test_decode_step_fused_write(ops, oracle, Hq, Hkv, D, dtype, splits, bs, monkeypatch):
    """One launch (KV write + attention) == reshape_and_cache followed by paged attention:
    cache bytes identical, outputs within the kernel tolerance.  Block sizes 8, 16 and 32 (the
    page sizes of the reference's scripts/diff_page_sizes.sh) all run the MFMA kernel for the head
    sizes made of whole 32-element k-steps (32, 64, 96, 128, 192, 256); 80, 112 and 120
    (paged_attn.py:37) run it with 16-token blocks."""
    if bs != 16 and D % 32:
        assert not ops._lib.load().pe_paged_decode_supports_fused_write(Hq, Hkv, D, bs, 2)
        pytest.skip("8- and 32-token blocks are built for head sizes made of whole 32-element k-steps")
    if splits:
        monkeypatch.setenv("PE_DECODE_SPLITS", str(splits))
    g = torch.Generator().manual_seed(Hq + D)
    nb = 6400 // bs
    assert ops._lib.load().pe_paged_decode_supports_fused_write(Hq, Hkv, D, bs, 1 if dtype == torch.float16 else 2)
    lens = [1, 16, 17, 32, 33, 700, 784, 2000]       # kv length including the new token
    S = len(lens)
    sc = D ** -0.5
    kc, vc = make_cache(nb, Hkv, D, bs, dtype, g)
    q = ((torch.rand(S, Hq, D, generator=g) * 2 - 1) * sc).to(dtype)
    k = ((torch.rand(S, Hkv, D, generator=g) * 2 - 1) * sc).to(dtype)
    v = ((torch.rand(S, Hkv, D, generator=g) * 2 - 1) * sc).to(dtype)
    bt = disjoint_tables(lens, bs, nb, g)
    slots = torch.tensor([int(bt[i, (L - 1) // bs]) * bs + (L - 1) % bs for i, L in enumerate(lens)],
                         dtype=torch.int64)
    slots[1] = -1                                     # padding token: nothing written
    lens_t = torch.tensor(lens, dtype=torch.int32)
    kc_d, vc_d = kc.cuda(), vc.cuda()
    out = torch.empty_like(q, device="cuda")
    ops.decode_step(out, q.cuda(), k.cuda(), v.cuda(), kc_d, vc_d, slots.cuda(), Hkv, sc, bt.cuda(),
                    lens_t.cuda(), bs, max(lens))
    oracle.reshape_and_cache(k, v, kc, vc, slots)
    ref = oracle.paged_attention(q, kc, vc, bt, lens_t, sc)
    assert torch.equal(kc_d.cpu(), kc) and torch.equal(vc_d.cpu(), vc)
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=ATOL[dtype], rtol=RTOL[dtype])


@pytest.mark.parametrize("dtype,bs", [(torch.bfloat16, 16), (torch.float16, 8), (torch.bfloat16, 32)])
def test_zero_slots_of_all_layers_in_one_launch(ops, dtype, bs):
    """pe_zero_slots_layers leaves in every layer's cache exactly the bytes the per-layer zeroing of
    pe_reshape_and_cache (zero_slots) leaves; padding (-1) and out-of-range slots are skipped."""
    g = torch.Generator().manual_seed(31 + bs)
    Hkv, D, nb, L = 4, 64, 12, 5
    shape = (2, nb, bs * Hkv * D)
    caches = [(torch.rand(shape, generator=g) - 0.5).to(dtype).cuda() for _ in range(L)]
    want = [c.clone() for c in caches]
    slots = torch.tensor([3, bs * 5 + 7, -1, bs * nb - 1, bs * nb + 4, bs + 1], dtype=torch.int64).cuda()
    empty = torch.zeros(0, Hkv, D, dtype=dtype, device="cuda")
    for c in want:
        kc = c[0].view(nb, Hkv, D // 8, bs, 8)
        vc = c[1].view(nb, Hkv, D, bs)
        ops.reshape_and_cache(empty, empty, kc, vc, slots[:0], zero_slots=slots)
    v_off = caches[0].stride(0) * caches[0].element_size()
    kp = torch.tensor([c.data_ptr() for c in caches], dtype=torch.int64).cuda()
    vp = torch.tensor([c.data_ptr() + v_off for c in caches], dtype=torch.int64).cuda()
    ops.zero_slots_layers(kp, vp, slots, Hkv, D, bs, nb, dtype)
    for a, b in zip(caches, want):
        assert torch.equal(a, b)
    assert int((caches[0] == 0).sum()) >= 4 * 2 * Hkv * D      # four valid slots, K and V rows


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("layout", [0, 1])
def test_block_score_two_step_head_sharded(oracle, dtype, layout):
    """pe_block_token_norm_sums + pe_block_score_from_token_sums: with all heads in one call they
    equal pe_block_score_value_l2 and the oracle bit for bit; with the heads split over two or
    four 'ranks' (sums added in f32, as the all-reduce does) they still do."""
    from vllm_paged_compression_amd import ops
    g = torch.Generator().manual_seed(5)
    H, D, bs, nblk = 8, 128, 16, 7
    if layout == 0:
        vals = (torch.rand(nblk * bs, H, D, generator=g) * 2 - 1).to(dtype)
        blocks = [vals[b * bs:(b + 1) * bs] for b in range(nblk)]
    else:
        vals = (torch.rand(nblk, H, D, bs, generator=g) * 2 - 1).to(dtype)
        blocks = [vals[b] for b in range(nblk)]
    dv = vals.cuda()
    ids = None if layout == 0 else torch.arange(nblk, dtype=torch.int32, device="cuda")
    stride = None if layout == 0 else dv.stride(0)
    fused = ops.block_score_value_l2(dv, ids, nblk, layout, H, D, bs, block_stride=stride).cpu()
    expect = torch.tensor([oracle.block_score(b, bs, H, D, layout) for b in blocks])
    assert torch.equal(fused, expect)
    sums = ops.block_token_norm_sums(dv, ids, nblk, layout, H, D, bs, block_stride=stride)
    osums = torch.cat([oracle.block_token_norm_sums(b, bs, H, D, layout) for b in blocks])
    assert torch.equal(sums.cpu(), osums)
    two = ops.block_score_from_token_sums(sums, nblk, bs, H, dtype).cpu()
    assert torch.equal(two, expect)
    for ranks in (2, 4):
        hl = H // ranks
        total = torch.zeros(nblk * bs, dtype=torch.float32, device="cuda")
        for r in range(ranks):
            if layout == 0:
                shard = dv[:, r * hl:(r + 1) * hl].contiguous()
                st = None
            else:
                shard = dv[:, r * hl:(r + 1) * hl].contiguous()
                st = shard.stride(0)
            total += ops.block_token_norm_sums(shard, ids, nblk, layout, hl, D, bs, block_stride=st)
        got = ops.block_score_from_token_sums(total, nblk, bs, H, dtype).cpu()
        assert torch.equal(got, expect), ranks
        assert oracle.block_score_from_token_sums(total[:bs].cpu(), H, dtype) == float(expect[0])


def test_evict_select_compact_with_caller_scores(oracle):
    """new_scores replaces the in-launch scoring of the just-filled block: same victims, tables
    and score bookkeeping as the launch that scores from the value cache."""
    from vllm_paged_compression_amd import ops
    g = torch.Generator().manual_seed(9)
    B, H, D, bs, nb, M = 4, 2, 64, 16, 64, 10
    vc = (torch.rand(nb, H, D, bs, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    tables = torch.stack([torch.randperm(nb, generator=g)[:M] for _ in range(B)]).to(torch.int32).cuda()
    nblk = torch.tensor([10, 9, 10, 8], dtype=torch.int32).cuda()
    seq_lens = torch.tensor([160, 144, 150, 128], dtype=torch.int32).cuda()   # row 2 not on a boundary
    budget = 96

    def fresh():
        t = torch.rand(B, 32, generator=torch.Generator().manual_seed(1)).cuda()
        c = (nblk - 1).clone()
        return t, c

    rows = torch.arange(B, dtype=torch.int32).cuda()
    t1, c1 = fresh()
    r1, o1, _ = ops.evict_select_compact(tables, seq_lens, nblk, "global", budget, bs, 3,
                                         score_table=t1, score_counts=c1, score_rows=rows,
                                         value_cache=vc, out_width=M - 1)
    last = tables.gather(1, (nblk.long() - 1).unsqueeze(1)).squeeze(1).contiguous()
    sums = ops.block_token_norm_sums(vc, last, B, 1, H, D, bs, block_stride=vc.stride(0))
    ns = ops.block_score_from_token_sums(sums, B, bs, H, torch.bfloat16)
    t2, c2 = fresh()
    r2, o2, _ = ops.evict_select_compact(tables, seq_lens, nblk, "global", budget, bs, 3,
                                         score_table=t2, score_counts=c2, score_rows=rows,
                                         value_cache=None, out_width=M - 1, new_scores=ns,
                                         dtype=torch.bfloat16)
    assert torch.equal(r1, r2) and torch.equal(o1, o2)
    assert torch.equal(t1, t2) and torch.equal(c1, c2)
    assert (r1.cpu() >= 0).sum() == 3 and r1[2].item() == -1


# ------------------------------------------------------------------ rotary embedding + KV write
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("is_neox", [True, False])
@pytest.mark.parametrize("Hq,Hkv,D,rot", [(32, 8, 128, 128), (12, 12, 64, 64), (8, 2, 128, 64)])
def test_rope_reshape_and_cache(ops, oracle, dtype, is_neox, Hq, Hkv, D, rot):
    """Fused rotary embedding + KV write: Q and K rotated in place exactly as the oracle rotates
    them (storage-type rounding of every product and sum), rotated K and V in the paged cache
    byte for byte; tokens without a slot are rotated but not cached; q/k/v are views of one fused
    qkv tensor (strided rows)."""
    from test_oracle_golden import rope_cache
    g = torch.Generator().manual_seed(Hq + D + rot + int(is_neox))
    T, bs, nb, max_pos = 53, 16, 9, 2048
    cs = rope_cache(max_pos, rot, dtype)
    pos = torch.randint(0, max_pos, (T,), generator=g)
    qkv = (torch.rand(T, (Hq + 2 * Hkv) * D, generator=g) * 2 - 1).to(dtype)
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    perm = torch.randperm(nb * bs, generator=g)
    slots = perm[(perm < 32) | (perm >= 48)][:T].to(torch.int64)   # every slot at most once
    slots[5] = -1
    slots[20:36] = torch.arange(32, 48)           # an aligned run: the vectorised V store path
    x = 16 // qkv.element_size()
    kc, vc = make_cache(nb, Hkv, D, bs, dtype, g)
    q_ref = oracle.rotary_embedding(pos, q, D, cs, is_neox)
    k_ref = oracle.rotary_embedding(pos, k, D, cs, is_neox)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    oracle.reshape_and_cache(k_ref.reshape(T, Hkv, D), v.reshape(T, Hkv, D).contiguous(), kc_ref, vc_ref, slots)
    qkv_d = qkv.cuda()
    qd, kd, vd = qkv_d.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    kc_d, vc_d = kc.cuda(), vc.cuda()
    ops.rotary_embedding_and_cache(pos.cuda(), qd, kd, vd, kc_d, vc_d, slots.cuda(), D, cs.cuda(), is_neox)
    if dtype == torch.float32:
        torch.testing.assert_close(qd.cpu(), q_ref, atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(kd.cpu(), k_ref, atol=1e-6, rtol=1e-6)
        torch.testing.assert_close(kc_d.cpu(), kc_ref, atol=1e-6, rtol=1e-6)
    else:
        assert torch.equal(qd.cpu(), q_ref) and torch.equal(kd.cpu(), k_ref)
        assert torch.equal(kc_d.cpu(), kc_ref)
    assert torch.equal(vd.cpu(), v)                 # V untouched
    assert torch.equal(vc_d.cpu(), vc_ref)
    # rotation only (the plain rotary_embedding op): same Q/K, no cache involved
    qkv_d2 = qkv.cuda()
    q2, k2, _ = qkv_d2.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    ops.rotary_embedding(pos.cuda(), q2, k2, D, cs.cuda(), is_neox)
    assert torch.equal(q2, qd) and torch.equal(k2, kd)
    # empty batch
    ops.rotary_embedding(pos[:0].cuda(), q2[:0], k2[:0], D, cs.cuda(), is_neox)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("is_neox", [True, False])
@pytest.mark.parametrize("Hq,Hkv,D,rot", [(32, 8, 128, 128), (12, 12, 64, 64), (8, 1, 128, 64),
                                          (8, 2, 128, 48)])
@pytest.mark.parametrize("splits", [0, 3])
@pytest.mark.parametrize("bs", [16, 8, 32])
def test_decode_step_rope(ops, oracle, dtype, is_neox, Hq, Hkv, D, rot, splits, bs, monkeypatch):
    """One launch (rotary embedding + KV write + attention) == the oracle's rotary embedding, KV
    write and paged attention: the cache receives the rotated key byte for byte, the output equals
    the two-launch form (rotary_embedding_and_cache, then decode_step's unfused twin) bit for bit
    and the oracle within the kernel tolerance; query and key stay unrotated.  rot 48 is a partial
    rotation the NeoX pairing cannot keep in one lane: refused, GPT-J pairing serves it."""
    from test_oracle_golden import rope_cache
    if splits:
        monkeypatch.setenv("PE_DECODE_SPLITS", str(splits))
    g = torch.Generator().manual_seed(Hq + D + rot + int(is_neox))
    nb, max_pos = 6400 // bs, 4096
    lens = [1, 16, 17, 32, 33, 700, 784, 2000]       # kv length including the new token
    S = len(lens)
    sc = D ** -0.5
    cs = rope_cache(max_pos, rot, dtype)
    pos = torch.randint(0, max_pos, (S,), generator=g)
    pos[0], pos[1] = 0, max_pos - 1
    kc, vc = make_cache(nb, Hkv, D, bs, dtype, g)
    qkv = ((torch.rand(S, (Hq + 2 * Hkv) * D, generator=g) * 2 - 1) * 4 * sc).to(dtype)
    q, k, v = (t.reshape(S, -1, D) for t in qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
    bt = disjoint_tables(lens, bs, nb, g)
    slots = torch.tensor([int(bt[i, (L - 1) // bs]) * bs + (L - 1) % bs for i, L in enumerate(lens)],
                         dtype=torch.int64)
    slots[2] = -1                                     # padding token: nothing written
    lens_t = torch.tensor(lens, dtype=torch.int32)
    qkv_d = qkv.cuda()
    qd, kd, vd = (t.reshape(S, -1, D) for t in qkv_d.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
    kc_d, vc_d = kc.cuda(), vc.cuda()
    out = torch.empty(S, Hq, D, dtype=dtype, device="cuda")
    args = (out, pos.cuda(), qd, kd, vd, kc_d, vc_d, slots.cuda(), Hkv, sc, bt.cuda(), lens_t.cuda(), bs,
            max(lens), cs.cuda(), is_neox)
    if is_neox and rot % 64:
        assert not ops.decode_step_rope_supported(Hq, Hkv, D, bs, dtype, rot, is_neox)
        with pytest.raises(RuntimeError, match="not supported"):
            ops.decode_step_rope(*args)
        assert torch.equal(kc_d.cpu(), kc)            # refused before anything was written
        return
    assert ops.decode_step_rope_supported(Hq, Hkv, D, bs, dtype, rot, is_neox)
    ops.decode_step_rope(*args)
    assert torch.equal(qkv_d.cpu(), qkv)              # inputs are left unrotated
    # oracle: rotate, write, attend
    q_ref = oracle.rotary_embedding(pos, q.reshape(S, -1), D, cs, is_neox).reshape(S, Hq, D)
    k_ref = oracle.rotary_embedding(pos, k.reshape(S, -1), D, cs, is_neox).reshape(S, Hkv, D)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    oracle.reshape_and_cache(k_ref, v.contiguous(), kc_ref, vc_ref, slots)
    ref = oracle.paged_attention(q_ref, kc_ref, vc_ref, bt, lens_t, sc)
    assert torch.equal(kc_d.cpu(), kc_ref) and torch.equal(vc_d.cpu(), vc_ref)
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=ATOL[dtype], rtol=RTOL[dtype])
    # the separate launches on the same inputs: rotation in place + KV write, then the fused
    # write/attention step on the rotated rows with nothing left to write differently
    qkv_2 = qkv.cuda()
    q2, k2, v2 = (t.reshape(S, -1, D) for t in qkv_2.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
    kc_2, vc_2 = kc.cuda(), vc.cuda()
    ops.rotary_embedding(pos.cuda(), q2, k2, D, cs.cuda(), is_neox)
    out2 = torch.empty_like(out)
    ops.decode_step(out2, q2, k2, v2, kc_2, vc_2, slots.cuda(), Hkv, sc, bt.cuda(), lens_t.cuda(), bs,
                    max(lens))
    assert torch.equal(kc_2, kc_d) and torch.equal(vc_2, vc_d)
    assert torch.equal(out2, out)
    # positions=None: the step's rows of the table handed over directly (what the layer launcher
    # does once per step for all layers): same bits
    kc_3, vc_3 = kc.cuda(), vc.cuda()
    out3 = torch.empty_like(out)
    step_cs = cs.cuda().index_select(0, pos.cuda())
    ops.decode_step_rope(out3, None, qd, kd, vd, kc_3, vc_3, slots.cuda(), Hkv, sc, bt.cuda(), lens_t.cuda(),
                         bs, max(lens), step_cs, is_neox)
    assert torch.equal(out3, out) and torch.equal(kc_3, kc_d) and torch.equal(vc_3, vc_d)


def test_device_diagnostics_report_clamps(oracle):
    """A seq_lens entry beyond the block-table row and a full score row are survived by clamping;
    both must be visible through pe_device_diagnostics (and cleared by reading with reset)."""
    from vllm_paged_compression_amd import ops
    assert ops.device_diagnostics(reset=True) in (0, 1, 2, 3)      # clear whatever earlier tests left
    assert ops.device_diagnostics(reset=False) == 0
    g = torch.Generator().manual_seed(2)
    Hq, Hkv, D, nb, M = 8, 2, 128, 16, 4
    kc = torch.rand(nb, Hkv, D // 8, 16, 8, generator=g).to(torch.bfloat16).cuda()
    vc = torch.rand(nb, Hkv, D, 16, generator=g).to(torch.bfloat16).cuda()
    q = torch.rand(2, Hq, D, generator=g).to(torch.bfloat16).cuda()
    bt = torch.arange(8, dtype=torch.int32).reshape(2, M).cuda()
    out = torch.empty_like(q)
    ok_lens = torch.tensor([64, 10], dtype=torch.int32).cuda()
    ops.paged_attention_v1(out, q, kc, vc, Hkv, D ** -0.5, bt, ok_lens, 16, 64, None)
    assert ops.device_diagnostics(reset=False) == 0
    bad_lens = torch.tensor([64, 65], dtype=torch.int32).cuda()       # 65 > 4 blocks x 16
    ops.paged_attention_v1(out, q, kc, vc, Hkv, D ** -0.5, bt, bad_lens, 16, 65, None)
    assert ops.device_diagnostics(reset=False) == ops.DIAG_DECODE_LEN_CLAMPED
    with pytest.raises(RuntimeError, match="beyond its block-table row"):
        ops.check_device_diagnostics()
    assert ops.device_diagnostics(reset=False) == 0
    # clamped result == result at the table's capacity
    ref = oracle.paged_attention(q.cpu(), kc.cpu(), vc.cpu(), bt.cpu(), torch.tensor([64, 64]), D ** -0.5)
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=1e-3, rtol=1.6e-2)
    # score row of width 4 already holding 4 scores: the 5th is dropped and flagged
    width = 4
    table = torch.tensor([[3.0, 1.0, 2.0, 5.0]], dtype=torch.float32).cuda()
    counts = torch.tensor([4], dtype=torch.int32).cuda()
    rows = torch.tensor([0], dtype=torch.int32).cuda()
    bt1 = torch.arange(10, dtype=torch.int32).reshape(1, 10).cuda()
    rmv, tabs, _ = ops.evict_select_compact(
        bt1, torch.tensor([160], dtype=torch.int32).cuda(), torch.tensor([10], dtype=torch.int32).cuda(),
        "global", 128, 16, score_table=table, score_counts=counts, score_rows=rows, value_cache=vc[:, :, :, :],
        out_width=9)
    assert width == table.shape[1]
    assert ops.device_diagnostics(reset=True) == ops.DIAG_SCORE_ROW_FULL
    assert int(rmv[0]) == 1          # arg-min over [1, n-1) of the stored scores
    ops.check_device_diagnostics()   # clean again


def test_decode_workspace_advertised_size_always_suffices(monkeypatch):
    """pe_paged_decode_workspace_bytes must cover whatever split count the launch picks - with the
    library's own choice and with every override of the split / wave tunables."""
    from vllm_paged_compression_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    Hq, Hkv, D = 8, 2, 128
    nb = 64
    kc = torch.rand(nb, Hkv, D // 8, 16, 8, generator=g).to(torch.bfloat16).cuda()
    vc = torch.rand(nb, Hkv, D, 16, generator=g).to(torch.bfloat16).cuda()
    st = torch.cuda.current_stream().cuda_stream
    checked = 0
    for forced in (None, "1", "2", "7", "64"):
        for waves in (None, "8", "16"):
            for env, val in (("PE_DECODE_SPLITS", forced), ("PE_DECODE_WAVES", waves)):
                if val is None:
                    monkeypatch.delenv(env, raising=False)
                else:
                    monkeypatch.setenv(env, val)
            for S in (1, 3, 40):
                for kv in (1, 31, 64, 129, 300, 776, 1000, 2055, 4104, 9000):
                    M = (kv + 15) // 16
                    q = torch.rand(S, Hq, D, generator=g).to(torch.bfloat16).cuda()
                    bt = torch.randint(0, nb, (S, M), generator=g, dtype=torch.int32).cuda()
                    lens = torch.full((S,), kv, dtype=torch.int32).cuda()
                    out = torch.empty_like(q)
                    need = lib.pe_paged_decode_workspace_bytes(S, Hq, Hkv, D, 16, kv)
                    ws = torch.zeros(max(need, 1), dtype=torch.uint8, device="cuda")
                    rc = lib.pe_paged_decode(out.data_ptr(), q.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                                             bt.data_ptr(), lens.data_ptr(), None, S, Hq, Hkv, D, 16, M,
                                             q.stride(0), kc.stride(0), kc.stride(1), D ** -0.5, kv,
                                             _lib.PE_BF16, ws.data_ptr() if need else None, need, st)
                    assert rc == 0, (forced, waves, S, kv, need, lib.pe_last_error())
                    checked += 1
    torch.cuda.synchronize()
    assert checked == 5 * 3 * 3 * 10


@pytest.mark.parametrize("method,topk,drop", [("streamingLLM", -1, True), ("streamingLLM-1", -1, True),
                                             ("global", -1, True), ("global", -1, False),
                                             ("local", 5, True), ("local", 3, False)])
@pytest.mark.parametrize("inplace", [False, True])
def test_evict_select_kernels_agree(ops, method, topk, drop, inplace, monkeypatch):
    """The register-resident form of the decision kernel and the general one (PE_EVICT_FAST=0)
    leave the same bits everywhere: victims, tables, score lists and counts, zeroed slots, freed
    block ids, block counts and lengths - on rows that evict, rows that do not, full score rows,
    host-chosen victims and tables near the width limit of the fast form."""
    import zlib
    # (a seed that does not depend on the interpreter's per-process string hashing)
    g = torch.Generator().manual_seed(zlib.crc32(repr((method, topk, drop, inplace)).encode()))
    bs, Hkv, D = 16, 4, 64
    for B, M, budget in [(7, 40, 320), (3, 300, 2048), (5, 2048, 4096), (4, 2100, 4096)]:
        nb = 64
        vc = (torch.rand(nb, Hkv, D, bs, generator=g) - 0.5).to(torch.bfloat16).cuda()
        nblk = torch.randint(budget // bs, min(M, budget // bs + 30) + 1, (B,), generator=g, dtype=torch.int32)
        nblk = torch.clamp(nblk, max=M)
        nblk[0] = min(M, budget // bs + 1)     # at least row 0 is past the budget and evicts
        # half the rows sit on a block boundary above the budget (they evict), the others do not
        seq = nblk * bs - torch.tensor([0 if i % 2 == 0 else 5 for i in range(B)], dtype=torch.int32)
        tables = torch.randint(0, nb, (B, M), generator=g, dtype=torch.int32)
        width = M + 2 if M < 2050 else M
        scores = torch.rand(B + 2, width, generator=g).to(torch.bfloat16).float()
        counts = torch.zeros(B + 2, dtype=torch.int32)
        rows = torch.randperm(B + 2, generator=g)[:B].to(torch.int32)
        for i in range(B):
            c = int(nblk[i]) - 1
            counts[int(rows[i])] = width if i == 2 else c     # row 2: the score row is already full
        ov = None
        if method == "global" and not drop:
            ov = torch.full((B,), -1, dtype=torch.int32)
            ov[0] = 3
        results = []
        for fast in ("1", "0"):
            monkeypatch.setenv("PE_EVICT_FAST", fast)
            st, ct = scores.clone().cuda(), counts.clone().cuda()
            kw = dict(victim_override=None if ov is None else ov.cuda(), drop_victim_score=drop,
                      score_table=st if method in ("local", "global") else None,
                      score_counts=ct if method in ("local", "global") else None,
                      score_rows=rows.cuda() if method in ("local", "global") else None,
                      value_cache=vc if method in ("local", "global") else None)
            if inplace:
                tb, nbk, kvl = tables.clone().cuda(), nblk.clone().cuda(), (seq.clone() - 1).cuda()
                rmv, freed, zero = ops.evict_select_compact_inplace(tb, nbk, kvl, seq.cuda(), B, method,
                                                                    budget, bs, topk, **kw)
                out = [rmv, freed, zero, tb, nbk, kvl, st, ct]
            else:
                rmv, tabs, zero = ops.evict_select_compact(tables.cuda(), seq.cuda(), nblk.cuda(), method,
                                                           budget, bs, topk, out_width=M, **kw)
                out = [rmv, tabs, zero, st, ct]
            torch.cuda.synchronize()
            ops.device_diagnostics(reset=True)    # the full score row of row 2 is reported, on purpose
            results.append([None if t is None else t.cpu() for t in out])
        for a, b in zip(*results):
            assert (a is None) == (b is None)
            if a is not None:
                assert torch.equal(a, b), (method, B, M)
        assert (results[0][0] >= 0).any() or method == "streamingLLM-1"


@pytest.mark.parametrize("inplace", [False, True])
def test_evict_select_override_with_caller_scores(ops, inplace, monkeypatch):
    """Caller-chosen victim + score table + caller-supplied new scores: the one combination in
    which the fast kernel reaches the score-list shift without passing a barrier of the scoring or
    arg-min branches.  Score rows of several hundred entries (the shift crosses wave boundaries),
    both kernels must leave the same bits."""
    g = torch.Generator().manual_seed(77)
    bs, Hkv, D, nb = 16, 4, 64, 64
    for B, M, budget in [(6, 300, 2048), (3, 1500, 8192)]:
        nblk = torch.full((B,), budget // bs + 1, dtype=torch.int32)
        nblk[1] = min(M, budget // bs + 40)
        seq = nblk * bs
        tables = torch.randint(0, nb, (B, M), generator=g, dtype=torch.int32)
        width = M + 2
        scores = torch.rand(B, width, generator=g)
        rows = torch.randperm(B, generator=g).to(torch.int32)
        counts = torch.zeros(B, dtype=torch.int32)
        counts[rows.long()] = nblk - 1                                         # indexed by score row
        ov = torch.tensor([3, 70, 128, 1, 127, 64][:B], dtype=torch.int32)     # around wave boundaries
        new_scores = torch.rand(B, generator=g)
        results = []
        for fast in ("1", "0"):
            monkeypatch.setenv("PE_EVICT_FAST", fast)
            st, ct = scores.clone().cuda(), counts.clone().cuda()
            kw = dict(victim_override=ov.cuda(), drop_victim_score=True, score_table=st, score_counts=ct,
                      score_rows=rows.cuda(), new_scores=new_scores.cuda())
            if inplace:
                tb, nbk, kvl = tables.clone().cuda(), nblk.clone().cuda(), (seq.clone() - 1).cuda()
                rmv, freed, zero = ops.evict_select_compact_inplace(tb, nbk, kvl, seq.cuda(), B, "global",
                                                                    budget, bs, -1, **kw)
                out = [rmv, freed, tb, nbk, kvl, st, ct]
            else:
                rmv, tabs, zero = ops.evict_select_compact(tables.cuda(), seq.cuda(), nblk.cuda(), "global",
                                                           budget, bs, -1, out_width=M, **kw)
                out = [rmv, tabs, st, ct]
            torch.cuda.synchronize()
            results.append([t.cpu() for t in out])
        for a, b in zip(*results):
            assert torch.equal(a, b), (B, M)
        assert results[0][0].tolist() == ov.tolist()
        # and against a plain model of the list: append the new score, delete the victim's entry
        st = results[0][-2]
        for i in range(B):
            r = int(rows[i])
            want = scores[r, :int(counts[r])].tolist() + [float(new_scores[i])]
            del want[int(ov[i])]
            assert st[r, :len(want)].tolist() == want and int(results[0][-1][r]) == len(want)
    ops.check_device_diagnostics()


def test_score_formulas_golden(ops, oracle):
    """global/local with the other score formulas (KVCachePruner.sub_evict_method,
    kvcache_prunner.py:29-66) against the reference's outputs: token-level prune through the
    kernels, block scores through the pruner (dense rows) and from the paged key / value caches."""
    from test_oracle_golden import FORMULAS
    from vllm_paged_compression_amd.pruner import KVCachePruner
    z = np.load(os.path.join(GOLDEN, "score_formulas.npz"))
    for t in range(2):
        L, budget, bs, H, D = [int(v) for v in z[f"t{t}_meta"]]
        k, v = from_bits(z[f"t{t}_k"], 0), from_bits(z[f"t{t}_v"], 0)
        n_kept = oracle.pruned_length(L, budget, bs)
        cu_q, cu_k = _cu([L]).cuda(), _cu([n_kept]).cuda()
        kd, vd = k.cuda(), v.cuda()
        for fi, f in enumerate(FORMULAS):
            idx = ops.prompt_token_select(kd, vd, cu_q, cu_k, n_kept, budget, bs, "global",
                                          compat_q2=True, formula=f)
            pk, pv = ops.prompt_prune_gather(kd, vd, idx, mode=2)
            assert torch.equal(pk.cpu(), from_bits(z[f"t{t}_f{fi}_pk"], 0)), (t, f)
            assert torch.equal(pv.cpu(), from_bits(z[f"t{t}_f{fi}_pv"], 0)), (t, f)
            # per-head compaction through the pruner's own API
            p = KVCachePruner("budget", "global", bs, budget, 1)
            p.sub_evict_method = f
            pk, pv = p.prune_prompt(kd, vd)
            ek, ev = oracle.prompt_prune(k, v, "global", budget, bs, formula=f)
            assert torch.equal(pk.cpu(), ek) and torch.equal(pv.cpu(), ev), (t, f)
            assert torch.equal(p.get_token_score(k, v, "global"), oracle.token_scores(k, v, "global", f))
    for b in range(8):
        bs, H, D, dt = [int(v) for v in z[f"b{b}_meta"]]
        k, v = from_bits(z[f"b{b}_k"], dt), from_bits(z[f"b{b}_v"], dt)
        x = 16 // k.element_size()
        # the same logical rows in the paged layouts: K [H, D/x, bs, x], V [H, D, bs]
        kc = k.permute(1, 2, 0).reshape(H, D // x, x, bs).permute(0, 1, 3, 2).contiguous().cuda()
        vc = v.permute(1, 2, 0).contiguous().cuda()
        for fi, f in enumerate(FORMULAS):
            expect = from_bits(z[f"b{b}_f{fi}_score"], dt).float().item()
            p = KVCachePruner("budget", "global", bs, 4 * bs, 1)
            p.sub_evict_method = f
            got = p.get_block_score(k.cuda(), v.cuda(), "global").float().item()
            ks = ops.block_score_value_l2(kc, None, 1, ops.SCORE_KEY_CACHE, H, D, bs)
            vs = ops.block_score_value_l2(vc, None, 1, ops.SCORE_CACHE, H, D, bs)
            got_c = ops.combine_block_scores(f, vs, ks, k.dtype).item()
            if dt == 0:
                assert got == pytest.approx(expect, rel=3e-6) and got_c == pytest.approx(expect, rel=3e-6)
            else:
                assert got == expect and got_c == expect, (b, f, got, got_c, expect)
        # head-separable form of the key norms (tensor-parallel scoring) on the paged key block
        sums = ops.block_token_norm_sums(kc, None, 1, ops.SCORE_KEY_CACHE, H, D, bs)
        two = ops.block_score_from_token_sums(sums, 1, bs, H, k.dtype).item()
        one = ops.block_score_value_l2(kc, None, 1, ops.SCORE_KEY_CACHE, H, D, bs).item()
        assert two == one
