"""Checks the C restatement against the reference's CPU kernels compiled from the
reference's sources (oracle/_ref/libpe_ref.so, built by oracle/build_ref.sh).
Skipped when that library is absent."""
import os

import pytest
import torch

from conftest import ROOT

REF = os.path.join(ROOT, "oracle", "_ref", "libpe_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def ref():
    torch.ops.load_library(REF)
    return torch.ops.pe_ref


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("H,D", [(8, 128), (12, 64), (2, 80)])
def test_reshape_and_cache_exact(oracle, ref, dtype, H, D):
    # recipe of the reference's tests/kernels/test_cache.py:111-202
    g = torch.Generator().manual_seed(0)
    nb, bs, T = 40, 16, 57
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    k = (torch.rand(T, H, D, generator=g) - 0.5).to(dtype)
    v = (torch.rand(T, H, D, generator=g) - 0.5).to(dtype)
    slots = torch.randperm(nb * bs, generator=g)[:T].to(torch.int64)
    slots[7] = -1
    kc = (torch.rand(nb, H, D // x, bs, x, generator=g) - 0.5).to(dtype)
    vc = (torch.rand(nb, H, D, bs, generator=g) - 0.5).to(dtype)
    kc2, vc2 = kc.clone(), vc.clone()
    oracle.reshape_and_cache(k, v, kc, vc, slots)
    ref.reshape_and_cache(k, v, kc2, vc2, slots, "auto", 1.0, 1.0)
    assert torch.equal(kc, kc2) and torch.equal(vc, vc2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("Hq,Hkv,D", [(32, 8, 128), (12, 12, 64), (8, 1, 128)])
def test_paged_attention_close(oracle, ref, dtype, Hq, Hkv, D):
    g = torch.Generator().manual_seed(1)
    nb, bs = 96, 16
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    sc = D ** -0.5
    kc = ((torch.rand(nb, Hkv, D // x, bs, x, generator=g) * 2 - 1) * sc).to(dtype)
    vc = ((torch.rand(nb, Hkv, D, bs, generator=g) * 2 - 1) * sc).to(dtype)
    lens = torch.tensor([1, 15, 16, 17, 300, 777], dtype=torch.int32)
    S = len(lens)
    q = ((torch.rand(S, Hq, D, generator=g) * 2 - 1) * sc).to(dtype)
    M = (777 + bs - 1) // bs
    bt = torch.randint(0, nb, (S, M), generator=g, dtype=torch.int32)
    out = oracle.paged_attention(q, kc, vc, bt, lens, sc)
    o2 = torch.zeros_like(q)
    ref.paged_attention_v1(o2, q, kc, vc, Hkv, sc, bt, lens, bs, 777, None, "auto", 1.0, 1.0,
                           0, 0, 0, 64, 0)
    atol = 1e-3 if dtype != torch.float32 else 1e-5
    torch.testing.assert_close(out.float(), o2.float(), atol=atol, rtol=1.6e-2)


def test_copy_blocks_exact(oracle, ref):
    g = torch.Generator().manual_seed(2)
    kc = torch.rand(10, 2, 8, 16, 8, generator=g).to(torch.bfloat16)
    vc = torch.rand(10, 2, 64, 16, generator=g).to(torch.bfloat16)
    kc2, vc2 = kc.clone(), vc.clone()
    mapping = torch.tensor([[0, 5], [0, 6], [3, 9]], dtype=torch.int64)
    oracle.copy_blocks(kc, vc, mapping)
    ref.copy_blocks([kc2], [vc2], mapping)
    assert torch.equal(kc, kc2) and torch.equal(vc, vc2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("neox", [True, False])
@pytest.mark.parametrize("Hq,Hkv,D,rot", [(32, 8, 128, 128), (12, 12, 64, 64), (8, 1, 128, 64)])
def test_rotary_embedding_close(oracle, ref, dtype, neox, Hq, Hkv, D, rot):
    """oracle.rotary_embedding (storage-type arithmetic of csrc/pos_encoding_kernels.cu:10-35)
    against the reference's CPU kernel (f32 arithmetic, one rounding, csrc/cpu/pos_encoding.cpp):
    within two units in the last place of the storage type."""
    g = torch.Generator().manual_seed(11)
    T, max_pos = 67, 4096
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, dtype=torch.float) / rot))
    freqs = torch.einsum("i,j -> ij", torch.arange(max_pos, dtype=torch.float), inv_freq)
    cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype)
    qkv = ((torch.rand(T, (Hq + 2 * Hkv) * D, generator=g) * 2 - 1) * 2).to(dtype)
    pos = torch.randint(0, max_pos, (T,), generator=g, dtype=torch.int64)
    q, k = qkv[:, :Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D]
    got_q = oracle.rotary_embedding(pos, q, D, cache, neox)
    got_k = oracle.rotary_embedding(pos, k, D, cache, neox)
    ref.rotary_embedding(pos, q, k, D, cache, neox)      # in place on the fused buffer's views
    atol = {torch.float32: 1e-6, torch.float16: 4e-3, torch.bfloat16: 3.2e-2}[dtype]
    torch.testing.assert_close(got_q.float(), q.float(), atol=atol, rtol=0)
    torch.testing.assert_close(got_k.float(), k.float(), atol=atol, rtol=0)
