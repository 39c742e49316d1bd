"""The operators registered as torch.ops.paged_evict.* (the reference's schemas,
csrc/torch_bindings.cpp:27-51,430-468) and the selector hook, driven the way a caller of
torch.ops._C / torch.ops._C_cache_ops drives them, against the oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BS = 16


def _caches(nb, Hkv, D, dtype, g):
    sc = D ** -0.5
    kc = ((torch.rand(nb, Hkv, D // (16 // dtype.itemsize), BS, 16 // dtype.itemsize, generator=g) * 2 - 1)
          * sc * 3).to(dtype)
    vc = (torch.rand(nb, Hkv, D, BS, generator=g) * 2 - 1).to(dtype)
    return kc, vc


@pytest.fixture(scope="module")
def pe_ops():
    from vllm_paged_compression_amd import ops
    ops.register_torch_ops()
    ops.register_torch_ops()      # idempotent
    return torch.ops.paged_evict


def test_schemas_match_reference(pe_ops):
    """Argument names, order and mutability as csrc/torch_bindings.cpp declares them."""
    def names(op):
        return [a.name for a in op.default._schema.arguments]
    assert names(pe_ops.paged_attention_v1) == [
        "out", "query", "key_cache", "value_cache", "num_kv_heads", "scale", "block_tables", "seq_lens",
        "block_size", "max_seq_len", "alibi_slopes", "kv_cache_dtype", "k_scale", "v_scale", "tp_rank",
        "blocksparse_local_blocks", "blocksparse_vert_stride", "blocksparse_block_size",
        "blocksparse_head_sliding_step"]
    assert names(pe_ops.paged_attention_v2)[:4] == ["out", "exp_sums", "max_logits", "tmp_out"]
    assert names(pe_ops.paged_attention_v2)[4:] == names(pe_ops.paged_attention_v1)[1:]
    assert names(pe_ops.reshape_and_cache) == ["key", "value", "key_cache", "value_cache",
                                               "slot_mapping", "kv_cache_dtype", "k_scale", "v_scale"]
    assert names(pe_ops.copy_blocks) == ["key_caches", "value_caches", "block_mapping"]
    assert names(pe_ops.swap_blocks) == ["src", "dst", "block_mapping"]
    assert names(pe_ops.rotary_embedding) == ["positions", "query", "key", "head_size", "cos_sin_cache",
                                              "is_neox"]
    ro = pe_ops.rotary_embedding.default._schema
    assert [a.alias_info is not None and a.alias_info.is_write for a in ro.arguments[:3]] == \
        [False, True, True]
    sch = pe_ops.paged_attention_v1.default._schema
    assert sch.arguments[0].alias_info is not None and sch.arguments[0].alias_info.is_write
    assert str(sch.arguments[10].type) == "Optional[Tensor]"
    rc = pe_ops.reshape_and_cache.default._schema
    assert [a.alias_info is not None and a.alias_info.is_write for a in rc.arguments[:5]] == \
        [False, False, True, True, False]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("use_alibi", [False, True])
def test_registered_paged_attention_v1_v2(oracle, pe_ops, dtype, use_alibi):
    g = torch.Generator().manual_seed(5)
    S, Hq, Hkv, D, nb, M = 5, 16, 4, 128, 300, 40
    kc, vc = _caches(nb, Hkv, D, dtype, g)
    q = ((torch.rand(S, Hq, D, generator=g) * 2 - 1) * D ** -0.5 * 3).to(dtype)
    bt = torch.randperm(nb, generator=g)[:S * M].to(torch.int32).reshape(S, M)
    lens = torch.tensor([1, 17, 333, 640, 97], dtype=torch.int32)
    slopes = (torch.rand(Hq, generator=g) * 0.1).float() if use_alibi else None
    ref = oracle.paged_attention(q, kc, vc, bt, lens, D ** -0.5, slopes)
    dq, dkc, dvc, dbt, dl = (t.cuda() for t in (q, kc, vc, bt, lens))
    ds = None if slopes is None else slopes.cuda()
    out1 = torch.zeros_like(dq)
    pe_ops.paged_attention_v1(out1, dq, dkc, dvc, Hkv, D ** -0.5, dbt, dl, BS, 640, ds, "auto",
                              1.0, 1.0, 0, 0, 0, 64, 0)
    torch.testing.assert_close(out1.cpu().float(), ref.float(), atol=1e-3, rtol=1.6e-2)
    # v2: the caller's partition buffers are accepted and left alone (partials live in the library's
    # workspace); same result
    P = 2
    exp_sums = torch.full((S, Hq, P), 7.0, device="cuda")
    max_logits = torch.full((S, Hq, P), 7.0, device="cuda")
    tmp_out = torch.full((S, Hq, P, D), 7.0, dtype=dtype, device="cuda")
    out2 = torch.zeros_like(dq)
    pe_ops.paged_attention_v2(out2, exp_sums, max_logits, tmp_out, dq, dkc, dvc, Hkv, D ** -0.5, dbt,
                              dl, BS, 640, ds, "auto", 1.0, 1.0, 0, 0, 0, 64, 0)
    assert torch.equal(out1, out2)
    assert float(exp_sums.min()) == 7.0 and float(tmp_out.float().min()) == 7.0
    # the reference's error conventions: fp8 KV and scales other than 1 are refused
    with pytest.raises(NotImplementedError):
        pe_ops.paged_attention_v1(out1, dq, dkc, dvc, Hkv, D ** -0.5, dbt, dl, BS, 640, ds, "fp8",
                                  1.0, 1.0, 0, 0, 0, 64, 0)
    with pytest.raises(RuntimeError):
        pe_ops.paged_attention_v1(out1, dq, dkc, dvc, Hkv, D ** -0.5, dbt, dl, BS, 640, ds, "auto",
                                  0.5, 1.0, 0, 0, 0, 64, 0)


def test_registered_cache_ops(oracle, pe_ops):
    g = torch.Generator().manual_seed(9)
    dtype = torch.bfloat16
    Hkv, D, nb, T, L = 4, 128, 24, 37, 3
    kcs, vcs = zip(*[_caches(nb, Hkv, D, dtype, g) for _ in range(L)])
    k = torch.rand(T, Hkv, D, generator=g).to(dtype)
    v = torch.rand(T, Hkv, D, generator=g).to(dtype)
    slots = torch.randperm(nb * BS, generator=g)[:T].to(torch.int64)
    slots[3] = -1                       # padding slot: skipped
    dk = [t.clone().cuda() for t in kcs]
    dv = [t.clone().cuda() for t in vcs]
    ck = [t.clone() for t in kcs]
    cv = [t.clone() for t in vcs]
    # reshape_and_cache
    pe_ops.reshape_and_cache(k.cuda(), v.cuda(), dk[0], dv[0], slots.cuda(), "auto", 1.0, 1.0)
    oracle.reshape_and_cache(k, v, ck[0], cv[0], slots)
    assert torch.equal(dk[0].cpu(), ck[0]) and torch.equal(dv[0].cpu(), cv[0])
    # copy_blocks over all layers (lists of tensors, mutated in place)
    mapping = torch.tensor([[0, 5], [0, 9], [7, 2]], dtype=torch.int64)
    pe_ops.copy_blocks(dk, dv, mapping.cuda())
    for li in range(L):
        oracle.copy_blocks(ck[li], cv[li], mapping)
        assert torch.equal(dk[li].cpu(), ck[li]) and torch.equal(dv[li].cpu(), cv[li])
    # swap_blocks: device -> host -> device, mapping on the CPU as the reference requires
    kv_dev = torch.stack([dk[1].reshape(nb, -1), dv[1].reshape(nb, -1)])         # [2, nb, block elems]
    kv_host = torch.zeros(2, 8, kv_dev.shape[2], dtype=dtype).pin_memory()
    out_map = torch.tensor([[3, 0], [11, 6]], dtype=torch.int64)
    for half in range(2):
        pe_ops.swap_blocks(kv_dev[half], kv_host[half], out_map)
    torch.cuda.synchronize()
    assert torch.equal(kv_host[0][0], dk[1].reshape(nb, -1)[3].cpu())
    assert torch.equal(kv_host[1][6], dv[1].reshape(nb, -1)[11].cpu())
    back = torch.zeros_like(kv_dev)
    in_map = torch.tensor([[0, 20], [6, 1]], dtype=torch.int64)
    for half in range(2):
        pe_ops.swap_blocks(kv_host[half], back[half], in_map)
    torch.cuda.synchronize()
    assert torch.equal(back[0][20].cpu(), kv_host[0][0]) and torch.equal(back[1][1].cpu(), kv_host[1][6])
    with pytest.raises(RuntimeError, match="block_mapping must be on CPU"):
        pe_ops.swap_blocks(kv_dev[0], back[0], out_map.cuda())


@pytest.mark.parametrize("is_neox", [True, False])
def test_registered_rotary_embedding(oracle, pe_ops, is_neox):
    """torch.ops.paged_evict.rotary_embedding rotates query and key in place as the oracle does."""
    from test_oracle_golden import rope_cache
    g = torch.Generator().manual_seed(11)
    T, Hq, Hkv, D = 37, 8, 2, 128
    cs = rope_cache(512, D, torch.bfloat16)
    pos = torch.randint(0, 512, (T,), generator=g)
    q = (torch.rand(T, Hq * D, generator=g) * 2 - 1).to(torch.bfloat16)
    k = (torch.rand(T, Hkv * D, generator=g) * 2 - 1).to(torch.bfloat16)
    qd, kd = q.cuda(), k.cuda()
    pe_ops.rotary_embedding(pos.cuda(), qd, kd, D, cs.cuda(), is_neox)
    assert torch.equal(qd.cpu(), oracle.rotary_embedding(pos, q, D, cs, is_neox))
    assert torch.equal(kd.cpu(), oracle.rotary_embedding(pos, k, D, cs, is_neox))


def test_selector_hook_and_backend_statics():
    """get_attn_backend_for_eviction is what the fork's selector calls in place of its TORCH_SDPA
    branch (vllm/attention/selector.py:138-146); the class it returns carries the static API of
    vllm/attention/backends/abstract.py:32-99."""
    from vllm_paged_compression_amd import backend as be
    from vllm_paged_compression_amd.config import CacheConfigView, PagedEvictConfig
    assert be.get_attn_backend_for_eviction(None) is None
    assert be.get_attn_backend_for_eviction(CacheConfigView(BS, None)) is None
    cfg = PagedEvictConfig(True, "budget", "streamingLLM", 256, -1)
    cls = be.get_attn_backend_for_eviction(CacheConfigView(BS, cfg))
    assert cls is be.PagedEvictBackend and cls.get_name() == "PAGED_EVICT_HIP"
    assert cls.get_impl_cls() is be.PagedEvictImpl and cls.get_metadata_cls() is be.PagedEvictMetadata
    assert cls.get_builder_cls() is be.PagedEvictMetadataBuilder
    assert cls.get_kv_cache_shape(10, BS, 8, 128) == (2, 10, BS * 8 * 128)
    # copy_blocks / swap_blocks of the backend on [2, nb, elems] caches
    g = torch.Generator().manual_seed(1)
    kv = [torch.rand(2, 6, BS * 2 * 64, generator=g).half().cuda() for _ in range(2)]
    before = [t.clone() for t in kv]
    cls.copy_blocks(kv, torch.tensor([[1, 4]], dtype=torch.int64).cuda())
    for t, b in zip(kv, before):
        assert torch.equal(t[:, 4], b[:, 1]) and torch.equal(t[:, 0], b[:, 0])
    dst = torch.zeros_like(kv[0])
    cls.swap_blocks(kv[0], dst, torch.tensor([[4, 2]], dtype=torch.int64))
    torch.cuda.synchronize()
    assert torch.equal(dst[:, 2], kv[0][:, 4])
