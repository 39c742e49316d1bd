"""The hot path at the full sizes of the benchmark configurations (8B: 64 sequences at kv
4096..4112 and 8192-token prompts; one 16.4k-token sequence), where the scalar oracle is only
run on a small sample.  Besides that sample the kernels are checked through properties that do
not depend on the size:
  * attention is a convex combination: with every V row equal to one vector the output is that
    vector (times sum/(sum+1e-6));
  * it is invariant under a permutation of the KV blocks of a sequence (no positional term);
  * split-KV partitions agree with the single-pass result;
  * causal prefill: rows <= j do not change when K/V rows > j change; its last row agrees with
    decode attention of that query over the paged copy of the same K/V;
  * KV write followed by a gather through the slot mapping returns the input bytes;
  * block-table compaction returns the row minus exactly the victim column;
  * the streamingLLM keep index is sorted, has the pruned length and keeps the first block, the
    budgeted tail of the middle and the last block.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

BS = 16


@pytest.fixture(scope="module")
def ops():
    import vllm_paged_compression_amd as pkg
    import vllm_paged_compression_amd.ops as ops
    pkg.load_library()
    return ops


def _cache(nb, H, D, dtype, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    sc = D ** -0.5
    kc = torch.empty(nb, H, D // 8, BS, 8, dtype=dtype, device="cuda").uniform_(-sc, sc, generator=g)
    vc = torch.empty(nb, H, D, BS, dtype=dtype, device="cuda").uniform_(-sc, sc, generator=g)
    return kc, vc


def _decode(ops, q, kc, vc, bt, lens, max_len):
    out = torch.empty_like(q)
    ops.paged_attention_v1(out, q, kc, vc, kc.shape[1], q.shape[2] ** -0.5, bt, lens, BS, max_len,
                           None, "auto", 1.0, 1.0)
    return out


@pytest.mark.parametrize("B,kv_lo,kv_hi,Hq,Hkv,D,dt,label", [
    (64, 4096, 4112, 32, 8, 128, torch.bfloat16, "8B 8192->1024 at 50%"),
    (2, 16384, 16400, 32, 8, 128, torch.bfloat16, "8B 32768->2048 at 50%"),
    (64, 3072, 3088, 8, 1, 128, torch.bfloat16, "70B TP=8, one rank's heads, 4096->512 at 25%"),
    (1, 1024, 1124, 12, 12, 64, torch.float16, "OPT-125m 1024->100, no eviction")])
def test_decode_full_size_properties(ops, oracle, B, kv_lo, kv_hi, Hq, Hkv, D, dt, label, monkeypatch):
    nblk = kv_hi // BS + 1
    nb = B * nblk + 4
    kc, vc = _cache(nb, Hkv, D, dt, 3)
    g = torch.Generator(device="cuda").manual_seed(4)
    q = torch.empty(B, Hq, D, dtype=dt, device="cuda").uniform_(-D ** -0.5, D ** -0.5, generator=g)
    bt = torch.randperm(nb - 4, device="cuda", generator=g)[:B * nblk].to(torch.int32).reshape(B, nblk)
    lens = torch.randint(kv_lo, kv_hi + 1, (B,), device="cuda", generator=g, dtype=torch.int32)
    lens[0], lens[-1] = kv_lo, kv_hi
    out = _decode(ops, q, kc, vc, bt, lens, kv_hi)
    assert torch.isfinite(out).all()
    # oracle on a sample of the sequences
    for s in sorted({0, B - 1}):
        ref = oracle.paged_attention(q[s:s + 1].cpu(), kc.cpu(), vc.cpu(), bt[s:s + 1].cpu(),
                                     lens[s:s + 1].cpu(), D ** -0.5)
        torch.testing.assert_close(out[s:s + 1].cpu().float(), ref.float(), atol=1e-3, rtol=1.6e-2)
    # split-KV partitions agree with the single pass
    for splits in ("1", "5"):
        monkeypatch.setenv("PE_DECODE_SPLITS", splits)
        o2 = _decode(ops, q, kc, vc, bt, lens, kv_hi)
        torch.testing.assert_close(o2.float(), out.float(), atol=1e-3, rtol=1.6e-2 if dt == torch.bfloat16 else 1e-3)
    monkeypatch.delenv("PE_DECODE_SPLITS")
    # permutation of the KV blocks (whole blocks only: use the block-aligned length)
    lens_al = torch.full_like(lens, kv_lo)
    base = _decode(ops, q, kc, vc, bt, lens_al, kv_lo)
    perm = torch.randperm(kv_lo // BS, device="cuda", generator=g)
    bt_p = bt.clone()
    bt_p[:, :kv_lo // BS] = bt[:, perm]
    permuted = _decode(ops, q, kc, vc, bt_p, lens_al, kv_lo)
    torch.testing.assert_close(permuted.float(), base.float(), atol=1e-3, rtol=1.6e-2 if dt == torch.bfloat16 else 1e-3)
    # convexity: constant V -> the constant comes back (softmax weights sum to 1)
    c = torch.empty(Hkv, D, dtype=dt, device="cuda").uniform_(-1, 1, generator=g)
    vc_c = c.view(1, Hkv, D, 1).expand(nb, Hkv, D, BS).contiguous()
    oc = _decode(ops, q, kc, vc_c, bt, lens, kv_hi).float()
    want = c.float().repeat_interleave(Hq // Hkv, dim=0).unsqueeze(0).expand(B, Hq, D)
    torch.testing.assert_close(oc, want, atol=8e-3, rtol=8e-3)   # one bf16 rounding of P and of O


def _rows_f32(q, k, v, rows, sc, Lkv=None):
    """f32 attention of single query rows (top-left causal: row i sees kv rows 0..min(i, Lkv-1))."""
    Hq, Hkv = q.shape[1], k.shape[1]
    outs = []
    for i in rows:
        n = i + 1 if Lkv is None else min(i + 1, Lkv)
        kk = k[:n].float().repeat_interleave(Hq // Hkv, dim=1)     # [n, Hq, D]
        vv = v[:n].float().repeat_interleave(Hq // Hkv, dim=1)
        s = torch.einsum("hd,nhd->hn", q[i].float(), kk) * sc
        outs.append(torch.einsum("hn,nhd->hd", torch.softmax(s, dim=-1), vv))
    return torch.stack(outs)


@pytest.mark.parametrize("Hq,Hkv,L,Lkv,label", [
    (32, 8, 8192, 4096, "8B 8192->1024 at 50%"),
    (32, 8, 32768, 16384, "8B 32768->2048 at 50%: one launch, 256 query tiles"),
    (8, 1, 4096, 3072, "70B TP=8, one rank's heads, 4096->512 at 25%")])
def test_prefill_full_size_properties(ops, oracle, Hq, Hkv, L, Lkv, label):
    D = 128
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(8)
    sc = D ** -0.5
    q = torch.empty(L, Hq, D, dtype=dt, device="cuda").uniform_(-1, 1, generator=g)
    k = torch.empty(L, Hkv, D, dtype=dt, device="cuda").uniform_(-1, 1, generator=g)
    v = torch.empty(L, Hkv, D, dtype=dt, device="cuda").uniform_(-1, 1, generator=g)
    cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
    out = ops.prefill_causal(q, k, v, cu, cu, L, sc)
    assert torch.isfinite(out).all()
    # oracle on the first rows (row i attends rows <= i only, so a prefix is self-contained)
    n = 200
    ref = oracle.prefill_attention(q[:n].cpu().float(), k[:n].cpu().float(), v[:n].cpu().float(), [n], [n], sc)
    torch.testing.assert_close(out[:n].cpu().float(), ref, atol=2e-2, rtol=1.6e-2)
    # rows sampled over the whole length (tile boundaries, the last rows) against f32 attention
    rows = sorted({63, 64, 127, 128, 1000, L // 3, L // 2 - 1, L // 2, L - 130, L - 65, L - 64, L - 2, L - 1})
    # (outputs of late rows are means over thousands of values, |o| ~ 1e-2: the absolute bar is the
    # decode kernel test's, not the 2e-2 of the short-prompt comparison)
    torch.testing.assert_close(out[rows].float(), _rows_f32(q, k, v, rows, sc), atol=2e-3, rtol=1.6e-2)
    # causality: changing K/V rows > j leaves rows <= j bit-identical
    j = (5 * L) // 8 + 8
    k2, v2 = k.clone(), v.clone()
    k2[j + 1:].uniform_(-3, 3, generator=g)
    v2[j + 1:].uniform_(-3, 3, generator=g)
    out2 = ops.prefill_causal(q, k2, v2, cu, cu, L, sc)
    assert torch.equal(out2[:j + 1], out[:j + 1])
    assert not torch.equal(out2[j + 1:], out[j + 1:])
    del k2, v2, out2
    # last row == decode attention of that query over the paged copy of the same K/V
    nblk = L // BS
    kc = torch.zeros(nblk + 2, Hkv, D // 8, BS, 8, dtype=dt, device="cuda")
    vc = torch.zeros(nblk + 2, Hkv, D, BS, dtype=dt, device="cuda")
    table = torch.randperm(nblk + 2, device="cuda", generator=g)[:nblk].to(torch.int32)
    slots = (table.to(torch.int64).unsqueeze(1) * BS + torch.arange(BS, device="cuda")).reshape(-1)
    ops.reshape_and_cache(k, v, kc, vc, slots)
    dec = _decode(ops, q[L - 1:L].contiguous(), kc, vc, table.unsqueeze(0).contiguous(),
                  torch.tensor([L], dtype=torch.int32, device="cuda"), L)
    torch.testing.assert_close(dec[0].float(), out[L - 1].float(), atol=2e-3, rtol=1.6e-2)
    # pruned-before-attention form (L_kv < L_q, top-left mask): rows < L_kv see the same prefix
    cu_kv = torch.tensor([0, Lkv], dtype=torch.int32, device="cuda")
    outp = ops.prefill_causal(q, k[:Lkv].contiguous(), v[:Lkv].contiguous(), cu, cu_kv, L, sc)
    assert torch.equal(outp[:Lkv], out[:Lkv])
    rows = sorted({Lkv, Lkv + 63, (L + Lkv) // 2, L - 1})
    torch.testing.assert_close(outp[rows].float(), _rows_f32(q, k, v, rows, sc, Lkv), atol=2e-3, rtol=1.6e-2)
    # rows >= L_kv all attend the whole pruned K/V: the last one equals decode over L_kv tokens
    dec2 = _decode(ops, q[L - 1:L].contiguous(), kc, vc, table[:Lkv // BS].unsqueeze(0).contiguous(),
                   torch.tensor([Lkv], dtype=torch.int32, device="cuda"), Lkv)
    torch.testing.assert_close(dec2[0].float(), outp[L - 1].float(), atol=2e-3, rtol=1.6e-2)


def test_prefill_batch_64x8192(ops):
    """One launch over the whole batch of the 8192->1024 configuration (64 prompts of 8192 tokens,
    524,288 query rows): every request holds the same data, so every request's output must be
    bit-identical to the single-request launch (no dependence on the position in the batch), and
    a request with different data in the middle of the batch must not disturb its neighbours."""
    Hq, Hkv, D, L, B = 32, 8, 128, 8192, 64
    dt = torch.bfloat16
    sc = D ** -0.5
    g = torch.Generator(device="cuda").manual_seed(15)
    q1 = torch.empty(L, Hq, D, dtype=dt, device="cuda").uniform_(-1, 1, generator=g)
    k1 = torch.empty(L, Hkv, D, dtype=dt, device="cuda").uniform_(-1, 1, generator=g)
    v1 = torch.empty(L, Hkv, D, dtype=dt, device="cuda").uniform_(-1, 1, generator=g)
    cu1 = torch.tensor([0, L], dtype=torch.int32, device="cuda")
    single = ops.prefill_causal(q1, k1, v1, cu1, cu1, L, sc)
    q = q1.repeat(B, 1, 1)
    k = k1.repeat(B, 1, 1)
    v = v1.repeat(B, 1, 1)
    odd = 37
    k[odd * L:(odd + 1) * L].uniform_(-1, 1, generator=g)
    cu = torch.arange(0, B + 1, dtype=torch.int32, device="cuda") * L
    out = ops.prefill_causal(q, k, v, cu, cu, L, sc).view(B, L, Hq, D)
    for b in range(B):
        if b == odd:
            assert not torch.equal(out[b], single)
        else:
            assert torch.equal(out[b], single), f"request {b} differs from the single-request launch"
    lone = ops.prefill_causal(q1, k[odd * L:(odd + 1) * L], v1, cu1, cu1, L, sc)
    assert torch.equal(out[odd], lone)


def test_kv_write_round_trip_full_size(ops):
    """64 prompts of 8192 tokens pruned to 4096: write through a gather index and a random slot
    permutation, then read every slot back with plain tensor indexing."""
    Hkv, D = 8, 128
    dt = torch.bfloat16
    B, kept = 64, 4096
    T = B * kept
    g = torch.Generator(device="cuda").manual_seed(12)
    nb = T // BS + 16
    k = torch.empty(T, Hkv, D, dtype=dt, device="cuda").uniform_(-1, 1, generator=g)
    v = torch.empty(T, Hkv, D, dtype=dt, device="cuda").uniform_(-1, 1, generator=g)
    kc = torch.zeros(nb, Hkv, D // 8, BS, 8, dtype=dt, device="cuda")
    vc = torch.zeros(nb, Hkv, D, BS, dtype=dt, device="cuda")
    blocks = torch.randperm(nb, device="cuda", generator=g)[:T // BS]
    slots = (blocks.to(torch.int64).unsqueeze(1) * BS + torch.arange(BS, device="cuda")).reshape(-1)
    src = torch.randperm(T, device="cuda", generator=g).to(torch.int32)     # fused row gather
    ops.reshape_and_cache(k, v, kc, vc, slots, src_index=src)
    blk, off = slots // BS, slots % BS
    k_back = kc[blk, :, :, off, :].reshape(T, Hkv, D)          # [T, H, D/8, 8]
    v_back = vc[blk, :, :, off]                                # [T, H, D]
    assert torch.equal(k_back, k[src.long()])
    assert torch.equal(v_back, v[src.long()])
    untouched = torch.ones(nb, dtype=torch.bool, device="cuda")
    untouched[blocks] = False
    assert not kc[untouched].any() and not vc[untouched].any()


def test_compaction_and_streaming_index_full_size(ops, oracle):
    B, M = 64, 8192 // BS + 1
    g = torch.Generator(device="cuda").manual_seed(21)
    tables = torch.stack([torch.randperm(100000, device="cuda", generator=g)[:M] for _ in range(B)]).to(torch.int32)
    # every sequence at a block boundary past the budget -> victim index 1 everywhere
    seq_lens = torch.full((B,), 8192 + BS, dtype=torch.int32, device="cuda")
    nblk = torch.full((B,), M, dtype=torch.int32, device="cuda")
    rmv, out, _ = ops.evict_select_compact(tables, seq_lens, nblk, "streamingLLM", 4096, BS, out_width=M - 1)
    assert (rmv == 1).all()
    assert torch.equal(out, torch.cat([tables[:, :1], tables[:, 2:]], dim=1))
    # host-chosen victims anywhere in the row: the row minus exactly that column
    ov = torch.randint(1, M - 1, (B,), device="cuda", generator=g, dtype=torch.int32)
    rmv, out, _ = ops.evict_select_compact(tables, seq_lens, nblk, "global", 4096, BS,
                                           victim_override=ov, out_width=M - 1)
    assert torch.equal(rmv, ov)
    cols = torch.arange(M - 1, device="cuda").unsqueeze(0)
    want = tables.gather(1, (cols + (cols >= ov.unsqueeze(1)).long()))
    assert torch.equal(out, want)
    # idempotence of "no eviction": lengths off the boundary leave the table as it is
    rmv, out, _ = ops.evict_select_compact(tables, seq_lens + 1, nblk, "streamingLLM", 4096, BS, out_width=M)
    assert (rmv == -1).all() and torch.equal(out, tables)
    # streamingLLM keep index for 64 prompts of 8192 at budget 4096
    L, budget = 8192, 4096
    kept = oracle.pruned_length(L, budget, BS)
    cu_q = torch.arange(0, B + 1, dtype=torch.int32, device="cuda") * L
    cu_k = torch.arange(0, B + 1, dtype=torch.int32, device="cuda") * kept
    idx = ops.prompt_keep_index_streaming(cu_q, cu_k, B * kept, budget, BS).view(B, kept).long()
    assert kept == budget
    assert (idx[:, 1:] > idx[:, :-1]).all()                         # sorted, no duplicates
    rel = idx - (torch.arange(B, device="cuda") * L).unsqueeze(1)
    assert torch.equal(rel[0], torch.tensor(oracle.streaming_keep_indices(L, budget, BS), device="cuda"))
    assert (rel == rel[0]).all()
    assert torch.equal(rel[0, :BS], torch.arange(BS, device="cuda"))
    assert torch.equal(rel[0, -BS:], torch.arange(L - BS, L, device="cuda"))


@pytest.mark.parametrize("method", ["cosine", "key_l2_div_value_l1"])
def test_readme_score_prune_full_size(ops, oracle, method):
    """The percentage type's prompt prune under a README-era token score at the headline's prefill
    size (8 prompts of 1024 tokens, 8 KV heads x 128): the kept rows of every prompt against the
    oracle's prune of that prompt (bit-exact: same scores, same order), and size-independent
    properties over the whole batch - every prompt keeps the same number of rows, its first block
    and its tail untouched."""
    from vllm_paged_compression_amd.pruner import KVCachePruner
    g = torch.Generator().manual_seed(31)
    n, L, H, D = 8, 1024, 8, 128
    k = (torch.rand(n * L, H, D, generator=g) * 2 - 1).to(torch.bfloat16)
    v = (torch.rand(n * L, H, D, generator=g) * 2 - 1).to(torch.bfloat16)
    p = KVCachePruner("percentage", method, BS, 0, 1, evict_freq=2, decode_evict_method="streamingLLM")
    pk, pv, kept = p.prune_prompts_batch(k.cuda(), v.cuda(), [L] * n)
    assert len(set(kept)) == 1 and sum(kept) == pk.shape[0] == pv.shape[0]
    K = kept[0]
    for i in (0, 3, 7):
        ek, ev = oracle.percentage_prompt_prune(k[i * L:(i + 1) * L], v[i * L:(i + 1) * L], method, BS, 2)
        assert ek.shape[0] == K
        assert torch.equal(pk[i * K:(i + 1) * K].cpu(), ek) and torch.equal(pv[i * K:(i + 1) * K].cpu(), ev)
    pk3, k3 = pk.view(n, K, H, D).cpu(), k.view(n, L, H, D)
    assert torch.equal(pk3[:, :BS], k3[:, :BS])                     # the first block stays
    end_first, end_mid, n_prune = oracle.percentage_prompt_slices(L, BS, 2, 1, 2)
    tail = L - end_mid
    if tail:
        assert torch.equal(pk3[:, K - tail:], k3[:, end_mid:])      # and so does the tail
    assert K == L - n_prune
