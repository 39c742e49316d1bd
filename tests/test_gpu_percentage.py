"""GPU parity of the "percentage" prune type (README-era flags): kept-row index, decode-time
block merge, table compaction and the backend driven end to end, against the oracle (pinned by
tests/test_percentage_mode.py) and the fixtures made from the reference's code for that mode."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN, from_bits

pytestmark = pytest.mark.gpu
BS = 16


@pytest.fixture(scope="module")
def ops():
    import vllm_paged_compression_amd.ops as ops
    import vllm_paged_compression_amd as pkg
    pkg.load_library()
    return ops


def _cfg(freq, nbm=2, decode="streamingLLM", disable_evict_prefill=True, prompt="streamingLLM"):
    from vllm_paged_compression_amd.config import PagedEvictConfig
    return PagedEvictConfig(disable_evict_prefill, "percentage", prompt, 0, -1, evict_freq=freq,
                            num_block_merge=nbm, decode_evict_method=decode)


def test_keep_index_percentage_golden(ops):
    tab = json.load(open(os.path.join(GOLDEN, "percentage_tables.json")))
    cases = tab["streaming_keep"]
    for bs in (8, 16):
        sel = [c for c in cases if c["bs"] == bs]
        cu_q = torch.tensor(np.cumsum([0] + [c["L"] for c in sel]), dtype=torch.int32).cuda()
        cu_k = torch.tensor(np.cumsum([0] + [len(c["kept"]) for c in sel]), dtype=torch.int32).cuda()
        total = int(cu_k[-1])
        idx = ops.prompt_keep_index_streaming(cu_q, cu_k, total, 0, bs).cpu().tolist()
        off_q = off_k = 0
        for c in sel:
            n = len(c["kept"])
            assert idx[off_k:off_k + n] == [off_q + t for t in c["kept"]], c
            off_q += c["L"]
            off_k += n


@pytest.mark.parametrize("freq", [2, 4])
def test_pruner_percentage_prompt(ops, oracle, freq):
    from vllm_paged_compression_amd.pruner import KVCachePruner
    g = torch.Generator().manual_seed(freq)
    p = KVCachePruner("percentage", "streamingLLM", BS, 0, 1, evict_freq=freq)
    lens = [1030, 40, 200, 48, 49]
    k = torch.rand(sum(lens), 2, 64, generator=g).to(torch.bfloat16)
    v = torch.rand(sum(lens), 2, 64, generator=g).to(torch.bfloat16)
    pk, pv, kept = p.prune_prompts_batch(k.cuda(), v.cuda(), lens)
    off, rows = 0, []
    for L in lens:
        rows += [off + t for t in oracle.percentage_keep_indices(L, BS, freq)]
        off += L
    assert kept == [len(oracle.percentage_keep_indices(L, BS, freq)) for L in lens]
    assert torch.equal(pk.cpu(), k[rows]) and torch.equal(pv.cpu(), v[rows])


@pytest.mark.parametrize("compat", [True, False])
def test_pruner_percentage_prompt_topk_golden(ops, oracle, compat):
    """Token-level prompt prune of the percentage type: the reference's outputs with its
    cross-head re-packing (compat), the oracle's per-head form otherwise."""
    from vllm_paged_compression_amd.pruner import KVCachePruner
    z = np.load(os.path.join(GOLDEN, "percentage_prompt_topk.npz"))
    n = len([k for k in z.files if k.endswith("_meta")])
    for c in range(n):
        L, freq, bs, nbm, H, D, mi = [int(v) for v in z[f"c{c}_meta"]]
        method = ("value_l2", "inverse_key_l2")[mi]
        k, v = from_bits(z[f"c{c}_k"], 0), from_bits(z[f"c{c}_v"], 0)
        p = KVCachePruner("percentage", method, bs, 0, 1, evict_freq=freq, num_block_merge=nbm,
                          decode_evict_method="streamingLLM", cross_head_compat=compat)
        pk, pv = p.prune_prompt(k.cuda(), v.cuda())
        if compat:
            ek, ev = from_bits(z[f"c{c}_pk"], 0), from_bits(z[f"c{c}_pv"], 0)
        else:
            ek, ev = oracle.percentage_prompt_prune(k, v, method, bs, freq, 1, nbm)
        assert torch.equal(pk.cpu(), ek) and torch.equal(pv.cpu(), ev), (c, method)
    # a batch of prompts, some too short to prune, 16-bit storage
    g = torch.Generator().manual_seed(3)
    lens = [300, 40, 130]
    k = torch.rand(sum(lens), 2, 64, generator=g).to(torch.bfloat16)
    v = torch.rand(sum(lens), 2, 64, generator=g).to(torch.bfloat16)
    p = KVCachePruner("percentage", "value_l2", 16, 0, 1, evict_freq=2, decode_evict_method="value_l2")
    pk, pv, kept = p.prune_prompts_batch(k.cuda(), v.cuda(), lens)
    off, ek, ev = 0, [], []
    for L in lens:
        a, b = oracle.percentage_prompt_prune(k[off:off + L], v[off:off + L], "value_l2", 16, 2)
        ek.append(a)
        ev.append(b)
        off += L
    assert kept == [t.shape[0] for t in ek]
    # 16-bit scores tie: oracle and kernel drop the lowest token first
    assert pk.shape[0] == sum(kept)
    gk, gv = pk.cpu(), pv.cpu()
    assert torch.equal(gk, torch.cat(ek)) and torch.equal(gv, torch.cat(ev))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("nbm,H,D,bs", [(2, 8, 128, 16), (3, 2, 64, 16), (4, 12, 64, 16), (2, 4, 80, 16),
                                         (2, 3, 32, 8), (8, 1, 128, 16)])
def test_merge_blocks_vs_oracle(ops, oracle, dtype, nbm, H, D, bs):
    g = torch.Generator().manual_seed(nbm * 100 + H + D)
    B, M = 5, nbm + 4
    nb = B * M + 3
    x = 16 // torch.tensor([], dtype=dtype).element_size()
    # coarse values so that 16-bit scores tie often (the tie rule is part of the contract)
    vals = 8 if dtype != torch.float32 else 1 << 20
    kc = (torch.randint(-vals, vals, (nb, H, D // x, bs, x), generator=g).float() / 8).to(dtype)
    vc = (torch.randint(-vals, vals, (nb, H, D, bs), generator=g).float() / 8).to(dtype)
    bt = torch.randperm(nb, generator=g)[:B * M].to(torch.int32).reshape(B, M)
    rows = [0, 2, 3]
    first = 1
    kd, vd = kc.cuda(), vc.cuda()
    ops.merge_blocks_value_l2(kd, vd, bt.cuda(), torch.tensor(rows, dtype=torch.int32).cuda(), first, nbm)
    for r in rows:
        oracle.merge_blocks_in_cache(kc, vc, bt[r, first:first + nbm].tolist())
    assert torch.equal(kd.cpu(), kc)
    assert torch.equal(vd.cpu(), vc)


def test_merge_blocks_golden_kept_sets(ops, oracle):
    """Inputs of the reference-made fixtures laid into a paged cache: the rows the kernel keeps
    are the rows the reference's prune_oldest_block keeps (per head)."""
    z = np.load(os.path.join(GOLDEN, "decode_merge.npz"))
    n = len([k for k in z.files if k.endswith("_meta")])
    for c in range(n):
        nbm, bs, H, D, prune_tokens = [int(v) for v in z[f"c{c}_meta"]]
        k, v = from_bits(z[f"c{c}_k"], 0), from_bits(z[f"c{c}_v"], 0)
        x = 4
        ids = list(range(2, 2 + nbm))
        kc = torch.zeros(nbm + 3, H, D // x, bs, x)
        vc = torch.zeros(nbm + 3, H, D, bs)
        slots = torch.tensor([b * bs + t for b in ids for t in range(bs)])
        oracle.reshape_and_cache(k, v, kc, vc, slots)
        bt = torch.tensor([[0] + ids + [1]], dtype=torch.int32)
        kd, vd = kc.cuda(), vc.cuda()
        ops.merge_blocks_value_l2(kd, vd, bt.cuda(), torch.zeros(1, dtype=torch.int32).cuda(), 1, nbm)
        ref_pk = from_bits(z[f"c{c}_pk"], 0)          # rows re-packed across heads (reference)
        keep = oracle.merge_keep_sets(v, bs)
        mask = torch.zeros(nbm * bs, H, dtype=torch.bool)
        for h in range(H):
            mask[keep[h], h] = True
        assert torch.equal(k[mask].view(-1, H, D), ref_pk)
        got_k = kd.cpu()[ids[0]].permute(2, 0, 1, 3).reshape(bs, H, D)
        got_v = vd.cpu()[ids[0]].permute(2, 0, 1)
        for h in range(H):
            assert torch.equal(got_k[:, h], k[keep[h], h]), (c, h)
            assert torch.equal(got_v[:, h], v[keep[h], h]), (c, h)


README_METHODS = ["cosine", "key_l1", "key_l2", "value_l1", "value_l2", "key_l1_div_value_l1",
                  "key_l1_div_value_l2", "key_l2_div_value_l1", "key_l2_div_value_l2", "value_l1_div_key_l1",
                  "value_l1_div_key_l2", "value_l2_div_key_l1", "value_l2_div_key_l2", "value_l1_plus_key_l1",
                  "value_l2_plus_key_l2", "streamingLLM", "inverse_key_l1", "inverse_key_l2"]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_readme_token_scores_vs_oracle_and_reference(ops, oracle, dtype):
    """pe_token_scores for every method name of the README-era get_score
    (kvcache_prunner_old.py:77-134): bit for bit the oracle's scores for 16-bit storage (same
    arithmetic, same order), and within two representable values of what the reference class
    itself returned (torch sums in another order; tests/golden/old_token_scores.npz)."""
    g = torch.Generator().manual_seed(11)
    T, H, D = 70, 3, 128
    k = (torch.rand(T, H, D, generator=g) * 2 - 1).to(dtype)
    v = (torch.rand(T, H, D, generator=g) * 4 - 2).to(dtype)
    k[5] = 0                                              # a zero row: norms 0, cosine 0
    for m in README_METHODS:
        got = ops.token_scores(k.cuda(), v.cuda(), m).cpu()
        want = oracle.readme_token_scores(k, v, m)
        if dtype == torch.float32:
            torch.testing.assert_close(got, want, rtol=2e-6, atol=1e-7)
        else:
            assert torch.equal(got, want), m
    with pytest.raises(ValueError):
        ops.token_scores(k.cuda(), v.cuda(), "key_l3")
    code = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dtype]
    z = np.load(os.path.join(GOLDEN, "old_token_scores.npz"))
    seen = 0
    for c in range(int(z["n_score_cases"])):
        if int(z[f"s{c}_meta"][3]) != code:
            continue
        kk, vv = from_bits(z[f"s{c}_k"], code), from_bits(z[f"s{c}_v"], code)
        for mi, m in enumerate(README_METHODS):
            want = from_bits(z[f"s{c}_m{mi}"], 0 if m == "streamingLLM" else code)
            got = ops.token_scores(kk.cuda(), vv.cuda(), m).cpu()
            if m == "streamingLLM":
                assert torch.equal(got.float(), want.float())
            elif code == 0:
                torch.testing.assert_close(got, want, rtol=2e-6, atol=1e-7)
            else:
                d = (got.view(torch.int16).int() - want.view(torch.int16).int()).abs()
                assert int(d.max()) <= 2, (m, int(d.max()))
            seen += 1
    assert seen == 2 * len(README_METHODS)


@pytest.mark.parametrize("method", ["cosine", "key_l1", "value_l1_div_key_l2", "value_l1_plus_key_l1",
                                    "inverse_key_l1", "key_l2_div_value_l2", "value_l1"])
def test_readme_scores_prompt_prune_and_merge(ops, oracle, method):
    """The percentage type's prompt prune and decode merge under README-era scores: the pruner's
    outputs against the oracle's (16-bit storage, scores that tie: lowest token first), the
    reference's own outputs where the fixtures hold them (f32, its cross-head re-packing), and the
    merge kernel against the oracle's merge on the paged layouts (negative scores included: the
    cosine)."""
    from vllm_paged_compression_amd.pruner import KVCachePruner
    g = torch.Generator().manual_seed(len(method))
    lens = [300, 40, 130, 97]
    k = (torch.rand(sum(lens), 2, 64, generator=g) * 2 - 1).to(torch.bfloat16)
    v = (torch.rand(sum(lens), 2, 64, generator=g) * 2 - 1).to(torch.bfloat16)
    p = KVCachePruner("percentage", method, 16, 0, 1, evict_freq=2, decode_evict_method=method)
    pk, pv, kept = p.prune_prompts_batch(k.cuda(), v.cuda(), lens)
    off, ek, ev = 0, [], []
    for L in lens:
        a, b = oracle.percentage_prompt_prune(k[off:off + L], v[off:off + L], method, 16, 2)
        ek.append(a)
        ev.append(b)
        off += L
    assert kept == [t.shape[0] for t in ek]
    assert torch.equal(pk.cpu(), torch.cat(ek)) and torch.equal(pv.cpu(), torch.cat(ev))
    assert torch.equal(p.get_score(k[:50].cuda(), v[:50].cuda(), method).cpu(),
                       oracle.readme_token_scores(k[:50], v[:50], method))
    # the reference's own prune_prompt outputs (compat form)
    z = np.load(os.path.join(GOLDEN, "old_token_scores.npz"))
    for c in range(int(z["n_prune_cases"])):
        L, freq, bs, nbm, H, D, mi = [int(x) for x in z[f"p{c}_meta"]]
        if README_METHODS[mi] != method:
            continue
        kk, vv = from_bits(z[f"p{c}_k"], 0), from_bits(z[f"p{c}_v"], 0)
        pc = KVCachePruner("percentage", method, bs, 0, 1, evict_freq=freq, num_block_merge=nbm,
                           decode_evict_method="streamingLLM", cross_head_compat=True)
        gk, gv = pc.prune_prompt(kk.cuda(), vv.cuda())
        assert torch.equal(gk.cpu(), from_bits(z[f"p{c}_pk"], 0)), (c, method)
        assert torch.equal(gv.cpu(), from_bits(z[f"p{c}_pv"], 0)), (c, method)
    # decode merge on the paged layouts
    for dtype, nbm, H, D, bs in ((torch.bfloat16, 2, 4, 128, 16), (torch.float16, 3, 2, 64, 16)):
        B, M = 4, nbm + 3
        nb = B * M + 2
        x = 8
        kc = ((torch.rand(nb, H, D // x, bs, x, generator=g) * 2 - 1)).to(dtype)
        vc = ((torch.rand(nb, H, D, bs, generator=g) * 2 - 1)).to(dtype)
        bt = torch.randperm(nb, generator=g)[:B * M].to(torch.int32).reshape(B, M)
        rows = [1, 3]
        kd, vd = kc.cuda(), vc.cuda()
        ops.merge_blocks_value_l2(kd, vd, bt.cuda(), torch.tensor(rows, dtype=torch.int32).cuda(), 1, nbm,
                                  formula=method)
        for r in rows:
            oracle.merge_blocks_in_cache(kc, vc, bt[r, 1:1 + nbm].tolist(), method)
        assert torch.equal(kd.cpu(), kc) and torch.equal(vd.cpu(), vc), (method, dtype)


def test_block_table_remove_range(ops, oracle):
    g = torch.Generator().manual_seed(1)
    bt = torch.randint(0, 1000, (7, 12), generator=g, dtype=torch.int32)
    for n_remove, start, width in [(1, [1, -1, 2, 11, 0, -1, 5], 12), (2, [1, 1, -1, 10, 3, -1, 0], 10),
                                   (3, [-1] * 7, 9), (2, [2] * 7, 10), (1, [1] * 7, 4)]:
        got = ops.block_table_remove_range(bt.cuda(), torch.tensor(start, dtype=torch.int32).cuda(),
                                           n_remove, width)
        assert torch.equal(got.cpu(), oracle.table_remove_range(bt, start, n_remove, width))
    with pytest.raises(RuntimeError):
        ops.block_table_remove_range(bt.cuda(), torch.zeros(7, dtype=torch.int32).cuda(), 0, 12)
    with pytest.raises(RuntimeError):
        ops.merge_blocks_value_l2(torch.zeros(4, 2, 16, 16, 8, dtype=torch.bfloat16).cuda(),
                                  torch.zeros(4, 2, 128, 16, dtype=torch.bfloat16).cuda(),
                                  bt.cuda(), torch.zeros(1, dtype=torch.int32).cuda(), 11, 2)


def _builder(cfg, inter):
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.config import CacheConfigView
    runner = SimpleNamespace(device=torch.device("cuda:0"),
                             cache_config=CacheConfigView(BS, cfg), pin_memory=False)
    ib = SimpleNamespace(runner=runner, sliding_window=None, block_size=BS,
                         chunked_prefill_enabled=False, inter_data_list=inter)
    return PagedEvictBackend.make_metadata_builder(ib)


def _mk(T, Hq, Hkv, D, g):
    sc = D ** -0.5
    q = ((torch.rand(T, Hq * D, generator=g) * 2 - 1) * sc * 3).to(torch.bfloat16)
    k = ((torch.rand(T, Hkv * D, generator=g) * 2 - 1) * sc * 3).to(torch.bfloat16)
    v = (torch.rand(T, Hkv * D, generator=g) * 2 - 1).to(torch.bfloat16)
    return q, k, v


@pytest.mark.parametrize("decode_method,nbm,freq,batched,prompt_method", [
    ("streamingLLM", 2, 2, False, "streamingLLM"), ("value_l2", 2, 2, False, "streamingLLM"),
    ("value_l2", 3, 1, False, "streamingLLM"), ("value_l2", 2, 2, True, "streamingLLM"),
    ("streamingLLM", 3, 2, True, "streamingLLM"), ("value_l2", 2, 2, False, "value_l2"),
    ("value_l2", 2, 4, True, "inverse_key_l2"), ("cosine", 2, 2, False, "key_l1"),
    ("key_l2_div_value_l1", 3, 2, True, "value_l1_plus_key_l1"), ("inverse_key_l1", 2, 2, False, "cosine")])
@pytest.mark.parametrize("disable_evict_prefill", [True, False])
def test_backend_percentage_trace(oracle, decode_method, nbm, freq, batched, prompt_method,
                                  disable_evict_prefill):
    """Builder -> forward over two layers -> get_rmv_block_idx -> scheduler commit, every step
    against a CPU walk: block tables and freed columns identical, KV caches byte-identical
    (merged blocks included), attention within the kernel-test tolerance."""
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler
    from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
    from vllm_paged_compression_amd.config import CacheConfigView

    g = torch.Generator().manual_seed(17 + nbm)
    Hq, Hkv, D, L_layers, nb = 8, 2, 128, 2, 96
    n_decode = 70
    cfg = _cfg(freq, nbm, decode_method, disable_evict_prefill, prompt_method)
    cache_cfg = CacheConfigView(BS, cfg)
    impls = [PagedEvictBackend.get_impl_cls()(
        Hq, D, D ** -0.5, Hkv, None, None, "auto", None, None, cache_config=cache_cfg,
        prefix=f"model.layers.{i}.self_attn.attn") for i in range(L_layers)]
    shape = PagedEvictBackend.get_kv_cache_shape(nb, BS, Hkv, D)
    kv_gpu = [torch.zeros(shape, dtype=torch.bfloat16, device="cuda") for _ in range(L_layers)]
    kc_cpu = [torch.zeros(nb, Hkv, D // 8, BS, 8, dtype=torch.bfloat16) for _ in range(L_layers)]
    vc_cpu = [torch.zeros(nb, Hkv, D, BS, dtype=torch.bfloat16) for _ in range(L_layers)]
    prompts = {"r0": 200, "r1": 130, "r2": 40}
    rids = list(prompts)
    if batched:
        sched = BatchEvictionScheduler(BS, nb, cfg, max_seqs=4, max_blocks_per_seq=32)
        table_of = lambda i: sched.get_block_table(rids[i])      # noqa: E731
        kvlen_of = lambda i: sched.get_seq_kv_len(rids[i])       # noqa: E731
    else:
        sched = EvictionScheduler(EvictBlockSpaceManager(BS, nb, cfg, watermark=0.0))
        table_of = lambda i: list(sched.block_manager.get_block_table(i))   # noqa: E731
        kvlen_of = lambda i: sched.block_manager.get_seq_kv_len(i)          # noqa: E731
    for i, (r, n) in enumerate(prompts.items()):
        sched.add_request(r, i, n)

    # ----------------------------------------------------------------- prefill
    lens = list(prompts.values())
    inter = [SimpleNamespace(request_id=r, is_prompt=True, block_tables={i: table_of(i)},
                             seq_kv_lens={i: kvlen_of(i)}, seq_ids=[i], input_tokens=[[0] * n],
                             orig_seq_lens=[n], seq_lens=[n], query_lens=[n], context_lens=[0])
             for i, (r, n) in enumerate(prompts.items())]
    meta = _builder(cfg, inter).build(lens, lens, -1, len(lens))
    slot_expect = []
    for it in inter:
        sid = it.seq_ids[0]
        slot_expect += oracle.prompt_slot_mapping(it.block_tables[sid], it.seq_lens[0], BS)
    assert meta.slot_mapping.cpu().tolist() == slot_expect
    T = sum(lens)
    rows, kept, off = [], [], 0
    for n in lens:
        keep = oracle.percentage_keep_indices(n, BS, freq, 1, nbm)
        rows += [off + t for t in keep]
        kept.append(len(keep))
        off += n
    assert meta.seq_kv_lens == kept
    for li in range(L_layers):
        q, k, v = _mk(T, Hq, Hkv, D, g)
        out = impls[li].forward(q.cuda(), k.cuda(), v.cuda(), kv_gpu[li], meta)
        q3, k3, v3 = q.view(T, Hq, D), k.view(T, Hkv, D), v.view(T, Hkv, D)
        pk, pv, off = [], [], 0
        for n in lens:
            a, b = oracle.percentage_prompt_prune(k3[off:off + n], v3[off:off + n], prompt_method, BS,
                                                  freq, 1, nbm)
            pk.append(a)
            pv.append(b)
            off += n
        pk, pv = torch.cat(pk), torch.cat(pv)
        if prompt_method == "streamingLLM":
            assert torch.equal(pk, k3[rows]) and torch.equal(pv, v3[rows])
        if disable_evict_prefill:
            ref = oracle.prefill_attention(q3.float(), k3.float(), v3.float(), lens, lens, D ** -0.5)
        else:
            ref = oracle.prefill_attention(q3.float(), pk.float(), pv.float(), lens, kept, D ** -0.5)
        torch.testing.assert_close(out.cpu().float().view(T, Hq, D), ref, atol=2e-2, rtol=1.6e-2)
        oracle.reshape_and_cache(pk, pv, kc_cpu[li], vc_cpu[li], torch.tensor(slot_expect))
        assert torch.equal(kv_gpu[li][0].view_as(kc_cpu[li]).cpu(), kc_cpu[li])
        assert torch.equal(kv_gpu[li][1].view_as(vc_cpu[li]).cpu(), vc_cpu[li])
    assert impls[0].get_rmv_block_idx() == {}

    # ----------------------------------------------------------------- decode
    seq_len = dict(prompts)
    n_prunes = 0
    for step in range(n_decode):
        if batched:
            batch = sched.schedule_decode(rids)
        else:
            sched.schedule_decode(rids)
        tbls = [table_of(i) for i in range(3)]
        kv_before = [kvlen_of(i) for i in range(3)]
        logical, kv_adj, first, merged = [], [], [], []
        for i, r in enumerate(rids):
            seq_len[r] += 1
            f, n_rm, mg = oracle.percentage_decode_blocks(seq_len[r], len(tbls[i]), BS, freq,
                                                          decode_method, 1, nbm)
            logical.append(seq_len[r])
            first.append(f)
            merged.append(mg)
            kv_adj.append(kv_before[i] - n_rm * BS)
        if batched:
            meta = _builder(cfg, []).build_decode_batch(batch, np.asarray(kv_adj))
        else:
            inter = [SimpleNamespace(request_id=r, is_prompt=False, block_tables={i: list(tbls[i])},
                                     seq_kv_lens={i: kv_adj[i]}, seq_ids=[i], input_tokens=[[0]],
                                     orig_seq_lens=[logical[i]], seq_lens=[logical[i]], query_lens=[1],
                                     context_lens=[kv_before[i] - 1]) for i, r in enumerate(rids)]
            meta = _builder(cfg, inter).build(logical, [1] * 3, -1, 3)
        slots = [oracle.decode_slot(t, kv_before[i] - 1, BS) for i, t in enumerate(tbls)]
        assert meta.slot_mapping.cpu().tolist() == slots
        M = max(len(t) for t in tbls)
        bt = torch.zeros((3, M), dtype=torch.int32)
        for i, t in enumerate(tbls):
            bt[i, :len(t)] = torch.tensor(t)
        after = [t[:f] + t[f + nbm - 1:] if f >= 0 else t for t, f in zip(tbls, first)]
        use_tables = oracle.table_remove_range(bt, first, nbm - 1, max(len(t) for t in after))
        for li in range(L_layers):
            q, k, v = _mk(3, Hq, Hkv, D, g)
            out = impls[li].forward(q.cuda(), k.cuda(), v.cuda(), kv_gpu[li], meta)
            oracle.reshape_and_cache(k.view(3, Hkv, D), v.view(3, Hkv, D), kc_cpu[li], vc_cpu[li],
                                     torch.tensor(slots))
            for i, mg in enumerate(merged):
                if mg is not None:
                    oracle.merge_blocks_in_cache(kc_cpu[li], vc_cpu[li], tbls[i][mg[0]:mg[1]], decode_method)
            assert torch.equal(kv_gpu[li][0].view_as(kc_cpu[li]).cpu(), kc_cpu[li]), (step, li)
            assert torch.equal(kv_gpu[li][1].view_as(vc_cpu[li]).cpu(), vc_cpu[li]), (step, li)
            ref = oracle.paged_attention(q.view(3, Hq, D), kc_cpu[li], vc_cpu[li], use_tables,
                                         torch.tensor(kv_adj, dtype=torch.int32), D ** -0.5)
            torch.testing.assert_close(out.cpu().float().view(3, Hq, D), ref.float(), atol=2e-3,
                                       rtol=1.6e-2, msg=lambda m: f"step {step} layer {li}: {m}")
        got = impls[0].get_rmv_block_idx()
        expect = {i: (r, f) for i, (r, f) in enumerate(zip(rids, first)) if f >= 0}
        assert got == expect, (step, got, expect)
        n_prunes += len(expect)
        sched.commit(got)
        for i in range(3):
            assert table_of(i) == after[i]
            assert kvlen_of(i) == kv_adj[i]
    assert n_prunes >= 3
