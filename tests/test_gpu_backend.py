"""End-to-end parity of the backend (metadata builder -> impl.forward over several layers ->
get_rmv_block_idx -> scheduler commit) against a CPU walk of the reference semantics built
from the oracle's pieces, on the same scheduler state: block tables and eviction indices must
be identical every step, attention outputs within the kernel-test tolerance."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

BS = 16


def _mk_inputs(T, Hq, Hkv, D, g):
    sc = D ** -0.5
    q = ((torch.rand(T, Hq * D, generator=g) * 2 - 1) * sc * 3).to(torch.bfloat16)
    k = ((torch.rand(T, Hkv * D, generator=g) * 2 - 1) * sc * 3).to(torch.bfloat16)
    v = ((torch.rand(T, Hkv * D, generator=g) * 2 - 1)).to(torch.bfloat16)
    return q, k, v


def _builder(cfg, inter, bs=None):
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.config import CacheConfigView
    bs = bs or BS
    runner = SimpleNamespace(device=torch.device("cuda:0"),
                             cache_config=CacheConfigView(bs, cfg), pin_memory=False)
    ib = SimpleNamespace(runner=runner, sliding_window=None, block_size=bs,
                         chunked_prefill_enabled=False, inter_data_list=inter)
    return PagedEvictBackend.make_metadata_builder(ib)


@pytest.mark.parametrize("method,formula", [("streamingLLM", "value_l2"), ("streamingLLM-1", "value_l2"),
                                            ("global", "value_l2"), ("local", "value_l2"),
                                            ("global", "key_l2"), ("global", "value_l2_plus_key_l2"),
                                            ("local", "inverse_key_l2")])
@pytest.mark.parametrize("disable_evict_prefill", [True, False])
def test_backend_generation_trace(oracle, method, formula, disable_evict_prefill):
    """A generation through the backend against a CPU walk of the reference's steps.  ``formula``
    is the score of global/local (KVCachePruner.sub_evict_method, kvcache_prunner.py:13,29-66)."""
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
    from vllm_paged_compression_amd.config import CacheConfigView, PagedEvictConfig
    from vllm_paged_compression_amd.page_evict_kv_util import adjust_seq_kv_len

    g = torch.Generator().manual_seed(31)
    Hq, Hkv, D, L_layers, nb = 8, 2, 128, 2, 96
    budget, n_decode = 128, 40
    cfg = PagedEvictConfig(disable_evict_prefill, "budget", method, budget, 3)
    cache_cfg = CacheConfigView(BS, cfg)
    impl_cls = PagedEvictBackend.get_impl_cls()
    impls = [impl_cls(Hq, D, D ** -0.5, Hkv, None, None, "auto", None, None, cache_config=cache_cfg,
                      prefix=f"model.layers.{i}.self_attn.attn") for i in range(L_layers)]
    for im in impls:
        im.kv_cache_pruner.sub_evict_method = formula
    shape = PagedEvictBackend.get_kv_cache_shape(nb, BS, Hkv, D)
    kv_gpu = [torch.zeros(shape, dtype=torch.bfloat16, device="cuda") for _ in range(L_layers)]
    x = 8
    kc_cpu = [torch.zeros(nb, Hkv, D // x, BS, x, dtype=torch.bfloat16) for _ in range(L_layers)]
    vc_cpu = [torch.zeros(nb, Hkv, D, BS, dtype=torch.bfloat16) for _ in range(L_layers)]

    sched = EvictionScheduler(EvictBlockSpaceManager(BS, nb, cfg, watermark=0.0))
    prompts = {"r0": 200, "r1": 130, "r2": 90}
    seq_of = {"r0": 0, "r1": 1, "r2": 2}
    for rid, n in prompts.items():
        assert sched.block_manager.can_allocate(n) == "OK"
        sched.add_request(rid, seq_of[rid], n)
    bm = sched.block_manager
    # expected allocation: pruned prompts (page_evict_kv_util.py:6-36)
    for rid, n in prompts.items():
        nblk, ntok = oracle.num_required_blocks_after_prune(n, budget, BS)
        assert len(bm.get_block_table(seq_of[rid])) == nblk and bm.get_seq_kv_len(seq_of[rid]) == ntok

    # ----------------------------------------------------------------- prefill step
    inter = []
    for rid, n in prompts.items():
        sid = seq_of[rid]
        inter.append(SimpleNamespace(
            request_id=rid, is_prompt=True, block_tables={sid: list(bm.get_block_table(sid))},
            seq_kv_lens={sid: bm.get_seq_kv_len(sid)}, seq_ids=[sid], input_tokens=[[0] * n],
            orig_seq_lens=[n], seq_lens=[n], query_lens=[n], context_lens=[0]))
    lens = list(prompts.values())
    meta = _builder(cfg, inter).build(lens, lens, -1, len(lens))
    slot_expect = []
    for it in inter:
        sid = it.seq_ids[0]
        slot_expect += oracle.prompt_slot_mapping(it.block_tables[sid], it.seq_lens[0], BS)
    assert meta.slot_mapping.cpu().tolist() == slot_expect
    assert meta.seq_kv_lens == [oracle.pruned_length(n, budget, BS) for n in lens]
    # host-side score model for global/local (per layer-0 only; decisions are per step)
    scores = {}
    T = sum(lens)
    for li in range(L_layers):
        q, k, v = _mk_inputs(T, Hq, Hkv, D, g)
        out = impls[li].forward(q.cuda(), k.cuda(), v.cuda(), kv_gpu[li], meta)
        k3, v3, q3 = k.view(T, Hkv, D), v.view(T, Hkv, D), q.view(T, Hq, D)
        pk, pv, off = [], [], 0
        for n in lens:
            a, b = oracle.prompt_prune(k3[off:off + n], v3[off:off + n], method, budget, BS,
                                       formula=formula)
            pk.append(a)
            pv.append(b)
            off += n
        kept = [t.shape[0] for t in pk]
        pk_all, pv_all = torch.cat(pk), torch.cat(pv)
        if disable_evict_prefill:
            ref = oracle.prefill_attention(q3.float(), k3.float(), v3.float(), lens, lens, D ** -0.5)
        else:
            ref = oracle.prefill_attention(q3.float(), pk_all.float(), pv_all.float(), lens, kept, D ** -0.5)
        torch.testing.assert_close(out.cpu().float().view(T, Hq, D), ref, atol=2e-2, rtol=1.6e-2)
        oracle.reshape_and_cache(pk_all, pv_all, kc_cpu[li], vc_cpu[li], torch.tensor(slot_expect))
        kc_g, vc_g = kv_gpu[li][0].view_as(kc_cpu[li]), kv_gpu[li][1].view_as(vc_cpu[li])
        assert torch.equal(kc_g.cpu(), kc_cpu[li]) and torch.equal(vc_g.cpu(), vc_cpu[li])
        if li == 0 and method in ("global", "local"):
            for i, t in enumerate(pv):
                scores[i] = [oracle.block_score_formula(pk[i][b * BS:(b + 1) * BS].contiguous(),
                                                        t[b * BS:(b + 1) * BS].contiguous(),
                                                        formula).float().item()
                             for b in range(t.shape[0] // BS)]
    assert impls[0].get_rmv_block_idx() == {}

    # ----------------------------------------------------------------- decode steps
    seq_len = dict(zip(prompts.keys(), lens))
    n_evictions = 0
    for step in range(n_decode):
        tables, kv_lens = sched.schedule_decode(list(prompts.keys()))
        inter, logical, kv_adj = [], [], []
        for rid in prompts:
            sid = seq_of[rid]
            seq_len[rid] += 1
            L = seq_len[rid]
            # the model runner's adjustment (model_runner.py:538-543), from the package helper
            kvl = adjust_seq_kv_len(cfg, L, kv_lens[sid], BS)
            assert kvl == kv_lens[sid] - (BS if oracle.decode_trigger(L, budget, BS) else 0)
            logical.append(L)
            kv_adj.append(kvl)
            inter.append(SimpleNamespace(
                request_id=rid, is_prompt=False, block_tables={sid: list(tables[sid])},
                seq_kv_lens={sid: kvl}, seq_ids=[sid], input_tokens=[[0]], orig_seq_lens=[L],
                seq_lens=[L], query_lens=[1], context_lens=[kv_lens[sid] - 1]))
        meta = _builder(cfg, inter).build(logical, [1] * 3, -1, 3)
        tbls = [list(tables[seq_of[r]]) for r in prompts]
        slots = [oracle.decode_slot(t, kv_lens[seq_of[r]] - 1, BS) for t, r in zip(tbls, prompts)]
        assert meta.slot_mapping.cpu().tolist() == slots
        M = max(len(t) for t in tbls)
        bt = torch.zeros((3, M), dtype=torch.int32)
        for i, t in enumerate(tbls):
            bt[i, :len(t)] = torch.tensor(t)
        # expected decision
        zero_slots = []
        if method in ("streamingLLM", "streamingLLM-1"):
            rmv, zero_slots = oracle.decode_streaming(logical, tbls, budget, BS, method)
        else:
            rmv = []
            for i, L in enumerate(logical):
                if oracle.decode_trigger(L, budget, BS):
                    blk = tbls[i][-1]        # the just-filled block, back in token-major rows
                    k_tm = kc_cpu[0][blk].permute(2, 0, 1, 3).reshape(BS, Hkv, D).contiguous()
                    v_tm = vc_cpu[0][blk].permute(2, 0, 1).contiguous()
                    new = oracle.block_score_formula(k_tm, v_tm, formula).float().item()
                    if formula == "value_l2":
                        assert new == oracle.block_score(vc_cpu[0][blk], BS, Hkv, D, 1)
                    rmv.append(oracle.decode_topk_victim(scores[i], new, method, 3, True))
                else:
                    rmv.append(-1)
        use_tables = bt
        if any(r >= 0 for r in rmv):
            use_tables = oracle.compact_block_tables(
                bt, rmv, out_width=oracle.surviving_width([len(t) for t in tbls], rmv))
        for li in range(L_layers):
            q, k, v = _mk_inputs(3, Hq, Hkv, D, g)
            out = impls[li].forward(q.cuda(), k.cuda(), v.cuda(), kv_gpu[li], meta)
            oracle.reshape_and_cache(k.view(3, Hkv, D), v.view(3, Hkv, D), kc_cpu[li], vc_cpu[li],
                                     torch.tensor(slots))
            if zero_slots:
                z = torch.zeros(len(zero_slots), Hkv, D, dtype=torch.bfloat16)
                oracle.reshape_and_cache(z, z, kc_cpu[li], vc_cpu[li], torch.tensor(zero_slots))
            assert torch.equal(kv_gpu[li][0].view_as(kc_cpu[li]).cpu(), kc_cpu[li]), (step, li)
            assert torch.equal(kv_gpu[li][1].view_as(vc_cpu[li]).cpu(), vc_cpu[li]), (step, li)
            ref = oracle.paged_attention(q.view(3, Hq, D), kc_cpu[li], vc_cpu[li], use_tables,
                                         torch.tensor(kv_adj, dtype=torch.int32), D ** -0.5)
            torch.testing.assert_close(out.cpu().float().view(3, Hq, D), ref.float(), atol=2e-3, rtol=1.6e-2,
                                       msg=lambda m: f"step {step} layer {li} rmv {rmv} kv {kv_adj} L {logical}: {m}")
        got = impls[0].get_rmv_block_idx()
        expect = {i: (rid, r) for i, (rid, r) in enumerate(zip(prompts, rmv)) if r >= 0}
        assert got == expect, (step, got, expect)
        n_evictions += len(expect)
        sched.commit(got)
        for i, rid in enumerate(prompts):
            if rmv[i] >= 0:
                del tbls[i][rmv[i]]
            assert bm.get_block_table(seq_of[rid]) == tbls[i]
    assert n_evictions >= 3
    # caches identical at the end (KV writes + zeroed slots are byte-exact)
    for li in range(L_layers):
        assert torch.equal(kv_gpu[li][0].view_as(kc_cpu[li]).cpu(), kc_cpu[li])
        assert torch.equal(kv_gpu[li][1].view_as(vc_cpu[li]).cpu(), vc_cpu[li])
    # steady state: every sequence holds between budget and budget + block tokens
    for rid in prompts:
        assert budget <= bm.get_seq_kv_len(seq_of[rid]) <= budget + BS or seq_len[rid] <= budget + BS
    # no kernel had to clamp a length to its table or drop a block score
    from vllm_paged_compression_amd import ops
    ops.check_device_diagnostics()


def test_backend_no_eviction_plain_path():
    """Eviction off: plain paged attention (prefill + decode) through the same backend."""
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.config import CacheConfigView
    from oracle import pe_oracle as oracle
    g = torch.Generator().manual_seed(3)
    Hq, Hkv, D, nb = 12, 12, 64, 32
    impl = PagedEvictBackend.get_impl_cls()(Hq, D, D ** -0.5, Hkv, None, None, "auto", None, None,
                                            cache_config=CacheConfigView(BS, None), prefix="")
    kv = torch.zeros(PagedEvictBackend.get_kv_cache_shape(nb, BS, Hkv, D), dtype=torch.float16, device="cuda")
    n = 70
    table = list(range(5, 10))
    inter = [SimpleNamespace(request_id="a", is_prompt=True, block_tables={0: table}, seq_kv_lens={0: n},
                             seq_ids=[0], input_tokens=[[0] * n], orig_seq_lens=[n], seq_lens=[n],
                             query_lens=[n], context_lens=[0])]
    meta = _builder(None, inter).build([n], [n], -1, 1)
    q = (torch.rand(n, Hq * D, generator=g) - 0.5).half()
    k = (torch.rand(n, Hkv * D, generator=g) - 0.5).half()
    v = (torch.rand(n, Hkv * D, generator=g) - 0.5).half()
    out = impl.forward(q.cuda(), k.cuda(), v.cuda(), kv, meta)
    ref = oracle.prefill_attention(q.view(n, Hq, D).float(), k.view(n, Hkv, D).float(),
                                   v.view(n, Hkv, D).float(), [n], [n], D ** -0.5)
    torch.testing.assert_close(out.cpu().float().view(n, Hq, D), ref, atol=3e-3, rtol=1e-2)
    # profiling run: empty cache tensor, output still produced
    out2 = impl.forward(q.cuda(), k.cuda(), v.cuda(), torch.empty(0, device="cuda", dtype=torch.float16), meta)
    torch.testing.assert_close(out2, out)


@pytest.mark.parametrize("per_layer", [False, True])
def test_backend_inverse_key_l2_decode_trace(oracle, per_layer):
    """Budget-type inverse_key_l2: the prompt is pruned per head by 1/(||K||+1e-8), and every
    decode step past the budget zeroes one random middle slot per sequence instead of removing
    a block (torch_cuda_sdpa.py:1435-1475,962-976).  The draws come from the global ``random``
    module seeded 1234 by the impl constructor (:796): the CPU walk replays them from
    ``random.Random(1234)``.  per_layer=True is HEAD's order (every layer draws its own slots,
    layer 0 first); the default draws once per step and zeroes the same slots in every layer.
    KV cache bytes must be identical after every layer of every step."""
    import random
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
    from vllm_paged_compression_amd.config import CacheConfigView, CompatFlags, PagedEvictConfig
    from vllm_paged_compression_amd.page_evict_kv_util import adjust_seq_kv_len

    g = torch.Generator().manual_seed(77)
    Hq, Hkv, D, L_layers, nb = 8, 2, 128, 3, 96
    budget, n_decode = 128, 36
    cfg = PagedEvictConfig(True, "budget", "inverse_key_l2", budget, -1)
    cache_cfg = CacheConfigView(BS, cfg)
    compat = CompatFlags(per_layer_decision=per_layer)
    impl_cls = PagedEvictBackend.get_impl_cls()
    impls = [impl_cls(Hq, D, D ** -0.5, Hkv, None, None, "auto", None, None, cache_config=cache_cfg,
                      prefix=f"model.layers.{i}.self_attn.attn", compat=compat) for i in range(L_layers)]
    rng = random.Random(1234)   # the state the last impl constructor left the global module in
    kv_gpu = [torch.zeros(PagedEvictBackend.get_kv_cache_shape(nb, BS, Hkv, D), dtype=torch.bfloat16,
                          device="cuda") for _ in range(L_layers)]
    kc_cpu = [torch.zeros(nb, Hkv, D // 8, BS, 8, dtype=torch.bfloat16) for _ in range(L_layers)]
    vc_cpu = [torch.zeros(nb, Hkv, D, BS, dtype=torch.bfloat16) for _ in range(L_layers)]
    sched = EvictionScheduler(EvictBlockSpaceManager(BS, nb, cfg, watermark=0.0))
    prompts = {"r0": 200, "r1": 120, "r2": 131}     # r1 starts under the budget and crosses it
    seq_of = {"r0": 0, "r1": 1, "r2": 2}
    for rid, n in prompts.items():
        sched.add_request(rid, seq_of[rid], n)
    bm = sched.block_manager
    inter = [SimpleNamespace(request_id=rid, is_prompt=True,
                             block_tables={seq_of[rid]: list(bm.get_block_table(seq_of[rid]))},
                             seq_kv_lens={seq_of[rid]: bm.get_seq_kv_len(seq_of[rid])},
                             seq_ids=[seq_of[rid]], input_tokens=[[0] * n], orig_seq_lens=[n],
                             seq_lens=[n], query_lens=[n], context_lens=[0])
             for rid, n in prompts.items()]
    lens = list(prompts.values())
    meta = _builder(cfg, inter).build(lens, lens, -1, len(lens))
    slot_expect = []
    for it in inter:
        sid = it.seq_ids[0]
        slot_expect += oracle.prompt_slot_mapping(it.block_tables[sid], it.seq_lens[0], BS)
    assert meta.slot_mapping.cpu().tolist() == slot_expect
    T = sum(lens)
    for li in range(L_layers):
        q, k, v = _mk_inputs(T, Hq, Hkv, D, g)
        impls[li].forward(q.cuda(), k.cuda(), v.cuda(), kv_gpu[li], meta)
        k3, v3 = k.view(T, Hkv, D), v.view(T, Hkv, D)
        pk, pv, off = [], [], 0
        for n in lens:
            a, b = oracle.prompt_prune(k3[off:off + n], v3[off:off + n], "inverse_key_l2", budget, BS)
            pk.append(a)
            pv.append(b)
            off += n
        oracle.reshape_and_cache(torch.cat(pk), torch.cat(pv), kc_cpu[li], vc_cpu[li],
                                 torch.tensor(slot_expect))
        assert torch.equal(kv_gpu[li][0].view_as(kc_cpu[li]).cpu(), kc_cpu[li])
        assert torch.equal(kv_gpu[li][1].view_as(vc_cpu[li]).cpu(), vc_cpu[li])

    seq_len = dict(zip(prompts.keys(), lens))
    n_zeroed = 0
    for step in range(n_decode):
        tables, kv_lens = sched.schedule_decode(list(prompts.keys()))
        inter, logical, kv_adj = [], [], []
        for rid in prompts:
            sid = seq_of[rid]
            seq_len[rid] += 1
            L = seq_len[rid]
            kvl = adjust_seq_kv_len(cfg, L, kv_lens[sid], BS)
            assert kvl == kv_lens[sid]      # this method never shortens the attention view
            logical.append(L)
            kv_adj.append(kvl)
            inter.append(SimpleNamespace(
                request_id=rid, is_prompt=False, block_tables={sid: list(tables[sid])},
                seq_kv_lens={sid: kvl}, seq_ids=[sid], input_tokens=[[0]], orig_seq_lens=[L],
                seq_lens=[L], query_lens=[1], context_lens=[kv_lens[sid] - 1]))
        meta = _builder(cfg, inter).build(logical, [1] * 3, -1, 3)
        tbls = [list(tables[seq_of[r]]) for r in prompts]
        slots = [oracle.decode_slot(t, kv_lens[seq_of[r]] - 1, BS) for t, r in zip(tbls, prompts)]
        assert meta.slot_mapping.cpu().tolist() == slots
        M = max(len(t) for t in tbls)
        bt = torch.zeros((3, M), dtype=torch.int32)
        for i, t in enumerate(tbls):
            bt[i, :len(t)] = torch.tensor(t)
        zero_step = None
        for li in range(L_layers):
            if per_layer or zero_step is None:
                zero_step = oracle.decode_inverse_key_l2(logical, tbls, budget, BS, rng)
            q, k, v = _mk_inputs(3, Hq, Hkv, D, g)
            out = impls[li].forward(q.cuda(), k.cuda(), v.cuda(), kv_gpu[li], meta)
            oracle.reshape_and_cache(k.view(3, Hkv, D), v.view(3, Hkv, D), kc_cpu[li], vc_cpu[li],
                                     torch.tensor(slots))
            if zero_step:
                z = torch.zeros(len(zero_step), Hkv, D, dtype=torch.bfloat16)
                oracle.reshape_and_cache(z, z, kc_cpu[li], vc_cpu[li], torch.tensor(zero_step))
                n_zeroed += len(zero_step)
            assert torch.equal(kv_gpu[li][0].view_as(kc_cpu[li]).cpu(), kc_cpu[li]), (step, li)
            assert torch.equal(kv_gpu[li][1].view_as(vc_cpu[li]).cpu(), vc_cpu[li]), (step, li)
            ref = oracle.paged_attention(q.view(3, Hq, D), kc_cpu[li], vc_cpu[li], bt,
                                         torch.tensor(kv_adj, dtype=torch.int32), D ** -0.5)
            torch.testing.assert_close(out.cpu().float().view(3, Hq, D), ref.float(), atol=2e-3,
                                       rtol=1.6e-2, msg=lambda m: f"step {step} layer {li}: {m}")
        assert impls[0].get_rmv_block_idx() == {}      # nothing to commit: no block leaves
        sched.commit(impls[0].get_rmv_block_idx())
        for i, rid in enumerate(prompts):
            assert bm.get_block_table(seq_of[rid]) == tbls[i]
    assert n_zeroed >= 2 * n_decode


@pytest.mark.parametrize("head_compat", [False, True])
def test_streaming_decode_table_on_eviction_steps(oracle, head_compat):
    """On a streamingLLM eviction step the default attends the compacted table; HEAD builds that
    table and then ignores it (the flag it sets, is_pruned_decode at torch_cuda_sdpa.py:903-911,
    is not the one the attention call tests, is_prune_decode at :858,1046), reading the ORIGINAL
    table with the kv length shortened by one block.  CompatFlags.streaming_decode_original_table
    reproduces that; the committed victim is block 1 either way."""
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
    from vllm_paged_compression_amd.config import CacheConfigView, CompatFlags, PagedEvictConfig
    from vllm_paged_compression_amd.page_evict_kv_util import adjust_seq_kv_len
    g = torch.Generator().manual_seed(5)
    Hq, Hkv, D, nb, budget, n = 8, 2, 128, 40, 128, 150
    cfg = PagedEvictConfig(True, "budget", "streamingLLM", budget, -1)
    impls = [PagedEvictBackend.get_impl_cls()(
        Hq, D, D ** -0.5, Hkv, None, None, "auto", None, None, cache_config=CacheConfigView(BS, cfg),
        prefix=f"model.layers.{i}.self_attn.attn",
        compat=CompatFlags(streaming_decode_original_table=head_compat)) for i in range(2)]
    kv_gpu = [torch.zeros(PagedEvictBackend.get_kv_cache_shape(nb, BS, Hkv, D), dtype=torch.bfloat16,
                          device="cuda") for _ in impls]
    kc = [torch.zeros(nb, Hkv, D // 8, BS, 8, dtype=torch.bfloat16) for _ in impls]
    vc = [torch.zeros(nb, Hkv, D, BS, dtype=torch.bfloat16) for _ in impls]
    sched = EvictionScheduler(EvictBlockSpaceManager(BS, nb, cfg, watermark=0.0))
    sched.add_request("r", 0, n)
    bm = sched.block_manager
    table = list(bm.get_block_table(0))
    meta = _builder(cfg, [SimpleNamespace(
        request_id="r", is_prompt=True, block_tables={0: table}, seq_kv_lens={0: bm.get_seq_kv_len(0)},
        seq_ids=[0], input_tokens=[[0] * n], orig_seq_lens=[n], seq_lens=[n], query_lens=[n],
        context_lens=[0])]).build([n], [n], -1, 1)
    for li, impl in enumerate(impls):
        q, k, v = _mk_inputs(n, Hq, Hkv, D, g)
        impl.forward(q.cuda(), k.cuda(), v.cuda(), kv_gpu[li], meta)
        pk, pv = oracle.prompt_prune(k.view(n, Hkv, D), v.view(n, Hkv, D), "streamingLLM", budget, BS)
        oracle.reshape_and_cache(pk, pv, kc[li], vc[li], torch.tensor(oracle.prompt_slot_mapping(table, n, BS)))
    L, evictions = n, 0
    for _ in range(30):
        tables, kvl = sched.schedule_decode(["r"])
        L += 1
        table = list(tables[0])
        kv_len = adjust_seq_kv_len(cfg, L, kvl[0], BS)
        meta = _builder(cfg, [SimpleNamespace(
            request_id="r", is_prompt=False, block_tables={0: table}, seq_kv_lens={0: kv_len}, seq_ids=[0],
            input_tokens=[[0]], orig_seq_lens=[L], seq_lens=[L], query_lens=[1],
            context_lens=[kvl[0] - 1])]).build([L], [1], -1, 1)
        rmv, _ = oracle.decode_streaming([L], [table], budget, BS, "streamingLLM")
        bt = torch.tensor([table], dtype=torch.int32)
        use = bt
        if rmv[0] >= 0 and not head_compat:
            use = oracle.compact_block_tables(bt, rmv, out_width=len(table) - 1)
        for li, impl in enumerate(impls):
            q, k, v = _mk_inputs(1, Hq, Hkv, D, g)
            out = impl.forward(q.cuda(), k.cuda(), v.cuda(), kv_gpu[li], meta)
            oracle.reshape_and_cache(k.view(1, Hkv, D), v.view(1, Hkv, D), kc[li], vc[li],
                                     torch.tensor([oracle.decode_slot(table, kvl[0] - 1, BS)]))
            assert torch.equal(kv_gpu[li][0].view_as(kc[li]).cpu(), kc[li])
            assert torch.equal(kv_gpu[li][1].view_as(vc[li]).cpu(), vc[li])
            ref = oracle.paged_attention(q.view(1, Hq, D), kc[li], vc[li], use,
                                         torch.tensor([kv_len], dtype=torch.int32), D ** -0.5)
            torch.testing.assert_close(out.cpu().float().view(1, Hq, D), ref.float(), atol=2e-3, rtol=1.6e-2)
        got = impls[0].get_rmv_block_idx()
        assert got == ({0: ("r", 1)} if rmv[0] >= 0 else {})
        evictions += len(got)
        sched.commit(got)
    assert evictions == 2


@pytest.mark.parametrize("method,bs,budget,evict_prefill", [
    ("global", 16, 128, False), ("streamingLLM-1", 16, 128, False), ("streamingLLM", 16, 128, False),
    ("streamingLLM", 8, 128, False), ("streamingLLM", 32, 128, False), ("global", 8, 128, False),
    ("global", 32, 128, False), ("streamingLLM", 16, 128, True), ("global", 16, 256, False),
    (None, 16, 0, False)])
def test_backend_forward_with_rope(oracle, method, bs, budget, evict_prefill):
    """forward(..., rope=(positions, cos_sin_cache, is_neox)) on unrotated query/key rows equals
    rotary embedding (the separate op) followed by forward(): outputs and KV caches bit for bit,
    over a prefill step and decode steps that include eviction steps.  'global' decodes on the
    one-launch path (rotation inside the attention kernel: query/key must stay unrotated),
    'streamingLLM-1' zeroes a slot per eviction step, which takes the separate-launch path
    (query/key rotated in place).  Prompts: the streamingLLM prune with attention over the full
    prompt, prompts within the budget (256) and no eviction at all rotate inside the KV-write
    launch; a token-level prune and a prune in front of attention rotate first.  Block sizes 8,
    16 and 32."""
    from test_oracle_golden import rope_cache
    from vllm_paged_compression_amd import ops
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler
    from vllm_paged_compression_amd.config import CacheConfigView, PagedEvictConfig

    g = torch.Generator().manual_seed(77)
    Hq, Hkv, D, L_layers, nb = 8, 2, 128, 2, 1536 // bs
    n_decode, max_pos = 40, 1024
    cfg = None if method is None else PagedEvictConfig(not evict_prefill, "budget", method, budget, 3)
    cache_cfg = CacheConfigView(bs, cfg)
    impl_cls = PagedEvictBackend.get_impl_cls()
    cs = rope_cache(max_pos, D, torch.bfloat16).cuda()
    prompts = {"r0": 200, "r1": 130, "r2": 90}
    lens = list(prompts.values())
    B = len(lens)

    class Side:
        def __init__(self):
            self.impls = [impl_cls(Hq, D, D ** -0.5, Hkv, None, None, "auto", None, None,
                                   cache_config=cache_cfg, prefix=f"model.layers.{i}.self_attn.attn",
                                   max_num_seqs=8, max_blocks_per_seq=64) for i in range(L_layers)]
            shape = PagedEvictBackend.get_kv_cache_shape(nb, bs, Hkv, D)
            self.kv = [torch.zeros(shape, dtype=torch.bfloat16, device="cuda") for _ in range(L_layers)]
            self.sched = BatchEvictionScheduler(bs, nb, cfg, max_seqs=8, max_blocks_per_seq=64)
            for i, (rid, n) in enumerate(prompts.items()):
                self.sched.add_request(rid, i, n)

        def prefill_meta(self):
            inter = [SimpleNamespace(request_id=rid, is_prompt=True,
                                     block_tables={i: self.sched.get_block_table(rid)},
                                     seq_kv_lens={i: self.sched.get_seq_kv_len(rid)}, seq_ids=[i],
                                     input_tokens=[[0] * n], orig_seq_lens=[n], seq_lens=[n],
                                     query_lens=[n], context_lens=[0])
                     for i, (rid, n) in enumerate(prompts.items())]
            return _builder(cfg, inter, bs).build(lens, lens, -1, B)

    fused, split = Side(), Side()
    T = sum(lens)
    pos_p = torch.cat([torch.arange(n) for n in lens]).cuda()
    meta_f, meta_s = fused.prefill_meta(), split.prefill_meta()
    for li in range(L_layers):
        q, k, v = (t.cuda() for t in _mk_inputs(T, Hq, Hkv, D, g))
        q2, k2 = q.clone(), k.clone()
        out_f = fused.impls[li].forward(q, k, v, fused.kv[li], meta_f, rope=(pos_p, cs, True))
        ops.rotary_embedding(pos_p, q2, k2, D, cs, True)
        out_s = split.impls[li].forward(q2, k2, v, split.kv[li], meta_s)
        assert torch.equal(q, q2) and torch.equal(k, k2)      # prompt rows are rotated in place
        assert torch.equal(out_f, out_s) and torch.equal(fused.kv[li], split.kv[li])
    rids = list(prompts)
    L = dict(zip(rids, lens))
    evictions = 0
    for step in range(n_decode):
        bf, bsp = fused.sched.schedule_decode(rids), split.sched.schedule_decode(rids)
        meta_f = _builder(cfg, [], bs).build_decode_batch(bf)
        meta_s = _builder(cfg, [], bs).build_decode_batch(bsp)
        pos = torch.tensor([L[r] for r in rids], dtype=torch.int64).cuda()   # position of the new token
        for r in rids:
            L[r] += 1
        for li in range(L_layers):
            q, k, v = (t.cuda() for t in _mk_inputs(B, Hq, Hkv, D, g))
            q0, k0 = q.clone(), k.clone()
            q2, k2 = q.clone(), k.clone()
            out_f = fused.impls[li].forward(q, k, v, fused.kv[li], meta_f, rope=(pos, cs, True))
            ops.rotary_embedding(pos, q2, k2, D, cs, True)
            out_s = split.impls[li].forward(q2, k2, v, split.kv[li], meta_s)
            assert torch.equal(out_f, out_s), (step, li)
            assert torch.equal(fused.kv[li], split.kv[li]), (step, li)
            if method in ("global", "streamingLLM", None):
                assert torch.equal(q, q0) and torch.equal(k, k0), "one-launch path rotates in registers"
        got_f, got_s = fused.impls[0].get_rmv_block_idx(), split.impls[0].get_rmv_block_idx()
        assert got_f == got_s
        evictions += len(got_f)
        fused.sched.commit(got_f)
        split.sched.commit(got_s)
    assert evictions >= 2 or method is None or budget == 256
    ops.check_device_diagnostics()


def test_prefill_with_rope_is_one_launch_fewer(monkeypatch):
    """Launch count of a prefill layer handed the rotary embedding: rotation + KV write are ONE
    launch (pe_rope_reshape_and_cache) when the kept rows are known from the lengths, i.e. the
    layer issues two launches (that one and attention) where rotary_embedding, attention and the
    KV write used to be three."""
    from test_oracle_golden import rope_cache
    from vllm_paged_compression_amd import _lib, ops
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler
    from vllm_paged_compression_amd.config import CacheConfigView, PagedEvictConfig
    lib = _lib.load()
    calls = []
    for name in ("pe_rope_reshape_and_cache", "pe_reshape_and_cache", "pe_prefill_causal",
                 "pe_prompt_prune_gather"):
        fn = getattr(lib, name)

        def wrapped(*a, _fn=fn, _name=name):
            calls.append(_name)
            return _fn(*a)
        monkeypatch.setattr(lib, name, wrapped)
    Hq, Hkv, D, nb = 8, 2, 128, 64
    cs = rope_cache(512, D, torch.bfloat16).cuda()
    for method, want in (("streamingLLM", ["pe_rope_reshape_and_cache", "pe_prefill_causal"]),
                         ("global", ["pe_rope_reshape_and_cache", "pe_prefill_causal", "pe_reshape_and_cache"])):
        cfg = PagedEvictConfig(True, "budget", method, 128, 3)
        ccfg = CacheConfigView(BS, cfg)
        impl = PagedEvictBackend.get_impl_cls()(Hq, D, D ** -0.5, Hkv, None, None, "auto", None, None,
                                                cache_config=ccfg, prefix="model.layers.0.self_attn.attn",
                                                max_num_seqs=4, max_blocks_per_seq=32)
        kv = torch.zeros(PagedEvictBackend.get_kv_cache_shape(nb, BS, Hkv, D), dtype=torch.bfloat16, device="cuda")
        sched = BatchEvictionScheduler(BS, nb, cfg, max_seqs=4, max_blocks_per_seq=32)
        n = 300
        sched.add_request("r", 0, n)
        inter = [SimpleNamespace(request_id="r", is_prompt=True, block_tables={0: sched.get_block_table("r")},
                                 seq_kv_lens={0: sched.get_seq_kv_len("r")}, seq_ids=[0], input_tokens=[[0] * n],
                                 orig_seq_lens=[n], seq_lens=[n], query_lens=[n], context_lens=[0])]
        meta = _builder(cfg, inter).build([n], [n], -1, 1)
        q, k, v = (t.cuda() for t in _mk_inputs(n, Hq, Hkv, D, torch.Generator().manual_seed(1)))
        del calls[:]
        impl.forward(q, k, v, kv, meta, rope=(torch.arange(n).cuda(), cs, True))
        if method == "streamingLLM":
            assert calls == want, calls
        else:       # token-level prune: rotation first (pe_rope_reshape_and_cache without a cache)
            assert calls[0] == "pe_rope_reshape_and_cache" and "pe_reshape_and_cache" in calls
    torch.cuda.synchronize()
