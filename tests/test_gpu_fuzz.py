"""Randomised continuous-batching runs of the backend against a CPU walk built from the oracle:
requests of random prompt length arrive at random steps, generate a random number of tokens
and leave, so the batch composition, the table widths and the score-table rows keep changing.
Every step is checked: slot mapping, KV cache bytes (exact), eviction victims (exact), the
scheduler's tables after the commit (exact), attention outputs (kernel-test tolerance).  Run for
each eviction method with the per-request and with the vectorised scheduler / metadata build."""
import random
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

BS = 16
HQ, HKV, D, LAYERS = 8, 2, 64, 2
BUDGET, TOPK = 128, 3
FREQ = 2           # percentage prune type ("pct:<decode method>" cases; "pct3:" merges 3 blocks)


def _rand(T, H, g, scale):
    return ((torch.rand(T, H * D, generator=g) * 2 - 1) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("vectorised", [False, True, "device"])
@pytest.mark.parametrize("method,seed,evict_before_prefill", [
    ("streamingLLM", 0, False), ("streamingLLM-1", 1, False), ("global", 2, False), ("global", 3, False),
    ("local", 4, False), ("streamingLLM", 5, True), ("global", 6, True),
    ("pct:streamingLLM", 7, False), ("pct:value_l2", 8, False), ("pct:value_l2", 9, True),
    ("pct3:value_l2", 10, False), ("pct3:streamingLLM", 11, False),
    # merges ranked by other token scores of the README era (kvcache_prunner_old.py:77-134)
    ("pct:cosine", 17, False), ("pct3:key_l1_div_value_l2", 18, False),
    # the other two page sizes of the reference's scripts/diff_page_sizes.sh
    ("bs8:global", 12, False), ("bs32:global", 13, False), ("bs8:streamingLLM-1", 14, False),
    ("bs32:streamingLLM", 15, True), ("bs32:local", 16, False)])
def test_continuous_batching_fuzz(oracle, method, seed, evict_before_prefill, vectorised):
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler
    from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
    from vllm_paged_compression_amd.config import CacheConfigView, PagedEvictConfig

    BS = 16
    if method.startswith("bs"):
        BS, method = int(method[2:method.index(":")]), method.split(":", 1)[1]
    rng = random.Random(100 + seed)
    g = torch.Generator().manual_seed(200 + seed)
    nb = 160 * 16 // BS
    pct = method.startswith("pct")
    NBM = 3 if method.startswith("pct3:") else 2
    if pct:
        method = method.split(":", 1)[1]
        cfg = PagedEvictConfig(not evict_before_prefill, "percentage", "streamingLLM", 0, -1,
                               evict_freq=FREQ, num_block_merge=NBM, decode_evict_method=method)
    else:
        cfg = PagedEvictConfig(not evict_before_prefill, "budget", method, BUDGET, TOPK)
    ccfg = CacheConfigView(BS, cfg)
    impls = [PagedEvictBackend.get_impl_cls()(HQ, D, D ** -0.5, HKV, None, None, "auto", None, None,
                                              cache_config=ccfg, prefix=f"model.layers.{i}.self_attn.attn",
                                              max_num_seqs=8) for i in range(LAYERS)]
    shape = PagedEvictBackend.get_kv_cache_shape(nb, BS, HKV, D)
    kv_gpu = [torch.zeros(shape, dtype=torch.bfloat16, device="cuda") for _ in range(LAYERS)]
    kc_cpu = [torch.zeros(nb, HKV, D // 8, BS, 8, dtype=torch.bfloat16) for _ in range(LAYERS)]
    vc_cpu = [torch.zeros(nb, HKV, D, BS, dtype=torch.bfloat16) for _ in range(LAYERS)]
    ref_sched = EvictionScheduler(EvictBlockSpaceManager(BS, nb, cfg, watermark=0.0))
    bm = ref_sched.block_manager
    # "device": the vectorised scheduler with the running batch's tables resident on the GPU
    vec_sched = BatchEvictionScheduler(BS, nb, cfg, max_seqs=8, max_blocks_per_seq=40 * max(1, 16 // BS),
                                       device="cuda:0" if vectorised == "device" else None) \
        if vectorised else None

    def builder(inter):
        runner = SimpleNamespace(device=torch.device("cuda:0"), cache_config=ccfg, pin_memory=False)
        ib = SimpleNamespace(runner=runner, sliding_window=None, block_size=BS,
                             chunked_prefill_enabled=False, inter_data_list=inter)
        return PagedEvictBackend.make_metadata_builder(ib)

    # arrival plan: (step, request id, prompt length, tokens to generate)
    lens_pool = [1, 5, 16, 17, 100, 128, 129, 140, 200, 260, 300, 131, 64, 223]
    plan = []
    step_t = 0
    for i in range(9):
        step_t += rng.choice([0, 0, 3, 10, 25])
        plan.append((step_t, f"req{i}", rng.choice(lens_pool), rng.randint(8, 70)))
    seq_of = {rid: i for i, (_, rid, _, _) in enumerate(plan)}
    waiting, running = [], []
    logical, remaining, scores = {}, {}, {}
    n_evictions = n_prefills = 0
    step = 0
    while plan or waiting or running:
        while plan and plan[0][0] <= step:
            waiting.append(plan.pop(0))
        if waiting and len(running) < 6:
            # ------------------------------------------------------------ prefill step
            grp = []
            while waiting and len(grp) < 3 and len(running) + len(grp) < 6:
                grp.append(waiting.pop(0))
            inter, lens = [], []
            for _, rid, n, n_out in grp:
                sid = seq_of[rid]
                ref_sched.add_request(rid, sid, n)
                if vec_sched is not None:
                    vec_sched.add_request(rid, sid, n)
                    assert vec_sched.get_block_table(rid) == bm.get_block_table(sid)
                logical[rid], remaining[rid] = n, n_out
                inter.append(SimpleNamespace(
                    request_id=rid, is_prompt=True, block_tables={sid: list(bm.get_block_table(sid))},
                    seq_kv_lens={sid: bm.get_seq_kv_len(sid)}, seq_ids=[sid], input_tokens=[[0] * n],
                    orig_seq_lens=[n], seq_lens=[n], query_lens=[n], context_lens=[0]))
                lens.append(n)
            meta = builder(inter).build(lens, lens, -1, len(lens))
            slot_expect = []
            for it in inter:
                sid = it.seq_ids[0]
                slot_expect += oracle.prompt_slot_mapping(it.block_tables[sid], it.seq_lens[0], BS)
            assert meta.slot_mapping.cpu().tolist() == slot_expect
            T = sum(lens)
            for li in range(LAYERS):
                q, k, v = _rand(T, HQ, g, 0.4), _rand(T, HKV, g, 0.4), _rand(T, HKV, g, 1.0)
                out = impls[li].forward(q.cuda(), k.cuda(), v.cuda(), kv_gpu[li], meta)
                q3, k3, v3 = q.view(T, HQ, D), k.view(T, HKV, D), v.view(T, HKV, D)
                pk, pv, off = [], [], 0
                for n in lens:
                    if pct:
                        keep = oracle.percentage_keep_indices(n, BS, FREQ, 1, NBM)
                        a, b = k3[off:off + n][keep], v3[off:off + n][keep]
                    else:
                        a, b = oracle.prompt_prune(k3[off:off + n], v3[off:off + n], method, BUDGET, BS)
                    pk.append(a)
                    pv.append(b)
                    off += n
                if evict_before_prefill:   # attention over the pruned K/V, top-left causal mask
                    ref = oracle.prefill_attention(q3.float(), torch.cat(pk).float(), torch.cat(pv).float(),
                                                   lens, [t.shape[0] for t in pk], D ** -0.5)
                else:
                    ref = oracle.prefill_attention(q3.float(), k3.float(), v3.float(), lens, lens, D ** -0.5)
                torch.testing.assert_close(out.cpu().float().view(T, HQ, D), ref, atol=2e-2, rtol=1.6e-2)
                oracle.reshape_and_cache(torch.cat(pk), torch.cat(pv), kc_cpu[li], vc_cpu[li],
                                         torch.tensor(slot_expect))
                assert torch.equal(kv_gpu[li][0].view_as(kc_cpu[li]).cpu(), kc_cpu[li]), (step, li)
                assert torch.equal(kv_gpu[li][1].view_as(vc_cpu[li]).cpu(), vc_cpu[li]), (step, li)
                if li == 0 and not pct and method in ("global", "local"):
                    for (_, rid, _, _), t in zip(grp, pv):
                        scores[rid] = [oracle.block_score(t[b * BS:(b + 1) * BS].contiguous(), BS, HKV, D, 0)
                                       for b in range(t.shape[0] // BS)]
            assert impls[0].get_rmv_block_idx() == {}
            running += [rid for _, rid, _, _ in grp]
            n_prefills += 1
        elif running:
            # ------------------------------------------------------------ decode step
            tables, kv_lens = ref_sched.schedule_decode(running)
            batch = vec_sched.schedule_decode(running) if vec_sched is not None else None
            B = len(running)
            L, kv_adj, tbls, merged = [], [], [], []
            n_rm = NBM - 1 if pct else 1
            for rid in running:
                sid = seq_of[rid]
                logical[rid] += 1
                L.append(logical[rid])
                tbls.append(list(tables[sid]))
                if pct:
                    first, _, mg = oracle.percentage_decode_blocks(logical[rid], len(tbls[-1]), BS, FREQ,
                                                                   method, 1, NBM)
                    merged.append(mg)
                    kv_adj.append(kv_lens[sid] - (n_rm * BS if first >= 0 else 0))
                else:
                    kv_adj.append(kv_lens[sid] - (BS if oracle.decode_trigger(logical[rid], BUDGET, BS) else 0))
            # what attention must see this step comes from the package (the model runner's
            # adjustment, model_runner.py:538-543); the oracle's figures above are the check
            from vllm_paged_compression_amd.page_evict_kv_util import adjust_seq_kv_len
            assert kv_adj == [adjust_seq_kv_len(cfg, logical[rid], kv_lens[seq_of[rid]], BS)
                              for rid in running]
            if batch is not None:
                meta = builder([]).build_decode_batch(batch)
                assert meta.seq_kv_lens == kv_adj
            else:
                inter = [SimpleNamespace(request_id=rid, is_prompt=False,
                                         block_tables={seq_of[rid]: list(tables[seq_of[rid]])},
                                         seq_kv_lens={seq_of[rid]: kv_adj[i]}, seq_ids=[seq_of[rid]],
                                         input_tokens=[[0]], orig_seq_lens=[L[i]], seq_lens=[L[i]],
                                         query_lens=[1], context_lens=[kv_lens[seq_of[rid]] - 1])
                         for i, rid in enumerate(running)]
                meta = builder(inter).build(L, [1] * B, -1, B)
            slots = [oracle.decode_slot(t, kv_lens[seq_of[r]] - 1, BS) for t, r in zip(tbls, running)]
            assert meta.slot_mapping.cpu().tolist() == slots
            zero_slots = []
            if pct:
                rmv = [oracle.percentage_decode_blocks(L[i], len(tbls[i]), BS, FREQ, method, 1, NBM)[0]
                       for i in range(B)]
            elif method in ("streamingLLM", "streamingLLM-1"):
                rmv, zero_slots = oracle.decode_streaming(L, tbls, BUDGET, BS, method)
            else:
                rmv = []
                for i, rid in enumerate(running):
                    if oracle.decode_trigger(L[i], BUDGET, BS):
                        new = oracle.block_score(vc_cpu[0][tbls[i][-1]], BS, HKV, D, 1)
                        rmv.append(oracle.decode_topk_victim(scores[rid], new, method, TOPK, True))
                    else:
                        rmv.append(-1)
            M = max(len(t) for t in tbls)
            bt = torch.zeros((B, M), dtype=torch.int32)
            for i, t in enumerate(tbls):
                bt[i, :len(t)] = torch.tensor(t)
            use_tables = bt
            if pct and any(r >= 0 for r in rmv):
                use_tables = oracle.table_remove_range(
                    bt, rmv, n_rm, max(len(t) - (n_rm if r >= 0 else 0) for t, r in zip(tbls, rmv)))
            elif any(r >= 0 for r in rmv):
                use_tables = oracle.compact_block_tables(
                    bt, rmv, out_width=oracle.surviving_width([len(t) for t in tbls], rmv))
            for li in range(LAYERS):
                q, k, v = _rand(B, HQ, g, 0.4), _rand(B, HKV, g, 0.4), _rand(B, HKV, g, 1.0)
                out = impls[li].forward(q.cuda(), k.cuda(), v.cuda(), kv_gpu[li], meta)
                oracle.reshape_and_cache(k.view(B, HKV, D), v.view(B, HKV, D), kc_cpu[li], vc_cpu[li],
                                         torch.tensor(slots))
                if zero_slots:
                    z = torch.zeros(len(zero_slots), HKV, D, dtype=torch.bfloat16)
                    oracle.reshape_and_cache(z, z, kc_cpu[li], vc_cpu[li], torch.tensor(zero_slots))
                for i, mg in enumerate(merged):
                    if mg is not None:
                        oracle.merge_blocks_in_cache(kc_cpu[li], vc_cpu[li], tbls[i][mg[0]:mg[1]], method)
                assert torch.equal(kv_gpu[li][0].view_as(kc_cpu[li]).cpu(), kc_cpu[li]), (step, li)
                assert torch.equal(kv_gpu[li][1].view_as(vc_cpu[li]).cpu(), vc_cpu[li]), (step, li)
                ref = oracle.paged_attention(q.view(B, HQ, D), kc_cpu[li], vc_cpu[li], use_tables,
                                             torch.tensor(kv_adj, dtype=torch.int32), D ** -0.5)
                torch.testing.assert_close(
                    out.cpu().float().view(B, HQ, D), ref.float(), atol=3e-3, rtol=1.6e-2,
                    msg=lambda m: f"step {step} layer {li} rmv {rmv} kv {kv_adj} L {L}: {m}")
            got = impls[0].get_rmv_block_idx()
            expect = {i: (rid, r) for i, (rid, r) in enumerate(zip(running, rmv)) if r >= 0}
            assert got == expect, (step, got, expect)
            n_evictions += len(expect)
            ref_sched.commit(got)
            if vec_sched is not None:
                vec_sched.commit(got)
            for i, rid in enumerate(running):
                if rmv[i] >= 0:
                    del tbls[i][rmv[i]:rmv[i] + n_rm]
                assert bm.get_block_table(seq_of[rid]) == tbls[i]
                if vec_sched is not None:
                    assert vec_sched.get_block_table(rid) == tbls[i]
                    assert vec_sched.get_seq_kv_len(rid) == bm.get_seq_kv_len(seq_of[rid])
            if vec_sched is not None and vec_sched.device_state is not None:
                # the resident rows hold exactly the committed tables and counters
                st = vec_sched.device_state
                dev_t, dev_n = st.tables[:B].cpu(), st.num_blocks[:B].cpu().tolist()
                assert dev_n == [len(t) for t in tbls]
                for i in range(B):
                    assert dev_t[i, :dev_n[i]].tolist() == tbls[i]
                    assert not dev_t[i, dev_n[i]:].any()
                assert st.kv_lens[:B].cpu().tolist() == [bm.get_seq_kv_len(seq_of[r]) for r in running]
                assert st.seq_lens[:B].cpu().tolist() == L
            for rid in list(running):
                remaining[rid] -= 1
                if remaining[rid] == 0:
                    running.remove(rid)
                    ref_sched.finish_request(rid)
                    if vec_sched is not None:
                        vec_sched.finish_request(rid)
        step += 1
        assert step < 2000
    assert n_prefills >= 3 and n_evictions >= 4, (n_prefills, n_evictions)
    from vllm_paged_compression_amd import ops
    ops.check_device_diagnostics()   # nothing was clamped or dropped on the way
    assert bm.get_num_free_gpu_blocks() == nb
