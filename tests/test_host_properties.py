"""Property tests (hypothesis) of the host-side eviction arithmetic and block bookkeeping: random
prompt lengths, budgets and operation sequences instead of hand-picked cases.  CPU only."""
import random

import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from vllm_paged_compression_amd import page_evict_kv_util as util
from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
from vllm_paged_compression_amd.config import PagedEvictConfig

BLOCK_SIZES = (8, 16, 32)
# derandomize: the same examples on every run (a suite that must stay green draws no fresh cases)
COMMON = dict(deadline=None, derandomize=True, database=None,
              suppress_health_check=[HealthCheck.function_scoped_fixture])


def _cfg(method, budget, topk=3, disable_prefill=True):
    """The configuration and the budget it ends up with (the constructor raises small budgets to
    128 tokens and rounds to whole 16-token blocks)."""
    cfg = PagedEvictConfig(disable_prefill, "budget", method, budget, topk)
    return cfg, cfg.cache_budget


@st.composite
def geometry(draw):
    bs = draw(st.sampled_from(BLOCK_SIZES))
    budget = bs * draw(st.integers(3, 300))
    length = draw(st.one_of(st.integers(1, 4 * budget + 5 * bs), st.integers(1, 70000)))
    return bs, budget, length


@settings(max_examples=300, **COMMON)
@given(geometry(), st.sampled_from(["streamingLLM", "streamingLLM-1", "global", "local", "inverse_key_l2"]))
def test_prompt_prune_arithmetic(oracle, geom, method):
    """Kept length / block count of a pruned prompt (page_evict_kv_util.py:6-36,
    kvcache_prunner.py:284-326): closed form, bounds, and the oracle's restatement."""
    from vllm_paged_compression_amd.pruner import KVCachePruner
    bs, budget, L = geom
    cfg, budget = _cfg(method, budget)
    blocks, kept = util.get_num_required_blocks_after_prune_promt(L, cfg, bs)
    assert 1 <= kept <= L and blocks == -(-kept // bs)
    if L <= budget:
        assert kept == L
    else:
        rem = L % bs
        assert kept == budget + rem          # first block + budgeted middle + last block + remainder
        first, end_mid, n_prune = oracle.prompt_slices(L, budget, bs)
        assert first == bs and end_mid == L - bs - rem
        assert n_prune == max(end_mid - bs - (budget - 2 * bs), 0) == L - kept
    assert kept == oracle.pruned_length(L, budget, bs)
    assert (blocks, kept) == tuple(oracle.num_required_blocks_after_prune(L, budget, bs))
    pruner = KVCachePruner("budget", method, bs, budget, 1)
    assert pruner.get_pruned_length(L) == kept and pruner.prunes(L) == (L > budget)
    if method.startswith("streamingLLM") and L <= 5000:
        idx = oracle.streaming_keep_indices(L, budget, bs)
        assert len(idx) == kept and idx == sorted(set(idx)) and idx[0] == 0 and idx[-1] == L - 1
        assert idx[:min(bs, L)] == list(range(min(bs, L)))


@settings(max_examples=300, **COMMON)
@given(geometry(), st.integers(0, 400), st.sampled_from(["streamingLLM", "streamingLLM-1", "global", "local"]))
def test_decode_trigger_and_kv_len_adjustment(oracle, geom, extra, method):
    """The decode trigger (torch_cuda_sdpa.py:1390-1403) and the model runner's kv-length
    adjustment on eviction steps (model_runner.py:538-547)."""
    bs, budget, L = geom
    cfg, budget = _cfg(method, budget)
    seq_len = L + extra
    trig = util.budget_decode_trigger(cfg, seq_len, bs)
    assert trig == (seq_len > budget and seq_len % bs == 0) == oracle.decode_trigger(seq_len, budget, bs)
    kv = min(seq_len, budget + bs)
    assert util.evicted_tokens(cfg, seq_len, kv, bs) == (bs if trig else 0)
    assert util.adjust_seq_kv_len(cfg, seq_len, kv, bs) == kv - (bs if trig else 0)
    import numpy as np
    lens = np.array([seq_len, seq_len + 1, budget, budget + bs], dtype=np.int64)
    kvs = np.minimum(lens, budget + bs)
    adj = util.adjust_seq_kv_lens(cfg, lens, kvs, bs)
    assert adj.tolist() == [util.adjust_seq_kv_len(cfg, int(a), int(b), bs) for a, b in zip(lens, kvs)]
    # inverse_key_l2 never removes a block at decode
    cfg_inv, _ = _cfg("inverse_key_l2", budget)
    assert util.evicted_tokens(cfg_inv, seq_len, kv, bs) == 0


@settings(max_examples=60, **COMMON)
@given(st.sampled_from(BLOCK_SIZES), st.integers(3, 12), st.integers(0, 2 ** 31 - 1),
       st.sampled_from(["streamingLLM", "global", "local"]), st.booleans())
def test_block_bookkeeping_invariants(bs, budget_blocks, seed, method, vectorised):
    """Random request turnover, decode steps and commits at the trigger, on the per-request
    scheduler and on the vectorised one: no block is ever owned twice, free + owned is the pool,
    the physical length follows the logical one minus what was evicted, and in steady state a
    sequence holds between budget and budget + block_size tokens."""
    from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler
    cfg, budget = _cfg(method, budget_blocks * bs)
    budget_blocks = budget // bs
    nb = 600
    if vectorised:
        sched = BatchEvictionScheduler(bs, nb, cfg, max_seqs=12, max_blocks_per_seq=budget_blocks + 40)
        table_of = sched.get_block_table
        kv_of = sched.get_seq_kv_len
        free = sched.get_num_free_gpu_blocks
    else:
        sched = EvictionScheduler(EvictBlockSpaceManager(bs, nb, cfg, watermark=0.0))
        seq_ids = {}
        table_of = lambda r: list(sched.block_manager.get_block_table(seq_ids[r]))  # noqa: E731
        kv_of = lambda r: sched.block_manager.get_seq_kv_len(seq_ids[r])            # noqa: E731
        free = sched.block_manager.get_num_free_gpu_blocks
    rng = random.Random(seed)
    logical, running, next_id = {}, [], 0

    def check():
        owned = []
        for r in running:
            t = table_of(r)
            owned += t
            assert len(t) == -(-kv_of(r) // bs), (r, len(t), kv_of(r))
            assert kv_of(r) <= logical[r]
            if logical[r] > budget + bs:
                assert budget <= kv_of(r) <= budget + bs
        assert len(owned) == len(set(owned))
        assert free() + len(owned) == nb

    for step in range(120):
        if len(running) < 6 and rng.random() < 0.3:
            r, n = f"r{next_id}", rng.randint(1, 3 * budget)
            if not vectorised:
                seq_ids[r] = next_id
            sched.add_request(r, next_id, n)
            running.append(r)
            logical[r] = n
            next_id += 1
            kept = util.get_num_required_blocks_after_prune_promt(n, cfg, bs)[1]
            assert kv_of(r) == kept
        if running and rng.random() < 0.05:
            r = running.pop(rng.randrange(len(running)))
            sched.finish_request(r)
            del logical[r]
        if not running:
            continue
        out = sched.schedule_decode(running)
        rmv = {}
        for i, r in enumerate(running):
            logical[r] += 1
            if util.budget_decode_trigger(cfg, logical[r], bs):
                n = len(table_of(r))
                hi = min(n - 1, 1 + cfg.topk_blocks) if method == "local" else n - 1
                rmv[i] = (r, 1 if method == "streamingLLM" else rng.randint(1, max(1, hi - 1)))
        sched.commit(rmv)
        del out
        check()
    for r in list(running):
        sched.finish_request(r)
    assert free() == nb


def test_percentage_prompt_arithmetic_matches_oracle(oracle):
    """README-era percentage type (page_evict_kv_util_bak.py:6-106): the kept length of a
    prompt for random lengths and evict frequencies."""
    rng = random.Random(5)
    for _ in range(400):
        bs = rng.choice(BLOCK_SIZES)
        freq = rng.randint(1, 8)
        merge = rng.randint(2, 4)
        L = rng.randint(1, 9000)
        pruned, kept = util.percentage_pruned_prompt(L, bs, freq, 1, merge)
        first, end_mid, n_prune = oracle.percentage_prompt_slices(L, bs, freq, 1, merge)
        idx = oracle.percentage_keep_indices(L, bs, freq, 1, merge)
        assert kept == len(idx) <= L and pruned == n_prune == L - kept and pruned % bs == 0
        assert idx == sorted(set(idx)) and idx[-1] == L - 1
        if pruned:
            assert first == bs and end_mid == L - merge * bs - L % bs
            assert idx[:bs] == list(range(bs)) and idx[bs] == bs + pruned
