"""The "percentage" prune type (README-era flags --cache-prune-type percentage --evict-freq N):
oracle and host arithmetic pinned against fixtures made by running the reference's own code
for that mode (vllm/attention/kvcache_prunner_old.py, vllm/core/page_evict_kv_util_bak.py;
tests/golden/make_golden.py section vii).  What those files do not pin: the order in which
torch.topk drops equal scores (the f32 fixtures are tie-free), and the decode-step trigger,
which only survives as commented-out code (torch_cuda_sdpa.py:1310) - parity unpinned there."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, from_bits


def _tables():
    return json.load(open(os.path.join(GOLDEN, "percentage_tables.json")))


def _cfg(freq, nbm=2, decode="streamingLLM"):
    from vllm_paged_compression_amd.config import PagedEvictConfig
    return PagedEvictConfig(True, "percentage", "streamingLLM", 0, -1, evict_freq=freq,
                            num_block_merge=nbm, decode_evict_method=decode)


def test_oracle_prompt_tables(oracle):
    tab = _tables()
    assert len(tab["required_blocks"]) >= 100
    for bs, freq, nbm, L, nblk, ntok in tab["required_blocks"]:
        kept = oracle.percentage_keep_indices(L, bs, freq, 1, nbm)
        assert len(kept) == ntok and -(-ntok // bs) == nblk, (bs, freq, nbm, L)
    for c in tab["streaming_keep"]:
        assert oracle.percentage_keep_indices(c["L"], c["bs"], c["freq"], 1, c["nbm"]) == c["kept"], c
    # worked example: 1024 tokens, block 16, evict_freq 2: middle = 1024-32-16 = 976,
    # int(976/2) = 488 -> 480 whole-block tokens go, 544 stay
    assert oracle.percentage_prompt_slices(1024, 16, 2) == (16, 992, 480)


def test_oracle_decode_blocks(oracle):
    tab = _tables()
    for method, nbm, s, e, tokens in tab["decode_blocks"]:      # blocks merged (old pruner)
        first, n, merged = oracle.percentage_decode_blocks(64, 100, 16, 2, "value_l2", 1, nbm)
        assert merged == (s, e) and n * 16 == tokens
    for method, nbm, s, e, tokens in tab["util_blocks"]:        # blocks freed (util)
        first, n, _ = oracle.percentage_decode_blocks(64, 100, 16, 2, method, 1, nbm)
        assert (first, first + n) == (s, e) and n * 16 == tokens
    # no prune off the period or while the sequence is short
    assert oracle.percentage_decode_blocks(63, 100, 16, 2, "value_l2")[0] == -1
    assert oracle.percentage_decode_blocks(48, 100, 16, 2, "value_l2")[0] == -1
    assert oracle.percentage_decode_blocks(64, 3, 16, 2, "value_l2")[0] == -1


def test_oracle_merge_golden(oracle):
    """The kept (token, head) sets of the oracle reproduce the reference's prune_oldest_block
    output rows (which re-pack across heads and repeat the keys as values)."""
    z = np.load(os.path.join(GOLDEN, "decode_merge.npz"))
    n = len([k for k in z.files if k.endswith("_meta")])
    assert n >= 8
    for c in range(n):
        nbm, bs, H, D, prune_tokens = [int(v) for v in z[f"c{c}_meta"]]
        k, v = from_bits(z[f"c{c}_k"], 0), from_bits(z[f"c{c}_v"], 0)
        keep = oracle.merge_keep_sets(v, nbm * bs - prune_tokens)
        assert all(len(s) == bs for s in keep)
        mask = torch.zeros(nbm * bs, H, dtype=torch.bool)
        for h in range(H):
            mask[keep[h], h] = True
        pk = k[mask].view(-1, H, D)
        assert torch.equal(pk, from_bits(z[f"c{c}_pk"], 0)), c
        assert torch.equal(pk, from_bits(z[f"c{c}_pv"], 0)), c


def test_oracle_prompt_topk_golden(oracle):
    """Token-level prompt prune of the percentage type against the reference's old pruner
    (cross-head re-packing as it does it; the per-head form keeps the same rows per head)."""
    z = np.load(os.path.join(GOLDEN, "percentage_prompt_topk.npz"))
    n = len([k for k in z.files if k.endswith("_meta")])
    assert n >= 8
    for c in range(n):
        L, freq, bs, nbm, H, D, mi = [int(v) for v in z[f"c{c}_meta"]]
        method = ("value_l2", "inverse_key_l2")[mi]
        k, v = from_bits(z[f"c{c}_k"], 0), from_bits(z[f"c{c}_v"], 0)
        pk, pv = oracle.percentage_prompt_prune(k, v, method, bs, freq, 1, nbm, compat_q2=True)
        assert torch.equal(pk, from_bits(z[f"c{c}_pk"], 0)), c
        assert torch.equal(pv, from_bits(z[f"c{c}_pv"], 0)), c
        ik, iv = oracle.percentage_prompt_prune(k, v, method, bs, freq, 1, nbm)
        assert ik.shape == pk.shape == (len(oracle.percentage_keep_indices(L, bs, freq, 1, nbm)), H, D)


def test_oracle_merge_in_cache(oracle):
    g = torch.Generator().manual_seed(5)
    nb, H, D, bs, x = 6, 2, 16, 16, 8
    kc = torch.rand(nb, H, D // x, bs, x, generator=g).to(torch.bfloat16)
    vc = torch.rand(nb, H, D, bs, generator=g).to(torch.bfloat16)
    k0, v0 = kc.clone(), vc.clone()
    ids = [4, 1, 3]
    oracle.merge_blocks_in_cache(kc, vc, ids)
    untouched = [b for b in range(nb) if b != 4]
    assert torch.equal(kc[untouched], k0[untouched]) and torch.equal(vc[untouched], v0[untouched])
    # every token of the merged block is one of the source tokens, in source order, K and V of
    # the same token together
    for h in range(H):
        src_v = torch.cat([v0[b, h] for b in ids], dim=1).t()             # [48, D]
        src_k = torch.cat([k0[b, h].permute(1, 0, 2).reshape(bs, D) for b in ids])
        pos = -1
        for t in range(bs):
            hit = [i for i in range(src_v.shape[0]) if torch.equal(src_v[i], vc[4, h, :, t])]
            assert hit and hit[0] > pos
            pos = hit[0]
            assert torch.equal(kc[4, h].permute(1, 0, 2).reshape(bs, D)[t], src_k[pos])


def test_host_arithmetic_matches_reference_tables():
    from vllm_paged_compression_amd.page_evict_kv_util import (
        get_blocks_to_prune, get_num_required_blocks_after_prune_promt, percentage_decode_trigger)
    from vllm_paged_compression_amd.pruner import KVCachePruner
    tab = _tables()
    for bs, freq, nbm, L, nblk, ntok in tab["required_blocks"]:
        cfg = _cfg(freq, nbm)
        assert get_num_required_blocks_after_prune_promt(L, cfg, bs) == (nblk, ntok)
        p = KVCachePruner("percentage", "streamingLLM", bs, 0, 1, evict_freq=freq, num_block_merge=nbm)
        assert p.get_pruned_length(L) == ntok and p.prunes(L) == (ntok < L)
    for method, nbm, s, e, tokens in tab["util_blocks"]:
        assert get_blocks_to_prune(_cfg(2, nbm, method), 4096, 16) == (s, e, tokens)
        assert get_blocks_to_prune(_cfg(2, nbm, method), (1 + nbm) * 16, 16) == (-1, -1, 0)
    for method, nbm, s, e, tokens in tab["decode_blocks"]:
        p = KVCachePruner("percentage", "streamingLLM", 16, 0, 1, evict_freq=2, num_block_merge=nbm,
                          decode_evict_method=method)
        assert p.get_blocks_to_prune_and_merge_decode(4096) == (s, e, tokens)
    cfg = _cfg(4)
    assert percentage_decode_trigger(cfg, 128, 100, 16) and not percentage_decode_trigger(cfg, 112, 100, 16)
    assert not percentage_decode_trigger(cfg, 128, 48, 16)


def test_config_percentage_validation():
    from vllm_paged_compression_amd.config import PagedEvictConfig
    # the five positional arguments alone keep HEAD's behaviour: "percentage" is rejected
    with pytest.raises(AssertionError):
        PagedEvictConfig(True, "percentage", "streamingLLM", 512, -1)
    cfg = _cfg(2, 2, "value_l2")
    assert cfg.evict_freq == 2 and cfg.prompt_evict_method == "streamingLLM" \
        and cfg.decode_evict_method == "value_l2" and "evict_freq=2" in repr(cfg)
    with pytest.raises(ValueError):
        PagedEvictConfig(True, "percentage", "streamingLLM", 0, -1, evict_freq=0)
    with pytest.raises(ValueError):
        PagedEvictConfig(True, "percentage", "streamingLLM", 0, -1, evict_freq=2, num_block_merge=1)
    with pytest.raises(ValueError):
        PagedEvictConfig(True, "percentage", "streamingLLM", 0, -1, evict_freq=2,
                         decode_evict_method="global")
    with pytest.raises(ValueError):
        PagedEvictConfig(True, "percentage", "key_l3", 0, -1, evict_freq=2,
                         decode_evict_method="value_l2")
    # every token score get_score of that era knows is a method name of either phase
    c = PagedEvictConfig(True, "percentage", "key_l1", 0, -1, evict_freq=2, decode_evict_method="cosine")
    assert (c.prompt_evict_method, c.decode_evict_method) == ("key_l1", "cosine")
    assert PagedEvictConfig(True, "percentage", "value_l2", 0, -1, evict_freq=2,
                            decode_evict_method="value_l2").prompt_evict_method == "value_l2"


@pytest.mark.parametrize("method", ["streamingLLM", "value_l2"])
@pytest.mark.parametrize("nbm", [2, 3])
def test_schedulers_percentage_commit(oracle, method, nbm):
    """Both schedulers walk the same block tables as a plain list model of the mode: pruned
    prompt allocation, one slot per step, num_block_merge-1 blocks freed at every trigger."""
    from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler
    from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
    bs, freq = 16, 2
    cfg = _cfg(freq, nbm, method)
    a = EvictionScheduler(EvictBlockSpaceManager(bs, 256, cfg, watermark=0.0))
    b = BatchEvictionScheduler(bs, 256, cfg, max_seqs=4, max_blocks_per_seq=64)
    prompts = {"r0": 200, "r1": 70, "r2": 45}
    for i, (r, n) in enumerate(prompts.items()):
        a.add_request(r, i, n)
        b.add_request(r, i, n)
        kept = len(oracle.percentage_keep_indices(n, bs, freq, 1, nbm))
        assert a.block_manager.get_seq_kv_len(i) == kept == b.get_seq_kv_len(r)
        assert a.block_manager.get_block_table(i) == b.get_block_table(r)
    L = dict(prompts)
    n_freed = 0
    for step in range(90):
        a.schedule_decode(list(prompts))
        batch = b.schedule_decode(list(prompts))
        rmv = {}
        for i, r in enumerate(prompts):
            L[r] += 1
            tl = len(a.block_manager.get_block_table(i))
            assert tl == int(batch.table_lens[i])
            first, n, _ = oracle.percentage_decode_blocks(L[r], tl, bs, freq, method, 1, nbm)
            if first >= 0:
                rmv[i] = (r, first)
                n_freed += n
        before = {r: list(a.block_manager.get_block_table(i)) for i, r in enumerate(prompts)}
        a.commit(rmv)
        b.commit(rmv)
        for i, r in enumerate(prompts):
            expect = before[r]
            if i in rmv:
                expect = expect[:rmv[i][1]] + expect[rmv[i][1] + nbm - 1:]
            assert a.block_manager.get_block_table(i) == expect == b.get_block_table(r)
            assert a.block_manager.get_seq_kv_len(i) == b.get_seq_kv_len(r)
    assert n_freed >= 6
    assert a.block_manager.get_num_free_gpu_blocks() == b.get_num_free_gpu_blocks()


OLD_METHODS = ["cosine", "key_l1", "key_l2", "value_l1", "value_l2", "key_l1_div_value_l1",
               "key_l1_div_value_l2", "key_l2_div_value_l1", "key_l2_div_value_l2", "value_l1_div_key_l1",
               "value_l1_div_key_l2", "value_l2_div_key_l1", "value_l2_div_key_l2", "value_l1_plus_key_l1",
               "value_l2_plus_key_l2", "streamingLLM", "inverse_key_l1", "inverse_key_l2"]


def _ulp_apart(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in representable values of a 16-bit dtype (same sign assumed near equality)."""
    ia, ib = a.view(torch.int16).to(torch.int32), b.view(torch.int16).to(torch.int32)
    return (ia - ib).abs()


def test_oracle_readme_token_scores_golden(oracle):
    """The oracle's restatement of get_score (kvcache_prunner_old.py:77-134) against the
    reference class's own outputs for every method name (tests/golden/make_golden_old_scores.py).
    The oracle sums in f32 one element after the other (what the kernels do), torch vectorises
    its sums: after rounding to a 16-bit type the two agree except where the f32 sums straddle a
    rounding boundary - at most one representable value apart, on few entries; the divisions can
    amplify that by one more.  f32 inputs: relative 2e-6."""
    z = np.load(os.path.join(GOLDEN, "old_token_scores.npz"))
    assert [str(m) for m in z["methods"]] == OLD_METHODS
    assert sorted(oracle.README_SCORES) == sorted(OLD_METHODS)
    checked = 0
    for c in range(int(z["n_score_cases"])):
        T, H, D, code = [int(x) for x in z[f"s{c}_meta"]]
        k, v = from_bits(z[f"s{c}_k"], code), from_bits(z[f"s{c}_v"], code)
        for mi, m in enumerate(OLD_METHODS):
            want = from_bits(z[f"s{c}_m{mi}"], 0 if m == "streamingLLM" else code)
            got = oracle.readme_token_scores(k, v, m)
            if m == "streamingLLM":
                assert torch.equal(got.float(), want.float())
            elif code == 0:
                torch.testing.assert_close(got, want, rtol=2e-6, atol=1e-7)
            else:
                d = _ulp_apart(got, want)
                assert int(d.max()) <= 2, (m, code, int(d.max()))
                assert float((d == 0).float().mean()) >= 0.9, (m, code)
            checked += 1
    assert checked == 6 * len(OLD_METHODS)


def test_oracle_readme_prune_and_merge_golden(oracle):
    """Prompt prune and decode merge under README-era scores against the reference's own
    prune_prompt / prune_oldest_block (tie-free f32 inputs): the reference re-packs the kept
    (token, head) rows across heads (compat form), so the kept rows are compared as it lays them
    out; for the merge the kept key rows."""
    z = np.load(os.path.join(GOLDEN, "old_token_scores.npz"))
    for c in range(int(z["n_prune_cases"])):
        L, freq, bs, nbm, H, D, mi = [int(x) for x in z[f"p{c}_meta"]]
        k, v = from_bits(z[f"p{c}_k"], 0), from_bits(z[f"p{c}_v"], 0)
        pk, pv = oracle.percentage_prompt_prune(k, v, OLD_METHODS[mi], bs, freq, 1, nbm, compat_q2=True)
        assert torch.equal(pk, from_bits(z[f"p{c}_pk"], 0)), (c, OLD_METHODS[mi])
        assert torch.equal(pv, from_bits(z[f"p{c}_pv"], 0)), (c, OLD_METHODS[mi])
    for c in range(int(z["n_merge_cases"])):
        nbm, bs, H, D, prune_tokens, mi = [int(x) for x in z[f"d{c}_meta"]]
        k, v = from_bits(z[f"d{c}_k"], 0), from_bits(z[f"d{c}_v"], 0)
        keep = oracle.merge_keep_sets(v, nbm * bs - prune_tokens, k, OLD_METHODS[mi])
        mask = torch.zeros(nbm * bs, H, dtype=torch.bool)
        for h in range(H):
            mask[keep[h], h] = True
        assert torch.equal(k[mask].view(-1, H, D), from_bits(z[f"d{c}_pk"], 0)), (c, OLD_METHODS[mi])


def test_percentage_decode_victims_follow_the_trigger_and_the_reference_columns(oracle):
    """backend.percentage_decode_victims (what a graph-replaying engine commits without a
    read-back): a row is listed exactly when page_evict_kv_util.percentage_decode_trigger fires
    for it, with the first freed column of the reference's get_blocks_to_prune
    (page_evict_kv_util_bak.py:66-106; fixtures in percentage_tables.json via the oracle)."""
    import numpy as np
    from vllm_paged_compression_amd.backend import percentage_decode_victims
    from vllm_paged_compression_amd.config import PagedEvictConfig
    from vllm_paged_compression_amd.page_evict_kv_util import percentage_decode_trigger
    BS = 16
    rng = np.random.default_rng(3)
    for method, freq, nbm in (("streamingLLM", 2, 2), ("value_l2", 2, 2), ("value_l2", 4, 3), ("cosine", 3, 2)):
        cfg = PagedEvictConfig(True, "percentage", "streamingLLM", 0, 1, evict_freq=freq, num_block_merge=nbm,
                               prompt_evict_method="streamingLLM", decode_evict_method=method)
        n_tab = rng.integers(1, 12, 40)
        L = n_tab * BS + rng.integers(0, 5, 40) * BS * freq      # many rows on a multiple of freq blocks
        L[::3] += 5
        ids = [f"r{i}" for i in range(40)]
        got = percentage_decode_victims(cfg, L, n_tab, ids, BS)
        for i in range(40):
            first, n, _ = oracle.percentage_decode_blocks(int(L[i]), int(n_tab[i]), BS, freq, method,
                                                          cfg.initial_blocks, nbm)
            fires = percentage_decode_trigger(cfg, int(L[i]), int(n_tab[i]) * BS, BS)
            assert (i in got) == fires == (first >= 0)
            if fires:
                assert got[i] == (ids[i], first) and n == nbm - 1
        assert got
