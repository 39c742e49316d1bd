"""Device-resident block tables (batch_scheduler.DeviceTableState): what crosses PCIe per decode
step must not depend on the table width, and a generation driven through the resident tables must
evict exactly what the host-table form evicts."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BS = 16


def _generation(method, prompt_len, budget, n_decode, device_tables, B=6, Hq=8, Hkv=2, D=128, layers=2,
                formula="value_l2"):
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler
    from vllm_paged_compression_amd.config import CacheConfigView, PagedEvictConfig
    dev = torch.device("cuda:0")
    cfg = PagedEvictConfig(True, "budget", method, budget, 3)
    ccfg = CacheConfigView(BS, cfg)
    per_seq = (prompt_len + n_decode) // BS + 2
    nb = B * per_seq + 4
    impls = [PagedEvictBackend.get_impl_cls()(Hq, D, D ** -0.5, Hkv, None, None, "auto", None, None,
                                              cache_config=ccfg, prefix=f"model.layers.{i}.self_attn.attn",
                                              max_num_seqs=B, max_blocks_per_seq=per_seq + 2)
             for i in range(layers)]
    for impl in impls:
        impl.kv_cache_pruner.sub_evict_method = formula
    kv = [torch.zeros(PagedEvictBackend.get_kv_cache_shape(nb, BS, Hkv, D), dtype=torch.bfloat16, device=dev)
          for _ in range(layers)]
    sched = BatchEvictionScheduler(BS, nb, cfg, max_seqs=B, max_blocks_per_seq=per_seq + 2,
                                   device=dev if device_tables else None)
    rids = [f"r{i}" for i in range(B)]
    lens = [prompt_len + 3 * i for i in range(B)]
    for i, r in enumerate(rids):
        sched.add_request(r, i, lens[i])

    def builder(inter):
        runner = SimpleNamespace(device=dev, cache_config=ccfg, pin_memory=False)
        ib = SimpleNamespace(runner=runner, sliding_window=None, block_size=BS,
                             chunked_prefill_enabled=False, inter_data_list=inter)
        return PagedEvictBackend.make_metadata_builder(ib)

    g = torch.Generator(device=dev).manual_seed(3)

    def rnd(T, H):
        return torch.empty(T, H * D, dtype=torch.bfloat16, device=dev).uniform_(-1, 1, generator=g)

    inter = [SimpleNamespace(request_id=r, is_prompt=True, block_tables={i: sched.get_block_table(r)},
                             seq_kv_lens={i: sched.get_seq_kv_len(r)}, seq_ids=[i], input_tokens=[range(lens[i])],
                             orig_seq_lens=[lens[i]], seq_lens=[lens[i]], query_lens=[lens[i]], context_lens=[0])
             for i, r in enumerate(rids)]
    meta = builder(inter).build(lens, lens, -1, B)
    T = sum(lens)
    for li, impl in enumerate(impls):
        impl.forward(rnd(T, Hq), rnd(T, Hkv), rnd(T, Hkv), kv[li], meta)
    victims, outs, h2d = [], [], []
    for _ in range(n_decode):
        before = sched.device_state.h2d_bytes if device_tables else 0
        batch = sched.schedule_decode(rids)
        meta = builder([]).build_decode_batch(batch)
        q, k, v = rnd(B, Hq), rnd(B, Hkv), rnd(B, Hkv)
        for li, impl in enumerate(impls):
            o = impl.forward(q, k, v, kv[li], meta)
            if li == 0:
                outs.append(o.clone())
        got = impls[0].get_rmv_block_idx()
        victims.append(dict(got))
        sched.commit(got)
        if device_tables:
            h2d.append(sched.device_state.h2d_bytes - before)
    tables = [sched.get_block_table(r) for r in rids]
    if device_tables:
        st = sched.device_state
        n = st.num_blocks[:B].cpu().tolist()
        assert [st.tables[i, :n[i]].cpu().tolist() for i in range(B)] == tables
    torch.cuda.synchronize()
    return victims, outs, tables, [t.cpu() for t in kv], h2d


@pytest.mark.parametrize("method,formula", [("streamingLLM", "value_l2"), ("streamingLLM-1", "value_l2"),
                                            ("global", "value_l2"), ("local", "value_l2"),
                                            ("inverse_key_l2", "value_l2"), ("global", "key_l2"),
                                            ("local", "value_l2_plus_key_l2")])
def test_device_tables_match_host_tables(method, formula):
    import random
    from vllm_paged_compression_amd import ops
    random.seed(1234)
    a = _generation(method, 150, 128, 40, device_tables=False, formula=formula)
    random.seed(1234)
    b = _generation(method, 150, 128, 40, device_tables=True, formula=formula)
    assert a[0] == b[0], "victims differ"
    assert a[2] == b[2], "final tables differ"
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y)                 # same kernels on the same tables: bit-identical
    for x, y in zip(a[3], b[3]):
        assert torch.equal(x, y)                 # KV cache bytes
    if method != "inverse_key_l2":
        assert sum(len(v) for v in a[0]) >= 6
    ops.check_device_diagnostics()


def test_device_tables_h2d_bytes_independent_of_table_width():
    """Steady-state decode steps upload 4 bytes per sequence whatever the prompt length (the host
    form uploads the whole [B, M] matrix every step)."""
    short = _generation("streamingLLM", 300, 256, 20, device_tables=True)[4]
    long = _generation("streamingLLM", 2400, 2048, 20, device_tables=True)[4]
    B = 6
    # the first step loads the rows (the batch was just formed); every later step sends the
    # new-block ids only
    assert short[0] > 4 * B and long[0] > short[0]
    assert set(short[1:]) == {4 * B} and set(long[1:]) == {4 * B}


def test_table_state_kernels_against_host_model(oracle):
    """pe_table_state_advance + pe_evict_select_compact_inplace on random states against a plain
    Python model (the same arithmetic as block_table.EvictBlockTable / the oracle's streaming
    decision)."""
    from vllm_paged_compression_amd import ops
    rng = np.random.default_rng(5)
    B, M, budget = 37, 24, 128
    dev = torch.device("cuda:0")
    n_blocks = rng.integers(9, 12, B)
    kv = np.array([rng.integers((n - 1) * BS, n * BS + 1) for n in n_blocks])   # may sit on a boundary
    L = kv + rng.integers(0, 200, B)
    tables = np.zeros((B, M), dtype=np.int32)
    nxt = 1000
    for b in range(B):
        tables[b, :n_blocks[b]] = rng.permutation(500)[:n_blocks[b]] + 1
    new_blocks = np.full(B, -1, dtype=np.int32)
    for b in range(B):
        if kv[b] == n_blocks[b] * BS:           # full: the step's token needs a fresh block
            new_blocks[b] = nxt
            nxt += 1
    t = lambda a, dt=torch.int32: torch.tensor(np.asarray(a), dtype=dt, device=dev)  # noqa: E731
    d_tab, d_n, d_kv, d_L = t(tables), t(n_blocks), t(kv), t(L)
    slots = torch.empty(B, dtype=torch.int64, device=dev)
    attn = torch.empty(B, dtype=torch.int32, device=dev)
    last = torch.empty(B, dtype=torch.int32, device=dev)
    ops.table_state_advance(d_tab, d_n, d_kv, d_L, t(new_blocks), B, slots, attn, last, 1, budget, BS)
    exp_tab, exp_n = tables.copy(), n_blocks.copy()
    exp_slots, exp_attn = [], []
    for b in range(B):
        if new_blocks[b] >= 0:
            exp_tab[b, exp_n[b]] = new_blocks[b]
            exp_n[b] += 1
        k1, L1 = kv[b] + 1, L[b] + 1
        exp_slots.append(oracle.decode_slot(exp_tab[b, :exp_n[b]].tolist(), k1 - 1, BS))
        exp_attn.append(k1 - (BS if oracle.decode_trigger(L1, budget, BS) else 0))
    assert d_tab.cpu().numpy().tolist() == exp_tab.tolist()
    assert d_n.cpu().tolist() == exp_n.tolist() and d_kv.cpu().tolist() == (kv + 1).tolist()
    assert d_L.cpu().tolist() == (L + 1).tolist()
    assert slots.cpu().tolist() == exp_slots and attn.cpu().tolist() == exp_attn
    assert last.cpu().tolist() == [int(exp_tab[b, exp_n[b] - 1]) for b in range(B)]
    # in-place streamingLLM eviction on that state
    rmv, freed, _ = ops.evict_select_compact_inplace(d_tab, d_n, d_kv, d_L, B, "streamingLLM", budget, BS)
    exp_rmv, _ = oracle.decode_streaming((L + 1).tolist(), [r.tolist() for r in exp_tab], budget, BS,
                                         "streamingLLM")
    assert rmv.cpu().tolist() == exp_rmv
    for b in range(B):
        row = exp_tab[b, :exp_n[b]].tolist()
        if exp_rmv[b] >= 0:
            assert int(freed[b]) == row[1]
            del row[1]
        else:
            assert int(freed[b]) == -1
        got_n = int(d_n[b])
        assert got_n == len(row) and d_tab[b, :got_n].cpu().tolist() == row
        assert not d_tab[b, got_n:].any()
        assert int(d_kv[b]) == kv[b] + 1 - (BS if exp_rmv[b] >= 0 else 0)
    # a token that cannot fall into the row's last block is flagged, not written
    bad_n = t(np.full(B, 2))
    ops.table_state_advance(d_tab, bad_n, d_kv, d_L, t(np.full(B, -1)), B, slots, attn, None, 0, 0, BS)
    assert ops.device_diagnostics(reset=True) == ops.DIAG_TABLE_STATE_MISMATCH
    assert (slots == -1).all()


def test_device_rows_follow_a_request_id_that_returns():
    """finish_request + add_request of the same id between two decode steps: the resident rows are
    rebuilt (the id tuple of the batch is unchanged, the blocks behind it are not)."""
    from vllm_paged_compression_amd import ops
    from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler
    dev = torch.device("cuda:0")
    sched = BatchEvictionScheduler(BS, 64, None, max_seqs=4, max_blocks_per_seq=16, device=dev)
    sched.add_request("r", 0, 40)
    sched.schedule_decode(["r"])
    sched.finish_request("r")
    sched.add_request("x", 1, 20)
    sched.add_request("r", 2, 100)
    batch = sched.schedule_decode(["r"])
    st = sched.device_state
    n = int(st.num_blocks[0])
    assert st.loads == 2
    assert st.tables[0, :n].cpu().tolist() == sched.get_block_table("r")
    assert int(st.kv_lens[0]) == 101 == int(batch.kv_lens[0]) and int(st.seq_lens[0]) == 101
    table = sched.get_block_table("r")
    assert int(st.slot_mapping[0]) == table[100 // BS] * BS + 100 % BS
    ops.check_device_diagnostics()


def test_table_state_advance_ring_form():
    """pe_table_state_advance_ring: sequence b takes the step's entry from the ring row its new
    logical length selects (pinned host memory read in place, or a device ring); many steps queued
    back to back, checked against the plain form fed step by step."""
    from vllm_paged_compression_amd import ops
    rng = np.random.default_rng(9)
    B, M, R, steps = 21, 40, 8, 6
    dev = torch.device("cuda:0")
    n_blocks = rng.integers(3, 6, B)
    kv = np.array([rng.integers((n - 1) * BS + 1, n * BS + 1) for n in n_blocks])
    L = kv + rng.integers(0, 100, B)
    tables = np.zeros((B, M), dtype=np.int32)
    for b in range(B):
        tables[b, :n_blocks[b]] = rng.permutation(500)[:n_blocks[b]] + 1
    t = lambda a, dt=torch.int32: torch.tensor(np.asarray(a), dtype=dt, device=dev)  # noqa: E731
    per_step, kv_h, n_h, nxt = [], kv.copy(), n_blocks.copy(), 1000
    for s in range(steps):
        nb = np.full(B, -1, dtype=np.int32)
        for b in range(B):
            if kv_h[b] == n_h[b] * BS:
                nb[b] = nxt
                nxt += 1
                n_h[b] += 1
        kv_h += 1
        per_step.append(nb)

    def run(ring_tensor):
        st = [t(tables), t(n_blocks), t(kv), t(L)]
        slots = torch.empty(B, dtype=torch.int64, device=dev)
        attn = torch.empty(B, dtype=torch.int32, device=dev)
        seen = []
        for s in range(steps):
            if ring_tensor is None:
                ops.table_state_advance(*st, t(per_step[s]), B, slots, attn, None, 1, 64, BS)
            else:
                ops.table_state_advance(*st, ring_tensor, B, slots, attn, None, 1, 64, BS)
            seen.append((slots.clone(), attn.clone()))
        return [x.cpu() for x in st], [(a.cpu(), b.cpu()) for a, b in seen]

    ring = np.full((R, B + 3), -1, dtype=np.int32)
    for s in range(steps):                      # all steps stored up front: steps < R
        ring[(L + 1 + s) % R, np.arange(B)] = per_step[s]
    want = run(None)
    for ring_tensor in (torch.from_numpy(ring).pin_memory(), torch.from_numpy(ring).to(dev)):
        got = run(ring_tensor)
        for x, y in zip(want[0], got[0]):
            assert torch.equal(x, y)
        for (a, b), (c, d) in zip(want[1], got[1]):
            assert torch.equal(a, c) and torch.equal(b, d)
    assert sum(int((p >= 0).sum()) for p in per_step) >= 3
    with pytest.raises(RuntimeError):           # a pageable host ring is refused
        run(torch.from_numpy(ring))
    ops.check_device_diagnostics()


def test_deferred_advance_in_a_replayed_graph():
    """defer_advance: schedule_decode only stores the step's entries in the pinned ring and the
    launch is the first node of a HIP graph replayed once per step, far more steps than the ring
    has rows; state and slot mappings equal those of the scheduler that launches every step
    itself."""
    from vllm_paged_compression_amd import ops
    from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler, DeviceTableState
    from vllm_paged_compression_amd.config import PagedEvictConfig
    dev = torch.device("cuda:0")
    cfg = PagedEvictConfig(True, "budget", "streamingLLM", 64, 3)
    steps = 3 * DeviceTableState.RING + 5
    res = []
    for deferred in (False, True):
        sched = BatchEvictionScheduler(BS, 400, cfg, max_seqs=4, max_blocks_per_seq=32, device=dev)
        rids = ["a", "b", "c"]
        for i, r in enumerate(rids):
            sched.add_request(r, i, 50 + 7 * i)
        st = sched.device_state
        st.defer_advance = deferred
        graph, slots = None, []
        for s in range(steps):
            sched.schedule_decode(rids)
            if deferred:
                if graph is None:
                    st.launch_advance()           # eager first step (warm-up), then capture
                    graph = torch.cuda.CUDAGraph()
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side), torch.cuda.graph(graph, stream=side):
                        st.launch_advance()
                    torch.cuda.current_stream().wait_stream(side)
                else:
                    graph.replay()
            rmv, freed, _ = ops.evict_select_compact_inplace(st.tables, st.num_blocks, st.kv_lens, st.seq_lens,
                                                             3, "streamingLLM", 64, BS)
            slots.append(st.slot_mapping[:3].clone())
            rl = rmv.cpu().tolist()               # (a real engine reads the victims back like this)
            sched.commit({i: (rids[i], r) for i, r in enumerate(rl) if r >= 0})
        n = st.num_blocks[:3].cpu().tolist()
        assert [st.tables[i, :n[i]].cpu().tolist() for i in range(3)] == [sched.get_block_table(r) for r in rids]
        res.append((torch.stack(slots).cpu(), st.kv_lens[:3].cpu(), st.seq_lens[:3].cpu()))
    for x, y in zip(*res):
        assert torch.equal(x, y)
    ops.check_device_diagnostics()
