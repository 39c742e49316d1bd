"""CPU-side checks: the C-ABI library loads and exports every symbol include/paged_evict.h
declares (no compute calls), the host logic of the path (config, pruned lengths, block-table
counters, scheduler, slot mapping, score bookkeeping) against the oracle, and that the product
package never reaches into oracle/."""
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "paged_evict.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vllm_paged_compression_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 14
    assert sorted(_lib.SIGNATURES) == syms, "binding table and header disagree"
    lib = _lib.load()
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.pe_version() >= 101
    # workspace helpers are pure host functions
    # sized for the largest split count a launch may use: one split per 4 block pairs of 32 tokens,
    # at most 64
    assert lib.pe_paged_decode_workspace_bytes(64, 32, 8, 128, 16, 776) == 6 * 64 * 32 * (128 + 2) * 4
    assert lib.pe_paged_decode_workspace_bytes(64, 32, 8, 128, 16, 100000) == 64 * 64 * 32 * (128 + 2) * 4
    assert lib.pe_paged_decode_workspace_bytes(64, 32, 8, 128, 16, 220) == 0      # 7 pairs: one split
    prev = 0
    for n in range(1, 40000, 97):       # never shrinks as the maximum length grows
        cur = lib.pe_paged_decode_workspace_bytes(3, 32, 8, 128, 16, n)
        assert cur >= prev
        prev = cur
    assert lib.pe_paged_decode_workspace_bytes(0, 32, 8, 128, 16, 776) == 0
    assert lib.pe_prompt_token_select_workspace_bytes(100, 8) == ((100 * 8 * 4 + 255) // 256) * 256 + 800


def test_argument_checks_without_gpu():
    """Entry points validate before touching the device: bad arguments come back as a status
    + message even on a machine without a GPU."""
    from vllm_paged_compression_amd import _lib
    lib = _lib.load()
    rc = lib.pe_paged_decode(None, None, None, None, None, None, None, 4, 32, 8, 128, 16, 4, 4096, 1, 1,
                             0.1, 64, 2, None, 0, None)
    assert rc == 1 and b"null pointer" in lib.pe_last_error()
    rc = lib.pe_evict_select_compact(1, 1, None, 1, 1, 1, None, 2, 8, 7, 99, 128, 16, -1, 1, 0, 1,
                                     None, None, None, 0, None, 0, 8, 128, None, 2, None)
    assert rc == 1 and b"unknown eviction method" in lib.pe_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc, "evict_select_compact")
    assert lib.pe_reshape_and_cache(None, None, None, None, None, 0, 0, 0, None, 0, None, 0, 8, 128, 16,
                                    4, 2, None) == 1
    # a fused decode step the MFMA kernel cannot serve (block size 4; 8, 16 and 32 are served) must
    # be refused, not run as attention-only with the step's token left unwritten; nothing is
    # dereferenced before that
    p = 1 << 20
    assert lib.pe_paged_decode_supports_fused_write(32, 8, 128, 4, 2) == 0
    assert [lib.pe_paged_decode_supports_fused_write(32, 8, 128, b, 2) for b in (8, 16, 32)] == [1, 1, 1]
    # 8- and 32-token blocks: the head sizes whose rows are whole 32-element k-steps
    assert [lib.pe_paged_decode_supports_fused_write(32, 8, d, b, 2) for d in (32, 96, 192, 256)
            for b in (8, 32)] == [1] * 8
    assert [lib.pe_paged_decode_supports_fused_write(32, 8, d, b, 2) for d in (80, 112, 120)
            for b in (8, 32)] == [0] * 6
    # every head size of the reference's op (paged_attn.py:37) with 16-token blocks; nothing else
    assert [lib.pe_paged_decode_supports_fused_write(32, 8, d, 16, 2)
            for d in (32, 64, 80, 96, 112, 120, 128, 192, 256)] == [1] * 9
    assert [lib.pe_paged_decode_supports_fused_write(32, 8, d, 16, 2) for d in (16, 48, 72, 160, 512)] == [0] * 5
    rc = lib.pe_decode_step(p, p, p, p, p, p, p, p, p, None, 1, 32, 8, 128, 4, 4, 4, 4096, 1024, 1024,
                            8192, 1024, 0.1, 32, 2, None, 0, None)
    assert rc == 2 and b"fused KV write" in lib.pe_last_error()
    # the rotating decode step refuses, before touching the device, what its fragment layout
    # cannot rotate in one lane (NeoX pairing with rot_dim not a multiple of 64), head size 256
    # and shapes outside the MFMA path; the Python-side predicate says the same beforehand
    import vllm_paged_compression_amd.ops as ops

    def step_rope(head, block, rot, neox):
        return lib.pe_decode_step_rope(p, p, p, p, p, p, p, p, p, None, p, p, rot, neox, 4096, 1, 32, 8,
                                       head, block, 4, 4, 4096, 1024, 1024, 8192, 1024, 0.1, 32, 2, None,
                                       0, None)
    assert step_rope(128, 16, 48, 1) == 2 and b"rot_dim 48" in lib.pe_last_error()
    assert step_rope(256, 16, 128, 1) == 2 and b"head_size 256" in lib.pe_last_error()
    assert step_rope(128, 4, 128, 1) == 2 and b"MFMA decode path" in lib.pe_last_error()
    assert step_rope(96, 16, 64, 1) == 2 and b"head sizes 64 and 128" in lib.pe_last_error()
    # (positions may be NULL - the per-step table form - so the missing pointer here is the table;
    # a NULL positions with a table of fewer rows than sequences is refused as well)
    assert lib.pe_decode_step_rope(p, p, p, p, p, p, p, p, p, None, p, None, 128, 1, 4096, 1, 32, 8, 128,
                                   16, 4, 4, 4096, 1024, 1024, 8192, 1024, 0.1, 32, 2, None, 0,
                                   None) == 1 and b"null pointer" in lib.pe_last_error()
    assert lib.pe_decode_step_rope(p, p, p, p, p, p, p, p, p, None, None, p, 128, 1, 2, 3, 32, 8, 128,
                                   16, 4, 4, 4096, 1024, 1024, 8192, 1024, 0.1, 32, 2, None, 0,
                                   None) == 1 and b"one row per sequence" in lib.pe_last_error()
    bf = torch.bfloat16
    assert ops.decode_step_rope_supported(32, 8, 128, 16, bf, 128, True)
    assert ops.decode_step_rope_supported(32, 8, 128, 16, bf, 64, True)
    assert ops.decode_step_rope_supported(32, 8, 128, 16, bf, 48, False)
    assert not ops.decode_step_rope_supported(32, 8, 128, 16, bf, 48, True)
    assert not ops.decode_step_rope_supported(32, 8, 256, 16, bf, 128, True)
    assert ops.decode_step_rope_supported(32, 8, 128, 8, bf, 128, True)       # 8- and 32-token blocks too
    assert ops.decode_step_rope_supported(32, 8, 64, 32, bf, 64, True)
    assert not ops.decode_step_rope_supported(32, 8, 96, 16, bf, 64, True)    # head sizes 64 and 128 only
    assert not ops.decode_step_rope_supported(32, 8, 128, 16, torch.float32, 128, True)


def test_ops_reject_cpu_tensors():
    import vllm_paged_compression_amd.ops as ops
    t = torch.zeros(2, 8, 128, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.reshape_and_cache(t, t, torch.zeros(4, 8, 16, 16, 8, dtype=torch.bfloat16),
                              torch.zeros(4, 8, 128, 16, dtype=torch.bfloat16),
                              torch.zeros(2, dtype=torch.int64))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vllm-paged-compression_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pe_oracle" not in src and "oracle/" not in src and "import oracle" not in src, f
    # the measurement and tuning scripts time the product too: none of them may load the checker
    # (tools/sanitize_oracle.sh builds the checker itself, for the CPU suite)
    tools = os.path.join(ROOT, "tools")
    for dirpath, dirs, files in os.walk(tools):
        dirs[:] = [d for d in dirs if d != "_exp"]
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "pe_oracle" not in src and "import oracle" not in src, f
    # bench.py: only inside cpu_baseline()
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for node in tree.body:
        seg = ast.get_source_segment(open(os.path.join(ROOT, "bench.py")).read(), node) or ""
        if "pe_oracle" in seg or "oracle" in seg.replace("the oracle", ""):
            assert isinstance(node, ast.FunctionDef) and node.name == "cpu_baseline", \
                f"bench.py touches the checker outside cpu_baseline(): {seg[:80]}"


def test_paged_evict_config_validation():
    from vllm_paged_compression_amd.config import PagedEvictConfig, budget_from_percentage
    c = PagedEvictConfig(True, "budget", "global", 64, 5)
    assert c.cache_budget == 128 and c.topk_blocks == -1 and c.initial_blocks == 1
    with pytest.raises(ValueError):
        PagedEvictConfig(True, "budget", "local", 512, 1)
    with pytest.raises(ValueError):
        PagedEvictConfig(True, "nope", "local", 512, 3)
    with pytest.raises(AssertionError):
        PagedEvictConfig(True, "percentage", "streamingLLM", 512, 3)
    assert budget_from_percentage(0.25, 1024, 16) == 768
    assert budget_from_percentage(0.5, 1024, 16) == 512
    assert budget_from_percentage(0.5, 8192, 16) == 4096


def test_pruned_lengths_match_oracle(oracle):
    from vllm_paged_compression_amd.config import PagedEvictConfig
    from vllm_paged_compression_amd.page_evict_kv_util import get_num_required_blocks_after_prune_promt
    from vllm_paged_compression_amd.pruner import KVCachePruner
    for bs in (8, 16, 32):
        for budget in (128, 256, 768, 4096):
            cfg = PagedEvictConfig(True, "budget", "streamingLLM", budget, -1)
            pr = KVCachePruner("budget", "streamingLLM", bs, budget, 1)
            for L in (1, bs, budget - 1, budget, budget + 1, budget + bs + 3, 1024, 1030, 8192, 32768):
                assert pr.get_pruned_length(L) == oracle.pruned_length(L, budget, bs)
                assert get_num_required_blocks_after_prune_promt(L, cfg, bs) == \
                    oracle.num_required_blocks_after_prune(L, budget, bs)


def test_block_table_counters_match_oracle(oracle):
    from vllm_paged_compression_amd.block_table import BlockAllocator, EvictBlockTable
    from vllm_paged_compression_amd.config import PagedEvictConfig
    for prompt, budget in [(1024, 768), (1030, 512), (100, 128), (200, 128), (513, 512)]:
        cfg = PagedEvictConfig(True, "budget", "streamingLLM", budget, -1)
        bt = EvictBlockTable(16, BlockAllocator(4096), cfg)
        ref = oracle.BlockTableCounters(16, budget)
        bt.allocate(prompt)
        ref.allocate(prompt)
        for _ in range(150):
            bt.append_token_ids(1)
            ref.append_token()
            assert (bt.num_full_slots, bt.get_seq_kv_len(), len(bt.physical_block_ids)) == \
                (ref.num_full_slots, ref.num_cached_slots, len(ref.blocks))
            if oracle.decode_trigger(bt.num_full_slots, budget, 16):
                bt.free_prunned_blocks(1)
                ref.free_pruned_block(1)
        assert budget <= bt.get_seq_kv_len() <= budget + 16 or prompt + 150 <= budget + 16


def test_scheduler_admission_and_commit():
    from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
    from vllm_paged_compression_amd.config import PagedEvictConfig
    cfg = PagedEvictConfig(True, "budget", "streamingLLM", 128, -1)
    bm = EvictBlockSpaceManager(16, 20, cfg, watermark=0.0)
    assert bm.num_required_blocks(1024) == 8          # pruned admission size, not 64
    assert bm.can_allocate(1024) == "OK"
    sched = EvictionScheduler(bm)
    sched.add_request("a", 0, 1024)
    sched.add_request("b", 1, 1000)
    assert bm.can_allocate(1024) == "LATER" and bm.can_allocate(100000) == "LATER"
    assert EvictBlockSpaceManager(16, 4, cfg, 0.0).can_allocate(1024) == "NEVER"
    free0 = bm.get_num_free_gpu_blocks()
    t, kv = sched.schedule_decode(["a", "b"])
    assert kv[0] == 129 and len(t[0]) == 9
    victim = t[0][1]
    sched.commit({0: ("a", 1)})
    assert victim not in bm.get_block_table(0) and bm.get_seq_kv_len(0) == 113
    assert bm.get_num_free_gpu_blocks() == free0  # one block appended for "a", one freed
    assert 0.0 < bm.gpu_cache_usage() <= 1.0
    sched.finish_request("a")
    sched.finish_request("b")
    assert bm.get_num_free_gpu_blocks() == 20


def test_slot_mapping_matches_oracle(oracle):
    from vllm_paged_compression_amd.backend import compute_slot_mapping
    table = {7: [5, 9, 2, 11]}
    for qlen in (64, 50, 49):
        out = []
        compute_slot_mapping(False, True, out, 7, qlen, 0, 16, table)
        assert out == oracle.prompt_slot_mapping(table[7], qlen, 16)
    out = []
    compute_slot_mapping(False, False, out, 7, 1, 53, 16, table)
    assert out == [oracle.decode_slot(table[7], 53, 16)]
    out = []
    compute_slot_mapping(True, True, out, 7, 5, 0, 16, None)
    assert out == [-1] * 5


def test_metadata_roundtrip_through_kwargs():
    """Non-driver TP workers rebuild the metadata from asdict_zerocopy() kwargs."""
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    m = PagedEvictBackend.make_metadata(
        num_prefills=0, num_prefill_tokens=0, num_decode_tokens=2, slot_mapping=torch.tensor([3, 4]),
        seq_lens=[10, 20], seq_lens_tensor=torch.tensor([10, 20], dtype=torch.int32),
        max_decode_seq_len=20, block_tables=torch.zeros(2, 2, dtype=torch.int32),
        seq_kv_lens=[10, 4], seq_kv_lens_tensor=torch.tensor([10, 4], dtype=torch.int32),
        reqid_mapping_blocktables={"a": {0: [1]}, "b": {1: [2, 3]}})
    m2 = PagedEvictBackend.make_metadata(**m.asdict_zerocopy())
    assert m2.seq_kv_lens == [10, 4] and m2.reqid_mapping_blocktables == m.reqid_mapping_blocktables
    assert m2.decode_metadata.seq_kv_lens_tensor.tolist() == [10, 4] and m2.prefill_metadata is None
    assert m2.get_seq_lens() == ([10, 20], [10, 4])
    assert PagedEvictBackend.get_kv_cache_shape(10, 16, 8, 128) == (2, 10, 16 * 8 * 128)
    assert PagedEvictBackend.get_name() == "PAGED_EVICT_HIP"


def test_impl_constructor_errors():
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    impl_cls = PagedEvictBackend.get_impl_cls()
    with pytest.raises(ValueError):
        impl_cls(8, 128, 0.1, 8, None, None, "auto", {"x": 1}, None)
    with pytest.raises(ValueError):
        impl_cls(8, 100, 0.1, 8, None, None, "auto", None, None)
    with pytest.raises(NotImplementedError):
        impl_cls(8, 128, 0.1, 8, None, None, "fp8", None, None)
    impl = impl_cls(32, 128, 0.1, 8, None, None, "auto", None, None, prefix="model.layers.7.self_attn.attn")
    assert impl.layer_idx == 7 and impl.get_rmv_block_idx() == {}


def test_random_victim_stream(oracle):
    """Layer-0 draws of the reference's random mode: seeded 1234, the other layers consume
    the same number of draws from the shared generator."""
    import random
    s = oracle.RandomVictimStream(num_layers=4)
    got = [s.step([(1, 10), (1, 46)]) for _ in range(3)]
    rng = random.Random(1234)
    exp = []
    for _ in range(3):
        layers = [[rng.randint(1, 10), rng.randint(1, 46)] for _ in range(4)]
        exp.append(layers[0])
    assert got == exp


@pytest.mark.parametrize("nbm", [2, 3, 4])
def test_batch_scheduler_percentage_frees_blocks_in_the_engine_order(nbm):
    """Percentage prune type: a prune frees num_block_merge-1 blocks of a request, and several
    requests prune on the same step.  The blocks must return to the free list request by request
    (llm_engine.py:1189-1193 commits one request after the other), or the block ids handed out
    afterwards differ from the per-request scheduler's."""
    from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler
    from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
    from vllm_paged_compression_amd.config import PagedEvictConfig
    from vllm_paged_compression_amd.page_evict_kv_util import percentage_decode_trigger
    bs, nb, freq = 16, 200, 2
    cfg = PagedEvictConfig(True, "percentage", "streamingLLM", 0, -1, evict_freq=freq, num_block_merge=nbm,
                           decode_evict_method="value_l2")
    ref = EvictionScheduler(EvictBlockSpaceManager(bs, nb, cfg, watermark=0.0))
    vec = BatchEvictionScheduler(bs, nb, cfg, max_seqs=8, max_blocks_per_seq=40)
    prompts = {"a": 200, "b": 200, "c": 232, "d": 140}      # a, b, c prune on the same steps
    seq_of = {r: i for i, r in enumerate(prompts)}
    for r, n in prompts.items():
        ref.add_request(r, seq_of[r], n)
        vec.add_request(r, seq_of[r], n)
    running, logical, prunes = list(prompts), dict(prompts), 0
    for step in range(100):
        tables, kv_lens = ref.schedule_decode(running)
        batch = vec.schedule_decode(running)
        rmv = {}
        for i, r in enumerate(running):
            logical[r] += 1
            assert batch.table_row(i) == list(tables[seq_of[r]]), (step, r)
            assert int(batch.kv_lens[i]) == kv_lens[seq_of[r]]
            if percentage_decode_trigger(cfg, logical[r], kv_lens[seq_of[r]], bs):
                rmv[i] = (r, cfg.initial_blocks + 1)
        prunes += len(rmv) > 1
        ref.commit(rmv)
        vec.commit(rmv)
        assert vec.get_num_free_gpu_blocks() == ref.block_manager.get_num_free_gpu_blocks()
    assert prunes >= 2


def test_batch_scheduler_matches_per_request_scheduler():
    """The vectorised scheduler (struct of arrays, one numpy step per decode step) walks the same
    states as the per-request one: block tables, physical/logical lengths, free-block counts,
    through allocation, appends across block boundaries, evictions at any index (including the
    last block) and request turnover; and build_decode_batch yields the metadata build() yields."""
    import random
    from types import SimpleNamespace
    import numpy as np
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler
    from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
    from vllm_paged_compression_amd.config import CacheConfigView, PagedEvictConfig
    bs, nb, budget = 16, 400, 64
    cfg = PagedEvictConfig(True, "budget", "global", budget, 3)
    ref = EvictionScheduler(EvictBlockSpaceManager(bs, nb, cfg, watermark=0.0))
    vec = BatchEvictionScheduler(bs, nb, cfg, max_seqs=8, max_blocks_per_seq=32)
    rng = random.Random(3)
    prompts = {"a": 70, "b": 64, "c": 200, "d": 17, "e": 100}
    seq_of = {r: i for i, r in enumerate(prompts)}
    for r, n in prompts.items():
        ref.add_request(r, seq_of[r], n)
        vec.add_request(r, seq_of[r], n)
    running = list(prompts)
    ccfg = CacheConfigView(bs, cfg)

    def builder(inter):
        runner = SimpleNamespace(device=torch.device("cpu"), cache_config=ccfg, pin_memory=False)
        ib = SimpleNamespace(runner=runner, sliding_window=None, block_size=bs,
                             chunked_prefill_enabled=False, inter_data_list=inter)
        return PagedEvictBackend.make_metadata_builder(ib)

    logical = dict(prompts)
    for step in range(90):
        tables, kv_lens = ref.schedule_decode(running)
        batch = vec.schedule_decode(running)
        assert vec.get_num_free_gpu_blocks() == ref.block_manager.get_num_free_gpu_blocks()
        for i, r in enumerate(running):
            logical[r] += 1
            sid = seq_of[r]
            assert batch.block_tables[i, :batch.table_lens[i]].tolist() == tables[sid], (step, r)
            assert not batch.block_tables[i, batch.table_lens[i]:].any()
            assert int(batch.kv_lens[i]) == kv_lens[sid] and int(batch.seq_lens[i]) == logical[r]
        # metadata equality (per-request build vs batched build)
        inter = [SimpleNamespace(request_id=r, is_prompt=False, block_tables={seq_of[r]: tables[seq_of[r]]},
                                 seq_kv_lens={seq_of[r]: kv_lens[seq_of[r]]}, seq_ids=[seq_of[r]],
                                 input_tokens=[[0]], orig_seq_lens=[logical[r]], seq_lens=[logical[r]],
                                 query_lens=[1], context_lens=[kv_lens[seq_of[r]] - 1]) for r in running]
        m1 = builder(inter).build([logical[r] for r in running], [1] * len(running), -1, len(running))
        m2 = builder([]).build_decode_batch(batch, batch.kv_lens)
        for f in ("slot_mapping", "seq_lens_tensor", "query_start_loc", "seq_start_loc",
                  "context_lens_tensor", "block_tables", "seq_kv_lens_tensor"):
            assert torch.equal(getattr(m1, f), getattr(m2, f)), (step, f)
        assert m1.seq_lens == m2.seq_lens and m1.seq_kv_lens == m2.seq_kv_lens
        assert (m1.max_decode_seq_len, m1.num_decode_tokens) == (m2.max_decode_seq_len, m2.num_decode_tokens)
        # evict somewhere in some sequences (any index >= 1, the last block included)
        rmv = {}
        for i, r in enumerate(running):
            n = int(batch.table_lens[i])
            if n > 3 and rng.random() < 0.3:
                rmv[i] = (r, rng.randint(1, n - 1))
        ref.commit(rmv)
        vec.commit(rmv)
        for r in running:
            assert vec.get_block_table(r) == ref.block_manager.get_block_table(seq_of[r])
            assert vec.get_seq_kv_len(r) == ref.block_manager.get_seq_kv_len(seq_of[r])
        if step == 40:      # turnover: one request leaves, a new one arrives
            ref.finish_request("b")
            vec.finish_request("b")
            running.remove("b")
            ref.add_request("f", 9, 130)
            vec.add_request("f", 9, 130)
            seq_of["f"], logical["f"] = 9, 130
            running.append("f")
    for r in running:
        ref.finish_request(r)
        vec.finish_request(r)
    assert vec.get_num_free_gpu_blocks() == nb == ref.block_manager.get_num_free_gpu_blocks()


def test_batch_scheduler_reloads_device_rows_when_a_request_id_returns():
    """A request that finishes and comes back under the same id (or a lone request preempted and
    recomputed) gives the next decode batch the same id tuple as the resident rows: the device
    rows must be rebuilt from the host record all the same, because the blocks and lengths
    behind the id are new."""
    import numpy as np
    from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler

    class RecordingState:
        """Stands in for DeviceTableState (which needs a GPU): records what is loaded."""
        def __init__(self):
            self.key, self.score_rows_key, self.score_rows, self.loads = None, None, None, []

        def load(self, key, tables, n_blocks, kv_lens, seq_lens):
            self.loads.append((key, tables.copy(), n_blocks.copy(), kv_lens.copy(), seq_lens.copy()))
            self.key = key

        def advance(self, *a, **k):
            pass

    sched = BatchEvictionScheduler(16, 64, None, max_seqs=4, max_blocks_per_seq=16)
    st = sched.device_state = RecordingState()
    sched.add_request("r", 0, 40)
    sched.schedule_decode(["r"])
    sched.schedule_decode(["r"])
    assert len(st.loads) == 1 and st.loads[0][3].tolist() == [40]
    st.score_rows_key = st.key               # as the backend caches its score-row indices
    sched.finish_request("r")
    assert st.key is None and st.score_rows_key is None
    sched.add_request("x", 1, 20)            # takes the blocks "r" gave back
    sched.add_request("r", 2, 100)
    sched.schedule_decode(["r"])             # same id tuple as the resident rows
    assert len(st.loads) == 2
    key, tables, n_blocks, kv_lens, seq_lens = st.loads[1]
    assert key == ("r",) and kv_lens.tolist() == [100] and seq_lens.tolist() == [100]
    assert tables[0, :n_blocks[0]].tolist() == sched.get_block_table("r")[:int(n_blocks[0])]
    assert not set(tables[0, :n_blocks[0]].tolist()) & set(sched.get_block_table("x"))


def test_launcher_signature_is_view_invariant():
    """The per-step decode launcher is built from [T, H, D] views and matched against the
    [T, H*D] tensors the next layers pass: both must give the same signature."""
    from vllm_paged_compression_amd.ops import DecodeLayerLauncher
    q = torch.zeros(5, 4 * 16)
    k = torch.zeros(5, 2 * 16)
    kv = torch.zeros(2, 7, 16 * 2 * 16)
    a = DecodeLayerLauncher.signature(q, k, k, kv)
    b = DecodeLayerLauncher.signature(q.view(5, 4, 16), k.view(5, 2, 16), k.view(5, 2, 16), kv)
    assert a == b
    assert a != DecodeLayerLauncher.signature(q[:4], k[:4], k[:4], kv)


def test_budget_not_a_multiple_of_the_block_size():
    """A budget that is not a whole number of blocks must not reach the kernels: with block 16,
    budget 200 and a 201-token prompt the pruned-length formula gives 209 > 201 kept tokens
    (vllm/attention/kvcache_prunner.py:284-326; the reference then fails in a copy_)."""
    from vllm_paged_compression_amd.config import PagedEvictConfig
    from vllm_paged_compression_amd.page_evict_kv_util import get_num_required_blocks_after_prune_promt
    from vllm_paged_compression_amd.pruner import KVCachePruner
    cfg = PagedEvictConfig(True, "budget", "streamingLLM", 1000, -1)
    assert cfg.cache_budget == 992                      # rounded down, like the < 128 clamp
    assert PagedEvictConfig(True, "budget", "global", 200, -1).cache_budget == 192
    with pytest.raises(ValueError, match="multiple of the block size"):
        KVCachePruner("budget", "streamingLLM", 16, 200, 1)
    with pytest.raises(ValueError, match="multiple of the block size"):
        KVCachePruner("budget", "global", 32, 784, 1)    # a multiple of 16 but not of 32
    # kept never exceeds the prompt, whatever the budget object says
    raw = PagedEvictConfig(True, "budget", "streamingLLM", 192, -1)
    raw.cache_budget = 200                              # as a foreign config object could carry it
    for L in range(180, 260):
        nblk, kept = get_num_required_blocks_after_prune_promt(L, raw, 16)
        assert kept <= L and nblk == -(-kept // 16)
    pr = KVCachePruner("budget", "streamingLLM", 16, 192, 1)
    for L in range(1, 400):
        assert pr.get_pruned_length(L) <= L


def test_kv_len_adjustment_helpers(oracle):
    """page_evict_kv_util.adjust_seq_kv_len(s) = the model runner's pre-attention adjustment
    (vllm/worker/model_runner.py:538-547): one block less on the steps that evict, nothing for
    inverse_key_l2 or with eviction off."""
    import numpy as np
    from vllm_paged_compression_amd.config import PagedEvictConfig
    from vllm_paged_compression_amd.page_evict_kv_util import (
        adjust_seq_kv_len, adjust_seq_kv_lens, budget_decode_trigger, evicted_tokens,
        percentage_decode_trigger)
    bs = 16
    Ls = np.arange(100, 400)
    kv = np.minimum(Ls, 150)
    for method in ("streamingLLM", "streamingLLM-1", "global", "local"):
        cfg = PagedEvictConfig(True, "budget", method, 128, 3)
        for L, k in zip(Ls.tolist(), kv.tolist()):
            trig = oracle.decode_trigger(L, 128, bs)
            assert budget_decode_trigger(cfg, L, bs) == trig
            assert evicted_tokens(cfg, L, k, bs) == (bs if trig else 0)
            assert adjust_seq_kv_len(cfg, L, k, bs) == k - (bs if trig else 0)
        want = [adjust_seq_kv_len(cfg, L, k, bs) for L, k in zip(Ls.tolist(), kv.tolist())]
        assert adjust_seq_kv_lens(cfg, Ls, kv, bs).tolist() == want
    cfg = PagedEvictConfig(True, "budget", "inverse_key_l2", 128, -1)
    assert not any(budget_decode_trigger(cfg, int(L), bs) for L in Ls)
    assert adjust_seq_kv_lens(cfg, Ls, kv, bs).tolist() == kv.tolist()
    assert adjust_seq_kv_lens(None, Ls, kv, bs).tolist() == kv.tolist()
    assert adjust_seq_kv_len(None, 160, 150, bs) == 150
    pct = PagedEvictConfig(True, "percentage", "streamingLLM", 0, -1, evict_freq=2, num_block_merge=3)
    want = [k - (2 * bs if percentage_decode_trigger(pct, L, k, bs) else 0)
            for L, k in zip(Ls.tolist(), kv.tolist())]
    assert adjust_seq_kv_lens(pct, Ls, kv, bs).tolist() == want
    assert [adjust_seq_kv_len(pct, L, k, bs) for L, k in zip(Ls.tolist(), kv.tolist())] == want
    assert any(w != k for w, k in zip(want, kv.tolist()))
