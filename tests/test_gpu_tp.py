"""Tensor-parallel eviction on the GPU: two ranks (two processes sharing the test GPU, gloo
collectives staged through the host) each own half of the heads and run the value_l2 ('global')
eviction path; the victims, the block tables and each rank's slice of the attention output must
be exactly what one rank holding all heads produces.  Covers the rank-invariant scores (token
sums all-reduced) of the backend's tensor-parallel path."""
import os
import socket
import sys
from types import SimpleNamespace

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

BS = 16
HQ, HKV, D, LAYERS = 8, 2, 128, 2
BUDGET, N_DECODE = 128, 40
PROMPTS = {"r0": 200, "r1": 130, "r2": 90}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(T, g):
    sc = D ** -0.5
    q = ((torch.rand(T, HQ, D, generator=g) * 2 - 1) * sc * 3).to(torch.bfloat16)
    k = ((torch.rand(T, HKV, D, generator=g) * 2 - 1) * sc * 3).to(torch.bfloat16)
    v = (torch.rand(T, HKV, D, generator=g) * 2 - 1).to(torch.bfloat16)
    return q, k, v


def run_trace(rank, world, group, mode="budget"):
    """One generation on this rank's head shard; returns ([victims per step], [output per step
    and layer, all on the CPU], final block tables).  mode "budget": value_l2 block eviction
    ("budget:<formula>": the same with another score formula of KVCachePruner.sub_evict_method);
    "percentage": the README-era type with the per-head value_l2 block merge at decode."""
    sys.path.insert(0, ROOT)
    from vllm_paged_compression_amd import parallel
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
    from vllm_paged_compression_amd.config import CacheConfigView, PagedEvictConfig
    dev = torch.device("cuda:0")
    sh = parallel.shard_heads(HQ, HKV, world, rank)
    from vllm_paged_compression_amd.page_evict_kv_util import adjust_seq_kv_len
    if mode == "percentage":
        cfg = PagedEvictConfig(True, "percentage", "streamingLLM", 0, -1, evict_freq=2,
                               num_block_merge=2, decode_evict_method="value_l2")
    else:
        cfg = PagedEvictConfig(True, "budget", "global", BUDGET, 3)
    ccfg = CacheConfigView(BS, cfg)
    nb = 96
    impls = [PagedEvictBackend.get_impl_cls()(
        sh.num_heads, D, D ** -0.5, sh.num_kv_heads, None, None, "auto", None, None,
        cache_config=ccfg, prefix=f"model.layers.{i}.self_attn.attn", tp_group=group,
        total_num_kv_heads=HKV) for i in range(LAYERS)]
    if ":" in mode:
        for impl in impls:
            impl.kv_cache_pruner.sub_evict_method = mode.split(":", 1)[1]
    shape = PagedEvictBackend.get_kv_cache_shape(nb, BS, sh.num_kv_heads, D)
    kv = [torch.zeros(shape, dtype=torch.bfloat16, device=dev) for _ in range(LAYERS)]
    sched = EvictionScheduler(EvictBlockSpaceManager(BS, nb, cfg, watermark=0.0))
    bm = sched.block_manager
    seq_of = {r: i for i, r in enumerate(PROMPTS)}
    for rid, n in PROMPTS.items():
        sched.add_request(rid, seq_of[rid], n)

    def build(inter, seq_lens, q_lens):
        runner = SimpleNamespace(device=dev, cache_config=ccfg, pin_memory=False)
        ib = SimpleNamespace(runner=runner, sliding_window=None, block_size=BS,
                             chunked_prefill_enabled=False, inter_data_list=inter)
        return PagedEvictBackend.make_metadata_builder(ib).build(seq_lens, q_lens, -1, len(seq_lens))

    def shard(q, k, v):
        qs = q[:, sh.q_head_start:sh.q_head_start + sh.num_heads].reshape(q.shape[0], -1)
        ks = k[:, sh.kv_head_start:sh.kv_head_start + sh.num_kv_heads].reshape(k.shape[0], -1)
        vs = v[:, sh.kv_head_start:sh.kv_head_start + sh.num_kv_heads].reshape(v.shape[0], -1)
        return qs.contiguous().to(dev), ks.contiguous().to(dev), vs.contiguous().to(dev)

    g = torch.Generator().manual_seed(77)   # the same full-head data on every rank
    victims, outputs = [], []
    lens = list(PROMPTS.values())
    inter = [SimpleNamespace(request_id=rid, is_prompt=True,
                             block_tables={seq_of[rid]: list(bm.get_block_table(seq_of[rid]))},
                             seq_kv_lens={seq_of[rid]: bm.get_seq_kv_len(seq_of[rid])},
                             seq_ids=[seq_of[rid]], input_tokens=[[0] * n], orig_seq_lens=[n],
                             seq_lens=[n], query_lens=[n], context_lens=[0])
             for rid, n in PROMPTS.items()]
    meta = build(inter, lens, lens)
    for li in range(LAYERS):
        q, k, v = shard(*_inputs(sum(lens), g))
        outputs.append(impls[li].forward(q, k, v, kv[li], meta).cpu())
    seq_len = dict(zip(PROMPTS, lens))
    for _ in range(N_DECODE):
        tables, kv_lens = sched.schedule_decode(list(PROMPTS))
        inter, logical = [], []
        for rid in PROMPTS:
            sid = seq_of[rid]
            seq_len[rid] += 1
            L = seq_len[rid]
            kvl = adjust_seq_kv_len(cfg, L, kv_lens[sid], BS)
            logical.append(L)
            inter.append(SimpleNamespace(request_id=rid, is_prompt=False,
                                         block_tables={sid: list(tables[sid])}, seq_kv_lens={sid: kvl},
                                         seq_ids=[sid], input_tokens=[[0]], orig_seq_lens=[L],
                                         seq_lens=[L], query_lens=[1], context_lens=[kv_lens[sid] - 1]))
        meta = build(inter, logical, [1] * 3)
        for li in range(LAYERS):
            q, k, v = shard(*_inputs(3, g))
            outputs.append(impls[li].forward(q, k, v, kv[li], meta).cpu())
        got = impls[0].get_rmv_block_idx()
        victims.append(dict(got))
        sched.commit(got)
    final_tables = [list(bm.get_block_table(seq_of[r])) for r in PROMPTS]
    return victims, outputs, final_tables


def _worker(rank, world, port, q, mode="budget"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        victims, outputs, tables = run_trace(rank, world, dist.group.WORLD, mode)
        q.put((rank, "ok", victims, [o.float().numpy() for o in outputs], tables))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc(), None, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["budget", "percentage", "budget:value_l2_plus_key_l2"])
def test_two_rank_value_l2_eviction_matches_single_rank(mode):
    import torch.multiprocessing as mp
    ref_victims, ref_out, ref_tables = run_trace(0, 1, None, mode)
    assert sum(len(v) for v in ref_victims) >= 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=110) for _ in procs)
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), [r[2] for r in res if r[1] != "ok"]
    hq = HQ // 2
    for rank, _, victims, outs, tables in res:
        assert victims == ref_victims, rank          # identical decisions, every step
        assert tables == ref_tables, rank
        for o, full in zip(outs, ref_out):
            want = full.float().view(full.shape[0], HQ, D)[:, rank * hq:(rank + 1) * hq]
            got = torch.from_numpy(o).view(full.shape[0], hq, D)
            assert torch.equal(got, want), rank      # per-head attention does not depend on the shard
