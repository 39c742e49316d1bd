"""Multi-rank behaviour of the path on CPU (gloo, world size 2): head sharding, the
o_proj-output all-reduce and the agreement of data-dependent eviction victims.  The
per-rank attention itself is stood in for by the oracle here (tests may use it); the GPU
kernels are rank-local and covered by the single-GPU parity tests."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pe_oracle as oracle
        from vllm_paged_compression_amd import parallel
        Hq, Hkv, D, bs, nb = 8, 2, 64, 16, 12
        sh = parallel.shard_heads(Hq, Hkv, world, rank)
        assert sh.num_heads == Hq // world and sh.num_kv_heads == 1
        g = torch.Generator().manual_seed(0)   # same data on every rank
        kc = (torch.rand(nb, Hkv, D // 8, bs, 8, generator=g) - 0.5).to(torch.bfloat16)
        vc = (torch.rand(nb, Hkv, D, bs, generator=g) - 0.5).to(torch.bfloat16)
        q_all = (torch.rand(3, Hq, D, generator=g) - 0.5).to(torch.bfloat16)
        bt = torch.randint(0, nb, (3, 5), generator=g, dtype=torch.int32)
        lens = torch.tensor([70, 33, 80], dtype=torch.int32)
        wo = torch.rand(Hq * D, 16, generator=g) - 0.5      # row-parallel o_proj weight
        full = oracle.paged_attention(q_all, kc, vc, bt, lens, D ** -0.5)
        # rank-local attention over this rank's heads only
        qs = q_all[:, sh.q_head_start:sh.q_head_start + sh.num_heads].contiguous()
        ks = kc[:, sh.kv_head_start:sh.kv_head_start + sh.num_kv_heads].contiguous()
        vs = vc[:, sh.kv_head_start:sh.kv_head_start + sh.num_kv_heads].contiguous()
        local = oracle.paged_attention(qs, ks, vs, bt, lens, D ** -0.5)
        assert torch.equal(local, full[:, sh.q_head_start:sh.q_head_start + sh.num_heads])
        # o_proj partial products summed over ranks == full projection
        part = local.float().reshape(3, -1) @ wo[sh.q_head_start * D:(sh.q_head_start + sh.num_heads) * D]
        parallel.all_reduce_sum(part)
        torch.testing.assert_close(part, full.float().reshape(3, -1) @ wo, atol=1e-4, rtol=1e-4)
        # value_l2 scores over ALL heads from head-sharded caches: per-token sums of this rank's
        # heads, added over the ranks, then the head mean / token sum with their roundings -
        # bit-identical to scoring the unsharded block, on every rank
        for b in range(nb):
            sums = oracle.block_token_norm_sums(vs[b], bs, sh.num_kv_heads, D, 1)
            parallel.all_reduce_sum(sums)
            s_tp = oracle.block_score_from_token_sums(sums, Hkv, torch.bfloat16)
            assert s_tp == oracle.block_score(vc[b], bs, Hkv, D, 1), (b, s_tp)
        # reference-compatible fallback: each rank scores with its own heads, rank 0's choice is used
        scores = [oracle.block_score(vs[b], bs, 1, D, 1) for b in range(6)]
        mine = oracle.argmin_first(scores, 1, 5)
        rmv = torch.tensor([mine, -1, mine], dtype=torch.int32)
        parallel.broadcast_victims(rmv, None, 0)
        gathered = [torch.zeros_like(rmv) for _ in range(world)]
        dist.all_gather(gathered, rmv)
        assert all(torch.equal(t, gathered[0]) for t in gathered)
        q.put((rank, "ok", mine, rmv.tolist()))
    except Exception as e:  # pragma: no cover
        q.put((rank, "fail", repr(e), None))
    finally:
        dist.destroy_process_group()


def test_two_rank_head_sharding_and_agreement():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == "ok" for r in res), res
    res.sort()
    assert res[0][3] == res[1][3]            # both ranks apply rank 0's victims
    assert res[0][3][0] == res[0][2]


def test_shard_heads_layouts():
    from vllm_paged_compression_amd.parallel import shard_heads
    s = shard_heads(64, 8, 8, 3)     # Llama-3-70B over 8 ranks
    assert (s.num_heads, s.num_kv_heads, s.q_head_start, s.kv_head_start) == (8, 1, 24, 3)
    s = shard_heads(32, 8, 2, 1)
    assert (s.num_heads, s.num_kv_heads, s.kv_head_start) == (16, 4, 4)
    s = shard_heads(32, 4, 8, 5)     # fewer KV heads than ranks: replicated
    assert (s.num_heads, s.num_kv_heads, s.kv_head_start) == (4, 1, 2)
    with pytest.raises(ValueError):
        shard_heads(30, 8, 4, 0)
