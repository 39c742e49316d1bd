"""Golden decode-time eviction decisions, produced by running the *reference's own methods*.

The backend module ``vllm/attention/backends/torch_cuda_sdpa.py`` cannot be imported in the build
container (its package imports need ``msgspec``), but the four methods that take the decode
decision only touch torch / numpy / random and a handful of attributes of ``self``:

    TorchCUDASDPABackendImpl._prune_decode_reqs_streamingLLM   (:1375-1433)
    TorchCUDASDPABackendImpl._prune_decode_reqs_inverseKeyL2   (:1435-1475)
    TorchCUDASDPABackendImpl._prune_decode_reqs_topk           (:1478-1560)
    TorchCUDASDPABackendImpl._create_updated_block_tables      (:1562-1586)

``host_logic()`` does the same for three pieces of host arithmetic (admission size of a pruned
prompt, slot mapping, kv-length adjustment on eviction steps).

This script reads the file from the reference checkout at generation time, compiles exactly those
function definitions (located with ``ast``; nothing of the file is stored), binds them to a
stand-in ``self`` that carries the reference's real ``KVCachePruner`` and ``BlockL2NormManager``
(both importable as they are), and drives them through multi-step decode traces the way the engine
does (layer 0's victims are committed to the tables, ``llm_engine.py:1179-1193``).  Only data is
written: per step the inputs (logical lengths, block tables) and what the methods returned
(victims, the compacted table tensor, zeroed slots, the score lists afterwards).

Run:  python tests/golden/make_golden_decode.py  [--reference /root/reference]
"""
import argparse
import ast
import importlib.util
import json
import os
import random
import textwrap
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
METHODS = ("_prune_decode_reqs_streamingLLM", "_prune_decode_reqs_inverseKeyL2",
           "_prune_decode_reqs_topk", "_create_updated_block_tables")


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def reference_methods(path):
    """The four method definitions of the impl class, compiled from the reference file."""
    src = open(path).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TorchCUDASDPABackendImpl")
    ns = {"torch": torch, "np": np, "random": random, "TorchCUDASDPAMetadata": object}
    for node in cls.body:
        if isinstance(node, ast.FunctionDef) and node.name in METHODS:
            exec(compile(textwrap.dedent(ast.get_source_segment(src, node)), path, "exec"), ns)
    return {m: ns[m] for m in METHODS}


def reference_function(path, name, cls_name=None, extra=None):
    """One function (or method of ``cls_name``) compiled from a reference file."""
    src = open(path).read()
    body = ast.parse(src).body
    if cls_name is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls_name).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = dict(extra or {})
    exec(compile(textwrap.dedent(ast.get_source_segment(src, node)), path, "exec"), ns)
    return ns[name]


def host_logic(ref):
    """Host-side integer arithmetic of the path, from the reference's own functions:
    get_num_required_blocks_after_prune_promt (vllm/core/page_evict_kv_util.py:6-36),
    TorchCUDASDPAMetadataBuilder._compute_slot_mapping (torch_cuda_sdpa.py:565-598) and the
    kv-length adjustment of ModelInputForGPUBuilder._compute_lens (vllm/worker/model_runner.py:493-558)."""
    import math
    from typing import List
    out = {}
    req = reference_function(f"{ref}/vllm/core/page_evict_kv_util.py",
                             "get_num_required_blocks_after_prune_promt",
                             extra={"math": math, "cdiv": lambda a, b: -(a // -b), "PagedEvictConfig": object})
    rows = []
    for bs in (8, 16, 32):
        for budget in (3 * bs, 4 * bs, 128, 512, 768, 4096):
            if budget < 3 * bs or budget % bs:
                continue
            for method in ("streamingLLM", "streamingLLM-1", "inverse_key_l2", "local", "global"):
                cfg = SimpleNamespace(cache_budget=budget, initial_blocks=1, evict_method=method,
                                      cache_prune_type="budget")
                for L in (1, bs, budget - 1, budget, budget + 1, budget + bs, budget + bs + 3, 1024, 1030,
                          2 * budget + 5, 8192, 32768):
                    nblk, ntok = req(L, cfg, bs)
                    rows.append([bs, budget, method, L, int(nblk), int(ntok)])
    out["required_blocks"] = rows

    slot = reference_function(f"{ref}/vllm/attention/backends/torch_cuda_sdpa.py", "_compute_slot_mapping",
                              "TorchCUDASDPAMetadataBuilder", extra={"List": List, "PAD_SLOT_ID": -1})
    me = SimpleNamespace(sliding_window=None)
    rng = random.Random(7)
    cases = []
    for bs in (8, 16):
        for kept in (1, bs - 1, bs, bs + 1, 5 * bs, 5 * bs + 3, 48 * bs, 48 * bs + 7):
            table = rng.sample(range(1000), -(-kept // bs))
            sm = []
            slot(me, False, True, sm, 4, kept, 0, bs, {4: table})
            cases.append({"prompt": True, "bs": bs, "query_len": kept, "context_len": 0, "table": table,
                          "slots": sm})
        for ctx in (0, 1, bs - 1, bs, bs + 5, 47 * bs + 15, 48 * bs):
            table = rng.sample(range(1000), ctx // bs + 1)
            sm = []
            slot(me, False, False, sm, 9, 1, ctx, bs, {9: table})
            cases.append({"prompt": False, "bs": bs, "query_len": 1, "context_len": ctx, "table": table,
                          "slots": sm})
    sm = []
    slot(me, True, True, sm, 0, 5, 0, 16, {0: [3]})
    cases.append({"prompt": True, "profile": True, "bs": 16, "query_len": 5, "context_len": 0, "table": [3],
                  "slots": sm})
    out["slot_mapping"] = cases

    lens_fn = reference_function(f"{ref}/vllm/worker/model_runner.py", "_compute_lens", "ModelInputForGPUBuilder",
                                 extra={"InterDataForSeqGroup": object, "SequenceGroupMetadata": object})
    rows = []
    for method in ("streamingLLM", "streamingLLM-1", "global", "local", "inverse_key_l2"):
        for bs, budget in ((16, 64), (16, 768), (8, 48)):
            cfg = SimpleNamespace(evict_method=method, cache_budget=budget)
            runner = SimpleNamespace(scheduler_config=SimpleNamespace(is_multi_step=False),
                                     model_config=SimpleNamespace(is_encoder_decoder=False),
                                     cache_config=SimpleNamespace(paged_evict_config=cfg, block_size=bs))
            for seq_len in (budget - 1, budget, budget + 1, budget + bs - 1, budget + bs, budget + bs + 1,
                            budget + 2 * bs, budget + 5 * bs, 2 * budget + 3):
                kv_before = min(seq_len, budget + bs)
                seq_data = SimpleNamespace(get_len=lambda n=seq_len: n,
                                           get_num_computed_tokens=lambda n=seq_len: n - 1,
                                           get_token_ids=lambda n=seq_len: list(range(n)),
                                           mrope_position_delta=None)
                sgm = SimpleNamespace(seq_data={5: seq_data}, token_chunk_size=1, token_type_ids=None)
                inter = SimpleNamespace(seq_ids=[5], is_prompt=False, seq_lens=[0], orig_seq_lens=[0],
                                        context_lens=[0], input_tokens=[[]], input_positions=[[]],
                                        token_types=[[]], query_lens=[0], seq_kv_lens={5: kv_before},
                                        mrope_input_positions=None, n_seqs=1)
                lens_fn(SimpleNamespace(runner=runner), inter, 0, sgm)
                rows.append([method, bs, budget, seq_len, kv_before, inter.seq_kv_lens[5], inter.query_lens[0],
                             inter.context_lens[0], inter.input_positions[0][0]])
    out["kv_len_adjust"] = rows

    # admission: SelfAttnBlockSpaceManager.can_allocate (vllm/core/block_manager.py:113-156) with the
    # names it takes from its module given as the reference's own function / plain stand-ins
    import enum

    class AllocStatus(enum.Enum):
        OK = enum.auto()
        LATER = enum.auto()
        NEVER = enum.auto()

    class PlainTable:
        @staticmethod
        def get_num_required_blocks(token_ids, block_size, num_lookahead_slots=0):
            return -((len(token_ids) + num_lookahead_slots) // -block_size)     # block_table.py:70-92

    can = reference_function(
        f"{ref}/vllm/core/block_manager.py", "can_allocate", "SelfAttnBlockSpaceManager",
        extra={"SequenceGroup": object, "AllocStatus": AllocStatus, "BlockTable": PlainTable,
               "SequenceStatus": SimpleNamespace(WAITING=0), "Device": SimpleNamespace(GPU=0),
               "check_no_caching_or_swa_for_blockmgr_encdec": lambda *a: None,
               "get_num_required_blocks_after_prune_promt": req})
    rows = []
    for bs, budget in ((16, 64), (16, 768), (8, 48)):
        for evict in (False, True):
            cfg = SimpleNamespace(cache_budget=budget, initial_blocks=1, evict_method="global",
                                  cache_prune_type="budget") if evict else None
            for total, watermark in ((64, 0), (64, 1), (100, 1), (1000, 10)):
                for free in (0, 3, 5, 6, 49, 50, 63, total):
                    for L in (1, budget - 1, budget, budget + 1, 100, 1024, 1030, 16 * total, 16 * total + 1):
                        mgr = SimpleNamespace(
                            paged_evict_config=cfg, block_size=bs, max_block_sliding_window=None,
                            num_total_gpu_blocks=total, watermark_blocks=watermark,
                            block_allocator=SimpleNamespace(get_num_free_blocks=lambda device, f=free: f))
                        seq = SimpleNamespace(get_token_ids=lambda n=L: [0] * n)
                        grp = SimpleNamespace(get_seqs=lambda status, s=seq: [s], is_encoder_decoder=lambda: False)
                        rows.append([bs, budget, int(evict), total, watermark, free, L, can(mgr, grp).name])
    out["can_allocate"] = rows
    with open(os.path.join(HERE, "host_logic.json"), "w") as f:
        json.dump(out, f)
    print(f"host_logic: {len(out['required_blocks'])} pruned-size rows, {len(out['slot_mapping'])} slot "
          f"mappings, {len(out['kv_len_adjust'])} kv-length rows, {len(out['can_allocate'])} admission rows")


def block_table_traces(ref):
    """Block bookkeeping from the reference's own classes: ``BlockTable``
    (vllm/core/block/block_table.py) on a ``NaiveBlockAllocator`` (vllm/core/block/naive_block.py),
    loaded from their files with the three names they take from ``vllm.utils`` (Device, cdiv,
    chunk_list) and ``PagedEvictConfig`` provided by stub modules, and a four-method adapter that
    drops the ``device`` argument the device-aware allocator front end would dispatch on.  Every
    trace shares one allocator between several sequences: pruned prompt allocation, one token per
    step, the victim block freed after the step where the trigger fires (llm_engine.py:1179-1193),
    a sequence finishing and a new one admitted.  Recorded after every step: block ids, logical
    length, cached length, free blocks."""
    import enum
    import sys

    class Device(enum.Enum):
        GPU = enum.auto()
        CPU = enum.auto()

    def chunk_list(lst, n):
        for i in range(0, len(lst), n):
            yield lst[i:i + n]

    utils = types.ModuleType("vllm.utils")
    utils.Device, utils.cdiv, utils.chunk_list = Device, (lambda a, b: -(a // -b)), chunk_list
    cfgm = types.ModuleType("vllm.config")
    cfgm.PagedEvictConfig = object
    names = ("vllm", "vllm.core", "vllm.core.block", "vllm.utils", "vllm.config",
             "vllm.core.block.interfaces", "vllm.core.block.common", "vllm.core.block.naive_block",
             "vllm.core.page_evict_kv_util", "vllm.core.block.block_table")
    saved = {k: sys.modules.get(k) for k in names}

    def load_as(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    try:
        for n in ("vllm", "vllm.core", "vllm.core.block"):
            sys.modules[n] = types.ModuleType(n)
            sys.modules[n].__path__ = []
        sys.modules["vllm.utils"], sys.modules["vllm.config"] = utils, cfgm
        load_as("vllm.core.block.interfaces", f"{ref}/vllm/core/block/interfaces.py")
        load_as("vllm.core.block.common", f"{ref}/vllm/core/block/common.py")
        nbm = load_as("vllm.core.block.naive_block", f"{ref}/vllm/core/block/naive_block.py")
        load_as("vllm.core.page_evict_kv_util", f"{ref}/vllm/core/page_evict_kv_util.py")
        btm = load_as("vllm.core.block.block_table", f"{ref}/vllm/core/block/block_table.py")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v

    class OnGpu:
        def __init__(self, a):
            self.a = a

        def allocate_mutable_block(self, prev_block, device, extra_hash=None):
            return self.a.allocate_mutable_block(prev_block, extra_hash=extra_hash)

        def allocate_immutable_block(self, prev_block, token_ids, device, extra_hash=None):
            return self.a.allocate_immutable_block(prev_block, token_ids, extra_hash=extra_hash)

        def allocate_immutable_blocks(self, prev_block, block_token_ids, device, extra_hash=None):
            return self.a.allocate_immutable_blocks(prev_block, block_token_ids, extra_hash=extra_hash)

        def free(self, block):
            self.a.free(block)

    traces = []
    for ti, (bs, budget, nblocks, prompts, late, policy) in enumerate([
            (16, 64, 48, [100, 61, 130], 75, "second"), (16, 64, 48, [100, 61, 130], 75, "middle"),
            (8, 48, 64, [33, 200, 48, 49], 90, "middle"), (16, 128, 64, [1024, 130, 128], 300, "second"),
            (16, 64, 40, [64, 65, 80], 17, None)]):
        cfg = None if policy is None else SimpleNamespace(
            cache_budget=budget, initial_blocks=1, evict_method="global", cache_prune_type="budget")
        alloc = nbm.NaiveBlockAllocator(create_block=nbm.NaiveBlock, num_blocks=nblocks, block_size=bs)
        rng = random.Random(50 + ti)
        live = {}

        def admit(name, n):
            t = btm.BlockTable(bs, OnGpu(alloc), paged_evict_config=cfg)
            t.allocate(list(range(n)))
            live[name] = t

        def snap():
            return {k: [list(t.physical_block_ids), t._num_full_slots, t.get_seq_kv_len()]
                    for k, t in live.items()}

        for i, n in enumerate(prompts):
            admit(f"s{i}", n)
        tr = {"bs": bs, "budget": budget, "num_blocks": nblocks, "prompts": prompts, "late_prompt": late,
              "evict": policy is not None, "after_prefill": snap(), "free_after_prefill": alloc.get_num_free_blocks(),
              "steps": []}
        for step in range(60):
            ev = {}
            if step == 20:
                live.pop("s1").free()
                ev["finished"] = "s1"
            if step == 23:
                admit("late", late)
                ev["admitted"] = "late"
            for t in live.values():
                t.append_token_ids([7])
            rmv = {}
            if cfg is not None:
                for k, t in live.items():
                    L = t._num_full_slots
                    if L > budget and L % bs == 0:
                        n = len(t.physical_block_ids)
                        rmv[k] = 1 if policy == "second" else rng.randint(1, n - 2)
                for k, idx in rmv.items():
                    live[k].free_prunned_blocks(idx)
            ev.update(rmv=rmv, state=snap(), free=alloc.get_num_free_blocks())
            tr["steps"].append(ev)
        traces.append(tr)
    with open(os.path.join(HERE, "block_table_traces.json"), "w") as f:
        json.dump(traces, f)
    print(f"block_table_traces: {len(traces)} traces, "
          f"{sum(len(s['rmv']) for t in traces for s in t['steps'])} evictions")


def prefill_driver(ref, kp, bm):
    """The batch driver of the prompt prune, TorchCUDASDPABackendImpl._prune_prompts_keys_and_values_new
    (torch_cuda_sdpa.py:1222-1296), compiled from the reference file and run on CPU tensors.  The
    method spreads the requests over CUDA side streams (a scheduling detail without effect on the
    result); the ``torch`` it sees here forwards everything to the real module except
    ``torch.cuda.stream`` / ``current_stream``, which do nothing.  Written per case: K/V of the
    batch, the concatenated pruned K/V the method returned and the per-request block-score lists
    it left in the BlockL2NormManager."""
    import contextlib

    class TorchOnCpu:
        cuda = SimpleNamespace(stream=lambda s: contextlib.nullcontext(),
                               current_stream=lambda: SimpleNamespace(wait_stream=lambda s: None))

        def __getattr__(self, name):
            return getattr(torch, name)

    drv = reference_function(f"{ref}/vllm/attention/backends/torch_cuda_sdpa.py",
                             "_prune_prompts_keys_and_values_new", "TorchCUDASDPABackendImpl",
                             extra={"torch": TorchOnCpu(), "TorchCUDASDPAMetadata": object})
    bs, H, D, budget = 16, 2, 16, 64
    arrays, cases = {}, []
    for ci, (method, formula, lens) in enumerate([
            ("streamingLLM", None, [100, 40, 130]), ("global", "value_l2", [100, 40, 130]),
            ("global", "key_l2", [97, 64, 65]), ("local", "value_l2_plus_key_l2", [200, 81]),
            ("global", "inverse_key_l2", [128, 129]), ("inverse_key_l2", None, [100, 70]),
            ("global", "value_l2", [30, 64])]):
        g = torch.Generator().manual_seed(600 + ci)
        T = sum(lens)
        # token-level prunes rank (token, head) norms: f32 inputs keep them free of ties (the order
        # torch.topk gives equal scores is pinned by nothing); the slice-only method runs in bf16
        dt = torch.bfloat16 if method == "streamingLLM" else torch.float32
        key = (torch.rand(T, H, D, generator=g) * 2 - 1).to(dt)
        value = (torch.rand(T, H, D, generator=g) * 2 - 1).to(dt)
        pruner = kp.KVCachePruner("budget", method, bs, budget, 1)
        if formula is not None:
            pruner.sub_evict_method = formula
        mgr = bm.BlockL2NormManager()
        mapping = {f"r{i}": {i: [0] * 8} for i in range(len(lens))}
        mgr.update_reqs(mapping)
        me = SimpleNamespace(streams=[None, None], kv_cache_pruner=pruner, num_kv_heads=H, head_size=D,
                             paged_evict_config=SimpleNamespace(evict_method=method),
                             cache_config=SimpleNamespace(block_size=bs), enable_random_evict=False,
                             block_l2norm_mgr=mgr)
        pk, pv = drv(me, key, value, SimpleNamespace(seq_lens=list(lens)), False)
        scores = None
        if method in ("local", "global") and pk is not key:
            scores = [[float(x) for x in mgr.reqid_mapping_block_l2norms[f"r{i}"][i]] for i in range(len(lens))]
        arrays[f"c{ci}_k"], arrays[f"c{ci}_v"] = raw(key), raw(value)
        arrays[f"c{ci}_pk"], arrays[f"c{ci}_pv"] = raw(pk), raw(pv)
        cases.append({"method": method, "formula": formula, "lens": lens, "returned_input": pk is key,
                      "dtype": 2 if dt == torch.bfloat16 else 0, "scores": scores})
    np.savez_compressed(os.path.join(HERE, "prefill_driver.npz"), **arrays)
    with open(os.path.join(HERE, "prefill_driver.json"), "w") as f:
        json.dump({"block_size": bs, "num_kv_heads": H, "head_size": D, "budget": budget, "cases": cases}, f)
    print(f"prefill_driver: {len(cases)} cases")


def sdpa_inputs(seed, lens_q, lens_kv, Hq, Hkv, D):
    """bf16 inputs of an sdpa_forward case, regenerated from the seed by the tests."""
    g = torch.Generator().manual_seed(seed)
    sc = D ** -0.5
    q = ((torch.rand(sum(lens_q), Hq, D, generator=g) * 2 - 1) * 3 * sc).to(torch.bfloat16)
    k = ((torch.rand(sum(lens_kv), Hkv, D, generator=g) * 2 - 1) * 3 * sc).to(torch.bfloat16)
    v = (torch.rand(sum(lens_kv), Hkv, D, generator=g) * 2 - 1).to(torch.bfloat16)
    return q, k, v


def sdpa_forward(ref):
    """Prefill attention as the reference composes it: TorchCUDASDPABackendImpl._run_sdpa_forward
    (torch_cuda_sdpa.py:1069-1127) compiled from the reference file and run on CPU in f32 over
    bf16-representable inputs - grouped heads through repeat_interleave, one
    scaled_dot_product_attention(is_causal=True) per request, kv offsets advancing by the PRUNED
    lengths when the prompt was pruned before attention (top-left aligned mask for L_q != L_kv),
    or by the query lengths with disable_evict_prefill.  Stored: the f32 outputs (inputs come
    from sdpa_inputs(seed))."""
    from torch.nn.functional import scaled_dot_product_attention
    fn = reference_function(f"{ref}/vllm/attention/backends/torch_cuda_sdpa.py", "_run_sdpa_forward",
                            "TorchCUDASDPABackendImpl",
                            extra={"torch": torch, "TorchCUDASDPAMetadata": object,
                                   "AttentionType": SimpleNamespace(DECODER="decoder"),
                                   "scaled_dot_product_attention": scaled_dot_product_attention})
    arrays, cases = {}, []
    for ci, (Hq, Hkv, D, lens_q, lens_kv, disable) in enumerate([
            (4, 2, 64, [90, 24, 50], [58, 24, 50], False), (4, 2, 64, [90, 24, 50], [90, 24, 50], True),
            (8, 1, 128, [20, 70], [20, 38], False), (4, 4, 64, [1, 17, 64], [1, 17, 64], True)]):
        q, k, v = sdpa_inputs(900 + ci, lens_q, lens_kv, Hq, Hkv, D)
        me = SimpleNamespace(num_heads=Hq, num_kv_heads=Hkv, num_queries_per_kv=Hq // Hkv,
                             alibi_slopes=None, sliding_window=None, enabled_paged_eviction=True,
                             paged_evict_config=SimpleNamespace(disable_evict_prefill=disable),
                             scale=D ** -0.5)
        md = SimpleNamespace(get_attn_bias=lambda t: None, set_attn_bias=lambda m, t: None,
                             get_seq_lens=lambda t, a=lens_q, b=lens_kv: (list(a), list(b)),
                             seq_lens=list(lens_q))
        out = torch.zeros(sum(lens_q), Hq, D, dtype=torch.float32)
        fn(me, out, q.float(), k.float(), v.float(), md)
        arrays[f"c{ci}_out"] = out.numpy().copy()
        cases.append({"seed": 900 + ci, "Hq": Hq, "Hkv": Hkv, "D": D, "lens_q": lens_q, "lens_kv": lens_kv,
                      "disable_evict_prefill": disable})
    np.savez_compressed(os.path.join(HERE, "sdpa_forward.npz"), **arrays)
    with open(os.path.join(HERE, "sdpa_forward.json"), "w") as f:
        json.dump(cases, f)
    print(f"sdpa_forward: {len(cases)} cases")


def bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def raw(t):
    return bits(t) if t.dtype == torch.bfloat16 else t.contiguous().numpy().copy()


def pad_tables(tables, width):
    out = torch.zeros((len(tables), width), dtype=torch.int32)
    for i, t in enumerate(tables):
        out[i, :len(t)] = torch.tensor(t, dtype=torch.int32)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ref = ap.parse_args().reference
    fns = reference_methods(f"{ref}/vllm/attention/backends/torch_cuda_sdpa.py")
    kp = load("ref_kvcache_prunner", f"{ref}/vllm/attention/kvcache_prunner.py")
    bm = load("ref_block_l2norm_manager", f"{ref}/vllm/attention/block_l2norm_manager.py")

    bs, Hkv, D, nb = 16, 2, 32, 40
    budget, topk, n_steps = 64, 3, 40
    g = torch.Generator().manual_seed(4242)
    kv_cache = (torch.rand(2, nb, bs * Hkv * D, generator=g) * 2 - 1).to(torch.bfloat16)
    arrays = {"kv_cache": bits(kv_cache)}
    meta = {"block_size": bs, "num_kv_heads": Hkv, "head_size": D, "num_blocks": nb, "budget": budget,
            "topk_blocks": topk, "traces": []}

    cases = [("streamingLLM", None, False), ("streamingLLM-1", None, False),
             ("local", "value_l2", False), ("global", "value_l2", False),
             ("global", "key_l2", False), ("global", "value_l2_plus_key_l2", False),
             ("local", "inverse_key_l2", False),
             ("global", "value_l2", True), ("local", "value_l2", True),
             ("inverse_key_l2", None, True)]
    for ci, (method, formula, use_random) in enumerate(cases):
        pruner = kp.KVCachePruner("budget", method, bs, budget, 1)
        if formula is not None:
            pruner.sub_evict_method = formula
        me = SimpleNamespace(
            block_l2norm_mgr=bm.BlockL2NormManager(), kv_cache_pruner=pruner,
            paged_evict_config=SimpleNamespace(cache_budget=budget, evict_method=method, topk_blocks=topk),
            cache_config=SimpleNamespace(block_size=bs), num_kv_heads=Hkv, head_size=D,
            enable_random_evict=use_random)
        me._create_updated_block_tables = types.MethodType(fns["_create_updated_block_tables"], me)
        decide = types.MethodType(fns[{"streamingLLM": "_prune_decode_reqs_streamingLLM",
                                       "streamingLLM-1": "_prune_decode_reqs_streamingLLM",
                                       "inverse_key_l2": "_prune_decode_reqs_inverseKeyL2"}.get(
                                           method, "_prune_decode_reqs_topk")], me)
        random.seed(1234)            # what the impl constructor does (:796)
        # three sequences (two requests, one of them with two sequences) at staggered lengths;
        # every table holds the blocks of its physical tokens, free blocks are handed out in order
        rng = random.Random(100 + ci)
        free = list(range(nb))
        rng.shuffle(free)
        seqs = [("reqA", 0, 61), ("reqB", 3, 50), ("reqB", 4, 77)]      # (request, sequence id, logical length)
        logical = [L for _, _, L in seqs]
        physical = [min(L, budget + L % bs) if L > budget else L for L in logical]
        tables = [[free.pop() for _ in range(-(-p // bs))] for p in physical]
        init_scores = [[round(rng.uniform(5.0, 60.0), 3) for _ in t[:-1]] for t in tables]
        mapping = {}
        for (rid, sid, _), t in zip(seqs, tables):
            mapping.setdefault(rid, {})[sid] = t
        me.block_l2norm_mgr.update_reqs(mapping)
        if method in ("local", "global"):
            # the prefill driver leaves one score per full prompt block (:1283-1296)
            for i, sc in enumerate(init_scores):
                me.block_l2norm_mgr.update_block_l2norms(i, list(sc))
        trace = {"method": method, "formula": formula, "random": use_random,
                 "seqs": [[r, s] for r, s, _ in seqs], "init_scores": init_scores, "steps": []}
        for _ in range(n_steps):
            # the step's new token: a block is appended when the last one is full
            for i in range(len(seqs)):
                if physical[i] % bs == 0:
                    tables[i].append(free.pop())
                physical[i] += 1
                logical[i] += 1
            me.block_l2norm_mgr.update_reqs(mapping)
            width = max(len(t) for t in tables)
            md = SimpleNamespace(seq_lens=list(logical), reqid_mapping_blocktables=mapping,
                                 decode_metadata=SimpleNamespace(block_tables=pad_tables(tables, width)))
            step = {"seq_lens": list(logical), "tables": [list(t) for t in tables], "width": width}
            if method in ("streamingLLM", "streamingLLM-1"):
                n_zero, zero_map, pruned, upd, rmv = decide(kv_cache, md)
                step["zero_slots"] = [] if zero_map is None else zero_map.tolist()
            elif method == "inverse_key_l2":
                n_zero, zero_map, pruned = decide(kv_cache, md)
                step["zero_slots"] = [] if zero_map is None else zero_map.tolist()
                upd, rmv = None, {}
            else:
                upd, rmv, pruned = decide(kv_cache, md)
            step["is_pruned_decode"] = bool(pruned)
            step["rmv"] = {str(i): [r, int(b)] for i, (r, b) in rmv.items()}
            step["updated_tables"] = None if upd is None else upd.tolist()
            if method in ("local", "global") and not use_random:
                step["scores_after"] = [
                    [float(x) for x in me.block_l2norm_mgr.reqid_mapping_block_l2norms[r][s]]
                    for r, s, _ in seqs]
            trace["steps"].append(step)
            # engine commit of the layer-0 decision: the victim block leaves the table
            for i, (_, b) in rmv.items():
                free.insert(0, tables[i].pop(b))
                physical[i] -= bs
        meta["traces"].append(trace)

    np.savez_compressed(os.path.join(HERE, "decode_decisions.npz"), **arrays)
    with open(os.path.join(HERE, "decode_decisions.json"), "w") as f:
        json.dump(meta, f)
    n_ev = sum(len(s["rmv"]) for t in meta["traces"] for s in t["steps"])
    print(f"decode_decisions: {len(meta['traces'])} traces, {n_ev} evictions")
    host_logic(ref)
    block_table_traces(ref)
    prefill_driver(ref, kp, bm)
    sdpa_forward(ref)


if __name__ == "__main__":
    main()
