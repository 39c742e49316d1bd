#!/usr/bin/env python3
"""Benchmark of the paged-KV-eviction attention hot path on MI355X.

One "step" = one full generation of a batch of synthetic requests through the hot path
of every layer: prefill attention + prompt prune + KV write for the prompts, then
output_len decode steps of (eviction decision + block-table compaction + KV write + paged
decode attention), driven by the same host code a serving engine would run (scheduler,
metadata builder, backend impl).  The model's GEMMs/MLP are outside the path and not run.

Default workload (BASELINE.json configs[1]): Llama-3-8B attention geometry (32 query heads,
8 KV heads, head 128, 32 layers, bf16), block size 16, input 1024 -> output 100, 64 requests
(the reference's sweep scripts use 64), 25 % eviction => cache budget 768, streamingLLM
prompt prune, value_l2 ("global") block eviction at decode.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N   (tensor parallel:
      heads sharded over the ranks, N x 64 requests - weak scaling)

Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--num-prompts", type=int, default=64)
    ap.add_argument("--input-len", type=int, default=1024)
    ap.add_argument("--output-len", type=int, default=100)
    ap.add_argument("--evict", type=float, default=0.25, help="fraction of the prompt evicted (0 = off)")
    ap.add_argument("--cache-budget", type=int, default=0, help="overrides --evict when > 0")
    ap.add_argument("--prompt-evict-method", default="streamingLLM")
    ap.add_argument("--decode-evict-method", default="value_l2",
                    help="streamingLLM | streamingLLM-1 | value_l2 (= global) | local | global")
    ap.add_argument("--cache-prune-type", choices=["budget", "percentage"], default="budget",
                    help="percentage = the README-era mode driven by --evict-freq / --num-block-merge "
                         "(prompt: 1/evict_freq of the middle goes; decode: every evict_freq blocks "
                         "num_block_merge blocks are merged into one)")
    ap.add_argument("--evict-freq", type=int, default=0,
                    help="percentage prune type: the divisor above (README: 2 and 4); ignored for budget")
    ap.add_argument("--num-block-merge", type=int, default=2)
    # flags of the reference's benchmark_throughput.py command lines, accepted so that the README
    # commands can be pasted; they select nothing here
    ap.add_argument("--enable-paged-eviction", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--enforce-eager", action="store_true", help=argparse.SUPPRESS)
    for flag in ("--backend", "--model", "--distributed-executor-backend", "--tensor-parallel-size"):
        ap.add_argument(flag, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--evict-prefill", action="store_true",
                    help="prune before prefill attention (default: attend the full prompt, as the "
                         "reference's scripts do with --disable-evict-prefill)")
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--num-heads", type=int, default=32)
    ap.add_argument("--num-kv-heads", type=int, default=8)
    ap.add_argument("--head-size", type=int, default=128)
    ap.add_argument("--block-size", type=int, default=16)
    ap.add_argument("--prefill-batch", type=int, default=8, help="prompts per prefill step")
    ap.add_argument("--mode", choices=["tp", "tp-strong", "dp"], default="tp",
                    help="N>1: tp = heads sharded over the ranks, requests and prefill batch scaled by N "
                         "so that every rank streams as many KV bytes as the single GPU does (weak "
                         "scaling); tp-strong = heads sharded, the same 64 requests; dp = replicas")
    ap.add_argument("--oproj-allreduce", action="store_true",
                    help="tp: also all-reduce a [tokens, hidden] tensor after every layer, the exchange "
                         "of the row-parallel o_proj that follows attention in the model (that GEMM is "
                         "outside the path and not run, so its reduction is off by default)")
    ap.add_argument("--per-request-scheduler", action="store_true",
                    help="drive the decode steps through the per-request scheduler / metadata builder "
                         "(the reference's shape of the host code) instead of the vectorised ones")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--alt-evict", type=float, default=0.5,
                    help="second eviction fraction measured after the timed region and reported under "
                         "config.alt (the metric is quoted at 25 %% and 50 %%); 0 = skip")
    return ap.parse_args()


class Workload:
    """Drives one generation through the backend with a private scheduler and KV cache."""

    def __init__(self, a, dev, tp_size, tp_rank, tp_group):
        from vllm_paged_compression_amd.backend import PagedEvictBackend
        from vllm_paged_compression_amd.config import (CacheConfigView, PagedEvictConfig,
                                                       budget_from_percentage)
        from vllm_paged_compression_amd.parallel import shard_heads
        self.a, self.dev = a, dev
        self.tp_size, self.tp_group = tp_size, tp_group
        sh = shard_heads(a.num_heads, a.num_kv_heads, tp_size, tp_rank)
        self.Hq, self.Hkv, self.D, self.bs = sh.num_heads, sh.num_kv_heads, a.head_size, a.block_size
        self.hidden = a.num_heads * a.head_size
        self.cfg = None
        self.budget = 0
        method = {"value_l2": "global"}.get(a.decode_evict_method, a.decode_evict_method)
        self.percentage = a.cache_prune_type == "percentage"
        if self.percentage:
            from vllm_paged_compression_amd.page_evict_kv_util import percentage_pruned_prompt
            if a.evict_freq < 1:
                raise SystemExit("--cache-prune-type percentage needs --evict-freq >= 1")
            self.cfg = PagedEvictConfig(not a.evict_prefill, "percentage", "streamingLLM", 0, -1,
                                        evict_freq=a.evict_freq, num_block_merge=a.num_block_merge,
                                        prompt_evict_method=a.prompt_evict_method,
                                        decode_evict_method=a.decode_evict_method)
            # "budget" here = tokens of the prompt that stay (sizes the CPU baseline's kv length)
            self.budget = percentage_pruned_prompt(a.input_len, a.block_size, a.evict_freq, 1,
                                                   a.num_block_merge)[1]
        elif a.evict > 0 or a.cache_budget > 0:
            self.budget = a.cache_budget or budget_from_percentage(a.evict, a.input_len, a.block_size)
            self.cfg = PagedEvictConfig(not a.evict_prefill, "budget", method, self.budget, 3)
        self.ccfg = CacheConfigView(a.block_size, self.cfg)
        self.backend = PagedEvictBackend
        B = a.num_prompts
        kept = a.input_len if self.cfg is None else min(a.input_len, self.budget + a.input_len % a.block_size)
        max_kv = (a.input_len + a.output_len) if self.cfg is None else kept + a.block_size + 1
        if self.percentage:
            max_kv = self.budget + a.output_len + a.block_size
        self.blocks_per_seq = -(-max_kv // a.block_size) + 1
        self.blocks_per_seq_full = -(-(a.input_len + a.output_len) // a.block_size)
        self.num_blocks = B * self.blocks_per_seq + 8
        shape = PagedEvictBackend.get_kv_cache_shape(self.num_blocks, self.bs, self.Hkv, self.D)
        self.kv = [torch.zeros(shape, dtype=torch.bfloat16, device=dev) for _ in range(a.layers)]
        self.impls = [PagedEvictBackend.get_impl_cls()(
            self.Hq, self.D, self.D ** -0.5, self.Hkv, None, None, "auto", None, None,
            cache_config=self.ccfg, prefix=f"model.layers.{i}.self_attn.attn",
            prompt_evict_method=(a.prompt_evict_method
                                 if self.cfg is not None and not self.percentage else None),
            tp_group=tp_group, total_num_kv_heads=a.num_kv_heads,
            kv_head_owner=tp_rank % max(1, tp_size // a.num_kv_heads) == 0,
            max_num_seqs=B, max_blocks_per_seq=4 * self.blocks_per_seq_full)
            for i in range(a.layers)]
        g = torch.Generator(device=dev).manual_seed(1234 + tp_rank)
        sc = self.D ** -0.5

        def rnd(T, H):
            return torch.empty(T, H * self.D, dtype=torch.bfloat16, device=dev).uniform_(-sc, sc, generator=g)

        Tp = a.prefill_batch * a.input_len
        self.qp, self.kp, self.vp = rnd(Tp, self.Hq), rnd(Tp, self.Hkv), rnd(Tp, self.Hkv)
        self.qd, self.kd, self.vd = rnd(B, self.Hq), rnd(B, self.Hkv), rnd(B, self.Hkv)
        self.o_partial_p = torch.zeros(Tp, self.hidden, dtype=torch.bfloat16, device=dev)
        self.o_partial_d = torch.zeros(B, self.hidden, dtype=torch.bfloat16, device=dev)
        self.allreduce = tp_size > 1 and a.oproj_allreduce
        # decode-kernel timing (HIP events on the launch stream) of one layer
        self.timed = []          # (start_event, end_event, algorithmic_bytes)
        self.timing_on = False
        self.kv_blocks_peak = 0

    # ---- metadata -------------------------------------------------------------------------
    def _builder(self, inter):
        runner = SimpleNamespace(device=self.dev, cache_config=self.ccfg, pin_memory=False)
        ib = SimpleNamespace(runner=runner, sliding_window=None, block_size=self.bs,
                             chunked_prefill_enabled=False, inter_data_list=inter)
        return self.backend.make_metadata_builder(ib)

    def _build(self, inter, seq_lens, q_lens):
        return self._builder(inter).build(seq_lens, q_lens, -1, len(seq_lens))

    def _layers(self, q, k, v, meta, o_partial, decode_bytes=None):
        """All layers of one step.  For decode steps of the timed region, layers 1..L-1 (layer 0
        also carries the step's eviction launch) are bracketed by one pair of HIP events on the
        launch stream: every launch between them is the decode-attention kernel, so the span
        divided by the launch count is that kernel's average launch duration."""
        from vllm_paged_compression_amd import parallel
        span = (decode_bytes is not None and self.timing_on and not self.allreduce
                and len(self.impls) > 2)
        for li, impl in enumerate(self.impls):
            if span and li == 1:
                s_ev = torch.cuda.Event(enable_timing=True)
                s_ev.record()
            impl.forward(q, k, v, self.kv[li], meta)
            if self.allreduce:
                parallel.all_reduce_sum(o_partial, self.tp_group)
        if span:
            e_ev = torch.cuda.Event(enable_timing=True)
            e_ev.record()
            self.timed.append((s_ev, e_ev, decode_bytes, len(self.impls) - 1))

    def _evicted_now(self, L, kv_len) -> int:
        """Tokens that leave the attention view at the decode step whose logical length is L
        (the kv-length adjustment of model_runner.py:538-543; every request of the batch has the
        same lengths)."""
        if self.cfg is None:
            return 0
        bs = self.bs
        if self.percentage:
            from vllm_paged_compression_amd.page_evict_kv_util import percentage_decode_trigger
            return ((self.cfg.num_block_merge - 1) * bs
                    if percentage_decode_trigger(self.cfg, L, int(kv_len), bs) else 0)
        trig = self.cfg.evict_method != "inverse_key_l2" and L > self.budget and L % bs == 0
        return bs if trig else 0

    # ---- one generation ---------------------------------------------------------------------
    def generate(self):
        if self.a.per_request_scheduler:
            return self._generate_per_request()
        import numpy as np
        from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler
        a, bs = self.a, self.bs
        B, L0 = a.num_prompts, a.input_len
        sched = BatchEvictionScheduler(bs, self.num_blocks, self.cfg, max_seqs=B,
                                       max_blocks_per_seq=self.blocks_per_seq_full + 2)
        rids = [f"req{i}" for i in range(B)]
        for i, r in enumerate(rids):
            sched.add_request(r, i, L0)
        # prefill, --prefill-batch prompts at a time
        for p0 in range(0, B, a.prefill_batch):
            grp = list(range(p0, min(B, p0 + a.prefill_batch)))
            inter = [SimpleNamespace(request_id=rids[i], is_prompt=True,
                                     block_tables={i: sched.get_block_table(rids[i])},
                                     seq_kv_lens={i: sched.get_seq_kv_len(rids[i])}, seq_ids=[i],
                                     input_tokens=[range(L0)], orig_seq_lens=[L0], seq_lens=[L0],
                                     query_lens=[L0], context_lens=[0]) for i in grp]
            meta = self._build(inter, [L0] * len(grp), [L0] * len(grp))
            T = len(grp) * L0
            self._layers(self.qp[:T], self.kp[:T], self.vp[:T], meta, self.o_partial_p[:T])
        # decode: one vectorised scheduler step + one metadata build per step
        L = L0
        es = 2
        for _ in range(a.output_len):
            batch = sched.schedule_decode(rids)
            L += 1
            kv_adj = batch.kv_lens - self._evicted_now(L, batch.kv_lens[0])
            meta = self._builder([]).build_decode_batch(batch, kv_adj)
            nbytes = int((2 * kv_adj * self.Hkv * self.D * es + -(-kv_adj // bs) * 4).sum()) \
                + 2 * B * self.Hq * self.D * es
            self._layers(self.qd, self.kd, self.vd, meta, self.o_partial_d, decode_bytes=nbytes)
            if self.cfg is not None:
                sched.commit(self.impls[0].get_rmv_block_idx())
            used = self.num_blocks - sched.get_num_free_gpu_blocks()
            self.kv_blocks_peak = max(self.kv_blocks_peak, used)
        for r in rids:
            sched.finish_request(r)

    def _generate_per_request(self):
        from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
        a, bs = self.a, self.bs
        B, L0 = a.num_prompts, a.input_len
        sched = EvictionScheduler(EvictBlockSpaceManager(bs, self.num_blocks, self.cfg, watermark=0.0))
        bm = sched.block_manager
        rids = [f"req{i}" for i in range(B)]
        for i, r in enumerate(rids):
            sched.add_request(r, i, L0)
        # prefill, --prefill-batch prompts at a time
        for p0 in range(0, B, a.prefill_batch):
            grp = list(range(p0, min(B, p0 + a.prefill_batch)))
            inter = [SimpleNamespace(request_id=rids[i], is_prompt=True,
                                     block_tables={i: list(bm.get_block_table(i))},
                                     seq_kv_lens={i: bm.get_seq_kv_len(i)}, seq_ids=[i],
                                     input_tokens=[range(L0)], orig_seq_lens=[L0], seq_lens=[L0],
                                     query_lens=[L0], context_lens=[0]) for i in grp]
            meta = self._build(inter, [L0] * len(grp), [L0] * len(grp))
            T = len(grp) * L0
            self._layers(self.qp[:T], self.kp[:T], self.vp[:T], meta, self.o_partial_p[:T])
        # decode
        L = L0
        for _ in range(a.output_len):
            tables, kv_lens = sched.schedule_decode(rids)
            L += 1
            drop = self._evicted_now(L, kv_lens[0])
            inter, kv_adj = [], []
            for i in range(B):
                kvl = kv_lens[i] - drop
                kv_adj.append(kvl)
                inter.append(SimpleNamespace(request_id=rids[i], is_prompt=False,
                                             block_tables={i: tables[i]}, seq_kv_lens={i: kvl},
                                             seq_ids=[i], input_tokens=[(0,)], orig_seq_lens=[L],
                                             seq_lens=[L], query_lens=[1], context_lens=[kv_lens[i] - 1]))
            meta = self._build(inter, [L] * B, [1] * B)
            es = 2
            nbytes = sum(2 * kvl * self.Hkv * self.D * es + -(-kvl // bs) * 4 for kvl in kv_adj) \
                + 2 * B * self.Hq * self.D * es
            self._layers(self.qd, self.kd, self.vd, meta, self.o_partial_d, decode_bytes=nbytes)
            if self.cfg is not None:
                sched.commit(self.impls[0].get_rmv_block_idx())
            used = self.num_blocks - bm.get_num_free_gpu_blocks()
            self.kv_blocks_peak = max(self.kv_blocks_peak, used)
        for r in rids:
            sched.finish_request(r)


def cpu_baseline(a, budget):
    """Reference CPU path (csrc/cpu paged_attention_v1 + reshape_and_cache compiled from the
    reference's sources, torch CPU SDPA for prefill, composed like the reference's CPU
    TORCH_SDPA backend) — or the oracle's C port when oracle/_ref is absent — timed on a
    bounded sample of the same workload and scaled to the whole job."""
    import torch.nn.functional as F
    from oracle import pe_oracle
    ref_lib = os.path.join(ROOT, "oracle", "_ref", "libpe_ref.so")
    kind = "port"
    if os.path.exists(ref_lib):
        try:
            torch.ops.load_library(ref_lib)
            kind = "reference"
        except Exception:
            kind = "port"
    cores = torch.get_num_threads() if kind == "reference" else 1
    B, Hq, Hkv, D, bs = a.num_prompts, a.num_heads, a.num_kv_heads, a.head_size, a.block_size
    kv_len = (budget + bs // 2) if budget else a.input_len + a.output_len // 2
    nblk = -(-kv_len // bs)
    dt = torch.bfloat16
    sc = D ** -0.5
    g = torch.Generator().manual_seed(0)
    nb = B * nblk
    kc = ((torch.rand(nb, Hkv, D // 8, bs, 8, generator=g) * 2 - 1) * sc).to(dt)
    vc = ((torch.rand(nb, Hkv, D, bs, generator=g) * 2 - 1) * sc).to(dt)
    q = ((torch.rand(B, Hq, D, generator=g) * 2 - 1) * sc).to(dt)
    k1 = ((torch.rand(B, Hkv, D, generator=g) * 2 - 1) * sc).to(dt)
    bt = torch.randperm(nb, generator=g).to(torch.int32).reshape(B, nblk)
    lens = torch.full((B,), kv_len, dtype=torch.int32)
    slots = (bt[:, -1].to(torch.int64) * bs + (kv_len - 1) % bs)
    out = torch.empty_like(q)
    n_dec = 3 if kind == "reference" else 1

    def decode_step():
        if kind == "reference":
            torch.ops.pe_ref.reshape_and_cache(k1, k1, kc, vc, slots, "auto", 1.0, 1.0)
            torch.ops.pe_ref.paged_attention_v1(out, q, kc, vc, Hkv, sc, bt, lens, bs, kv_len, None,
                                                "auto", 1.0, 1.0, 0, 0, 0, 64, 0)
        else:
            pe_oracle.reshape_and_cache(k1, k1, kc, vc, slots)
            pe_oracle.paged_attention(q[:4], kc, vc, bt[:4], lens[:4], sc)

    decode_step()
    t0 = time.perf_counter()
    for _ in range(n_dec):
        decode_step()
    t_dec = (time.perf_counter() - t0) / n_dec
    if kind == "port":
        t_dec *= B / 4.0
    # prefill: one prompt, one layer, torch CPU SDPA with GQA expansion as the CPU backend does
    Lp = a.input_len
    qp = torch.rand(1, Hq, Lp, D, generator=g).to(dt)
    kp = torch.rand(1, Hkv, Lp, D, generator=g).to(dt).repeat_interleave(Hq // Hkv, dim=1)
    t0 = time.perf_counter()
    F.scaled_dot_product_attention(qp, kp, kp, is_causal=True, scale=sc)
    t_pre = time.perf_counter() - t0
    t_job = a.layers * (t_pre * B + t_dec * a.output_len)
    return {
        "value": round(B * a.output_len / t_job, 3), "unit": "output tok/s (attention path only)",
        "cores": cores, "kind": kind,
        "sample": f"1 layer: {n_dec} decode steps (KV write + paged_attention_v1, batch "
                  f"{B if kind == 'reference' else 4}, kv {kv_len}) + SDPA prefill of 1 prompt of {Lp}; "
                  f"scaled to {a.layers} layers x ({B} prompts + {a.output_len} steps); no eviction logic "
                  "(the reference's CPU backend has none)"}


def workload_name(a, wl) -> str:
    return (f"Llama-3-8B attention geometry (Hq {a.num_heads}, Hkv {a.num_kv_heads}, D "
            f"{a.head_size}, {a.layers} layers), block {a.block_size}, {a.num_prompts} "
            f"requests {a.input_len}->{a.output_len}, "
            + ((f"percentage prune type, evict-freq {a.evict_freq}, num-block-merge "
                f"{a.num_block_merge} ({wl.budget} prompt tokens kept)" if wl.percentage else
                f"{'budget ' + str(wl.budget)} ({a.evict:.0%} eviction)")
               + f", prompt {a.prompt_evict_method} / decode {a.decode_evict_method}"
               if wl.cfg is not None else "no eviction"))


def main():
    a = parse()
    if os.environ.get("PE_BENCH_HANG_DUMP"):  # stacks of every thread if the run is still going then
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PE_BENCH_HANG_DUMP"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # PE_BENCH_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than
    # ranks (ranks share devices, collectives go through the host); the measured runs use RCCL
    backend = os.environ.get("PE_BENCH_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    n_gpus = world
    tp = a.mode in ("tp", "tp-strong") and world > 1
    if tp and a.mode == "tp":
        # weak scaling under tensor parallelism: N times the requests over 1/N of the heads
        a.num_prompts *= world
        a.prefill_batch *= world
    import vllm_paged_compression_amd as pkg
    pkg.load_library()
    wl = Workload(a, dev, world if tp else 1, rank if tp else 0, dist.group.WORLD if tp else None)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        wl.generate()
    sync()
    # everything allocated so far (library, caches, launch tables) is long-lived: take it out of
    # the collector's sight so that a full collection (tens of ms) cannot land in a generation
    import gc
    gc.collect()
    gc.freeze()
    wl.timing_on = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wl.generate()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    replicas = 1 if (tp or world == 1) else world
    total_tokens = a.num_prompts * a.output_len * a.steps * replicas
    value = total_tokens / dt

    # decode kernel: measured launch durations -> achieved algorithmic bandwidth
    durs = [s.elapsed_time(e) * 1e-3 / n for s, e, _, n in wl.timed]   # seconds per launch
    byts = [b for _, _, b, _ in wl.timed]
    roof = None
    if durs:
        avg_t = sum(durs) / len(durs)
        avg_b = sum(byts) / len(byts)
        ach = avg_b / avg_t / 1e9
        # HBM bytes per launch from the PMC passes kept under profiles/ (FETCH_SIZE x2 per the
        # gfx950 correction + WRITE_SIZE, averaged over the decode launches of this same bench
        # command), as a ratio to the algorithmic bytes; applied only to the workload it was
        # measured on
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "round1_decode_pmc.json")))
            if pmc["workload"] == workload_name(a, wl) and \
                    abs(avg_b / pmc["algorithmic_bytes_per_launch"] - 1.0) < 0.05:
                traffic = int(avg_b * pmc["traffic_over_algorithmic"])
        except Exception:
            traffic = None
        roof = {"bound": "hbm", "kernel": "paged_decode_mfma_kernel", "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": traffic, "avg_launch_us": round(avg_t * 1e6, 2),
                "algorithmic_bytes_per_launch": int(avg_b),
                "launches_timed": sum(n for _, _, _, n in wl.timed),
                "includes": "fused KV write of the step's token (pe_decode_step)"}
    es = 2
    blk_bytes = 2 * a.block_size * a.num_kv_heads * a.head_size * es * a.layers
    full_blocks = a.num_prompts * wl.blocks_per_seq_full
    kv = {"kv_gb_held_peak": round(wl.kv_blocks_peak * blk_bytes / 2 ** 30, 3),
          "kv_gb_without_eviction": round(full_blocks * blk_bytes / 2 ** 30, 3)}
    kv["kv_gb_saved"] = round(kv["kv_gb_without_eviction"] - kv["kv_gb_held_peak"], 3)
    out = {
        "metric": "output tok/s + KV GB saved, Llama-3-8B 1k->100 @25/50% evict, 1-8 MI355X",
        "value": round(value, 2), "unit": "output tok/s (attention hot path)",
        "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong" if a.mode == "tp-strong" and world > 1 else "weak", "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": workload_name(a, wl),
            "step": "one full generation of the batch through prefill + decode attention of all layers",
            "parallelism": (f"tp{world}" if tp else f"dp{world}") if world > 1 else "single",
            "oproj_allreduce": bool(wl.allreduce), **kv},
    }
    main_budget = wl.budget
    if a.alt_evict > 0 and a.cache_budget == 0 and a.cache_prune_type == "budget" \
            and abs(a.alt_evict - a.evict) > 1e-9:
        # the same job at the metric's other eviction fraction (outside the timed region above)
        import copy
        a2 = copy.copy(a)
        a2.evict = a.alt_evict
        main_budget = wl.budget
        del wl
        torch.cuda.empty_cache()
        wl2 = Workload(a2, dev, world if tp else 1, rank if tp else 0, dist.group.WORLD if tp else None)
        wl2.generate()
        sync()
        t0 = time.perf_counter()
        n2 = max(1, min(a.steps, 2))
        for _ in range(n2):
            wl2.generate()
        sync()
        dt2 = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        held2 = wl2.kv_blocks_peak * blk_bytes / 2 ** 30
        out["config"]["alt"] = {
            "evict": a.alt_evict, "budget": wl2.budget, "steps": n2,
            "value": round(a.num_prompts * a.output_len * n2 * replicas / dt2, 2),
            "ms_per_step": round(dt2 / n2 * 1e3, 3), "kv_gb_held_peak": round(held2, 3),
            "kv_gb_saved": round(kv["kv_gb_without_eviction"] - held2, 3)}
    if roof is not None:
        out["roofline"] = roof
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(a, main_budget)
        except Exception as e:  # the baseline never decides the run
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
