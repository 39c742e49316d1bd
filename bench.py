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
  python bench.py --gpus N            (no launcher: starts the N ranks itself, before any GPU call)
  python bench.py --config c3         (the other BASELINE.json configurations: c1 OPT-125m without
      eviction, c2 = the default, c3 8192->1024 x 64 at 50 %, c4 70B geometry 4096->512 at 25 %
      (one rank's heads of TP=8 with --emulate-tp 8), c5 32768->2048 at 50 %)

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

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16 matrix peak (no sparsity)

# BASELINE.json configurations as flag presets (explicit flags still override them)
PRESETS = {
    "c1": dict(num_prompts=1, input_len=1024, output_len=100, evict=0.0, alt_evict=0.0, layers=12,
               num_heads=12, num_kv_heads=12, head_size=64, dtype="fp16", prefill_batch=1,
               model="OPT-125m", graph_decode=True),
    "c2": dict(),
    "c3": dict(num_prompts=64, input_len=8192, output_len=1024, evict=0.5, alt_evict=0.0, steps=1,
               warmup=1),
    "c4": dict(num_prompts=64, input_len=4096, output_len=512, evict=0.25, alt_evict=0.0, layers=80,
               num_heads=64, num_kv_heads=8, steps=1, warmup=1, model="Llama-3-70B"),
    "c5": dict(num_prompts=64, input_len=32768, output_len=2048, evict=0.5, alt_evict=0.0, steps=1,
               warmup=0, prefill_batch=1, decode_evict_method="streamingLLM"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(PRESETS), default="c2",
                    help="BASELINE.json configuration (default c2 = the headline: 64 requests "
                         "1024->100 at 25 %% eviction)")
    ap.add_argument("--emulate-tp", type=int, default=1,
                    help="run ONE rank's head shard of a TP=N deployment on this GPU (no collectives)")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16")
    ap.add_argument("--model", default="Llama-3-8B", help="label only")
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
    for flag in ("--backend", "--distributed-executor-backend", "--tensor-parallel-size"):
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
    ap.add_argument("--host-tables", action="store_true",
                    help="vectorised scheduler with the block tables kept on the host and uploaded "
                         "every decode step (default: tables resident on the GPU, one int per "
                         "sequence uploaded per step)")
    ap.add_argument("--graph-decode", dest="graph_decode", action="store_true", default=None,
                    help="replay the launches of a decode step (all layers) from HIP graphs: one for the "
                         "steps that evict nothing, one (budget prune type, streamingLLM / global / "
                         "local, single GPU) for the steps with an eviction decision, whose victims "
                         "are read back after the replay; other eviction steps run eagerly.  Needs the "
                         "resident tables (the captured launches read the step's metadata from the same "
                         "device buffers every step).  Default: on when a decode launch moves less than "
                         "150 MB, i.e. when the step would be bound by the host's launch rate (few "
                         "sequences, or one rank's heads of a tensor-parallel job); the headline "
                         "workload (204 MB per launch) runs eagerly")
    ap.add_argument("--no-graph-decode", dest="graph_decode", action="store_false",
                    help="launch every decode step eagerly")
    ap.add_argument("--rope", choices=["off", "separate", "fused"], default="off",
                    help="also run the rotary embedding that precedes attention in the model "
                         "(llama.py:202-203) on every layer's query/key rows: separate = its own "
                         "launch in front of the backend (the reference's order), fused = handed to "
                         "the backend, which rotates a decode layer's rows inside the attention launch "
                         "(pe_decode_step_rope).  off (default) = the headline workload, rows arrive "
                         "rotated")
    ap.add_argument("--reference-head-compat", action="store_true",
                    help="run with every switch that reproduces the reference's HEAD "
                         "(config.CompatFlags.reference_head(): a decision per layer with random victims, "
                         "the in-loop table slice and its strided read, unfused KV write, ...) instead of "
                         "the intended semantics the headline is measured on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--alt-evict", type=float, default=0.5,
                    help="second eviction fraction measured after the timed region and reported under "
                         "config.alt (the metric is quoted at 25 %% and 50 %%); 0 = skip")
    pre, _ = ap.parse_known_args()
    ap.set_defaults(**PRESETS[pre.config])
    return ap.parse_args()


class Workload:
    """Drives one generation through the backend with a private scheduler and KV cache."""

    def __init__(self, a, dev, tp_size, tp_rank, tp_group):
        from vllm_paged_compression_amd.backend import PagedEvictBackend
        from vllm_paged_compression_amd.config import (CacheConfigView, PagedEvictConfig,
                                                       budget_from_percentage)
        from vllm_paged_compression_amd.parallel import shard_heads
        self.a, self.dev = a, dev
        self.tp_size, self.tp_group, self.tp_rank = tp_size, tp_group, tp_rank
        self.dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float16
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
        elif self.cfg is not None and method == "inverse_key_l2":
            # zeroes one slot per step past the budget and never frees a block
            # (torch_cuda_sdpa.py:1435-1475, model_runner.py:545-547): the pruned prompt keeps growing
            max_kv = kept + a.output_len + a.block_size
        self.blocks_per_seq = -(-max_kv // a.block_size) + 1
        self.blocks_per_seq_full = -(-(a.input_len + a.output_len) // a.block_size)
        self.num_blocks = B * self.blocks_per_seq + 8
        shape = PagedEvictBackend.get_kv_cache_shape(self.num_blocks, self.bs, self.Hkv, self.D)
        self.kv = [torch.zeros(shape, dtype=self.dtype, device=dev) for _ in range(a.layers)]
        compat = None
        if a.reference_head_compat:
            from vllm_paged_compression_amd.config import CompatFlags
            compat = CompatFlags.reference_head()
        self.impls = [PagedEvictBackend.get_impl_cls()(
            self.Hq, self.D, self.D ** -0.5, self.Hkv, None, None, "auto", None, None,
            cache_config=self.ccfg, prefix=f"model.layers.{i}.self_attn.attn", compat=compat,
            prompt_evict_method=(a.prompt_evict_method
                                 if self.cfg is not None and not self.percentage else None),
            tp_group=tp_group, total_num_kv_heads=a.num_kv_heads,
            kv_head_owner=tp_rank % max(1, tp_size // a.num_kv_heads) == 0,
            max_num_seqs=B, max_blocks_per_seq=2 * self.blocks_per_seq_full)
            for i in range(a.layers)]
        g = torch.Generator(device=dev).manual_seed(1234 + tp_rank)
        sc = self.D ** -0.5

        def rnd(T, H):
            return torch.empty(T, H * self.D, dtype=self.dtype, device=dev).uniform_(-sc, sc, generator=g)

        Tp = a.prefill_batch * a.input_len
        self.qp, self.kp, self.vp = rnd(Tp, self.Hq), rnd(Tp, self.Hkv), rnd(Tp, self.Hkv)
        self.qd, self.kd, self.vd = rnd(B, self.Hq), rnd(B, self.Hkv), rnd(B, self.Hkv)
        self.allreduce = tp_size > 1 and a.oproj_allreduce and tp_group is not None
        self.o_partial_p = self.o_partial_d = None
        if self.allreduce:
            from vllm_paged_compression_amd import parallel
            if id(tp_group) not in parallel._one_shot and os.environ.get("PE_AR_DISABLE", "0") in ("", "0"):
                # decode-sized partials (<= 8 MiB) go through the exchange over the peer-mapped
                # buffers (one stage, or two from 256-512 KiB), larger ones through the group's
                # backend (RCCL)
                parallel.enable_one_shot_all_reduce(tp_group, dev, 8 << 20)
            self.o_partial_p = torch.zeros(Tp, self.hidden, dtype=self.dtype, device=dev)
            self.o_partial_d = torch.zeros(B, self.hidden, dtype=self.dtype, device=dev)
        self.cos_sin = None
        if a.rope != "off":
            # cos/sin table of the model's rotary embedding (rot_dim = head size, NeoX pairing)
            max_pos = a.input_len + a.output_len + 1
            inv = 1.0 / (10000.0 ** (torch.arange(0, self.D, 2, dtype=torch.float32) / self.D))
            fr = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv)
            self.cos_sin = torch.cat([fr.cos(), fr.sin()], dim=-1).to(self.dtype).to(dev)
            self.pos_p = torch.arange(a.input_len, dtype=torch.int64, device=dev).repeat(a.prefill_batch)
            self.pos_d = torch.zeros(B, dtype=torch.int64, device=dev)
        # decode-kernel timing (HIP events on the launch stream) of one layer
        self.timed = []          # (start_event, end_event, algorithmic_bytes)
        self.timed_prefill = []  # (start_event, end_event, flops) per prefill-attention launch
        self.timing_on = False
        self.kv_blocks_peak = 0
        # launch-bound decode steps are replayed from HIP graphs (see --graph-decode); the graphs
        # and the device-resident table state they read live as long as the workload, so a graph
        # captured in one generation serves the next ones too
        self._dev_state = None
        self._graphs = {}
        self.graph_replays = 0
        self.graph_evict_replays = 0
        per_launch = B * (2 * max_kv * self.Hkv * self.D * 2)      # K and V bytes a decode launch reads
        self.use_graph = (a.graph_decode if a.graph_decode is not None else per_launch < 150e6) \
            and not a.host_tables and not a.reference_head_compat and not a.per_request_scheduler \
            and not self.allreduce and a.rope == "off"

    # ---- metadata -------------------------------------------------------------------------
    def _builder(self, inter):
        runner = SimpleNamespace(device=self.dev, cache_config=self.ccfg, pin_memory=False)
        ib = SimpleNamespace(runner=runner, sliding_window=None, block_size=self.bs,
                             chunked_prefill_enabled=False, inter_data_list=inter)
        return self.backend.make_metadata_builder(ib)

    def _build(self, inter, seq_lens, q_lens):
        return self._builder(inter).build(seq_lens, q_lens, -1, len(seq_lens))

    def _layers(self, q, k, v, meta, o_partial, decode_bytes=None, prefill_flops=None, positions=None):
        """All layers of one step.  For decode steps of the timed region, layers 1..L-1 (layer 0
        also carries the step's eviction launch) are bracketed by one pair of HIP events on the
        launch stream: every launch between them is the decode-attention kernel, so the span
        divided by the launch count is that kernel's average launch duration."""
        from vllm_paged_compression_amd import ops, parallel
        span = (decode_bytes is not None and self.timing_on and not self.allreduce
                and len(self.impls) > 2 and self.a.rope != "separate")
        rope = None
        if self.cos_sin is not None:
            rope = (positions, self.cos_sin, True)
        sink = None
        if prefill_flops is not None and self.timing_on:
            sink = ops.prefill_event_sink = []
        for li, impl in enumerate(self.impls):
            if span and li == 1:
                s_ev = torch.cuda.Event(enable_timing=True)
                s_ev.record()
            if rope is not None and self.a.rope == "separate":
                ops.rotary_embedding(positions, q, k, self.D, self.cos_sin, True)
                impl.forward(q, k, v, self.kv[li], meta)
            elif rope is not None:
                impl.forward(q, k, v, self.kv[li], meta, rope=rope)
            else:
                impl.forward(q, k, v, self.kv[li], meta)
            if self.allreduce:
                parallel.all_reduce_sum(o_partial, self.tp_group)
        if span:
            e_ev = torch.cuda.Event(enable_timing=True)
            e_ev.record()
            self.timed.append((s_ev, e_ev, decode_bytes, len(self.impls) - 1))
        if sink is not None:
            ops.prefill_event_sink = None
            self.timed_prefill += [(s, e, prefill_flops) for s, e in sink]

    def _capture_decode(self, meta, max_len: int):
        """The launches of one decode step, all layers, captured into a HIP graph.  The step's
        metadata tensors are the scheduler's resident device buffers, so later steps replay the
        same launches on new contents; the length that sizes the KV splits is fixed at the
        generation's maximum.  On a step with an eviction decision the first layer's decision
        launch (victim choice + in-place table compaction, which tests every sequence's trigger
        on the device) is part of the graph; returns (graph, device tensor of the victims)."""
        meta.max_decode_seq_len = max_len
        dm = meta.decode_metadata
        if dm is not None and dm is not meta:
            dm.max_decode_seq_len = max_len
        if self.cfg is not None and dm is not None and dm.seq_kv_lens:
            # the impl sizes the splits by the longest physical length: pin it for all replays
            dm.seq_kv_lens = [min(max_len, self.blocks_per_seq * self.bs)] * len(dm.seq_kv_lens)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                # the step's table append is the graph's first node: it reads the step's entries
                # from the scheduler's pinned ring, so a replay is the only submission of a step
                self._dev_state.launch_advance()
                for li, impl in enumerate(self.impls):
                    impl.forward(self.qd, self.kd, self.vd, self.kv[li], meta)
        torch.cuda.current_stream().wait_stream(side)
        dec = meta._pe_step.get("decode") or {}
        # (the third entry keeps alive what the captured decision launch reads besides the
        # resident table state: the score-row indices of the batch)
        return graph, dec.get("rmv_dev"), getattr(self._dev_state, "score_rows", None)

    def _replay_decode(self, graph, decode_bytes: int) -> None:
        if self.timing_on:
            s_ev = torch.cuda.Event(enable_timing=True)
            e_ev = torch.cuda.Event(enable_timing=True)
            s_ev.record()
            graph.replay()
            e_ev.record()
            self.timed.append((s_ev, e_ev, decode_bytes, len(self.impls)))
        else:
            graph.replay()
        self.graph_replays += 1

    def _prefill_flops(self, n_req: int) -> float:
        """4 * Hq * D * sum_i min(i+1, L_kv) per request: both contractions, causal, the kv length
        being the pruned one when the prompt is pruned before attention."""
        a = self.a
        L = a.input_len
        Lkv = L
        if self.cfg is not None and not self.cfg.disable_evict_prefill:
            Lkv = self.impls[0].kv_cache_pruner.get_pruned_length(L)
        pairs = Lkv * (Lkv + 1) // 2 + (L - Lkv) * Lkv
        return 4.0 * self.Hq * self.D * pairs * n_req

    @staticmethod
    def _op(buf, n):
        return None if buf is None else buf[:n]

    def _progress(self, what: str) -> None:
        """Long configurations report to stderr so that a watcher sees the run is alive."""
        if self.a.input_len * self.a.num_prompts >= 1 << 18 and self.tp_rank == 0:
            print(f"[bench] {what}", file=sys.stderr, flush=True)

    # ---- one generation ---------------------------------------------------------------------
    def generate(self):
        if self.a.per_request_scheduler:
            return self._generate_per_request()
        import numpy as np
        from vllm_paged_compression_amd import staging
        from vllm_paged_compression_amd.batch_scheduler import BatchEvictionScheduler
        from vllm_paged_compression_amd.backend import percentage_decode_victims
        from vllm_paged_compression_amd.page_evict_kv_util import budget_decode_trigger
        a, bs = self.a, self.bs
        B, L0 = a.num_prompts, a.input_len
        sched = BatchEvictionScheduler(bs, self.num_blocks, self.cfg, max_seqs=B,
                                       max_blocks_per_seq=self.blocks_per_seq_full + 2,
                                       device=None if (a.host_tables or a.reference_head_compat)
                                       else self.dev, device_state=self._dev_state)
        self._dev_state = sched.device_state
        if self.use_graph:
            self._dev_state.defer_advance = True      # issued below: eagerly, or by the step's graph
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
            self._layers(self.qp[:T], self.kp[:T], self.vp[:T], meta, self._op(self.o_partial_p, T),
                         prefill_flops=self._prefill_flops(len(grp)),
                         positions=None if self.cos_sin is None else self.pos_p[:T])
        self._progress("prefill done")
        # decode: one vectorised scheduler step + one metadata build per step
        L = L0
        es = 2
        use_graph = self.use_graph
        # an eviction step can be replayed too when its decision needs nothing from the host:
        # budget type with victims chosen on the device (or fixed), percentage type (the device
        # derives each row's trigger from its resident counters); one GPU (under tensor
        # parallelism the scores take an all-reduce, left eager)
        evict_graph = (use_graph and self.cfg is not None and self.tp_group is None
                       and (self.percentage or self.cfg.evict_method in ("streamingLLM", "global", "local")))
        warmed = "plain" in self._graphs
        warmed_ev = "evict" in self._graphs
        for it in range(a.output_len):
            batch = sched.schedule_decode(rids)
            L += 1
            # kv lengths as attention sees them (the model runner's adjustment on evict steps,
            # model_runner.py:538-543) come from the package: build_decode_batch applies
            # page_evict_kv_util.adjust_seq_kv_lens
            meta = self._builder([]).build_decode_batch(batch)
            kv_adj = np.asarray(meta.seq_kv_lens, dtype=np.int64)
            if it % 256 == 255:
                self._progress(f"decode step {it + 1}/{a.output_len}")
            nbytes = int((2 * kv_adj * self.Hkv * self.D * es + -(-kv_adj // bs) * 4).sum()) \
                + 2 * B * self.Hq * self.D * es
            # a step whose sequences all stay below their eviction trigger launches nothing but
            # the layers' attention, always on the same device buffers: it can be replayed
            if self.percentage:      # a prune can only fall on a multiple of evict_freq blocks
                plain = L % (bs * self.cfg.evict_freq) != 0
            else:
                plain = self.cfg is None or (self.cfg.evict_method in ("streamingLLM", "global", "local")
                                             and not budget_decode_trigger(self.cfg, L, bs))
            if self.cos_sin is not None:
                self.pos_d.fill_(L - 1)      # position of the step's token
            if use_graph and plain and warmed:
                if "plain" not in self._graphs:
                    self._graphs["plain"] = self._capture_decode(meta, L0 + a.output_len)
                self._replay_decode(self._graphs["plain"][0], nbytes)
            elif evict_graph and not plain and warmed_ev:
                if "evict" not in self._graphs:
                    self._graphs["evict"] = self._capture_decode(meta, L0 + a.output_len)
                graph_ev, rmv_dev = self._graphs["evict"][:2]
                graph_ev.replay()
                self.graph_replays += 1
                self.graph_evict_replays += 1
                if self.percentage:
                    # nothing depends on the data: the host knows the freed columns
                    sched.commit(percentage_decode_victims(self.cfg, batch.seq_lens, batch.table_lens,
                                                           rids, bs))
                else:
                    # the victims feed the host's free list: one small read-back per eviction step
                    rmv = staging.AsyncReadback(rmv_dev[:B]).tolist()
                    sched.commit({i: (rids[i], r) for i, r in enumerate(rmv) if r >= 0})
            else:
                if use_graph:
                    self._dev_state.launch_advance()
                self._layers(self.qd, self.kd, self.vd, meta, self.o_partial_d, decode_bytes=nbytes,
                             positions=None if self.cos_sin is None else self.pos_d)
                if plain:
                    warmed = True
                else:
                    warmed_ev = True
                if self.cfg is not None:
                    sched.commit(self.impls[0].get_rmv_block_idx())
            used = self.num_blocks - sched.get_num_free_gpu_blocks()
            self.kv_blocks_peak = max(self.kv_blocks_peak, used)
        for r in rids:
            sched.finish_request(r)

    def _generate_per_request(self):
        from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
        from vllm_paged_compression_amd.page_evict_kv_util import adjust_seq_kv_len
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
            self._layers(self.qp[:T], self.kp[:T], self.vp[:T], meta, self._op(self.o_partial_p, T),
                         prefill_flops=self._prefill_flops(len(grp)),
                         positions=None if self.cos_sin is None else self.pos_p[:T])
        # decode
        L = L0
        for _ in range(a.output_len):
            tables, kv_lens = sched.schedule_decode(rids)
            L += 1
            inter, kv_adj = [], []
            for i in range(B):
                kvl = adjust_seq_kv_len(self.cfg, L, kv_lens[i], bs)
                kv_adj.append(kvl)
                inter.append(SimpleNamespace(request_id=rids[i], is_prompt=False,
                                             block_tables={i: tables[i]}, seq_kv_lens={i: kvl},
                                             seq_ids=[i], input_tokens=[(0,)], orig_seq_lens=[L],
                                             seq_lens=[L], query_lens=[1], context_lens=[kv_lens[i] - 1]))
            meta = self._build(inter, [L] * B, [1] * B)
            es = 2
            nbytes = sum(2 * kvl * self.Hkv * self.D * es + -(-kvl // bs) * 4 for kvl in kv_adj) \
                + 2 * B * self.Hq * self.D * es
            if self.cos_sin is not None:
                self.pos_d.fill_(L - 1)
            self._layers(self.qd, self.kd, self.vd, meta, self.o_partial_d, decode_bytes=nbytes,
                         positions=None if self.cos_sin is None else self.pos_d)
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
    bounded sample of the same workload and scaled to the whole job.  Every timed piece is run
    once untimed first; the figure is the median of the repeats."""
    import statistics
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
    threads = torch.get_num_threads()
    cores = threads if kind == "reference" else 1
    B, Hq, Hkv, D, bs = a.num_prompts, a.num_heads, a.num_kv_heads, a.head_size, a.block_size
    kv_len = (budget + bs // 2) if budget else a.input_len + a.output_len // 2
    # bounded sample: at most 16 sequences of the batch (decode time scales with the batch)
    Bs = min(B, 16) if kind == "reference" else min(B, 4)
    nblk = -(-kv_len // bs)
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    x = 16 // 2
    sc = D ** -0.5
    g = torch.Generator().manual_seed(0)
    nb = Bs * nblk
    kc = ((torch.rand(nb, Hkv, D // x, bs, x, generator=g) * 2 - 1) * sc).to(dt)
    vc = ((torch.rand(nb, Hkv, D, bs, generator=g) * 2 - 1) * sc).to(dt)
    q = ((torch.rand(Bs, Hq, D, generator=g) * 2 - 1) * sc).to(dt)
    k1 = ((torch.rand(Bs, Hkv, D, generator=g) * 2 - 1) * sc).to(dt)
    bt = torch.randperm(nb, generator=g).to(torch.int32).reshape(Bs, nblk)
    lens = torch.full((Bs,), kv_len, dtype=torch.int32)
    slots = (bt[:, -1].to(torch.int64) * bs + (kv_len - 1) % bs)
    out = torch.empty_like(q)

    def decode_step():
        if kind == "reference":
            torch.ops.pe_ref.reshape_and_cache(k1, k1, kc, vc, slots, "auto", 1.0, 1.0)
            torch.ops.pe_ref.paged_attention_v1(out, q, kc, vc, Hkv, sc, bt, lens, bs, kv_len, None,
                                                "auto", 1.0, 1.0, 0, 0, 0, 64, 0)
        else:
            pe_oracle.reshape_and_cache(k1, k1, kc, vc, slots)
            pe_oracle.paged_attention(q, kc, vc, bt, lens, sc)

    def median_time(fn, repeats, budget_s):
        fn()                                    # warm (page-in, thread pool start)
        ts = []
        t_all = time.perf_counter()
        for _ in range(repeats):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all > budget_s and len(ts) >= 3:
                break
        return statistics.median(ts), len(ts)

    t_dec, n_dec = median_time(decode_step, 7, 8.0)
    t_dec *= B / Bs
    # prefill: one prompt, one layer, torch CPU SDPA with GQA expansion as the CPU backend does;
    # prompts longer than 4096 are sampled at 4096 and scaled by the causal pair count
    Lp = min(a.input_len, 4096)
    qp = torch.rand(1, Hq, Lp, D, generator=g).to(dt)
    kp = torch.rand(1, Hkv, Lp, D, generator=g).to(dt).repeat_interleave(Hq // Hkv, dim=1)
    t_pre, n_pre = median_time(
        lambda: F.scaled_dot_product_attention(qp, kp, kp, is_causal=True, scale=sc), 5, 10.0)
    t_pre *= (a.input_len / Lp) ** 2
    t_job = a.layers * (t_pre * B + t_dec * a.output_len)
    return {
        "value": round(B * a.output_len / t_job, 3), "unit": "output tok/s (attention path only)",
        "cores": cores, "torch_num_threads": threads, "kind": kind,
        "sample": f"1 layer: median of {n_dec} decode steps (KV write + paged_attention_v1, {Bs} of "
                  f"the {B} sequences, kv {kv_len}; scaled by {B}/{Bs}) + median of {n_pre} SDPA "
                  f"prefills of 1 prompt of {Lp}"
                  + (f" (scaled by ({a.input_len}/{Lp})^2)" if Lp != a.input_len else "")
                  + f", each after one untimed call; scaled to {a.layers} layers x ({B} prompts + "
                  f"{a.output_len} steps); no eviction logic (the reference's CPU backend has none)"}


def workload_name(a, wl) -> str:
    tp = f", one rank's heads of TP={a.emulate_tp}" if a.emulate_tp > 1 else ""
    return (f"{a.model} attention geometry (Hq {a.num_heads}, Hkv {a.num_kv_heads}, D "
            f"{a.head_size}, {a.layers} layers{tp}), block {a.block_size}, {a.num_prompts} "
            f"requests {a.input_len}->{a.output_len}, "
            + ((f"percentage prune type, evict-freq {a.evict_freq}, num-block-merge "
                f"{a.num_block_merge} ({wl.budget} prompt tokens kept)" if wl.percentage else
                f"{'budget ' + str(wl.budget)} ({a.evict:.0%} eviction)")
               + f", prompt {a.prompt_evict_method} / decode {a.decode_evict_method}"
               if wl.cfg is not None else "no eviction"))


def kernel_source_hash(files=("paged_decode.hip", "paged_decode_kernel.h", "pe_common.h")) -> str:
    """SHA-256 over a kernel's sources (default: the decode kernel's): a stored counter figure is
    only reused for the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "vllm-paged-compression_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


PREFILL_SOURCES = ("prefill.hip", "pe_common.h")


def stored_prefill_busy(flops_per_launch: float):
    """Matrix-pipe busy fraction of the prefill kernel (SQ_VALU_MFMA_BUSY_CYCLES over SIMD cycles)
    from the counter passes kept under profiles/, for the tracked launch shape with the same flop
    count, measured on the same kernel source; otherwise None."""
    import glob
    src = kernel_source_hash(PREFILL_SOURCES)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_prefill_pmc.json")), reverse=True):
        try:
            pmc = json.load(open(path))
            if pmc.get("kernel_source_sha256") != src:
                continue
            for shape in pmc["shapes"].values():
                f = shape["tflops"] * 1e12 * shape["us_per_launch"] * 1e-6
                if abs(f / flops_per_launch - 1.0) < 0.02:
                    return shape["mfma_pipe_busy_frac"], os.path.basename(path)
        except Exception:
            continue
    return None, None


def stored_traffic(workload: str, avg_bytes: float):
    """HBM bytes per decode launch from the PMC passes kept under profiles/ (FETCH_SIZE x2 per
    the gfx950 correction + WRITE_SIZE, separate runs of this same bench command), as a ratio to
    the algorithmic bytes.  Used only when workload, byte count and kernel source all match the
    run the counters were taken on; otherwise traffic is reported as null."""
    import glob
    src = kernel_source_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_decode_pmc*.json")), reverse=True):
        try:
            pmc = json.load(open(path))
            if pmc.get("workload") == workload and pmc.get("kernel_source_sha256") == src and \
                    abs(avg_bytes / pmc["algorithmic_bytes_per_launch"] - 1.0) < 0.05:
                return int(avg_bytes * pmc["traffic_over_algorithmic"]), os.path.basename(path)
        except Exception:
            continue
    return None, None


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes
    (torch.distributed.run) before this process has made any GPU call, and pass their exit code
    on."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           *sys.argv[1:]]
    return subprocess.call(cmd)


def wl_comm_lost(dist, dev) -> bool:
    """Whether the peer-buffer exchange of any rank gave up a wait (its outputs were NaN): such a
    run must not pass for a measurement."""
    from vllm_paged_compression_amd import parallel
    comm = parallel._one_shot.get(id(dist.group.WORLD))
    lost = 0
    if comm is not None and comm._own is not None:
        lost = comm.timeouts()
    t = torch.tensor([lost], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item()) > 0


def timed_generations(wl, n, sync, dist, dev, world):
    t0 = time.perf_counter()
    for _ in range(n):
        wl.generate()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def main():
    a = parse()
    if os.environ.get("PE_BENCH_HANG_DUMP"):  # stacks of every thread if the run is still going then
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PE_BENCH_HANG_DUMP"]), exit=True)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # PE_BENCH_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than
    # ranks (ranks share devices, collectives go through the host); the measured runs use RCCL
    backend = os.environ.get("PE_BENCH_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, {torch.cuda.device_count()} visible "
                         "(PE_BENCH_DIST_BACKEND=gloo rehearses with shared devices)")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    devices = [dev_index]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        gathered = [None] * world
        dist.all_gather_object(gathered, dev_index)
        devices = gathered
    n_gpus = world
    tp = a.mode in ("tp", "tp-strong") and world > 1
    base_prompts, base_prefill = a.num_prompts, a.prefill_batch
    if tp and a.mode == "tp":
        # weak scaling under tensor parallelism: N times the requests over 1/N of the heads
        a.num_prompts *= world
        a.prefill_batch *= world
    import vllm_paged_compression_amd as pkg
    pkg.load_library()
    tp_size, tp_rank, tp_group = (world, rank, dist.group.WORLD) if tp else (1, 0, None)
    if a.emulate_tp > 1:
        if world > 1:
            raise SystemExit("--emulate-tp is a single-GPU view of one rank of a TP deployment")
        tp_size, tp_rank = a.emulate_tp, 0
    wl = Workload(a, dev, tp_size, tp_rank, tp_group)

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
    dt = timed_generations(wl, a.steps, sync, dist, dev, world)
    wl.timing_on = False
    replicas = 1 if (tp or world == 1) else world
    total_tokens = a.num_prompts * a.output_len * a.steps * replicas
    value = total_tokens / dt

    # decode kernel: measured launch durations -> achieved algorithmic bandwidth
    durs = [s.elapsed_time(e) * 1e-3 / n for s, e, _, n in wl.timed]   # seconds per launch
    byts = [b for _, _, b, _ in wl.timed]
    roof = None
    name = workload_name(a, wl)
    if durs:
        avg_t = sum(durs) / len(durs)
        avg_b = sum(byts) / len(byts)
        ach = avg_b / avg_t / 1e9
        traffic, traffic_src = stored_traffic(name, avg_b) if world == 1 else (None, None)
        roof = {"bound": "hbm", "kernel": "paged_decode_mfma_kernel", "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_us": round(avg_t * 1e6, 2),
                "algorithmic_bytes_per_launch": int(avg_b),
                "launches_timed": sum(n for _, _, _, n in wl.timed),
                "includes": "fused KV write of the step's token (pe_decode_step)"
                            + (" and the rotary embedding of its query/key rows (pe_decode_step_rope)"
                               if a.rope == "fused" else "")}
    roof_p = None
    if wl.timed_prefill:
        pd = [s.elapsed_time(e) * 1e-3 for s, e, _ in wl.timed_prefill]
        pf = [f for _, _, f in wl.timed_prefill]
        ach = sum(pf) / sum(pd) / 1e12
        busy, busy_src = stored_prefill_busy(sum(pf) / len(pf)) if world == 1 else (None, None)
        roof_p = {"bound": "mfma", "kernel": "prefill_mfma_kernel", "achieved": round(ach, 1),
                  "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                  "traffic": None, "mfma_busy": busy, "mfma_busy_source": busy_src,
                  "avg_launch_us": round(sum(pd) / len(pd) * 1e6, 2),
                  "flops_per_launch": int(sum(pf) / len(pf)), "launches_timed": len(pd)}
    es = 2
    blk_bytes = 2 * a.block_size * a.num_kv_heads * a.head_size * es * a.layers // max(a.emulate_tp, 1)
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
        "dtype": a.dtype,
        "data": "synthetic",
        "config": {
            "workload": name, "preset": a.config,
            "step": "one full generation of the batch through prefill + decode attention of all layers",
            "parallelism": (f"tp{world}" if tp else f"dp{world}") if world > 1 else "single",
            "world_size": dist.get_world_size() if world > 1 else 1,
            "dist_backend": (dist.get_backend() if world > 1 else None),
            "devices": devices,
            "oproj_allreduce": bool(wl.allreduce), "rope": a.rope,
            "reference_head_compat": bool(a.reference_head_compat),
            "graph_decode": bool(wl.use_graph), "graph_replays": wl.graph_replays,
            "graph_evict_replays": wl.graph_evict_replays, **kv},
    }
    if os.environ.get("PE_BENCH_KV_CHECKSUM"):
        # bit-level sum of every layer's KV cache after the last generation (A/B of launch modes)
        out["config"]["kv_checksum"] = int(sum(int(t.view(torch.int16).to(torch.int64).sum().item())
                                               for t in wl.kv))
    main_budget = wl.budget
    del wl
    torch.cuda.empty_cache()
    if tp and a.mode == "tp":
        # the same heads-sharded job with the single-GPU request count (what the reference's
        # --tensor-parallel-size N run of a fixed request set measures): strong scaling
        import copy
        a3 = copy.copy(a)
        a3.num_prompts, a3.prefill_batch = base_prompts, base_prefill
        wl3 = Workload(a3, dev, world, rank, dist.group.WORLD)
        wl3.generate()
        sync()
        n3 = max(1, min(a.steps, 2))
        dt3 = timed_generations(wl3, n3, sync, dist, dev, world)
        out["config"]["tp_strong"] = {
            "requests": base_prompts, "steps": n3, "scaling": "strong",
            "value": round(base_prompts * a.output_len * n3 / dt3, 2),
            "ms_per_step": round(dt3 / n3 * 1e3, 3)}
        del wl3
        torch.cuda.empty_cache()
    exit_code = 0
    if tp and not a.oproj_allreduce:
        # the exchange that follows attention in the model: the same weak-scaling job with the
        # all-reduce of a [tokens, hidden] o_proj partial after every layer (the GEMM itself is
        # outside the path and not run).  Reported beside the attention-only value, never as it.
        import copy
        from vllm_paged_compression_amd import parallel
        a4 = copy.copy(a)
        a4.oproj_allreduce = True
        leg = {"path_decode": None, "path_prefill": None, "ar_timeouts": None}
        try:
            wl4 = Workload(a4, dev, world, rank, dist.group.WORLD)
            comm = parallel._one_shot.get(id(dist.group.WORLD))
            es4 = 2
            msg_d = a4.num_prompts * wl4.hidden * es4
            msg_p = a4.prefill_batch * a4.input_len * wl4.hidden * es4
            leg.update(decode_message_bytes=msg_d, prefill_message_bytes=msg_p,
                       path_decode=comm.path(msg_d) if comm is not None else "backend",
                       path_prefill=comm.path(msg_p) if comm is not None else "backend",
                       peer_buffers=("off (PE_AR_DISABLE)" if comm is None else
                                     "on" if comm.enabled else f"off: {comm.disabled_reason}"))
            wl4.generate()
            sync()
            n4 = max(1, min(a.steps, 2))
            dt4 = timed_generations(wl4, n4, sync, dist, dev, world)
            lost = 0
            if comm is not None and comm._own is not None:
                lost = comm.timeouts()
            t = torch.tensor([lost], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            leg.update(value=round(a4.num_prompts * a.output_len * n4 / dt4, 2), steps=n4,
                       ms_per_step=round(dt4 / n4 * 1e3, 3), ar_timeouts=int(t.item()))
            if leg["ar_timeouts"] > 0:
                exit_code = 3
            del wl4
            torch.cuda.empty_cache()
        except Exception as e:          # reported on the line; the attention-only value stands
            leg["error"] = repr(e)[:300]
            exit_code = 3
        out["config"]["with_oproj_allreduce"] = leg
    if a.alt_evict > 0 and a.cache_budget == 0 and a.cache_prune_type == "budget" \
            and abs(a.alt_evict - a.evict) > 1e-9:
        # the same job at the metric's other eviction fraction (outside the timed region above)
        import copy
        a2 = copy.copy(a)
        a2.evict = a.alt_evict
        wl2 = Workload(a2, dev, tp_size, tp_rank, tp_group)
        wl2.generate()
        sync()
        n2 = max(1, min(a.steps, 2))
        dt2 = timed_generations(wl2, n2, sync, dist, dev, world)
        held2 = wl2.kv_blocks_peak * blk_bytes / 2 ** 30
        out["config"]["alt"] = {
            "evict": a.alt_evict, "budget": wl2.budget, "steps": n2,
            "value": round(a.num_prompts * a.output_len * n2 * replicas / dt2, 2),
            "ms_per_step": round(dt2 / n2 * 1e3, 3), "kv_gb_held_peak": round(held2, 3),
            "kv_gb_saved": round(kv["kv_gb_without_eviction"] - held2, 3)}
        del wl2
    if roof is not None:
        out["roofline"] = roof
    if roof_p is not None:
        out["roofline_prefill"] = roof_p
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(a, main_budget)
        except Exception as e:  # the baseline never decides the run
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if world > 1 and wl_comm_lost(dist, dev):
        exit_code = 3
        out["config"]["ar_timeouts_main"] = True
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        from vllm_paged_compression_amd import parallel
        try:
            parallel.disable_one_shot_all_reduce(dist.group.WORLD)
        except parallel.AllReduceError as e:
            print(f"[bench] {e}", file=sys.stderr, flush=True)
            exit_code = 3
        dist.destroy_process_group()
    if exit_code:
        raise SystemExit(exit_code)


if __name__ == "__main__":
    main()
