"""The backend against end-to-end traces of the reference's OWN backend code
(tests/golden/make_golden_backend.py: TorchCUDASDPABackendImpl.forward and everything it calls,
run on CPU over the reference's CPU kernels, block bookkeeping by the reference's BlockTable).

Our side runs the real scheduler -> metadata builder -> PagedEvictImpl.forward on the GPU with
the switches that reproduce HEAD (config.CompatFlags.reference_head(); ``random_victim`` as the
trace's ``enable_random_evict``) on the same inputs (regenerated from the trace's seeds).  Every
step must give the reference's block tables (ids included), slot mapping, kv lengths and victims,
the KV cache of every layer byte for byte (SHA-256 of the whole cache), and attention outputs
within tolerance (the reference's numbers come from CPU kernels: bf16 SDPA for prefill, a paged
attention with f32 probabilities and no epsilon for decode - stated tolerances: prefill atol 2e-2
rtol 1.6e-2, decode atol 4e-3 rtol 1.6e-2, atol relative to the tensor's largest magnitude)."""
import hashlib
import importlib.util
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN, from_bits

pytestmark = pytest.mark.gpu
BS = 16


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_backend",
                                                  os.path.join(GOLDEN, "make_golden_backend.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _sha(t):
    return hashlib.sha256(t.contiguous().view(torch.int16).cpu().numpy().tobytes()).hexdigest()


def _close(got, ref, atol, what):
    """atol is relative to the largest magnitude of the reference tensor (the prompts' value rows
    span a 36x range of norms, see prompt_inputs), rtol 1.6e-2."""
    scale = max(1.0, float(ref.abs().max()))
    torch.testing.assert_close(got / scale, ref / scale, atol=atol, rtol=1.6e-2,
                               msg=lambda m: f"{what}: {m}")


TRACES = json.load(open(os.path.join(GOLDEN, "backend_traces.json")))


@pytest.mark.parametrize("trace", TRACES, ids=[t["name"] for t in TRACES])
def test_backend_reproduces_reference_forward(trace):
    from vllm_paged_compression_amd import ops
    from vllm_paged_compression_amd.backend import PagedEvictBackend
    from vllm_paged_compression_amd.block_table import EvictBlockSpaceManager, EvictionScheduler
    from vllm_paged_compression_amd.config import CacheConfigView, CompatFlags, PagedEvictConfig
    from vllm_paged_compression_amd.page_evict_kv_util import adjust_seq_kv_len
    gen = _gen()
    name, method, budget = trace["name"], trace["method"], trace["budget"]
    tag = next(("_" + g for g in ("llama", "full", "staged") if name.startswith(g + "_")), "")
    dt, code = {"bf16": (torch.bfloat16, 2), "fp16": (torch.float16, 1)}[trace["dtype"]]
    zp = np.load(os.path.join(GOLDEN, f"backend_traces{tag}_prefill.npz"))
    zd = np.load(os.path.join(GOLDEN, f"backend_traces{tag}_decode.npz"))
    Hq, Hkv, D, layers, nb = trace["Hq"], trace["Hkv"], trace["D"], trace["layers"], trace["num_blocks"]
    prompts, seed = trace["prompts"], trace["seed"]
    cfg = None                  # method None: paged eviction off, the full-cache baseline
    if method is not None:
        cfg = PagedEvictConfig(trace["disable_evict_prefill"], "budget", method, budget, 3)
        assert cfg.cache_budget == budget
    compat = CompatFlags.reference_head()
    compat.random_victim = bool(trace["random_evict"])
    cache_cfg = CacheConfigView(BS, cfg)
    impl_cls = PagedEvictBackend.get_impl_cls()
    impls = [impl_cls(Hq, D, D ** -0.5, Hkv, None, None, "auto", None, None, cache_config=cache_cfg,
                      prefix=f"model.layers.{i}.self_attn.attn", compat=compat) for i in range(layers)]
    kv = [torch.zeros(PagedEvictBackend.get_kv_cache_shape(nb, BS, Hkv, D), dtype=dt, device="cuda")
          for _ in range(layers)]
    sched = EvictionScheduler(EvictBlockSpaceManager(BS, nb, cfg, watermark=0.0))
    bm = sched.block_manager
    rids = [f"r{i}" for i in range(len(prompts))]

    def builder(inter):
        runner = SimpleNamespace(device=torch.device("cuda:0"), cache_config=cache_cfg, pin_memory=False)
        ib = SimpleNamespace(runner=runner, sliding_window=None, block_size=BS, chunked_prefill_enabled=False,
                             inter_data_list=inter)
        return PagedEvictBackend.make_metadata_builder(ib)

    # ---- prefill steps (a request is admitted, and its blocks allocated, at its step)
    step_rows = trace["prefill_row_step"]
    for gi, gp in enumerate(trace["prefills"]):
        idxs = gp["requests"]
        assert idxs == trace["groups"][gi]
        lens = [prompts[i] for i in idxs]
        for i in idxs:
            sched.add_request(rids[i], i, prompts[i])
        assert [list(bm.get_block_table(i)) for i in idxs] == gp["tables"]
        assert [bm.get_seq_kv_len(i) for i in idxs] == gp["seq_kv_lens"]
        inter = [SimpleNamespace(request_id=rids[i], is_prompt=True, block_tables={i: list(bm.get_block_table(i))},
                                 seq_kv_lens={i: bm.get_seq_kv_len(i)}, seq_ids=[i], input_tokens=[[0] * prompts[i]],
                                 orig_seq_lens=[prompts[i]], seq_lens=[prompts[i]], query_lens=[prompts[i]],
                                 context_lens=[0])
                 for i in idxs]
        meta = builder(inter).build(list(lens), list(lens), -1, len(idxs))
        assert meta.slot_mapping.cpu().tolist() == gp["slot_mapping"]
        pre = f"{name}_prefill" if gi == 0 else f"{name}_prefill{gi}"
        for li in range(layers):
            if trace["plain_prompts"]:
                q, k, v = gen.step_inputs(seed + li + 100 * gi, sum(lens), Hq, Hkv, D, dt)
            else:
                q, k, v = gen.prompt_inputs(seed + li + 100 * gi, lens, Hq, Hkv, D, dt)
            out = impls[li].forward(q.cuda(), k.cuda(), v.cuda(), kv[li], meta)
            assert _sha(kv[li]) == gp["cache_sha"][li], (name, "prefill cache", gi, li)
            if f"{pre}_out{li}" in zp.files:
                ref = from_bits(zp[f"{pre}_out{li}"], code).float()
                _close(out[::step_rows].cpu().float(), ref, 2e-2, f"{name} prefill {gi} layer {li}")
        assert impls[0].get_rmv_block_idx() == {}

    # ---- decode
    logical = {i: prompts[i] for g in trace["groups"] for i in g}
    active = [i for g in trace["groups"] for i in g]
    finish = {int(k): v for k, v in trace["finish"].items()}
    evictions = 0
    for step, gs in enumerate(trace["steps"]):
        for i in [i for i in active if finish.get(i) == step]:
            sched.finish_request(rids[i])
            active.remove(i)
        assert active == gs["requests"]
        tables, kv_lens = sched.schedule_decode([rids[i] for i in active])
        inter, adj = [], []
        for i in active:
            logical[i] += 1
            kvl = adjust_seq_kv_len(cfg, logical[i], kv_lens[i], BS)
            adj.append(kvl)
            inter.append(SimpleNamespace(request_id=rids[i], is_prompt=False, block_tables={i: list(tables[i])},
                                         seq_kv_lens={i: kvl}, seq_ids=[i], input_tokens=[[0]],
                                         orig_seq_lens=[logical[i]], seq_lens=[logical[i]], query_lens=[1],
                                         context_lens=[logical[i] - 1]))
        lens = [logical[i] for i in active]
        assert lens == gs["seq_lens"] and adj == gs["seq_kv_lens"], (name, step)
        assert [list(tables[i]) for i in active] == gs["tables"], (name, step)
        meta = builder(inter).build(list(lens), [1] * len(active), -1, len(active))
        assert meta.slot_mapping.cpu().tolist() == gs["slot_mapping"], (name, step)
        for li in range(layers):
            q, k, v = gen.step_inputs(seed + 1000 * (step + 1) + li, len(active), Hq, Hkv, D, dt)
            out = impls[li].forward(q.cuda(), k.cuda(), v.cuda(), kv[li], meta)
            got = {str(i): [r, int(b)] for i, (r, b) in impls[li].get_rmv_block_idx().items()}
            assert got == gs["rmv_by_layer"][li], (name, step, li)
            assert _sha(kv[li]) == gs["cache_sha"][li], (name, step, li)
            key = f"{name}_s{step}_out{li}"
            if key in zd.files:
                ref = from_bits(zd[key], code).float()
                _close(out.cpu().float(), ref, 4e-3, f"{name} step {step} layer {li}")
        rmv = impls[0].get_rmv_block_idx()
        evictions += len(rmv)
        sched.commit(rmv)        # by the reported request id, as the engine does
    assert evictions == sum(len(s["rmv"]) for s in trace["steps"])
    # HEAD's table forms (width M-1 for every row, the strided re-read) make the decode kernel clamp
    # lengths to what a row addresses: that flag is expected here, any other one is not
    flags = ops.device_diagnostics(reset=True)
    assert flags & ~ops.DIAG_DECODE_LEN_CLAMPED == 0, f"unexpected device diagnostics {flags:#x} in {name}"
