"""Generates the golden vectors under tests/golden/ by running the *reference's own
code* in the build container:

* ``vllm/attention/kvcache_prunner.py`` and ``vllm/attention/block_l2norm_manager.py``
  are loaded straight from the reference checkout (they import only torch/stdlib);
* the reference's CPU kernels (``csrc/cpu/cache.cpp``, ``csrc/cpu/attention.cpp``,
  ``csrc/cpu/pos_encoding.cpp``) are used through ``oracle/_ref/libpe_ref.so`` (built by ``oracle/build_ref.sh``).

Only data (inputs and expected outputs) is written; no reference source travels.
Run:  python tests/golden/make_golden.py  [--reference /root/reference]
"""
import argparse
import importlib.util
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def bits(t: torch.Tensor) -> np.ndarray:
    """Raw storage of a tensor as a numpy array (bf16/fp16 -> uint16)."""
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()
    return t.contiguous().numpy().copy()


def tagged_kv(L, H, D, dtype):
    """K/V whose every (token, head) row is identifiable: element 0 = token index,
    element 1 = head index (exact in bf16 for the sizes used via two-part split)."""
    k = torch.zeros(L, H, D, dtype=torch.float32)
    t = torch.arange(L, dtype=torch.float32)
    k[:, :, 0] = (t // 64)[:, None]
    k[:, :, 1] = (t % 64)[:, None]
    k[:, :, 2] = torch.arange(H, dtype=torch.float32)[None, :]
    return k.to(dtype), (k + 0).to(dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    ref = args.reference
    kp = load("ref_kvcache_prunner", f"{ref}/vllm/attention/kvcache_prunner.py")
    bm = load("ref_block_l2norm_manager", f"{ref}/vllm/attention/block_l2norm_manager.py")

    # ---- (ii) get_pruned_length table + streamingLLM kept-token maps -------------
    table = []
    for bs in (8, 16, 32):
        for budget in (3 * bs, 4 * bs, 128, 256, 512, 768, 4096):
            if budget < 3 * bs or budget % bs:
                continue
            for method in ("streamingLLM", "streamingLLM-1", "inverse_key_l2", "local", "global"):
                p = kp.KVCachePruner("budget", method, bs, budget, 1)
                for L in (1, bs, 3 * bs, budget - 1, budget, budget + 1, budget + bs, budget + bs + 3,
                          1024, 1030, 2 * budget + 5, 8192):
                    table.append([bs, budget, method, L, int(p.get_pruned_length(L))])
    keep = []
    for (L, budget, bs) in [(80, 48, 16), (100, 48, 16), (1024, 768, 16), (1024, 512, 16),
                            (1030, 512, 16), (1030, 768, 16), (512, 512, 16), (513, 512, 16),
                            (200, 96, 32), (77, 24, 8), (64, 48, 16), (70, 48, 16)]:
        p = kp.KVCachePruner("budget", "streamingLLM", bs, budget, 1)
        k, v = tagged_kv(L, 2, 8, torch.float32)
        pk, pv = p.prune_prompt(k, v)
        src = (pk[:, 0, 0] * 64 + pk[:, 0, 1]).to(torch.int64).tolist()
        ok = torch.empty(p.get_pruned_length(L), 2, 8)
        ov = torch.empty_like(ok)
        p.prune_prompt_inplace(k, v, ok, ov)
        assert torch.equal(ok, pk) and torch.equal(ov, pv)
        keep.append({"L": L, "budget": budget, "bs": bs, "kept": src})
    with open(os.path.join(HERE, "prune_tables.json"), "w") as f:
        json.dump({"pruned_length": table, "streaming_keep": keep}, f)

    # ---- (i) token-level prompt prune (local/global/inverse_key_l2), tie-free f32 --
    out = {}
    case = 0
    for method in ("global", "local", "inverse_key_l2"):
        for (L, budget, bs, H, D) in [(80, 48, 16, 4, 16), (100, 48, 16, 2, 16), (150, 64, 16, 3, 8)]:
            for seed in (0, 1):
                g = torch.Generator().manual_seed(1000 * case + seed)
                k = torch.rand(L, H, D, generator=g) * 2 - 1
                v = torch.rand(L, H, D, generator=g) * 2 - 1
                p = kp.KVCachePruner("budget", method, bs, budget, 1)
                pk, pv = p.prune_prompt(k, v)
                name = f"c{case}"
                out[name + "_meta"] = np.array([L, budget, bs, H, D, ["global", "local", "inverse_key_l2"].index(method)])
                out[name + "_k"], out[name + "_v"] = bits(k), bits(v)
                out[name + "_pk"], out[name + "_pv"] = bits(pk), bits(pv)
                case += 1
    np.savez_compressed(os.path.join(HERE, "prompt_prune_topk.npz"), **out)

    # ---- (i-b) the same at the benchmark's sizes: [L, 8, 128] and [L, 12, 64] ---------------
    # The inputs are regenerated from their seed; stored are the SHA-256 of the reference's
    # outputs and, per head, the bitmap of middle tokens it kept.  f32 inputs are tie-free and
    # must match exactly; 16-bit inputs have tied scores (torch.topk's choice among ties is
    # not specified), so for them the tests compare the kept scores, not the kept indices.
    import hashlib
    out = {}
    case = 0
    for method in ("global", "inverse_key_l2"):
        for (L, budget, H, D, dtype) in [(1024, 768, 8, 128, torch.float32), (1030, 512, 8, 128, torch.float32),
                                         (1024, 512, 12, 64, torch.float32),
                                         (1024, 768, 8, 128, torch.bfloat16), (1030, 512, 12, 64, torch.float16)]:
            bs = 16
            g = torch.Generator().manual_seed(5000 + case)
            k = (torch.rand(L, H, D, generator=g) * 2 - 1).to(dtype)
            v = (torch.rand(L, H, D, generator=g) * 2 - 1).to(dtype)
            p = kp.KVCachePruner("budget", method, bs, budget, 1)
            pk, pv = p.prune_prompt(k, v)
            assert pk.shape[0] == p.get_pruned_length(L)
            # kept bitmap of the middle, recovered by matching output rows to input rows
            rem = L % bs
            end_mid = L - bs - rem
            mid_k = k[bs:end_mid]
            keyed = {}
            for t in range(mid_k.shape[0]):
                for h in range(H):
                    keyed.setdefault(bits(mid_k[t, h]).tobytes(), []).append((t, h))
            kept = np.zeros((mid_k.shape[0], H), dtype=np.uint8)
            out_mid = pk[bs:pk.shape[0] - bs - rem].reshape(-1, D)
            for r in range(out_mid.shape[0]):
                t, h = keyed[bits(out_mid[r]).tobytes()].pop(0)
                kept[t, h] = 1
            out[f"c{case}_meta"] = np.array([L, budget, bs, H, D, ["global", "local", "inverse_key_l2"].index(method),
                                             {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dtype], 5000 + case])
            out[f"c{case}_kept"] = np.packbits(kept, axis=0)
            out[f"c{case}_sha_k"] = np.frombuffer(hashlib.sha256(bits(pk).tobytes()).digest(), dtype=np.uint8)
            out[f"c{case}_sha_v"] = np.frombuffer(hashlib.sha256(bits(pv).tobytes()).digest(), dtype=np.uint8)
            case += 1
    np.savez_compressed(os.path.join(HERE, "prompt_prune_topk_full.npz"), **out)

    # ---- (iii) block scores ---------------------------------------------------------
    out = {}
    case = 0
    p = kp.KVCachePruner("budget", "global", 16, 512, 1)
    for dtype, H, D, bs in [(torch.bfloat16, 8, 128, 16), (torch.float16, 12, 64, 16),
                            (torch.float32, 8, 128, 16), (torch.bfloat16, 1, 128, 16),
                            (torch.bfloat16, 8, 128, 32), (torch.float16, 4, 64, 8)]:
        for seed in (0, 1, 2):
            g = torch.Generator().manual_seed(7 + 13 * case)
            scale = D ** -0.5
            v = ((torch.rand(bs, H, D, generator=g) * 2 - 1) * scale).to(dtype)
            s = p.get_block_score(v, v, "global")
            out[f"c{case}_v"] = bits(v)
            out[f"c{case}_score"] = bits(s.reshape(1))
            out[f"c{case}_meta"] = np.array([bs, H, D, {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dtype]])
            case += 1
    np.savez_compressed(os.path.join(HERE, "block_scores.npz"), **out)

    # ---- (iii-b) the other score formulas of global/local (sub_evict_method, :29-66) ------
    # inputs are stored once; per formula the pruned prompt / the block score
    out = {}
    formulas = ["value_l2", "key_l2", "value_l2_plus_key_l2", "inverse_key_l2"]
    for ti, (L, budget, bs, H, D) in enumerate([(80, 48, 16, 4, 16), (150, 64, 16, 3, 8)]):
        g = torch.Generator().manual_seed(9000 + ti)
        k = torch.rand(L, H, D, generator=g) * 2 - 1      # tie-free f32
        v = torch.rand(L, H, D, generator=g) * 2 - 1
        out[f"t{ti}_meta"] = np.array([L, budget, bs, H, D])
        out[f"t{ti}_k"], out[f"t{ti}_v"] = bits(k), bits(v)
        for fi, formula in enumerate(formulas):
            pp = kp.KVCachePruner("budget", "global", bs, budget, 1)
            pp.sub_evict_method = formula
            pk, pv = pp.prune_prompt(k, v)
            out[f"t{ti}_f{fi}_pk"], out[f"t{ti}_f{fi}_pv"] = bits(pk), bits(pv)
    bi = 0
    for dtype, H, D, bs in [(torch.bfloat16, 8, 128, 16), (torch.float16, 12, 64, 16),
                            (torch.float32, 4, 64, 16), (torch.bfloat16, 2, 128, 32)]:
        for seed in (0, 1):
            g = torch.Generator().manual_seed(9500 + 7 * bi + seed)
            scale = D ** -0.5
            k = ((torch.rand(bs, H, D, generator=g) * 2 - 1) * scale * 3).to(dtype)
            v = ((torch.rand(bs, H, D, generator=g) * 2 - 1) * scale).to(dtype)
            out[f"b{bi}_k"], out[f"b{bi}_v"] = bits(k), bits(v)
            out[f"b{bi}_meta"] = np.array([bs, H, D, {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dtype]])
            for fi, formula in enumerate(formulas):
                p = kp.KVCachePruner("budget", "global", bs, 4 * bs, 1)
                p.sub_evict_method = formula
                out[f"b{bi}_f{fi}_score"] = bits(p.get_block_score(k, v, "global").reshape(1))
            bi += 1
    np.savez_compressed(os.path.join(HERE, "score_formulas.npz"), **out)

    # ---- (iv) BlockL2NormManager bookkeeping ----------------------------------------
    mgr = bm.BlockL2NormManager()
    steps = [
        {"r0": {0: [1, 2, 3]}, "r1": {1: [4, 5]}},
        {"r0": {0: [1, 2, 3]}, "r1": {1: [4, 5]}},
        {"r1": {1: [4, 5, 6]}, "r2": {2: [7]}},
        {"r2": {2: [7, 8]}, "r3": {3: [9, 10]}, "r1": {1: [4, 5, 6]}},
    ]
    log = []
    for st in steps:
        mgr.update_reqs(st)
        log.append({"input": {k: {str(s): b for s, b in v.items()} for k, v in st.items()},
                    "flat_sequence_map": [[r, s] for r, s in mgr.flat_sequence_map],
                    "norm_keys": sorted(mgr.reqid_mapping_block_l2norms.keys())})
    mgr.update_block_l2norms(0, [1.0, 2.0, 3.0])
    after_append = list(mgr.update_last_block_l2norm(0, 0.5))
    with open(os.path.join(HERE, "l2norm_manager.json"), "w") as f:
        json.dump({"steps": log, "after_append": after_append,
                   "seq0": list(mgr.get_reqid_and_seqid(0))}, f)

    # ---- (vii) "percentage" prune type (README-era flags) ----------------------------------
    # The code of this mode is kept in kvcache_prunner_old.py / page_evict_kv_util_bak.py.  The
    # old pruner's __init__ reads names that only its __main__ block defines, so the object is
    # made with __new__ and given the attributes its methods read; the util module imports two
    # names from the vllm package (not importable here), which are supplied as stubs - its
    # functions only read attributes of the config object they are handed.
    import sys
    import types
    from types import SimpleNamespace
    old = load("ref_kvcache_prunner_old", f"{ref}/vllm/attention/kvcache_prunner_old.py")
    stub_utils = types.ModuleType("vllm.utils")
    stub_utils.cdiv = lambda a, b: -(a // -b)
    stub_cfg = types.ModuleType("vllm.config")
    stub_cfg.PagedEvictConfig = object
    saved = {k: sys.modules.get(k) for k in ("vllm", "vllm.utils", "vllm.config")}
    sys.modules.update({"vllm": types.ModuleType("vllm"), "vllm.utils": stub_utils,
                        "vllm.config": stub_cfg})
    try:
        bak = load("ref_page_evict_kv_util_bak", f"{ref}/vllm/core/page_evict_kv_util_bak.py")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v

    def old_pruner(prompt_method, decode_method, bs, freq, ib, nbm):
        p = old.KVCachePruner.__new__(old.KVCachePruner)
        p.cache_prune_type = "percentage"
        p.prompt_evict_method, p.decode_evict_method = prompt_method, decode_method
        p.orig_block_size, p.evict_frequency = bs, freq
        p.initial_blocks, p.num_block_merge = ib, nbm
        return p

    pct = {"required_blocks": [], "streaming_keep": [], "decode_blocks": [], "util_blocks": []}
    for bs in (8, 16):
        for freq in (1, 2, 3, 4):
            for nbm in (2, 3):
                cfg = SimpleNamespace(initial_blocks=1, num_block_merge=nbm, original_block_size=bs,
                                      prompt_evict_method="streamingLLM", cache_prune_type="percentage",
                                      evict_freq=freq, decode_evict_method="streamingLLM")
                for L in (1, bs, (1 + nbm) * bs, (1 + nbm) * bs + 1, 5 * bs, 100, 1024, 1030, 8192):
                    nblk, ntok = bak.get_num_required_blocks_after_prune_promt(L, cfg)
                    pct["required_blocks"].append([bs, freq, nbm, L, int(nblk), int(ntok)])
    for (L, freq, bs, nbm) in [(80, 2, 16, 2), (100, 2, 16, 2), (100, 4, 16, 2), (1024, 2, 16, 2),
                               (1024, 4, 16, 2), (1030, 4, 16, 2), (1030, 3, 16, 3), (48, 2, 16, 2),
                               (49, 2, 16, 2), (200, 2, 8, 3), (77, 1, 8, 2)]:
        p = old_pruner("streamingLLM", "streamingLLM", bs, freq, 1, nbm)
        k, v = tagged_kv(L, 2, 8, torch.float32)
        pk, pv = p.prune_prompt(k, v)
        assert torch.equal(pk, pv)
        src = (pk[:, 0, 0] * 64 + pk[:, 0, 1]).to(torch.int64).tolist()
        pct["streaming_keep"].append({"L": L, "freq": freq, "bs": bs, "nbm": nbm, "kept": src})
    for method in ("streamingLLM", "value_l2"):
        for nbm in (2, 3, 4):
            p = old_pruner("streamingLLM", method, 16, 2, 1, nbm)
            pct["decode_blocks"].append([method, nbm, *[int(x) for x in p.get_blocks_to_prune_and_merge_decode(4096)]])
            cfg = SimpleNamespace(initial_blocks=1, num_block_merge=nbm, original_block_size=16,
                                  cache_prune_type="percentage", decode_evict_method=method)
            pct["util_blocks"].append([method, nbm, *[int(x) for x in bak.get_blocks_to_prune(cfg, 4096)]])
    with open(os.path.join(HERE, "percentage_tables.json"), "w") as f:
        json.dump(pct, f)

    # token-level prompt prune of the same mode (value_l2 / inverse_key_l2), tie-free f32 inputs
    out = {}
    case = 0
    for mi, method in enumerate(("value_l2", "inverse_key_l2")):
        for (L, freq, bs, nbm, H, D) in [(100, 2, 16, 2, 4, 16), (150, 4, 16, 2, 2, 16), (130, 3, 16, 3, 3, 8),
                                         (200, 2, 8, 2, 2, 8)]:
            g = torch.Generator().manual_seed(7000 + case)
            k = torch.rand(L, H, D, generator=g) * 2 - 1
            v = torch.rand(L, H, D, generator=g) * 2 - 1
            p = old_pruner(method, "streamingLLM", bs, freq, 1, nbm)
            pk, pv = p.prune_prompt(k, v)
            out[f"c{case}_meta"] = np.array([L, freq, bs, nbm, H, D, mi])
            out[f"c{case}_k"], out[f"c{case}_v"] = bits(k), bits(v)
            out[f"c{case}_pk"], out[f"c{case}_pv"] = bits(pk), bits(pv)
            case += 1
    np.savez_compressed(os.path.join(HERE, "percentage_prompt_topk.npz"), **out)

    # decode-time merge (prune_oldest_block): tie-free f32 inputs, the reference's output rows
    # (kept (token, head) rows re-packed across heads, keys also returned as values - both are
    # properties of the reference's code, kvcache_prunner_old.py:312-313)
    out = {}
    case = 0
    for (nbm, bs, H, D) in [(2, 16, 4, 16), (3, 16, 2, 32), (2, 8, 3, 8), (4, 16, 2, 128)]:
        for seed in (0, 1):
            g = torch.Generator().manual_seed(9000 + 10 * case + seed)
            T = nbm * bs
            k = torch.rand(T, H, D, generator=g) * 2 - 1
            v = torch.rand(T, H, D, generator=g) * 2 - 1
            p = old_pruner("streamingLLM", "value_l2", bs, 2, 1, nbm)
            _, _, prune_tokens = p.get_blocks_to_prune_and_merge_decode(4096)
            pk, pv = p.prune_oldest_block(k, v, prune_tokens)
            out[f"c{case}_meta"] = np.array([nbm, bs, H, D, prune_tokens])
            out[f"c{case}_k"], out[f"c{case}_v"] = bits(k), bits(v)
            out[f"c{case}_pk"], out[f"c{case}_pv"] = bits(pk), bits(pv)
            case += 1
    np.savez_compressed(os.path.join(HERE, "decode_merge.npz"), **out)

    # ---- (vi) reference CPU kernels: KV write + paged attention ----------------------
    lib = os.path.join(ROOT, "oracle", "_ref", "libpe_ref.so")
    torch.ops.load_library(lib)
    out = {}
    case = 0
    for dtype, Hq, Hkv, D in [(torch.bfloat16, 8, 2, 64), (torch.float32, 4, 2, 64), (torch.bfloat16, 4, 1, 128)]:
        g = torch.Generator().manual_seed(100 + case)
        nb, bs = 12, 16
        x = 16 // torch.tensor([], dtype=dtype).element_size()
        scale = D ** -0.5
        T = nb * bs
        k = ((torch.rand(T + 1, Hkv, D, generator=g) * 2 - 1) * scale).to(dtype)
        v = ((torch.rand(T + 1, Hkv, D, generator=g) * 2 - 1) * scale).to(dtype)
        slots = torch.randperm(nb * bs + 1, generator=g).to(torch.int64)
        slots[slots == nb * bs] = -1  # one padding token, every real slot written once
        kc = torch.zeros(nb, Hkv, D // x, bs, x, dtype=dtype)
        vc = torch.zeros(nb, Hkv, D, bs, dtype=dtype)
        torch.ops.pe_ref.reshape_and_cache(k, v, kc, vc, slots, "auto", 1.0, 1.0)
        S = 4
        q = ((torch.rand(S, Hq, D, generator=g) * 2 - 1) * scale).to(dtype)
        lens = torch.tensor([1, 16, 37, 90], dtype=torch.int32)
        M = 6
        bt = torch.randint(0, nb, (S, M), generator=g, dtype=torch.int32)
        o = torch.zeros_like(q)
        torch.ops.pe_ref.paged_attention_v1(o, q, kc, vc, Hkv, scale, bt, lens, bs, 90, None, "auto",
                                            1.0, 1.0, 0, 0, 0, 64, 0)
        out[f"c{case}_meta"] = np.array([Hq, Hkv, D, bs, nb, {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dtype]])
        for n, t in [("k", k), ("v", v), ("slots", slots), ("kc_out", kc), ("vc_out", vc), ("q", q),
                     ("lens", lens), ("bt", bt), ("o", o)]:
            out[f"c{case}_{n}"] = bits(t)
        case += 1
    np.savez_compressed(os.path.join(HERE, "ref_cpu_kernels.npz"), **out)

    # ---- (vii) reference CPU rotary embedding (csrc/cpu/pos_encoding.cpp) -------------
    # cos/sin table as vllm/model_executor/layers/rotary_embedding.py:_compute_cos_sin_cache
    # builds it (base 10000); query and key are column slices of one fused qkv buffer, as in
    # models/llama.py:202-205
    out = {}
    case = 0
    DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
    for dtype, Hq, Hkv, D, rot, neox in [(torch.bfloat16, 4, 2, 128, 128, True),
                                         (torch.bfloat16, 4, 2, 128, 128, False),
                                         (torch.float16, 6, 6, 64, 64, True),
                                         (torch.bfloat16, 4, 1, 128, 64, True),
                                         (torch.float32, 2, 2, 64, 32, False)]:
        g = torch.Generator().manual_seed(300 + case)
        T, max_pos = 11, 64
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, dtype=torch.float) / rot))
        freqs = torch.einsum("i,j -> ij", torch.arange(max_pos, dtype=torch.float), inv_freq)
        cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype)
        qkv = ((torch.rand(T, (Hq + 2 * Hkv) * D, generator=g) * 2 - 1) * 2).to(dtype)
        pos = torch.randint(0, max_pos, (T,), generator=g, dtype=torch.int64)
        q, k = qkv[:, :Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D]
        q_in, k_in = q.clone(), k.clone()
        torch.ops.pe_ref.rotary_embedding(pos, q, k, D, cache, neox)
        out[f"c{case}_meta"] = np.array([Hq, Hkv, D, rot, int(neox), DT[dtype]])
        for n, t in [("pos", pos), ("cache", cache), ("q_in", q_in), ("k_in", k_in),
                     ("q_out", q.contiguous()), ("k_out", k.contiguous())]:
            out[f"c{case}_{n}"] = bits(t)
        case += 1
    np.savez_compressed(os.path.join(HERE, "ref_cpu_rope.npz"), **out)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
