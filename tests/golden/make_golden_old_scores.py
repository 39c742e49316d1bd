"""Golden vectors for the token scores of the README-era pruner: runs get_score of the REFERENCE's
own class (vllm/attention/kvcache_prunner_old.py:77-134; its __init__ reads names only its
__main__ block defines, so the object is made with __new__) for every method name it knows, on
random [T, H, D] tensors in bf16, fp16 and f32, and its prune_prompt / prune_oldest_block for a few
of them on tie-free f32 inputs.  Only data is written.
Run:  python tests/golden/make_golden_old_scores.py  [--reference /root/reference]"""
import argparse
import os

import numpy as np
import torch

from make_golden import HERE, bits, load

METHODS = ["cosine", "key_l1", "key_l2", "value_l1", "value_l2", "key_l1_div_value_l1",
           "key_l1_div_value_l2", "key_l2_div_value_l1", "key_l2_div_value_l2", "value_l1_div_key_l1",
           "value_l1_div_key_l2", "value_l2_div_key_l1", "value_l2_div_key_l2", "value_l1_plus_key_l1",
           "value_l2_plus_key_l2", "streamingLLM", "inverse_key_l1", "inverse_key_l2"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ref = ap.parse_args().reference
    old = load("ref_kvcache_prunner_old", f"{ref}/vllm/attention/kvcache_prunner_old.py")

    def pruner(prompt_method, decode_method, bs, freq, nbm):
        p = old.KVCachePruner.__new__(old.KVCachePruner)
        p.cache_prune_type = "percentage"
        p.prompt_evict_method, p.decode_evict_method = prompt_method, decode_method
        p.orig_block_size, p.evict_frequency = bs, freq
        p.initial_blocks, p.num_block_merge = 1, nbm
        return p

    out = {"methods": np.array(METHODS)}
    p = pruner("value_l2", "value_l2", 16, 2, 2)
    case = 0
    for code, dt in ((2, torch.bfloat16), (1, torch.float16), (0, torch.float32)):
        for (T, H, D) in ((24, 2, 64), (9, 2, 128)):
            g = torch.Generator().manual_seed(4100 + case)
            k = (torch.rand(T, H, D, generator=g) * 2 - 1).to(dt)
            v = (torch.rand(T, H, D, generator=g) * 4 - 2).to(dt)
            out[f"s{case}_meta"] = np.array([T, H, D, code])
            out[f"s{case}_k"], out[f"s{case}_v"] = bits(k), bits(v)
            for mi, m in enumerate(METHODS):
                sc = p.get_score(k, v, m)
                if m == "streamingLLM":
                    sc = sc.to(torch.float32)
                out[f"s{case}_m{mi}"] = bits(sc.contiguous())
            case += 1
    out["n_score_cases"] = np.array(case)
    # prompt prune (kvcache_prunner_old.py:201-262) and decode merge (:297-315) under some of the
    # scores, tie-free f32 inputs
    case = 0
    for m in ("cosine", "key_l1", "value_l1_div_key_l2", "value_l1_plus_key_l1", "inverse_key_l1"):
        for (L, freq, bs, nbm, H, D) in ((100, 2, 16, 2, 4, 16), (130, 3, 16, 3, 3, 8)):
            g = torch.Generator().manual_seed(5100 + case)
            k = torch.rand(L, H, D, generator=g) * 2 - 1
            v = torch.rand(L, H, D, generator=g) * 2 - 1
            pk, pv = pruner(m, "streamingLLM", bs, freq, nbm).prune_prompt(k, v)
            out[f"p{case}_meta"] = np.array([L, freq, bs, nbm, H, D, METHODS.index(m)])
            out[f"p{case}_k"], out[f"p{case}_v"] = bits(k), bits(v)
            out[f"p{case}_pk"], out[f"p{case}_pv"] = bits(pk), bits(pv)
            case += 1
    out["n_prune_cases"] = np.array(case)
    case = 0
    for m in ("cosine", "key_l2_div_value_l1", "value_l1"):
        for (nbm, bs, H, D) in ((2, 16, 4, 16), (3, 16, 2, 32)):
            g = torch.Generator().manual_seed(6100 + case)
            T = nbm * bs
            k = torch.rand(T, H, D, generator=g) * 2 - 1
            v = torch.rand(T, H, D, generator=g) * 2 - 1
            pr = pruner("streamingLLM", m, bs, 2, nbm)
            _, _, prune_tokens = pr.get_blocks_to_prune_and_merge_decode(4096)
            pk, _ = pr.prune_oldest_block(k, v, prune_tokens)
            out[f"d{case}_meta"] = np.array([nbm, bs, H, D, prune_tokens, METHODS.index(m)])
            out[f"d{case}_k"], out[f"d{case}_v"], out[f"d{case}_pk"] = bits(k), bits(v), bits(pk)
            case += 1
    out["n_merge_cases"] = np.array(case)
    np.savez_compressed(os.path.join(HERE, "old_token_scores.npz"), **out)
    print("wrote old_token_scores.npz")


if __name__ == "__main__":
    main()
