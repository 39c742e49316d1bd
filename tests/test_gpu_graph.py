"""A decode step (KV write + paged attention of every layer) captured into a HIP graph and
replayed with new metadata and inputs: the library calls allocate nothing and synchronise
nothing, so a serving engine can capture them.  The reference runs eager only
(--enforce-eager in every script; its graph-capture metadata lacks seq_kv_lens,
vllm/attention/backends/utils.py:316-336), so the check is replay == eager launch, bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,kv0", [(16, 200), (3, 1500)])
def test_decode_step_graph_replay_matches_eager(B, kv0):
    import vllm_paged_compression_amd as pkg
    import vllm_paged_compression_amd.ops as ops
    pkg.load_library()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(B)
    Hq, Hkv, D, bs, layers, steps = 8, 2, 128, 16, 4, 5
    M = (kv0 + steps) // bs + 2
    nb = B * M + 1

    def rnd(*shape):
        return torch.empty(*shape, dtype=torch.bfloat16, device=dev).uniform_(-0.1, 0.1, generator=g)

    kv_graph = [rnd(2, nb, bs * Hkv * D) for _ in range(layers)]
    kv_eager = [t.clone() for t in kv_graph]
    perm = torch.randperm(nb - 1, device=dev, generator=g)[:B * M].to(torch.int32).reshape(B, M)
    # static buffers the graph reads; their contents change between replays
    q_s, k_s, v_s = rnd(layers, B, Hq * D), rnd(layers, B, Hkv * D), rnd(layers, B, Hkv * D)
    out_s = torch.zeros(layers, B, Hq * D, dtype=torch.bfloat16, device=dev)
    bt_s = perm.clone()
    lens_s = torch.zeros(B, dtype=torch.int32, device=dev)
    slots_s = torch.zeros(B, dtype=torch.int64, device=dev)
    scale = D ** -0.5
    max_len = kv0 + steps + bs  # upper bound baked into the captured launch (sizes the splits)

    def set_step(step):
        kv = kv0 + step + torch.arange(B, device=dev) % 3        # ragged lengths
        lens_s.copy_(kv.to(torch.int32))
        last = bt_s.gather(1, ((kv - 1) // bs).unsqueeze(1)).squeeze(1).to(torch.int64)
        slots_s.copy_(last * bs + (kv - 1) % bs)

    def eager(kv_caches, outs):
        # the decode kernel takes a table whose last used column holds the step's slot
        launcher = ops.DecodeLayerLauncher(q_s[0], k_s[0], v_s[0], kv_caches[0], Hq, Hkv, D, slots_s,
                                           None, bt_s, lens_s, max_len, scale)
        for li in range(layers):
            launcher.run(q_s[li], k_s[li], v_s[li], kv_caches[li], outs[li])

    set_step(0)
    # warm-up on the capture stream (library load, workspace), then capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager([t.clone() for t in kv_graph], torch.zeros_like(out_s))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eager(kv_graph, out_s)
    # the capture itself launched nothing: the caches are still equal
    for a, b in zip(kv_graph, kv_eager):
        assert torch.equal(a, b)

    for step in range(steps):
        set_step(step)
        if step == 2:   # a compacted table, as after an eviction: column 1 removed in place
            bt_s[:, 1:-1] = bt_s[:, 2:].clone()
            set_step(step)
        q_s.copy_(rnd(layers, B, Hq * D))
        k_s.copy_(rnd(layers, B, Hkv * D))
        v_s.copy_(rnd(layers, B, Hkv * D))
        graph.replay()
        got = out_s.clone()
        ref = torch.zeros_like(out_s)
        eager(kv_eager, ref)
        torch.cuda.synchronize()
        assert torch.equal(got, ref), step
        for a, b in zip(kv_graph, kv_eager):
            assert torch.equal(a, b), step
