"""pe_paged_decode_noalloc (split-KV partials in the library's own device array, no caller
workspace) against pe_paged_decode, through the C ABI: same kernels and the same split count,
so the outputs must be bit-identical; two calls in a row, the second smaller, and a call
whose partials exceed the built-in array must be refused without launching."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(num_seqs, kv_len, Hq, Hkv, D, seed, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    bs = 16
    nblk = -(-kv_len // bs)
    nb = num_seqs * nblk + 3
    sc = D ** -0.5
    kc = ((torch.rand(nb, Hkv, D // 8, bs, 8, generator=g) * 2 - 1) * sc * 3).to(dtype)
    vc = (torch.rand(nb, Hkv, D, bs, generator=g) * 2 - 1).to(dtype)
    q = ((torch.rand(num_seqs, Hq, D, generator=g) * 2 - 1) * sc * 3).to(dtype)
    bt = torch.randperm(nb, generator=g)[:num_seqs * nblk].to(torch.int32).reshape(num_seqs, nblk)
    lens = torch.randint(max(1, kv_len - 40), kv_len + 1, (num_seqs,), generator=g).to(torch.int32)
    lens[0] = kv_len
    return [t.cuda() for t in (q, kc, vc, bt, lens)]


def _run_both(lib, _lib, q, kc, vc, bt, lens, kv_len):
    S, Hq, D = q.shape
    Hkv = vc.shape[1]
    dt = _lib.PE_BF16 if q.dtype == torch.bfloat16 else _lib.PE_F16
    st = torch.cuda.current_stream().cuda_stream
    need = lib.pe_paged_decode_workspace_bytes(S, Hq, Hkv, D, 16, kv_len)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    out_a = torch.full_like(q, float("nan"))
    out_b = torch.full_like(q, float("nan"))
    args = (q.data_ptr(), kc.data_ptr(), vc.data_ptr(), bt.data_ptr(), lens.data_ptr(), None, S, Hq,
            Hkv, D, 16, bt.shape[1], q.stride(0), kc.stride(0), kc.stride(1), D ** -0.5, kv_len, dt)
    _lib.check(lib.pe_paged_decode(out_a.data_ptr(), *args, ws.data_ptr(), ws.numel(), st), "paged_decode")
    _lib.check(lib.pe_paged_decode_noalloc(out_b.data_ptr(), *args, st), "paged_decode_noalloc")
    torch.cuda.synchronize()
    return out_a, out_b, need


def test_noalloc_decode_matches_workspace_entry_point(oracle):
    from vllm_paged_compression_amd import _lib
    lib = _lib.load()
    assert lib.pe_paged_decode_noalloc_max_bytes() == 8 << 20   # also resolves this device's array
    # first call: 3 sequences of up to 4104 tokens -> several KV splits (partials in the array);
    # second call, right after it: fewer, shorter sequences (fewer splits, smaller partials) - it
    # must not see anything the first call left in the array
    needs = []
    for (S, kv, seed) in ((3, 4104, 1), (2, 1500, 2)):
        q, kc, vc, bt, lens = _case(S, kv, 32, 8, 128, seed)
        a, b, need = _run_both(lib, _lib, q, kc, vc, bt, lens, kv)
        needs.append(need)
        assert not torch.isnan(b.float()).any()
        assert torch.equal(a, b), f"S={S} kv={kv}: no-alloc output differs from pe_paged_decode"
        ref = oracle.paged_attention(q.cpu(), kc.cpu(), vc.cpu(), bt.cpu(), lens.cpu(), 128 ** -0.5)
        torch.testing.assert_close(b.cpu().float(), ref.float(), atol=1e-3, rtol=1.6e-2)
    assert needs[0] > 0 and needs[1] > 0, "both calls must take the split-KV path"
    # an unsplit call (no partials at all) works too
    q, kc, vc, bt, lens = _case(70, 300, 8, 8, 64, 3, torch.float16)
    a, b, need = _run_both(lib, _lib, q, kc, vc, bt, lens, 300)
    assert torch.equal(a, b)


def test_noalloc_decode_allocates_nothing_and_runs_under_graph_capture():
    """Device memory in use does not change across calls (hipMemGetInfo through torch), and a call
    captured into a HIP graph replays to the bits of the eager call."""
    from vllm_paged_compression_amd import _lib
    lib = _lib.load()
    lib.pe_paged_decode_noalloc_max_bytes()          # set-up call, outside capture
    q, kc, vc, bt, lens = _case(3, 4104, 32, 8, 128, 5)
    S, Hq, D = q.shape
    args = (q.data_ptr(), kc.data_ptr(), vc.data_ptr(), bt.data_ptr(), lens.data_ptr(), None, S, Hq,
            8, D, 16, bt.shape[1], q.stride(0), kc.stride(0), kc.stride(1), D ** -0.5, 4104, _lib.PE_BF16)
    eager = torch.full_like(q, float("nan"))
    replayed = torch.full_like(q, float("nan"))
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        _lib.check(lib.pe_paged_decode_noalloc(eager.data_ptr(), *args, torch.cuda.current_stream().cuda_stream),
                   "paged_decode_noalloc")
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, f"device memory in use changed by {free0 - free1} bytes across the calls"
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            _lib.check(lib.pe_paged_decode_noalloc(replayed.data_ptr(), *args, side.cuda_stream),
                       "paged_decode_noalloc under capture")
    for _ in range(2):
        replayed.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(replayed, eager)
    # a stream of another device than the current one is refused (one-GPU boxes cannot show it)
    if torch.cuda.device_count() > 1:
        other = torch.cuda.Stream(device=1)
        rc = lib.pe_paged_decode_noalloc(eager.data_ptr(), *args, other.cuda_stream)
        assert rc == 1 and b"is current" in lib.pe_last_error()


def test_noalloc_decode_rejects_oversized_partials(monkeypatch):
    from vllm_paged_compression_amd import _lib
    lib = _lib.load()
    S, kv = 40, 16400
    q = torch.zeros(S, 32, 128, dtype=torch.bfloat16, device="cuda")
    kc = torch.zeros(4, 8, 16, 16, 8, dtype=torch.bfloat16, device="cuda")
    vc = torch.zeros(4, 8, 128, 16, dtype=torch.bfloat16, device="cuda")
    bt = torch.zeros(S, 1025, dtype=torch.int32, device="cuda")
    lens = torch.ones(S, dtype=torch.int32, device="cuda")
    out = torch.zeros_like(q)
    monkeypatch.setenv("PE_DECODE_SPLITS", "64")   # 64 x 40 x 32 x 130 x 4 B = 42.6 MB of partials
    rc = lib.pe_paged_decode_noalloc(
        out.data_ptr(), q.data_ptr(), kc.data_ptr(), vc.data_ptr(), bt.data_ptr(), lens.data_ptr(),
        None, S, 32, 8, 128, 16, 1025, q.stride(0), kc.stride(0), kc.stride(1), 128 ** -0.5, kv,
        _lib.PE_BF16, torch.cuda.current_stream().cuda_stream)
    assert rc == 4, rc   # PE_ERR_WORKSPACE
    assert b"exceed the built-in scratch" in lib.pe_last_error()
    torch.cuda.synchronize()
    assert float(out.abs().sum()) == 0.0
