"""One-shot all-reduce over HIP-IPC mapped buffers: two ranks (two processes sharing the test
GPU; the buffers are exchanged as IPC handles exactly as ranks on different GPUs of a node would)
must produce, bit for bit, the f32-accumulated sum in rank order - for two ranks that is also
what a gloo all-reduce of the same bf16/fp16/f32 tensors gives.  xGMI bandwidth is not what this
measures; it pins correctness of the flag protocol, the slot reuse and the dispatch by size."""
import os
import socket
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vllm_paged_compression_amd import parallel
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        group = dist.group.WORLD
        comm = parallel.enable_one_shot_all_reduce(group, dev, max_bytes=1 << 20)
        g = torch.Generator().manual_seed(1000 + rank)
        checked = 0
        # decode-sized o_proj partials [B, 8192] and odd small sizes; several calls queued back to
        # back without a host synchronisation in between exercise the alternating slots
        cases = [(torch.bfloat16, (1, 8192)), (torch.bfloat16, (64, 8192)), (torch.float16, (7, 4096)),
                 (torch.float32, (3, 1000)), (torch.bfloat16, (8,)), (torch.bfloat16, (63, 8192))]
        for rep in range(4):
            pending = []
            for dtype, shape in cases:
                x_cpu = (torch.rand(shape, generator=g) * 2 - 1).to(dtype)
                x = x_cpu.to(dev)
                assert comm.should_use(x)
                seq_before = comm.seq
                parallel.all_reduce_sum(x, group)          # dispatched to the one-shot path
                assert comm.seq == seq_before + 1
                pending.append((x_cpu, x))
            for x_cpu, x in pending:
                parts = [torch.empty_like(x_cpu) for _ in range(world)]
                dist.all_gather(parts, x_cpu)
                acc = torch.zeros(x_cpu.shape, dtype=torch.float32)
                for p in parts:                              # rank order, f32 accumulation
                    acc += p.float()
                want = acc.to(x_cpu.dtype)
                got = x.cpu()
                assert torch.equal(got, want), (rank, x_cpu.dtype, tuple(x_cpu.shape), rep)
                red = x_cpu.clone()
                dist.all_reduce(red)                         # gloo's own sum of the same tensors
                assert torch.equal(got, red)
                checked += 1
        # a message beyond the slot goes to the group's backend (gloo here: staged through the host)
        big = (torch.rand(1 << 20, generator=g)).to(torch.bfloat16)      # 2 MiB
        xb = big.to(dev)
        assert not comm.should_use(xb)
        seq_before = comm.seq
        parallel.all_reduce_sum(xb, group)
        assert comm.seq == seq_before
        red = big.clone()
        dist.all_reduce(red)
        assert torch.equal(xb.cpu(), red)
        # the C entry point refuses what does not fit or is not 16-byte granular
        from vllm_paged_compression_amd import _lib
        lib = _lib.load()
        rc = lib.pe_allreduce_oneshot(xb.data_ptr(), xb.data_ptr(), xb.numel(), _lib.PE_BF16, comm._ptrs,
                                      rank, world, comm.buffer_bytes, 99, None, 0, None)
        assert rc == 4 and b"exceeds" in lib.pe_last_error()
        assert comm.enabled and comm.disabled_reason is None and comm.path(1 << 19) == "oneshot"
        comm.check()
        assert comm.timeouts() == 0
        parallel.disable_one_shot_all_reduce(group)
        q.put((rank, "ok", checked))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_one_shot_all_reduce_two_ranks_one_gpu():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), [r[2] for r in res if r[1] != "ok"]
    assert all(r[2] == 24 for r in res)


def _worker_two_stage(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vllm_paged_compression_amd import parallel
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        group = dist.group.WORLD
        comm = parallel.enable_one_shot_all_reduce(group, dev, max_bytes=4 << 20)
        g = torch.Generator().manual_seed(2000 + rank)
        # with four ranks the reference's dispatch takes two stages from 512 KiB: decode-sized
        # partials of a large batch, vector counts that do not divide by the ranks, and small
        # messages below the threshold in between (both forms share the slots and the sequence)
        cases = [(torch.bfloat16, (128, 8192), True), (torch.bfloat16, (8, 8192), False),
                 (torch.bfloat16, (33, 8200), True), (torch.float32, (3, 50000), True),
                 (torch.float16, (512, 4096), True), (torch.bfloat16, (64, 8192), True),
                 (torch.bfloat16, (7, 4096), False)]
        checked = 0
        for rep in range(3):
            comm.force_two_stage = rep == 2          # last round: tiny messages in two stages too
            pending = []
            for dtype, shape, two in cases + ([(torch.bfloat16, (8,), True), (torch.float32, (12,), True)]
                                              if rep == 2 else []):
                x_cpu = (torch.rand(shape, generator=g) * 2 - 1).to(dtype)
                x = x_cpu.to(dev)
                assert comm.should_use(x)
                before = comm.calls_two_stage
                parallel.all_reduce_sum(x, group)
                assert comm.calls_two_stage - before == int(two or rep == 2)
                pending.append((x_cpu, x))
            for x_cpu, x in pending:
                parts = [torch.empty_like(x_cpu) for _ in range(world)]
                dist.all_gather(parts, x_cpu)
                acc = torch.zeros(x_cpu.shape, dtype=torch.float32)
                for p in parts:                              # rank order, f32 accumulation
                    acc += p.float()
                assert torch.equal(x.cpu(), acc.to(x_cpu.dtype)), (rank, x_cpu.dtype, tuple(x_cpu.shape), rep)
                checked += 1
        # what does not fit the slot with its reduced chunk is refused by the C entry point
        from vllm_paged_compression_amd import _lib
        lib = _lib.load()
        big = torch.zeros(5 << 20, dtype=torch.uint8, device=dev).view(torch.bfloat16)
        assert not comm.should_use(big)
        rc = lib.pe_allreduce_twostage(big.data_ptr(), big.data_ptr(), big.numel(), _lib.PE_BF16, comm._ptrs,
                                       rank, world, comm.buffer_bytes, 999, None, 0, None)
        assert rc == 4 and b"two-stage" in lib.pe_last_error()
        comm.force_two_stage = False
        assert comm.enabled and comm.path(1 << 19) == "twostage" and comm.path(1 << 12) == "oneshot"
        assert comm.path(64 << 20) == "backend"
        comm.check()
        assert comm.timeouts() == 0
        parallel.disable_one_shot_all_reduce(group)
        q.put((rank, "ok", checked))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_stage_all_reduce_four_ranks_one_gpu():
    """Reduce-scatter + all-gather over the same peer-mapped buffers (four processes sharing the
    test GPU): bit for bit the rank-order f32 sum, chunk boundaries that do not divide evenly,
    one-stage and two-stage calls interleaved without host synchronisation."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_two_stage, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=200) for _ in procs)
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), [r[2] for r in res if r[1] != "ok"]
    assert all(r[2] == 23 for r in res)


def test_one_shot_all_reduce_single_rank_is_identity():
    """world = 1 through the C ABI: out = in (the degenerate sum), no peers to wait for."""
    import ctypes
    from vllm_paged_compression_amd import _lib
    lib = _lib.load()
    nbytes = lib.pe_ar_buffer_bytes(1 << 16)
    assert nbytes == 8192 + 2 * (1 << 16)
    buf = ctypes.c_void_p()
    _lib.check(lib.pe_ar_alloc(ctypes.byref(buf), nbytes), "ar_alloc")
    try:
        ptrs = (ctypes.c_void_p * 1)(buf.value)
        x = torch.randn(4, 1024, device="cuda").to(torch.bfloat16)
        y = torch.empty_like(x)
        for seq in (1, 2, 3):
            _lib.check(lib.pe_allreduce_oneshot(y.data_ptr(), x.data_ptr(), x.numel(), _lib.PE_BF16, ptrs,
                                                0, 1, nbytes, seq, None, 0,
                                                torch.cuda.current_stream().cuda_stream),
                       "allreduce")
            torch.cuda.synchronize()
            assert torch.equal(x, y)
        n = ctypes.c_uint(7)
        _lib.check(lib.pe_ar_timeouts(buf, ctypes.byref(n)), "ar_timeouts")
        assert n.value == 0
        assert lib.pe_allreduce_oneshot(y.data_ptr(), x.data_ptr(), x.numel(), _lib.PE_BF16, ptrs, 0, 1,
                                        nbytes, 0, None, 0, None) == 1          # seq starts at 1
    finally:
        _lib.check(lib.pe_ar_free(buf), "ar_free")


def _rccl_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from vllm_paged_compression_amd import parallel
        x = torch.arange(4096, dtype=torch.float32, device=dev)
        y = parallel.all_reduce_sum(x.clone(), dist.group.WORLD)         # RCCL all-reduce on the GPU
        v = parallel.broadcast_victims(torch.tensor([3, -1, 7], dtype=torch.int32, device=dev),
                                       dist.group.WORLD, 0)
        torch.cuda.synchronize()
        ok = torch.equal(x, y) and v.cpu().tolist() == [3, -1, 7] and dist.get_backend() == "nccl"
        # the one-shot communicator on an RCCL group (handles travel through all_gather_object)
        comm = parallel.enable_one_shot_all_reduce(dist.group.WORLD, dev, 1 << 16)
        z = torch.randn(8, 1024, device=dev).to(torch.bfloat16)
        z2 = parallel.all_reduce_sum(z.clone(), dist.group.WORLD)
        torch.cuda.synchronize()
        ok = ok and torch.equal(z, z2) and comm.seq == 1 and comm.timeouts() == 0
        parallel.disable_one_shot_all_reduce(dist.group.WORLD)
        q.put(("ok" if ok else "mismatch", ""))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put(("fail", traceback.format_exc()))


def test_rccl_backend_initialises_and_serves_the_exchange_steps():
    """The exchange helpers over a real RCCL ("nccl") process group - with the one rank a one-GPU
    box can hold, so what this pins is that RCCL loads, initialises on the device and runs the
    collectives the tensor-parallel path issues; multi-rank behaviour is covered with gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    status, detail = q.get(timeout=150)
    p.join(60)
    assert status == "ok", detail


def _worker_missing_peer(rank, world, port, q, two_stage):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vllm_paged_compression_amd import parallel
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        group = dist.group.WORLD
        comm = parallel.enable_one_shot_all_reduce(group, dev, max_bytes=1 << 20, timeout_s=0.5)
        assert comm.enabled
        comm.force_two_stage = two_stage
        x = torch.full((64, 4096), float(rank + 1), dtype=torch.bfloat16, device=dev)
        parallel.all_reduce_sum(x, group)                 # everybody there: the plain sum
        comm.check()
        assert bool((x == world * (world + 1) / 2).all())
        dist.barrier()
        y = torch.full((64, 4096), 1.0, dtype=torch.bfloat16, device=dev)
        outcome = "skipped"
        if rank != world - 1:                             # the last rank never makes this call
            parallel.all_reduce_sum(y, group)
            try:
                comm.check()
                outcome = "no error"
            except parallel.AllReduceError as e:
                outcome = "raised"
                assert f"rank {world - 1} did not arrive" in str(e)
            # no partial sum: every workgroup that gave up wrote NaN (a chunk whose owner saw all
            # its peers can still be whole - the missing rank owns the last chunk)
            assert bool(torch.isnan(y.float()).any())
            assert not bool(((y == 1.0) | (y == float(world - 1))).all())
            assert comm.timeouts() > 0
            try:                                          # the communicator refuses further use
                parallel.all_reduce_sum(torch.zeros(8, dtype=torch.float32, device=dev), group)
                outcome = "second call passed"
            except parallel.AllReduceError:
                pass
        dist.barrier()
        try:
            parallel.disable_one_shot_all_reduce(group)   # close() reports it once more
            closed = "quiet"
        except parallel.AllReduceError:
            closed = "raised"
        q.put((rank, "ok", (outcome, closed)))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,two_stage", [(2, False), (4, True)])
def test_missing_peer_poisons_the_output_and_raises(world, two_stage):
    """One rank skips a call: the others must neither hang nor return a partial sum.  Their wait
    gives up after the communicator's bound (0.5 s here), the output carries NaN, and check(), the
    next all_reduce and close() raise AllReduceError."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_missing_peer, args=(r, world, port, q, two_stage)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=200) for _ in procs)
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), [r[2] for r in res if r[1] != "ok"]
    for rank, _, (outcome, closed) in res:
        if rank == world - 1:
            assert (outcome, closed) == ("skipped", "quiet")
        else:
            assert (outcome, closed) == ("raised", "raised")


def _worker_bad_mapping(rank, world, port, q, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vllm_paged_compression_amd import parallel
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        group = dist.group.WORLD
        if mode == "wrong_pattern" and rank == 1:
            # this rank writes its check patterns as if it were rank 0: what its peers read through
            # their mapping of its buffer is then not what rank 1 should have written
            orig = parallel.OneShotAllReduce._verify_mappings

            def corrupt(self):
                true_rank, self.rank = self.rank, 0
                lib = self._lib
                lib.pe_ar_selfcheck(self._ptrs, 0, self.world, 0, None,
                                    torch.cuda.current_stream(self.device).cuda_stream)
                self.rank = true_rank
                import ctypes
                bad = torch.zeros(1, dtype=torch.int32, device=self.device)
                torch.cuda.synchronize()
                dist.barrier(group=self.group)
                lib.pe_ar_selfcheck(self._ptrs, self.rank, self.world, 1, bad.data_ptr(),
                                    torch.cuda.current_stream(self.device).cuda_stream)
                torch.cuda.synchronize()
                dist.barrier(group=self.group)
                n = int(bad.item())
                return f"{n} threads read a wrong pattern" if n else None
            parallel.OneShotAllReduce._verify_mappings = corrupt
        comm = parallel.enable_one_shot_all_reduce(group, dev, max_bytes=1 << 20, timeout_s=1.0)
        # every rank has switched the communicator off, whoever saw the problem
        assert comm is not None and not comm.enabled and "wrong pattern" in comm.disabled_reason
        x_cpu = torch.full((16, 1024), float(rank + 1), dtype=torch.bfloat16)
        x = x_cpu.to(dev)
        assert not comm.should_use(x) and comm.path(x.numel() * 2) == "backend"
        parallel.all_reduce_sum(x, group)                 # the group's own backend
        assert bool((x == world * (world + 1) / 2).all()) and comm.seq == 0
        parallel.disable_one_shot_all_reduce(group)
        q.put((rank, "ok", comm.disabled_reason))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_failed_mapping_check_switches_every_rank_to_the_backend():
    """The set-up self-check: a rank whose buffer does not show the expected patterns through the
    mappings makes ALL ranks switch the communicator off and use the group's backend."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bad_mapping, args=(r, 2, port, q, "wrong_pattern")) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), [r[2] for r in res if r[1] != "ok"]
    assert res[0][2] == res[1][2]


def test_pe_ar_disable_keeps_every_message_on_the_backend(monkeypatch):
    from vllm_paged_compression_amd import parallel
    monkeypatch.setenv("PE_AR_DISABLE", "1")
    assert parallel.enable_one_shot_all_reduce(None, "cuda:0") is None


def test_sequence_number_wraps_onto_the_other_slot():
    """The call number handed to the kernels wraps from 0xfffffffe to 1: consecutive calls always
    differ in parity (slot), and it never becomes 0 (which the entry points refuse)."""
    from vllm_paged_compression_amd.parallel import OneShotAllReduce
    seq = OneShotAllReduce.SEQ_WRAP - 2
    seen = []
    for _ in range(5):
        seq = seq % OneShotAllReduce.SEQ_WRAP + 1
        seen.append(seq)
    assert seen == [0xFFFFFFFD, 0xFFFFFFFE, 1, 2, 3]
    assert all((a ^ b) & 1 for a, b in zip(seen, seen[1:])) and 0 not in seen
