"""N > 1 path on CPU: trainer.GradExchange (the bucketed, backward-overlapped gradient all-reduce that
bench.py runs over RCCL) exercised with world_size = 2 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, bucket, offsets, out_q, wire="fp32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import wav2vec_s_amd  # noqa: F401
    from wav2vec_s_amd.trainer import GradExchange
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    if rank == 1:
        flat[n // 3: n // 2] = 0          # a LayerDrop-ped layer on one rank only: zeros still take part
        mine = flat.clone()
    ex = GradExchange(flat, dist, bucket_elems=bucket, wire_dtype=wire)
    for step in range(2):                  # two steps: state must reset between them
        flat.copy_(mine)
        ex.begin_step()
        for off in offsets:
            ex.on_ready(off)
        ex.finish()
        covered = np.zeros(n, dtype=np.int32)
        for lo, hi in ex.launched:
            covered[lo:hi] += 1
        assert (covered == 1).all(), "every element must be reduced exactly once"
        assert all(hi - lo >= min(bucket, n) or lo == 0 for lo, hi in ex.launched)
    gathered = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if wire == "bf16":     # each rank's range is rounded to bf16, summed in bf16 on the wire, widened back (exactly representable)
        bf = torch.bfloat16
        want = (gathered[0].to(bf).float() + gathered[1].to(bf).float()).to(bf).float()
        ok = bool(torch.equal(flat.to(bf).float(), flat)) and bool(((flat - want).abs() <= 2.0 ** -7 * want.abs() + 1e-30).all())
        ok = ok and bool((flat[n // 3: n // 2] == gathered[0].to(bf).float()[n // 3: n // 2]).all())   # zeros of the other rank take part
    else:
        want = sum(gathered)
        ok = bool(torch.allclose(flat, want, atol=1e-6))
    out_q.put((rank, ok, len(ex.launched)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
@pytest.mark.parametrize("n,bucket,offsets", [
    (10_000, 3_000, [9_500, 8_000, 6_100, 4_000, 2_500, 900, 0]),     # milestones as the backward reports them
    (10_000, 50_000, [9_000, 5_000, 0]),                              # bucket larger than the arena: one collective
    (4_097, 1_024, [4_000, 4_000, 10]),                               # repeated / missing final milestone
])
def test_grad_exchange_gloo_world2(n, bucket, offsets, wire):
    """Both wire types: "fp32" all-reduces the arena in place; "bf16" packs each range into a bf16 image, all-reduces that (the
    reference reduces in the model dtype, fs/distributed/legacy_distributed_data_parallel.py:100-115) and unpacks it."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, bucket, offsets, q, wire)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == res[1][2]          # both ranks issued the same sequence of collectives


def test_arena_is_in_forward_order_so_backward_finalises_a_suffix():
    import wav2vec_s_amd as w
    from wav2vec_s_amd import engine
    m = w.Wav2VecSModel(w.base_librispeech_config())
    W = {n: p for n, p in m.named_parameters()}
    A = engine.Arena(engine.grad_shapes(m.cfg, W), "cpu")
    off = lambda pre: engine.milestone_offset(A, pre)  # noqa: E731
    seq = [off("feature_extractor."), off("layer_norm."), off("post_extract_proj."), off("mask_emb"),
           off("encoder.layer_norm.")] + [off(f"encoder.layers.{i}.") for i in range(12)] + \
          [off("quantizer."), off("project_q."), off("final_proj.")]
    assert seq == sorted(seq) and seq[0] == 0
    # fused QKV block: q, k, v weights adjacent
    o = A.offsets
    e2 = 768 * 768
    assert o["encoder.layers.3.self_attn.k_proj.weight"][0] == o["encoder.layers.3.self_attn.q_proj.weight"][0] + e2
    assert o["encoder.layers.3.self_attn.v_proj.weight"][0] == o["encoder.layers.3.self_attn.q_proj.weight"][0] + 2 * e2
    assert A.numel >= 90_325_120


def test_grad_exchange_flushes_before_the_tail():
    """flush_at: when a milestone reaches it, what is pending goes out even below a bucket, so that only the range below
    flush_at (the conv extractor, final last) is left for after the backward.  No process group needed: a stub records."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import wav2vec_s_amd  # noqa: F401
    from wav2vec_s_amd.trainer import GradExchange

    class _Work:
        def wait(self):
            pass

    class _Dist:
        class ReduceOp:
            SUM = 0

        def all_reduce(self, t, op=None, group=None, async_op=False):
            return _Work()

    flat = torch.zeros(10_000)
    for flush_at, want in ((-1, [(7000, 10000), (4000, 7000), (1000, 4000), (0, 1000)]),     # the last two only at finish()
                           (1500, [(7000, 10000), (4000, 7000), (1500, 4000), (0, 1500)])):
        ex = GradExchange(flat, _Dist(), bucket_elems=3_000, flush_at=flush_at)
        ex.begin_step()
        for off in (9_500, 8_000, 6_100, 4_000, 2_500, 1_500):
            ex.on_ready(off)
        before_finish = list(ex.launched)
        ex.finish()
        assert ex.launched == want, (flush_at, ex.launched)
        covered = np.zeros(10_000, dtype=np.int32)
        for lo, hi in ex.launched:
            covered[lo:hi] += 1
        assert (covered == 1).all()
        if flush_at >= 0:
            assert before_finish[-1] == (1500, 4000)          # launched at the milestone, not at finish()


def test_grad_exchange_ranges_do_not_depend_on_the_milestones_a_rank_saw():
    """Two ranks whose LayerDrop decisions differ report different milestones above flush_at (the one that dropped the
    first layer never reports flush_at itself: its next milestone is already the extractor side's).  The all-reduce ranges must still pair up: the same list, in the same
    order, on both - only WHEN a range goes out may differ."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import wav2vec_s_amd  # noqa: F401
    from wav2vec_s_amd.trainer import GradExchange

    class _Work:
        def wait(self):
            pass

    class _Dist:
        class ReduceOp:
            SUM = 0

        def all_reduce(self, t, op=None, group=None, async_op=False):
            return _Work()

    flat = torch.zeros(10_000)
    tail = (1_200, 700, 0)                     # the extractor's milestones: the same on every rank
    seen = []
    for encoder_offsets in ((9_500, 8_000, 6_100, 4_000, 2_500, 1_500), (9_000, 6_100, 2_500), (8_000, 1_600), (1_500,), ()):
        ex = GradExchange(flat, _Dist(), bucket_elems=3_000, flush_at=1_500)
        ex.begin_step()
        for off in encoder_offsets + tail:
            ex.on_ready(off)
        ex.finish()
        covered = np.zeros(10_000, dtype=np.int32)
        for lo, hi in ex.launched:
            covered[lo:hi] += 1
        assert (covered == 1).all()
        seen.append(list(ex.launched))
    assert all(s == seen[0] for s in seen), seen
    assert seen[0][:3] == [(7000, 10000), (4000, 7000), (1500, 4000)]
