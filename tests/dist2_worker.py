"""One rank of the two-process data-parallel test (tests/test_a_dist_gpu.py::test_two_ranks_on_one_gpu_equal_accumulated_local_step).
Not a test module.  python tests/dist2_worker.py RANK PORT OUT WIRE UPDATES"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, port, out, wire, updates = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
    import datetime
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=2,
                            timeout=datetime.timedelta(seconds=300))   # a dead peer fails the collective, not the test's clock
    from test_a_dist_gpu import SMALL, BF, KEEP2, _build, _draws, batch_of
    from wav2vec_s_amd import trainer
    w, cfg, model, crit = _build(SMALL)                    # the same seed on both ranks: identical initial weights
    step = trainer.TrainStep(model, crit, world_size=2, lr=1e-3, clip_norm=0.05, update_freq=1, arena_gib=1.0, wire_dtype=wire)
    step.exchange.bucket = 50_000                          # several buckets on this small model
    assert step.exchange is not None
    B, L = 2, 16000
    launched = []
    for u in range(updates):
        src = batch_of(u, rank, B, L).to(BF).cuda()
        model.inject_draws(_draws(cfg, B, L, KEEP2[u], seed=7 + 10 * u + rank)())
        step({"net_input": {"source": src}})
        launched.append(list(step.exchange.launched))
        if u == 0:
            torch.cuda.synchronize()
            first = {"p32": step.flat.p32.cpu(), "m": step.flat.m.cpu(), "v": step.flat.v.cpu(), "arena": step.flat.arena.flat.cpu()}
    step.check()
    torch.cuda.synchronize()
    torch.save({"p32": step.flat.p32.cpu(), "m": step.flat.m.cpu(), "v": step.flat.v.cpu(), "p16": step.flat.p16.float().cpu(),
                "gn": float(step.grad_norm()), "arena": step.flat.arena.flat.cpu(), "offsets": dict(step.flat.arena.offsets), "launched": launched, "first": first, "numel": step.flat.arena.numel, "step": step.flat.step}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
