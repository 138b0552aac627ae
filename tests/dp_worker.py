"""Child process of tests/test_gpu_dp.py: one data-parallel rank of the real model on the one GPU of the box
(gloo backend, every rank on cuda:0 -- RCCL refuses two ranks on one device).  Started fresh by the test with
RANK / WORLD_SIZE / MASTER_* in the environment; writes its results to <out_dir>/rank<r>.pt.

    python tests/dp_worker.py <out_dir> <case_name> <mode: eager|graph> <steps>
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, case_name, mode, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    import torch
    from oracle import seld_oracle as O          # closed-form fill / input only (test infrastructure)
    from tests.golden.cases import MODEL_CASES, train_target
    from tests.helpers import PKG, build_model
    pkg = importlib.import_module(PKG)
    DP, T = pkg.dp, pkg.train
    rank, _, world = DP.init_from_env("gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    case = dict(next(c for c in MODEL_CASES if c["name"] == case_name), B=4)
    m = build_model(case)
    O.closed_form_fill_(list(m.state_dict().items()))
    if rank != 0:                                 # replicas start different: the broadcast must fix that
        with torch.no_grad():
            for p in m.parameters():
                p.mul_(1.5)
    m = m.to(dev).train()
    DP.seed_rank_streams(rank)
    opt = T.FlatAdam(m.parameters(), lr=1e-3, late=DP.late_parameters(m))
    DP.broadcast_parameters(opt.flat_param)
    sync = DP.BucketedGradSync(opt, m)
    assert sync.cut is not None and 0 < opt.late_numel < opt.flat_grad.numel()
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"]))
    target = train_target(case)
    lo, hi = DP.shard_batch(case["B"])
    xs, ts = x[lo:hi].contiguous().to(dev), target[lo:hi].contiguous().to(dev)
    n_sed = int(case["output_classes"] * 3)
    losses, first_grad = [], None
    if mode == "graph":
        runner = T.GraphedTrainStep(m, opt, xs, ts, n_sed, 1.0, 5.0, sync=sync, warmup=1)
        assert len(runner.graphs) == 3
        # the warm-up inside the constructor took one real step: results below are "after 1 + steps steps"
        for _ in range(steps):
            losses.append(float(runner().item()))
    else:
        for s in range(steps):
            loss = DP.dp_train_step(m, opt, sync, xs, ts, n_sed, T.seld_loss_fn)
            if s == 0:
                torch.cuda.synchronize()
                first_grad = opt.flat_grad.detach().cpu().clone()      # summed over ranks (before the 1/world of Adam)
            losses.append(float(loss.item()))
    torch.cuda.synchronize()
    names = [n for n, _ in m.named_parameters()]
    torch.save(dict(param=opt.flat_param.detach().cpu(), first_grad=first_grad, losses=losses, names=names,
                    offsets=opt.offsets, late_numel=opt.late_numel, step_count=opt.step_count,
                    state=pkg.hip_ops.philox.state(dev).cpu()),
               os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
