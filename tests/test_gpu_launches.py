"""Launch budget of one training step at the config-3 widths: how many kernels a step issues and
which of them are not this library's.  Counted with torch.profiler on an eager step (a recorded graph replays the same
nodes).  The reference's step is torch eager throughout (train.py:84-104); this guards the fused host path against
regressions that tests of values cannot see: a stray fill, a gradient autograd has to add, a pooled buffer that leaks."""
import collections

import pytest
import torch

from tests.golden.cases import MODEL_CASES, train_target
from tests.helpers import build_model, pkg
from oracle import seld_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASE = next(c for c in MODEL_CASES if c["name"] == "c3w_train")


def test_step_launch_budget():
    T, H = pkg().train, pkg().hip_ops
    case = dict(CASE, dropout_perc=0.3, spatial_dropout_rate=0.5)
    torch.manual_seed(5)
    m = build_model(case).to(DEV).train()
    opt = T.FlatAdam(m.parameters(), lr=1e-4)
    x = O.closed_form_input((case["B"], case["input_channels"], case["freq_dim"], case["time_dim"])).to(DEV)
    target = train_target(case).to(DEV)
    n_sed = int(case["output_classes"] * 3)

    DP = pkg().dp
    sync = DP.BucketedGradSync(opt, m)

    def step():                                   # the step bench.py times (single rank: the gradient sync is a no-op)
        DP.dp_train_step(m, opt, sync, x, target, n_sed, T.seld_loss_fn)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    names = collections.Counter(ev.name for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA)
    total = sum(names.values())
    foreign = {k: v for k, v in names.items() if "seld::" not in k}
    fills = {k: v for k, v in foreign.items() if "fill" in k.lower() or "memset" in k.lower() or "copy" in k.lower()}
    assert not fills, fills                      # no zero-fills, no device copies inside a step
    # nothing is left to autograd's own kernels: the attention's q / k / v projections are one stacked convolution, the sum
    # of the two classifier heads' input gradients is hip_ops.FanOut2Fn (this library's add)
    assert not foreign, foreign
    assert total <= 200, (total, names.most_common(12))

