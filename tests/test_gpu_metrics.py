"""Post-processing + test metrics on the device against the reference's evaluate_test fixtures
(metrics.npz, make_golden.gen_metrics) and the oracle."""
import time
import types

import numpy as np
import pytest
import torch

from oracle import seld_oracle as O
from tests.golden.cases import METRIC_CASES, metric_inputs
from tests.helpers import pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _counts(acc):
    H = pkg().hip_ops
    return dict(zip(H.METRIC_COUNTERS, acc[0].cpu().tolist())), float(acc[1].item())


@pytest.mark.parametrize("case", METRIC_CASES, ids=[c[0] for c in METRIC_CASES])
def test_counters_and_results_match_reference(case, golden):
    """Integer counters bit-exact; _total_DE and the 16 results to 1e-12 (double atomics add the same per-track
    averages in another order; acos / sqrt are the device's)."""
    H, T = pkg().hip_ops, pkg().train
    name, clips, frames, seed, variant = case
    g = golden("metrics")
    sed, doa, target = (torch.from_numpy(a).to(DEV) for a in metric_inputs(clips, frames, seed, variant))
    acc = H.metrics_new(DEV)
    H.metrics_accumulate(acc, sed, doa, target, frames)
    counts, total_de = _counts(acc)
    assert [counts[k] for k in H.METRIC_COUNTERS] == g[name + ".counters"].tolist()
    assert abs(total_de - g[name + ".total_DE"][0]) <= 1e-12 * max(1.0, abs(total_de))
    results = T.test_results_from_counters(counts, total_de, epoch=7)
    assert np.allclose(np.array(results, dtype=np.float64), g[name + ".results"], rtol=1e-12, atol=1e-12)
    # recording by recording (the reference's batch size 1) gives the same totals as one batched call
    acc2 = H.metrics_new(DEV)
    for k in range(clips):
        H.metrics_accumulate(acc2, sed[k], doa[k], target[k], frames)
    assert torch.equal(acc2[0], acc[0]) and abs(float(acc2[1]) - total_de) <= 1e-12 * max(1.0, abs(total_de))


def test_evaluate_test_mirror_matches_reference(golden):
    """train.evaluate_test with a stand-in model that replays the fixture predictions, as make_golden drove the
    reference's evaluate_test."""
    T = pkg().train
    name, clips, frames, seed, variant = METRIC_CASES[0]
    sed, doa, target = (torch.from_numpy(a) for a in metric_inputs(clips, frames, seed, variant))

    class Replay:
        k = 0

        def eval(self):
            return self

        def __call__(self, x):
            k = self.k
            self.k += 1
            return sed[k:k + 1].to(DEV), doa[k:k + 1].to(DEV)

    loader = [(torch.zeros(1, 1), target[k:k + 1]) for k in range(clips)]
    args = types.SimpleNamespace(output_classes=14, class_overlaps=3, Dcase21_metrics_DOA_threshold=20)
    out = T.evaluate_test(Replay(), torch.device(DEV), loader, epoch=7, max_loc_value=2., num_frames=frames,
                          spatial_threshold=2., args=args)
    assert len(out) == 16
    assert np.allclose(np.array(out, dtype=np.float64), golden("metrics")[name + ".results"], rtol=1e-12, atol=1e-12)


def test_edge_cases():
    H, L, T = pkg().hip_ops, pkg()._lib, pkg().train
    sed, doa, target = (torch.from_numpy(a).to(DEV) for a in metric_inputs(1, 20, 3, "mixed"))
    acc = H.metrics_new(DEV)
    H.metrics_accumulate(acc, sed[:0], doa[:0], target[:0], 20)                  # no recording
    assert int(acc[0].abs().sum()) == 0
    with pytest.raises(L.SeldHipError):
        H.metrics_accumulate(acc, sed, doa, target, 19)                          # an event beyond num_frames: KeyError there
    with pytest.raises(L.SeldHipError):
        H.metrics_accumulate(acc, sed, doa[..., :-1], target, 20)
    with pytest.raises(L.SeldHipError):
        H.metrics_accumulate(acc, sed.cpu(), doa, target, 20)
    H.metrics_accumulate(acc, sed, doa, torch.zeros_like(target), 20)            # no reference event at all
    counts, de = _counts(acc)
    assert counts["TP"] == 0 and counts["FN"] == 0 and counts["dc_Nref"] == 0 and counts["FP"] > 0
    with pytest.raises(ZeroDivisionError):                                        # train.py:136 divides by Nref
        T.test_results_from_counters(counts, de)
    # num_frames larger than the data: trailing blocks are empty
    acc3, acc4 = H.metrics_new(DEV), H.metrics_new(DEV)
    H.metrics_accumulate(acc3, sed, doa, target, 20)
    H.metrics_accumulate(acc4, sed, doa, target, 600)
    assert torch.equal(acc3[0], acc4[0])


def test_test_set_scale_against_oracle_sample():
    """A test set of 500 recordings x 600 frames (L3DAS21 Task 2 size) in one call; the oracle on 5 of them, timed."""
    H = pkg().hip_ops
    sed, doa, target = metric_inputs(500, 600, 21, "mixed")
    sd, dd, td = (torch.from_numpy(a).to(DEV) for a in (sed, doa, target))
    acc = H.metrics_new(DEV)
    H.metrics_accumulate(acc, sd[:5], dd[:5], td[:5], 600)
    t0 = time.perf_counter()
    _, ref_counts, ref_de = O.evaluate_clips(sed[:5], doa[:5], target[:5], num_frames=600)
    t_cpu = (time.perf_counter() - t0) / 5
    counts, de = _counts(acc)
    assert counts == ref_counts and abs(de - ref_de) <= 1e-12 * ref_de
    acc = H.metrics_new(DEV)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    H.metrics_accumulate(H.metrics_new(DEV), sd, dd, td, 600)                     # warm-up
    ev[0].record()
    H.metrics_accumulate(acc, sd, dd, td, 600)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1])
    counts, de = _counts(acc)
    assert counts["dc_Nref"] > 0 and counts["TP"] > 0
    # linearity: two halves add up to the whole
    a1, a2 = H.metrics_new(DEV), H.metrics_new(DEV)
    H.metrics_accumulate(a1, sd[:250], dd[:250], td[:250], 600)
    H.metrics_accumulate(a2, sd[250:], dd[250:], td[250:], 600)
    assert torch.equal(a1[0] + a2[0], acc[0]) and abs(float(a1[1] + a2[1]) - de) <= 1e-11 * de
    mb = (sd.numel() + dd.numel() + td.numel()) * 4 / 1e6
    print(f"N4 500 recordings x 600 frames: {ms:.3f} ms on the device ({mb / ms:.0f} GB/s of {mb:.0f} MB); "
          f"oracle {t_cpu * 1e3:.0f} ms per recording = {t_cpu * 500:.1f} s for the set")
