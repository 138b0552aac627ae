"""CPU-only checks: the C ABI library exports what include/seld_hip.h declares, the host-side mirror has the
reference's state-dict layout and initialisation, config handling, and the product path refuses to run
without the HIP device (no fallback)."""
import importlib
import os
import re

import numpy as np
import pytest
import torch

from tests.golden.cases import MODEL_CASES, model_kwargs
from tests.helpers import PKG, build_model, pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import ctypes
    lib_path = os.path.join(ROOT, PKG, "csrc", "libseld_hip.so")
    if not os.path.exists(lib_path):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(lib_path)
    header = open(os.path.join(ROOT, "include", "seld_hip.h")).read()
    names = sorted(set(re.findall(r"\b(seld_\w+)\s*\(", header)))
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.seld_build_arch.restype = ctypes.c_char_p
    assert lib.seld_build_arch() == b"gfx950"
    assert lib.seld_abi_version() >= 1


def test_descriptor_validation_without_gpu():
    """Pure host-side argument checking of the ABI (no kernel is launched)."""
    import ctypes
    L = pkg()._lib
    H = pkg().hip_ops
    d = H.make_conv_desc((2, 16, 40), 32, 8, (3,), 1, 1, 1)
    assert H.conv_out_shape(d) == (1, 40)
    d2 = H.make_conv_desc((2, 8, 12, 20), 16, 4, (3, 3), 1, 1, 1)
    assert H.conv_out_shape(d2) == (12, 20)
    bad = H.make_conv_desc((2, 12, 40), 32, 8, (3,), 1, 1, 1)      # 12 channels: not a multiple of 8
    out = (ctypes.c_int32 * 2)()
    assert L.lib().seld_hc_conv_out_shape(ctypes.byref(bad), out) == -1
    assert L.lib().seld_stft_frames(6400, 512, 112) == 16
    assert L.lib().seld_stft_frames(32000 * 60, 512, 112) == 4800


@pytest.mark.parametrize("case", MODEL_CASES, ids=[c["name"] for c in MODEL_CASES])
def test_state_dict_layout_matches_reference(case, golden):
    g = golden("model_" + case["name"])
    m = build_model(case)
    sd = m.state_dict()
    assert list(sd.keys()) == str(g["sd_names"]).split("\n")
    shapes = [",".join(str(d) for d in v.shape) for v in sd.values()]
    assert shapes == str(g["sd_shapes"]).split("\n")


@pytest.mark.parametrize("case", [c for c in MODEL_CASES if c.get("train")], ids=lambda c: c["name"])
def test_default_initialisation_matches_reference(case, golden):
    """Same seeds -> same weights: the initialisers reproduce the reference's random draws."""
    g = golden("init")
    M = importlib.import_module(PKG + ".model")
    np.random.seed(1)
    torch.manual_seed(1)
    m = M.SELD_Model(**model_kwargs(case))
    names = str(g[case["name"] + ".names"]).split("\n")
    cks = g[case["name"] + ".checksums"]
    params = dict(m.named_parameters())
    assert list(params.keys()) == names
    for i, n in enumerate(names):
        d = params[n].detach().double()
        got = np.array([d.sum().item(), (d ** 2).sum().item(), d.flatten()[0].item(), d.flatten()[-1].item()])
        assert np.allclose(got, cks[i], rtol=1e-6, atol=1e-7), (n, got, cks[i])


def test_model_names_and_receptive_field():
    m = build_model(MODEL_CASES[3])
    assert m.model_name == "DualQSELD-TCN-PHI-S1_BN_RF287_10RB"
    assert (m.receptive_field, m.total_n_resblocks) == (287, 10)
    M = importlib.import_module(PKG + ".model")
    assert M.TC_Block.dilation_schedule([10], "fibonacci") == [1, 1, 2, 3, 5, 8, 13, 21, 34, 55]
    assert M.TC_Block.dilation_schedule([4], "pow2") == [1, 2, 4, 8]
    assert M.TC_Block.dilation_schedule([[1, 7], 2], "fibonacci") == [1, 7, 1, 1]


def test_no_cpu_fallback():
    L = pkg()._lib
    m = build_model(MODEL_CASES[3]).eval()
    x = torch.zeros(1, 8, 128, 64)
    with pytest.raises(L.SeldHipError):
        m(x)
    with pytest.raises(L.SeldHipError):
        pkg().hip_ops.stft_magphase(torch.zeros(2, 4096))


def test_readfile_and_configs(tmp_path):
    T = pkg().train
    p = tmp_path / "a.txt"
    p.write_text("--domain=DQ\n--use_bias_conv=False\n--fixed_seed=True\n--D=[10]\n#comment\n--lr=0.001\n")
    toks = T.readFile(str(p))
    assert toks == ['--domain', 'DQ', '--use_bias_conv', 0, '--fixed_seed', '1', '--D', '[10]', '--lr', '0.001']
    args = T.parse_args(["--TextArgs=" + str(p)])
    assert args.domain == "DQ" and args.use_bias_conv == 0 and args.fixed_seed == 1 and args.D == [10] and args.lr == 1e-3
    cfg_dir = os.path.join(ROOT, PKG, "config")
    for f in sorted(os.listdir(cfg_dir)):
        a = T.parse_args(["--TextArgs=" + os.path.join(cfg_dir, f)])
        m = T.model_from_args(a)
        n_params = sum(p.numel() for p in m.parameters())
        assert n_params > 0
        if "DQSELD-TCN-S1-PHI_8ch" in f and f.startswith("SERVER"):
            assert a.U == 384 and a.domain == "DQ" and m.model_name.startswith("DualQSELD-TCN-PHI-S1_BN_RF287_10RB")
        if "QSELD-TCN-S1-PHI_parallel" in f and f.startswith("SERVER"):
            # `--parallel_ConvTC_block=True` becomes '1', which is NOT a two-stream mode
            assert a.parallel_ConvTC_block == '1' and hasattr(m, "seld_block") and "_1_" in m.model_name


def test_flat_adam_aliases_parameters():
    """Host-side bookkeeping of the flat optimiser (the update itself is a HIP kernel, tested on the GPU)."""
    T = pkg().train
    lin = torch.nn.Linear(5, 3)
    w0 = lin.weight.detach().clone()
    opt = T.FlatAdam(lin.parameters(), lr=1e-3)
    assert opt.flat_param.numel() == 18
    assert torch.equal(lin.weight.detach(), w0)
    assert lin.weight.data_ptr() == opt.flat_param.data_ptr()
    assert lin.bias.data_ptr() == opt.flat_param.data_ptr() + 15 * 4
    lin(torch.ones(2, 5)).sum().backward()
    assert lin.weight.grad.data_ptr() == opt.flat_grad.data_ptr()      # autograd accumulated in place
    assert float(opt.flat_grad[:15].abs().sum()) > 0
    opt.zero_grad()
    assert float(opt.flat_grad.abs().sum()) == 0.0
    s = T.StepLR(opt, 2, 0.5)
    for _ in range(4):
        s.step()
    assert abs(opt.param_groups[0]["lr"] - 0.25e-3) < 1e-12


def test_shapes_beyond_32bit_offsets_are_not_given_to_the_fast_kernels():
    """The fast-product kernels index a tensor with 32-bit byte offsets (csrc/hcq_conv.hip, hcq_wgrad_grp.hip): the host must
    refuse them a tensor of 4 GB or more -- e.g. the first layer at batch 64, F = 256 (6.4 GB of output; /root/reference
    model.py:273-283 at twice the benchmark batch) -- so that such a call runs on the 64-bit-indexed generic kernels instead.
    Host-side queries only: nothing is launched."""
    import ctypes
    pkg = importlib.import_module(PKG)
    H, L = pkg.hip_ops, pkg._lib
    big = H.make_conv_desc((64, 8, 256, 512), 192, 8, (3, 3), 1, 1, 1)           # y: 64*192*256*512*4 B = 6.4 GB
    ok = H.make_conv_desc((32, 8, 128, 512), 192, 8, (3, 3), 1, 1, 1)            # the benchmark's first layer: 1.6 GB
    assert H.hcq_pack_floats(ok, 2) > 0 and H.hcq_pack_floats(big, 2) == 0       # mode 2: the pooling first-layer kernel
    ok1 = H.make_conv_desc((32, 192, 16, 512), 192, 8, (3, 3), 1, 1, 1)
    big1_2d = H.make_conv_desc((256, 192, 32, 512), 192, 8, (3, 3), 1, 1, 1)     # 3.2 GB each way ... x2 batch: 6.4 GB
    big2_2d = H.make_conv_desc((512, 192, 32, 512), 192, 8, (3, 3), 1, 1, 1)
    for mode in (0, 1):
        assert H.hcq_pack_floats(ok1, mode) > 0
        assert H.hcq_pack_floats(big2_2d, mode) == 0
    del big1_2d
    big1 = H.make_conv_desc((32, 192, 16384), 384, 8, (3,), 1, 1, 1)             # 1-D: y = 32*384*16384*4 B = 805 MB per ... ok
    assert H.hcq_pack_floats(big1, 0) > 0
    huge1 = H.make_conv_desc((256, 192, 16384), 384, 8, (3,), 1, 1, 1)           # y = 6.4 GB
    assert H.hcq_pack_floats(huge1, 0) == 0 and H.hcq_pack_floats(huge1, 1) == 0
    # grouped weight gradient: one image of either operand stays below 2 GB, the step space below 2^30
    fam = L.lib().seld_hcq_wgrad_group_family
    assert fam(ctypes.byref(H.make_conv_desc((32, 192, 512), 384, 8, (3,), 1, 1, 1))) == 0
    assert fam(ctypes.byref(H.make_conv_desc((32, 192, 16, 512), 192, 8, (3, 3), 1, 1, 1))) == 2
    assert fam(ctypes.byref(H.make_conv_desc((1, 192, 4096, 1024), 192, 8, (3, 3), 1, 1, 1))) == -1      # 3.2 GB per image
