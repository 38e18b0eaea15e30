"""CPU-only tests: host logic of the package (module tree, state-dict contract, DRW schedule, sampler, metrics),
the C-ABI library (loads, exports every symbol of include/mi355x_disrupt.h, rejects bad arguments without a GPU) and
the loud failure of the product path without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import losses as ol, r2plus1d as orc, step as ostep
from src import _native
from src.models.R2Plus1D import R2Plus1DClassifier
from src.train import drw_class_weights
from src.utils.metrics import macro_f1
from src.utils.sampler import ImbalancedDatasetSampler

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mi355x_disrupt.h")).read()
    declared = set(re.findall(r"\b(md_[a-z0-9_]+)\s*\(", hdr)) - {"md_cpad"}
    assert declared, "header parse failed"
    lib = _native.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _native.SIGNATURES, f"{name} has no ctypes prototype"
    arch = C.c_char_p()
    assert lib.md_version(C.byref(arch)) >= 1 and arch.value == b"gfx950"


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    lib = _native.lib()
    d = _native.MdConvDesc(1, 4, 8, 8, 4, 4, 8, 8, 8, 1, 3, 3, 1, 1, 1, 0, 1, 1)
    assert lib.md_conv_wpack_fwd_floats(C.byref(d)) > 0
    bad = _native.MdConvDesc(1, 4, 8, 8, 4, 9, 8, 8, 8, 1, 3, 3, 1, 1, 1, 0, 1, 1)      # To inconsistent
    assert lib.md_conv_wpack_fwd_floats(C.byref(bad)) == 0
    assert lib.md_conv_fwd(C.byref(bad), None, None, None, None, None) == -1             # MD_ERR_BAD_SHAPE
    assert lib.md_conv_fwd(C.byref(d), None, None, None, None, None) == -5                # MD_ERR_NULL
    assert lib.md_bn_act(None, 10, 4, None, None) == -5
    ls = (C.c_int32 * 4)(1, 2, 2, 1)
    h = C.c_void_p()
    assert lib.md_plan_create(8, 21, 128, 128, ls, 0.01, C.byref(h)) == 0
    assert lib.md_plan_num_units(h) == 32                  # 32 Conv3d at [1,2,2,1]
    assert lib.md_plan_feat_dim(h) == 128
    assert lib.md_plan_workspace_bytes(h) > 1 << 30
    u = _native.MdConvDesc()
    assert lib.md_plan_unit_desc(h, 2, C.byref(u)) == 0 and (u.Cin, u.Cout, u.kh, u.kw) == (32, 72, 3, 3)
    lib.md_plan_destroy(h)
    assert lib.md_plan_create(0, 21, 128, 128, ls, 0.01, C.byref(h)) == -1


def test_module_tree_matches_reference_contract():
    m = R2Plus1DClassifier(input_size=(3, 21, 128, 128), num_classes=2, layer_sizes=[1, 2, 2, 1], alpha=0.01)
    sd = m.state_dict()
    want = dict(orc.param_shapes([1, 2, 2, 1])); want.update(orc.buffer_shapes([1, 2, 2, 1]))
    assert set(sd) == set(want) and len(sd) == 201         # 201 state-dict tensors
    for k, shp in want.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    assert sum(p.numel() for p in m.parameters()) == 1587523
    # attribute names pinned by the reference's side tools
    assert hasattr(m.res2plus1d, "conv5") and hasattr(m, "linear") and hasattr(m, "encode") and hasattr(m, "summary")
    # init follows the reference: BN gamma 1 / beta 0, Kaiming-normal convs
    assert float(m.res2plus1d.conv1.spatio_conv.bn.weight.min()) == 1.0
    assert abs(float(m.res2plus1d.conv2.block1.conv1.spatio_conv.conv.weight.std()) - (2.0 / (32 * 9)) ** 0.5) < 0.01
    # quirk kept: inner units use LeakyReLU(0.01) regardless of alpha; stem and block closers use alpha
    m2 = R2Plus1DClassifier(input_size=(3, 8, 32, 32), layer_sizes=[1, 1, 1, 1], alpha=0.3)
    assert m2.res2plus1d.conv1.spatio_conv.relu.negative_slope == 0.3
    assert m2.res2plus1d.conv2.block1.relu.negative_slope == 0.3
    assert m2.res2plus1d.conv2.block1.conv1.spatio_conv.relu.negative_slope == 0.01


def test_product_path_fails_loudly_on_cpu():
    m = R2Plus1DClassifier(input_size=(3, 4, 32, 32), layer_sizes=[1, 1, 1, 1], alpha=0.01)
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 3, 4, 32, 32))
    from src.loss import FocalLoss
    with pytest.raises(RuntimeError):
        FocalLoss(torch.ones(2))(torch.zeros(2, 2), torch.zeros(2, dtype=torch.int64))


def test_drw_schedule_bit_exact_against_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "drw.npz"))
    for n in (8, 50, 128):
        tab = np.stack([drw_class_weights(e, n, [0, 0.25, 0.75, 0.9], [100, 2000]) for e in range(n)])
        assert np.array_equal(tab, g[f"w{n}"])


def test_macro_f1_and_sampler_bookkeeping():
    rng = np.random.default_rng(0)
    for _ in range(20):
        y = rng.integers(0, 2, 50); p = rng.integers(0, 2, 50)
        assert abs(macro_f1(y, p) - ostep.macro_f1(y, p)) < 1e-15
    assert macro_f1(np.array([0, 0, 1]), np.array([0, 0, 0])) == pytest.approx((0.8 + 0.0) / 2)

    class DS:
        labels = [0] * 5 + [1] * 95

        def __len__(self):
            return 100

    torch.manual_seed(3)
    s = ImbalancedDatasetSampler(DS())
    idx = list(iter(s))
    assert len(idx) == 100 and abs(sum(1 for i in idx if i < 5) - 50) < 20      # classes re-balanced
    torch.manual_seed(3)
    w = torch.DoubleTensor([1.0 / 5] * 5 + [1.0 / 95] * 95)
    assert idx == torch.multinomial(w, 100, replacement=True).tolist()           # index-exact with the reference recipe
