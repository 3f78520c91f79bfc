"""CPU checks of the host-side mirror of the reference interface: constructor signatures, state_dict
keys/shapes (checkpoint compatibility) and the relative_pos constant."""
import numpy as np
import pytest
import torch

from util import load_fixture, state_from

GRAPHER_CASES = ["f1_grapher_cfg1", "f2_grapher_g4", "f3_grapher_dil3", "f4a_grapher_r2", "f4b_grapher_r4",
                 "f7_grapher_bf16in", "f11_grapher_edgeconv"]
LABEL_CASES = ["f5_label_g2", "f5b_label_g1"]


def make_grapher(meta):
    from gkgnet_amd.grapher import Grapher
    return Grapher(meta["C"], meta["k"], meta["dilation"], meta["conv"], "gelu", "batch", True, False, 0.2,
                   meta["r"], n=meta["n"], drop_path=0.0, relative_pos=True,
                   use_multi_group=meta["use_multi_group"], num_group=meta["G"])


def make_label(meta):
    from gkgnet_amd.grapher import GrapherLabel
    return GrapherLabel(meta["C"], meta["k"], 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=meta["n"],
                        drop_path=0.0, relative_pos=False, num_nodes=meta["L"],
                        use_multi_group=meta["use_multi_group"], num_group=meta["G"])


@pytest.mark.parametrize("name", GRAPHER_CASES + LABEL_CASES)
def test_state_dict_is_checkpoint_compatible(name):
    meta, a = load_fixture(name)
    mod = make_grapher(meta) if meta["kind"] == "grapher" else make_label(meta)
    ref = state_from(a)
    mine = mod.state_dict()
    assert list(mine.keys()) == list(ref.keys())            # same keys, same order
    for k in ref:
        assert mine[k].shape == ref[k].shape and mine[k].dtype == ref[k].dtype, k
    mod.load_state_dict(ref, strict=True)
    if "relative_pos" in ref:
        assert not mod.relative_pos.requires_grad


def test_relative_pos_constants_bit_exact():
    from gkgnet_amd.relpos import build_relative_pos, resize_relative_pos
    meta, a = load_fixture("f9_relpos")
    for C, n, r in meta["combos"]:
        assert torch.equal(build_relative_pos(C, n, r), torch.from_numpy(a[f"rp_{C}_{n}_{r}"])), (C, n, r)
    got = resize_relative_pos(build_relative_pos(32, 64, 2), 64, 2, 10, 10)
    assert torch.allclose(got, torch.from_numpy(a["rp_runtime_32_64_2_to_10x10"]), atol=1e-6)


def test_forward_on_cpu_fails_loudly():
    from gkgnet_amd import _lib
    meta, a = load_fixture("f2_grapher_g4")
    mod = make_grapher(meta)
    with pytest.raises(_lib.GkgError):
        mod(torch.from_numpy(a["x"]))
