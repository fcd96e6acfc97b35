"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol (no compute calls),
the product package never touches the oracle, config mirrors, flat-parameter data-parallel plumbing over gloo."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from sast_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from sast_amd import build
        build.build()
    lib = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib._SIGNATURES, f"{n} declared in include/sast_hip.h but not bound in sast_amd/_lib.py"
    assert lib.sast_version() >= 100


def test_struct_layouts_match_header():
    """ctypes mirrors must list the header's struct fields in the same order."""
    from sast_amd import _lib
    src = open(_lib.HEADER_PATH).read()
    for name in ("SastDownArgs", "SastScoreArgs", "SastSel", "SastMswsaArgs", "SastLstmArgs", "SastConvBnArgs"):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = re.findall(r"\*?\s*([A-Za-z_][A-Za-z0-9_]*)\s*(?:,|$)", decl.split(None, 1)[1] if " " in decl else decl)
            fields += names
        got = [f[0] for f in getattr(_lib, name)._fields_]
        assert got == fields, (name, got, fields)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sast_amd")
    for d, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h")):
                txt = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), os.path.join(d, f)
                assert "sast_oracle" not in txt, os.path.join(d, f)


def test_hot_path_fails_loudly_without_gpu():
    from sast_amd import functional as SF
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SF.non_zero_ratio(torch.zeros(1, 20, 32, 32))
    from sast_amd.layers import SAST_block
    from sast_amd.detection import PositionEmbeddingSine
    blk = SAST_block(64, dict(partition_size=(4, 5), mlp_activation="gelu"), first_block=True)
    pe = PositionEmbeddingSine(32, normalize=True, input_size=(1, 8, 10))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        blk(torch.randn(1, 8, 10, 64), pe, torch.rand(1, 20), None)


def test_modified_hw_and_state_dict_keys():
    from sast_amd.config import modified_hw, backbone_config
    assert modified_hw((360, 640)) == ((384, 640), (6, 10))       # 1Mpx, config/modifier.py:27-41
    assert modified_hw((240, 304), 1) == ((256, 320), (8, 10))    # Gen1: partition_split_32 = 1 (config/experiment/gen1)
    from sast_amd.detection import RNNDetector, YOLOPAFPN, PositionEmbeddingSine
    from oracle import sast_oracle as O
    net = RNNDetector(backbone_config((128, 160), (4, 5), embed_dim=32))
    ref = O.init_backbone_params(O.BackboneCfg(in_res_hw=(128, 160), partition_size=(4, 5), embed_dim=32))
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items() if "sub_layers" not in k}
    assert mine == {k: tuple(v.shape) for k, v in ref.items()}    # reference names + shapes
    assert "stages.0.att_blocks.0.att.win_attn.sub_layers.3.net.0.proj.weight" in net.state_dict()   # aliased duplicates
    fpn = YOLOPAFPN(depth=0.67, in_channels=(64, 128, 256))
    fr = O.init_pafpn_params((64, 128, 256))
    assert {k for k in fpn.state_dict() if "num_batches" not in k} == set(fr)
    pe = PositionEmbeddingSine(32, normalize=True, input_size=(1, 16, 20))
    assert torch.equal(pe.pos_embedding, O.position_embedding_sine(16, 20, 64))


def test_partition_maps_match_oracle():
    from sast_amd.layers import ops
    from oracle import sast_oracle as O
    x = torch.arange(2 * 12 * 20 * 3, dtype=torch.float32).view(2, 12, 20, 3)
    for mine, ref in ((ops.window_partition, O.window_partition), (ops.grid_partition, O.grid_partition)):
        assert torch.equal(mine(x, (6, 10)), ref(x, (6, 10)))
    w = ops.window_partition(x, (6, 10))
    assert torch.equal(ops.window_reverse(w, (6, 10), (12, 20)), x)
    g = ops.grid_partition(x, (6, 10))
    assert torch.equal(ops.grid_reverse(g, (6, 10), (12, 20)), x)


_DIST_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from sast_amd.dist import FlatParams, FusedAdamW
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)
m = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
conv = torch.nn.Module(); w = torch.empty(4, 3, 3, 2); torch.nn.init.normal_(w); conv.weight = torch.nn.Parameter(w.permute(0, 3, 1, 2))
fp = FlatParams([m, conv])
assert conv.weight.permute(0, 2, 3, 1).is_contiguous() and conv.weight.grad.stride() == conv.weight.stride()
opt = FusedAdamW(fp, lr=1e-2)
ref = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4)); ref.load_state_dict(m.state_dict())
ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.0)
xs = [torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
for step in range(3):
    fp.zero_grad()
    out = m(xs[rank]); loss = (out ** 2).mean()
    # emulate the fused in-place accumulation of the HIP backward: grads are added into the flat views
    gs = torch.autograd.grad(loss, list(m.parameters()))
    for p, g in zip(m.parameters(), gs): p.grad.add_(g)
    fp.all_reduce(); opt.step(grad_scale=1.0 / world)
    ropt.zero_grad(); sum((ref(x) ** 2).mean() for x in xs).div(world).backward(); ropt.step()
for a, b in zip(m.parameters(), ref.parameters()):
    assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_flat_params_allreduce_adamw_gloo_world2(tmp_path):
    """N>1 path on CPU: 2 ranks, gloo; averaged gradients + fused AdamW == single-process AdamW on the mean loss."""
    script = tmp_path / "worker.py"
    script.write_text(_DIST_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", str(script), ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2


def test_positive_linear_forward_matches_reference_expression():
    """SAST.py:325-328: F.linear(input, exp(weight), bias) -- the stand-alone module computes (inside SAST_block it is fused)"""
    import torch
    from sast_amd.layers.sast import PositiveLinear
    torch.manual_seed(0)
    m = PositiveLinear(20, 8, bias=True)
    x = torch.rand(3, 20)
    assert torch.equal(m(x), torch.nn.functional.linear(x, m.weight.exp(), m.bias))
