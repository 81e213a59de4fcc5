"""INTEGRATION.md section B shows the ctypes stub a maintainer of the reference would write against include/mcamd.h.
The stub is executed here as written (only the library path is pointed at the in-tree build), so the document cannot
drift from the ABI: its two structures must list the header's fields in order (CPU check against _lib.py's mirror),
and the function must reproduce F.conv2d(x, weight * mask, bias) on a GPU."""
import os
import re

import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B."):]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "no python block in INTEGRATION.md section B"
    return m.group(1)


def test_doc_structs_list_the_abi_fields_in_order():
    from modelcompression_amd import _lib
    src = _stub_source()
    ns = {}
    exec(src.split("lib.mcamd_packed_elems_fwd.restype")[0].replace('lib = C.CDLL("libmcamd.so")', "lib = None"), ns)
    assert [f[0] for f in ns["Geom"]._fields_] == [f[0] for f in _lib.ConvGeom._fields_]
    assert [f[0] for f in ns["Epi"]._fields_] == [f[0] for f in _lib.ConvEpilogue._fields_]
    import ctypes as C
    assert C.sizeof(ns["Geom"]) == C.sizeof(_lib.ConvGeom) and C.sizeof(ns["Epi"]) == C.sizeof(_lib.ConvEpilogue)


@pytest.mark.gpu
def test_doc_stub_runs_and_matches_conv2d(dev):
    from modelcompression_amd import _lib
    _lib.lib()                                   # builds / loads the in-tree library (raises when it is missing)
    so = os.path.join(ROOT, "modelcompression_amd", "libmcamd.so")
    ns = {}
    exec(_stub_source().replace('C.CDLL("libmcamd.so")', "C.CDLL(%r)" % so), ns)
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(2, 48, 13, 17, generator=gen)
    w = torch.randn(40, 48, 3, 3, generator=gen) * 0.05
    mask = (torch.rand(40, 48, 3, 3, generator=gen) > 0.4).float()
    bias = torch.randn(40, generator=gen)
    y = ns["conv2d_mi355x"](x.to(dev), w.to(dev).contiguous(), mask.to(dev).contiguous(), bias.to(dev))
    torch.cuda.synchronize()
    ref = F.conv2d(x.half().float(), (w * mask).half().float(), bias, 1, 1)
    err = float((y.cpu().double() - ref.double()).norm() / ref.double().norm())
    assert err < 1e-3, err
