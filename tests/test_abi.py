"""CPU-only checks of the C-ABI boundary: the shared library loads, exports every symbol that
include/creamfl_hip.h declares, and the ctypes signature table covers exactly that set.  No
compute call is made (there is no GPU in the build container)."""
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'creamfl_hip.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cfl_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_something():
    syms = declared_symbols()
    assert 'cfl_pair_loss_fwd' in syms and 'cfl_rank_count' in syms and len(syms) >= 25


def test_library_built_and_exports_every_declared_symbol():
    from creamfl_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() / make -C creamfl_amd/csrc'
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f'{name} declared in creamfl_hip.h but not exported'
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_identity_calls_without_gpu():
    from creamfl_amd import _lib
    lib = _lib.load()
    assert lib.cfl_version() >= 100
    assert lib.cfl_arch() == b'gfx950'
    names = _lib.kernel_names()
    assert len(names) == lib.cfl_num_kernels() and all(n.startswith('cfl_') for n in names)
    # workspace-size helpers are pure host arithmetic
    assert lib.cfl_pair_loss_ws_bytes(256, 512) >= 5 * 256 * 4
    assert lib.cfl_bank_ws_bytes(128, 50000, 256) > 0
    # con_w scratch = split partials + the two pre-split operand images (M * D * 4 bytes each): never the [M, M] logits (10 GB)
    assert lib.cfl_conw_ws_bytes(50000, 50000, 256) < 3 * 50000 * 256 * 4
    assert lib.cfl_rank_ws_bytes(5000, 25000, 512) >= 5000 * 8


def test_ops_refuse_cpu_tensors():
    import torch
    from creamfl_amd import _lib, ops
    x = torch.randn(4, 8)
    with pytest.raises(_lib.CreamflHipError):
        ops.pair_loss(x, x, torch.ones(1), torch.ones(1))
    with pytest.raises(_lib.CreamflHipError):
        ops.l2_normalize(x)


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under creamfl_amd/ may reference it."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, 'creamfl_amd')):
        for fn in fns:
            if fn.endswith('.py'):
                txt = open(os.path.join(dp, fn)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_documented_binding_stub_agrees_with_the_signature_table():
    """INTEGRATION.md section B prints the ctypes stub a reference maintainer would add.  It is executed here (the library loads
    without a GPU; no compute call): every `argtypes` / `restype` it declares must equal the package's own table, so the document
    cannot drift from the ABI; tests/test_gpu_parity.py runs the same stubs against the oracle on the GPU."""
    import ctypes
    from conftest import integration_stubs
    from creamfl_amd import _lib
    ns = integration_stubs()
    assert 'PairLoss' in ns and 'ClientContrast' in ns
    lib = ns['_lib']
    seen = 0
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        if fn.argtypes is None:
            continue                                   # not bound by the stub
        seen += 1
        assert list(fn.argtypes) == list(argtypes), name
        if restype is not ctypes.c_int:
            assert fn.restype is restype, name
    assert seen >= 7
