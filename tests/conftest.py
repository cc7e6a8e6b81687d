import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _miopen_env():
    """The product's own library set-up (creamfl_amd/runtime.py: find mode 2, recorded find-db / kernel cache, cudnn.benchmark),
    so that the library convolutions the trunk tests compare against are the ones the product and the bench run.  Must happen
    before the first convolution."""
    os.environ.setdefault('MIOPEN_LOG_LEVEL', '1')       # the fallback warnings of untuned test shapes otherwise bury the log
    os.environ.setdefault('CFL_RUNTIME_TAG', 'tests')    # the tests' odd shapes are recorded apart from the bench's find-db
    from creamfl_amd import runtime
    runtime.configure_env()


_miopen_env()


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu via gpurun)')


# Collection order: the tests that compare a HIP kernel with the oracle / the reference-generated goldens come first,
# the trunk-glue tests (BatchNorm, BERT glue, pooling vs the library kernels) last -- with `-x` a glue failure must not
# hide the parity run.
_ORDER = ['test_oracle_golden', 'test_abi', 'test_host_logic', 'test_gpu_parity', 'test_gpu_configs', 'test_gpu_framework',
          'test_gpu_optimizer', 'test_dist_gloo', 'test_gpu_multirank', 'test_gpu_bert', 'test_gpu_bnorm']


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(mod) if mod in _ORDER else len(_ORDER)
    items.sort(key=key)                      # stable: the order inside a module is kept


@pytest.fixture(autouse=True)
def _library_flags_do_not_leak():
    """creamfl_amd.runtime.configure() -- called whenever an engine or a client trainer is built -- switches cudnn.benchmark on for
    the process (that is the product's set-up).  In a test process that would make every LATER test's library convolutions depend
    on which tests ran before it (MIOpen picks other kernels in benchmark mode; the bf16-ulp comparisons against library results
    in test_gpu_bnorm.py are sensitive to that): every test starts from the flag's default and configure() applies again."""
    import torch
    from creamfl_amd import runtime
    old = torch.backends.cudnn.benchmark
    yield
    torch.backends.cudnn.benchmark = old
    runtime._STATE['torch'] = False


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def golden_files(prefix):
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith('.npz'))


def reference_main_namespace(**overrides):
    """argparse.Namespace exactly as the reference's src/main.py:38-110 would hand it to MMFL: every flag with its default
    (tests/golden/main_flags.json, extracted from main.py by tests/golden/make_golden.py), computed defaults filled the way
    main.py computes them, plus `overrides` (sizes for a small test; build-defined extras such as cnn_type)."""
    import argparse
    import json
    spec = json.load(open(os.path.join(GOLDEN, 'main_flags.json')))
    parser = argparse.ArgumentParser()
    for f in spec['flags']:
        kw = {}
        if 'action' in f:
            kw['action'] = f['action']
        else:
            kw['type'] = {'int': int, 'float': float, 'str': str}[f.get('type', 'str')]
        if 'nargs' in f:
            kw['nargs'] = f['nargs']
        if 'choices' in f:
            kw['choices'] = f['choices']
        default = f.get('default')
        if f.get('computed'):
            default = {'seed': 1234, 'data_root': os.path.expanduser('~/data/')}[f['dest']]
        parser.add_argument(*f['options'], default=default, **kw)
    ns = parser.parse_args([])
    for k, v in overrides.items():
        setattr(ns, k, v)
    return ns, spec


def integration_stubs():
    """The reference-side ctypes binding stubs printed in INTEGRATION.md section B, EXECUTED: every ```python block between the
    heading of section B and the next `## ` heading, in document order, in one namespace (the second stub builds on the first's
    `_lib` / `P`), with the library name resolved to the in-tree build.  Returns that namespace (PairLoss, ClientContrast, _lib)."""
    import re
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = doc[doc.index('## B. Bind the C ABI directly'):]
    nxt = re.search(r'^## (?!B\.)', sec, flags=re.M)
    sec = sec[:nxt.start()] if nxt else sec
    blocks = re.findall(r'```python\n(.*?)```', sec, flags=re.S)
    assert len(blocks) >= 2, 'INTEGRATION.md section B lost its binding stubs'
    from creamfl_amd import _lib
    ns = {'__name__': 'integration_stub'}
    for b in blocks:
        assert 'creamfl_amd' not in b, 'a reference-side stub must not import this package'
        exec(compile(b.replace('ctypes.CDLL("libcreamfl_hip.so")', 'ctypes.CDLL(%r)' % _lib.LIB_PATH), 'INTEGRATION.md#B', 'exec'), ns)
    return ns
