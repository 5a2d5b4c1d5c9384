"""The C-ABI library builds, loads, and exports exactly what include/ndcn_hip.h declares.  No compute
calls (there is no GPU in the CPU suite)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'ndcn_hip.h')).read()
    return sorted(set(re.findall(r'NDCN_API[^;(]*?\b(ndcn_\w+)\s*\(', text)))


def test_header_and_binding_agree():
    from ndcn_amd import _lib
    assert header_functions() == sorted(_lib.SIGNATURES)


def test_library_loads_and_exports_every_symbol():
    from ndcn_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'build it: python -c "import __graft_entry__ as g; g.build()"'
    lib = _lib.load()                      # resolves every symbol of SIGNATURES
    assert lib.ndcn_abi_version() == _lib.ABI_VERSION
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r' T (ndcn_\w+)', out)))
    assert exported == header_functions()


def test_library_has_gfx950_code_object():
    from ndcn_amd import _lib
    out = subprocess.run(['strings', '-a', _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert 'gfx950' in out


def test_host_tensors_are_refused():
    import torch
    from ndcn_amd import _lib, hip
    from ndcn_amd import torchdiffeq as ode
    with pytest.raises(_lib.NdcnHipError):
        hip.combine(torch.ones(4), [torch.ones(4)], [1.0])
    with pytest.raises(_lib.NdcnHipError):
        ode.odeint(lambda t, y: -y, torch.ones(4), torch.tensor([0., 1.]), method='euler')


def test_argument_errors_match_reference():
    import torch
    from ndcn_amd import torchdiffeq as ode
    f = lambda t, y: -y
    y0 = torch.ones(3)
    with pytest.raises(TypeError):
        ode.odeint(f, torch.ones(3, dtype=torch.int64), torch.tensor([0., 1.]), method='euler')
    with pytest.raises(TypeError):
        ode.odeint(f, y0, torch.tensor([0, 1]), method='euler')
    with pytest.raises(ValueError):
        ode.odeint(f, y0, torch.tensor([0., 1.]), options={'step_size': 0.1})
    with pytest.raises(KeyError):
        ode.odeint(f, y0, torch.tensor([0., 1.]), method='nope')
    with pytest.raises(NotImplementedError):
        ode.odeint(f, y0, torch.tensor([0., 1.]), method='tsit5')
    with pytest.raises(AssertionError):
        ode.odeint(f, [y0], torch.tensor([0., 1.]), method='euler')
    with pytest.raises(ValueError):
        ode.odeint_adjoint(f, y0, torch.tensor([0., 1.]))          # func must be an nn.Module (adjoint.py:109-110)


def test_dropin_install_redirects_the_reference_import_lines():
    """after ndcn_amd.dropin.install() the reference drivers' own import lines
    (heat_dynamics.py:12-14, dgnn.py:8,19,22) resolve to this package - checked in a fresh interpreter."""
    import subprocess
    import sys
    code = '''
import sys
sys.path.insert(0, %r)
import ndcn_amd.dropin
ndcn_amd.dropin.install(models=True)
import torchdiffeq as ode
from neural_dynamics import *
from models import *
import ndcn_amd.neural_dynamics as nd, ndcn_amd.torchdiffeq as td, ndcn_amd.models as md
assert ode is td and ode.odeint is td.odeint and ode.odeint_adjoint is td.odeint_adjoint
assert sys.modules['neural_dynamics'] is nd and sys.modules['models'] is md
assert ODEFunc is nd.ODEFunc and ODEBlock is nd.ODEBlock and ODEBlock2 is nd.ODEBlock2 and NDCN is nd.NDCN
assert GCN is md.GCN and GraphConvolution is md.GraphConvolution      # the later star import wins, as in Python
import torch
m = NDCN(input_size=1, hidden_size=4, A=torch.eye(3), num_classes=1)
assert sorted(m.state_dict()) == sorted(['input_layer.0.weight', 'input_layer.0.bias', 'input_layer.2.weight',
    'input_layer.2.bias', 'neural_dynamic_layer.odefunc.wt.weight', 'neural_dynamic_layer.odefunc.wt.bias',
    'output_layer.weight', 'output_layer.bias'])
print('ok')
''' % ROOT
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-2000:]


def test_documented_bindings_state_the_current_abi_version():
    """the stub a reference maintainer would copy (examples/ndcn_hip_binding.py) asserts the ABI version:
    it must be the library's"""
    from ndcn_amd import _lib
    for rel in ('examples/ndcn_hip_binding.py', 'INTEGRATION.md'):
        text = open(os.path.join(ROOT, rel)).read()
        found = re.findall(r'ndcn_abi_version\(\) == (\d+)', text)
        assert found and all(int(v) == _lib.ABI_VERSION for v in found), (rel, found, _lib.ABI_VERSION)
