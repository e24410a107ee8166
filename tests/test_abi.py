"""The C ABI: include/strive_hip.h, the ctypes binding and the hipcc-built library stay in sync.  No compute
is issued here (there is no GPU in the build container); loading the library and resolving symbols is enough."""
import os
import re

import pytest

from strive_amd import _lib as L

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def header_functions():
    hdr = open(os.path.join(REPO, 'include', 'strive_hip.h')).read()
    return sorted(set(re.findall(r'\b(strive_[a-z0-9_]+)\s*\(', hdr)))


def test_binding_lists_every_header_symbol():
    assert header_functions() == sorted(L.PROTOTYPES.keys())


def test_hip_library_builds_loads_and_exports_everything():
    import __graft_entry__ as ge
    path = ge.build(verbose=False)
    assert os.path.exists(path)
    lib = L.StriveLib(path)            # raises if any symbol is missing
    assert lib.missing == []
    assert lib.query("strive_abi_version") == L.ABI_VERSION == 17


def test_library_reads_no_environment_variable():
    """ABI 17: every switch of the library is a declared option (strive_set_option); no getenv anywhere in csrc."""
    csrc = os.path.join(REPO, 'strive_amd', 'csrc')
    hits = []
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith(('.hip', '.h')):
            for i, line in enumerate(open(os.path.join(csrc, fn)), 1):
                code = line.split('//')[0]
                if re.search(r'\bgetenv\s*\(', code) or 'secure_getenv' in code:
                    hits.append('%s:%d' % (fn, i))
    assert hits == []


def test_options_are_declared_defaulted_and_range_checked(monkeypatch):
    import __graft_entry__ as ge
    lib = L.StriveLib(ge.build(verbose=False))
    names = lib.option_names()
    hdr = open(os.path.join(REPO, 'include', 'strive_hip.h')).read()
    assert len(names) == len(set(names)) >= 19
    for n in names:                                    # every option is documented in the header
        assert re.search(r'\b%s\b' % n, hdr), n
    lib.query('strive_reset_options')
    assert lib.get_option('conv_wsx') == 1 and lib.get_option('cnn_small_batch') == 96 and lib.get_option('sweep_step') == -1
    lib.set_option('cnn_small_batch', 0)
    assert lib.get_option('cnn_small_batch') == 0
    with pytest.raises(L.StriveHipError):
        lib.set_option('cnn_chunk', 4)                 # below the option's range
    with pytest.raises(L.StriveHipError):
        lib.set_option('no_such_option', 1)
    # the host maps STRIVE_<NAME> onto the options (and back to the defaults when the variable goes away)
    monkeypatch.setenv('STRIVE_SCENE_KERNELS', '0')
    lib.sync_options_from_env()
    assert lib.get_option('scene_kernels') == 0 and lib.get_option('cnn_small_batch') == 96
    monkeypatch.delenv('STRIVE_SCENE_KERNELS')
    lib.sync_options_from_env()
    assert lib.get_option('scene_kernels') == 1


def test_every_entry_point_cites_the_reference():
    hdr = open(os.path.join(REPO, 'include', 'strive_hip.h')).read()
    assert hdr.count('reference src/') >= 12


def test_product_has_no_cpu_path():
    import torch
    from strive_amd import ops
    with pytest.raises(L.StriveHipError):
        ops.map_crop(None, torch.zeros((1, 4)), torch.zeros((1,), dtype=torch.long))


def test_product_never_imports_the_oracle():
    import subprocess
    out = subprocess.run(['grep', '-rln', '--include=*.py', '-E', r'^\s*(from|import)\s+oracle', os.path.join(REPO, 'strive_amd')],
                         capture_output=True, text=True).stdout.strip()
    assert out == '', 'strive_amd imports oracle: %s' % out


def test_no_packed_fp32_ops_in_the_device_code():
    """MI355X returned wrong values in lanes 48-63 for v_pk_add_f32 with crossed op_sel halves when >= 4 waves shared
    a SIMD with MFMA-issuing neighbours (tools/pk_waw_probe.hip, profiles/r01_pk_add_opsel_probe.txt).  The library
    is therefore built without auto-formed packed fp32 arithmetic; this checks the disassembly of what was built."""
    from strive_amd import build
    counts = build.audit_packed_ops()
    if counts is None:
        pytest.skip('llvm-objdump not available')
    assert counts.get('_code_objects', 0) >= 1 and counts.get('_mfma', 0) > 0, 'no gfx950 code objects found: %r' % (counts,)
    packed = {k: v for k, v in counts.items() if not k.startswith('_')}
    assert not packed, 'packed fp32 VALU ops in the device code: %r' % (packed,)
