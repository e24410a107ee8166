"""Build libstrive_hip.so for gfx950 with hipcc (in-tree, next to this file).

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the tree.  No torch headers are involved: the library
is a plain C ABI (include/strive_hip.h) loaded through ctypes.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libstrive_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

SOURCES = ['capi.hip', 'map_crop.hip', 'map_raster.hip', 'map_cnn.hip', 'mlp_gnn.hip', 'rollout.hip', 'losses.hip', 'planner.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-fast-math', '-ffp-contract=on',
         '-fhip-fp32-correctly-rounded-divide-sqrt', '-Wno-unused-result', '-Wno-unused-value',
         # no auto-formed v_pk_*_f32: on MI355X a v_pk_add_f32 with crossed op_sel halves returned wrong values in
         # lanes 48-63 whenever >= 4 waves shared the SIMD and a neighbour issued MFMAs (tools/pk_waw_probe.hip,
         # profiles/r01_pk_add_opsel_probe.txt); audit_packed_ops() checks the built library
         '-fno-slp-vectorize', '-fno-vectorize']


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stamp():
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(HERE, '..', 'include')):
        for fn in sorted(os.listdir(root)):
            if fn.endswith(('.hip', '.h')):
                with open(os.path.join(root, fn), 'rb') as f:
                    h.update(fn.encode())
                    h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def audit_packed_ops(lib=LIB):
    """Disassemble the gfx950 code objects inside `lib`; return {mnemonic: count} of packed-fp32 VALU ops."""
    import re
    import shutil
    import subprocess
    import tempfile
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        return None
    counts = {}
    with tempfile.TemporaryDirectory() as td:
        cp = os.path.join(td, os.path.basename(lib))
        shutil.copy(lib, cp)
        subprocess.run([objdump, '--offloading', cp], cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        for fn in sorted(os.listdir(td)):
            if 'amdgcn' not in fn:
                continue
            dis = subprocess.run([objdump, '-d', os.path.join(td, fn)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 check=False).stdout.decode('utf-8', 'replace')
            for m in re.finditer(r'\b(v_pk_(?:add|mul|fma)_f32)\b', dis):
                counts[m.group(1)] = counts.get(m.group(1), 0) + 1
            counts['_code_objects'] = counts.get('_code_objects', 0) + 1
            counts['_mfma'] = counts.get('_mfma', 0) + len(re.findall(r'\bv_mfma_', dis))
    return counts


def build(force=False, verbose=True):
    stamp_file = LIB + '.stamp'
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, '_obj'), exist_ok=True)
    for src in _sources():
        obj = os.path.join(HERE, '_obj', os.path.basename(src) + '.o')
        cmd = [HIPCC] + FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors='replace'))
            raise RuntimeError('hipcc failed on %s' % src)
        elif verbose and out.strip():
            sys.stderr.write(out.decode(errors='replace'))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp_file, 'w') as f:
        f.write(stamp)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
