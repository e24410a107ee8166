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

SOURCES = ['capi.hip', 'map_crop.hip', 'map_cnn.hip', 'mlp_gnn.hip', 'rollout.hip', 'losses.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-fast-math', '-ffp-contract=on',
         '-fhip-fp32-correctly-rounded-divide-sqrt', '-Wno-unused-result', '-Wno-unused-value',
         # no auto-formed v_pk_*_f32: see DESIGN.md "packed-fp32 write-after-write" (a half-dead packed result
         # followed by a scalar write of the same VGPR lost the race for lanes 48-63 when two workgroups shared a CU)
         '-fno-slp-vectorize']


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
