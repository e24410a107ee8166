"""TEST INFRASTRUCTURE ONLY: compile strive_amd/csrc/*.hip, unmodified, as host C++ against
tests/hipemu/hip/hip_runtime.h so kernel logic can be exercised without a GPU.  The product loader
never loads this library."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
CSRC = os.path.join(REPO, 'strive_amd', 'csrc')
OUT = os.path.join(HERE, '_build')
LIB = os.path.join(OUT, 'libstrive_emu.so')
CXX = os.environ.get('EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
FLAGS = ['-x', 'c++', '-std=c++17', '-O2', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-I', HERE,
         '-Wno-unused-value', '-Wno-unknown-attributes', '-Wno-deprecated-declarations']


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip')) + [os.path.join(HERE, 'hipemu.cpp')]
    h = hashlib.sha256()
    for root in (CSRC, HERE, os.path.join(HERE, 'hip'), os.path.join(REPO, 'include')):
        for fn in sorted(os.listdir(root)):
            p = os.path.join(root, fn)
            if os.path.isfile(p) and fn.endswith(('.hip', '.h', '.cpp')):
                h.update(fn.encode())
                h.update(open(p, 'rb').read())
    stamp = h.hexdigest()
    sf = LIB + '.stamp'
    if not force and os.path.exists(LIB) and os.path.exists(sf) and open(sf).read() == stamp:
        return LIB
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + '.o')
        procs.append((s, subprocess.Popen([CXX] + FLAGS + ['-c', s, '-o', o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors='replace'))
            raise RuntimeError('emu compile failed: %s' % s)
    subprocess.check_call([CXX, '-shared', '-fPIC', '-o', LIB] + objs)
    open(sf, 'w').write(stamp)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
