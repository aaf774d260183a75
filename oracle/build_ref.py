"""Builds oracle/lib/libs3ref_<isa>.so from oracle/conv_ref.c (TEST / BASELINE
INFRASTRUCTURE).  Two portable ISA levels are built so the library compiled in
the GPU-less build container also runs on the GPU box's host CPU:
x86-64-v3 (AVX2 + FMA) and x86-64-v4 (AVX-512); ``load()`` picks the widest one
the running CPU supports.  ``native=True`` additionally tries -march=native
into a temp dir (bench.py's cpu_baseline leg, on the box it runs on)."""
import ctypes as C
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'conv_ref.c')
LIBDIR = os.path.join(HERE, 'lib')
LEVELS = ('x86-64-v3', 'x86-64-v4')


def _compile(march, out):
    cmd = ['gcc', '-O3', '-fopenmp', '-fPIC', '-shared', f'-march={march}',
           '-ffp-contract=fast', SRC, '-o', out]
    subprocess.run(cmd, check=True, capture_output=True)


def build():
    os.makedirs(LIBDIR, exist_ok=True)
    built = []
    for lv in LEVELS:
        out = os.path.join(LIBDIR, f'libs3ref_{lv.replace("-", "_")}.so')
        if not os.path.exists(out) or \
                os.path.getmtime(out) < os.path.getmtime(SRC):
            _compile(lv, out)
        built.append(out)
    return built


def _cpu_flags():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('flags'):
                    return set(line.split(':', 1)[1].split())
    except OSError:
        pass
    return set()


def load(native=False):
    """-> (ctypes library, description of the build that was loaded)"""
    if native:
        try:
            out = os.path.join(tempfile.mkdtemp(prefix='s3ref_'),
                               'libs3ref_native.so')
            _compile('native', out)
            return _bind(C.CDLL(out)), 'gcc -O3 -fopenmp -march=native'
        except Exception:
            pass
    flags = _cpu_flags()
    want = 'x86-64-v4' if {'avx512f', 'avx512bw', 'avx512vl',
                           'avx512dq'} <= flags else 'x86-64-v3'
    path = os.path.join(LIBDIR, f'libs3ref_{want.replace("-", "_")}.so')
    if not os.path.exists(path):
        build()
    return _bind(C.CDLL(path)), f'gcc -O3 -fopenmp -march={want}'


def _bind(lib):
    i64, i32, pf = C.c_int64, C.c_int, C.POINTER(C.c_float)
    lib.s3ref_conv_valid.restype = i32
    lib.s3ref_conv_valid.argtypes = [pf, i64, i64, i64, i64, i64, pf, pf, i32,
                                     i32, i32, i32, i32, i32, i64, pf]
    lib.s3ref_threads.restype = i32
    return lib


if __name__ == '__main__':
    print(build())
