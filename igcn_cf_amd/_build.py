"""Builds libigcn_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so sits
next to the package so that it travels to the GPU box with the source tree."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libigcn_hip.so')
ROOF_LIB = os.path.join(PKG, 'libigcn_roof.so')
SOURCES = [f for f in ('spmm.hip', 'bpr.hip', 'score_topk.hip', 'topk_order.hip', 'sampler.hip', 'csr_util.hip')
           if os.path.exists(os.path.join(CSRC, f))]
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-Wall', '-Wno-unused-function',
         '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
# per source: the d = 128 candidate sweep runs one wave per SIMD with 396 registers; with MFMA results in VGPRs (this option)
# the selection reads the accumulators where they are, the item tiles and user planes spill over into AGPRs instead
# (left to itself the allocator puts the accumulators in AGPRs and copies 32 of them to VGPRs every tile step)
EXTRA_FLAGS = {'score_topk.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1']}


def _newer(src_list, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in src_list)


def _msgpack_unpack(buf):
    """The subset of msgpack the AMDGPU metadata note uses (maps, arrays, strings, ints, bools, nil, floats, bin): no third-party
    module is needed to build (the `msgpack` package is used instead when it is importable; the test suite compares the two)."""
    import struct

    def take(o):
        b = buf[o]
        if b <= 0x7f:
            return b, o + 1
        if b >= 0xe0:
            return b - 0x100, o + 1
        if 0x80 <= b <= 0x8f:
            return take_map(o + 1, b & 0x0f)
        if 0x90 <= b <= 0x9f:
            return take_array(o + 1, b & 0x0f)
        if 0xa0 <= b <= 0xbf:
            n = b & 0x1f
            return buf[o + 1:o + 1 + n].decode('utf-8', 'replace'), o + 1 + n
        if b == 0xc0:
            return None, o + 1
        if b in (0xc2, 0xc3):
            return b == 0xc3, o + 1
        if b in (0xc4, 0xc5, 0xc6, 0xd9, 0xda, 0xdb):                       # bin 8/16/32, str 8/16/32
            w = {0xc4: 1, 0xc5: 2, 0xc6: 4, 0xd9: 1, 0xda: 2, 0xdb: 4}[b]
            n = int.from_bytes(buf[o + 1:o + 1 + w], 'big')
            raw = bytes(buf[o + 1 + w:o + 1 + w + n])
            return (raw if b <= 0xc6 else raw.decode('utf-8', 'replace')), o + 1 + w + n
        if b == 0xca:
            return struct.unpack_from('>f', buf, o + 1)[0], o + 5
        if b == 0xcb:
            return struct.unpack_from('>d', buf, o + 1)[0], o + 9
        if 0xcc <= b <= 0xcf:
            w = 1 << (b - 0xcc)
            return int.from_bytes(buf[o + 1:o + 1 + w], 'big'), o + 1 + w
        if 0xd0 <= b <= 0xd3:
            w = 1 << (b - 0xd0)
            return int.from_bytes(buf[o + 1:o + 1 + w], 'big', signed=True), o + 1 + w
        if b in (0xdc, 0xdd):
            w = 2 if b == 0xdc else 4
            return take_array(o + 1 + w, int.from_bytes(buf[o + 1:o + 1 + w], 'big'))
        if b in (0xde, 0xdf):
            w = 2 if b == 0xde else 4
            return take_map(o + 1 + w, int.from_bytes(buf[o + 1:o + 1 + w], 'big'))
        raise ValueError('msgpack type 0x%02x is not part of the AMDGPU metadata subset' % b)

    def take_array(o, n):
        out = []
        for _ in range(n):
            v, o = take(o)
            out.append(v)
        return out, o

    def take_map(o, n):
        out = {}
        for _ in range(n):
            k, o = take(o)
            v, o = take(o)
            out[k] = v
        return out, o
    return take(0)[0]


def _unpack_note(desc):
    try:
        import msgpack
    except ImportError:
        return _msgpack_unpack(desc)
    return msgpack.unpackb(desc, raw=False, strict_map_key=False)


def kernel_metadata(lib_path=None):
    """name -> metadata dict of every gfx950 kernel in a built library (.private_segment_fixed_size, .vgpr_count, .agpr_count,
    .sgpr_count, .group_segment_fixed_size ...): the code objects are read out of the clang offload bundles inside the .so and
    their NT_AMDGPU_METADATA notes decoded — no GPU, no external tool."""
    import struct
    blob = open(lib_path or LIB, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    kernels = {}
    at = blob.find(magic)
    while at >= 0:
        n, = struct.unpack_from('<Q', blob, at + 24)
        o = at + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from('<QQQ', blob, o)
            o += 24
            triple = blob[o:o + tlen].decode()
            o += tlen
            if size == 0 or 'amdgcn' not in triple:
                continue
            elf = blob[at + off: at + off + size]
            assert elf[:4] == b'\x7fELF' and elf[4] == 2, 'not a 64-bit ELF code object'
            shoff, = struct.unpack_from('<Q', elf, 0x28)
            shentsize, shnum = struct.unpack_from('<HH', elf, 0x3A)
            for i in range(shnum):
                sh = shoff + i * shentsize
                sh_type, = struct.unpack_from('<I', elf, sh + 4)
                if sh_type != 7:                                   # SHT_NOTE
                    continue
                s_off, s_size = struct.unpack_from('<QQ', elf, sh + 0x18)
                q = s_off
                while q + 12 <= s_off + s_size:
                    namesz, descsz, ntype = struct.unpack_from('<III', elf, q)
                    q += 12
                    name = elf[q:q + namesz].rstrip(b'\0')
                    q += (namesz + 3) // 4 * 4
                    desc = elf[q:q + descsz]
                    q += (descsz + 3) // 4 * 4
                    if ntype == 32 and name == b'AMDGPU':          # NT_AMDGPU_METADATA (msgpack)
                        for k in _unpack_note(desc).get('amdhsa.kernels', []):
                            kernels[k['.name']] = k
        at = blob.find(magic, at + 1)
    return kernels


def scratch_report(lib_path=None):
    """Kernels of the built library that carry a private segment (scratch): [(demangled name, bytes per lane)].  The library is
    meant to have none — a kernel with scratch cannot be replayed from a HIP graph on a queue that never ran one (csrc/common.h)."""
    # (rocPRIM's radix sort kernels — igcn_csr_transpose, a graph-build utility — do carry one; that entry point refuses a capturing
    # stream, IGCN_E_CAPTURE)
    bad = [(name, int(k.get('.private_segment_fixed_size', 0))) for name, k in sorted(kernel_metadata(lib_path).items())
           if (int(k.get('.private_segment_fixed_size', 0)) > 0 or k.get('.uses_dynamic_stack')) and 'rocprim' not in name]
    if bad:
        try:
            dem = subprocess.run(['c++filt'] + [b[0] for b in bad], capture_output=True, text=True).stdout.split('\n')
            bad = [(dem[i].split('(')[0] or bad[i][0], bad[i][1]) for i in range(len(bad))]
        except OSError:
            pass
    return bad


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(ROOT, 'include', 'igcn_hip.h'))
    srcs = [os.path.join(CSRC, name) for name in SOURCES]
    # The shipped library is current (the GPU box gets the .so but not the objects: *.o is in .gpurunignore): nothing to do there
    main_current = not force and not _newer(srcs + headers, LIB)
    objs = []
    for name in ([] if main_current else SOURCES):
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            raise FileNotFoundError(src)
        obj = os.path.join(CSRC, name.replace('.hip', '.o'))
        if force or _newer([src] + headers, obj):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(name, []) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if not main_current and (force or _newer(objs, LIB)):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    bad = scratch_report(LIB)
    if bad:
        raise RuntimeError('kernels with a private segment in %s (none is allowed: HIP-graph replays fault on them): %s' % (LIB, bad))
    # measurement kernels of bench.py (the in-run gather roof): a library of their own, never loaded by the product
    roof_src = os.path.join(CSRC, 'roof_probe.hip')
    if os.path.exists(roof_src) and (force or _newer([roof_src], ROOF_LIB)):
        cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-o', ROOF_LIB, roof_src]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
