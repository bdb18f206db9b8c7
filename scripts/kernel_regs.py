"""Developer: registers / scratch / occupancy of every kernel of one csrc/*.hip (compiles the device side to assembly).
Usage: python scripts/kernel_regs.py score_topk.hip [name filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'igcn_cf_amd', 'csrc', sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = os.path.join(tempfile.gettempdir(), 'kernel_regs.s')
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-I' + os.path.join(ROOT, 'include'),
                       '-I' + os.path.dirname(src), '-S', '--cuda-device-only', '-o', out, src] + sys.argv[3:], stderr=subprocess.DEVNULL)
txt = open(out).read()
for m in re.finditer(r'\.size\t(\S+), \.Lfunc_end\d+-\1\n\s*; -- End function\n(.*?)\n; (?:NumSgprs|TotalNumSgprs)(.*?)\n\t\.(?:text|section)', txt, re.S):
    name, tail = m.group(1), m.group(2) + m.group(3)
    if flt not in name:
        continue
    def g(k):
        mm = re.search(r'; ' + k + r':\s*(\S+)', tail) or re.search(name + r'\.' + k + r', (\S+)', tail)
        return mm.group(1) if mm else '?'
    demangled = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    print('%-60s vgpr %3s agpr %3s sgpr %3s scratch %4s occupancy %s' % (demangled.split('(')[0][-60:], g('num_vgpr'), g('num_agpr'), g('numbered_sgpr'),
                                                                      g('private_seg_size'), g('Occupancy')))
