"""Developer: compile ONE instantiation of score_topk_kernel to assembly (seconds instead of the file's minute) and print its
registers / scratch / occupancy.  Usage: python scripts/dev_one_variant.py 64,2,true,3,false [-DNAME ...]   (asm kept in /tmp/ov/)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'igcn_cf_amd', 'csrc')
variant = sys.argv[1]
src = open(os.path.join(CSRC, 'score_topk.hip')).read()
cut = src.index('template <int D, int NG, bool FULL, int MODE = 0, bool BOUNDED = false>\nstatic int launch_topk')
one = src[:cut] + 'template __global__ void score_topk_kernel<%s>(const TopkArgs);\n}\n' % variant
os.makedirs('/tmp/ov', exist_ok=True)
tag = variant.replace(',', '_')
path = '/tmp/ov/one_%s.hip' % tag
open(path, 'w').write(one)
out = '/tmp/ov/one_%s.s' % tag
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
                       '-mllvm', '-amdgpu-mfma-vgpr-form=1', '-S', '--cuda-device-only', '-o', out, path] + sys.argv[2:], stderr=subprocess.DEVNULL)
txt = open(out).read()
m = re.search(r'(_ZN4igcn17score_topk_kernel\S+):.*?\n; (?:NumSgprs|TotalNumSgprs)(.*?)\n\t\.(?:text|section)', txt, re.S)
body = txt[m.start():m.end()]
def g(k):
    mm = re.search(r'; ' + k + r':\s*(\S+)', body) or re.search(r'\.' + k + r', (\S+)', body)
    return mm.group(1) if mm else '?'
print('score_topk_kernel<%s>: vgpr %s agpr %s sgpr %s scratch %s occupancy %s  scratch ops %d  writelane %d  mfma %d  lines %d' % (
    variant, g('num_vgpr'), g('num_agpr'), g('numbered_sgpr'), g('private_seg_size'), g('Occupancy'),
    len(re.findall(r'scratch_', body)), len(re.findall(r'v_writelane', body)), len(re.findall(r'v_mfma', body)), body.count('\n')))
