"""
Instruction mix of the kernels of one HIP source file, read off the gfx950 ISA that hipcc emits: per kernel the registers,
spills and, for every loop that contains MFMAs, the number of MFMA / other VALU / scalar / LDS / vector-memory / scratch
instructions of its body.  fp64 MFMAs and VALU instructions share one issue pipe on this part, so "VALU per MFMA" is the
figure DESIGN.md 4.1 / 4.2 argue with.

    python3 tools/isa_mix.py grates_amd/csrc/synthesis_rot.hip [kernel-name-substring]        (needs hipcc only, no GPU)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(t):
    if t.startswith('v_mfma'):
        return 'mfma'
    if t.startswith('v_'):
        return 'valu'
    if t.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_cbranch', 's_branch', 's_endpgm')):
        return None
    if t.startswith('s_load') or t.startswith('s_buffer_load'):
        return 'smem'
    if t.startswith('s_'):
        return 'salu'
    if t.startswith('ds_'):
        return 'lds'
    if t.startswith('scratch_'):
        return 'scratch'
    if t.startswith(('global_', 'buffer_', 'flat_')):
        return 'vmem'
    return None


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-I' + os.path.join(ROOT, 'include'),
               '-S', '--cuda-device-only', '-o', out, src]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split('\n')
    meta = {}
    name = None
    for l in lines:                                    # kernel descriptors at the end of the file
        m = re.match(r'\s+\.name:\s+(\S+)', l)
        if m:
            name = m.group(1)
        for key in ('vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'private_segment_fixed_size'):
            m = re.match(r'\s+\.' + key + r':\s+(\d+)', l)
            if m:
                meta.setdefault('pending', {})[key] = int(m.group(1))
        if name and 'pending' in meta and re.match(r'\s+\.wavefront_size', l):
            meta[name] = meta.pop('pending')
            name = None
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and want in l]
    for i in starts:
        kernel = lines[i].split(':')[0]
        j = i
        while j < len(lines) and not lines[j].startswith('.Lfunc_end'):
            j += 1
        body = lines[i:j]
        print(subprocess.run(['c++filt', kernel], capture_output=True, text=True).stdout.strip()[:150])
        if kernel in meta:
            print('   registers:', meta[kernel])
        for h, l in enumerate(body):
            if 'Loop Header' not in l:
                continue
            label = l.split(':')[0].strip()
            ends = [k for k, x in enumerate(body) if k > h and re.search(r's_cbranch_\w+ ' + re.escape(label) + r'\s*$', x)]
            if not ends:
                continue
            mix = {}
            for x in body[h:ends[-1] + 1]:
                t = x.strip()
                if not t or t.startswith((';', '.')):
                    continue
                c = classify(t)
                if c:
                    mix[c] = mix.get(c, 0) + 1
            if mix.get('mfma'):
                depth = re.search(r'Depth=(\d+)', l)
                print('   loop %-10s depth %s: %s   (VALU per MFMA %.2f)' % (label, depth.group(1) if depth else '?',
                      '  '.join('%s %d' % kv for kv in sorted(mix.items())), mix.get('valu', 0) / mix['mfma']))


if __name__ == '__main__':
    main()
