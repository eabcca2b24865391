#!/usr/bin/env python
"""Run-length summary of the memory / MFMA / barrier instruction stream of gfx950 kernels.

usage: hipcc ... -save-temps -c file.hip ; python tools/isa_summary.py file-hip-amdgcn-amd-amdhsa-gfx950.s [name-substring]

Used to check, without a GPU, that a "one round trip" kernel really issues all of its global loads
before the first s_waitcnt vmcnt and that the MFMA loop has no memory waits inside.
"""
import re
import sys

PAT = re.compile(r'(global_load_\w+|buffer_load_\w+|s_waitcnt|v_mfma_\w+|s_barrier|ds_write\w*|ds_read\w*|'
                 r'global_store_\w+|buffer_store_\w+|s_cbranch\w+|s_branch|scratch_\w+)')


def summarize(body):
    seq = []
    for line in body.split('\n'):
        line = line.strip()
        m = PAT.match(line)
        if not m:
            continue
        tok = m.group(1)
        if tok == 's_waitcnt':
            tok = 'W(' + line.split(None, 1)[1].split(';')[0].strip() + ')'
        elif tok.startswith('global_load') or tok.startswith('buffer_load'):
            tok = 'LD' + tok.rsplit('_', 1)[1].replace('dword', '')
            tok = tok if tok != 'LD' else 'LD1'
        elif tok.startswith('global_store') or tok.startswith('buffer_store'):
            tok = 'ST'
        elif tok.startswith('v_mfma'):
            tok = 'MFMA'
        elif tok.startswith('ds_write'):
            tok = 'dsW'
        elif tok.startswith('ds_read'):
            tok = 'dsR'
        elif tok.startswith('s_cbranch') or tok == 's_branch':
            tok = 'BR'
        seq.append(tok)
    out, prev, c = [], None, 0
    for t in seq:
        if t == prev:
            c += 1
        else:
            if prev:
                out.append('%sx%d' % (prev, c) if c > 1 else prev)
            prev, c = t, 1
    if prev:
        out.append('%sx%d' % (prev, c) if c > 1 else prev)
    return ' '.join(out)


def main():
    src = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    parts = re.split(r'\n(?=_Z\w+:)', src)
    for part in parts:
        name = part.split(':', 1)[0]
        if not name.startswith('_Z') or want not in name:
            continue
        end = part.find('.Lfunc_end')
        body = part[:end] if end > 0 else part
        print(name[:150])
        print('   ', summarize(body))
        print()


if __name__ == '__main__':
    main()
