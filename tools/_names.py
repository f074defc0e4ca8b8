"""Kernel names as rocprofv3 reports them -> short readable ones.  The profiler demangles most names itself; the ones whose template
arguments include __bf16 (Itanium `DF16b`) come out mangled, and binutils' c++filt does not know that type either: the few argument
kinds our kernels use (integers, bools, float, __bf16) are decoded here."""
import re


def _template_args(s):
    out, i = [], 0
    while i < len(s) and s[i] != 'E':
        if s.startswith('DF16b', i):
            out.append('bf16')
            i += 5
        elif s[i] == 'f':
            out.append('float')
            i += 1
        elif s[i] == 'L':
            m = re.match(r'L([a-z])(n?\d+)E', s[i:])
            if not m:
                return None
            v = m.group(2).replace('n', '-')
            out.append({'0': 'false', '1': 'true'}[v] if m.group(1) == 'b' else v)
            i += m.end()
        else:
            return None
    return out


def pretty(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', name) or re.match(r'_Z(\d+)', name)
    if m:
        n = int(m.group(1))
        base, rest = name[m.end():m.end() + n], name[m.end() + n:]
        if rest.startswith('I'):
            args = _template_args(rest[1:])
            return '%s<%s>' % (base, ', '.join(args)) if args is not None else base
        return base
    return re.sub(r'\(.*$', '', name)
