"""Small stand-ins for the reference's `utils.*` helpers that the hot-path modules import.

In *overlay* mode (our cvae.py + module/ copied over a reference checkout) the reference's own `utils`
package is importable and is used; in *standalone* mode (this repository alone, e.g. on the GPU box) these
minimal equivalents are used instead.  See INTEGRATION.md.
"""
import re


def texify_str(s, num=False, space=None, underscore=None, verbatim=False):
    """LaTeX-friendly rendering used by `__format__(..., 'x')` of Sigma / Optimizer / priors."""
    if not isinstance(s, str):
        return s
    try:
        float(s)
    except ValueError:
        pass
    else:
        return s
    out = s.replace('->', '\\ensuremath{\\to{}}')
    if space:
        out = out.replace(' ', space)
    if underscore:
        out = out.replace('_', underscore)
    if num:
        out = re.sub(r'[-+]?\d*\.\d+', lambda m: '\\num{' + m.group(0) + '}', out)
    return out


try:                                   # overlay mode: defer to the reference's helper
    from utils.print_log import texify_str as _ref_texify   # noqa: F401
    texify_str = _ref_texify
except Exception:                      # standalone mode
    pass
