"""Small stand-ins for the reference's `utils.*` helpers that the hot-path modules import.

The hot-path modules never import the reference's `utils` package (it pulls torchvision, pandas, ...): the one
helper they need (`texify_str`, used by `__format__(..., 'x')` of Sigma / Optimizer) lives here, so overlay and
standalone use behave the same.  See INTEGRATION.md.
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
