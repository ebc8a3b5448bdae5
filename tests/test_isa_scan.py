"""Build check (CPU): no built kernel contains the packed-fp32 operand-select form that is unreliable on MI355X (profiles/NOTES.md,
round 5; tools/isa_opsel_scan.py).  The Makefile runs the same scan; this keeps it in the test suite."""
import glob
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))


def test_scan_logic_on_a_disassembly_fragment():
    import isa_opsel_scan as S
    text = '''0000000000001000 <kernel_a>:
	v_pk_fma_f32 v[70:71], v[70:71], v[26:27], v[30:31] op_sel:[0,1,1]      // 000000001000: D3B04046
	v_pk_fma_f32 v[60:61], v[60:61], v[26:27], v[30:31] op_sel_hi:[1,0,0]   // 000000001008: D3B0003C
	v_pk_fma_f32 v[48:49], s[12:13], v[58:59], v[48:49] op_sel:[1,0,0]      // 000000001010: D3B00030
	v_pk_add_f32 v[2:3], v[2:3], v[8:9] op_sel:[0,1]                        // 000000001018: D3B20002
	v_pk_mul_f32 v[4:5], v[4:5], v[6:7]                                     // 000000001020: D3B10004
'''
    flagged, scalar = S.scan(text)
    assert [k for k, _ in flagged] == ['kernel_a', 'kernel_a'] and 'op_sel:[0,1,1]' in flagged[0][1] and 'v_pk_add_f32' in flagged[1][1]
    assert len(scalar) == 1 and 's[12:13]' in scalar[0][1]


def test_no_built_kernel_selects_the_high_register_of_a_vector_pair_for_a_low_half():
    import isa_opsel_scan as S
    objs = sorted(glob.glob(os.path.join(REPO, 'joint-vae_amd', 'csrc', 'build', '*.o')))
    objs = [o for o in objs if not os.path.basename(o).startswith('stamps_')]
    if not objs or not os.path.exists(S.LLVM + '/llvm-objdump'):
        pytest.skip('no built objects / no llvm-objdump here (run __graft_entry__.build() first)')
    assert len(objs) >= 20
    bad = []
    for o in objs:
        flagged, _ = S.scan(S.device_isa(o))
        bad += [(os.path.basename(o),) + f for f in flagged]
    assert not bad, bad[:5]
