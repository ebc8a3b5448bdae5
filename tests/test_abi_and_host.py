"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/jvae_hip.h declares,
the host logic (layer DSL, shapes, state_dict contract, optimiser formatting) matches the reference's goldens,
and the product path refuses to compute without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest
import torch

from oracle.cases import CASES, DSL_CASES, get_case

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(REPO, 'include', 'jvae_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(jvae_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from jvae_hip import lib
    names = _header_functions()
    assert len(names) >= 20
    handle = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f'{n} declared in include/jvae_hip.h but not exported'
    assert sorted(lib.exported_symbols()) == names, 'ctypes signature table out of sync with the header'
    assert b'gfx950' in lib.load().jvae_version()


def test_ctypes_signatures_match_the_header():
    """Argument count and pointer / integer / float class of every ctypes signature against the prototype in the header
    (a mismatch would pass garbage registers to the kernels' launchers)."""
    import ctypes
    from jvae_hip import lib
    src = open(os.path.join(REPO, 'include', 'jvae_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = dict(re.findall(r'\b(jvae_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S))
    assert set(protos) == set(lib._SIGS)
    for name, (res, args) in lib._SIGS.items():
        plist = [a.strip() for a in protos[name].replace('\n', ' ').split(',')]
        if plist == ['void'] or plist == ['']:
            plist = []
        assert len(plist) == len(args), (name, len(plist), len(args))
        for decl, ct in zip(plist, args):
            is_ptr = '*' in decl
            if is_ptr:
                assert ct in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(ct, '_type_') and hasattr(ct, 'contents'), (name, decl)
            elif re.match(r'^(const\s+)?float\b', decl):
                assert ct is ctypes.c_float, (name, decl)
            elif re.match(r'^(const\s+)?size_t\b', decl):
                assert ct is ctypes.c_size_t, (name, decl)
            elif re.match(r'^(const\s+)?long\b', decl):
                assert ct is ctypes.c_long, (name, decl)
            else:
                assert ct is ctypes.c_int, (name, decl)


def test_out_shape_entry_point_is_host_only():
    from jvae_hip import ops
    assert ops.ConvSpec(3, 32, 5, 1, 2).out_hw(32, 32) == (32, 32)
    assert ops.ConvSpec(32, 32, 5, 2, 2).out_hw(32, 32) == (16, 16)
    assert ops.ConvSpec(64, 200, 7, 1, 0).out_hw(8, 8) == (2, 2)
    assert ops.ConvSpec(64, 64, 5, 2, 2, 1, transposed=True).out_hw(8, 8) == (16, 16)
    assert ops.ConvSpec(64, 64, 8, 1, 0, 0, transposed=True).out_hw(1, 1) == (8, 8)


def test_batchnorm_launch_plan_partitions_every_batch_size():
    """The reference keeps the ragged last batch (cvae.py:2245-2249; BatchNorm2d works for any N, conv.py:214-220), so
    every batch size must partition into image ranges that are never negative, never overlap and cover [0, N) exactly -
    including the plans whose trailing parts are EMPTY ((parts-1)*ceil(N/parts) >= N, e.g. N=49 over 64 parts: the round-2
    regression).  Host-only: walks the same jvae_bn_plan / image_range() the kernels use, for every layer shape of
    configs 2 and 5 (encoder batch n, decoder batch 2n) and the label-free evaluation's 129*512 decoder batch."""
    import ctypes
    from jvae_hip import lib
    L = lib.load()
    ns, nc, nb, ne = (ctypes.c_int() for _ in range(4))
    layers2 = [(32, 1024), (32, 256), (64, 256), (64, 64), (200, 4), (64, 64), (64, 256), (32, 256), (32, 1024), (3, 1024)]
    layers5 = [(32, 4096), (32, 1024), (64, 1024), (64, 256), (128, 256), (128, 64), (200, 36), (3, 4096)]
    empties = 0

    def check(N, C, P, b8=False):
        nonlocal empties
        rc = L.jvae_bn_plan_b8(N, C, P, ns, nc) if b8 else L.jvae_bn_plan(N, C, P, ns, nc)
        assert rc == 0
        for parts in (ns.value, nc.value):
            assert 1 <= parts <= max(N, 1)
            pos = 0
            for j in range(parts):
                assert L.jvae_image_range(N, parts, j, nb, ne) == 0
                assert 0 <= nb.value <= ne.value <= N, (N, C, P, parts, j, nb.value, ne.value)
                if nb.value == ne.value:
                    empties += 1
                else:
                    assert nb.value == pos, (N, parts, j)
                    pos = ne.value
            assert pos == N, (N, C, P, parts)

    for n in range(1, 513):
        for C, P in layers2:
            check(n, C, P)
            check(2 * n, C, P)
    for n in list(range(1, 257, 5)) + [255, 256]:
        for C, P in layers5:
            check(n, C, P)
            check(2 * n, C, P)
            check(2 * n, C, P, b8=True)
    check(129 * 512, 3, 1024)
    check(129 * 512, 32, 1024)
    for N, parts in ((49, 64), (98, 64), (66048, 1366), (1, 1), (5, 5), (0, 3)):
        pos = 0
        for j in range(parts):
            assert L.jvae_image_range(N, parts, j, nb, ne) == 0
            assert 0 <= nb.value <= ne.value <= N
            pos += ne.value - nb.value
        assert pos == N
    assert empties > 0          # the sweep really contains plans with empty trailing parts
    assert L.jvae_image_range(4, 0, 0, nb, ne) < 0 and L.jvae_image_range(4, 2, 2, nb, ne) < 0


def test_batch_size_bounds_and_evaluation_slabs_follow_from_the_tensors():
    """Host logic only (the model is built on the CPU, nothing is launched): `max_batch_sizes` is derived from the tensors -
    powers of two, as the reference's halving search returns (cvae.py:1087-1153; it hard-wires 32, cvae.py:1145-1147) - such that
    the decoder batch times the widest activation (train) and the returned reconstruction (test) stay below 2^31 elements; the
    label-free evaluation decodes at most 2^28 activation elements per slab."""
    from cvae import ClassificationVariationalNetwork as Net
    from oracle.cases import full_config, get_case
    net = Net(**full_config(2, 8)['net'])
    assert net._widest_decoder_activation() == 32 * 32 * 32 and net._eval_slab_rows() == 8192
    mb = net.max_batch_sizes
    assert mb == {'train': 16384, 'test': 65536}            # L = 1 for both: 2 * 16384 * 32768 = 2^30, 2 * 65536 * 3072 < 2^31
    kw = dict(get_case('e2_n8_L3')['net'], test_latent_sampling=128)
    net = Net(**kw)
    assert net.max_batch_sizes['test'] == 4096               # 129 * 4096 * 3072 = 1.6e9 < 2^31 <= 129 * 8192 * 3072
    c100 = Net(**dict(full_config(3, 8)['net'], test_latent_sampling=128))
    assert c100.max_batch_sizes['test'] == 2048              # (L, C, N, K) = 128 * 100 * 2048 * 64 = 1.7e9 < 2^31
    os.environ['JVAE_EVAL_SLAB_ROWS'] = '7'
    try:
        assert net._eval_slab_rows() == 7
    finally:
        del os.environ['JVAE_EVAL_SLAB_ROWS']
    mlp = Net(**get_case('c1_n16_mlp')['net'])
    assert mlp._widest_decoder_activation() >= 784 and mlp.max_batch_sizes['train'] == 1 << 16


def test_save_signature_serves_the_fine_tuning_subclasses(tmp_path):
    """cvae.py:2650-2675 `save(dir_name=None, except_optimizer=False, except_state=False)`; the fine-tuning job classes
    override it the way ft/job.py:154-158 does - `super().save(*a, except_state=except_state, except_optimizer=True)` plus a
    json file of their own - so the drop-in must accept both keywords with the reference's meaning: tensors only once
    `trained`, state.pth unless except_state, optimizer.pth unless except_optimizer.  Host logic only (CPU tensors)."""
    import json
    from cvae import ClassificationVariationalNetwork as Net

    class Job(Net):
        ft_param_file = 'ft.json'
        ft_params = {'alpha': 0.5}

        def save(self, *a, except_state=True, **kw):
            kw['except_optimizer'] = True
            dir_name = super().save(*a, except_state=except_state, **kw)
            with open(os.path.join(dir_name, self.ft_param_file), 'w') as f:
                json.dump(self.ft_params, f)
            return dir_name
    job = Job(**get_case('c2_n8')['net'])
    jsons = {'params.json', 'train_params.json', 'test.json', 'ood.json', 'history.json', 'ft.json'}
    job.trained = 3
    a = job.save(str(tmp_path / 'a'))
    assert a == str(tmp_path / 'a') and set(os.listdir(a)) == jsons
    b = job.save(str(tmp_path / 'b'), except_state=False)
    assert set(os.listdir(b)) == jsons | {'state.pth'}
    assert list(torch.load(os.path.join(b, 'state.pth')).keys()) == list(job.state_dict().keys())
    plain = Net(**get_case('c2_n8')['net'])
    c = plain.save(str(tmp_path / 'c'))                     # untrained: json files only (cvae.py:2667 `if self.trained`)
    assert set(os.listdir(c)) == jsons - {'ft.json'}
    plain.trained = 1
    d = plain.save(str(tmp_path / 'd'))
    assert set(os.listdir(d)) == (jsons - {'ft.json'}) | {'state.pth', 'optimizer.pth'}
    assert plain.save() == d                                 # dir_name=None: the directory of the last save / load
    again = Net.load(d)
    assert again.trained == plain.train_history['epochs'] == 0   # `trained` follows history.json, as in the reference


def test_layer_dsl_shapes():
    from module.vae_layers.conv import build_de_conv_layers, find_input_shape, parse_conv_layer_name
    f = build_de_conv_layers((3, 32, 32), 'conv32', batch_norm=True)
    assert f.name == 'conv32' and tuple(f.output_shape) == (200, 2, 2)
    assert f.shapes == [(3, 32, 32), (32, 32, 32), (32, 16, 16), (64, 16, 16), (64, 8, 8), (200, 2, 2)]
    assert find_input_shape('deconv32', (32, 32)) == (1, 1)
    assert find_input_shape('deconv32+', (64, 64)) == (5, 5)
    g = build_de_conv_layers((64, 1, 1), 'deconv32', batch_norm=True, where='output', output_activation='linear')
    assert tuple(g.output_shape) == (3, 32, 32)
    assert [type(m).__name__ for m in list(g)[-3:]] == ['HipConv2d', 'HipBatchNorm2d', 'HipIdentity']
    p = parse_conv_layer_name('64:2++1', where='output', kernel_size=5, padding=2)
    assert p == dict(ltype='deconv', kernel_size=5, padding=2, stride=2, out_channels=64, output_padding=1)
    p = parse_conv_layer_name('!3x5+2', where='output')
    assert p == dict(ltype='conv', kernel_size=5, padding=2, stride=1, out_channels=3)
    anon = build_de_conv_layers((3, 32, 32), '[x5+2]16-16:2')
    assert anon.name == '16x5-16x5:2'
    # pooling / up-sampling tokens (vgg*, ivgg*: conv-models.ini:13-18,28-30): shapes and names as in the reference
    v = build_de_conv_layers((3, 32, 32), 'vgg11')
    assert v.name == 'vgg11' and tuple(v.output_shape) == (512, 1, 1)
    assert [type(m).__name__ for m in list(v)[:3]] == ['HipConv2d', 'HipReLU', 'HipPool2d'] or \
        [type(m).__name__ for m in list(v)[:3]][0::2] == ['HipConv2d', 'HipPool2d']
    assert v.shapes[:3] == [(3, 32, 32), (64, 32, 32), (64, 16, 16)] and v.shapes[-1] == (512, 1, 1)
    u = build_de_conv_layers((8, 2, 2), '[!x3+1-U:2]U-!4-U-!3', where='output')
    assert tuple(u.output_shape) == (3, 8, 8) and u.name == 'u:2-4x3-u:2-3x3'
    anon = build_de_conv_layers((3, 8, 8), '[x3-Mx2]4-M-Ax2')
    assert anon.name == '4x3-Mx2-Ax2' and tuple(anon.output_shape) == (4, 2, 2)
    with pytest.raises(NotImplementedError):
        build_de_conv_layers((3, 32, 32), 'resnet18')


@pytest.mark.parametrize('name', list(CASES) + list(DSL_CASES))
def test_state_dict_contract(name, golden_dir):
    """Same keys, order and shapes as the reference model (checkpoint round-trip)."""
    from cvae import ClassificationVariationalNetwork as Net
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    net = Net(**get_case(name)['net'])
    sd = net.state_dict()
    assert list(sd.keys()) == list(g['state_keys'])
    assert [','.join(str(s) for s in v.shape) for v in sd.values()] == list(g['state_shapes'])
    assert net.nparams == int(g['nparams'])
    trainable = {n for n, p in net.named_parameters() if p.requires_grad}
    assert set(g['grad_names']) <= trainable


def test_no_cpu_fallback():
    from cvae import ClassificationVariationalNetwork as Net
    from jvae_hip import JvaeHipError
    case = get_case('c1_n16_mlp')
    net = Net(**case['net'])
    net.train()
    x = torch.rand(4, 1, 28, 28)
    y = torch.randint(0, 10, (4,))
    with pytest.raises(JvaeHipError):
        net.evaluate(x, y)


def test_sigma_and_optimizer_host_api():
    from module.vae_layers import Sigma
    from module.optimizers import Optimizer
    s = Sigma(value=1.0, learned=True)
    assert s.requires_grad and s.is_log and abs(s.value - 1.0) < 1e-7 and f'{s:i}' == 'l'
    s2 = Sigma(value=0.5)
    assert not s2.requires_grad and str(s2) == '0.5' and abs(s2.value - 0.5) < 1e-7
    s3 = Sigma(value=2.0, decay=0.1, reach=1.5)
    assert str(s3) == '2->1.5*rmse[-0.1*]'
    p = torch.nn.Parameter(torch.zeros(3))
    o = Optimizer([p], optim_type='adam', lr=1e-3, weight_decay=3e-5, grad_clipping=100, lr_decay=0.01)
    assert o.lr == 1e-3 and o.params['grad_clipping'] == 100
    assert f'{o:10}' == 'adam--lr=0.001--decay=0.01--betas=(0.9, 0.999)--weight_decay=3e-05'
    o.update_lr()
    assert abs(o.lr - 1e-3 * 0.99) < 1e-12
    sgd = Optimizer([p], optim_type='sgd', momentum=0.9, weight_decay=1e-4)
    assert sgd.lr == 0.01 and f'{sgd:10}' == 'sgd--lr=0.01--momentum=0.9--weight_decay=0.0001'
    with pytest.raises(ValueError):
        Optimizer([p], optim_type='rmsprop')


def test_sigma_matches_reference_forms(golden_dir):
    """Sigma: printed forms, `params` keys / values, storage and the update rules against what the
    REFERENCE's class produced for the same constructor arguments (tests/golden/sigma_forms.json, written by
    oracle/gen_sigma_fixture.py) - fixed, decayed (reach / max_step), learned, rmse, coded scalar / mask, per-dimension."""
    import copy
    import json
    import warnings
    from module.vae_layers.layers import Sigma
    fx = json.load(open(os.path.join(golden_dir, 'sigma_forms.json')))

    def check(s, want):
        assert str(s) == want['str'] and repr(s) == want['repr'], (str(s), want['str'], repr(s), want['repr'])
        for f, txt in want['formats'].items():
            assert format(s, f) == txt, (f, format(s, f), txt)
        p = s.params
        assert list(p) == want['param_keys']
        for k, v in want['params'].items():
            mine = list(p[k]) if isinstance(p[k], (tuple, list)) else p[k]
            assert mine == pytest.approx(v) if isinstance(v, float) else mine == v, (k, mine, v)
        assert list(s.shape) == want['shape'] and s.requires_grad == want['requires_grad']
        assert s.data.flatten()[:4].tolist() == pytest.approx(want['data'], rel=1e-6, abs=1e-7)

    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for rec in fx['cases']:
            kw = {a: (tuple(b) if isinstance(b, list) else b) for a, b in rec['kwargs'].items()}
            s = Sigma(**kw)
            check(s, rec['initial'])
            if s.coded:
                n = int(np.prod(s.output_dim))
                s.update(v=torch.linspace(-1, 1, 5 * n).reshape(5, *s.output_dim))
                check(s, rec['after'][0])
            else:
                for r, want in zip(fx['rmses'], rec['after']):
                    s.update(rmse=torch.tensor(r))
                    check(s, want)
            twin = copy.deepcopy(s)
            assert str(twin) == str(s) and torch.equal(twin.data, s.data)


def test_loss_recorder_reads_and_writes_the_reference_format(tmp_path, golden_dir):
    """`record-<set>.pth` (utils/save_load/recorders.py:107-174).  tests/golden/record_ref was written by
    the REFERENCE's LossRecorder (oracle/gen_record_fixture.py).  The drop-in recorder loads both files (cut / uncut),
    returns what the reference's accessors returned, re-records the same batches into a file with the same dictionary
    (keys, scalars, tensors bit for bit), and merges like the reference."""
    from jvae_compat.recorders import LossRecorder
    src = os.path.join(golden_dir, 'record_ref')
    exp = np.load(os.path.join(src, 'expected.npz'))
    for name in ('record-demo.pth', 'record-demo-uncut.pth'):
        r = LossRecorder.load(os.path.join(src, name))
        assert len(r) == int(exp['len']) and r.recorded_samples == int(exp['recorded_samples']) and r.batch_size == 4
        assert list(r.keys()) == ['total', 'kl', 'cross_x', 'logits', 'y_true']
        for k in r:
            assert np.array_equal(r[k].numpy(), exp['all.' + k]), (name, k)
            assert np.array_equal(r.get_batch(3, k).numpy(), exp['b3.' + k]) and np.array_equal(r.get_batch(1, k).numpy(), exp['b1.' + k])
        assert r.has_batch(3) and not r.has_batch(3, only_full=True) and not r.has_batch(4)
    # re-record the same batches (as oracle/gen_record_fixture.py made them) and compare the files' dictionaries
    C, B, sizes = 3, 4, [4, 4, 4, 2]

    def batch(i, n):
        g = torch.Generator().manual_seed(50 + i)
        return dict(total=torch.randn(C, n, generator=g), kl=torch.randn(C, n, generator=g), cross_x=torch.randn(n, generator=g),
                    logits=torch.randn(C, n, generator=g), y_true=torch.randint(0, C, (n,), generator=g))
    mine = LossRecorder(B, **batch(0, B))
    for i, n in enumerate(sizes):
        mine.append_batch(**batch(i, n))
    assert mine.num_batch == int(exp['num_batch_before_save'])
    for cut, name in ((False, 'record-demo-uncut.pth'), (True, 'record-demo.pth')):
        out = os.path.join(tmp_path, name)
        mine.save(out, cut=cut)
        a = torch.load(out, weights_only=False)
        b = torch.load(os.path.join(src, name), weights_only=False)
        assert set(a) == set(b), (set(a) ^ set(b))
        for k in b:
            if k == '_tensors':
                assert list(a[k]) == list(b[k])
                for t in b[k]:
                    assert a[k][t].dtype == b[k][t].dtype and torch.equal(a[k][t], b[k][t]), (name, t)
            elif k != '_seed':                     # the seed is random by design
                assert a[k] == b[k], (name, k, a[k], b[k])
    assert mine.num_batch == int(exp['num_batch_after_cut'])
    other = LossRecorder(B, **batch(0, B))
    other.append_batch(**batch(7, 3))
    merged = LossRecorder.load(os.path.join(tmp_path, 'record-demo.pth'))
    merged.merge(other)
    assert len(merged) == int(exp['merged.len']) and merged.recorded_samples == int(exp['merged.recorded_samples'])
    assert merged.last_batch_size == int(exp['merged.last_batch_size'])
    assert np.array_equal(merged['total'].numpy(), exp['merged.total'])
    assert set(LossRecorder.loadall(str(tmp_path))) == {'demo', 'demo-uncut'}
