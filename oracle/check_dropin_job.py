#!/usr/bin/env python3
"""TEST INFRASTRUCTURE - build container only (needs /root/reference).

Hands a job directory written by the DROP-IN's train_model() (tests/test_2_model_gpu.py::test_train_model_as_train_py_drives_it and
::test_train_model_job_of_a_small_model_for_the_reference_to_load; copied off the GPU box through JVAE_KEEP_JOB_DIR) to the
REFERENCE's own ClassificationVariationalNetwork.load() (cvae.py:2677-2857) and performs the reads of train.py:224-229 on the
result - what `train.py --resume` does with a job of ours.  A directory without state.pth (the CIFAR-recipe job: its tensors are
18 MB and are not committed) is loaded with load_state=False, as the reference's own tools do for listings.

usage: python oracle/check_dropin_job.py DIR [DIR ...]      (default: tests/golden/job_dropin/*)"""
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from oracle.gen_golden import import_reference      # noqa: E402


def cpu_copy(d):
    """The job was written on the GPU box: its state.pth / optimizer.pth hold CUDA tensors, and the reference's load() reads
    optimizer.pth without a map_location (cvae.py:2845) - in this GPU-less container the two files are re-saved with their tensors
    on the CPU, in a temporary copy of the directory; nothing else is touched."""
    import shutil
    import tempfile
    import torch
    t = tempfile.mkdtemp(prefix='dropin_job_')
    dst = os.path.join(t, os.path.basename(d))
    shutil.copytree(d, dst)
    for f in ('state.pth', 'optimizer.pth'):
        fp = os.path.join(dst, f)
        if os.path.exists(fp):
            torch.save(torch.load(fp, map_location='cpu'), fp)
    return dst


def check(Net, d):
    d = os.path.abspath(d)
    shown = d
    has_state = os.path.exists(os.path.join(d, 'state.pth'))
    if has_state:
        import torch
        if not torch.cuda.is_available():
            d = cpu_copy(d)
    net = Net.load(d, load_state=has_state)
    tp = net.training_parameters
    # train.py:224-229
    dataset, transformer = tp['set'], tp['transformer']
    validation = tp['validation']
    data_augmentation = tp['data_augmentation']
    latent_sampling = tp['latent_sampling']
    ours = json.load(open(os.path.join(d, 'train_params.json')))
    assert dataset == ours['set'] and transformer == ours['transformer'] and validation == ours['validation']
    assert data_augmentation == ours['data_augmentation'] and latent_sampling == ours['latent_sampling']
    assert tp['full_test_every'] == ours['full_test_every'] and tp['batch_size'] == ours['batch_size']
    hist = json.load(open(os.path.join(d, 'history.json')))
    assert net.trained == hist['epochs'] == net.train_history['epochs']
    assert set(net.train_history) == {'epochs'} | set(range(hist['epochs'] + 1))        # int keys, epoch == epochs entry included
    last = net.train_history[hist['epochs'] - 1]
    assert 'train_loss' in last and 'lr' in last
    n_state = None
    if has_state:                       # every tensor of our state.pth went into the reference's modules (strict load)
        import torch
        sd = torch.load(os.path.join(d, 'state.pth'), map_location='cpu')
        mine = net.state_dict()
        assert set(sd) == set(mine), set(sd) ^ set(mine)
        for k, v in sd.items():
            assert torch.equal(v.cpu(), mine[k].cpu()), k
        n_state = len(sd)
        # and the reference can take a training step from there (optimizer.pth restored: Adam moments of every parameter)
        net.train()
        x = torch.rand(4, *net.input_shape)
        y = torch.randint(0, net.num_labels, (4,))
        net.optimizer.zero_grad()
        out = net.evaluate(x, y, with_beta=True)
        out[2]['total'].mean().backward()
        net.optimizer.clip(net.parameters())
        net.optimizer.step()
    print('OK  {}: reference load(load_state={}) -> set={!r} transformer={!r} validation={} data_augmentation={} latent_sampling={} '
          'trained={} history keys={} lr={:.3e}{}'.format(os.path.relpath(shown, REPO), has_state, dataset, transformer, validation,
                                                         data_augmentation, latent_sampling, net.trained, sorted(map(str, net.train_history)),
                                                         net.optimizer.lr, '' if n_state is None else ' state tensors={} + one reference step'.format(n_state)))


def main():
    dirs = sys.argv[1:] or sorted(glob.glob(os.path.join(REPO, 'tests', 'golden', 'job_dropin', '*')))
    dirs = [os.path.abspath(d) for d in dirs]
    Net = import_reference()            # (changes the working directory to the reference's)
    for d in dirs:
        check(Net, d)


if __name__ == '__main__':
    main()
