"""CPU: the job directories under tests/golden/job_dropin/ were WRITTEN by the drop-in's train_model() on the GPU box
(tests/test_2_model_gpu.py::test_train_model_as_train_py_drives_it / ::test_train_model_job_of_a_small_model_for_the_reference_to_load,
copied through JVAE_KEEP_JOB_DIR).  oracle/check_dropin_job.py hands them to the REFERENCE's own load() in the build container
(profiles/r05_dropin_job_reference_load.txt); here: they hold what train.py:224-229 reads on --resume and what cvae.py:2108-2167
records."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
JOBS = os.path.join(HERE, 'golden', 'job_dropin')


@pytest.mark.parametrize('job,expect', [('cifar_conv32', dict(set='cifar10', transformer='default', validation=88,
                                                              data_augmentation=['flip', 'crop'], epochs=2, batch_size=64)),
                                        ('toy8_mlp', dict(set='toy8', transformer='simple', validation=16, data_augmentation=[],
                                                          epochs=2, batch_size=16))])
def test_job_written_by_train_model_has_the_reference_bookkeeping(job, expect):
    d = os.path.join(JOBS, job)
    tp = json.load(open(os.path.join(d, 'train_params.json')))
    for k, v in expect.items():
        assert tp[k] == v, k
    for k in ('latent_sampling', 'full_test_every', 'validation_split_seed', 'warmup', 'warmup_gamma', 'sigma', 'optimizer'):
        assert k in tp, k
    assert 0 <= tp['validation_split_seed'] < 2 ** 12
    hist = json.load(open(os.path.join(d, 'history.json')))
    assert hist['epochs'] == 2 and set(hist) == {'epochs', '0', '1', '2'}        # cvae.py:2293-2296: epoch == epochs has an entry too
    assert 'train_loss' in hist['1'] and 'validation_accuracy' in hist['0'] and 'train_loss' not in hist['2']
    assert {'test_accuracy', 'test_loss', 'validation_loss'} <= set(hist['2'])
    arch = json.load(open(os.path.join(d, 'params.json')))
    assert arch['type'] == 'cvae' and 'prior' in arch and 'latent_dim' in arch
    for f in ('test.json', 'ood.json'):
        assert os.path.exists(os.path.join(d, f))


def test_reference_load_report_is_committed():
    rep = open(os.path.join(os.path.dirname(HERE), 'profiles', 'r05_dropin_job_reference_load.txt')).read()
    assert rep.count('OK  ') == 2 and 'load_state=True' in rep and 'one reference step' in rep
