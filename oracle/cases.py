"""TEST INFRASTRUCTURE — the parity cases (model constructor kwargs + batch geometry).

Shared by oracle/gen_golden.py (runs the *reference* in the build container and writes tests/golden/*.npz),
by the CPU-oracle tests and by the GPU parity tests.  Config numbering follows BASELINE.json `configs`
/ SURVEY.md §8d.
"""
import copy

_ADAM = dict(optim_type='adam', lr=1e-3, weight_decay=3e-5, grad_clipping=100)


def _conv(C, K=64, **kw):
    d = dict(input_shape=(3, 32, 32), num_labels=C, type='cvae',
             features='conv32', upsampler='deconv32', encoder=[], decoder=[], classifier=[],
             batch_norm='both', latent_dim=K, latent_sampling=1, test_latent_sampling=1,
             sigma={'value': 1.0, 'learned': True}, gamma=0, beta=1., output_activation='linear',
             prior=dict(distribution='gaussian', init_mean=0., learned_means=True, var_dim='scalar',
                        freeze_means=0),
             optimizer=dict(_ADAM))
    d.update(kw)
    return d


CASES = {
    # config 2 (CIFAR-10 conv CVAE) at a batch the oracle runs in < 1 s
    'c2_n8': dict(net=_conv(10), N=8, kl_var_weighting=1.0, gamma_weighting=1.0),
    # same, warm-up weight != 1 and L=2 latent samples, fixed sigma
    'c2_n6_L2_warm': dict(net=_conv(10, latent_sampling=2, test_latent_sampling=2, sigma={'value': 0.7}),
                          N=6, kl_var_weighting=0.25, gamma_weighting=1.0),
    # config 3 (CIFAR-100): scalar / diag / full prior variance
    'c3_n8_scalar': dict(net=_conv(100), N=8, kl_var_weighting=1.0, gamma_weighting=1.0),
    'c3_n8_diag': dict(net=_conv(100, prior=dict(distribution='gaussian', init_mean=0., learned_means=True,
                                                   var_dim='diag', freeze_means=0)),
                       N=8, kl_var_weighting=1.0, gamma_weighting=1.0),
    'c3_n8_full': dict(net=_conv(100, prior=dict(distribution='gaussian', init_mean=0., learned_means=True,
                                                   var_dim='full', freeze_means=0)),
                       N=8, kl_var_weighting=1.0, gamma_weighting=1.0),
    # gamma > 0: classifier on z enters the loss
    'c2_n8_gamma': dict(net=_conv(10, gamma=2.0, classifier=[20]), N=8, kl_var_weighting=1.0, gamma_weighting=0.5),
    # alternative priors
    'c2_n8_tilted': dict(net=_conv(10, prior=dict(distribution='tilted', init_mean=0., learned_means=True,
                                                   tau=5., freeze_means=0)),
                         N=8, kl_var_weighting=1.0, gamma_weighting=1.0),
    # activation='leaky' (nn.LeakyReLU(), slope 0.01) with the tilted prior: the shipped conv32 / deconv32 recipe of config.ini:96-115
    'c2_n8_leaky': dict(net=_conv(10, activation='leaky', prior=dict(distribution='tilted', init_mean=0., learned_means=True,
                                                                      tau=5., freeze_means=0)),
                        N=8, kl_var_weighting=1.0, gamma_weighting=1.0),
    'c2_n8_uniform': dict(net=_conv(10, prior=dict(distribution='uniform', init_mean=0., learned_means=True,
                                                    tau=3., freeze_means=0)),
                          N=8, kl_var_weighting=0.5, gamma_weighting=1.0),
    # encoder-only batch norm, no learned means
    'c2_n8_bnenc': dict(net=_conv(10, batch_norm='encoder',
                                  prior=dict(distribution='gaussian', init_mean=0., learned_means=False,
                                             var_dim='scalar', freeze_means=0)),
                        N=8, kl_var_weighting=1.0, gamma_weighting=1.0),
    # sigma kinds beyond the scalar value (cvae.py:626-670): sigma_n = the sample's own rmse, and log sigma_n
    # coded by one more head of the encoder; a decayed sigma with a step cap
    'c2_n8_rmse': dict(net=_conv(10, sigma={'is_rmse': True}), N=8, kl_var_weighting=1.0, gamma_weighting=1.0),
    'c2_n8_coded': dict(net=_conv(10, sigma={'input_dim': (3, 32, 32)}), N=8, kl_var_weighting=1.0, gamma_weighting=1.0),
    'c2_n8_decay': dict(net=_conv(10, sigma={'value': 1.0, 'decay': 0.1, 'reach': 2, 'max_step': 0.05}), N=8,
                        kl_var_weighting=1.0, gamma_weighting=1.0),
    # config 1: MNIST-shape MLP (784-512-256 -> 16 -> 256-512-784), gamma=1000, sigmoid output, fixed sigma
    'c1_n16_mlp': dict(net=dict(input_shape=(1, 28, 28), num_labels=10, type='cvae',
                                features=None, upsampler=None, encoder=[512, 256], decoder=[256, 512],
                                classifier=[], batch_norm=False, latent_dim=16, latent_sampling=1,
                                test_latent_sampling=1, sigma={'value': 0.1}, gamma=1000., beta=1.,
                                output_activation='sigmoid',
                                prior=dict(distribution='gaussian', init_mean=0., learned_means=True,
                                           var_dim='scalar', freeze_means=0),
                                optimizer=dict(_ADAM)),
                       N=16, kl_var_weighting=1.0, gamma_weighting=1.0),
    # config 5 geometry (ours): 3x64x64, conv32+/deconv32+, K=200, C=20 (fp32 here)
    'c5_n4': dict(net=dict(_conv(20, K=200), input_shape=(3, 64, 64), features='conv32+', upsampler='deconv32+'),
                  N=4, kl_var_weighting=1.0, gamma_weighting=1.0),
}


# Evaluation-mode cases: evaluate(x) WITHOUT labels in eval mode -> all-class losses (C, N), the
# importance-weighted bound `iws`, predictions and OOD scores.  BatchNorm runs on its running statistics.
EVAL_CASES = {
    'e2_n8_L3': dict(net=_conv(10, test_latent_sampling=3), N=8),
    'e3_n6_diag_L2': dict(net=_conv(100, test_latent_sampling=2,
                                    prior=dict(distribution='gaussian', init_mean=0., learned_means=True,
                                               var_dim='diag', freeze_means=0)), N=6),
    'e2_n8_gamma_L2': dict(net=_conv(10, gamma=2.0, classifier=[20], test_latent_sampling=2), N=8),
    # the sampling the evaluation path exists for (L >> 1; x_reco kept as per-image checksums)
    'e2_n16_L16': dict(net=_conv(10, test_latent_sampling=16), N=16),
    'e3_n8_L32': dict(net=_conv(100, test_latent_sampling=32), N=8),
    'e2_n8_rmse_L2': dict(net=_conv(10, sigma={'is_rmse': True}, test_latent_sampling=2), N=8),
    'e2_n8_coded_L2': dict(net=_conv(10, sigma={'input_dim': (3, 32, 32)}, test_latent_sampling=2), N=8),
}
EVAL_OOD_METHODS = ['iws', 'mse', 'elbo', 'soft', 'zdist', 'iws-2s', 'elbo-a-4-1']


# WIM fine-tuning step (ft/wim.py:215-255, ft/job.py:380-399): one batch evaluated under the original
# class-conditional prior, a second ("mixture") batch under an alternate, non-conditional prior N(mean_shift, I), one
# backward on  total_in.mean() + alpha * total_mix.mean(),  then optimizer.step() BEFORE optimizer.clip() (sic).
WIM_CASES = {
    'w2_n8': dict(net=_conv(10), N=8, alpha=0.1,
                  alternate_prior=dict(distribution='gaussian', init_mean=0., mean_shift=1.5, var_dim='scalar')),
}


# Layer-DSL tokens outside conv32 / deconv32 (SURVEY.md §8b "accept" list): max / average pooling (`M`, `A`: vgg*-style
# features), nearest up-sampling with plain convolutions (`U`, `!C`: ivgg*-style upsampler) and optim_type='sgd'.
# Pinned by the reference directly (tests/test_model_gpu.py); the CPU oracle restates only the conv / deconv tokens.
DSL_CASES = {
    'v2_n6_vgg_sgd': dict(net=_conv(10, K=16, input_shape=(3, 16, 16), features='[x3-Mx2]8-M-16-Ax2-32-M-Ax1',
                                    upsampler='[!x3+1-U:2]U-!16-U-!8-U-!3',
                                    optimizer=dict(optim_type='sgd', lr=0.05, weight_decay=1e-4, grad_clipping=100,
                                                   momentum=0.9)),
                          N=6, kl_var_weighting=1.0, gamma_weighting=1.0),
}


# type='vae' (single, non-conditional prior; cvae.py:188-205,274-275): one training step and one evaluation pass
DSL_CASES['a2_n8_vae'] = dict(net=_conv(10, type='vae', prior=dict(distribution='gaussian', init_mean=0., var_dim='scalar')),
                              N=8, kl_var_weighting=1.0, gamma_weighting=1.0)


# type='jvae': labels one-hot coded into the encoder input, single prior, classifier on z always in the loss
DSL_CASES['j2_n8_jvae'] = dict(net=_conv(10, type='jvae', y_is_coded=True, gamma=2.0, classifier=[20],
                                         prior=dict(distribution='gaussian', init_mean=0., var_dim='scalar')),
                               N=8, kl_var_weighting=1.0, gamma_weighting=0.5)

# type='xvae': class-conditional prior as cvae, classifier on z always in the loss (cvae.py:196-199,557-563)
DSL_CASES['x2_n8_xvae'] = dict(net=_conv(10, type='xvae', gamma=2.0, classifier=[20]), N=8, kl_var_weighting=1.0,
                               gamma_weighting=0.5)

# type='vib' (no decoder: the classifier on z and the KL to a single prior are the whole loss; cvae.py:189,201,222-224,
# 490-505,891-896): one training step
DSL_CASES['b2_n8_vib'] = dict(net=_conv(10, type='vib', gamma=2.0, classifier=[20], upsampler=None,
                                        prior=dict(distribution='gaussian', init_mean=0., var_dim='scalar')),
                              N=8, kl_var_weighting=1.0, gamma_weighting=0.5)

# output_distribution='categorical': the decoder ends in 256 x C channels of level logits (conv.py:180-185,228-230), -log p(x|z)
# is the per-pixel 256-way cross entropy (cvae.py:654-660,752-753)
DSL_CASES['g2_n4_categorical'] = dict(net=_conv(10, output_distribution='categorical', sigma={'value': 1.0}), N=4,
                                      kl_var_weighting=1.0, gamma_weighting=1.0)

DSL_EVAL_CASES = {
    # NOT here: evaluate(x) without labels for models with CODED labels (jvae, y_is_coded).  The reference cannot run it:
    # cvae.py:593-600 builds the (C, N) label grid, cvae.py:451 then does y.view(N) on it -> "RuntimeError: shape '[N]' is
    # invalid for input of size C*N" (probed for conv and MLP models in the build container); the drop-in raises as well.
    'ex2_n8_xvae_L2': dict(net=_conv(10, type='xvae', gamma=2.0, classifier=[20], test_latent_sampling=2), N=8),
    'eb2_n8_vib_L2': dict(net=_conv(10, type='vib', gamma=2.0, classifier=[20], upsampler=None, test_latent_sampling=2,
                                    prior=dict(distribution='gaussian', init_mean=0., var_dim='scalar')), N=8),
    'eg2_n4_categorical_L2': dict(net=_conv(10, output_distribution='categorical', sigma={'value': 1.0}, test_latent_sampling=2),
                                  N=4),
    'ea2_n8_vae_L3': dict(net=_conv(10, type='vae', test_latent_sampling=3,
                                    prior=dict(distribution='gaussian', init_mean=0., var_dim='scalar')), N=8),
}


# Full-size workloads of BASELINE.json (configs[1], configs[2] and the per-rank batch of configs[4]): the reference runs
# them in seconds on the CPU; the goldens are COMPACT (per-sample losses, measures, per-tensor gradient norms, small
# gradients, parameter norms after the step - a few hundred KB) and pin the headline workload to the reference itself.
FULL_CASES = {
    'c2_n512': dict(net=_conv(10), N=512, kl_var_weighting=1.0, gamma_weighting=1.0),
    'c3_n512': dict(net=_conv(100), N=512, kl_var_weighting=1.0, gamma_weighting=1.0),
    'c3_n512_diag': dict(net=_conv(100, prior=dict(distribution='gaussian', init_mean=0., learned_means=True,
                                                     var_dim='diag', freeze_means=0)),
                         N=512, kl_var_weighting=1.0, gamma_weighting=1.0),
    'c5_n256': dict(net=dict(_conv(20, K=200), input_shape=(3, 64, 64), features='conv32+', upsampler='deconv32+'),
                    N=256, kl_var_weighting=1.0, gamma_weighting=1.0),
}


def get_case(name):
    if name in FULL_CASES:
        return copy.deepcopy(FULL_CASES[name])
    if name in DSL_EVAL_CASES:
        return copy.deepcopy(DSL_EVAL_CASES[name])
    if name in DSL_CASES:
        return copy.deepcopy(DSL_CASES[name])
    if name in WIM_CASES:
        return copy.deepcopy(WIM_CASES[name])
    return copy.deepcopy(CASES[name] if name in CASES else EVAL_CASES[name])


def full_config(which, N=512):
    """Full-size configs of BASELINE.json (no goldens: checked through size-independent properties)."""
    if which == 2:
        return dict(net=_conv(10), N=N, kl_var_weighting=1.0, gamma_weighting=1.0)
    if which == 3:
        return dict(net=_conv(100), N=N, kl_var_weighting=1.0, gamma_weighting=1.0)
    raise KeyError(which)
