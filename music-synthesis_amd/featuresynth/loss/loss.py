"""GAN losses of the hot path with the call signatures of the reference's
featuresynth/loss/loss.py (hinge :9-18, least squares :5-14, mel_gan_disc_loss :21-25,
mel_gan_feature_loss :28-65, mel_gan_gen_loss :68-78).  Each returns a 0-d tensor with autograd;
the reductions are wavefront-shuffle HIP kernels and the composite losses are single autograd
nodes (one fused backward over the 18 feature maps).
"""
from .._ops import functional as F_


def least_squares_generator_loss(j):
    """0.5 * mean((j - 1)^2)"""
    return F_.LsGFn.apply(j)


def hinge_generator_loss(j):
    """mean(-j)"""
    return F_.NegMeanFn.apply(j)


def least_squares_disc_loss(r_j, f_j):
    """0.5 * (mean((r - 1)^2) + mean(f^2))"""
    return F_.LsDFn.apply(r_j, f_j)


def hinge_discriminator_loss(r_j, f_j):
    """mean(relu(1 - r) + relu(1 + f))"""
    return F_.HingeDFn.apply(r_j, f_j)


def mel_gan_disc_loss(real_judgements, fake_judgements, gan_loss=hinge_discriminator_loss):
    real_judgements, fake_judgements = list(real_judgements), list(fake_judgements)
    if gan_loss is hinge_discriminator_loss and len(real_judgements) == len(fake_judgements) > 0:
        return F_.MelGanDiscLossFn.apply(len(real_judgements), *real_judgements, *fake_judgements)
    total = None
    for r, f in zip(real_judgements, fake_judgements):
        term = gan_loss(r, f)
        total = term if total is None else total + term
    return total


def _balanced(real_features, fake_features):
    if len(real_features) != len(fake_features) or not real_features:
        return False
    n = len(real_features[0])
    return n > 0 and all(len(g) == n for g in real_features) and all(len(g) == n for g in fake_features)


def mel_gan_feature_loss(real_features, fake_features):
    """sum over discriminators d and layers l of (1/D)(1/L_d) * l1(real, fake); the reference
    settles on this scaling after the discussion in its comment block (loss.py:41-60)."""
    nd = 1.0 / len(real_features)
    total = None
    for r_group, f_group in zip(real_features, fake_features):
        nl = 1.0 / len(r_group)
        for r_f, f_f in zip(r_group, f_group):
            term = (nl * nd) * F_.L1MeanFn.apply(r_f, f_f)
            total = term if total is None else total + term
    return total


def mel_gan_gen_loss(real_features, fake_features, real_judgements, fake_judgements,
                     gan_loss=hinge_generator_loss, feature_loss_weight=10):
    real_features = [list(g) for g in real_features]
    fake_features = [list(g) for g in fake_features]
    fake_judgements = list(fake_judgements)
    if gan_loss is hinge_generator_loss and _balanced(real_features, fake_features) and \
            len(fake_judgements) == len(fake_features):
        S, Lyr = len(fake_features), len(fake_features[0])
        flat_r = [t for g in real_features for t in g]
        flat_f = [t for g in fake_features for t in g]
        return F_.MelGanGenLossFn.apply(S, Lyr, float(feature_loss_weight), *flat_r, *flat_f,
                                        *fake_judgements)
    j_loss = None
    for _, f in zip(real_judgements, fake_judgements):
        term = gan_loss(f)
        j_loss = term if j_loss is None else j_loss + term
    return j_loss + feature_loss_weight * mel_gan_feature_loss(real_features, fake_features)
